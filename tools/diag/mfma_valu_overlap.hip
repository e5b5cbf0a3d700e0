// Does a wave's own VALU work run under its own MFMAs on gfx950, or only another wave's?  The question behind the fused ViT MLP
// (ymk_vit_mlp.hip: one wave per SIMD, 72 MFMAs + ~450 VALU instructions per chunk at 1.9 x the longer of the two) and behind the
// fp32 -> (h, l) conversions inside the convolution loops.  A standalone program (no torch): every instruction of the timed
// loop is volatile inline assembly, so the sequence in the binary is the sequence written here
// (check: /opt/rocm/lib/llvm/bin/llvm-objdump -d on the code object).
//
//   body of one iteration = 4 x { one v_mfma_f32_32x32x16_f16 (own accumulator), NV VALU instructions of one kind }
//   kinds: v_fma_f32, v_pk_fma_f32, v_exp_f32, v_cvt_f16_f32 (first set); with any argument the second set: v_fma_mixlo_f16
//   (f32 and f16 third source), v_cvt_pk_f16_f32, v_pk_mul_f32, v_pk_add_f32, ds_read_b128, v_cvt_f32_f16, v_rcp_f32,
//   v_pk_fma_f16, v_mul_f32, and MFMA chains on 4 / 2 / 1 accumulators
//   MF = 0: the VALU instructions alone (their own cost); NV = 0: the MFMAs alone
//   waves per SIMD: 1 / 2 / 4 (block of 256 / 512 / 1024 threads, 128 KB of LDS so that a CU holds one block)
//
// Results and what they meant for the kernels: profiles/r05_conv_dma_ablation.md section 2.
// Output: one JSON line per case with core cycles per MFMA group (s_memtime around the loop, mean over waves), the core clock
// (s_memtime against the 100 MHz s_memrealtime), and the wall time of the launch.
//
// build: hipcc --offload-arch=gfx950 -O2 -o scratch/mfma_valu_overlap tools/diag/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

template <int KIND>
__device__ __forceinline__ void valu(float& x, float& y, f2v& px, f2v& py, const float c, const char* lp) {
  if constexpr (KIND == 0) {
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(c));
  } else if constexpr (KIND == 1) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(px) : "v"(py), "v"(py));
  } else if constexpr (KIND == 2) {
    asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(y));
  } else if constexpr (KIND == 3) {
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(x) : "v"(y));
  } else if constexpr (KIND == 4) {  // f16(y * c + 0): the scale and the cut in one instruction
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(x) : "v"(y), "v"(c));
  } else if constexpr (KIND == 5) {  // f16(y * c - f16 source): the low plane in one instruction
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(x) : "v"(y), "v"(c), "v"(px.x));
  } else if constexpr (KIND == 6) {
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(c));
  } else if constexpr (KIND == 7) {
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(px) : "v"(py), "v"(py));
  } else if constexpr (KIND == 8) {
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(px) : "v"(py), "v"(py));
  } else if constexpr (KIND == 9) {  // a fragment read
    f4v r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((unsigned)(size_t)lp));
    px.x = r.x;
  } else if constexpr (KIND == 10) {
    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x) : "v"(y));
  } else if constexpr (KIND == 11) {
    asm volatile("v_rcp_f32 %0, %1" : "=v"(x) : "v"(y));
  } else if constexpr (KIND == 12) {
    asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(c));
  } else {
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(c));
  }
}

template <int NV, int KIND, int MF, int NACC = 4>
__global__ void k_overlap(float* sink, long long* cyc, long long* real, int iters) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  h8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (lane + i));
    b[i] = (_Float16)(0.002f * (lane - i));
  }
  f16v acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float x[8], y[8];
  f2v px[8], py[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    x[i] = 0.5f + i;
    y[i] = 1e-3f * (lane + i);
    px[i] = f2v{0.5f, 0.25f};
    py[i] = f2v{1e-3f * lane, 1e-3f * i};
  }
  const float c = 0.999f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();  // s_memtime
  const long long r0 = wall_clock64();                // s_memrealtime (100 MHz)
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (MF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[q % NACC]) : "v"(a), "v"(b));
#pragma unroll
      for (int n = 0; n < NV; ++n) valu<KIND>(x[(q * NV + n) & 7], y[(q * NV + n) & 7], px[(q * NV + n) & 7], py[(q * NV + n) & 7], c, lds + 16 * lane);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long r1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[q][r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i] + px[i].x + px[i].y;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (lane == 0) {
    cyc[gw] = t1 - t0;
    real[gw] = r1 - r0;
  }
  if (s == 123.456f) sink[0] = s + lds[0];
}

struct Case {
  int nv, kind, mf, wps;
};

template <int NV, int KIND, int MF, int NACC = 4>
static int run(int wps, float* sink, long long* cyc, long long* real, int iters) {
  const int threads = 256 * wps, blocks = 256;
  const size_t lds = 128 * 1024;
  auto kern = k_overlap<NV, KIND, MF, NACC>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, sink, cyc, real, iters);  // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, sink, cyc, real, iters);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = blocks * threads / 64;
  std::vector<long long> hc(nw), hr(nw);
  CHECK(hipMemcpy(hc.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hr.data(), real, nw * sizeof(long long), hipMemcpyDeviceToHost));
  double sc = 0, sr = 0;
  for (int i = 0; i < nw; ++i) {
    sc += (double)hc[i];
    sr += (double)hr[i];
  }
  sc /= nw;
  sr /= nw;
  const double groups = (double)iters * 4;
  static const char* kinds[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_cvt_f16_f32", "v_fma_mixlo_f16", "v_fma_mixlo_f16(f16 src)",
                                "v_cvt_pk_f16_f32", "v_pk_mul_f32", "v_pk_add_f32", "ds_read_b128", "v_cvt_f32_f16", "v_rcp_f32", "v_pk_fma_f16", "v_mul_f32"};
  std::printf("{\"mfma\": %d, \"accumulators\": %d, \"valu_per_mfma\": %d, \"valu\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_group\": %.2f, "
              "\"memtime_ticks_per_us\": %.1f, \"wave_us\": %.1f, \"launch_us\": %.1f, \"ns_per_group_per_simd\": %.2f}\n",
              MF, NACC, NV, kinds[KIND], wps, sc / groups, sc / (sr / 100.0), sr / 100.0, ms * 1e3, ms * 1e6 / (groups * wps));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return 0;
}

template <int KIND>
static int sweep(int wps, float* sink, long long* cyc, long long* real, int iters) {
  if (run<6, KIND, 0>(wps, sink, cyc, real, iters)) return 1;  // alone
  if (run<3, KIND, 1>(wps, sink, cyc, real, iters)) return 1;
  if (run<6, KIND, 1>(wps, sink, cyc, real, iters)) return 1;
  return 0;
}

int main(int argc, char** argv) {
  float* sink;
  long long *cyc, *real;
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&cyc, 256 * 16 * sizeof(long long)));
  CHECK(hipMalloc(&real, 256 * 16 * sizeof(long long)));
  const int iters = 4000;
  const bool second = argc > 1;  // any argument: the second set (conversion / packed / LDS instructions, accumulator chains)
  for (int wps : {1, 2, 4}) {
    if (!second) {
      if (run<0, 0, 1>(wps, sink, cyc, real, iters)) return 1;  // the MFMAs alone
      if (run<6, 0, 0>(wps, sink, cyc, real, iters)) return 1;  // six v_fma alone
      if (run<6, 1, 0>(wps, sink, cyc, real, iters)) return 1;  // six v_pk_fma alone
      if (run<6, 2, 0>(wps, sink, cyc, real, iters)) return 1;  // six v_exp alone
      if (run<2, 0, 1>(wps, sink, cyc, real, iters)) return 1;
      if (run<4, 0, 1>(wps, sink, cyc, real, iters)) return 1;
      if (run<6, 0, 1>(wps, sink, cyc, real, iters)) return 1;
      if (run<8, 0, 1>(wps, sink, cyc, real, iters)) return 1;
      if (run<12, 0, 1>(wps, sink, cyc, real, iters)) return 1;
      if (run<6, 1, 1>(wps, sink, cyc, real, iters)) return 1;
      if (run<6, 2, 1>(wps, sink, cyc, real, iters)) return 1;
      if (run<2, 2, 1>(wps, sink, cyc, real, iters)) return 1;
      if (run<6, 3, 1>(wps, sink, cyc, real, iters)) return 1;
    } else {
      if (run<0, 0, 1, 4>(wps, sink, cyc, real, iters)) return 1;  // four accumulators in turn
      if (run<0, 0, 1, 2>(wps, sink, cyc, real, iters)) return 1;  // two
      if (run<0, 0, 1, 1>(wps, sink, cyc, real, iters)) return 1;  // every MFMA waits for the one before it
      if (run<4, 0, 1, 1>(wps, sink, cyc, real, iters)) return 1;  // .. with four v_fma in between
      if (sweep<3>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<4>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<5>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<6>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<7>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<8>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<9>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<10>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<11>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<12>(wps, sink, cyc, real, iters)) return 1;
      if (sweep<13>(wps, sink, cyc, real, iters)) return 1;
    }
  }
  return 0;
}
