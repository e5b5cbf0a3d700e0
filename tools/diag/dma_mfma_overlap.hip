// Does an LDS-DMA stream (buffer_load ... lds) run under the MFMAs of the waves that issued it?  The ablation copies of
// conv_f16_dma / conv_f16_dma_wide (tools/jobs/r05_l.sh) say the two ADD: 256 x 256 tiles, MFMAs on made-up fragments + the DMA
// stream 722 us, the DMA stream alone 370, the MFMAs alone ~465.  This program has the loop of those kernels and nothing else:
//
//   per iteration and wave:  s_waitcnt vmcnt(0); s_barrier; ND x buffer_load_dwordx4 .. lds (1 KB each, a stream through a
//   buffer of SRC_MB); NM x v_mfma_f32_32x32x16_f16 (four accumulators)
//
// one block of 8 waves per CU (128 KB of LDS), 256 blocks.  Output: ns per iteration for (ND, NM) pairs; overlap means
// t(ND, NM) = max(t(ND, 0), t(0, NM)).
// build: hipcc --offload-arch=gfx950 -O2 -o scratch/dma_mfma_overlap tools/diag/dma_mfma_overlap.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

#define CHECK(x)                                                   \
  do {                                                             \
    hipError_t e_ = (x);                                           \
    if (e_ != hipSuccess) {                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                                    \
    }                                                              \
  } while (0)

template <int ND, int NM, int WAIT_AT_END>
__global__ __launch_bounds__(512, 1) void k_loop(const float* src, unsigned src_bytes, float* sink, int iters) {
  extern __shared__ char lds[];
  const int t = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
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
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
  const unsigned gw = blockIdx.x * 8 + wv;        // global wave
  const unsigned stride = gridDim.x * 8 * ND * 1024;  // bytes all waves move per iteration
  unsigned base = gw * ND * 1024 + lane * 16;
  int st = 0;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (!WAIT_AT_END) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    char* dst = lds + st * 65536 + wv * ND * 1024;
#pragma unroll
    for (int i = 0; i < ND; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + i * 1024), 16, (int)((base + i * 1024) % src_bytes), 0, 0, 0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < NM; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
    if (WAIT_AT_END) {  // the DMA of this iteration is waited for right after its own MFMAs (no tile in flight across the barrier)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    base += stride;
    st ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  if (s == 123.456f) sink[0] = s + lds[lane];
}

template <int ND, int NM, int WAIT_AT_END = 0>
static int run(const float* src, size_t src_bytes, float* sink, int iters, const char* what) {
  auto kern = k_loop<ND, NM, WAIT_AT_END>;
  const size_t lds = 128 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, src, (unsigned)src_bytes, sink, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, src, (unsigned)src_bytes, sink, iters);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double ns_it = ms * 1e6 / iters;
  const double tbs = 256.0 * 8 * ND * 1024 / ns_it / 1e3;  // TB/s moved
  const double mfma_ns = 2.0 * NM * 18.5;                    // two waves per SIMD at the measured 18.5 ns per MFMA
  std::printf("{\"source\": \"%s\", \"dma_per_wave\": %d, \"mfma_per_wave\": %d, \"wait_after_own_mfmas\": %d, \"ns_per_iteration\": %.1f, "
              "\"dma_tb_s\": %.2f, \"mfma_alone_ns_expected\": %.0f}\n",
              what, ND, NM, WAIT_AT_END, ns_it, tbs, mfma_ns);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return 0;
}


// the ring form: NS slots of ND KB per wave-set, the DMA of step h + PD threaded through the MFMAs of step h (one instruction
// per NM / ND MFMAs), counted wait: at the top of step h the instructions of steps h + 1 .. h + PD - 1 may stay in flight
template <int ND, int NM, int PD>
__global__ __launch_bounds__(512, 1) void k_ring(const float* src, unsigned src_bytes, float* sink, int iters) {
  extern __shared__ char lds[];
  const int t = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
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
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
  const unsigned gw = blockIdx.x * 8 + wv;
  const unsigned stride = gridDim.x * 8 * ND * 1024;
  unsigned base = gw * ND * 1024 + lane * 16;
  constexpr int NS = PD + 1, SLOT = 8 * ND * 1024;
  int slot = 0;
  for (int h = 0; h < PD; ++h) {  // prologue: steps 0 .. PD - 1
    char* dst = lds + slot * SLOT + wv * ND * 1024;
#pragma unroll
    for (int i = 0; i < ND; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + i * 1024), 16, (int)((base + i * 1024) % src_bytes), 0, 0, 0);
    base += stride;
    slot = slot == NS - 1 ? 0 : slot + 1;
  }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * ND) : "memory");
    __builtin_amdgcn_s_barrier();
    char* dst = lds + slot * SLOT + wv * ND * 1024;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (m % (NM / ND) == 0) {
        const int i = m / (NM / ND);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + i * 1024), 16, (int)((base + i * 1024) % src_bytes), 0, 0, 0);
        asm volatile("" ::: "memory");
      }
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
    }
    base += stride;
    slot = slot == NS - 1 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  if (s == 123.456f) sink[0] = s + lds[lane];
}

template <int ND, int NM, int PD>
static int run_ring(const float* src, size_t src_bytes, float* sink, int iters, const char* what) {
  auto kern = k_ring<ND, NM, PD>;
  const size_t lds = 128 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, src, (unsigned)src_bytes, sink, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, src, (unsigned)src_bytes, sink, iters);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double ns_it = ms * 1e6 / iters;
  std::printf("{\"source\": \"%s\", \"form\": \"ring\", \"dma_per_wave\": %d, \"mfma_per_wave\": %d, \"steps_ahead\": %d, \"ns_per_iteration\": %.1f, "
              "\"dma_tb_s\": %.2f, \"mfma_alone_ns_expected\": %.0f}\n",
              what, ND, NM, PD, ns_it, 256.0 * 8 * ND * 1024 / ns_it / 1e3, 2.0 * NM * 18.5);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return 0;
}

int main() {
  float *src, *sink;
  const size_t big = (size_t)1 << 30;  // 1 GiB: a stream from HBM
  CHECK(hipMalloc(&src, big));
  CHECK(hipMemset(src, 0x11, big));
  CHECK(hipMalloc(&sink, 64));
  const int iters = 400;
  for (int pass = 0; pass < 2; ++pass) {
    const size_t bytes = pass == 0 ? big : ((size_t)16 << 20);  // second pass: a 16 MiB window (256 CUs x 8 KB x 8: L2 / MALL hits)
    const char* what = pass == 0 ? "1 GiB stream" : "16 MiB window";
    if (run<8, 0>(src, bytes, sink, iters, what)) return 1;
    if (run<4, 0>(src, bytes, sink, iters, what)) return 1;
    if (run<0, 48>(src, bytes, sink, iters, what)) return 1;
    if (run<0, 24>(src, bytes, sink, iters, what)) return 1;
    if (run<8, 12>(src, bytes, sink, iters, what)) return 1;
    if (run<8, 24>(src, bytes, sink, iters, what)) return 1;
    if (run<8, 48>(src, bytes, sink, iters, what)) return 1;
    if (run<8, 96>(src, bytes, sink, iters, what)) return 1;
    if (run<4, 24>(src, bytes, sink, iters, what)) return 1;
    if (run<4, 48>(src, bytes, sink, iters, what)) return 1;
    if (run<8, 48, 1>(src, bytes, sink, iters, what)) return 1;
    if (run_ring<4, 24, 1>(src, bytes, sink, 2 * iters, what)) return 1;
    if (run_ring<4, 24, 2>(src, bytes, sink, 2 * iters, what)) return 1;
    if (run_ring<4, 24, 3>(src, bytes, sink, 2 * iters, what)) return 1;
    if (run_ring<8, 48, 1>(src, bytes, sink, iters, what)) return 1;
    if (run_ring<2, 24, 3>(src, bytes, sink, 2 * iters, what)) return 1;
    if (run_ring<4, 12, 3>(src, bytes, sink, 2 * iters, what)) return 1;
  }
  return 0;
}
