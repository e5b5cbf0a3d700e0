// One launch for the MLP half of a ViT block on gfx950: x <- x + fc2(GELU(fc1(LayerNorm(x)))), fp16-split arithmetic (two scaled
// fp16 planes per fp32 operand, three v_mfma_f32_32x32x16_f16 per product tile, fp32 accumulate - ymk_conv_split.hip), the
// hidden state never leaving the chip.  Replaces k_layernorm + the fc1 GEMM (GELU epilogue) + the fc2 GEMM (residual epilogue)
// of timm's Block (models/layers/parseq_transformer.py:188-204: norm2 -> mlp.fc1 -> act -> mlp.fc2, drop-path identity at
// inference), which wrote and re-read a [tokens][4 D] fp32 tensor per block: 1.05 GB each way at a wave's 342 624 tokens -
// the two largest items of the by-layer table (profiles/r05_conv_two_roof_by_layer.md: fc1 0.26, fc2 0.35 of the HBM roof).
//
// A block owns 128 token rows, a wave 32 of them, for the WHOLE layer (one block per CU, one wave per SIMD, the 512-register
// budget of that regime):
//   * prologue: the wave's rows come in once, are LayerNorm-ed in registers (a row lives in lanes li and li + 32: one
//     xor-shuffle per statistic) and cut into the two planes - 16 VGPRs per 32 channels, held for the whole kernel;
//   * the hidden units go by in chunks of 32.  Per chunk the 32 x D slab of fc1's planes and the D x 32 slab of fc2's planes
//     (48 KB at D = 192) arrive by LDS-DMA, three stages, two chunks ahead, one barrier per chunk;
//   * first product TRANSPOSED: H^T = W1[chunk] . X^T (A operand = the weight rows from LDS, B operand = the row planes in
//     registers), so that a lane's sixteen accumulator registers all belong to ITS row (column l & 31 of H^T) and are - after
//     scale, bias, GELU and the cut into planes - exactly the A operand of the second product, no shuffle, no LDS: register
//     8 s + j of lane (li, lh) is hidden unit 16 s + 8 (j >> 2) + 4 lh + (j & 3) of the chunk, and the k order of a product
//     is free, so fc2's planes are stored with the hidden units of every 32-chunk in THAT order (k_split_panel_f16 `perm`);
//   * the planes of the hidden values need their scale before the values exist: the bound pl_a(fc1) x (the LayerNorm's static
//     output bound) + pl_b(fc1) - GELU only shrinks - as for the planes in HBM (ymk_common.h, Tensor::planes);
//   * second product accumulates the wave's 32 x D outputs over all chunks (D / 2 accumulator registers); the epilogue adds
//     bias and the residual row (re-read: an L2 hit) and stores.
#include <string>

#include "ymk_conv_kernel.h"

namespace ymk {

typedef _Float16 mlp_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 mlp_h8 __attribute__((ext_vector_type(8)));
typedef float mlp_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void mlp_lds_void;

struct MlpK {
  const float* x;   // [M][ld] token rows in
  float* out;       // [M][ld] out (may be x)
  int M, ld;
  const float *ln_g, *ln_b;  // [D]
  float ln_eps, ln_bound;    // LayerNorm epsilon; its static output bound (scale of the row planes)
  const uint4* w1;           // fc1 planes [F^256][KT][2][32] halves (the standard fp16 panel: rows = hidden units)
  unsigned w1_bytes;
  const float *s1, *b1;      // [F]: epilogue scale (row's power of two taken back out) and bias of fc1
  const uint4* w2;           // fc2 planes [D^256][F / 32][2][32] halves, hidden units of every 32-chunk in accumulator order
  unsigned w2_bytes;
  const float *s2, *b2;      // [D]
  float g_bound;             // bound on |GELU(fc1(..))|: scale of the hidden planes
};

__device__ __forceinline__ void mlp_split8(const f32x4 u, const f32x4 v, float sa, mlp_h8& hi, mlp_h8& lo) {
  mlp_f2 x[4] = {{u.x, u.y}, {u.z, u.w}, {v.x, v.y}, {v.z, v.w}};
  mlp_h2 h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[i] *= sa;
    h[i] = __builtin_convertvector(x[i], mlp_h2);
    x[i] -= __builtin_convertvector(h[i], mlp_f2);  // exact
    l[i] = __builtin_convertvector(x[i], mlp_h2);
  }
  hi = mlp_h8{h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
  lo = mlp_h8{l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
}

// KT = D / 32 (the model width in 32-channel tiles), NCH = F / 32 (hidden chunks)
template <int KT, int NCH>
__global__ __launch_bounds__(256, 1) void k_vit_mlp_f16(MlpK p) {
  constexpr int D = 32 * KT, F = 32 * NCH;
  constexpr int W1_B = KT * 4096;  // a chunk of fc1 in LDS: KT K-tiles x [32 hidden rows][128 B]
  constexpr int W2_B = D * 128;    // a chunk of fc2 in LDS: [D output rows][128 B] (one 32-hidden tile)
  constexpr int STAGE = W1_B + W2_B;
  constexpr int NST = 3;
  constexpr int TAB_B = 2 * F * 4;  // fc1's scale and bias vectors, read per chunk by every lane
  constexpr int NDMA = (KT * 4 + D / 8) / 4;  // LDS-DMA instructions per wave per chunk (each moves 1 KB)
  static_assert((KT * 4) % 4 == 0 && (D / 8) % 4 == 0, "the chunk's DMA instructions divide among four waves");
  static_assert(NST * STAGE + TAB_B <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE + TAB_B];
  float* const tab = reinterpret_cast<float*>(lds + NST * STAGE);  // [F] scale | [F] bias

  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
  const float2 sx = f16_plane_scales(__float_as_uint(p.ln_bound));
  const float2 sg = f16_plane_scales(__float_as_uint(p.g_bound));
  const int m0 = blockIdx.x * 128 + 32 * wv;  // first row of this wave

  // ---- weight slabs: chunk c -> stage st.  A DMA instruction writes 1 KB = 8 rows x 128 B linearly; the lane's 16 bytes are
  // PHYSICAL slot js of row jr of the eight, and must hold LOGICAL slot js ^ ((row >> 1) & 7) of that row (the swizzle the
  // fragment reads undo: ymk_conv_dma.hip)
  const __amdgpu_buffer_rsrc_t rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.w1), 0, p.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.w2), 0, p.w2_bytes, 0x00020000);
  const int jr = lane >> 3, js = lane & 7;
  auto issue = [&](int c, int st) {
    char* base = lds + st * STAGE;
#pragma unroll
    for (int q = 0; q < KT; ++q) {  // fc1: instruction i = wv + 4 q of KT * 4: K tile i >> 2, rows 8 (i & 3) .. + 7 of the chunk
      const int i = wv + 4 * q, kt = i >> 2, row = 8 * (i & 3) + jr;
      const unsigned off = (unsigned)(32 * c + row) * (unsigned)(KT * 128) + (unsigned)(kt * 128) + (unsigned)((js ^ ((row >> 1) & 7)) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (mlp_lds_void*)(base + kt * 4096 + (8 * (i & 3)) * 128), 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < D / 32; ++q) {  // fc2: instruction i = wv + 4 q of D / 8: output rows 8 i .. + 7, K tile c
      const int i = wv + 4 * q, row = 8 * i + jr;
      const unsigned off = (unsigned)row * (unsigned)(NCH * 128) + (unsigned)(c * 128) + (unsigned)((js ^ ((row >> 1) & 7)) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (mlp_lds_void*)(base + W1_B + (8 * i) * 128), 16, (int)off, 0, 0, 0);
    }
  };
  issue(0, 0);
  issue(1, 1);

  // fc1's scale | bias into LDS (published by the first chunk's barrier)
  for (int i = t; i < F; i += 256) {
    tab[i] = p.s1[i] * sx.y;  // with 1 / (scale of the row planes) folded in
    tab[F + i] = p.b1[i];
  }

  // ---- the wave's 32 rows: load, LayerNorm (as k_layernorm / conv_f16_astat<.., LN>), planes
  mlp_h8 xh[KT][2], xl[KT][2];
  {
    const size_t in_bytes = (size_t)p.M * (size_t)p.ld * 4;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (unsigned)in_bytes, 0x00020000);
    const int m = m0 + li;
    const unsigned row_off = m < p.M ? (unsigned)m * (unsigned)p.ld * 4u : OOB_OFFSET;
    f32x4 u[KT][2], v[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int c = kt * 32 + s * 16 + lh * 8;
        const unsigned o0 = row_off != OOB_OFFSET ? row_off + (unsigned)c * 4u : OOB_OFFSET;
        const unsigned o1 = row_off != OOB_OFFSET ? row_off + (unsigned)(c + 4) * 4u : OOB_OFFSET;
        u[kt][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)o0, 0, 0));
        v[kt][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)o1, 0, 0));
      }
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        sum += ((u[kt][s].x + u[kt][s].y) + (u[kt][s].z + u[kt][s].w)) + ((v[kt][s].x + v[kt][s].y) + (v[kt][s].z + v[kt][s].w));
    sum += __shfl_xor(sum, 32);
    const float mean = sum / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u[kt][s] -= mean;
        v[kt][s] -= mean;
        sq += ((u[kt][s].x * u[kt][s].x + u[kt][s].y * u[kt][s].y) + (u[kt][s].z * u[kt][s].z + u[kt][s].w * u[kt][s].w)) +
              ((v[kt][s].x * v[kt][s].x + v[kt][s].y * v[kt][s].y) + (v[kt][s].z * v[kt][s].z + v[kt][s].w * v[kt][s].w));
      }
    sq += __shfl_xor(sq, 32);
    const float rstd = 1.f / sqrtf(sq / (float)D + p.ln_eps);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int c = kt * 32 + s * 16 + lh * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_g + c), g1 = *reinterpret_cast<const f32x4*>(p.ln_g + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln_b + c), b1 = *reinterpret_cast<const f32x4*>(p.ln_b + c + 4);
        mlp_split8(u[kt][s] * rstd * g0 + b0, v[kt][s] * rstd * g1 + b1, sx.x, xh[kt][s], xl[kt][s]);
      }
  }

  f32x16 acc2[KT];
#pragma unroll
  for (int ct = 0; ct < KT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[ct][r] = 0.f;
  const int swz = (li >> 1) & 7;  // rows li, 32 + li, .. of a slab: (row >> 1) & 7 is the same for all of them

  int st = 0, stp = 2;
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    // chunk c has landed once this wave's own DMAs of it have (the next chunk's may stay in flight) and every wave says so;
    // the same barrier tells that every wave is done with the stage chunk c + 2 is about to overwrite
    if (c + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 2 < NCH) issue(c + 2, stp);
    const char* W1s = lds + st * STAGE + li * 128;
    const char* W2s = lds + st * STAGE + W1_B + li * 128;

    // ---- H^T chunk = W1[chunk] . X^T: three accumulators (one per product term) keep consecutive MFMAs independent
    f32x16 h0, h1, h2;
#pragma unroll
    for (int r = 0; r < 16; ++r) h0[r] = h1[r] = h2[r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const mlp_h8 wh = *reinterpret_cast<const mlp_h8*>(W1s + kt * 4096 + (((s * 2 + lh) ^ swz) * 16));
        const mlp_h8 wl = *reinterpret_cast<const mlp_h8*>(W1s + kt * 4096 + (((4 + s * 2 + lh) ^ swz) * 16));
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[kt][s], h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[kt][s], h1, 0, 0, 0);
        h2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[kt][s], h2, 0, 0, 0);
      }
    // ---- scale, bias, GELU, planes: register r of lane (li, lh) is hidden unit (r & 3) + 8 (r >> 2) + 4 lh of the chunk
    mlp_h8 gh[2], gl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 a[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {  // registers 8 s + 4 q .. + 3: four consecutive hidden units
        const int hid = 32 * c + 8 * (2 * s + q) + 4 * lh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + hid), bi = *reinterpret_cast<const f32x4*>(tab + F + hid);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 8 * s + 4 * q + e;
          a[q][e] = gelu_f32(((h0[r] + h1[r]) + h2[r]) * sc[e] + bi[e]);
        }
      }
      mlp_split8(a[0], a[1], sg.x, gh[s], gl[s]);
    }
    // ---- out += G[chunk] . W2[chunk]^T over the D / 32 column tiles
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      mlp_h8 bh[KT], bl[KT];
#pragma unroll
      for (int ct = 0; ct < KT; ++ct) {
        bh[ct] = *reinterpret_cast<const mlp_h8*>(W2s + ct * 4096 + (((s * 2 + lh) ^ swz) * 16));
        bl[ct] = *reinterpret_cast<const mlp_h8*>(W2s + ct * 4096 + (((4 + s * 2 + lh) ^ swz) * 16));
      }
#pragma unroll
      for (int ct = 0; ct < KT; ++ct) acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[s], bh[ct], acc2[ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < KT; ++ct) acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[s], bl[ct], acc2[ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < KT; ++ct) acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[s], bh[ct], acc2[ct], 0, 0, 0);
    }
    st = st == NST - 1 ? 0 : st + 1;
    stp = stp == NST - 1 ? 0 : stp + 1;
  }

  // ---- epilogue: out[row][col] = x[row][col] + acc * (1 / scale of the hidden planes) * s2[col] + b2[col]; accumulator
  // register r of lane (li, lh) is row (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32, column 32 ct + li
  const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (unsigned)((size_t)p.M * p.ld * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (unsigned)((size_t)p.M * p.ld * 4), 0x00020000);
#pragma unroll
  for (int ct = 0; ct < KT; ++ct) {
    const int co = 32 * ct + li;
    const float sc = p.s2[co] * sg.y, bi = p.b2[co];
    float rr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const unsigned ro = row < p.M ? ((unsigned)row * (unsigned)p.ld + (unsigned)co) * 4u : OOB_OFFSET;
      rr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, (int)ro, 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const unsigned oo = row < p.M ? ((unsigned)row * (unsigned)p.ld + (unsigned)co) * 4u : OOB_OFFSET;
      const float y = (acc2[ct][r] * sc + bi) + rr[r];
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rsrc_o, (int)oo, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // one column tile's residual values in flight at a time
  }
}

// the launch; the caller (ymk_conv_split.hip: vit_mlp_fused) has resolved the planes.  D = 192 / F = 768 (PARSeq-tiny's ViT)
// is the instantiation the register file and the LDS hold; anything else is refused.
bool vit_mlp_f16_launch(hipStream_t s, const MlpK& k, int D, int F) {
  if (D != 192 || F != 768 || k.M <= 0 || (size_t)k.M * (size_t)k.ld * 4 >= (size_t)OOB_OFFSET) return false;
  hipLaunchKernelGGL((k_vit_mlp_f16<6, 24>), dim3((k.M + 127) / 128), dim3(256), 0, s, k);
  return true;
}

}  // namespace ymk
