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

// gelu_f32 (ymk_common.h) on four values at once, written on vectors so that the multiplies and fused multiply-adds become
// packed instructions (v_pk_mul_f32 / v_pk_fma_f32: two values per issue slot); the same arithmetic per value.
__device__ __forceinline__ f32x4 gelu_f32x4(const f32x4 v) {
  f32x4 z, t, e;
#pragma unroll
  for (int i = 0; i < 4; ++i) z[i] = fabsf(v[i]);
  z = z * 0.70710678118654752440f;
  const f32x4 d = z * 0.39032074649205456f + 1.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_rcpf(d[i]);
  f32x4 q = t * -0.22690855651422961f + 0.8816638035254636f;
  q = q * t + -0.6277749224408846f;
  q = q * t + 0.6443424378640197f;
  q = q * t + 0.09342759526711675f;
  q = q * t + 0.23524963446014596f;
  const f32x4 w = z * z * -1.4426950408889634f;
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(w[i]);
  const f32x4 r = 1.f - (t * q) * e;  // erf(|v| / sqrt 2)
  const f32x4 hv = v * 0.5f;
  f32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = fmaf(hv[i], copysignf(r[i], v[i]), hv[i]);
  return out;
}

__device__ __forceinline__ mlp_f2 gelu_f32x2(const mlp_f2 v) {  // the same on a pair
  mlp_f2 z = {fabsf(v.x), fabsf(v.y)};
  z = z * 0.70710678118654752440f;
  const mlp_f2 d = z * 0.39032074649205456f + 1.f;
  const mlp_f2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  mlp_f2 q = t * -0.22690855651422961f + 0.8816638035254636f;
  q = q * t + -0.6277749224408846f;
  q = q * t + 0.6443424378640197f;
  q = q * t + 0.09342759526711675f;
  q = q * t + 0.23524963446014596f;
  const mlp_f2 w = z * z * -1.4426950408889634f;
  const mlp_f2 e = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const mlp_f2 r = 1.f - (t * q) * e;  // erf(|v| / sqrt 2)
  const mlp_f2 hv = v * 0.5f;
  return mlp_f2{fmaf(hv.x, copysignf(r.x, v.x), hv.x), fmaf(hv.y, copysignf(r.y, v.y), hv.y)};
}

// KT = D / 32 (the model width in 32-channel tiles), NCH = F / 32 (hidden chunks)
template <int KT, int NCH>
__global__ __launch_bounds__(256, 1) void k_vit_mlp_f16(MlpK p) {
  constexpr int D = 32 * KT, F = 32 * NCH;
  constexpr int W1_B = KT * 4096;  // a chunk of fc1 in LDS: KT K-tiles x [32 hidden rows][128 B]
  constexpr int W2_B = D * 128;    // a chunk of fc2 in LDS: [D output rows][128 B] (one 32-hidden tile)
  constexpr int N1 = 2, N2 = 3;    // ring depths: fc1's slab of chunk k is read ONE iteration before chunk k's turn, fc2's one AFTER
  constexpr int RING2 = N1 * W1_B;  // byte offset of the fc2 ring
  constexpr int TAB = RING2 + N2 * W2_B;
  constexpr int TAB_B = 2 * F * 4;  // fc1's scale and bias vectors, read per chunk by every lane
  constexpr int NDMA = (KT * 4 + D / 8) / 4;  // LDS-DMA instructions per wave per iteration (each moves 1 KB)
  static_assert((KT * 4) % 4 == 0 && (D / 8) % 4 == 0, "the slabs' DMA instructions divide among four waves");
  static_assert(TAB + TAB_B <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) char lds[TAB + TAB_B];
  float* const tab = reinterpret_cast<float*>(lds + TAB);  // [F] scale | [F] bias

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
  auto issue1 = [&](int c, int slot) {  // fc1's slab of chunk c: instruction i = wv + 4 q of KT * 4 - K tile i >> 2, rows 8 (i & 3) .. + 7
    char* base = lds + slot * W1_B;
#pragma unroll
    for (int q = 0; q < KT; ++q) {
      const int i = wv + 4 * q, kt = i >> 2, row = 8 * (i & 3) + jr;
      const unsigned off = (unsigned)(32 * c + row) * (unsigned)(KT * 128) + (unsigned)(kt * 128) + (unsigned)((js ^ ((row >> 1) & 7)) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (mlp_lds_void*)(base + kt * 4096 + (8 * (i & 3)) * 128), 16, (int)off, 0, 0, 0);
    }
  };
  auto issue2 = [&](int c, int slot) {  // fc2's slab of chunk c: instruction i = wv + 4 q of D / 8 - output rows 8 i .. + 7, K tile c
    char* base = lds + RING2 + slot * W2_B;
#pragma unroll
    for (int q = 0; q < D / 32; ++q) {
      const int i = wv + 4 * q, row = 8 * i + jr;
      const unsigned off = (unsigned)row * (unsigned)(NCH * 128) + (unsigned)(c * 128) + (unsigned)((js ^ ((row >> 1) & 7)) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (mlp_lds_void*)(base + (8 * i) * 128), 16, (int)off, 0, 0, 0);
    }
  };
  issue1(0, 0);
  issue1(1, 1);
  issue2(0, 0);

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

  // ---- the chunk's work as EIGHT slots of nine MFMAs and six fragment reads each.  Slots 0-3: the first product of chunk c + 1
  // (H^T = W1 . X^T: steps 3 j .. 3 j + 2 of its 2 KT k-steps, three MFMAs each - one per product term, into three
  // accumulators so that consecutive MFMAs are independent); slots 4-7: the second product of chunk c - 1 (three column tiles
  // of one k-step each).  The fragments of slot j + 1 are read at the START of slot j into the other half of a double buffer:
  // one wave per SIMD has nobody to cover an LDS round trip (~130 cycles), and a fragment read issued right before its MFMA
  // (what the compiler does left to itself: the first forms of this kernel, 1015-1048 us) stalls every MFMA for one.
  static_assert(KT % 3 == 0, "slots of three k-steps / three column tiles");
  constexpr int S1 = 2 * KT / 3, S2 = 2 * KT / 3, NSLOT = S1 + S2;  // 4 + 4 at KT = 6
  auto load_slot = [&](int j, int slot1, int slot2, mlp_h8 (&f)[6]) {
    if (j < S1) {
      const char* W1s = lds + slot1 * W1_B + li * 128;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int step = 3 * j + u, kt = step >> 1, s = step & 1;
        f[2 * u] = *reinterpret_cast<const mlp_h8*>(W1s + kt * 4096 + (((s * 2 + lh) ^ swz) * 16));
        f[2 * u + 1] = *reinterpret_cast<const mlp_h8*>(W1s + kt * 4096 + (((4 + s * 2 + lh) ^ swz) * 16));
      }
    } else {
      const char* W2s = lds + RING2 + slot2 * W2_B + li * 128;
      const int jj = j - S1, s = jj / (KT / 3), ct0 = 3 * (jj % (KT / 3));
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        f[2 * u] = *reinterpret_cast<const mlp_h8*>(W2s + (ct0 + u) * 4096 + (((s * 2 + lh) ^ swz) * 16));
        f[2 * u + 1] = *reinterpret_cast<const mlp_h8*>(W2s + (ct0 + u) * 4096 + (((4 + s * 2 + lh) ^ swz) * 16));
      }
    }
  };
  auto mfma_slot = [&](int j, const mlp_h8 (&f)[6], f32x16& h0, f32x16& h1, f32x16& h2, const mlp_h8 (&gh)[2], const mlp_h8 (&gl)[2]) {
    if (j < S1) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int step = 3 * j + u, kt = step >> 1, s = step & 1;
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * u + 1], xh[kt][s], h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * u], xl[kt][s], h1, 0, 0, 0);
        h2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * u], xh[kt][s], h2, 0, 0, 0);
      }
    } else {
      const int jj = j - S1, s = jj / (KT / 3), ct0 = 3 * (jj % (KT / 3));
#pragma unroll
      for (int u = 0; u < 3; ++u) acc2[ct0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[s], f[2 * u], acc2[ct0 + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 3; ++u) acc2[ct0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[s], f[2 * u + 1], acc2[ct0 + u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 3; ++u) acc2[ct0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[s], f[2 * u], acc2[ct0 + u], 0, 0, 0);
    }
  };

  // Software pipeline, one wave per SIMD: iteration c issues the first product of chunk c + 1, the scale / bias / GELU / plane
  // cut of chunk c (a sixteenth of it per MFMA pair: ~360 VALU instructions spread under 12 KT MFMAs) and the second product of
  // chunk c - 1.
  //   fc1 ring (2 slots): slab k is read at iteration k - 1, DMA-ed at iteration k - 2 into the slot slab k - 2 left at k - 3
  //   fc2 ring (3 slots): slab k is read at iteration k + 1, DMA-ed at iteration k - 1 into the slot slab k - 3 left at k - 2
  f32x16 c0, c1, c2;    // the three accumulators of the CURRENT chunk's first product
  mlp_h8 ph[2], pl[2];  // planes of the PREVIOUS chunk's hidden values (zero before the first chunk: its product adds nothing)
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) ph[s][e] = pl[s][e] = (_Float16)0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // fc1 slabs 0 and 1, fc2 slab 0
  __builtin_amdgcn_s_barrier();
  {  // the first chunk's first product (nothing to hide it under)
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = 0.f;
#pragma unroll
    for (int j = 0; j < S1; ++j) {
      mlp_h8 f[6];
      load_slot(j, 0, 0, f);
      mfma_slot(j, f, c0, c1, c2, ph, pl);
    }
  }
  int s1n = 1, s2p = 0, s2n = 1;  // ring slots: fc1 slab c + 1; fc2 slab c - 1 (slab 0 while c == 0: times zero planes); fc2 slab c + 1
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    // everything DMA-ed during the last iteration has landed once every wave's own share has; the same barrier tells that every
    // wave is done with the slots this iteration's DMAs overwrite
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 2 < NCH) issue1(c + 2, s1n ^ 1);
    if (c + 1 < NCH) issue2(c + 1, s2n);
    mlp_h8 fa[2][6];
    load_slot(0, s1n, s2p, fa[0]);
    // pre-activations of chunk c: register r of lane (li, lh) is hidden unit (r & 3) + 8 (r >> 2) + 4 lh
    f32x4 v[4], a[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // registers 4 g .. 4 g + 3: four consecutive hidden units
      const int hid = 32 * c + 8 * g + 4 * lh;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + hid), bi = *reinterpret_cast<const f32x4*>(tab + F + hid);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[g][e] = (c0[4 * g + e] + c1[4 * g + e]) + c2[4 * g + e];
      v[g] = v[g] * sc + bi;
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 n0, n1, n2;
#pragma unroll
    for (int r = 0; r < 16; ++r) n0[r] = n1[r] = n2[r] = 0.f;
    mlp_h8 gh[2], gl[2];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      if (j + 1 < NSLOT) load_slot(j + 1, s1n, s2p, fa[(j + 1) & 1]);  // (the last iteration multiplies a stale fc1 slab: no branch)
      mfma_slot(j, fa[j & 1], n0, n1, n2, ph, pl);
      {  // GELU of two of the sixteen values per slot (packed pairs); the plane cut once eight are through
        const int g = j >> 1, e0 = 2 * (j & 1);
        const mlp_f2 pair = gelu_f32x2(mlp_f2{v[g][e0], v[g][e0 + 1]});
        a[g][e0] = pair.x;
        a[g][e0 + 1] = pair.y;
        if (j == NSLOT / 2 - 1) mlp_split8(a[0], a[1], sg.x, gh[0], gl[0]);
        if (j == NSLOT - 1) mlp_split8(a[2], a[3], sg.x, gh[1], gl[1]);
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) {  // inside a slot: an MFMA, then its share of the slot's VALU work
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    c0 = n0;
    c1 = n1;
    c2 = n2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ph[s] = gh[s];
      pl[s] = gl[s];
    }
    s1n ^= 1;
    s2p = c == 0 ? 0 : (s2p == N2 - 1 ? 0 : s2p + 1);
    s2n = s2n == N2 - 1 ? 0 : s2n + 1;
  }
  {  // the last chunk's second product
    f32x16 d0, d1, d2;
#pragma unroll
    for (int j = S1; j < NSLOT; ++j) {
      mlp_h8 f[6];
      load_slot(j, 0, s2p, f);
      mfma_slot(j, f, d0, d1, d2, ph, pl);
    }
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
