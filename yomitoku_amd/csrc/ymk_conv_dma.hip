// fp16-split implicit-GEMM convolution, LDS-DMA form (gfx950): the same arithmetic as conv_igemm_split<.., FMT = 1> of
// ymk_conv_split.hip (two scaled fp16 planes per fp32 operand, 3 x v_mfma_f32_32x32x16_f16 per product tile, fp32 accumulate;
// models/dbnet_plus.py:33-38,56-127, rtdetr_backbone.py, parseq_transformer.py), restructured around what the PMC passes of
// that kernel showed (profiles/r03_conv_bf16_pmc_pass{1,2}.csv): MFMA pipe 41 % busy with the LDS 50 % busy and the waves
// parked at the per-K-tile barrier - sixteen 32 x 32 wave tiles read a whole A and B fragment per three MFMAs, and every K
// tile passes through registers (load, convert, ds_write) between two barriers.
//
//   * both operands travel global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write, no VALU): the
//     weight planes are stored in LDS layout already; the ACTIVATIONS land in LDS as the fp32 they are in HBM - a padding tap
//     or a row past M is an out-of-range buffer offset, for which the DMA writes zeros;
//   * a wave owns 32 rows x BN columns of a 256 x BN block tile (8 waves): it converts the fp32 A fragment it reads into the
//     two fp16 planes IN REGISTERS, once per 16-k step, for BN / 32 column tiles x 3 MFMAs (no other wave converts the same
//     rows), and it DMA-loads exactly the 32 A rows it will read itself;
//   * three LDS stages of one 32-k tile each (48 KB: 144 KB per block, one block per CU), loads two tiles ahead, ONE raw
//     s_barrier per K tile, counted `s_waitcnt vmcnt` (the DMA of tiles t+1 / t+2 stays in flight across the barrier);
//   * LDS rows are 128 B (32 fp32 of A; 2 planes x 32 halves of B) with the 16-byte slot index XOR-ed by (row >> 1) & 7 -
//     applied on the DMA's SOURCE address (the destination of a wave's DMA is linear) and on the fragment reads - so that
//     the 16 rows a ds_read_b128 phase touches fall on 16 different slots of the 256-byte bank line.
#include <atomic>
#include <string>

#include "ymk_conv_kernel.h"

namespace ymk {

typedef _Float16 hf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 hf16x8_t __attribute__((ext_vector_type(8)));
typedef float hf32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;


__device__ __forceinline__ float2 dma_f16_scales(unsigned amax_bits) {  // as f16_scales of ymk_conv_split.hip
  int e = (int)(amax_bits >> 23);
  e = e < 27 ? 27 : (e > 227 ? 227 : e);
  float2 r;
  r.x = __uint_as_float((unsigned)(268 - e) << 23);
  r.y = __uint_as_float((unsigned)(e - 14) << 23);
  return r;
}

// 8 fp32 (two 16-byte LDS slots), times the power of two sa -> hi and lo planes of 8 halves each
__device__ __forceinline__ void split8(const f32x4 u, const f32x4 v, float sa, hf16x8_t& hi, hf16x8_t& lo) {
  hf32x2_t x[4] = {{u.x, u.y}, {u.z, u.w}, {v.x, v.y}, {v.z, v.w}};
  hf16x2_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[i] *= sa;
    h[i] = __builtin_convertvector(x[i], hf16x2_t);
    x[i] -= __builtin_convertvector(h[i], hf32x2_t);  // exact
    l[i] = __builtin_convertvector(x[i], hf16x2_t);
  }
  hi = hf16x8_t{h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
  lo = hf16x8_t{l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
}

// DMA_WAVES waves of 32 rows x BN columns (block tile 32 DMA_WAVES x BN), DMA_NST LDS stages of one 32-k tile, loads
// DMA_NST - 1 tiles ahead.  <BN, 8, 3>: 256-row tiles, 144 KB, one block per CU.  <BN, 4, 2>: 128-row tiles, 68 KB, TWO blocks
// per CU - another block's main loop covers a block's prologue and epilogue (what the short-K layers need).
// APL: the ACTIVATIONS are fp16 planes in HBM already (Tensor::planes: the 128 bytes of a pixel's 32-channel slice are 32 high
// halves then 32 low halves, written by the producing convolution's epilogue under the scale of the record p.amax): the
// fragment read is two ds_read_b128 straight into the MFMA operands - no fp32 -> (h, l) conversion in the loop, where the fp32
// form spends 12 VALU instructions per 16-k step and tap (6.8 VALU per MFMA on the 3 x 3 layers, the MFMA pipe 0.47 busy:
// profiles/r04_conv_f16_short_k_pmc_pass*.csv).  Same DMA addressing, same swizzle, same products in the same order.
// OPL: the epilogue writes planes (epilogue_tile<.., PL>).
template <int BN, int DMA_WAVES, int DMA_NST, bool APL = false, bool OPL = false>
__global__ __launch_bounds__(64 * DMA_WAVES, 2) void conv_f16_dma(ConvK p, const uint4* __restrict__ wsplit, unsigned w_bytes) {
  constexpr int DMA_BM = 32 * DMA_WAVES, DMA_NT = 64 * DMA_WAVES;
  constexpr int TN = BN / 32;                 // 32-column MFMA tiles of a wave
  constexpr int A_STAGE = DMA_BM * 128;       // bytes: BM rows x 32 fp32
  constexpr int B_STAGE = BN * 128;           // bytes: BN rows x 2 planes x 32 halves
  constexpr int STAGE_B = A_STAGE + B_STAGE;
  constexpr int BROWS = BN / DMA_WAVES;       // B rows a wave loads per K tile
  constexpr int BI = BROWS / 8;               // B DMA instructions per wave per K tile (8 rows each)
  constexpr int NLOAD = 4 + BI;               // DMA instructions a wave issues per K tile
  constexpr int LDC = BN + 4;
  constexpr int PD = DMA_NST - 1;             // prefetch distance in K tiles
  constexpr int EPI_B = DMA_BM * LDC * 4;     // the fp32 output tile of the epilogue
  constexpr int LDS_B = DMA_NST * STAGE_B > EPI_B ? DMA_NST * STAGE_B : EPI_B;
  static_assert(BROWS % 8 == 0 && PD >= 1 && PD <= 2, "shape");
  __shared__ __attribute__((aligned(16))) char lds[LDS_B];

  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
  const float2 sc = dma_f16_scales((unsigned)__builtin_amdgcn_readfirstlane((int)amax_read(p.amax, t)));
  const float sa = sc.x, inv_sa = sc.y;
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / p.ntiles_n, tile_n = tile - tile_m * p.ntiles_n;
  const int m0 = tile_m * DMA_BM, n0 = tile_n * BN;

  // ---- DMA geometry of this lane: A instruction i covers rows 32 wv + 8 i + (lane >> 3) (the wave's own rows), 16 bytes
  // each; the LDS destination of lane j is row (j >> 3), PHYSICAL slot (j & 7), which must receive LOGICAL slot
  // (j & 7) ^ ((row >> 1) & 7) of the row's 128 bytes
  const int jr = lane >> 3, js = lane & 7;
  int pixb[4], ih0[4], iw0[4];
  unsigned cbyte[4];  // byte offset of the lane's logical slot within a 32-channel tile
  const bool pointwise = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.stride_w == 1 && p.pad == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 32 * wv + 8 * i + jr, m = m0 + row;
    cbyte[i] = (unsigned)((js ^ ((row >> 1) & 7)) * 16);
    if (m < p.M && pointwise) {
      pixb[i] = m;
      ih0[i] = 0;
      iw0[i] = 0;
    } else if (m < p.M) {
      const int ohw = p.OH * p.OW;
      const int n = m / ohw, rem = m - n * ohw;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      pixb[i] = n * p.H * p.W;
      ih0[i] = oh * p.stride - p.pad;
      iw0[i] = ow * p.stride_w - p.pad;
    } else {
      pixb[i] = 0;
      ih0[i] = -(1 << 20);
      iw0[i] = 0;
    }
  }
  const int ktiles = p.Kpad >> 5;
  unsigned boff[BI];  // byte offset of the lane's 16 bytes of B within the weight panel, K tile 0
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int row = BROWS * wv + 8 * j + jr;  // the wave's share of the B rows
    boff[j] = (unsigned)(n0 + row) * (unsigned)(ktiles * 128) + (unsigned)((js ^ ((row >> 1) & 7)) * 16);
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wsplit), 0, w_bytes, 0x00020000);

  int cur_kh = 0, cur_kw = 0, cur_cc = 0;
  unsigned voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) voff[i] = OOB_OFFSET;

  // K tile kt -> LDS stage st: 4 + BI LDS-DMA instructions of this wave
  auto issue = [&](int kt, int st) {
    if (cur_cc == 0 || p.KH * p.KW > 1) {  // wave-uniform: a new filter tap - every K tile of a k x k layer (channel-major panels)
      const int dh = cur_kh * p.dil, dw = cur_kw * p.dil;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ih = ih0[i] + dh, iw = iw0[i] + dw;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const unsigned off = (unsigned)(pixb[i] + ih * p.W + iw) * (unsigned)p.in_ld * 4u + cbyte[i];
        voff[i] = ok ? off : OOB_OFFSET;
      }
    }
    char* As = lds + st * STAGE_B + (32 * wv) * 128;
    const int soff = cur_cc * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool chan_ok = cur_cc * 128 + (int)cbyte[i] < p.C * 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(As + i * 1024), 16, (int)(chan_ok ? voff[i] : OOB_OFFSET), soff, 0, 0);
    }
    char* Bs = lds + st * STAGE_B + A_STAGE + (BROWS * wv) * 128;
#pragma unroll
    for (int j = 0; j < BI; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void*)(Bs + j * 1024), 16, (int)boff[j], kt * 128, 0, 0);
    if (++cur_kw == p.KW) {  // the taps of one 32-channel slice back to back (k_split_panel_f16: channel-major K order)
      cur_kw = 0;
      if (++cur_kh == p.KH) {
        cur_kh = 0;
        ++cur_cc;
      }
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  // fragment addresses: A row 32 wv + li (the wave's own rows), B rows 32 b + li; slot swizzle by (row >> 1) & 7
  const int arow = 32 * wv + li;
  const int aswz = (arow >> 1) & 7, bswz = (li >> 1) & 7;  // (32 b + li) >> 1 & 7 == (li >> 1) & 7
  // one 16-k step (s = 0, 1) of the K tile in stage st; the lane holds k = 8 lh .. + 7 of the step
  auto compute = [&](int st, int s) {
    const char* As = lds + st * STAGE_B + arow * 128;
    const char* Bs = lds + st * STAGE_B + A_STAGE + li * 128;
    {
      hf16x8_t ah, al;
      if constexpr (APL) {  // k = 16 s + 8 lh .. + 7 of the slice: high halves in slots 0-3, low halves in slots 4-7
        ah = *reinterpret_cast<const hf16x8_t*>(As + (((s * 2 + lh) ^ aswz) * 16));
        al = *reinterpret_cast<const hf16x8_t*>(As + (((4 + s * 2 + lh) ^ aswz) * 16));
      } else {
        const int ca = s * 4 + lh * 2;
        const f32x4 u = *reinterpret_cast<const f32x4*>(As + ((ca ^ aswz) * 16));
        const f32x4 v = *reinterpret_cast<const f32x4*>(As + (((ca + 1) ^ aswz) * 16));
        split8(u, v, sa, ah, al);
      }
      hf16x8_t bh[TN], bl[TN];
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        bh[b] = *reinterpret_cast<const hf16x8_t*>(Bs + b * 32 * 128 + (((s * 2 + lh) ^ bswz) * 16));
        bl[b] = *reinterpret_cast<const hf16x8_t*>(Bs + b * 32 * 128 + (((4 + s * 2 + lh) ^ bswz) * 16));
      }
      // smallest terms first, term-major: consecutive MFMAs go to different accumulators
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[b], acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[b], acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[b], acc[b], 0, 0, 0);
    }
  };

  issue(0, 0);
  if (PD > 1 && ktiles > 1) issue(1, 1);
  int st = 0, stp = PD;  // stage of tile kt / of tile kt + PD
  for (int kt = 0; kt < ktiles; ++kt) {
    // tile kt has landed once this wave's own DMAs of it have (a younger tile's may stay in flight) and every wave has
    // said so; the same barrier tells that every wave is done reading the stage tile kt + PD is about to overwrite
    if (PD > 1 && kt + 1 < ktiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (PD == 1 && kt + 1 < ktiles) issue(kt + 1, stp);  // one tile ahead: as early as the barrier allows
    compute(st, 0);
    if (PD > 1 && kt + PD < ktiles) issue(kt + PD, stp);  // behind the first step's MFMAs: the address arithmetic rides in their shadow
    compute(st, 1);
    st = st == DMA_NST - 1 ? 0 : st + 1;
    stp = stp == DMA_NST - 1 ? 0 : stp + 1;
  }

  // ---- epilogue: accumulators (times 1 / sa: exact) -> LDS as [256][BN + 4] fp32 -> the shared coalesced epilogue
  __syncthreads();
  float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * lh;
      Cs[row * LDC + b * 32 + li] = acc[b][r] * inv_sa;
    }
  __syncthreads();
  epilogue_tile<DMA_BM, BN, DMA_NT, false, 4, OPL>(p, Cs, m0, n0, t);
}

template <int BN, int WAVES, int NST, bool APL = false, bool OPL = false>
static void launch_dma(hipStream_t s, ConvK& k, const void* wsplit, size_t w_bytes) {
  const int mt = (k.M + 32 * WAVES - 1) / (32 * WAVES), nt = (k.Cout + BN - 1) / BN;
  k.ntiles_n = nt;
  hipLaunchKernelGGL((conv_f16_dma<BN, WAVES, NST, APL, OPL>), dim3(mt * nt), dim3(64 * WAVES), 0, s, k, reinterpret_cast<const uint4*>(wsplit), (unsigned)w_bytes);
}

template <int BN>
static void launch_dma_planes(hipStream_t s, ConvK& k, const void* wsplit, size_t w_bytes) {
  if (k.in_planes && k.out_planes) launch_dma<BN, 4, 2, true, true>(s, k, wsplit, w_bytes);
  else if (k.in_planes) launch_dma<BN, 4, 2, true, false>(s, k, wsplit, w_bytes);
  else launch_dma<BN, 4, 2, false, true>(s, k, wsplit, w_bytes);
}

// the caller (conv2d_split) has resolved the panel and the input's max|x| record.  rows: 128 (4 waves, two stages, two or three
// blocks per CU: the form the dispatch uses) or 256 (8 waves, three stages, one block per CU: kept for A/B runs - level on the
// long-K layers, behind wherever a block's prologue / epilogue weighs, profiles/r04_conv_sweep_f16_lds_dma.txt; a three-stage
// 128 x 64 form measured no better than the two-stage one and was dropped)
bool conv2d_f16_dma(hipStream_t s, ConvK& k, const void* wsplit, size_t w_bytes, bool narrow, int rows) {
  if (w_bytes >= (size_t)OOB_OFFSET) return false;
  if (k.in_planes || k.out_planes) {  // the plane forms exist for the dispatch's own tile (128 rows) only
    if (rows != 128) return false;
    if (narrow) launch_dma_planes<64>(s, k, wsplit, w_bytes);
    else launch_dma_planes<128>(s, k, wsplit, w_bytes);
    return true;
  }
  if (rows == 128) {
    if (narrow) launch_dma<64, 4, 2>(s, k, wsplit, w_bytes);
    else launch_dma<128, 4, 2>(s, k, wsplit, w_bytes);
  } else {
    if (narrow) launch_dma<64, 8, 3>(s, k, wsplit, w_bytes);
    else launch_dma<128, 8, 3>(s, k, wsplit, w_bytes);
  }
  return true;
}

}  // namespace ymk
