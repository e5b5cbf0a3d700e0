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

// tools/jobs: -DYMK_ABLATE=n builds a copy of the kernel with one part of its loop removed, to time the parts against the whole
// (1: no DMA inside the loop, 2: no fragment reads / MFMAs, 3: no barrier, 4: fragment reads and conversions but no MFMA,
// 5: the MFMAs on made-up fragments, no LDS read; wide form only: 6 / 7 / 8: the A / B / both DMA streams read contiguous
// memory instead of the tile's rows).  The results of such a build are wrong by construction; the product build has n = 0.
#ifndef YMK_ABLATE
#define YMK_ABLATE 0
#endif

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
    if constexpr (YMK_ABLATE == 2) return;
    if constexpr (YMK_ABLATE == 5) {  // the MFMAs alone, on made-up fragments
      hf16x8_t f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (_Float16)(0.01f * (float)(lane + e + s));
      asm volatile("" : "+v"(f));
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, f, acc[b], 0, 0, 0);
      return;
    }
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
      if constexpr (YMK_ABLATE == 4) {
        asm volatile("" ::"v"(ah), "v"(al));
#pragma unroll
        for (int b = 0; b < TN; ++b) asm volatile("" ::"v"(bh[b]), "v"(bl[b]));
        return;
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
    if constexpr (YMK_ABLATE != 3) __builtin_amdgcn_s_barrier();
    if (YMK_ABLATE != 1 && PD == 1 && kt + 1 < ktiles) issue(kt + 1, stp);  // one tile ahead: as early as the barrier allows
    compute(st, 0);
    if (YMK_ABLATE != 1 && PD > 1 && kt + PD < ktiles) issue(kt + PD, stp);  // behind the first step's MFMAs: the address arithmetic rides in their shadow
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

// ---- the wide form: WM x WN waves of (32 TM) x (32 TN) outputs each on a (32 TM WM) x (32 TN WN) block tile - <4, 2, 2, 4>: 256 x 256,
// eight waves of 64 x 128, two LDS stages of 64 KB, one block per CU.  Why: the copies of the 128-row form with one part of the
// loop removed (profiles/r05_conv_dma_ablation.md) - on the 3 x 3 512 -> 512 layer the DMA stream and its barriers ALONE take
// 570 us of the kernel's 962, the fragment reads + MFMAs alone 659: a 128 x 128 tile asks the L2 for 32 KB per 3.1 MFLOP of
// MFMA work (8.7 GB for the layer, 15 TB/s in that run), twice what the LDS can hold in flight covers at the MFMA rate.  A
// 256 x 256 tile moves half the bytes per product (64 KB per 12.6 MFLOP), and a 64 x 128 wave tile reads 12 fragments per 24
// MFMAs where the 32 x 128 one reads 10 per 12.  Same DMA addressing, same swizzle, same conversion, the same three products per
// accumulator in the same order: bit-identical outputs.
template <int WM, int WN, int TM, int TN, bool APL = false, bool OPL = false>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_f16_dma_wide(ConvK p, const uint4* __restrict__ wsplit, unsigned w_bytes) {
  constexpr int NW = WM * WN, NT = 64 * NW, BM = 32 * TM * WM, BN = 32 * TN * WN, WTM = 32 * TM, WTN = 32 * TN;
  constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128, STAGE_B = A_STAGE + B_STAGE;
  constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;  // DMA instructions (8 rows x 128 B each) per wave and K tile
  constexpr int EROWS = BM < 128 ? BM : 128, LDC = BN + 4, EPI_B = EROWS * LDC * 4;
  constexpr int LDS_B = 2 * STAGE_B > EPI_B ? 2 * STAGE_B : EPI_B;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && LDS_B <= 160 * 1024 && WTM <= EROWS && EROWS % WTM == 0, "shape");
  __shared__ __attribute__((aligned(16))) char lds[LDS_B];

  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
  const int wm = wv / WN, wn = wv - wm * WN;
  const float2 sc = dma_f16_scales((unsigned)__builtin_amdgcn_readfirstlane((int)amax_read(p.amax, t)));
  const float sa = sc.x, inv_sa = sc.y;
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / p.ntiles_n, tile_n = tile - tile_m * p.ntiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // DMA geometry (as conv_f16_dma): instruction i of this wave moves rows 8 (AI wv + i) .. + 7 of the A tile, instruction j
  // rows 8 (BI wv + j) .. + 7 of the B tile; lane (jr, js) carries the 16 bytes whose LOGICAL slot is js ^ ((row >> 1) & 7)
  const int jr = lane >> 3, js = lane & 7;
  int pixb[AI], ih0[AI], iw0[AI];
  unsigned cbyte[AI];
  const bool pointwise = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.stride_w == 1 && p.pad == 0;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = 8 * (AI * wv + i) + jr, m = m0 + row;
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
  unsigned boff[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int row = 8 * (BI * wv + j) + jr;
    boff[j] = (unsigned)(n0 + row) * (unsigned)(ktiles * 128) + (unsigned)((js ^ ((row >> 1) & 7)) * 16);
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wsplit), 0, w_bytes, 0x00020000);

  // ---- the DMA of K tile kt + 1 is PREPARED once per iteration (tap arithmetic: VALU) and ISSUED one instruction at a time
  // inside the MFMA stream of tile kt.  Issued in one go after the barrier (the first form of this kernel, and the 128-row
  // kernel above) a wave sits at its third or fourth buffer_load until the CU's texture path has drained the others' - 64 KB
  // per iteration and CU at the ~30 B / clock that path moves into LDS (tools/diag/dma_mfma_overlap.hip: 19 TB/s over the chip
  // from an L2 / MALL window) is a third of the iteration's MFMA time during which its in-order instruction stream feeds no
  // MFMA: the DMA stream's time ADDED to the MFMAs' (ablation copies: 722 us with both, 370 / ~480 each alone).
  int cur_kh = 0, cur_kw = 0, cur_cc = 0;
  unsigned a_off[AI];
  int a_soff = 0, b_soff = 0;
  auto prep = [&](int kt, bool live) {  // live = false past the last K tile: every offset out of range (the DMA writes zeros)
    const int dh = cur_kh * p.dil, dw = cur_kw * p.dil;
    const int cc = __builtin_amdgcn_readfirstlane(cur_cc);
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int ih = ih0[i] + dh, iw = iw0[i] + dw;
      const bool ok = live && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && cc * 128 + (int)cbyte[i] < p.C * 4;
      const unsigned off = (unsigned)(pixb[i] + ih * p.W + iw) * (unsigned)p.in_ld * 4u + cbyte[i];
      a_off[i] = ok ? off : OOB_OFFSET;
    }
    a_soff = cc * 128;
    b_soff = live ? kt * 128 : 0;
    if (++cur_kw == p.KW) {  // the taps of one 32-channel slice back to back (k_split_panel_f16: channel-major K order)
      cur_kw = 0;
      if (++cur_kh == p.KH) {
        cur_kh = 0;
        ++cur_cc;
      }
    }
  };
  const int wv_u = __builtin_amdgcn_readfirstlane(wv);  // (the LDS address of a DMA travels in M0)
  unsigned seq = 0;  // (YMK_ABLATE 6 - 8: the DMA instructions read a contiguous stream instead of the tile's rows)
  auto dma_one = [&](int idx, int st, bool live) {  // instruction idx of AI + BI into stage st
    if (idx < AI) {
      char* As = lds + st * STAGE_B + (8 * AI * wv_u) * 128;
      if constexpr (YMK_ABLATE == 6 || YMK_ABLATE == 8) {
        const unsigned o = ((((unsigned)tile * (unsigned)ktiles + seq) * NW + wv_u) * AI + idx) % (p.in_bytes >> 10);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(As + idx * 1024), 16, (int)(o * 1024u + lane * 16u), 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(As + idx * 1024), 16, (int)a_off[idx], a_soff, 0, 0);
      }
    } else {
      char* Bs = lds + st * STAGE_B + A_STAGE + (8 * BI * wv_u) * 128;
      if constexpr (YMK_ABLATE == 7 || YMK_ABLATE == 8) {
        const unsigned o = ((((unsigned)tile * (unsigned)ktiles + seq) * NW + wv_u) * BI + (idx - AI)) % (w_bytes >> 10);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void*)(Bs + (idx - AI) * 1024), 16, (int)(o * 1024u + lane * 16u), 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void*)(Bs + (idx - AI) * 1024), 16, (int)(live ? boff[idx - AI] : OOB_OFFSET), b_soff, 0, 0);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment rows: A row WTM wm + 32 a + li, B row WTN wn + 32 b + li; the swizzle term (row >> 1) & 7 is (li >> 1) & 7 for all
  const int swz = (li >> 1) & 7;
  constexpr int NMF = 3 * TM * TN;                  // MFMAs of one 16-k step
  constexpr int NDMA = AI + BI;                     // DMA instructions per K tile and wave
  static_assert(NDMA % 2 == 0 && NMF % (NDMA / 2) == 0, "the DMA instructions spread evenly over the two steps' MFMAs");
  constexpr int MPD = NMF / (NDMA / 2);             // MFMAs between two DMA instructions
  // one 16-k step (s = 0, 1) of the K tile in stage st, with its half of the next tile's DMA (into stage st ^ 1) threaded through
  auto compute = [&](int st, int s, bool live) {
    if constexpr (YMK_ABLATE == 2) return;
    const char* As = lds + st * STAGE_B + (WTM * wm + li) * 128;
    const char* Bs = lds + st * STAGE_B + A_STAGE + (WTN * wn + li) * 128;
    hf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      if constexpr (APL) {
        ah[a] = *reinterpret_cast<const hf16x8_t*>(As + a * 32 * 128 + (((s * 2 + lh) ^ swz) * 16));
        al[a] = *reinterpret_cast<const hf16x8_t*>(As + a * 32 * 128 + (((4 + s * 2 + lh) ^ swz) * 16));
      } else {
        const int ca = s * 4 + lh * 2;
        const f32x4 u = *reinterpret_cast<const f32x4*>(As + a * 32 * 128 + ((ca ^ swz) * 16));
        const f32x4 v = *reinterpret_cast<const f32x4*>(As + a * 32 * 128 + (((ca + 1) ^ swz) * 16));
        split8(u, v, sa, ah[a], al[a]);
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      bh[b] = *reinterpret_cast<const hf16x8_t*>(Bs + b * 32 * 128 + (((s * 2 + lh) ^ swz) * 16));
      bl[b] = *reinterpret_cast<const hf16x8_t*>(Bs + b * 32 * 128 + (((4 + s * 2 + lh) ^ swz) * 16));
    }
    // smallest terms first, term-major (per accumulator: al bh, ah bl, ah bh - the order of every fp16-split kernel)
#pragma unroll
    for (int q = 0; q < NMF; ++q) {
      if (q % MPD == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (YMK_ABLATE != 1) dma_one(s * (NDMA / 2) + q / MPD, st ^ 1, live);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int term = q / (TM * TN), a = (q % (TM * TN)) / TN, b = q % TN;
      if constexpr (YMK_ABLATE == 4) {
        asm volatile("" ::"v"(ah[a]), "v"(al[a]), "v"(bh[b]), "v"(bl[b]));
      } else {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al[a] : ah[a], term == 1 ? bl[b] : bh[b], acc[a][b], 0, 0, 0);
      }
    }
  };

  prep(0, true);
#pragma unroll
  for (int idx = 0; idx < NDMA; ++idx) dma_one(idx, 0, true);
  int st = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    // tile kt has landed once every wave's own DMAs of it have; the same barrier tells that every wave is done reading the
    // other stage (tile kt - 1), which tile kt + 1 overwrites from here on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (YMK_ABLATE != 3) __builtin_amdgcn_s_barrier();
    const bool live = kt + 1 < ktiles;
    prep(kt + 1, live);
    seq = (unsigned)kt + 1;
    compute(st, 0, live);
    compute(st, 1, live);
    st ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the zeros the last iteration's DMA wrote: the epilogue reuses the LDS)

  // ---- epilogue, EROWS rows at a time: accumulators (times 1 / sa: exact) -> LDS as [EROWS][BN + 4] fp32 -> the shared epilogue
  float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int e0 = 0; e0 < BM; e0 += EROWS) {
    __syncthreads();
    if (WTM * wm >= e0 && WTM * wm < e0 + EROWS) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = WTM * wm - e0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
            Cs[row * LDC + WTN * wn + 32 * b + li] = acc[a][b][r] * inv_sa;
          }
    }
    __syncthreads();
    epilogue_tile<EROWS, BN, NT, false, 4, OPL>(p, Cs, m0 + e0, n0, t);
  }
}

template <bool APL, bool OPL>
static void launch_dma_wide(hipStream_t s, ConvK& k, const void* wsplit, size_t w_bytes) {
  const int mt = (k.M + 255) / 256, nt = (k.Cout + 255) / 256;
  k.ntiles_n = nt;
  hipLaunchKernelGGL((conv_f16_dma_wide<4, 2, 2, 4, APL, OPL>), dim3(mt * nt), dim3(512), 0, s, k, reinterpret_cast<const uint4*>(wsplit), (unsigned)w_bytes);
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
  if (rows == 512) {  // the wide form: 256 x 256 tiles (rows names it; the caller has set ntiles_n for 256 columns)
    if (k.in_planes && k.out_planes) launch_dma_wide<true, true>(s, k, wsplit, w_bytes);
    else if (k.in_planes) launch_dma_wide<true, false>(s, k, wsplit, w_bytes);
    else if (k.out_planes) launch_dma_wide<false, true>(s, k, wsplit, w_bytes);
    else launch_dma_wide<false, false>(s, k, wsplit, w_bytes);
    return true;
  }
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
