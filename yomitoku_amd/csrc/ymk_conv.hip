// Implicit-GEMM convolution / linear layer for gfx950 (CDNA4), exact fp32 on the matrix cores.
//
// Replaces the ATen conv2d / linear calls of the reference's nets
// (models/dbnet_plus.py:33-38,56-127; models/layers/rtdetr_backbone.py; parseq_transformer.py).
//
//   out[m][co] = act( scale[co] * sum_k A[m][k] * Wp[co][k] + bias[co] + res[m][co] )
//
//   m  = (n, oh, ow) output pixel           (GEMM M, NHWC so a pixel's channels are contiguous)
//   k  = (kh, kw, c) filter tap x channel   (GEMM K, gathered on the fly - no im2col buffer)
//   co = output channel                     (GEMM N)
//
// Matrix instruction: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bit-identical to an fmaf
// chain in k order; 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak). Operand layout (guide §3):
//   A: lane l holds A[i = l&31][k = l>>5],  B: lane l holds B[k = l>>5][j = l&31],
//   D: lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
// K is consumed in chunks of 8 with the in-chunk order permuted so that each half-wave reads
// 4 consecutive k with one ds_read_b128: step s of chunk kc multiplies k = kc*8 + 4*(l>>5) + s.
//
// Block = 4, 8 or 16 waves (the wide 128 x 128 tile runs with 8: 32 x 64 per wave); block tile BM x BN, K tile 32, LDS rows padded to 36 floats
// (144 B = 9 x 16 B slots: the 16-lane groups of ds_read_b128 hit 16 distinct slots, guide G4).
// Global -> register -> LDS double buffering, one barrier per K tile.
// blockIdx is remapped so that each XCD (private L2) owns a contiguous range of tiles, n fastest,
// i.e. the blocks that re-read one A row panel run on the same L2 (guide T1, bijective form).
//
// Two kernels share the operand layout and the epilogue: conv_igemm (above) for launches that fill the
// chip, conv_splitk for grid-starved ones (K split over the waves of a block, see its comment);
// conv2d() picks per launch from the GEMM shape alone, so a given shape always takes the same path.
#include <atomic>
#include <deque>
#include <mutex>

#include "ymk_conv_kernel.h"

namespace ymk {

// PF: K tiles the global loads run ahead of the MFMAs (1: the tile consumed next; 2: one more - two register sets, so a
// load has two compute phases to come back from MALL / HBM before its ds_write waits for it)
// OPT bit 0: accumulators go straight to global memory (every wave on its own: no LDS staging, no block barrier in the
//            epilogue; a half-wave writes 128 contiguous bytes of an output row per store) - EPI_STORE launches only
// OPT bit 1: K-tile rows of 32 floats with the 16-byte slots XOR-swizzled over row pairs instead of rows padded to 36:
//            the 128 x 64 tile then takes 48 KB of LDS and THREE blocks share a CU
template <int BM, int BN, int WM, int WN, int MODE, int PF = 1, int OPT = 0>
__global__ __launch_bounds__(64 * WM * WN, ((OPT & 2) ? blocks_per_cu(BM, BN, OPT) : 2 * (BM + BN) * LDK * 4 <= 80 * 1024 ? 2 : 1) * WM * WN / 4) void conv_igemm(ConvK p) {
  static_assert(PF == 1 || PF == 2, "prefetch distance 1 or 2");
  constexpr int LDR = lds_row(OPT);
  static_assert(!(OPT & 2) || (64 * WM * WN / 8) % 16 == 0, "swizzle: staging passes must keep (row >> 1) & 7");
  constexpr int NT = 64 * WM * WN;             // 4, 8 or 16 waves
  static_assert(NT == 256 || NT == 512 || NT == 1024, "4, 8 or 16 waves per block");
  constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
  constexpr int RPP = NT / 8;                  // rows staged per pass (8 threads x 16 B per 32-float row)
  constexpr int APASS = BM / RPP, BPASS = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must divide by the staging pass");
  constexpr int STAGE = (BM + BN) * LDR;
  constexpr int LDC = BN + 4;  // epilogue staging row (floats)
  constexpr int EROWS = (2 * STAGE / LDC) / 32 * 32 < BM ? (2 * STAGE / LDC) / 32 * 32 : BM;  // rows per epilogue pass
  static_assert(EROWS >= WTM && EROWS % WTM == 0, "an epilogue pass must hold whole wave tiles");
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int t = threadIdx.x;
  // XCD-aware bijective remap of the block id (block b runs on XCD b % 8)
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / p.ntiles_n, tile_n = tile - tile_m * p.ntiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (!tile_needed<BM, NT>(p, m0, t)) return;

  // ---- staging coordinates: thread loads 16 B (4 k) of row (t>>3)+32*i
  const int colq = t & 7, rowb = t >> 3;
  int pixb[APASS], ih0[APASS], iw0[APASS];
  // 1x1, stride 1, no padding (every nn.Linear and most bottleneck convs): input pixel = output pixel = m, no divisions
  const bool pointwise = (p.fast & 1) && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.stride_w == 1 && p.pad == 0;
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = m0 + rowb + RPP * i;
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
      ih0[i] = -(1 << 20);  // fails every bounds test
      iw0[i] = 0;
    }
  }
  // weight rows of this thread (the panel is zero padded to a multiple of 128 rows)
  const float* wrow = p.w + (size_t)(n0 + rowb) * p.Kpad + colq * 4;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  f32x4 ra[PF][APASS], rb[PF][BPASS];

  // K-tile cursor (wave-uniform, kept in scalar registers): tap (kh, kw) and channel tile cc
  int cur_kh = 0, cur_kw = 0, cur_cc = 0;
  // byte offset of this thread's 16 B inside each of its rows for the current TAP, or OOB_OFFSET when the tap falls
  // outside the image / the row is a tail row: recomputed once per tap, not once per K tile - the channel tile within
  // a tap only moves the buffer instruction's scalar offset (the range check ignores soffset, so masked lanes stay
  // out of range and valid lanes stay inside the view)
  unsigned voff[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) voff[i] = OOB_OFFSET;

  // branch-free gather of one K tile: out-of-image / padded taps read a safe address and are zeroed
  auto load_tile = [&](int kt, int set) {
    if (MODE == 0) {
      if (cur_cc == 0) {  // wave-uniform: a new filter tap
        const int dh = cur_kh * p.dil, dw = cur_kw * p.dil;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
          const int ih = ih0[i] + dh, iw = iw0[i] + dw;
          const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          const unsigned off = ((unsigned)(pixb[i] + ih * p.W + iw) * (unsigned)p.in_ld + (unsigned)(colq * 4)) * 4u;
          voff[i] = ok ? off : OOB_OFFSET;
        }
      }
      const bool chan_ok = cur_cc * 32 + colq * 4 < p.C;  // the last channel tile of a C % 32 != 0 layer is partial
      const int soff = cur_cc * 128;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(chan_ok ? voff[i] : OOB_OFFSET), soff, 0);
        ra[set][i] = __builtin_bit_cast(f32x4, v);
      }
      if (++cur_cc == p.ctiles) {
        cur_cc = 0;
        if (++cur_kw == p.KW) {
          cur_kw = 0;
          ++cur_kh;
        }
      }
    } else {
      const int tap = kt * 8 + colq;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
      const int dh = kh * p.dil, dw = kw * p.dil;
      const bool tapok = tap < p.KH * p.KW;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const int ih = ih0[i] + dh, iw = iw0[i] + dw;
        const bool ok = tapok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const unsigned off = ((unsigned)(pixb[i] + ih * p.W + iw) * (unsigned)p.in_ld) * 4u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok ? off : OOB_OFFSET), 0, 0);
        ra[set][i] = __builtin_bit_cast(f32x4, v);
      }
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j)
      rb[set][j] = *reinterpret_cast<const f32x4*>(wrow + (size_t)(RPP * j) * p.Kpad + kt * 32);
  };

  const int st_off = lds_slot<OPT>(rowb, colq);  // RPP is a multiple of 16 rows: the passes only add whole lines
  auto store_tile = [&](int buf, int set) {
    float* As = lds + buf * STAGE + st_off;
    float* Bs = As + BM * LDR;
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      *reinterpret_cast<f32x4*>(As + RPP * i * LDR) = ra[set][i];
#pragma unroll
    for (int j = 0; j < BPASS; ++j)
      *reinterpret_cast<f32x4*>(Bs + RPP * j * LDR) = rb[set][j];
  };

  // ---- MFMA coordinates
  const int wv = t >> 6, lane = t & 63;
  const int wm = wv / WN, wn = wv - wm * WN;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int ktiles = p.Kpad >> 5;

  int a_off[4], b_off[4];  // chunk kc of this lane's rows (tile rows step by 32: whole lines, the swizzle term stays)
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    a_off[kc] = lds_slot<OPT>(wm * WTM + li, 2 * kc + lh);
    b_off[kc] = BM * LDR + lds_slot<OPT>(wn * WTN + li, 2 * kc + lh);
  }
  auto compute = [&](int buf) {
    const float* Ts = lds + buf * STAGE;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        fa[a] = *reinterpret_cast<const f32x4*>(Ts + a_off[kc] + a * 32 * LDR);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        fb[b] = *reinterpret_cast<const f32x4*>(Ts + b_off[kc] + b * 32 * LDR);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
        }
    }
  };

  load_tile(0, 0);
  store_tile(0, 0);
  if (PF == 2 && ktiles > 1) load_tile(1, 1);
  __syncthreads();

  using Epi = EpiRows<EROWS, BN, NT>;
  constexpr bool CAN_PRE = PF == 1 && EROWS == BM && Epi::NR <= 8 && NT <= 512;  // the 16-wave tiles have no registers to spare
  const bool pre_res = CAN_PRE && (p.fast & 2) && p.res != nullptr && p.vec && p.epi == EPI_STORE;  // block-uniform
  float4 rpre[CAN_PRE ? Epi::NR : 1];
  if (PF == 1) {
    for (int kt = 0; kt < ktiles; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < ktiles) load_tile(kt + 1, 0);
      if (CAN_PRE && kt + 1 == ktiles && pre_res) prefetch_residual<EROWS, BN, NT>(p, m0, n0, t, rpre);
      compute(buf);
      if (kt + 1 < ktiles) store_tile(buf ^ 1, 0);
      if (!(OPT & 1) || kt + 1 < ktiles) __syncthreads();  // the direct epilogue does not reuse the LDS
    }
  } else {
    // tile t travels in register set t & 1: loaded at the top of step t - 2, written to LDS[t & 1] at the end of step t - 1
    for (int kt = 0; kt < ktiles; kt += 2) {
      if (kt + 2 < ktiles) load_tile(kt + 2, 0);
      compute(0);
      if (kt + 1 < ktiles) store_tile(1, PF - 1);
      __syncthreads();
      if (kt + 1 >= ktiles) break;
      if (kt + 3 < ktiles) load_tile(kt + 3, PF - 1);
      compute(1);
      if (kt + 2 < ktiles) store_tile(0, 0);
      __syncthreads();
    }
  }

  if (OPT & 1) {
    epilogue_direct<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, li, lh);
    return;
  }
  // ---- epilogue: accumulators -> LDS tile [EROWS][LDC] -> 16 B per lane, full rows coalesced; a tile taller than the
  // staging LDS goes in BM / EROWS passes (the waves owning the rows of a pass write, every thread stores).
  // D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cs = lds;
#pragma unroll
  for (int e0 = 0; e0 < BM; e0 += EROWS) {
    if (e0 > 0) __syncthreads();
    if (wm * WTM >= e0 && wm * WTM < e0 + EROWS) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm * WTM - e0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            Cs[row * LDC + wn * WTN + b * 32 + li] = acc[a][b][r];
          }
    }
    __syncthreads();
    if (CAN_PRE && pre_res) epilogue_tile<EROWS, BN, NT, true>(p, Cs, m0 + e0, n0, t, rpre);
    else epilogue_tile<EROWS, BN, NT>(p, Cs, m0 + e0, n0, t);
  }
}

// ---- grid-starved shapes (batch-1 RT-DETR stages, decoder linears, the PARSeq head): split K over the
// waves of a block.  conv_igemm gives such a launch fewer than one 4-wave block per CU, and each wave
// then walks the whole K extent with one global->LDS round trip per K tile (measured ~0.9 us per tile,
// twice its MFMA time).  Here a block owns a (32 TM) x (32 TN) tile and wave w takes K tiles
// w, w + NW, ...: NW x more waves in flight, and no LDS in the main loop at all - with the in-chunk K
// permutation each lane's MFMA operands for 4 consecutive k are one 16 B global load (A: its pixel row,
// B: its weight row), PF K tiles deep in registers.  The NW partial tiles are summed pairwise through
// LDS in a fixed order (deterministic), then the usual epilogue runs on the block tile.
template <int TM, int TN, int NW, int PF>
__global__ __launch_bounds__(64 * NW) void conv_splitk(ConvK p) {
  constexpr int BM = 32 * TM, BN = 32 * TN, NT = 64 * NW;
  constexpr int LDC = BN + 4;
  constexpr int PART = TM * TN * 16 * 64;  // floats of one wave's accumulators
  constexpr int RED = (NW / 2) * PART > BM * LDC ? (NW / 2) * PART : BM * LDC;
  __shared__ __attribute__((aligned(16))) float red[RED];

  const int t = threadIdx.x;
  const int lane = t & 63, li = lane & 31, lh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / p.ntiles_n, tile_n = tile - tile_m * p.ntiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (!tile_needed<BM, NT>(p, m0, t)) return;

  int pixb[TM], ih0[TM], iw0[TM];
  const bool pointwise = (p.fast & 1) && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.stride_w == 1 && p.pad == 0;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int m = m0 + 32 * a + li;
    if (m < p.M && pointwise) {
      pixb[a] = m;
      ih0[a] = 0;
      iw0[a] = 0;
    } else if (m < p.M) {
      const int ohw = p.OH * p.OW;
      const int n = m / ohw, rem = m - n * ohw;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      pixb[a] = n * p.H * p.W;
      ih0[a] = oh * p.stride - p.pad;
      iw0[a] = ow * p.stride_w - p.pad;
    } else {
      pixb[a] = 0;
      ih0[a] = -(1 << 20);
      iw0[a] = 0;
    }
  }
  const float* wrow = p.w + (size_t)(n0 + li) * p.Kpad + lh * 4;  // panel rows are padded to 128: always in range
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);

  const int ktiles = p.Kpad >> 5;
  const int nt = wv < ktiles ? (ktiles - wv + NW - 1) / NW : 0;  // K tiles of this wave
  f32x4 sa[PF][TM][4], sb[PF][TN][4];

  // K tile j of this wave -> registers of stage s; j past the end re-reads the last tile (never consumed)
  auto load = [&](int s, int j) {
    int kt = wv + NW * j;
    kt = kt < ktiles ? kt : ktiles - 1;
    const int tap = kt / p.ctiles, cc = kt - tap * p.ctiles;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int dh = kh * p.dil, dw = kw * p.dil;
    const int c0 = cc * 32 + lh * 4;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int ih = ih0[a] + dh, iw = iw0[a] + dw;
      const bool ok = (unsigned)ih < (unsigned)p.H & (unsigned)iw < (unsigned)p.W;
      // padding taps / tail rows: a row base past the descriptor's range, so all four loads return zeros
      const unsigned row = ok ? (unsigned)(pixb[a] + ih * p.W + iw) * (unsigned)p.in_ld * 4u : SPLITK_OOB_ROW;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        // chunks past C (C % 32 != 0) re-read the last real chunk: finite data against zero-padded weights
        const int c = min(c0 + kc * 8, p.C - 4);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(row + (unsigned)c * 4u), 0, 0);
        sa[s][a][kc] = __builtin_bit_cast(f32x4, v);
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        sb[s][b][kc] = *reinterpret_cast<const f32x4*>(wrow + (size_t)(32 * b) * p.Kpad + kt * 32 + kc * 8);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto compute = [&](int s) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[s][a][kc].x, sb[s][b][kc].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[s][a][kc].y, sb[s][b][kc].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[s][a][kc].z, sb[s][b][kc].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[s][a][kc].w, sb[s][b][kc].w, acc[a][b], 0, 0, 0);
        }
  };

#pragma unroll
  for (int s = 0; s < PF; ++s) load(s, s);
  int j0 = 0;
  for (; j0 + PF <= nt; j0 += PF) {  // full groups: one basic block, so the load waits stay precise
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      compute(s);
      load(s, j0 + s + PF);
    }
  }
  {
    const int rem = nt - j0;
#pragma unroll
    for (int s = 0; s < PF - 1; ++s)
      if (s < rem) compute(s);
  }

  // ---- sum the NW partial tiles: upper half of the live waves hands its registers to the lower half
#pragma unroll
  for (int half = NW / 2; half >= 1; half >>= 1) {
    if (wv >= half && wv < 2 * half) {
      float* dst = red + (wv - half) * PART + lane;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((a * TN + b) * 16 + r) * 64] = acc[a][b][r];
    }
    __syncthreads();
    if (wv < half) {
      const float* src = red + wv * PART + lane;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] += src[((a * TN + b) * 16 + r) * 64];
    }
    __syncthreads();
  }
  if (wv == 0) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          red[row * LDC + b * 32 + li] = acc[a][b][r];
        }
  }
  __syncthreads();
  epilogue_tile<BM, BN, NT>(p, red, m0, n0, t);
}

// ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream
struct ProfState {
  std::mutex mu;  // launches may come from several host threads (page workers, recogniser lanes)
  bool on = false;
  std::deque<std::pair<hipEvent_t, hipEvent_t>> ev;  // deque: a span's pointer stays valid while others are added
  size_t used = 0;
  double flop = 0.0;
  double bytes = 0.0;  // algorithmic HBM bytes: input view, weights, output and residual, each touched once
  std::vector<std::string> desc;
  std::vector<double> lflop, lbytes, lms, lprod;  // per launch (lms filled by prof_end): what ymk_prof_launch_table hands out
};
static ProfState g_prof;

// conv_igemm launches with fewer blocks than this go to the split-K kernel (1.5 blocks per CU)
constexpr int SPLITK_MAX_GRID = 384;
// test / measurement knobs (ymk_debug_option): process-wide, read per launch, never set on the product path
static std::atomic<int> g_splitk_force{-1};  // >= 0: that split-K candidate for every eligible launch (kernel tests)
static std::atomic<int> g_no_splitk{0};      // 1: every launch through conv_igemm (A/B profiling, kernel tests)
static std::atomic<int> g_conv_fast{27};     // ConvK::fast
static std::atomic<int> g_conv_variant{0};   // conv_igemm schedule selector for A/B runs, see launch_wide()
static std::atomic<int> g_prof_dump{0};      // 1: ymk_prof_end prints one line per launch to stderr
static std::atomic<int> g_conv_split{-1};     // -1: unset (a model runs its own default); 0: exact fp32 MFMA everywhere; 2 / 3 / 16: split operands (ymk_conv_split.hip)
static std::atomic<int> g_gemm_row_limit{0};  // > 0: gemm() cuts its rows into chunks of at most this many (tests of the 4 GiB chunking)
static int gemm_row_limit() { return g_gemm_row_limit.load(std::memory_order_relaxed); }
static int splitk_forced() { return g_splitk_force.load(std::memory_order_relaxed); }
static bool no_splitk() { return g_no_splitk.load(std::memory_order_relaxed) != 0; }
static thread_local int t_conv_split = -1;  // >= 0: set by a ConvSplitScope on this thread (a model's own parameter)
static thread_local SplitCtx* t_split_ctx = nullptr;
static thread_local int t_split_default = 0;
ConvSplitScope::ConvSplitScope(int split, SplitCtx* ctx, int dflt) : prev_(t_conv_split), prev_dflt_(t_split_default), prev_ctx_(t_split_ctx) {
  t_conv_split = split;
  if (ctx) t_split_ctx = ctx;
  if (dflt >= 0) t_split_default = dflt;
}
ConvSplitScope::~ConvSplitScope() {
  t_conv_split = prev_;
  t_split_default = prev_dflt_;
  t_split_ctx = prev_ctx_;
}

int conv_effective_split() {
  int split = t_conv_split;
  if (split < 0) split = g_conv_split.load(std::memory_order_relaxed);
  if (split < 0) split = t_split_default;
  return split;
}

bool conv_debug_option(const std::string& key, int value) {
  if (key == "splitk_force") g_splitk_force = value;
  else if (key == "no_splitk") g_no_splitk = value;
  else if (key == "conv_variant") g_conv_variant = value;
  else if (key == "conv_fast") g_conv_fast = value;
  else if (key == "prof_dump") g_prof_dump = value;
  else if (key == "conv_split") g_conv_split = value;
  else if (key == "gemm_row_limit") g_gemm_row_limit = value;
  else return false;
  return true;
}

void prof_begin() {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  g_prof.on = true;
  g_prof.used = 0;
  g_prof.flop = 0.0;
  g_prof.bytes = 0.0;
}
double prof_bytes() {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  return g_prof.bytes;
}
void prof_end(double* ms, double* flop, int64_t* launches) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  double total = 0.0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    YMK_HIP(hipEventSynchronize(g_prof.ev[i].second));
    float t = 0.f;
    YMK_HIP(hipEventElapsedTime(&t, g_prof.ev[i].first, g_prof.ev[i].second));
    total += t;
    g_prof.lms[i] = t;
    if (g_prof_dump.load(std::memory_order_relaxed))
      fprintf(stderr, "[ymk-prof] %3zu %s  %8.1f us  %6.1f TFLOP/s  %7.1f MB\n", i, g_prof.desc[i].c_str(), t * 1e3,
              g_prof.lflop[i] / (t * 1e-3) / 1e12, g_prof.lbytes[i] / 1e6);
  }
  *ms = total;
  *flop = g_prof.flop;
  *launches = (int64_t)g_prof.used;
  g_prof.on = false;
}
// the launches of the span prof_end closed last, one row each; returns how many there are (rows beyond `capacity` are not written)
int64_t prof_launch_table(double* ms, double* flop, double* bytes, double* products, int64_t capacity) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  const int64_t n = g_prof.on ? 0 : (int64_t)g_prof.used;
  for (int64_t i = 0; i < n && i < capacity; ++i) {
    ms[i] = g_prof.lms[i];
    flop[i] = g_prof.lflop[i];
    bytes[i] = g_prof.lbytes[i];
    products[i] = g_prof.lprod[i];
  }
  return n;
}

// open a timed span for one launch when profiling is on (returns the event pair to close it with)
std::pair<hipEvent_t, hipEvent_t>* conv_prof_open(hipStream_t s, const ConvK& k, int BM, int BN, int grid, int ksplit) {
  std::pair<hipEvent_t, hipEvent_t>* e = nullptr;
  if (g_prof.on) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    if (g_prof.used == g_prof.ev.size()) {
      std::pair<hipEvent_t, hipEvent_t> n;
      YMK_HIP(hipEventCreate(&n.first));
      YMK_HIP(hipEventCreate(&n.second));
      g_prof.ev.push_back(n);
    }
    e = &g_prof.ev[g_prof.used++];
    const int creal = k.mode == 0 ? k.C : 3;
    const double fl = 2.0 * (double)k.M * (double)k.Cout * (double)(k.KH * k.KW * creal);
    g_prof.flop += fl;
    const double images = (double)k.M / ((double)k.OH * k.OW);
    const double outs = (double)k.M * k.Cout;
    const double by = 4.0 * (images * k.H * k.W * creal + (double)k.Cout * k.KH * k.KW * creal + outs * (k.res ? 2.0 : 1.0));
    g_prof.bytes += by;
    char buf[176];
    snprintf(buf, sizeof buf, "M=%7d Cin=%4d Cout=%4d k=%dx%d s=%d d=%d res=%d tile=%dx%d ksplit=%d grid=%d", k.M, k.C, k.Cout,
             k.KH, k.KW, k.stride, k.dil, k.res ? 1 : 0, BM, BN, ksplit, grid);
    if (g_prof.desc.size() < g_prof.used) {
      g_prof.desc.resize(g_prof.used);
      g_prof.lflop.resize(g_prof.used);
      g_prof.lbytes.resize(g_prof.used);
      g_prof.lms.resize(g_prof.used);
      g_prof.lprod.resize(g_prof.used);
    }
    g_prof.desc[g_prof.used - 1] = buf;
    g_prof.lflop[g_prof.used - 1] = fl;
    g_prof.lbytes[g_prof.used - 1] = by;
    g_prof.lms[g_prof.used - 1] = 0.0;
    // MFMA products behind one fp32-grade product: the split kernels tag their spans through `ksplit` (160 / 161: two fp16
    // planes, 20 / 30: two / three bf16 planes); everything else is the exact fp32 MFMA
    g_prof.lprod[g_prof.used - 1] = ksplit >= 160 ? 3.0 : (ksplit == 30 ? 6.0 : (ksplit == 20 ? 3.0 : 0.0));
    YMK_HIP(hipEventRecord(e->first, s));
  }
  return e;
}

// the same for a launch that is not one convolution (the fused ViT MLP): description, algorithmic FLOPs and bytes, and the
// MFMA products behind one fp32-grade product, as given
std::pair<hipEvent_t, hipEvent_t>* conv_prof_open_raw(hipStream_t s, const char* desc, double flops, double bytes, double products) {
  if (!g_prof.on) return nullptr;
  std::lock_guard<std::mutex> lock(g_prof.mu);
  if (g_prof.used == g_prof.ev.size()) {
    std::pair<hipEvent_t, hipEvent_t> n;
    YMK_HIP(hipEventCreate(&n.first));
    YMK_HIP(hipEventCreate(&n.second));
    g_prof.ev.push_back(n);
  }
  auto* e = &g_prof.ev[g_prof.used++];
  g_prof.flop += flops;
  g_prof.bytes += bytes;
  if (g_prof.desc.size() < g_prof.used) {
    g_prof.desc.resize(g_prof.used);
    g_prof.lflop.resize(g_prof.used);
    g_prof.lbytes.resize(g_prof.used);
    g_prof.lms.resize(g_prof.used);
    g_prof.lprod.resize(g_prof.used);
  }
  g_prof.desc[g_prof.used - 1] = desc;
  g_prof.lflop[g_prof.used - 1] = flops;
  g_prof.lbytes[g_prof.used - 1] = bytes;
  g_prof.lms[g_prof.used - 1] = 0.0;
  g_prof.lprod[g_prof.used - 1] = products;
  YMK_HIP(hipEventRecord(e->first, s));
  return e;
}

template <int TM, int TN, int NW, int PF>
static void launch_splitk(hipStream_t s, ConvK& k) {
  const int mt = (k.M + 32 * TM - 1) / (32 * TM), nt = (k.Cout + 32 * TN - 1) / (32 * TN);
  k.ntiles_n = nt;
  auto* e = conv_prof_open(s, k, 32 * TM, 32 * TN, mt * nt, NW);
  hipLaunchKernelGGL((conv_splitk<TM, TN, NW, PF>), dim3(mt * nt), dim3(64 * NW), 0, s, k);
  if (e) YMK_HIP(hipEventRecord(e->second, s));
}

// Pick the split-K shape for a launch conv_igemm cannot spread over the chip.  Cost model: a wave issues
// one 32x32x2 MFMA per 64 cycles, so its time is (its K tiles) x TM x TN x 1024 cycles plus a fixed
// prologue / reduction / epilogue cost, and the 1024 SIMDs take ceil(waves / 1024) waves each.
static bool try_splitk(hipStream_t s, ConvK& k) {
  struct Cand {
    int tm, tn, nw;
    void (*fn)(hipStream_t, ConvK&);
  };
  static const Cand cands[] = {
      {2, 2, 4, launch_splitk<2, 2, 4, 2>}, {2, 1, 4, launch_splitk<2, 1, 4, 3>}, {2, 1, 8, launch_splitk<2, 1, 8, 3>},
      {1, 1, 4, launch_splitk<1, 1, 4, 4>}, {1, 1, 8, launch_splitk<1, 1, 8, 4>},
  };
  if (k.in_bytes >= SPLITK_OOB_ROW) return false;  // masked rows must stay out of the descriptor's range
  const int ktiles = k.Kpad >> 5;
  const Cand* best = nullptr;
  double best_cost = 0.0;
  for (const Cand& c : cands) {
    const long blocks = (long)((k.M + 32 * c.tm - 1) / (32 * c.tm)) * ((k.Cout + 32 * c.tn - 1) / (32 * c.tn));
    const long waves = blocks * c.nw;
    const double per_wave = (double)((ktiles + c.nw - 1) / c.nw) * c.tm * c.tn * 1024.0 + 3000.0 + (c.nw == 8 ? 1800.0 : 1200.0);
    const double cost = (double)((waves + 1023) / 1024) * per_wave;
    if (!best || cost < best_cost * 0.97) {  // candidates are listed widest tile first: ties keep the fewer loads
      best = &c;
      best_cost = cost;
    }
  }
  const int forced = splitk_forced();
  if (forced >= 0) best = &cands[forced % (int)(sizeof(cands) / sizeof(cands[0]))];
  best->fn(s, k);
  return true;
}

template <int BM, int BN, int WM, int WN, int MODE = 0, int PF = 1>
static void launch(hipStream_t s, ConvK& k) {
  const int mt = (k.M + BM - 1) / BM, nt = (k.Cout + BN - 1) / BN;
  k.ntiles_n = nt;
  if (MODE == 0 && (mt * nt < SPLITK_MAX_GRID || splitk_forced() >= 0) && !no_splitk() && try_splitk(s, k)) return;
  // ConvK::fast bits 2 / 4: direct epilogue (plain stores only; everywhere / ragged Cout); bit 3: swizzled K tiles.
  // (A persistent tile loop over the swizzled tile - next tile's first loads issued before this tile's epilogue - was
  // measured in round 4 and removed: bit-identical, 5-30 % slower on 15 of 17 shapes, profiles/r04_conv_sweep_persistent_tile_loop.txt)
  constexpr bool HAS_OPT = MODE == 0 && PF == 1;
  constexpr bool HAS_SWZ = HAS_OPT && BM == 128 && BN == 64 && WM * WN == 8;
  const bool direct = HAS_OPT && k.epi == EPI_STORE && ((k.fast & 4) || ((k.fast & 16) && !k.vec));
  const bool swz = HAS_SWZ && (k.fast & 8);
  auto* e = conv_prof_open(s, k, BM, BN, mt * nt, 1);
  const dim3 grid(mt * nt), block(64 * WM * WN);
  if constexpr (HAS_SWZ) {
    if (swz && direct) hipLaunchKernelGGL((conv_igemm<BM, BN, WM, WN, MODE, PF, 3>), grid, block, 0, s, k);
    else if (swz) hipLaunchKernelGGL((conv_igemm<BM, BN, WM, WN, MODE, PF, 2>), grid, block, 0, s, k);
  }
  if constexpr (HAS_OPT) {
    if (direct && !swz) hipLaunchKernelGGL((conv_igemm<BM, BN, WM, WN, MODE, PF, 1>), grid, block, 0, s, k);
  }
  if (!direct && !swz) hipLaunchKernelGGL((conv_igemm<BM, BN, WM, WN, MODE, PF>), grid, block, 0, s, k);
  if (e) YMK_HIP(hipEventRecord(e->second, s));
}

// Round 3: the 128 x 64 tile with swizzled K-tile rows (48 KB of LDS: three blocks per CU, six waves per SIMD) replaced the
// 16-wave wide tile for Kpad <= 512 (conv2d below); accumulators of ragged-Cout launches go straight to global memory.
// Schedules.  Measured on MI355X (tools/conv_sweep.py, random operands; profiles/r02_conv_sweep*.txt): the 128 x 128
// tile with EIGHT waves (32 x 64 wave tiles, 4 waves per SIMD at 2 blocks per CU) beats the four-wave form of round 1
// everywhere - +3..8 % on the K >= 512 layers of DBNet, +10 % on the K = 192 GEMMs of PARSeq, +30 % on the memory-bound
// 64 -> 256 expands; SIXTEEN waves add a few per cent more where K is short (<= 512: the loop is mostly prologue and
// epilogue) and lose a little on the long-K layers; the 128 x 64 tile likewise gains from eight waves.  Interleaving the
// MFMA order across accumulators, s_setprio around the MFMA cluster and a 256 x 128 tile changed nothing or lost.
// conv_variant (ymk_debug_option) keeps the alternatives reachable for A/B runs:
//   wide path (Cout > 64):  0 = by K (default)   1 = 4 waves (round 1)   2 = 16 waves   4 = 8 waves   5 = 8 waves, loads 2 K tiles ahead   8 = 128 x 64 tiles
//   narrow path (128 x 64): 0 = 8 waves (default)   3 = 4 waves (round 1)   6 = 8 waves, loads 2 K tiles ahead
static void launch_wide(hipStream_t s, ConvK& k) {
  switch (g_conv_variant.load(std::memory_order_relaxed)) {
    case 1: launch<128, 128, 2, 2>(s, k); break;
    case 2: launch<128, 128, 4, 4>(s, k); break;
    case 4: launch<128, 128, 4, 2>(s, k); break;
    case 5: launch<128, 128, 4, 2, 0, 2>(s, k); break;
    case 8: launch<128, 64, 4, 2>(s, k); break;  // A/B runs: 64-wide tiles whatever Cout
    default:
      if (k.Kpad <= 512) launch<128, 128, 4, 4>(s, k);
      else launch<128, 128, 4, 2>(s, k);
      break;
  }
}
static void launch_n64(hipStream_t s, ConvK& k) {
  switch (g_conv_variant.load(std::memory_order_relaxed)) {
    case 3: launch<128, 64, 2, 2>(s, k); break;
    case 6: launch<128, 64, 4, 2, 0, 2>(s, k); break;
    default: launch<128, 64, 4, 2>(s, k); break;
  }
}

// conv2d proper.  With a LayerNorm to fuse (a.ln_g) only the fp16-split path is tried and the result says whether it took the
// launch; without one every launch is taken by some kernel (true).
static bool conv2d_impl(hipStream_t s, const Tensor& in, const ConvW& w, const ConvArgs& a, const Tensor& out) {
  ConvK k{};
  k.ln_g = a.ln_g;
  k.ln_b = a.ln_b;
  k.ln_eps = a.ln_eps;
  k.in_planes = in.planes ? 1 : 0;
  k.out_planes = a.out_planes ? 1 : 0;
  k.pl_a = w.pl_a;
  k.pl_b = w.pl_b;
  if (a.out_planes)
    YMK_CHECK(a.epi == EPI_STORE && a.res == nullptr && out.ld == w.cout && w.cout % 32 == 0 && out.amax != nullptr,
              "conv: fp16 planes are written by plain stores into a whole tensor of 32-channel slices that has a record");
  if (in.planes) YMK_CHECK(in.ld == in.c && in.c % 32 == 0 && in.amax != nullptr, "conv: malformed tensor of fp16 planes");
  k.in = in.p;
  k.w = w.w;
  k.scale = w.scale;
  k.bias = w.bias;
  k.res = a.res ? a.res->p : nullptr;
  k.res_ld = a.res ? a.res->ld : 0;
  k.res_post = a.res_post ? 1 : 0;
  k.out = out.p;
  k.H = in.h;
  k.W = in.w;
  k.C = in.c;
  k.in_ld = in.ld;
  k.KH = w.kh;
  k.KW = w.kw;
  k.stride = a.stride;
  k.stride_w = a.stride_w > 0 ? a.stride_w : a.stride;
  k.pad = a.pad;
  k.dil = a.dil;
  k.OH = conv_out_dim(in.h, w.kh, a.stride, a.pad, a.dil);
  k.OW = conv_out_dim(in.w, w.kw, k.stride_w, a.pad, a.dil);
  k.Cout = w.cout;
  k.out_ld = out.ld;
  k.Kpad = w.kpad;
  k.ctiles = w.ctiles;
  k.mode = w.mode;
  k.M = in.n * k.OH * k.OW;
  k.act = a.act;
  k.epi = a.epi;
  k.row_group = a.row_group;
  k.group_open = a.group_open;
  k.amax = in.amax ? in.amax : a.amax_in;
  k.amax_out = a.epi == EPI_ROWMAX ? nullptr : (out.amax ? out.amax : a.amax_out);
  YMK_CHECK((a.row_group == nullptr) == (a.group_open == nullptr), "conv: row_group and group_open come together");
  YMK_CHECK(w.w != nullptr, "conv weight not packed");
  if (w.mode == 0) {
    YMK_CHECK(in.c == w.cin, "conv: input channels " + std::to_string(in.c) + " != weight cin " + std::to_string(w.cin));
    YMK_CHECK(in.c % 4 == 0 && in.ld % 4 == 0, "conv: channels/ld must be multiples of 4");
  } else {
    YMK_CHECK(in.c == 4 && in.ld == 4, "conv tap4 mode wants a 4-channel packed input");
  }
  YMK_CHECK(((uintptr_t)in.p & 15) == 0, "conv: input not 16B aligned");
  if (a.epi == EPI_ROWMAX) {
    YMK_CHECK(a.res == nullptr && a.act == ACT_NONE && w.mode == 0, "row-max epilogue: plain linear layers only");
    YMK_CHECK(out.ld == 2 * ((w.cout + ROWMAX_TILE_N - 1) / ROWMAX_TILE_N), "row-max epilogue: out must be [M][2 * ceil(Cout / 64)]");
  } else if (a.epi == EPI_STORE) {
    YMK_CHECK(out.n == in.n && out.h == k.OH && out.w == k.OW && out.c == w.cout, "conv: bad output shape");
    if (a.res && a.res->ld != 0)  // ld == 0: one row broadcast over every output pixel
      YMK_CHECK(a.res->n == out.n && a.res->h == out.h && a.res->w == out.w && a.res->c == out.c, "conv: bad residual shape");
  } else {
    YMK_CHECK(w.kh == 1 && w.kw == 1 && a.stride == 1 && a.pad == 0, "deconv epilogue wants a 1x1 panel");
    YMK_CHECK(out.n == in.n && out.h == 2 * in.h && out.w == 2 * in.w && out.c * 4 == w.cout, "deconv: bad output shape");
  }
  if (k.M == 0) return true;
  {
    const size_t ib = ((in.pixels() - 1) * (size_t)in.ld + in.c) * sizeof(float);
    if (a.ln_g != nullptr && ib >= (size_t)OOB_OFFSET) return false;
    YMK_CHECK(ib < (size_t)OOB_OFFSET, "conv input view must stay below 4 GiB (split the batch)");
    k.in_bytes = (unsigned)ib;
  }
  {
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const int cq = a.epi == EPI_DECONV2X2 ? w.cout / 4 : w.cout;
    k.vec = (cq % 4 == 0) && (out.ld % 4 == 0) && al16(out.p) && (!w.scale || al16(w.scale)) &&
            (!w.bias || al16(w.bias)) && (!a.res || (a.res->ld % 4 == 0 && al16(a.res->p)));
    if (a.epi == EPI_DECONV2X2) YMK_CHECK(k.vec, "deconv epilogue needs 16 B aligned channels");
  }
  k.fast = g_conv_fast.load(std::memory_order_relaxed);

  {  // split operands (fp16 / bf16 planes) with fp32 accumulation for the launches that fill the chip - the models' default
    int split = t_conv_split;
    if (split < 0) split = g_conv_split.load(std::memory_order_relaxed);
    if (split < 0) split = t_split_default;
    if (split != 0 && t_split_ctx != nullptr && conv2d_split(s, k, w, split, t_split_ctx)) return true;
    if (a.ln_g != nullptr) return false;
    YMK_CHECK(!in.planes && !a.out_planes, "conv: fp16 planes outside the fp16-split path (conv_planes_pair_ok decides before the launch)");
  }
  if (a.epi == EPI_ROWMAX) {  // fixed 64-column tiles (the caller sized the partial table for them), never split-K
    k.vec = 0;
    const int mt = (k.M + 63) / 64, nt = (k.Cout + ROWMAX_TILE_N - 1) / ROWMAX_TILE_N;
    k.ntiles_n = nt;
    auto* e = conv_prof_open(s, k, 64, 64, mt * nt, 1);
    hipLaunchKernelGGL((conv_igemm<64, 64, 2, 2, 0, 1>), dim3(mt * nt), dim3(256), 0, s, k);
    if (e) YMK_HIP(hipEventRecord(e->second, s));
    YMK_HIP(hipGetLastError());
    return true;
  }
  // tile selection: wide tiles when there is enough work to fill 256 CUs x 2 blocks
  const long blocks128 = (long)((k.M + 127) / 128) * ((w.cout + 127) / 128);
  if (w.mode == 0 && g_conv_variant.load(std::memory_order_relaxed) == 7) {  // A/B runs: 64 x 64 tiles for any shape
    launch<64, 64, 2, 2>(s, k);
  } else if (w.mode == 1) {  // 4-channel stems: Cout is 32 or 64
    if (w.cout <= 32) launch<128, 32, 4, 1, 1>(s, k);
    else launch<128, 64, 2, 2, 1>(s, k);
  } else if (w.cout <= 32) {
    launch<128, 32, 4, 1>(s, k);
  } else if (w.cout <= 64) {
    if ((k.M + 127) / 128 >= 256) launch_n64(s, k);
    else launch<64, 64, 2, 2>(s, k);
  } else if (blocks128 >= 384) {
    // a 128-wide N tile wastes (-Cout mod 128) columns of MFMA work: 25 % at Cout = 192 (the PARSeq-tiny projections
    // and FFN outputs).  64-wide tiles cover such a Cout exactly, at a slightly lower rate per tile.
    const int waste = (128 - w.cout % 128) % 128;
    // short K (the 1x1 expansions, the ViT projections): the swizzled 128 x 64 tile runs three blocks per CU and beat the
    // 16-wave 128 x 128 tile on every such shape - qkv 192->576 +13 %, 256->1024 +10 %, 128->512 +9 %, 64->256 +8 %,
    // 512->2048 +5 %, level elsewhere (profiles/r03_conv_sweep_swizzled_tiles.txt); long K stays on the wide tile
    const bool short_k = (k.fast & 8) && k.Kpad <= 512 && g_conv_variant.load(std::memory_order_relaxed) == 0;
    if ((waste >= 48 && waste * 5 >= w.cout) || short_k) launch_n64(s, k);
    else launch_wide(s, k);
  } else {
    const long blocks64 = (long)((k.M + 127) / 128) * ((w.cout + 63) / 64);
    // few rows, many columns (the vocabulary head inside the greedy loop: 655 x 7119): 128-row tiles would idle through
    // the padding rows of the last tile (15 % at 655 rows); 64 x 64 tiles measured 51 against 56 us there
    const int pad_rows = (k.M + 127) / 128 * 128 - k.M;
    if (blocks64 >= 256 && !(k.M <= 1024 && pad_rows * 8 >= k.M)) launch_n64(s, k);
    else launch<64, 64, 2, 2>(s, k);
  }
  YMK_HIP(hipGetLastError());
  return true;
}

int conv_split_route_with_planes(long M, int cout, int kpad, int taps, bool* auto_tile);  // ymk_conv_split.hip

bool conv_planes_pair_ok(const Tensor& in, const ConvW& w1, const ConvArgs& a1, const ConvW& w2, const ConvArgs& a2) {
  if (conv_effective_split() != SPLIT_F16X2 || t_split_ctx == nullptr) return false;
  if (w1.mode != 0 || w2.mode != 0 || a1.epi != EPI_STORE || a2.epi != EPI_STORE || a1.res != nullptr) return false;
  if (a1.row_group != nullptr || a2.row_group != nullptr || a1.ln_g != nullptr) return false;
  if (a1.act != ACT_NONE && a1.act != ACT_RELU && a1.act != ACT_SILU && a1.act != ACT_GELU) return false;
  if (w1.cout % 32 != 0 || w2.cin != w1.cout || w2.kh * w2.kw <= 1 || in.planes) return false;
  const int sw1 = a1.stride_w > 0 ? a1.stride_w : a1.stride, sw2 = a2.stride_w > 0 ? a2.stride_w : a2.stride;
  const int h1 = conv_out_dim(in.h, w1.kh, a1.stride, a1.pad, a1.dil), v1 = conv_out_dim(in.w, w1.kw, sw1, a1.pad, a1.dil);
  const int h2 = conv_out_dim(h1, w2.kh, a2.stride, a2.pad, a2.dil), v2 = conv_out_dim(v1, w2.kw, sw2, a2.pad, a2.dil);
  if (h1 <= 0 || v1 <= 0 || h2 <= 0 || v2 <= 0) return false;
  if ((size_t)in.n * h1 * v1 * w1.cout * sizeof(float) >= (size_t)OOB_OFFSET) return false;  // the consumer's input view
  bool auto1 = false, auto2 = false;
  const int r1 = conv_split_route_with_planes((long)in.n * h1 * v1, w1.cout, w1.kpad, w1.kh * w1.kw, &auto1);
  const int r2 = conv_split_route_with_planes((long)in.n * h2 * v2, w2.cout, w2.kpad, w2.kh * w2.kw, &auto2);
  return auto1 && auto2 && (r1 == 2 || r1 == 3) && r2 == 2;  // producer: LDS-DMA or register-staged kernel; consumer: LDS-DMA
}

void conv2d(hipStream_t s, const Tensor& in, const ConvW& w, const ConvArgs& a, const Tensor& out) {
  YMK_CHECK(a.ln_g == nullptr, "conv2d: a fused LayerNorm goes through gemm_ln_fused");
  (void)conv2d_impl(s, in, w, a, out);
}

bool gemm_ln_fused(hipStream_t s, const float* X, int M, int K, int ldx, const float* ln_g, const float* ln_b, float ln_eps, const ConvW& w,
                   int act, const float* res, int res_ld, float* out, int out_ld, const unsigned* amax_in, unsigned* amax_out) {
  YMK_CHECK(K == w.cin && ln_g != nullptr && ln_b != nullptr, "gemm_ln_fused: K must equal the weight's in-features; gamma and beta are required");
  if (w.kh != 1 || w.kw != 1 || w.mode != 0 || amax_in == nullptr) return false;
  Tensor in{const_cast<float*>(X), 1, 1, M, K, ldx};
  Tensor o{out, 1, 1, M, w.cout, out_ld};
  Tensor r{const_cast<float*>(res), 1, 1, M, w.cout, res_ld};
  ConvArgs a;
  a.act = act;
  a.res = res ? &r : nullptr;
  a.amax_in = amax_in;
  a.amax_out = amax_out;
  a.ln_g = ln_g;
  a.ln_b = ln_b;
  a.ln_eps = ln_eps;
  return conv2d_impl(s, in, w, a, o);
}

bool vit_mlp_split_launch(hipStream_t s, SplitCtx* ctx, float* x, int M, int ld, const float* ln_g, const float* ln_b, float ln_eps,
                          float ln_bound, const ConvW& fc1, const ConvW& fc2);  // ymk_conv_split.hip

bool vit_mlp_fused(hipStream_t s, float* x, int M, int ld, const float* ln_g, const float* ln_b, float ln_eps, float ln_bound, const ConvW& fc1,
                   const ConvW& fc2) {
  if (conv_effective_split() != SPLIT_F16X2 || t_split_ctx == nullptr) return false;
  return vit_mlp_split_launch(s, t_split_ctx, x, M, ld, ln_g, ln_b, ln_eps, ln_bound, fc1, fc2);
}

void gemm(hipStream_t s, const float* A, int M, int K, int lda, const ConvW& w, int act, const float* res, int res_ld,
          float* out, int out_ld, const int* row_group, const int* group_open, int epi, const unsigned* amax_in, unsigned* amax_out) {
  YMK_CHECK(K == w.cin, "gemm: K " + std::to_string(K) + " != weight in-features " + std::to_string(w.cin));
  // The kernels address their A operand through a buffer descriptor with 32-bit byte offsets (conv2d: "input view must stay
  // below 4 GiB"); a row-major GEMM has no such natural bound - the fc2 input of a 2048-line PARSeq forward is 1.6 M rows x
  // 3 KB - so rows are processed in chunks whose view fits.  Rows are independent and a chunk is a multiple of 1024 rows
  // (every tile height divides it), so the chunked result is the unchunked one bit for bit; the max|y| record is a maximum.
  const size_t in_row = (size_t)std::max(lda, 1) * sizeof(float);
  size_t max_rows = ((size_t)OOB_OFFSET - 4096) / in_row;
  if (gemm_row_limit() > 0) max_rows = std::min(max_rows, (size_t)gemm_row_limit());
  max_rows = std::max<size_t>(max_rows / 1024 * 1024, 1024);
  for (size_t m0 = 0; m0 < (size_t)std::max(M, 0); m0 += max_rows) {
    const int rows = (int)std::min(max_rows, (size_t)M - m0);
    Tensor in{const_cast<float*>(A) + m0 * lda, 1, 1, rows, K, lda};
    Tensor o{out + m0 * out_ld, 1, 1, rows, w.cout, out_ld};
    Tensor r{res ? const_cast<float*>(res) + m0 * res_ld : nullptr, 1, 1, rows, w.cout, res_ld};
    ConvArgs a;
    a.act = act;
    a.res = res ? &r : nullptr;
    a.row_group = row_group ? row_group + m0 : nullptr;
    a.group_open = group_open;
    a.epi = epi;
    a.amax_in = amax_in;
    a.amax_out = amax_out;
    conv2d(s, in, w, a, o);
  }
}

// ------------------------------------------------------------------ host packing
void pack_conv_weight(const float* oihw, int cout, int cin, int kh, int kw, bool tap4,
                      std::vector<float>& panel, int& kpad, int& ctiles) {
  const int taps = kh * kw;
  if (tap4) {
    YMK_CHECK(cin <= 4, "tap4 packing wants cin <= 4");
    ctiles = 0;
    kpad = ((taps * 4 + 31) / 32) * 32;
    panel.assign((size_t)((cout + 127) / 128 * 128) * kpad, 0.f);
    for (int co = 0; co < cout; ++co)
      for (int c = 0; c < cin; ++c)
        for (int tp = 0; tp < taps; ++tp)
          panel[(size_t)co * kpad + tp * 4 + c] = oihw[((size_t)co * cin + c) * taps + tp];
  } else {
    ctiles = (cin + 31) / 32;
    kpad = taps * ctiles * 32;
    panel.assign((size_t)((cout + 127) / 128 * 128) * kpad, 0.f);  // rows padded: no N guard in the kernel
    for (int co = 0; co < cout; ++co)
      for (int c = 0; c < cin; ++c)
        for (int tp = 0; tp < taps; ++tp)
          panel[(size_t)co * kpad + (size_t)tp * ctiles * 32 + c] = oihw[((size_t)co * cin + c) * taps + tp];
  }
}

}  // namespace ymk
