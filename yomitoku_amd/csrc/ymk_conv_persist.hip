// Persistent form of the swizzled 128 x 64 implicit-GEMM tile (ymk_conv.hip, OPT = 2): a block walks a SEQUENCE of output
// tiles and issues the first K tile's global loads of tile i + 1 before it runs the epilogue of tile i.
//
// Why: the K <= 512 layers (six or fewer K tiles per output tile) spend a large share of a block's life in the prologue
// (pixel index arithmetic, first global loads: ~2 us of HBM latency under load) and in the epilogue; the one-tile-per-block
// kernel hides them only behind the other two blocks of the CU (MFMA pipe 0.665 busy, profiles/r03_analyzer_pmc_mfma_busy.csv,
// against 0.754 on the long-K tile).  Here the next tile's loads fly while the accumulators of this one drain, and no block is
// launched or retired between tiles.
//
// Same arithmetic as conv_igemm: same K order per output element (v_mfma_f32_32x32x2_f32 chain over the K tiles in order),
// same epilogue (epilogue_tile) - every output bit equals the one-tile kernel's.
//
// STATUS: opt-in (ymk_debug_option("conv_fast") bit 5), written at the end of round 3 after the GPU budget of the round was
// spent: compiles for gfx950 (80 VGPRs at six waves per SIMD; 30 dwords spilled in the per-tile prologue / epilogue, none
// inside the K loop), NOT yet run on hardware.  tests/test_ops_gpu.py holds its bit-identity
// test behind YMK_EXPERIMENTAL=1 and tools/jobs/r04_persistent.sh the first measurement; nothing on the product path
// reaches this file until both have been run.
#include "ymk_conv_kernel.h"

namespace ymk {

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, blocks_per_cu(BM, BN, 2) * WM * WN / 4) void conv_igemm_persist(ConvK p) {
  constexpr int OPT = 2;  // swizzled K-tile rows
  constexpr int LDR = lds_row(OPT);
  constexpr int NT = 64 * WM * WN;
  static_assert((NT / 8) % 16 == 0, "swizzle: staging passes must keep (row >> 1) & 7");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RPP = NT / 8;
  constexpr int APASS = BM / RPP, BPASS = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must divide by the staging pass");
  constexpr int STAGE = (BM + BN) * LDR;
  constexpr int LDC = BN + 4;
  constexpr int EROWS = (2 * STAGE / LDC) / 32 * 32 < BM ? (2 * STAGE / LDC) / 32 * 32 : BM;
  static_assert(EROWS >= WTM && EROWS % WTM == 0, "an epilogue pass must hold whole wave tiles");
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int t = threadIdx.x;
  // tiles of this block: XCD x (blocks x, x + 8, ...) owns a contiguous range of tiles, n fastest (the split of conv_igemm's
  // bijective remap); its P blocks take tiles start + idx, start + idx + P, ... - at any moment the XCD works on a window of
  // P consecutive tiles, so the A row panels of a window are shared in its L2
  int tile, tile_end;
  const int tile_step = (int)gridDim.x >> 3;  // the host launches a multiple of 8 blocks
  {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = p.ntiles >> 3, r = p.ntiles & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    tile = start + idx;
    tile_end = start + q + (xcd < r ? 1 : 0);
  }
  if (tile >= tile_end) return;  // block-uniform

  const int colq = t & 7, rowb = t >> 3;
  int pixb[APASS], ih0[APASS], iw0[APASS];
  unsigned voff[APASS];
  const float* wrow;
  int m0, n0;
  int cur_kh, cur_kw, cur_cc;  // K-tile cursor (wave-uniform): filter tap and channel tile
  const bool pointwise = (p.fast & 1) && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.stride_w == 1 && p.pad == 0;

  // staging coordinates of output tile `tl` (conv_igemm's prologue)
  auto set_tile = [&](int tl) {
    const int tile_m = tl / p.ntiles_n, tile_n = tl - tile_m * p.ntiles_n;
    m0 = tile_m * BM;
    n0 = tile_n * BN;
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
      voff[i] = OOB_OFFSET;
    }
    wrow = p.w + (size_t)(n0 + rowb) * p.Kpad + colq * 4;
    cur_kh = cur_kw = cur_cc = 0;
  };

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  f32x4 ra[APASS], rb[BPASS];

  // gather of K tile kt of the current tile (conv_igemm's load_tile, MODE 0): padding taps / tail rows read an offset past
  // the descriptor's range and come back as zeros
  auto load_tile = [&](int kt) {
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
    const bool chan_ok = cur_cc * 32 + colq * 4 < p.C;
    const int soff = cur_cc * 128;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(chan_ok ? voff[i] : OOB_OFFSET), soff, 0);
      ra[i] = __builtin_bit_cast(f32x4, v);
    }
    if (++cur_cc == p.ctiles) {
      cur_cc = 0;
      if (++cur_kw == p.KW) {
        cur_kw = 0;
        ++cur_kh;
      }
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j)
      rb[j] = *reinterpret_cast<const f32x4*>(wrow + (size_t)(RPP * j) * p.Kpad + kt * 32);
  };

  const int st_off = lds_slot<OPT>(rowb, colq);
  auto store_tile = [&](int buf) {
    float* As = lds + buf * STAGE + st_off;
    float* Bs = As + BM * LDR;
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      *reinterpret_cast<f32x4*>(As + RPP * i * LDR) = ra[i];
#pragma unroll
    for (int j = 0; j < BPASS; ++j)
      *reinterpret_cast<f32x4*>(Bs + RPP * j * LDR) = rb[j];
  };

  const int wv = t >> 6, lane = t & 63;
  const int wm = wv / WN, wn = wv - wm * WN;
  const int li = lane & 31, lh = lane >> 5;
  // chunk kc of a lane's row sits at slot (2 kc + lh) ^ swizzle: the chunk index only flips bits 1-2 of the slot, so one
  // offset per operand and an XOR per read replace four offsets per operand (registers are what limits this kernel)
  const int a_off = lds_slot<OPT>(wm * WTM + li, lh);
  const int b_off = BM * LDR + lds_slot<OPT>(wn * WTN + li, lh);
  f32x16 acc[TM][TN];
  auto compute = [&](int buf) {
    const float* Ts = lds + buf * STAGE;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        fa[a] = *reinterpret_cast<const f32x4*>(Ts + (a_off ^ (kc * 8)) + a * 32 * LDR);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        fb[b] = *reinterpret_cast<const f32x4*>(Ts + (b_off ^ (kc * 8)) + b * 32 * LDR);
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

  const int ktiles = p.Kpad >> 5;
  set_tile(tile);
  load_tile(0);
  for (;;) {
    store_tile(0);
    __syncthreads();
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int kt = 0; kt < ktiles; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < ktiles) load_tile(kt + 1);
      compute(buf);
      if (kt + 1 < ktiles) store_tile(buf ^ 1);
      __syncthreads();
    }

    // the next tile's first K tile is on its way while this tile's accumulators drain
    const int em0 = m0, en0 = n0;
    tile += tile_step;
    const bool more = tile < tile_end;  // block-uniform
    if (more) {
      set_tile(tile);
      load_tile(0);
    }

    // epilogue of conv_igemm: accumulators -> LDS tile [EROWS][LDC] -> 16 B per lane, full rows coalesced
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
      epilogue_tile<EROWS, BN, NT, false, 2>(p, Cs, em0 + e0, en0, t);  // two rows in flight: the next tile's operands hold 12 registers
    }
    if (!more) break;
    __syncthreads();  // every thread has read its rows of Cs before stage 0 of the next tile is written
  }
}

// Host side: taken for plain-store / deconvolution launches of the swizzled 128 x 64 tile with at least two tiles per
// resident block (below that there is nothing to overlap); the row predicate of the greedy loop's head stays on conv_igemm.
bool conv2d_persistent(hipStream_t s, ConvK& k) {
  constexpr int BM = 128, BN = 64, SLOTS = 256 * 3;  // three blocks per CU
  if (k.mode != 0 || k.row_group != nullptr || k.epi == EPI_ROWMAX) return false;
  const int mt = (k.M + BM - 1) / BM, nt = (k.Cout + BN - 1) / BN;
  if ((long)mt * nt < 2L * SLOTS) return false;
  k.ntiles_n = nt;
  k.ntiles = mt * nt;
  auto* e = conv_prof_open(s, k, BM, BN, SLOTS, 1);
  hipLaunchKernelGGL((conv_igemm_persist<BM, BN, 4, 2>), dim3(SLOTS), dim3(512), 0, s, k);
  if (e) YMK_HIP(hipEventRecord(e->second, s));
  YMK_HIP(hipGetLastError());
  return true;
}

}  // namespace ymk
