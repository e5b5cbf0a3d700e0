// Implicit-GEMM convolution with SPLIT operands and fp32 accumulation (gfx950), the alternative to the exact fp32-MFMA
// kernel of ymk_conv.hip for the same layers (models/dbnet_plus.py:33-38,56-127; rtdetr_backbone.py;
// parseq_transformer.py): same GEMM view, same gathers, same epilogue.
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TFLOP/s); the 16-bit MFMAs are 16 x faster.  An fp32 value
// splits exactly into short pieces (round-to-nearest at every cut, remainders exact in fp32), so a product is a sum of
// short x short products, each exact in fp32:
//   bf16, 3 planes, 6 MFMAs  x_h y_h + (x_h y_m + x_m y_h) + (x_m y_m + x_h y_l + x_l y_h)   dropped terms <= 2^-23 |xy|
//   bf16, 2 planes, 3 MFMAs  x_h y_h + (x_h y_l + x_l y_h)                                   dropped terms <= 2^-15 |xy|
//   fp16, 2 planes, 3 MFMAs  the same three terms on 11-bit pieces                           dropped terms <= 2^-21 |xy|
// fp16 pieces carry 22 of the 24 significand bits in the same 3 MFMAs that two bf16 planes need (16 bits), i.e. fp32-grade
// products at 16 / 3 = 5.3 x the fp32 matrix rate: for K >= 64 the fp32 accumulation's own rounding (2^-24 of a partial sum
// that is ~sqrt(K) products large) exceeds the 2^-21 per product.  What fp16 lacks is range, so the operands are scaled by
// powers of two (exact): the activations by ONE scale per launch that puts max|x| of the input view into [2^14, 2^15) -
// a pass over the input (k_absmax) ahead of the convolution, its result read by the kernel from device memory, no host
// round trip -, the weights per output channel at panel-build time (row maximum into [2^14, 2^15); the inverse goes into the
// epilogue's per-channel scale).  Elements more than 2^18 below the tensor's maximum lose low-plane bits gradually (fp16
// subnormals); nothing overflows.  Accumulation is fp32 inside the MFMA (not an fmaf chain in k order: results agree
// with the fp32 kernel to rounding, not bit for bit).
//
// Data path: activations stay fp32 in HBM; a thread splits the 4 floats it stages into NS x 4 halves on the way to LDS
// (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).  Weights are split once per (panel, format), the first time a launch asks, into
// panels [Cout^128][K tiles][NS][32], so a K tile's B rows are copied to LDS verbatim.  LDS row = NS x 64 B + 16 B pad
// (conflict-free ds_read_b128: row stride 36 or 52 dwords = 4 x odd), one ds_read_b128 per (32-row tile, plane, 16-k
// step) feeds v_mfma_f32_32x32x16_{bf16,f16}: lane l holds row l & 31, k = 8 (l >> 5) .. + 7 of the step, for A and B alike.
#include <atomic>
#include <string>
#include <type_traits>
#include <unordered_map>

#include "ymk_conv_kernel.h"

namespace ymk {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// FMT 0: bf16 pieces, 1: fp16 pieces of the scaled operand
template <int FMT> struct Half;
template <> struct Half<0> { typedef bf16x2_t v2; typedef bf16x8_t v8; };
template <> struct Half<1> { typedef f16x2_t v2; typedef f16x8_t v8; };

template <int FMT>
__device__ __forceinline__ f32x16 mfma16(const typename Half<FMT>::v8 a, const typename Half<FMT>::v8 b, const f32x16 c) {
  if constexpr (FMT == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// 4 floats (times sa, a power of two: exact) -> NS planes of 4 halves (8 B each)
template <int FMT, int NS>
__device__ __forceinline__ void split4(const f32x4 v, float sa, uint2* planes) {
  typedef typename Half<FMT>::v2 h2;
  f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
  if (FMT == 1) {
    a *= sa;
    b *= sa;
  }
#pragma unroll
  for (int pl = 0; pl < NS; ++pl) {
    const h2 pa = __builtin_convertvector(a, h2), pb = __builtin_convertvector(b, h2);
    planes[pl].x = __builtin_bit_cast(unsigned, pa);
    planes[pl].y = __builtin_bit_cast(unsigned, pb);
    if (pl + 1 < NS) {
      a -= __builtin_convertvector(pa, f32x2_t);  // exact: the remainder of a round-to-nearest cut fits fp32
      b -= __builtin_convertvector(pb, f32x2_t);
    }
  }
}

__device__ __forceinline__ float2 f16_scales(unsigned amax_bits) { return f16_plane_scales(amax_bits); }  // ymk_conv_kernel.h

// blocks of this shape a CU's 160 KB of LDS holds (at most 2 are asked for) -> minimum waves per SIMD for the register allocator
template <int BM, int BN, int WM, int WN, int NS, int KS>
constexpr int bf16_waves_per_simd() {
  constexpr int lds = 2 * (BM + BN) * (KS * NS * 64 + 16);
  constexpr int blocks = 2 * lds <= 160 * 1024 ? 2 : 1;
  return blocks * 64 * WM * WN / 256;
}

// KS: 32-k tiles per LDS stage (a stage = one barrier interval: KS = 2 halves the barriers per k and doubles the MFMAs a wave
// issues between them); PF: stages the global loads run ahead of the MFMAs (2 = two register sets, as conv_igemm's PF).
// OPL (fp16 form): the epilogue writes fp16 planes (epilogue_tile<.., PL>, Tensor::planes).
template <int BM, int BN, int WM, int WN, int NS, int KS, int PF, int FMT, bool OPL = false>
__global__ __launch_bounds__(64 * WM * WN, (bf16_waves_per_simd<BM, BN, WM, WN, NS, KS>())) void conv_igemm_split(ConvK p, const uint4* __restrict__ wsplit) {
  typedef typename Half<FMT>::v8 h8;
  static_assert(PF >= 1 && PF <= 3, "prefetch: 1 = one stage ahead, 2 = two ahead, 3 = two ahead with the LDS stores threaded through the MFMAs");
  constexpr int NSET = PF == 1 ? 1 : 2;  // register sets of staged loads
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int TILEB = NS * 64;               // bytes of one 32-k tile of a row: NS planes of 32 bf16
  constexpr int ROWB = KS * TILEB + 16;        // bytes of an LDS row: KS tiles + pad (row stride = 4 x odd dwords)
  constexpr int STAGE_B = (BM + BN) * ROWB;    // bytes of one stage
  constexpr int RPP = NT / 8;                  // A rows staged per pass (8 threads x 16 B of fp32 per 32-k row)
  constexpr int APASS = BM / RPP;
  static_assert(BM % RPP == 0, "tile rows must divide by the staging pass");
  constexpr int PPR = NS * 4;                  // 16 B pieces of a B row per 32-k tile
  constexpr int BPASS = (BN * PPR + NT - 1) / NT;
  constexpr int LDC = BN + 4;
  constexpr int ECAP = 2 * STAGE_B / 4 / LDC;  // rows of the fp32 output tile the two stages can hold
  constexpr int EROWS = ECAP >= BM ? BM : (ECAP >= BM / 2 ? BM / 2 : BM / 4);  // rows per epilogue pass: divides BM
  static_assert(EROWS >= WTM && EROWS % WTM == 0 && EROWS <= ECAP, "an epilogue pass must hold whole wave tiles");
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_B];

  const int t = threadIdx.x;
  float sa = 1.f, inv_sa = 1.f;
  if (FMT == 1) {
    const float2 sc = f16_scales((unsigned)__builtin_amdgcn_readfirstlane((int)amax_read(p.amax, t)));  // wave-uniform: scalar registers
    sa = sc.x;
    inv_sa = sc.y;
  }
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / p.ntiles_n, tile_n = tile - tile_m * p.ntiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (!tile_needed<BM, NT>(p, m0, t)) return;

  const int colq = t & 7, rowb = t >> 3;
  int pixb[APASS], ih0[APASS], iw0[APASS];
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
      ih0[i] = -(1 << 20);
      iw0[i] = 0;
    }
  }
  const int ktiles = p.Kpad >> 5;   // a multiple of KS (the launcher checks)
  const int nstages = ktiles / KS;
  // B pieces of this thread: piece q = t + NT * j of the tile's BN x PPR, row-major
  const uint4* wsrc[BPASS];
  int boff[BPASS];  // byte offset inside the B half of a stage, or -1 past the tile
#pragma unroll
  for (int j = 0; j < BPASS; ++j) {
    const int q = t + NT * j;
    const int row = q / PPR, piece = q - row * PPR;
    const bool ok = q < BN * PPR;
    wsrc[j] = wsplit + ((size_t)(n0 + (ok ? row : 0)) * ktiles) * PPR + piece;
    boff[j] = ok ? row * ROWB + piece * 16 : -1;
  }

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  f32x4 ra[NSET][KS][APASS];
  uint4 rb[NSET][KS][BPASS];
  int cur_kh = 0, cur_kw = 0, cur_cc = 0;
  unsigned voff[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) voff[i] = OOB_OFFSET;

  // stage st (K tiles st * KS .. + KS - 1) -> register set `set`; the tap cursor walks K tile by K tile
  auto load_stage = [&](int st, int set) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (FMT == 1 ? (cur_cc == 0 || p.KH * p.KW > 1) : cur_cc == 0) {  // wave-uniform: a new filter tap (fp16 panels: every K tile of a k x k layer)
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
        ra[set][ks][i] = __builtin_bit_cast(f32x4, v);
      }
      if (FMT == 1) {  // channel-major panels: the taps of one channel slice back to back
        if (++cur_kw == p.KW) {
          cur_kw = 0;
          if (++cur_kh == p.KH) {
            cur_kh = 0;
            ++cur_cc;
          }
        }
      } else if (++cur_cc == p.ctiles) {
        cur_cc = 0;
        if (++cur_kw == p.KW) {
          cur_kw = 0;
          ++cur_kh;
        }
      }
#pragma unroll
      for (int j = 0; j < BPASS; ++j) rb[set][ks][j] = wsrc[j][(size_t)(st * KS + ks) * PPR];
    }
  };

  auto store_stage = [&](int buf, int set) {
    char* As = lds + buf * STAGE_B;
    char* Bs = As + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        uint2 pl[NS];
        split4<FMT, NS>(ra[set][ks][i], sa, pl);
        char* row = As + (rowb + RPP * i) * ROWB + ks * TILEB + colq * 8;
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<uint2*>(row + q * 64) = pl[q];
      }
#pragma unroll
      for (int j = 0; j < BPASS; ++j)
        if (boff[j] >= 0) *reinterpret_cast<uint4*>(Bs + boff[j] + ks * TILEB) = rb[set][ks][j];
    }
  };

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

  // product terms, smallest first: (A plane, B plane)
  constexpr int NTERM = NS == 3 ? 6 : 3;
  constexpr int TA[6] = {NS - 1, 0, 1, 1, 0, 0};
  constexpr int TB[6] = {0, NS - 1, 1, 0, 1, 0};
  constexpr int T0 = NS == 3 ? 0 : 3;

  auto compute = [&](int buf) {
    const char* As = lds + buf * STAGE_B + (wm * WTM + li) * ROWB + lh * 16;
    const char* Bs = lds + buf * STAGE_B + BM * ROWB + (wn * WTN + li) * ROWB + lh * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int s = 0; s < 2; ++s) {  // two 16-k steps per 32-k tile
        h8 fa[TM][NS], fb[TN][NS];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int q = 0; q < NS; ++q) fa[a][q] = *reinterpret_cast<const h8*>(As + a * 32 * ROWB + ks * TILEB + q * 64 + s * 32);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int q = 0; q < NS; ++q) fb[b][q] = *reinterpret_cast<const h8*>(Bs + b * 32 * ROWB + ks * TILEB + q * 64 + s * 32);
        // term-major: consecutive MFMAs go to different accumulators whenever a wave owns more than one 32 x 32 tile
#pragma unroll
        for (int tm = 0; tm < NTERM; ++tm)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
              acc[a][b] = mfma16<FMT>(fa[a][TA[T0 + tm]], fb[b][TB[T0 + tm]], acc[a][b]);
      }
  };

  // PF == 3: the same MFMAs with the next stage's conversion + LDS stores slotted between them, one chunk (one A pass: split
  // + NS ds_write_b64, or one B piece: ds_write_b128) every `stride` MFMAs, pinned by scheduling fences.  A wave issues in
  // order and an MFMA occupies the matrix pipe for 32 cycles: the few VALU / DS instructions behind it ride in its shadow
  // instead of forming a separate phase between the last MFMA and the barrier (the staged data was loaded a whole stage
  // earlier, so no chunk waits on memory).
  auto compute_store = [&](int buf, int sbuf, int set, auto do_store) {
    constexpr bool STORE = decltype(do_store)::value;
    const char* As = lds + buf * STAGE_B + (wm * WTM + li) * ROWB + lh * 16;
    const char* Bs = lds + buf * STAGE_B + BM * ROWB + (wn * WTN + li) * ROWB + lh * 16;
    char* SA = lds + sbuf * STAGE_B;
    char* SB = SA + BM * ROWB;
    constexpr int NCH = KS * (APASS + BPASS);
    constexpr int NM = KS * 2 * NTERM * TM * TN;
    constexpr int STRIDE = NM / NCH > 0 ? NM / NCH : 1;
    auto chunk = [&](int c) {
      const int ks = c / (APASS + BPASS), r = c - ks * (APASS + BPASS);
      if (r < APASS) {
        uint2 pl[NS];
        split4<FMT, NS>(ra[set][ks][r], sa, pl);
        char* row = SA + (rowb + RPP * r) * ROWB + ks * TILEB + colq * 8;
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<uint2*>(row + q * 64) = pl[q];
      } else {
        const int j = r - APASS;
        if (boff[j] >= 0) *reinterpret_cast<uint4*>(SB + boff[j] + ks * TILEB) = rb[set][ks][j];
      }
    };
    int mi = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        h8 fa[TM][NS], fb[TN][NS];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int q = 0; q < NS; ++q) fa[a][q] = *reinterpret_cast<const h8*>(As + a * 32 * ROWB + ks * TILEB + q * 64 + s * 32);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int q = 0; q < NS; ++q) fb[b][q] = *reinterpret_cast<const h8*>(Bs + b * 32 * ROWB + ks * TILEB + q * 64 + s * 32);
#pragma unroll
        for (int tm = 0; tm < NTERM; ++tm)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
              acc[a][b] = mfma16<FMT>(fa[a][TA[T0 + tm]], fb[b][TB[T0 + tm]], acc[a][b]);
              if (STORE && mi % STRIDE == STRIDE - 1 && mi / STRIDE < NCH) {
                __builtin_amdgcn_sched_barrier(0);
                chunk(mi / STRIDE);
                __builtin_amdgcn_sched_barrier(0);
              }
              ++mi;
            }
      }
    if (STORE) {
#pragma unroll
      for (int c = NM / STRIDE; c < NCH; ++c) chunk(c);  // more chunks than MFMA slots (short stages): the rest at the end
    }
  };

  load_stage(0, 0);
  store_stage(0, 0);
  if (PF >= 2 && nstages > 1) load_stage(1, 1);
  __syncthreads();
  if (PF == 3) {
    for (int st = 0; st < nstages; st += 2) {
      if (st + 2 < nstages) load_stage(st + 2, 0);
      if (st + 1 < nstages) compute_store(0, 1, NSET - 1, std::true_type{});
      else compute_store(0, 1, NSET - 1, std::false_type{});
      __syncthreads();
      if (st + 1 >= nstages) break;
      if (st + 3 < nstages) load_stage(st + 3, NSET - 1);
      if (st + 2 < nstages) compute_store(1, 0, 0, std::true_type{});
      else compute_store(1, 0, 0, std::false_type{});
      __syncthreads();
    }
  } else if (PF == 1) {
    for (int st = 0; st < nstages; ++st) {
      const int buf = st & 1;
      if (st + 1 < nstages) load_stage(st + 1, 0);
      compute(buf);
      if (st + 1 < nstages) store_stage(buf ^ 1, 0);
      __syncthreads();
    }
  } else {
    // stage s travels in register set s & 1: loaded at the top of step s - 2, written to LDS[s & 1] at the end of step s - 1
    for (int st = 0; st < nstages; st += 2) {
      if (st + 2 < nstages) load_stage(st + 2, 0);
      compute(0);
      if (st + 1 < nstages) store_stage(1, NSET - 1);
      __syncthreads();
      if (st + 1 >= nstages) break;
      if (st + 3 < nstages) load_stage(st + 3, NSET - 1);
      compute(1);
      if (st + 2 < nstages) store_stage(0, 0);
      __syncthreads();
    }
  }

  float* Cs = reinterpret_cast<float*>(lds);
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
            Cs[row * LDC + wn * WTN + b * 32 + li] = FMT == 1 ? acc[a][b][r] * inv_sa : acc[a][b][r];
          }
    }
    __syncthreads();
    epilogue_tile<EROWS, BN, NT, false, 4, OPL>(p, Cs, m0 + e0, n0, t);
  }
}

// ---- fp32 panel [rows][kpad] -> split panel [rows][kpad / 32][NS][32] halves (once per panel and format)
template <int NS>
__global__ void k_split_panel(const float* __restrict__ w, unsigned short* __restrict__ out, size_t n_elems, int kpad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_elems) return;
  const size_t row = i / kpad;
  const int k = (int)(i - row * kpad), kt = k >> 5, kk = k & 31;
  float r = w[i];
#pragma unroll
  for (int pl = 0; pl < NS; ++pl) {
    const __bf16 h = (__bf16)r;
    out[((row * (kpad >> 5) + kt) * NS + pl) * 32 + kk] = __builtin_bit_cast(unsigned short, h);
    r -= (float)h;
  }
}

// fp16 form: one block per panel row.  The row is scaled by the power of two that puts its largest |w| into [2^14, 2^15);
// scale_out[row] = (BatchNorm scale of the channel, or 1) / that power - what the epilogue multiplies the accumulators by.
// K ORDER of the fp16 panels: CHANNEL-major - K tile (channel slice cc, tap) sits at cc * taps + tap, where the fp32 panel has
// it at tap * ctiles + cc.  The kernels then walk the taps of one 32-channel slice back to back: the nine shifted windows of
// a 3 x 3 layer overlap in all but one pixel column / row, and with only 128 bytes per pixel live they stay in L1 / L2, where
// the tap-major order streams the whole Cin x 4 B of every pixel past between two visits (section 9 item 3 of round 3).
// perm = 1 (1 x 1 panels only; the fused ViT MLP, ymk_vit_mlp.hip): inside every 32-wide K tile, source position k goes to the
// slot whose MFMA k index reads it as "accumulator register order": slot 16 s + 8 lh + j holds k = 16 s + 8 (j >> 2) + 4 lh +
// (j & 3) - the order in which a lane of the TRANSPOSED first product holds the hidden units of its row.
__global__ void k_split_panel_f16(const float* __restrict__ w, unsigned short* __restrict__ out, int kpad, const float* __restrict__ scale,
                                  int cout, float* __restrict__ scale_out, int taps, int ctiles, int perm) {
  __shared__ unsigned red[4];
  const int row = blockIdx.x, t = threadIdx.x;
  const float* wr = w + (size_t)row * kpad;
  unsigned m = 0;
  for (int k = t; k < kpad; k += 256) m = max(m, __float_as_uint(fabsf(wr[k])));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  m = max(max(red[0], red[1]), max(red[2], red[3]));
  const float2 sc = f16_scales(m);
  for (int k = t; k < kpad; k += 256) {
    const int kt_src = k >> 5;
    int kk = k & 31;
    const int tap = kt_src / ctiles, cc = kt_src - tap * ctiles;
    const int kt = cc * taps + tap;
    if (perm) {  // source k = 16 s + 8 q + 4 lh + e  ->  slot 16 s + 8 lh + 4 q + e
      const int rem = kk & 15;
      kk = (kk & 16) | (((rem >> 2) & 1) << 3) | ((rem >> 3) << 2) | (rem & 3);
    }
    float r = wr[k] * sc.x;
    const _Float16 h = (_Float16)r;
    r -= (float)h;
    const _Float16 l = (_Float16)r;
    unsigned short* o = out + (((size_t)row * (kpad >> 5) + kt) * 2) * 32 + kk;
    o[0] = __builtin_bit_cast(unsigned short, h);
    o[32] = __builtin_bit_cast(unsigned short, l);
  }
  if (t == 0 && row < cout) scale_out[row] = (scale ? scale[row] : 1.f) * sc.y;
}

// ---- max|x| over an NHWC view (pixels x c floats, pixel stride ld; c, ld multiples of 4) -> atomicMax of the fp32 bit
// patterns into word 0 of the record `slot` (zero before the launch; the other lines of these records stay zero); the
// launch also clears word 0 of `next`, the record the following launch will use
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ in, size_t items, int c4, int ld, unsigned* __restrict__ slot,
                                                unsigned* __restrict__ next) {
  __shared__ unsigned red[4];
  const int t = threadIdx.x;
  if (blockIdx.x == 0 && t == 0) *next = 0u;
  unsigned m = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  auto fold = [&](const float4 v) {
    m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
  };
  size_t i = (size_t)blockIdx.x * 256 + t;
  if (ld == c4 * 4) {  // contiguous view
    const float4* p = reinterpret_cast<const float4*>(in);
    for (; i + 3 * stride < items; i += 4 * stride) {  // four loads in flight per thread
      const float4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
      fold(a);
      fold(b);
      fold(c);
      fold(d);
    }
    for (; i < items; i += stride) fold(p[i]);
  } else {
    for (; i < items; i += stride) {
      const size_t pix = i / (unsigned)c4;
      const int q = (int)(i - pix * (unsigned)c4);
      fold(*reinterpret_cast<const float4*>(in + pix * ld + q * 4));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  if (t == 0) atomicMax(slot, max(max(red[0], red[1]), max(red[2], red[3])));
}

void absmax_record(hipStream_t s, const float* x, size_t n, unsigned* rec) {
  const size_t items = n / 4;
  const int blocks = (int)std::min<size_t>(1024, (items + 1023) / 1024);
  hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(256), 0, s, x, items, 1, 4, rec, rec + AMAX_LINE_WORDS);  // (clears a word nobody reads)
  YMK_HIP(hipGetLastError());
}

// ---- self-check of the producers' records (ymk_debug_option("amax_check", 1); tests/test_conv_split_gpu.py): a launch whose
// input came with a record ALSO makes the pass, and a one-wave kernel compares the two: counters[0] = launches checked,
// [1] = records BELOW the true max|x| (a stale or incomplete record: the fp16 planes may overflow - a bug), [2] = records more
// than 2^8 above it (a derived bound so loose that it costs low-plane bits), [3] = the largest record / truth ratio as a power of two
static std::atomic<int> g_amax_check{0};
static unsigned long long* g_amax_counters = nullptr;  // device, 4 words
__global__ void k_amax_verify(const unsigned* __restrict__ rec, const unsigned* __restrict__ truth, unsigned long long* __restrict__ counters) {
  const unsigned got = amax_read(rec, threadIdx.x), want = amax_read(truth, threadIdx.x);
  if (threadIdx.x != 0) return;
  atomicAdd(&counters[0], 1ull);
  if (got < want) atomicAdd(&counters[1], 1ull);
  const int slack = (int)(got >> 23) - (int)(want >> 23);  // exponent distance
  if (want != 0u && slack > 8) atomicAdd(&counters[2], 1ull);
  if (want != 0u && slack > 0) atomicMax(&counters[3], (unsigned long long)slack);
}
void amax_check_counters(long long* out4) {
  for (int i = 0; i < 4; ++i) out4[i] = 0;
  if (!g_amax_counters) return;
  YMK_HIP(hipDeviceSynchronize());
  YMK_HIP(hipMemcpy(out4, g_amax_counters, 4 * sizeof(long long), hipMemcpyDeviceToHost));
}

// cache key of the fc2 panel of the fused ViT MLP: fp16 planes with the accumulator-order K permutation (never a "conv_split" code)
constexpr int SPLIT_F16X2_PERM = 17;

// ---- per-model state
class SplitCtx {
 public:
  struct Panels {
    void* planes = nullptr;
    float* scale = nullptr;  // fp16 form only: the epilogue's per-channel scale with the row's power of two taken back out
  };
  ~SplitCtx() {
    for (void* q : allocs_) dev_free(q);
    if (moved_) (void)hipEventDestroy(moved_);
  }
  // split copy of panel w for `code`.  Built at ymk_model_finalize for the precision the model runs (prebuild below); a pair
  // that is first asked for inside a forward - the process-wide precision switched after finalize, a single-operator call -
  // is built here on stream s, counted ("lazy_panel_builds"), and waited for
  const Panels& panels(hipStream_t s, const ConvW& w, int code) {
    const Key key{w.w, code};
    auto it = cache_.find(key);
    if (it != cache_.end()) return it->second;
    note_lazy_panel_build();
    const Panels pn = build(s, w, code);
    // on the stream of the forward that asked first: wait for it here, so that a later forward of this model on ANOTHER
    // stream (a second lane, a caller that changed streams) can never read a panel that is still being written
    forward_sync(s);
    return cache_.emplace(key, pn).first->second;
  }
  // every (panel, code) pair a forward of the owning model can ask for, and the max|x| words, on stream s (ymk_model_finalize:
  // the null stream, followed by a device synchronisation) - so that a forward never allocates, never builds, never waits
  void prebuild(hipStream_t s, const std::vector<ConvW>& convs, const std::vector<ConvW>& perm, int code) {
    if (code != 2 && code != 3 && code != SPLIT_F16X2) return;
    for (const ConvW& w : convs)
      if (w.mode == 0 && cache_.find(Key{w.w, code}) == cache_.end()) cache_.emplace(Key{w.w, code}, build(s, w, code));
    if (code == SPLIT_F16X2)
      for (const ConvW& w : perm)
        if (w.mode == 0 && cache_.find(Key{w.w, SPLIT_F16X2_PERM}) == cache_.end())
          cache_.emplace(Key{w.w, SPLIT_F16X2_PERM}, build(s, w, SPLIT_F16X2_PERM));
    ensure_slots(s);
  }
  // max|x| of the input view of launch k, on stream s; returns the device word the convolution kernel reads
  const unsigned* absmax(hipStream_t s, const ConvK& k) {
    ensure_slots(s);
    // the two words alternate along ONE stream (each launch clears the other word for its successor); when the context
    // moves to another stream, that stream waits for the old one first
    if (have_stream_ && s != stream_) {  // an event in the old stream's order, the new stream behind it: the host does not wait
      YMK_HIP(hipEventRecord(moved_, stream_));
      YMK_HIP(hipStreamWaitEvent(s, moved_, 0));
    }
    stream_ = s;
    have_stream_ = true;
    const size_t pixels = (size_t)(k.in_bytes / 4 - k.C) / k.in_ld + 1;
    const size_t items = pixels * (size_t)(k.C / 4);
    const int blocks = (int)std::min<size_t>(1024, (items + 1023) / 1024);
    unsigned* cur = slots_ + parity_ * AMAX_REC_WORDS;
    hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(256), 0, s, k.in, items, k.C / 4, k.in_ld, cur, slots_ + (parity_ ^ 1) * AMAX_REC_WORDS);
    parity_ ^= 1;
    return cur;
  }

 private:
  struct Key {
    const float* w;
    int code;
    bool operator==(const Key& o) const { return w == o.w && code == o.code; }
  };
  struct KeyHash {
    size_t operator()(const Key& k) const { return std::hash<const void*>()(k.w) * 31 + (size_t)k.code; }
  };
  void* alloc(size_t bytes) {
    void* q = dev_malloc(bytes);
    allocs_.push_back(q);
    return q;
  }
  Panels build(hipStream_t s, const ConvW& w, int code) {
    const int ns = code == 3 ? 3 : 2;
    const size_t rows = (size_t)((w.cout + 255) / 256 * 256), real = (size_t)((w.cout + 127) / 128 * 128), n = real * w.kpad;
    Panels pn;
    pn.planes = alloc(rows * w.kpad * ns * 2);  // rows padded to 256: the 256-wide tile reads whole tiles
    YMK_HIP(hipMemsetAsync(pn.planes, 0, rows * w.kpad * ns * 2, s));
    if (code == SPLIT_F16X2 || code == SPLIT_F16X2_PERM) {
      pn.scale = reinterpret_cast<float*>(alloc(real * sizeof(float)));
      hipLaunchKernelGGL(k_split_panel_f16, dim3((unsigned)real), dim3(256), 0, s, w.w, reinterpret_cast<unsigned short*>(pn.planes), w.kpad,
                         w.scale, w.cout, pn.scale, w.kh * w.kw, w.ctiles, code == SPLIT_F16X2_PERM ? 1 : 0);
    } else {
      const int blocks = (int)((n + 255) / 256);
      if (ns == 2) hipLaunchKernelGGL(k_split_panel<2>, dim3(blocks), dim3(256), 0, s, w.w, reinterpret_cast<unsigned short*>(pn.planes), n, w.kpad);
      else hipLaunchKernelGGL(k_split_panel<3>, dim3(blocks), dim3(256), 0, s, w.w, reinterpret_cast<unsigned short*>(pn.planes), n, w.kpad);
    }
    YMK_HIP(hipGetLastError());
    return pn;
  }
  // the two max|x| words of launches whose input came without a record
  void ensure_slots(hipStream_t s) {
    if (slots_) return;
    slots_ = reinterpret_cast<unsigned*>(alloc(2 * AMAX_REC_WORDS * sizeof(unsigned)));
    YMK_HIP(hipEventCreateWithFlags(&moved_, hipEventDisableTiming));
    if (debug_hazard_null_memset()) {
      // the round-5 form, reachable only through YMK_DEBUG_HAZARD_NULL_MEMSET (tools/stress_call.py): the fill goes to the NULL
      // stream, which the callers' non-blocking streams do not order with - it can land after the first k_absmax has written
      // the word, and the model's first split launch then scales its operands for max|x| = 0
      YMK_HIP(hipMemset(slots_, 0, 2 * AMAX_REC_WORDS * sizeof(unsigned)));
      return;
    }
    // ON stream s, ahead of the first k_absmax that will use the words
    YMK_HIP(hipMemsetAsync(slots_, 0, 2 * AMAX_REC_WORDS * sizeof(unsigned), s));
  }
  std::unordered_map<Key, Panels, KeyHash> cache_;
  std::vector<void*> allocs_;
  unsigned* slots_ = nullptr;
  int parity_ = 0;
  hipStream_t stream_ = nullptr;
  bool have_stream_ = false;
  hipEvent_t moved_ = nullptr;
};

SplitCtx* SplitCtxOwner::get() {
  if (!p_) p_ = new SplitCtx();
  return p_;
}
SplitCtxOwner::~SplitCtxOwner() { delete p_; }

void Model::prebuild_split() {
  if (debug_lazy_split()) return;
  SplitCtx* ctx = split_ctx.get();
  int code;
  {
    ConvSplitScope scope(conv_split(), ctx, SPLIT_MODEL_DEFAULT);
    code = conv_effective_split();
  }
  ctx->prebuild(nullptr, pool.convs(), pool.perm_convs(), code);
  const int enc = (int)param("conv_split_encoder", -1);  // the recogniser's evaluation switch: a second precision for the ViT blocks
  if (enc > 0 && enc != code) ctx->prebuild(nullptr, pool.convs(), pool.perm_convs(), enc);
  YMK_HIP(hipStreamSynchronize(nullptr));
}

// amax_check mode: the launch's record against a pass over its input (counters above).  Level 2 also waits for the answer
// and names the launch whose record lies below the truth on stderr (a debugging aid: it serialises the stream).
static void amax_verify(hipStream_t s, const ConvK& k, SplitCtx* ctx) {
  if (!g_amax_counters) {
    YMK_HIP(hipMalloc((void**)&g_amax_counters, 4 * sizeof(unsigned long long)));
    YMK_HIP(hipMemset(g_amax_counters, 0, 4 * sizeof(unsigned long long)));
  }
  const unsigned* truth = ctx->absmax(s, k);
  hipLaunchKernelGGL(k_amax_verify, dim3(1), dim3(64), 0, s, k.amax, truth, g_amax_counters);
  if (g_amax_check.load(std::memory_order_relaxed) >= 2) {
    unsigned got[AMAX_REC_WORDS], want[AMAX_REC_WORDS];
    YMK_HIP(hipStreamSynchronize(s));
    YMK_HIP(hipMemcpy(got, k.amax, sizeof(got), hipMemcpyDeviceToHost));
    YMK_HIP(hipMemcpy(want, truth, sizeof(want), hipMemcpyDeviceToHost));
    unsigned g = 0, w = 0;
    for (int i = 0; i < AMAX_LINES; ++i) {
      g = std::max(g, got[i * AMAX_LINE_WORDS]);
      w = std::max(w, want[i * AMAX_LINE_WORDS]);
    }
    if (g < w) {
      float gf, wf;
      std::memcpy(&gf, &g, 4);
      std::memcpy(&wf, &w, 4);
      std::fprintf(stderr, "amax_check: record BELOW the truth: %g < %g  M=%d C=%d ld=%d Cout=%d k=%dx%d stride=%d H=%d W=%d act=%d res=%d rec=%p in=%p\n", gf, wf,
                   k.M, k.C, k.in_ld, k.Cout, k.KH, k.KW, k.stride, k.H, k.W, k.act, k.res ? 1 : 0, (const void*)k.amax, (const void*)k.in);
    }
  }
}

template <int FMT, int BM, int BN, int WM, int WN, int NS, int KS = 1, int PF = 1, bool OPL = false>
static void launch_split(hipStream_t s, ConvK& k, const void* wsplit, SplitCtx* ctx = nullptr) {
  if (KS > 1 && (k.Kpad >> 5) % KS != 0) {  // a stage holds KS whole K tiles: odd tile counts take the one-tile form
    launch_split<FMT, BM, BN, WM, WN, NS, 1, PF, OPL>(s, k, wsplit, ctx);
    return;
  }
  const int mt = (k.M + BM - 1) / BM, nt = (k.Cout + BN - 1) / BN;
  k.ntiles_n = nt;
  auto* e = conv_prof_open(s, k, BM, BN, mt * nt, (FMT ? 160 : 10 * NS));
  if (FMT == 1 && k.amax == nullptr) {
    k.amax = ctx->absmax(s, k);  // no record from the producer: a pass over the input (inside the timed span)
  } else if (FMT == 1 && g_amax_check.load(std::memory_order_relaxed)) {
    amax_verify(s, k, ctx);
  }
  hipLaunchKernelGGL((conv_igemm_split<BM, BN, WM, WN, NS, KS, PF, FMT, OPL>), dim3(mt * nt), dim3(64 * WM * WN), 0, s, k,
                     reinterpret_cast<const uint4*>(wsplit));
  if (e) YMK_HIP(hipEventRecord(e->second, s));
}

// tile shapes, ymk_debug_option("conv_split_tile", v): 0 = the measured best per format (profiles/
// r03_conv_sweep_bf16_split*.txt); for A/B runs
//   1 = 128 x 64, 8 waves          2 = 256 x 128, 16 waves        3 = 128 x 128, 16 waves        4 = 128 x 128, 8 waves
//   5 = 128 x 128, 16 waves, 64-k stages              6 = the same, loads two stages ahead
//   7 = 128 x 128, 8 waves, 64-k stages, two ahead    8 = 256 x 128, 16 waves, 32-k stages, two ahead
//   9 = 128 x 128, 8 waves, 32-k stages, two ahead   10 = 128 x 128, 16 waves, 32-k stages, two ahead
//   11 = 256 x 256, 16 waves (64 x 64 per wave), two planes
//   12 / 13 / 14 = 128 x 128 x 16 waves / 256 x 128 x 16 waves / 128 x 128 x 8 waves with the stores threaded through the MFMAs
// (bf16 only; the fp16 form keeps the shapes that won there: 0 / 3 = 128 x 128 x 16 waves, 1 = 128 x 64, 2 = 256 x 128, 11)
static std::atomic<int> g_split_tile{0};
// ... or, for the calling thread only, a ConvSplitTileScope (single-operator entry points: the process-wide word would
// reroute forwards that other threads have in flight)
static thread_local int t_split_tile = 0;
static int split_tile_now() { return t_split_tile != 0 ? t_split_tile : g_split_tile.load(std::memory_order_relaxed); }
ConvSplitTileScope::ConvSplitTileScope(int tile) : prev_(t_split_tile) { t_split_tile = tile; }
ConvSplitTileScope::~ConvSplitTileScope() { t_split_tile = prev_; }
// ymk_debug_option("astat", v): 1 (default) = the short-K pointwise layers the A-stationary kernel measured ahead on take it;
// 0 = none does (the round-4 routing, for A/B runs); "conv_split_tile" 30 forces it for every launch it can run
static std::atomic<int> g_astat{1};
// ymk_debug_option("act_planes", v): 1 (default) = the tensor between a bottleneck's 1 x 1 reduction and its 3 x 3 convolution
// lives in HBM as the two fp16 planes the 3 x 3 multiplies with (written by the reduction's epilogue under a bound, read by
// LDS-DMA without any conversion: ymk_conv_dma.hip APL) wherever conv_planes_pair_ok allows; 0 = fp32 activations everywhere
static std::atomic<int> g_act_planes{1};
// ymk_debug_option("rowmax_tile", v): the kernel of a row-max launch (the greedy loop's vocabulary head) where 128 x 128 tiles fill
// the chip: 0 = the A-stationary kernel, one column block per block, where it runs (K <= 192: ymk_conv_astat.hip astat_rowmax),
// else the 128 x 64 x 8-wave tile; for A/B runs 1 = 128 x 128 x 16 waves, 2 = 128 x 128 x 8 waves, 3 = 256 x 128 x 16 waves, 4 = the
// A-stationary kernel with its column blocks dealt to groups, 6 = 128 x 64 always (rounds 3-5) - the same (max, column) pairs per
// 64-column sub-tile from each (ymk_conv_kernel.h, EPI_ROWMAX).  profiles/r06_rowmax_head_tiles.md: 35.2 us per step at 1234 rows
// on 128 x 64, 36.3 / 44.4 / 55.5 on the wider tiles, 38.1 on (4), 28.2 on (0)
static std::atomic<int> g_rowmax_tile{0};
// launch counters since the process started (ymk_stat; tests assert that a route was really taken)
static std::atomic<long long> g_n_astat{0}, g_n_ln_fused{0}, g_n_planes_read{0}, g_n_planes_written{0}, g_n_rowmax_wide{0};
long long vit_mlp_fused_launches();
bool conv_split_stat(const std::string& key, long long* value) {
  if (key == "astat_launches") *value = g_n_astat.load();
  else if (key == "ln_fused_launches") *value = g_n_ln_fused.load();
  else if (key == "planes_read_launches") *value = g_n_planes_read.load();
  else if (key == "planes_written_launches") *value = g_n_planes_written.load();
  else if (key == "rowmax_wide_launches") *value = g_n_rowmax_wide.load();
  else if (key == "mlp_fused_launches") *value = vit_mlp_fused_launches();
  else return false;
  return true;
}
void astat_rowmax_dealt(int on);  // ymk_conv_astat.hip
bool conv_split_debug_option(const std::string& key, int value) {
  if (key == "conv_split_tile") g_split_tile = value;
  else if (key == "amax_check") g_amax_check = value;
  else if (key == "astat") g_astat = value;
  else if (key == "act_planes") g_act_planes = value;
  else if (key == "rowmax_tile") {
    g_rowmax_tile = value;
    astat_rowmax_dealt(value == 4);
  }
  else return false;
  return true;
}

template <int NS>
static void dispatch_bf16(hipStream_t s, ConvK& k, const void* ws, int tile, bool narrow) {
  if (narrow) {
    launch_split<0, 128, 64, 4, 2, NS>(s, k, ws);
    return;
  }
  switch (tile) {
    case 2: launch_split<0, 256, 128, 4, 4, NS>(s, k, ws); break;
    case 4: launch_split<0, 128, 128, 4, 2, NS>(s, k, ws); break;
    // 64-k stages of three planes do not fit the CU's LDS (205 KB): those selectors keep 32-k stages there
    case 5: launch_split<0, 128, 128, 4, 4, NS, NS == 2 ? 2 : 1, 1>(s, k, ws); break;
    case 6: launch_split<0, 128, 128, 4, 4, NS, NS == 2 ? 2 : 1, 2>(s, k, ws); break;
    case 7: launch_split<0, 128, 128, 4, 2, NS, NS == 2 ? 2 : 1, 2>(s, k, ws); break;
    case 8: launch_split<0, 256, 128, 4, 4, NS, 1, 2>(s, k, ws); break;
    case 9: launch_split<0, 128, 128, 4, 2, NS, 1, 2>(s, k, ws); break;
    case 10: launch_split<0, 128, 128, 4, 4, NS, 1, 2>(s, k, ws); break;
    case 12: launch_split<0, 128, 128, 4, 4, NS, 1, 3>(s, k, ws); break;
    case 13: launch_split<0, 256, 128, 4, 4, NS, 1, 3>(s, k, ws); break;
    case 14: launch_split<0, 128, 128, 4, 2, NS, 1, 3>(s, k, ws); break;
    case 11:  // 256 x 256, 16 waves of 64 x 64 (two planes only: three do not fit the LDS); Cout < 256 keeps 128-wide tiles
      if (NS == 2 && k.Cout >= 256) launch_split<0, 256, 256, 4, 4, 2>(s, k, ws);
      else launch_split<0, 128, 128, 4, 4, NS>(s, k, ws);
      break;
    default: launch_split<0, 128, 128, 4, 4, NS>(s, k, ws); break;
  }
}

static void dispatch_f16(hipStream_t s, ConvK& k, const void* ws, int tile, bool narrow, SplitCtx* ctx) {
  if (k.out_planes) {  // plane-writing epilogues exist for the two shapes the automatic choice uses
    if (narrow) launch_split<1, 128, 64, 4, 2, 2, 1, 1, true>(s, k, ws, ctx);
    else launch_split<1, 128, 128, 4, 4, 2, 1, 1, true>(s, k, ws, ctx);
    return;
  }
  if (narrow) {
    launch_split<1, 128, 64, 4, 2, 2>(s, k, ws, ctx);
    return;
  }
  switch (tile) {
    case 2: launch_split<1, 256, 128, 4, 4, 2>(s, k, ws, ctx); break;
    case 4: launch_split<1, 128, 128, 4, 2, 2>(s, k, ws, ctx); break;
    case 11:
      if (k.Cout >= 256) launch_split<1, 256, 256, 4, 4, 2>(s, k, ws, ctx);
      else launch_split<1, 128, 128, 4, 4, 2>(s, k, ws, ctx);
      break;
    default: launch_split<1, 128, 128, 4, 4, 2>(s, k, ws, ctx); break;
  }
}

bool conv2d_f16_dma(hipStream_t s, ConvK& k, const void* wsplit, size_t w_bytes, bool narrow, int rows);  // ymk_conv_dma.hip

// ---- which kernel an fp16-split launch runs on: ONE rule, asked by conv2d_split for the launch itself and by
// conv_planes_pair_ok (ymk_conv.hip) for a producer / consumer pair before either is launched
//   none: not a launch that fills the chip (>= 256 tiles of 128 x 128, or - few - of 128 x 64): the exact fp32 paths keep it
//   astat (ymk_conv_astat.hip): pointwise layers with K <= 192 and more than 64 output channels - the ViT blocks' qkv / proj /
//     fc1 and the final vocabulary head (K = 192), the ResNet expands 64 -> 256 / 128 -> 512 with their residual: A rows are
//     fetched and cut into planes once per 128 rows instead of once per tile (profiles/r05_conv_astat_timing.jsonl: qkv 371 us
//     against 514, 64 -> 256 + residual 403 against 475, the 124 634 x 7119 head 1.76 ms against 3.22).  At K = 256 the A
//     planes leave registers for 64-column blocks only, and the per-layer table of the serial pass
//     (profiles/r05_conv_two_roof_by_layer*.md) has that form BEHIND the register-staged kernel wherever a residual is read
//     (256 -> 1024 + residual 2.67 ms against 2.27): K = 256 stays where it was.  A launch with a LayerNorm to fuse has no
//     other kernel to go to: taken there or refused.  It neither reads nor writes planes.
//   dma (ymk_conv_dma.hip; 128-row tiles, two or three blocks per CU) where it measured ahead of the register-staged kernel
//     (profiles/r04_conv_sweep_f16_lds_dma.txt): every k x k layer (+2..17 %), long-K 1 x 1 reductions (+3..7 %) and the
//     64-column 1 x 1 layers (+1..6 %)
//   staged (conv_igemm_split): the rest - wide 1 x 1 layers with K > 256 (the ViT fc2, the 512 -> 128 reductions)
enum SplitRoute : int { ROUTE_NONE = 0, ROUTE_ASTAT = 1, ROUTE_DMA = 2, ROUTE_STAGED = 3 };
struct RouteQuery {
  long M;
  int cout, kpad, taps;
  bool rowmax, row_group, astat_can, ln, planes_io;
};
static SplitRoute route_f16(const RouteQuery& q, int& tile, bool& narrow) {
  const long mt128 = (q.M + 127) / 128;
  const long blocks128 = mt128 * ((q.cout + 127) / 128), blocks64 = mt128 * ((q.cout + 63) / 64);
  tile = split_tile_now();
  const bool auto_tile = tile == 0;
  const bool astat_forced = tile == 30;  // tests: the A-stationary kernel at any size it can run
  bool few = false;
  if (blocks128 < 256 && !astat_forced) {
    if (blocks64 < 256) return ROUTE_NONE;
    few = true;
  }
  if (tile == 0) tile = 3;
  const int rowmax_tile = q.rowmax && auto_tile && !few ? g_rowmax_tile.load(std::memory_order_relaxed) : 0;
  if (rowmax_tile == 2) tile = 4;
  if (rowmax_tile == 3) tile = 2;
  narrow = q.cout <= 64 || tile == 1 || (q.rowmax && (rowmax_tile < 1 || rowmax_tile > 3)) || few;
  if ((!q.rowmax || ((rowmax_tile == 0 || rowmax_tile == 4) && auto_tile)) && q.astat_can && !q.planes_io &&
      (astat_forced || q.ln || (auto_tile && g_astat.load(std::memory_order_relaxed) != 0 && q.cout > 64 && !few && q.kpad <= 192)))
    return ROUTE_ASTAT;
  if (q.ln) return ROUTE_NONE;
  if (tile == 30) {
    if (blocks128 < 256) return ROUTE_NONE;  // forced, but not a launch the kernel runs: as without the option
    tile = 3;
  }
  const bool dma_shape = q.taps > 1 || (q.kpad >= 1024 && q.cout <= 512) || q.cout <= 64;
  const bool dma_fills = blocks128 >= 256 || (q.cout <= 64 && mt128 >= 256);
  if (!q.rowmax && !q.row_group && (tile == 20 || tile == 21 || (auto_tile && dma_shape && dma_fills))) return ROUTE_DMA;
  if (tile == 20 || tile == 21) tile = 3;
  return ROUTE_STAGED;
}

// for conv_planes_pair_ok: the route of a plain-store fp16-split launch that reads or writes planes; `auto_tile`: no tile is forced
int conv_split_route_with_planes(long M, int cout, int kpad, int taps, bool* auto_tile) {
  int tile = 0;
  bool narrow = false;
  *auto_tile = split_tile_now() == 0 && g_act_planes.load(std::memory_order_relaxed) != 0;
  return (int)route_f16(RouteQuery{M, cout, kpad, taps, false, false, false, false, true}, tile, narrow);
}

bool conv2d_split(hipStream_t s, ConvK& k, const ConvW& w, int code, SplitCtx* ctx) {
  if (w.mode != 0 || ctx == nullptr || (code != 2 && code != 3 && code != SPLIT_F16X2)) return false;  // 4-channel stems keep the fp32 kernel
  const bool planes_io = k.in_planes != 0 || k.out_planes != 0;
  if ((k.ln_g != nullptr || planes_io) && code != SPLIT_F16X2) return false;  // fused LayerNorm / planes in HBM: fp16 form only
  const bool rowmax = k.epi == EPI_ROWMAX;  // (max, column) per 64-column tile: the 128 x 64 tile
  const size_t w_bytes_f16 = (size_t)((w.cout + 255) / 256 * 256) * w.kpad * 2 * 2;
  int tile = 0;
  bool narrow = false;
  SplitRoute route = ROUTE_STAGED;
  if (code == SPLIT_F16X2) {
    route = route_f16(RouteQuery{k.M, w.cout, k.Kpad, k.KH * k.KW, rowmax, k.row_group != nullptr, conv2d_f16_astat_can(k, w_bytes_f16),
                                 k.ln_g != nullptr, planes_io}, tile, narrow);
  } else {  // bf16 evaluation forms: the register-staged kernel, for launches that fill the chip
    const long mt128 = (k.M + 127) / 128;
    const long blocks128 = mt128 * ((w.cout + 127) / 128), blocks64 = mt128 * ((w.cout + 63) / 64);
    tile = split_tile_now();
    if (tile == 0) tile = code == 3 ? 2 : 3;
    if (tile == 20 || tile == 21 || tile == 30) tile = 3;
    if (blocks128 < 256 && blocks64 < 256) return false;
    narrow = w.cout <= 64 || tile == 1 || rowmax || blocks128 < 256;
  }
  if (route == ROUTE_NONE) return false;
  YMK_CHECK(!k.in_planes || (route == ROUTE_DMA && tile != 20), "a tensor stored as fp16 planes reached a kernel that cannot read it");
  YMK_CHECK(!k.out_planes || (route == ROUTE_DMA && tile != 20) || (route == ROUTE_STAGED && (narrow || tile == 3)),
            "fp16 planes asked of a kernel that cannot write them");
  const SplitCtx::Panels& pn = ctx->panels(s, w, code);
  if (k.in_planes) ++g_n_planes_read;
  if (k.out_planes) ++g_n_planes_written;
  if (route == ROUTE_ASTAT) {
    k.scale = pn.scale;
    const int bn = conv2d_f16_astat_columns(k);
    auto* e = conv_prof_open(s, k, 128, bn, ((k.M + 127) / 128) * ((k.Cout + bn - 1) / bn), 162);
    if (k.amax == nullptr) k.amax = ctx->absmax(s, k);
    else if (g_amax_check.load(std::memory_order_relaxed) && k.ln_g == nullptr) amax_verify(s, k, ctx);
    const bool taken = conv2d_f16_astat(s, k, pn.planes, w_bytes_f16);
    if (e) YMK_HIP(hipEventRecord(e->second, s));
    YMK_HIP(hipGetLastError());
    YMK_CHECK(taken, "conv_f16_astat refused a launch it had accepted");
    ++g_n_astat;
    if (rowmax) ++g_n_rowmax_wide;
    if (k.ln_g != nullptr) ++g_n_ln_fused;
    return true;
  }
  if (route == ROUTE_DMA) {
    k.scale = pn.scale;
    const bool nar = w.cout <= 64;
    const int rows = tile == 20 ? 256 : 128;
    k.ntiles_n = (k.Cout + (nar ? 64 : 128) - 1) / (nar ? 64 : 128);
    auto* e = conv_prof_open(s, k, rows, nar ? 64 : 128, ((k.M + rows - 1) / rows) * k.ntiles_n, 161);
    YMK_CHECK(!k.in_planes || k.amax != nullptr, "a tensor of fp16 planes without the record that holds its scale");
    if (k.amax == nullptr) k.amax = ctx->absmax(s, k);
    else if (g_amax_check.load(std::memory_order_relaxed) && !k.in_planes) amax_verify(s, k, ctx);
    const bool taken = conv2d_f16_dma(s, k, pn.planes, w_bytes_f16, nar, rows);
    if (e) YMK_HIP(hipEventRecord(e->second, s));
    YMK_HIP(hipGetLastError());
    YMK_CHECK(taken, "conv_f16_dma refused a launch");
    return true;
  }
  if (code == SPLIT_F16X2) {
    k.scale = pn.scale;
    if (rowmax && !narrow) ++g_n_rowmax_wide;
    dispatch_f16(s, k, pn.planes, tile, narrow, ctx);
  } else if (code == 2) {
    dispatch_bf16<2>(s, k, pn.planes, tile, narrow);
  } else {
    dispatch_bf16<3>(s, k, pn.planes, tile, narrow);
  }
  YMK_HIP(hipGetLastError());
  return true;
}

// ---- the fused ViT MLP (ymk_vit_mlp.hip): planes of fc1 (standard) and fc2 (permuted), the bounds, the launch
struct MlpK {  // as in ymk_vit_mlp.hip
  const float* x;
  float* out;
  int M, ld;
  const float *ln_g, *ln_b;
  float ln_eps, ln_bound;
  const uint4* w1;
  unsigned w1_bytes;
  const float *s1, *b1;
  const uint4* w2;
  unsigned w2_bytes;
  const float *s2, *b2;
  float g_bound;
};
bool vit_mlp_f16_launch(hipStream_t s, const MlpK& k, int D, int F);
std::pair<hipEvent_t, hipEvent_t>* conv_prof_open_raw(hipStream_t s, const char* desc, double flops, double bytes, double products);

static std::atomic<long long> g_n_mlp_fused{0};
long long vit_mlp_fused_launches() { return g_n_mlp_fused.load(); }

bool vit_mlp_split_launch(hipStream_t s, SplitCtx* ctx, float* x, int M, int ld, const float* ln_g, const float* ln_b, float ln_eps,
                          float ln_bound, const ConvW& fc1, const ConvW& fc2) {
  const int D = fc1.cin, F = fc1.cout;
  if (ctx == nullptr || fc1.mode != 0 || fc2.mode != 0 || fc1.kh * fc1.kw != 1 || fc2.kh * fc2.kw != 1) return false;
  if (fc2.cin != F || fc2.cout != D || fc1.kpad != D || fc2.kpad != F || fc1.scale != nullptr || fc2.scale != nullptr) return false;
  if (fc1.bias == nullptr || fc2.bias == nullptr || D != 192 || F != 768) return false;
  if ((M + 127) / 128 < 256) return false;  // one block per CU: fewer blocks than CUs leave the layer to the GEMM kernels
  const SplitCtx::Panels& p1 = ctx->panels(s, fc1, SPLIT_F16X2);
  const SplitCtx::Panels& p2 = ctx->panels(s, fc2, SPLIT_F16X2_PERM);
  MlpK k{};
  k.x = x;
  k.out = x;
  k.M = M;
  k.ld = ld;
  k.ln_g = ln_g;
  k.ln_b = ln_b;
  k.ln_eps = ln_eps;
  k.ln_bound = ln_bound;
  k.w1 = reinterpret_cast<const uint4*>(p1.planes);
  k.w1_bytes = (unsigned)((size_t)((F + 255) / 256 * 256) * fc1.kpad * 4);
  k.s1 = p1.scale;
  k.b1 = fc1.bias;
  k.w2 = reinterpret_cast<const uint4*>(p2.planes);
  k.w2_bytes = (unsigned)((size_t)((D + 255) / 256 * 256) * fc2.kpad * 4);
  k.s2 = p2.scale;
  k.b2 = fc2.bias;
  k.g_bound = fc1.pl_a * ln_bound + fc1.pl_b;  // |GELU(v)| <= |v| <= pl_a max|LayerNorm output| + pl_b
  char desc[160];
  // (the by-layer tools parse this line: the fused layer shows as M x D -> D with a residual on a 128 x D tile, ksplit 163)
  snprintf(desc, sizeof desc, "M=%7d Cin=%4d Cout=%4d k=1x1 s=1 d=1 res=1 tile=128x%d ksplit=163 grid=%d", M, D, D, D, (M + 127) / 128);
  // algorithmic work of the fused layer: both products; bytes: the rows in and out, the residual read, the two weight matrices
  auto* e = conv_prof_open_raw(s, desc, 4.0 * (double)M * D * F, 4.0 * (3.0 * (double)M * D + 2.0 * (double)D * F), 3.0);
  const bool taken = vit_mlp_f16_launch(s, k, D, F);
  if (e) YMK_HIP(hipEventRecord(e->second, s));
  YMK_HIP(hipGetLastError());
  if (taken) ++g_n_mlp_fused;
  return taken;
}

}  // namespace ymk
