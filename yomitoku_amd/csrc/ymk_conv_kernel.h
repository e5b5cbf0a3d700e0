// Device-side pieces shared by the implicit-GEMM convolution kernels (ymk_conv.hip: exact fp32 MFMA; ymk_conv_split.hip:
// bf16 / fp16 split operands, fp32 accumulate): the launch record, the row predicate and the epilogue.
#pragma once
#include <utility>

#include "ymk_common.h"

namespace ymk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvK {
  const float* in;
  const float* w;
  const float* scale;
  const float* bias;
  const float* res;
  float* out;
  int H, W, C, in_ld;
  int OH, OW, Cout, out_ld, res_ld, res_post;
  int KH, KW, stride, stride_w, pad, dil;
  int Kpad, ctiles, mode;
  int M;        // n*oh*ow
  int act, epi;
  int ntiles_n;
  unsigned in_bytes;  // extent of the input view (buffer descriptor range)
  int vec;      // 1: Cout, leading dims and pointers allow 16 B epilogue accesses
  const int* row_group;   // optional (ConvArgs): row m -> group id ...
  const int* group_open;  // ... and the per-group "still needed" word; M tiles without a needed row return at once
  int fast;     // ymk_debug_option("conv_fast"), default 27: bit 0 pointwise index shortcut, bit 1 residual prefetch,
                // bit 2 direct epilogue for every plain store (A/B runs; slower than the staged one where 16-byte stores
                // are possible), bit 3 swizzled K tiles / three 128 x 64 blocks per CU, bit 4 direct epilogue for the
                // launches that cannot store 16 bytes per lane (ragged Cout: the 7119-wide vocabulary head)
  const unsigned* amax;  // fp16-split kernels only: max|x| record of the input view (ymk_common.h; ymk_conv_split.hip)
  unsigned* amax_out;    // every kernel: record to fold max|y| of the stored outputs into, or null
  // conv_f16_astat only (ymk_conv_astat.hip): LayerNorm(gamma, beta, eps) over the C channels of every input row, applied on
  // the way into the operand planes - the launch then computes act(LayerNorm(x) . W^T ...) without the normalised tensor
  // ever existing in memory; null = plain input.  No other kernel looks at these: conv2d_split refuses what it cannot fuse.
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  // fp16 PLANES in HBM (Tensor::planes): in_planes - the input view holds planes (conv_f16_dma<.., APL> only); out_planes - the
  // epilogue writes planes scaled from the bound pl_a max|x_in| + pl_b and leaves THAT bound in amax_out (epilogue_tile<.., PL>)
  int in_planes, out_planes;
  float pl_a, pl_b;
};

// true when some row of the block's M tile [m0, m0 + BM) is still needed (or no row predicate was given); block-uniform
template <int BM, int NT>
__device__ __forceinline__ bool tile_needed(const ConvK& p, int m0, int t) {
  if (!p.row_group) return true;
  int live = 0;
  for (int r = t; r < BM; r += NT) {
    const int m = m0 + r;
    if (m < p.M && p.group_open[p.row_group[m]] != 0) live = 1;
  }
  return __syncthreads_or(live) != 0;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SILU: return v / (1.f + expf(-v));
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case ACT_GELU: return gelu_f32(v);
    default: return v;
  }
}

// max|x| bits -> {sa, 1 / sa}, sa the power of two that puts max|x| into [2^14, 2^15); both kept normal whatever the input
// (the one definition behind f16_scales / dma_f16_scales / astat_scales of the three fp16 kernels)
__device__ __forceinline__ float2 f16_plane_scales(unsigned amax_bits) {
  int e = (int)(amax_bits >> 23);  // biased exponent, 0 .. 255
  e = e < 27 ? 27 : (e > 227 ? 227 : e);
  float2 r;
  r.x = __uint_as_float((unsigned)(268 - e) << 23);  // 2^(14 - (e - 127))
  r.y = __uint_as_float((unsigned)(e - 14) << 23);
  return r;
}

constexpr int LDK = 36;  // padded K-tile row (floats)

// ---- K-tile layout in LDS (conv_igemm OPT bit 1)
constexpr int lds_row(int opt) { return (opt & 2) ? 32 : LDK; }
constexpr int blocks_per_cu(int bm, int bn, int opt) {
  const int bytes = 2 * (bm + bn) * lds_row(opt) * 4;
  return 3 * bytes <= 160 * 1024 ? 3 : 2 * bytes <= 160 * 1024 ? 2 : 1;
}
// float offset of 16-byte slot `slot` (0..7) of K-tile row `row`.  Swizzled form: a pair of rows is one 256-byte line
// (all 64 banks), slot index (row parity, slot) XOR (pair index mod 8): the 16 rows a ds_read_b128 phase touches at one
// slot land in 16 distinct slots of the line, and so do the 2 rows x 8 slots of a ds_write_b128 phase
template <int OPT>
__device__ __forceinline__ int lds_slot(int row, int slot) {
  if (OPT & 2) return (row >> 1) * 64 + ((((row & 1) << 3) | slot) ^ ((row >> 1) & 7)) * 4;
  return row * LDK + slot * 4;
}

// A-operand gathers go through a raw buffer descriptor: padding taps and tail rows use an offset
// beyond num_records, for which the hardware returns zeros - no branch, no select after the load,
// so the loaded registers flow straight to ds_write and their wait can sit after the MFMAs.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in SSA registers
constexpr unsigned OOB_OFFSET = 0xFFFFFFF0u;
constexpr unsigned SPLITK_OOB_ROW = 0xC0000000u;  // conv_splitk: row base of a masked pixel (+ channel bytes, no wrap)

// ---- epilogue over a block tile staged in LDS as Cs[BM][BN + 4]: scale/bias, residual (before or after
// the activation), activation, plain or 2x2 pixel-shuffle store; NT threads, 16 B per lane, full rows coalesced.
// rows a thread stores in epilogue_tile (row r0 + RPP * i, i < NR)
template <int BM, int BN, int NT>
struct EpiRows {
  static constexpr int TPR = BN / 4, RPP = NT / TPR, NR = (BM + RPP - 1) / RPP;
};

// the residual values of the thread's epilogue rows, fetched ahead of the last K tile's MFMAs (short-K layers: the
// epilogue is a large share of the block's life, and its only long-latency operation is this read)
template <int BM, int BN, int NT>
__device__ __forceinline__ void prefetch_residual(const ConvK& p, int m0, int n0, int t, float4* rr) {
  using E = EpiRows<BM, BN, NT>;
  const int c4 = t % E::TPR, r0 = t / E::TPR;
  const int co = n0 + c4 * 4;
#pragma unroll
  for (int i = 0; i < E::NR; ++i) {
    const int m = m0 + r0 + E::RPP * i;
    rr[i] = (co < p.Cout && m < p.M && r0 + E::RPP * i < BM) ? *reinterpret_cast<const float4*>(p.res + (size_t)m * p.res_ld + co)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// PL (fp16 kernels only): the outputs leave as the two fp16 planes of Tensor::planes instead of fp32 - 4 high halves and 4 low
// halves per thread and row (two 8-byte stores where the fp32 form has one 16-byte store), scaled by the power of two of the
// BOUND pl_a max|x_in| + pl_b, which is also what the output's record receives: the consumer derives the same scale from it.
// Plain stores with 16-byte channel groups only (the launcher checks: EPI_STORE, vec, no residual, out_ld == Cout % 32 == 0).
template <int BM, int BN, int NT, bool PRE = false, int UNROLL = 4, bool PL = false>
__device__ __forceinline__ void epilogue_tile(const ConvK& p, const float* Cs, int m0, int n0, int t, const float4* pre = nullptr) {
  constexpr int LDC = BN + 4;
  if constexpr (PL) {
    typedef _Float16 ep_h2 __attribute__((ext_vector_type(2)));
    typedef float ep_f2 __attribute__((ext_vector_type(2)));
    constexpr int TPR = BN / 4, RPP = NT / TPR;
    const int c4 = t % TPR, r0 = t / TPR;
    const int co = n0 + c4 * 4;
    const unsigned bound_bits = __float_as_uint(fmaf(__uint_as_float(amax_read(p.amax, t)), p.pl_a, p.pl_b));  // every lane: shuffles inside
    const float so = f16_plane_scales(bound_bits).x;
    if (co < p.Cout) {
      const float4 sc = p.scale ? *reinterpret_cast<const float4*>(p.scale + co) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 bi = p.bias ? *reinterpret_cast<const float4*>(p.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
      char* const obase = reinterpret_cast<char*>(p.out) + (size_t)(co >> 5) * 128 + (size_t)(co & 31) * 2;
#pragma unroll 4
      for (int i = 0; i < EpiRows<BM, BN, NT>::NR; ++i) {
        const int row = r0 + RPP * i;
        const int m = m0 + row;
        if (row >= BM || m >= p.M) break;
        float4 v = *reinterpret_cast<const float4*>(Cs + row * LDC + c4 * 4);
        v.x = apply_act(v.x * sc.x + bi.x, p.act);
        v.y = apply_act(v.y * sc.y + bi.y, p.act);
        v.z = apply_act(v.z * sc.z + bi.z, p.act);
        v.w = apply_act(v.w * sc.w + bi.w, p.act);
        ep_f2 a = {v.x * so, v.y * so}, b = {v.z * so, v.w * so};
        const ep_h2 ha = __builtin_convertvector(a, ep_h2), hb = __builtin_convertvector(b, ep_h2);
        a -= __builtin_convertvector(ha, ep_f2);  // exact
        b -= __builtin_convertvector(hb, ep_f2);
        const ep_h2 la = __builtin_convertvector(a, ep_h2), lb = __builtin_convertvector(b, ep_h2);
        char* o = obase + (size_t)m * p.out_ld * 4;
        *reinterpret_cast<uint2*>(o) = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
        *reinterpret_cast<uint2*>(o + 64) = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
      }
    }
    if (p.amax_out) amax_commit(p.amax_out, bound_bits, t);
    return;
  }
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int TPR = BN / 4;        // threads per output row
  constexpr int RPP = NT / TPR;     // rows per pass
  const int c4 = t % TPR, r0 = t / TPR;
  const int co = n0 + c4 * 4;
  if (p.epi == EPI_ROWMAX) {
    // every row of the tile -> (max over a 64-column sub-tile of scale * acc + bias, its column); the G = 16 lanes that hold a
    // row's sub-tile are neighbours in a wave (G divides TPR divides 64), so the row reduction is G / 2 .. 1 xor-shuffles; ties:
    // lowest column.  A 128-column tile writes two pairs per row: the table is laid out in 64-column tiles whatever BN is
    // (launchers only ask it of tiles of whole sub-tiles: BN = 64, 128, 256)
    constexpr int G = (BN < ROWMAX_TILE_N ? BN : ROWMAX_TILE_N) / 4;
    const int ntn = (p.Cout + ROWMAX_TILE_N - 1) / ROWMAX_TILE_N, tn = co / ROWMAX_TILE_N;
    for (int row = r0; row < BM; row += RPP) {  // BM % RPP == 0: every lane of a row group takes the same trips
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = co + e;
        if (c < p.Cout) {
          const float v = Cs[row * LDC + c4 * 4 + e] * (p.scale ? p.scale[c] : 1.f) + (p.bias ? p.bias[c] : 0.f);
          if (v > best) {
            best = v;
            bi = c;
          }
        }
      }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) {
          best = ov;
          bi = oi;
        }
      }
      const int m = m0 + row;
      if ((c4 & (G - 1)) == 0 && m < p.M && co < p.Cout) {
        float2 pr;
        pr.x = best;
        pr.y = __int_as_float(bi);
        *reinterpret_cast<float2*>(p.out + ((size_t)m * ntn + tn) * 2) = pr;
      }
    }
    return;
  }
  unsigned am = 0u;  // max|y| of what this thread stores (bit pattern), for the output's record
  const bool live = co < p.Cout;  // (no early return: the record's wave reduction below wants all 64 lanes)
  const int ohw = p.OH * p.OW;
  if (live && p.vec) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = zero4;
    if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + co);
    if (p.bias) bi = *reinterpret_cast<const float4*>(p.bias + co);
    int cq = co, ab = 0;
    if (p.epi == EPI_DECONV2X2) {
      const int cq_n = p.Cout >> 2;
      ab = co / cq_n;
      cq = co - ab * cq_n;
    }
#pragma unroll(PRE ? EpiRows<BM, BN, NT>::NR : UNROLL)
    for (int i = 0; i < EpiRows<BM, BN, NT>::NR; ++i) {
      const int row = r0 + RPP * i;
      const int m = m0 + row;
      if (row >= BM || m >= p.M) break;
      float4 v = *reinterpret_cast<const float4*>(Cs + row * LDC + c4 * 4);
      v.x = v.x * sc.x + bi.x;
      v.y = v.y * sc.y + bi.y;
      v.z = v.z * sc.z + bi.z;
      v.w = v.w * sc.w + bi.w;
      size_t o;
      float4 rr = zero4;
      if (p.epi == EPI_STORE) {
        if (p.res) {
          rr = PRE ? pre[i] : *reinterpret_cast<const float4*>(p.res + (size_t)m * p.res_ld + co);
          if (!p.res_post) {
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
        }
        o = (size_t)m * p.out_ld + co;
      } else {  // ConvTranspose2d(k=2, s=2): co = (a2*2+b2)*Cq + cq -> pixel (2oh+a2, 2ow+b2)
        const int n = m / ohw, rem = m - n * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const size_t opix = ((size_t)n * (2 * p.OH) + 2 * oh + (ab >> 1)) * (2 * p.OW) + 2 * ow + (ab & 1);
        o = opix * p.out_ld + cq;
      }
      v.x = apply_act(v.x, p.act);
      v.y = apply_act(v.y, p.act);
      v.z = apply_act(v.z, p.act);
      v.w = apply_act(v.w, p.act);
      if (p.res_post) {  // y = res + act(conv): CSPRep "x_1 + conv2(x)" (rtdetr_hybrid_encoder.py:209-213)
        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
      }
      *reinterpret_cast<float4*>(p.out + o) = v;
      amax_fold4(am, v);
      if (PRE) __builtin_amdgcn_sched_barrier(0);  // one row at a time: the 16-wave tiles have 64 registers per lane
    }
  } else if (live) {  // ragged Cout / unaligned rows: scalar stores (EPI_STORE only)
    for (int row = r0; row < BM; row += RPP) {
      const int m = m0 + row;
      if (m >= p.M) break;
      for (int e = 0; e < 4 && co + e < p.Cout; ++e) {
        float v = Cs[row * LDC + c4 * 4 + e];
        v = v * (p.scale ? p.scale[co + e] : 1.f) + (p.bias ? p.bias[co + e] : 0.f);
        const float rr = p.res ? p.res[(size_t)m * p.res_ld + co + e] : 0.f;
        if (!p.res_post) v += rr;
        v = apply_act(v, p.act);
        if (p.res_post) v += rr;
        p.out[(size_t)m * p.out_ld + co + e] = v;
        amax_fold(am, v);
      }
    }
  }
  if (p.amax_out) amax_commit(p.amax_out, am, t);
}

// ---- epilogue straight from the accumulators (EPI_STORE): lane (li, lh) of a wave holds, for MFMA tile (a, b), register
// r = D[row 32a + (r & 3) + 8 (r >> 2) + 4 lh][column 32b + li]; a store instruction therefore writes two output rows,
// 128 contiguous bytes each.  Same arithmetic per element as epilogue_tile.  One instantiation per activation (the
// activation switch sits outside the unrolled element loops: the GELU expansion appears once per element, not five
// functions per element).
template <int ACT, int TM, int TN>
__device__ __forceinline__ void epilogue_direct_act(const ConvK& p, const f32x16 (&acc)[TM][TN], int mw, int nw, int li, int lh) {
  unsigned am = 0u;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int co = nw + 32 * b + li;
    const bool cok = co < p.Cout;
    const float sc = (p.scale && cok) ? p.scale[co] : 1.f;
    const float bi = (p.bias && cok) ? p.bias[co] : 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int mb = mw + 32 * a + 4 * lh;
      const float* rp = p.res ? p.res + (size_t)mb * p.res_ld + co : nullptr;
      float* op = p.out + (size_t)mb * p.out_ld + co;
      float rr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // the residual reads of a tile are issued together
        const int dr = (r & 3) + 8 * (r >> 2);
        rr[r] = (rp && cok && mb + dr < p.M) ? rp[(size_t)dr * p.res_ld] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        float v = acc[a][b][r] * sc + bi;
        if (rp && !p.res_post) v += rr[r];
        v = apply_act(v, ACT);
        if (p.res_post) v += rr[r];
        if (cok && mb + dr < p.M) {
          op[(size_t)dr * p.out_ld] = v;
          amax_fold(am, v);
        }
      }
    }
  }
  if (p.amax_out) amax_commit(p.amax_out, am, threadIdx.x);
}

template <int TM, int TN>
__device__ __forceinline__ void epilogue_direct(const ConvK& p, const f32x16 (&acc)[TM][TN], int mw, int nw, int li, int lh) {
  switch (p.act) {  // block-uniform
    case ACT_RELU: epilogue_direct_act<ACT_RELU, TM, TN>(p, acc, mw, nw, li, lh); break;
    case ACT_SILU: epilogue_direct_act<ACT_SILU, TM, TN>(p, acc, mw, nw, li, lh); break;
    case ACT_SIGMOID: epilogue_direct_act<ACT_SIGMOID, TM, TN>(p, acc, mw, nw, li, lh); break;
    case ACT_GELU: epilogue_direct_act<ACT_GELU, TM, TN>(p, acc, mw, nw, li, lh); break;
    default: epilogue_direct_act<ACT_NONE, TM, TN>(p, acc, mw, nw, li, lh); break;
  }
}

// per-launch timing hooks (bench.py roofline leg), defined in ymk_conv.hip
std::pair<hipEvent_t, hipEvent_t>* conv_prof_open(hipStream_t s, const ConvK& k, int BM, int BN, int grid, int ksplit);

// split-operand path (ymk_conv_split.hip): true when the launch was taken.  code 2 / 3: bf16 planes (3 / 6 MFMAs per
// product tile), SPLIT_F16X2: two scaled fp16 planes (3 MFMAs)
bool conv2d_split(hipStream_t s, ConvK& k, const ConvW& w, int code, SplitCtx* ctx);
// A-stationary short-K 1 x 1 kernel (ymk_conv_astat.hip): false = this launch is not one it runs
bool conv2d_f16_astat_can(const ConvK& k, size_t w_bytes);
bool conv2d_f16_astat(hipStream_t s, ConvK& k, const void* planes, size_t w_bytes);
int conv2d_f16_astat_columns(const ConvK& k);

}  // namespace ymk
