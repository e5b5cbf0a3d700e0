// A-stationary fp16-split 1 x 1 convolution for short K (the kernel runs K <= 256; the router sends it K <= 192: at K = 256 its
// 64-column form measured behind the register-staged kernel wherever a residual is read, ymk_conv_split.hip route_f16).  Round 4
// built and validated it behind a test operator; round 5 routes the library's short-K pointwise layers to it (conv2d_split: the
// PARSeq encoder's qkv / proj / fc1, the ResNet expands 64 -> 256 / 128 -> 512 with their residual, the vocabulary head) and folds the LayerNorm in front of
// a linear layer into its operand load.  Why it exists: profiles/r04_conv_two_roof_by_layer.md (the K = 192 linear layers of
// the PARSeq encoder at 0.22-0.45 of their HBM roof, the ResNet expands at 0.38-0.57) and
// profiles/r04_conv_f16_short_k_pmc_pass*.csv (their waves wait two thirds of their cycles, 16 VALU + 12.6 SALU instructions
// per MFMA: every 128 x 128 tile pays the A fetch, the fp32 -> (h, l) conversion and the prologue again for a K loop of 2-8
// steps).  References for the layers: models/layers/parseq_transformer.py:188-204 (timm ViT blocks: norm1 -> qkv, norm2 ->
// fc1), models/dbnet_plus.py:33-38 (ResNet-50 bottlenecks).
//
// Round 6: the greedy loop's vocabulary head (EPI_ROWMAX: (max, column) per 64-column sub-tile instead of the tile) runs here too -
// astat_rowmax folds the accumulators across lanes, one column block per block, a group's row blocks on one XCD
// (profiles/r06_rowmax_head_tiles.md: 28.2 us per step at 1234 rows against 35.2 on the 128 x 64 tile, 38.8 against 56.3 at 2048).
//
// Same arithmetic as conv_f16_dma (ymk_conv_dma.hip): two scaled fp16 planes per fp32 operand, the three MFMAs of a product
// tile in the same order, the same K order - the results must equal that kernel's bit for bit.  What changes is who waits:
//   * a block owns 128 rows (4 waves x 32) for ALL column blocks of the layer (or of its column group).  Each wave loads ITS
//     32 rows of A once, straight into registers (buffer loads; a row past M or a channel past C is an out-of-range offset ->
//     zeros), and converts them to the two planes once: 8 VGPRs per 16-k step, 16 KT VGPRs for K = 32 KT (96 at K = 192).
//     No LDS, no barrier and no second conversion for A, whatever the number of column blocks (6 for fc1: the tiled kernels
//     fetch and convert its A rows six times);
//   * the weight slabs (BN columns x 32 k x 2 planes = 16 KB at BN = 128) flow through three LDS stages by LDS-DMA as ONE
//     sequence over (column block, K tile) - the pipeline does not drain between column blocks;
//   * a column block's accumulators go out straight from registers, four rows at a time through buffer descriptors; with two
//     blocks per CU (K = 192 at 128 columns: 255 VGPRs, 48 KB of LDS) the other block's MFMAs cover this one's epilogue.
#include <atomic>
#include <string>

#include "ymk_conv_kernel.h"

namespace ymk {

typedef _Float16 as_hf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 as_hf16x8_t __attribute__((ext_vector_type(8)));
typedef float as_hf32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void as_lds_void;

__device__ __forceinline__ float2 astat_scales(unsigned amax_bits) {  // f16_scales of ymk_conv_split.hip
  int e = (int)(amax_bits >> 23);
  e = e < 27 ? 27 : (e > 227 ? 227 : e);
  float2 r;
  r.x = __uint_as_float((unsigned)(268 - e) << 23);
  r.y = __uint_as_float((unsigned)(e - 14) << 23);
  return r;
}

__device__ __forceinline__ void astat_split8(const f32x4 u, const f32x4 v, float sa, as_hf16x8_t& hi, as_hf16x8_t& lo) {
  as_hf32x2_t x[4] = {{u.x, u.y}, {u.z, u.w}, {v.x, v.y}, {v.z, v.w}};
  as_hf16x2_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[i] *= sa;
    h[i] = __builtin_convertvector(x[i], as_hf16x2_t);
    x[i] -= __builtin_convertvector(h[i], as_hf32x2_t);  // exact
    l[i] = __builtin_convertvector(x[i], as_hf16x2_t);
  }
  hi = as_hf16x8_t{h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
  lo = as_hf16x8_t{l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
}

// one column block of a wave (32 rows x 32 TN columns) out of the accumulators: scale / bias / residual / activation as
// epilogue_tile does (same expression per value), through buffer descriptors with 32-bit offsets, four rows at a time -
// the direct epilogue of ymk_conv_kernel.h keeps sixteen 64-bit addresses and sixteen residual values live, which is
// what pushed the first form of this kernel past 256 registers
template <int ACT, int TN, int RG>
__device__ __forceinline__ void astat_store(const ConvK& p, const f32x16 (&acc)[TN], float inv_sa, int mw, int n0, int li, int lh,
                                            __amdgpu_buffer_rsrc_t rsrc_o, __amdgpu_buffer_rsrc_t rsrc_r, unsigned& am) {
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int co = n0 + 32 * b + li;
    const bool cok = co < p.Cout;
    const float sc = (p.scale && cok) ? p.scale[co] : 1.f;
    const float bi = (p.bias && cok) ? p.bias[co] : 0.f;
    const int row0 = mw + 4 * lh;
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += RG) {  // RG accumulator rows at a time: their residual values are fetched together
      float rr[RG];
#pragma unroll
      for (int i = 0; i < RG; ++i) {
        const int r = r0 + i, row = row0 + (r & 3) + 8 * (r >> 2);
        const unsigned ro = (cok && row < p.M) ? ((unsigned)row * (unsigned)p.res_ld + (unsigned)co) * 4u : OOB_OFFSET;
        rr[i] = p.res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, (int)ro, 0, 0)) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < RG; ++i) {
        const int r = r0 + i, row = row0 + (r & 3) + 8 * (r >> 2);
        const bool ok = cok && row < p.M;
        const unsigned oo = ok ? ((unsigned)row * (unsigned)p.out_ld + (unsigned)co) * 4u : OOB_OFFSET;
        float v = (acc[b][r] * inv_sa) * sc + bi;
        if (p.res && !p.res_post) v += rr[i];
        v = apply_act(v, ACT);
        if (p.res_post) v += rr[i];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_o, (int)oo, 0, 0);
        if (ok) amax_fold(am, v);
      }
      // a group at a time: without the fence the scheduler hoists every residual load of the tile, and without pinning
      // `am` the max-chain is re-associated into a tree over all 16 TN values of the wave (56 registers: measured)
      asm volatile("" : "+v"(am));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// EPI_ROWMAX out of the accumulators (BN = 128: two 64-column sub-tiles): every row's (largest scale * acc + bias, its column;
// ties: lowest column) per sub-tile - the pair table epilogue_tile writes (ymk_conv_kernel.h), from the same expression per
// value.  Per sub-tile a lane holds 16 candidates, slot r = accumulator row, each the best of its two columns; the 32 lanes of
// a half-wave then fold them in exchange steps in which a lane keeps HALF of its slots and hands the other half to its partner
// (xor 16, 8, 4, 2: 8 + 4 + 2 + 1 exchanged pairs, then one more with xor 1: 16 in all instead of 16 x 5), after which lane li
// holds slot li >> 1, folded over all 32 lanes.  (value, lowest column) is a total order, so the order of the folds is
// immaterial.  One sub-tile at a time behind a scheduling fence: 32 registers of candidates next to the A planes, not 64.
__device__ __forceinline__ void astat_fold(float& v, int& c, float ov, int oc) {
  const bool take = ov > v || (ov == v && oc < c);
  v = take ? ov : v;
  c = take ? oc : c;
}

__device__ __forceinline__ void astat_rowmax(const ConvK& p, const f32x16 (&acc)[4], float inv_sa, int mw, int n0, int li, int lh) {
  const int ntn = (p.Cout + ROWMAX_TILE_N - 1) / ROWMAX_TILE_N;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float val[16];
    int col[16];
    const int c0 = n0 + 64 * j + li, c1 = c0 + 32;
    const bool ok0 = c0 < p.Cout, ok1 = c1 < p.Cout;
    const float s0 = (p.scale && ok0) ? p.scale[c0] : 1.f, s1 = (p.scale && ok1) ? p.scale[c1] : 1.f;
    const float b0 = (p.bias && ok0) ? p.bias[c0] : 0.f, b1 = (p.bias && ok1) ? p.bias[c1] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float best = -INFINITY;
      int bc = 0x7fffffff;
      const float v0 = (acc[2 * j][r] * inv_sa) * s0 + b0, v1 = (acc[2 * j + 1][r] * inv_sa) * s1 + b1;
      if (ok0 && v0 > best) {
        best = v0;
        bc = c0;
      }
      if (ok1 && v1 > best) {
        best = v1;
        bc = c1;
      }
      val[r] = best;
      col[r] = bc;
    }
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1) {      // h slots survive in this lane; partner: lane ^ 2 h
      const bool up = (li & (2 * h)) != 0;  // this lane keeps slots [h, 2 h) of the 2 h it holds, its partner [0, h)
#pragma unroll
      for (int q = 0; q < h; ++q) {
        const float send_v = up ? val[q] : val[q + h];
        const int send_c = up ? col[q] : col[q + h];
        float v = up ? val[q + h] : val[q];
        int c = up ? col[q + h] : col[q];
        astat_fold(v, c, __shfl_xor(send_v, 2 * h), __shfl_xor(send_c, 2 * h));
        val[q] = v;
        col[q] = c;
      }
    }
    astat_fold(val[0], col[0], __shfl_xor(val[0], 1), __shfl_xor(col[0], 1));
    // lanes li and li ^ 1 now hold slot (li >> 1) & 15; the even one writes it
    const int r = (li >> 1) & 15, row = mw + 4 * lh + (r & 3) + 8 * (r >> 2);
    const int tn = n0 / ROWMAX_TILE_N + j;
    if ((li & 1) == 0 && row < p.M && tn < ntn) {
      float2 pr;
      pr.x = val[0];
      pr.y = __int_as_float(col[0]);
      *reinterpret_cast<float2*>(p.out + ((size_t)row * ntn + tn) * 2) = pr;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// 1 x 1, stride 1, no padding (the caller checks); KT = Kpad / 32 K tiles (compile time: the A planes live in registers);
// BN columns per column block; grid = ceil(M / 128) x column groups (p.ntiles_n column blocks are dealt to gridDim.y groups)
// LN: the rows are LayerNorm-ed on their way into the planes (p.ln_g / p.ln_b / p.ln_eps over the C = 32 KT channels of a row:
// a row lives in the two lanes li and li + 32 of its wave, so mean and variance are one xor-shuffle away) - what
// k_layernorm + this kernel computed through a [M][C] round trip; p.amax is then the LayerNorm's static output bound.
// RM (BN = 128): the row-max epilogue above instead of stores (the greedy loop's vocabulary head, EPI_ROWMAX), and a block whose
// 128 rows all belong to finished groups (p.row_group / p.group_open) returns at once
template <int BN, int KT, bool LN = false, bool RM = false>
__global__ __launch_bounds__(256, 2) void conv_f16_astat(ConvK p, const uint4* __restrict__ wsplit, unsigned w_bytes) {
  constexpr int TN = BN / 32;
  constexpr int RG = KT * 16 + TN * 24 <= 160 ? 16 : 4;  // residual values in flight per lane: as many as the registers allow
  constexpr int NST = 3, PD = 2;
  constexpr int B_STAGE = BN * 128;  // bytes: BN rows x 2 planes x 32 halves
  constexpr int BROWS = BN / 4;      // B rows a wave loads per slab
  constexpr int BI = BROWS / 8;      // its LDS-DMA instructions per slab
  __shared__ __attribute__((aligned(16))) char lds[NST * B_STAGE];

  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
  const float2 sc = astat_scales((unsigned)__builtin_amdgcn_readfirstlane((int)amax_read(p.amax, t)));
  const float sa = sc.x, inv_sa = sc.y;
  int tile_m, group = (int)blockIdx.y;
  if constexpr (RM) {
    // few row blocks, many column groups: the row blocks of ONE group share an XCD (workgroups go to the XCDs round-robin in
    // x-fastest order), so that an XCD's L2 holds the weights of its ~ gridDim.y / 8 groups and not the whole 5.5 MB panel
    const int gx = gridDim.x, nblk = gx * (int)gridDim.y, bid = (int)blockIdx.x + gx * (int)blockIdx.y;
    const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    group = v / gx;
    tile_m = v - group * gx;
  } else {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    tile_m = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = tile_m * 128;
  if constexpr (RM) {
    if (!tile_needed<128, 256>(p, m0, t)) return;
  }
  // column blocks of this block: [nb0, nb1)
  const int per = (p.ntiles_n + (int)gridDim.y - 1) / (int)gridDim.y;
  const int nb0 = group * per, nb1 = min(p.ntiles_n, nb0 + per);
  if (nb0 >= nb1) return;

  // ---- A: the lane's fragments of all KT tiles, loaded once.  MFMA 32x32x16 A operand: row li, k = 8 lh .. + 7 of the step
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const int m = m0 + 32 * wv + li;
  const unsigned row_off = m < p.M ? (unsigned)m * (unsigned)p.in_ld * 4u : OOB_OFFSET;
  as_hf16x8_t ah[KT][2], al[KT][2];
  {
    f32x4 u[KT][2], v[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int c = kt * 32 + s * 16 + lh * 8;  // first channel of the lane's 8
        const unsigned o0 = (row_off != OOB_OFFSET && c < p.C) ? row_off + (unsigned)c * 4u : OOB_OFFSET;
        const unsigned o1 = (row_off != OOB_OFFSET && c + 4 < p.C) ? row_off + (unsigned)(c + 4) * 4u : OOB_OFFSET;
        u[kt][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)o0, 0, 0));
        v[kt][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)o1, 0, 0));
      }
    if constexpr (LN) {
      // torch.nn.LayerNorm as k_layernorm computes it: mean, then the biased variance of the centred values, then
      // (x - mean) * rstd * gamma + beta; C == 32 KT here (the launcher checks), so no lane holds a padding channel
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          sum += ((u[kt][s].x + u[kt][s].y) + (u[kt][s].z + u[kt][s].w)) + ((v[kt][s].x + v[kt][s].y) + (v[kt][s].z + v[kt][s].w));
      sum += __shfl_xor(sum, 32);
      const float mean = sum / (float)(32 * KT);
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
      const float rstd = 1.f / sqrtf(sq / (float)(32 * KT) + p.ln_eps);
      // scale / shift and cut one K tile at a time behind a scheduling fence: the staging registers of a tile die as its planes
      // are born.  (At K = 192 the compiler still parks 44 registers in scratch across this prologue - 256 are not quite enough
      // for a whole fp32 row next to its planes; visiting the row a second time instead, tile by tile, measured no fewer in the
      // resource report and adds an L2 round trip per tile.)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int c = kt * 32 + s * 16 + lh * 8;
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_g + c), g1 = *reinterpret_cast<const f32x4*>(p.ln_g + c + 4);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln_b + c), b1 = *reinterpret_cast<const f32x4*>(p.ln_b + c + 4);
          astat_split8(u[kt][s] * rstd * g0 + b0, v[kt][s] * rstd * g1 + b1, sa, ah[kt][s], al[kt][s]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) astat_split8(u[kt][s], v[kt][s], sa, ah[kt][s], al[kt][s]);
    }
  }

  // ---- B: slab q = (nb - nb0) KT + kt of this block's sequence -> stage q % 3, two slabs ahead
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wsplit), 0, w_bytes, 0x00020000);
  const int jr = lane >> 3, js = lane & 7;
  unsigned boff[BI];  // the lane's 16 bytes within column block 0, K tile 0 (panel rows are KT x 128 bytes)
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int row = BROWS * wv + 8 * j + jr;
    boff[j] = (unsigned)row * (unsigned)(KT * 128) + (unsigned)((js ^ ((row >> 1) & 7)) * 16);
  }
  auto issue = [&](int nb, int kt, int st) {
    char* Bs = lds + st * B_STAGE + (BROWS * wv) * 128;
    const int soff = nb * BN * (KT * 128) + kt * 128;
#pragma unroll
    for (int j = 0; j < BI; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (as_lds_void*)(Bs + j * 1024), 16, (int)boff[j], soff, 0, 0);
  };

  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (unsigned)p.M * (unsigned)p.out_ld * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res : p.out), 0,
      p.res ? (p.res_ld ? (unsigned)p.M * (unsigned)p.res_ld * 4u : (unsigned)p.Cout * 4u) : (unsigned)p.M * (unsigned)p.out_ld * 4u, 0x00020000);
  unsigned am = 0u;
  f32x16 acc[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  const int bswz = (li >> 1) & 7;

  const int total = (nb1 - nb0) * KT;
  // the slab after (nb, kt) in the sequence
  int inb = nb0, ikt = 0;
  auto advance = [&]() {
    if (++ikt == KT) {
      ikt = 0;
      ++inb;
    }
  };
  issue(inb, ikt, 0);
  advance();
  if (total > 1) {
    issue(inb, ikt, 1);
    advance();
  }
  int st = 0, stp = PD, issued = total > 1 ? 2 : 1;
  for (int nb = nb0; nb < nb1; ++nb) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int q = (nb - nb0) * KT + kt;
      // the counted wait assumes that everything outstanding is a slab DMA (loads return in order).  Right after a column
      // block's epilogue the wave also has STORES in flight, which may return out of order with the loads: drain there
      if (q + 1 < total && !(kt == 0 && nb > nb0)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const char* Bs = lds + st * B_STAGE + li * 128;
        as_hf16x8_t bh[TN], bl[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          bh[b] = *reinterpret_cast<const as_hf16x8_t*>(Bs + b * 32 * 128 + (((s * 2 + lh) ^ bswz) * 16));
          bl[b] = *reinterpret_cast<const as_hf16x8_t*>(Bs + b * 32 * 128 + (((4 + s * 2 + lh) ^ bswz) * 16));
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kt][s], bh[b], acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt][s], bl[b], acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt][s], bh[b], acc[b], 0, 0, 0);
        if (s == 0 && issued < total) {  // the slab two ahead, behind the first step's MFMAs
          issue(inb, ikt, stp);
          advance();
          ++issued;
        }
      }
      st = st == NST - 1 ? 0 : st + 1;
      stp = stp == NST - 1 ? 0 : stp + 1;
    }
    // ---- this column block's outputs, straight from the accumulators; the stores drain under the next block's MFMAs
    if constexpr (RM) {
      static_assert(!RM || TN == 4, "row-max epilogue: 128-column blocks");
      astat_rowmax(p, acc, inv_sa, m0 + 32 * wv, nb * BN, li, lh);
    } else {
      switch (p.act) {  // block-uniform
        case ACT_RELU: astat_store<ACT_RELU, TN, RG>(p, acc, inv_sa, m0 + 32 * wv, nb * BN, li, lh, rsrc_o, rsrc_r, am); break;
        case ACT_GELU: astat_store<ACT_GELU, TN, RG>(p, acc, inv_sa, m0 + 32 * wv, nb * BN, li, lh, rsrc_o, rsrc_r, am); break;
        case ACT_SILU: astat_store<ACT_SILU, TN, RG>(p, acc, inv_sa, m0 + 32 * wv, nb * BN, li, lh, rsrc_o, rsrc_r, am); break;
        case ACT_SIGMOID: astat_store<ACT_SIGMOID, TN, RG>(p, acc, inv_sa, m0 + 32 * wv, nb * BN, li, lh, rsrc_o, rsrc_r, am); break;
        default: astat_store<ACT_NONE, TN, RG>(p, acc, inv_sa, m0 + 32 * wv, nb * BN, li, lh, rsrc_o, rsrc_r, am); break;
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  }
  if (p.amax_out) amax_commit(p.amax_out, am, t);  // once per wave, over all its column blocks
}


static std::atomic<int> g_astat_rowmax_dealt{0};  // "rowmax_tile" 4 (A/B runs): the head's column blocks dealt to groups as a store launch's are
void astat_rowmax_dealt(int on) { g_astat_rowmax_dealt = on; }

template <int BN, int KT, bool LN, bool RM = false>
static void launch_astat(hipStream_t s, ConvK& k, const void* planes, size_t w_bytes, int groups) {
  hipLaunchKernelGGL((conv_f16_astat<BN, KT, LN, RM>), dim3((k.M + 127) / 128, groups), dim3(256), 0, s, k, reinterpret_cast<const uint4*>(planes), (unsigned)w_bytes);
}

// column width of the launch: 64 where the registers are needed elsewhere - K > 192 (A planes of 7-8 K tiles), a residual at
// K > 128 (sixteen residual values in flight per lane instead of four: what the K = 192 projection lost to in round 4), a
// fused LayerNorm (the fp32 staging of a whole row next to its planes) - and for Cout <= 64
static bool astat_narrow(const ConvK& k) { return k.epi == EPI_ROWMAX ? false : k.Kpad > 192 || k.Cout <= 64 || (k.res != nullptr && k.Kpad > 128) || k.ln_g != nullptr; }

int conv2d_f16_astat_columns(const ConvK& k) { return astat_narrow(k) ? 64 : 128; }

// The launch (conv2d_split routes to it; the caller has set k.scale to the fp16 panels' epilogue scale and k.amax).  A 1 x 1,
// stride-1, unpadded layer with Kpad <= 256, plain stores, views below 4 GiB; with k.ln_g: C == Kpad.  False = not taken.
bool conv2d_f16_astat_can(const ConvK& k, size_t w_bytes) {
  if (k.KH != 1 || k.KW != 1 || k.stride != 1 || k.stride_w != 1 || k.pad != 0) return false;
  if (k.epi == EPI_ROWMAX) {  // the pair table instead of the tile: 128-column blocks, K <= 192, nothing else in the epilogue
    if (k.Kpad > 192 || k.res != nullptr || k.ln_g != nullptr || k.act != ACT_NONE || k.Cout <= 64) return false;
  } else if (k.epi != EPI_STORE || k.row_group != nullptr) {
    return false;
  }
  if (k.Kpad % 32 != 0 || k.Kpad < 32 || k.Kpad > 256 || k.C % 4 != 0 || k.M <= 0) return false;
  if ((size_t)k.M * (size_t)k.out_ld * 4 >= (size_t)OOB_OFFSET || (size_t)k.M * (size_t)k.res_ld * 4 >= (size_t)OOB_OFFSET) return false;
  if (w_bytes >= (size_t)OOB_OFFSET) return false;
  if (k.ln_g != nullptr && (k.C != k.Kpad || k.ln_b == nullptr || (k.Kpad != 128 && k.Kpad != 192))) return false;
  return true;
}

// the same question for a row-major GEMM out[M][w.cout] = X[M][K] . W^T (leading dimension ld for output and residual)
bool gemm_takes_astat(int M, int K, const ConvW& w, bool with_res, int ld) {
  ConvK k{};
  k.KH = k.KW = k.stride = k.stride_w = 1;
  k.epi = EPI_STORE;
  k.Kpad = w.kpad;
  k.C = K;
  k.M = M;
  k.out_ld = ld;
  k.res_ld = with_res ? ld : 0;
  return w.kh == 1 && w.kw == 1 && w.mode == 0 && conv2d_f16_astat_can(k, (size_t)((w.cout + 255) / 256 * 256) * w.kpad * 4);
}

bool conv2d_f16_astat(hipStream_t s, ConvK& k, const void* planes, size_t w_bytes) {
  if (!conv2d_f16_astat_can(k, w_bytes)) return false;
  const bool narrow = astat_narrow(k);
  const int bn = narrow ? 64 : 128;
  k.ntiles_n = (k.Cout + bn - 1) / bn;
  // few row blocks: deal the column blocks to groups so that the launch still covers the chip (A is then loaded per group)
  const int mblocks = (k.M + 127) / 128;
  int groups = 1;
  while (mblocks * groups < 512 && groups * 2 <= k.ntiles_n) groups *= 2;
  // the row-max head: ONE column block per block (job AE, 655 / 1234 / 2048 rows x 7119 columns: 24.3 / 28.2 / 38.8 us against 33.4 /
  // 38.1 / 43.8 with two per block) - its blocks are short and latency-bound, the more of them in flight the better
  if (k.epi == EPI_ROWMAX && !g_astat_rowmax_dealt.load(std::memory_order_relaxed)) groups = k.ntiles_n;
  const int kt = k.Kpad / 32;
  if (k.epi == EPI_ROWMAX) {
    switch (kt) {
      case 1: launch_astat<128, 1, false, true>(s, k, planes, w_bytes, groups); break;
      case 2: launch_astat<128, 2, false, true>(s, k, planes, w_bytes, groups); break;
      case 3: launch_astat<128, 3, false, true>(s, k, planes, w_bytes, groups); break;
      case 4: launch_astat<128, 4, false, true>(s, k, planes, w_bytes, groups); break;
      case 5: launch_astat<128, 5, false, true>(s, k, planes, w_bytes, groups); break;
      default: launch_astat<128, 6, false, true>(s, k, planes, w_bytes, groups); break;
    }
  } else if (k.ln_g != nullptr) {  // the ViT widths that fit the registers
    switch (kt) {
      case 4: launch_astat<64, 4, true>(s, k, planes, w_bytes, groups); break;
      case 6: launch_astat<64, 6, true>(s, k, planes, w_bytes, groups); break;
      default: return false;
    }
  } else if (narrow) {
    switch (kt) {
      case 1: launch_astat<64, 1, false>(s, k, planes, w_bytes, groups); break;
      case 2: launch_astat<64, 2, false>(s, k, planes, w_bytes, groups); break;
      case 3: launch_astat<64, 3, false>(s, k, planes, w_bytes, groups); break;
      case 4: launch_astat<64, 4, false>(s, k, planes, w_bytes, groups); break;
      case 5: launch_astat<64, 5, false>(s, k, planes, w_bytes, groups); break;
      case 6: launch_astat<64, 6, false>(s, k, planes, w_bytes, groups); break;
      case 7: launch_astat<64, 7, false>(s, k, planes, w_bytes, groups); break;
      default: launch_astat<64, 8, false>(s, k, planes, w_bytes, groups); break;
    }
  } else {
    switch (kt) {
      case 1: launch_astat<128, 1, false>(s, k, planes, w_bytes, groups); break;
      case 2: launch_astat<128, 2, false>(s, k, planes, w_bytes, groups); break;
      case 3: launch_astat<128, 3, false>(s, k, planes, w_bytes, groups); break;
      case 4: launch_astat<128, 4, false>(s, k, planes, w_bytes, groups); break;
      case 5: launch_astat<128, 5, false>(s, k, planes, w_bytes, groups); break;
      default: launch_astat<128, 6, false>(s, k, planes, w_bytes, groups); break;
    }
  }
  return true;
}

}  // namespace ymk
