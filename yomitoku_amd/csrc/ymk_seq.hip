// Token-sequence kernels for the transformers on the path (PARSeq ViT encoder + AR decoder,
// RT-DETR AIFI / decoder): LayerNorm, position embedding, fp32-MFMA flash attention, a masked
// small-query attention, and the on-device greedy-decode bookkeeping of PARSeq.
#include "ymk_common.h"
#include "ymk_seq.h"

namespace ymk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------- LayerNorm over the last dim
// one wave per row; D <= 1024, D % 4 == 0.  torch: (x - mean) / sqrt(var_biased + eps) * g + b
// `in_rows_mod` > 0 reads row (m % in_rows_mod): a [rows_mod][D] table broadcast over the batch.
__global__ void k_layernorm(const float* __restrict__ x, int ldx, int in_rows_mod, const float* __restrict__ g,
                            const float* __restrict__ b, float eps, float* __restrict__ y, int ldy, int M, int D) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)(in_rows_mod > 0 ? row % in_rows_mod : row) * ldx;
  float4 v[4];
  const int nv = D >> 2;  // float4 count, <= 256
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      v[i] = *reinterpret_cast<const float4*>(xr + c * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + bb * bb) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float4 gg = *reinterpret_cast<const float4*>(g + c * 4);
      const float4 be = *reinterpret_cast<const float4*>(b + c * 4);
      float4 o;
      o.x = (v[i].x - mean) * rstd * gg.x + be.x;
      o.y = (v[i].y - mean) * rstd * gg.y + be.y;
      o.z = (v[i].z - mean) * rstd * gg.z + be.z;
      o.w = (v[i].w - mean) * rstd * gg.w + be.w;
      *reinterpret_cast<float4*>(y + (size_t)row * ldy + c * 4) = o;
    }
  }
}
void layernorm(hipStream_t s, const float* x, int ldx, int in_rows_mod, const float* g, const float* b, float eps,
               float* y, int ldy, int M, int D) {
  YMK_CHECK(D % 4 == 0 && D <= 1024, "layernorm: D must be a multiple of 4, <= 1024");
  if (M == 0) return;
  hipLaunchKernelGGL(k_layernorm, dim3((M + 3) / 4), dim3(256), 0, s, x, ldx, in_rows_mod, g, b, eps, y, ldy, M, D);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- x[b, r, c, :] += pos[r*full_gw + c, :]
__global__ void k_add_pos(float* __restrict__ x, const float* __restrict__ pos, int gh, int gw, int full_gw, int D4,
                          size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % D4);
    const size_t tok = i / D4;
    const int cc = (int)(tok % gw);
    const int rr = (int)((tok / gw) % gh);
    float4 v = reinterpret_cast<float4*>(x)[i];
    const float4 p = reinterpret_cast<const float4*>(pos)[(size_t)(rr * full_gw + cc) * D4 + c4];
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    reinterpret_cast<float4*>(x)[i] = v;
  }
}
void add_pos_embed(hipStream_t s, float* x, const float* pos, int B, int gh, int gw, int full_gw, int D) {
  const size_t total = (size_t)B * gh * gw * (D / 4);
  size_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_add_pos, dim3((int)g), dim3(256), 0, s, x, pos, gh, gw, full_gw, D / 4, total);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- flash attention, exact fp32 on the MFMA pipe
// One wave owns 32 queries of one (batch, head); 4 waves per block share K/V tiles of 64 keys in LDS.
// Scores are computed transposed, S^T = K Q^T, so that a lane's 16 accumulator registers all belong to
// ITS query (column lane&31): the running max / sum of the online softmax are lane-local plus one
// exchange with lane^32, and P never leaves registers - register r of S^T is exactly the B operand
// of step r of O^T += V^T P^T (the k order of that product is free, so it follows the D layout:
// key(r, half) = (r&3) + 8*(r>>2) + 4*half).
struct AttnP {
  const float *q, *k, *v;
  float* o;
  int ldq, ldk, ldv, ldo;          // row strides (floats)
  long bsq, bsk, bsv, bso;         // batch strides (floats)
  int Lq, Lk, H;
  float scale;
  // ragged batches (grouped PARSeq forward): per-sample row offsets / lengths replace the batch strides and Lq / Lk.
  // koff/klen: keys and values of sample b start at row koff[b] and number klen[b]; qoff/qlen: the same for the
  // queries and the output rows.  Null = the uniform strided form.
  const int *qoff, *qlen, *koff, *klen;
  // fp16-split form only (k_flash_attn_f16): max|x| records (ymk_common.h) bounding q, k and v
  const unsigned *amax_q, *amax_k, *amax_v;
};

// HD: head dim as laid out in LDS / the accumulators (a multiple of 32); HDR <= HD: the real head dim (a multiple of 8).
// HDR < HD (48 in 64, parseq-small) leaves LDS columns HDR..HD-1 unwritten: they only feed output rows d >= HDR, which
// are never stored.
template <int HD, int HDR = HD>
__global__ __launch_bounds__(256, 1) void k_flash_attn(AttnP p) {
  constexpr int KT = 64;           // keys per LDS tile
  constexpr int LDH = HD + 4;      // padded row
  constexpr int NKC = HDR / 8;     // k-chunks of 8 in the QK^T product
  constexpr int NDC = HD / 32;     // 32-wide d chunks of the output
  constexpr int LPT = (KT * HDR / 4) / 256;  // float4 loads per thread per operand
  static_assert(HD % 32 == 0 && HDR % 8 == 0 && HDR <= HD && (KT * HDR / 4) % 256 == 0, "head dim");
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * KT * LDH];

  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wv * 32;
  const float* qb = p.q + (size_t)b * p.bsq + h * HDR;
  const float* kb = p.k + (size_t)b * p.bsk + h * HDR;
  const float* vb = p.v + (size_t)b * p.bsv + h * HDR;
  float* ob = p.o + (size_t)b * p.bso + h * HDR;
  int Lq = p.Lq, Lk = p.Lk;
  if (p.koff) {
    const size_t r = (size_t)p.koff[b];
    kb = p.k + r * p.ldk + h * HDR;
    vb = p.v + r * p.ldv + h * HDR;
    Lk = p.klen[b];
  }
  if (p.qoff) {
    const size_t r = (size_t)p.qoff[b];
    qb = p.q + r * p.ldq + h * HDR;
    ob = p.o + r * p.ldo + h * HDR;
    Lq = p.qlen[b];
  }
  if ((int)blockIdx.x * 128 >= Lq) return;  // block-uniform: the grid is sized for the longest sample

  // this lane's query row, pre-scaled: Q[q][kc*8 + 4*lh + s]
  f32x4 qf[NKC];
  {
    const int qi = min(q0 + li, Lq - 1);
    const float* qr = qb + (size_t)qi * p.ldq;
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      qf[kc] = *reinterpret_cast<const f32x4*>(qr + kc * 8 + lh * 4);
      qf[kc] *= p.scale;
    }
  }

  f32x16 oacc[NDC];
#pragma unroll
  for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dc][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  f32x4 rk[LPT], rv[LPT];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int idx = t + 256 * i;
      const int row = idx / (HDR / 4), c4 = idx - row * (HDR / 4);
      const int key = min(k0 + row, Lk - 1);
      rk[i] = *reinterpret_cast<const f32x4*>(kb + (size_t)key * p.ldk + c4 * 4);
      rv[i] = *reinterpret_cast<const f32x4*>(vb + (size_t)key * p.ldv + c4 * 4);
    }
  };
  auto store_kv = [&](int buf) {
    float* Ks = lds + buf * (2 * KT * LDH);
    float* Vs = Ks + KT * LDH;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int idx = t + 256 * i;
      const int row = idx / (HDR / 4), c4 = idx - row * (HDR / 4);
      *reinterpret_cast<f32x4*>(Ks + row * LDH + c4 * 4) = rk[i];
      *reinterpret_cast<f32x4*>(Vs + row * LDH + c4 * 4) = rv[i];
    }
  };

  const int ntiles = (Lk + KT - 1) / KT;
  load_kv(0);
  store_kv(0);
  __syncthreads();
  for (int tt = 0; tt < ntiles; ++tt) {
    const int buf = tt & 1;
    if (tt + 1 < ntiles) load_kv((tt + 1) * KT);
    const float* Ks = lds + buf * (2 * KT * LDH);
    const float* Vs = Ks + KT * LDH;
    // a wave whose 32 queries all lie past the sample's last query still stages K / V and meets the barriers, but
    // skips the products (ragged batches: the last 128-query block of a sample is mostly such waves)
#pragma unroll
    for (int sub = 0; sub < (q0 < Lq ? KT / 32 : 0); ++sub) {
      const int kbase = tt * KT + sub * 32;
      if (kbase >= Lk) break;  // wave-uniform
      // ---- S^T[key][q] for 32 keys x 32 queries
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const float* kr = Ks + (sub * 32 + li) * LDH + lh * 4;
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(kr + kc * 8);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kc].x, sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kc].y, sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kc].z, sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kc].w, sacc, 0, 0, 0);
      }
      // ---- online softmax for this lane's query
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= Lk) sacc[r] = -INFINITY;
        mt = fmaxf(mt, sacc[r]);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);  // finite: every tile has >= 1 valid key
      const float alpha = __expf(m_run - m_new);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] = __expf(sacc[r] - m_new);
        ps += sacc[r];
      }
      ps += __shfl_xor(ps, 32, 64);
      l_run = l_run * alpha + ps;
      m_run = m_new;
      // ---- O^T[d][q] = alpha * O^T + V^T P^T
#pragma unroll
      for (int dc = 0; dc < NDC; ++dc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dc][r] *= alpha;
        const float* vr = Vs + (sub * 32 + 4 * lh) * LDH + dc * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float vf = vr[((r & 3) + 8 * (r >> 2)) * LDH];
          oacc[dc] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sacc[r], oacc[dc], 0, 0, 0);
        }
      }
    }
    if (tt + 1 < ntiles) store_kv(buf ^ 1);
    __syncthreads();
  }
  // ---- normalise and store: lane holds d = dc*32 + (r&3) + 8*(r>>2) + 4*lh of query q0 + li
  const int qi = q0 + li;
  if (qi < Lq) {
    const float inv = 1.f / l_run;
    float* orow = ob + (size_t)qi * p.ldo;
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w;
        w.x = oacc[dc][4 * g + 0] * inv;
        w.y = oacc[dc][4 * g + 1] * inv;
        w.z = oacc[dc][4 * g + 2] * inv;
        w.w = oacc[dc][4 * g + 3] * inv;
        if (dc * 32 + 8 * g + 4 * lh < HDR) *reinterpret_cast<f32x4*>(orow + dc * 32 + 8 * g + 4 * lh) = w;
      }
  }
}

// ---------------------------------------------------------------- flash attention, fp16-split operands (round 4)
// The same algorithm with both products on the 16-bit MFMA pipe, fp32-grade: S^T = K Q^T and O^T += V^T P^T multiply
// fp32 operands as two scaled fp16 planes each (three v_mfma_f32_32x32x16_f16 per 16-k step: lo x hi, hi x lo, hi x hi -
// ymk_conv_split.hip has the error analysis: products to 2^-21), fp32 accumulate, softmax arithmetic in fp32 as before.
// Scales (powers of two, exact): q (times the softmax scale), k and v each by the one that puts the max|x| record of their
// producer into [2^14, 2^15), P in (0, 1] by 2^14.  The D layout of S^T puts keys
// (r & 3) + 8 (r >> 2) + 4 lh into register r of lane half lh: registers 8 j .. 8 j + 7 ARE the lane's eight k values of
// 16-key step j of the P^T operand, provided V^T is gathered with the same key order - so P never leaves registers here
// either.  32 x 32 x 16 MFMAs: 6 per 32-key sub-tile per 16 head dims of Q K^T ... 12 + 6 NDC instead of 4 HDR / 2 + 16 NDC
// twice-as-long fp32 ones.
typedef _Float16 ah16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ah16x8 __attribute__((ext_vector_type(8)));
typedef float af32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 attn_f16_scales(unsigned amax_bits) {  // as f16_scales of ymk_conv_split.hip
  int e = (int)(amax_bits >> 23);
  e = e < 27 ? 27 : (e > 227 ? 227 : e);
  float2 r;
  r.x = __uint_as_float((unsigned)(268 - e) << 23);
  r.y = __uint_as_float((unsigned)(e - 14) << 23);
  return r;
}
// 8 floats times the power of two sc -> hi / lo planes
__device__ __forceinline__ void attn_split8(const float (&x)[8], float sc, ah16x8& hi, ah16x8& lo) {
  ah16x2 h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    af32x2 v = {x[2 * i] * sc, x[2 * i + 1] * sc};
    h[i] = __builtin_convertvector(v, ah16x2);
    v -= __builtin_convertvector(h[i], af32x2);  // exact
    l[i] = __builtin_convertvector(v, ah16x2);
  }
  hi = ah16x8{h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
  lo = ah16x8{l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
}

template <int HD, int HDR = HD>
__global__ __launch_bounds__(256, (HD <= 32 ? 3 : (HD <= 64 ? 2 : 1))) void k_flash_attn_f16(AttnP p) {
  constexpr int KT = 64;
  constexpr int LDH = HD + 4;
  constexpr int NKS = HDR / 16;    // 16-k steps of the QK^T product
  constexpr int NDC = HD / 32;
  constexpr int LPT = (KT * HDR / 4) / 256;
  static_assert(HD % 32 == 0 && HDR % 16 == 0 && HDR <= HD && (KT * HDR / 4) % 256 == 0, "head dim");
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * KT * LDH];

  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wv * 32;
  const float* qb = p.q + (size_t)b * p.bsq + h * HDR;
  const float* kb = p.k + (size_t)b * p.bsk + h * HDR;
  const float* vb = p.v + (size_t)b * p.bsv + h * HDR;
  float* ob = p.o + (size_t)b * p.bso + h * HDR;
  int Lq = p.Lq, Lk = p.Lk;
  if (p.koff) {
    const size_t r = (size_t)p.koff[b];
    kb = p.k + r * p.ldk + h * HDR;
    vb = p.v + r * p.ldv + h * HDR;
    Lk = p.klen[b];
  }
  if (p.qoff) {
    const size_t r = (size_t)p.qoff[b];
    qb = p.q + r * p.ldq + h * HDR;
    ob = p.o + r * p.ldo + h * HDR;
    Lq = p.qlen[b];
  }
  if ((int)blockIdx.x * 128 >= Lq) return;  // block-uniform: the grid is sized for the longest sample

  const float2 cq = attn_f16_scales((unsigned)__builtin_amdgcn_readfirstlane((int)amax_read(p.amax_q, t)));
  const float2 ck = attn_f16_scales((unsigned)__builtin_amdgcn_readfirstlane((int)amax_read(p.amax_k, t)));
  const float2 cv = attn_f16_scales((unsigned)__builtin_amdgcn_readfirstlane((int)amax_read(p.amax_v, t)));
  const float s_q = cq.x, s_k = ck.x, s_v = cv.x;
  const float s_unscale = cq.y * ck.y;                // S = sacc / (s_q s_k)
  const float o_unscale = cv.y * (1.f / 16384.f);     // O = oacc / (s_v 2^14)

  // this lane's query row, times the softmax scale (<= 1: the launcher checks) and s_q, as planes: d = ks*16 + 8*lh + e
  ah16x8 qh[NKS], ql[NKS];
  {
    const int qi = min(q0 + li, Lq - 1);
    const float* qr = qb + (size_t)qi * p.ldq;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qr + ks * 16 + lh * 8);
      const f32x4 c = *reinterpret_cast<const f32x4*>(qr + ks * 16 + lh * 8 + 4);
      const float x[8] = {a.x * p.scale, a.y * p.scale, a.z * p.scale, a.w * p.scale, c.x * p.scale, c.y * p.scale, c.z * p.scale, c.w * p.scale};
      attn_split8(x, s_q, qh[ks], ql[ks]);
    }
  }

  f32x16 oacc[NDC];
#pragma unroll
  for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dc][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  f32x4 rk[LPT], rv[LPT];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int idx = t + 256 * i;
      const int row = idx / (HDR / 4), c4 = idx - row * (HDR / 4);
      const int key = min(k0 + row, Lk - 1);
      rk[i] = *reinterpret_cast<const f32x4*>(kb + (size_t)key * p.ldk + c4 * 4);
      rv[i] = *reinterpret_cast<const f32x4*>(vb + (size_t)key * p.ldv + c4 * 4);
    }
  };
  auto store_kv = [&](int buf) {
    float* Ks = lds + buf * (2 * KT * LDH);
    float* Vs = Ks + KT * LDH;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int idx = t + 256 * i;
      const int row = idx / (HDR / 4), c4 = idx - row * (HDR / 4);
      *reinterpret_cast<f32x4*>(Ks + row * LDH + c4 * 4) = rk[i];
      *reinterpret_cast<f32x4*>(Vs + row * LDH + c4 * 4) = rv[i];
    }
  };

  const int ntiles = (Lk + KT - 1) / KT;
  load_kv(0);
  store_kv(0);
  __syncthreads();
  for (int tt = 0; tt < ntiles; ++tt) {
    const int buf = tt & 1;
    if (tt + 1 < ntiles) load_kv((tt + 1) * KT);
    const float* Ks = lds + buf * (2 * KT * LDH);
    const float* Vs = Ks + KT * LDH;
#pragma unroll
    for (int sub = 0; sub < (q0 < Lq ? KT / 32 : 0); ++sub) {
      const int kbase = tt * KT + sub * 32;
      if (kbase >= Lk) break;  // wave-uniform
      // ---- S^T[key][q] for 32 keys x 32 queries: A = K rows (key li, d = ks*16 + 8*lh + e), B = the query planes
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const float* kr = Ks + (sub * 32 + li) * LDH + lh * 8;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(kr + ks * 16);
        const f32x4 c = *reinterpret_cast<const f32x4*>(kr + ks * 16 + 4);
        const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        ah16x8 kh, kl;
        attn_split8(x, s_k, kh, kl);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], sacc, 0, 0, 0);
      }
      // ---- online softmax for this lane's query
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
        sacc[r] = key >= Lk ? -INFINITY : sacc[r] * s_unscale;
        mt = fmaxf(mt, sacc[r]);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);  // finite: every tile has >= 1 valid key
      const float alpha = __expf(m_run - m_new);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] = __expf(sacc[r] - m_new);
        ps += sacc[r];
      }
      ps += __shfl_xor(ps, 32, 64);
      l_run = l_run * alpha + ps;
      m_run = m_new;
      // ---- P^T planes: registers 8 j .. 8 j + 7 are the lane's eight k values of 16-key step j
      ah16x8 ph[2], pl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float x[8] = {sacc[8 * j], sacc[8 * j + 1], sacc[8 * j + 2], sacc[8 * j + 3], sacc[8 * j + 4], sacc[8 * j + 5], sacc[8 * j + 6], sacc[8 * j + 7]};
        attn_split8(x, 16384.f, ph[j], pl[j]);
      }
      // ---- O^T[d][q] = alpha * O^T + V^T P^T: A = V^T (d = dc*32 + li; the same key order as P's registers)
#pragma unroll
      for (int dc = 0; dc < NDC; ++dc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dc][r] *= alpha;
        const float* vr = Vs + (sub * 32 + 4 * lh) * LDH + dc * 32 + li;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = 8 * j + e;
            x[e] = vr[((r & 3) + 8 * (r >> 2)) * LDH];
          }
          ah16x8 vh, vl;
          attn_split8(x, s_v, vh, vl);
          oacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[j], oacc[dc], 0, 0, 0);
          oacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[j], oacc[dc], 0, 0, 0);
          oacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[j], oacc[dc], 0, 0, 0);
        }
      }
    }
    if (tt + 1 < ntiles) store_kv(buf ^ 1);
    __syncthreads();
  }
  const int qi = q0 + li;
  if (qi < Lq) {
    const float inv = o_unscale / l_run;
    float* orow = ob + (size_t)qi * p.ldo;
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w;
        w.x = oacc[dc][4 * g + 0] * inv;
        w.y = oacc[dc][4 * g + 1] * inv;
        w.z = oacc[dc][4 * g + 2] * inv;
        w.w = oacc[dc][4 * g + 3] * inv;
        if (dc * 32 + 8 * g + 4 * lh < HDR) *reinterpret_cast<f32x4*>(orow + dc * 32 + 8 * g + 4 * lh) = w;
      }
  }
}

void flash_attention(hipStream_t s, const float* q, const float* k, const float* v, float* o, int B, int H, int Lq, int Lk,
                     int hd, int ldq, int ldk, int ldv, int ldo, long bsq, long bsk, long bsv, long bso, float scale,
                     const SeqTab* tab, const unsigned* amax_q, const unsigned* amax_k, const unsigned* amax_v) {
  if (B == 0 || Lq == 0) return;
  YMK_CHECK(Lk > 0, "attention: no keys");
  YMK_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "attention: strides must be multiples of 4");
  AttnP p{q, k, v, o, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, Lq, Lk, H, scale, nullptr, nullptr, nullptr, nullptr, amax_q, amax_k, amax_v};
  if (tab) {
    p.qoff = tab->qoff;
    p.qlen = tab->qlen;
    p.koff = tab->koff;
    p.klen = tab->klen;
    YMK_CHECK((p.qoff == nullptr) == (p.qlen == nullptr) && (p.koff == nullptr) == (p.klen == nullptr), "attention: offset and length tables come in pairs");
  }
  dim3 grid((Lq + 127) / 128, H, B);
  // fp16-split products when the caller knows bounds of q, k and v (their producers' max|x| records) and the thread's
  // convolutions run in that form too (conv_effective_split: the model's "conv_split" / the process-wide option)
  const bool f16 = amax_q && amax_k && amax_v && scale <= 1.f && scale > 0.f && conv_effective_split() == SPLIT_F16X2 && ldq % 4 == 0;
  if (f16 && (hd == 32 || hd == 64 || hd == 96 || hd == 48)) {
    if (hd == 32) hipLaunchKernelGGL(k_flash_attn_f16<32>, grid, dim3(256), 0, s, p);
    else if (hd == 64) hipLaunchKernelGGL(k_flash_attn_f16<64>, grid, dim3(256), 0, s, p);
    else if (hd == 96) hipLaunchKernelGGL(k_flash_attn_f16<96>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_flash_attn_f16<64, 48>), grid, dim3(256), 0, s, p);
    YMK_HIP(hipGetLastError());
    return;
  }
  if (hd == 32) hipLaunchKernelGGL(k_flash_attn<32>, grid, dim3(256), 0, s, p);
  else if (hd == 64) hipLaunchKernelGGL(k_flash_attn<64>, grid, dim3(256), 0, s, p);
  else if (hd == 96) hipLaunchKernelGGL(k_flash_attn<96>, grid, dim3(256), 0, s, p);
  else if (hd == 48) hipLaunchKernelGGL((k_flash_attn<64, 48>), grid, dim3(256), 0, s, p);
  else {
    // any other head dim (46: the legacy parseq-tiny) takes the one-wave-per-query kernel - correct, not fast
    YMK_CHECK(Lk <= 1024 && hd <= 128, "attention: head dim " + std::to_string(hd) + " is only supported for <= 1024 keys per sample");
    small_attention(s, q, k, v, o, B, H, Lq, Lk, hd, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, scale, nullptr, 0, nullptr, 0, tab);
    return;
  }
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- masked attention for a handful of queries
// One wave per (batch, head, query).  mask_qk[q][k] != 0 blocks key k for query q (shared by the
// batch, row stride ld_mask); kpm[b][k] != 0 blocks key k for the whole sample.  Lk <= 1024.
struct SmallAttnP {
  const float *q, *k, *v;
  float* o;
  int ldq, ldk, ldv, ldo;
  long bsq, bsk, bsv, bso;  // bsq may be 0: queries shared by the batch
  int Lq, Lk, H, hd;
  float scale;
  const unsigned char* mask_qk;
  int ld_mask;
  const unsigned char* kpm;
  int ld_kpm;
  const int *koff, *klen;  // ragged keys/values: sample b reads klen[b] rows from row koff[b] (null = strided, Lk)
  const int *qoff, *qlen;  // ragged queries / outputs, likewise (null = strided, Lq)
  int vec4;                // 1: head dim and every base / stride allow 16 B key loads
};
__global__ __launch_bounds__(64) void k_small_attn(SmallAttnP p) {
  __shared__ float prob[1024];
  __shared__ float qs[128];
  const int lane = threadIdx.x;
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  if (p.qoff && qi >= p.qlen[b]) return;  // block-uniform: the grid is sized for the longest sample
  const size_t qrow0 = p.qoff ? (size_t)p.qoff[b] : 0;
  const float* qr = (p.qoff ? p.q + qrow0 * p.ldq : p.q + (size_t)b * p.bsq) + (size_t)qi * p.ldq + h * p.hd;
  for (int d = lane; d < p.hd; d += 64) qs[d] = qr[d] * p.scale;
  __syncthreads();
  const float* kb = p.k + (size_t)b * p.bsk + h * p.hd;
  const float* vb = p.v + (size_t)b * p.bsv + h * p.hd;
  int Lk = p.Lk;
  if (p.koff) {
    kb = p.k + (size_t)p.koff[b] * p.ldk + h * p.hd;
    vb = p.v + (size_t)p.koff[b] * p.ldv + h * p.hd;
    Lk = p.klen[b];
  }
  float mx = -INFINITY;
  for (int k = lane; k < Lk; k += 64) {
    bool blocked = false;
    if (p.mask_qk) blocked = p.mask_qk[(size_t)qi * p.ld_mask + k] != 0;
    if (p.kpm) blocked = blocked || p.kpm[(size_t)b * p.ld_kpm + k] != 0;
    float sc = -INFINITY;
    if (!blocked) {
      const float* kr = kb + (size_t)k * p.ldk;
      float a = 0.f;
      if (p.vec4) {
        for (int d = 0; d < p.hd; d += 4) {
          const float4 kk = *reinterpret_cast<const float4*>(kr + d);
          a += qs[d] * kk.x + qs[d + 1] * kk.y + qs[d + 2] * kk.z + qs[d + 3] * kk.w;
        }
      } else {  // head dim 46 (the legacy parseq-tiny: 368 / 8): rows are only 8 B aligned
        for (int d = 0; d < p.hd; ++d) a += qs[d] * kr[d];
      }
      sc = a;
    }
    prob[k] = sc;
    mx = fmaxf(mx, sc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < Lk; k += 64) {
    const float e = __expf(prob[k] - mx);
    prob[k] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  const float inv = 1.f / sum;
  float* orow = (p.qoff ? p.o + qrow0 * p.ldo : p.o + (size_t)b * p.bso) + (size_t)qi * p.ldo + h * p.hd;
  for (int d = lane; d < p.hd; d += 64) {
    float a = 0.f;
    for (int k = 0; k < Lk; ++k) a += prob[k] * vb[(size_t)k * p.ldv + d];
    orow[d] = a * inv;
  }
}
void small_attention(hipStream_t s, const float* q, const float* k, const float* v, float* o, int B, int H, int Lq, int Lk,
                     int hd, int ldq, int ldk, int ldv, int ldo, long bsq, long bsk, long bsv, long bso, float scale,
                     const unsigned char* mask_qk, int ld_mask, const unsigned char* kpm, int ld_kpm, const SeqTab* tab) {
  if (B == 0 || Lq == 0) return;
  YMK_CHECK(Lk > 0 && Lk <= 1024 && hd <= 128, "small attention: Lk <= 1024, hd <= 128");
  const int vec4 = hd % 4 == 0 && ldk % 4 == 0 && bsk % 4 == 0 && ((uintptr_t)k & 15) == 0;
  SmallAttnP p{q, k, v, o, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, Lq, Lk, H, hd, scale, mask_qk, ld_mask, kpm, ld_kpm,
               tab ? tab->koff : nullptr, tab ? tab->klen : nullptr, tab ? tab->qoff : nullptr, tab ? tab->qlen : nullptr, vec4};
  YMK_CHECK(!tab || (!mask_qk && !kpm), "small attention: ragged batches come without masks");
  hipLaunchKernelGGL(k_small_attn, dim3(Lq, H, B), dim3(64), 0, s, p);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- PARSeq decode bookkeeping
// content row of position `pos` for every sample (models/parseq.py:143-146), then norm_c:
//   pos == 0: sqrt(D) * emb[tok]            pos >= 1: pos_queries[pos-1] + sqrt(D) * emb[tok]
// tokens: [B][ld_tok]; out rows b*out_rows + pos_out.  One wave per (sample, position).
__global__ void k_ctx_embed_ln(const int* __restrict__ tok, int ld_tok, int pos0, int npos, const float* __restrict__ emb,
                               const float* __restrict__ posq, const float* __restrict__ g, const float* __restrict__ be,
                               float eps, float sqrt_d, float* __restrict__ out, int out_rows, int D, int B) {
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= B * npos) return;
  const int b = w / npos, pos = pos0 + (w - b * npos);
  const int token = tok[(size_t)b * ld_tok + pos];
  const float* er = emb + (size_t)token * D;
  const float* pr = pos > 0 ? posq + (size_t)(pos - 1) * D : nullptr;
  float v[16];
  float s = 0.f;
  const int per = (D + 63) / 64;  // <= 16 (D <= 1024)
  for (int i = 0; i < per; ++i) {
    const int c = lane + 64 * i;
    float x = 0.f;
    if (c < D) {
      x = sqrt_d * er[c];
      if (pr) x = pr[c] + x;
      s += x;
    }
    v[i] = x;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = 0; i < per; ++i) {
    const int c = lane + 64 * i;
    if (c < D) q += (v[i] - mean) * (v[i] - mean);
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
  float* orow = out + ((size_t)b * out_rows + pos) * D;
  for (int i = 0; i < per; ++i) {
    const int c = lane + 64 * i;
    if (c < D) orow[c] = (v[i] - mean) * rstd * g[c] + be[c];
  }
}
void ctx_embed_ln(hipStream_t s, const int* tok, int ld_tok, int pos0, int npos, const float* emb, const float* posq,
                  const float* g, const float* be, float eps, float* out, int out_rows, int D, int B) {
  YMK_CHECK(D <= 1024, "ctx_embed: D <= 1024");
  const int waves = B * npos;
  if (waves == 0) return;
  hipLaunchKernelGGL(k_ctx_embed_ln, dim3((waves + 3) / 4), dim3(256), 0, s, tok, ld_tok, pos0, npos, emb, posq, g, be, eps,
                     sqrtf((float)D), out, out_rows, D, B);
  YMK_HIP(hipGetLastError());
}

// ---- one block of 256 threads reduces one row of C logits: arg-max (first maximal index, as torch.argmax) and, on
// request, sum(exp(x - max)).  Thread t owns c = t, t + 256, ...; rows of up to 256 * ROW_RV values are held in
// registers, so all of a thread's loads are in flight together (a greedy step is a chain of such reductions: 28
// dependent L2 round trips per row otherwise) and the softmax sum needs no second pass over memory.  The per-thread
// order and the LDS tree are the same on both paths, so the results do not depend on which one ran.
constexpr int ROW_RV = 32;
template <bool WITH_SUM>
__device__ __forceinline__ void row_reduce_256(const float* __restrict__ row, int C, float* sv, int* si, float& mx, int& am,
                                               float& sum) {
  const int t = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  float v[ROW_RV];
  const bool in_regs = C <= 256 * ROW_RV;
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < ROW_RV; ++i) {
      const int c = t + 256 * i;
      v[i] = c < C ? row[c] : -INFINITY;
    }
#pragma unroll
    for (int i = 0; i < ROW_RV; ++i)
      if (v[i] > best) {  // strict: keeps the lowest index among equal values within a thread
        best = v[i];
        bi = t + 256 * i;
      }
  } else {
    for (int c = t; c < C; c += 256) {
      const float x = row[c];
      if (x > best) {
        best = x;
        bi = c;
      }
    }
  }
  sv[t] = best;
  si[t] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      const float v2 = sv[t + o];
      const int i2 = si[t + o];
      if (v2 > sv[t] || (v2 == sv[t] && i2 < si[t])) {  // torch.argmax returns the first maximal index
        sv[t] = v2;
        si[t] = i2;
      }
    }
    __syncthreads();
  }
  mx = sv[0];
  am = si[0];
  if (WITH_SUM) {
    __syncthreads();
    float acc = 0.f;
    if (in_regs) {
#pragma unroll
      for (int i = 0; i < ROW_RV; ++i)
        if (t + 256 * i < C) acc += expf(v[i] - mx);
    } else {
      for (int c = t; c < C; c += 256) acc += expf(row[c] - mx);
    }
    sv[t] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) sv[t] += sv[t + o];
      __syncthreads();
    }
    sum = sv[0];
  }
}

// greedy step (models/parseq.py:222-250): argmax over C of logits[b][step][:], write the raw argmax,
// the context token for position step+1 (forced to <eos> when a repetition loop is detected), and the
// per-sample flags; one block per sample.  state[b] = {has_eos, rep_done, rep_cut(-1 = none)}.
__global__ __launch_bounds__(256) void k_greedy_step(const float* __restrict__ logits, long ld_b, int C, int step,
                                                     int num_steps, int* __restrict__ tok, int* __restrict__ raw,
                                                     int ld_tok, int* __restrict__ state, int eos_id, int rep_on,
                                                     int period_max, int min_run_p1, int min_repeats,
                                                     int* __restrict__ not_done, const int* __restrict__ prev_not_done,
                                                     int* __restrict__ arrived, int* __restrict__ host_flag,
                                                     const int* __restrict__ gid, int* __restrict__ gopen, int ng, int partials,
                                                     int flags) {
  const int b = blockIdx.x, t = threadIdx.x;
  // speculative step issued after every row already held an <eos>: change nothing (not_done stays 0)
  if (prev_not_done && *prev_not_done == 0) return;
  // grouped forward: a row whose mini-batch finished at an earlier step is frozen - its own loop would have stopped
  // there (models/parseq.py:245-250), so neither tokens nor the repetition detector may advance any further
  const int g = gid ? gid[b] : 0;
  const bool frozen = gid && step > 0 && gopen[(size_t)(step - 1) * ng + g] == 0;
  __shared__ float sv[256];
  __shared__ int si[256];
  if (!frozen && !partials) {  // block-uniform
    float mx, unused;
    int am;
    row_reduce_256<false>(logits + (size_t)b * ld_b, C, sv, si, mx, am, unused);
  }
  if (!frozen && partials) {
    // the vocabulary head already reduced each 64-column tile to (max, column) (EPI_ROWMAX): C <= 256 pairs per row here;
    // (value, lowest column) is a total order, so this tree gives the arg-max row_reduce_256 finds on the full row
    const float2* pr = reinterpret_cast<const float2*>(logits + (size_t)b * ld_b);
    float v = -INFINITY;
    int i = 0x7fffffff;
    if (t < C) {
      const float2 q = pr[t];
      v = q.x;
      i = __float_as_int(q.y);
    }
    sv[t] = v;
    si[t] = i;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) {
        const float v2 = sv[t + o];
        const int i2 = si[t + o];
        if (v2 > sv[t] || (v2 == sv[t] && i2 < si[t])) {
          sv[t] = v2;
          si[t] = i2;
        }
      }
      __syncthreads();
    }
  }
  // the repetition detector below is ONE thread walking back through the row's tokens: from global memory that is a chain of
  // some tens of dependent loads (~0.4 us each: 27 us per step on average over a 101-step loop, as long as the vocabulary
  // head next to it); the block fetches the row once, side by side, and the walk reads LDS
  __shared__ int srow[260];
  const bool cached = num_steps <= 256;
  if (cached && rep_on && !frozen)
    for (int i = t; i <= step; i += 256) srow[i] = tok[(size_t)b * ld_tok + i];
  __syncthreads();
  if (t != 0) return;
  int* st = state + b * 4;
  // a frozen row still owns a slot of the shared token table at every step: keep it a valid id (it lies behind the
  // row's <eos>, so the refinement masks it - but it is used to index the embedding table before the mask applies)
  if (frozen) raw[(size_t)b * ld_tok + step] = eos_id;
  if (!frozen) {
    const int am = si[0];
    raw[(size_t)b * ld_tok + step] = am;
    const int j = step + 1;
    if (j < num_steps) {
      int* trow = tok + (size_t)b * ld_tok;
      int next = am;
      if (rep_on && !st[1] && next != eos_id) {
        // _detect_repeat_onset on seq = trow[1..j] with trow[j] = next (models/parseq.py:108-128)
        trow[j] = next;
        if (cached) srow[j] = next;
        const int* seq = trow + 1;
        auto at = [&](int i) { return cached ? srow[1 + i] : seq[i]; };
        const int n = j;
        for (int pp = 1; pp <= period_max; ++pp) {
          if (n < 2 * pp) continue;
          int k = 1, tpos = n - pp;
          while (tpos - pp >= 0) {
            bool same = true;
            for (int u = 0; u < pp; ++u)
              if (at(tpos - pp + u) != at(n - pp + u)) {
                same = false;
                break;
              }
            if (!same) break;
            ++k;
            tpos -= pp;
          }
          if (k >= (pp == 1 ? min_run_p1 : min_repeats)) {
            st[2] = tpos + pp;  // rep_cut: keep the prefix + one unit
            st[1] = 1;
            next = eos_id;
            break;
          }
        }
      }
      trow[j] = next;
      if (next == eos_id) st[0] = 1;
    }
    if (!st[0]) {
      // "some row is still open" / "some row of mini-batch g is still open": every reader tests these words against zero only.
      // `flags` (the recogniser's loop since round 6): an open row STORES 1 (zero before the loop) - a thousand atomic
      // increments of one word per step queue up at one L2 channel; else (ymk_debug_option("ar_publish", 0): the form of
      // rounds 1-5, kept for A/B runs) the rows count
      if (flags) {
        __hip_atomic_store(not_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gopen) __hip_atomic_store(gopen + (size_t)step * ng + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        atomicAdd(not_done, 1);
        if (gopen) atomicAdd(gopen + (size_t)step * ng + g, 1);
      }
    }
  }
  if (host_flag) {
    // (rounds 1-5) the last block to arrive publishes (rows still open) + 1 to the mapped host word the AR loop polls: a
    // device-scope fence and an atomic per block - 17 of this kernel's 27 us on a multi-XCD part; publish_open replaces it
    __threadfence();
    if (atomicAdd(arrived, 1) == (int)gridDim.x - 1) {
      const int open = atomicAdd(not_done, 0);
      __hip_atomic_store(host_flag, open + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// (any row still open after a step: 0 / 1) + 1 into the mapped host word the AR loop polls, as a launch of its own behind the
// step's greedy kernel.  Inside that kernel the same store needs "the last block to arrive": a device-scope fence and an atomic
// per block, which on this multi-XCD part means L2 write-backs - 17 of the kernel's 27 us (kernel trace with and without:
// 26.7 against 9.6 us over 404 launches).  The kernel boundary orders the words for free.  A speculative step (issued after
// every row already held an <eos>) writes nothing, as before: only steps that really ran report.
__global__ void k_publish_open(const int* __restrict__ not_done, const int* __restrict__ prev_not_done, int* __restrict__ host_flag) {
  if (prev_not_done && *prev_not_done == 0) return;
  __hip_atomic_store(host_flag, *not_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void publish_open(hipStream_t s, const int* not_done, const int* prev_not_done, int* host_flag) {
  hipLaunchKernelGGL(k_publish_open, dim3(1), dim3(1), 0, s, not_done, prev_not_done, host_flag);
  YMK_HIP(hipGetLastError());
}

void greedy_step(hipStream_t s, const float* logits, long ld_b, int C, int step, int num_steps, int* tok, int* raw,
                 int ld_tok, int* state, int eos_id, int rep_on, int period_max, int min_run_p1, int min_repeats,
                 int* not_done, const int* prev_not_done, int* arrived, int* host_flag, int B, const int* gid, int* gopen,
                 int ng, int partials) {
  // host_flag null: the open-row words are flags and publish_open reports the step; non-null: they count and the kernel reports
  YMK_CHECK(!partials || C <= 256, "greedy step: at most 256 partial (max, column) pairs per row");
  hipLaunchKernelGGL(k_greedy_step, dim3(B), dim3(256), 0, s, logits, ld_b, C, step, num_steps, tok, raw, ld_tok, state,
                     eos_id, rep_on, period_max, min_run_p1, min_repeats, not_done, prev_not_done, arrived, host_flag, gid,
                     gopen, ng, partials, host_flag == nullptr ? 1 : 0);
  YMK_HIP(hipGetLastError());
}

// refinement inputs (models/parseq.py:286-292): tok2 = [bos, raw[0..S-2]]; kpm = cumsum(tok2 == eos) > 0
// gid / gsteps (grouped forward): row b belongs to mini-batch gid[b], whose own greedy loop ran gsteps[gid[b]] <= S steps; its
// own forward hands the refinement a context of exactly that many tokens (models/parseq.py:264-278: tgt_in is built from
// logits[:, :-1] of ITS loop), so context positions beyond it are masked for the row - they hold what later steps of OTHER
// mini-batches left in the shared token buffer (for a row stopped by the repetition detector that is an arg-max, not <eos>).
__global__ void k_refine_prep(const int* __restrict__ raw, int ld_tok, int S, int bos_id, int eos_id, int* __restrict__ tok2,
                              unsigned char* __restrict__ kpm, int B, const int* __restrict__ gid,
                              const int* __restrict__ gsteps) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int own = gid ? gsteps[gid[b]] : S;
  bool seen = false;
  for (int t = 0; t < S; ++t) {
    const int v = t == 0 ? bos_id : raw[(size_t)b * ld_tok + t - 1];
    tok2[(size_t)b * ld_tok + t] = v;
    seen = seen || (v == eos_id) || t >= own;
    kpm[(size_t)b * ld_tok + t] = seen ? 1 : 0;
  }
}
void refine_prep(hipStream_t s, const int* raw, int ld_tok, int S, int bos_id, int eos_id, int* tok2, unsigned char* kpm,
                 int B, const int* gid, const int* gsteps) {
  hipLaunchKernelGGL(k_refine_prep, dim3((B + 63) / 64), dim3(64), 0, s, raw, ld_tok, S, bos_id, eos_id, tok2, kpm, B, gid,
                     gsteps);
  YMK_HIP(hipGetLastError());
}

// raw[b][t] = argmax(logits[b][t][:]) for t < S (used between refinement iterations)
__global__ __launch_bounds__(256) void k_row_argmax(const float* __restrict__ logits, int C, int* __restrict__ out) {
  const size_t rowi = blockIdx.x;
  __shared__ float sv[256];
  __shared__ int si[256];
  float mx, unused;
  int am;
  row_reduce_256<false>(logits + rowi * C, C, sv, si, mx, am, unused);
  if (threadIdx.x == 0) out[rowi] = am;
}
void row_argmax(hipStream_t s, const float* logits, int rows, int C, int* out) {
  if (rows == 0) return;
  hipLaunchKernelGGL(k_row_argmax, dim3(rows), dim3(256), 0, s, logits, C, out);
  YMK_HIP(hipGetLastError());
}

// repetition cut (models/parseq.py:301-309): logits[b][cut][:] = -30, [eos] = +30
__global__ void k_rep_cut(float* __restrict__ logits, long ld_b, int C, int S, const int* __restrict__ state, int eos_id) {
  const int b = blockIdx.x;
  const int cut = state[b * 4 + 2];
  if (cut < 0 || cut >= S) return;
  float* row = logits + (size_t)b * ld_b + (size_t)cut * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) row[c] = c == eos_id ? 30.f : -30.f;
}
void rep_cut(hipStream_t s, float* logits, long ld_b, int C, int S, const int* state, int eos_id, int B) {
  hipLaunchKernelGGL(k_rep_cut, dim3(B), dim3(256), 0, s, logits, ld_b, C, S, state, eos_id);
  YMK_HIP(hipGetLastError());
}

// softmax statistics the tokenizer needs (postprocessor/parseq_tokenizer.py:79-87): per row the
// arg-max class and its probability max(softmax(x)) = 1 / sum(exp(x - max)).
__global__ __launch_bounds__(256) void k_row_maxprob(const float* __restrict__ logits, int C, int* __restrict__ ids,
                                                     float* __restrict__ probs) {
  const size_t rowi = blockIdx.x;
  __shared__ float sv[256];
  __shared__ int si[256];
  float mx, sum;
  int am;
  row_reduce_256<true>(logits + rowi * C, C, sv, si, mx, am, sum);
  if (threadIdx.x == 0) {
    ids[rowi] = am;
    probs[rowi] = 1.f / sum;
  }
}
void row_maxprob(hipStream_t s, const float* logits, int rows, int C, int* ids, float* probs) {
  if (rows == 0) return;
  hipLaunchKernelGGL(k_row_maxprob, dim3(rows), dim3(256), 0, s, logits, C, ids, probs);
  YMK_HIP(hipGetLastError());
}

__global__ void k_tile_rows(const float4* __restrict__ src, float4* __restrict__ dst, size_t per, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i % per];
}
// dst[b][r][:] = src[r][:] for b < B
void tile_rows(hipStream_t s, const float* src, int rows, int D, float* dst, int B) {
  const size_t per = (size_t)rows * D / 4, total = per * B;
  if (total == 0) return;
  size_t g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_tile_rows, dim3((int)g), dim3(256), 0, s, (const float4*)src, (float4*)dst, per, total);
  YMK_HIP(hipGetLastError());
}

__global__ void k_init_decode(int* __restrict__ tok, int ld_tok, int* __restrict__ state, int bos_id, int pad_id, int B) {
  const size_t n = (size_t)B * ld_tok;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    tok[i] = (i % ld_tok) == 0 ? bos_id : pad_id;
    if (i < (size_t)B * 4) state[i] = (i & 3) == 2 ? -1 : 0;
  }
}
void init_decode(hipStream_t s, int* tok, int ld_tok, int* state, int bos_id, int pad_id, int B) {
  YMK_CHECK(ld_tok >= 4, "init_decode: ld_tok >= 4");
  const size_t n = (size_t)B * ld_tok;
  if (n == 0) return;
  size_t g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_init_decode, dim3((int)g), dim3(256), 0, s, tok, ld_tok, state, bos_id, pad_id, B);
  YMK_HIP(hipGetLastError());
}

__global__ void k_fill_i32(int* p, int v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void fill_i32(hipStream_t s, int* p, int v, size_t n) {
  if (n == 0) return;
  size_t g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_fill_i32, dim3((int)g), dim3(256), 0, s, p, v, n);
  YMK_HIP(hipGetLastError());
}

}  // namespace ymk
