// One greedy step of the PARSeq decoder as ONE kernel (models/parseq.py:204-221 ->
// parseq_transformer.py:69-99 for the query stream of the single decoder layer):
//
//   content row i: sqrt(D) emb[tok] (+ pos_queries[i-1])  -> norm_c -> K|V projection -> cache row i
//   query i = pos_queries[i] + out_proj(self-attention(W_q norm_q(pos_queries[i]); K|V rows 0..i))
//   query  += out_proj(cross-attention(W_q norm1(query); memory K|V))
//   query  += linear2(gelu(linear1(norm2(query))))
//   out     = decoder.norm(query)                                   (the vocabulary head runs after, as a GEMM)
//
// The step-by-step path issues ~13 dependent launches for this, each working on a B x 192 slab:
// pure launch latency.  Here one 1024-thread block (16 waves) owns one sample and walks the ~12
// dependent phases itself; every phase is a chain of L2 round trips (~0.7 us each), so the design rule
// is memory-level parallelism: all 1024 threads issue 16 B loads, 8-16 of them in flight per thread,
// no data-dependent tail loops (ragged ends re-read a valid address and multiply by zero).
// The small matrices are transposed at load ([in][out]: consecutive lanes read consecutive outputs),
// vectors live in LDS.  Used when the decoder width is <= 256 (the lite recogniser); wider models keep
// the GEMM path.
#include <algorithm>
#include <atomic>
#include <string>

#include "ymk_common.h"
#include "ymk_decstep.h"

namespace ymk {

constexpr int DMAX = 256, FMAX = 1024, LMAX = 1024, HMAX = 8;
constexpr int NT = 1024, NWV = NT / 64;

__device__ __forceinline__ float w_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float w_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// LayerNorm of an LDS vector of D <= 256 floats by wave 0 alone (4 elements per lane, no LDS round trip);
// x must be complete (barrier) on entry, y is visible to the block on return.  In place allowed.
__device__ void block_ln(const float* x, const float* __restrict__ g, const float* __restrict__ b, float eps, float* y, int D) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float v[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < D ? x[c] : 0.f;
      s += v[i];
    }
    const float mean = w_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < D) q += (v[i] - mean) * (v[i] - mean);
    const float rstd = 1.f / sqrtf(w_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < D) y[c] = (v[i] - mean) * rstd * g[c] + b[c];
    }
  }
  __syncthreads();
}

// y[o] = act(sum_k Wt[k][o] * x[k] + bias[o]) (+ res[o]); Wt is [K][N] (transposed nn.Linear weight), N <= 4096.
// A thread owns 4 adjacent outputs (16 B loads, consecutive lanes = consecutive columns); the N/4 column
// groups are replicated G = 1024 / (N/4) times along K (group g takes k = g, g + G, ...), 16 loads in
// flight per thread; the G partial sums meet in LDS (`part`, 4 * NT floats).
template <int ACT>
__device__ void matvec(const float* __restrict__ Wt, const float* __restrict__ bias, const float* x, int K, int N,
                       const float* res, float* y, float* part) {
  constexpr int U = 16;
  const int t = threadIdx.x;
  const int nv = N >> 2;
  const int G = NT / nv;
  const int cg = t % nv, g = t / nv;
  if (g < G) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wp = Wt + cg * 4;
    const int per = (K + G - 1) / G;  // k values per group (the last ones may fall past K)
    for (int i0 = 0; i0 < per; i0 += U) {
      float4 w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = min(g + G * (i0 + u), K - 1);
        w[u] = *reinterpret_cast<const float4*>(wp + (size_t)k * N);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = g + G * (i0 + u);
        const float xv = k < K ? x[k] : 0.f;
        acc.x = fmaf(w[u].x, xv, acc.x); acc.y = fmaf(w[u].y, xv, acc.y);
        acc.z = fmaf(w[u].z, xv, acc.z); acc.w = fmaf(w[u].w, xv, acc.w);
      }
    }
    *reinterpret_cast<float4*>(part + (size_t)g * N + cg * 4) = acc;
  }
  __syncthreads();
  for (int o = t; o < N; o += NT) {
    float a = 0.f;
    for (int gg = 0; gg < G; ++gg) a += part[(size_t)gg * N + o];
    a += bias[o];
    if (ACT == ACT_GELU) a = gelu_f32(a);
    if (res) a += res[o];
    y[o] = a;
  }
  __syncthreads();
}

// Multi-head attention of ONE query over n key/value rows in global memory (row r: K at base + r * stride,
// V at + D), wave-cooperative: a wave takes rows wv, wv + 16, ...; lane l < D/4 holds 4 channels, the
// D/(4H) lanes of a head reduce by shuffles.  qv: this lane's 4 query channels, already scaled.
// Result (un-normalised sum and 1/denominator folded in) lands in y[0..D).
__device__ void attend(float4 qv, const float* __restrict__ base, size_t stride, int n, int D, int H, float* sc, float* red,
                       float* part, float* y) {
  constexpr int U = 8;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int nv = D >> 2, hd = D / H, gl = hd >> 2;
  const int ln = min(lane, nv - 1);
  const float* col = base + ln * 4;
  for (int j0 = wv; j0 < n; j0 += NWV * U) {
    float4 kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kk[u] = *reinterpret_cast<const float4*>(col + (size_t)min(j0 + NWV * u, n - 1) * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float a = qv.x * kk[u].x + qv.y * kk[u].y + qv.z * kk[u].z + qv.w * kk[u].w;
      for (int o = gl >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      const int j = j0 + NWV * u;
      if (j < n && lane < nv && (lane % gl) == 0) sc[(lane / gl) * LMAX + j] = a;
    }
  }
  __syncthreads();
  for (int h = wv; h < H; h += NWV) {
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, sc[h * LMAX + j]);
    mx = w_max(mx);
    float sm = 0.f;
    for (int j = lane; j < n; j += 64) {
      const float e = __expf(sc[h * LMAX + j] - mx);
      sc[h * LMAX + j] = e;
      sm += e;
    }
    sm = w_sum(sm);
    if (lane == 0) red[h] = 1.f / sm;
  }
  __syncthreads();
  {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = sc + (ln / gl) * LMAX;
    for (int j0 = wv; j0 < n; j0 += NWV * U) {
      float4 vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        vv[u] = *reinterpret_cast<const float4*>(col + D + (size_t)min(j0 + NWV * u, n - 1) * stride);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + NWV * u;
        const float w = j < n ? p[j] : 0.f;
        acc.x = fmaf(w, vv[u].x, acc.x); acc.y = fmaf(w, vv[u].y, acc.y);
        acc.z = fmaf(w, vv[u].z, acc.z); acc.w = fmaf(w, vv[u].w, acc.w);
      }
    }
    if (lane < nv) *reinterpret_cast<float4*>(part + wv * DMAX + lane * 4) = acc;
  }
  __syncthreads();
  for (int c = t; c < D; c += NT) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) a += part[w * DMAX + c];
    y[c] = a * red[c / hd];
  }
  __syncthreads();
}

__global__ __launch_bounds__(NT) void k_parseq_dec_step(DecStepW W, const int* __restrict__ tok, int ld_tok, int step,
                                                        float* __restrict__ skv, int NS, const float* __restrict__ memkv,
                                                        int L, const int* __restrict__ mem_off,
                                                        const int* __restrict__ mem_len, float* __restrict__ out,
                                                        const int* __restrict__ prev_not_done,
                                                        const int* __restrict__ gid, const int* __restrict__ gopen, int ng) {
  if (prev_not_done && *prev_not_done == 0) return;  // speculative step after the batch finished
  // grouped forward: the row's mini-batch finished at an earlier step - its own loop would not run this step at all
  if (gid && step > 0 && gopen[(size_t)(step - 1) * ng + gid[blockIdx.x]] == 0) return;
  __shared__ __attribute__((aligned(16))) float xa[DMAX], xb[DMAX], q[DMAX], kvcur[2 * DMAX], hid[FMAX], sc[HMAX * LMAX],
      part[4 * NT];
  __shared__ float red[HMAX];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
  const int D = W.D, H = W.H, hd = D / H;
  const int nv = D >> 2, ln = min(lane, nv - 1);
  const float scale = 1.f / sqrtf((float)hd);

  // ---- content row of position `step` -> norm_c -> K|V, appended to the cache
  float* cache = skv + (size_t)b * NS * 2 * D;
  {
    const int token = tok[(size_t)b * ld_tok + step];
    const float sq = sqrtf((float)D);
    for (int c = t; c < D; c += NT) {
      float v = sq * W.emb[(size_t)token * D + c];
      if (step > 0) v = W.posq[(size_t)(step - 1) * D + c] + v;
      xa[c] = v;
    }
    __syncthreads();
    block_ln(xa, W.ncg, W.ncb, 1e-5f, xb, D);
    matvec<ACT_NONE>(W.Wkv_t, W.bkv, xb, D, 2 * D, nullptr, kvcur, part);
    for (int c = t; c < 2 * D; c += NT) cache[(size_t)step * 2 * D + c] = kvcur[c];
    __syncthreads();  // the new row is read back from the cache by the whole block below
  }
  // ---- self attention of query `step` over context rows 0..step; query = pos_queries[step] + out_proj(attn)
  {
    float4 qv = *reinterpret_cast<const float4*>(W.qsa + (size_t)step * D + ln * 4);  // W_q norm_q(pos_queries[step]) + b
    qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
    attend(qv, cache, (size_t)2 * D, step + 1, D, H, sc, red, part, xa);
    matvec<ACT_NONE>(W.Wo1_t, W.bo1, xa, D, D, W.posq + (size_t)step * D, q, part);
  }
  // ---- cross attention over the encoder memory
  {
    block_ln(q, W.n1g, W.n1b, 1e-5f, xa, D);
    matvec<ACT_NONE>(W.Wq_t, W.bq, xa, D, D, nullptr, xb, part);
    float4 qv = *reinterpret_cast<const float4*>(xb + ln * 4);
    qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
    // encoder memory of this sample: row mem_off[b], mem_len[b] rows (ragged mini-batches), else b * L, L rows
    const size_t mrow = mem_off ? (size_t)mem_off[b] : (size_t)b * L;
    attend(qv, memkv + mrow * 2 * D, (size_t)2 * D, mem_len ? mem_len[b] : L, D, H, sc, red, part, xa);
    matvec<ACT_NONE>(W.Wo2_t, W.bo2, xa, D, D, q, q, part);  // each thread reads q[o] before it writes q[o]
  }
  // ---- feed forward
  block_ln(q, W.n2g, W.n2b, 1e-5f, xa, D);
  matvec<ACT_GELU>(W.W1_t, W.b1, xa, D, W.F, nullptr, hid, part);
  matvec<ACT_NONE>(W.W2_t, W.b2, hid, W.F, D, q, q, part);
  // ---- decoder.norm -> rows for the vocabulary head
  block_ln(q, W.dng, W.dnb, 1e-5f, xa, D);
  for (int c = t; c < D; c += NT) out[(size_t)b * D + c] = xa[c];
}

// ---------------------------------------------------------------------------------------------------------------
// The same step for R samples per block (R = 2; 4 as an A/B option), for forwards with more rows than the chip has block slots
// (a grouped forward over the pages of a wave: ~650 rows against 256 resident blocks of the kernel above, i.e. three
// block generations per step).  A block streams the 1.9 MB of decoder matrices ONCE for its R rows (R accumulators per
// thread), LayerNorms run one wave per row, and the 16 waves split into 16 / R waves per row for the two attentions.
// Every value is produced by the same chain of operations in the same order as in the one-row kernel - the matvec's
// K split over thread groups, the attention's key rows dealt to 16 (here: virtual) waves and summed in wave order -
// so the two kernels agree bit for bit (tests/test_parseq_gpu.py), and a mini-batch decodes to the same tokens whether
// it runs alone or inside a grouped forward.
template <int ACT, int R>
__device__ void matvec_rows(const float* __restrict__ Wt, const float* __restrict__ bias, const float* x, int xs, int K, int N,
                            const float* res, int rs, float* y, int ys, float* part) {
  constexpr int U = 8;  // 8 rows of the matrix in flight per thread: R accumulators share the registers
  const int t = threadIdx.x;
  const int nv = N >> 2;
  const int G = NT / nv;
  const int cg = t % nv, g = t / nv;
  if (g < G) {
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wp = Wt + cg * 4;
    const int per = (K + G - 1) / G;
    for (int i0 = 0; i0 < per; i0 += U) {
      float4 w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = min(g + G * (i0 + u), K - 1);
        w[u] = *reinterpret_cast<const float4*>(wp + (size_t)k * N);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = g + G * (i0 + u);
        const int kk = min(k, K - 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float xv = k < K ? x[r * xs + kk] : 0.f;
          acc[r].x = fmaf(w[u].x, xv, acc[r].x); acc[r].y = fmaf(w[u].y, xv, acc[r].y);
          acc[r].z = fmaf(w[u].z, xv, acc[r].z); acc[r].w = fmaf(w[u].w, xv, acc[r].w);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(part + (size_t)r * 4 * NT + (size_t)g * N + cg * 4) = acc[r];
  }
  __syncthreads();
  for (int o = t; o < N * R; o += NT) {
    const int r = o / N, oo = o - r * N;
    const float* pr = part + (size_t)r * 4 * NT;
    float a = 0.f;
    for (int gg = 0; gg < G; ++gg) a += pr[(size_t)gg * N + oo];
    a += bias[oo];
    if (ACT == ACT_GELU) a = gelu_f32(a);
    if (res) a += res[r * rs + oo];
    y[r * ys + oo] = a;
  }
  __syncthreads();
}

// LayerNorm of R LDS vectors, wave r takes row r (same arithmetic as block_ln)
template <int R>
__device__ void rows_ln(const float* x, const float* __restrict__ g, const float* __restrict__ b, float eps, float* y, int D,
                        int rs /* floats between the rows of x and of y */) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wv < R) {
    const float* xr = x + wv * rs;
    float* yr = y + wv * rs;
    float v[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < D ? xr[c] : 0.f;
      s += v[i];
    }
    const float mean = w_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < D) q += (v[i] - mean) * (v[i] - mean);
    const float rstd = 1.f / sqrtf(w_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < D) yr[c] = (v[i] - mean) * rstd * g[c] + b[c];
    }
  }
  __syncthreads();
}

// attention of R queries (one per row of the block) over their own key/value rows: the 16 / R waves of row r do what the
// 16 waves of `attend` do for its single row.  sc: [R][H * lcap] scores, pv: [R][NWV][D] partial sums, red: [R][HMAX].
// n = key rows of THIS wave's row (0 for a row that takes no part); base = its K|V rows.
template <int R>
__device__ void attend_rows(float4 qv, const float* __restrict__ base, size_t stride, int n, int D, int H, float* sc_all,
                            int lcap, float* red_all, float* pv_all, float* y_all, int ys) {
  constexpr int WPR = NWV / R;  // waves per row
  constexpr int U = 8, U2 = 8 / R;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int r = wv / WPR, wr = wv - r * WPR;
  const int nv = D >> 2, hd = D / H, gl = hd >> 2;
  const int ln = min(lane, nv - 1);
  float* sc = sc_all + (size_t)r * H * lcap;
  float* red = red_all + r * HMAX;
  float* pv = pv_all + (size_t)r * NWV * D;
  const float* col = base + ln * 4;
  for (int j0 = wr; j0 < n; j0 += WPR * U) {
    float4 kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kk[u] = *reinterpret_cast<const float4*>(col + (size_t)min(j0 + WPR * u, n - 1) * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float a = qv.x * kk[u].x + qv.y * kk[u].y + qv.z * kk[u].z + qv.w * kk[u].w;
      for (int o = gl >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      const int j = j0 + WPR * u;
      if (j < n && lane < nv && (lane % gl) == 0) sc[(lane / gl) * lcap + j] = a;
    }
  }
  __syncthreads();
  for (int h = wr; h < H; h += WPR) {
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, sc[h * lcap + j]);
    mx = w_max(mx);
    float sm = 0.f;
    for (int j = lane; j < n; j += 64) {
      const float e = __expf(sc[h * lcap + j] - mx);
      sc[h * lcap + j] = e;
      sm += e;
    }
    sm = w_sum(sm);
    if (lane == 0) red[h] = 1.f / sm;
  }
  __syncthreads();
  {
    // virtual wave v = wr + WPR * i keeps the key rows j = v, v + 16, ... of the one-row kernel's wave v, in that order
    float4 acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = sc + (ln / gl) * lcap;
    for (int j0 = 0; j0 < n; j0 += NWV * U2) {
      float4 vv[R][U2];
#pragma unroll
      for (int i = 0; i < R; ++i)
#pragma unroll
        for (int u = 0; u < U2; ++u)
          vv[i][u] = *reinterpret_cast<const float4*>(col + D + (size_t)min(j0 + wr + WPR * i + NWV * u, n - 1) * stride);
#pragma unroll
      for (int i = 0; i < R; ++i)
#pragma unroll
        for (int u = 0; u < U2; ++u) {
          const int j = j0 + wr + WPR * i + NWV * u;
          const float w = j < n ? p[j] : 0.f;
          acc[i].x = fmaf(w, vv[i][u].x, acc[i].x); acc[i].y = fmaf(w, vv[i][u].y, acc[i].y);
          acc[i].z = fmaf(w, vv[i][u].z, acc[i].z); acc[i].w = fmaf(w, vv[i][u].w, acc[i].w);
        }
    }
    if (lane < nv) {
#pragma unroll
      for (int i = 0; i < R; ++i) *reinterpret_cast<float4*>(pv + (size_t)(wr + WPR * i) * D + lane * 4) = acc[i];
    }
  }
  __syncthreads();
  for (int c = t; c < D * R; c += NT) {
    const int rr = c / D, cc = c - rr * D;
    const float* pr = pv_all + (size_t)rr * NWV * D;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) a += pr[w * D + cc];
    y_all[rr * ys + cc] = a * red_all[rr * HMAX + cc / hd];
  }
  __syncthreads();
}

template <int R>
__global__ __launch_bounds__(NT) void k_parseq_dec_step_rows(DecStepW W, const int* __restrict__ tok, int ld_tok, int step,
                                                             float* __restrict__ skv, int NS, const float* __restrict__ memkv,
                                                             int L, const int* __restrict__ mem_off,
                                                             const int* __restrict__ mem_len, float* __restrict__ out,
                                                             const int* __restrict__ prev_not_done,
                                                             const int* __restrict__ gid, const int* __restrict__ gopen, int ng,
                                                             int B, int lcap) {
  if (prev_not_done && *prev_not_done == 0) return;
  constexpr int WPR = NWV / R;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int DS = W.D, FS = W.F;         // row strides of the LDS vectors: the model's widths, not the kernel's maxima
  float* xa = smem;                     // [R][D]
  float* xb = xa + R * DS;              // [R][D]
  float* q = xb + R * DS;               // [R][D]
  float* kvcur = q + R * DS;            // [R][2 D]
  float* hid = kvcur + R * 2 * DS;      // [R][F]
  float* red = hid + R * FS;            // [R][HMAX]
  float* uni = red + R * HMAX;          // matvec: part [R][4 * NT]  |  attention: sc [R][H * lcap] + pv [R][NWV][D]
  __shared__ int live_s[R];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int D = W.D, H = W.H, hd = D / H;
  const int nv = D >> 2, ln = min(lane, nv - 1);
  const float scale = 1.f / sqrtf((float)hd);
  const int b0 = blockIdx.x * R;
  if (t < R) {
    const int b = b0 + t;
    int live = b < B;
    if (live && gid && step > 0 && gopen[(size_t)(step - 1) * ng + gid[b]] == 0) live = 0;  // its mini-batch has finished
    live_s[t] = live;
  }
  __syncthreads();
  bool any = false;
#pragma unroll
  for (int r = 0; r < R; ++r) any = any || live_s[r] != 0;
  if (!any) return;
  const int my_r = wv / WPR;                     // the row this wave serves in the attentions
  const bool my_live = live_s[my_r] != 0;
  const int my_b = min(b0 + my_r, B - 1);
  float* sc = uni;
  float* pv = uni + (size_t)R * H * lcap;

  // ---- content rows of position `step` -> norm_c -> K|V, appended to the caches
  {
    const float sq = sqrtf((float)D);
    for (int c = t; c < D * R; c += NT) {
      const int r = c / D, cc = c - r * D;
      float v = 0.f;
      if (live_s[r]) {
        const int token = tok[(size_t)(b0 + r) * ld_tok + step];
        v = sq * W.emb[(size_t)token * D + cc];
        if (step > 0) v = W.posq[(size_t)(step - 1) * D + cc] + v;
      }
      xa[r * DS + cc] = v;
    }
    __syncthreads();
    rows_ln<R>(xa, W.ncg, W.ncb, 1e-5f, xb, D, DS);
    matvec_rows<ACT_NONE, R>(W.Wkv_t, W.bkv, xb, DS, D, 2 * D, nullptr, 0, kvcur, 2 * DS, uni);
    for (int c = t; c < 2 * D * R; c += NT) {
      const int r = c / (2 * D), cc = c - r * 2 * D;
      if (live_s[r]) skv[((size_t)(b0 + r) * NS + step) * 2 * D + cc] = kvcur[r * 2 * DS + cc];
    }
    __syncthreads();  // the new rows are read back from the caches below
  }
  // ---- self attention of query `step` over context rows 0..step; query = pos_queries[step] + out_proj(attn)
  {
    float4 qv = *reinterpret_cast<const float4*>(W.qsa + (size_t)step * D + ln * 4);
    qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
    attend_rows<R>(qv, skv + (size_t)my_b * NS * 2 * D, (size_t)2 * D, my_live ? step + 1 : 0, D, H, sc, lcap, red, pv, xa, DS);
    matvec_rows<ACT_NONE, R>(W.Wo1_t, W.bo1, xa, DS, D, D, W.posq + (size_t)step * D, 0, q, DS, uni);
  }
  // ---- cross attention over the encoder memory
  {
    rows_ln<R>(q, W.n1g, W.n1b, 1e-5f, xa, D, DS);
    matvec_rows<ACT_NONE, R>(W.Wq_t, W.bq, xa, DS, D, D, nullptr, 0, xb, DS, uni);
    float4 qv = *reinterpret_cast<const float4*>(xb + my_r * DS + ln * 4);
    qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
    const size_t mrow = mem_off ? (size_t)mem_off[my_b] : (size_t)my_b * L;
    const int mlen = mem_len ? mem_len[my_b] : L;
    attend_rows<R>(qv, memkv + mrow * 2 * D, (size_t)2 * D, my_live ? mlen : 0, D, H, sc, lcap, red, pv, xa, DS);
    matvec_rows<ACT_NONE, R>(W.Wo2_t, W.bo2, xa, DS, D, D, q, DS, q, DS, uni);
  }
  // ---- feed forward
  rows_ln<R>(q, W.n2g, W.n2b, 1e-5f, xa, D, DS);
  matvec_rows<ACT_GELU, R>(W.W1_t, W.b1, xa, DS, D, W.F, nullptr, 0, hid, FS, uni);
  matvec_rows<ACT_NONE, R>(W.W2_t, W.b2, hid, FS, W.F, D, q, DS, q, DS, uni);
  // ---- decoder.norm -> rows for the vocabulary head
  rows_ln<R>(q, W.dng, W.dnb, 1e-5f, xa, D, DS);
  for (int c = t; c < D * R; c += NT) {
    const int r = c / D, cc = c - r * D;
    if (live_s[r]) out[(size_t)(b0 + r) * D + cc] = xa[r * DS + cc];
  }
}

static std::atomic<int> g_dec_rows{0};  // ymk_debug_option("dec_rows", R): rows per block of the fused step (0 = by batch size)
bool decstep_debug_option(const std::string& key, int value) {
  if (key != "dec_rows") return false;
  g_dec_rows = value;
  return true;
}

template <int R>
static size_t rows_smem_bytes(int D, int F, int H, int lcap) {
  const size_t fixed = (size_t)R * (3 * D + 2 * D + F + HMAX);
  const size_t uni = std::max((size_t)R * 4 * NT, (size_t)R * ((size_t)H * lcap + (size_t)NWV * D));
  return (fixed + uni) * sizeof(float);
}

template <int R>
static bool launch_rows(hipStream_t s, const DecStepW& W, const int* tok, int ld_tok, int step, float* skv, int NS,
                        const float* memkv, int L, const int* mem_off, const int* mem_len, float* out, const int* prev_not_done,
                        int B, const int* gid, const int* gopen, int ng) {
  const int lcap = (std::max(L, NS) + 63) / 64 * 64;
  const size_t bytes = rows_smem_bytes<R>(W.D, W.F, W.H, lcap);
  // per device: the LDS a workgroup may declare (gfx950: 160 KB; asked of the device, not assumed) minus live_s and
  // alignment slack; 0 = not asked yet, -1 = the attribute could not be raised -> the one-row kernel serves the forward
  static std::atomic<long> lds_max[64];
  int dev = 0;
  YMK_HIP(hipGetDevice(&dev));
  YMK_CHECK(dev >= 0 && dev < 64, "fused decoder step: device index out of range");
  long cap = lds_max[dev].load(std::memory_order_acquire);
  if (cap == 0) {  // idempotent: a race asks twice and stores the same value
    int per_block = 0;
    cap = -1;
    if (hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && per_block > 4096 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_parseq_dec_step_rows<R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            per_block - 64) == hipSuccess)
      cap = per_block - 64;
    (void)hipGetLastError();  // a refused attribute is a routing decision, not an error of this forward
    lds_max[dev].store(cap, std::memory_order_release);
  }
  if (cap < 0 || bytes > (size_t)cap) return false;
  hipLaunchKernelGGL(k_parseq_dec_step_rows<R>, dim3((B + R - 1) / R), dim3(NT), bytes, s, W, tok, ld_tok, step, skv, NS, memkv,
                     L, mem_off, mem_len, out, prev_not_done, gid, gopen, ng, B, lcap);
  YMK_HIP(hipGetLastError());
  return true;
}

void parseq_dec_step(hipStream_t s, const DecStepW& W, const int* tok, int ld_tok, int step, float* skv, int NS,
                     const float* memkv, int L, const int* mem_off, const int* mem_len, float* out, const int* prev_not_done,
                     int B, const int* gid, const int* gopen, int ng) {
  YMK_CHECK(parseq_dec_step_supported(W.D, W.H, W.F, L, NS), "fused decoder step: unsupported geometry");
  // rows per block: one while every row gets its own resident block (256 CUs x 1 block of this register footprint),
  // two when the rows would otherwise queue up behind each other.  Measured per step, serial, 655 rows / ~400 / a handful
  // still open (profiles/README.md): 1 row 175 / 120 / 58 us, 2 rows 148 / 84 / 72 us, 4 rows 183 / 122 / 109 us - with
  // four rows the block's own chain of phases is twice as long, so that variant stays a test / A-B option.
  int rows = g_dec_rows.load(std::memory_order_relaxed);
  if (rows == 0) rows = B > 1152 ? 4 : (B > 288 ? 2 : 1);  // (> 1152 rows - waves of 16 pages - even two rows per block queue up 2.5 deep)
  if (rows >= 4 && launch_rows<4>(s, W, tok, ld_tok, step, skv, NS, memkv, L, mem_off, mem_len, out, prev_not_done, B, gid, gopen, ng))
    return;
  if (rows >= 2 && launch_rows<2>(s, W, tok, ld_tok, step, skv, NS, memkv, L, mem_off, mem_len, out, prev_not_done, B, gid, gopen, ng))
    return;
  hipLaunchKernelGGL(k_parseq_dec_step, dim3(B), dim3(NT), 0, s, W, tok, ld_tok, step, skv, NS, memkv, L, mem_off, mem_len,
                     out, prev_not_done, gid, gopen, ng);
  YMK_HIP(hipGetLastError());
}

bool parseq_dec_step_supported(int D, int H, int F, int L, int NS) {
  if (!(D <= DMAX && D % 4 == 0 && F <= FMAX && F % 4 == 0 && L <= LMAX && NS <= LMAX && H <= HMAX && D % H == 0)) return false;
  const int hd = D / H, gl = hd / 4;
  return hd % 4 == 0 && (gl & (gl - 1)) == 0 && D / 4 <= 64;  // shuffle reduction wants a power-of-two lane group per head
}

}  // namespace ymk
