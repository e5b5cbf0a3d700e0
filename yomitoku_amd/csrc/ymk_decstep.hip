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
// pure launch latency.  Here one 256-thread block owns one sample; the small matrices (transposed at
// load so that consecutive lanes read consecutive outputs) stream from L2, vectors live in LDS.
// Used when the decoder width is <= 256 (the lite recogniser); wider models keep the GEMM path.
#include "ymk_common.h"
#include "ymk_decstep.h"

namespace ymk {

constexpr int DMAX = 256, FMAX = 1024, LMAX = 1024, HMAX = 8;

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

// LayerNorm of an LDS vector (in place allowed); all 256 threads call it
__device__ void block_ln(const float* x, const float* g, const float* b, float eps, float* y, int D, float* red) {
  const int t = threadIdx.x;
  float s = 0.f;
  for (int c = t; c < D; c += 256) s += x[c];
  s = w_sum(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)D;
  __syncthreads();
  float q = 0.f;
  for (int c = t; c < D; c += 256) q += (x[c] - mean) * (x[c] - mean);
  q = w_sum(q);
  if ((t & 63) == 0) red[t >> 6] = q;
  __syncthreads();
  const float rstd = 1.f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
  for (int c = t; c < D; c += 256) y[c] = (x[c] - mean) * rstd * g[c] + b[c];
  __syncthreads();
}

// y[o] = act(sum_k Wt[k][o] * x[k] + bias[o]) (+ res[o]); Wt is [K][N] (transposed nn.Linear weight).
// Memory-level parallelism is what matters (the weights stream from L2 at ~0.3 us per dependent
// round trip): a thread owns 4 adjacent outputs (16 B loads, consecutive lanes = consecutive columns),
// the N/4 column groups are replicated G = 256 / (N/4) times along K, 16 loads are kept in flight,
// and the G partial sums meet in LDS (`part`, >= 4 * 256 floats... sized G * N by the caller).
template <int ACT>
__device__ void matvec(const float* __restrict__ Wt, const float* __restrict__ bias, const float* x, int K, int N,
                       const float* res, float* y, float* part) {
  const int t = threadIdx.x;
  const int nv = N >> 2;                       // float4 column groups
  const int passes = (nv + 255) / 256;          // > 1 only when N > 1024
  const int G = passes > 1 ? 1 : 256 / nv;      // K-splits (nv <= 256)
  const int cg = t % nv, g = t / nv;
  if (passes == 1) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < G) {
      const int k0 = (int)((long)K * g / G), k1 = (int)((long)K * (g + 1) / G);
      const float* wp = Wt + (size_t)k0 * N + cg * 4;
      int k = k0;
      for (; k + 16 <= k1; k += 16) {
        float4 w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = *reinterpret_cast<const float4*>(wp + (size_t)u * N);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float xv = x[k + u];
          acc.x = fmaf(w[u].x, xv, acc.x); acc.y = fmaf(w[u].y, xv, acc.y);
          acc.z = fmaf(w[u].z, xv, acc.z); acc.w = fmaf(w[u].w, xv, acc.w);
        }
        wp += (size_t)16 * N;
      }
      for (; k < k1; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(wp);
        const float xv = x[k];
        acc.x = fmaf(w.x, xv, acc.x); acc.y = fmaf(w.y, xv, acc.y); acc.z = fmaf(w.z, xv, acc.z); acc.w = fmaf(w.w, xv, acc.w);
        wp += N;
      }
      *reinterpret_cast<float4*>(part + (size_t)g * N + cg * 4) = acc;
    }
    __syncthreads();
    for (int o = t; o < N; o += 256) {
      float a = 0.f;
      for (int gg = 0; gg < G; ++gg) a += part[(size_t)gg * N + o];
      a += bias[o];
      if (ACT == ACT_GELU) a = 0.5f * a * (1.f + erff(a * 0.70710678118654752440f));
      if (res) a += res[o];
      y[o] = a;
    }
  } else {
    for (int o = t; o < N; o += 256) {
      float a = 0.f;
      for (int k = 0; k < K; ++k) a = fmaf(Wt[(size_t)k * N + o], x[k], a);
      a += bias[o];
      if (ACT == ACT_GELU) a = 0.5f * a * (1.f + erff(a * 0.70710678118654752440f));
      if (res) a += res[o];
      y[o] = a;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_parseq_dec_step(DecStepW W, const int* __restrict__ tok, int ld_tok, int step,
                                                         float* __restrict__ skv, int NS, const float* __restrict__ memkv,
                                                         int L, float* __restrict__ out, const int* __restrict__ prev_not_done) {
  if (prev_not_done && *prev_not_done == 0) return;  // speculative step after the batch finished
  __shared__ float xa[DMAX], xb[DMAX], q[DMAX], kvcur[2 * DMAX], hid[FMAX], sc[HMAX * LMAX], red[8], part[4 * DMAX];
  const int b = blockIdx.x, t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int D = W.D, H = W.H, hd = D / H;
  const float scale = 1.f / sqrtf((float)hd);

  // ---- content row of position `step` -> norm_c -> K|V, appended to the cache
  {
    const int token = tok[(size_t)b * ld_tok + step];
    const float sq = sqrtf((float)D);
    for (int c = t; c < D; c += 256) {
      float v = sq * W.emb[(size_t)token * D + c];
      if (step > 0) v = W.posq[(size_t)(step - 1) * D + c] + v;
      xa[c] = v;
    }
    __syncthreads();
    block_ln(xa, W.ncg, W.ncb, 1e-5f, xb, D, red);
    matvec<ACT_NONE>(W.Wkv_t, W.bkv, xb, D, 2 * D, nullptr, kvcur, part);
    float* row = skv + ((size_t)b * NS + step) * 2 * D;
    for (int c = t; c < 2 * D; c += 256) row[c] = kvcur[c];
  }
  // ---- self attention of query `step` over context rows 0..step
  {
    const int nk = step + 1;
    const float* qs = W.qsa + (size_t)step * D;  // W_q norm_q(pos_queries[step]) + b (batch invariant)
    const float* cache = skv + (size_t)b * NS * 2 * D;
    for (int j = t; j < nk; j += 256) {
      const float* kr = j == step ? kvcur : cache + (size_t)j * 2 * D;
      for (int h = 0; h < H; ++h) {
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a = fmaf(qs[h * hd + d] * scale, kr[h * hd + d], a);
        sc[h * LMAX + j] = a;
      }
    }
    __syncthreads();
    for (int h = wv; h < H; h += 4) {
      float mx = -INFINITY;
      for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, sc[h * LMAX + j]);
      mx = w_max(mx);
      float sm = 0.f;
      for (int j = lane; j < nk; j += 64) {
        const float e = __expf(sc[h * LMAX + j] - mx);
        sc[h * LMAX + j] = e;
        sm += e;
      }
      sm = w_sum(sm);
      if (lane == 0) red[h] = 1.f / sm;
    }
    __syncthreads();
    for (int c = t; c < D; c += 256) {
      const int h = c / hd;
      float a = 0.f;
      for (int j = 0; j < nk; ++j) {
        const float* vr = j == step ? kvcur + D : cache + (size_t)j * 2 * D + D;
        a = fmaf(sc[h * LMAX + j], vr[c], a);
      }
      xa[c] = a * red[h];
    }
    __syncthreads();
    // query = pos_queries[step] + out_proj(attn)
    matvec<ACT_NONE>(W.Wo1_t, W.bo1, xa, D, D, W.posq + (size_t)step * D, q, part);
  }
  // ---- cross attention over the encoder memory (wave-cooperative: 48..64 lanes read one K / V row)
  {
    block_ln(q, W.n1g, W.n1b, 1e-5f, xa, D, red);
    matvec<ACT_NONE>(W.Wq_t, W.bq, xa, D, D, nullptr, xb, part);
    const float* mem = memkv + (size_t)b * L * 2 * D;
    const int nv = D >> 2;  // float4 lanes per row
    const int gl = hd >> 2; // lanes per head
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < nv) {
      qv = *reinterpret_cast<const float4*>(xb + lane * 4);
      qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
    }
    for (int j = wv; j < L; j += 4) {
      float a = 0.f;
      if (lane < nv) {
        const float4 kk = *reinterpret_cast<const float4*>(mem + (size_t)j * 2 * D + lane * 4);
        a = qv.x * kk.x + qv.y * kk.y + qv.z * kk.z + qv.w * kk.w;
      }
      for (int o = gl >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (lane < nv && (lane % gl) == 0) sc[(lane / gl) * LMAX + j] = a;
    }
    __syncthreads();
    for (int h = wv; h < H; h += 4) {
      float mx = -INFINITY;
      for (int j = lane; j < L; j += 64) mx = fmaxf(mx, sc[h * LMAX + j]);
      mx = w_max(mx);
      float sm = 0.f;
      for (int j = lane; j < L; j += 64) {
        const float e = __expf(sc[h * LMAX + j] - mx);
        sc[h * LMAX + j] = e;
        sm += e;
      }
      sm = w_sum(sm);
      if (lane == 0) red[h] = 1.f / sm;
    }
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < nv) {
      const int h = lane / gl;
      for (int j = wv; j < L; j += 4) {
        const float p = sc[h * LMAX + j];
        const float4 vv = *reinterpret_cast<const float4*>(mem + (size_t)j * 2 * D + D + lane * 4);
        acc.x = fmaf(p, vv.x, acc.x); acc.y = fmaf(p, vv.y, acc.y); acc.z = fmaf(p, vv.z, acc.z); acc.w = fmaf(p, vv.w, acc.w);
      }
      *reinterpret_cast<float4*>(part + wv * DMAX + lane * 4) = acc;
    }
    __syncthreads();
    for (int c = t; c < D; c += 256)
      xa[c] = (part[c] + part[DMAX + c] + part[2 * DMAX + c] + part[3 * DMAX + c]) * red[c / hd];
    __syncthreads();
    matvec<ACT_NONE>(W.Wo2_t, W.bo2, xa, D, D, q, q, part);  // each thread reads q[o] before it writes q[o]
  }
  // ---- feed forward
  block_ln(q, W.n2g, W.n2b, 1e-5f, xa, D, red);
  matvec<ACT_GELU>(W.W1_t, W.b1, xa, D, W.F, nullptr, hid, part);
  matvec<ACT_NONE>(W.W2_t, W.b2, hid, W.F, D, q, q, part);
  // ---- decoder.norm -> rows for the vocabulary head
  block_ln(q, W.dng, W.dnb, 1e-5f, xa, D, red);
  for (int c = t; c < D; c += 256) out[(size_t)b * D + c] = xa[c];
}

void parseq_dec_step(hipStream_t s, const DecStepW& W, const int* tok, int ld_tok, int step, float* skv, int NS,
                     const float* memkv, int L, float* out, const int* prev_not_done, int B) {
  YMK_CHECK(W.D <= DMAX && W.D % 4 == 0 && W.F <= FMAX && L <= LMAX && NS <= LMAX && W.H <= HMAX && (W.D / W.H) % 4 == 0,
            "fused decoder step: unsupported geometry");
  hipLaunchKernelGGL(k_parseq_dec_step, dim3(B), dim3(256), 0, s, W, tok, ld_tok, step, skv, NS, memkv, L, out,
                     prev_not_done);
  YMK_HIP(hipGetLastError());
}

bool parseq_dec_step_supported(int D, int H, int F, int L, int NS) {
  return D <= DMAX && D % 4 == 0 && F <= FMAX && L <= LMAX && NS <= LMAX && H <= HMAX && (D / H) % 4 == 0 && (D / 4) <= 64;
}

}  // namespace ymk
