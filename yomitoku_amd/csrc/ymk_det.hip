// Detection-transformer kernels for RT-DETRv2 (models/layers/rtdetrv2_decoder.py): row-wise
// elementwise helpers, per-image top-k over the encoder tokens (radix select + bitonic sort of the winners in LDS), gathers,
// box refinement, and the multi-scale deformable-attention sampler.
//
// Token rows are stored LEVEL-MAJOR: all images' tokens of level 0, then level 1, then level 2 -
// exactly the NHWC outputs of the three input_proj convolutions laid end to end - so that the
// per-token linear layers run as one GEMM over every row and nothing has to be re-packed into the
// reference's [batch][8400] order.  row(b, k) = B*off[l] + b*hw[l] + (k - off[l]).
#include "ymk_common.h"
#include "ymk_det.h"

namespace ymk {

static inline int gridsz(size_t work, int block = 256, int cap = 4096) {
  size_t g = (work + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g == 0) g = 1;
  return (int)g;
}

__device__ __forceinline__ int token_row(const DetGeom& g, int b, int k) {
  const int l = k >= g.off[2] ? 2 : (k >= g.off[1] ? 1 : 0);
  return g.B * g.off[l] + b * g.hw[l] + (k - g.off[l]);
}

// ---------------------------------------------------------------- out[m][:] = a[m][:] + b[m % rows_mod][:]
__global__ void k_add_bcast(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, int D4,
                            size_t mod_elems, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float4 x = a[i];
    const float4 y = b[i % mod_elems];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    out[i] = x;
  }
}
void add_bcast(hipStream_t s, const float* a, const float* b, int rows_mod, float* out, int M, int D) {
  const size_t total = (size_t)M * D / 4;
  if (total == 0) return;
  hipLaunchKernelGGL(k_add_bcast, dim3(gridsz(total)), dim3(256), 0, s, (const float4*)a, (const float4*)b, (float4*)out,
                     D / 4, (size_t)rows_mod * D / 4, total);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- masked copy: out[row] = valid[k] ? in[row] : 0
__global__ void k_mask_rows(const float4* __restrict__ in, const float* __restrict__ valid, float4* __restrict__ out,
                            DetGeom g, int D4, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / D4);
    // invert the level-major row map
    int l = 0;
    if (row >= g.B * g.off[2]) l = 2;
    else if (row >= g.B * g.off[1]) l = 1;
    const int k = g.off[l] + (row - g.B * g.off[l]) % g.hw[l];
    const float v = valid[k];
    float4 x = in[i];
    x.x *= v; x.y *= v; x.z *= v; x.w *= v;
    out[i] = x;
  }
}
void mask_rows(hipStream_t s, const float* in, const float* valid, float* out, const DetGeom& g, int D) {
  const size_t total = (size_t)g.B * g.ntok * D / 4;
  hipLaunchKernelGGL(k_mask_rows, dim3(gridsz(total)), dim3(256), 0, s, (const float4*)in, valid, (float4*)out, g, D / 4,
                     total);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- top-k per image over max-class logits
// torch.topk(enc_outputs_logits.max(-1).values, K) per image (rtdetrv2_decoder.py:752-756): the K best tokens in rank
// order (descending value, ascending token id among equal values).  One 1024-thread block per image, any token count
// (8400 at 640^2, 18900 at 960^2 - the cell detector), K <= 2048 (300 / 1500):
//   1. radix select on the order-preserving 32-bit image of the value (12 + 12 + 8 bits, 4096-bin LDS histograms):
//      the exact K-th largest value T, how many tokens lie strictly above it, how many of the tokens equal to T are
//      still needed;
//   2. the tokens above T are appended to an LDS list in any order; of the ties, the `need` smallest token ids are taken
//      through a block-wide prefix count in token order (exact also when every value is equal);
//   3. the K (value, id) keys are bitonic sorted in LDS and written out in rank order.
__device__ __forceinline__ unsigned topk_key(const float* __restrict__ logits, int nc, const DetGeom& g, int b, int k) {
  const float* row = logits + (size_t)token_row(g, b, k) * nc;
  float m = row[0];
  for (int c = 1; c < nc; ++c) m = fmaxf(m, row[c]);
  const unsigned u = __float_as_uint(m);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // larger float <=> larger unsigned
}

__global__ __launch_bounds__(1024) void k_topk_tokens(const float* __restrict__ logits, int nc, DetGeom g, int K,
                                                      unsigned* __restrict__ keys_scratch, int* __restrict__ out_idx) {
  constexpr int NT = 1024, BINS = 4096, KMAX = 2048;
  __shared__ unsigned hist[BINS];
  __shared__ unsigned long long sel[KMAX];
  __shared__ unsigned s_prefix, s_mask, s_need, s_count;
  __shared__ int scan[NT];
  const int b = blockIdx.x, t = threadIdx.x, N = g.ntok;
  unsigned* keys = keys_scratch + (size_t)b * N;
  for (int k = t; k < N; k += NT) keys[k] = topk_key(logits, nc, g, b, k);
  if (t == 0) {
    s_prefix = 0;
    s_mask = 0;
    s_need = (unsigned)K;  // tokens still to be taken among those matching the prefix
  }
  __syncthreads();
  // ---- 1. radix select, most significant digits first
  const int shifts[3] = {20, 8, 0}, widths[3] = {12, 12, 8};
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = t; i < BINS; i += NT) hist[i] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix, mask = s_mask;
    const int sh = shifts[pass];
    const unsigned dm = (1u << widths[pass]) - 1u;
    for (int k = t; k < N; k += NT) {
      const unsigned u = keys[k];
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> sh) & dm], 1u);
    }
    __syncthreads();
    if (t == 0) {
      unsigned need = s_need, d = dm;
      for (;; --d) {  // walk the digits from the top: the digit whose bin contains the need-th largest
        const unsigned c = hist[d];
        if (c >= need) break;
        need -= c;
        if (d == 0) break;  // cannot happen (K <= N)
      }
      s_need = need;
      s_prefix = prefix | (d << sh);
      s_mask = mask | (dm << sh);
    }
    __syncthreads();
  }
  const unsigned T = s_prefix, need = s_need;  // exact K-th largest value; ties with T still to take
  // ---- 2. collect: everything above T, then the first `need` ties in token order
  if (t == 0) s_count = 0;
  __syncthreads();
  const int per = (N + NT - 1) / NT;  // contiguous token range of this thread
  const int k0 = t * per, k1 = min(N, k0 + per);
  int ties = 0;
  for (int k = k0; k < k1; ++k) {
    const unsigned u = keys[k];
    if (u > T) sel[atomicAdd(&s_count, 1u)] = ((unsigned long long)(~u) << 32) | (unsigned)k;
    else if (u == T) ++ties;
  }
  scan[t] = ties;
  __syncthreads();
  for (int o = 1; o < NT; o <<= 1) {  // inclusive Hillis-Steele scan of the per-thread tie counts
    const int v = t >= o ? scan[t - o] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  {
    int rank = scan[t] - ties;  // ties before this thread's range
    for (int k = k0; k < k1 && rank < (int)need; ++k)
      if (keys[k] == T) {
        sel[atomicAdd(&s_count, 1u)] = ((unsigned long long)(~T) << 32) | (unsigned)k;
        ++rank;
      }
  }
  __syncthreads();
  // ---- 3. rank order: ascending (~value, id) == descending value, ascending id
  int P = 1;
  while (P < K) P <<= 1;
  for (int i = K + t; i < P; i += NT) sel[i] = ~0ull;
  __syncthreads();
  for (int kk = 2; kk <= P; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int idx = t; idx < P / 2; idx += NT) {
        const int i = 2 * idx - (idx & (j - 1));
        const int q = i + j;
        const unsigned long long a = sel[i], c = sel[q];
        const bool up = (i & kk) == 0;
        if ((a > c) == up) {
          sel[i] = c;
          sel[q] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int r = t; r < K; r += NT) out_idx[(size_t)b * K + r] = (int)(sel[r] & 0xFFFFFFFFu);
}
void topk_tokens(hipStream_t s, const float* logits, int nc, const DetGeom& g, int K, unsigned* keys_scratch, int* out_idx) {
  YMK_CHECK(K >= 1 && K <= 2048 && K <= g.ntok, "topk: 1 <= K <= min(2048, tokens)");
  hipLaunchKernelGGL(k_topk_tokens, dim3(g.B), dim3(1024), 0, s, logits, nc, g, K, keys_scratch, out_idx);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- gathers of the selected queries
// content[b][q][:] = om[row(b, idx)][:];  ref_unact[b][q][:] = bbox[row][:] + anchors[idx][:];  ref = sigmoid(ref_unact)
__global__ void k_gather_queries(const float* __restrict__ om, const float* __restrict__ bbox,
                                 const float* __restrict__ anchors, const int* __restrict__ idx, DetGeom g, int K, int D,
                                 float* __restrict__ content, float* __restrict__ ref) {
  const int q = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int k = idx[(size_t)b * K + q];
  const int row = token_row(g, b, k);
  const float* src = om + (size_t)row * D;
  float* dst = content + ((size_t)b * K + q) * D;
  for (int c = t; c < D; c += blockDim.x) dst[c] = src[c];
  if (t < 4) {
    const float u = bbox[(size_t)row * 4 + t] + anchors[(size_t)k * 4 + t];
    ref[((size_t)b * K + q) * 4 + t] = 1.f / (1.f + expf(-u));
  }
}
void gather_queries(hipStream_t s, const float* om, const float* bbox, const float* anchors, const int* idx,
                    const DetGeom& g, int K, int D, float* content, float* ref) {
  hipLaunchKernelGGL(k_gather_queries, dim3(K, g.B), dim3(64), 0, s, om, bbox, anchors, idx, g, K, D, content, ref);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- box refinement
// box = sigmoid(delta + inverse_sigmoid(ref)), inverse_sigmoid(x) = log(clip(x,1e-5) / clip(1-x,1e-5)) on x in [0,1]
__global__ void k_refine_boxes(const float* __restrict__ delta, const float* __restrict__ ref, float* __restrict__ out,
                               size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float x = fminf(fmaxf(ref[i], 0.f), 1.f);
    const float inv = logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
    out[i] = 1.f / (1.f + expf(-(delta[i] + inv)));
  }
}
void refine_boxes(hipStream_t s, const float* delta, const float* ref, float* out, size_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_refine_boxes, dim3(gridsz(n)), dim3(256), 0, s, delta, ref, out, n);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- multi-scale deformable attention sampling
// One wave per (image, query): lane = head*8 + c4 owns 4 channels of one head.  12 points = 3 levels x 4;
// weights are soft-maxed over the 12; taps follow F.grid_sample(bilinear, zeros, align_corners=False).
__global__ __launch_bounds__(256) void k_deform_sample(const float* __restrict__ offs, const float* __restrict__ attw,
                                                       const float* __restrict__ ref, const float* __restrict__ value,
                                                       int ldv, DetGeom g, int K, float* __restrict__ out) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= g.B * K) return;
  const int lane = threadIdx.x & 63, h = lane >> 3, c4 = lane & 7;
  const int b = w / K;
  const float4 r = *reinterpret_cast<const float4*>(ref + (size_t)w * 4);
  const float* ow = offs + (size_t)w * 192 + h * 24;
  const float* aw = attw + (size_t)w * 96 + h * 12;
  float wts[12];
  float mx = -INFINITY;
#pragma unroll
  for (int p = 0; p < 12; ++p) {
    wts[p] = aw[p];
    mx = fmaxf(mx, wts[p]);
  }
  float sum = 0.f;
#pragma unroll
  for (int p = 0; p < 12; ++p) {
    wts[p] = expf(wts[p] - mx);
    sum += wts[p];
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < 12; ++p) {
    const int l = p >> 2;
    const int H = g.h[l], W = g.w[l];
    const float lx = r.x + ow[2 * p] * 0.25f * r.z * 0.5f;
    const float ly = r.y + ow[2 * p + 1] * 0.25f * r.w * 0.5f;
    const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
    const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
    const float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const float wnw = (1.f - tx) * (1.f - ty), wne = tx * (1.f - ty), wsw = (1.f - tx) * ty, wse = tx * ty;
    const float* base = value + ((size_t)(g.B * g.off[l] + b * g.hw[l])) * ldv + h * 32 + c4 * 4;
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
    auto tap = [&](int yy, int xx, float wt) {
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)(yy * W + xx) * ldv);
        sv.x += v.x * wt; sv.y += v.y * wt; sv.z += v.z * wt; sv.w += v.w * wt;
      }
    };
    tap(y0, x0, wnw);
    tap(y0, x0 + 1, wne);
    tap(y0 + 1, x0, wsw);
    tap(y0 + 1, x0 + 1, wse);
    const float a = wts[p] / sum;
    acc.x += sv.x * a; acc.y += sv.y * a; acc.z += sv.z * a; acc.w += sv.w * a;
  }
  *reinterpret_cast<float4*>(out + (size_t)w * 256 + h * 32 + c4 * 4) = acc;
}
void deform_sample(hipStream_t s, const float* offs, const float* attw, const float* ref, const float* value, int ldv,
                   const DetGeom& g, int K, float* out) {
  const int waves = g.B * K;
  if (waves == 0) return;
  hipLaunchKernelGGL(k_deform_sample, dim3((waves + 3) / 4), dim3(256), 0, s, offs, attw, ref, value, ldv, g, K, out);
  YMK_HIP(hipGetLastError());
}

}  // namespace ymk
