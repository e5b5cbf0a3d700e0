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
// pure launch latency.  Here one 1024-thread block (16 waves) owns R samples and walks the ~12
// dependent phases itself; every phase is a chain of L2 round trips (~0.7 us each), so the design rule
// is memory-level parallelism: all 1024 threads issue 16 B loads, 8-16 of them in flight per thread,
// no data-dependent tail loops (ragged ends re-read a valid address and multiply by zero).
// The small matrices are transposed at load ([in][out]: consecutive lanes read consecutive outputs),
// vectors live in LDS.  Used when the decoder width is <= 256 (the lite recogniser); wider models keep
// the GEMM path.
//
// Round 6: ONE kernel template for R = 1 ... 4 rows per block, and the two attentions as ONE pass over the key / value rows with a
// running softmax per wave: no score buffer in LDS, and the K and V halves of a row - adjacent in memory - are read together.
// Per step of a 16-page wave (1234 rows, 101 steps; rocprofv3, profiles/r06_decoder_step_*): 153 us with the two-pass
// attention at four rows per block -> 119 us with one pass -> 104 us at TWO rows per block.  (The idea this rewrite started
// from - five or six rows per block so that all rows sit in ONE generation of 256 blocks - measured WORSE: 124 / 148 us; the
// generations overlap anyway, and a block's chain of phases grows with its rows.  The floor is the rows' own K | V stream:
// 1234 x ~277 tokens x 1536 B = 525 MB per step, 83 us at the achievable HBM rate.)
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

// y[r][o] = act(sum_k Wt[k][o] * x[r][k] + bias[o]) (+ res[r][o]) for the R rows of the block; Wt is [K][N] (transposed
// nn.Linear weight), N <= 4096.  A thread owns 4 adjacent outputs (16 B loads, consecutive lanes = consecutive columns); the
// N/4 column groups are replicated G = 1024 / (N/4) times along K (group g takes k = g, g + G, ...), 8 loads in flight per
// thread and R accumulators on them: a block streams the 1.9 MB of decoder matrices ONCE for its R rows; the G partial
// sums meet in LDS (`part`, 4 * NT floats per row).  Every value is the same chain of operations whatever R is.
template <int ACT, int R>
__device__ void matvec_rows(const float* __restrict__ Wt, const float* __restrict__ bias, const float* x, int xs, int K, int N,
                            const float* res, int rs, float* y, int ys, float* part) {
  constexpr int U = 8;  // matrix rows in flight per thread (the order of the sums does not depend on it)
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

// LayerNorm of R LDS vectors of D <= 256 floats, wave r takes row r (4 elements per lane, no LDS round trip); x must be
// complete (barrier) on entry, y is visible to the block on return.  In place allowed.
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

// Multi-head attention of R queries (one per row of the block), each over its OWN n key / value rows in global memory (row j:
// K at base + j * stride, V at + D), in ONE pass with a running softmax.  The key rows of a query are dealt to 16 VIRTUAL
// waves (virtual wave v takes j = v, v + 16, ...), four rows at a time (their K and V halves in flight together: 32 registers of
// the 128 a 1024-thread block leaves a lane): per batch of four one maximum, one rescale of the
// running sums, four weights - so what a virtual wave computes does not depend on R; the WPR = 16 / R (rounded down)
// physical waves of a row walk its virtual waves v = wr, wr + WPR, ... and leave each one's (maximum, denominator, weighted V
// sum) in LDS, where they are merged in the order v = 0 ... 15.  Hence every R gives the same bits
// (tests/test_parseq_gpu.py::test_fused_step_rows_per_block_agree_bit_for_bit).  Lane l < D/4 holds 4 channels; the
// D/(4H) lanes of a head share their dot products by shuffles.  qv: this lane's 4 query channels, already scaled.
// pv: [R][NWV][D] weighted sums, ml: [R][NWV][2 * HMAX] (maximum, denominator) per head.  n = key rows of THIS wave's row
// (0 for a row that takes no part, and for the waves beyond R * WPR, which serve no row).
template <int R>
__device__ void attend_rows(float4 qv, const float* __restrict__ base, size_t stride, int n, int D, int H, float* pv_all,
                            float* ml_all, float* y_all, int ys) {
  constexpr int WPR = NWV / R;
  constexpr int U = 4;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int r = wv / WPR, wr = wv - r * WPR;
  const int nv = D >> 2, hd = D / H, gl = hd >> 2;
  const int ln = min(lane, nv - 1);
  if (r < R) {
    float* pv = pv_all + (size_t)r * NWV * D;
    float* ml = ml_all + (size_t)r * NWV * 2 * HMAX;
    const float* col = base + ln * 4;
    for (int v = wr; v < NWV; v += WPR) {
      float m = -INFINITY, l = 0.f;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j0 = v; j0 < n; j0 += NWV * U) {
        float4 kk[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float* row = col + (size_t)min(j0 + NWV * u, n - 1) * stride;
          kk[u] = *reinterpret_cast<const float4*>(row);
          vv[u] = *reinterpret_cast<const float4*>(row + D);
        }
        float sc[U];
        float mx = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float a = qv.x * kk[u].x + qv.y * kk[u].y + qv.z * kk[u].z + qv.w * kk[u].w;
          for (int o = gl >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
          sc[u] = j0 + NWV * u < n ? a : -INFINITY;
          mx = fmaxf(mx, sc[u]);
        }
        const float keep = __expf(m - mx);  // (m = -inf on the first batch: 0; mx is finite, row j0 exists)
        l *= keep;
        acc.x *= keep; acc.y *= keep; acc.z *= keep; acc.w *= keep;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float p = __expf(sc[u] - mx);
          l += p;
          acc.x = fmaf(p, vv[u].x, acc.x); acc.y = fmaf(p, vv[u].y, acc.y);
          acc.z = fmaf(p, vv[u].z, acc.z); acc.w = fmaf(p, vv[u].w, acc.w);
        }
        m = mx;
      }
      if (lane < nv) {
        *reinterpret_cast<float4*>(pv + (size_t)v * D + lane * 4) = acc;
        if ((lane % gl) == 0) {
          ml[v * 2 * HMAX + lane / gl] = m;
          ml[v * 2 * HMAX + HMAX + lane / gl] = l;
        }
      }
    }
  }
  __syncthreads();
  for (int c = t; c < D * R; c += NT) {
    const int rr = c / D, cc = c - rr * D, h = cc / hd;
    const float* pr = pv_all + (size_t)rr * NWV * D;
    const float* mr = ml_all + (size_t)rr * NWV * 2 * HMAX;
    float mx = -INFINITY;
#pragma unroll
    for (int v = 0; v < NWV; ++v) mx = fmaxf(mx, mr[v * 2 * HMAX + h]);
    float den = 0.f, a = 0.f;
    if (mx > -INFINITY) {
#pragma unroll
      for (int v = 0; v < NWV; ++v) {
        const float w = __expf(mr[v * 2 * HMAX + h] - mx);  // a virtual wave without rows: exp(-inf) = 0
        den = fmaf(mr[v * 2 * HMAX + HMAX + h], w, den);
        a = fmaf(pr[v * D + cc], w, a);
      }
    }
    y_all[rr * ys + cc] = den > 0.f ? a / den : 0.f;  // (a row that takes no part: nothing reads it)
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
                                                             int B) {
  if (prev_not_done && *prev_not_done == 0) return;  // speculative step after the batch finished
  constexpr int WPR = NWV / R;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int DS = W.D, FS = W.F;         // row strides of the LDS vectors: the model's widths, not the kernel's maxima
  float* xa = smem;                     // [R][D]
  float* xb = xa + R * DS;              // [R][D]
  float* q = xb + R * DS;               // [R][D]
  float* kvcur = q + R * DS;            // [R][2 D]
  float* hid = kvcur + R * 2 * DS;      // [R][F]
  float* uni = hid + R * FS;            // matvec: part [R][4 * NT]  |  attention: pv [R][NWV][D] + ml [R][NWV][2 HMAX]
  __shared__ int live_s[R];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int D = W.D, H = W.H, hd = D / H;
  const int nv = D >> 2, ln = min(lane, nv - 1);
  const float scale = 1.f / sqrtf((float)hd);
  const int b0 = blockIdx.x * R;
  if (t < R) {
    const int b = b0 + t;
    int live = b < B;
    // grouped forward: the row's mini-batch finished at an earlier step - its own loop would not run this step at all
    if (live && gid && step > 0 && gopen[(size_t)(step - 1) * ng + gid[b]] == 0) live = 0;
    live_s[t] = live;
  }
  __syncthreads();
  bool any = false;
#pragma unroll
  for (int r = 0; r < R; ++r) any = any || live_s[r] != 0;
  if (!any) return;
  const int my_r = min(wv / WPR, R - 1);         // the row this wave serves in the attentions (waves beyond R * WPR: none)
  const bool my_live = wv / WPR < R && live_s[my_r] != 0;
  const int my_b = min(b0 + my_r, B - 1);
  float* pv = uni;
  float* ml = uni + (size_t)R * NWV * D;

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
    float4 qv = *reinterpret_cast<const float4*>(W.qsa + (size_t)step * D + ln * 4);  // W_q norm_q(pos_queries[step]) + b
    qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
    attend_rows<R>(qv, skv + (size_t)my_b * NS * 2 * D, (size_t)2 * D, my_live ? step + 1 : 0, D, H, pv, ml, xa, DS);
    matvec_rows<ACT_NONE, R>(W.Wo1_t, W.bo1, xa, DS, D, D, W.posq + (size_t)step * D, 0, q, DS, uni);
  }
  // ---- cross attention over the encoder memory
  {
    rows_ln<R>(q, W.n1g, W.n1b, 1e-5f, xa, D, DS);
    matvec_rows<ACT_NONE, R>(W.Wq_t, W.bq, xa, DS, D, D, nullptr, 0, xb, DS, uni);
    float4 qv = *reinterpret_cast<const float4*>(xb + my_r * DS + ln * 4);
    qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
    // encoder memory of this sample: row mem_off[b], mem_len[b] rows (ragged mini-batches), else b * L, L rows
    const size_t mrow = mem_off ? (size_t)mem_off[my_b] : (size_t)my_b * L;
    const int mlen = mem_len ? mem_len[my_b] : L;
    attend_rows<R>(qv, memkv + mrow * 2 * D, (size_t)2 * D, my_live ? mlen : 0, D, H, pv, ml, xa, DS);
    matvec_rows<ACT_NONE, R>(W.Wo2_t, W.bo2, xa, DS, D, D, q, DS, q, DS, uni);  // each thread reads q[o] before it writes q[o]
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

constexpr int MAX_ROWS = 4;  // LDS: 23.3 KB per row at D = 192, F = 768 (the vectors and the matvec's partial sums) of 160 KB

template <int R>
static size_t rows_smem_bytes(int D, int F) {
  const size_t fixed = (size_t)R * (3 * D + 2 * D + F);
  const size_t uni = std::max((size_t)R * 4 * NT, (size_t)R * ((size_t)NWV * D + (size_t)NWV * 2 * HMAX));
  return (fixed + uni) * sizeof(float);
}

template <int R>
static bool launch_rows(hipStream_t s, const DecStepW& W, const int* tok, int ld_tok, int step, float* skv, int NS,
                        const float* memkv, int L, const int* mem_off, const int* mem_len, float* out, const int* prev_not_done,
                        int B, const int* gid, const int* gopen, int ng) {
  const size_t bytes = rows_smem_bytes<R>(W.D, W.F);
  // per device: the LDS a workgroup may declare (gfx950: 160 KB; asked of the device, not assumed) minus live_s and
  // alignment slack; 0 = not asked yet, -1 = the attribute could not be raised -> fewer rows per block serve the forward
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
                     L, mem_off, mem_len, out, prev_not_done, gid, gopen, ng, B);
  YMK_HIP(hipGetLastError());
  return true;
}

void parseq_dec_step(hipStream_t s, const DecStepW& W, const int* tok, int ld_tok, int step, float* skv, int NS,
                     const float* memkv, int L, const int* mem_off, const int* mem_len, float* out, const int* prev_not_done,
                     int B, const int* gid, const int* gopen, int ng) {
  YMK_CHECK(parseq_dec_step_supported(W.D, W.H, W.F, L, NS), "fused decoder step: unsupported geometry");
  // rows per block, by measurement (one launch per step of the bench's recogniser forwards, rocprofv3; us per step):
  //   rows   150 / 300     655           1234                  2048
  //   R = 1  51 / 52.5
  //   R = 2       61.3     79.2          104.0                 163.3
  //   R = 3                83.0          116.8                 157.1
  //   R = 4                92.7          119.2                 154.4
  // a block streams the 1.9 MB of decoder matrices once for its R rows, but its chain of phases grows with them and the
  // attentions of a row get 16 / R waves: two rows per block from ~320 rows on, four only where the rows outnumber the
  // chip's blocks six to one
  int rows = g_dec_rows.load(std::memory_order_relaxed);
  if (rows <= 0) rows = B > 1536 ? 4 : (B > 320 ? 2 : 1);
  rows = std::min(rows, MAX_ROWS);
#define YMK_TRY_ROWS(R) \
  if (rows >= R && launch_rows<R>(s, W, tok, ld_tok, step, skv, NS, memkv, L, mem_off, mem_len, out, prev_not_done, B, gid, gopen, ng)) return
  YMK_TRY_ROWS(4);
  YMK_TRY_ROWS(3);
  YMK_TRY_ROWS(2);
  YMK_TRY_ROWS(1);
#undef YMK_TRY_ROWS
  throw Error("fused decoder step: not even one row per block fits the LDS this device grants a workgroup");
}

bool parseq_dec_step_supported(int D, int H, int F, int L, int NS) {
  if (!(D <= DMAX && D % 4 == 0 && F <= FMAX && F % 4 == 0 && L <= LMAX && NS <= LMAX && H <= HMAX && D % H == 0)) return false;
  const int hd = D / H, gl = hd / 4;
  return hd % 4 == 0 && (gl & (gl - 1)) == 0 && D / 4 <= 64;  // shuffle reduction wants a power-of-two lane group per head
}

}  // namespace ymk
