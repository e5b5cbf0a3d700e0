// Internal header of libymk_hip.so (gfx950 only). Not part of the C ABI.
//
// Conventions
//   * activations live in HBM as NHWC fp32 ("pixel-major": one pixel's channels are
//     contiguous) with an explicit pixel stride `ld` so a tensor may be a channel slice
//     of a wider concat buffer;
//   * weights are repacked once at load into the K-major panel layout ymk_conv.hip wants;
//   * nothing on the per-call path allocates: every model owns an Arena sized at first use.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <stdexcept>
#include <cmath>

namespace ymk {

void set_error(const std::string& msg);

struct Error : public std::runtime_error {
  explicit Error(const std::string& m) : std::runtime_error(m) {}
};

#define YMK_HIP(expr)                                                                 \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess) {                                                          \
      throw ::ymk::Error(std::string(#expr) + " failed: " + hipGetErrorString(e__) +  \
                         " at " + __FILE__ + ":" + std::to_string(__LINE__));         \
    }                                                                                 \
  } while (0)

#define YMK_CHECK(cond, msg)                                                          \
  do {                                                                                \
    if (!(cond)) throw ::ymk::Error(std::string("check failed: ") + #cond + ": " + (msg)); \
  } while (0)

// ---------------------------------------------------------------- device allocations and the forwards
// Contract since round 6: a forward (ymk_*_forward) neither allocates nor frees device / pinned memory, builds no weight copy
// and waits for no stream when the caller sized the workspace first (ymk_model_reserve) - everything a forward needs beyond
// its workspace exists when ymk_model_finalize returns.  The fallbacks remain (a caller that never reserves, a precision
// switched through the process-wide option after finalize) and are COUNTED: every hipMalloc / hipFree / hipHostMalloc /
// hipHostFree of the library goes through these wrappers, a forward opens a ForwardScope, and ymk_stat reports
//   "allocs_in_forward"      allocations + frees made while the calling thread was inside a forward
//   "arena_grows_in_forward" of those, workspace growth (the forward's shape was not covered by a reservation)
//   "lazy_panel_builds"      split weight copies built on first use inside a forward instead of at finalize
//   "syncs_in_forward"       hipStreamSynchronize calls those fallbacks made
// (tests/test_serving_gpu.py asserts zeros over a job; tools/stress_call.py records them per run)
void* dev_malloc(size_t bytes);
void dev_free(void* p);
void* host_malloc_pinned(size_t bytes, unsigned flags);
void host_free_pinned(void* p);
void forward_sync(hipStream_t s);  // hipStreamSynchronize, counted when inside a forward
void note_lazy_panel_build();
void note_arena_grow();
bool in_forward();
struct ForwardScope {
  ForwardScope();
  ~ForwardScope();
};
bool runtime_stat(const std::string& key, long long* value);
// diagnostics of tools/stress_call.py, read once from the environment: YMK_DEBUG_LAZY_SPLIT=1 skips the finalize-time build
// of the split weight copies and the max|x| words (the round-5 behaviour: built inside the first forward that asks);
// YMK_DEBUG_HAZARD_NULL_MEMSET=1 re-opens the ordering hazard closed at the end of round 5 (the max|x| words zeroed by a
// null-stream hipMemset that the forward's non-blocking stream does not order with) - to show what it does, never for use
// YMK_DEBUG_HAZARD_NO_FINALIZE_SYNC=1 likewise drops the device synchronisation at the end of ymk_model_finalize (the weights'
// null-stream uploads then order with nothing a first forward does on its non-blocking stream).
bool debug_lazy_split();
bool debug_hazard_null_memset();
bool debug_hazard_no_finalize_sync();

// ---------------------------------------------------------------- max|x| records
// The fp16-split convolutions scale their input by a power of two taken from max|x| over the whole input view
// (ymk_conv_split.hip).  A producer that writes a tensor can leave that maximum behind for free: an "amax record" is 32
// words on 32 separate 128-byte lines (so that the atomicMax of thousands of waves do not queue on one address), zeroed at
// the start of the forward; every wave of the producing kernel folds |v| of what it stores into word (its wave number mod
// 32), the consumer reads the 32 words and takes their maximum.  A tensor without a record (`amax == nullptr`) costs the
// consumer a pass over its input (k_absmax).  Only kernels that write the WHOLE tensor fill a record, and a tensor that is
// modified in place afterwards must drop it (set amax = nullptr).
constexpr int AMAX_LINES = 32, AMAX_LINE_WORDS = 32, AMAX_REC_WORDS = AMAX_LINES * AMAX_LINE_WORDS;

// ---------------------------------------------------------------- tensors
struct Tensor {
  float* p = nullptr;
  int n = 0, h = 0, w = 0, c = 0;
  int ld = 0;  // floats between consecutive pixels (>= c)
  unsigned* amax = nullptr;  // record of max|x| over the tensor, filled by its producer (see above), or null
  // true: the values are stored as the two scaled fp16 planes the fp16-split kernels multiply with, not as fp32 - per pixel and
  // 32-channel slice 32 high halves then 32 low halves (128 bytes, where the fp32 form has its 32 floats: same size, same ld),
  // scaled by the power of two that f16 scales derive from the record `amax` (which then holds the producer's BOUND, below).
  // Only a k x k convolution on the LDS-DMA kernel reads such a tensor (conv_planes_pair_ok decides; ymk_conv_dma.hip).
  bool planes = false;
  size_t pixels() const { return (size_t)n * h * w; }
  Tensor slice_c(int c0, int cn) const {
    Tensor t = *this;
    t.p = p + c0;
    t.c = cn;
    return t;
  }
};

// Bump allocator over one hipMalloc'd slab. reset() at the start of a forward.
class Arena {
 public:
  ~Arena();
  void reserve(size_t bytes);  // grows (re-allocates) if needed; only legal when empty
  void reset() { off_ = 0; }
  float* alloc_f(size_t count);
  void* alloc_bytes(size_t bytes);
  Tensor tensor(int n, int h, int w, int c);
  size_t used() const { return off_; }
  size_t high_water() const { return high_; }
  size_t capacity() const { return cap_; }
  bool dry_run = false;  // when true only measures (returns fake pointers)
  // max|x| records of this forward: `records` of them carved from the arena and zeroed on stream s (call once, right
  // after reset()); amax_next() hands them out, null once they are used up (the consumer then makes its own pass)
  void amax_begin(hipStream_t s, int records);
  unsigned* amax_next();
 private:
  char* base_ = nullptr;
  size_t cap_ = 0, off_ = 0, high_ = 0;
  unsigned* amax_pool_ = nullptr;
  int amax_n_ = 0, amax_used_ = 0;
};

// device side of the records
__device__ __forceinline__ void amax_fold(unsigned& am, float v) { am = max(am, __float_as_uint(fabsf(v))); }
__device__ __forceinline__ void amax_fold4(unsigned& am, const float4 v) {
  am = max(max(am, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
}
// all 64 lanes of a wave call it (t = thread index in the block): one atomicMax per wave, on the wave's line of the record
__device__ __forceinline__ void amax_commit(unsigned* rec, unsigned am, int t) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, o));
  if ((t & 63) == 0 && am != 0u)
    atomicMax(rec + ((blockIdx.x * (blockDim.x >> 6) + (t >> 6)) & (AMAX_LINES - 1)) * AMAX_LINE_WORDS, am);
}
// the record's maximum, in every lane
__device__ __forceinline__ unsigned amax_read(const unsigned* rec, int lane) {
  unsigned m = rec[(lane & (AMAX_LINES - 1)) * AMAX_LINE_WORDS];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  return m;
}

// ---------------------------------------------------------------- activations
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_SIGMOID = 3, ACT_GELU = 4 };

// GELU, exact (erf) form: 0.5 v (1 + erf(v / sqrt 2)) - timm's ViT blocks and the PARSeq decoder (nn.GELU(), parseq_transformer.py:
// 57, 188-204).  erf through erfc(z) = t P(t) exp(-z^2), t = 1 / (1 + p z) (the Abramowitz-Stegun 7.1.26 form with one more
// coefficient, refitted: 8e-9 in exact arithmetic, 4e-7 absolute as evaluated in fp32 - the rounding of a libm-grade erff times v
// is the same size, and both sit at the 2^-21 of one fp16-plane product).  Branch-free: 13 VALU instructions and two
// quarter-rate ones (v_rcp_f32, v_exp_f32) where the library erff costs ~40 with both of its ranges evaluated; the GELU epilogue
// of the ViT fc1 layers was as long as their MFMA work.  Every kernel of the library uses this one function.
__device__ __forceinline__ float gelu_f32(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.39032074649205456f, z, 1.f));
  float q = -0.22690855651422961f;
  q = fmaf(q, t, 0.8816638035254636f);
  q = fmaf(q, t, -0.6277749224408846f);
  q = fmaf(q, t, 0.6443424378640197f);
  q = fmaf(q, t, 0.09342759526711675f);
  q = fmaf(q, t, 0.23524963446014596f);
  const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  const float r = fmaf(-(t * q), e, 1.f);  // erf(|v| / sqrt 2)
  const float hv = 0.5f * v;
  return fmaf(hv, copysignf(r, v), hv);
}

// ---------------------------------------------------------------- conv / linear
// Packed weight panel for the implicit-GEMM kernel (ymk_conv.hip).
//   mode 0 ("chunk"): K runs tap-major, each tap owns ctiles*32 slots (channels zero padded)
//   mode 1 ("tap4") : input has exactly 4 channels per pixel; K = tap*4 + c, padded to 32
struct ConvW {
  float* w = nullptr;      // [cout][kpad] device
  float* scale = nullptr;  // [cout] or null (folded BN gamma/sqrt(var+eps))
  float* bias = nullptr;   // [cout] or null
  int cout = 0, cin = 0, kh = 1, kw = 1;
  int kpad = 0, ctiles = 0, mode = 0;
  // |y| <= pl_a max|x| + pl_b for every output of the layer before a residual (pl_a = max over output channels of |scale|
  // times the L1 norm of the channel's weights, pl_b = max |bias|; ReLU / SiLU / GELU only shrink it): the bound a producer
  // of fp16 PLANES scales its outputs by - the scale must be known before the first output exists
  float pl_a = 0.f, pl_b = 0.f;
};

// EPI_ROWMAX: instead of the tile, every row's (largest value, its column) within the tile: out[m][tile_n] = {max as float,
// column as int bits}, ld = 2 * ntiles_n floats, 64-column tiles - the greedy loop's vocabulary head, whose logits are only
// ever arg-maxed (models/parseq.py:224); ties keep the lowest column, as torch.argmax does.
enum Epi : int { EPI_STORE = 0, EPI_DECONV2X2 = 1, EPI_ROWMAX = 2 };
constexpr int ROWMAX_TILE_N = 64;

struct ConvArgs {
  int stride = 1, pad = 0, dil = 1;
  int stride_w = 0;  // 0: same as stride (vertical); PARSeq-tiny patchify is 4 x 8
  int act = ACT_NONE;
  int epi = EPI_STORE;
  const Tensor* res = nullptr;  // residual added before the activation ...
  bool res_post = false;        // ... or after it (y = res + act(conv))
  // GEMM rows that may be skipped (grouped PARSeq greedy loop): output row m belongs to group row_group[m]; an M tile
  // all of whose rows sit in groups with group_open[g] == 0 is not computed (its outputs keep their old contents)
  const int* row_group = nullptr;
  const int* group_open = nullptr;
  // max|x| records for callers without Tensor objects (gemm): of the input (filled by its producer) / to fill for the output
  const unsigned* amax_in = nullptr;
  unsigned* amax_out = nullptr;
  // LayerNorm(gamma, beta, eps) over the input's channels fused into the operand load (gemm_ln_fused only; ConvK::ln_g)
  const float* ln_g = nullptr;
  const float* ln_b = nullptr;
  float ln_eps = 0.f;
  // the output is written as fp16 planes (Tensor::planes) under the bound pl_a max|x_in| + pl_b; the caller marks the tensor
  bool out_planes = false;
};

// Split-operand convolutions (ymk_conv_split.hip).  "conv_split" codes: 0 = exact fp32 MFMA; 2 / 3 = operands cut into
// 2 / 3 bf16 planes (3 / 6 MFMAs per product tile); 16 = two fp16 planes of the operands scaled by a power of two
// (per tensor for the activations, from a max|x| pass; per output channel for the weights): 3 MFMAs, products good to
// 2^-21 - fp32 grade for every K >= 64, where the fp32 accumulation's own rounding is the larger term.
constexpr int SPLIT_F16X2 = 16;
// Device-side state of that path for ONE model (one host thread, one stream at a time): the split copies of the weight
// panels, built the first time a (panel, code) pair is used, and the two max|x| words the launches alternate between.
class SplitCtx;
struct SplitCtxOwner {
  SplitCtx* get();
  ~SplitCtxOwner();
 private:
  SplitCtx* p_ = nullptr;
};
// The models' default: what a forward runs with when neither its "conv_split" parameter nor the process-wide
// ymk_debug_option("conv_split") / YMK_CONV_SPLIT says otherwise (round 4: the fp16 split; 0 selects the exact fp32 kernels)
constexpr int SPLIT_MODEL_DEFAULT = SPLIT_F16X2;
// Operand precision of the calling thread's conv2d / gemm launches while the scope lives.  A model's forward opens one with
// its "conv_split" parameter, its SplitCtx and SPLIT_MODEL_DEFAULT; a nested scope without a context / default keeps the
// enclosing ones.  Resolution per launch: the scope's split if >= 0, else the process-wide option if set (>= 0), else the
// scope's default (0 outside every scope).  Without a context the split path is not taken.
class ConvSplitScope {
 public:
  explicit ConvSplitScope(int split, SplitCtx* ctx = nullptr, int dflt = -1);
  ~ConvSplitScope();
 private:
  int prev_, prev_dflt_;
  SplitCtx* prev_ctx_;
};

// "conv_split_tile" (ymk_conv_split.hip) for the calling thread's launches while the scope lives; 0 = no force
class ConvSplitTileScope {
 public:
  explicit ConvSplitTileScope(int tile);
  ~ConvSplitTileScope();
 private:
  int prev_;
};

// the "conv_split" code conv2d / gemm launches of the calling thread resolve to right now (scope, process-wide option, default)
int conv_effective_split();
// max|x| over n floats (n % 4 == 0, 16-byte aligned) into word 0 of the zeroed record rec (ymk_conv_split.hip; tests / tools)
void absmax_record(hipStream_t s, const float* x, size_t n, unsigned* rec);

// out must be pre-shaped (n, oh, ow, cout[, ld]); for EPI_DECONV2X2 out is (n, 2h, 2w, cout/4).
void conv2d(hipStream_t s, const Tensor& in, const ConvW& w, const ConvArgs& a, const Tensor& out);

// A 1 x 1 (or any) convolution `w1` whose ONLY consumer is the k x k convolution `w2`: may the tensor between them live as fp16
// planes?  True when the process runs the fp16-split form with the "act_planes" option on, both launches fill the chip, the
// producer lands on a kernel whose epilogue writes planes (not the A-stationary one) with an activation that cannot grow its
// input (none / ReLU / SiLU / GELU), no residual, and the consumer lands on the LDS-DMA kernel.  The producer then gets
// ConvArgs::out_planes, its output Tensor::planes = true.  `in`: the producer's input.
bool conv_planes_pair_ok(const Tensor& in, const ConvW& w1, const ConvArgs& a1, const ConvW& w2, const ConvArgs& a2);

// Row-major GEMM view of the same kernel: out[m][:] = act(A[m][:] . W^T * scale + bias + res[m][:]).
// `res_ld == 0` broadcasts one residual row to every m.
void gemm(hipStream_t s, const float* A, int M, int K, int lda, const ConvW& w, int act, const float* res, int res_ld,
          float* out, int out_ld, const int* row_group = nullptr, const int* group_open = nullptr, int epi = EPI_STORE,
          const unsigned* amax_in = nullptr, unsigned* amax_out = nullptr);

// out[m][:] = act(LayerNorm(X[m][:]; gamma, beta, eps) . W^T * scale + bias + res[m][:]) in ONE launch, the normalised rows never
// written (conv_f16_astat<.., LN>, ymk_conv_astat.hip: fp16-split mode, K = 128 / 192, a launch that fills the chip).
// amax_in: the STATIC bound of the LayerNorm's output (make_layernorm_amax_record).  False = nothing was launched: the caller
// runs layernorm() and gemm() (timm ViT blocks norm1 -> attn.qkv and norm2 -> mlp.fc1, parseq_transformer.py:188-204).
bool gemm_ln_fused(hipStream_t s, const float* X, int M, int K, int ldx, const float* ln_g, const float* ln_b, float ln_eps, const ConvW& w,
                   int act, const float* res, int res_ld, float* out, int out_ld, const unsigned* amax_in, unsigned* amax_out = nullptr);

// x[m][:] <- x[m][:] + fc2(GELU(fc1(LayerNorm(x[m][:])))) in ONE launch, the hidden state never written (k_vit_mlp_f16,
// ymk_vit_mlp.hip: fp16-split mode, D = 192, F = 768, a launch of >= 256 row blocks); ln_bound: the LayerNorm's static output
// bound (the number make_layernorm_amax_record stores).  False = nothing was launched: the caller runs the three launches.
bool vit_mlp_fused(hipStream_t s, float* x, int M, int ld, const float* ln_g, const float* ln_b, float ln_eps, float ln_bound, const ConvW& fc1,
                   const ConvW& fc2);
float layernorm_output_bound(const std::vector<float>& gamma, const std::vector<float>& beta);

// Host-side packing: OIHW fp32 -> panel. `cin_pad4` packs for the tap4 mode (cin<=4).
void pack_conv_weight(const float* oihw, int cout, int cin, int kh, int kw, bool tap4,
                      std::vector<float>& panel, int& kpad, int& ctiles);

inline int conv_out_dim(int in, int k, int stride, int pad, int dil) {
  return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

// ---------------------------------------------------------------- elementwise (ymk_elem.hip)
void nchw3_to_nhwc4(hipStream_t s, const float* in, int n, int h, int w, const Tensor& out);
void maxpool3x3s2(hipStream_t s, const Tensor& in, const Tensor& out);
// out = bilinear_resize(in -> out dims, align_corners=False) [+ add]   (torch F.interpolate)
void upsample_bilinear(hipStream_t s, const Tensor& in, const Tensor& out, const Tensor* add);
void upsample_nearest2x(hipStream_t s, const Tensor& in, const Tensor& out);
void avgpool2x2_ceil(hipStream_t s, const Tensor& in, const Tensor& out);
// per-image per-channel mean over h*w: out[n][c]; scratch = GAP_CHUNKS*n*c floats
constexpr int GAP_CHUNKS = 256;
void global_avgpool(hipStream_t s, const Tensor& in, float* scratch, float* out_nc);
void add_act(hipStream_t s, const Tensor& a, const Tensor& b, int act, const Tensor& out);
// dst record = max(dst record, src record), line by line (a tensor that also holds values copied from src's tensor)
void amax_merge(hipStream_t s, unsigned* dst, const unsigned* src);

// Adaptive-scale-fusion pieces (DBNet++), see ymk_dbnet.cpp
void asf_channel_gate(hipStream_t s, const float* gap_nc, const float* w1, const float* w2, int n,
                      int c, int cmid, float* gate_nc);
void asf_channel_mean(hipStream_t s, const Tensor& x, const float* gate_nc, float* mean_nhw);
void asf_apply(hipStream_t s, const Tensor& x, const float* gate_nc, const float* mean_nhw,
               const float* w_sp3x3, float w_sp1x1, const float* w_att /*[4][c]*/,
               const Tensor& fuse /*4*c ch*/, const Tensor& out);
// final ConvTranspose2d(c->1, 2, 2) + bias + sigmoid: in (n,h,w,c) -> out plane (n, 2h, 2w)
void deconv2x2_to1_sigmoid(hipStream_t s, const Tensor& in, const float* w_c4 /*[c][4]*/, float bias,
                           float* out);

// ---------------------------------------------------------------- weight store
struct HostTensor {
  std::vector<int64_t> dims;
  std::vector<float> data;
  size_t numel() const { return data.size(); }
};

class WeightStore {
 public:
  void put(const std::string& name, const float* data, int ndim, const int64_t* dims);
  const HostTensor& get(const std::string& name) const;
  bool has(const std::string& name) const { return t_.count(name) != 0; }
  void clear() { t_.clear(); }
  size_t size() const { return t_.size(); }
 private:
  std::map<std::string, HostTensor> t_;
};

// Device buffer pool owned by a model (weights). Freed with the model.
class DevicePool {
 public:
  ~DevicePool();
  float* upload(const std::vector<float>& v);
  float* upload(const float* p, size_t n);
  float* alloc(size_t n);
  size_t bytes() const { return bytes_; }
  // every packed panel of the model, as make_conv / make_linear_raw built it (a caller that patches scale / bias afterwards
  // notes the panel again): what Model::prebuild_split builds the split copies from at finalize.  `perm`: the fc2 panels of
  // the ViT blocks the fused MLP kernel runs, which also need the accumulator-order copy.
  void note(const ConvW& c, bool perm = false);
  const std::vector<ConvW>& convs() const { return convs_; }
  const std::vector<ConvW>& perm_convs() const { return perm_; }
 private:
  std::vector<void*> ptrs_;
  size_t bytes_ = 0;
  std::vector<ConvW> convs_, perm_;
};

// conv (+ optional BatchNorm folded to scale/bias) from a state-dict
ConvW make_conv(DevicePool& pool, const WeightStore& ws, const std::string& conv_prefix,
                const std::string& bn_prefix /* "" = none */, bool tap4 = false, float bn_eps = 1e-5f);
// nn.Linear(in,out): weight [out][in] (+bias) -> 1x1 conv panel
ConvW make_linear(DevicePool& pool, const WeightStore& ws, const std::string& prefix, bool has_bias = true);
ConvW make_linear_raw(DevicePool& pool, const float* w_out_in, const float* bias, int out, int in);
// A STATIC max|x| record for the output of LayerNorm(gamma, beta) over d elements: a normalised element is at most sqrt(d - 1)
// in magnitude, so |y| <= sqrt(d) max|gamma| + max|beta| whatever the input - a bound a few times above the real maximum,
// which is all the power-of-two scale of the fp16-split kernels needs (ymk_conv_split.hip).  No pass, no atomics.
unsigned* make_layernorm_amax_record(DevicePool& pool, const std::vector<float>& gamma, const std::vector<float>& beta);

// ---------------------------------------------------------------- models
class Model {
 public:
  virtual ~Model() {}
  virtual const char* kind() const = 0;
  virtual void finalize() = 0;
  // size the workspace once for the largest forward the caller will issue (see ymk_model_reserve in include/ymk.h)
  virtual void reserve(int n, int h, int w, hipStream_t s) = 0;
  WeightStore ws;
  std::map<std::string, double> params;
  double param(const std::string& k, double dflt) const {
    auto it = params.find(k);
    return it == params.end() ? dflt : it->second;
  }
  DevicePool pool;
  Arena arena;
  bool finalized = false;
  // "conv_split" parameter (ymk_model_set_param; may be changed between forwards): operand precision of this model's convs
  int conv_split() const { return (int)param("conv_split", -1); }
  SplitCtxOwner split_ctx;
  // the split copies of every noted panel for the precision(s) this model's forwards resolve to right now, and the context's
  // max|x| words: called by ymk_model_finalize and again when "conv_split" / "conv_split_encoder" change (ymk_conv_split.hip)
  void prebuild_split();
};

Model* create_dbnet();

}  // namespace ymk
