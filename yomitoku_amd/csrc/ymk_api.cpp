// extern "C" surface of libymk_hip.so (declared in include/ymk.h).
#include "../../include/ymk.h"
#include "ymk_common.h"
#include "ymk_seq.h"

namespace ymk {
const std::string& last_error();
void dbnet_forward(Model* m, const float* x, int n, int h, int w, float* prob, hipStream_t s);
void parseq_forward(Model* m, const float* x, int B, int W, float* logits, int* out_len, int* ar_steps, hipStream_t s);
void parseq_forward_groups(Model* m, const float* const* x, const int* b, const int* w, int ng, float* logits, int* out_len,
                           int* ar_steps, hipStream_t s);
void parseq_dims(Model* m, int* num_steps, int* num_classes);
Model* create_parseq();
Model* create_rtdetr();
void rtdetr_forward(Model* m, const float* x, int B, int H, int W, float* logits, float* boxes, hipStream_t s);
void row_maxprob(hipStream_t s, const float* logits, int rows, int C, int* ids, float* probs);
void prof_begin();
void prof_end(double* ms, double* flop, int64_t* launches);
double prof_bytes();
int64_t prof_launch_table(double* ms, double* flop, double* bytes, double* products, int64_t capacity);
bool gemm_takes_astat(int M, int K, const ConvW& w, bool with_res, int ld);
bool conv_debug_option(const std::string& key, int value);
bool parseq_debug_option(const std::string& key, int value);
bool decstep_debug_option(const std::string& key, int value);
bool conv_split_debug_option(const std::string& key, int value);
bool conv_split_stat(const std::string& key, long long* value);
void amax_check_counters(long long* out4);
}  // namespace ymk

struct ymk_model {
  ymk::Model* impl = nullptr;
  int device = 0;
};

// roctx ranges around the forwards and the pre-processing entry points ("rocprofv3 / roctx ranges around each
// C-ABI call"), so that a `rocprofv3 --marker-trace` timeline shows which call a kernel belongs to.  Off unless YMK_ROCTX=1: the
// library links only libamdhip64, the marker library (librocprofiler-sdk-roctx.so, else libroctx64.so) is opened at the first
// range and a box without it simply gets no ranges.
#include <dlfcn.h>
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* on = std::getenv("YMK_ROCTX");
    if (on == nullptr || on[0] == '\0' || on[0] == '0') return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr;
        pop = nullptr;
      }
    }
  }
};
struct RoctxRange {
  explicit RoctxRange(const char* name) {
    static Roctx r;
    pop_ = r.pop;
    if (r.push) (void)r.push(name);
    else pop_ = nullptr;
  }
  ~RoctxRange() {
    if (pop_) (void)pop_();
  }
  int (*pop_)() = nullptr;
};
}  // namespace

#define YMK_API_BEGIN try {
#define YMK_API_END                                 \
  return 0;                                         \
  }                                                 \
  catch (const std::exception& e) {                 \
    ymk::set_error(e.what());                       \
    return 1;                                       \
  }                                                 \
  catch (...) {                                     \
    ymk::set_error("unknown C++ exception");        \
    return 2;                                       \
  }

extern "C" {

int ymk_version(void) { return 100; }

const char* ymk_last_error(void) { return ymk::last_error().c_str(); }

int ymk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

ymk_model* ymk_model_create(const char* kind, int device) {
  try {
    YMK_CHECK(kind != nullptr, "kind is null");
    YMK_HIP(hipSetDevice(device));
    std::string k(kind);
    ymk::Model* impl = nullptr;
    if (k == "dbnet") impl = ymk::create_dbnet();
    else if (k == "parseq") impl = ymk::create_parseq();
    else if (k == "rtdetr") impl = ymk::create_rtdetr();
    else throw ymk::Error("unknown model kind: " + k);
    auto* m = new ymk_model();
    m->impl = impl;
    m->device = device;
    return m;
  } catch (const std::exception& e) {
    ymk::set_error(e.what());
    return nullptr;
  }
}

void ymk_model_destroy(ymk_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  delete m->impl;
  delete m;
}

int ymk_model_set_param(ymk_model* m, const char* key, double value) {
  YMK_API_BEGIN
  YMK_CHECK(m && key, "null argument");
  m->impl->params[key] = value;
  const std::string k(key);
  if (m->impl->finalized && (k == "conv_split" || k == "conv_split_encoder")) {  // a precision switched between forwards: its copies now
    YMK_HIP(hipSetDevice(m->device));
    m->impl->prebuild_split();
  }
  YMK_API_END
}

int ymk_model_set_tensor(ymk_model* m, const char* name, const float* host_data, int ndim, const int64_t* dims) {
  YMK_API_BEGIN
  YMK_CHECK(m && name && host_data && (ndim == 0 || dims), "null argument");
  YMK_CHECK(!m->impl->finalized, "model already finalized");
  m->impl->ws.put(name, host_data, ndim, dims);
  YMK_API_END
}

int ymk_model_finalize(ymk_model* m) {
  YMK_API_BEGIN
  RoctxRange roctx_range("ymk_model_finalize");
  YMK_CHECK(m, "null model");
  YMK_HIP(hipSetDevice(m->device));
  m->impl->finalize();
  // the split copies of the weight panels for the precision the model runs, and its max|x| words: built here, so that no
  // forward ever allocates, builds or waits for them (ymk_common.h: "device allocations and the forwards")
  m->impl->prebuild_split();
  // finalize() uploads the weights with synchronous copies from pageable host memory and fills a few words with hipMemset: all
  // work of the null stream, which the forwards' non-blocking streams do not order with.  Whatever the runtime's guarantee at
  // the return of such a call (staged vs. landed), after this line every byte is in place before a first forward can start
  if (!ymk::debug_hazard_no_finalize_sync()) YMK_HIP(hipDeviceSynchronize());
  YMK_API_END
}

int ymk_model_reserve(ymk_model* m, int n, int h, int w, void* stream) {
  YMK_API_BEGIN
  RoctxRange roctx_range("ymk_model_reserve");
  YMK_CHECK(m && m->impl, "null argument");
  YMK_HIP(hipSetDevice(m->device));
  m->impl->reserve(n, h, w, (hipStream_t)stream);
  YMK_API_END
}

int64_t ymk_model_weight_bytes(const ymk_model* m) { return m ? (int64_t)m->impl->pool.bytes() : -1; }
int64_t ymk_model_workspace_bytes(const ymk_model* m) { return m ? (int64_t)m->impl->arena.capacity() : -1; }

int ymk_dbnet_forward(ymk_model* m, const float* x_dev, int n, int h, int w, float* prob_dev, void* stream) {
  YMK_API_BEGIN
  RoctxRange roctx_range("ymk_dbnet_forward");
  YMK_CHECK(m && x_dev && prob_dev, "null argument");
  YMK_HIP(hipSetDevice(m->device));
  ymk::dbnet_forward(m->impl, x_dev, n, h, w, prob_dev, (hipStream_t)stream);
  YMK_API_END
}

int ymk_parseq_dims(ymk_model* m, int* num_steps, int* num_classes) {
  YMK_API_BEGIN
  YMK_CHECK(m && num_steps && num_classes, "null argument");
  ymk::parseq_dims(m->impl, num_steps, num_classes);
  YMK_API_END
}

int ymk_parseq_forward(ymk_model* m, const float* x_dev, int b, int w, float* logits_dev, int* out_len, int* ar_steps,
                       void* stream) {
  YMK_API_BEGIN
  RoctxRange roctx_range("ymk_parseq_forward");
  YMK_CHECK(m && x_dev && logits_dev && out_len && ar_steps, "null argument");
  YMK_HIP(hipSetDevice(m->device));
  ymk::parseq_forward(m->impl, x_dev, b, w, logits_dev, out_len, ar_steps, (hipStream_t)stream);
  YMK_API_END
}

int ymk_parseq_forward_groups(ymk_model* m, const float* const* x_dev, const int* b, const int* w, int n_groups,
                              float* logits_dev, int* out_len, int* ar_steps, void* stream) {
  YMK_API_BEGIN
  RoctxRange roctx_range("ymk_parseq_forward_groups");
  YMK_CHECK(m && x_dev && b && w && logits_dev && out_len && ar_steps, "null argument");
  YMK_HIP(hipSetDevice(m->device));
  ymk::parseq_forward_groups(m->impl, x_dev, b, w, n_groups, logits_dev, out_len, ar_steps, (hipStream_t)stream);
  YMK_API_END
}

int ymk_parseq_token_stats(const float* logits_dev, int rows, int num_classes, int* ids_dev, float* probs_dev,
                           void* stream) {
  YMK_API_BEGIN
  RoctxRange roctx_range("ymk_parseq_token_stats");
  YMK_CHECK(logits_dev && ids_dev && probs_dev, "null argument");
  ymk::row_maxprob((hipStream_t)stream, logits_dev, rows, num_classes, ids_dev, probs_dev);
  YMK_API_END
}

int ymk_rtdetr_forward(ymk_model* m, const float* x_dev, int b, int h, int w, float* logits_dev, float* boxes_dev,
                       void* stream) {
  YMK_API_BEGIN
  RoctxRange roctx_range("ymk_rtdetr_forward");
  YMK_CHECK(m && x_dev && logits_dev && boxes_dev, "null argument");
  YMK_HIP(hipSetDevice(m->device));
  ymk::rtdetr_forward(m->impl, x_dev, b, h, w, logits_dev, boxes_dev, (hipStream_t)stream);
  YMK_API_END
}

int ymk_debug_option(const char* key, int value) {
  YMK_API_BEGIN
  YMK_CHECK(key != nullptr, "null key");
  const std::string k(key);
  YMK_CHECK(ymk::conv_debug_option(k, value) || ymk::parseq_debug_option(k, value) || ymk::decstep_debug_option(k, value) ||
                ymk::conv_split_debug_option(k, value),
            "unknown debug option: " + k);
  YMK_API_END
}

int ymk_stat(const char* key, int64_t* value) {
  YMK_API_BEGIN
  YMK_CHECK(key != nullptr && value != nullptr, "null argument");
  long long v = 0;
  YMK_CHECK(ymk::conv_split_stat(std::string(key), &v) || ymk::runtime_stat(std::string(key), &v), std::string("unknown counter: ") + key);
  *value = v;
  YMK_API_END
}

int ymk_amax_check_counters(int64_t* out4) {
  YMK_API_BEGIN
  YMK_CHECK(out4 != nullptr, "null argument");
  long long v[4];
  ymk::amax_check_counters(v);
  for (int i = 0; i < 4; ++i) out4[i] = v[i];
  YMK_API_END
}

int ymk_prof_begin(void) {
  YMK_API_BEGIN
  ymk::prof_begin();
  YMK_API_END
}

int ymk_prof_end(double* conv_ms, double* conv_flop, int64_t* conv_launches) {
  YMK_API_BEGIN
  YMK_CHECK(conv_ms && conv_flop && conv_launches, "null argument");
  ymk::prof_end(conv_ms, conv_flop, conv_launches);
  YMK_API_END
}

int ymk_prof_bytes(double* conv_bytes) {
  YMK_API_BEGIN
  YMK_CHECK(conv_bytes, "null argument");
  *conv_bytes = ymk::prof_bytes();
  YMK_API_END
}

int ymk_prof_launch_table(double* ms, double* flop, double* bytes, double* mfma_products, int64_t capacity, int64_t* count) {
  YMK_API_BEGIN
  YMK_CHECK(count != nullptr && capacity >= 0 && (capacity == 0 || (ms && flop && bytes && mfma_products)), "null argument");
  *count = ymk::prof_launch_table(ms, flop, bytes, mfma_products, capacity);
  YMK_API_END
}

// ------------------------------------------------------------------ single operators
int ymk_op_conv2d(const float* x_dev, int n, int h, int w, int c, const float* w_host_oihw, int cout, int cin, int kh,
                  int kw, const float* scale_host, const float* bias_host, const float* res_dev, int stride, int pad,
                  int dil, int act, int tap4, float* y_dev, void* stream) {
  YMK_API_BEGIN
  using namespace ymk;
  DevicePool pool;
  ConvW cw;
  cw.cout = cout;
  cw.cin = tap4 ? 4 : cin;
  cw.kh = kh;
  cw.kw = kw;
  cw.mode = tap4 ? 1 : 0;
  std::vector<float> panel;
  pack_conv_weight(w_host_oihw, cout, cin, kh, kw, tap4 != 0, panel, cw.kpad, cw.ctiles);
  cw.w = pool.upload(panel);
  if (scale_host) cw.scale = pool.upload(scale_host, cout);
  if (bias_host) cw.bias = pool.upload(bias_host, cout);
  Tensor in{const_cast<float*>(x_dev), n, h, w, c, c};
  const int oh = conv_out_dim(h, kh, stride, pad, dil), ow = conv_out_dim(w, kw, stride, pad, dil);
  Tensor out{y_dev, n, oh, ow, cout, cout};
  Tensor res{const_cast<float*>(res_dev), n, oh, ow, cout, cout};
  ConvArgs a;
  a.stride = stride;
  a.pad = pad;
  a.dil = dil;
  a.act = act;
  a.res = res_dev ? &res : nullptr;
  SplitCtxOwner split_ctx;  // split copies of this call's panel live and die with it
  ConvSplitScope scope(-1, split_ctx.get(), 0);  // single operators: exact fp32 unless the process-wide option says otherwise
  conv2d((hipStream_t)stream, in, cw, a, out);
  YMK_HIP(hipStreamSynchronize((hipStream_t)stream));  // pool frees the panel on return
  YMK_API_END
}

int ymk_op_conv1x1_astat(const float* x_dev, int m, int c, const float* w_host_oc, int cout, const float* scale_host,
                         const float* bias_host, const float* res_dev, int act, const float* ln_g_host, const float* ln_b_host, float ln_eps,
                         float* y_dev, int reps, float* kernel_ms, void* stream) {
  YMK_API_BEGIN
  using namespace ymk;
  YMK_CHECK(x_dev && w_host_oc && y_dev && m > 0 && c > 0 && cout > 0, "bad argument");
  YMK_CHECK((ln_g_host == nullptr) == (ln_b_host == nullptr), "LayerNorm: gamma and beta come together");
  hipStream_t s = (hipStream_t)stream;
  DevicePool pool;
  ConvW cw;
  cw.cout = cout;
  cw.cin = c;
  std::vector<float> panel;
  YMK_CHECK(c % 4 == 0, "astat: channels must be a multiple of 4");
  pack_conv_weight(w_host_oc, cout, c, 1, 1, false, panel, cw.kpad, cw.ctiles);
  YMK_CHECK(cw.kpad <= 256, "astat: K <= 256");
  cw.w = pool.upload(panel);
  if (scale_host) cw.scale = pool.upload(scale_host, cout);
  if (bias_host) cw.bias = pool.upload(bias_host, cout);
  // the input's max|x| record: measured - or, in front of a fused LayerNorm, the static bound of its output, as the models do
  unsigned* rec = nullptr;
  const float *g_dev = nullptr, *b_dev = nullptr;
  if (ln_g_host) {
    const std::vector<float> g(ln_g_host, ln_g_host + c), b(ln_b_host, ln_b_host + c);
    g_dev = pool.upload(g);
    b_dev = pool.upload(b);
    rec = make_layernorm_amax_record(pool, g, b);
  } else {
    rec = reinterpret_cast<unsigned*>(pool.alloc(AMAX_REC_WORDS));
    YMK_HIP(hipMemsetAsync(rec, 0, AMAX_REC_WORDS * sizeof(unsigned), s));
    absmax_record(s, x_dev, (size_t)m * c, rec);
  }
  struct Events {  // freed on every way out (a failing HIP call throws)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } ev;
  YMK_HIP(hipEventCreate(&ev.e0));
  YMK_HIP(hipEventCreate(&ev.e1));
  SplitCtxOwner split_ctx;  // the fp16 planes of this call's panel live and die with it
  ConvSplitScope scope(SPLIT_F16X2, split_ctx.get(), 0);
  ConvSplitTileScope tile_scope(30);  // the A-stationary kernel for whatever it can run - this thread's launches only
  for (int r = 0; r < std::max(1, reps); ++r) {
    if (r == std::max(1, reps) - 1) YMK_HIP(hipEventRecord(ev.e0, s));
    if (ln_g_host) {
      YMK_CHECK(gemm_ln_fused(s, x_dev, m, c, c, g_dev, b_dev, ln_eps, cw, act, res_dev, cout, y_dev, cout, rec),
                "astat: this launch cannot carry a fused LayerNorm (C must be 128 or 192)");
    } else {
      YMK_CHECK(gemm_takes_astat(m, c, cw, res_dev != nullptr, cout), "astat: not a launch the A-stationary kernel runs");
      gemm(s, x_dev, m, c, c, cw, act, res_dev, cout, y_dev, cout, nullptr, nullptr, EPI_STORE, rec);
    }
    if (r == std::max(1, reps) - 1) YMK_HIP(hipEventRecord(ev.e1, s));
  }
  YMK_HIP(hipStreamSynchronize(s));
  float ms = 0.f;
  YMK_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  if (kernel_ms) *kernel_ms = ms;
  YMK_API_END
}

int ymk_op_vit_mlp(const float* x_dev, int m, int d, int f, const float* ln_g_host, const float* ln_b_host, float ln_eps,
                   const float* w1_host_fd, const float* b1_host, const float* w2_host_df, const float* b2_host, float* y_dev, int reps,
                   float* kernel_ms, void* stream) {
  YMK_API_BEGIN
  using namespace ymk;
  YMK_CHECK(x_dev && y_dev && ln_g_host && ln_b_host && w1_host_fd && b1_host && w2_host_df && b2_host && m > 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  DevicePool pool;
  const std::vector<float> g(ln_g_host, ln_g_host + d), b(ln_b_host, ln_b_host + d);
  const float* g_dev = pool.upload(g);
  const float* b_dev = pool.upload(b);
  ConvW fc1 = make_linear_raw(pool, w1_host_fd, b1_host, f, d), fc2 = make_linear_raw(pool, w2_host_df, b2_host, d, f);
  struct Events {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } ev;
  YMK_HIP(hipEventCreate(&ev.e0));
  YMK_HIP(hipEventCreate(&ev.e1));
  SplitCtxOwner split_ctx;
  ConvSplitScope scope(SPLIT_F16X2, split_ctx.get(), 0);
  YMK_HIP(hipMemcpyAsync(y_dev, x_dev, (size_t)m * d * sizeof(float), hipMemcpyDeviceToDevice, s));
  for (int r = 0; r < std::max(1, reps); ++r) {  // in place on y: repetitions (timing) keep transforming it
    if (r == std::max(1, reps) - 1) YMK_HIP(hipEventRecord(ev.e0, s));
    YMK_CHECK(vit_mlp_fused(s, y_dev, m, d, g_dev, b_dev, ln_eps, layernorm_output_bound(g, b), fc1, fc2),
              "the fused ViT MLP runs D = 192, F = 768 and at least 256 blocks of 128 rows");
    if (r == std::max(1, reps) - 1) YMK_HIP(hipEventRecord(ev.e1, s));
  }
  YMK_HIP(hipStreamSynchronize(s));
  float ms = 0.f;
  YMK_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  if (kernel_ms) *kernel_ms = ms;
  YMK_API_END
}

int ymk_op_layernorm(const float* x_dev, int rows, int d, const float* g_dev, const float* b_dev, float eps,
                     float* y_dev, void* stream) {
  YMK_API_BEGIN
  ymk::layernorm((hipStream_t)stream, x_dev, d, 0, g_dev, b_dev, eps, y_dev, d, rows, d);
  YMK_API_END
}

int ymk_op_attention(const float* q_dev, const float* k_dev, const float* v_dev, float* o_dev, int b, int heads, int lq,
                     int lk, int hd, float scale, const unsigned char* mask_qk_dev, const unsigned char* kpm_dev,
                     int use_small, void* stream) {
  YMK_API_BEGIN
  const int D = heads * hd;
  if (use_small || mask_qk_dev || kpm_dev)
    ymk::small_attention((hipStream_t)stream, q_dev, k_dev, v_dev, o_dev, b, heads, lq, lk, hd, D, D, D, D, (long)lq * D,
                         (long)lk * D, (long)lk * D, (long)lq * D, scale, mask_qk_dev, lk, kpm_dev, lk);
  else if (ymk::conv_effective_split() == ymk::SPLIT_F16X2) {
    // ymk_debug_option("conv_split", 16): the fp16-split form, its three max|x| records measured here (tests)
    unsigned* rec = (unsigned*)ymk::dev_malloc(3 * ymk::AMAX_REC_WORDS * sizeof(unsigned));
    YMK_HIP(hipMemsetAsync(rec, 0, 3 * ymk::AMAX_REC_WORDS * sizeof(unsigned), (hipStream_t)stream));
    ymk::absmax_record((hipStream_t)stream, q_dev, (size_t)b * lq * D, rec);
    ymk::absmax_record((hipStream_t)stream, k_dev, (size_t)b * lk * D, rec + ymk::AMAX_REC_WORDS);
    ymk::absmax_record((hipStream_t)stream, v_dev, (size_t)b * lk * D, rec + 2 * ymk::AMAX_REC_WORDS);
    ymk::flash_attention((hipStream_t)stream, q_dev, k_dev, v_dev, o_dev, b, heads, lq, lk, hd, D, D, D, D, (long)lq * D,
                         (long)lk * D, (long)lk * D, (long)lq * D, scale, nullptr, rec, rec + ymk::AMAX_REC_WORDS, rec + 2 * ymk::AMAX_REC_WORDS);
    YMK_HIP(hipStreamSynchronize((hipStream_t)stream));
    ymk::dev_free(rec);
  } else
    ymk::flash_attention((hipStream_t)stream, q_dev, k_dev, v_dev, o_dev, b, heads, lq, lk, hd, D, D, D, D, (long)lq * D,
                         (long)lk * D, (long)lk * D, (long)lq * D, scale);
  YMK_API_END
}

int ymk_op_maxpool3x3s2(const float* x_dev, int n, int h, int w, int c, float* y_dev, void* stream) {
  YMK_API_BEGIN
  using namespace ymk;
  Tensor in{const_cast<float*>(x_dev), n, h, w, c, c};
  Tensor out{y_dev, n, (h + 2 - 3) / 2 + 1, (w + 2 - 3) / 2 + 1, c, c};
  maxpool3x3s2((hipStream_t)stream, in, out);
  YMK_API_END
}

int ymk_op_upsample_bilinear(const float* x_dev, int n, int h, int w, int c, int oh, int ow, const float* add_dev,
                             float* y_dev, void* stream) {
  YMK_API_BEGIN
  using namespace ymk;
  Tensor in{const_cast<float*>(x_dev), n, h, w, c, c};
  Tensor out{y_dev, n, oh, ow, c, c};
  Tensor add{const_cast<float*>(add_dev), n, oh, ow, c, c};
  upsample_bilinear((hipStream_t)stream, in, out, add_dev ? &add : nullptr);
  YMK_API_END
}

}  // extern "C"
