// Host runtime of libymk_hip.so: error slot, arena, weight store, state-dict -> packed panels.
#include "ymk_common.h"
#include <atomic>
#include <cmath>
#include <cstdlib>

namespace ymk {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const std::string& last_error() { return g_err; }

// ---------------------------------------------------------------- allocation accounting (ymk_common.h)
static thread_local int t_forward_depth = 0;
static std::atomic<long long> g_allocs_in_forward{0}, g_arena_grows_in_forward{0}, g_lazy_panel_builds{0}, g_syncs_in_forward{0};
ForwardScope::ForwardScope() { ++t_forward_depth; }
ForwardScope::~ForwardScope() { --t_forward_depth; }
bool in_forward() { return t_forward_depth > 0; }
static void count_alloc() {
  if (t_forward_depth > 0) ++g_allocs_in_forward;
}
void* dev_malloc(size_t bytes) {
  void* p = nullptr;
  count_alloc();
  YMK_HIP(hipMalloc(&p, bytes ? bytes : 4));
  return p;
}
void dev_free(void* p) {
  if (!p) return;
  count_alloc();
  (void)hipFree(p);
}
void* host_malloc_pinned(size_t bytes, unsigned flags) {
  void* p = nullptr;
  count_alloc();
  YMK_HIP(hipHostMalloc(&p, bytes ? bytes : 4, flags));
  return p;
}
void host_free_pinned(void* p) {
  if (!p) return;
  count_alloc();
  (void)hipHostFree(p);
}
void forward_sync(hipStream_t s) {
  if (t_forward_depth > 0) ++g_syncs_in_forward;
  YMK_HIP(hipStreamSynchronize(s));
}
void note_lazy_panel_build() {
  if (t_forward_depth > 0) ++g_lazy_panel_builds;
}
void note_arena_grow() {
  if (t_forward_depth > 0) ++g_arena_grows_in_forward;
}
bool runtime_stat(const std::string& key, long long* value) {
  if (key == "allocs_in_forward") *value = g_allocs_in_forward.load();
  else if (key == "arena_grows_in_forward") *value = g_arena_grows_in_forward.load();
  else if (key == "lazy_panel_builds") *value = g_lazy_panel_builds.load();
  else if (key == "syncs_in_forward") *value = g_syncs_in_forward.load();
  else return false;
  return true;
}
static bool env_flag(const char* name) {
  const char* v = std::getenv(name);
  return v != nullptr && v[0] != '\0' && v[0] != '0';
}
bool debug_lazy_split() {
  static const bool on = env_flag("YMK_DEBUG_LAZY_SPLIT");
  return on;
}
bool debug_hazard_null_memset() {
  static const bool on = env_flag("YMK_DEBUG_HAZARD_NULL_MEMSET");
  return on;
}
bool debug_hazard_no_finalize_sync() {
  static const bool on = env_flag("YMK_DEBUG_HAZARD_NO_FINALIZE_SYNC");
  return on;
}

// ---------------------------------------------------------------- Arena
Arena::~Arena() { dev_free(base_); }
void Arena::reserve(size_t bytes) {
  if (bytes <= cap_) return;
  YMK_CHECK(off_ == 0, "arena reserve while in use");
  note_arena_grow();
  dev_free(base_);
  base_ = nullptr;
  cap_ = 0;
  base_ = (char*)dev_malloc(bytes);
  cap_ = bytes;
}
void* Arena::alloc_bytes(size_t bytes) {
  const size_t a = (bytes + 255) & ~(size_t)255;
  const size_t at = off_;
  off_ += a;
  if (off_ > high_) high_ = off_;
  if (dry_run) return (void*)(uintptr_t)(4096 + at);  // aligned fake address, never dereferenced
  YMK_CHECK(off_ <= cap_, "arena overflow: need " + std::to_string(off_) + " have " + std::to_string(cap_));
  return base_ + at;
}
void Arena::amax_begin(hipStream_t s, int records) {
  amax_n_ = records;
  amax_used_ = 0;
  const size_t bytes = (size_t)records * AMAX_REC_WORDS * sizeof(unsigned);
  amax_pool_ = (unsigned*)alloc_bytes(bytes);
  if (!dry_run) YMK_HIP(hipMemsetAsync(amax_pool_, 0, bytes, s));
}
unsigned* Arena::amax_next() {
  if (amax_pool_ == nullptr || amax_used_ >= amax_n_) return nullptr;
  return amax_pool_ + (size_t)(amax_used_++) * AMAX_REC_WORDS;
}
float* Arena::alloc_f(size_t count) { return (float*)alloc_bytes(count * sizeof(float)); }
Tensor Arena::tensor(int n, int h, int w, int c) {
  Tensor t;
  t.n = n;
  t.h = h;
  t.w = w;
  t.c = c;
  t.ld = c;
  t.p = alloc_f((size_t)n * h * w * c);
  return t;
}

// ---------------------------------------------------------------- DevicePool
DevicePool::~DevicePool() {
  for (void* p : ptrs_) dev_free(p);
}
void DevicePool::note(const ConvW& c, bool perm) {
  std::vector<ConvW>& list = perm ? perm_ : convs_;
  for (ConvW& have : list)
    if (have.w == c.w) {
      have = c;
      return;
    }
  list.push_back(c);
}
float* DevicePool::alloc(size_t n) {
  const size_t b = (n ? n : 1) * sizeof(float);
  void* p = dev_malloc(b);
  ptrs_.push_back(p);
  bytes_ += b;
  return (float*)p;
}
float* DevicePool::upload(const float* src, size_t n) {
  float* d = alloc(n);
  if (n) YMK_HIP(hipMemcpy(d, src, n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
float* DevicePool::upload(const std::vector<float>& v) { return upload(v.data(), v.size()); }

// ---------------------------------------------------------------- WeightStore
void WeightStore::put(const std::string& name, const float* data, int ndim, const int64_t* dims) {
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    t.dims.push_back(dims[i]);
    n *= (size_t)dims[i];
  }
  t.data.assign(data, data + n);
  t_[name] = std::move(t);
}
const HostTensor& WeightStore::get(const std::string& name) const {
  auto it = t_.find(name);
  if (it == t_.end()) throw Error("missing weight tensor: " + name);
  return it->second;
}

// ---------------------------------------------------------------- packing helpers
// ConvW::pl_a / pl_b from the OIHW weights and the folded scale / bias (1 / 0 where absent); a little head room for the
// rounding of the fp32 sums and of the bound's own arithmetic
static void plane_bound(ConvW& c, const float* oihw, size_t per_out, const std::vector<float>& scale, const std::vector<float>& bias) {
  double a = 0.0, b = 0.0;
  for (int o = 0; o < c.cout; ++o) {
    double l1 = 0.0;
    for (size_t i = 0; i < per_out; ++i) l1 += std::fabs((double)oihw[(size_t)o * per_out + i]);
    a = std::max(a, l1 * (scale.empty() ? 1.0 : std::fabs((double)scale[o])));
    if (!bias.empty()) b = std::max(b, std::fabs((double)bias[o]));
  }
  c.pl_a = (float)(a * 1.001);
  c.pl_b = (float)(b * 1.001);
}

ConvW make_conv(DevicePool& pool, const WeightStore& ws, const std::string& conv_prefix, const std::string& bn_prefix,
                bool tap4, float bn_eps) {
  const HostTensor& w = ws.get(conv_prefix + ".weight");
  YMK_CHECK(w.dims.size() == 4, conv_prefix + ".weight must be OIHW");
  ConvW c;
  c.cout = (int)w.dims[0];
  c.cin = (int)w.dims[1];
  c.kh = (int)w.dims[2];
  c.kw = (int)w.dims[3];
  c.mode = tap4 ? 1 : 0;
  std::vector<float> panel;
  pack_conv_weight(w.data.data(), c.cout, c.cin, c.kh, c.kw, tap4, panel, c.kpad, c.ctiles);
  if (tap4) c.cin = 4;
  c.w = pool.upload(panel);
  std::vector<float> scale, bias;
  const bool has_cb = ws.has(conv_prefix + ".bias");
  if (!bn_prefix.empty()) {
    // eval BatchNorm: y = (x - mean) / sqrt(var + eps) * gamma + beta    (x may carry a conv bias)
    const HostTensor& g = ws.get(bn_prefix + ".weight");
    const HostTensor& b = ws.get(bn_prefix + ".bias");
    const HostTensor& m = ws.get(bn_prefix + ".running_mean");
    const HostTensor& v = ws.get(bn_prefix + ".running_var");
    YMK_CHECK((int)g.numel() == c.cout, bn_prefix + ": channel mismatch");
    scale.resize(c.cout);
    bias.resize(c.cout);
    for (int i = 0; i < c.cout; ++i) {
      const float s = g.data[i] / std::sqrt(v.data[i] + bn_eps);
      const float cb = has_cb ? ws.get(conv_prefix + ".bias").data[i] : 0.f;
      scale[i] = s;
      bias[i] = b.data[i] + (cb - m.data[i]) * s;
    }
    c.scale = pool.upload(scale);
    c.bias = pool.upload(bias);
  } else if (has_cb) {
    bias = ws.get(conv_prefix + ".bias").data;
    c.bias = pool.upload(bias);
  }
  plane_bound(c, w.data.data(), w.numel() / (size_t)w.dims[0], scale, bias);
  pool.note(c);
  return c;
}

ConvW make_linear_raw(DevicePool& pool, const float* w_out_in, const float* bias, int out, int in) {
  ConvW c;
  c.cout = out;
  c.cin = in;
  c.kh = c.kw = 1;
  c.mode = 0;
  std::vector<float> panel;
  pack_conv_weight(w_out_in, out, in, 1, 1, false, panel, c.kpad, c.ctiles);
  c.w = pool.upload(panel);
  if (bias) c.bias = pool.upload(bias, out);
  plane_bound(c, w_out_in, (size_t)in, {}, bias ? std::vector<float>(bias, bias + out) : std::vector<float>());
  pool.note(c);
  return c;
}

float layernorm_output_bound(const std::vector<float>& gamma, const std::vector<float>& beta) {
  float g = 0.f, b = 0.f;
  for (float v : gamma) g = std::max(g, std::fabs(v));
  for (float v : beta) b = std::max(b, std::fabs(v));
  return std::sqrt((float)gamma.size()) * g + b;
}

unsigned* make_layernorm_amax_record(DevicePool& pool, const std::vector<float>& gamma, const std::vector<float>& beta) {
  const float bound = layernorm_output_bound(gamma, beta);
  std::vector<float> rec(AMAX_REC_WORDS, 0.f);  // all-zero bit patterns but word 0
  rec[0] = bound;                                // the record holds fp32 bit patterns: upload the float as it is
  return reinterpret_cast<unsigned*>(pool.upload(rec));
}

ConvW make_linear(DevicePool& pool, const WeightStore& ws, const std::string& prefix, bool has_bias) {
  const HostTensor& w = ws.get(prefix + ".weight");
  YMK_CHECK(w.dims.size() == 2, prefix + ".weight must be [out][in]");
  const float* b = nullptr;
  if (has_bias && ws.has(prefix + ".bias")) b = ws.get(prefix + ".bias").data.data();
  return make_linear_raw(pool, w.data.data(), b, (int)w.dims[0], (int)w.dims[1]);
}

}  // namespace ymk
