// PARSeq text recogniser forward on gfx950.
// Follows models/parseq.py:159-311 (encode, greedy AR decode with early stop and repetition stop,
// refinement with the reference's integer-indexed mask, repetition cut) and
// models/layers/parseq_transformer.py:69-169,206-234 (two-stream decoder layer, ViT encoder with the
// dynamic-width position-embedding crop).  Differences in *how* (not what) it is computed:
//   - K/V of the context and of the encoder memory are projected once and cached (the reference
//     re-projects them every step); rows are bit-identical because each output row of the MFMA GEMM
//     is an independent fmaf chain;
//   - tokens, <eos> bookkeeping and the repetition detector live on the device; the host polls one
//     mapped word per step, two steps behind the device (the "every row has an <eos>" test,
//     models/parseq.py:245-250); the steps queued past the stop are no-ops;
//   - decoder widths <= 256 run a whole AR step as one kernel (ymk_decstep.hip), wider ones as GEMMs.
#include <sched.h>

#include <atomic>

#include <optional>

#include "ymk_common.h"
#include "ymk_seq.h"
#include "ymk_decstep.h"

namespace ymk {

void tile_rows(hipStream_t s, const float* src, int rows, int D, float* dst, int B);

static std::atomic<int> g_parseq_unfused{0};  // ymk_debug_option("parseq_unfused", 1): per-op decoder path for every width (tests)
static bool parseq_unfused() { return g_parseq_unfused.load(std::memory_order_relaxed) != 0; }
static std::atomic<int> g_parseq_no_rowmax{0};  // ymk_debug_option("parseq_no_rowmax", 1): keep the AR logits (A/B, tests)
static bool parseq_no_rowmax() { return g_parseq_no_rowmax.load(std::memory_order_relaxed) != 0; }
static std::atomic<int> g_parseq_no_mlp_fusion{0};  // ymk_debug_option("parseq_no_mlp_fusion", 1): fc1 / fc2 as GEMMs of their own (A/B, tests)
static bool parseq_no_mlp_fusion() { return g_parseq_no_mlp_fusion.load(std::memory_order_relaxed) != 0; }
static std::atomic<int> g_parseq_no_ln_fusion{0};  // ymk_debug_option("parseq_no_ln_fusion", 1): LayerNorm as its own launch (A/B, tests)
static bool parseq_no_ln_fusion() { return g_parseq_no_ln_fusion.load(std::memory_order_relaxed) != 0; }
// ymk_debug_option("ar_publish", v): how a greedy step tells the host whether rows are still open.  1 (default since round 6) =
// the rows store a flag and a one-thread launch behind the greedy kernel writes the mapped host word; 0 = the rows count and the
// kernel's last-arriving block writes it (rounds 1-5; A/B runs and tools/stress_call.py)
static std::atomic<int> g_ar_publish{1};
bool parseq_debug_option(const std::string& key, int value) {
  if (key == "parseq_unfused") g_parseq_unfused = value;
  else if (key == "parseq_no_rowmax") g_parseq_no_rowmax = value;
  else if (key == "parseq_no_ln_fusion") g_parseq_no_ln_fusion = value;
  else if (key == "parseq_no_mlp_fusion") g_parseq_no_mlp_fusion = value;
  else if (key == "ar_publish") g_ar_publish = value;
  else return false;
  return true;
}

struct PGroup {
  const float* x;
  int B, W;
};

namespace {

struct EncBlock {
  float *ln1g, *ln1b, *ln2g, *ln2b;
  unsigned *ln1_rec, *ln2_rec;  // static max|x| records of the two LayerNorm outputs (make_layernorm_amax_record)
  float ln2_bound = 0.f;        // the number ln2_rec holds (layernorm_output_bound): scale of the fused MLP's row planes
  ConvW qkv, proj, fc1, fc2;
};

class ParseqModel : public Model {
 public:
  const char* kind() const override { return "parseq"; }

  void finalize() override {
    ph_ = (int)param("patch_h", 4);
    pw_ = (int)param("patch_w", 8);
    img_h_ = (int)param("img_h", 32);
    img_w_ = (int)param("img_w", 800);
    D_ = (int)param("enc_dim", 192);
    eh_ = (int)param("enc_heads", 6);
    depth_ = (int)param("enc_depth", 12);
    Dd_ = (int)param("dec_dim", 192);
    dh_ = (int)param("dec_heads", 6);
    ntok_ = (int)param("num_tokens", 7121);
    maxlen_ = (int)param("max_label_length", 100);
    refine_ = (int)param("refine_iters", 1);
    rep_on_ = (int)param("repetition_stop", 1);
    rep_pmax_ = (int)param("rep_period_max", 8);
    rep_p1_ = (int)param("rep_min_run_p1", 8);
    rep_min_ = (int)param("rep_min_repeats", 3);
    YMK_CHECK((int)param("dec_depth", 1) == 1, "only decoder depth 1 is implemented (all shipped configs)");
    YMK_CHECK((int)param("decode_ar", 1) == 1, "only decode_ar=1 is implemented (all shipped configs)");
    YMK_CHECK(D_ == Dd_, "encoder and decoder widths must match (cross attention has no kdim)");
    YMK_CHECK(D_ % eh_ == 0 && Dd_ % dh_ == 0, "embed dim must divide by heads");
    C_ = ntok_ - 2;
    eos_ = 0;
    bos_ = ntok_ - 2;
    pad_ = ntok_ - 1;
    nsteps_ = maxlen_ + 1;
    gh_ = img_h_ / ph_;
    full_gw_ = img_w_ / pw_;

    const std::string e = "encoder.";
    patch_ = make_conv(pool, ws, e + "patch_embed.proj", "", /*tap4=*/true);
    {
      const HostTensor& pe = ws.get(e + "pos_embed");
      YMK_CHECK((int)pe.numel() == gh_ * full_gw_ * D_, "pos_embed size");
      pos_embed_ = pool.upload(pe.data);
    }
    blocks_.resize(depth_);
    for (int i = 0; i < depth_; ++i) {
      const std::string p = e + "blocks." + std::to_string(i) + ".";
      EncBlock& b = blocks_[i];
      b.ln1g = pool.upload(ws.get(p + "norm1.weight").data);
      b.ln1b = pool.upload(ws.get(p + "norm1.bias").data);
      b.ln2g = pool.upload(ws.get(p + "norm2.weight").data);
      b.ln2b = pool.upload(ws.get(p + "norm2.bias").data);
      b.ln1_rec = make_layernorm_amax_record(pool, ws.get(p + "norm1.weight").data, ws.get(p + "norm1.bias").data);
      b.ln2_rec = make_layernorm_amax_record(pool, ws.get(p + "norm2.weight").data, ws.get(p + "norm2.bias").data);
      b.ln2_bound = layernorm_output_bound(ws.get(p + "norm2.weight").data, ws.get(p + "norm2.bias").data);
      b.qkv = make_linear(pool, ws, p + "attn.qkv");
      b.proj = make_linear(pool, ws, p + "attn.proj");
      b.fc1 = make_linear(pool, ws, p + "mlp.fc1");
      b.fc2 = make_linear(pool, ws, p + "mlp.fc2");
      if (b.fc1.cin == 192 && b.fc1.cout == 768) pool.note(b.fc2, /*perm=*/true);  // the fused MLP kernel's fc2 copy (ymk_vit_mlp.hip)
    }
    enc_ng_ = pool.upload(ws.get(e + "norm.weight").data);
    enc_nb_ = pool.upload(ws.get(e + "norm.bias").data);
    enc_n_rec_ = make_layernorm_amax_record(pool, ws.get(e + "norm.weight").data, ws.get(e + "norm.bias").data);

    const std::string d = "decoder.layers.0.";
    auto split_mha = [&](const std::string& name, ConvW& wq, ConvW& wkv, ConvW& wo) {
      const HostTensor& w = ws.get(d + name + ".in_proj_weight");
      const HostTensor& b = ws.get(d + name + ".in_proj_bias");
      YMK_CHECK((int)w.dims[0] == 3 * Dd_ && (int)w.dims[1] == Dd_, name + ".in_proj_weight shape");
      wq = make_linear_raw(pool, w.data.data(), b.data.data(), Dd_, Dd_);
      wkv = make_linear_raw(pool, w.data.data() + (size_t)Dd_ * Dd_, b.data.data() + Dd_, 2 * Dd_, Dd_);
      wo = make_linear(pool, ws, d + name + ".out_proj");
    };
    split_mha("self_attn", sa_q_, sa_kv_, sa_o_);
    split_mha("cross_attn", ca_q_, ca_kv_, ca_o_);
    {
      // transposed copies ([in][out]) for the fused one-kernel decoder step (ymk_decstep.hip)
      auto tr = [&](const float* w, int out, int in) {
        std::vector<float> t((size_t)out * in);
        for (int o = 0; o < out; ++o)
          for (int k = 0; k < in; ++k) t[(size_t)k * out + o] = w[(size_t)o * in + k];
        return pool.upload(t);
      };
      const HostTensor& sw = ws.get(d + "self_attn.in_proj_weight");
      const HostTensor& sb = ws.get(d + "self_attn.in_proj_bias");
      const HostTensor& cw = ws.get(d + "cross_attn.in_proj_weight");
      const HostTensor& cb = ws.get(d + "cross_attn.in_proj_bias");
      const int D = Dd_, F = (int)ws.get(d + "linear1.weight").dims[0];
      fw_.D = D;
      fw_.H = dh_;
      fw_.F = F;
      fw_.Wkv_t = tr(sw.data.data() + (size_t)D * D, 2 * D, D);
      fw_.bkv = pool.upload(sb.data.data() + D, 2 * D);
      fw_.Wo1_t = tr(ws.get(d + "self_attn.out_proj.weight").data.data(), D, D);
      fw_.bo1 = pool.upload(ws.get(d + "self_attn.out_proj.bias").data);
      fw_.Wq_t = tr(cw.data.data(), D, D);
      fw_.bq = pool.upload(cb.data.data(), D);
      fw_.Wo2_t = tr(ws.get(d + "cross_attn.out_proj.weight").data.data(), D, D);
      fw_.bo2 = pool.upload(ws.get(d + "cross_attn.out_proj.bias").data);
      fw_.W1_t = tr(ws.get(d + "linear1.weight").data.data(), F, D);
      fw_.b1 = pool.upload(ws.get(d + "linear1.bias").data);
      fw_.W2_t = tr(ws.get(d + "linear2.weight").data.data(), D, F);
      fw_.b2 = pool.upload(ws.get(d + "linear2.bias").data);
    }
    lin1_ = make_linear(pool, ws, d + "linear1");
    lin2_ = make_linear(pool, ws, d + "linear2");
    auto up = [&](const std::string& n) { return pool.upload(ws.get(n).data); };
    n1g_ = up(d + "norm1.weight"); n1b_ = up(d + "norm1.bias");
    n2g_ = up(d + "norm2.weight"); n2b_ = up(d + "norm2.bias");
    nqg_ = up(d + "norm_q.weight"); nqb_ = up(d + "norm_q.bias");
    ncg_ = up(d + "norm_c.weight"); ncb_ = up(d + "norm_c.bias");
    dng_ = up("decoder.norm.weight"); dnb_ = up("decoder.norm.bias");
    n1_rec_ = make_layernorm_amax_record(pool, ws.get(d + "norm1.weight").data, ws.get(d + "norm1.bias").data);
    n2_rec_ = make_layernorm_amax_record(pool, ws.get(d + "norm2.weight").data, ws.get(d + "norm2.bias").data);
    dn_rec_ = make_layernorm_amax_record(pool, ws.get("decoder.norm.weight").data, ws.get("decoder.norm.bias").data);
    head_ = make_linear(pool, ws, "head");
    YMK_CHECK(head_.cout == C_, "head width must be num_tokens - 2");
    {
      const HostTensor& em = ws.get("text_embed.embedding.weight");
      YMK_CHECK((int)em.dims[0] == ntok_ && (int)em.dims[1] == Dd_, "text_embed shape");
      emb_ = pool.upload(em.data);
      const HostTensor& pq = ws.get("pos_queries");
      YMK_CHECK((int)pq.numel() == nsteps_ * Dd_, "pos_queries shape");
      posq_ = pool.upload(pq.data);
      fw_.emb = emb_;
      fw_.posq = posq_;
      fw_.ncg = ncg_; fw_.ncb = ncb_; fw_.n1g = n1g_; fw_.n1b = n1b_; fw_.n2g = n2g_; fw_.n2b = n2b_;
      fw_.dng = dng_; fw_.dnb = dnb_;
    }
    {
      // refinement query mask (models/parseq.py:267-277): triu(1) with rows 0 and 1 cleared
      std::vector<float> tmp;
      std::vector<unsigned char> m((size_t)nsteps_ * nsteps_, 0);
      for (int q = 2; q < nsteps_; ++q)
        for (int k = q + 1; k < nsteps_; ++k) m[(size_t)q * nsteps_ + k] = 1;
      void* dm = dev_malloc(m.size());
      YMK_HIP(hipMemcpy(dm, m.data(), m.size(), hipMemcpyHostToDevice));
      qmask_ = (unsigned char*)dm;
      host_flags_ = (int*)host_malloc_pinned((size_t)nsteps_ * sizeof(int), hipHostMallocMapped);
      YMK_HIP(hipHostGetDevicePointer((void**)&host_flags_dev_, host_flags_, 0));
    }
    ws.clear();
    finalized = true;
  }

  ~ParseqModel() override {
    dev_free(qmask_);
    host_free_pinned(host_flags_);
    host_free_pinned(stage_);
  }

  int num_classes() const { return C_; }
  int num_steps() const { return nsteps_; }

  // A forward over `ng` mini-batches at once ("groups": each keeps its own padded width, so the padding columns every
  // crop sees are those of its own mini-batch).  Rows of a GEMM are independent, attention is per
  // sample, so the groups share every launch: token rows are laid end to end, attention kernels read per-sample
  // (offset, length) tables, and ONE greedy loop runs until every row of every group holds an <eos>.
  // groups[g].x: device fp32 [B_g][3][img_h][W_g]; logits: device [sum B_g][nsteps][C];
  // out_len[g] / ar_steps[g]: rows valid per sample / greedy steps of group g as its own loop would have run them.
  void forward_groups(const PGroup* groups, int ng, float* logits, int* out_len, int* ar_steps, hipStream_t s) {
    YMK_CHECK(finalized, "model not finalized");
    ForwardScope forward_scope;
    ConvSplitScope split_scope(conv_split(), split_ctx.get(), SPLIT_MODEL_DEFAULT);
    YMK_CHECK(ng > 0, "parseq: no mini-batch");
    uint64_t key = 1469598103934665603ull;
    for (int g = 0; g < ng; ++g) {
      YMK_CHECK(groups[g].x != nullptr && groups[g].B > 0 && groups[g].W >= pw_ && groups[g].W % pw_ == 0 && groups[g].W <= img_w_,
                "parseq input width must be a multiple of the patch width, <= img_w");
      key = (key ^ (((uint64_t)groups[g].B << 32) | (uint64_t)groups[g].W)) * 1099511628211ull;
    }
    if (key != shape_key_) {
      arena.dry_run = true;
      arena.reset();
      run(groups, ng, logits, out_len, ar_steps, s);
      arena.dry_run = false;
      const size_t need = arena.used();
      arena.reset();
      if (need > arena.capacity()) {
        forward_sync(s);
        arena.reserve(need + need / 4);  // ragged workloads change shape every call: leave head room, grow rarely
      }
      shape_key_ = key;
    }
    arena.reset();
    run(groups, ng, logits, out_len, ar_steps, s);
  }

  // Size the workspace once for the largest forward the caller will ever issue (max_lines rows, each max_w wide; every
  // buffer of run() is monotone in the row count, the token-row count and the largest group's pixels), so that
  // ragged forwards - whose shape changes on every call - never reach hipMalloc / hipFree on the serving path.
  void reserve(int max_lines, int /*h*/, int max_w, hipStream_t s) override {
    YMK_CHECK(finalized, "model not finalized");
    YMK_CHECK(max_lines > 0 && max_w >= pw_ && max_w <= img_w_, "parseq reserve: bad bounds");
    max_w -= max_w % pw_;
    // one group holding every line: the largest input staging buffer; the per-group tables of up to max_lines groups on top
    const PGroup g{reinterpret_cast<const float*>(uintptr_t(4096)), max_lines, max_w};
    arena.dry_run = true;
    arena.reset();
    run(&g, 1, nullptr, nullptr, nullptr, s);
    arena.dry_run = false;
    const size_t need = arena.used() + (size_t)nsteps_ * max_lines * sizeof(int) + 512;
    arena.reset();
    if (need > arena.capacity()) {
      YMK_HIP(hipStreamSynchronize(s));
      arena.reserve(need);
    }
    const size_t want = (size_t)3 * max_lines + (size_t)nsteps_ * max_lines + max_lines;
    if (want > stage_cap_) {
      YMK_HIP(hipStreamSynchronize(s));
      host_free_pinned(stage_);
      stage_ = nullptr;
      stage_cap_ = 0;
      stage_ = (int*)host_malloc_pinned(want * sizeof(int), hipHostMallocDefault);
      stage_cap_ = want;
    }
    shape_key_ = 0;
  }

 private:
  void ln(hipStream_t s, const float* x, const float* g, const float* b, float eps, float* y, int M, int D) {
    layernorm(s, x, D, 0, g, b, eps, y, D, M, D);
  }

  // out = act(LayerNorm(x) . W^T + bias): one launch with the LayerNorm folded into the operand load where the A-stationary
  // fp16-split kernel runs the layer (gemm_ln_fused: chip-filling launches at widths 128 / 192), else LayerNorm into
  // `tmp` and the GEMM on it - the same values either way up to the order of the K sums
  void ln_gemm(hipStream_t s, const float* x, const float* g, const float* b, float eps, float* tmp, int M, int D, const ConvW& w, int act,
               float* out, int out_ld, const unsigned* ln_rec, unsigned* out_rec) {
    if (!parseq_no_ln_fusion() && gemm_ln_fused(s, x, M, D, D, g, b, eps, w, act, nullptr, 0, out, out_ld, ln_rec, out_rec)) return;
    ln(s, x, g, b, eps, tmp, M, D);
    gemm(s, tmp, M, D, D, w, act, nullptr, 0, out, out_ld, nullptr, nullptr, EPI_STORE, ln_rec, out_rec);
  }

  // query-stream tail shared by the AR step and the refinement pass:
  //   q (in/out, [M][Dd]) already holds query + self-attention; adds cross attention and the FFN,
  //   then decoder.norm + head -> out rows (ld_out floats apart)
  void stream_tail(hipStream_t s, float* q, int M, int B, int Lq, const float* memkv, int L, const SeqTab* mem, float* t,
                   float* t2, float* h, float* out, int ld_out) {
    const int D = Dd_, hd = D / dh_;
    const float scale = 1.f / std::sqrt((float)hd);
    // max|x| records of the GEMM inputs (fp16-split launches: the refinement pass's 60 000 rows): LayerNorm outputs have
    // static ones, the cross-attention output is a convex combination of memory V rows, the FFN hidden state gets one
    ln(s, q, n1g_, n1b_, 1e-5f, t, M, D);
    unsigned* q_rec = Lq >= 32 ? arena.amax_next() : nullptr;  // the refinement pass: bounds the queries for the fp16-split attention
    gemm(s, t, M, D, D, ca_q_, ACT_NONE, nullptr, 0, t2, D, nullptr, nullptr, EPI_STORE, n1_rec_, q_rec);
    if (Lq >= 32)
      flash_attention(s, t2, memkv, memkv + D, t, B, dh_, Lq, L, hd, D, 2 * D, 2 * D, D, (long)Lq * D, (long)L * 2 * D,
                      (long)L * 2 * D, (long)Lq * D, scale, mem, q_rec, memkv_rec_, memkv_rec_);
    else
      small_attention(s, t2, memkv, memkv + D, t, B, dh_, Lq, L, hd, D, 2 * D, 2 * D, D, (long)Lq * D, (long)L * 2 * D,
                      (long)L * 2 * D, (long)Lq * D, scale, nullptr, 0, nullptr, 0, mem);
    gemm(s, t, M, D, D, ca_o_, ACT_NONE, q, D, q, D, nullptr, nullptr, EPI_STORE, memkv_rec_);
    ln(s, q, n2g_, n2b_, 1e-5f, t, M, D);
    unsigned* h_rec = arena.amax_next();
    gemm(s, t, M, D, D, lin1_, ACT_GELU, nullptr, 0, h, lin1_.cout, nullptr, nullptr, EPI_STORE, n2_rec_, h_rec);
    gemm(s, h, M, lin1_.cout, lin1_.cout, lin2_, ACT_NONE, q, D, q, D, nullptr, nullptr, EPI_STORE, h_rec);
    ln(s, q, dng_, dnb_, 1e-5f, t, M, D);
    gemm(s, t, M, D, D, head_, ACT_NONE, nullptr, 0, out, ld_out, nullptr, nullptr, EPI_STORE, dn_rec_);
  }

  void run(const PGroup* groups, int ng, float* logits, int* out_len, int* ar_steps, hipStream_t s) {
    const bool dry = arena.dry_run;
    const int D = D_, hd = D / eh_;
    const int NS = nsteps_, C = C_;
    // ---------------- geometry: token rows of the groups end to end, one (offset, length) pair per sample
    int B = 0, M = 0, Lmax = 0;
    size_t x4_max = 0;
    for (int g = 0; g < ng; ++g) {
      const int L = gh_ * (groups[g].W / pw_);
      B += groups[g].B;
      M += groups[g].B * L;
      Lmax = std::max(Lmax, L);
      x4_max = std::max(x4_max, (size_t)groups[g].B * img_h_ * groups[g].W * 4);
    }
    const bool ragged = ng > 1;
    // ---------------- encoder
    arena.amax_begin(s, 160);  // max|x| records (ymk_common.h): two per encoder block, one per decoder-tail call, K|V
    float* x4buf = arena.alloc_f(x4_max);
    float* xs = arena.alloc_f((size_t)M * D);  // [M][D] token stream, updated in place
    float* y = arena.alloc_f((size_t)M * D);
    float* qkv = arena.alloc_f((size_t)M * 3 * D);
    float* att = arena.alloc_f((size_t)M * D);
    float* hbuf = arena.alloc_f((size_t)M * blocks_[0].fc1.cout);
    float* mem = arena.alloc_f((size_t)M * D);
    float* memkv = arena.alloc_f((size_t)M * 2 * D);
    int* tab = (int*)arena.alloc_bytes((size_t)3 * B * sizeof(int));  // [B] first token row | [B] token rows | [B] group
    int* gopen = (int*)arena.alloc_bytes((size_t)nsteps_ * ng * sizeof(int));  // [step][group]: rows still open
    // ---------------- decoder buffers
    const int MR = B * NS;
    float* qsa = arena.alloc_f((size_t)NS * D);        // W_q(norm_q(pos_queries)) - shared by the batch
    float* posq_t = arena.alloc_f((size_t)MR * D);     // pos_queries tiled over the batch (refinement residual)
    float* cn = arena.alloc_f((size_t)MR * D);         // norm_c(content)
    float* skv = arena.alloc_f((size_t)MR * 2 * D);    // self-attention K|V cache
    float* qcur = arena.alloc_f((size_t)MR * D);
    float* t1 = arena.alloc_f((size_t)MR * D);
    float* t2 = arena.alloc_f((size_t)MR * D);
    float* hdec = arena.alloc_f((size_t)MR * lin1_.cout);
    // greedy steps through the fused decoder kernel, with a refinement pass to follow: the AR logits are only ever arg-maxed
    // (models/parseq.py:224), so the vocabulary head reduces each 64-column tile to (max, column) in its epilogue and no
    // [B][steps][C] logit buffer exists at all (1.9 GB at 655 rows); otherwise the steps' logits are kept - they are the output
    const bool fused = parseq_dec_step_supported(D, dh_, lin1_.cout, Lmax, NS) && !parseq_unfused();
    const bool ar_rowmax = fused && refine_ > 0 && !parseq_no_rowmax();
    const int head_tiles = (C + ROWMAX_TILE_N - 1) / ROWMAX_TILE_N;
    float* arlog = ar_rowmax ? nullptr : arena.alloc_f((size_t)MR * C);
    float* armax = ar_rowmax ? arena.alloc_f((size_t)B * head_tiles * 2) : nullptr;
    int* tok = (int*)arena.alloc_bytes((size_t)MR * sizeof(int));
    int* raw = (int*)arena.alloc_bytes((size_t)MR * sizeof(int));
    int* tok2 = (int*)arena.alloc_bytes((size_t)MR * sizeof(int));
    int* state = (int*)arena.alloc_bytes((size_t)B * 4 * sizeof(int));
    int* not_done = (int*)arena.alloc_bytes((size_t)2 * NS * sizeof(int));
    unsigned char* kpm = (unsigned char*)arena.alloc_bytes((size_t)MR);
    if (dry) return;

    SeqTab enc_tab, mem_tab;
    if (ragged) {
      // the tables travel through a pinned staging buffer owned by the model: a forward returns only after its
      // greedy loop has been observed to finish, so the previous call's copy has long left the buffer
      const size_t want = (size_t)3 * B + (size_t)NS * ng + ng;  // tables out | per-step group counters back | group step counts out
      if (want > stage_cap_) {
        host_free_pinned(stage_);
        stage_ = nullptr;
        stage_cap_ = 0;
        stage_ = (int*)host_malloc_pinned(2 * want * sizeof(int), hipHostMallocDefault);
        stage_cap_ = 2 * want;
      }
      int row = 0, b = 0;
      for (int g = 0; g < ng; ++g) {
        const int L = gh_ * (groups[g].W / pw_);
        for (int i = 0; i < groups[g].B; ++i, ++b) {
          stage_[b] = row;
          stage_[B + b] = L;
          stage_[2 * B + b] = g;
          row += L;
        }
      }
      YMK_HIP(hipMemcpyAsync(tab, stage_, (size_t)3 * B * sizeof(int), hipMemcpyHostToDevice, s));
      YMK_HIP(hipMemsetAsync(gopen, 0, (size_t)NS * ng * sizeof(int), s));
      enc_tab.qoff = enc_tab.koff = mem_tab.koff = tab;
      enc_tab.qlen = enc_tab.klen = mem_tab.klen = tab + B;
    }
    const SeqTab* enc_t = ragged ? &enc_tab : nullptr;
    const SeqTab* mem_t = ragged ? &mem_tab : nullptr;
    const int* gid = ragged ? tab + 2 * B : nullptr;
    int* gop = ragged ? gopen : nullptr;
    const int L = Lmax;  // uniform length when not ragged

    {
      size_t row = 0;
      for (int g = 0; g < ng; ++g) {
        const int Bg = groups[g].B, W = groups[g].W, gw = W / pw_;
        Tensor x4{x4buf, Bg, img_h_, W, 4, 4};
        Tensor tk{xs + row * D, Bg, gh_, gw, D, D};
        nchw3_to_nhwc4(s, groups[g].x, Bg, img_h_, W, x4);
        ConvArgs a;
        a.stride = ph_;
        a.stride_w = pw_;
        conv2d(s, x4, patch_, a, tk);
        add_pos_embed(s, tk.p, pos_embed_, Bg, gh_, gw, full_gw_, D);
        row += (size_t)Bg * gh_ * gw;
      }
    }
    const float scale = 1.f / std::sqrt((float)hd);
    // "conv_split_encoder" (>= 0): operand precision of the ViT blocks' linear layers alone - the memory K|V projection, the
    // decoder and the vocabulary head then follow "conv_split" (the logits come straight out of the head GEMM, so its
    // products are the ones whose rounding shows; the encoder's pass through twelve LayerNorms on the way)
    const int enc_split = (int)param("conv_split_encoder", -1);
    std::optional<ConvSplitScope> enc_scope;
    if (enc_split >= 0) enc_scope.emplace(enc_split);
    for (const EncBlock& b : blocks_) {
      // records of the GEMM inputs: static for the LayerNorm outputs; the q|k|v GEMM leaves one that also bounds the attention
      // output (a convex combination of V rows); fc1 leaves one for the hidden state
      unsigned *qkv_rec = arena.amax_next(), *h_rec = arena.amax_next();
      ln_gemm(s, xs, b.ln1g, b.ln1b, 1e-6f, y, M, D, b.qkv, ACT_NONE, qkv, 3 * D, b.ln1_rec, qkv_rec);
      flash_attention(s, qkv, qkv + D, qkv + 2 * D, att, B, eh_, L, L, hd, 3 * D, 3 * D, 3 * D, D, (long)L * 3 * D,
                      (long)L * 3 * D, (long)L * 3 * D, (long)L * D, scale, enc_t, qkv_rec, qkv_rec, qkv_rec);
      gemm(s, att, M, D, D, b.proj, ACT_NONE, xs, D, xs, D, nullptr, nullptr, EPI_STORE, qkv_rec);
      // norm2 -> fc1 -> GELU -> fc2 -> + residual: one launch with the hidden state on chip where the fused kernel runs the
      // layer (ymk_vit_mlp.hip), else LayerNorm (folded or not) + the two GEMMs through the [M][4 D] buffer
      if (parseq_no_mlp_fusion() || !vit_mlp_fused(s, xs, M, D, b.ln2g, b.ln2b, 1e-6f, b.ln2_bound, b.fc1, b.fc2)) {
        ln_gemm(s, xs, b.ln2g, b.ln2b, 1e-6f, y, M, D, b.fc1, ACT_GELU, hbuf, b.fc1.cout, b.ln2_rec, h_rec);
        gemm(s, hbuf, M, b.fc1.cout, b.fc1.cout, b.fc2, ACT_NONE, xs, D, xs, D, nullptr, nullptr, EPI_STORE, h_rec);
      }
    }
    ln(s, xs, enc_ng_, enc_nb_, 1e-6f, mem, M, D);
    enc_scope.reset();

    // ---------------- decoder: batch-invariant pieces + memory K|V
    memkv_rec_ = arena.amax_next();
    gemm(s, mem, M, D, D, ca_kv_, ACT_NONE, nullptr, 0, memkv, 2 * D, nullptr, nullptr, EPI_STORE, enc_n_rec_, memkv_rec_);
    ln(s, posq_, nqg_, nqb_, 1e-5f, t1, NS, D);
    gemm(s, t1, NS, D, D, sa_q_, ACT_NONE, nullptr, 0, qsa, D);
    init_decode(s, tok, NS, state, bos_, pad_, B);  // tok[:, 0] = bos, rest pad; state = {0, 0, -1, 0}
    const int dhd = D / dh_;
    const float dscale = 1.f / std::sqrt((float)dhd);
    // Early stop without stalling the queue: step i notes in not_done[i] whether any row still lacks an <eos> (0 / 1: rows
    // store, they do not count - ymk_seq.hip) and publishes it to mapped pinned host memory; the host reads the flag of step
    // i - LAG, so up to LAG speculative steps are in flight.  A speculative step's greedy kernel sees
    // not_done[i-1] == 0 and leaves tokens / repetition state untouched, hence `steps` and every
    // result are exactly those of the reference's step-by-step test (models/parseq.py:245-250).
    constexpr int LAG = 2;
    YMK_HIP(hipMemsetAsync(not_done, 0, (size_t)2 * NS * sizeof(int), s));  // not_done[NS] | arrived[NS]
    int* arrived = not_done + NS;
    // host_flags_ is mapped pinned memory a one-thread launch behind step i's greedy kernel writes
    // (that word + 1, so 0 = "not reported yet"): no copy-engine hop, no event per step.  Only
    // non-speculative steps write, and all of those have completed before a forward returns.
    for (int i = 0; i < NS; ++i) host_flags_[i] = 0;
    auto wait_flag = [&](int i) -> int {
      for (long spin = 0;; ++spin) {
        const int v = __atomic_load_n(&host_flags_[i], __ATOMIC_ACQUIRE);
        if (v != 0) return v - 1;
        if (spin < 4096) __builtin_ia32_pause();  // ~20 us of pure spinning, then give the core away between polls:
        else sched_yield();                       // 8 ranks x 8 pages in flight must not pin 64 host cores
        if ((spin & 0xFFFF) == 0xFFFF) {  // the step may have faulted: do not spin forever
          hipError_t e = hipStreamQuery(s);
          if (e != hipSuccess && e != hipErrorNotReady) YMK_HIP(e);
          if (e == hipSuccess && __atomic_load_n(&host_flags_[i], __ATOMIC_ACQUIRE) == 0)
            throw Error("parseq: AR step " + std::to_string(i) + " never reported its <eos> count");
        }
      }
    };
    int steps = NS;
    const bool publish_launch = g_ar_publish.load(std::memory_order_relaxed) != 0;
    for (int i = 0; i < NS; ++i) {
      const int* prev = i > 0 ? not_done + i - 1 : nullptr;
      if (fused) {
        // the whole query stream of step i in one launch, then the vocabulary head as a GEMM
        DecStepW w = fw_;
        w.qsa = qsa;
        parseq_dec_step(s, w, tok, NS, i, skv, NS, memkv, L, mem_tab.koff, mem_tab.klen, t1, prev, B, gid, gop, ng);
        // the vocabulary head skips the M tiles whose rows all sit in mini-batches that finished at an earlier step
        const int* open_prev = (gop && i > 0) ? gop + (size_t)(i - 1) * ng : nullptr;
        if (ar_rowmax)
          gemm(s, t1, B, D, D, head_, ACT_NONE, nullptr, 0, armax, 2 * head_tiles, open_prev ? gid : nullptr, open_prev, EPI_ROWMAX, dn_rec_);
        else
          gemm(s, t1, B, D, D, head_, ACT_NONE, nullptr, 0, arlog + (size_t)i * C, NS * C, open_prev ? gid : nullptr, open_prev, EPI_STORE, dn_rec_);
      } else {
        // content row i (token tok[:, i]) -> norm_c -> K|V cache row i
        ctx_embed_ln(s, tok, NS, i, 1, emb_, posq_, ncg_, ncb_, 1e-5f, cn, NS, D, B);
        gemm(s, cn + (size_t)i * D, B, D, NS * D, sa_kv_, ACT_NONE, nullptr, 0, skv + (size_t)i * 2 * D, NS * 2 * D);
        // query i attends context 0..i (mask row i of triu(1) blocks nothing among those keys)
        small_attention(s, qsa + (size_t)i * D, skv, skv + D, t1, B, dh_, 1, i + 1, dhd, D, 2 * D, 2 * D, D, 0,
                        (long)NS * 2 * D, (long)NS * 2 * D, D, dscale, nullptr, 0, nullptr, 0);
        gemm(s, t1, B, D, D, sa_o_, ACT_NONE, posq_ + (size_t)i * D, 0, qcur, D);
        stream_tail(s, qcur, B, B, 1, memkv, L, mem_t, t1, t2, hdec, arlog + (size_t)i * C, NS * C);
      }
      // who tells the host: a one-thread launch behind the greedy kernel (default), or - "ar_publish" 0 - that kernel's
      // last-arriving block as in rounds 1-5 (the last step's word is never read)
      int* in_kernel_flag = (!publish_launch && i + 1 < NS) ? host_flags_dev_ + i : nullptr;
      if (ar_rowmax)
        greedy_step(s, armax, (long)2 * head_tiles, head_tiles, i, NS, tok, raw, NS, state, eos_, rep_on_, rep_pmax_, rep_p1_, rep_min_,
                    not_done + i, prev, arrived + i, in_kernel_flag, B, gid, gop, ng, 1);
      else
        greedy_step(s, arlog + (size_t)i * C, (long)NS * C, C, i, NS, tok, raw, NS, state, eos_, rep_on_, rep_pmax_, rep_p1_,
                    rep_min_, not_done + i, prev, arrived + i, in_kernel_flag, B, gid, gop, ng);
      if (publish_launch && i + 1 < NS) publish_open(s, not_done + i, prev, host_flags_dev_ + i);
      if (i + 1 < NS && i >= LAG && wait_flag(i - LAG) == 0) {  // every row held an <eos> after step i - LAG
        steps = i - LAG + 1;
        break;
      }
    }
    if (steps == NS) {  // not stopped inside the loop: the last LAG flags have not been looked at yet
      for (int i = std::max(0, NS - 1 - LAG); i + 1 < NS; ++i)
        if (wait_flag(i) == 0) {
          steps = i + 1;
          break;
        }
    }
    // per-group step counts: group g's own loop stops after the first step that leaves none of ITS rows open
    // (gopen[step][g] == 0); its rows were frozen from then on.  One small copy, before the refinement is queued.
    if (ng > 1) {
      int* back = stage_ + 3 * B;
      YMK_HIP(hipMemcpyAsync(back, gopen, (size_t)NS * ng * sizeof(int), hipMemcpyDeviceToHost, s));
      YMK_HIP(hipStreamSynchronize(s));  // the stream is idle here: the loop above has seen the last real step finish
      for (int g = 0; g < ng; ++g) {
        int sg = steps;
        for (int i = 0; i + 1 < steps; ++i)
          if (back[(size_t)i * ng + g] == 0) {
            sg = i + 1;
            break;
          }
        ar_steps[g] = sg;
      }
      // the refinement masks every row's context beyond ITS mini-batch's step count: the counts go back to the device
      // (into the head of the group-counter buffer, which has been read out)
      int* gs = back + (size_t)NS * ng;
      for (int g = 0; g < ng; ++g) gs[g] = ar_steps[g];
      YMK_HIP(hipMemcpyAsync(gopen, gs, (size_t)ng * sizeof(int), hipMemcpyHostToDevice, s));
    } else {
      ar_steps[0] = steps;
    }

    if (refine_ > 0) {
      tile_rows(s, posq_, NS, D, posq_t, B);
      int S_in = steps;
      const int* prev_raw = raw;
      for (int it = 0; it < refine_; ++it) {
        if (it > 0) {
          row_argmax(s, logits, MR, C, raw);
          S_in = NS;
        }
        refine_prep(s, prev_raw, NS, S_in, bos_, eos_, tok2, kpm, B, (it == 0 && ng > 1) ? gid : nullptr, gopen);
        // rows >= S_in of the context buffer are never written: whatever the workspace held there before (another forward's
        // activations - or, after the arena grew, any bit pattern, NaNs included) would set the fp16 scale of the K|V
        // projection below, which takes max|x| over ALL its input rows: zero them (100 MB at wave scale: ~20 us)
        if (S_in < NS) YMK_HIP(hipMemsetAsync(cn, 0, (size_t)MR * D * sizeof(float), s));
        ctx_embed_ln(s, tok2, NS, 0, S_in, emb_, posq_, ncg_, ncb_, 1e-5f, cn, NS, D, B);
        // project every row of the [B][NS] context buffer; rows >= S_in are stale but never attended (Lk = S_in)
        gemm(s, cn, MR, D, D, sa_kv_, ACT_NONE, nullptr, 0, skv, 2 * D);
        small_attention(s, qsa, skv, skv + D, t1, B, dh_, NS, S_in, dhd, D, 2 * D, 2 * D, D, 0, (long)NS * 2 * D,
                        (long)NS * 2 * D, (long)NS * D, dscale, qmask_, NS, kpm, NS);
        gemm(s, t1, MR, D, D, sa_o_, ACT_NONE, posq_t, D, qcur, D);
        stream_tail(s, qcur, MR, B, NS, memkv, L, mem_t, t1, t2, hdec, logits, C);
      }
      if (rep_on_) rep_cut(s, logits, (long)NS * C, C, NS, state, eos_, B);
      for (int g = 0; g < ng; ++g) out_len[g] = NS;
    } else {
      YMK_HIP(hipMemcpyAsync(logits, arlog, (size_t)MR * C * sizeof(float), hipMemcpyDeviceToDevice, s));
      if (rep_on_) rep_cut(s, logits, (long)NS * C, C, steps, state, eos_, B);  // a cut lies before its row's <eos>
      for (int g = 0; g < ng; ++g) out_len[g] = ar_steps[g];
    }
  }

  int ph_ = 4, pw_ = 8, img_h_ = 32, img_w_ = 800, D_ = 192, eh_ = 6, depth_ = 12, Dd_ = 192, dh_ = 6;
  int ntok_ = 7121, maxlen_ = 100, refine_ = 1, rep_on_ = 1, rep_pmax_ = 8, rep_p1_ = 8, rep_min_ = 3;
  int C_ = 0, eos_ = 0, bos_ = 0, pad_ = 0, nsteps_ = 101, gh_ = 8, full_gw_ = 100;
  ConvW patch_;
  float* pos_embed_ = nullptr;
  std::vector<EncBlock> blocks_;
  float *enc_ng_ = nullptr, *enc_nb_ = nullptr;
  ConvW sa_q_, sa_kv_, sa_o_, ca_q_, ca_kv_, ca_o_, lin1_, lin2_, head_;
  DecStepW fw_{};  // transposed decoder weights for the fused step
  float *n1g_, *n1b_, *n2g_, *n2b_, *nqg_, *nqb_, *ncg_, *ncb_, *dng_, *dnb_;
  unsigned *n1_rec_ = nullptr, *n2_rec_ = nullptr, *dn_rec_ = nullptr, *enc_n_rec_ = nullptr;  // static LayerNorm output records
  unsigned* memkv_rec_ = nullptr;  // this forward's record of the memory K|V projection (what cross-attention outputs are bounded by)
  float *emb_ = nullptr, *posq_ = nullptr;
  unsigned char* qmask_ = nullptr;
  int* host_flags_ = nullptr;      // mapped pinned: (rows still open) + 1 per AR step
  int* host_flags_dev_ = nullptr;  // the same words as the device addresses them
  int* stage_ = nullptr;           // pinned: ragged tables out (3B ints at 0), per-step group counters back (after them)
  size_t stage_cap_ = 0;
  uint64_t shape_key_ = 0;
};

}  // namespace

Model* create_parseq() { return new ParseqModel(); }

void parseq_forward(Model* m, const float* x, int B, int W, float* logits, int* out_len, int* ar_steps, hipStream_t s) {
  auto* p = dynamic_cast<ParseqModel*>(m);
  YMK_CHECK(p != nullptr, "model is not a parseq");
  const PGroup g{x, B, W};
  p->forward_groups(&g, 1, logits, out_len, ar_steps, s);
}
void parseq_forward_groups(Model* m, const float* const* x, const int* b, const int* w, int ng, float* logits, int* out_len,
                           int* ar_steps, hipStream_t s) {
  auto* p = dynamic_cast<ParseqModel*>(m);
  YMK_CHECK(p != nullptr, "model is not a parseq");
  YMK_CHECK(ng > 0 && ng <= 4096, "parseq: 1..4096 mini-batches per call");
  std::vector<PGroup> g((size_t)ng);
  for (int i = 0; i < ng; ++i) g[i] = PGroup{x[i], b[i], w[i]};
  p->forward_groups(g.data(), ng, logits, out_len, ar_steps, s);
}
void parseq_dims(Model* m, int* num_steps, int* num_classes) {
  auto* p = dynamic_cast<ParseqModel*>(m);
  YMK_CHECK(p != nullptr, "model is not a parseq");
  *num_steps = p->num_steps();
  *num_classes = p->num_classes();
}

}  // namespace ymk
