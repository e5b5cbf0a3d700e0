// RT-DETRv2 forward on gfx950 (layout parser and table-structure recogniser share it).
// Follows models/rtdetr.py:16-21: PResNet-50vd (models/layers/rtdetr_backbone.py:245-334) ->
// HybridEncoder (rtdetr_hybrid_encoder.py:365-414: AIFI on the 20x20 level, CSPRep FPN + PAN) ->
// RTDETRTransformerv2 eval path (rtdetrv2_decoder.py:782-814: encoder-token scoring, top-300 query
// selection, 6 decoder layers of self-attention + multi-scale deformable attention + FFN with
// iterative box refinement).  How it differs from the reference's op sequence (same arithmetic):
//   - RepVGG 3x3 + 1x1 branches are folded into one 3x3 kernel at load (the reference keeps them apart);
//   - torch.concat of FPN/PAN inputs never happens: producers write channel slices of one buffer;
//   - the six per-layer value projections of the (layer-invariant) memory run as ONE GEMM (N = 6*256);
//   - token rows are level-major (ymk_det.hip) so the per-token heads are single GEMMs over all images.
#include "ymk_common.h"
#include "ymk_det.h"
#include "ymk_seq.h"

namespace ymk {

namespace {

struct VdBlock {
  ConvW a, b, c, shortc;
  bool has_short = false, pool = false;
  int stride = 1;
};
struct Csp {
  ConvW conv1, conv2, rep[3];
};
struct Mlp3 {
  ConvW l0, l1, l2;
};
struct DecLayer {
  ConvW sa_qk, sa_v, sa_o, offs, attw, outp, lin1, lin2;
  float *n1g, *n1b, *n2g, *n2b, *n3g, *n3b;
  Mlp3 bbox;
};

class RtdetrModel : public Model {
 public:
  const char* kind() const override { return "rtdetr"; }

  ConvW cn(const std::string& name, bool tap4 = false) { return make_conv(pool, ws, name + ".conv", name + ".norm", tap4); }
  float* up(const std::string& n) { return pool.upload(ws.get(n).data); }

  // RepVggBlock.get_equivalent_kernel_bias (rtdetr_hybrid_encoder.py:145-175): 3x3+BN and 1x1+BN -> one 3x3 + bias
  ConvW fused_rep(const std::string& name) {
    const HostTensor& w3 = ws.get(name + ".conv1.conv.weight");
    const HostTensor& w1 = ws.get(name + ".conv2.conv.weight");
    const int co = (int)w3.dims[0], ci = (int)w3.dims[1];
    std::vector<float> k((size_t)co * ci * 9), bias(co);
    auto bn = [&](const std::string& p, int o, float& s, float& t) {
      const float g = ws.get(p + ".weight").data[o], be = ws.get(p + ".bias").data[o];
      const float m = ws.get(p + ".running_mean").data[o], v = ws.get(p + ".running_var").data[o];
      const float sd = std::sqrt(v + 1e-5f);
      s = g / sd;
      t = be - m * g / sd;
    };
    for (int o = 0; o < co; ++o) {
      float s3, t3, s1, t1;
      bn(name + ".conv1.norm", o, s3, t3);
      bn(name + ".conv2.norm", o, s1, t1);
      for (int c = 0; c < ci; ++c) {
        for (int tp = 0; tp < 9; ++tp) k[((size_t)o * ci + c) * 9 + tp] = w3.data[((size_t)o * ci + c) * 9 + tp] * s3;
        k[((size_t)o * ci + c) * 9 + 4] += w1.data[(size_t)o * ci + c] * s1;
      }
      bias[o] = t3 + t1;
    }
    ConvW c;
    c.cout = co;
    c.cin = ci;
    c.kh = c.kw = 3;
    std::vector<float> panel;
    pack_conv_weight(k.data(), co, ci, 3, 3, false, panel, c.kpad, c.ctiles);
    c.w = pool.upload(panel);
    c.bias = pool.upload(bias);
    pool.note(c);
    return c;
  }
  Csp csp(const std::string& name) {
    Csp c;
    c.conv1 = cn(name + ".conv1");
    c.conv2 = cn(name + ".conv2");
    for (int j = 0; j < 3; ++j) c.rep[j] = fused_rep(name + ".bottlenecks." + std::to_string(j));
    return c;
  }
  Mlp3 mlp3(const std::string& name) {
    Mlp3 m;
    m.l0 = make_linear(pool, ws, name + ".layers.0");
    m.l1 = make_linear(pool, ws, name + ".layers.1");
    m.l2 = make_linear(pool, ws, name + ".layers.2");
    return m;
  }
  void split_qkv(const std::string& name, ConvW& wqk, ConvW& wv, ConvW& wo) {
    const HostTensor& w = ws.get(name + ".in_proj_weight");
    const HostTensor& b = ws.get(name + ".in_proj_bias");
    const int D = (int)w.dims[1];
    wqk = make_linear_raw(pool, w.data.data(), b.data.data(), 2 * D, D);
    wv = make_linear_raw(pool, w.data.data() + (size_t)2 * D * D, b.data.data() + 2 * D, D, D);
    wo = make_linear(pool, ws, name + ".out_proj");
  }

  void finalize() override {
    nc_ = (int)param("num_classes", 6);
    nq_ = (int)param("num_queries", 300);
    nl_ = (int)param("num_layers", 6);
    YMK_CHECK((int)param("hidden_dim", 256) == 256, "RT-DETR hidden_dim must be 256");
    const std::string b = "backbone.";
    stem_[0] = cn(b + "conv1.conv1_1", true);
    stem_[1] = cn(b + "conv1.conv1_2");
    stem_[2] = cn(b + "conv1.conv1_3");
    const int counts[4] = {3, 4, 6, 3};
    for (int s = 0; s < 4; ++s)
      for (int i = 0; i < counts[s]; ++i) {
        const std::string p = b + "res_layers." + std::to_string(s) + ".blocks." + std::to_string(i) + ".";
        VdBlock k;
        k.a = cn(p + "branch2a");
        k.b = cn(p + "branch2b");
        k.c = cn(p + "branch2c");
        k.stride = (i == 0 && s != 0) ? 2 : 1;
        if (i == 0) {
          k.has_short = true;
          k.pool = k.stride == 2;  // variant d: AvgPool2d(2, 2, ceil) then 1x1
          k.shortc = cn(p + (k.pool ? "short.conv" : "short"));
        }
        stages_[s].push_back(k);
      }
    const std::string e = "encoder.";
    for (int i = 0; i < 3; ++i) enc_in_[i] = cn(e + "input_proj." + std::to_string(i));
    const std::string a = e + "encoder.0.layers.0.";
    split_qkv(a + "self_attn", aifi_qk_, aifi_v_, aifi_o_);
    aifi_l1_ = make_linear(pool, ws, a + "linear1");
    aifi_l2_ = make_linear(pool, ws, a + "linear2");
    aifi_n1g_ = up(a + "norm1.weight"); aifi_n1b_ = up(a + "norm1.bias");
    aifi_n2g_ = up(a + "norm2.weight"); aifi_n2b_ = up(a + "norm2.bias");
    {
      const HostTensor& pe = ws.get("__aifi_pos_embed");  // [tokens][256], built by the host wrapper
      aifi_pos_rows_ = (int)(pe.numel() / 256);
      aifi_pos_ = pool.upload(pe.data);
    }
    for (int i = 0; i < 2; ++i) {
      lateral_[i] = cn(e + "lateral_convs." + std::to_string(i));
      fpn_[i] = csp(e + "fpn_blocks." + std::to_string(i));
      down_[i] = cn(e + "downsample_convs." + std::to_string(i));
      pan_[i] = csp(e + "pan_blocks." + std::to_string(i));
    }
    const std::string d = "decoder.";
    for (int i = 0; i < 3; ++i) dec_in_[i] = cn(d + "input_proj." + std::to_string(i));
    enc_proj_ = make_linear(pool, ws, d + "enc_output.proj");
    eng_ = up(d + "enc_output.norm.weight"); enb_ = up(d + "enc_output.norm.bias");
    enc_score_ = make_linear(pool, ws, d + "enc_score_head");
    YMK_CHECK(enc_score_.cout == nc_, "enc_score_head width != num_classes");
    enc_bbox_ = mlp3(d + "enc_bbox_head");
    {
      const HostTensor& an = ws.get(d + "anchors");
      const HostTensor& vm = ws.get(d + "valid_mask");
      ntok_ = (int)(an.numel() / 4);
      YMK_CHECK((int)vm.numel() == ntok_, "valid_mask size");
      anchors_ = pool.upload(an.data);
      valid_ = pool.upload(vm.data);
    }
    qph0_ = make_linear(pool, ws, d + "query_pos_head.layers.0");
    qph1_ = make_linear(pool, ws, d + "query_pos_head.layers.1");
    layers_.resize(nl_);
    std::vector<float> vw((size_t)nl_ * 256 * 256), vb((size_t)nl_ * 256);
    for (int i = 0; i < nl_; ++i) {
      const std::string p = d + "decoder.layers." + std::to_string(i) + ".";
      DecLayer& L = layers_[i];
      split_qkv(p + "self_attn", L.sa_qk, L.sa_v, L.sa_o);
      L.offs = make_linear(pool, ws, p + "cross_attn.sampling_offsets");
      L.attw = make_linear(pool, ws, p + "cross_attn.attention_weights");
      YMK_CHECK(L.offs.cout == 192 && L.attw.cout == 96, "deformable attention wants 8 heads x 3 levels x 4 points");
      L.outp = make_linear(pool, ws, p + "cross_attn.output_proj");
      const HostTensor& w = ws.get(p + "cross_attn.value_proj.weight");
      const HostTensor& bb = ws.get(p + "cross_attn.value_proj.bias");
      std::copy(w.data.begin(), w.data.end(), vw.begin() + (size_t)i * 256 * 256);
      std::copy(bb.data.begin(), bb.data.end(), vb.begin() + (size_t)i * 256);
      L.lin1 = make_linear(pool, ws, p + "linear1");
      L.lin2 = make_linear(pool, ws, p + "linear2");
      L.n1g = up(p + "norm1.weight"); L.n1b = up(p + "norm1.bias");
      L.n2g = up(p + "norm2.weight"); L.n2b = up(p + "norm2.bias");
      L.n3g = up(p + "norm3.weight"); L.n3b = up(p + "norm3.bias");
      L.bbox = mlp3(d + "dec_bbox_head." + std::to_string(i));
    }
    value_all_ = make_linear_raw(pool, vw.data(), vb.data(), nl_ * 256, 256);
    score_ = make_linear(pool, ws, d + "dec_score_head." + std::to_string(nl_ - 1));
    ws.clear();
    finalized = true;
  }

  // x: device fp32 [B][3][H][W]; logits: [B][nq][nc]; boxes: [B][nq][4] (cxcywh in [0,1])
  void forward(const float* x, int B, int H, int W, float* logits, float* boxes, hipStream_t s) {
    YMK_CHECK(finalized, "model not finalized");
    ForwardScope forward_scope;
    ConvSplitScope split_scope(conv_split(), split_ctx.get(), SPLIT_MODEL_DEFAULT);
    YMK_CHECK(B > 0 && H % 32 == 0 && W % 32 == 0, "rtdetr input must be a multiple of 32");
    const int tok = (H / 8) * (W / 8) + (H / 16) * (W / 16) + (H / 32) * (W / 32);
    YMK_CHECK(tok == ntok_, "input size does not match the checkpoint's anchors (eval_spatial_size)");
    YMK_CHECK((H / 32) * (W / 32) == aifi_pos_rows_, "AIFI position table does not match the input size");
    const uint64_t key = ((uint64_t)B << 40) | ((uint64_t)H << 20) | (uint64_t)W;
    if (key != shape_key_) {
      arena.dry_run = true;
      arena.reset();
      run(x, B, H, W, logits, boxes, s);
      arena.dry_run = false;
      const size_t need = arena.used();
      arena.reset();
      if (need > arena.capacity()) {
        forward_sync(s);
        arena.reserve(need);
      }
      shape_key_ = key;
    }
    arena.reset();
    run(x, B, H, W, logits, boxes, s);
  }

  void reserve(int n, int h, int w, hipStream_t s) override {
    YMK_CHECK(finalized, "model not finalized");
    YMK_CHECK(n > 0 && h % 32 == 0 && w % 32 == 0, "rtdetr reserve: sizes must be multiples of 32");
    arena.dry_run = true;
    arena.reset();
    run(nullptr, n, h, w, nullptr, nullptr, s);
    arena.dry_run = false;
    const size_t need = arena.used();
    arena.reset();
    if (need > arena.capacity()) {
      YMK_HIP(hipStreamSynchronize(s));
      arena.reserve(need);
    }
    shape_key_ = 0;
  }

 private:
  bool dry() const { return arena.dry_run; }

  Tensor conv(hipStream_t s, const Tensor& in, const ConvW& w, int stride, int pad, int act, const Tensor* res = nullptr,
              const Tensor* into = nullptr, bool res_post = false, bool out_planes = false) {
    Tensor out = into ? *into
                      : arena.tensor(in.n, conv_out_dim(in.h, w.kh, stride, pad, 1), conv_out_dim(in.w, w.kw, stride, pad, 1),
                                     w.cout);
    if (!into) out.amax = arena.amax_next();  // a fresh output gets a max|x| record (ymk_common.h); a slice keeps its buffer's
    out.planes = out_planes && !into && out.amax != nullptr;
    if (dry()) return out;
    ConvArgs a;
    a.stride = stride;
    a.pad = pad;
    a.act = act;
    a.res = res;
    a.res_post = res_post;
    a.out_planes = out.planes;
    conv2d(s, in, w, a, out);
    return out;
  }

  Tensor csp_fwd(hipStream_t s, const Tensor& x, const Csp& c) {
    Tensor x1 = conv(s, x, c.conv1, 1, 0, ACT_SILU);
    for (int j = 0; j < 3; ++j) x1 = conv(s, x1, c.rep[j], 1, 1, ACT_SILU);
    return conv(s, x, c.conv2, 1, 0, ACT_SILU, &x1, nullptr, /*res_post=*/true);
  }

  float* lin(hipStream_t s, const float* in, int M, const ConvW& w, int act, const float* res = nullptr, float* out = nullptr) {
    if (!out) out = arena.alloc_f((size_t)M * w.cout);
    if (dry()) return out;
    gemm(s, in, M, w.cin, w.cin, w, act, res, res ? w.cout : 0, out, w.cout);
    return out;
  }
  float* mlp3_fwd(hipStream_t s, const float* in, int M, const Mlp3& m) {
    float* a = lin(s, in, M, m.l0, ACT_RELU);
    float* b = lin(s, a, M, m.l1, ACT_RELU);
    return lin(s, b, M, m.l2, ACT_NONE);
  }
  float* ln(hipStream_t s, const float* in, int M, const float* g, const float* b) {
    float* out = arena.alloc_f((size_t)M * 256);
    if (!dry()) layernorm(s, in, 256, 0, g, b, 1e-5f, out, 256, M, 256);
    return out;
  }

  void run(const float* x, int B, int H, int W, float* logits, float* boxes, hipStream_t s) {
    const int D = 256;
    arena.amax_begin(s, 128);  // max|x| records: one per convolution output / concat buffer (about 100)
    // ---------------- PResNet-50vd
    Tensor x4 = arena.tensor(B, H, W, 4);
    if (!dry()) nchw3_to_nhwc4(s, x, B, H, W, x4);
    Tensor c = conv(s, x4, stem_[0], 2, 1, ACT_RELU);
    c = conv(s, c, stem_[1], 1, 1, ACT_RELU);
    c = conv(s, c, stem_[2], 1, 1, ACT_RELU);
    Tensor p = arena.tensor(B, (c.h - 1) / 2 + 1, (c.w - 1) / 2 + 1, c.c);
    p.amax = c.amax;  // window maxima of c: bounded by c's max|x|
    if (!dry()) maxpool3x3s2(s, c, p);
    Tensor feat[4];
    Tensor cur = p;
    for (int st = 0; st < 4; ++st) {
      for (const VdBlock& k : stages_[st]) {
        // branch2a's output has one reader, the 3 x 3 branch2b: fp16 planes in HBM where both launches allow (Tensor::planes)
        ConvArgs pa, pb;
        pa.act = pb.act = ACT_RELU;
        pb.stride = k.stride;
        pb.pad = 1;
        const bool planes = conv_planes_pair_ok(cur, k.a, pa, k.b, pb);
        Tensor t1 = conv(s, cur, k.a, 1, 0, ACT_RELU, nullptr, nullptr, false, planes);
        Tensor t2 = conv(s, t1, k.b, k.stride, 1, ACT_RELU);
        Tensor sh = cur;
        if (k.has_short) {
          Tensor src = cur;
          if (k.pool) {
            src = arena.tensor(B, (cur.h + 1) / 2, (cur.w + 1) / 2, cur.c);
            src.amax = cur.amax;  // window means of cur
            if (!dry()) avgpool2x2_ceil(s, cur, src);
          }
          sh = conv(s, src, k.shortc, 1, 0, ACT_NONE);
        }
        cur = conv(s, t2, k.c, 1, 0, ACT_RELU, &sh);
      }
      feat[st] = cur;
    }
    const Tensor &C3 = feat[1], &C4 = feat[2], &C5 = feat[3];

    // ---------------- HybridEncoder
    Tensor P5 = conv(s, C5, enc_in_[2], 1, 0, ACT_NONE);
    const int T = P5.h * P5.w, M5 = B * T;
    {  // AIFI: post-LN encoder layer on the coarsest level, q = k = src + pos, v = src
      float* sp = arena.alloc_f((size_t)M5 * D);
      if (!dry()) add_bcast(s, P5.p, aifi_pos_, T, sp, M5, D);
      float* qk = lin(s, sp, M5, aifi_qk_, ACT_NONE);
      float* v = lin(s, P5.p, M5, aifi_v_, ACT_NONE);
      float* att = arena.alloc_f((size_t)M5 * D);
      if (!dry())
        flash_attention(s, qk, qk + D, v, att, B, 8, T, T, 32, 2 * D, 2 * D, D, D, (long)T * 2 * D, (long)T * 2 * D,
                        (long)T * D, (long)T * D, 1.f / std::sqrt(32.f));
      float* t = lin(s, att, M5, aifi_o_, ACT_NONE, P5.p);
      float* s1 = ln(s, t, M5, aifi_n1g_, aifi_n1b_);
      float* h = lin(s, s1, M5, aifi_l1_, ACT_GELU);
      float* t2 = lin(s, h, M5, aifi_l2_, ACT_NONE, s1);
      if (!dry()) layernorm(s, t2, D, 0, aifi_n2g_, aifi_n2b_, 1e-5f, P5.p, D, M5, D);  // back into the feature map
      P5.amax = nullptr;  // rewritten in place: the convolution's record no longer describes it
    }
    Tensor catA = arena.tensor(B, C4.h, C4.w, 2 * D);     // [up(L5) | P4]
    Tensor catB = arena.tensor(B, C3.h, C3.w, 2 * D);     // [up(L4) | P3]
    Tensor catP0 = arena.tensor(B, C4.h, C4.w, 2 * D);    // [down(F3) | L4]
    Tensor catP1 = arena.tensor(B, C5.h, C5.w, 2 * D);    // [down(N4) | L5]
    // one max|x| record per concat buffer, filled by the convolutions that write its halves; a half that is a nearest
    // up-sampling of another buffer's half takes that buffer's record over (amax_merge) once its producer has run
    catA.amax = arena.amax_next();
    catB.amax = arena.amax_next();
    catP0.amax = arena.amax_next();
    catP1.amax = arena.amax_next();
    Tensor sA0 = catA.slice_c(0, D), sA1 = catA.slice_c(D, D), sB0 = catB.slice_c(0, D), sB1 = catB.slice_c(D, D);
    Tensor sP00 = catP0.slice_c(0, D), sP01 = catP0.slice_c(D, D), sP10 = catP1.slice_c(0, D), sP11 = catP1.slice_c(D, D);
    conv(s, C4, enc_in_[1], 1, 0, ACT_NONE, nullptr, &sA1);
    conv(s, C3, enc_in_[0], 1, 0, ACT_NONE, nullptr, &sB1);
    conv(s, P5, lateral_[0], 1, 0, ACT_SILU, nullptr, &sP11);  // L5
    if (!dry()) {
      upsample_nearest2x(s, sP11, sA0);
      amax_merge(s, catA.amax, catP1.amax);  // (catP1's record holds L5 alone at this point)
    }
    Tensor F4 = csp_fwd(s, catA, fpn_[0]);
    conv(s, F4, lateral_[1], 1, 0, ACT_SILU, nullptr, &sP01);  // L4
    if (!dry()) {
      upsample_nearest2x(s, sP01, sB0);
      amax_merge(s, catB.amax, catP0.amax);
    }
    Tensor F3 = csp_fwd(s, catB, fpn_[1]);
    conv(s, F3, down_[0], 2, 1, ACT_SILU, nullptr, &sP00);
    Tensor N4 = csp_fwd(s, catP0, pan_[0]);
    conv(s, N4, down_[1], 2, 1, ACT_SILU, nullptr, &sP10);
    Tensor N5 = csp_fwd(s, catP1, pan_[1]);
    const Tensor* outs[3] = {&F3, &N4, &N5};

    // ---------------- decoder input: level-major token memory
    DetGeom g;
    g.B = B;
    g.ntok = 0;
    for (int l = 0; l < 3; ++l) {
      g.h[l] = outs[l]->h;
      g.w[l] = outs[l]->w;
      g.hw[l] = g.h[l] * g.w[l];
      g.off[l] = g.ntok;
      g.ntok += g.hw[l];
    }
    const int R = B * g.ntok;
    float* mem = arena.alloc_f((size_t)R * D);
    for (int l = 0; l < 3; ++l) {
      Tensor dst{mem + (size_t)B * g.off[l] * D, B, g.h[l], g.w[l], D, D};
      conv(s, *outs[l], dec_in_[l], 1, 0, ACT_NONE, nullptr, &dst);
    }
    float* mm = arena.alloc_f((size_t)R * D);
    if (!dry()) mask_rows(s, mem, valid_, mm, g, D);
    float* om = ln(s, lin(s, mm, R, enc_proj_, ACT_NONE), R, eng_, enb_);
    float* elog = lin(s, om, R, enc_score_, ACT_NONE);
    float* ebox = mlp3_fwd(s, om, R, enc_bbox_);
    int* idx = (int*)arena.alloc_bytes((size_t)B * nq_ * sizeof(int));
    unsigned* topk_keys = (unsigned*)arena.alloc_bytes((size_t)R * sizeof(unsigned));
    const int MQ = B * nq_;
    float* tgt = arena.alloc_f((size_t)MQ * D);
    float* ref = arena.alloc_f((size_t)MQ * 4);
    if (!dry()) {
      topk_tokens(s, elog, nc_, g, nq_, topk_keys, idx);
      gather_queries(s, om, ebox, anchors_, idx, g, nq_, D, tgt, ref);
    }
    float* vall = lin(s, mem, R, value_all_, ACT_NONE);  // every layer's value projection at once
    const int ldv = nl_ * D;

    for (int i = 0; i < nl_; ++i) {
      const DecLayer& L = layers_[i];
      float* qpe = lin(s, lin(s, ref, MQ, qph0_, ACT_RELU), MQ, qph1_, ACT_NONE);
      float* q = arena.alloc_f((size_t)MQ * D);
      if (!dry()) add_bcast(s, tgt, qpe, MQ, q, MQ, D);
      float* qk = lin(s, q, MQ, L.sa_qk, ACT_NONE);
      float* v = lin(s, tgt, MQ, L.sa_v, ACT_NONE);
      float* att = arena.alloc_f((size_t)MQ * D);
      if (!dry())
        flash_attention(s, qk, qk + D, v, att, B, 8, nq_, nq_, 32, 2 * D, 2 * D, D, D, (long)nq_ * 2 * D, (long)nq_ * 2 * D,
                        (long)nq_ * D, (long)nq_ * D, 1.f / std::sqrt(32.f));
      tgt = ln(s, lin(s, att, MQ, L.sa_o, ACT_NONE, tgt), MQ, L.n1g, L.n1b);
      float* q2 = arena.alloc_f((size_t)MQ * D);
      if (!dry()) add_bcast(s, tgt, qpe, MQ, q2, MQ, D);
      float* of = lin(s, q2, MQ, L.offs, ACT_NONE);
      float* aw = lin(s, q2, MQ, L.attw, ACT_NONE);
      float* smp = arena.alloc_f((size_t)MQ * D);
      if (!dry()) deform_sample(s, of, aw, ref, vall + (size_t)i * D, ldv, g, nq_, smp);
      tgt = ln(s, lin(s, smp, MQ, L.outp, ACT_NONE, tgt), MQ, L.n2g, L.n2b);
      float* hh = lin(s, tgt, MQ, L.lin1, ACT_RELU);
      tgt = ln(s, lin(s, hh, MQ, L.lin2, ACT_NONE, tgt), MQ, L.n3g, L.n3b);
      float* delta = mlp3_fwd(s, tgt, MQ, L.bbox);
      const bool last = i == nl_ - 1;
      float* nref = last ? boxes : arena.alloc_f((size_t)MQ * 4);
      if (!dry()) refine_boxes(s, delta, ref, nref, (size_t)MQ * 4);
      if (last) {
        lin(s, tgt, MQ, score_, ACT_NONE, nullptr, logits);
      }
      ref = nref;
    }
  }

  int nc_ = 6, nq_ = 300, nl_ = 6, ntok_ = 8400, aifi_pos_rows_ = 400;
  ConvW stem_[3];
  std::vector<VdBlock> stages_[4];
  ConvW enc_in_[3], lateral_[2], down_[2], dec_in_[3];
  Csp fpn_[2], pan_[2];
  ConvW aifi_qk_, aifi_v_, aifi_o_, aifi_l1_, aifi_l2_;
  float *aifi_n1g_, *aifi_n1b_, *aifi_n2g_, *aifi_n2b_, *aifi_pos_ = nullptr;
  ConvW enc_proj_, enc_score_, qph0_, qph1_, value_all_, score_;
  Mlp3 enc_bbox_;
  float *eng_, *enb_, *anchors_ = nullptr, *valid_ = nullptr;
  std::vector<DecLayer> layers_;
  uint64_t shape_key_ = 0;
};

}  // namespace

Model* create_rtdetr() { return new RtdetrModel(); }

void rtdetr_forward(Model* m, const float* x, int B, int H, int W, float* logits, float* boxes, hipStream_t s) {
  auto* p = dynamic_cast<RtdetrModel*>(m);
  YMK_CHECK(p != nullptr, "model is not an rtdetr");
  p->forward(x, B, H, W, logits, boxes, s);
}

}  // namespace ymk
