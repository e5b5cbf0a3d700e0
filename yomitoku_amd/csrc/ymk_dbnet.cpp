// DBNet++ text detector forward on gfx950.
// Follows the reference's graph: models/dbnet_plus.py:30-38 (torchvision resnet50 v1.5 with
// replace_stride_with_dilation=[F,F,T]), :200-230 (FPN-style decoder), :110-119 (binarize head) and
// models/layers/dbnet_feature_attention.py:69-79,150-160 (adaptive scale fusion).
// Weights arrive under the reference's state-dict names.
#include "ymk_common.h"

namespace ymk {

namespace {

struct Bottleneck {
  ConvW c1, c2, c3, down;
  bool has_down = false;
  int stride = 1, dil = 1;
};

class DBNetModel : public Model {
 public:
  const char* kind() const override { return "dbnet"; }

  void finalize() override {
    const std::string bb = "backbone.body.";
    stem_ = make_conv(pool, ws, bb + "conv1", bb + "bn1", /*tap4=*/true);
    const int nblocks[4] = {3, 4, 6, 3};
    for (int L = 0; L < 4; ++L) {
      layers_[L].clear();
      for (int i = 0; i < nblocks[L]; ++i) {
        const std::string p = bb + "layer" + std::to_string(L + 1) + "." + std::to_string(i) + ".";
        Bottleneck b;
        b.c1 = make_conv(pool, ws, p + "conv1", p + "bn1");
        b.c2 = make_conv(pool, ws, p + "conv2", p + "bn2");
        b.c3 = make_conv(pool, ws, p + "conv3", p + "bn3");
        b.has_down = ws.has(p + "downsample.0.weight");
        if (b.has_down) b.down = make_conv(pool, ws, p + "downsample.0", p + "downsample.1");
        // layer2/3 stride 2 on block 0; layer4 trades its stride for dilation 2 (blocks >= 1)
        b.stride = (i == 0 && (L == 1 || L == 2)) ? 2 : 1;
        b.dil = (L == 3 && i > 0) ? 2 : 1;
        layers_[L].push_back(b);
      }
    }
    const std::string d = "decoder.";
    for (int L = 0; L < 4; ++L) {
      in_proj_[L] = make_conv(pool, ws, d + "input_proj.layer" + std::to_string(L + 1), "");
      out_proj_[L] = make_conv(pool, ws, d + "out_proj.layer" + std::to_string(L + 1) + (L == 0 ? "" : ".0"), "");
    }
    asf_conv_ = make_conv(pool, ws, d + "concat_attention.conv", "");
    const std::string ea = d + "concat_attention.enhanced_attention.";
    {
      const HostTensor& w1 = ws.get(ea + "channel_wise.1.weight");  // [16][64][1][1]
      const HostTensor& w2 = ws.get(ea + "channel_wise.3.weight");  // [64][16][1][1]
      asf_c_ = (int)w1.dims[1];
      asf_cmid_ = (int)w1.dims[0];
      YMK_CHECK(asf_c_ == 64 && (int)w2.dims[0] == 64, "ASF expects 64 inner channels");
      asf_w1_ = pool.upload(w1.data);
      asf_w2_ = pool.upload(w2.data);
      asf_sp33_ = pool.upload(ws.get(ea + "spatial_wise.0.weight").data);
      asf_sp11_ = ws.get(ea + "spatial_wise.2.weight").data[0];
      const HostTensor& wa = ws.get(ea + "attention_wise.0.weight");  // [4][64][1][1]
      YMK_CHECK((int)wa.dims[0] == 4, "ASF expects 4 scales");
      asf_watt_ = pool.upload(wa.data);
    }
    bin_conv_ = make_conv(pool, ws, d + "binarize.0", d + "binarize.1");
    {
      // ConvTranspose2d(64,64,2,2): weight [ci][co][a][b] -> 1x1 panel with N = (a*2+b)*64 + co
      const HostTensor& w = ws.get(d + "binarize.3.weight");
      const HostTensor& b = ws.get(d + "binarize.3.bias");
      const int ci = (int)w.dims[0], co = (int)w.dims[1];
      YMK_CHECK(w.dims[2] == 2 && w.dims[3] == 2, "binarize.3 must be a 2x2 transposed conv");
      std::vector<float> lin((size_t)4 * co * ci);
      for (int c = 0; c < ci; ++c)
        for (int o = 0; o < co; ++o)
          for (int ab = 0; ab < 4; ++ab) lin[((size_t)ab * co + o) * ci + c] = w.data[((size_t)c * co + o) * 4 + ab];
      const HostTensor& g = ws.get(d + "binarize.4.weight");
      const HostTensor& be = ws.get(d + "binarize.4.bias");
      const HostTensor& m = ws.get(d + "binarize.4.running_mean");
      const HostTensor& v = ws.get(d + "binarize.4.running_var");
      std::vector<float> sc(4 * co), bi(4 * co);
      for (int ab = 0; ab < 4; ++ab)
        for (int o = 0; o < co; ++o) {
          const float s = g.data[o] / std::sqrt(v.data[o] + 1e-5f);
          sc[ab * co + o] = s;
          bi[ab * co + o] = be.data[o] + (b.data[o] - m.data[o]) * s;
        }
      deconv1_ = make_linear_raw(pool, lin.data(), nullptr, 4 * co, ci);
      deconv1_.scale = pool.upload(sc);
      deconv1_.bias = pool.upload(bi);
      pool.note(deconv1_);  // (the panel with its scale: what the split copy folds the row's power of two into)
    }
    {
      const HostTensor& w = ws.get(d + "binarize.6.weight");  // [64][1][2][2] == [c][ab]
      YMK_CHECK(w.dims[0] == 64 && w.dims[1] == 1, "binarize.6 must be 64->1");
      deconv2_w_ = pool.upload(w.data);
      deconv2_b_ = ws.get(d + "binarize.6.bias").data[0];
    }
    ws.clear();
    finalized = true;
  }

  // x: device NCHW fp32 [n][3][h][w] (h, w multiples of 32); prob: device [n][1][h][w]
  void forward(const float* x_nchw, int n, int h, int w, float* prob, hipStream_t s) {
    YMK_CHECK(finalized, "model not finalized");
    ForwardScope forward_scope;
    ConvSplitScope split_scope(conv_split(), split_ctx.get(), SPLIT_MODEL_DEFAULT);
    YMK_CHECK(n > 0 && h % 32 == 0 && w % 32 == 0 && h >= 32 && w >= 32, "dbnet input must be a multiple of 32");
    const uint64_t key = ((uint64_t)n << 40) | ((uint64_t)h << 20) | (uint64_t)w;
    if (key != shape_key_) {
      arena.dry_run = true;
      arena.reset();
      run(x_nchw, n, h, w, prob, s);
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
    run(x_nchw, n, h, w, prob, s);
  }

  void reserve(int n, int h, int w, hipStream_t s) override {
    YMK_CHECK(finalized, "model not finalized");
    YMK_CHECK(n > 0 && h % 32 == 0 && w % 32 == 0 && h >= 32 && w >= 32, "dbnet reserve: sizes must be multiples of 32");
    size_t need = 0;
    for (int turn = 0; turn < 2; ++turn) {  // either orientation of the page
      arena.dry_run = true;
      arena.reset();
      run(nullptr, n, turn ? w : h, turn ? h : w, nullptr, s);
      arena.dry_run = false;
      need = std::max(need, arena.used());
      arena.reset();
    }
    if (need > arena.capacity()) {
      YMK_HIP(hipStreamSynchronize(s));
      arena.reserve(need);
    }
    shape_key_ = 0;
  }

 private:
  // `rec`: the max|x| record the launch folds its outputs into (ymk_common.h); by default a fresh one for a fresh output
  Tensor conv(hipStream_t s, const Tensor& in, const ConvW& w, int stride, int pad, int dil, int act,
              const Tensor* res = nullptr, const Tensor* into = nullptr, unsigned* rec = nullptr, bool out_planes = false) {
    Tensor out;
    if (into) {
      out = *into;
    } else {
      out = arena.tensor(in.n, conv_out_dim(in.h, w.kh, stride, pad, dil), conv_out_dim(in.w, w.kw, stride, pad, dil),
                         w.cout);
      out.amax = arena.amax_next();
    }
    if (rec) out.amax = rec;
    out.planes = out_planes && out.amax != nullptr;
    if (arena.dry_run) return out;
    ConvArgs a;
    a.stride = stride;
    a.pad = pad;
    a.dil = dil;
    a.act = act;
    a.res = res;
    a.out_planes = out.planes;
    conv2d(s, in, w, a, out);
    return out;
  }

  Tensor bottleneck(hipStream_t s, const Tensor& x, const Bottleneck& b) {
    // the tensor between the 1 x 1 reduction and the 3 x 3 has ONE reader: where both launches run fp16-split kernels that
    // support it, it lives in HBM as the two fp16 planes the 3 x 3 multiplies with (Tensor::planes, conv_planes_pair_ok)
    ConvArgs a1, a2;
    a1.act = a2.act = ACT_RELU;
    a2.stride = b.stride;
    a2.pad = a2.dil = b.dil;
    const bool planes = conv_planes_pair_ok(x, b.c1, a1, b.c2, a2);
    Tensor t1 = conv(s, x, b.c1, 1, 0, 1, ACT_RELU, nullptr, nullptr, nullptr, planes);
    Tensor t2 = conv(s, t1, b.c2, b.stride, b.dil, b.dil, ACT_RELU);
    Tensor idn = x;
    if (b.has_down) idn = conv(s, x, b.down, b.stride, 0, 1, ACT_NONE);
    return conv(s, t2, b.c3, 1, 0, 1, ACT_RELU, &idn);
  }

  void run(const float* x_nchw, int n, int h, int w, float* prob, hipStream_t s) {
    const bool dry = arena.dry_run;
    arena.amax_begin(s, 96);  // one max|x| record per convolution output (about 70 of them)
    Tensor x4 = arena.tensor(n, h, w, 4);
    if (!dry) nchw3_to_nhwc4(s, x_nchw, n, h, w, x4);
    Tensor c1 = conv(s, x4, stem_, 2, 3, 1, ACT_RELU);
    Tensor p = arena.tensor(n, (c1.h + 2 - 3) / 2 + 1, (c1.w + 2 - 3) / 2 + 1, c1.c);
    p.amax = c1.amax;  // a maximum over windows of c1: bounded by c1's own max|x|
    if (!dry) maxpool3x3s2(s, c1, p);
    Tensor feat[4];
    Tensor cur = p;
    for (int L = 0; L < 4; ++L) {
      for (const Bottleneck& b : layers_[L]) cur = bottleneck(s, cur, b);
      feat[L] = cur;
    }
    // ---- decoder (dbnet_plus.py:200-230)
    Tensor p4 = conv(s, feat[3], in_proj_[3], 1, 0, 1, ACT_NONE);
    Tensor p3;
    if (feat[2].h == p4.h && feat[2].w == p4.w) {
      p3 = conv(s, feat[2], in_proj_[2], 1, 0, 1, ACT_NONE, &p4);
    } else {
      Tensor up = arena.tensor(n, feat[2].h, feat[2].w, p4.c);
      if (!dry) upsample_bilinear(s, p4, up, nullptr);
      p3 = conv(s, feat[2], in_proj_[2], 1, 0, 1, ACT_NONE, &up);
    }
    Tensor up3 = arena.tensor(n, feat[1].h, feat[1].w, p3.c);
    if (!dry) upsample_bilinear(s, p3, up3, nullptr);
    Tensor p2 = conv(s, feat[1], in_proj_[1], 1, 0, 1, ACT_NONE, &up3);
    Tensor up2 = arena.tensor(n, feat[0].h, feat[0].w, p2.c);
    if (!dry) upsample_bilinear(s, p2, up2, nullptr);
    Tensor p1 = conv(s, feat[0], in_proj_[0], 1, 0, 1, ACT_NONE, &up2);

    const int fh = p1.h, fw = p1.w;
    Tensor fuse = arena.tensor(n, fh, fw, 256);
    // ONE record for the four parts of the concat buffer: three of them are bilinear up-samplings (convex combinations: no
    // value beyond their source's), the fourth a convolution writing its slice
    fuse.amax = arena.amax_next();
    {
      Tensor o4 = conv(s, p4, out_proj_[3], 1, 1, 1, ACT_NONE, nullptr, nullptr, fuse.amax);
      Tensor o3 = conv(s, p3, out_proj_[2], 1, 1, 1, ACT_NONE, nullptr, nullptr, fuse.amax);
      Tensor o2 = conv(s, p2, out_proj_[1], 1, 1, 1, ACT_NONE, nullptr, nullptr, fuse.amax);
      Tensor s0 = fuse.slice_c(0, 64), s1 = fuse.slice_c(64, 64), s2 = fuse.slice_c(128, 64), s3 = fuse.slice_c(192, 64);
      if (!dry) {
        // nn.Upsample(scale_factor=4 / 4 / 2): output = floor(in * scale)
        YMK_CHECK(o4.h * 4 == fh && o3.h * 4 == fh && o2.h * 2 == fh, "decoder pyramid shape");
        upsample_bilinear(s, o4, s0, nullptr);
        upsample_bilinear(s, o3, s1, nullptr);
        upsample_bilinear(s, o2, s2, nullptr);
      }
      conv(s, p1, out_proj_[0], 1, 1, 1, ACT_NONE, nullptr, &s3);
    }
    // ---- adaptive scale fusion
    Tensor ax = conv(s, fuse, asf_conv_, 1, 1, 1, ACT_NONE);
    float* gap_scr = arena.alloc_f((size_t)GAP_CHUNKS * n * 64);
    float* gap = arena.alloc_f((size_t)n * 64);
    float* gate = arena.alloc_f((size_t)n * 64);
    float* cmean = arena.alloc_f((size_t)n * fh * fw);
    Tensor fused = arena.tensor(n, fh, fw, 256);
    fused.amax = fuse.amax;  // fuse times attention weights in (0, 1) (sigmoids): bounded by fuse's max|x|
    if (!dry) {
      global_avgpool(s, ax, gap_scr, gap);
      asf_channel_gate(s, gap, asf_w1_, asf_w2_, n, asf_c_, asf_cmid_, gate);
      asf_channel_mean(s, ax, gate, cmean);
      asf_apply(s, ax, gate, cmean, asf_sp33_, asf_sp11_, asf_watt_, fuse, fused);
    }
    // ---- binarize head
    Tensor b0 = conv(s, fused, bin_conv_, 1, 1, 1, ACT_RELU);
    Tensor b1 = arena.tensor(n, 2 * fh, 2 * fw, 64);
    if (!dry) {
      ConvArgs a;
      a.act = ACT_RELU;
      a.epi = EPI_DECONV2X2;
      conv2d(s, b0, deconv1_, a, b1);
      YMK_CHECK(4 * fh == h && 4 * fw == w, "binarize output shape");
      deconv2x2_to1_sigmoid(s, b1, deconv2_w_, deconv2_b_, prob);
    }
  }

  ConvW stem_;
  std::vector<Bottleneck> layers_[4];
  ConvW in_proj_[4], out_proj_[4];
  ConvW asf_conv_, bin_conv_, deconv1_;
  float *asf_w1_ = nullptr, *asf_w2_ = nullptr, *asf_sp33_ = nullptr, *asf_watt_ = nullptr;
  float asf_sp11_ = 0.f;
  int asf_c_ = 64, asf_cmid_ = 16;
  float* deconv2_w_ = nullptr;
  float deconv2_b_ = 0.f;
  uint64_t shape_key_ = 0;
};

}  // namespace

Model* create_dbnet() { return new DBNetModel(); }

void dbnet_forward(Model* m, const float* x, int n, int h, int w, float* prob, hipStream_t s) {
  auto* d = dynamic_cast<DBNetModel*>(m);
  YMK_CHECK(d != nullptr, "model is not a dbnet");
  d->forward(x, n, h, w, prob, s);
}

}  // namespace ymk
