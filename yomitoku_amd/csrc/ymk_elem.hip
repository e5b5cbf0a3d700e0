// HBM-bound NHWC fp32 kernels around the convolutions: layout change, pooling, bilinear
// resize(+add), and the Adaptive-Scale-Fusion gates of DBNet++.  All are one-pass, 16 B per lane,
// a wave covers consecutive channels of consecutive pixels (coalesced 1 KiB per instruction).
#include "ymk_common.h"

namespace ymk {

static inline int grid_for(size_t work, int block = 256, int cap = 256 * 16) {
  size_t g = (work + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g == 0) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- NCHW(3) -> NHWC(4)
__global__ void k_nchw3_to_nhwc4(const float* __restrict__ in, float4* __restrict__ out, size_t hw, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / hw, r = i - n * hw;
    const float* b = in + n * 3 * hw + r;
    out[i] = make_float4(b[0], b[hw], b[2 * hw], 0.f);
  }
}
void nchw3_to_nhwc4(hipStream_t s, const float* in, int n, int h, int w, const Tensor& out) {
  YMK_CHECK(out.c == 4 && out.ld == 4, "nchw3_to_nhwc4 wants packed 4-channel output");
  const size_t hw = (size_t)h * w, total = hw * n;
  hipLaunchKernelGGL(k_nchw3_to_nhwc4, dim3(grid_for(total)), dim3(256), 0, s, in, (float4*)out.p, hw, total);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- maxpool 3x3 / 2 / pad 1
__global__ void k_maxpool3x3s2(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C4, int in_ld,
                               int OH, int OW, int out_ld, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t pix = i / C4;
    const int ow = (int)(pix % OW);
    pix /= OW;
    const int oh = (int)(pix % OH);
    const int n = (int)(pix / OH);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = 0; dy < 3; ++dy) {
      const int ih = oh * 2 - 1 + dy;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int iw = ow * 2 - 1 + dx;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(n * H + ih) * W + iw) * in_ld + c4 * 4);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4*>(out + ((size_t)(n * OH + oh) * OW + ow) * out_ld + c4 * 4) = m;
  }
}
void maxpool3x3s2(hipStream_t s, const Tensor& in, const Tensor& out) {
  YMK_CHECK(in.c % 4 == 0 && out.c == in.c, "maxpool: channels");
  YMK_CHECK(out.h == (in.h + 2 - 3) / 2 + 1 && out.w == (in.w + 2 - 3) / 2 + 1, "maxpool: shape");
  const size_t total = out.pixels() * (in.c / 4);
  hipLaunchKernelGGL(k_maxpool3x3s2, dim3(grid_for(total)), dim3(256), 0, s, in.p, out.p, in.h, in.w, in.c / 4, in.ld,
                     out.h, out.w, out.ld, total);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- avgpool 2x2 / 2, ceil_mode (RT-DETR "vd" shortcut)
// torch AvgPool2d(2, 2, 0, ceil_mode=True): divisor counts only in-bounds taps of a clipped window.
__global__ void k_avgpool2(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C4, int in_ld,
                           int OH, int OW, int out_ld, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t pix = i / C4;
    const int ow = (int)(pix % OW);
    pix /= OW;
    const int oh = (int)(pix % OH);
    const int n = (int)(pix / OH);
    float4 a = make_float4(0, 0, 0, 0);
    int cnt = 0;
    for (int dy = 0; dy < 2; ++dy) {
      const int ih = oh * 2 + dy;
      if (ih >= H) continue;
      for (int dx = 0; dx < 2; ++dx) {
        const int iw = ow * 2 + dx;
        if (iw >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(n * H + ih) * W + iw) * in_ld + c4 * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        ++cnt;
      }
    }
    const float d = (float)cnt;
    a.x /= d; a.y /= d; a.z /= d; a.w /= d;
    *reinterpret_cast<float4*>(out + ((size_t)(n * OH + oh) * OW + ow) * out_ld + c4 * 4) = a;
  }
}
void avgpool2x2_ceil(hipStream_t s, const Tensor& in, const Tensor& out) {
  YMK_CHECK(in.c % 4 == 0 && out.c == in.c, "avgpool: channels");
  YMK_CHECK(out.h == (in.h + 1) / 2 && out.w == (in.w + 1) / 2, "avgpool: shape");
  const size_t total = out.pixels() * (in.c / 4);
  hipLaunchKernelGGL(k_avgpool2, dim3(grid_for(total)), dim3(256), 0, s, in.p, out.p, in.h, in.w, in.c / 4, in.ld, out.h,
                     out.w, out.ld, total);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- bilinear resize, align_corners=False (+ add)
// torch: src = max((dst + 0.5) * scale - 0.5, 0), scale = in/out; i1 = min(i0 + 1, in - 1)
__global__ void k_bilinear(const float* __restrict__ in, const float* __restrict__ add, float* __restrict__ out, int H,
                           int W, int C4, int in_ld, int OH, int OW, int out_ld, int add_ld, float sh, float sw,
                           size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t pix = i / C4;
    const int ow = (int)(pix % OW);
    pix /= OW;
    const int oh = (int)(pix % OH);
    const int n = (int)(pix / OH);
    float fy = ((float)oh + 0.5f) * sh - 0.5f;
    float fx = ((float)ow + 0.5f) * sw - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* b = in + (size_t)n * H * W * in_ld + c4 * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * W + x0) * in_ld);
    const float4 v01 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * W + x1) * in_ld);
    const float4 v10 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * W + x0) * in_ld);
    const float4 v11 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * W + x1) * in_ld);
    float4 r;
    // same association as ATen's upsample_bilinear2d: h0*(w0*a + w1*b) + h1*(w0*c + w1*d)
    r.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    r.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    r.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    r.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    const size_t opix = (size_t)(n * OH + oh) * OW + ow;
    if (add) {
      const float4 t = *reinterpret_cast<const float4*>(add + opix * add_ld + c4 * 4);
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    *reinterpret_cast<float4*>(out + opix * out_ld + c4 * 4) = r;
  }
}
void upsample_bilinear(hipStream_t s, const Tensor& in, const Tensor& out, const Tensor* add) {
  YMK_CHECK(in.c % 4 == 0 && out.c == in.c && in.n == out.n, "bilinear: channels");
  YMK_CHECK(out.ld % 4 == 0 && in.ld % 4 == 0, "bilinear: ld");
  const size_t total = out.pixels() * (in.c / 4);
  const float sh = (float)in.h / (float)out.h, sw = (float)in.w / (float)out.w;
  hipLaunchKernelGGL(k_bilinear, dim3(grid_for(total)), dim3(256), 0, s, in.p, add ? add->p : nullptr, out.p, in.h,
                     in.w, in.c / 4, in.ld, out.h, out.w, out.ld, add ? add->ld : 0, sh, sw, total);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- nearest x2 (RT-DETR FPN)
__global__ void k_nearest2(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C4, int in_ld,
                           int out_ld, size_t total) {
  const int OH = 2 * H, OW = 2 * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t pix = i / C4;
    const int ow = (int)(pix % OW);
    pix /= OW;
    const int oh = (int)(pix % OH);
    const int n = (int)(pix / OH);
    const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(n * H + (oh >> 1)) * W + (ow >> 1)) * in_ld + c4 * 4);
    *reinterpret_cast<float4*>(out + ((size_t)(n * OH + oh) * OW + ow) * out_ld + c4 * 4) = v;
  }
}
void upsample_nearest2x(hipStream_t s, const Tensor& in, const Tensor& out) {
  YMK_CHECK(in.c % 4 == 0 && out.c == in.c && out.h == 2 * in.h && out.w == 2 * in.w, "nearest2x: shape");
  const size_t total = out.pixels() * (in.c / 4);
  hipLaunchKernelGGL(k_nearest2, dim3(grid_for(total)), dim3(256), 0, s, in.p, out.p, in.h, in.w, in.c / 4, in.ld, out.ld,
                     total);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- a + b (+act)
__global__ void k_add_act(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int C4,
                          int a_ld, int b_ld, int out_ld, int act, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const size_t pix = i / C4;
    float4 x = *reinterpret_cast<const float4*>(a + pix * a_ld + c4 * 4);
    const float4 y = *reinterpret_cast<const float4*>(b + pix * b_ld + c4 * 4);
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    if (act == ACT_RELU) {
      x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
    }
    *reinterpret_cast<float4*>(out + pix * out_ld + c4 * 4) = x;
  }
}
void add_act(hipStream_t s, const Tensor& a, const Tensor& b, int act, const Tensor& out) {
  YMK_CHECK(a.c % 4 == 0 && a.c == b.c && a.c == out.c, "add: channels");
  YMK_CHECK(act == ACT_NONE || act == ACT_RELU, "add: act");
  const size_t total = out.pixels() * (a.c / 4);
  hipLaunchKernelGGL(k_add_act, dim3(grid_for(total)), dim3(256), 0, s, a.p, b.p, out.p, a.c / 4, a.ld, b.ld, out.ld, act,
                     total);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- global average pool (per image, per channel)
// two-stage: block partial sums over pixel ranges -> atomics-free second pass
__global__ void k_gap_partial(const float* __restrict__ in, float* __restrict__ part, int HW, int C, int ld, int chunks) {
  // grid: (chunks, n); block 256 threads: thread handles channel (t % C) for pixel stripe (t / C)
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int per = (HW + chunks - 1) / chunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const int lanes_per_pix = C;  // C <= 256 and 256 % C == 0
  const int c = threadIdx.x % lanes_per_pix, sub = threadIdx.x / lanes_per_pix, nsub = blockDim.x / lanes_per_pix;
  float acc = 0.f;
  const float* b = in + (size_t)n * HW * ld;
  for (int p = p0 + sub; p < p1; p += nsub) acc += b[(size_t)p * ld + c];
  __shared__ float sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (sub == 0) {
    for (int k = 1; k < nsub; ++k) acc += sm[k * lanes_per_pix + c];
    part[((size_t)n * chunks + chunk) * C + c] = acc;
  }
}
__global__ void k_gap_final(const float* __restrict__ part, float* __restrict__ out, int C, int chunks, float inv) {
  const int n = blockIdx.x, c = threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
  for (int k = 0; k < chunks; ++k) acc += part[((size_t)n * chunks + k) * C + c];
  out[(size_t)n * C + c] = acc * inv;
}
void global_avgpool(hipStream_t s, const Tensor& in, float* scratch, float* out_nc) {
  YMK_CHECK(in.c <= 256 && 256 % in.c == 0, "gap: C must divide 256");
  const int HW = in.h * in.w;
  const int chunks = GAP_CHUNKS;  // scratch holds GAP_CHUNKS * n * c floats
  hipLaunchKernelGGL(k_gap_partial, dim3(chunks, in.n), dim3(256), 0, s, in.p, scratch, HW, in.c, in.ld, chunks);
  hipLaunchKernelGGL(k_gap_final, dim3(in.n), dim3(256), 0, s, scratch, out_nc, in.c, chunks, 1.f / (float)HW);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- ASF (models/layers/dbnet_feature_attention.py:69-79,150-160)
// gate[n][c] = sigmoid(W2 relu(W1 gap[n]))            (channel_wise + .sigmoid())
__global__ void k_asf_gate(const float* __restrict__ gap, const float* __restrict__ w1, const float* __restrict__ w2,
                           int C, int Cm, float* __restrict__ gate) {
  const int n = blockIdx.x, t = threadIdx.x;
  __shared__ float mid[64];
  if (t < Cm) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w1[t * C + c] * gap[n * C + c];
    mid[t] = fmaxf(a, 0.f);
  }
  __syncthreads();
  if (t < C) {
    float a = 0.f;
    for (int k = 0; k < Cm; ++k) a += w2[t * Cm + k] * mid[k];
    gate[n * C + t] = 1.f / (1.f + expf(-a));
  }
}
void asf_channel_gate(hipStream_t s, const float* gap_nc, const float* w1, const float* w2, int n, int c, int cmid,
                      float* gate_nc) {
  YMK_CHECK(c <= 256 && cmid <= 64, "asf gate dims");
  hipLaunchKernelGGL(k_asf_gate, dim3(n), dim3(256), 0, s, gap_nc, w1, w2, c, cmid, gate_nc);
  YMK_HIP(hipGetLastError());
}

// mean over channels of (x + gate): one 16-lane group per pixel (C = 64 -> 16 lanes x float4)
__global__ void k_asf_mean(const float* __restrict__ x, const float* __restrict__ gate, float* __restrict__ mean, int HW,
                           int C, int ld, size_t npix) {
  const int lanes = C / 4;  // 16
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x / lanes;
  const int l = (int)(gtid % lanes);
  for (size_t pix = gtid / lanes; pix < npix; pix += stride) {
    const int n = (int)(pix / HW);
    const float4 v = *reinterpret_cast<const float4*>(x + pix * ld + l * 4);
    const float4 g = *reinterpret_cast<const float4*>(gate + (size_t)n * C + l * 4);
    float a = (v.x + g.x) + (v.y + g.y) + (v.z + g.z) + (v.w + g.w);
    for (int o = lanes >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (l == 0) mean[pix] = a / (float)C;
  }
}
void asf_channel_mean(hipStream_t s, const Tensor& x, const float* gate_nc, float* mean_nhw) {
  YMK_CHECK(x.c == 64, "asf mean: C must be 64");
  const size_t npix = x.pixels();
  hipLaunchKernelGGL(k_asf_mean, dim3(grid_for(npix * 16)), dim3(256), 0, s, x.p, gate_nc, mean_nhw, x.h * x.w, x.c, x.ld,
                     npix);
  YMK_HIP(hipGetLastError());
}

// per pixel: sp = sigmoid(w1 * relu(conv3x3(mean))) ; score_i = sigmoid(sum_c Wa[i][c] (x+gate+sp)) ;
//            out[:, i*64 + c] = score_i * fuse[:, i*64 + c]
// 16 lanes per pixel; lane l owns channels 4l..4l+3 of x and of each of the 4 fuse groups.
__global__ void k_asf_apply(const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ mean,
                            const float* __restrict__ w33, float w11, const float* __restrict__ watt,
                            const float* __restrict__ fuse, float* __restrict__ out, int H, int W, int x_ld, int fuse_ld,
                            int out_ld, size_t npix) {
  const int C = 64, lanes = 16;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x / lanes;
  const int l = (int)(gtid % lanes);
  float k33[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k33[i] = w33[i];
  float4 wa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wa[i] = *reinterpret_cast<const float4*>(watt + i * C + l * 4);
  for (size_t pix = gtid / lanes; pix < npix; pix += stride) {
    const int hw = H * W;
    const int n = (int)(pix / hw);
    const int rem = (int)(pix - (size_t)n * hw);
    const int h = rem / W, w = rem - h * W;
    float conv = 0.f;
    const float* mb = mean + (size_t)n * hw;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = h - 1 + dy;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = w - 1 + dx;
        const float mv = ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) ? mb[y * W + xx] : 0.f;
        conv += k33[dy * 3 + dx] * mv;
      }
    }
    const float sp = 1.f / (1.f + expf(-(w11 * fmaxf(conv, 0.f))));
    const float4 v = *reinterpret_cast<const float4*>(x + pix * x_ld + l * 4);
    const float4 g = *reinterpret_cast<const float4*>(gate + (size_t)n * C + l * 4);
    const float gx = (v.x + g.x) + sp, gy = (v.y + g.y) + sp, gz = (v.z + g.z) + sp, gw = (v.w + g.w) + sp;
    float sc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = wa[i].x * gx + wa[i].y * gy + wa[i].z * gz + wa[i].w * gw;
      for (int o = lanes >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      sc[i] = 1.f / (1.f + expf(-a));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 f = *reinterpret_cast<const float4*>(fuse + pix * fuse_ld + i * C + l * 4);
      f.x *= sc[i]; f.y *= sc[i]; f.z *= sc[i]; f.w *= sc[i];
      *reinterpret_cast<float4*>(out + pix * out_ld + i * C + l * 4) = f;
    }
  }
}
void asf_apply(hipStream_t s, const Tensor& x, const float* gate_nc, const float* mean_nhw, const float* w_sp3x3,
               float w_sp1x1, const float* w_att, const Tensor& fuse, const Tensor& out) {
  YMK_CHECK(x.c == 64 && fuse.c == 256 && out.c == 256, "asf apply: channel counts");
  const size_t npix = x.pixels();
  hipLaunchKernelGGL(k_asf_apply, dim3(grid_for(npix * 16)), dim3(256), 0, s, x.p, gate_nc, mean_nhw, w_sp3x3, w_sp1x1,
                     w_att, fuse.p, out.p, x.h, x.w, x.ld, fuse.ld, out.ld, npix);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- final ConvTranspose2d(C->1,2,2)+bias+sigmoid
// 16 lanes per input pixel (C=64); each input pixel emits a 2x2 output patch.
__global__ void k_deconv_to1(const float* __restrict__ in, const float* __restrict__ w, float bias, float* __restrict__ out,
                             int H, int W, int ld, size_t npix) {
  const int lanes = 16;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x / lanes;
  const int l = (int)(gtid % lanes);
  // w layout [c][4] (ab fastest): lane owns c = 4l..4l+3
  float4 wc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wc[j] = *reinterpret_cast<const float4*>(w + (l * 4 + j) * 4);
  for (size_t pix = gtid / lanes; pix < npix; pix += stride) {
    const float4 v = *reinterpret_cast<const float4*>(in + pix * ld + l * 4);
    float a0 = v.x * wc[0].x + v.y * wc[1].x + v.z * wc[2].x + v.w * wc[3].x;
    float a1 = v.x * wc[0].y + v.y * wc[1].y + v.z * wc[2].y + v.w * wc[3].y;
    float a2 = v.x * wc[0].z + v.y * wc[1].z + v.z * wc[2].z + v.w * wc[3].z;
    float a3 = v.x * wc[0].w + v.y * wc[1].w + v.z * wc[2].w + v.w * wc[3].w;
    for (int o = lanes >> 1; o > 0; o >>= 1) {
      a0 += __shfl_xor(a0, o, 64);
      a1 += __shfl_xor(a1, o, 64);
      a2 += __shfl_xor(a2, o, 64);
      a3 += __shfl_xor(a3, o, 64);
    }
    if (l == 0) {
      const int hw = H * W;
      const size_t n = pix / hw;
      const int rem = (int)(pix - n * hw);
      const int h = rem / W, ww = rem - h * W;
      float* o = out + n * (size_t)(4 * hw) + (size_t)(2 * h) * (2 * W) + 2 * ww;
      o[0] = 1.f / (1.f + expf(-(a0 + bias)));
      o[1] = 1.f / (1.f + expf(-(a1 + bias)));
      o[2 * W] = 1.f / (1.f + expf(-(a2 + bias)));
      o[2 * W + 1] = 1.f / (1.f + expf(-(a3 + bias)));
    }
  }
}
void deconv2x2_to1_sigmoid(hipStream_t s, const Tensor& in, const float* w_c4, float bias, float* out) {
  YMK_CHECK(in.c == 64, "deconv_to1: C must be 64");
  const size_t npix = in.pixels();
  hipLaunchKernelGGL(k_deconv_to1, dim3(grid_for(npix * 16)), dim3(256), 0, s, in.p, w_c4, bias, out, in.h, in.w, in.ld,
                     npix);
  YMK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- max|x| records (ymk_common.h)
__global__ void k_amax_merge(unsigned* __restrict__ dst, const unsigned* __restrict__ src) {
  const int i = threadIdx.x * AMAX_LINE_WORDS;
  dst[i] = max(dst[i], src[i]);
}
void amax_merge(hipStream_t s, unsigned* dst, const unsigned* src) {
  if (dst == nullptr || src == nullptr) return;
  hipLaunchKernelGGL(k_amax_merge, dim3(1), dim3(AMAX_LINES), 0, s, dst, src);
  YMK_HIP(hipGetLastError());
}

}  // namespace ymk
