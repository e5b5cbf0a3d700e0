// Page / crop pre-processing fused into coalesced HBM kernels (uint8 pages in, network tensors out).
//
//   k_det_preprocess      TextDetector.preprocess (text_detector.py:99-107): BGR uint8 page ->
//                         cv2.resize(INTER_AREA) on float32 -> /255 -> (x - mean) / std -> NCHW fp32
//   k_pil_resize_to_chw   LayoutParser / TableStructureRecognizer.preprocess (layout_parser.py:195-199,
//                         table_structure_recognizer.py:169-186): crop -> PIL bilinear (antialiased,
//                         8-bit two-pass, 22-bit coefficients) resize to 640x640 -> ToTensor
//   k_warp_quads / k_crop_resize_norm   ParseqDataset (data/dataset.py:105-124, data/functions.py:301-439):
//                         perspective warp of each text quad (cv2 fixed-point bilinear), optional 90 deg
//                         rotation, down-scale-only INTER_AREA to 32 px height, paste on a black canvas,
//                         ToTensor + Normalize(0.5, 0.5), right padding with -1 to the batch width
#include "ymk_common.h"

// The resamplers round to uint8 (or feed thresholds) after short float sums; cv2 / NumPy evaluate them
// with separate multiplies and adds, so fused multiply-add contraction is switched off in this file.
#pragma clang fp contract(off)

namespace ymk {

// ------------------------------------------------------------------ cv2 INTER_AREA tap generator
// computeResizeAreaTab for one destination index: up to MAXT (src index, float weight) taps.
constexpr int MAXT = 24;
struct Taps {
  int n;
  int idx[MAXT];
  float wt[MAXT];
};
__device__ __forceinline__ void area_taps(int d, double scale, int ssize, Taps& t) {
  const double fsx1 = d * scale, fsx2 = fsx1 + scale;
  const double cell = fmin(scale, ssize - fsx1);
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  sx2 = min(sx2, ssize - 1);
  sx1 = min(sx1, sx2);
  t.n = 0;
  if (sx1 - fsx1 > 1e-3) {
    t.idx[t.n] = sx1 - 1;
    t.wt[t.n++] = (float)((sx1 - fsx1) / cell);
  }
  for (int sx = sx1; sx < sx2 && t.n < MAXT - 1; ++sx) {
    t.idx[t.n] = sx;
    t.wt[t.n++] = (float)(1.0 / cell);
  }
  if (fsx2 - sx2 > 1e-3) {
    t.idx[t.n] = sx2;
    t.wt[t.n++] = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell);
  }
}
// The same taps without the MAXT cap, for the recogniser crops (a tall text line or a whole page handed to
// TextRecognizer is scaled down by far more than 22): tap i of n = left partial cell, the full cells, right partial
// cell, in computeResizeAreaTab's order, so sums accumulate exactly as with the array form.
struct AreaSpan {
  int sx1, sx2, n;
  int has_l, has_r;
  float wl, wf, wr;
  __device__ __forceinline__ void tap(int i, int& idx, float& wt) const {
    if (i < has_l) {
      idx = sx1 - 1;
      wt = wl;
    } else if (i - has_l < sx2 - sx1) {
      idx = sx1 + (i - has_l);
      wt = wf;
    } else {
      idx = sx2;
      wt = wr;
    }
  }
};
__device__ __forceinline__ AreaSpan area_span(int d, double scale, int ssize) {
  const double fsx1 = d * scale, fsx2 = fsx1 + scale;
  const double cell = fmin(scale, ssize - fsx1);
  AreaSpan a;
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  sx2 = min(sx2, ssize - 1);
  sx1 = min(sx1, sx2);
  a.sx1 = sx1;
  a.sx2 = sx2;
  a.has_l = sx1 - fsx1 > 1e-3 ? 1 : 0;
  a.has_r = fsx2 - sx2 > 1e-3 ? 1 : 0;
  a.wl = (float)((sx1 - fsx1) / cell);
  a.wf = (float)(1.0 / cell);
  a.wr = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell);
  a.n = a.has_l + (sx2 - sx1) + a.has_r;
  return a;
}
// INTER_AREA used for enlarging = bilinear taps with the area coefficient rule
__device__ __forceinline__ void linear_area_tap(int d, double scale, int ssize, int& s0, float& f) {
  const double inv = 1.0 / scale;
  int sx = (int)floor(d * scale);
  float fx = (float)((d + 1) - (sx + 1) * inv);
  fx = fx <= 0.f ? 0.f : fx - floorf(fx);
  if (sx < 0) {
    fx = 0.f;
    sx = 0;
  }
  if (sx >= ssize - 1) {
    fx = 0.f;
    sx = ssize - 1;
  }
  s0 = sx;
  f = fx;
}

struct DetPre {
  const unsigned char* src;  // [h][w][3] BGR
  float* dst;                // [3][oh][ow]
  int h, w, oh, ow;
  double sx, sy;             // src / dst
  double mean[3], stdv[3];
};

__global__ void k_det_preprocess(DetPre p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= p.ow) return;
  float v[3];
  if (p.sx >= 1.0 && p.sy >= 1.0) {
    Taps tx, ty;
    area_taps(x, p.sx, p.w, tx);
    area_taps(y, p.sy, p.h, ty);
    float sum[3] = {0.f, 0.f, 0.f};
    for (int r = 0; r < ty.n; ++r) {
      const unsigned char* row = p.src + (size_t)ty.idx[r] * p.w * 3;
      float buf[3] = {0.f, 0.f, 0.f};
      for (int c = 0; c < tx.n; ++c) {
        const unsigned char* px = row + (size_t)tx.idx[c] * 3;
        buf[0] = buf[0] + (float)px[0] * tx.wt[c];
        buf[1] = buf[1] + (float)px[1] * tx.wt[c];
        buf[2] = buf[2] + (float)px[2] * tx.wt[c];
      }
      if (r == 0) {
        sum[0] = ty.wt[r] * buf[0]; sum[1] = ty.wt[r] * buf[1]; sum[2] = ty.wt[r] * buf[2];
      } else {
        sum[0] = sum[0] + ty.wt[r] * buf[0]; sum[1] = sum[1] + ty.wt[r] * buf[1]; sum[2] = sum[2] + ty.wt[r] * buf[2];
      }
    }
    v[0] = sum[0]; v[1] = sum[1]; v[2] = sum[2];
  } else {
    int x0, y0;
    float fx, fy;
    linear_area_tap(x, p.sx, p.w, x0, fx);
    linear_area_tap(y, p.sy, p.h, y0, fy);
    const int x1 = min(x0 + 1, p.w - 1), y1 = min(y0 + 1, p.h - 1);
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const unsigned char* r0 = p.src + (size_t)y0 * p.w * 3;
    const unsigned char* r1 = p.src + (size_t)y1 * p.w * 3;
    for (int c = 0; c < 3; ++c) {
      const float h0 = (float)r0[x0 * 3 + c] * a0 + (float)r0[x1 * 3 + c] * a1;
      const float h1 = (float)r1[x0 * 3 + c] * a0 + (float)r1[x1 * 3 + c] * a1;
      v[c] = h0 * b0 + h1 * b1;
    }
  }
  // standardization_image (data/functions.py:230-247): /255 in fp32, (x - mean) / std in fp64, back to fp32.
  // The two channel flips of the reference cancel: tensor channel c is BGR channel c with mean/std[c].
  const size_t plane = (size_t)p.oh * p.ow;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t = v[c] / 255.0f;
    p.dst[c * plane + (size_t)y * p.ow + x] = (float)(((double)t - p.mean[c]) / p.stdv[c]);
  }
}

void det_preprocess(hipStream_t s, const unsigned char* bgr, int h, int w, int oh, int ow, float* out) {
  DetPre p;
  p.src = bgr;
  p.dst = out;
  p.h = h;
  p.w = w;
  p.oh = oh;
  p.ow = ow;
  p.sx = (double)w / ow;
  p.sy = (double)h / oh;
  YMK_CHECK(p.sx < MAXT - 3 && p.sy < MAXT - 3, "detector preprocess: down-scale factor too large");
  const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
  for (int c = 0; c < 3; ++c) {
    p.mean[c] = mean[c];
    p.stdv[c] = stdv[c];
  }
  hipLaunchKernelGGL(k_det_preprocess, dim3((ow + 255) / 256, oh), dim3(256), 0, s, p);
  YMK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ PIL bilinear (antialiased) resize -> CHW fp32 in [0,1]
// Coefficients (ImagingResample precompute_coeffs + normalize_coeffs_8bpc) are built on the host in
// double and passed as int tables; the kernel does the two 8-bit passes (horizontal, then vertical,
// each rounded to uint8 through >> 22 with the 1 << 21 bias) for one output pixel.
struct PilRes {
  const unsigned char* src;  // page [H][W][3] BGR
  int W;                     // page row length in pixels
  int x0, y0;                // crop origin inside the page
  const int* xb;             // [ow][2] (first tap, count) relative to the crop
  const int* xk;             // [ow][ksx]
  const int* yb;
  const int* yk;
  int ksx, ksy, oh, ow;
  float* dst;                // [3][oh][ow], RGB order (cv2.cvtColor BGR2RGB first)
};
__device__ __forceinline__ int clip8(int v) {
  v >>= 22;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}
__global__ void k_pil_resize_to_chw(PilRes p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= p.ow) return;
  const int xmin = p.xb[2 * x], xn = p.xb[2 * x + 1];
  const int ymin = p.yb[2 * y], yn = p.yb[2 * y + 1];
  const int* kx = p.xk + (size_t)x * p.ksx;
  const int* ky = p.yk + (size_t)y * p.ksy;
  int acc[3] = {1 << 21, 1 << 21, 1 << 21};
  for (int r = 0; r < yn; ++r) {
    const unsigned char* row = p.src + ((size_t)(p.y0 + ymin + r) * p.W + p.x0 + xmin) * 3;
    int h0 = 1 << 21, h1 = 1 << 21, h2 = 1 << 21;
    for (int c = 0; c < xn; ++c) {
      const int k = kx[c];
      h0 += row[c * 3 + 2] * k;  // R
      h1 += row[c * 3 + 1] * k;  // G
      h2 += row[c * 3 + 0] * k;  // B
    }
    acc[0] += clip8(h0) * ky[r];
    acc[1] += clip8(h1) * ky[r];
    acc[2] += clip8(h2) * ky[r];
  }
  const size_t plane = (size_t)p.oh * p.ow;
#pragma unroll
  for (int c = 0; c < 3; ++c) p.dst[c * plane + (size_t)y * p.ow + x] = (float)clip8(acc[c]) / 255.0f;  // ToTensor
}
// The same for n crops in ONE launch (a wave's table crops: ~170, each with its own size and coefficient tables).  `blob`: int32
// words on the device - n records of PIL_BATCH_REC words {page address low, high; page width; x0; y0; ksx; ksy; and the word
// offsets into the blob of xbounds, xcoefs, ybounds, ycoefs}, then the tables themselves.  Crop z writes out[z][3][oh][ow].
constexpr int PIL_BATCH_REC = 12;
__global__ void k_pil_resize_batch(const int* __restrict__ blob, int oh, int ow, float* __restrict__ out) {
  const int* rec = blob + (size_t)blockIdx.z * PIL_BATCH_REC;
  PilRes p;
  p.src = reinterpret_cast<const unsigned char*>(((unsigned long long)(unsigned)rec[1] << 32) | (unsigned long long)(unsigned)rec[0]);
  p.W = rec[2];
  p.x0 = rec[3];
  p.y0 = rec[4];
  p.ksx = rec[5];
  p.ksy = rec[6];
  p.xb = blob + rec[7];
  p.xk = blob + rec[8];
  p.yb = blob + rec[9];
  p.yk = blob + rec[10];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  const int xmin = p.xb[2 * x], xn = p.xb[2 * x + 1];
  const int ymin = p.yb[2 * y], yn = p.yb[2 * y + 1];
  const int* kx = p.xk + (size_t)x * p.ksx;
  const int* ky = p.yk + (size_t)y * p.ksy;
  int acc[3] = {1 << 21, 1 << 21, 1 << 21};
  for (int r = 0; r < yn; ++r) {
    const unsigned char* row = p.src + ((size_t)(p.y0 + ymin + r) * p.W + p.x0 + xmin) * 3;
    int h0 = 1 << 21, h1 = 1 << 21, h2 = 1 << 21;
    for (int c = 0; c < xn; ++c) {
      const int k = kx[c];
      h0 += row[c * 3 + 2] * k;  // R
      h1 += row[c * 3 + 1] * k;  // G
      h2 += row[c * 3 + 0] * k;  // B
    }
    acc[0] += clip8(h0) * ky[r];
    acc[1] += clip8(h1) * ky[r];
    acc[2] += clip8(h2) * ky[r];
  }
  const size_t plane = (size_t)oh * ow;
  float* dst = out + (size_t)blockIdx.z * 3 * plane;
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[c * plane + (size_t)y * ow + x] = (float)clip8(acc[c]) / 255.0f;  // ToTensor
}
void pil_resize_batch_to_chw(hipStream_t s, const int* blob, int n, int oh, int ow, float* out) {
  if (n <= 0) return;
  YMK_CHECK(n <= 65535, "pil_resize_batch: at most 65535 crops per launch");
  hipLaunchKernelGGL(k_pil_resize_batch, dim3((ow + 255) / 256, oh, n), dim3(256), 0, s, blob, oh, ow, out);
  YMK_HIP(hipGetLastError());
}

void pil_resize_to_chw(hipStream_t s, const unsigned char* page, int W, int x0, int y0, const int* xb, const int* xk, int ksx,
                       const int* yb, const int* yk, int ksy, int oh, int ow, float* out) {
  PilRes p{page, W, x0, y0, xb, xk, yb, yk, ksx, ksy, oh, ow, out};
  hipLaunchKernelGGL(k_pil_resize_to_chw, dim3((ow + 255) / 256, oh), dim3(256), 0, s, p);
  YMK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ text-line crops for the recogniser
// One descriptor per quad, prepared on the host (integer geometry only, data/functions.py:301-376):
struct CropDesc {
  double minv[9];     // inverse of the perspective matrix (dst -> src), src relative to the bbox crop
  int bx, by, bw, bh; // bbox crop inside the page (roi_img)
  int ww, wh;         // warp output size (width, height) BEFORE the optional rotation
  int rot;            // 1: rotate 90 deg counter-clockwise afterwards (h > 2 w)
  int rw, rh;         // size after rotation
  int nw, nh;         // size after the down-scale-only INTER_AREA resize
  int fast_x, fast_y; // both non-zero: exact integer factors -> cv2's ResizeAreaFast path
  long long warp_off; // offset (bytes) of this crop's warped pixels in the scratch buffer ([wh][ww][3] RGB)
  int slot;           // row of the batch tensor
  int flip;           // 1: cv2.rotate(ROTATE_180) of the (rotated) crop before the resize - orientation fallback retry
  int level;          // pyramid level the crop is cut from (source_downscale, data/dataset.py:26-41,64-79); 0 = the page
};

constexpr int MAX_LEVELS = 4;  // _calc_source_levels clips to max_level = 3
struct PageLevels {
  const unsigned char* p[MAX_LEVELS];
  int H[MAX_LEVELS], W[MAX_LEVELS];
};

// cv2.warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0) on uint8: source coordinates quantised to 1/32 px,
// bilinear weights as 15-bit integers summing to 32768.
__global__ void k_warp_quads(PageLevels lv, const CropDesc* __restrict__ descs, unsigned char* __restrict__ scratch) {
  const CropDesc& d = descs[blockIdx.z];
  const unsigned char* __restrict__ page = lv.p[d.level];
  const int W = lv.W[d.level];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= d.ww || y >= d.wh) return;
  const double X0 = d.minv[0] * x + d.minv[1] * y + d.minv[2];
  const double Y0 = d.minv[3] * x + d.minv[4] * y + d.minv[5];
  double Wd = d.minv[6] * x + d.minv[7] * y + d.minv[8];
  Wd = Wd != 0.0 ? 32.0 / Wd : 0.0;
  const double fX = fmax(-2147483648.0, fmin(2147483647.0, X0 * Wd));
  const double fY = fmax(-2147483648.0, fmin(2147483647.0, Y0 * Wd));
  const long long Xi = (long long)rint(fX), Yi = (long long)rint(fY);
  const int sx = (int)(Xi >> 5), sy = (int)(Yi >> 5);
  const float ax = (float)(Xi & 31) / 32.f, ay = (float)(Yi & 31) / 32.f;
  const int w00 = (int)rintf((1.f - ax) * (1.f - ay) * 32768.f), w01 = (int)rintf(ax * (1.f - ay) * 32768.f);
  const int w10 = (int)rintf((1.f - ax) * ay * 32768.f), w11 = 32768 - w00 - w01 - w10;
  int acc[3] = {0, 0, 0};
  auto tap = [&](int yy, int xx, int wt) {
    if ((unsigned)xx < (unsigned)d.bw && (unsigned)yy < (unsigned)d.bh) {
      const unsigned char* px = page + ((size_t)(d.by + yy) * W + d.bx + xx) * 3;
      acc[0] += px[2] * wt;  // ParseqDataset crops from the RGB view of the page (dataset.py:68)
      acc[1] += px[1] * wt;
      acc[2] += px[0] * wt;
    }
  };
  tap(sy, sx, w00);
  tap(sy, sx + 1, w01);
  tap(sy + 1, sx, w10);
  tap(sy + 1, sx + 1, w11);
  unsigned char* o = scratch + d.warp_off + ((size_t)y * d.ww + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int v = (acc[c] + (1 << 14)) >> 15;
    o[c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

// rotate (optional) + INTER_AREA down-scale on uint8 (rounded back to uint8) + canvas + normalise.
// Writes the full 32 x batch_w row block of the crop's slot: content, then black canvas (-1), then -1 padding.
__global__ void k_crop_resize_norm(const CropDesc* __restrict__ descs, const unsigned char* __restrict__ scratch,
                                   float* __restrict__ out, int batch_w, int out_h) {
  const CropDesc& d = descs[blockIdx.z];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= batch_w) return;
  float v[3] = {-1.f, -1.f, -1.f};
  if (x < d.nw && y < d.nh) {
    const unsigned char* img = scratch + d.warp_off;
    // pixel (yy, xx) of the (possibly rotated) crop; ROTATE_90_COUNTERCLOCKWISE: dst(i, j) = src(j, W - 1 - i)
    auto px = [&](int yy, int xx, int c) -> float {
      if (d.flip) {  // ROTATE_180: dst(i, j) = src(rh - 1 - i, rw - 1 - j)
        yy = d.rh - 1 - yy;
        xx = d.rw - 1 - xx;
      }
      if (d.rot) return (float)img[((size_t)xx * d.ww + (d.ww - 1 - yy)) * 3 + c];
      return (float)img[((size_t)yy * d.ww + xx) * 3 + c];
    };
    if (d.nw == d.rw && d.nh == d.rh) {
      for (int c = 0; c < 3; ++c) v[c] = px(y, x, c);
    } else if (d.fast_x > 0 && d.fast_y > 0) {
      // ResizeAreaFast: integer block sums; 2x2 rounds as (s + 2) >> 2, other factors as rint(s * (1.f / area))
      for (int c = 0; c < 3; ++c) {
        int sacc = 0;
        for (int dy = 0; dy < d.fast_y; ++dy)
          for (int dx = 0; dx < d.fast_x; ++dx) sacc += (int)px(y * d.fast_y + dy, x * d.fast_x + dx, c);
        if (d.fast_x == 2 && d.fast_y == 2) v[c] = (float)((sacc + 2) >> 2);
        else v[c] = fminf(fmaxf(rintf((float)sacc * (1.f / (float)(d.fast_x * d.fast_y))), 0.f), 255.f);
      }
    } else {
      const AreaSpan tx = area_span(x, (double)d.rw / d.nw, d.rw);
      const AreaSpan ty = area_span(y, (double)d.rh / d.nh, d.rh);
      float sum[3] = {0.f, 0.f, 0.f};
      for (int r = 0; r < ty.n; ++r) {
        int yi, xi;
        float wy, wx;
        ty.tap(r, yi, wy);
        float buf[3] = {0.f, 0.f, 0.f};
        for (int c2 = 0; c2 < tx.n; ++c2) {
          tx.tap(c2, xi, wx);
          for (int c = 0; c < 3; ++c) buf[c] = buf[c] + px(yi, xi, c) * wx;
        }
        for (int c = 0; c < 3; ++c) sum[c] = r == 0 ? wy * buf[c] : sum[c] + wy * buf[c];
      }
      for (int c = 0; c < 3; ++c) v[c] = fminf(fmaxf(rintf(sum[c]), 0.f), 255.f);  // saturate_cast<uchar>
    }
    // ToTensor (/255) then Normalize(0.5, 0.5)
    for (int c = 0; c < 3; ++c) v[c] = (v[c] / 255.0f - 0.5f) / 0.5f;
  }
  const size_t plane = (size_t)out_h * batch_w;
  float* o = out + (size_t)d.slot * 3 * plane + (size_t)y * batch_w + x;
  o[0] = v[0];
  o[plane] = v[1];
  o[2 * plane] = v[2];
}

// cv2.resize(img, None, fx=0.5, fy=0.5, interpolation=INTER_AREA) on uint8 HxWx3 (data/dataset.py:73-79): the scale is
// exactly 2, i.e. ResizeAreaFast - full 2x2 cells round as (s + 2) >> 2; a cell cut by the right / bottom edge (odd
// sizes whose half rounds up) averages the pixels it has, rint(sum / count); a cell starting past the edge is 0.
__global__ void k_halve_u8c3(const unsigned char* __restrict__ src, int H, int W, unsigned char* __restrict__ dst, int dh, int dw) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dw) return;
  const int sx0 = 2 * x, sy0 = 2 * y;
  unsigned char* o = dst + ((size_t)y * dw + x) * 3;
  if (sx0 >= W || sy0 >= H) {
    o[0] = o[1] = o[2] = 0;
    return;
  }
  const int nx = sx0 + 2 <= W ? 2 : 1, ny = sy0 + 2 <= H ? 2 : 1;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int sum = 0;
    for (int dy = 0; dy < ny; ++dy)
      for (int dx = 0; dx < nx; ++dx) sum += src[((size_t)(sy0 + dy) * W + sx0 + dx) * 3 + c];
    if (nx == 2 && ny == 2) o[c] = (unsigned char)((sum + 2) >> 2);
    else o[c] = (unsigned char)fminf(fmaxf(rintf((float)sum / (float)(nx * ny)), 0.f), 255.f);
  }
}
void halve_u8c3(hipStream_t s, const unsigned char* src, int H, int W, unsigned char* dst, int dh, int dw) {
  hipLaunchKernelGGL(k_halve_u8c3, dim3((dw + 255) / 256, dh), dim3(256), 0, s, src, H, W, dst, dh, dw);
  YMK_HIP(hipGetLastError());
}

void crop_batch(hipStream_t s, const PageLevels& lv, const CropDesc* descs_dev, int n, int max_ww, int max_wh,
                unsigned char* scratch, float* out, int batch_w, int out_h) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_warp_quads, dim3((max_ww + 63) / 64, max_wh, n), dim3(64), 0, s, lv, descs_dev, scratch);
  hipLaunchKernelGGL(k_crop_resize_norm, dim3((batch_w + 63) / 64, out_h, n), dim3(64), 0, s, descs_dev, scratch, out,
                     batch_w, out_h);
  YMK_HIP(hipGetLastError());
}

}  // namespace ymk

// ------------------------------------------------------------------ C ABI
#include "../../include/ymk.h"
namespace ymk {
void det_preprocess(hipStream_t s, const unsigned char* bgr, int h, int w, int oh, int ow, float* out);
}
extern "C" {
int ymk_det_preprocess(const unsigned char* bgr_dev, int h, int w, int oh, int ow, float* x_dev, void* stream) {
  try {
    YMK_CHECK(bgr_dev && x_dev && h > 0 && w > 0 && oh > 0 && ow > 0, "bad argument");
    ymk::det_preprocess((hipStream_t)stream, bgr_dev, h, w, oh, ow, x_dev);
    return 0;
  } catch (const std::exception& e) {
    ymk::set_error(e.what());
    return 1;
  }
}
int ymk_pil_resize_to_chw(const unsigned char* page_dev, int page_w, int x0, int y0, const int* xbounds_dev,
                          const int* xcoef_dev, int ksize_x, const int* ybounds_dev, const int* ycoef_dev, int ksize_y,
                          int oh, int ow, float* x_dev, void* stream) {
  try {
    YMK_CHECK(page_dev && xbounds_dev && xcoef_dev && ybounds_dev && ycoef_dev && x_dev, "null argument");
    ymk::pil_resize_to_chw((hipStream_t)stream, page_dev, page_w, x0, y0, xbounds_dev, xcoef_dev, ksize_x, ybounds_dev,
                           ycoef_dev, ksize_y, oh, ow, x_dev);
    return 0;
  } catch (const std::exception& e) {
    ymk::set_error(e.what());
    return 1;
  }
}
int ymk_pil_resize_batch_to_chw(const int* blob_dev, int n, int oh, int ow, float* x_dev, void* stream) {
  try {
    YMK_CHECK(blob_dev && x_dev && n >= 0 && oh > 0 && ow > 0, "bad argument");
    ymk::pil_resize_batch_to_chw((hipStream_t)stream, blob_dev, n, oh, ow, x_dev);
    return 0;
  } catch (const std::exception& e) {
    ymk::set_error(e.what());
    return 1;
  }
}
int ymk_pil_batch_record_words(void) { return ymk::PIL_BATCH_REC; }
int ymk_crop_batch(const unsigned char* page_dev, int page_h, int page_w, const void* descs_dev, int n, int max_warp_w,
                   int max_warp_h, unsigned char* scratch_dev, float* out_dev, int batch_w, int out_h, void* stream) {
  try {
    YMK_CHECK(page_dev && descs_dev && scratch_dev && out_dev, "null argument");
    ymk::PageLevels lv{};
    for (int i = 0; i < ymk::MAX_LEVELS; ++i) {  // a descriptor naming a level > 0 reads the page: callers with a pyramid use _levels
      lv.p[i] = page_dev;
      lv.H[i] = page_h;
      lv.W[i] = page_w;
    }
    ymk::crop_batch((hipStream_t)stream, lv, (const ymk::CropDesc*)descs_dev, n, max_warp_w, max_warp_h, scratch_dev, out_dev,
                    batch_w, out_h);
    return 0;
  } catch (const std::exception& e) {
    ymk::set_error(e.what());
    return 1;
  }
}
int ymk_crop_batch_levels(const unsigned char* const* level_pages_dev, const int* level_h, const int* level_w, int n_levels,
                          const void* descs_dev, int n, int max_warp_w, int max_warp_h, unsigned char* scratch_dev,
                          float* out_dev, int batch_w, int out_h, void* stream) {
  try {
    YMK_CHECK(level_pages_dev && level_h && level_w && descs_dev && scratch_dev && out_dev, "null argument");
    YMK_CHECK(n_levels >= 1 && n_levels <= ymk::MAX_LEVELS, "1..4 pyramid levels");
    ymk::PageLevels lv{};
    for (int i = 0; i < ymk::MAX_LEVELS; ++i) {
      const int k = i < n_levels && level_pages_dev[i] ? i : 0;  // unused levels alias the page
      lv.p[i] = level_pages_dev[k];
      lv.H[i] = level_h[k];
      lv.W[i] = level_w[k];
    }
    YMK_CHECK(lv.p[0] != nullptr, "level 0 (the page) is required");
    ymk::crop_batch((hipStream_t)stream, lv, (const ymk::CropDesc*)descs_dev, n, max_warp_w, max_warp_h, scratch_dev, out_dev,
                    batch_w, out_h);
    return 0;
  } catch (const std::exception& e) {
    ymk::set_error(e.what());
    return 1;
  }
}
int ymk_halve_u8c3(const unsigned char* src_dev, int h, int w, unsigned char* dst_dev, int dst_h, int dst_w, void* stream) {
  try {
    YMK_CHECK(src_dev && dst_dev && h > 0 && w > 0 && dst_h > 0 && dst_w > 0, "bad argument");
    ymk::halve_u8c3((hipStream_t)stream, src_dev, h, w, dst_dev, dst_h, dst_w);
    return 0;
  } catch (const std::exception& e) {
    ymk::set_error(e.what());
    return 1;
  }
}
int ymk_crop_desc_size(void) { return (int)sizeof(ymk::CropDesc); }
}
