// Host geometry of the table cell detector's post-processing: the parts of a table crop that no detected cell covers,
// as rectangles (find_holes_as_rects, table_cell_detector.py:116-143).  The reference composes OpenCV calls -
// rectangle(thickness=-1), morphologyEx(MORPH_OPEN, 5x5 rect, iterations=3), floodFill from the corner,
// findContours(RETR_EXTERNAL), boundingRect - restated here on a byte mask (OpenCV is not a dependency):
//   - filled rectangles cover the inclusive pixel range [x1, x2] x [y1, y2], clipped to the mask;
//   - OPEN with `it` iterations of a k x k box = erosion then dilation by one (it*(k-1)+1)-wide box; OpenCV's default
//     morphology border never wins a min / max, i.e. windows are simply clipped to the image;
//   - floodFill(seed (0,0), newVal 0, 4-connected, zero tolerance) clears the white region connected to the corner
//     (nothing changes when the corner pixel is black);
//   - RETR_EXTERNAL keeps the outer border of every 8-connected white component that is not enclosed by another
//     component; boundingRect is (min x, min y, extent + 1).
// Order: contours come back newest first (a raster scan meets components top-to-bottom, left-to-right; the list is
// returned in reverse), as in ymk_dbpost.cpp - unpinned against OpenCV itself, like the rest of that file.
#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../include/ymk.h"
#include "ymk_common.h"

namespace ymk {

namespace {

// out(y, x) = min (erode) or max (dilate) of in over the k x k window centred on (y, x), clipped to the image.  The mask
// only holds 0 and 255, so a window's min / max follows from how many of its pixels are white: separable, one
// prefix-count pass per axis.
void box_morph(std::vector<uint8_t>& img, int h, int w, int k, bool erode) {
  const int r = k / 2;
  std::vector<int> pre((size_t)std::max(h, w) + 1);
  std::vector<uint8_t> tmp((size_t)h * w);
  auto decide = [erode](int white, int total) -> uint8_t { return erode ? (white == total ? 255 : 0) : (white > 0 ? 255 : 0); };
  for (int y = 0; y < h; ++y) {
    pre[0] = 0;
    for (int x = 0; x < w; ++x) pre[x + 1] = pre[x] + (img[(size_t)y * w + x] != 0);
    for (int x = 0; x < w; ++x) {
      const int a = std::max(0, x - r), b = std::min(w - 1, x + r);
      tmp[(size_t)y * w + x] = decide(pre[b + 1] - pre[a], b - a + 1);
    }
  }
  for (int x = 0; x < w; ++x) {
    pre[0] = 0;
    for (int y = 0; y < h; ++y) pre[y + 1] = pre[y] + (tmp[(size_t)y * w + x] != 0);
    for (int y = 0; y < h; ++y) {
      const int a = std::max(0, y - r), b = std::min(h - 1, y + r);
      img[(size_t)y * w + x] = decide(pre[b + 1] - pre[a], b - a + 1);
    }
  }
}

}  // namespace

void table_hole_rects(int h, int w, const int* boxes, int n, int pad, int close_ksize, int min_area, std::vector<int>& rects) {
  rects.clear();
  if (h <= 0 || w <= 0) return;
  std::vector<uint8_t> mask((size_t)h * w, 255);
  for (int i = 0; i < n; ++i) {
    int x1 = boxes[4 * i], y1 = boxes[4 * i + 1], x2 = boxes[4 * i + 2], y2 = boxes[4 * i + 3];
    if (x1 > x2) std::swap(x1, x2);
    if (y1 > y2) std::swap(y1, y2);
    x1 = std::max(x1, 0);
    y1 = std::max(y1, 0);
    x2 = std::min(x2, w - 1);
    y2 = std::min(y2, h - 1);
    for (int y = y1; y <= y2; ++y)
      for (int x = x1; x <= x2; ++x) mask[(size_t)y * w + x] = 0;
  }
  if (close_ksize > 1) {
    const int k = 3 * (close_ksize - 1) + 1;  // iterations = 3
    box_morph(mask, h, w, k, /*erode=*/true);
    box_morph(mask, h, w, k, /*erode=*/false);
  }
  std::vector<int> stack;
  if (mask[0] != 0) {  // floodFill from (0, 0), 4-connected, exact value
    const uint8_t seed = mask[0];
    stack.push_back(0);
    mask[0] = 0;
    while (!stack.empty()) {
      const int p = stack.back();
      stack.pop_back();
      const int y = p / w, x = p - y * w;
      const int nb[4][2] = {{y - 1, x}, {y + 1, x}, {y, x - 1}, {y, x + 1}};
      for (auto& q : nb)
        if (q[0] >= 0 && q[0] < h && q[1] >= 0 && q[1] < w && mask[(size_t)q[0] * w + q[1]] == seed) {
          mask[(size_t)q[0] * w + q[1]] = 0;
          stack.push_back(q[0] * w + q[1]);
        }
    }
  }
  // 8-connected white components in raster order of their first pixel
  std::vector<int> label((size_t)h * w, 0);
  struct Comp {
    int x0, y0, x1, y1;
  };
  std::vector<Comp> comps;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      if (mask[(size_t)y * w + x] == 0 || label[(size_t)y * w + x] != 0) continue;
      const int id = (int)comps.size() + 1;
      Comp c{x, y, x, y};
      label[(size_t)y * w + x] = id;
      stack.push_back(y * w + x);
      while (!stack.empty()) {
        const int p = stack.back();
        stack.pop_back();
        const int py = p / w, px = p - py * w;
        c.x0 = std::min(c.x0, px);
        c.x1 = std::max(c.x1, px);
        c.y0 = std::min(c.y0, py);
        c.y1 = std::max(c.y1, py);
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int qy = py + dy, qx = px + dx;
            if ((dy || dx) && qy >= 0 && qy < h && qx >= 0 && qx < w && mask[(size_t)qy * w + qx] != 0 &&
                label[(size_t)qy * w + qx] == 0) {
              label[(size_t)qy * w + qx] = id;
              stack.push_back(qy * w + qx);
            }
          }
      }
      comps.push_back(c);
    }
  // RETR_EXTERNAL: drop components enclosed by another one, i.e. not reachable from the frame through black pixels
  // (4-connected background) - a component is external iff one of its pixels touches the outside background or the frame
  std::vector<uint8_t> outside((size_t)h * w, 0);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      if ((y != 0 && y != h - 1 && x != 0 && x != w - 1) || mask[(size_t)y * w + x] != 0 || outside[(size_t)y * w + x]) continue;
      outside[(size_t)y * w + x] = 1;
      stack.push_back(y * w + x);
      while (!stack.empty()) {
        const int p = stack.back();
        stack.pop_back();
        const int py = p / w, px = p - py * w;
        const int nb[4][2] = {{py - 1, px}, {py + 1, px}, {py, px - 1}, {py, px + 1}};
        for (auto& q : nb)
          if (q[0] >= 0 && q[0] < h && q[1] >= 0 && q[1] < w && mask[(size_t)q[0] * w + q[1]] == 0 &&
              !outside[(size_t)q[0] * w + q[1]]) {
            outside[(size_t)q[0] * w + q[1]] = 1;
            stack.push_back(q[0] * w + q[1]);
          }
      }
    }
  std::vector<uint8_t> external(comps.size() + 1, 0);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int id = label[(size_t)y * w + x];
      if (id == 0 || external[id]) continue;
      if (y == 0 || y == h - 1 || x == 0 || x == w - 1) {
        external[id] = 1;
        continue;
      }
      if (outside[(size_t)(y - 1) * w + x] || outside[(size_t)(y + 1) * w + x] || outside[(size_t)y * w + x - 1] ||
          outside[(size_t)y * w + x + 1])
        external[id] = 1;
    }
  for (int i = (int)comps.size() - 1; i >= 0; --i) {  // newest first
    if (!external[i + 1]) continue;
    const Comp& c = comps[i];
    const int rw = c.x1 - c.x0 + 1, rh = c.y1 - c.y0 + 1;
    if (rw * rh < min_area) continue;
    rects.push_back(c.x0 - pad);
    rects.push_back(c.y0 - pad);
    rects.push_back(c.x0 + rw + pad);
    rects.push_back(c.y0 + rh + pad);
  }
}

}  // namespace ymk

extern "C" int ymk_table_hole_rects(int h, int w, const int* cell_boxes, int n, int pad, int close_ksize, int min_area,
                                    int* rects_out, int capacity, int* count) {
  try {
    YMK_CHECK(count != nullptr && (n == 0 || cell_boxes != nullptr) && (capacity == 0 || rects_out != nullptr), "null argument");
    YMK_CHECK(h >= 0 && w >= 0 && (long)h * w <= (1L << 28), "table crop too large");
    std::vector<int> rects;
    ymk::table_hole_rects(h, w, cell_boxes, n, pad, close_ksize, min_area, rects);
    const int found = (int)rects.size() / 4;
    YMK_CHECK(found <= capacity, "ymk_table_hole_rects: " + std::to_string(found) + " rectangles, capacity " + std::to_string(capacity));
    std::copy(rects.begin(), rects.end(), rects_out);
    *count = found;
    return 0;
  } catch (const std::exception& e) {
    ymk::set_error(e.what());
    return 1;
  }
}
