// DB text-detector post-processing on the host (C++), replacing the Python/cv2/pyclipper loop of
// postprocessor/dbnet_postporcessor.py:32-138:
//   binarise -> border following (cv2.findContours RETR_LIST; Suzuki-Abe 1985, 8-connected) ->
//   minimum-area rectangle (cv2.minAreaRect + boxPoints) -> polygon-mean score (cv2.fillPoly +
//   cv2.mean) -> unclip (shapely area/length + Clipper round-join offset) -> minimum-area
//   rectangle -> scale to the original page with np.round -> int16 quads.
// cv2 / pyclipper / shapely are third-party code that is not in the reference tree; the semantics
// restated here are listed in SURVEY.md Appendix A and marked "parity unpinned" in DESIGN.md.
#include "../../include/ymk.h"
#include "ymk_common.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace ymk {

struct IPt {
  int x, y;
};
struct FPt {
  float x, y;
};

// ---------------------------------------------------------------- border following
// img: (h+2) x (w+2) int32 with a zero frame; non-zero = foreground.  Returns every outer and hole
// border as a closed 8-connected pixel chain (image coordinates, frame removed), in raster
// discovery order.
static void find_borders(std::vector<int>& f, int h, int w, std::vector<std::vector<IPt>>& out) {
  const int W = w + 2;
  // 8 neighbours in clockwise order starting east (dy, dx); index arithmetic mod 8
  static const int DY[8] = {0, 1, 1, 1, 0, -1, -1, -1};
  static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
  auto dir_of = [&](int dy, int dx) {
    for (int d = 0; d < 8; ++d)
      if (DY[d] == dy && DX[d] == dx) return d;
    return 0;
  };
  int nbd = 1;
  for (int i = 1; i <= h; ++i) {
    for (int j = 1; j <= w; ++j) {
      const int v = f[i * W + j];
      if (v == 0) continue;
      int i2, j2;
      if (v == 1 && f[i * W + j - 1] == 0) {
        i2 = i;
        j2 = j - 1;
      } else if (v >= 1 && f[i * W + j + 1] == 0) {
        i2 = i;
        j2 = j + 1;
      } else {
        continue;
      }
      ++nbd;
      std::vector<IPt> chain;
      // (3.1) clockwise from (i2, j2) around (i, j): first non-zero neighbour
      int d0 = dir_of(i2 - i, j2 - j), d1 = -1;
      for (int k = 0; k < 8; ++k) {
        const int d = (d0 + k) & 7;
        if (f[(i + DY[d]) * W + j + DX[d]] != 0) {
          d1 = d;
          break;
        }
      }
      if (d1 < 0) {
        f[i * W + j] = -nbd;
        chain.push_back({j - 1, i - 1});
        out.push_back(std::move(chain));
        continue;
      }
      const int i1 = i + DY[d1], j1 = j + DX[d1];
      int pi = i1, pj = j1;  // (i2, j2)
      int ci = i, cj = j;    // (i3, j3)
      for (;;) {
        // (3.3) counter-clockwise around (ci, cj), starting after (pi, pj)
        const int ds = dir_of(pi - ci, pj - cj);
        int dn = -1;
        bool east_zero_seen = false;
        for (int k = 1; k <= 8; ++k) {
          const int d = (ds - k) & 7;
          if (f[(ci + DY[d]) * W + cj + DX[d]] != 0) {
            dn = d;
            break;
          }
          if (d == 0) east_zero_seen = true;  // (ci, cj + 1) examined and found zero
        }
        chain.push_back({cj - 1, ci - 1});
        // (3.4)
        if (east_zero_seen) f[ci * W + cj] = -nbd;
        else if (f[ci * W + cj] == 1) f[ci * W + cj] = nbd;
        const int ni = ci + DY[dn], nj = cj + DX[dn];
        // (3.5)
        if (ni == i && nj == j && ci == i1 && cj == j1) break;
        pi = ci;
        pj = cj;
        ci = ni;
        cj = nj;
      }
      out.push_back(std::move(chain));
    }
  }
}

// ---------------------------------------------------------------- convex hull + minimum-area rectangle
static long long cross(const IPt& o, const IPt& a, const IPt& b) {
  return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x);
}
static std::vector<IPt> convex_hull(std::vector<IPt> p) {
  std::sort(p.begin(), p.end(), [](const IPt& a, const IPt& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
  p.erase(std::unique(p.begin(), p.end(), [](const IPt& a, const IPt& b) { return a.x == b.x && a.y == b.y; }), p.end());
  const int n = (int)p.size();
  if (n < 3) return p;
  std::vector<IPt> hcv(2 * n);
  int k = 0;
  for (int i = 0; i < n; ++i) {
    while (k >= 2 && cross(hcv[k - 2], hcv[k - 1], p[i]) <= 0) --k;
    hcv[k++] = p[i];
  }
  for (int i = n - 2, t = k + 1; i >= 0; --i) {
    while (k >= t && cross(hcv[k - 2], hcv[k - 1], p[i]) <= 0) --k;
    hcv[k++] = p[i];
  }
  hcv.resize(k - 1);
  return hcv;
}

// cv2.minAreaRect + cv2.boxPoints: 4 corners (float32) and the short side length
static void min_area_rect(const std::vector<IPt>& pts, FPt box[4], float& short_side) {
  std::vector<IPt> hull = convex_hull(pts);
  const int n = (int)hull.size();
  if (n == 1) {
    for (int i = 0; i < 4; ++i) box[i] = {(float)hull[0].x, (float)hull[0].y};
    short_side = 0.f;
    return;
  }
  if (n == 2) {
    box[0] = box[3] = {(float)hull[0].x, (float)hull[0].y};
    box[1] = box[2] = {(float)hull[1].x, (float)hull[1].y};
    short_side = 0.f;
    return;
  }
  double best = 1e300;
  double bc[4][2] = {{0}};
  double bw = 0, bh = 0;
  for (int e = 0; e < n; ++e) {
    const IPt& a = hull[e];
    const IPt& b = hull[(e + 1) % n];
    const double dx = b.x - a.x, dy = b.y - a.y;
    const double len = std::sqrt(dx * dx + dy * dy);
    if (len == 0) continue;
    const double ux = dx / len, uy = dy / len;  // edge direction; normal = (-uy, ux)
    double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
    for (const IPt& q : hull) {
      const double u = q.x * ux + q.y * uy, v = -q.x * uy + q.y * ux;
      umin = std::min(umin, u);
      umax = std::max(umax, u);
      vmin = std::min(vmin, v);
      vmax = std::max(vmax, v);
    }
    const double area = (umax - umin) * (vmax - vmin);
    if (area < best) {
      best = area;
      bw = umax - umin;
      bh = vmax - vmin;
      const double us[4] = {umin, umax, umax, umin}, vs[4] = {vmin, vmin, vmax, vmax};
      for (int c = 0; c < 4; ++c) {
        bc[c][0] = us[c] * ux - vs[c] * uy;
        bc[c][1] = us[c] * uy + vs[c] * ux;
      }
    }
  }
  for (int c = 0; c < 4; ++c) box[c] = {(float)bc[c][0], (float)bc[c][1]};
  short_side = (float)std::min(bw, bh);
}

// get_mini_boxes (dbnet_postporcessor.py:100-124): sort by x, then fix the clockwise order from top-left
static void order_box(FPt b[4]) {
  std::stable_sort(b, b + 4, [](const FPt& p, const FPt& q) { return p.x < q.x; });
  int i1, i2, i3, i4;
  if (b[1].y > b[0].y) { i1 = 0; i4 = 1; } else { i1 = 1; i4 = 0; }
  if (b[3].y > b[2].y) { i2 = 2; i3 = 3; } else { i2 = 3; i3 = 2; }
  const FPt o[4] = {b[i1], b[i2], b[i3], b[i4]};
  for (int c = 0; c < 4; ++c) b[c] = o[c];
}

// ---------------------------------------------------------------- polygon mean (box_score_fast :126-138)
// mask = border pixels of the chain + lattice points strictly inside the closed polygon
static double polygon_mean(const float* pred, int h, int w, const std::vector<IPt>& chain) {
  int xmin = w, xmax = -1, ymin = h, ymax = -1;
  for (const IPt& p : chain) {
    xmin = std::min(xmin, p.x);
    xmax = std::max(xmax, p.x);
    ymin = std::min(ymin, p.y);
    ymax = std::max(ymax, p.y);
  }
  xmin = std::max(0, std::min(xmin, w - 1));
  xmax = std::max(0, std::min(xmax, w - 1));
  ymin = std::max(0, std::min(ymin, h - 1));
  ymax = std::max(0, std::min(ymax, h - 1));
  const int bw = xmax - xmin + 1, bh = ymax - ymin + 1;
  std::vector<unsigned char> mask((size_t)bw * bh, 0);
  for (const IPt& p : chain) mask[(size_t)(p.y - ymin) * bw + (p.x - xmin)] = 1;
  const int n = (int)chain.size();
  std::vector<double> xs;
  for (int y = ymin; y <= ymax; ++y) {
    xs.clear();
    for (int e = 0; e < n; ++e) {
      const IPt& a = chain[e];
      const IPt& b = chain[(e + 1) % n];
      if ((a.y <= y) == (b.y <= y)) continue;  // half-open rule: horizontal edges never cross
      xs.push_back(a.x + (double)(y - a.y) * (b.x - a.x) / (double)(b.y - a.y));
    }
    std::sort(xs.begin(), xs.end());
    for (size_t k = 0; k + 1 < xs.size(); k += 2) {
      int x0 = (int)std::floor(xs[k]) + 1, x1 = (int)std::ceil(xs[k + 1]) - 1;
      x0 = std::max(x0, xmin);
      x1 = std::min(x1, xmax);
      for (int x = x0; x <= x1; ++x) mask[(size_t)(y - ymin) * bw + (x - xmin)] = 1;
    }
  }
  double sum = 0.0;
  long cnt = 0;
  for (int y = 0; y < bh; ++y)
    for (int x = 0; x < bw; ++x)
      if (mask[(size_t)y * bw + x]) {
        sum += pred[(size_t)(y + ymin) * w + (x + xmin)];
        ++cnt;
      }
  return cnt ? sum / (double)cnt : 0.0;
}

// ---------------------------------------------------------------- Clipper-style round-join offset of one closed polygon
static long long cround(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }

static std::vector<IPt> offset_round(const std::vector<IPt>& in, double delta) {
  std::vector<IPt> src;
  {  // AddPath: drop closing duplicates and consecutive duplicates
    int hi = (int)in.size() - 1;
    while (hi > 0 && in[0].x == in[hi].x && in[0].y == in[hi].y) --hi;
    for (int i = 0; i <= hi; ++i)
      if (src.empty() || src.back().x != in[i].x || src.back().y != in[i].y) src.push_back(in[i]);
  }
  const int len = (int)src.size();
  std::vector<IPt> dst;
  if (len < 3) return dst;
  {  // FixOrientations: Clipper's Area() must be >= 0
    double a = 0;
    for (int i = 0, j = len - 1; i < len; j = i++) a += ((double)src[j].x + src[i].x) * ((double)src[j].y - src[i].y);
    if (-a * 0.5 < 0) std::reverse(src.begin(), src.end());
  }
  const double pi = 3.141592653589793238, two_pi = 2 * pi;
  // ClipperOffset::DoOffset: y = ArcTolerance (0.25, the pyclipper default) unless that exceeds |delta| * 0.25 -
  // for |delta| < 1 the tolerance scales down with the offset, so the acos argument stays in [0.75, 1)
  const double arc_tol = std::min(0.25, std::fabs(delta) * 0.25);
  double steps = pi / std::acos(1 - arc_tol / std::fabs(delta));
  if (steps > std::fabs(delta) * pi) steps = std::fabs(delta) * pi;
  double m_sin = std::sin(two_pi / steps);
  const double m_cos = std::cos(two_pi / steps), steps_per_rad = steps / two_pi;
  if (delta < 0) m_sin = -m_sin;
  std::vector<double> nx(len), ny(len);
  for (int j = 0; j < len; ++j) {
    const IPt& p1 = src[j];
    const IPt& p2 = src[(j + 1) % len];
    double dx = (double)(p2.x - p1.x), dy = (double)(p2.y - p1.y);
    const double f = 1.0 / std::sqrt(dx * dx + dy * dy);
    dx *= f;
    dy *= f;
    nx[j] = dy;
    ny[j] = -dx;
  }
  int k = len - 1;
  for (int j = 0; j < len; ++j) {
    double sinA = nx[k] * ny[j] - nx[j] * ny[k];
    bool done = false;
    if (std::fabs(sinA * delta) < 1.0) {
      const double cosA = nx[k] * nx[j] + ny[j] * ny[k];
      if (cosA > 0) {
        dst.push_back({(int)cround(src[j].x + nx[k] * delta), (int)cround(src[j].y + ny[k] * delta)});
        done = true;
      }
    } else if (sinA > 1.0) sinA = 1.0;
    else if (sinA < -1.0) sinA = -1.0;
    if (!done) {
      if (sinA * delta < 0) {
        dst.push_back({(int)cround(src[j].x + nx[k] * delta), (int)cround(src[j].y + ny[k] * delta)});
        dst.push_back(src[j]);
        dst.push_back({(int)cround(src[j].x + nx[j] * delta), (int)cround(src[j].y + ny[j] * delta)});
      } else {  // jtRound
        const double a = std::atan2(sinA, nx[k] * nx[j] + ny[k] * ny[j]);
        const int st = std::max((int)cround(steps_per_rad * std::fabs(a)), 1);
        double X = nx[k], Y = ny[k];
        for (int i = 0; i < st; ++i) {
          dst.push_back({(int)cround(src[j].x + X * delta), (int)cround(src[j].y + Y * delta)});
          const double X2 = X;
          X = X * m_cos - m_sin * Y;
          Y = X2 * m_sin + Y * m_cos;
        }
        dst.push_back({(int)cround(src[j].x + nx[j] * delta), (int)cround(src[j].y + ny[j] * delta)});
      }
    }
    k = j;
  }
  return dst;
}

// unclip (dbnet_postporcessor.py:84-98): heuristic offset distance, then the round-join offset
static std::vector<IPt> unclip(const FPt box[4], float unclip_ratio) {
  double area = 0, length = 0;
  for (int i = 0; i < 4; ++i) {
    const FPt& a = box[i];
    const FPt& b = box[(i + 1) & 3];
    area += (double)a.x * (double)b.y - (double)b.x * (double)a.y;
    length += std::sqrt(((double)b.x - a.x) * ((double)b.x - a.x) + ((double)b.y - a.y) * ((double)b.y - a.y));
  }
  area = std::fabs(area) * 0.5;
  float xmin = box[0].x, xmax = box[0].x, ymin = box[0].y, ymax = box[0].y;
  for (int i = 1; i < 4; ++i) {
    xmin = std::min(xmin, box[i].x);
    xmax = std::max(xmax, box[i].x);
    ymin = std::min(ymin, box[i].y);
    ymax = std::max(ymax, box[i].y);
  }
  const float bw = xmax - xmin, bh = ymax - ymin;  // float32 like the numpy array
  const double ratio = (double)unclip_ratio / std::sqrt((double)std::min(bw, bh));
  const double distance = area * ratio / length;
  std::vector<IPt> poly(4);
  for (int i = 0; i < 4; ++i) poly[i] = {(int)box[i].x, (int)box[i].y};  // pyclipper truncates to integers
  return offset_round(poly, distance);
}

int db_postprocess(const float* pred, int h, int w, float thresh, float box_thresh, int min_size, int max_candidates,
                   float unclip_ratio, int dest_w, int dest_h, int16_t* quads, double* scores, int cap) {
  std::vector<int> f((size_t)(h + 2) * (w + 2), 0);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      if (pred[(size_t)y * w + x] > thresh) f[(size_t)(y + 1) * (w + 2) + x + 1] = 1;
  std::vector<std::vector<IPt>> borders;
  find_borders(f, h, w, borders);
  // cv2 returns RETR_LIST contours newest first
  std::reverse(borders.begin(), borders.end());
  const int ncand = std::min((int)borders.size(), max_candidates);
  int count = 0;
  for (int c = 0; c < ncand; ++c) {
    const std::vector<IPt>& chain = borders[c];
    FPt box[4];
    float sside;
    min_area_rect(chain, box, sside);
    order_box(box);
    if (sside < (float)min_size) continue;
    const double score = polygon_mean(pred, h, w, chain);
    if ((double)box_thresh > score) continue;
    std::vector<IPt> grown = unclip(box, unclip_ratio);
    if (grown.empty()) continue;
    FPt box2[4];
    min_area_rect(grown, box2, sside);
    order_box(box2);
    if (sside < (float)(min_size + 2)) continue;
    if (count >= cap) break;
    for (int k = 0; k < 4; ++k) {
      float x = std::nearbyintf(box2[k].x / (float)w * (float)dest_w);
      float y = std::nearbyintf(box2[k].y / (float)h * (float)dest_h);
      x = std::min(std::max(x, 0.f), (float)dest_w);
      y = std::min(std::max(y, 0.f), (float)dest_h);
      quads[(size_t)count * 8 + 2 * k] = (int16_t)x;
      quads[(size_t)count * 8 + 2 * k + 1] = (int16_t)y;
    }
    scores[count] = score;  // double, as cv2.mean returns it
    ++count;
  }
  return count;
}

}  // namespace ymk

extern "C" int ymk_db_postprocess(const float* prob_host, int h, int w, float thresh, float box_thresh, int min_size,
                                  int max_candidates, float unclip_ratio, int dest_w, int dest_h, int16_t* quads_out,
                                  double* scores_out, int capacity, int* count) {
  try {
    YMK_CHECK(prob_host && quads_out && scores_out && count, "null argument");
    const int n = ymk::db_postprocess(prob_host, h, w, thresh, box_thresh, min_size, max_candidates, unclip_ratio, dest_w,
                                      dest_h, quads_out, scores_out, capacity);
    *count = n;
    return 0;
  } catch (const std::exception& e) {
    ymk::set_error(e.what());
    return 1;
  }
}
