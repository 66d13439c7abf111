/*
 * svc_imageops.c -- CPU statement of the per-call image operations (include/svc_hip.h "Image operations"): what each
 * OpenCV call of the reference's per-frame loop computes, one function per call.
 * TEST INFRASTRUCTURE ONLY (see svc_oracle.h).
 *
 * Reference call sites: cv::cvtColor(BGR2YUV) libs/encoder.cpp:449, :468; cv::morphologyEx :524-527; cv::kmeans :575-576;
 * cv::connectedComponents :607-610.  All of them live in OpenCV 3.4.x, which is neither vendored nor installed:
 * PARITY UNPINNED, as for the fused forms.  The definitions are the ones oracle/svc_segment.c already states for the
 * fused segmentation (tests/test_cpu_baseline.py checks that composing these functions the way the reference composes
 * the cv:: calls gives svc_oracle_segment's region ids), written a second time here for ANY point dimension up to 4,
 * any 8-bit image and any labelling image -- the generality the per-call C ABI has.
 */
#include <stdlib.h>
#include <string.h>

#include "svc_oracle.h"

/* cv::cvtColor(COLOR_BGR2YUV), 8-bit: OpenCV 3.4's integer path with 14 fractional bits (from its published source,
 * unverifiable offline): Y as svc_oracle_luma; U = descale((B - Y) * 8061 + (128 << 14)), V = descale((R - Y) * 14369
 * + (128 << 14)), descale(x) = (x + 8192) >> 14, saturated. */
void svc_oracle_bgr2yuv(const uint8_t* bgr, uint32_t w, uint32_t h, uint8_t* yuv) {
  const uint64_t n = (uint64_t)w * h;
  for (uint64_t i = 0; i < n; ++i) {
    const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
    const int y = (1868 * b + 9617 * g + 4899 * r + 8192) >> 14;
    int u = ((b - y) * 8061 + (128 << 14) + 8192) >> 14;
    int v = ((r - y) * 14369 + (128 << 14) + 8192) >> 14;
    u = u < 0 ? 0 : u > 255 ? 255 : u;
    v = v < 0 ? 0 : v > 255 ? 255 : v;
    yuv[3 * i] = (uint8_t)y;
    yuv[3 * i + 1] = (uint8_t)u;
    yuv[3 * i + 2] = (uint8_t)v;
  }
}

static void morph_pass(const uint8_t* src, uint8_t* dst, int w, int h, int kw, int kh, int dilate) {
  const int ax = kw / 2, ay = kh / 2;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int v = dilate ? 0 : 255;
      for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx) {
          const int sx = x + kx - ax, sy = y + ky - ay;
          if (sx < 0 || sy < 0 || sx >= w || sy >= h) continue; /* outside the image: ignored */
          const int p = src[sy * w + sx];
          v = dilate ? (p > v ? p : v) : (p < v ? p : v);
        }
      dst[y * w + x] = (uint8_t)v;
    }
}

/* cv::erode / cv::dilate / cv::morphologyEx(OPEN | CLOSE), rectangular element anchored at its centre; op: 0 erode,
 * 1 dilate, 2 open (erode, dilate), 3 close (dilate, erode).  src may equal dst. */
void svc_oracle_morph_rect(const uint8_t* src, uint32_t w, uint32_t h, uint32_t kw, uint32_t kh, uint32_t op, uint8_t* dst) {
  const size_t n = (size_t)w * h;
  uint8_t* a = (uint8_t*)malloc(n ? n : 1);
  uint8_t* b = (uint8_t*)malloc(n ? n : 1);
  if (op <= 1) {
    morph_pass(src, a, (int)w, (int)h, (int)kw, (int)kh, (int)op);
    memcpy(dst, a, n);
  } else {
    const int first = op == 3 ? 1 : 0;
    morph_pass(src, a, (int)w, (int)h, (int)kw, (int)kh, first);
    morph_pass(a, b, (int)w, (int)h, (int)kw, (int)kh, 1 - first);
    memcpy(dst, b, n);
  }
  free(a);
  free(b);
}

static uint64_t io_hash(uint64_t x) { /* splitmix64 finaliser */
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

typedef struct { int64_t f[4]; } pt4;

static uint64_t d2_int(const pt4* a, const pt4* b, uint32_t dims) {
  uint64_t s = 0;
  for (uint32_t d = 0; d < dims; ++d) {
    const int64_t t = a->f[d] - b->f[d];
    s += (uint64_t)(t * t);
  }
  return s;
}

static double d2_dbl(const pt4* p, const double* c, uint32_t dims) {
  const double t0 = (double)p->f[0] - c[0];
  double s = t0 * t0;
  for (uint32_t d = 1; d < dims; ++d) {
    const double t = (double)p->f[d] - c[d];
    s = s + t * t;
  }
  return s;
}

static uint64_t attempt(const pt4* pts, uint32_t n, uint32_t dims, uint32_t k, uint32_t max_iter, double eps2, uint64_t seed,
                        uint32_t att, int32_t* labels) {
  double c[256][4];
  uint32_t chosen[256];
  const uint64_t aseed = seed ^ ((uint64_t)att << 32);
  chosen[0] = (uint32_t)(io_hash(aseed) % n);
  for (uint32_t j = 1; j < k; ++j) { /* k-means++ with exact integer weights */
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
      uint64_t m = UINT64_MAX;
      for (uint32_t q = 0; q < j; ++q) {
        const uint64_t d = d2_int(&pts[i], &pts[chosen[q]], dims);
        if (d < m) m = d;
      }
      total += m;
    }
    uint32_t pick = 0;
    if (total == 0) {
      pick = j < n ? j : 0;
    } else {
      const uint64_t r = io_hash(aseed ^ j) % total;
      uint64_t acc = 0;
      for (uint32_t i = 0; i < n; ++i) {
        uint64_t m = UINT64_MAX;
        for (uint32_t q = 0; q < j; ++q) {
          const uint64_t d = d2_int(&pts[i], &pts[chosen[q]], dims);
          if (d < m) m = d;
        }
        acc += m;
        if (acc > r) { pick = i; break; }
      }
    }
    chosen[j] = pick;
  }
  for (uint32_t j = 0; j < k; ++j)
    for (uint32_t d = 0; d < dims; ++d) c[j][d] = (double)pts[chosen[j]].f[d];
  uint64_t compact = 0;
  for (uint32_t it = 0;; ++it) {
    int64_t sum[256][4];
    uint32_t cnt[256];
    memset(sum, 0, sizeof(sum));
    memset(cnt, 0, sizeof(cnt));
    compact = 0;
    for (uint32_t i = 0; i < n; ++i) {
      double best = d2_dbl(&pts[i], c[0], dims);
      uint32_t bj = 0;
      for (uint32_t j = 1; j < k; ++j) {
        const double d = d2_dbl(&pts[i], c[j], dims);
        if (d < best) { best = d; bj = j; }
      }
      labels[i] = (int32_t)bj;
      cnt[bj]++;
      for (uint32_t d = 0; d < dims; ++d) sum[bj][d] += pts[i].f[d];
      compact += (uint64_t)(best * 256.0);
    }
    if (it + 1 >= max_iter) break;
    double shift = 0.0;
    for (uint32_t j = 0; j < k; ++j) {
      if (!cnt[j]) continue;
      double s = 0.0;
      for (uint32_t d = 0; d < dims; ++d) {
        const double nc = (double)sum[j][d] / (double)cnt[j];
        const double t = nc - c[j][d];
        s = s + t * t;
        c[j][d] = nc;
      }
      if (s > shift) shift = s;
    }
    if (shift <= eps2) break;
  }
  return compact;
}

/* cv::kmeans(KMEANS_PP_CENTERS, COUNT | EPS) by oracle/svc_segment.c's definition on n points of `dims` (1..4) integral
 * coordinates.  Returns 0, or 1 for parameters outside the definition (non-integral features included). */
int svc_oracle_kmeans(const float* features, uint32_t n, uint32_t dims, uint32_t k, uint32_t attempts, uint32_t max_iter,
                      float epsilon, uint64_t seed, int32_t* labels, double* compactness) {
  if (!n || !k || k > 255 || k > n || dims < 1 || dims > 4 || !attempts || !max_iter || !(epsilon > 0)) return 1;
  pt4* pts = (pt4*)malloc(sizeof(pt4) * n);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t d = 0; d < 4; ++d) {
      const float f = d < dims ? features[(size_t)i * dims + d] : 0.0f;
      if (f != (float)(int64_t)f || f <= -32768.0f || f >= 32768.0f) { free(pts); return 1; }
      pts[i].f[d] = (int64_t)f;
    }
  int32_t* lab = (int32_t*)malloc(sizeof(int32_t) * n);
  uint64_t best = UINT64_MAX;
  const double eps2 = (double)epsilon * (double)epsilon;
  for (uint32_t a = 0; a < attempts; ++a) {
    const uint64_t cpt = attempt(pts, n, dims, k, max_iter, eps2, seed, a, lab);
    if (cpt < best) { best = cpt; memcpy(labels, lab, sizeof(int32_t) * n); } /* ties -> the earlier attempt */
  }
  if (compactness) *compactness = (double)best / 256.0;
  free(pts);
  free(lab);
  return 0;
}

static uint32_t root_of(uint32_t* parent, uint32_t i) {
  while (parent[i] != i) {
    parent[i] = parent[parent[i]];
    i = parent[i];
  }
  return i;
}

/* cv::connectedComponents(image, labels, connectivity, CV_32S): labels 1..n in raster order of each component's first
 * pixel, 0 = background; returns n + 1 (OpenCV counts the background label). */
uint32_t svc_oracle_connected_components(const uint8_t* image, uint32_t w, uint32_t h, uint32_t connectivity, int32_t* labels) {
  const uint32_t n = w * h;
  uint32_t* parent = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  for (uint32_t i = 0; i < n; ++i) parent[i] = i;
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      const uint32_t i = y * w + x;
      if (!image[i]) continue;
      const int nb[4][2] = {{-1, 0}, {0, -1}, {-1, -1}, {1, -1}};
      for (int q = 0; q < (connectivity == 8 ? 4 : 2); ++q) {
        const int nx = (int)x + nb[q][0], ny = (int)y + nb[q][1];
        if (nx < 0 || ny < 0 || nx >= (int)w) continue;
        const uint32_t j = (uint32_t)ny * w + (uint32_t)nx;
        if (!image[j]) continue;
        const uint32_t ra = root_of(parent, i), rb = root_of(parent, j);
        if (ra != rb) { if (ra < rb) parent[rb] = ra; else parent[ra] = rb; }
      }
    }
  uint32_t count = 0;
  for (uint32_t i = 0; i < n; ++i) labels[i] = 0;
  for (uint32_t i = 0; i < n; ++i) /* raster order: a root is its component's first pixel */
    if (image[i] && root_of(parent, i) == i) labels[i] = (int32_t)(++count);
  for (uint32_t i = 0; i < n; ++i)
    if (image[i]) labels[i] = labels[root_of(parent, i)];
  free(parent);
  return count + 1;
}
