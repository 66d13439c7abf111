/*
 * svc_segment.c -- CPU statement of the segmentation glue between RANSAC and the
 * quantiser: RANSAC inliers + motion field -> region id per MV block.
 * TEST INFRASTRUCTURE ONLY (see svc_oracle.h).
 *
 * Reference: libs/encoder.cpp:507-623.  The in-repo parts are restated exactly:
 *   :507-513  foreground mask = 255 everywhere except the RANSAC inliers
 *   :538-551  foreground index list in raster order; every block starts as background (0)
 *   :316-319  feature of a foreground block = (0, mv.x, x_px, y_px) -- BuildMvFeatures
 *             overwrites the mv.y slot with x_px, so mv.y never reaches k-means
 *   :555      cluster_count = min(kmeans.cluster_count, #foreground)
 *   :597-623  per cluster: connected components; type = component label + offset;
 *             offset += component COUNT INCLUDING the background label 0, so one id is
 *             skipped between clusters
 * The OpenCV parts are PARITY UNPINNED (OpenCV 3.4.x is neither vendored nor installed) and
 * are replaced by this repo's own definitions of the same operations:
 *   cv::morphologyEx CLOSE then OPEN (:524-527): rectangular element, anchor at
 *       (w/2, h/2), pixels outside the image ignored (OpenCV's default border value for
 *       erode/dilate behaves this way);
 *   cv::kmeans (:575-576, KMEANS_PP_CENTERS, COUNT|EPS criteria, `attempts` restarts):
 *       k-means++ seeding and Lloyd iterations, but made deterministic and independent of
 *       summation order so that a parallel implementation reproduces it bit for bit:
 *       features are rounded to integers (lossless for block-matching output), seeding uses
 *       exact 64-bit integer distance sums and a counter hash instead of cv::theRNG(),
 *       centres are integer sums / counts in double, assignment distances are doubles in a
 *       fixed operation order, the compactness that ranks attempts is a fixed-point integer;
 *   cv::connectedComponents (:607-610, 4- or 8-connectivity): labels 1..n in raster order
 *       of each component's first block.
 */
#include "svc_oracle.h"

#include <stdlib.h>
#include <string.h>

static uint64_t seg_hash(uint64_t x) { /* splitmix64 finaliser */
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

static void morph(const uint8_t* src, uint8_t* dst, int w, int h, int kw, int kh, int dilate) {
  const int ax = kw / 2, ay = kh / 2;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int v = dilate ? 0 : 255;
      for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx) {
          int sx = x + kx - ax, sy = y + ky - ay;
          if (sx < 0 || sy < 0 || sx >= w || sy >= h) continue;
          int p = src[sy * w + sx];
          v = dilate ? (p > v ? p : v) : (p < v ? p : v);
        }
      dst[y * w + x] = (uint8_t)v;
    }
}

typedef struct { int64_t f[3]; } pt3; /* (mv.x, x_px, y_px); the constant 0 slot is dropped */

static uint64_t dist2_int(const pt3* a, const pt3* b) {
  uint64_t s = 0;
  for (int d = 0; d < 3; ++d) {
    int64_t t = a->f[d] - b->f[d];
    s += (uint64_t)(t * t);
  }
  return s;
}

static double dist2_dbl(const pt3* p, const double* c) {
  double dx = (double)p->f[0] - c[0], dy = (double)p->f[1] - c[1], dz = (double)p->f[2] - c[2];
  double s = dx * dx;
  s = s + dy * dy;
  s = s + dz * dz;
  return s;
}

/* One k-means attempt; writes labels, returns the fixed-point compactness. */
static uint64_t kmeans_attempt(const pt3* pts, uint32_t n, uint32_t k, uint32_t max_iter, double eps2,
                               uint64_t seed, uint32_t attempt, uint8_t* labels) {
  double c[256][3];
  uint32_t chosen[256];
  /* k-means++ seeding with exact integer weights */
  chosen[0] = (uint32_t)(seg_hash(seed ^ ((uint64_t)attempt << 32)) % n);
  for (uint32_t j = 1; j < k; ++j) {
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
      uint64_t m = UINT64_MAX;
      for (uint32_t q = 0; q < j; ++q) {
        uint64_t d = dist2_int(&pts[i], &pts[chosen[q]]);
        if (d < m) m = d;
      }
      total += m;
    }
    uint32_t pick = 0;
    if (total == 0) { /* every point coincides with a centre: take the first unused index */
      pick = j < n ? j : 0;
    } else {
      uint64_t r = seg_hash(seed ^ ((uint64_t)attempt << 32) ^ j) % total, acc = 0;
      for (uint32_t i = 0; i < n; ++i) {
        uint64_t m = UINT64_MAX;
        for (uint32_t q = 0; q < j; ++q) {
          uint64_t d = dist2_int(&pts[i], &pts[chosen[q]]);
          if (d < m) m = d;
        }
        acc += m;
        if (acc > r) { pick = i; break; }
      }
    }
    chosen[j] = pick;
  }
  for (uint32_t j = 0; j < k; ++j)
    for (int d = 0; d < 3; ++d) c[j][d] = (double)pts[chosen[j]].f[d];

  uint64_t compact = 0;
  for (uint32_t it = 0;; ++it) {
    int64_t sum[256][3];
    uint32_t cnt[256];
    memset(sum, 0, sizeof(sum));
    memset(cnt, 0, sizeof(cnt));
    compact = 0;
    for (uint32_t i = 0; i < n; ++i) {
      double best = dist2_dbl(&pts[i], c[0]);
      uint32_t bj = 0;
      for (uint32_t j = 1; j < k; ++j) {
        double d = dist2_dbl(&pts[i], c[j]);
        if (d < best) { best = d; bj = j; } /* ties -> lowest cluster index */
      }
      labels[i] = (uint8_t)bj;
      cnt[bj]++;
      for (int d = 0; d < 3; ++d) sum[bj][d] += pts[i].f[d];
      compact += (uint64_t)(best * 256.0); /* fixed point: exact, order-independent */
    }
    if (it + 1 >= max_iter) break; /* COUNT criterion: at most max_iter assignment passes */
    double shift = 0.0;
    for (uint32_t j = 0; j < k; ++j) {
      if (!cnt[j]) continue; /* empty cluster keeps its centre */
      double s = 0.0;
      for (int d = 0; d < 3; ++d) {
        double nc = (double)sum[j][d] / (double)cnt[j];
        double t = nc - c[j][d];
        s = s + t * t;
        c[j][d] = nc;
      }
      if (s > shift) shift = s;
    }
    if (shift <= eps2) break; /* EPS criterion: the labels of this pass stand */
  }
  return compact;
}

static uint32_t find_root(uint32_t* parent, uint32_t i) {
  while (parent[i] != i) {
    parent[i] = parent[parent[i]];
    i = parent[i];
  }
  return i;
}

int svc_oracle_segment(const uint8_t* inlier_mask, const svc_oracle_vec2f* mv, uint32_t mfw, uint32_t mfh,
                       uint32_t mv_bw, uint32_t mv_bh, uint32_t morph_w, uint32_t morph_h,
                       uint32_t cluster_count, uint32_t attempts, uint32_t max_iter, float epsilon,
                       uint32_t connectivity, uint64_t seed, uint32_t* block_types) {
  const uint32_t n = mfw * mfh;
  if (!n || cluster_count == 0 || cluster_count > 255 || !attempts || !max_iter || !(epsilon > 0) ||
      (connectivity != 4 && connectivity != 8) || !morph_w || !morph_h)
    return 1; /* libs/encoder.cpp:39-61, :92-97 (Validate) */
  uint8_t* fg = (uint8_t*)malloc(n);
  uint8_t* tmp = (uint8_t*)malloc(n);
  for (uint32_t i = 0; i < n; ++i) fg[i] = inlier_mask[i] ? 0 : 255; /* :507-513 */
  morph(fg, tmp, (int)mfw, (int)mfh, (int)morph_w, (int)morph_h, 1);  /* close = dilate, erode */
  morph(tmp, fg, (int)mfw, (int)mfh, (int)morph_w, (int)morph_h, 0);
  morph(fg, tmp, (int)mfw, (int)mfh, (int)morph_w, (int)morph_h, 0);  /* open = erode, dilate */
  morph(tmp, fg, (int)mfw, (int)mfh, (int)morph_w, (int)morph_h, 1);

  uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * n);
  uint32_t nf = 0;
  for (uint32_t i = 0; i < n; ++i) {
    block_types[i] = 0; /* :549-551 */
    if (fg[i] == 255) idx[nf++] = i; /* :540-546 */
  }
  if (nf) {
    const uint32_t k = cluster_count < nf ? cluster_count : nf; /* :555 */
    pt3* pts = (pt3*)malloc(sizeof(pt3) * nf);
    for (uint32_t i = 0; i < nf; ++i) { /* :300-321 BuildMvFeatures */
      uint32_t b = idx[i];
      float mx = mv[b].x;
      pts[i].f[0] = (int64_t)(mx < 0 ? mx - 0.5f : mx + 0.5f); /* round half away; exact for integral MVs */
      pts[i].f[1] = (int64_t)((b % mfw) * mv_bw);
      pts[i].f[2] = (int64_t)((b / mfw) * mv_bh);
    }
    uint8_t* lab = (uint8_t*)malloc(nf);
    uint8_t* best_lab = (uint8_t*)malloc(nf);
    uint64_t best_c = UINT64_MAX;
    const double eps2 = (double)epsilon * (double)epsilon;
    for (uint32_t a = 0; a < attempts; ++a) {
      uint64_t cpt = kmeans_attempt(pts, nf, k, max_iter, eps2, seed, a, lab);
      if (cpt < best_c) { best_c = cpt; memcpy(best_lab, lab, nf); } /* ties -> earlier attempt */
    }
    /* per cluster connected components (:597-623) */
    int32_t* cl = (int32_t*)malloc(sizeof(int32_t) * n);
    uint32_t* parent = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint32_t* label = (uint32_t*)malloc(sizeof(uint32_t) * n);
    for (uint32_t i = 0; i < n; ++i) cl[i] = -1;
    for (uint32_t i = 0; i < nf; ++i) cl[idx[i]] = best_lab[i];
    for (uint32_t i = 0; i < n; ++i) parent[i] = i;
    for (uint32_t y = 0; y < mfh; ++y)
      for (uint32_t x = 0; x < mfw; ++x) {
        uint32_t i = y * mfw + x;
        if (cl[i] < 0) continue;
        const int nb[4][2] = {{-1, 0}, {0, -1}, {-1, -1}, {1, -1}};
        for (int q = 0; q < (connectivity == 8 ? 4 : 2); ++q) {
          int nx = (int)x + nb[q][0], ny = (int)y + nb[q][1];
          if (nx < 0 || ny < 0 || nx >= (int)mfw) continue;
          uint32_t j = (uint32_t)ny * mfw + (uint32_t)nx;
          if (cl[j] != cl[i]) continue;
          uint32_t ra = find_root(parent, i), rb = find_root(parent, j);
          if (ra != rb) { if (ra < rb) parent[rb] = ra; else parent[ra] = rb; } /* root = first block */
        }
      }
    uint32_t offset = 0; /* BLOCK_TYPE_BACKGROUND, libs/codec.hpp:6 */
    for (uint32_t cid = 0; cid < k; ++cid) {
      uint32_t ncomp = 0;
      for (uint32_t i = 0; i < n; ++i) /* raster order: a root is its component's first block */
        if (cl[i] == (int32_t)cid && find_root(parent, i) == i) label[i] = ++ncomp;
      for (uint32_t i = 0; i < n; ++i)
        if (cl[i] == (int32_t)cid) block_types[i] = label[find_root(parent, i)] + offset; /* :617 */
      offset += ncomp + 1; /* :620: connectedComponents' count includes label 0 */
    }
    free(cl); free(parent); free(label); free(lab); free(best_lab); free(pts);
  }
  free(fg); free(tmp); free(idx);
  return 0;
}
