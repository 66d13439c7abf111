/*
 * svc_oracle.c -- CPU restatement of the reference encode hot path (plain C).
 * TEST INFRASTRUCTURE ONLY; see svc_oracle.h for the parity status of each part.
 *
 * Build: gcc -O3 -std=c11 -msse2 -ffp-contract=off (oracle/Makefile).  The
 * reference is built by g++ for baseline x86-64, which has no FMA, so float
 * expressions here must not be contracted either.
 */
#include "svc_oracle.h"

#include <emmintrin.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- MAD --------------------------------------------------------------- */

/* libs/motion.cpp:27-42: integer SAD in a 32-bit unsigned, one f32 divide by
 * the (unsigned -> float) pixel count. */
static inline uint32_t sad_block(const uint8_t* a, const uint8_t* b,
                                 uint32_t stride, uint32_t bw, uint32_t bh) {
  uint32_t sad = 0;
  for (uint32_t r = 0; r < bh; ++r) {
    const uint8_t* pa = a + (size_t)r * stride;
    const uint8_t* pb = b + (size_t)r * stride;
    for (uint32_t c = 0; c < bw; ++c) {
      int d = (int)pa[c] - (int)pb[c];
      sad += (uint32_t)(d < 0 ? -d : d); /* libs/math.hpp:60-65 AbsDiff */
    }
  }
  return sad;
}

float svc_oracle_mad(const uint8_t* a_frame, const uint8_t* b_frame,
                     uint32_t frame_w, uint32_t ax, uint32_t ay, uint32_t bx,
                     uint32_t by, uint32_t block_w, uint32_t block_h) {
  uint32_t sad = sad_block(a_frame + (size_t)ay * frame_w + ax,
                           b_frame + (size_t)by * frame_w + bx, frame_w,
                           block_w, block_h);
  uint32_t count = block_w * block_h;
  return (float)sad / (float)count; /* :38-40 */
}

/* ---- whole-frame global motion (no caller in the reference) ------------- */

/* libs/motion.cpp:45-53 */
void svc_oracle_global_avg(const svc_oracle_vec2f* mv, uint32_t n, svc_oracle_vec2f* avg) {
  float ax = 0.0f, ay = 0.0f; /* Vec2f avg = {} */
  for (uint32_t i = 0; i < n; ++i) {
    float r = 1.0f / (float)(i + 1); /* 1.0f / (i + 1): unsigned -> float, f32 divide */
    ax = ax + (mv[i].x - ax) * r;    /* avg += (mv - avg) * r, componentwise (math.hpp) */
    ay = ay + (mv[i].y - ay) * r;
  }
  avg->x = ax;
  avg->y = ay;
}

/* libs/motion.cpp:55-99 */
void svc_oracle_global_ebma(const uint8_t* tracked, const uint8_t* anchor, uint32_t w,
                            uint32_t h, uint32_t search_range, int reference_loop,
                            svc_oracle_vec2f* gm, float* min_mad) {
  gm->x = 0.0f; /* :66 */
  gm->y = 0.0f;
  *min_mad = FLT_MAX; /* :67 */
  const int r = (int)search_range;
  for (int dy = -r;; ++dy) {
    /* :72 `dy <= search_range` with dy int, search_range unsigned */
    if (reference_loop ? !((unsigned)dy <= search_range) : !(dy <= r)) break;
    uint32_t ty0 = (uint32_t)(dy > 0 ? dy : 0);              /* Max(0, dy) :73 */
    uint32_t ty1 = (uint32_t)((int)h + (dy < 0 ? dy : 0));    /* :74 */
    uint32_t bh = ty1 - ty0;                                  /* :76 */
    uint32_t ay0 = (uint32_t)((int)ty0 - dy);                 /* :79 */
    for (int dx = -r;; ++dx) {
      if (reference_loop ? !((unsigned)dx <= search_range) : !(dx <= r)) break; /* :81 */
      uint32_t tx0 = (uint32_t)(dx > 0 ? dx : 0);
      uint32_t tx1 = (uint32_t)((int)w + (dx < 0 ? dx : 0));
      uint32_t bw = tx1 - tx0;
      uint32_t ax0 = (uint32_t)((int)tx0 - dx);
      float mad = svc_oracle_mad(tracked, anchor, w, tx0, ty0, ax0, ay0, bw, bh); /* :89-90 */
      if (mad < *min_mad) { /* :92 strict: the first minimum in raster order stays */
        *min_mad = mad;
        gm->x = (float)dx;
        gm->y = (float)dy;
      }
    }
  }
}

/* libs/motion.cpp:101-142 */
void svc_oracle_global_hbma(const uint8_t* const* tracked_pyr,
                            const uint8_t* const* anchor_pyr, uint32_t levels, uint32_t w,
                            uint32_t h, uint32_t search_range, int reference_loop,
                            svc_oracle_vec2f* gm) {
  uint32_t f = 1;
  for (uint32_t i = 0; i + 1 < levels; ++i) f <<= 1; /* :114-117 */
  uint32_t fw = w / f, fh = h / f;
  float mm;
  svc_oracle_global_ebma(tracked_pyr[levels - 1], anchor_pyr[levels - 1], fw, fh,
                         search_range / f, reference_loop, gm, &mm); /* :124-129 */
  for (int l = (int)levels - 2; l >= 0; --l) {
    svc_oracle_vec2f corr;
    fh *= 2;
    fw *= 2;
    svc_oracle_global_ebma(tracked_pyr[l], anchor_pyr[l], fw, fh, 1, reference_loop, &corr, &mm); /* :136-138 */
    gm->x = 2.0f * gm->x + corr.x; /* :140 */
    gm->y = 2.0f * gm->y + corr.y;
  }
}

/* libs/motion.cpp:472-510 (Mad16x16Sse2): two row pairs per step, two
 * independent psadbw accumulators, horizontal add, divide by 256.0f. */
static inline float mad16_sse2(const uint8_t* a, const uint8_t* b,
                               uint32_t stride) {
  __m128i acc_even = _mm_setzero_si128();
  __m128i acc_odd = _mm_setzero_si128();
  for (uint32_t r = 0; r < 16; r += 2) {
    __m128i ta = _mm_loadu_si128((const __m128i*)(a + (size_t)r * stride));
    __m128i tb = _mm_loadu_si128((const __m128i*)(b + (size_t)r * stride));
    __m128i ua = _mm_loadu_si128((const __m128i*)(a + (size_t)(r + 1) * stride));
    __m128i ub = _mm_loadu_si128((const __m128i*)(b + (size_t)(r + 1) * stride));
    acc_even = _mm_add_epi64(acc_even, _mm_sad_epu8(ta, tb));
    acc_odd = _mm_add_epi64(acc_odd, _mm_sad_epu8(ua, ub));
  }
  __m128i s = _mm_add_epi64(acc_even, acc_odd);
  long long sad = _mm_cvtsi128_si64(_mm_add_epi64(s, _mm_srli_si128(s, 8)));
  return (float)sad / 256.0f; /* :507 */
}

/* libs/motion.cpp:516-550 (Mad8x8Sse2): two 8-byte rows interleaved into one
 * 16-byte vector per frame, one psadbw per row pair, divide by 64.0f. */
static inline float mad8_sse2(const uint8_t* a, const uint8_t* b,
                              uint32_t stride) {
  __m128i acc = _mm_setzero_si128();
  for (uint32_t r = 0; r < 8; r += 2) {
    __m128i va = _mm_unpacklo_epi8(
        _mm_loadl_epi64((const __m128i*)(a + (size_t)r * stride)),
        _mm_loadl_epi64((const __m128i*)(a + (size_t)(r + 1) * stride)));
    __m128i vb = _mm_unpacklo_epi8(
        _mm_loadl_epi64((const __m128i*)(b + (size_t)r * stride)),
        _mm_loadl_epi64((const __m128i*)(b + (size_t)(r + 1) * stride)));
    acc = _mm_add_epi64(acc, _mm_sad_epu8(va, vb));
  }
  long long sad = _mm_cvtsi128_si64(_mm_add_epi64(acc, _mm_srli_si128(acc, 8)));
  return (float)sad / 64.0f; /* :547 */
}

/* ---- search windows ---------------------------------------------------- */

/* The reference forms every window bound the same way (:297-299, :308-310,
 * :375-385): begin = Max(int 0, int centre - int R) stored to an unsigned;
 * end = Min(unsigned dim - block + 1, unsigned centre + R + 1). */
static inline void window(uint32_t centre, uint32_t range, uint32_t dim,
                          uint32_t block, uint32_t* begin, uint32_t* end) {
  int lo = (int)centre - (int)range;
  *begin = (uint32_t)(lo < 0 ? 0 : lo);
  uint32_t cap = dim - block + 1u;
  uint32_t hi = centre + range + 1u;
  *end = hi < cap ? hi : cap;
}

/* ---- EBMA -------------------------------------------------------------- */

void svc_oracle_ebma(const uint8_t* tracked, const uint8_t* anchor,
                     uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                     uint32_t block_w, uint32_t block_h, svc_oracle_vec2f* mv,
                     float* min_mad) {
  uint32_t fw = frame_w / block_w, fh = frame_h / block_h;
  float area = (float)(block_w * block_h);

  /* :288-291: every block is initialised before any block is searched. */
  for (uint32_t i = 0; i < fw * fh; ++i) {
    mv[i].x = 0.0f;
    mv[i].y = 0.0f;
    min_mad[i] = FLT_MAX;
  }

  for (uint32_t by = 0; by < fh; ++by) {
    uint32_t ay = by * block_h, y0, y1;
    window(ay, search_range, frame_h, block_h, &y0, &y1);
    for (uint32_t bx = 0; bx < fw; ++bx) {
      uint32_t ax = bx * block_w, x0, x1;
      window(ax, search_range, frame_w, block_w, &x0, &x1);
      uint32_t i = by * fw + bx;
      const uint8_t* pa = anchor + (size_t)ay * frame_w + ax;
      uint32_t updates = 0;
      for (uint32_t y = y0; y < y1; ++y) {
        for (uint32_t x = x0; x < x1; ++x) {
          uint32_t sad = sad_block(tracked + (size_t)y * frame_w + x, pa,
                                   frame_w, block_w, block_h);
          float mad = (float)sad / area;
          if (mad <= min_mad[i]) { /* :324 non-strict: last minimum wins */
            min_mad[i] = mad;
            mv[i].x = (float)((int)x - (int)ax); /* :326-327 */
            mv[i].y = (float)((int)y - (int)ay);
            ++updates;
          }
        }
      }
      /* :333-337: every candidate updated <=> MADs non-increasing in raster
       * order; the MV is zeroed, min_mad keeps the last (smallest) MAD. */
      if (updates == (y1 - y0) * (x1 - x0)) {
        mv[i].x = 0.0f;
        mv[i].y = 0.0f;
      }
    }
  }
}

/* ---- refinement -------------------------------------------------------- */

typedef float (*mad_fn)(const uint8_t*, const uint8_t*, uint32_t);

static inline int round_to_int(float v) { return (int)roundf(v); } /* math.hpp:15-18 */

static void refine_impl(const uint8_t* tracked, const uint8_t* anchor,
                        uint32_t frame_w, uint32_t frame_h, uint32_t block_w,
                        uint32_t block_h, uint32_t search_range,
                        svc_oracle_vec2f* mv, float* min_mad, mad_fn fast) {
  uint32_t fw = frame_w / block_w, fh = frame_h / block_h;
  float area = (float)(block_w * block_h);
  for (uint32_t by = 0; by < fh; ++by) {
    uint32_t ay = by * block_h;
    for (uint32_t bx = 0; bx < fw; ++bx) {
      uint32_t ax = bx * block_w;
      uint32_t i = by * fw + bx;
      /* :372-373: centre = anchor + round(mv), held in unsigned */
      uint32_t cx = (uint32_t)((int)ax + round_to_int(mv[i].x));
      uint32_t cy = (uint32_t)((int)ay + round_to_int(mv[i].y));
      uint32_t x0, x1, y0, y1;
      window(cy, search_range, frame_h, block_h, &y0, &y1);
      window(cx, search_range, frame_w, block_w, &x0, &x1);
      const uint8_t* pa = anchor + (size_t)ay * frame_w + ax;
      for (uint32_t y = y0; y < y1; ++y) {
        for (uint32_t x = x0; x < x1; ++x) {
          const uint8_t* pt = tracked + (size_t)y * frame_w + x;
          float mad = fast ? fast(pt, pa, frame_w)
                           : (float)sad_block(pt, pa, frame_w, block_w, block_h) / area;
          /* :401 strict, against the value CARRIED from the coarser level */
          if (mad < min_mad[i]) {
            min_mad[i] = mad;
            mv[i].x = (float)((int)x - (int)ax);
            mv[i].y = (float)((int)y - (int)ay);
          }
        }
      }
    }
  }
}

void svc_oracle_refine(const uint8_t* tracked, const uint8_t* anchor,
                       uint32_t frame_w, uint32_t frame_h, uint32_t block_w,
                       uint32_t block_h, uint32_t search_range,
                       svc_oracle_vec2f* mv, float* min_mad) {
  refine_impl(tracked, anchor, frame_w, frame_h, block_w, block_h, search_range,
              mv, min_mad, NULL);
}

/* ---- HBMA -------------------------------------------------------------- */

int svc_oracle_hbma(const uint8_t* const* tracked_pyr,
                    const uint8_t* const* anchor_pyr, uint32_t level_count,
                    uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                    uint32_t block_w, uint32_t block_h, svc_oracle_vec2f* mv,
                    float* min_mad) {
  if (!tracked_pyr || !anchor_pyr || !mv || !min_mad) return 1;
  if (level_count == 0 || level_count > 16 || !block_w || !block_h) return 1;
  if (!frame_w || !frame_h || frame_w % block_w || frame_h % block_h) return 1;
  uint32_t f = 1u << (level_count - 1); /* :431 */
  if (search_range < f) return 1;       /* :433 */
  uint32_t r_top = search_range / f;    /* :435 */
  uint32_t fw = frame_w / f, fh = frame_h / f;
  uint32_t bw = block_w / f, bh = block_h / f;
  if (!bw || !bh) return 1;

  svc_oracle_ebma(tracked_pyr[level_count - 1], anchor_pyr[level_count - 1], fw,
                  fh, r_top, bw, bh, mv, min_mad); /* :443-445 */

  uint32_t n = (frame_w / block_w) * (frame_h / block_h);
  for (int l = (int)level_count - 2; l >= 0; --l) { /* :451-464 */
    fw *= 2; fh *= 2; bw *= 2; bh *= 2;
    for (uint32_t i = 0; i < n; ++i) {
      mv[i].x *= 2.0f;
      mv[i].y *= 2.0f;
    }
    /* the TOP level's range is reused at every level (:462-463) */
    svc_oracle_refine(tracked_pyr[l], anchor_pyr[l], fw, fh, bw, bh, r_top, mv,
                      min_mad);
  }
  return 0;
}

int svc_oracle_hbma16_sse2(const uint8_t* const* tracked_pyr,
                           const uint8_t* const* anchor_pyr, uint32_t frame_w,
                           uint32_t frame_h, uint32_t search_range,
                           svc_oracle_vec2f* mv, float* min_mad) {
  if (!tracked_pyr || !anchor_pyr || !mv || !min_mad) return 1;
  if (!frame_w || !frame_h || frame_w % 16 || frame_h % 16) return 1;
  if (search_range < 8) return 1; /* :712 */
  uint32_t r_top = search_range / 8;
  uint32_t fw = frame_w / 8, fh = frame_h / 8;
  uint32_t n = (frame_w / 16) * (frame_h / 16);

  svc_oracle_ebma(tracked_pyr[3], anchor_pyr[3], fw, fh, r_top, 2, 2, mv,
                  min_mad); /* :719-720 */
  for (int l = 2; l >= 0; --l) {
    fw *= 2; fh *= 2;
    for (uint32_t i = 0; i < n; ++i) {
      mv[i].x *= 2.0f;
      mv[i].y *= 2.0f;
    }
    uint32_t b = 16u >> l;
    mad_fn fast = l == 0 ? mad16_sse2 : (l == 1 ? mad8_sse2 : NULL); /* :731-748 */
    refine_impl(tracked_pyr[l], anchor_pyr[l], fw, fh, b, b, r_top, mv, min_mad,
                fast);
  }
  return 0;
}

/* ---- RANSAC ------------------------------------------------------------ */

uint32_t svc_oracle_ransac_iter_count(svc_oracle_ransac_params p) {
  float num = logf(1 - p.success_prob);                         /* :145 */
  float den = logf(1 - powf(p.inlier_ratio, (float)p.subset_sz)); /* :146 */
  return (uint32_t)ceilf(num / den);                            /* :147 */
}

static inline float sqrf(float v) { return v * v; }

/* :151-163: sequential f32 sum in index order, then one multiply by 1/n. */
static svc_oracle_vec2f mean_motion(const svc_oracle_vec2f* mf,
                                    const uint32_t* idx, uint32_t n) {
  svc_oracle_vec2f s = {0.0f, 0.0f};
  for (uint32_t i = 0; i < n; ++i) {
    s.x = s.x + mf[idx[i]].x;
    s.y = s.y + mf[idx[i]].y;
  }
  float inv = 1.0f / (float)n;
  s.x = s.x * inv;
  s.y = s.y * inv;
  return s;
}

/* :165-180 */
static float rmse_of(const svc_oracle_vec2f* mf, const uint32_t* idx,
                     uint32_t n, svc_oracle_vec2f est) {
  float acc = 0;
  for (uint32_t i = 0; i < n; ++i) {
    svc_oracle_vec2f m = mf[idx[i]];
    acc += sqrf(m.x - est.x) + sqrf(m.y - est.y);
  }
  return sqrtf(acc / (float)n);
}

static uint32_t collect_inliers(const svc_oracle_vec2f* mf, uint32_t n,
                                svc_oracle_vec2f gm, float thresh,
                                uint32_t limit, uint32_t* out) {
  uint32_t k = 0;
  for (uint32_t i = 0; i < n && k < limit; ++i) {
    if (sqrf(gm.x - mf[i].x) + sqrf(gm.y - mf[i].y) < sqrf(thresh)) out[k++] = i;
  }
  return k;
}

void svc_oracle_ransac(const svc_oracle_vec2f* motion_field, uint32_t n,
                       svc_oracle_ransac_params params, const uint32_t* samples,
                       uint32_t iter_count, float* rmse,
                       svc_oracle_vec2f* global_motion, uint32_t* inliers,
                       uint32_t* inlier_count) {
  uint32_t ns = params.subset_sz;
  uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t* best = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t best_n = 0;
  const uint32_t* best_subset = samples;
  svc_oracle_vec2f best_gm = {0.0f, 0.0f};

  for (uint32_t it = 0; it < iter_count; ++it) { /* :210 */
    const uint32_t* subset = samples + (size_t)it * ns;
    svc_oracle_vec2f gm = mean_motion(motion_field, subset, ns); /* :222 */
    uint32_t k = collect_inliers(motion_field, n, gm, params.inlier_thresh, n, cur);
    if (k >= best_n) { /* :233 ties go to the later iteration */
      best_gm = gm;
      best_subset = subset;
      uint32_t* t = best; best = cur; cur = t;
      best_n = k;
    }
  }

  if (best_n < ns) { /* :240-242: rmse against the caller's incoming value */
    *rmse = rmse_of(motion_field, best_subset, ns, *global_motion);
  } else { /* :244-261 */
    uint32_t k = collect_inliers(motion_field, n, best_gm, params.inlier_thresh,
                                 best_n, cur);
    best_gm = mean_motion(motion_field, cur, k);
    *rmse = rmse_of(motion_field, cur, k, best_gm);
    uint32_t* t = best; best = cur; cur = t;
    best_n = k;
  }
  *global_motion = best_gm; /* :264 */
  memcpy(inliers, best, sizeof(uint32_t) * best_n);
  *inlier_count = best_n;
  free(cur);
  free(best);
}

/* ---- luma + pyramid (OpenCV steps, parity unpinned: see svc_oracle.h) ----- */

void svc_oracle_luma(const uint8_t* bgr, uint32_t w, uint32_t h, uint8_t* y) {
  const uint64_t n = (uint64_t)w * h;
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
    y[i] = (uint8_t)((b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14); /* <= 255: the weights sum to 2^14 */
  }
}

static int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

void svc_oracle_pyr_down(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst) {
  static const int k[5] = {1, 4, 6, 4, 1};
  const uint32_t dw = (w + 1) / 2, dh = (h + 1) / 2;
  int* rows = (int*)malloc((size_t)h * dw * sizeof(int)); /* horizontal pass of every source row */
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < dw; ++x) {
      int s = 0;
      for (int t = 0; t < 5; ++t) s += k[t] * src[(size_t)y * w + reflect101(2 * (int)x + t - 2, (int)w)];
      rows[(size_t)y * dw + x] = s;
    }
  for (uint32_t y = 0; y < dh; ++y)
    for (uint32_t x = 0; x < dw; ++x) {
      int s = 0;
      for (int t = 0; t < 5; ++t) s += k[t] * rows[(size_t)reflect101(2 * (int)y + t - 2, (int)h) * dw + x];
      dst[(size_t)y * dw + x] = (uint8_t)((s + 128) >> 8);
    }
  free(rows);
}

void svc_oracle_luma_pyramid(const uint8_t* bgr, uint32_t w, uint32_t h, uint32_t levels, uint8_t* packed) {
  svc_oracle_luma(bgr, w, h, packed);
  uint8_t* lvl = packed;
  for (uint32_t l = 1; l < levels; ++l) {
    uint8_t* next = lvl + (size_t)w * h;
    svc_oracle_pyr_down(lvl, w, h, next);
    lvl = next;
    w = (w + 1) / 2;
    h = (h + 1) / 2;
  }
}

void svc_oracle_fg_mask(const uint32_t* inliers, uint32_t inlier_count,
                        uint32_t n, uint8_t* mask) {
  memset(mask, 255, n);
  for (uint32_t i = 0; i < inlier_count; ++i) mask[inliers[i]] = 0;
}

/* ---- quantisation ------------------------------------------------------ */

void svc_oracle_quant(float* coeffs, uint64_t n, uint32_t step) {
  for (uint64_t i = 0; i < n; ++i) {
    float c = coeffs[i];
    c = c / (float)step; /* decoder.cpp:141 (float /= unsigned) */
    c = roundf(c);       /* :142 std::round, half away from zero */
    c = c * (float)step; /* :143 */
    coeffs[i] = c;
  }
}

void svc_oracle_quant_frame(float* planes, uint32_t w, uint32_t h,
                            uint32_t mv_bw, uint32_t mv_bh,
                            const uint32_t* block_types, uint32_t fg_step,
                            uint32_t bg_step) {
  uint32_t mv_fw = w / mv_bw;
  for (uint32_t p = 0; p < 3; ++p) {
    float* plane = planes + (size_t)p * w * h;
    for (uint32_t y = 0; y < h; ++y) {
      for (uint32_t x = 0; x < w; ++x) {
        uint32_t type = block_types[(y / mv_bh) * mv_fw + x / mv_bw];
        uint32_t step = type == 0 ? bg_step : fg_step; /* decoder.cpp:130-135, codec.hpp:6 */
        svc_oracle_quant(plane + (size_t)y * w + x, 1, step);
      }
    }
  }
}

/* ---- wire format ------------------------------------------------------- */

/* libs/encoder.cpp:222-269 (SerializeEncodedFrame), literally: tiles are visited over
 * frame_w x frame_h AS PASSED (the encoder passes the UNPADDED size, :647-650), each tile is
 * the u32 type of its MV block followed, per channel, by `transform_block_w` rows of
 * `transform_block_h` floats read at ch[y * frame_w + tb_x] -- the passed width is also the
 * row stride, and the w/h of the transform block are swapped (:257-262).  Returns the number
 * of bytes written. */
uint64_t svc_oracle_serialize_frame(const float* planes, uint64_t plane_elems, uint32_t channels,
                                    const uint32_t* block_types, uint32_t frame_w, uint32_t frame_h,
                                    uint32_t transform_block_w, uint32_t transform_block_h,
                                    uint32_t mv_field_w, uint32_t mv_block_w, uint32_t mv_block_h,
                                    uint8_t* out) {
  uint8_t* o = out;
  for (uint32_t tb_y = 0; tb_y < frame_h; tb_y += transform_block_h) {
    for (uint32_t tb_x = 0; tb_x < frame_w; tb_x += transform_block_w) {
      uint32_t type = block_types[(tb_y / mv_block_h) * mv_field_w + tb_x / mv_block_w]; /* :243-247 */
      memcpy(o, &type, 4);
      o += 4;
      for (uint32_t c = 0; c < channels; ++c) {
        const float* ch = planes + (size_t)c * plane_elems;
        for (uint32_t y = tb_y; y < tb_y + transform_block_w; ++y) { /* :257 */
          memcpy(o, ch + (size_t)y * frame_w + tb_x, sizeof(float) * transform_block_h); /* :258-262 */
          o += sizeof(float) * transform_block_h;
        }
      }
    }
  }
  return (uint64_t)(o - out);
}

/* ---- DCT --------------------------------------------------------------- */

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Orthonormal DCT-II basis C[k][n] = s_k cos(pi (2n+1) k / 2N), s_0 = sqrt(1/N),
 * s_k = sqrt(2/N): what cv::dct(src, dst, 0) computes (encoder.cpp:335). */
static void dct_basis(uint32_t n, double* c) {
  for (uint32_t k = 0; k < n; ++k) {
    double s = k == 0 ? sqrt(1.0 / n) : sqrt(2.0 / n);
    for (uint32_t i = 0; i < n; ++i)
      c[k * n + i] = s * cos(M_PI * (2.0 * i + 1.0) * k / (2.0 * n));
  }
}

void svc_oracle_dct_frame_f64(const uint8_t* bgr, uint32_t w, uint32_t h,
                              uint32_t block_w, uint32_t block_h,
                              double* planes64) {
  double* cw = (double*)malloc(sizeof(double) * block_w * block_w);
  double* ch = (double*)malloc(sizeof(double) * block_h * block_h);
  double* tmp = (double*)malloc(sizeof(double) * block_w * block_h);
  dct_basis(block_w, cw);
  dct_basis(block_h, ch);
  for (uint32_t p = 0; p < 3; ++p) { /* cv::split order B, G, R (:328) */
    double* plane = planes64 + (size_t)p * w * h;
    for (uint32_t ty = 0; ty < h; ty += block_h) {
      for (uint32_t tx = 0; tx < w; tx += block_w) {
        /* rows: tmp[y][u] = sum_x X[y][x] Cw[u][x] */
        for (uint32_t y = 0; y < block_h; ++y)
          for (uint32_t u = 0; u < block_w; ++u) {
            double acc = 0.0;
            for (uint32_t x = 0; x < block_w; ++x)
              acc += (double)bgr[((size_t)(ty + y) * w + tx + x) * 3 + p] *
                     cw[u * block_w + x];
            tmp[y * block_w + u] = acc;
          }
        /* columns: Y[v][u] = sum_y Ch[v][y] tmp[y][u] */
        for (uint32_t v = 0; v < block_h; ++v)
          for (uint32_t u = 0; u < block_w; ++u) {
            double acc = 0.0;
            for (uint32_t y = 0; y < block_h; ++y)
              acc += ch[v * block_h + y] * tmp[y * block_w + u];
            plane[(size_t)(ty + v) * w + tx + u] = acc;
          }
      }
    }
  }
  free(cw);
  free(ch);
  free(tmp);
}

void svc_oracle_dct_frame_f32(const uint8_t* bgr, uint32_t w, uint32_t h,
                              uint32_t block_w, uint32_t block_h,
                              float* planes32) {
  size_t n = (size_t)3 * w * h;
  double* p64 = (double*)malloc(sizeof(double) * n);
  svc_oracle_dct_frame_f64(bgr, w, h, block_w, block_h, p64);
  for (size_t i = 0; i < n; ++i) planes32[i] = (float)p64[i];
  free(p64);
}

/* ---- decoder-side inverse path (headless) -------------------------------- */

/* libs/decoder.cpp:128-149 (DecodeBlock) over every tile of a frame, as Decoder::operator()
 * walks them (:183-207): step = gazed ? 1 : (type == 0 ? bg : fg) with
 * gazed = gaze_rect.contains(tile origin) (:202), quantise-round-dequantise (:140-144),
 * cv::idct restated as the float64 inverse of the orthonormal DCT-II (X = C^T Y C), cv::merge
 * into interleaved B,G,R.  gaze_w == 0 or gaze_h == 0 means no gaze rectangle.
 * out64: H x W x 3 doubles. */
void svc_oracle_decode_frame(const float* planes, uint32_t w, uint32_t h, uint32_t block_w, uint32_t block_h,
                             const uint32_t* block_types, uint32_t mv_bw, uint32_t mv_bh, uint32_t fg_step,
                             uint32_t bg_step, uint32_t gaze_x, uint32_t gaze_y, uint32_t gaze_w,
                             uint32_t gaze_h, double* out64) {
  double* cw = (double*)malloc(sizeof(double) * block_w * block_w);
  double* ch = (double*)malloc(sizeof(double) * block_h * block_h);
  double* q = (double*)malloc(sizeof(double) * block_w * block_h);
  double* tmp = (double*)malloc(sizeof(double) * block_w * block_h);
  dct_basis(block_w, cw);
  dct_basis(block_h, ch);
  const uint32_t mv_fw = w / mv_bw;
  for (uint32_t ty = 0; ty < h; ty += block_h)
    for (uint32_t tx = 0; tx < w; tx += block_w) {
      const int gazed = gaze_w && gaze_h && tx >= gaze_x && tx < gaze_x + gaze_w && ty >= gaze_y && ty < gaze_y + gaze_h;
      const uint32_t type = block_types[(ty / mv_bh) * mv_fw + tx / mv_bw];
      const uint32_t step = gazed ? 1u : (type == 0 ? bg_step : fg_step); /* decoder.cpp:130-135 */
      for (uint32_t p = 0; p < 3; ++p) {
        const float* plane = planes + (size_t)p * w * h;
        for (uint32_t v = 0; v < block_h; ++v)
          for (uint32_t u = 0; u < block_w; ++u) {
            float c = plane[(size_t)(ty + v) * w + tx + u];
            c = c / (float)step;
            c = roundf(c);
            c = c * (float)step;
            q[v * block_w + u] = (double)c;
          }
        /* rows: tmp[v][x] = sum_u Y[v][u] Cw[u][x] */
        for (uint32_t v = 0; v < block_h; ++v)
          for (uint32_t x = 0; x < block_w; ++x) {
            double acc = 0.0;
            for (uint32_t u = 0; u < block_w; ++u) acc += q[v * block_w + u] * cw[u * block_w + x];
            tmp[v * block_w + x] = acc;
          }
        /* columns: X[y][x] = sum_v Ch[v][y] tmp[v][x] */
        for (uint32_t y = 0; y < block_h; ++y)
          for (uint32_t x = 0; x < block_w; ++x) {
            double acc = 0.0;
            for (uint32_t v = 0; v < block_h; ++v) acc += ch[v * block_h + y] * tmp[v * block_w + x];
            out64[((size_t)(ty + y) * w + tx + x) * 3 + p] = acc;
          }
      }
    }
  free(cw); free(ch); free(q); free(tmp);
}

/* Sum of squared errors between the source frame and the reconstruction rounded to u8
 * (clamp(round half away)), over the top-left region_w x region_h pixels: exact integer. */
uint64_t svc_oracle_sse_frame(const uint8_t* src_bgr, const float* rec_bgr, uint32_t w, uint32_t region_w,
                              uint32_t region_h) {
  uint64_t sse = 0;
  for (uint32_t y = 0; y < region_h; ++y)
    for (uint32_t x = 0; x < region_w; ++x)
      for (uint32_t c = 0; c < 3; ++c) {
        size_t i = ((size_t)y * w + x) * 3 + c;
        float r = roundf(rec_bgr[i]);
        int v = r < 0.f ? 0 : (r > 255.f ? 255 : (int)r);
        int d = (int)src_bgr[i] - v;
        sse += (uint64_t)(d * d);
      }
  return sse;
}
