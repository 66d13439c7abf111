/*
 * svc_cpu_dct.c -- the CPU BASELINE's transform: an f32 separable 8x8 / 16x16 forward DCT-II with an AVX2 + FMA path.
 * TEST / BASELINE INFRASTRUCTURE ONLY (see svc_oracle.h): bench.py's cpu_baseline leg times it, tests check it against
 * the f64 oracle.  It is NOT the oracle of the transform (that is svc_oracle_dct_frame_f64) and not cv::dct either --
 * OpenCV is not installed, so the reference's own transform cannot be timed; this is the fastest honest stand-in for
 * "what one CPU core needs for libs/encoder.cpp:323-339 + :638" (SURVEY.md 8d "CPU DCT baseline = build's own
 * scalar / AVX2 separable DCT"; round 3's VERDICT, Missing 4).
 *
 * Per tile and channel:  T = C X  (rows of T as sums of scaled rows of X), then  Y = T C^T  (rows of Y as sums of scaled
 * rows of C^T): both passes are "vector += scalar * vector" over N contiguous floats, which GCC turns into 8-wide FMAs in
 * the target("avx2,fma") instantiation (picked at run time with __builtin_cpu_supports; the other one is baseline
 * x86-64 = SSE2).
 * u8 -> f32 conversion and the B,G,R de-interleave (cv::split) are part of the timed work, as in the reference.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "svc_oracle.h"

static void basis(int n, float* c, float* ct) { /* c[k][m] = s_k cos(pi (2m + 1) k / 2n); ct = its transpose */
  for (int k = 0; k < n; ++k)
    for (int m = 0; m < n; ++m) {
      const double s = k == 0 ? sqrt(1.0 / n) : sqrt(2.0 / n);
      const float v = (float)(s * cos(M_PI * (2 * m + 1) * k / (2.0 * n)));
      c[k * n + m] = v;
      ct[m * n + k] = v;
    }
}

/* One band of N rows: de-interleave B,G,R into three f32 strips (x - 128: halves the magnitudes the f32 sums carry; the
 * DCT is linear, so the offset comes back as 128 N on each tile's DC term), then tile by tile and channel by channel
 * T = C X, Y = T C^T.  Instantiated twice: for baseline x86-64 and with target("avx2,fma"). */
#define DCT_BAND(N, SUFFIX, ATTR)                                                                                  \
  ATTR static void dct_band_##N##SUFFIX(const uint8_t* restrict bgr, uint32_t w, const float* restrict c,          \
                                        const float* restrict ct, float* restrict strip, float* restrict planes,   \
                                        size_t plane) {                                                            \
    const size_t n = (size_t)w * N;                                                                                \
    float* restrict sb = strip;                                                                                    \
    float* restrict sg = strip + n;                                                                                \
    float* restrict sr = strip + 2 * n;                                                                            \
    for (size_t i = 0; i < n; ++i) {                                                                               \
      sb[i] = (float)((int)bgr[3 * i] - 128);                                                                      \
      sg[i] = (float)((int)bgr[3 * i + 1] - 128);                                                                  \
      sr[i] = (float)((int)bgr[3 * i + 2] - 128);                                                                  \
    }                                                                                                              \
    for (int ch = 0; ch < 3; ++ch) {                                                                               \
      const float* restrict xs = strip + (size_t)ch * n;                                                           \
      float* restrict out = planes + (size_t)ch * plane;                                                           \
      for (uint32_t x0 = 0; x0 < w; x0 += N) {                                                                     \
        float t[N][N];                                                                                             \
        for (int v = 0; v < N; ++v) {                                                                              \
          float acc[N];                                                                                            \
          for (int u = 0; u < N; ++u) acc[u] = 0.0f;                                                               \
          for (int m = 0; m < N; ++m) {                                                                            \
            const float s = c[v * N + m];                                                                          \
            const float* restrict row = xs + (size_t)m * w + x0;                                                   \
            for (int u = 0; u < N; ++u) acc[u] += s * row[u];                                                      \
          }                                                                                                        \
          for (int u = 0; u < N; ++u) t[v][u] = acc[u];                                                            \
        }                                                                                                          \
        for (int v = 0; v < N; ++v) {                                                                              \
          float acc[N];                                                                                            \
          for (int u = 0; u < N; ++u) acc[u] = 0.0f;                                                               \
          for (int k = 0; k < N; ++k) {                                                                            \
            const float s = t[v][k];                                                                               \
            const float* restrict row = ct + k * N;                                                                \
            for (int u = 0; u < N; ++u) acc[u] += s * row[u];                                                      \
          }                                                                                                        \
          float* restrict o = out + (size_t)v * w + x0;                                                            \
          for (int u = 0; u < N; ++u) o[u] = acc[u];                                                               \
        }                                                                                                          \
        out[x0] += 128.0f * N; /* the offset's DC term: sqrt(1/N) sqrt(1/N) * 128 * N * N */                       \
      }                                                                                                            \
    }                                                                                                              \
  }
#define NOATTR
DCT_BAND(8, _base, NOATTR)
DCT_BAND(16, _base, NOATTR)

/* The AVX2 + FMA form of the same two passes with the rows held in registers: a tile row is one (8x8) or two (16x16)
 * 8-float vectors, a coefficient is broadcast from memory, every product is one vfmadd. */
#define AVX2 __attribute__((target("avx2,fma")))

AVX2 static inline void tile8_avx2(const float* restrict x, size_t xs, const float* restrict c, const float* restrict ct,
                                   float* restrict y, size_t ys) {
  __m256 r[8], cr[8];
  for (int m = 0; m < 8; ++m) { r[m] = _mm256_loadu_ps(x + m * xs); cr[m] = _mm256_load_ps(ct + 8 * m); }
  float t[8][8] __attribute__((aligned(32)));
  for (int v = 0; v < 8; ++v) {
    __m256 acc = _mm256_mul_ps(_mm256_broadcast_ss(c + 8 * v), r[0]);
    for (int m = 1; m < 8; ++m) acc = _mm256_fmadd_ps(_mm256_broadcast_ss(c + 8 * v + m), r[m], acc);
    _mm256_store_ps(t[v], acc);
  }
  for (int v = 0; v < 8; ++v) {
    __m256 acc = _mm256_mul_ps(_mm256_broadcast_ss(&t[v][0]), cr[0]);
    for (int k = 1; k < 8; ++k) acc = _mm256_fmadd_ps(_mm256_broadcast_ss(&t[v][k]), cr[k], acc);
    _mm256_storeu_ps(y + v * ys, acc);
  }
}

AVX2 static inline void tile16_avx2(const float* restrict x, size_t xs, const float* restrict c, const float* restrict ct,
                                    float* restrict y, size_t ys) {
  float t[16][16] __attribute__((aligned(32)));
  for (int v = 0; v < 16; ++v) {
    __m256 a0 = _mm256_setzero_ps(), a1 = _mm256_setzero_ps();
    for (int m = 0; m < 16; ++m) {
      const __m256 s = _mm256_broadcast_ss(c + 16 * v + m);
      a0 = _mm256_fmadd_ps(s, _mm256_loadu_ps(x + m * xs), a0);
      a1 = _mm256_fmadd_ps(s, _mm256_loadu_ps(x + m * xs + 8), a1);
    }
    _mm256_store_ps(t[v], a0);
    _mm256_store_ps(t[v] + 8, a1);
  }
  for (int v = 0; v < 16; ++v) {
    __m256 a0 = _mm256_setzero_ps(), a1 = _mm256_setzero_ps();
    for (int k = 0; k < 16; ++k) {
      const __m256 s = _mm256_broadcast_ss(&t[v][k]);
      a0 = _mm256_fmadd_ps(s, _mm256_load_ps(ct + 16 * k), a0);
      a1 = _mm256_fmadd_ps(s, _mm256_load_ps(ct + 16 * k + 8), a1);
    }
    _mm256_storeu_ps(y + v * ys, a0);
    _mm256_storeu_ps(y + v * ys + 8, a1);
  }
}

#define DCT_BAND_AVX2(N)                                                                                           \
  AVX2 static void dct_band_##N##_avx2(const uint8_t* restrict bgr, uint32_t w, const float* restrict c,           \
                                       const float* restrict ct, float* restrict strip, float* restrict planes,    \
                                       size_t plane) {                                                             \
    const size_t n = (size_t)w * N;                                                                                \
    float* restrict sb = strip;                                                                                    \
    float* restrict sg = strip + n;                                                                                \
    float* restrict sr = strip + 2 * n;                                                                            \
    for (size_t i = 0; i < n; ++i) {                                                                               \
      sb[i] = (float)((int)bgr[3 * i] - 128);                                                                      \
      sg[i] = (float)((int)bgr[3 * i + 1] - 128);                                                                  \
      sr[i] = (float)((int)bgr[3 * i + 2] - 128);                                                                  \
    }                                                                                                              \
    for (int ch = 0; ch < 3; ++ch) {                                                                               \
      const float* xs = strip + (size_t)ch * n;                                                                    \
      float* out = planes + (size_t)ch * plane;                                                                    \
      for (uint32_t x0 = 0; x0 < w; x0 += N) {                                                                     \
        tile##N##_avx2(xs + x0, w, c, ct, out + x0, w);                                                            \
        out[x0] += 128.0f * N;                                                                                     \
      }                                                                                                            \
    }                                                                                                              \
  }
DCT_BAND_AVX2(8)
DCT_BAND_AVX2(16)

static int has_avx2(void) {
  __builtin_cpu_init();
  return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
}

/* libs/encoder.cpp:638 (convertTo) + :323-339 (Dct: cv::split + cv::dct per tile) for block = 8 or 16; planes32 gets
 * 3 planar h x w floats in B, G, R order.  Returns 0, or 1 for another block size (the f64 oracle covers those). */
int svc_cpu_dct_frame_f32(const uint8_t* bgr, uint32_t w, uint32_t h, uint32_t block, float* planes32) {
  enum { kMaxW = 8192 };
  if ((block != 8 && block != 16) || w % block || h % block || w > kMaxW) return 1;
  static float c8[64] __attribute__((aligned(32))), ct8[64] __attribute__((aligned(32))), c16[256] __attribute__((aligned(32))),
      ct16[256] __attribute__((aligned(32)));
  static int ready = 0;
  if (!ready) { basis(8, c8, ct8); basis(16, c16, ct16); ready = 1; } /* idempotent: a race only repeats the same stores */
  static _Thread_local float strip[3 * 16 * kMaxW];
  const size_t plane = (size_t)w * h;
  const int avx2 = has_avx2();
  for (uint32_t y0 = 0; y0 < h; y0 += block) {
    const uint8_t* src = bgr + (size_t)y0 * w * 3;
    float* out = planes32 + (size_t)y0 * w;
    if (block == 8) {
      if (avx2) dct_band_8_avx2(src, w, c8, ct8, strip, out, plane);
      else dct_band_8_base(src, w, c8, ct8, strip, out, plane);
    } else {
      if (avx2) dct_band_16_avx2(src, w, c16, ct16, strip, out, plane);
      else dct_band_16_base(src, w, c16, ct16, strip, out, plane);
    }
  }
  return 0;
}

/* libs/decoder.cpp:130-144 over a planar frame, as svc_oracle_quant_frame (same bits: tests/test_cpu_baseline.py), at the
 * speed a CPU deployment would run it: per row segment of one MV block the step is constant, c / step -> round half away
 * from zero -> * step, eight coefficients per instruction in the AVX2 form.  std::round(q) = trunc(q + copysign(0.5 - 2^-25,
 * q)) for every float q (the sum is exact below 2^23 except at the tie 0.5 - 2^-25 + 0.5, which rounds to even = 1). */
AVX2 static void quant_run_avx2(float* restrict p, uint32_t n, float step) {
  const __m256 vs = _mm256_set1_ps(step), half = _mm256_set1_ps(0.49999997f), sign = _mm256_set1_ps(-0.0f);
  uint32_t i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m256 q = _mm256_div_ps(_mm256_loadu_ps(p + i), vs);
    const __m256 r = _mm256_round_ps(_mm256_add_ps(q, _mm256_or_ps(half, _mm256_and_ps(q, sign))), _MM_FROUND_TO_ZERO | _MM_FROUND_NO_EXC);
    _mm256_storeu_ps(p + i, _mm256_mul_ps(r, vs));
  }
  for (; i < n; ++i) p[i] = roundf(p[i] / step) * step;
}

static void quant_run_base(float* restrict p, uint32_t n, float step) {
  for (uint32_t i = 0; i < n; ++i) p[i] = roundf(p[i] / step) * step;
}

void svc_cpu_quant_frame_f32(float* planes, uint32_t w, uint32_t h, uint32_t mv_bw, uint32_t mv_bh, const uint32_t* block_types,
                             uint32_t fg_step, uint32_t bg_step) {
  const int avx2 = has_avx2();
  const uint32_t mfw = w / mv_bw;
  for (int ch = 0; ch < 3; ++ch)
    for (uint32_t y = 0; y < h; ++y) {
      float* row = planes + ((size_t)ch * h + y) * w;
      const uint32_t* trow = block_types + (size_t)(y / mv_bh) * mfw;
      for (uint32_t bx = 0; bx < mfw; ++bx) {
        const float step = (float)(trow[bx] == 0 ? bg_step : fg_step); /* libs/decoder.cpp:130-135, :141 */
        if (avx2) quant_run_avx2(row + (size_t)bx * mv_bw, mv_bw, step);
        else quant_run_base(row + (size_t)bx * mv_bw, mv_bw, step);
      }
    }
}

/* which instantiation runs on this machine: 2 = avx2 + fma, 0 = baseline x86-64 (SSE2) */
int svc_cpu_dct_isa(void) { return has_avx2() ? 2 : 0; }
