// ref_shim.cpp -- extern "C" doorway into the UNMODIFIED reference libs/motion.cpp.
//
// TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles /root/reference/libs/motion.cpp
// where it lies (never copied) together with this file into oracle/_ref/libsvc_ref.so.
// This file contains no reference code: it only calls the reference's public
// functions (libs/motion.hpp:100-153) so that Python (ctypes) can reach them.
//
// One test seam: the reference seeds a function-local static engine from
// std::random_device (libs/motion.cpp:186-187), which makes its RANSAC
// unrepeatable.  This library defines std::random_device::_M_getval() itself and
// is linked -Bsymbolic, so inside this .so (only) the "entropy" is the constant
// kRefSeed.  The reference's std::default_random_engine stream is then a pure
// function of how many draws it has made, and svc_ref_ransac_draw() mirrors it
// with the same libstdc++ engine + distribution types, so a test can hand the
// identical accepted sample indices to the restatement in oracle/svc_oracle.c.
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "motion.hpp"  // the reference header, via -I/root/reference/libs

static constexpr unsigned kRefSeed = 0x5C0DEC0Du;

std::random_device::result_type std::random_device::_M_getval() { return kRefSeed; }

namespace {
// Mirror of the reference's function-local statics (motion.cpp:186-187).
std::default_random_engine& MirrorEngine() {
  static std::default_random_engine eng(kRefSeed);
  return eng;
}
}  // namespace

extern "C" {

unsigned svc_ref_seed(void) { return kRefSeed; }

int svc_ref_has_sse2(void) {
#ifdef __SSE2__
  return 1;
#else
  return 0;
#endif
}

void svc_ref_ebma(const uint8_t* tracked, const uint8_t* anchor, uint32_t w,
                  uint32_t h, uint32_t r, uint32_t bw, uint32_t bh, float* mv_xy,
                  float* min_mad) {
  EstimateMotionExhaustiveSearch(tracked, anchor, w, h, r, bw, bh,
                                 reinterpret_cast<Vec2f*>(mv_xy), min_mad);
}

void svc_ref_hbma(const uint8_t* const* tracked_pyr,
                  const uint8_t* const* anchor_pyr, uint32_t levels, uint32_t w,
                  uint32_t h, uint32_t r, uint32_t bw, uint32_t bh, float* mv_xy,
                  float* min_mad) {
  EstimateMotionHierarchical(tracked_pyr, anchor_pyr, levels, w, h, r, bw, bh,
                             reinterpret_cast<Vec2f*>(mv_xy), min_mad);
}

void svc_ref_hbma16_sse2(const uint8_t* const* tracked_pyr,
                         const uint8_t* const* anchor_pyr, uint32_t w, uint32_t h,
                         uint32_t r, float* mv_xy, float* min_mad) {
#ifdef __SSE2__
  EstimateMotionHierarchical16x16Sse2(tracked_pyr, anchor_pyr, w, h, r,
                                      reinterpret_cast<Vec2f*>(mv_xy), min_mad);
#else
  (void)tracked_pyr; (void)anchor_pyr; (void)w; (void)h; (void)r; (void)mv_xy; (void)min_mad;
#endif
}

// The three whole-frame estimators (libs/motion.hpp:38-59); no caller in the reference.
void svc_ref_global_avg(const float* mv_xy, uint32_t n, float* avg_xy) {
  Vec2f a = EstimateGlobalMotionAvg(reinterpret_cast<const Vec2f*>(mv_xy), n);
  avg_xy[0] = a.x;
  avg_xy[1] = a.y;
}

void svc_ref_global_ebma(const uint8_t* tracked, const uint8_t* anchor, uint32_t w, uint32_t h,
                         uint32_t r, float* gm_xy, float* min_mad) {
  EstimateGlobalMotionExhaustiveSearch(tracked, anchor, w, h, r, reinterpret_cast<Vec2f*>(gm_xy), min_mad);
}

void svc_ref_global_hbma(const uint8_t* const* tracked_pyr, const uint8_t* const* anchor_pyr,
                         uint32_t levels, uint32_t w, uint32_t h, uint32_t r, float* gm_xy) {
  EstimateGlobalMotionHierarchical(tracked_pyr, anchor_pyr, levels, w, h, r, reinterpret_cast<Vec2f*>(gm_xy));
}

// Runs the reference RANSAC.  `mv_xy` must hold n + 1 vectors: the reference
// draws indices from [0, n] inclusive (motion.cpp:208) and so may read entry n.
// `gm_xy` is in/out (the reference reads it uninitialised at :241-242).
// `inliers` must hold n entries; returns the inlier count.
uint32_t svc_ref_ransac(const float* mv_xy, uint32_t n, uint32_t subset_sz,
                        float inlier_thresh, float success_prob,
                        float inlier_ratio, float* rmse, float* gm_xy,
                        uint32_t* inliers) {
  RansacParams p;
  p.subset_sz = subset_sz;
  p.inlier_thresh = inlier_thresh;
  p.success_prob = success_prob;
  p.inlier_ratio = inlier_ratio;
  Vec2f gm{gm_xy[0], gm_xy[1]};
  std::vector<uint> idx;
  EstimateGlobalMotionRansac(reinterpret_cast<const Vec2f*>(mv_xy), n, p, rmse, &gm,
                             &idx);
  gm_xy[0] = gm.x;
  gm_xy[1] = gm.y;
  std::memcpy(inliers, idx.data(), idx.size() * sizeof(uint32_t));
  return static_cast<uint32_t>(idx.size());
}

// Advances the mirror engine exactly as ONE call of the reference RANSAC with
// these parameters advances the reference's engine, and returns the accepted
// draws (iter_count * subset_sz, iteration-major).  Call once per svc_ref_ransac
// call, in the same order, to stay in lockstep.
void svc_ref_ransac_draw(uint32_t n, uint32_t subset_sz, uint32_t iter_count,
                         uint32_t* samples) {
  std::uniform_int_distribution<uint> distrib(0, n);  // inclusive, as motion.cpp:208
  auto& eng = MirrorEngine();
  for (uint32_t it = 0; it < iter_count; ++it) {
    uint32_t* s = samples + static_cast<size_t>(it) * subset_sz;
    for (uint32_t i = 0; i < subset_sz; ++i) {
      bool dup;
      do {  // redraw until distinct from the earlier picks of this iteration
        s[i] = distrib(eng);
        dup = false;
        for (uint32_t j = 0; j < i; ++j) dup = dup || (s[j] == s[i]);
      } while (dup);
    }
  }
}

}  // extern "C"
