// motion_hip.cpp -- the reference's C++ entry points (include/svc/motion.hpp) as thin
// callers of the C ABI (include/svc_hip.h).  No arithmetic of the hot path lives
// here except RANSAC's sample drawing, which is host-side by nature.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "svc/motion.hpp"
#include "svc_hip.h"

namespace {

[[noreturn]] void Die(const char* what, int rc) {
  std::fprintf(stderr, "svc-hip: %s failed (status %d): %s\n", what, rc, svc_hip_last_error());
  std::abort();
}

struct RansacRng {
  std::default_random_engine engine;  // the reference's engine type (motion.cpp:187)
  bool seeded = false;
};

RansacRng& Rng() {
  static thread_local RansacRng rng;
  if (!rng.seeded) {
    std::random_device rd;
    rng.engine.seed(rd());
    rng.seeded = true;
  }
  return rng;
}

}  // namespace

namespace {
thread_local bool g_literal_global_search = false;
}

void SvcReferenceLiteralGlobalSearch(bool on) { g_literal_global_search = on; }

void SvcSeedRansac(uint seed) {
  static_cast<void>(Rng());
  Rng().engine.seed(seed);
}

void EstimateMotionHierarchical(const uchar* const* tracked_pyramid, const uchar* const* anchor_pyramid,
                                uint level_count, uint frame_w, uint frame_h, uint search_range,
                                uint block_w, uint block_h, Vec2f* motion_field, float* min_mad) {
  int rc = svc_hip_hbma_host(tracked_pyramid, anchor_pyramid, level_count, frame_w, frame_h, search_range,
                             block_w, block_h, reinterpret_cast<float*>(motion_field), min_mad, SVC_HBMA_AUTO);
  if (rc) Die("EstimateMotionHierarchical", rc);
}

void EstimateMotionHierarchical16x16Sse2(const uchar* const* tracked_pyramid,
                                         const uchar* const* anchor_pyramid, uint frame_w, uint frame_h,
                                         uint search_range, Vec2f* mv_field, float* min_mad) {
  int rc = svc_hip_hbma_host(tracked_pyramid, anchor_pyramid, 4, frame_w, frame_h, search_range, 16, 16,
                             reinterpret_cast<float*>(mv_field), min_mad, SVC_HBMA_AUTO);
  if (rc) Die("EstimateMotionHierarchical16x16Sse2", rc);
}

void EstimateMotionExhaustiveSearch(const uchar* tracked_frame, const uchar* anchor_frame, uint frame_w,
                                    uint frame_h, uint search_range, uint block_w, uint block_h,
                                    Vec2f* motion_field, float* min_mad) {
  int rc = svc_hip_ebma_host(tracked_frame, anchor_frame, frame_w, frame_h, search_range, block_w, block_h,
                             reinterpret_cast<float*>(motion_field), min_mad);
  if (rc) Die("EstimateMotionExhaustiveSearch", rc);
}

void EstimateGlobalMotionRansac(const Vec2f* motion_field, uint motion_field_sz, RansacParams params,
                                float* rmse, Vec2f* global_motion, std::vector<uint>* inlier_indices) {
  svc_ransac_params p{params.subset_sz, params.inlier_thresh, params.success_prob, params.inlier_ratio};
  const uint iters = svc_hip_ransac_iter_count(p);
  // distinct indices per iteration by rejection, as motion.cpp:211-220, from [0, N-1]
  std::vector<uint> samples(static_cast<size_t>(iters) * p.subset_sz);
  if (motion_field_sz > 0) {
    std::uniform_int_distribution<uint> pick(0, motion_field_sz - 1);
    auto& eng = Rng().engine;
    for (uint it = 0; it < iters; ++it) {
      uint* s = samples.data() + static_cast<size_t>(it) * p.subset_sz;
      for (uint i = 0; i < p.subset_sz; ++i) {
        bool again;
        do {
          s[i] = pick(eng);
          again = false;
          for (uint j = 0; j < i; ++j) again = again || s[j] == s[i];
        } while (again);
      }
    }
  }
  std::vector<uint> inliers(motion_field_sz ? motion_field_sz : 1);
  uint count = 0;
  float gm[2] = {global_motion->x, global_motion->y};
  int rc = svc_hip_ransac_host(reinterpret_cast<const float*>(motion_field), motion_field_sz, p, samples.data(),
                               iters, gm, rmse, inliers.data(), &count);
  if (rc) Die("EstimateGlobalMotionRansac", rc);
  global_motion->x = gm[0];
  global_motion->y = gm[1];
  inliers.resize(count);
  inlier_indices->swap(inliers);  // motion.cpp:265
}

Vec2f EstimateGlobalMotionAvg(const Vec2f* motion_field, uint sz) {
  Vec2f avg{0.0f, 0.0f};
  if (sz == 0) return avg;  // motion.cpp:48: the loop does not run
  int rc = svc_hip_global_avg_host(reinterpret_cast<const float*>(motion_field), sz, &avg.x);
  if (rc) Die("EstimateGlobalMotionAvg", rc);
  return avg;
}

void EstimateGlobalMotionExhaustiveSearch(const uchar* tracked_frame, const uchar* anchor_frame, uint frame_w,
                                          uint frame_h, uint search_range, Vec2f* global_motion, float* min_mad) {
  if (g_literal_global_search && search_range > 0) {
    // what the reference's loops leave behind when they do not run (`int dy <= uint search_range`, motion.cpp:66-67, :72)
    *global_motion = Vec2f{0.0f, 0.0f};
    *min_mad = std::numeric_limits<float>::max();
    return;
  }
  int rc = svc_hip_global_ebma_host(tracked_frame, anchor_frame, frame_w, frame_h, search_range,
                                    reinterpret_cast<float*>(global_motion), min_mad);
  if (rc) Die("EstimateGlobalMotionExhaustiveSearch", rc);
}

void EstimateGlobalMotionHierarchical(const uchar* const* tracked_pyramid, const uchar* const* anchor_pyramid,
                                      uint num_levels, uint base_frame_w, uint base_frame_h, uint base_search_range,
                                      Vec2f* global_motion) {
  if (g_literal_global_search) {
    // every level's search either does not run (range > 0) or sees the single candidate (0, 0): motion.cpp:118-141
    *global_motion = Vec2f{0.0f, 0.0f};
    return;
  }
  int rc = svc_hip_global_hbma_host(tracked_pyramid, anchor_pyramid, num_levels, base_frame_w, base_frame_h,
                                    base_search_range, reinterpret_cast<float*>(global_motion));
  if (rc) Die("EstimateGlobalMotionHierarchical", rc);
}

void Dct(const uchar* bgr, uint frame_w, uint frame_h, uint block_w, uint block_h, float* const planes[3]) {
  int rc = svc_hip_dct_planes_host(bgr, frame_w, frame_h, block_w, block_h, planes);  // D2H lands in the caller's planes
  if (rc) Die("Dct", rc);
}

void QuantizeDequantize(float* coeffs, unsigned long long count, uint quant_step) {
  int rc = svc_hip_quant_host(coeffs, count, quant_step);
  if (rc) Die("QuantizeDequantize", rc);
}
