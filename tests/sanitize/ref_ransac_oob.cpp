// ref_ransac_oob.cpp -- calls the UNMODIFIED reference EstimateGlobalMotionRansac (compiled from /root/reference by
// tests/sanitize/Makefile, never copied) on a heap field of exactly N vectors.  The reference draws sample indices
// from [0, N] inclusive (libs/motion.cpp:208), so sooner or later it reads motion_field[N]: AddressSanitizer reports a
// heap-buffer-overflow.  tests/test_sanitizers.py expects that report -- it is the behaviour the product's RANSAC
// deliberately does not reproduce (include/svc_hip.h).
#include <cstdio>
#include <vector>

#include "motion.hpp"  // the reference header, via -I/root/reference/libs

int main() {
  const uint n = 7;  // small field: a draw of index n comes within a few calls
  RansacParams p;
  p.subset_sz = 3;
  p.inlier_thresh = 7.5f;
  p.success_prob = 0.99f;
  p.inlier_ratio = 0.5f;
  for (int call = 0; call < 2000; ++call) {
    Vec2f* field = new Vec2f[n];
    for (uint i = 0; i < n; ++i) field[i] = Vec2f{(float)i, 1.0f};
    float rmse = 0;
    Vec2f gm{0, 0};
    std::vector<uint> inliers;
    EstimateGlobalMotionRansac(field, n, p, &rmse, &gm, &inliers);
    delete[] field;
  }
  std::puts("no out-of-bounds read in 2000 calls");
  return 0;
}
