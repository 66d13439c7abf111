// compat/src/internal.hpp -- helpers shared by the adapter's translation units (not installed).
#pragma once

#include <functional>

#include "opencv2/core/mat.hpp"

namespace cv {
namespace detail {
void Abi(int rc, const char* where);  // a non-zero svc_status aborts with svc_hip_last_error()
Mat Continuous(const Mat& m);         // m itself, or a tightly packed copy of a view

// job(y0, y1) over the rows [0, rows) of a host-side pass that moves `bytes` in all: by a few threads above 1 MB (the reference's
// frame loop moves 150 MB per 1080p frame through such passes: clone, convertTo, split ...), by the caller alone below.
void ParallelRows(int rows, size_t bytes, const std::function<void(int, int)>& job);

// SVC_COMPAT_PROFILE=1 in the environment: wall time per adapter call, summed over the run, printed to stderr at exit --
// where an application written against this adapter spends its host time (the reference's encoder is host-bound).
struct Timed {
  const char* name;
  double t0;
  explicit Timed(const char* n);
  ~Timed();
};
}  // namespace detail
}  // namespace cv
