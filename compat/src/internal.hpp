// compat/src/internal.hpp -- helpers shared by the adapter's translation units (not installed).
#pragma once

#include "opencv2/core/mat.hpp"

namespace cv {
namespace detail {
void Abi(int rc, const char* where);  // a non-zero svc_status aborts with svc_hip_last_error()
Mat Continuous(const Mat& m);         // m itself, or a tightly packed copy of a view

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
