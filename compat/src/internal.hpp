// compat/src/internal.hpp -- helpers shared by the adapter's translation units (not installed).
#pragma once

#include "opencv2/core/mat.hpp"

namespace cv {
namespace detail {
void Abi(int rc, const char* where);  // a non-zero svc_status aborts with svc_hip_last_error()
Mat Continuous(const Mat& m);         // m itself, or a tightly packed copy of a view
}  // namespace detail
}  // namespace cv
