// compat/opencv2/imgproc.hpp -- the cv:: image-processing functions the reference's encoder calls
// (libs/encoder.cpp:186-187, :449-451, :468-470, :524-527, :607-610), as thin callers of this repo's HIP entry points.
// PRODUCT-SIDE ADAPTER, NOT AN ORACLE: see core/mat.hpp.
#ifndef SVC_COMPAT_OPENCV2_IMGPROC_HPP
#define SVC_COMPAT_OPENCV2_IMGPROC_HPP

#include <vector>

#include "opencv2/core.hpp"

namespace cv {

enum ColorConversionCodes { COLOR_BGR2YUV = 82 };
enum MorphTypes { MORPH_ERODE = 0, MORPH_DILATE = 1, MORPH_OPEN = 2, MORPH_CLOSE = 3 };
enum MorphShapes { MORPH_RECT = 0, MORPH_CROSS = 1, MORPH_ELLIPSE = 2 };
enum ConnectedComponentsAlgorithmsTypes { CCL_WU = 0, CCL_DEFAULT = -1, CCL_GRANA = 1 };

// libs/encoder.cpp:449, :468.  COLOR_BGR2YUV on 8-bit BGR (svc_hip_bgr2yuv_host); every other code aborts with a message.
void cvtColor(const Mat& src, Mat& dst, int code);

// cv::pyrDown: 5x5 Gaussian, BORDER_REFLECT_101, every second sample (svc_hip_build_pyramid_host, one level)
void pyrDown(const Mat& src, Mat& dst);

// libs/encoder.cpp:451, :470.  dst[0] becomes a header over src's storage (as in OpenCV), dst[1..maxlevel] are
// create()d -- existing planes of the right size are reused, which the encoder's cached `data` pointers rely on
// (libs/encoder.cpp:205-218) -- and filled by ONE call of svc_hip_build_pyramid_host.
namespace detail { void BuildPyramidInto(const Mat& src, Mat* const* levels, int maxlevel); }
void buildPyramid(const Mat& src, std::vector<Mat>& dst, int maxlevel);
template <typename T> void buildPyramid(const Mat& src, std::vector<Mat_<T>>& dst, int maxlevel) {
  dst.resize((size_t)maxlevel + 1);
  std::vector<Mat*> p;
  for (auto& m : dst) p.push_back(&m);
  detail::BuildPyramidInto(src, p.data(), maxlevel);
}

// libs/encoder.cpp:186-187: MORPH_RECT only (an all-ones 8-bit matrix)
Mat getStructuringElement(int shape, Size ksize, Point anchor = Point(-1, -1));

// libs/encoder.cpp:524-527 and their single steps, rectangular elements anchored at the centre, one iteration, default
// border (svc_hip_morph_rect_host).  src may be dst.
void morphologyEx(const Mat& src, Mat& dst, int op, const Mat& kernel);
void erode(const Mat& src, Mat& dst, const Mat& kernel);
void dilate(const Mat& src, Mat& dst, const Mat& kernel);

// libs/encoder.cpp:607-610 (svc_hip_connected_components_host): labels CV_32S, components numbered in raster order of
// their first pixel; returns their number INCLUDING the background label, as OpenCV does.
int connectedComponents(const Mat& image, Mat& labels, int connectivity = 8, int ltype = CV_32S, int ccltype = CCL_DEFAULT);

}  // namespace cv

#endif  // SVC_COMPAT_OPENCV2_IMGPROC_HPP
