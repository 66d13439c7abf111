// compat/src/imgproc.cpp -- compat/opencv2/imgproc.hpp on top of include/svc_hip.h.
// PRODUCT-SIDE ADAPTER, NOT AN ORACLE (see compat/opencv2/core/mat.hpp).
#include <vector>

#include "internal.hpp"
#include "opencv2/imgproc.hpp"
#include "svc_hip.h"

namespace cv {

void cvtColor(const Mat& src, Mat& dst, int code) {
  detail::Timed timed("cv::cvtColor (GPU)");
  if (code != COLOR_BGR2YUV) detail::Fail("cv::cvtColor", "only COLOR_BGR2YUV (libs/encoder.cpp:449, :468)");
  if (src.empty() || src.type() != CV_8UC3) detail::Fail("cv::cvtColor", "COLOR_BGR2YUV takes an 8-bit 3-channel matrix");
  const Mat in = detail::Continuous(src);
  Mat out = dst.data == src.data ? Mat() : dst;
  out.create(src.rows, src.cols, CV_8UC3);
  out.sync();
  if (out.isContinuous()) {
    detail::Abi(svc_hip_bgr2yuv_host(in.data, (uint32_t)in.cols, (uint32_t)in.rows, out.data), "cv::cvtColor");
  } else {
    Mat tmp(src.rows, src.cols, CV_8UC3);
    detail::Abi(svc_hip_bgr2yuv_host(in.data, (uint32_t)in.cols, (uint32_t)in.rows, tmp.data), "cv::cvtColor");
    tmp.copyTo(out);
  }
  dst = out;
}

void detail::BuildPyramidInto(const Mat& src, Mat* const* levels, int maxlevel) {
  Timed timed("cv::buildPyramid (GPU)");
  if (src.empty() || src.type() != CV_8UC1) Fail("cv::buildPyramid", "only 8-bit single-channel matrices (libs/encoder.cpp:451)");
  if (maxlevel < 0 || maxlevel > 15) Fail("cv::buildPyramid", "maxlevel out of range");
  const uint32_t f = 1u << maxlevel;
  if ((uint32_t)src.cols % f || (uint32_t)src.rows % f)
    Fail("cv::buildPyramid", "the size must be divisible by 2^maxlevel (the encoder pads to that, libs/encoder.cpp:164-168)");
  src.sync();
  *levels[0] = src;  // OpenCV: `_dst.getMatRef(0) = src` -- level 0 shares the source's storage
  std::vector<uint8_t*> ptrs((size_t)maxlevel + 1, nullptr);
  std::vector<Mat> packed((size_t)maxlevel + 1);
  for (int l = 1; l <= maxlevel; ++l) {
    levels[l]->create(src.rows >> l, src.cols >> l, CV_8UC1);
    levels[l]->sync();
    packed[(size_t)l] = levels[l]->isContinuous() ? *levels[l] : Mat(src.rows >> l, src.cols >> l, CV_8UC1);
    ptrs[(size_t)l] = packed[(size_t)l].data;
  }
  if (maxlevel == 0) return;
  const Mat in = Continuous(src);
  Abi(svc_hip_build_pyramid_host(in.data, (uint32_t)in.cols, (uint32_t)in.rows, (uint32_t)maxlevel + 1, ptrs.data()), "cv::buildPyramid");
  for (int l = 1; l <= maxlevel; ++l)
    if (packed[(size_t)l].data != levels[l]->data) packed[(size_t)l].copyTo(*levels[l]);
}

void buildPyramid(const Mat& src, std::vector<Mat>& dst, int maxlevel) {
  dst.resize((size_t)maxlevel + 1);
  std::vector<Mat*> p;
  for (auto& m : dst) p.push_back(&m);
  detail::BuildPyramidInto(src, p.data(), maxlevel);
}

void pyrDown(const Mat& src, Mat& dst) {
  if (src.empty() || src.type() != CV_8UC1 || src.cols % 2 || src.rows % 2)
    detail::Fail("cv::pyrDown", "only 8-bit single-channel matrices of even size");
  Mat out = dst.data == src.data ? Mat() : dst;
  Mat level0 = src;
  Mat* levels[2] = {&level0, &out};
  detail::BuildPyramidInto(src, levels, 1);
  dst = out;
}

Mat getStructuringElement(int shape, Size ksize, Point) {
  if (shape != MORPH_RECT) detail::Fail("cv::getStructuringElement", "only MORPH_RECT (libs/encoder.cpp:186-187)");
  if (ksize.width < 1 || ksize.height < 1) detail::Fail("cv::getStructuringElement", "empty element");
  Mat k(ksize.height, ksize.width, CV_8UC1);
  k.setTo(Scalar(1.0));
  return k;
}

static void Morph(const Mat& src, Mat& dst, uint32_t op, const Mat& kernel, const char* who) {
  detail::Timed timed("cv::morphologyEx (GPU)");
  if (src.empty() || src.type() != CV_8UC1) detail::Fail(who, "only 8-bit single-channel matrices (libs/encoder.cpp:524-527)");
  if (kernel.empty() || kernel.type() != CV_8UC1) detail::Fail(who, "the structuring element must be an 8-bit matrix");
  kernel.sync();
  for (int y = 0; y < kernel.rows; ++y)
    for (int x = 0; x < kernel.cols; ++x)
      if (!kernel.data[(size_t)y * kernel.step + x]) detail::Fail(who, "only rectangular (all non-zero) structuring elements");
  const Mat in = detail::Continuous(src);
  Mat out = dst;
  out.create(src.rows, src.cols, CV_8UC1);  // src == dst: in place, as the reference calls it
  out.sync();
  Mat packed = out.isContinuous() ? out : Mat(src.rows, src.cols, CV_8UC1);
  detail::Abi(svc_hip_morph_rect_host(in.data, (uint32_t)in.cols, (uint32_t)in.rows, (uint32_t)kernel.cols, (uint32_t)kernel.rows, op,
                                      packed.data), who);
  if (packed.data != out.data) packed.copyTo(out);
  dst = out;
}

void morphologyEx(const Mat& src, Mat& dst, int op, const Mat& kernel) {
  if (op < MORPH_ERODE || op > MORPH_CLOSE) detail::Fail("cv::morphologyEx", "only MORPH_ERODE, _DILATE, _OPEN, _CLOSE");
  Morph(src, dst, (uint32_t)op, kernel, "cv::morphologyEx");
}
void erode(const Mat& src, Mat& dst, const Mat& kernel) { Morph(src, dst, SVC_MORPH_ERODE, kernel, "cv::erode"); }
void dilate(const Mat& src, Mat& dst, const Mat& kernel) { Morph(src, dst, SVC_MORPH_DILATE, kernel, "cv::dilate"); }

int connectedComponents(const Mat& image, Mat& labels, int connectivity, int ltype, int) {
  detail::Timed timed("cv::connectedComponents (GPU)");
  if (image.empty() || image.type() != CV_8UC1) detail::Fail("cv::connectedComponents", "only 8-bit single-channel images");
  if (ltype != CV_32S) detail::Fail("cv::connectedComponents", "only CV_32S labels (libs/encoder.cpp:609)");
  const Mat in = detail::Continuous(image);
  Mat out = labels;
  out.create(image.rows, image.cols, CV_32SC1);
  out.sync();
  Mat packed = out.isContinuous() ? out : Mat(image.rows, image.cols, CV_32SC1);
  uint32_t count = 0;
  detail::Abi(svc_hip_connected_components_host(in.data, (uint32_t)in.cols, (uint32_t)in.rows, (uint32_t)connectivity,
                                                reinterpret_cast<int32_t*>(packed.data), &count), "cv::connectedComponents");
  if (packed.data != out.data) packed.copyTo(out);
  labels = out;
  return (int)count;
}

}  // namespace cv
