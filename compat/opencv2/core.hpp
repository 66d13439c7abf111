// compat/opencv2/core.hpp -- the cv:: core functions the reference's encoder calls (libs/encoder.cpp), as thin callers
// of this repo's HIP entry points (include/svc_hip.h).  PRODUCT-SIDE ADAPTER, NOT AN ORACLE: see core/mat.hpp.
#ifndef SVC_COMPAT_OPENCV2_CORE_HPP
#define SVC_COMPAT_OPENCV2_CORE_HPP

#include <vector>

#include "opencv2/core/mat.hpp"

namespace cv {

enum BorderTypes { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4,
                   BORDER_REFLECT101 = BORDER_REFLECT_101, BORDER_DEFAULT = BORDER_REFLECT_101 };
enum KmeansFlags { KMEANS_RANDOM_CENTERS = 0, KMEANS_PP_CENTERS = 2, KMEANS_USE_INITIAL_LABELS = 1 };
enum DftFlags { DCT_INVERSE = 1, DCT_ROWS = 4 };

// libs/encoder.cpp:447, :459: pads the source frame to the MV-block grid.  Data movement only (host); BORDER_CONSTANT.
void copyMakeBorder(const Mat& src, Mat& dst, int top, int bottom, int left, int right, int borderType,
                    const Scalar& value = Scalar());

// libs/encoder.cpp:450, :469: one channel of an interleaved matrix.  Data movement only (host).
void extractChannel(const Mat& src, Mat& dst, int coi);

// libs/encoder.cpp:328: interleaved -> planar.  Data movement only (host).  Existing planes of the right size and
// type are reused, like OpenCV's OutputArrayOfArrays::create.
namespace detail { void SplitInto(const Mat& src, Mat* const* planes, int n); }
void split(const Mat& src, std::vector<Mat>& mv);
template <typename T> void split(const Mat& src, std::vector<Mat_<T>>& mv) {
  mv.resize((size_t)src.channels());
  std::vector<Mat*> p;
  for (auto& m : mv) p.push_back(&m);
  detail::SplitInto(src, p.data(), (int)p.size());
}

// libs/encoder.cpp:335: the forward orthonormal DCT-II (flags 0) of a single-channel f32 matrix of even size -- in the
// reference always a transform-block view of a plane, in place.  The call is COLLECTED on the storage of `dst` and runs
// with every other pending one in a single GPU launch (svc_hip_dct_tiles_host) when the data is next looked at; see
// detail::Buffer in core/mat.hpp.  A matrix over caller-owned memory is transformed at once.
void dct(const Mat& src, Mat& dst, int flags = 0);

// cv::theRNG(): OpenCV's per-thread multiply-with-carry generator.  cv::kmeans takes its seed from it.
struct RNG {
  uint64_t state;
  RNG() : state(0xffffffff) {}
  explicit RNG(uint64_t s) : state(s ? s : 0xffffffff) {}
  unsigned next() {
    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  operator unsigned() { return next(); }
};
RNG& theRNG();
void setRNGSeed(int seed);

// libs/encoder.cpp:575-576: k-means++ seeded clustering of the foreground features, by this repo's deterministic
// definition (include/svc_hip.h, svc_hip_kmeans_host; oracle/svc_segment.c): data is N points of up to four f32
// coordinates holding integers (N x 1 of up to 4 channels, or N x d single-channel); KMEANS_PP_CENTERS and a
// COUNT | EPS criterion only.  The seed is theRNG().state, and theRNG() advances by one step per call -- so a
// process that makes the same calls produces the same labels, like OpenCV with a fixed RNG seed.
double kmeans(const Mat& data, int K, Mat& bestLabels, TermCriteria criteria, int attempts, int flags);

}  // namespace cv

#endif  // SVC_COMPAT_OPENCV2_CORE_HPP
