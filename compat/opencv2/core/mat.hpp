// compat/opencv2/core/mat.hpp -- the slice of OpenCV's matrix types that the reference's encoder uses
// (/root/reference libs/encoder.hpp:5, :42, :55-94; libs/encoder.cpp; apps/encoder.cpp:126-133), so that those files
// compile UNCHANGED against this directory and run their arithmetic on the MI355X through include/svc_hip.h.
//
// PRODUCT-SIDE ADAPTER, NOT AN ORACLE.  Nothing here is OpenCV and nothing here may ever be used to check parity with
// OpenCV: the cv:: functions of this directory are thin callers of this repo's HIP entry points, so comparing the
// kernels with them would compare the kernels with themselves.  oracle/ and tests/golden/ never include this
// directory (tests/test_abi.py checks that).  What it pins is the INTEGRATION: the reference's own Encoder class and
// main() drive the GPU path without an edit.
//
// Scope: exactly the types, members and overloads libs/encoder.{hpp,cpp} and apps/encoder.cpp (non-VISUALIZE) touch,
// plus what they imply (copy semantics, ROI views, create() reuse).  Anything else OpenCV offers is absent on purpose.
#ifndef SVC_COMPAT_OPENCV2_CORE_MAT_HPP
#define SVC_COMPAT_OPENCV2_CORE_MAT_HPP

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace cv {

typedef unsigned char uchar;
typedef unsigned short ushort;
typedef std::string String;

// ---- type codes (OpenCV's encoding: depth in the low 3 bits, channels - 1 above) -------------------------------------
#define CV_CN_SHIFT 3
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAT_DEPTH(flags) ((flags) & 7)
#define CV_MAT_CN(flags) ((((flags) >> CV_CN_SHIFT) & 511) + 1)
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

// ---- small value types -----------------------------------------------------------------------------------------------
template <typename T, int N> struct Vec {
  T val[N];
  Vec() { for (int i = 0; i < N; ++i) val[i] = T(); }
  Vec(T a, T b) : Vec() { static_assert(N >= 2, ""); val[0] = a; val[1] = b; }
  Vec(T a, T b, T c) : Vec() { static_assert(N >= 3, ""); val[0] = a; val[1] = b; val[2] = c; }
  Vec(T a, T b, T c, T d) : Vec() { static_assert(N >= 4, ""); val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
typedef Vec<uchar, 3> Vec3b;
typedef Vec<int, 2> Vec2i;
typedef Vec<float, 2> Vec2f;
typedef Vec<float, 3> Vec3f;
typedef Vec<float, 4> Vec4f;

template <typename T> struct Scalar_ : public Vec<T, 4> {
  Scalar_() {}
  Scalar_(T a) { this->val[0] = a; }
  Scalar_(T a, T b, T c = T(), T d = T()) : Vec<T, 4>(a, b, c, d) {}
  static Scalar_ all(T v) { return Scalar_(v, v, v, v); }
};
typedef Scalar_<double> Scalar;

template <typename T> struct Point_ {
  T x, y;
  Point_() : x(), y() {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point2i;
typedef Point2i Point;

template <typename T> struct Size_ {
  T width, height;
  Size_() : width(), height() {}
  Size_(T w, T h) : width(w), height(h) {}
  bool operator==(const Size_& o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size_& o) const { return !(*this == o); }
};
typedef Size_<int> Size2i;
typedef Size2i Size;

template <typename T> struct Rect_ {
  T x, y, width, height;
  Rect_() : x(), y(), width(), height() {}
  Rect_(T x_, T y_, T w, T h) : x(x_), y(y_), width(w), height(h) {}
  Rect_(const Point_<T>& p, const Size_<T>& s) : x(p.x), y(p.y), width(s.width), height(s.height) {}
};
typedef Rect_<int> Rect2i;
typedef Rect2i Rect;

struct TermCriteria {
  enum Type { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
  int type, maxCount;
  double epsilon;
  TermCriteria() : type(0), maxCount(0), epsilon(0) {}
  TermCriteria(int t, int n, double e) : type(t), maxCount(n), epsilon(e) {}
};

// depth / channel count of an element type (OpenCV's DataType<> traits)
template <typename T> struct DataType;
template <> struct DataType<uchar> { enum { depth = CV_8U, channels = 1 }; };
template <> struct DataType<int> { enum { depth = CV_32S, channels = 1 }; };
template <> struct DataType<float> { enum { depth = CV_32F, channels = 1 }; };
template <> struct DataType<double> { enum { depth = CV_64F, channels = 1 }; };
template <typename T, int N> struct DataType<Vec<T, N>> { enum { depth = DataType<T>::depth, channels = N }; };

namespace detail {

[[noreturn]] void Fail(const char* where, const char* what);  // message to stderr, std::abort() -- OpenCV would throw cv::Exception

// The storage behind one or more matrix headers.  Besides the bytes it holds the list of cv::dct calls that have been
// ISSUED on sub-rectangles of it but not yet EXECUTED: the reference runs cv::dct once per 8x8 block (97 920 calls per
// 1080p frame, libs/encoder.cpp:330-337); each would be a PCIe round trip of 256 bytes, so the calls are collected
// here and leave as ONE launch (svc_hip_dct_tiles_host) the first time anybody looks at the data -- Mat::ptr / at /
// clone / copyTo / convertTo and every cv:: function of this directory call Mat::sync() first.  (Only a read of the
// public `data` member itself cannot be intercepted; the reference reads float matrices through ptr<>() and clone().)
struct DeferredDct { uint32_t x, y, w, h; };
struct Buffer {
  uchar* base = nullptr;
  size_t bytes = 0, step = 0;  // step: row pitch of the allocation this buffer was created for
  int rows = 0, cols = 0, type = 0;
  std::vector<DeferredDct> pending;  // invariant: one tile shape, origins on that shape's grid, in strictly increasing raster order
                                     // -- so no two collected calls overlap and running them together equals running them in turn
  uint64_t last_key = 0;             // (y << 32 | x) + 1 of the newest collected tile
  std::shared_ptr<void> keeps;        // set for storage this buffer does not own (a mapped clip file): what keeps it alive
  Buffer(int rows_, int cols_, int type_);
  Buffer(uchar* memory, int rows_, int cols_, int type_, std::shared_ptr<void> keeps_);  // videoio.cpp: a frame where the file is mapped
  ~Buffer();
  Buffer(const Buffer&) = delete;
  Buffer& operator=(const Buffer&) = delete;
};
void Flush(Buffer& b);  // runs the collected cv::dct calls (core.cpp)

}  // namespace detail

// ---- cv::Mat: a 2-D matrix header over reference-counted storage ----------------------------------------------------
class Mat {
 public:
  enum { AUTO_STEP = 0 };

  Mat() {}
  Mat(int rows_, int cols_, int type_) { create(rows_, cols_, type_); }
  Mat(Size s, int type_) { create(s.height, s.width, type_); }
  // a header over memory the caller owns (libs/encoder.cpp:565-567 wraps its feature vector): no copy, no ownership
  Mat(int rows_, int cols_, int type_, void* data_, size_t step_ = AUTO_STEP)
      : flags(type_), dims(2), rows(rows_), cols(cols_), data(static_cast<uchar*>(data_)) {
    step = step_ == AUTO_STEP ? (size_t)cols_ * elemSize() : step_;
  }
  Mat(const Mat& m, const Rect& roi);  // a view: shares storage (libs/encoder.cpp:333-334)
  Mat(const Mat&) = default;           // shares storage, like OpenCV's reference-counted copy
  Mat& operator=(const Mat&) = default;
  Mat(Mat&& m) noexcept { *this = std::move(m); }
  Mat& operator=(Mat&& m) noexcept {
    if (this != &m) {
      flags = m.flags; dims = m.dims; rows = m.rows; cols = m.cols; data = m.data; step = m.step; buf_ = std::move(m.buf_);
      m.dims = 0; m.rows = m.cols = 0; m.data = nullptr; m.step = 0;  // the moved-from header keeps its type (Mat_<T>)
    }
    return *this;
  }

  // OpenCV's rule: an existing allocation of the same size and type is REUSED (even when shared); anything else is
  // dropped and allocated anew.  The reference depends on the reuse: buildPyramid must not move the planes whose
  // `data` pointers the encoder caches (libs/encoder.cpp:205-218).
  void create(int rows_, int cols_, int type_);
  void create(Size s, int type_) { create(s.height, s.width, type_); }
  void release() { buf_.reset(); data = nullptr; rows = cols = 0; dims = 0; step = 0; }

  Mat clone() const;
  void copyTo(Mat& dst) const;
  void convertTo(Mat& dst, int rtype, double alpha = 1.0, double beta = 0.0) const;
  Mat& setTo(const Scalar& value);
  Mat operator()(const Rect& roi) const { return Mat(*this, roi); }

  static Mat zeros(int rows_, int cols_, int type_);
  static Mat ones(int rows_, int cols_, int type_);

  int type() const { return flags & 4095; }
  int depth() const { return CV_MAT_DEPTH(flags); }
  int channels() const { return CV_MAT_CN(flags); }
  size_t elemSize1() const { static const size_t sz[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return sz[depth()]; }
  size_t elemSize() const { return elemSize1() * (size_t)channels(); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  size_t total() const { return (size_t)rows * (size_t)cols; }
  bool isContinuous() const { return rows <= 1 || step == (size_t)cols * elemSize(); }
  Size size() const { return Size(cols, rows); }

  // Row pointers.  Every accessor first executes the cv::dct calls still collected on the storage (see detail::Buffer).
  uchar* ptr(int y = 0) { sync(); return data + (size_t)y * step; }
  const uchar* ptr(int y = 0) const { sync(); return data + (size_t)y * step; }
  template <typename T> T* ptr(int y = 0) { sync(); return reinterpret_cast<T*>(data + (size_t)y * step); }
  template <typename T> const T* ptr(int y = 0) const { sync(); return reinterpret_cast<const T*>(data + (size_t)y * step); }
  template <typename T> T& at(int y, int x) { sync(); return reinterpret_cast<T*>(data + (size_t)y * step)[x]; }
  template <typename T> const T& at(int y, int x) const { sync(); return reinterpret_cast<const T*>(data + (size_t)y * step)[x]; }

  void sync() const { if (buf_ && !buf_->pending.empty()) detail::Flush(*buf_); }
  detail::Buffer* compat_buffer() const { return buf_.get(); }  // adapter-internal (core.cpp / imgproc.cpp)
  static Mat compat_over(std::shared_ptr<detail::Buffer> b);    // adapter-internal (videoio.cpp): a header over a ready buffer

  // OpenCV's public fields
  int flags = 0;  // the type code (OpenCV keeps more bits here; only the type is modelled)
  int dims = 0;
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  size_t step = 0;  // bytes per row

 private:
  std::shared_ptr<detail::Buffer> buf_;  // null for an empty header and for headers over caller-owned memory
};

void swap(Mat& a, Mat& b);

// ---- cv::Mat_<T>: a Mat whose element type is fixed at compile time (no data members of its own) -------------------
template <typename T> class Mat_ : public Mat {
 public:
  enum { kType = CV_MAKETYPE(DataType<T>::depth, DataType<T>::channels) };
  Mat_() { flags = kType; }
  Mat_(int rows_, int cols_) : Mat(rows_, cols_, kType) {}
  explicit Mat_(Size s) : Mat(s.height, s.width, kType) {}
  Mat_(int rows_, int cols_, T* data_, size_t step_ = AUTO_STEP) : Mat(rows_, cols_, kType, data_, step_) {}
  Mat_(const Mat_& m) = default;
  Mat_(Mat_&& m) noexcept : Mat(std::move(m)) { flags = kType; }
  Mat_(const Mat& m) : Mat() { flags = kType; *this = m; }
  Mat_(Mat&& m) : Mat() { flags = kType; *this = static_cast<const Mat&>(m); }
  Mat_& operator=(const Mat_& m) = default;
  Mat_& operator=(Mat_&& m) noexcept { Mat::operator=(std::move(m)); flags = kType; return *this; }
  Mat_& operator=(const Mat& m) {
    if (m.empty()) { release(); flags = kType; return *this; }
    if (m.type() == kType) Mat::operator=(m);   // shares storage
    else m.convertTo(*this, kType);              // OpenCV converts on a depth mismatch
    return *this;
  }

  void create(int rows_, int cols_) { Mat::create(rows_, cols_, kType); }
  void create(Size s) { Mat::create(s.height, s.width, kType); }
  Mat_ clone() const { return Mat_(Mat::clone()); }
  int channels() const { return DataType<T>::channels; }
  int type() const { return kType; }

  // Element access.  No bounds or emptiness check, like a release build of OpenCV: libs/encoder.cpp:183 evaluates
  // `foreground_cluster_mask_(h, w)` on an EMPTY matrix and discards the reference (a typo for a constructor call).
  T& operator()(int row, int col) {
    if (!data) { static T nothing; return nothing; }  // see above: no null arithmetic for the sanitizers to trip over
    sync();
    return reinterpret_cast<T*>(data + (size_t)row * step)[col];
  }
  const T& operator()(int row, int col) const { return const_cast<Mat_*>(this)->operator()(row, col); }
  Mat_ operator()(const Rect& roi) const { return Mat_(Mat(*this, roi)); }
  T* operator[](int y) { sync(); return reinterpret_cast<T*>(data + (size_t)y * step); }
  const T* operator[](int y) const { sync(); return reinterpret_cast<const T*>(data + (size_t)y * step); }

  // OpenCV returns lazy MatExpr objects here; the reference only ever assigns them to a matrix at once
  // (`Mat1b::ones(h, w) * 255`, libs/encoder.cpp:507), so plain matrices do.
  static Mat_ zeros(int rows_, int cols_) { return Mat_(Mat::zeros(rows_, cols_, kType)); }
  static Mat_ ones(int rows_, int cols_) { return Mat_(Mat::ones(rows_, cols_, kType)); }
};

typedef Mat_<uchar> Mat1b;
typedef Mat_<Vec3b> Mat3b;
typedef Mat_<int> Mat1i;
typedef Mat_<float> Mat1f;
typedef Mat_<Vec3f> Mat3f;
typedef Mat_<Vec4f> Mat4f;
typedef Mat_<double> Mat1d;

Mat operator*(const Mat& m, double s);  // every element times s, saturated to the element type
inline Mat operator*(double s, const Mat& m) { return m * s; }
template <typename T> Mat_<T> operator*(const Mat_<T>& m, double s) { return Mat_<T>(static_cast<const Mat&>(m) * s); }
template <typename T> Mat_<T> operator*(double s, const Mat_<T>& m) { return m * s; }

inline Mat::Mat(const Mat& m, const Rect& roi)
    : flags(m.flags), dims(2), rows(roi.height), cols(roi.width),
      data(m.data + (size_t)roi.y * m.step + (size_t)roi.x * m.elemSize()), step(m.step), buf_(m.buf_) {
  if (roi.x < 0 || roi.y < 0 || roi.width < 0 || roi.height < 0 || roi.x + roi.width > m.cols || roi.y + roi.height > m.rows)
    detail::Fail("cv::Mat::Mat(const Mat&, const Rect&)", "the rectangle leaves the matrix");
}

}  // namespace cv

#endif  // SVC_COMPAT_OPENCV2_CORE_MAT_HPP
