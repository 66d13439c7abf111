// Host-only behaviour of compat/opencv2 that the reference's encoder relies on (no GPU call is made here): header
// sharing, create() reuse, views, the moved-from / empty corner cases, the data-movement functions, the clip containers.
// Built and run by tests/test_compat_host.py.
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "opencv2/core.hpp"
#include "opencv2/imgproc.hpp"
#include "opencv2/videoio.hpp"

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

int main(int argc, char** argv) {
  using namespace cv;
  {  // headers share storage; create() of the same size and type keeps the allocation (libs/encoder.cpp:205-218, :470)
    Mat1b y(8, 16);
    std::vector<Mat1b> pyr(3);
    pyr[0] = y;
    CHECK(pyr[0].data == y.data);
    uchar* p = y.data;
    y.create(8, 16);
    CHECK(y.data == p);
    Mat1b z = y;
    z(2, 3) = 77;
    CHECK(y(2, 3) == 77 && y.ptr<uchar>(2)[3] == 77);
    Mat1b c = y.clone();
    c(2, 3) = 1;
    CHECK(y(2, 3) == 77 && c.data != y.data);
    y.create(4, 4);
    CHECK(y.data != p && z.data == p && z(2, 3) == 77);  // the old allocation lives on in its other headers
    Mat1b a(2, 2), b(3, 3);
    uchar *pa = a.data, *pb = b.data;
    cv::swap(a, b);
    CHECK(a.data == pb && b.data == pa && a.rows == 3 && b.rows == 2);
  }
  {  // typed matrices: channel counts of empty headers, element access on an empty matrix (libs/encoder.cpp:183), MatExpr stand-ins
    Mat3b f;
    CHECK(f.channels() == 3 && f.empty() && f.type() == CV_8UC3);
    Mat1b empty;
    (void)empty(5, 7);  // the reference evaluates this and discards the reference
    Mat1b ones = Mat1b::ones(3, 4) * 255;
    CHECK(ones.rows == 3 && ones.cols == 4 && ones(2, 3) == 255 && ones(0, 0) == 255);
    Mat1b zeros = Mat1b::zeros(3, 4);
    CHECK(zeros(1, 1) == 0);
    Mat k = getStructuringElement(MORPH_RECT, Size(3, 2));
    CHECK(k.rows == 2 && k.cols == 3 && k.type() == CV_8UC1 && k.data[5] == 1);
    Mat3b moved = Mat3b(2, 2);
    Mat3b taken = std::move(moved);
    CHECK(moved.empty() && moved.channels() == 3 && taken.rows == 2);
    std::vector<float> feat(8, 1.5f);
    Mat4f wrap(2, 1, reinterpret_cast<Vec4f*>(feat.data()));  // libs/encoder.cpp:565-567
    CHECK((void*)wrap.data == (void*)feat.data() && wrap.channels() == 4 && wrap.rows == 2 && wrap.isContinuous());
    Mat1i labels;
    CHECK(labels.empty() && labels.type() == CV_32SC1);
  }
  {  // views (libs/encoder.cpp:333-334) write through to the plane
    Mat1f plane(16, 24);
    plane.setTo(Scalar(0.0));
    const Mat1f& cplane = plane;
    Mat1f block = cplane(Rect(8, 8, 8, 8));
    CHECK(block.data == plane.data + 8 * plane.step + 8 * 4 && block.step == plane.step && !block.isContinuous());
    block(1, 2) = 3.5f;
    CHECK(plane(9, 10) == 3.5f);
  }
  {  // copyMakeBorder / convertTo / split / extractChannel (libs/encoder.cpp:447, :638, :328, :450): data movement, exact
    Mat3b src(3, 5);
    for (int y = 0; y < 3; ++y)
      for (int x = 0; x < 5; ++x) src(y, x) = Vec3b((uchar)(10 * y + x), (uchar)(100 + x), (uchar)(200 + y));
    Mat3b pad(4, 8);
    uchar* keep = pad.data;
    copyMakeBorder(src, pad, 0, 1, 0, 3, BORDER_CONSTANT, Scalar(0, 0, 0));
    CHECK(pad.data == keep && pad.rows == 4 && pad.cols == 8);  // the preallocated padded frame is reused
    CHECK(pad(2, 4)[0] == 24 && pad(2, 4)[2] == 202 && pad(3, 0)[1] == 0 && pad(0, 5)[0] == 0 && pad(0, 7)[2] == 0);
    Mat3f f(4, 8);
    float* fkeep = reinterpret_cast<float*>(f.data);
    pad.convertTo(f, CV_32FC3);
    CHECK(reinterpret_cast<float*>(f.data) == fkeep && f(2, 4)[1] == 104.0f && f(3, 7)[0] == 0.0f);
    std::vector<Mat1f> planes(3);
    for (auto& p : planes) p = Mat1f(4, 8);
    float* p1 = reinterpret_cast<float*>(planes[1].data);
    split(f, planes);
    CHECK(reinterpret_cast<float*>(planes[1].data) == p1 && planes[0](2, 4) == 24.0f && planes[1](2, 4) == 104.0f && planes[2](2, 4) == 202.0f);
    Mat1b g;
    extractChannel(pad, g, 1);
    CHECK(g.rows == 4 && g.cols == 8 && g(1, 3) == 103 && g(3, 3) == 0);
    Mat3b yuv_like = pad;
    Mat1b first(4, 8);
    uchar* fk = first.data;
    extractChannel(yuv_like, first, 0);
    CHECK(first.data == fk && first(2, 1) == 21);
  }
  {  // theRNG: OpenCV's multiply-with-carry step from its default state
    RNG r;
    CHECK(r.state == 0xffffffffull);
    r.next();
    CHECK(r.state == 0xffffffffull * 4164903690ull);
    TermCriteria tc(TermCriteria::COUNT | TermCriteria::EPS, 10, 1.0);
    CHECK(tc.type == 3 && tc.maxCount == 10);
  }
  {  // frame-sized matrices: the data-movement passes run on several threads above 1 MB (compat/src/core.cpp: ParallelRows) -- every byte arrives
    const int h = 1080, w = 1920;
    Mat3b frame(h, w);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) frame(y, x) = Vec3b((uchar)(x + 3 * y), (uchar)(x * 7 + y), (uchar)(x ^ y));
    Mat3b copy = frame.clone();
    CHECK(copy.data != frame.data && std::memcmp(copy.data, frame.data, (size_t)h * w * 3) == 0);
    Mat3b padded;
    copyMakeBorder(frame, padded, 0, 8, 0, 16, BORDER_CONSTANT, Scalar(0, 0, 0));
    bool ok = padded.rows == h + 8 && padded.cols == w + 16;
    for (int y = 0; ok && y < padded.rows; ++y)
      for (int x = 0; x < padded.cols; ++x) {
        const Vec3b v = padded(y, x), want = (y < h && x < w) ? frame(y, x) : Vec3b(0, 0, 0);
        if (v[0] != want[0] || v[1] != want[1] || v[2] != want[2]) { ok = false; break; }
      }
    CHECK(ok);
    Mat3f f;
    padded.convertTo(f, CV_32FC3);
    std::vector<Mat1f> planes;
    split(f, planes);
    Mat1b green;
    extractChannel(padded, green, 1);
    ok = planes.size() == 3;
    for (int y = 0; ok && y < padded.rows; y += 1)
      for (int x = 0; x < padded.cols; ++x) {
        const Vec3b v = padded(y, x);
        if (f(y, x)[0] != (float)v[0] || f(y, x)[2] != (float)v[2] || planes[0](y, x) != (float)v[0] || planes[1](y, x) != (float)v[1] ||
            planes[2](y, x) != (float)v[2] || green(y, x) != v[1]) { ok = false; break; }
      }
    CHECK(ok);
    Mat3b view = padded(Rect(16, 8, 1024, 512));  // a view (pitch != row bytes) through the same passes
    Mat3b view_copy = view.clone();
    Mat3f view_f;
    view.convertTo(view_f, CV_32FC3);
    ok = view_copy.isContinuous() && view_copy.rows == 512 && view_copy.cols == 1024;
    for (int y = 0; ok && y < 512; ++y)
      for (int x = 0; x < 1024; ++x)
        if (view_copy(y, x)[1] != padded(y + 8, x + 16)[1] || view_f(y, x)[2] != (float)padded(y + 8, x + 16)[2]) { ok = false; break; }
    CHECK(ok);
  }
  if (argc == 3) {  // the two clip containers written by the Python side: same frames, B,G,R order, fresh storage per read
    VideoCapture a(argv[1]), b(argv[2]);
    CHECK(a.isOpened() && b.isOpened());
    CHECK(a.get(CAP_PROP_FRAME_WIDTH) == 6 && a.get(CAP_PROP_FRAME_HEIGHT) == 4 && a.get(CAP_PROP_FRAME_COUNT) == 3);
    CHECK(b.get(VideoCaptureProperties::CAP_PROP_FRAME_WIDTH) == 6 && b.get(CAP_PROP_FRAME_HEIGHT) == 4 && b.get(CAP_PROP_FRAME_COUNT) == 3);
    Mat3b fa, fb, first;
    int n = 0;
    while (a.read(fa)) {
      CHECK(b.read(fb));
      CHECK(fa.rows == 4 && fa.cols == 6 && std::memcmp(fa.data, fb.data, 72) == 0);
      CHECK(fa(1, 2)[0] == (uchar)(n * 50 + 1 * 6 * 3 + 2 * 3) && fa(1, 2)[2] == (uchar)(n * 50 + 1 * 6 * 3 + 2 * 3 + 2));
      if (n == 0) first = fa;
      else CHECK(first.data != fa.data && first(0, 0)[0] == 0);  // a queued header keeps its frame (apps/encoder.cpp:139-145)
      ++n;
    }
    CHECK(n == 3 && !b.read(fb));
    {  // SVCBGR1 frames are headers over the file where it is mapped: writable (copy-on-write, the file never changes), alive after the
       // capture has gone, and the file's real length -- not the header's count -- ends the clip
      Mat3b kept;
      {
        VideoCapture c(argv[1]);
        CHECK(c.read(kept));
        kept(0, 0)[0] = 77;
        Mat3b again;
        VideoCapture d(argv[1]);
        CHECK(d.read(again) && again(0, 0)[0] == 0 && again.data != kept.data);  // neither the file nor another mapping saw the write
      }
      CHECK(kept(0, 0)[0] == 77 && kept(1, 2)[0] == (uchar)(1 * 6 * 3 + 2 * 3));
      Mat3b copy = kept.clone();
      CHECK(copy.data != kept.data && std::memcmp(copy.data, kept.data, 72) == 0);
      const std::string cut = std::string(argv[1]) + ".cut";
      std::FILE* in = std::fopen(argv[1], "rb");
      std::FILE* out = std::fopen(cut.c_str(), "wb");
      CHECK(in && out);
      if (in && out) {
        char bytes[24 + 72 + 40];  // the header (which still says 3 frames), one whole frame and a piece of the second
        CHECK(std::fread(bytes, 1, sizeof(bytes), in) == sizeof(bytes) && std::fwrite(bytes, 1, sizeof(bytes), out) == sizeof(bytes));
        std::fclose(in);
        std::fclose(out);
        VideoCapture e(cut);
        Mat3b f;
        CHECK(e.isOpened() && e.get(CAP_PROP_FRAME_COUNT) == 3 && e.read(f) && f(0, 0)[0] == 0 && !e.read(f) && f.empty());
        std::remove(cut.c_str());
      }
    }
    VideoCapture none("/nonexistent/clip");
    CHECK(!none.isOpened());
  }
  if (g_fail) return 1;
  std::puts("compat host semantics ok");
  return 0;
}
