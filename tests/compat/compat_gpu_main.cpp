// compat/opencv2 on the GPU: every cv:: function that forwards to the C ABI gives what the C ABI gives when called directly
// (the C ABI itself is checked against the oracle by tests/test_gpu_imageops.py), and the deferred cv::dct list keeps OpenCV's
// eager semantics -- calls that depend on each other run in turn.  Built and run by tests/test_gpu_compat.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "opencv2/core.hpp"
#include "opencv2/imgproc.hpp"
#include "svc_hip.h"

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

static uint32_t g_lcg = 99;
static uint32_t Rnd() { g_lcg = g_lcg * 1664525u + 1013904223u; return g_lcg >> 8; }

static std::vector<float> DirectTiles(const cv::Mat1f& m, uint32_t bw, uint32_t bh, const std::vector<uint32_t>& xy) {
  std::vector<float> img((size_t)m.rows * m.cols);
  for (int y = 0; y < m.rows; ++y) std::memcpy(&img[(size_t)y * m.cols], m.data + (size_t)y * m.step, (size_t)m.cols * 4);
  CHECK(svc_hip_dct_tiles_host(img.data(), (uint32_t)m.cols, (uint32_t)m.rows, bw, bh, xy.empty() ? nullptr : xy.data(), (uint32_t)(xy.size() / 2)) == SVC_OK);
  return img;
}

static bool Same(const cv::Mat1f& m, const std::vector<float>& img) {
  for (int y = 0; y < m.rows; ++y)
    if (std::memcmp(m.ptr<float>(y), &img[(size_t)y * m.cols], (size_t)m.cols * 4) != 0) return false;
  return true;
}

int main() {
  using namespace cv;
  const int W = 48, H = 32;
  Mat1f src(H, W);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) src(y, x) = (float)(Rnd() % 256);

  {  // the reference's pattern: every 8 x 8 tile in raster order, in place, read back through clone()
    Mat1f a = src.clone();
    for (int y = 0; y < H; y += 8)
      for (int x = 0; x < W; x += 8) { Mat1f t = a(Rect(x, y, 8, 8)); dct(t, t); }
    CHECK(a.compat_buffer()->pending.size() == 24);  // collected, not run
    Mat1f c = a.clone();                             // runs them (one launch) and copies
    CHECK(a.compat_buffer()->pending.empty() && Same(c, DirectTiles(src, 8, 8, {})));
  }
  {  // the same tile twice: the second call must see the first one's result
    Mat1f a = src.clone();
    Mat1f t = a(Rect(8, 8, 8, 8));
    dct(t, t);
    dct(t, t);
    std::vector<float> once = DirectTiles(src, 8, 8, {8, 8});
    Mat1f tmp(H, W);
    for (int y = 0; y < H; ++y) std::memcpy(tmp.ptr<float>(y), &once[(size_t)y * W], W * 4);
    CHECK(Same(a, DirectTiles(tmp, 8, 8, {8, 8})));
  }
  {  // another shape, an off-grid tile, reverse order: each runs what was collected first
    Mat1f a = src.clone();
    std::vector<float> want(src.ptr<float>(0), src.ptr<float>(0) + H * W);
    auto apply = [&](int x, int y, int w, int h) {
      Mat1f t = a(Rect(x, y, w, h));
      dct(t, t);
      Mat1f cur(H, W);
      std::memcpy(cur.ptr<float>(0), want.data(), want.size() * 4);
      want = DirectTiles(cur, (uint32_t)w, (uint32_t)h, {(uint32_t)x, (uint32_t)y});
    };
    apply(16, 0, 8, 8); apply(0, 0, 8, 8);   // out of raster order
    apply(0, 16, 16, 16);                    // another shape
    apply(20, 16, 8, 8);                     // off the 8 x 8 grid, next to ...
    apply(24, 16, 8, 8);                     // ... an overlapping on-grid tile: depends on it
    apply(2, 2, 4, 2);                       // small, off-grid, overlaps the first tile's result
    CHECK(Same(a, want));
  }
  {  // out of place, and a matrix over caller-owned memory (transformed at once)
    Mat1f dst;
    dct(src(Rect(0, 0, 16, 16)), dst);
    Mat1f full = src.clone();
    std::vector<float> want = DirectTiles(full, 16, 16, {0, 0});
    bool ok = dst.rows == 16 && dst.cols == 16;
    for (int y = 0; ok && y < 16; ++y) ok = std::memcmp(dst.ptr<float>(y), &want[(size_t)y * W], 64) == 0;
    CHECK(ok);
    std::vector<float> mine(64);
    for (auto& v : mine) v = (float)(Rnd() % 256);
    std::vector<float> img = mine;
    CHECK(svc_hip_dct_tiles_host(img.data(), 8, 8, 8, 8, nullptr, 0) == SVC_OK);
    Mat1f wrap(8, 8, mine.data());
    dct(wrap, wrap);
    CHECK(std::memcmp(mine.data(), img.data(), 256) == 0);
  }
  {  // cvtColor -> extractChannel -> buildPyramid == the C ABI on the same bytes; planes are reused across calls
    Mat3b bgr(64, 96);
    for (int y = 0; y < 64; ++y)
      for (int x = 0; x < 96; ++x) bgr(y, x) = Vec3b((uchar)Rnd(), (uchar)Rnd(), (uchar)Rnd());
    Mat3b yuv(64, 96);
    cvtColor(bgr, yuv, COLOR_BGR2YUV);
    std::vector<uchar> want((size_t)64 * 96 * 3);
    CHECK(svc_hip_bgr2yuv_host(bgr.data, 96, 64, want.data()) == SVC_OK && std::memcmp(yuv.data, want.data(), want.size()) == 0);
    Mat1b y0(64, 96);
    extractChannel(yuv, y0, 0);
    std::vector<Mat1b> pyr(3);
    pyr[0] = y0; pyr[1] = Mat1b(32, 48); pyr[2] = Mat1b(16, 24);
    uchar *p1 = pyr[1].data, *p2 = pyr[2].data;
    buildPyramid(y0, pyr, 2);
    CHECK(pyr[0].data == y0.data && pyr[1].data == p1 && pyr[2].data == p2);
    std::vector<uchar> l1(32 * 48), l2(16 * 24);
    uint8_t* outs[3] = {nullptr, l1.data(), l2.data()};
    CHECK(svc_hip_build_pyramid_host(y0.data, 96, 64, 3, outs) == SVC_OK);
    CHECK(std::memcmp(pyr[1].data, l1.data(), l1.size()) == 0 && std::memcmp(pyr[2].data, l2.data(), l2.size()) == 0);
  }
  {  // morphologyEx in place, kmeans seeded from theRNG (one step per call), connectedComponents
    const int fw = 30, fh = 17;
    Mat1b mask(fh, fw);
    for (int i = 0; i < fw * fh; ++i) mask.data[i] = (Rnd() % 3) ? 255 : 0;
    std::vector<uchar> want(fw * fh), tmp(fw * fh);
    CHECK(svc_hip_morph_rect_host(mask.data, fw, fh, 3, 3, SVC_MORPH_CLOSE, tmp.data()) == SVC_OK);
    CHECK(svc_hip_morph_rect_host(tmp.data(), fw, fh, 3, 3, SVC_MORPH_OPEN, want.data()) == SVC_OK);
    Mat k = getStructuringElement(MORPH_RECT, Size(3, 3));
    morphologyEx(mask, mask, MORPH_CLOSE, k);
    morphologyEx(mask, mask, MORPH_OPEN, k);
    CHECK(std::memcmp(mask.data, want.data(), want.size()) == 0);
    Mat1i labels;
    const int n = connectedComponents(mask, labels, 4, CV_32S, ConnectedComponentsAlgorithmsTypes::CCL_DEFAULT);
    std::vector<int32_t> wl(fw * fh);
    uint32_t wc = 0;
    CHECK(svc_hip_connected_components_host(mask.data, fw, fh, 4, wl.data(), &wc) == SVC_OK);
    CHECK(n == (int)wc && labels.rows == fh && labels.cols == fw && std::memcmp(labels.ptr<int>(), wl.data(), wl.size() * 4) == 0);
    std::vector<float> feats(200 * 4, 0.0f);
    for (int i = 0; i < 200; ++i) { feats[4 * i + 1] = (float)(int)(Rnd() % 9) - 4; feats[4 * i + 2] = (float)(16 * (Rnd() % 30)); feats[4 * i + 3] = (float)(16 * (Rnd() % 17)); }
    Mat4f data(200, 1, reinterpret_cast<Vec4f*>(feats.data()));
    for (int call = 0; call < 2; ++call) {  // theRNG advances one step per call: the second call uses the next state
      const uint64_t seed = theRNG().state;
      Mat1i ids;
      const double compact = kmeans(data, 5, ids, TermCriteria(TermCriteria::COUNT | TermCriteria::EPS, 10, 1.0), 3, KMEANS_PP_CENTERS);
      std::vector<int32_t> wi(200);
      double wcpt = 0;
      CHECK(svc_hip_kmeans_host(feats.data(), 200, 4, 5, 3, 10, 1.0f, seed, wi.data(), &wcpt) == SVC_OK);
      CHECK(ids.rows == 200 && ids.cols == 1 && std::memcmp(ids.ptr<int>(), wi.data(), 800) == 0 && compact == wcpt);
      CHECK(theRNG().state != seed);
    }
  }
  if (g_fail) return 1;
  std::puts("compat gpu semantics ok");
  return 0;
}
