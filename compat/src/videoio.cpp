// compat/src/videoio.cpp -- cv::VideoCapture over uncompressed clips (see compat/opencv2/videoio.hpp).
// PRODUCT-SIDE ADAPTER, NOT AN ORACLE (see compat/opencv2/core/mat.hpp).
#include <sys/mman.h>
#include <sys/stat.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "internal.hpp"
#include "opencv2/videoio.hpp"

namespace cv {

namespace {

constexpr size_t kRawHeader = 24;  // "SVCBGR1\0" + four u32

// one PPM header token (whitespace- and '#'-comment-separated); false at end of file
bool PpmToken(std::FILE* f, long* out) {
  int c = std::fgetc(f);
  for (;;) {
    while (c == ' ' || c == '\t' || c == '\n' || c == '\r') c = std::fgetc(f);
    if (c != '#') break;
    while (c != '\n' && c != EOF) c = std::fgetc(f);
  }
  if (c < '0' || c > '9') return false;
  long v = 0;
  while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); c = std::fgetc(f); }
  *out = v;  // the single whitespace byte after the token has been consumed (what the format asks for after maxval)
  return true;
}

bool PpmHeader(std::FILE* f, int* w, int* h) {
  const int a = std::fgetc(f), b = std::fgetc(f);
  if (a != 'P' || b != '6') return false;
  long ww, hh, maxval;
  if (!PpmToken(f, &ww) || !PpmToken(f, &hh) || !PpmToken(f, &maxval)) return false;
  if (ww < 1 || hh < 1 || ww > 65535 || hh > 65535 || maxval != 255) return false;
  *w = (int)ww; *h = (int)hh;
  return true;
}

}  // namespace

bool VideoCapture::open(const String& filename) {
  release();
  std::FILE* f = std::fopen(filename.c_str(), "rb");
  if (!f) return false;
  char magic[8] = {};
  if (std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "SVCBGR1", 8) == 0) {
    uint32_t hdr[4];
    if (std::fread(hdr, 4, 4, f) != 4 || hdr[0] == 0 || hdr[1] == 0 || hdr[0] > 65535 || hdr[1] > 65535) { std::fclose(f); return false; }
    w_ = (int)hdr[0]; h_ = (int)hdr[1]; count_ = (int)hdr[2];
    ppm_ = false;
    struct stat st;
    if (fstat(fileno(f), &st) == 0 && st.st_size > (off_t)kRawHeader) {
      const size_t n = (size_t)st.st_size;
      void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE, fileno(f), 0);
      if (m != MAP_FAILED) {
        madvise(m, n, MADV_SEQUENTIAL);
        map_ = std::shared_ptr<void>(m, [n](void* p) { munmap(p, n); });
        map_bytes_ = n;
      }
    }
  } else {
    std::rewind(f);
    if (!PpmHeader(f, &w_, &h_)) { std::fclose(f); return false; }
    const long body = std::ftell(f);  // header length of the first image; the stream repeats it per frame
    std::fseek(f, 0, SEEK_END);
    const long total = std::ftell(f);
    count_ = (int)(total / (body + (long)w_ * h_ * 3));
    std::rewind(f);
    ppm_ = true;
  }
  f_ = f;
  pos_ = 0;
  return true;
}

void VideoCapture::release() {
  if (f_) std::fclose(f_);
  f_ = nullptr;
  map_.reset();  // frames still alive keep the mapping
  map_bytes_ = 0;
  w_ = h_ = count_ = pos_ = 0;
}

double VideoCapture::get(int propId) const {
  switch (propId) {
    case CAP_PROP_FRAME_WIDTH: return w_;
    case CAP_PROP_FRAME_HEIGHT: return h_;
    case CAP_PROP_FRAME_COUNT: return count_;
    case CAP_PROP_POS_FRAMES: return pos_;
    default: return 0.0;
  }
}

bool VideoCapture::read(Mat& image) {
  detail::Timed timed("VideoCapture::read");
  if (!f_) { image.release(); return false; }
  if (ppm_) {
    int w = 0, h = 0;
    if (!PpmHeader(f_, &w, &h) || w != w_ || h != h_) { image.release(); return false; }
  }
  const size_t bytes = (size_t)w_ * h_ * 3;
  if (map_) {
    const size_t at = kRawHeader + (size_t)pos_ * bytes;
    if (at + bytes > map_bytes_) { image.release(); return false; }  // the end of the file (whatever the header's count says)
#ifdef MADV_POPULATE_READ
    {  // map the frame's pages here, on the reader's thread, rather than by page faults under whoever reads the frame first
      const uintptr_t lo = (reinterpret_cast<uintptr_t>(map_.get()) + at) & ~(uintptr_t)4095;
      madvise(reinterpret_cast<void*>(lo), reinterpret_cast<uintptr_t>(map_.get()) + at + bytes - lo, MADV_POPULATE_READ);  // best effort
    }
#endif
    image = Mat::compat_over(std::make_shared<detail::Buffer>(static_cast<uchar*>(map_.get()) + at, h_, w_, CV_8UC3, map_));
    ++pos_;
    return true;
  }
  Mat frame(h_, w_, CV_8UC3);  // a fresh allocation per frame: queued headers keep theirs
  if (std::fread(frame.data, 1, bytes, f_) != bytes) { image.release(); return false; }
  if (ppm_)
    for (size_t i = 0; i < bytes; i += 3) { const uchar t = frame.data[i]; frame.data[i] = frame.data[i + 2]; frame.data[i + 2] = t; }
  image = frame;
  ++pos_;
  return true;
}

}  // namespace cv
