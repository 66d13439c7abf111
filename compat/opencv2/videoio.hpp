// compat/opencv2/videoio.hpp -- cv::VideoCapture as apps/encoder.cpp uses it (:126-133, :192-204): an uncompressed clip
// read from a file.  PRODUCT-SIDE ADAPTER, NOT AN ORACLE: see core/mat.hpp.  No codec is involved: the container
// formats are
//   * "SVCBGR1\0" + u32 width, height, frame count, 0 (little endian), then frames of height x width x 3 bytes B,G,R;
//   * a stream of binary PPM images ("P6", maxval 255; R,G,B -> swapped to B,G,R) of one size, e.g. what
//     `ffmpeg -i in.mp4 -f image2pipe -vcodec ppm out.ppm` writes.
#ifndef SVC_COMPAT_OPENCV2_VIDEOIO_HPP
#define SVC_COMPAT_OPENCV2_VIDEOIO_HPP

#include <cstdio>
#include <memory>
#include <string>

#include "opencv2/core/mat.hpp"

namespace cv {

enum VideoCaptureProperties { CAP_PROP_POS_FRAMES = 1, CAP_PROP_FRAME_WIDTH = 3, CAP_PROP_FRAME_HEIGHT = 4, CAP_PROP_FPS = 5,
                              CAP_PROP_FRAME_COUNT = 7 };

class VideoCapture {
 public:
  VideoCapture() {}
  explicit VideoCapture(const String& filename) { open(filename); }
  ~VideoCapture() { release(); }
  VideoCapture(const VideoCapture&) = delete;
  VideoCapture& operator=(const VideoCapture&) = delete;

  bool open(const String& filename);
  bool isOpened() const { return f_ != nullptr; }
  void release();
  double get(int propId) const;
  // Every frame is storage of its own: apps/encoder.cpp:139-145 pushes the same cv::Mat3b header into its queue after
  // each read, and a header shares its storage -- reusing an allocation would overwrite frames still queued.  A frame of
  // an SVCBGR1 file is a header over the file where it is MAPPED (private, copy-on-write: writable like any matrix, the
  // file never changes): no copy at all on the reader's thread, and the mapping lives as long as any frame does.  PPM
  // frames (channel order swapped) are read into a fresh allocation.
  bool read(Mat& image);
  VideoCapture& operator>>(Mat& image) { read(image); return *this; }

 private:
  std::FILE* f_ = nullptr;
  int w_ = 0, h_ = 0, count_ = 0, pos_ = 0;
  bool ppm_ = false;
  std::shared_ptr<void> map_;  // the whole SVCBGR1 file; null for PPM streams or where mapping failed (then: fread)
  size_t map_bytes_ = 0;
};

}  // namespace cv

#endif  // SVC_COMPAT_OPENCV2_VIDEOIO_HPP
