// stream_encoder.hpp -- the hot path for a clip that lives in HOST memory, batched and overlapped
// with PCIe: the C++ entry point a host application uses instead of calling the reference's
// one-frame-per-call functions (include/svc/motion.hpp) in a loop.
//
// What it runs per encoded frame is the middle of Encoder::operator() (reference
// libs/encoder.cpp:449-650): pad, luma + pyramid, EstimateMotionHierarchical,
// EstimateGlobalMotionRansac, the segmentation glue -> region ids, Dct (+ the decoder's quant
// lines), optionally SerializeEncodedFrame -- every stage through the C ABI of include/svc_hip.h.
// Frames go pinned -> device on a copy stream while the previous batch is in the kernels and the
// one before is on its way back; the one-frame overlap between batches is copied on the device.
//
// RANSAC's draws (the reference seeds std::random_device, libs/motion.cpp:186-187) and the k-means
// seeds are a deterministic function of `seed` and the frame index, so a clip encodes to the same
// result whatever the batch size.
#ifndef SVC_STREAM_ENCODER_HPP
#define SVC_STREAM_ENCODER_HPP

#include <cstdint>
#include <functional>
#include <memory>

#include "svc_hip.h"

namespace svc {

struct StreamEncoderConfig {
  uint32_t width = 0, height = 0;  // source frame size; padded as libs/encoder.cpp:164-168 does
  uint32_t levels = 3;             // pyr-lvl-count
  uint32_t mv_block = 16;          // MV block width (and height, unless mv_block_h is set); apps/encoder.cpp:28-58 defaults from here on
  uint32_t search_range = 8;
  uint32_t dct_block = 8;          // transform block width (and height, unless dct_block_h is set): anything the C ABI's Dct takes
  uint32_t mv_block_h = 0;         // non-square blocks (--mv-block-h / --transform-block-h differing from the widths): 0 = square.
  uint32_t dct_block_h = 0;        // Non-square transform blocks with `wire` take the planes + serialiser route (the fused record kernel is square)
  uint32_t fg_step = 1, bg_step = 640;  // apps/decoder.cpp:22-23
  bool wire = false;               // serialised records (libs/encoder.cpp:222-269) instead of planes: RAW
                                   // coefficients, as the reference's encoder emits them (the decoder picks the
                                   // quant step per tile, libs/decoder.cpp:130-135), over the PADDED tile grid its
                                   // decoder parses (libs/decoder.cpp:185-186); fg_step / bg_step apply to planes only
  bool reference_stream = false;   // with `wire`: the records EXACTLY as the reference's ENCODER serialises them
                                   // (libs/encoder.cpp:647-650 hands SerializeEncodedFrame the UNPADDED size: tile loops over it,
                                   // and the unpadded width as the row stride of the padded planes) -- the stream
                                   // apps/encoder.cpp writes to stdout; record_bytes then counts the unpadded tile grid
  uint32_t batch = 16;             // encoded frames per batch
  uint32_t depth = 3;              // batches in flight, >= 3 (H2D, kernels and D2H of three batches overlap)
  uint32_t copy_threads = 4;       // threads that move a source frame into the pinned batch buffer (the caller's included; 1 = the caller alone)
  uint64_t seed = 0;
  svc_ransac_params ransac{1, 7.5f, 0.99f, 0.5f};
  svc_segment_params segment{3, 3, 10, 3, 10, 1.0f, 4};
};

// One finished batch; the pointers are pinned host memory owned by the encoder and stay valid
// until depth - 2 more batches have been delivered (a slot is re-staged one iteration before its
// turn to deliver comes again): with the default depth of 3, until the NEXT delivery returns.
struct EncodedBatch {
  uint32_t first_frame = 0;  // clip index of the batch's first encoded frame (frame 0 is tracked-only)
  uint32_t count = 0;
  uint32_t padded_w = 0, padded_h = 0, mv_field_w = 0, mv_field_h = 0;
  const float* mv_xy = nullptr;          // [count][blocks][2]
  const float* global_motion = nullptr;  // [count][2]
  const uint32_t* block_types = nullptr; // [count][blocks], 0 = background
  const float* coeffs = nullptr;         // [count][3][padded_h][padded_w], planes B,G,R (wire == false)
  const uint8_t* records = nullptr;      // [count][record_bytes] (wire == true)
  uint64_t record_bytes = 0;
  const svc_wire_header* header = nullptr;  // wire == true, first batch of a clip only: the 32 bytes that
                                            // open the reference's stream (libs/codec.hpp:8-17, encoder.cpp:360-381)
};

// Where the time of one Encode() went (round 6: the PCIe-inclusive rate explains itself).  Host clocks are wall time of the CALLING thread;
// the three device figures are HIP-event times summed over the batches (the streams overlap: they do not add up to the wall time, the
// largest of them bounds it).
struct EncodeStats {
  uint32_t batches = 0, encoded_frames = 0;
  uint32_t copy_threads = 0;  // threads that stage a source frame into the pinned buffer (the caller's included)
  uint32_t host_cores = 0;    // cores this process may run on (sched_getaffinity)
  double wall_ms = 0;
  double staging_ms = 0;       // host: source frames -> pinned batch buffer (the copy crew), RANSAC draws
  double slot_wait_ms = 0;     // host: waiting for a batch slot whose previous results are still on their way back
  double deliver_wait_ms = 0;  // host: waiting for a batch's results before handing it to the sink
  double sink_ms = 0;          // host: inside the caller's sink
  double h2d_ms = 0, kernels_ms = 0, d2h_ms = 0;  // device, per stream
  uint64_t h2d_bytes = 0, d2h_bytes = 0;
};

class StreamEncoder {
 public:
  using Sink = std::function<void(const EncodedBatch&)>;
  // Hands out the clip's next source frame (unpadded B,G,R u8, height x width x 3, tightly packed, anywhere in host memory),
  // or nullptr when the clip has ended; the pointer must stay valid until the next call.  May block (a capture queue).
  using Source = std::function<const uint8_t*()>;

  // Allocates every buffer; throws std::runtime_error with the C ABI's message on failure.
  explicit StreamEncoder(const StreamEncoderConfig& config);
  ~StreamEncoder();
  StreamEncoder(const StreamEncoder&) = delete;
  StreamEncoder& operator=(const StreamEncoder&) = delete;

  // Encodes a clip of n_frames >= 2 unpadded B,G,R u8 frames (height x width x 3, tightly packed,
  // anywhere in host memory); sink is called once per batch, in clip order, from this thread.
  void Encode(const uint8_t* bgr, uint32_t n_frames, const Sink& sink);

  // The same for a clip that ARRIVES frame by frame (the reference's reader thread -> queue -> Encoder::operator(),
  // apps/encoder.cpp:125-148): frames are pulled from `next` as the batches need them.  header_frame_count is what the
  // stream's header announces (the reference takes it from the container, libs/encoder.cpp:361-367), not a limit.
  void Encode(const Source& next, uint32_t header_frame_count, const Sink& sink);

  uint32_t padded_width() const;
  uint32_t padded_height() const;
  const EncodeStats& last_stats() const;  // of the last Encode() that returned

 private:
  struct Impl;
  std::unique_ptr<Impl> p_;
};

}  // namespace svc

#endif  // SVC_STREAM_ENCODER_HPP
