// encoder_hip.cpp -- the reference's `class Encoder` and its three Validate functions (libs/encoder.hpp:14, :23, :38, :52-95)
// implemented on svc::StreamEncoder: the drop-in one level above libsvc_motion.so.  Compiled against the REFERENCE'S OWN
// header (-I<reference>/libs; never copied) so that apps/encoder.cpp -- unchanged -- constructs this Encoder, starts its reader
// and writer threads and calls operator() exactly as it does with the reference's libs/encoder.cpp (apps/encoder.cpp:213-228).
//
// What operator() does is the reference's per-frame loop (libs/encoder.cpp:341-664) batched: frames are pulled from the
// reader's queue sixteen at a time, go to the GPU once as 8-bit B,G,R, and come back as the serialised records themselves --
// Header first (libs/codec.hpp:8-17), then one byte vector per encoded frame, bit-compatible with what
// SerializeEncodedFrame emits for the reference's own arguments (the unpadded tile loops and row stride of
// libs/encoder.cpp:647-650 included).  No cv:: arithmetic is called: the adapter under compat/ only supplies the matrix TYPE
// the two queues carry.  RANSAC draws and k-means seeds come from one seed (std::random_device unless SvcEncoderSeed() was
// called), as the reference's come from its own generators: two runs of the reference differ in region ids the same way.
//
// Scope: every configuration the reference's Validate admits -- non-square MV and transform blocks included (those take the per-level
// search kernel and the planes + serialiser route instead of the tuned ones).

#include <algorithm>
#include <chrono>
#include <initializer_list>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "codec.hpp"    // the reference's (Header)
#include "encoder.hpp"  // the reference's (class Encoder, EncoderConfig, Validate)
#include "svc/stream_encoder.hpp"
#include "svc_hip.h"  // svc_hip_tune_host_allocator (opt-in)

namespace {
bool g_seeded = false;
uint64_t g_seed = 0;

// A fatal error inside operator(): the reader and writer threads of apps/encoder.cpp are alive and may be blocked on the two
// queues, which are objects of static storage -- std::exit would destroy them under the waiters.  Leave without destructors.
[[noreturn]] void Die() {
  std::fflush(stderr);
  std::_Exit(EXIT_FAILURE);
}

Error Invalid(std::string what) { return Error{ErrorCode::kInvalidParameter, std::move(what)}; }
}  // namespace

// Reproducible runs (tests): fixes the seed of RANSAC's draws and of the k-means seeding for every Encoder of the process.
extern "C" void SvcEncoderSeed(unsigned long long seed) {
  g_seed = seed;
  g_seeded = true;
}

// ---- the configuration rules of libs/encoder.cpp:20-142: rule for rule, in the reference's order, with the reference's MESSAGES -- the
// application prints them (apps/encoder.cpp:185-190), so they are part of what a drop-in answers with
// (tests/test_compat_host.py::test_both_encoder_builds_refuse_a_configuration_with_the_same_words holds the two builds to each other).
namespace {
struct Rule { bool broken; const char* what; const char* must; };
Error FirstBroken(std::initializer_list<Rule> rules) {
  for (const Rule& r : rules)
    if (r.broken) return Invalid(std::string("invalid ") + r.what + ": " + r.must);
  return Error{ErrorCode::kOk};
}
}  // namespace

Error Validate(const RansacParams& p) {
  return FirstBroken({{p.inlier_thresh < 0, "inlier threshold", "must be >= 0"},
                      {p.success_prob < 0, "success probability", "must be >= 0"},
                      {p.inlier_ratio < 0, "inlier ratio", "must be >= 0"}});
}

Error Validate(const KMeansParams& p) {
  return FirstBroken({{p.cluster_count == 0, "cluster count", "must be > 0"},
                      {p.attempt_count == 0, "attempt count", "must be > 0"},
                      {p.max_iter_count == 0, "maximum iteration count", "must be > 0"},
                      {p.epsilon <= 0, "epsilon", "must be > 0"}});
}

Error Validate(const EncoderConfig& c) {
  Error e = FirstBroken({{c.mv_block_w < 1, "mv block width", "must be > 0"},
                         {c.mv_block_h < 1, "mv block height", "must be > 0"},
                         {c.pyr_lvl_count < 1, "pyramid level count", "must be > 0"}});
  if (e.code != ErrorCode::kOk) return e;
  if (c.mv_search_range / Pow2(c.pyr_lvl_count - 1) == 0)
    return Invalid("invalid mv search and pyramid level count: the quotient from dividing the mv search range by the pyramid level reduction "
                   "factor must be > 0");
  e = Validate(c.ransac);
  if (e.code != ErrorCode::kOk) return Error{e.code, "validating RANSAC parameters: " + e.message};
  e = Validate(c.kmeans);
  if (e.code != ErrorCode::kOk) return Error{e.code, "validating k-means parameters: " + e.message};
  const uint cc = c.connected_components_connectivity;
  // transform blocks larger than, or not dividing, the MV block would overlap several MV blocks: a tile's region id would be ambiguous
  return FirstBroken({{cc != 4 && cc != 8, "connected components connectivity", "must be either 4 or 8"},
                      {c.transform_block_w < 1, "transform block width", "must be > 0"},
                      {c.transform_block_h < 1, "transform block height", "must be > 0"},
                      {c.transform_block_w > c.mv_block_w, "transform block width and mv block width", "transform block width must be <= mv block width"},
                      {c.transform_block_h > c.mv_block_h, "transform block height and mv block height", "transform block height must be <= mv block height"},
                      {c.transform_block_w >= 1 && c.mv_block_w % c.transform_block_w != 0, "mv block width and transform block width",
                       "mv block width must be divisible by transform block width"},
                      {c.transform_block_h >= 1 && c.mv_block_h % c.transform_block_h != 0, "mv block height and transform block height",
                       "mv block height must be divisible by transform block height"}});
}

Encoder::Encoder(const EncoderConfig& cfg, const VideoProperties& vidprops, CircularQueue<cv::Mat3b>& in_queue,
                 std::future<void> attempted_first_frame_read, CircularQueue<std::vector<uchar>>& out_queue)
    : cfg_{cfg},
      vidprops_{vidprops},
      in_queue_{in_queue},
      attempted_first_frame_read_{std::move(attempted_first_frame_read)},
      out_queue_{out_queue} {
  const uint f = Pow2(cfg_.pyr_lvl_count - 1);
  padded_frame_w_ = ClosestLargerDivisible(vidprops_.frame_w, cfg_.mv_block_w, f);  // libs/encoder.cpp:164-168
  padded_frame_h_ = ClosestLargerDivisible(vidprops_.frame_h, cfg_.mv_block_h, f);
  frame_excess_w_ = padded_frame_w_ - vidprops_.frame_w;
  frame_excess_h_ = padded_frame_h_ - vidprops_.frame_h;
  mv_field_w_ = padded_frame_w_ / cfg_.mv_block_w;
  mv_field_h_ = padded_frame_h_ / cfg_.mv_block_h;
}

void Encoder::operator()() {
  attempted_first_frame_read_.wait();
  if (in_queue_.IsEmpty()) return;  // the reader found no frame: no output at all (libs/encoder.cpp:344-348)

  svc::StreamEncoderConfig c;
  c.width = vidprops_.frame_w; c.height = vidprops_.frame_h;
  c.levels = cfg_.pyr_lvl_count;
  c.mv_block = cfg_.mv_block_w; c.mv_block_h = cfg_.mv_block_h;
  c.search_range = cfg_.mv_search_range;
  c.dct_block = cfg_.transform_block_w; c.dct_block_h = cfg_.transform_block_h;
  // sixteen frames per batch; fewer when the container says the clip is shorter (the pinned buffers of a batch are 31 MB per frame at
  // 1080p, and page-locking them is most of a short run's start-up)
  c.batch = vidprops_.frame_count > 1 ? std::min<uint32_t>(16u, vidprops_.frame_count - 1) : 16u;
  c.wire = true;
  c.reference_stream = true;  // the bytes SerializeEncodedFrame emits for the reference's arguments (libs/encoder.cpp:647-650)
  c.ransac = svc_ransac_params{cfg_.ransac.subset_sz, cfg_.ransac.inlier_thresh, cfg_.ransac.success_prob, cfg_.ransac.inlier_ratio};
  c.segment = svc_segment_params{cfg_.morph_rect_w, cfg_.morph_rect_h, cfg_.kmeans.cluster_count, cfg_.kmeans.attempt_count,
                                 cfg_.kmeans.max_iter_count, cfg_.kmeans.epsilon, cfg_.connected_components_connectivity};
  if (g_seeded) {
    c.seed = g_seed;
  } else {
    std::random_device rd;  // the reference's own source of randomness (libs/motion.cpp:186)
    c.seed = ((uint64_t)rd() << 32) | rd();
  }

  {  // the stream's header goes out as soon as the first frame has arrived, before anything is encoded (libs/encoder.cpp:360-381)
    uint frame_count = vidprops_.frame_count;
    if (frame_count > 0) --frame_count;  // the first frame is tracked only
    Header h{frame_count, vidprops_.frame_w, vidprops_.frame_h, frame_excess_w_, frame_excess_h_,
             cfg_.transform_block_w, cfg_.transform_block_h, 3u};
    const uchar* p = reinterpret_cast<const uchar*>(&h);
    out_queue_.Push(std::vector<uchar>(p, p + sizeof(h)));
  }

  // Every encoded frame leaves as a fresh 25 MB std::vector (the queue's element type, libs/encoder.hpp:57) that the writer thread frees
  // a moment later.  glibc serves such sizes with mmap / munmap -- a page fault per 4 KB on every frame; keeping them on ONE heap lets
  // the next frame reuse the block the writer just returned.  That is the host process's malloc policy: applied only where the process
  // opted in (SVC_KEEP_LARGE_BLOCKS=1, or its own svc_hip_tune_host_allocator call before this one).
  if (svc_hip_host_tuning_requested()) svc_hip_tune_host_allocator(SVC_HOST_KEEP_LARGE_BLOCKS | SVC_HOST_ONE_ARENA);
  try {
    const auto t_begin = std::chrono::steady_clock::now();
    svc::StreamEncoder enc(c);  // page-locks the batch buffers: most of a short run's time
    const auto t_ready = std::chrono::steady_clock::now();
    uint64_t encoded_frames = 0;
    cv::Mat3b frame;  // keeps the frame handed to the encoder alive until it asks for the next one
    const size_t want = (size_t)vidprops_.frame_w * vidprops_.frame_h * 3;
    double wait_reader = 0, wait_helpers = 0, wait_writer = 0;  // seconds this thread spent blocked on each neighbour (SVC_ENCODER_REPORT)
    auto clock = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(clock() - t).count(); };
    auto next = [&]() -> const uint8_t* {
      const auto t0 = clock();
      const bool more = in_queue_.Pop(frame);
      wait_reader += since(t0);
      if (!more) return nullptr;  // queue empty and the reader is done
      if (frame.empty() || (size_t)frame.rows * frame.cols * 3 != want || !frame.isContinuous()) {
        std::fprintf(stderr, "svc Encoder: a frame of %d x %d does not match the capture's %u x %u\n", frame.cols, frame.rows,
                     vidprops_.frame_w, vidprops_.frame_h);
        Die();
      }
      return frame.data;
    };
    // One vector per frame, as the reference pushes them (:652).  Filling a 25 MB vector from the pinned batch buffer is 1.2 ms of one
    // core -- more than a frame's share of everything else this thread does -- so a batch's vectors are filled by a few helper threads
    // while this thread goes on staging the next batch's source frames; they are pushed, in clip order, when the next batch is
    // delivered (a delivered batch's buffers stay valid until the following delivery returns, include/svc/stream_encoder.hpp).
    const uint32_t copiers = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
    std::vector<std::vector<uchar>> filled;
    std::vector<std::thread> helpers;
    struct JoinAll {  // an exception on its way out of Encode must not meet a running thread
      std::vector<std::thread>& threads;
      ~JoinAll() { for (auto& t : threads) if (t.joinable()) t.join(); }
    } join_all{helpers};
    auto push_filled = [&]() {
      auto t0 = clock();
      for (auto& h : helpers) h.join();
      wait_helpers += since(t0);
      helpers.clear();
      t0 = clock();
      for (auto& v : filled) out_queue_.Push(std::move(v));
      wait_writer += since(t0);
      encoded_frames += filled.size();
      filled.clear();
    };
    auto sink = [&](const svc::EncodedBatch& b) {
      push_filled();  // the previous batch
      filled.resize(b.count);
      const uint8_t* records = b.records;
      const uint64_t bytes = b.record_bytes;
      const uint32_t count = b.count;
      for (uint32_t t = 0; t < copiers && t < count; ++t)
        helpers.emplace_back([&filled, records, bytes, count, copiers, t]() {
          for (uint32_t i = t; i < count; i += copiers) filled[i] = std::vector<uchar>(records + (size_t)i * bytes, records + (size_t)(i + 1) * bytes);
        });
    };
    enc.Encode(next, vidprops_.frame_count, sink);
    push_filled();  // the last batch
    if (std::getenv("SVC_ENCODER_REPORT")) {  // what bench.py's end_to_end object reads: the loop's own clock, start-up apart
      const double setup = std::chrono::duration<double>(t_ready - t_begin).count();
      const double loop = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ready).count();
      std::fprintf(stderr, "svc Encoder: %llu frames encoded in %.4f s (%.1f frames/s) after %.3f s of set-up; batch %u; this thread waited "
                   "%.3f s for the reader, %.3f s for its copy helpers, %.3f s for the writer's queue\n",
                   (unsigned long long)encoded_frames, loop, loop > 0 ? encoded_frames / loop : 0.0, setup, c.batch, wait_reader, wait_helpers,
                   wait_writer);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "svc Encoder: %s\n", e.what());
    Die();
  }
  out_queue_.SignalProducerIsDone();
}
