// clip_encoder.hpp -- the hot path for one rank's shard of a clip that is RESIDENT IN HBM: the
// C++ driver behind bench.py and the multi-GPU form of the path (BASELINE
// configs 3-5).
//
// What one step runs is the middle of Encoder::operator() (reference libs/encoder.cpp:453-664)
// over every frame of the shard at once: luma + pyramid, EstimateMotionHierarchical,
// EstimateGlobalMotionRansac, the segmentation glue -> region ids, Dct + the decoder's quant
// lines -- every stage a call into the C ABI of include/svc_hip.h, nothing computed here.
//
// Sharding.  The reference's only cross-frame state is the previous SOURCE frame's Y pyramid
// (libs/encoder.cpp:661-663), so a clip is cut into `world` consecutive chunks (PlanShard) and
// rank r needs one thing from rank r - 1: the pyramid of the frame just before its chunk.  That
// halo is shifted rank r -> r + 1 once per step with RCCL send/recv (svc_hip_halo_shift) on a
// communication stream of its own.  RANSAC draws and k-means seeds are a function of (seed, clip
// frame index) only, so a sharded clip encodes to exactly the bytes of the unsharded one.
//
// Schedule.  kSerial runs a step's stages back to back on one stream.  kPipelined is a software pipeline over consecutive steps, as a
// streaming encoder runs consecutive pieces of a stream: per call of Step() the main stream carries the HBM-bound kernels back to back --
// at one rank luma+pyramid and the motion search of the newest step (nothing to wait for between them) and the transform of the step three
// back; on a multi-rank run luma+pyramid of the newest step, the motion search of the one before (its halo has had an iteration to cross
// xGMI) and the transform of the step four back -- while RANSAC + segmentation (latency-bound: one workgroup per frame) of an earlier step
// start beside them on a stream of their own (forked at the start of the iteration, in front of the luma launch) and have two iterations to
// finish, consecutive steps alternating between two such streams.  The small per-frame outputs exist in four sets, pyramids in two; Flush()
// drains the pipeline.
//
// Chunks (round 6).  With ClipEncoderConfig::chunk_pairs a pipelined step at one rank is cut into chunks of that many frame pairs and the
// pipeline runs over the CHUNKS: every stage launch covers one chunk, RANSAC + segmentation of chunk c run beside the motion search of chunk
// c + 1 and the transform of chunk c - 1 -- the stages of ONE step overlap each other, so LoadFrames / one Step() / Sync() (a clip encoded
// once, as the reference encodes it: libs/encoder.cpp:453-664) is no longer a serial chain of whole-shard kernels.  Same bytes for every
// chunk size (tests/test_gpu_clip.py::test_chunked_steps_equal_whole_steps).  Default: one chunk -- the latency-bound stages take as long
// for half a shard as for a whole one, so two chunks hold CUs beside the bandwidth kernels twice as long: at 1080p x 300 frames a
// once-through clip gains up to 5 %, the steady state of back-to-back steps loses 0-4 % (profiles/r06_ab_chunks.txt).  Hence the
// idle-pipeline rule: only a step that finds the pipeline EMPTY (the first after LoadFrames / Sync) is cut in two, on big shards in the
// two-pass order; steps that follow each other keep whole-shard launches.
//
// Wire output (records of the RAW coefficients, what the reference's encoder emits: libs/encoder.cpp:638-650) with the tuned 8x8 /
// 16x16 transform reads the BGR clip ONCE per step: the record-emitting transform of step s runs at the FRONT of the step and stores
// the luma plane as a by-product (svc_hip_dct_records_luma_frames), the pyramid's other levels follow from it, and the type words --
// region ids that exist only after motion search, RANSAC and segmentation of the same step -- are stored into the records where the
// transform used to run, 2 + depth iterations later (svc_hip_wire_patch_types_frames).  A step's records are complete when its patch
// has run, so they exist in as many sets as the small per-frame outputs.
//
// Planes + quant reads the clip once per step too, by SPECULATION, where that pays: the quantiser's step is the tile's region id's, so the
// transform at the front of the step quantises every tile as background and leaves the luma plane (svc_hip_dct_quant_luma_frames), and the
// tiles of foreground MV blocks are redone once the ids exist (svc_hip_dct_quant_redo_frames).  The redo moves scattered 16-pixel pieces
// and costs ~5 ms per clip's worth of foreground at C3 against ~0.14 ms saved per step: it pays below ~2.7 % foreground MV blocks
// (profiles/r05_ab_speculative_quant.txt: C3 0.5 % -> 2.43 -> 2.31 ms; C3b / C5 13 % -> 2.67 -> 3.18 ms).  So the driver measures: after every
// segmentation a counting kernel leaves the step's foreground share in pinned host memory (no wait), and a step speculates only if the
// newest share that has arrived is at most 2 % (none yet: the plain order) and the shard is at least 50 M pixels x frames (below ~25 frames
// of 1080p the fixed cost of the extra launches eats the saving).  Results never depend on the choice.
#ifndef SVC_CLIP_ENCODER_HPP
#define SVC_CLIP_ENCODER_HPP

#include <cstdint>
#include <functional>
#include <memory>

#include "svc_hip.h"

namespace svc {

struct Shard {
  uint32_t first_frame = 0;  // clip index of the shard's first source frame
  uint32_t frames = 0;       // source frames held by the rank
  bool needs_halo = false;   // first_frame > 0: the pyramid of frame first_frame - 1 comes from rank - 1
  uint32_t pairs = 0;        // encoded frames = frame pairs of the rank (clip frame 0 is tracked-only,
                             // libs/encoder.cpp:361-367)
  uint32_t first_encoded = 0;  // clip index of the first encoded frame
};

// Consecutive chunks, the first (clip_frames % world) ranks one frame longer.
Shard PlanShard(uint32_t clip_frames, uint32_t world, uint32_t rank);

enum class Schedule : uint32_t { kSerial = 0, kPipelined = 1 };

struct ClipEncoderConfig {
  uint32_t width = 0, height = 0;  // source frame size; padded as libs/encoder.cpp:164-168 does
  uint32_t levels = 3;             // pyr-lvl-count
  uint32_t mv_block = 16;          // apps/encoder.cpp:28-58 defaults from here on
  uint32_t search_range = 8;
  uint32_t dct_block_w = 8, dct_block_h = 8;  // transform block; 0 = no transform
  uint32_t fg_step = 1, bg_step = 640;        // apps/decoder.cpp:22-23
  bool wire = false;          // serialised records (libs/encoder.cpp:222-269) of the RAW coefficients, as the
                              // reference's encoder emits them, instead of quantised planes
  bool segmentation = true;   // false: region ids from the in-repo part only (foreground = one region)
  uint64_t seed = 0;
  svc_ransac_params ransac{1, 7.5f, 0.99f, 0.5f};
  svc_segment_params segment{3, 3, 10, 3, 10, 1.0f, 4};
  uint32_t clip_frames = 0;   // frames of the WHOLE clip
  uint32_t rank = 0, world = 1;
  Schedule schedule = Schedule::kPipelined;
  // Tuning (the A/B switches of the measurements under profiles/; results never depend on them)
  uint32_t hbma_flags = SVC_HBMA_AUTO;  // kernel choice of the motion search, SVC_HBMA_*
  uint32_t lat_depth = 0;               // pipelined: iterations RANSAC + segmentation get, 1..3; 0 = 2
  bool standalone_shapes = false;       // pipelined: keep the latency-bound stages' stand-alone launch shapes
  bool segment_fork = false;            // pipelined: let the segmentation fork its heavy attempts to a side stream
  bool inline_rmse = false;             // pipelined: keep RANSAC's in-order RMSE sum inside its kernel instead of beside the segmentation
  bool narrow_attempts = false;         // segmentation: one workgroup per (frame, attempt) whatever the shard size (SVC_LAUNCH_NO_WIDE)
  bool two_bgr_passes = false;          // never the one-pass forms: luma + pyramid, later the transform, two passes over the BGR clip (A/B)
  bool always_speculate = false;        // planes + quant: the speculative one-pass form on every step, whatever the foreground share
  // Not tuning but a statement about the input: LoadFrames() brings consecutive pieces of ONE stream, so the foreground share measured on the
  // last piece stays the policy's prior for the next (default: a load voids it; the first step over new frames is then two passes)
  bool keep_foreground_prior = false;
  uint32_t chunk_pairs = 0;             // pipelined, one rank: frame pairs per chunk of EVERY step; 0 = whole-shard launches, except that a
                                        // step which finds the pipeline empty (a clip encoded once) runs in two chunks on big shards in the
                                        // two-pass order (the idle-pipeline rule, Step())
  bool whole_shard_steps = false;       // never the idle-pipeline rule (A/B)
  bool idle_rule_any_size = false;  // the idle-pipeline rule (Step()) whatever the shard's size (tests: it is tuned for >= 400 M pixels x frames)
  bool fork_behind_front = false;   // one rank, A/B: RANSAC + segmentation fork behind the front-of-step transform, not in front of it
  bool random_policy = false;       // tests: the speculation policy answers yes / no by a fixed pseudo-random sequence over the chunk launches
  bool mixed_steps = false;         // a step into an empty pipeline that knows nothing about the clip: first half two passes, second half one
                                    // pass, blind (A/B: - 2 % at 0.5 % foreground, + 5-8 % at 13 %; profiles/r06_ab_mixed_step.txt)
  bool search_after_transform = false;  // one rank: the motion search, not a pyramid pass, right behind the transform kernel (A/B)
};

enum class Stage : uint32_t { kLumaPyramid = 0, kHalo, kHbma, kRansac, kSegment, kTransform, kTypePatch, kCount };
enum class Buffer : uint32_t { kMv = 0, kMinMad, kGlobalMotion, kRmse, kInlierMask, kInlierCount, kBlockTypes,
                               kCoeffs, kRecords, kPyramids, kBgr, kCount };

class ClipEncoder {
 public:
  // The halo transport: called with the communication stream once per step; must enqueue, on
  // that stream, the send of `bytes` from d_send to rank + 1 (if any) and the receive into d_recv
  // from rank - 1 (if any).  Default: svc_hip_halo_shift on the communicator given to SetComm().
  using HaloFn = std::function<void(const uint8_t* d_send, uint8_t* d_recv, uint64_t bytes, void* stream)>;

  // Allocates every buffer on the current device; throws std::runtime_error on failure.
  explicit ClipEncoder(const ClipEncoderConfig& config);
  ~ClipEncoder();
  ClipEncoder(const ClipEncoder&) = delete;
  ClipEncoder& operator=(const ClipEncoder&) = delete;

  const Shard& shard() const;
  uint32_t padded_width() const;
  uint32_t padded_height() const;
  uint32_t blocks() const;
  uint64_t pyramid_stride() const;

  // Copies `n` PADDED B,G,R u8 frames (padded_height x padded_width x 3) into shard slots
  // [first_local, first_local + n); synchronous.
  void LoadFrames(const uint8_t* src, uint32_t first_local, uint32_t n, bool src_on_device);

  void SetComm(void* nccl_comm);  // an ncclComm_t of `world` ranks (svc_hip_comm_create)
  void SetHaloTransport(HaloFn fn);

  // Enqueues one step (a pass over the whole shard); returns without waiting for the GPU.
  // timed: HIP events around every stage this call enqueues, on the stream it is launched on (in
  // the pipelined schedule those are stages of up to four different steps; Flush() keeps timing
  // what it drains, so K timed steps from an empty pipeline time every stage exactly K times).
  void Step(bool timed = false);
  // A stream of clips, each encoded ONCE (libs/encoder.cpp:453-664 never looks at a clip twice): the step encodes the shard's frames where
  // the caller has them in DEVICE memory (the layout LoadFrames fills: frames x padded_h x padded_w x 3, frame 0 of an unsharded clip
  // tracked only) -- no copy, and above all no drain: LoadFrames synchronises (the resident buffer is being read), so load / step / load /
  // step runs every step into an empty pipeline, while StepFrames over a rotation of buffers keeps consecutive clips overlapped like
  // consecutive steps (RANSAC + segmentation of one clip beside the kernels of the next).  The speculation policy carries over from clip to
  // clip (a stream: what keep_foreground_prior gives the load / step form).  The frames must stay untouched until WaitStep(the returned step)
  // returns (or Sync()): their last reader is the step's transform, up to depth + 1 steps later.  Results: as after Step() -- Output() gives the
  // newest finished step's; a step's small outputs live for depth + 2 steps, its planes / records for output_sets() steps.
  uint32_t StepFrames(const uint8_t* device_frames, bool timed = false);
  void WaitStep(uint32_t step);  // returns once nothing reads that step's frames any more
  void Flush();  // pipelined schedule: enqueues what is left of the steps in flight
  void Sync();   // Flush() + waits for every stream

  // Sum over timed steps of a stage's event time / number of launches timed; Sync()s first.
  void StageTime(Stage s, double* total_ms, uint32_t* launches);
  uint64_t StagePairs(Stage s);  // frame pairs those launches covered (a step's launches are its chunks): time per step = total x pairs / this
  void ResetTimers();
  uint32_t steps_submitted() const;
  uint32_t chunks_per_step() const;  // launches of every stage per Step()
  uint32_t output_sets() const;      // sets the coefficient planes / records exist in

  // The speculation policy: forget what it has measured (Sync()s first); what it did so far -- chunk launches that had the choice, those that
  // speculated, the newest foreground share known (-1: none).
  void ResetPolicy();
  void PolicyInfo(uint64_t* chunks_decided, uint64_t* chunks_speculated, double* foreground_share);

  // Device pointer + size of the NEWEST finished step's output; Sync()s first.
  void* Output(Buffer b, uint64_t* bytes);

 private:
  struct Impl;
  std::unique_ptr<Impl> p_;
  void StepOn(const uint8_t* frames, bool timed);
};

}  // namespace svc

#endif  // SVC_CLIP_ENCODER_HPP
