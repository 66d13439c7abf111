// clip_encoder.cpp -- include/svc/clip_encoder.hpp + include/svc_clip.h: buffers, streams, the
// step schedule and the halo of one rank's shard.  No arithmetic of the hot path lives here;
// every stage is a call into the C ABI (include/svc_hip.h).
#include "svc/clip_encoder.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "svc_clip.h"

namespace svc {
namespace {

void Hip(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("svc::ClipEncoder: ") + what + ": " + hipGetErrorString(e));
}
// an event that has not completed yet is "not ready"; any other status is an error of an earlier launch and is reported HERE, where it
// is first seen, not swallowed as "no news yet"
bool Ready(hipEvent_t e) {
  const hipError_t q = hipEventQuery(e);
  if (q == hipErrorNotReady) return false;
  Hip(q, "hipEventQuery");
  return true;
}
void Abi(int rc, const char* what) {
  if (rc) throw std::runtime_error(std::string("svc::ClipEncoder: ") + what + ": " + svc_hip_last_error());
}

// libs/math.hpp:276-283 (ClosestLargerDivisible)
uint32_t ClosestLargerDivisible(uint32_t dim, uint32_t a, uint32_t b) {
  while (dim % a != 0 || dim % b != 0) ++dim;
  return dim;
}

uint32_t Hash32(uint64_t x) {  // the harness's stateless mixer (synth.py:hash32, stream_encoder.cpp)
  uint32_t v = (uint32_t)x;
  v ^= v >> 16; v *= 0x7FEB352Du;
  v ^= v >> 15; v *= 0x846CA68Bu;
  v ^= v >> 16;
  return v;
}

template <typename T> struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  void Alloc(size_t count) {
    n = count;
    Hip(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)), "hipMalloc");
  }
  uint64_t bytes() const { return (uint64_t)n * sizeof(T); }
  ~DevBuf() { if (p) (void)hipFree(p); }
};

constexpr uint32_t kStages = (uint32_t)Stage::kCount;

}  // namespace

Shard PlanShard(uint32_t clip_frames, uint32_t world, uint32_t rank) {
  Shard s;
  if (world == 0 || rank >= world) return s;
  const uint32_t base = clip_frames / world, extra = clip_frames % world;
  s.frames = base + (rank < extra ? 1u : 0u);
  s.first_frame = rank * base + std::min(rank, extra);
  s.needs_halo = s.first_frame > 0 && s.frames > 0;
  s.pairs = s.frames == 0 ? 0 : s.frames - (s.needs_halo ? 0u : 1u);
  s.first_encoded = s.needs_halo ? s.first_frame : s.first_frame + 1;
  return s;
}

struct ClipEncoder::Impl {
  ClipEncoderConfig c;
  Shard sh;
  uint32_t pw = 0, ph = 0, mfw = 0, mfh = 0, blocks = 0, iters = 0;
  uint64_t pyr_stride = 0, frame_bytes = 0, plane_elems = 0, record_bytes = 0, seg_ws_bytes = 0;
  // Pipelined schedule: RANSAC + segmentation of a step get `depth` iterations to finish, on `depth` streams (step l on
  // stream l % depth), and the small per-step buffers exist in depth + 2 sets (step s uses set s % nsets).
  static constexpr int kMaxDepth = 3, kSets = kMaxDepth + 2;
  int depth = 1, nsets = 1;
  // Chunks (round 6).  A step's pass over the shard may be cut into consecutive runs of frame pairs, and what the pipeline moves is a
  // MICRO-STEP = (step, chunk): every stage launch covers one chunk.  One chunk per step is the schedule of rounds 2-5 (a pipeline over
  // consecutive whole-shard steps).  With more, the stages of ONE step overlap each other -- RANSAC + segmentation of chunk c run beside the
  // motion search of chunk c + 1 and the transform of chunk c - 1 -- so a clip that is encoded ONCE (LoadFrames, Step, Sync) no longer pays
  // the latency-bound stages end to end.  Buffers: a chunk lives at its offset p0 inside set (step % nsets); events and "pending" flags are
  // per micro-step slot m % nsets (at most depth + 2 micro-steps are in flight).
  uint32_t nch = 1, cp = 0;  // the configured plan (chunk_pairs); a step may be cut differently (Step(): the idle-pipeline rule), so a
  // micro-step carries its own description from the moment it enters the pipeline until its last stage has been enqueued
  // order: who decides whether the micro-step reads its frames once -- 0 the policy, 1 two passes, 2 once (speculating BLIND: Step()'s
  // mixed form of a step into an empty pipeline)
  // frames: the shard's B,G,R frames this micro-step's step encodes -- the resident buffer (LoadFrames) or the caller's own device buffer
  // (StepFrames: a stream of clips, each encoded once, without a copy and without draining the pipeline between them)
  struct Micro { uint32_t step = 0, p0 = 0, pn = 0; bool first = true; uint8_t order = 0; const uint8_t* frames = nullptr; };
  static constexpr int kRing = 8;  // > depth + 2 micro-steps in flight
  Micro ring[kRing], next_micro;
  uint32_t n_steps = 0;
  const uint8_t* step_frames = nullptr;   // frames of the step being submitted (Step: the resident buffer; StepFrames: the caller's)
  // "the frames of step s are no longer read": an event on the main stream behind the step's last transform (which has joined the step's
  // RANSAC + segmentation + redo by then); kRing of them, by step -- a slot reused by a later step stands for the earlier one too (stream order)
  hipEvent_t e_step[kRing] = {};
  uint64_t step_last_micro[kRing] = {};  // the micro-step whose transform is the step's last reader
  void MarkStepDone(uint64_t d) {        // called where the transform of micro-step d has just been enqueued
    const uint32_t s = StepOf(d);
    if (step_last_micro[s % kRing] == d) Hip(hipEventRecord(e_step[s % kRing], sM), "hipEventRecord");
  }
  const Micro& At(uint64_t m) const { return ring[(int)(m % (uint64_t)kRing)]; }
  uint32_t StepOf(uint64_t m) const { return At(m).step; }
  uint32_t P0(uint64_t m) const { return At(m).p0; }  // first pair of the micro-step's chunk
  uint32_t Pn(uint64_t m) const { return At(m).pn; }  // its pairs
  int Slot(uint64_t m) const { return (int)(m % (uint64_t)nsets); }
  hipStream_t sM = nullptr, sL[kMaxDepth] = {nullptr, nullptr, nullptr}, sC = nullptr;
  DevBuf<uint8_t> bgr, pyr[2], mask[kSets], seg_ws[kMaxDepth], records[kSets + 1];
  DevBuf<float> mv[kSets], mad[kSets], gm[kSets], rmse[kSets], coeffs[kSets + 1];
  DevBuf<uint8_t> redo_ws[kMaxDepth];  // spec_quant: the foreground list of the step being finished, one per latency stream (as seg_ws)
  DevBuf<uint32_t> count[kSets], types[kSets], samples;
  hipEvent_t e_pyr[2] = {nullptr, nullptr}, e_halo[2] = {nullptr, nullptr}, e_fork = nullptr, e_join[kSets] = {}, e_rfork = nullptr, e_rmse[kSets] = {};
  bool halo_recorded[2] = {false, false}, join_pending[kSets] = {}, rmse_pending[kSets] = {};
  bool defer_rmse = false;  // pipelined, large fields: RANSAC leaves its in-order RMSE sum to a later kernel (nothing downstream
                            // needs it): on sC beside the segmentation at world 1, on the latency stream behind it on a multi-rank run
  uint64_t iter = 0, fork_iter[kSets] = {};
  void* comm = nullptr;
  HaloFn halo;
  // pipeline progress: steps whose stage has been enqueued
  uint64_t n_luma = 0, n_hbma = 0, n_lat = 0, n_dct = 0;
  // RANSAC + segmentation fork off the main stream where the previous iteration's main-stream work ends (in front of the
  // luma launch), not behind the motion search: measured at C3 (ms per step, late / early): 150 frames 1.353 / 1.310, 75
  // frames 0.726 / 0.689, 38 frames 0.419 / 0.376 (round 2); whole clips on the round-3 build: C3 2.41 / 2.38, C3b 2.52 /
  // 2.50, C5 2.35 / 2.34, and C5's MAD kernel runs alone (0.272 -> 0.238 ms): profiles/r03_ab_fork_full.txt, r03_ab_fork_4k.txt.
  // pipelined schedule: RANSAC + segmentation run beside the main stream's kernels and ask for shapes that fit there
  uint32_t lat_flags = 0;
  bool fused_records = false;  // wire output straight from the transform kernel (square transform blocks)
  bool one_bgr_pass = false;   // wire, tuned transform blocks: records + luma plane from ONE kernel at the front of the step, type words
                               // stored once the step's region ids exist (clip_encoder.hpp); records then exist in `rec_sets` sets
  // nsets + 1 of them: the front of step s + nsets + 1 rewrites the set in the iteration AFTER the one whose last launch completed step s's
  // records (with nsets it would be the same iteration, and the front runs first)
  int rec_sets = 1;
  DevBuf<uint8_t>& Records(uint64_t m) { return records[(int)(StepOf(m) % (uint32_t)rec_sets)]; }
  // The same for planes + quant: the transform runs at the front of the step with every tile quantised as background and leaves the luma
  // plane (svc_hip_dct_quant_luma_frames); the tiles of foreground MV blocks are redone with fg_step where the transform used to run
  // (svc_hip_dct_quant_redo_frames).  Coefficient planes then exist in rec_sets sets too.
  bool spec_quant = false;  // the configuration CAN speculate; whether a step does is spec_step[]
  // The extra coefficient sets are allocated the first time the policy decides to speculate (a clip that never does -- C3b and C5 measure
  // 13 % foreground -- keeps one set: 4 x 7.5 GB at C3, 4 x 6.3 GB at C5 that round 5 allocated for nothing); until then every step uses set 0.
  int coeff_sets = 1;  // sets that exist
  DevBuf<float>& Coeffs(uint64_t m) { return coeffs[coeff_sets > 1 ? (int)(StepOf(m) % (uint32_t)coeff_sets) : 0]; }
  // Called at the top of an iteration that is about to speculate for the first time, before anything of it is enqueued: the pipeline is
  // drained first (every earlier micro-step wrote, or will write, set 0 -- once their launches are all in the streams, stream order keeps
  // the rotation that starts now behind them).  If the memory is not there, the shard simply never speculates.
  bool GrowCoeffSets() {
    if (coeff_sets == rec_sets) return true;
    while (n_dct < n_luma) Iterate(false, last_timed);
    for (int b = 1; b < rec_sets; ++b) {
      float* q = nullptr;
      if (hipMalloc(reinterpret_cast<void**>(&q), std::max<size_t>(coeffs[0].n, 1) * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();  // not sticky: clear it
        for (int k = 1; k < b; ++k) { (void)hipFree(coeffs[k].p); coeffs[k].p = nullptr; coeffs[k].n = 0; }
        spec_quant = false;       // two passes from here on
        return false;
      }
      coeffs[b].p = q; coeffs[b].n = coeffs[0].n;
    }
    coeff_sets = rec_sets;
    return true;
  }
  bool spec_step[kSets + 1] = {};  // micro-step m speculated (a ring over the micro-steps in flight)
  bool& SpecStep(uint64_t m) { return spec_step[(int)(m % (uint64_t)(kSets + 1))]; }
  // The policy's feedback: after the segmentation of a step a counting kernel + a 4-byte copy leave the step's number of foreground MV
  // blocks in pinned host memory; nobody waits for it -- a step decides on the newest count that has arrived by then.
  static constexpr int kFgSlots = 8;
  static constexpr double kSpecMaxShare = 0.02;
  // small shards do not pay: the front-of-step transform costs a fixed ~10 us more than it saves below ~25 frames of 1080p
  // (profiles/r05_ab_speculative_quant.txt: 1080p shards of 150 / 75 / 38 / 19 frames -6 / -4 / -2 / +-1 %; C2's 29 frames of 720p +5 %)
  static constexpr uint64_t kSpecMinPixels = 50000000ull;  // encoded frames x padded pixels of the shard
  // The driver's own choice (chunk_pairs = 0) for steps that FOLLOW EACH OTHER is one chunk: whole-shard launches, the schedule of rounds
  // 2-5.  Measured (profiles/r06_ab_chunks.txt, r06_z_final_*): two chunks at C3 encode a clip once in 2.48-2.53 ms instead of 2.55-2.64, but
  // cost the steady state 0-4 % box to box (RANSAC + segmentation are latency-bound, so two chunks hold CUs beside the bandwidth kernels
  // twice as long, and the pyramid pass pays the write-back behind the transform twice); three and five chunks lose more; C5 loses 4 % to two.
  // Hence the idle-pipeline rule (ClipEncoder::Step): only a step that finds the pipeline empty is cut in two.
  static constexpr uint64_t kChunkMinPixels = 300000000ull;  // (what an automatic choice for every step would ask of a chunk)
  // the idle-pipeline rule applies from this many pixels x frames per shard: C3 / C3b (625 M: - 3.6 / - 4.8 %) and C5 (522 M: - 4.5 %, profiles/
  // r06_ab_chunks.txt: 2.83-2.85 -> 2.70-2.73 ms) were measured; smaller shards were not
  static constexpr uint64_t kIdleRuleMinPixels = 400000000ull;
  static constexpr uint32_t kMaxAutoChunks = 1;              // chunks of a step that follows another one
  DevBuf<uint32_t> fg_dev;
  uint32_t* fg_host = nullptr;
  hipEvent_t e_fg[kFgSlots] = {};
  bool fg_pending[kFgSlots] = {};
  uint64_t fg_blocks[kFgSlots] = {};  // MV blocks the slot's count was taken over (a chunk's)
  uint64_t n_fg = 0;
  double fg_share = -1.0;  // newest foreground share known (-1: none yet)
  uint64_t n_spec = 0, n_decided = 0;  // micro-steps that speculated / that had the choice (PolicyInfo)
  void MeasureForeground(uint64_t m, hipStream_t st) {
    if (!spec_quant || c.two_bgr_passes || c.always_speculate || !Pn(m)) return;  // only the adaptive policy asks
    const int slot = (int)(n_fg % kFgSlots);
    if (fg_pending[slot] && !Ready(e_fg[slot])) return;  // eight measurements in flight: skip this one
    Abi(svc_hip_count_foreground(types[Set(m)].p + (uint64_t)P0(m) * blocks, (uint64_t)Pn(m) * blocks, fg_dev.p + slot, st), "svc_hip_count_foreground");
    Hip(hipMemcpyAsync(fg_host + slot, fg_dev.p + slot, 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
    Hip(hipEventRecord(e_fg[slot], st), "hipEventRecord");
    fg_pending[slot] = true;
    fg_blocks[slot] = (uint64_t)Pn(m) * blocks;
    ++n_fg;
  }
  void PollForeground() {  // the newest measurement that has landed, if any, becomes the share the policy acts on; never waits
    for (uint64_t k = n_fg; k > 0 && k + kFgSlots > n_fg; --k) {  // newest first
      const int slot = (int)((k - 1) % kFgSlots);
      if (!fg_pending[slot]) continue;
      if (!Ready(e_fg[slot])) continue;
      fg_share = (double)fg_host[slot] / (double)fg_blocks[slot];
      break;
    }
  }
  bool DecideSpeculation(uint64_t m) {
    if (!spec_quant || c.two_bgr_passes) return false;
    ++n_decided;
    bool yes = c.always_speculate;
    if (c.random_policy) {
      // test switch, whatever was measured: step 0 never, step 1 on every chunk but its first (the first speculation of a shard -- it
      // allocates the extra coefficient sets and starts their rotation -- then falls INSIDE a step whose set is not set 0), later steps by
      // a fixed pseudo-random sequence over the chunk launches (a strict alternation would lock onto the number of chunks per step)
      const uint32_t s = StepOf(m);
      yes = s == 0 ? false : s == 1 ? !At(m).first : (Hash32(n_decided * 0x9E3779B1ull) & 1u) != 0;
    } else if (!yes) {
      PollForeground();
      yes = fg_share >= 0.0 && fg_share <= kSpecMaxShare;
    }
    n_spec += yes;
    return yes;
  }
  // Would a step enqueued now read the BGR clip once?  (no counters touched: Step()'s chunk plan asks before the micro-steps decide)
  bool WouldReadOnce() {
    if (one_bgr_pass) return true;
    if (!spec_quant || c.two_bgr_passes) return false;
    if (c.always_speculate) return true;
    PollForeground();
    return fg_share >= 0.0 && fg_share <= kSpecMaxShare;
  }
  void ResetPolicy() {  // what the policy knew is void (the caller has Sync()ed: every measurement has landed)
    fg_share = -1.0;
    n_fg = 0;
    for (bool& p : fg_pending) p = false;
  }
  bool last_timed = false;  // Flush() times the rest of a step that was submitted timed
  // timing
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed[kStages];
  std::vector<hipEvent_t> event_pool;

  ~Impl() {
    for (hipStream_t s : {sM, sL[0], sL[1], sL[2], sC})
      if (s) (void)hipStreamSynchronize(s);
    for (auto& v : timed)
      for (auto& pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : {e_pyr[0], e_pyr[1], e_halo[0], e_halo[1], e_fork, e_rfork, e_join[0], e_join[1], e_join[2], e_join[3], e_join[4],
                         e_rmse[0], e_rmse[1], e_rmse[2], e_rmse[3], e_rmse[4]})
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : e_fg)
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : e_step)
      if (e) (void)hipEventDestroy(e);
    if (fg_host) (void)hipHostFree(fg_host);
    for (hipStream_t s : {sM, sL[0], sL[1], sL[2], sC})
      if (s) (void)hipStreamDestroy(s);
  }

  hipEvent_t TimingEvent() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    Hip(hipEventCreate(&e), "hipEventCreate");
    return e;
  }

  // Runs fn with HIP events around it on `stream` when timing is on.
  uint64_t timed_pairs[kStages] = {};  // frame pairs the timed launches of a stage covered (a step's launches are its chunks)
  uint64_t cur_pairs = 0;              // pairs of the micro-step whose stages are being enqueued
  template <typename F> void Run(Stage st, hipStream_t stream, bool timing, F&& fn) {
    if (!timing) { fn(); return; }
    timed_pairs[(uint32_t)st] += cur_pairs;
    hipEvent_t a = TimingEvent(), b = TimingEvent();
    Hip(hipEventRecord(a, stream), "hipEventRecord");
    fn();
    Hip(hipEventRecord(b, stream), "hipEventRecord");
    timed[(uint32_t)st].emplace_back(a, b);
  }

  // pyramid set of micro-step m's step: the pipelined schedule alternates two, the serial one has one
  int Par(uint64_t m) const { return c.schedule == Schedule::kPipelined ? (int)(StepOf(m) & 1u) : 0; }
  // set of the small per-step buffers (motion field, RANSAC outputs, region ids) of micro-step m's step
  int Set(uint64_t m) const { return (int)(StepOf(m) % (uint32_t)nsets); }

  // ---- the stages; `m` is the micro-step, its buffers are chunk P0(m) .. + Pn(m) of the sets Par(m) / Set(m) ---------------
  // own frame j lives in pyramid slot 1 + j; encoded frame (pair) p is own frame p + skip (skip = 1: frame 0 of the clip is tracked only,
  // skip = 0: the tracked frame of pair 0 is the halo in slot 0)
  uint32_t Skip() const { return sh.needs_halo ? 0u : 1u; }
  bool decided = false;  // this micro-step speculates
  void Decide(uint64_t m) {
    if (At(m).order && spec_quant && Pn(m)) {
      // Step() chose (the mixed form of a step into an empty pipeline).  No extra coefficient sets are needed for it: both chunks belong to
      // ONE step, so they write different ranges of the same set, every earlier step's launches are already in the streams in front of
      // them, and a later step that speculates by the policy grows the sets behind a drained pipeline (GrowCoeffSets) as ever.
      decided = At(m).order == 2;
      ++n_decided;
      n_spec += decided;
      return;
    }
    decided = spec_quant && Pn(m) && DecideSpeculation(m);
    // The rotation over the extra coefficient sets starts with a STEP, never inside one: the chunks of a step that came before this one
    // have written (or, drained by GrowCoeffSets, will write) set 0, and a step's planes must not end up in two sets.  So the first
    // speculation of a shard waits for a step's first chunk (found by tests/helpers/driver_fuzz.py: chunked steps, the foreground count of
    // the previous step landing between the decisions of two chunks of one step).
#ifndef SVC_CLIP_ALLOW_MIDSTEP_ROTATION  // 1: the behaviour before the fix, to show that test_the_policy_may_flip_at_every_chunk sees it
#define SVC_CLIP_ALLOW_MIDSTEP_ROTATION 0
#endif
    if (decided && coeff_sets != rec_sets && ((!At(m).first && !SVC_CLIP_ALLOW_MIDSTEP_ROTATION) || !GrowCoeffSets())) {
      decided = false;
      --n_spec;
    }
  }
  template <typename Between> void Luma(uint64_t m, hipStream_t st, bool timing, Between&& between) {
    cur_pairs = Pn(m);
    const int b = Par(m);
    const uint32_t p0 = P0(m), pn = Pn(m), skip = Skip();
    const bool first_chunk = At(m).first;
    if (spec_quant) SpecStep(m) = pn ? decided : false;  // decided at the top of the iteration (Iterate / SerialStep)
    if (pn && (one_bgr_pass || (spec_quant && SpecStep(m)))) {
      const uint8_t* enc = At(m).frames + (uint64_t)(skip + p0) * frame_bytes;
      uint8_t* slots = pyr[b].p + (uint64_t)(1 + skip + p0) * pyr_stride;
      Run(Stage::kTransform, st, timing, [&] {
        if (!one_bgr_pass)
          Abi(svc_hip_dct_quant_luma_frames(enc, frame_bytes, pn, pw, ph, c.dct_block_w, c.bg_step,
                                            Coeffs(m).p + (uint64_t)p0 * 3 * plane_elems, slots, pyr_stride, st), "svc_hip_dct_quant_luma_frames");
        else
          Abi(svc_hip_dct_records_luma_frames(enc, frame_bytes, pn, pw, ph, c.dct_block_w, ph,
                                              Records(m).p + (uint64_t)p0 * record_bytes, record_bytes, slots, pyr_stride, st), "svc_hip_dct_records_luma_frames");
      });
      between();  // (search_after_transform: the motion search of the previous micro-step goes here, right behind the transform)
      cur_pairs = pn;
      Run(Stage::kLumaPyramid, st, timing, [&] {
        if (skip && first_chunk)  // the tracked-only first frame of the clip has no records: its pyramid the usual way
          Abi(svc_hip_luma_pyramid_frames(At(m).frames, frame_bytes, 1, pw, ph, c.levels, pyr[b].p + pyr_stride, pyr_stride, st),
              "svc_hip_luma_pyramid_frames");
        Abi(svc_hip_pyramid_levels_frames(slots, pyr_stride, pn, pw, ph, c.levels, st), "svc_hip_pyramid_levels_frames");
      });
      return;
    }
    // the chunk's own frames (the first chunk of a shard without a halo also holds the clip's tracked-only frame 0)
    const uint32_t f0 = first_chunk ? 0u : p0 + skip, f1 = p0 + pn + skip;
    if (f1 <= f0) return;
    Run(Stage::kLumaPyramid, st, timing, [&] {
      Abi(svc_hip_luma_pyramid_frames(At(m).frames + (uint64_t)f0 * frame_bytes, frame_bytes, f1 - f0, pw, ph, c.levels,
                                      pyr[b].p + (uint64_t)(1 + f0) * pyr_stride, pyr_stride, st), "svc_hip_luma_pyramid_frames");
    });
  }

  // my last pyramid -> rank + 1's slot 0; slot 0 <- rank - 1's last pyramid  (world > 1: one chunk per step)
  void Halo(uint64_t m, bool timing) {
    cur_pairs = 0;
    const int b = Par(m);
    Hip(hipEventRecord(e_pyr[b], sM), "hipEventRecord");
    Hip(hipStreamWaitEvent(sC, e_pyr[b], 0), "hipStreamWaitEvent");
    Run(Stage::kHalo, sC, timing, [&] {
      const uint8_t* send = pyr[b].p + (uint64_t)sh.frames * pyr_stride;
      if (halo) halo(send, pyr[b].p, pyr_stride, sC);
      else if (comm) Abi(svc_hip_halo_shift(comm, send, pyr[b].p, pyr_stride, c.rank, c.world, 0, sC), "svc_hip_halo_shift");
      else throw std::runtime_error("svc::ClipEncoder: world > 1 needs SetComm() or SetHaloTransport() before Step()");
    });
    Hip(hipEventRecord(e_halo[b], sC), "hipEventRecord");
    halo_recorded[b] = true;
  }

  void Hbma(uint64_t m, hipStream_t st, bool timing) {
    cur_pairs = Pn(m);
    const uint32_t p0 = P0(m), pn = Pn(m);
    if (!pn) return;
    const int b = Par(m), q = Set(m), k = Slot(m);
    const uint64_t t0 = (sh.needs_halo ? 0 : 1) + p0;  // slot of the chunk's first tracked pyramid
    if (rmse_pending[k]) {  // a deferred RMSE kernel may still read the motion field this rewrites: the slot's newest one is later on its
      Hip(hipStreamWaitEvent(st, e_rmse[k], 0), "hipStreamWaitEvent");  // stream than the one of this chunk's previous step
      rmse_pending[k] = false;
    }
    Run(Stage::kHbma, st, timing, [&] {
      Abi(svc_hip_hbma_pairs(pyr[b].p + t0 * pyr_stride, pyr[b].p + (t0 + 1) * pyr_stride, pyr_stride, pn, c.levels, pw, ph,
                             c.search_range, c.mv_block, c.mv_block, mv[q].p + (uint64_t)p0 * blocks * 2, mad[q].p + (uint64_t)p0 * blocks,
                             c.hbma_flags, st), "svc_hip_hbma_pairs");
    });
  }

  // RANSAC + region ids: one workgroup per frame, latency-bound
  void Lat(uint64_t m, hipStream_t st, bool timing) {
    cur_pairs = Pn(m);
    const uint32_t p0 = P0(m), pn = Pn(m);
    if (!pn) return;
    const int b = Set(m), k = Slot(m);
    DevBuf<uint8_t>& ws = seg_ws[c.schedule == Schedule::kPipelined ? (int)(m % (uint64_t)depth) : 0];  // one per stream
    const uint64_t g0 = sh.first_encoded - 1 + p0;  // clip-wide index of the chunk's first pair
    float* gm_c = gm[b].p + (uint64_t)p0 * 2;
    float* rmse_c = rmse[b].p + p0;
    const float* mv_c = mv[b].p + (uint64_t)p0 * blocks * 2;
    uint8_t* mask_c = mask[b].p + (uint64_t)p0 * blocks;
    uint32_t* count_c = count[b].p + p0;
    uint32_t* types_c = types[b].p + (uint64_t)p0 * blocks;
    const uint32_t* samples_c = samples.p + (uint64_t)p0 * iters * c.ransac.subset_sz;
    Run(Stage::kRansac, st, timing, [&] {
      Hip(hipMemsetAsync(gm_c, 0, (size_t)pn * 2 * sizeof(float), st), "hipMemsetAsync");  // in/out, libs/motion.cpp:241-242
      Abi(svc_hip_ransac_frames_ex(mv_c, blocks, pn, c.ransac, samples_c, iters, gm_c, rmse_c, mask_c, count_c,
                                   lat_flags | (defer_rmse ? SVC_LAUNCH_DEFER_RMSE : 0u), st), "svc_hip_ransac_frames");
    });
    if (defer_rmse && c.world == 1) {
      // the serial tail of RANSAC (one dependent f32 add per MV block) beside the segmentation: it reads what the launch
      // above left (gm, mask, count: final) and is waited for only where the set's motion field is next rewritten (Hbma).
      // On the COMMUNICATION stream, not a stream of its own: HIP multiplexes streams onto four hardware queues, and a
      // fifth stream shares the main stream's queue -- its 0.13 ms single-wave kernel then holds every main-stream kernel
      // back (measured: C5 2.43 -> 2.95 ms per step).  Only while that stream carries no halo (one rank): on a multi-rank
      // run the next step's ncclSend / ncclRecv would queue behind this rank's RANSAC + this kernel and stall the
      // NEIGHBOUR's receive with it (round 3's ADVICE) -- there the kernel goes behind the segmentation instead (ForkLat).
      Hip(hipEventRecord(e_rfork, st), "hipEventRecord");
      Hip(hipStreamWaitEvent(sC, e_rfork, 0), "hipStreamWaitEvent");
      Abi(svc_hip_ransac_rmse_frames(mv_c, blocks, pn, c.ransac, gm_c, mask_c, count_c, rmse_c, sC), "svc_hip_ransac_rmse_frames");
      Hip(hipEventRecord(e_rmse[k], sC), "hipEventRecord");
      rmse_pending[k] = true;
    }
    Run(Stage::kSegment, st, timing, [&] {
      if (c.segmentation)
        Abi(svc_hip_segment_frames_ex(mask_c, mv_c, mfw, mfh, pn, c.mv_block, c.mv_block, c.segment,
                                      c.seed * 1000003ull + g0, ws.p, seg_ws_bytes, types_c, lat_flags, st),
            "svc_hip_segment_frames");
      else
        Abi(svc_hip_block_types_frames(mask_c, blocks, pn, types_c, st), "svc_hip_block_types_frames");
    });
    MeasureForeground(m, st);
    if (c.schedule == Schedule::kPipelined && OnePassStep(m)) FinishOnePass(m, st, timing);
  }

  bool OnePassStep(uint64_t m) { return Pn(m) && (one_bgr_pass || (spec_quant && SpecStep(m))); }

  // What a one-pass step still owes once its region ids exist: the type words of the records it emitted at its front (wire), or the tiles
  // of its foreground MV blocks once more with fg_step (planes).  Latency-bound (a list, a few thousand scattered tiles): in the pipelined
  // schedule it runs on the latency stream right behind the segmentation, beside the main stream's kernels, and the main stream only joins.
  void FinishOnePass(uint64_t m, hipStream_t st, bool timing) {
    cur_pairs = Pn(m);
    const uint32_t p0 = P0(m), pn = Pn(m);
    if (!pn || !c.dct_block_w) return;
    const int b = Set(m);
    const uint8_t* enc = At(m).frames + (uint64_t)(Skip() + p0) * frame_bytes;  // encoded frame of pair p: own frame p + skip
    const uint32_t* types_c = types[b].p + (uint64_t)p0 * blocks;
    DevBuf<uint8_t>& rws = redo_ws[c.schedule == Schedule::kPipelined ? (int)(m % (uint64_t)depth) : 0];
    Run(Stage::kTypePatch, st, timing, [&] {
      if (one_bgr_pass)
        Abi(svc_hip_wire_patch_types_frames(types_c, pn, pw, ph, ph, c.dct_block_w, c.mv_block, c.mv_block,
                                            Records(m).p + (uint64_t)p0 * record_bytes, record_bytes, 0, st), "svc_hip_wire_patch_types_frames");
      else
        Abi(svc_hip_dct_quant_redo_frames(enc, frame_bytes, pn, pw, ph, c.dct_block_w, types_c, c.mv_block, c.mv_block, c.fg_step,
                                          Coeffs(m).p + (uint64_t)p0 * 3 * plane_elems, rws.p, rws.bytes(), st), "svc_hip_dct_quant_redo_frames");
    });
  }

  void Transform(uint64_t m, hipStream_t st, bool timing) {
    cur_pairs = Pn(m);
    const uint32_t p0 = P0(m), pn = Pn(m);
    if (!pn || !c.dct_block_w) return;
    const int b = Set(m);
    const uint8_t* enc = At(m).frames + (uint64_t)(Skip() + p0) * frame_bytes;  // encoded frame of pair p: own frame p + skip
    const uint32_t* types_c = types[b].p + (uint64_t)p0 * blocks;
    if (OnePassStep(m)) {  // the transform ran at the front of the step; its finish follows the segmentation (pipelined: on that stream, Lat)
      if (c.schedule != Schedule::kPipelined) FinishOnePass(m, st, timing);
      return;
    }
    Run(Stage::kTransform, st, timing, [&] {
      // records carry RAW coefficients, as the reference's encoder serialises them (libs/encoder.cpp:638-650:
      // the decoder picks the step per tile, libs/decoder.cpp:130-135); planes carry the quantised ones
      if (c.wire && fused_records)
        Abi(svc_hip_dct_records_frames(enc, frame_bytes, pn, pw, ph, c.dct_block_w, types_c, c.mv_block, c.mv_block,
                                       0, 0, ph, Records(m).p + (uint64_t)p0 * record_bytes, record_bytes, st), "svc_hip_dct_records_frames");
      else if (c.wire) {  // any other transform block: Dct, then SerializeEncodedFrame
        float* planes = coeffs[0].p + (uint64_t)p0 * 3 * plane_elems;
        Abi(svc_hip_dct_frames(enc, frame_bytes, pn, pw, ph, c.dct_block_w, c.dct_block_h, planes, st), "svc_hip_dct_frames");
        Abi(svc_hip_serialize_frames(planes, plane_elems, pn, types_c, pw, ph, c.dct_block_w, c.dct_block_h, mfw, mfh,
                                     c.mv_block, c.mv_block, Records(m).p + (uint64_t)p0 * record_bytes, record_bytes, st), "svc_hip_serialize_frames");
      } else
        Abi(svc_hip_dct_quant_frames(enc, frame_bytes, pn, pw, ph, c.dct_block_w, c.dct_block_h, types_c, c.mv_block,
                                     c.mv_block, c.fg_step, c.bg_step, Coeffs(m).p + (uint64_t)p0 * 3 * plane_elems, st), "svc_hip_dct_quant_frames");
    });
  }

  // RANSAC + segmentation of micro-step l on its latency stream, behind everything the main stream holds so far; the main
  // stream picks the result up (JoinLat) only where it is needed: in front of the transform of l, `depth` iterations later.
  void ForkLat(uint64_t l, bool timing) {
    hipStream_t st = sL[l % (uint64_t)depth];
    Hip(hipEventRecord(e_fork, sM), "hipEventRecord");
    Hip(hipStreamWaitEvent(st, e_fork, 0), "hipStreamWaitEvent");
    Lat(l, st, timing);
    Hip(hipEventRecord(e_join[Slot(l)], st), "hipEventRecord");
    join_pending[Slot(l)] = true;
    fork_iter[Slot(l)] = iter;
    if (defer_rmse && c.world > 1 && Pn(l)) {
      // multi-rank: the RMSE tail on this latency stream BEHIND the region ids (and behind the join event, so the transform
      // does not wait for it); the communication stream stays free for the halo
      const int b = Set(l);
      const uint32_t p0 = P0(l), pn = Pn(l);
      Abi(svc_hip_ransac_rmse_frames(mv[b].p + (uint64_t)p0 * blocks * 2, blocks, pn, c.ransac, gm[b].p + (uint64_t)p0 * 2,
                                     mask[b].p + (uint64_t)p0 * blocks, count[b].p + p0, rmse[b].p + p0, st), "svc_hip_ransac_rmse_frames");
      Hip(hipEventRecord(e_rmse[Slot(l)], st), "hipEventRecord");
      rmse_pending[Slot(l)] = true;
    }
  }
  void JoinLat(uint64_t l) {
    if (!join_pending[Slot(l)]) return;
    Hip(hipStreamWaitEvent(sM, e_join[Slot(l)], 0), "hipStreamWaitEvent");
    join_pending[Slot(l)] = false;
  }

  // One iteration of the software pipeline over micro-steps.  Multi-rank (a halo to wait for): luma + pyramid of m, motion search of
  // m - 1, fork of RANSAC + segmentation of m - 2, transform of m - 2 - depth.  One rank: the motion search of m follows its own pyramids
  // in the same iteration (nothing to wait for), so RANSAC + segmentation of m - 1 fork at the start of the next one and the transform
  // of m - 1 - depth joins them.  Buffer hazards (a micro-step's events live in slot m % (depth + 2), its data at its chunk's offset
  // in set step % (depth + 2)):
  //   lat(l) reads mv[l], writes mask / types[l] on stream l % depth, and must be done before transform(l) reads types[l]:
  //     joined there, `depth` iterations after its fork;
  //   hbma(h) writes its chunk of mv[set]: the last reader of that range, lat of the same chunk nsets steps earlier, was joined long before;
  //   up to `depth` lats are in flight on their own streams, each with its own segmentation workspace.
  void Iterate(bool new_step, bool timing) {
    if (new_step) {
      ring[(int)(n_luma % (uint64_t)kRing)] = next_micro;
      Decide(n_luma);  // may drain the pipeline (the first speculation allocates the extra coefficient sets)
    }
    const uint64_t lumas_before = n_luma, hbmas = n_hbma, lats = n_lat;
    const bool do_lat = n_lat < hbmas;
    const uint64_t l = n_lat, d = n_dct;
    // draining (no new step) joins at once; otherwise the transform of micro-step d waits until lat(d) has had its iterations
    const bool do_dct = n_dct < lats && (!new_step || iter - fork_iter[Slot(d)] >= (uint64_t)depth);
    // forked where the previous iteration's main-stream work ends -- or (fork_behind_front, A/B) behind the front-of-step transform of a
    // micro-step that reads its frames once, so that RANSAC + segmentation run beside the pyramid pass and the motion search instead of beside
    // the store-bound transform's first third
    const bool fork_late = c.fork_behind_front && c.world == 1 && new_step && next_micro.pn && (one_bgr_pass || (spec_quant && decided));
    if (do_lat && !fork_late) ForkLat(l, timing);
    // search_after_transform (one rank; A/B): whatever kernel follows the transform shares the memory system with the write-back of what it
    // left dirty (0.03 ms at C3, profiles/r06_ab_pyr_strip.txt).  With this switch that kernel is the motion search instead of a pyramid
    // pass: a one-pass micro-step runs transform(m) | search(m - 1) | pyramid levels(m) (the search one micro-step behind, as on a
    // multi-rank run), a two-pass one luma(m) | transform(m - 3) | search(m).
    const bool reorder = c.search_after_transform && c.world == 1;
    bool hbma_done = false;
    auto pending_searches = [&](uint64_t upto) {
      while (n_hbma < upto) { Hbma(n_hbma, sM, timing); ++n_hbma; }
    };
    if (new_step) {
      const uint64_t m = n_luma;
      const int b = Par(m);
      if (c.world > 1 && halo_recorded[b])  // the send out of pyr[b] two steps ago must have left
        Hip(hipStreamWaitEvent(sM, e_halo[b], 0), "hipStreamWaitEvent");
      Luma(m, sM, timing, [&] {
        if (do_lat && fork_late) ForkLat(l, timing);
        if (reorder) { pending_searches(lumas_before); hbma_done = true; }
      });
      if (c.world > 1) Halo(m, timing);
      ++n_luma;
    }
    if (reorder) {
      if (do_dct) { JoinLat(d); Transform(d, sM, timing); MarkStepDone(d); }
      if (!hbma_done) pending_searches(n_luma);  // two-pass micro-step (or draining): behind the transform of this iteration
      n_lat += do_lat; n_dct += do_dct;
      ++iter;
      return;
    }
    // one rank: the search of the micro-step whose pyramids were just enqueued; multi-rank: the previous one's, whose halo has had a
    // whole iteration to arrive
    const bool do_hbma = (c.world > 1 && new_step) ? n_hbma < lumas_before : n_hbma < n_luma;
    const uint64_t h = n_hbma;
    if (do_hbma && c.world > 1) Hip(hipStreamWaitEvent(sM, e_halo[Par(h)], 0), "hipStreamWaitEvent");
    if (do_hbma) Hbma(h, sM, timing);
    if (do_dct) {
      JoinLat(d);
      Transform(d, sM, timing);
      MarkStepDone(d);
    }
    n_hbma += do_hbma; n_lat += do_lat; n_dct += do_dct;
    ++iter;
  }

  void SerialStep(bool timing) {
    const uint64_t s = n_luma;
    ring[(int)(s % (uint64_t)kRing)] = Micro{n_steps++, 0u, sh.pairs, true, 0, step_frames};
    step_last_micro[(n_steps - 1) % kRing] = s;
    Decide(s);
    const int b = Par(s);
    if (c.world > 1 && halo_recorded[b]) Hip(hipStreamWaitEvent(sM, e_halo[b], 0), "hipStreamWaitEvent");
    Luma(s, sM, timing, [] {});
    if (c.world > 1) {
      Halo(s, timing);
      Hip(hipStreamWaitEvent(sM, e_halo[b], 0), "hipStreamWaitEvent");
    }
    Hbma(s, sM, timing);
    Lat(s, sM, timing);
    Transform(s, sM, timing);
    MarkStepDone(s);
    ++n_luma; ++n_hbma; ++n_lat; ++n_dct;
  }
};

ClipEncoder::ClipEncoder(const ClipEncoderConfig& config) : p_(new Impl) {
  Impl& m = *p_;
  m.c = config;
  const ClipEncoderConfig& c = m.c;
  if (!c.width || !c.height || !c.levels || c.levels > 16 || !c.mv_block || c.world == 0 || c.rank >= c.world)
    throw std::runtime_error("svc::ClipEncoder: invalid configuration");
  if (c.clip_frames < 2 || c.clip_frames < c.world)
    throw std::runtime_error("svc::ClipEncoder: a clip needs at least two frames and one frame per rank");
  if ((c.dct_block_w == 0) != (c.dct_block_h == 0))
    throw std::runtime_error("svc::ClipEncoder: transform block needs both sides");
  m.sh = PlanShard(c.clip_frames, c.world, c.rank);
  if (c.schedule == Schedule::kPipelined) {
    if (!c.standalone_shapes) m.lat_flags |= SVC_LAUNCH_BESIDE;
    if (!c.segment_fork) m.lat_flags |= SVC_LAUNCH_NO_FORK;
  }
  if (c.narrow_attempts) m.lat_flags |= SVC_LAUNCH_NO_WIDE;
  if (c.lat_depth > (uint32_t)Impl::kMaxDepth) throw std::runtime_error("svc::ClipEncoder: lat_depth must be 0..3");
  const uint32_t f = 1u << (c.levels - 1);
  m.pw = ClosestLargerDivisible(c.width, c.mv_block, f);   // libs/encoder.cpp:164-168
  m.ph = ClosestLargerDivisible(c.height, c.mv_block, f);
  m.mfw = m.pw / c.mv_block; m.mfh = m.ph / c.mv_block; m.blocks = m.mfw * m.mfh;
  m.pyr_stride = (svc_hip_pyramid_bytes(m.pw, m.ph, c.levels) + 255) / 256 * 256;
  m.frame_bytes = (uint64_t)m.pw * m.ph * 3;
  m.plane_elems = (uint64_t)m.pw * m.ph;
  const bool transform = c.dct_block_w != 0;
  m.record_bytes = (c.wire && transform) ? svc_hip_serialized_frame_bytes(m.pw, m.ph, c.dct_block_w, c.dct_block_h) : 0;
  m.iters = svc_hip_ransac_iter_count(c.ransac);
  const uint32_t P = m.sh.pairs, N = m.sh.frames;
  // (computed below, once the chunk size is known: the workspace of one launch)
  // All three at the default priority.  Measured on MI355X (38-frame shard, pipelined): raising the second and
  // the communication stream stretched the main stream's HBM-bound kernels 1.8x (0.41 -> 0.64 ms per step), and
  // a low-priority main stream was slower still (0.79).  Confining the second stream to every 2nd / 4th / 8th CU
  // (hipExtStreamCreateWithCUMask) cost as much: 0.37 -> 0.52-0.55 ms at 38 frames, 2.52 -> 2.76-2.78 ms at 300
  // (profiles/r02_cu_mask.txt).
  // How many iterations RANSAC + segmentation of a step may take (and how many streams they alternate on).  With the
  // shapes that fit beside the bandwidth kernels (fields up to 8 192 blocks) one iteration hides them unless the shard
  // is tiny; a 4K field keeps 1 024-lane workgroups that need whole CUs and get them only where a main-stream kernel
  // drains (profiles/r02_timeline_C5.txt), so its chain of four launches spans more than one iteration.  Two is never
  // worse and is what the tiny and the 4K shards need (ms per step at depth 1 / 2 / 3, profiles/r02_ab_lat_depth.txt:
  // C5 2.83 / 2.44 / 2.70, C5 8-frame shard 1.07 / 0.65 / 0.81, C3 2.57 / 2.56 / 2.56, C3 38 frames 0.371 / 0.370 /
  // 0.373, C3 19 frames 0.236 / 0.208 / 0.207).  ClipEncoderConfig::lat_depth overrides (A/B runs).
  const bool pipelined = c.schedule == Schedule::kPipelined;
  m.depth = 1;
  if (pipelined) m.depth = c.lat_depth ? (int)c.lat_depth : 2;
  m.nsets = !pipelined ? 1 : m.depth + 2;
  // Chunks per step.  Only where one rank holds the clip and the schedule pipelines (a multi-rank step is tied to its neighbour's by the
  // halo, and its shards are small already); default one (see kMaxAutoChunks), chunk_pairs asks for more.
  m.nch = 1;
  if (pipelined && c.world == 1 && P > 1) {
    if (c.chunk_pairs) m.nch = (P + c.chunk_pairs - 1) / c.chunk_pairs;
    else m.nch = (uint32_t)std::min<uint64_t>(Impl::kMaxAutoChunks, std::max<uint64_t>(1, (uint64_t)P * m.pw * m.ph / Impl::kChunkMinPixels));
    m.nch = std::min(m.nch, P);
  }
  m.cp = P ? (P + m.nch - 1) / m.nch : 0;
  if (m.cp) m.nch = (P + m.cp - 1) / m.cp;  // no empty chunk at the end
  // only where the chain is long enough to matter: 0.13 ms at 4K against 0.03 ms at 1080p, where it measures neutral
  // (profiles/r03_ab_defer_rmse.txt)
  m.defer_rmse = pipelined && !c.inline_rmse && m.blocks > 8192;
  Hip(hipStreamCreateWithFlags(&m.sM, hipStreamNonBlocking), "hipStreamCreate");
  Hip(hipStreamCreateWithFlags(&m.sC, hipStreamNonBlocking), "hipStreamCreate");
  if (m.defer_rmse) {
    Hip(hipEventCreateWithFlags(&m.e_rfork, hipEventDisableTiming), "hipEventCreate");
    for (int b = 0; b < m.nsets; ++b) Hip(hipEventCreateWithFlags(&m.e_rmse[b], hipEventDisableTiming), "hipEventCreate");
  }
  for (int k = 0; k < m.depth; ++k) Hip(hipStreamCreateWithFlags(&m.sL[k], hipStreamNonBlocking), "hipStreamCreate");
  for (hipEvent_t* e : {&m.e_pyr[0], &m.e_pyr[1], &m.e_halo[0], &m.e_halo[1], &m.e_fork})
    Hip(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate");
  for (int b = 0; b < m.nsets; ++b) Hip(hipEventCreateWithFlags(&m.e_join[b], hipEventDisableTiming), "hipEventCreate");
  for (hipEvent_t& e : m.e_step) Hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
  m.bgr.Alloc((size_t)N * m.frame_bytes);
  for (int b = 0; b < (pipelined ? 2 : 1); ++b) {
    m.pyr[b].Alloc((size_t)(N + 1) * m.pyr_stride);  // slot 0 = halo, slots 1..N = own frames
    Hip(hipMemset(m.pyr[b].p, 0, (size_t)(N + 1) * m.pyr_stride), "hipMemset");
  }
  for (int b = 0; b < m.nsets; ++b) {
    m.mv[b].Alloc((size_t)P * m.blocks * 2); m.mad[b].Alloc((size_t)P * m.blocks);
    m.gm[b].Alloc((size_t)P * 2); m.rmse[b].Alloc(P);
    m.mask[b].Alloc((size_t)P * m.blocks); m.count[b].Alloc(P); m.types[b].Alloc((size_t)P * m.blocks);
  }
  m.seg_ws_bytes = c.segmentation ? svc_hip_segment_workspace_bytes(m.mfw, m.mfh, std::max(m.cp, 1u), c.segment.attempt_count) : 0;
  for (int k = 0; k < m.depth; ++k) m.seg_ws[k].Alloc(m.seg_ws_bytes);
  m.fused_records = c.wire && c.dct_block_w == c.dct_block_h && c.dct_block_w <= 64 && c.dct_block_w % 2 == 0;
  // one pass over the BGR clip: the tuned record emitter (8x8 / 16x16 on widths that are whole 16-pixel segments) also leaves the luma plane
  m.one_bgr_pass = m.fused_records && !c.two_bgr_passes && transform && (c.dct_block_w == 8 || c.dct_block_w == 16) && m.pw % 16 == 0 &&
                   c.mv_block % c.dct_block_w == 0 && P > 0;
  // can this shard ever speculate?  (not when told never to; not -- unless told always to -- when it is too small to pay: then it keeps
  // one set of coefficients and measures nothing)
  m.spec_quant = !c.wire && !c.two_bgr_passes && transform && c.dct_block_w == c.dct_block_h && (c.dct_block_w == 8 || c.dct_block_w == 16) &&
                 m.pw % 16 == 0 && c.mv_block % 16 == 0 && c.mv_block % c.dct_block_w == 0 && c.fg_step > 0 && c.bg_step > 0 && P > 0 &&
                 (c.always_speculate || c.idle_rule_any_size || (uint64_t)P * m.pw * m.ph >= Impl::kSpecMinPixels);
  // a micro-step's output is written at its front and completed up to depth + 2 iterations later; the same chunk of the NEXT step that uses
  // the set comes nch iterations later per set: ceil((depth + 3) / nch) sets keep them apart (5 with whole-shard steps, 2 with three chunks)
  m.rec_sets = (m.one_bgr_pass || m.spec_quant) && pipelined ? (m.nsets + 1 + (int)m.nch - 1) / (int)m.nch : 1;
  if (transform) {
    if (c.wire)
      for (int b = 0; b < m.rec_sets; ++b) m.records[b].Alloc((size_t)P * m.record_bytes);
    if (!c.wire || !m.fused_records)
      m.coeffs[0].Alloc((size_t)P * 3 * m.plane_elems);  // the other rec_sets - 1 sets: Impl::GrowCoeffSets, on the first speculation
    if (m.spec_quant) {
      for (int k = 0; k < m.depth; ++k) m.redo_ws[k].Alloc(svc_hip_dct_redo_workspace_bytes(m.cp, m.pw, m.ph, c.mv_block, c.mv_block));
      m.fg_dev.Alloc(Impl::kFgSlots);
      Hip(hipHostMalloc(reinterpret_cast<void**>(&m.fg_host), Impl::kFgSlots * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
      for (hipEvent_t& e : m.e_fg) Hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    }
  }
  // RANSAC draws: distinct within an iteration, a function of (seed, clip frame, iteration) only --
  // the generator of stream_encoder.cpp / pipeline.ransac_samples, indexed by the CLIP-wide pair
  {
    const uint64_t g0 = m.sh.first_encoded - 1;
    const size_t n = (size_t)P * m.iters * c.ransac.subset_sz;
    std::vector<uint32_t> h(std::max<size_t>(n, 1));
    const uint32_t div = std::max<uint32_t>(1, (m.blocks - 1) / std::max<uint32_t>(1, c.ransac.subset_sz));
    for (size_t i = 0; i < (size_t)P * m.iters; ++i) {
      const uint64_t idx = g0 * m.iters + i;
      const uint32_t first = Hash32(idx * 0x9E3779B1ull + c.seed) % m.blocks;
      const uint32_t step = 1 + Hash32(idx * 0x85EBCA6Bull + c.seed + 1) % div;
      for (uint32_t k = 0; k < c.ransac.subset_sz; ++k)
        h[i * c.ransac.subset_sz + k] = (uint32_t)(((uint64_t)first + (uint64_t)step * k) % m.blocks);
    }
    m.samples.Alloc(n);
    if (n) Hip(hipMemcpy(m.samples.p, h.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy");
  }
  Hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
}

ClipEncoder::~ClipEncoder() = default;
const Shard& ClipEncoder::shard() const { return p_->sh; }
uint32_t ClipEncoder::padded_width() const { return p_->pw; }
uint32_t ClipEncoder::padded_height() const { return p_->ph; }
uint32_t ClipEncoder::blocks() const { return p_->blocks; }
uint64_t ClipEncoder::pyramid_stride() const { return p_->pyr_stride; }
uint32_t ClipEncoder::steps_submitted() const { return p_->n_steps; }
uint32_t ClipEncoder::chunks_per_step() const { return p_->nch; }
uint32_t ClipEncoder::output_sets() const { return (uint32_t)(p_->one_bgr_pass ? p_->rec_sets : p_->coeff_sets); }

void ClipEncoder::ResetPolicy() {
  Sync();
  p_->ResetPolicy();
}

void ClipEncoder::PolicyInfo(uint64_t* chunks_decided, uint64_t* chunks_speculated, double* foreground_share) {
  Sync();
  p_->PollForeground();
  if (chunks_decided) *chunks_decided = p_->n_decided;
  if (chunks_speculated) *chunks_speculated = p_->n_spec;
  if (foreground_share) *foreground_share = p_->fg_share;
}

void ClipEncoder::LoadFrames(const uint8_t* src, uint32_t first_local, uint32_t n, bool src_on_device) {
  Impl& m = *p_;
  if (!src || (uint64_t)first_local + n > m.sh.frames) throw std::runtime_error("svc::ClipEncoder: LoadFrames out of range");
  Sync();
  // other frames: what the speculation policy knew about the clip's foreground share is void (every measurement has landed: Sync above) --
  // unless the caller says the clips it loads are consecutive pieces of one stream, whose last measurement is a fair prior for the next
  // piece (speculation is correct at ANY share: a stale prior costs one slow step, never a wrong byte)
  if (!m.c.keep_foreground_prior) m.ResetPolicy();
  Hip(hipMemcpy(m.bgr.p + (size_t)first_local * m.frame_bytes, src, (size_t)n * m.frame_bytes,
                src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice), "hipMemcpy");
  // a device-to-device hipMemcpy returns before the copy has run, and the streams of Step() do not order behind the
  // null stream: without this the first luma kernel can read frames that have not landed (and `src` could be freed)
  Hip(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
}

void ClipEncoder::SetComm(void* nccl_comm) { p_->comm = nccl_comm; }
void ClipEncoder::SetHaloTransport(HaloFn fn) { p_->halo = std::move(fn); }

void ClipEncoder::Step(bool timed) { StepOn(p_->bgr.p, timed); }

uint32_t ClipEncoder::StepFrames(const uint8_t* device_frames, bool timed) {
  if (!device_frames) throw std::runtime_error("svc::ClipEncoder: StepFrames needs the shard's frames in device memory");
  // refused HERE, before anything of the step is in the pipeline (the kernels' own checks would refuse it stage by stage)
  if (reinterpret_cast<uintptr_t>(device_frames) % 16 != 0) throw std::runtime_error("svc::ClipEncoder: StepFrames: the frames must be 16-byte aligned");
  StepOn(device_frames, timed);
  return p_->n_steps - 1;
}

void ClipEncoder::WaitStep(uint32_t step) {
  Impl& m = *p_;
  if (step >= m.n_steps) throw std::runtime_error("svc::ClipEncoder: WaitStep of a step that has not been submitted");
  // older than the events kept: the oldest step still tracked stands for it (stream order; its last transform is long in the stream,
  // at most depth + 2 steps are in flight)
  if (m.n_steps - step > (uint32_t)Impl::kRing) step = m.n_steps - (uint32_t)Impl::kRing;
  // the step's last transform must be in the stream before its event means anything
  if (m.step_last_micro[step % Impl::kRing] >= m.n_dct) Flush();
  Hip(hipEventSynchronize(m.e_step[step % Impl::kRing]), "hipEventSynchronize");
}

void ClipEncoder::StepOn(const uint8_t* frames, bool timed) {
  Impl& m = *p_;
  m.last_timed = timed;
  m.step_frames = frames;
  if (m.c.schedule != Schedule::kPipelined) { m.SerialStep(timed); return; }
  // The step's chunk plan.  Configured: chunk_pairs (default: one chunk).  The idle-pipeline rule (round 6): a step that finds the pipeline
  // EMPTY -- the first one after LoadFrames / Sync: a clip encoded once is exactly that -- has no earlier step's kernels to overlap its
  // RANSAC + segmentation with, so on a big shard in the two-pass order it is cut in two and overlaps them with its own second half
  // (profiles/r06_ab_idle_rule.txt, same box: C3 2.58-2.63 -> 2.49-2.52 ms, C3b 2.96-3.12 -> 2.89-2.92, C5 2.80-2.85 -> 2.60-2.70); back-to-back
  // steps keep whole-shard launches (the steady state loses 0-4 % to chunks).  Not in the one-pass orders: there the transform runs at the
  // front and the step's tail cannot be hidden either way.  ("Empty" = every stage of every earlier step has been enqueued -- after a Flush()
  // without a Sync() the GPU may still be busy with them; the rule then costs a few launches and hides nothing: harmless.)
  //
  // The MIXED form (round 6; SVC_CLIP_TUNE_MIXED_STEPS, off by default): such a step that knows NOTHING about the clip runs its first half in
  // the two-pass order and its second half reading its frames ONCE, blind -- luma(c0) | search(c0) | transform+luma(c1) | pyramid(c1) |
  // search(c1) | transform(c0), the redo c1 owes beside the transform of c0.  Measured (profiles/r06_ab_mixed_step.txt, one box, three
  // repetitions): at C3's 0.5 % foreground the clip encoded once gains 1.2-2.2 % (2.53 -> 2.47-2.50 ms); at 13 % (C3b, C5) the redo does NOT
  // hide behind the first half's transform and the step loses 5-8 % (C3b 2.87 -> 3.03, C5 2.63 -> 2.84).  Deciding blind is a bet, not a
  // free lunch: the default stays two passes for a clip nothing is known about.
  const uint32_t P = m.sh.pairs;
  uint32_t n = m.nch, cp = m.cp;
  uint8_t order[2] = {0, 0};
  const bool idle = !m.c.chunk_pairs && m.c.world == 1 && !m.c.whole_shard_steps && m.n_dct == m.n_luma && P >= 2 &&
                    (m.c.idle_rule_any_size || (uint64_t)P * m.pw * m.ph >= Impl::kIdleRuleMinPixels);
  if (idle && !m.WouldReadOnce()) {  // (polls the newest foreground measurement)
    n = 2; cp = (P + 1) / 2;
    if (m.spec_quant && !m.c.two_bgr_passes && m.c.mixed_steps && m.fg_share < 0.0) { order[0] = 1; order[1] = 2; }  // nothing known: mixed
  }
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t p0 = std::min(P, k * cp);
    m.next_micro = Impl::Micro{m.n_steps, p0, std::min(cp, P - p0), k == 0, n == 2 ? order[k] : (uint8_t)0, frames};
    if (k + 1 == n) m.step_last_micro[m.n_steps % Impl::kRing] = m.n_luma;  // the micro-step this Iterate enters
    m.Iterate(true, timed);
  }
  ++m.n_steps;
}

void ClipEncoder::Flush() {
  Impl& m = *p_;
  while (m.n_dct < m.n_luma) m.Iterate(false, m.last_timed);
}

void ClipEncoder::Sync() {
  Flush();
  for (hipStream_t s : {p_->sC, p_->sL[0], p_->sL[1], p_->sL[2], p_->sM})
    if (s) Hip(hipStreamSynchronize(s), "hipStreamSynchronize");
}

uint64_t ClipEncoder::StagePairs(Stage s) {
  Sync();
  return p_->timed_pairs[(uint32_t)s];
}

void ClipEncoder::StageTime(Stage s, double* total_ms, uint32_t* launches) {
  Sync();
  double t = 0;
  for (auto& pr : p_->timed[(uint32_t)s]) {
    float ms = 0;
    Hip(hipEventElapsedTime(&ms, pr.first, pr.second), "hipEventElapsedTime");
    t += ms;
  }
  if (total_ms) *total_ms = t;
  if (launches) *launches = (uint32_t)p_->timed[(uint32_t)s].size();
}

void ClipEncoder::ResetTimers() {
  Sync();
  for (auto& v : p_->timed) {
    for (auto& pr : v) { p_->event_pool.push_back(pr.first); p_->event_pool.push_back(pr.second); }
    v.clear();
  }
  for (uint64_t& n : p_->timed_pairs) n = 0;
}

void* ClipEncoder::Output(Buffer b, uint64_t* bytes) {
  Impl& m = *p_;
  Sync();
  const int par = m.n_dct ? m.Set(m.n_dct - 1) : 0, ppar = m.n_dct ? m.Par(m.n_dct - 1) : 0;
  void* ptr = nullptr;
  uint64_t n = 0;
  switch (b) {
    case Buffer::kMv: ptr = m.mv[par].p; n = m.mv[par].bytes(); break;
    case Buffer::kMinMad: ptr = m.mad[par].p; n = m.mad[par].bytes(); break;
    case Buffer::kGlobalMotion: ptr = m.gm[par].p; n = m.gm[par].bytes(); break;
    case Buffer::kRmse: ptr = m.rmse[par].p; n = m.rmse[par].bytes(); break;
    case Buffer::kInlierMask: ptr = m.mask[par].p; n = m.mask[par].bytes(); break;
    case Buffer::kInlierCount: ptr = m.count[par].p; n = m.count[par].bytes(); break;
    case Buffer::kBlockTypes: ptr = m.types[par].p; n = m.types[par].bytes(); break;
    case Buffer::kCoeffs: { auto& q = m.n_dct ? m.Coeffs(m.n_dct - 1) : m.coeffs[0]; ptr = q.p; n = q.bytes(); break; }
    case Buffer::kRecords: { auto& r = m.n_dct ? m.Records(m.n_dct - 1) : m.records[0]; ptr = r.p; n = r.bytes(); break; }
    case Buffer::kPyramids: ptr = m.pyr[ppar].p; n = m.pyr[ppar].bytes(); break;
    case Buffer::kBgr: ptr = m.bgr.p; n = m.bgr.bytes(); break;
    default: throw std::runtime_error("svc::ClipEncoder: unknown buffer");
  }
  if (bytes) *bytes = n;
  return ptr;
}

}  // namespace svc

// ---- C handle API (include/svc_clip.h) ---------------------------------------------------------
static_assert(sizeof(svc_clip_config) == 136 && sizeof(svc_clip_info) == 80, "the ctypes binding (clip.py) mirrors this layout");
struct svc_clip {
  std::unique_ptr<svc::ClipEncoder> enc;
  svc::ClipEncoderConfig cfg;
};

namespace {
thread_local std::string g_clip_err;

template <typename F> int Guard(F&& fn) {
  try {
    fn();
    return 0;
  } catch (const std::exception& e) {
    g_clip_err = e.what();
    return 1;
  } catch (...) {
    g_clip_err = "unknown exception";
    return 1;
  }
}
}  // namespace

extern "C" {

const char* svc_clip_last_error(void) { return g_clip_err.c_str(); }

int svc_clip_plan_shard(uint32_t clip_frames, uint32_t world, uint32_t rank, uint32_t* first_frame, uint32_t* frames,
                        uint32_t* pairs, uint32_t* first_encoded) {
  return Guard([&] {
    if (world == 0 || rank >= world) throw std::runtime_error("svc_clip_plan_shard: rank out of range");
    const svc::Shard s = svc::PlanShard(clip_frames, world, rank);
    if (first_frame) *first_frame = s.first_frame;
    if (frames) *frames = s.frames;
    if (pairs) *pairs = s.pairs;
    if (first_encoded) *first_encoded = s.first_encoded;
  });
}

int svc_clip_create(const svc_clip_config* k, svc_clip** out) {
  return Guard([&] {
    if (!k || !out) throw std::runtime_error("svc_clip_create: null pointer");
    if (k->struct_size != sizeof(svc_clip_config))
      throw std::runtime_error("svc_clip_create: config->struct_size is " + std::to_string(k->struct_size) + ", this build's svc_clip_config has " +
                               std::to_string(sizeof(svc_clip_config)) + " bytes (set struct_size = sizeof(svc_clip_config))");
    constexpr uint32_t kHbmaBits = SVC_HBMA_FORCE_WAVE_PER_BLOCK | SVC_HBMA_FORCE_FUSED | SVC_HBMA_FORCE_TILED | SVC_HBMA_FORCE_LANE;
    constexpr uint32_t kTuneBits = SVC_CLIP_TUNE_STANDALONE_SHAPES | SVC_CLIP_TUNE_SEGMENT_FORK | SVC_CLIP_TUNE_NARROW_ATTEMPTS | SVC_CLIP_TUNE_INLINE_RMSE |
                                   SVC_CLIP_TUNE_TWO_BGR_PASSES | SVC_CLIP_TUNE_ALWAYS_SPECULATE | SVC_CLIP_KEEP_FOREGROUND_PRIOR |
                                   SVC_CLIP_TUNE_WHOLE_SHARD_STEPS | SVC_CLIP_TUNE_SEARCH_AFTER_TRANSFORM | SVC_CLIP_TUNE_IDLE_RULE_ANY_SIZE |
                                   SVC_CLIP_TUNE_MIXED_STEPS | SVC_CLIP_TUNE_RANDOM_POLICY | SVC_CLIP_TUNE_FORK_BEHIND_FRONT;
    if (k->hbma_flags & ~kHbmaBits) throw std::runtime_error("svc_clip_create: unknown hbma_flags bits");
    if (k->tuning & ~kTuneBits) throw std::runtime_error("svc_clip_create: unknown tuning bits");
    if (k->lat_depth > 3) throw std::runtime_error("svc_clip_create: lat_depth must be 0..3");
    if (k->schedule != SVC_CLIP_SERIAL && k->schedule != SVC_CLIP_PIPELINED) throw std::runtime_error("svc_clip_create: unknown schedule");
    svc::ClipEncoderConfig c;
    c.width = k->width; c.height = k->height; c.levels = k->levels; c.mv_block = k->mv_block;
    c.search_range = k->search_range; c.dct_block_w = k->dct_block_w; c.dct_block_h = k->dct_block_h;
    c.fg_step = k->fg_step; c.bg_step = k->bg_step; c.wire = k->wire != 0; c.segmentation = k->segmentation != 0;
    c.seed = k->seed; c.ransac = k->ransac; c.segment = k->segment; c.clip_frames = k->clip_frames;
    c.rank = k->rank; c.world = k->world;
    c.schedule = k->schedule == SVC_CLIP_SERIAL ? svc::Schedule::kSerial : svc::Schedule::kPipelined;
    c.hbma_flags = k->hbma_flags; c.lat_depth = k->lat_depth;
    c.standalone_shapes = (k->tuning & SVC_CLIP_TUNE_STANDALONE_SHAPES) != 0;
    c.segment_fork = (k->tuning & SVC_CLIP_TUNE_SEGMENT_FORK) != 0;
    c.narrow_attempts = (k->tuning & SVC_CLIP_TUNE_NARROW_ATTEMPTS) != 0;
    c.inline_rmse = (k->tuning & SVC_CLIP_TUNE_INLINE_RMSE) != 0;
    c.two_bgr_passes = (k->tuning & SVC_CLIP_TUNE_TWO_BGR_PASSES) != 0;
    c.always_speculate = (k->tuning & SVC_CLIP_TUNE_ALWAYS_SPECULATE) != 0;
    c.keep_foreground_prior = (k->tuning & SVC_CLIP_KEEP_FOREGROUND_PRIOR) != 0;
    c.whole_shard_steps = (k->tuning & SVC_CLIP_TUNE_WHOLE_SHARD_STEPS) != 0;
    c.search_after_transform = (k->tuning & SVC_CLIP_TUNE_SEARCH_AFTER_TRANSFORM) != 0;
    c.idle_rule_any_size = (k->tuning & SVC_CLIP_TUNE_IDLE_RULE_ANY_SIZE) != 0;
    c.mixed_steps = (k->tuning & SVC_CLIP_TUNE_MIXED_STEPS) != 0;
    c.random_policy = (k->tuning & SVC_CLIP_TUNE_RANDOM_POLICY) != 0;
    c.fork_behind_front = (k->tuning & SVC_CLIP_TUNE_FORK_BEHIND_FRONT) != 0;
    c.chunk_pairs = k->chunk_pairs;
    std::unique_ptr<svc_clip> h(new svc_clip);
    h->cfg = c;
    h->enc.reset(new svc::ClipEncoder(c));
    *out = h.release();
  });
}

void svc_clip_destroy(svc_clip* clip) { delete clip; }

int svc_clip_get_info(svc_clip* clip, svc_clip_info* o) {
  return Guard([&] {
    if (!clip || !o) throw std::runtime_error("svc_clip_get_info: null pointer");
    const svc::ClipEncoder& e = *clip->enc;
    const svc::Shard& s = e.shard();
    o->padded_w = e.padded_width(); o->padded_h = e.padded_height();
    o->mv_field_w = o->padded_w / clip->cfg.mv_block; o->mv_field_h = o->padded_h / clip->cfg.mv_block;
    o->blocks = e.blocks(); o->ransac_iters = svc_hip_ransac_iter_count(clip->cfg.ransac);
    o->pyramid_stride = e.pyramid_stride();
    o->frame_bytes = (uint64_t)o->padded_w * o->padded_h * 3;
    o->record_bytes = (clip->cfg.wire && clip->cfg.dct_block_w)
                          ? svc_hip_serialized_frame_bytes(o->padded_w, o->padded_h, clip->cfg.dct_block_w, clip->cfg.dct_block_h) : 0;
    o->first_frame = s.first_frame; o->frames = s.frames; o->pairs = s.pairs; o->first_encoded = s.first_encoded;
    o->needs_halo = s.needs_halo ? 1u : 0u;
    o->chunks_per_step = e.chunks_per_step();
    o->output_sets = e.output_sets();
    o->reserved = 0;
  });
}

int svc_clip_load_frames(svc_clip* clip, const uint8_t* src, uint32_t first_local, uint32_t n, int src_on_device) {
  return Guard([&] { clip->enc->LoadFrames(src, first_local, n, src_on_device != 0); });
}

int svc_clip_set_comm(svc_clip* clip, void* nccl_comm) { return Guard([&] { clip->enc->SetComm(nccl_comm); }); }

int svc_clip_set_halo_callback(svc_clip* clip, svc_clip_halo_fn fn, void* user) {
  return Guard([&] {
    if (!fn) { clip->enc->SetHaloTransport(nullptr); return; }
    clip->enc->SetHaloTransport([fn, user](const uint8_t* s, uint8_t* r, uint64_t bytes, void* stream) {
      if (fn(s, r, bytes, stream, user)) throw std::runtime_error("svc::ClipEncoder: the halo callback failed");
    });
  });
}

int svc_clip_step(svc_clip* clip, int timed) { return Guard([&] { clip->enc->Step(timed != 0); }); }
int svc_clip_step_frames(svc_clip* clip, const uint8_t* device_frames, int timed, uint32_t* step) {
  return Guard([&] {
    const uint32_t s = clip->enc->StepFrames(device_frames, timed != 0);
    if (step) *step = s;
  });
}
int svc_clip_wait_step(svc_clip* clip, uint32_t step) { return Guard([&] { clip->enc->WaitStep(step); }); }
int svc_clip_flush(svc_clip* clip) { return Guard([&] { clip->enc->Flush(); }); }
int svc_clip_sync(svc_clip* clip) { return Guard([&] { clip->enc->Sync(); }); }

int svc_clip_stage_time(svc_clip* clip, uint32_t stage, double* total_ms, uint32_t* launches) {
  return Guard([&] {
    if (stage >= SVC_STAGE_COUNT) throw std::runtime_error("svc_clip_stage_time: unknown stage");
    clip->enc->StageTime((svc::Stage)stage, total_ms, launches);
  });
}

int svc_clip_stage_pairs(svc_clip* clip, uint32_t stage, uint64_t* pairs) {
  return Guard([&] {
    if (stage >= SVC_STAGE_COUNT || !pairs) throw std::runtime_error("svc_clip_stage_pairs: bad argument");
    *pairs = clip->enc->StagePairs((svc::Stage)stage);
  });
}

int svc_clip_reset_timers(svc_clip* clip) { return Guard([&] { clip->enc->ResetTimers(); }); }

int svc_clip_reset_policy(svc_clip* clip) { return Guard([&] { clip->enc->ResetPolicy(); }); }

int svc_clip_policy_info(svc_clip* clip, uint64_t* chunks_decided, uint64_t* chunks_speculated, double* foreground_share) {
  return Guard([&] { clip->enc->PolicyInfo(chunks_decided, chunks_speculated, foreground_share); });
}

int svc_clip_output(svc_clip* clip, uint32_t buffer, void** d_ptr, uint64_t* bytes) {
  return Guard([&] {
    if (buffer >= SVC_BUF_COUNT || !d_ptr) throw std::runtime_error("svc_clip_output: bad argument");
    *d_ptr = clip->enc->Output((svc::Buffer)buffer, bytes);
  });
}

int svc_clip_read(svc_clip* clip, uint32_t buffer, uint64_t offset, void* dst, uint64_t bytes, int dst_on_device) {
  return Guard([&] {
    if (buffer >= SVC_BUF_COUNT || !dst) throw std::runtime_error("svc_clip_read: bad argument");
    uint64_t have = 0;
    const uint8_t* p = static_cast<const uint8_t*>(clip->enc->Output((svc::Buffer)buffer, &have));
    if (offset + bytes > have) throw std::runtime_error("svc_clip_read: range exceeds the buffer");
    svc::Hip(hipMemcpy(dst, p + offset, bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost), "hipMemcpy");
    if (dst_on_device) svc::Hip(hipStreamSynchronize(nullptr), "hipStreamSynchronize");  // the next Step() may overwrite the source
  });
}

}  // extern "C"
