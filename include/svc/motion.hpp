// motion.hpp -- the reference's motion-estimation API (libs/motion.hpp), served by
// the MI355X kernels.
//
// Every declaration below has the signature -- and therefore the Itanium-mangled
// symbol -- of its namesake in reference libs/motion.hpp, so libsvc_motion.so can be
// linked in place of the reference's `motion` static library
// (libs/CMakeLists.txt:3,11) and apps/encoder.cpp drives it unchanged.  Semantics
// are the reference's (tie rules, carried MAD, clamped windows: DESIGN.md), results
// bit-identical; the differences are listed per function.
//
// Errors: the reference's functions return void and assert their preconditions
// (libs/motion.cpp:417-433).  These wrappers print svc_hip_last_error() and abort()
// on any failure -- including "no GPU": there is no CPU fallback here.
#ifndef SVC_MOTION_HPP
#define SVC_MOTION_HPP

#include <vector>

#include "math.hpp"
#include "types.hpp"

// libs/motion.hpp:60-79
#ifndef SCALABLE_VIDEO_CODEC_MOTION_HPP
struct RansacParams {
  uint subset_sz;
  float inlier_thresh;
  float success_prob;
  float inlier_ratio;
};
#endif

// libs/motion.hpp:134-138 / motion.cpp:412-465.  One fused launch for 16x16 blocks
// with 3-4 levels, otherwise one LDS-staged launch per level.
void EstimateMotionHierarchical(const uchar* const* tracked_pyramid,
                                const uchar* const* anchor_pyramid,
                                uint level_count, uint frame_w, uint frame_h,
                                uint search_range, uint block_w, uint block_h,
                                Vec2f* motion_field, float* min_mad);

// libs/motion.hpp:149-153 / motion.cpp:691-749: level_count = 4, 16x16.  Kept under
// its historical name for link compatibility; nothing about it is SSE2 here.
void EstimateMotionHierarchical16x16Sse2(const uchar* const* tracked_pyramid,
                                         const uchar* const* anchor_pyramid,
                                         uint frame_w, uint frame_h,
                                         uint search_range, Vec2f* mv_field,
                                         float* min_mad);

// libs/motion.hpp:106-110 / motion.cpp:268-340.
void EstimateMotionExhaustiveSearch(const uchar* tracked_frame,
                                    const uchar* anchor_frame, uint frame_w,
                                    uint frame_h, uint search_range,
                                    uint block_w, uint block_h,
                                    Vec2f* motion_field, float* min_mad);

// libs/motion.hpp:100-103 / motion.cpp:182-266.  Differences, both documented in
// samples are drawn from [0, N-1] (the reference's [0, N] reads one past
// the field, motion.cpp:208), and the engine is per-thread, seeded from
// std::random_device unless SvcSeedRansac() was called on this thread.
void EstimateGlobalMotionRansac(const Vec2f* motion_field, uint motion_field_sz,
                                RansacParams params, float* rmse,
                                Vec2f* global_motion,
                                std::vector<uint>* inlier_indices);

// libs/motion.hpp:38 / motion.cpp:45-53: the f32 running mean, bit for bit.
Vec2f EstimateGlobalMotionAvg(const Vec2f* motion_field, uint sz);

// libs/motion.hpp:45-49 / motion.cpp:55-99.  DEVIATION: the reference's loops compare
// `int dy <= uint search_range` (motion.cpp:72, :81), so for search_range > 0 they never run and
// it returns {0, 0} and FLT_MAX whatever the frames hold.  This is the search the code evidently
// means: every (dx, dy) in [-R, R]^2, MAD of the overlap, strict `<` in raster order.  With
// search_range == 0 both give the whole-frame MAD at zero displacement.
void EstimateGlobalMotionExhaustiveSearch(const uchar* tracked_frame,
                                          const uchar* anchor_frame, uint frame_w,
                                          uint frame_h, uint search_range,
                                          Vec2f* global_motion, float* min_mad);

// libs/motion.hpp:55-59 / motion.cpp:101-142, on top of the search above (same deviation): top
// level with range / 2^(L-1), then gm = 2 gm + (a +-1 search around zero) per finer level.
void EstimateGlobalMotionHierarchical(const uchar* const* tracked_pyramid,
                                      const uchar* const* anchor_pyramid,
                                      uint num_levels, uint base_frame_w,
                                      uint base_frame_h, uint base_search_range,
                                      Vec2f* global_motion);

// ---- additions (no counterpart in the reference's headers) -------------------

// Makes this thread's RANSAC draws reproducible.
void SvcSeedRansac(uint seed);

// Compatibility switch for the two whole-frame searches above (per thread, default off).  On: they answer what the
// UNMODIFIED reference answers -- EstimateGlobalMotionExhaustiveSearch {0, 0} and FLT_MAX for every search_range > 0
// (its loops never run, libs/motion.cpp:72, :81), EstimateGlobalMotionHierarchical {0, 0} -- so that a caller who swaps
// this library in for the reference's `motion` target can have bit-identical outputs for these two (dead) functions
// too.  Off: the search the code evidently means.
void SvcReferenceLiteralGlobalSearch(bool on);

// The reference keeps these two file-static (Dct, libs/encoder.cpp:323-339; the
// quant lines of DecodeBlock, libs/decoder.cpp:130-144), so there is no signature
// to keep; parameter meaning follows the originals.  `bgr` is the padded frame as
// H x W x 3 interleaved u8 (what encoder.cpp:638 converts to f32); `planes[c]` are
// three caller-allocated H x W f32 planes in B, G, R order.
void Dct(const uchar* bgr, uint frame_w, uint frame_h, uint block_w, uint block_h,
         float* const planes[3]);

void QuantizeDequantize(float* coeffs, unsigned long long count, uint quant_step);

#endif  // SVC_MOTION_HPP
