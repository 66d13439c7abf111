/*
 * svc_oracle.h -- CPU restatement of the reference encode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the reported CPU
 * baseline.  The product path (scalable_video_codec_amd/csrc + include/) never
 * links, imports or falls back to it.
 *
 * Parity status
 *   HBMA / EBMA / refinement / RANSAC : PINNED.  Checked bit-for-bit against
 *       the unmodified reference libs/motion.cpp compiled in place into
 *       oracle/_ref/ (tests/test_oracle_vs_ref.py) and against golden vectors
 *       the reference produced (tests/golden/, made by tests/golden/make_golden.py).
 *   quantisation : restated from libs/decoder.cpp:130-144, pinned by the hand
 *       vectors of SURVEY.md 8(c) (the TU needs OpenCV, so it cannot be built).
 *   luma + pyramid : PARITY UNPINNED.  cv::cvtColor(BGR2YUV) + extractChannel +
 *       cv::buildPyramid (libs/encoder.cpp:468-470) live in OpenCV 3.4.x (README:
 *       3.4.16), neither vendored nor installed; restated from its published 8-bit
 *       fixed-point algorithms (svc_oracle_luma / svc_oracle_pyr_down below).
 *   DCT : PARITY UNPINNED.  The arithmetic lives in OpenCV 3.4.x cv::dct
 *       (libs/encoder.cpp:335), which is neither vendored nor installed.  The
 *       oracle of record is the float64 orthonormal DCT-II from its definition.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 */
#ifndef SVC_ORACLE_H
#define SVC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  float x;
  float y;
} svc_oracle_vec2f; /* libs/math.hpp:181-185 (Vec2f) */

typedef struct {
  uint32_t subset_sz;
  float inlier_thresh;
  float success_prob;
  float inlier_ratio;
} svc_oracle_ransac_params; /* libs/motion.hpp:60-79 (RansacParams) */

/* libs/motion.cpp:17-43 (Mad).  a = tracked, b = anchor in every caller. */
float svc_oracle_mad(const uint8_t* a_frame, const uint8_t* b_frame,
                     uint32_t frame_w, uint32_t ax, uint32_t ay, uint32_t bx,
                     uint32_t by, uint32_t block_w, uint32_t block_h);

/* libs/motion.cpp:268-340 (EstimateMotionExhaustiveSearch). */
void svc_oracle_ebma(const uint8_t* tracked, const uint8_t* anchor,
                     uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                     uint32_t block_w, uint32_t block_h, svc_oracle_vec2f* mv,
                     float* min_mad);

/* libs/motion.cpp:342-410 (RefineHierMotionEst). */
void svc_oracle_refine(const uint8_t* tracked, const uint8_t* anchor,
                       uint32_t frame_w, uint32_t frame_h, uint32_t block_w,
                       uint32_t block_h, uint32_t search_range,
                       svc_oracle_vec2f* mv, float* min_mad);

/* libs/motion.cpp:412-465 (EstimateMotionHierarchical).  Returns 0, or 1 when
 * a precondition the reference only asserts (:422-433) is violated. */
int svc_oracle_hbma(const uint8_t* const* tracked_pyr,
                    const uint8_t* const* anchor_pyr, uint32_t level_count,
                    uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                    uint32_t block_w, uint32_t block_h, svc_oracle_vec2f* mv,
                    float* min_mad);

/* libs/motion.cpp:691-749 (EstimateMotionHierarchical16x16Sse2): L = 4,
 * 16x16, psadbw at the two finest levels (:472-550).  CPU-baseline "port". */
int svc_oracle_hbma16_sse2(const uint8_t* const* tracked_pyr,
                           const uint8_t* const* anchor_pyr, uint32_t frame_w,
                           uint32_t frame_h, uint32_t search_range,
                           svc_oracle_vec2f* mv, float* min_mad);

/* libs/motion.cpp:144-149 (IterCount). */
uint32_t svc_oracle_ransac_iter_count(svc_oracle_ransac_params p);

/*
 * libs/motion.cpp:182-266 (EstimateGlobalMotionRansac) with the random draws
 * made explicit: `samples` holds iter_count * subset_sz indices, iteration-major,
 * i.e. exactly the accepted values of `subset[i] = distrib(reng)` (:215).  The
 * reference draws from [0, N] inclusive (:208, off by one); an index equal to N
 * is honoured here as a plain array index, so the caller owns motion_field[N]
 * in that case.  `global_motion` is in/out: its incoming value is what the
 * reference reads uninitialised at :241-242 when no iteration found
 * subset_sz inliers.  `inliers` must hold N entries.
 */
void svc_oracle_ransac(const svc_oracle_vec2f* motion_field, uint32_t n,
                       svc_oracle_ransac_params params, const uint32_t* samples,
                       uint32_t iter_count, float* rmse,
                       svc_oracle_vec2f* global_motion, uint32_t* inliers,
                       uint32_t* inlier_count);

/* libs/motion.cpp:45-53 (EstimateGlobalMotionAvg): the f32 running mean. */
void svc_oracle_global_avg(const svc_oracle_vec2f* motion_field, uint32_t n,
                           svc_oracle_vec2f* avg);

/*
 * libs/motion.cpp:55-99 (EstimateGlobalMotionExhaustiveSearch).  reference_loop != 0 keeps
 * the reference's loop conditions `int dy <= unsigned search_range` (:72, :81) literally: the
 * signed operand converts to unsigned, so for search_range > 0 no candidate is ever visited
 * and the outputs stay {0, 0}, FLT_MAX -- that form is pinned against the compiled reference.
 * reference_loop == 0 runs dy, dx over [-R, R] as evidently meant; that form is the checker
 * of the product's EstimateGlobalMotionExhaustiveSearch (a documented deviation).
 */
void svc_oracle_global_ebma(const uint8_t* tracked, const uint8_t* anchor, uint32_t w,
                            uint32_t h, uint32_t search_range, int reference_loop,
                            svc_oracle_vec2f* global_motion, float* min_mad);

/* libs/motion.cpp:101-142 (EstimateGlobalMotionHierarchical) over the search above. */
void svc_oracle_global_hbma(const uint8_t* const* tracked_pyr,
                            const uint8_t* const* anchor_pyr, uint32_t levels, uint32_t w,
                            uint32_t h, uint32_t search_range, int reference_loop,
                            svc_oracle_vec2f* global_motion);

/*
 * libs/encoder.cpp:468-469: cv::cvtColor(frame, yuv, COLOR_BGR2YUV); cv::extractChannel(yuv, y, 0).
 * OpenCV 3.4's 8-bit path is fixed point with 14 fractional bits: Y = (B * 1868 + G * 9617 +
 * R * 4899 + (1 << 13)) >> 14 (coefficients 0.114 / 0.587 / 0.299 scaled by 2^14 and rounded).
 * bgr: h x w x 3 interleaved B,G,R; y: h x w.
 */
void svc_oracle_luma(const uint8_t* bgr, uint32_t w, uint32_t h, uint8_t* y);

/*
 * One level of libs/encoder.cpp:470 cv::buildPyramid = cv::pyrDown with the default border:
 * the separable 5-tap kernel [1 4 6 4 1] / 16 in both directions on integers, taps outside the
 * plane taken by BORDER_REFLECT_101 (index -i -> i, n - 1 + i -> n - 1 - i), every second sample
 * kept, dst = (sum + 128) >> 8.  src: h x w, dst: ((h + 1) / 2) x ((w + 1) / 2).
 */
void svc_oracle_pyr_down(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst);

/*
 * libs/encoder.cpp:468-470 for one frame: luma, then `levels - 1` pyrDowns, the level planes written
 * back to back (level 0 first) -- the packed layout of include/svc_hip.h.  w and h must be divisible
 * by 2^(levels - 1) (the encoder pads to that, libs/encoder.cpp:164-168).
 */
void svc_oracle_luma_pyramid(const uint8_t* bgr, uint32_t w, uint32_t h, uint32_t levels, uint8_t* packed);

/* libs/encoder.cpp:507-513: fg mask = 255 everywhere except RANSAC inliers. */
void svc_oracle_fg_mask(const uint32_t* inliers, uint32_t inlier_count,
                        uint32_t n, uint8_t* mask);

/* libs/encoder.cpp:507-623: RANSAC inliers + motion field -> region id per MV block
 * (0 = background).  In-repo steps restated exactly; the OpenCV steps (morphology,
 * k-means, connected components) follow this repo's deterministic definitions --
 * see oracle/svc_segment.c.  inlier_mask[i] != 0 marks a RANSAC inlier. */
int svc_oracle_segment(const uint8_t* inlier_mask, const svc_oracle_vec2f* mv,
                       uint32_t mfw, uint32_t mfh, uint32_t mv_bw, uint32_t mv_bh,
                       uint32_t morph_w, uint32_t morph_h, uint32_t cluster_count,
                       uint32_t attempts, uint32_t max_iter, float epsilon,
                       uint32_t connectivity, uint64_t seed, uint32_t* block_types);

/* ---- the per-call forms (oracle/svc_imageops.c): one function per OpenCV call of libs/encoder.cpp:447-640, the checkers
 * of include/svc_hip.h's "Image operations".  PARITY UNPINNED like the fused forms above (OpenCV absent). ---- */
void svc_oracle_bgr2yuv(const uint8_t* bgr, uint32_t w, uint32_t h, uint8_t* yuv); /* cv::cvtColor(BGR2YUV), :449 */
/* cv::erode / dilate / morphologyEx OPEN / CLOSE (:524-527); op 0 erode, 1 dilate, 2 open, 3 close */
void svc_oracle_morph_rect(const uint8_t* src, uint32_t w, uint32_t h, uint32_t kw, uint32_t kh, uint32_t op, uint8_t* dst);
/* cv::kmeans (:575-576) on n points of dims (1..4) integral coordinates; 0 = ok */
int svc_oracle_kmeans(const float* features, uint32_t n, uint32_t dims, uint32_t k, uint32_t attempts, uint32_t max_iter,
                      float epsilon, uint64_t seed, int32_t* labels, double* compactness);
/* cv::connectedComponents (:607-610); returns the component count including the background label */
uint32_t svc_oracle_connected_components(const uint8_t* image, uint32_t w, uint32_t h, uint32_t connectivity, int32_t* labels);

/* libs/decoder.cpp:130-144 (quant lines of DecodeBlock), one coefficient run. */
void svc_oracle_quant(float* coeffs, uint64_t n, uint32_t step);

/* libs/decoder.cpp:130-135 step choice + :140-144, applied to a planar frame
 * laid out as libs/encoder.cpp:323-339 leaves it (3 planes, H x W f32); the
 * type of the tile at (x, y) is block_types[(y / mv_bh) * mv_fw + x / mv_bw]
 * (libs/encoder.cpp:243-249). */
void svc_oracle_quant_frame(float* planes, uint32_t w, uint32_t h,
                            uint32_t mv_bw, uint32_t mv_bh,
                            const uint32_t* block_types, uint32_t fg_step,
                            uint32_t bg_step);

/* libs/encoder.cpp:222-269 (SerializeEncodedFrame), argument for argument; `planes` holds
 * `channels` planes of `plane_elems` floats each.  Returns bytes written. */
uint64_t svc_oracle_serialize_frame(const float* planes, uint64_t plane_elems, uint32_t channels,
                                    const uint32_t* block_types, uint32_t frame_w, uint32_t frame_h,
                                    uint32_t transform_block_w, uint32_t transform_block_h,
                                    uint32_t mv_field_w, uint32_t mv_block_w, uint32_t mv_block_h,
                                    uint8_t* out);

/* libs/encoder.cpp:323-339 (Dct) with cv::dct restated as the float64
 * orthonormal DCT-II.  `bgr` is H x W x 3 u8 interleaved (what :638 converts
 * to f32); `planes64` receives 3 planar H x W doubles in B, G, R order. */
void svc_oracle_dct_frame_f64(const uint8_t* bgr, uint32_t w, uint32_t h,
                              uint32_t block_w, uint32_t block_h,
                              double* planes64);

/* Same result rounded once to f32 -- the CPU baseline the bench times. */
void svc_oracle_dct_frame_f32(const uint8_t* bgr, uint32_t w, uint32_t h,
                              uint32_t block_w, uint32_t block_h,
                              float* planes32);

/* The CPU BASELINE's transform (oracle/svc_cpu_dct.c): f32 separable 8x8 / 16x16 DCT-II, AVX2 + FMA where the host has it.
 * Timed by bench.py's cpu_baseline leg, checked against svc_oracle_dct_frame_f64; not the oracle and not cv::dct. */
int svc_cpu_dct_frame_f32(const uint8_t* bgr, uint32_t w, uint32_t h, uint32_t block, float* planes32);
/* svc_oracle_quant_frame's bits at CPU-deployment speed (same file) */
void svc_cpu_quant_frame_f32(float* planes, uint32_t w, uint32_t h, uint32_t mv_bw, uint32_t mv_bh, const uint32_t* block_types,
                             uint32_t fg_step, uint32_t bg_step);
int svc_cpu_dct_isa(void); /* 2 = the avx2 + fma clone runs on this host, 0 = the baseline x86-64 clone */

/* libs/decoder.cpp:128-149 + :183-207, headless: reconstructed B,G,R (H x W x 3 doubles). */
void svc_oracle_decode_frame(const float* planes, uint32_t w, uint32_t h, uint32_t block_w, uint32_t block_h,
                             const uint32_t* block_types, uint32_t mv_bw, uint32_t mv_bh, uint32_t fg_step,
                             uint32_t bg_step, uint32_t gaze_x, uint32_t gaze_y, uint32_t gaze_w,
                             uint32_t gaze_h, double* out64);

/* exact integer SSE of source vs reconstruction rounded to u8, over region_w x region_h */
uint64_t svc_oracle_sse_frame(const uint8_t* src_bgr, const float* rec_bgr, uint32_t w, uint32_t region_w,
                              uint32_t region_h);

#ifdef __cplusplus
}
#endif

#endif /* SVC_ORACLE_H */
