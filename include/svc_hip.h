/*
 * svc_hip.h -- C ABI of the MI355X (gfx950) encode hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI layer: its hot path is
 * plain C++ free functions (libs/motion.hpp:100-153) plus two file-static helpers
 * (Dct, libs/encoder.cpp:323-339; the quant lines of DecodeBlock,
 * libs/decoder.cpp:130-144).  include/svc/motion.hpp re-declares those C++
 * signatures verbatim and implements them on top of the *_host entry points
 * below; everything that touches the GPU goes through this header and nothing
 * else.  Plain pointers and sizes only -- no C++ or torch types.
 *
 * Conventions
 *  - Every function returns an svc_status (0 = ok).  svc_hip_last_error() gives
 *    the message of the calling thread's last failure.  The reference's functions
 *    return void and only assert their preconditions (libs/motion.cpp:417-433);
 *    here a violated precondition is SVC_ERR_INVALID_ARG, never UB.
 *  - "d_" pointers are device (HBM) pointers; everything else is host memory.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  The
 *    device-pointer entry points only enqueue work: no allocation, no
 *    synchronisation, safe to capture into a hipGraph.
 *  - A packed pyramid is the L level planes of one frame back to back, level 0
 *    (full resolution) first, each plane (W>>l) x (H>>l) u8, row stride = width
 *    (the layout of the cv::Mat1b planes the reference passes,
 *    libs/encoder.cpp:197-219).  Its size is svc_hip_pyramid_bytes().
 *  - A motion field is (W/block_w) x (H/block_h) row-major (libs/motion.cpp:284-306);
 *    MVs are {x, y} f32 pairs (Vec2f, libs/math.hpp:181-185).
 */
#ifndef SVC_HIP_H
#define SVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum svc_status {
  SVC_OK = 0,
  SVC_ERR_INVALID_ARG = 1, /* a precondition the reference asserts is violated */
  SVC_ERR_UNSUPPORTED = 2, /* valid for the reference, outside this build's kernels */
  SVC_ERR_HIP = 3,         /* a HIP runtime call failed (message has the hipError) */
  SVC_ERR_NO_DEVICE = 4
} svc_status;

/* libs/motion.hpp:60-79 (RansacParams), same field order and types. */
typedef struct svc_ransac_params {
  uint32_t subset_sz;
  float inlier_thresh;
  float success_prob;
  float inlier_ratio;
} svc_ransac_params;

/* The segmentation knobs of EncoderConfig / KMeansParams (libs/encoder.hpp:17-36);
 * defaults in apps/encoder.cpp:47-56: 3x3 rect, 10 clusters, 3 attempts, 10 iterations,
 * epsilon 1, 4-connectivity. */
typedef struct svc_segment_params {
  uint32_t morph_rect_w;
  uint32_t morph_rect_h;
  uint32_t cluster_count;
  uint32_t attempt_count;
  uint32_t max_iter_count;
  float epsilon;
  uint32_t connectivity; /* 4 or 8 */
} svc_segment_params;

/* flags for svc_hip_hbma_pairs / svc_hip_hbma_host */
#define SVC_HBMA_AUTO 0u
#define SVC_HBMA_FORCE_WAVE_PER_BLOCK 1u /* per-level LDS-staged kernel (any shape) */
#define SVC_HBMA_FORCE_FUSED 2u          /* fused all-level kernel; UNSUPPORTED if the shape does not fit */
#define SVC_HBMA_FORCE_TILED 4u          /* fused kernel, LDS-tiled form (4 levels, r_top 1); UNSUPPORTED elsewhere */
#define SVC_HBMA_FORCE_LANE 8u           /* fused kernel, lane-per-block form without LDS (every fused shape) */

const char* svc_hip_last_error(void);
int svc_hip_abi_version(void); /* 5 (round 5: additions only -- svc_hip_tune_host_allocator / svc_hip_host_tuning_requested; round 4 added svc_hip_dct_planes_host and the per-call image operations svc_hip_bgr2yuv_host ... svc_hip_dct_tiles_host) */
int svc_hip_device_count(int* count);

/* Host allocator tuning -- OPT-IN, never applied by loading a library or constructing an object.  An application that moves
 * frame-sized blocks at frame rate (the reference's SerializeEncodedFrame grows a 25 MB std::vector per frame,
 * libs/encoder.cpp:241-266; its queues move such vectors between threads, libs/encoder.hpp:57) pays a page fault per 4 KB on
 * every one of them with glibc's defaults (mmap / munmap above 128 KB).  This call changes the CALLING PROCESS's malloc policy:
 *   SVC_HOST_KEEP_LARGE_BLOCKS  mallopt(M_MMAP_THRESHOLD, 1 GiB) + mallopt(M_TRIM_THRESHOLD, 1 GiB): large blocks stay on the
 *                               heap and are reused; the heap is not trimmed again (RSS stays at its high-water mark);
 *   SVC_HOST_ONE_ARENA          mallopt(M_ARENA_MAX, 1): one heap for all threads (a block freed by the writer thread is reused by
 *                               the encoding thread); serialises malloc / free across threads.
 * Nothing in this repository calls it unless the process's environment has SVC_KEEP_LARGE_BLOCKS=1 (then compat/'s loader and
 * class Encoder apply the flags they used to apply unasked in round 4).  Returns SVC_OK, or
 * SVC_INVALID for unknown flag bits. */
#define SVC_HOST_KEEP_LARGE_BLOCKS 1u
#define SVC_HOST_ONE_ARENA 2u
int svc_hip_tune_host_allocator(uint32_t flags);
/* 1 when the environment opts in (SVC_KEEP_LARGE_BLOCKS=1), else 0: what the libraries consult before tuning anything. */
int svc_hip_host_tuning_requested(void);

/* Measurement aid, not part of the hot path: one launch of a plain streaming kernel (dwordx4 per lane,
 * contiguous across the workgroup) that per iteration reads `reads` x 16 B from d_in and writes `writes` x
 * 16 B to d_out, over buffers of `bytes` each.  Mixes: 1:0, 3:0, 0:1, 1:1, 3:1, 1:4.  bench.py times it
 * next to the kernels it reports: what a memory-bound kernel can reach differs from box to box. */
int svc_hip_probe_stream(const void* d_in, void* d_out, uint64_t bytes, uint32_t reads,
                         uint32_t writes, void* stream);

/* Bytes of one packed pyramid. */
uint64_t svc_hip_pyramid_bytes(uint32_t frame_w, uint32_t frame_h, uint32_t level_count);

/* ------------------------------------------------------------------------- *
 * Motion estimation, device-resident and batched over frame pairs.
 * Pair p reads the tracked pyramid at d_tracked + p * pair_stride_bytes and the
 * anchor pyramid at d_anchor + p * pair_stride_bytes.  For a clip stored as
 * consecutive packed pyramids, d_anchor = d_tracked + pair_stride_bytes gives the
 * reference's frame order (tracked = previous source frame, libs/encoder.cpp:661-663).
 * Outputs: d_mv_xy [n_pairs][blocks][2], d_min_mad [n_pairs][blocks], both fully
 * overwritten (libs/motion.cpp:288-291).
 * ------------------------------------------------------------------------- */

/* replaces EstimateMotionHierarchical, libs/motion.hpp:134-138 / motion.cpp:412-465 */
int svc_hip_hbma_pairs(const uint8_t* d_tracked, const uint8_t* d_anchor,
                       uint64_t pair_stride_bytes, uint32_t n_pairs,
                       uint32_t level_count, uint32_t frame_w, uint32_t frame_h,
                       uint32_t search_range, uint32_t block_w, uint32_t block_h,
                       float* d_mv_xy, float* d_min_mad, uint32_t flags,
                       void* stream);

/* Which kernel svc_hip_hbma_pairs launches for these parameters and flags when the pyramids are 16-byte
 * aligned with a pair stride that is a multiple of 16 (any hipMalloc'ed clip of packed pyramids whose frame
 * width is a multiple of 16): "hbma_tiled16_kernel", "hbma_fused_kernel" or "hbma_wave_level_kernel".  NULL
 * with svc_hip_last_error() set where svc_hip_hbma_pairs would return an error.  No GPU work. */
const char* svc_hip_hbma_kernel_name(uint32_t level_count, uint32_t frame_w, uint32_t frame_h,
                                     uint32_t search_range, uint32_t block_w, uint32_t block_h,
                                     uint32_t flags);

/* replaces EstimateMotionExhaustiveSearch, libs/motion.hpp:106-110 / motion.cpp:268-340.
 * Planes are single-level here: pair p's planes sit at base + p * pair_stride_bytes. */
int svc_hip_ebma_pairs(const uint8_t* d_tracked, const uint8_t* d_anchor,
                       uint64_t pair_stride_bytes, uint32_t n_pairs,
                       uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                       uint32_t block_w, uint32_t block_h, float* d_mv_xy,
                       float* d_min_mad, void* stream);

/* ------------------------------------------------------------------------- *
 * Global motion (RANSAC), batched over frames.  Replaces
 * EstimateGlobalMotionRansac, libs/motion.hpp:100-103 / motion.cpp:182-266, with
 * the random draws made explicit: d_samples is [n_frames][iter_count][subset_sz]
 * accepted sample indices (each < blocks; the reference's inclusive upper bound,
 * motion.cpp:208, is an out-of-bounds read and is not reproduced: svc_hip_ransac_host
 * rejects such an index, and the device form, which cannot inspect d_samples, reads
 * entry blocks - 1 for it, so no draw ever leaves the frame's own field).
 * Outputs per frame: d_gm_xy [2] (in/out, see motion.cpp:241-242), d_rmse,
 * d_inlier_mask [blocks] u8 (1 = inlier, i.e. background; the ascending index
 * list of motion.cpp:261-265 is the positions of the 1s), d_inlier_count.
 * ------------------------------------------------------------------------- */
uint32_t svc_hip_ransac_iter_count(svc_ransac_params params); /* motion.cpp:144-149 */

int svc_hip_ransac_frames(const float* d_mv_xy, uint32_t blocks, uint32_t n_frames,
                          svc_ransac_params params, const uint32_t* d_samples,
                          uint32_t iter_count, float* d_gm_xy, float* d_rmse,
                          uint8_t* d_inlier_mask, uint32_t* d_inlier_count,
                          void* stream);

/* The same with launch flags.  SVC_LAUNCH_BESIDE: the caller runs this launch on a second stream BESIDE
 * bandwidth-bound kernels (a software-pipelined encoder): pick workgroup shapes that fit on a CU next to them (256
 * lanes, one wave per SIMD, a few KB of LDS) instead of the shapes that are fastest alone (1024-lane workgroups that
 * need a whole CU's registers and would not start until the other kernel drains).  Results are identical. */
#define SVC_LAUNCH_BESIDE 1u
/* SVC_LAUNCH_NO_FORK: keep every kernel of the call on `stream` (the segmentation otherwise forks its heavy attempts to an
 * internal side stream).  For callers that already run the call on a stream of its own: HIP multiplexes streams onto a
 * few hardware queues, and an internal stream that lands on the queue of the caller's MAIN stream holds that stream's
 * next kernel back for the length of an attempt kernel. */
#define SVC_LAUNCH_NO_FORK 2u
/* Segmentation, fields above 8 192 blocks: a heavy frame's k-means attempts as launch sequences over several workgroups
 * (default: only when frames x attempts does not exceed the number of CUs).  _WIDE forces that form, _NO_WIDE the
 * one-workgroup form; results are identical. */
#define SVC_LAUNCH_WIDE 4u
#define SVC_LAUNCH_NO_WIDE 8u
/* SVC_LAUNCH_DEFER_RMSE (svc_hip_ransac_frames_ex only): leave out the in-order f32 RMSE sum over the inliers -- the
 * kernel's serial tail (one dependent add per MV block: 34 us at 1080p, 134 us at 4K), which nothing downstream of RANSAC
 * needs.  d_gm_xy, d_inlier_mask and d_inlier_count are final when the launch ends; d_rmse is final only for frames whose
 * inlier count is below subset_sz (libs/motion.cpp:240-242).  The caller completes it with svc_hip_ransac_rmse_frames() on
 * any stream ordered behind this launch; the bytes are the undeferred call's. */
#define SVC_LAUNCH_DEFER_RMSE 16u
int svc_hip_ransac_frames_ex(const float* d_mv_xy, uint32_t blocks, uint32_t n_frames,
                             svc_ransac_params params, const uint32_t* d_samples,
                             uint32_t iter_count, float* d_gm_xy, float* d_rmse,
                             uint8_t* d_inlier_mask, uint32_t* d_inlier_count, uint32_t flags,
                             void* stream);

/* The RMSE of libs/motion.cpp:258-259 (Rmse, :165-180) from the outputs of a RANSAC launch: sqrt(mean over the inliers,
 * summed in index order in f32, of |mv - gm|^2) for every frame with at least subset_sz inliers; other frames' d_rmse is
 * left as it is.  Completes svc_hip_ransac_frames_ex(..., SVC_LAUNCH_DEFER_RMSE, ...); idempotent after a full call. */
int svc_hip_ransac_rmse_frames(const float* d_mv_xy, uint32_t blocks, uint32_t n_frames,
                               svc_ransac_params params, const float* d_gm_xy,
                               const uint8_t* d_inlier_mask, const uint32_t* d_inlier_count,
                               float* d_rmse, void* stream);

/* In-repo part of the segmentation glue, libs/encoder.cpp:507-513 + :549-551:
 * foreground = not a RANSAC inlier; d_block_types [n_frames][blocks] gets 0
 * (BLOCK_TYPE_BACKGROUND, libs/codec.hpp:6) for inliers and region id 1 for the
 * foreground (the OpenCV-side clustering into several regions is SURVEY 8f-2). */
int svc_hip_block_types_frames(const uint8_t* d_inlier_mask, uint32_t blocks,
                               uint32_t n_frames, uint32_t* d_block_types,
                               void* stream);

/* The whole segmentation glue of libs/encoder.cpp:507-623: foreground mask, morphological
 * close + open, k-means on (0, mv.x, x_px, y_px), per-cluster connected components, region
 * ids numbered as the reference numbers them (0 = background).  The in-repo steps are the
 * reference's; the OpenCV steps follow this repo's deterministic definitions (DESIGN.md
 * section 4.6; parity with OpenCV 3.4's kmeans RNG cannot be pinned offline).  Frame f uses
 * seed + f.  d_workspace: svc_hip_segment_workspace_bytes() bytes of scratch.
 * cluster_count up to 64 and attempt_count up to 16 run as fused kernels on `stream` (only enqueued).  Beyond that, up to 255 clusters
 * and 64 attempts (what svc_hip_kmeans_host takes), the call composes the per-call entry points frame by frame through host memory --
 * the same definitions and bits, but SYNCHRONOUS: it waits for `stream` and returns with the ids in place (round 6; the reference's
 * Validate admits any positive count, libs/encoder.cpp:39-61).  Larger counts: SVC_ERR_UNSUPPORTED. */
uint64_t svc_hip_segment_workspace_bytes(uint32_t mv_field_w, uint32_t mv_field_h,
                                         uint32_t n_frames, uint32_t attempt_count);

int svc_hip_segment_frames(const uint8_t* d_inlier_mask, const float* d_mv_xy,
                           uint32_t mv_field_w, uint32_t mv_field_h, uint32_t n_frames,
                           uint32_t mv_block_w, uint32_t mv_block_h,
                           svc_segment_params params, uint64_t seed,
                           uint8_t* d_workspace, uint64_t workspace_bytes,
                           uint32_t* d_block_types, void* stream);

/* svc_hip_segment_frames with launch flags (SVC_LAUNCH_BESIDE, see svc_hip_ransac_frames_ex). */
int svc_hip_segment_frames_ex(const uint8_t* d_inlier_mask, const float* d_mv_xy,
                              uint32_t mv_field_w, uint32_t mv_field_h, uint32_t n_frames,
                              uint32_t mv_block_w, uint32_t mv_block_h,
                              svc_segment_params params, uint64_t seed,
                              uint8_t* d_workspace, uint64_t workspace_bytes,
                              uint32_t* d_block_types, uint32_t flags, void* stream);

/* ------------------------------------------------------------------------- *
 * Transform.  d_bgr: n_frames frames of H x W x 3 u8, interleaved B,G,R (the
 * padded frame the reference converts to f32 at libs/encoder.cpp:638), frame f at
 * d_bgr + f * frame_stride_bytes.  d_planes: [n_frames][3][H][W] f32, plane order
 * B,G,R (cv::split, encoder.cpp:328); coefficient (v,u) of the tile at (x,y) is at
 * row y+v, column x+u (in-place cv::dct on the ROI, encoder.cpp:330-337).
 * block_w x block_h: any transform block the reference's Validate admits (libs/encoder.cpp:62-142)
 * and cv::dct implements -- even sides, or a single row / column of even length -- dividing W and
 * H, up to 64 x 64; 8x8 and 16x16 on frames a multiple of 16 wide take the tuned kernels (and ask
 * for 16-byte aligned frames), every other shape a general one.  An odd side is
 * SVC_ERR_INVALID_ARG (cv::dct asserts there), a side above 64 SVC_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------- */

/* replaces static Dct, libs/encoder.cpp:323-339 */
int svc_hip_dct_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes,
                       uint32_t n_frames, uint32_t frame_w, uint32_t frame_h,
                       uint32_t block_w, uint32_t block_h, float* d_planes,
                       void* stream);

/* Dct followed by the decoder's quantise-round-dequantise (libs/decoder.cpp:130-144)
 * in one pass.  d_block_types: [n_frames][mv blocks] region ids, 0 = background
 * (libs/codec.hpp:6); the tile at (x,y) takes the type of MV block
 * (y / mv_block_h) * (W / mv_block_w) + x / mv_block_w (encoder.cpp:243-249). */
int svc_hip_dct_quant_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes,
                             uint32_t n_frames, uint32_t frame_w, uint32_t frame_h,
                             uint32_t block_w, uint32_t block_h,
                             const uint32_t* d_block_types, uint32_t mv_block_w,
                             uint32_t mv_block_h, uint32_t fg_step,
                             uint32_t bg_step, float* d_planes, void* stream);

/* replaces the quant lines of DecodeBlock, libs/decoder.cpp:140-144, in place */
int svc_hip_quant(float* d_coeffs, uint64_t n, uint32_t step, void* stream);

int svc_hip_quant_frames(float* d_planes, uint32_t n_frames, uint32_t frame_w,
                         uint32_t frame_h, uint32_t mv_block_w, uint32_t mv_block_h,
                         const uint32_t* d_block_types, uint32_t fg_step,
                         uint32_t bg_step, void* stream);

/* ------------------------------------------------------------------------- *
 * Wire format.
 * ------------------------------------------------------------------------- */

/* libs/codec.hpp:8-17 (Header), the first 32 bytes of the encoder's output. */
typedef struct svc_wire_header {
  uint32_t frame_count;
  uint32_t frame_w;
  uint32_t frame_h;
  uint32_t frame_excess_w;
  uint32_t frame_excess_h;
  uint32_t transform_block_w;
  uint32_t transform_block_h;
  uint32_t channel_count;
} svc_wire_header;

/* Fills the header as libs/encoder.cpp:360-381 does: frame_count = clip frames - 1 (the first
 * frame is tracked-only), the UNPADDED size, excess = padded - unpadded with the padding rule
 * of libs/encoder.cpp:164-168, 3 channels.  Host-only. */
int svc_hip_wire_header(uint32_t clip_frame_count, uint32_t frame_w, uint32_t frame_h,
                        uint32_t mv_block_w, uint32_t mv_block_h, uint32_t level_count,
                        uint32_t transform_block_w, uint32_t transform_block_h,
                        svc_wire_header* out);

/* Bytes SerializeEncodedFrame emits for one frame with these arguments. */
uint64_t svc_hip_serialized_frame_bytes(uint32_t frame_w, uint32_t frame_h,
                                        uint32_t transform_block_w, uint32_t transform_block_h);

/* replaces SerializeEncodedFrame, libs/encoder.cpp:222-269, argument for argument (3 channels).
 * d_planes: [n_frames][3][plane_elems] f32 (plane_elems = padded W * H, what the Dct entry
 * points write); frame_w / frame_h: the tile-loop bounds AND row stride exactly as the
 * reference uses them (the encoder passes the unpadded size, :647-650; pass the padded size
 * for a stream the reference's decoder can parse; none of the reference's asserts (:230-239, with their swapped w / h) is a
 * precondition -- its Release build compiles them out and configurations its own Validate admits trip them --, only every read must
 * stay inside planes and motion field: non-square tiles wider than tall make the reference walk past the planes' end on the last
 * tile row, which is SVC_ERR_INVALID_ARG here).  Frame f is written at
 * d_out + f * out_stride_bytes (>= svc_hip_serialized_frame_bytes, multiple of 4). */
int svc_hip_serialize_frames(const float* d_planes, uint64_t plane_elems, uint32_t n_frames,
                             const uint32_t* d_block_types, uint32_t frame_w, uint32_t frame_h,
                             uint32_t transform_block_w, uint32_t transform_block_h,
                             uint32_t mv_field_w, uint32_t mv_field_h, uint32_t mv_block_w,
                             uint32_t mv_block_h, uint8_t* d_out, uint64_t out_stride_bytes,
                             void* stream);

/* Dct (+ optional quant) that emits the serialised records DIRECTLY instead of coefficient
 * planes: one kernel = libs/encoder.cpp:638-650 (convertTo, Dct, SerializeEncodedFrame) with
 * no extra pass over HBM.  Same bytes as svc_hip_dct[_quant]_frames followed by
 * svc_hip_serialize_frames(frame_w, emit_frame_h, ...).  Square transform block (even side up to 64; 8 and 16 take the tuned kernel);
 * frame_w must already be the padded width (the fused path does not reproduce the reference's
 * unpadded-width row stride, libs/encoder.cpp:258 -- use svc_hip_serialize_frames for that);
 * emit_frame_h = the height SerializeEncodedFrame is given (the encoder passes the unpadded
 * one; pass frame_h for a stream the reference's decoder parses).  fg_step = bg_step = 0
 * skips the quantiser (the reference's encoder emits raw coefficients). */
int svc_hip_dct_records_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes,
                               uint32_t n_frames, uint32_t frame_w, uint32_t frame_h,
                               uint32_t block, const uint32_t* d_block_types,
                               uint32_t mv_block_w, uint32_t mv_block_h, uint32_t fg_step,
                               uint32_t bg_step, uint32_t emit_frame_h, uint8_t* d_records,
                               uint64_t records_stride_bytes, void* stream);

/* The record emitter with the luma plane as a by-product: ONE pass over the BGR bytes feeds both the transform (records of the RAW
 * coefficients, what the reference's encoder emits: libs/encoder.cpp:638-650) and cv::cvtColor + extractChannel (:468-469) -- Y is
 * pointwise, so the lane that holds 16 pixels of a row for the transform stores their 16 luma bytes into level 0 of the frame's packed
 * pyramid (frame f at d_pyr + f * pyr_stride_bytes; levels 1.. are then svc_hip_pyramid_levels_frames).  The clip is read once per
 * step instead of twice.  Region ids do not exist yet when this runs (they need the pyramid it produces): every record's type word
 * is written as 0 (background, libs/codec.hpp:6) and svc_hip_wire_patch_types_frames stores the foreground ids afterwards -- the two
 * calls together leave exactly the bytes of svc_hip_dct_records_frames(fg_step = bg_step = 0).  block: 8 or 16; frame_w a multiple of
 * 16; UNSUPPORTED otherwise (call svc_hip_luma_pyramid_frames + svc_hip_dct_records_frames). */
int svc_hip_dct_records_luma_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames,
                                    uint32_t frame_w, uint32_t frame_h, uint32_t block, uint32_t emit_frame_h,
                                    uint8_t* d_records, uint64_t records_stride_bytes, uint8_t* d_pyr,
                                    uint64_t pyr_stride_bytes, void* stream);
/* Type words of records that were emitted before the region ids were known (libs/encoder.cpp:243-249: the id of the MV block that
 * holds the tile).  Stores the word of every tile of a FOREGROUND block (id != 0); all_tiles != 0 stores the zeros too (records whose
 * type words hold anything else than 0). */
int svc_hip_wire_patch_types_frames(const uint32_t* d_block_types, uint32_t n_frames, uint32_t frame_w, uint32_t frame_h,
                                    uint32_t emit_frame_h, uint32_t block, uint32_t mv_block_w, uint32_t mv_block_h,
                                    uint8_t* d_records, uint64_t records_stride_bytes, int all_tiles, void* stream);

/* Dct + quant with the BGR clip read ONCE per step -- speculation on the region ids.  The quantiser's step is the tile's region id's
 * (libs/decoder.cpp:130-135), and the id exists only after luma -> pyramid -> motion search -> RANSAC -> segmentation of the same frame;
 * the plain order therefore reads every frame twice (svc_hip_luma_pyramid_frames, later svc_hip_dct_quant_frames).  Instead:
 *   svc_hip_dct_quant_luma_frames   at the FRONT of a step: every tile quantised as background (bg_step) into d_planes, and the luma
 *                                   plane (cv::cvtColor + extractChannel, libs/encoder.cpp:468-469) into level 0 of the frame's packed
 *                                   pyramid, from one pass over the B,G,R bytes (levels 1..: svc_hip_pyramid_levels_frames);
 *   svc_hip_dct_quant_redo_frames   once the ids exist: the tiles of every MV block whose id is not 0 are transformed again and
 *                                   quantised with fg_step.  d_ws: svc_hip_dct_redo_workspace_bytes(n_frames, frame_w, frame_h,
 *                                   mv_block_w, mv_block_h) bytes, 16-byte aligned.
 * The two calls together leave exactly the bytes of svc_hip_dct_quant_frames.  When it pays: by bytes alone (15 per FOREGROUND pixel moved
 * again against 3 per pixel of every frame saved) up to ~17 % foreground MV blocks, but MEASURED on MI355X only below ~2-3 %: the redo moves
 * scattered 16-pixel pieces at about a third of the streaming rate (profiles/r05_ab_speculative_quant.txt: 0.5 % foreground -> the step
 * 5 % faster, 13 % -> 19 % slower).  Decide with svc_hip_count_foreground on a recent batch's region ids (svc::ClipEncoder speculates at
 * <= 2 %).
 * block: 8 or 16; frame_w a multiple of 16; MV blocks whole 16-pixel segments wide and whole transform blocks tall; else UNSUPPORTED. */
int svc_hip_dct_quant_luma_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t frame_w,
                                  uint32_t frame_h, uint32_t block, uint32_t bg_step, float* d_planes, uint8_t* d_pyr,
                                  uint64_t pyr_stride_bytes, void* stream);
/* *d_count = how many of the n region ids are not 0 (foreground MV blocks): the feedback a driver decides on whether the next step
 * speculates (svc::ClipEncoder does: speculation pays while the share of foreground blocks is a few per cent). */
int svc_hip_count_foreground(const uint32_t* d_block_types, uint64_t n, uint32_t* d_count, void* stream);
uint64_t svc_hip_dct_redo_workspace_bytes(uint32_t n_frames, uint32_t frame_w, uint32_t frame_h, uint32_t mv_block_w,
                                          uint32_t mv_block_h);
int svc_hip_dct_quant_redo_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t frame_w,
                                  uint32_t frame_h, uint32_t block, const uint32_t* d_block_types, uint32_t mv_block_w,
                                  uint32_t mv_block_h, uint32_t fg_step, float* d_planes, uint8_t* d_ws, uint64_t ws_bytes,
                                  void* stream);

/* ------------------------------------------------------------------------- *
 * Decoder-side inverse path, headless: DecodeBlock over every tile
 * (libs/decoder.cpp:128-149, :183-207) without the GUI.  d_planes: coefficient planes as the
 * Dct entry points write them (raw, or already quantised: quantisation is idempotent);
 * step = gazed ? 1 : (type == 0 ? bg_step : fg_step), gazed = the gaze rectangle (in padded
 * frame coordinates; gaze_w or gaze_h == 0 = none) contains the tile origin.  d_bgr_f32:
 * [n_frames][H][W][3] reconstructed B,G,R (the decoder's upscaled_frame before its / 255).
 * ------------------------------------------------------------------------- */
int svc_hip_decode_frames(const float* d_planes, uint32_t n_frames, uint32_t frame_w,
                          uint32_t frame_h, uint32_t block, const uint32_t* d_block_types,
                          uint32_t mv_block_w, uint32_t mv_block_h, uint32_t fg_step,
                          uint32_t bg_step, uint32_t gaze_x, uint32_t gaze_y, uint32_t gaze_w,
                          uint32_t gaze_h, float* d_bgr_f32, void* stream);

/* Exact integer sum of squared errors per frame between the source frames (u8 B,G,R, as given
 * to the Dct entry points) and a reconstruction rounded to u8 (clamp(round)), over the top-left
 * region_w x region_h pixels (the unpadded picture).  PSNR = 10 log10(255^2 * 3 * region_w *
 * region_h / sse).  d_sse: [n_frames] u64, overwritten. */
int svc_hip_sse_frames(const uint8_t* d_src_bgr, uint64_t src_frame_stride_bytes,
                       const float* d_rec_bgr_f32, uint32_t n_frames, uint32_t frame_w,
                       uint32_t frame_h, uint32_t region_w, uint32_t region_h, uint64_t* d_sse,
                       void* stream);

/* ------------------------------------------------------------------------- *
 * Pre-step: luma + pyramid on the device, so the pyramid never
 * crosses PCIe.  Stands in for cv::cvtColor(BGR2YUV) + cv::extractChannel +
 * cv::buildPyramid (libs/encoder.cpp:468-470) with this repo's fixed-point
 * definitions (parity with OpenCV unpinned offline).
 * d_pyr: [n_frames] packed pyramids, frame f at d_pyr + f * pyr_stride_bytes.
 * ------------------------------------------------------------------------- */
int svc_hip_luma_pyramid_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes,
                                uint32_t n_frames, uint32_t frame_w,
                                uint32_t frame_h, uint32_t level_count,
                                uint8_t* d_pyr, uint64_t pyr_stride_bytes,
                                void* stream);
/* cv::buildPyramid (libs/encoder.cpp:470) from level-0 planes that already exist: levels 1 .. level_count - 1 of n_frames packed
 * pyramids (same layout, same kernels as the levels svc_hip_luma_pyramid_frames produces past its first). */
int svc_hip_pyramid_levels_frames(uint8_t* d_pyr, uint64_t pyr_stride_bytes, uint32_t n_frames, uint32_t frame_w,
                                  uint32_t frame_h, uint32_t level_count, void* stream);

/* ------------------------------------------------------------------------- *
 * Whole-frame global motion: the three estimators of libs/motion.hpp:38-59.  No caller in the
 * reference (the encoder uses RANSAC); kept so that the replacement library is complete.
 *  - exhaustive: MAD (libs/motion.cpp:17-43, 32-bit sums) of the overlap of the two frames
 *    shifted by (dx, dy) for dy, dx in [-R, R], raster order, strict `<` (first minimum);
 *    outputs {dx, dy} and the minimum MAD.  DEVIATION: the reference's loops compare an int with
 *    an unsigned (motion.cpp:72, :81) and never run for R > 0 -- it always returns {0, 0},
 *    FLT_MAX; this is the function as evidently meant.  R must be smaller than both frame sides.
 *  - hierarchical (motion.cpp:101-142): exhaustive search on the top level with R / 2^(L-1), then
 *    per finer level gm = 2 gm + (a +-1 exhaustive search around zero), as written there.
 *  - average (motion.cpp:45-53): the f32 running mean avg += (mv[i] - avg) * (1 / (i + 1)).
 * d_workspace: svc_hip_global_ebma_workspace_bytes() bytes, 8-byte aligned.
 * ------------------------------------------------------------------------- */
uint64_t svc_hip_global_ebma_workspace_bytes(uint32_t search_range, uint32_t n_pairs);

/* replaces EstimateGlobalMotionExhaustiveSearch, libs/motion.hpp:45-49, batched over pairs */
int svc_hip_global_ebma_pairs(const uint8_t* d_tracked, const uint8_t* d_anchor,
                              uint64_t pair_stride_bytes, uint32_t n_pairs, uint32_t frame_w,
                              uint32_t frame_h, uint32_t search_range, uint8_t* d_workspace,
                              uint64_t workspace_bytes, float* d_gm_xy, float* d_min_mad,
                              void* stream);

/* replaces EstimateGlobalMotionAvg, libs/motion.hpp:38, batched: d_avg_xy [n_frames][2] */
int svc_hip_global_avg_frames(const float* d_mv_xy, uint32_t blocks, uint32_t n_frames,
                              float* d_avg_xy, void* stream);

/* ------------------------------------------------------------------------- *
 * Multi-GPU: frames shard as consecutive chunks, one per rank, and the only
 * cross-rank dependency is the reference's only cross-frame state -- the previous SOURCE
 * frame's Y pyramid (libs/encoder.cpp:661-663).  Rank r therefore sends the packed pyramid of
 * its last frame to rank r + 1 and receives its predecessor's: one RCCL send/recv pair per
 * rank and step over one xGMI link per direction, no other collective anywhere on the path.
 * RCCL is bound at run time (librccl.so.1, the copy already in the process if there is one);
 * without it these return SVC_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------- */
#define SVC_COMM_ID_BYTES 128u /* sizeof(ncclUniqueId) */
#define SVC_SHIFT_CYCLIC 1u    /* the last rank also sends to rank 0 (frame-per-GPU round robin) */

/* SVC_OK when librccl could be bound in this process (no communicator is created: safe to call on any subset of
 * the ranks, unlike svc_hip_comm_create, which is a collective). */
int svc_hip_comm_available(void);
/* What the communicator says about itself: ncclCommCount / ncclCommUserRank / ncclCommCuDevice (any pointer may be
 * null).  A multi-rank run reports these so that "the halo really went through an N-rank RCCL communicator" is a
 * measured statement. */
int svc_hip_comm_info(void* comm, uint32_t* ranks, uint32_t* rank, int32_t* device);
/* ncclGetUniqueId: call on one rank, hand the bytes to all of them out of band. */
int svc_hip_comm_unique_id(uint8_t id[SVC_COMM_ID_BYTES]);
/* ncclCommInitRank on the calling thread's current device; *comm is an ncclComm_t. */
int svc_hip_comm_create(const uint8_t id[SVC_COMM_ID_BYTES], uint32_t rank, uint32_t world,
                        void** comm);
int svc_hip_comm_destroy(void* comm);

/* The halo shift: enqueues on `stream`, as ONE RCCL group, the send of `bytes` from d_send to
 * rank + 1 (if there is one) and the receive of `bytes` into d_recv from rank - 1 (if there is
 * one).  comm: an ncclComm_t of `world` ranks in which the caller is `rank`.  Only enqueues. */
int svc_hip_halo_shift(void* comm, const uint8_t* d_send, uint8_t* d_recv, uint64_t bytes,
                       uint32_t rank, uint32_t world, uint32_t flags, void* stream);

/* ------------------------------------------------------------------------- *
 * Host-pointer forms: what the C++ wrappers of include/svc/motion.hpp call.  They
 * stage through pinned buffers owned by the library, run the device entry point
 * on an internal stream and synchronise before returning (the reference's calls
 * are synchronous, libs/encoder.cpp:472-498).  Thread-safe; per-thread staging.
 * ------------------------------------------------------------------------- */
int svc_hip_hbma_host(const uint8_t* const* tracked_pyr,
                      const uint8_t* const* anchor_pyr, uint32_t level_count,
                      uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                      uint32_t block_w, uint32_t block_h, float* mv_xy,
                      float* min_mad, uint32_t flags);

int svc_hip_ebma_host(const uint8_t* tracked, const uint8_t* anchor,
                      uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                      uint32_t block_w, uint32_t block_h, float* mv_xy,
                      float* min_mad);

int svc_hip_ransac_host(const float* mv_xy, uint32_t blocks,
                        svc_ransac_params params, const uint32_t* samples,
                        uint32_t iter_count, float* gm_xy, float* rmse,
                        uint32_t* inlier_indices, uint32_t* inlier_count);

int svc_hip_dct_host(const uint8_t* bgr, uint32_t frame_w, uint32_t frame_h,
                     uint32_t block_w, uint32_t block_h, float* planes);

/* The same with D2H straight into the caller's three plane buffers (B, G, R order; any may alias none): what the C++
 * Dct() wrapper of include/svc/motion.hpp and the OpenCV-shaped adapter (compat/opencv2/) call -- no packed intermediate. */
int svc_hip_dct_planes_host(const uint8_t* bgr, uint32_t frame_w, uint32_t frame_h, uint32_t block_w,
                            uint32_t block_h, float* const planes[3]);

/* ------------------------------------------------------------------------- *
 * Image operations, host-pointer forms: ONE ENTRY POINT PER OpenCV CALL of the reference's per-frame loop
 * (libs/encoder.cpp:447-640) that does arithmetic -- for a host whose control flow stays the reference's own
 * Encoder::operator() (the adapter under compat/opencv2/ makes cv::cvtColor, cv::buildPyramid, cv::morphologyEx,
 * cv::kmeans, cv::connectedComponents and cv::dct thin callers of these).  All images are tightly packed
 * (row stride = width x channels); all calls are synchronous and thread-safe like the other *_host forms.  The OpenCV
 * semantics followed are the ones stated in oracle/svc_imageops.c / oracle/svc_segment.c (parity with OpenCV itself is
 * unpinned offline); the fused, batched device forms above (svc_hip_luma_pyramid_frames, svc_hip_segment_frames,
 * svc_hip_dct_frames) compute the same values and are what a throughput-minded caller uses.
 * ------------------------------------------------------------------------- */

/* cv::cvtColor(src, dst, COLOR_BGR2YUV), 8-bit (libs/encoder.cpp:449, :468): 14-bit fixed point,
 * Y = (1868 B + 9617 G + 4899 R + 8192) >> 14, U = ((B - Y) * 8061 + (128 << 14) + 8192) >> 14,
 * V = ((R - Y) * 14369 + (128 << 14) + 8192) >> 14, saturated to 0..255; dst interleaved Y,U,V. */
int svc_hip_bgr2yuv_host(const uint8_t* bgr, uint32_t w, uint32_t h, uint8_t* yuv);

/* cv::buildPyramid(level0, levels, level_count - 1) (libs/encoder.cpp:451, :470) on one 8-bit plane: out_levels[l] for
 * l = 1 .. level_count - 1 receives (w >> l) x (h >> l); out_levels[0] is ignored (OpenCV's level 0 IS the source).
 * w and h must be divisible by 2^(level_count - 1) (what the encoder pads to, libs/encoder.cpp:164-168). */
int svc_hip_build_pyramid_host(const uint8_t* level0, uint32_t w, uint32_t h, uint32_t level_count,
                               uint8_t* const* out_levels);

/* cv::erode / cv::dilate / cv::morphologyEx(MORPH_OPEN | MORPH_CLOSE) with a rectangular kernel_w x kernel_h element
 * anchored at its centre (kernel_w / 2, kernel_h / 2), one iteration, on an 8-bit single-channel image
 * (libs/encoder.cpp:524-527 on the MV-field mask).  Pixels outside the image are ignored
 * (cv::morphologyDefaultBorderValue()).  src == dst is allowed. */
#define SVC_MORPH_ERODE 0u
#define SVC_MORPH_DILATE 1u
#define SVC_MORPH_OPEN 2u
#define SVC_MORPH_CLOSE 3u
int svc_hip_morph_rect_host(const uint8_t* src, uint32_t w, uint32_t h, uint32_t kernel_w, uint32_t kernel_h,
                            uint32_t op, uint8_t* dst);

/* cv::kmeans(data, K, labels, TermCriteria(COUNT | EPS, max_iter, epsilon), attempts, KMEANS_PP_CENTERS)
 * (libs/encoder.cpp:575-576) by this repo's deterministic definition (oracle/svc_segment.c): `features` is n points of
 * `dims` (1..4) f32 coordinates, which must be integers of magnitude below 32768 (block-matching output and pixel
 * positions are); k-means++ seeding with exact integer weights and a counter hash of `seed` in place of cv::theRNG(),
 * Lloyd iterations with integer sums / counts and f64 distances in coordinate order, the attempt with the smallest
 * fixed-point compactness wins (ties: the earlier one).  labels: n cluster ids in [0, k); *compactness (may be NULL):
 * sum over the points of the squared distance to their centre, as cv::kmeans returns it (here in 1/256 steps).
 * n >= k >= 1, k <= 255, attempts <= 64. */
int svc_hip_kmeans_host(const float* features, uint32_t n, uint32_t dims, uint32_t k, uint32_t attempts,
                        uint32_t max_iter, float epsilon, uint64_t seed, int32_t* labels, double* compactness);

/* cv::connectedComponents(image, labels, connectivity, CV_32S) (libs/encoder.cpp:607-610): non-zero pixels of the
 * 8-bit image are foreground; labels[y][x] = 0 for background, 1 .. n for the components, numbered in raster order of
 * each component's first pixel; *count = n + 1 (OpenCV's return value counts the background label). */
int svc_hip_connected_components_host(const uint8_t* image, uint32_t w, uint32_t h, uint32_t connectivity,
                                      int32_t* labels, uint32_t* count);

/* cv::dct(tile, tile) (flags 0: forward orthonormal DCT-II, libs/encoder.cpp:335) over MANY tiles of one f32 image
 * in one launch, in place: tiles_xy holds n_tiles top-left corners {x, y} of block_w x block_h tiles inside the
 * w x h image (they must not overlap); tiles_xy == NULL means every tile of the regular (w / block_w) x (h / block_h)
 * grid.  Sides as svc_hip_dct_frames takes them (even, or a single row / column of even length, up to 64).  f64
 * accumulation, rounded once to f32. */
int svc_hip_dct_tiles_host(float* image, uint32_t w, uint32_t h, uint32_t block_w, uint32_t block_h,
                           const uint32_t* tiles_xy, uint32_t n_tiles);

int svc_hip_dct_quant_host(const uint8_t* bgr, uint32_t frame_w, uint32_t frame_h,
                           uint32_t block_w, uint32_t block_h,
                           const uint32_t* block_types, uint32_t mv_block_w,
                           uint32_t mv_block_h, uint32_t fg_step, uint32_t bg_step,
                           float* planes);

int svc_hip_quant_host(float* coeffs, uint64_t n, uint32_t step);

int svc_hip_global_ebma_host(const uint8_t* tracked, const uint8_t* anchor, uint32_t frame_w,
                             uint32_t frame_h, uint32_t search_range, float* gm_xy,
                             float* min_mad);

/* replaces EstimateGlobalMotionHierarchical, libs/motion.hpp:55-59 */
int svc_hip_global_hbma_host(const uint8_t* const* tracked_pyr,
                             const uint8_t* const* anchor_pyr, uint32_t level_count,
                             uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                             float* gm_xy);

int svc_hip_global_avg_host(const float* mv_xy, uint32_t blocks, float* avg_xy);

#ifdef __cplusplus
}
#endif

#endif /* SVC_HIP_H */
