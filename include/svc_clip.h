/*
 * svc_clip.h -- C handle API over svc::ClipEncoder (include/svc/clip_encoder.hpp): the driver of
 * one rank's shard of an HBM-resident clip.  This is what bench.py and the Python harness bind
 * with ctypes; a C++ host uses the class directly.  Built into libsvc_motion.so.
 *
 * The stages a step runs are the reference's per-frame loop, libs/encoder.cpp:453-664, batched
 * over the shard: luma + pyramid (:468-470), EstimateMotionHierarchical (:472-482),
 * EstimateGlobalMotionRansac (:491-498), segmentation glue -> region ids (:507-623), Dct (:638-640)
 * + the decoder's quant lines (libs/decoder.cpp:130-144) -- each a call into include/svc_hip.h.
 *
 * Every function returns 0 on success; svc_clip_last_error() has the calling thread's last message.
 */
#ifndef SVC_CLIP_H
#define SVC_CLIP_H

#include <stdint.h>

#include "svc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svc_clip svc_clip;

#define SVC_CLIP_SERIAL 0u    /* a step's stages back to back on one stream */
#define SVC_CLIP_PIPELINED 1u /* software pipeline over consecutive steps, see clip_encoder.hpp */

#define SVC_CLIP_TUNE_STANDALONE_SHAPES 1u /* pipelined: RANSAC / segmentation keep their stand-alone launch shapes */
#define SVC_CLIP_TUNE_SEGMENT_FORK 2u      /* pipelined: the segmentation may fork its heavy attempts to a side stream */
#define SVC_CLIP_TUNE_INLINE_RMSE 8u       /* pipelined: RANSAC keeps its in-order RMSE sum inside its kernel */
#define SVC_CLIP_TUNE_NARROW_ATTEMPTS 4u   /* segmentation: never spread a heavy frame's k-means attempts over workgroups */
#define SVC_CLIP_TUNE_TWO_BGR_PASSES 16u   /* never read the BGR clip once per step: luma + pyramid, later the transform (the plain order).  Default:
                                              wire output always reads it once (records + luma plane from one kernel, type words stored afterwards);
                                              planes + quant speculates -- every tile quantised as background at the front of the step, the foreground
                                              tiles redone -- while the foreground share of recent steps is small (clip_encoder.hpp) */
#define SVC_CLIP_TUNE_ALWAYS_SPECULATE 32u /* planes + quant: speculate on every step, whatever the foreground share (tests, A/B) */
/* Not a tuning switch but a statement about the input (same field): the clips handed to svc_clip_load_frames are consecutive pieces of ONE
 * stream, so the foreground share measured on the last piece stays the speculation policy's prior for the next (default: a load voids it and
 * the first step over new frames runs the plain two-pass order).  Safe at any share: a stale prior costs one slow step, never a byte. */
#define SVC_CLIP_KEEP_FOREGROUND_PRIOR 64u
#define SVC_CLIP_TUNE_IDLE_RULE_ANY_SIZE 512u /* the idle-pipeline rule whatever the shard's size (tests; default: from 400 M pixels x frames) */
#define SVC_CLIP_TUNE_FORK_BEHIND_FRONT 4096u /* one rank, A/B: RANSAC + segmentation of the previous micro-step fork behind the front-of-step transform */
#define SVC_CLIP_TUNE_RANDOM_POLICY 2048u /* tests: the speculation policy answers yes / no by a fixed pseudo-random sequence over the chunk launches */
#define SVC_CLIP_TUNE_MIXED_STEPS 1024u /* a step into an empty pipeline that knows nothing about the clip takes the mixed form (two-pass half +
                                           blind one-pass half): A/B, off by default (- 2 % at 0.5 % foreground, + 5-8 % at 13 %) */
#define SVC_CLIP_TUNE_SEARCH_AFTER_TRANSFORM 256u /* one rank: the motion search, not a pyramid pass, runs right behind the transform kernel (A/B) */
#define SVC_CLIP_TUNE_WHOLE_SHARD_STEPS 128u /* never the idle-pipeline rule (a step that finds the pipeline empty runs in two chunks on big
                                                shards in the two-pass order: clip_encoder.hpp); A/B */

typedef struct svc_clip_config {
  uint32_t struct_size;   /* sizeof(svc_clip_config) of the caller's build: svc_clip_create refuses any other value, so a
                             caller compiled against an older / newer layout fails loudly instead of feeding garbage into
                             the trailing fields */
  uint32_t width, height; /* source size (padded per libs/encoder.cpp:164-168) */
  uint32_t levels, mv_block, search_range;
  uint32_t dct_block_w, dct_block_h; /* 0 = no transform */
  uint32_t fg_step, bg_step;
  uint32_t wire;         /* 1: serialised records (libs/encoder.cpp:222-269) of the raw coefficients instead of quantised planes */
  uint32_t segmentation; /* 0: in-repo part only (foreground = one region) */
  uint64_t seed;
  svc_ransac_params ransac;
  svc_segment_params segment;
  uint32_t clip_frames; /* frames of the whole clip */
  uint32_t rank, world; /* this handle holds shard `rank` of `world` */
  uint32_t schedule;    /* SVC_CLIP_SERIAL / SVC_CLIP_PIPELINED */
  uint32_t chunk_pairs; /* pipelined, one rank: frame pairs per chunk of a step (a step's stages then overlap each other chunk by chunk, so a
                           clip encoded ONCE does not pay RANSAC + segmentation end to end; clip_encoder.hpp); 0 = whole-shard launches
                           (one chunk: best steady state; two chunks at 1080p x 300 frames trade 0-4 % of it for up to 5 % on a
                           once-through clip).  (Was `reserved0`, always 0, through round 5.) */
  /* Tuning; all zero = the defaults.  These are the A/B switches of the measurements under profiles/ -- they change
   * launch shapes and kernel choice, never results. */
  uint32_t hbma_flags;  /* SVC_HBMA_* passed to svc_hip_hbma_pairs (0 = SVC_HBMA_AUTO) */
  uint32_t lat_depth;   /* pipelined: iterations RANSAC + segmentation get to finish, 1..3 (0 = 2) */
  uint32_t tuning;      /* SVC_CLIP_TUNE_* bits */
} svc_clip_config;

typedef struct svc_clip_info {
  uint32_t padded_w, padded_h, mv_field_w, mv_field_h, blocks, ransac_iters;
  uint64_t pyramid_stride, frame_bytes, record_bytes;
  uint32_t first_frame, frames, pairs, first_encoded, needs_halo;
  uint32_t chunks_per_step; /* launches of every stage per svc_clip_step (1 = whole-shard launches) */
  uint32_t output_sets;     /* sets the coefficient planes / records exist in (memory: output_sets x pairs x frame's output) */
  uint32_t reserved;
} svc_clip_info;

/* stage ids for svc_clip_stage_time */
enum { SVC_STAGE_LUMA_PYRAMID = 0, SVC_STAGE_HALO, SVC_STAGE_HBMA, SVC_STAGE_RANSAC, SVC_STAGE_SEGMENT,
       SVC_STAGE_TRANSFORM, SVC_STAGE_TYPE_PATCH /* wire, one BGR pass: region ids into the records emitted before they existed */,
       SVC_STAGE_COUNT };
/* buffer ids for svc_clip_output / svc_clip_read */
enum { SVC_BUF_MV = 0, SVC_BUF_MIN_MAD, SVC_BUF_GLOBAL_MOTION, SVC_BUF_RMSE, SVC_BUF_INLIER_MASK,
       SVC_BUF_INLIER_COUNT, SVC_BUF_BLOCK_TYPES, SVC_BUF_COEFFS, SVC_BUF_RECORDS, SVC_BUF_PYRAMIDS,
       SVC_BUF_BGR, SVC_BUF_COUNT };

/* Halo transport override: must enqueue on `stream` the send of `bytes` from d_send to rank + 1
 * (if any) and the receive into d_recv from rank - 1 (if any). */
typedef int (*svc_clip_halo_fn)(const uint8_t* d_send, uint8_t* d_recv, uint64_t bytes, void* stream,
                                void* user);

const char* svc_clip_last_error(void);

/* Consecutive chunks; the first (clip_frames % world) ranks hold one frame more. */
int svc_clip_plan_shard(uint32_t clip_frames, uint32_t world, uint32_t rank, uint32_t* first_frame,
                        uint32_t* frames, uint32_t* pairs, uint32_t* first_encoded);

/* 1 (with svc_clip_last_error set) for a null pointer, config->struct_size != sizeof(svc_clip_config), an hbma_flags / tuning
 * bit this build does not know, lat_depth > 3, or any configuration svc::ClipEncoder rejects. */
int svc_clip_create(const svc_clip_config* config, svc_clip** out);
void svc_clip_destroy(svc_clip* clip);
int svc_clip_get_info(svc_clip* clip, svc_clip_info* out);

/* n PADDED B,G,R u8 frames into shard slots [first_local, first_local + n); synchronous. */
int svc_clip_load_frames(svc_clip* clip, const uint8_t* src, uint32_t first_local, uint32_t n,
                         int src_on_device);

int svc_clip_set_comm(svc_clip* clip, void* nccl_comm); /* from svc_hip_comm_create */
int svc_clip_set_halo_callback(svc_clip* clip, svc_clip_halo_fn fn, void* user);

int svc_clip_step(svc_clip* clip, int timed); /* enqueue one pass over the shard */
/* A stream of clips, each encoded once (svc::ClipEncoder::StepFrames): one pass over the shard's frames where the caller has them in DEVICE
   memory (frames x padded_h x padded_w x 3, the layout svc_clip_load_frames fills) -- no copy and no drain of the pipeline between clips.
   *step (optional) receives the step's number; the frames must stay untouched until svc_clip_wait_step(clip, that number) or svc_clip_sync. */
int svc_clip_step_frames(svc_clip* clip, const uint8_t* device_frames, int timed, uint32_t* step);
int svc_clip_wait_step(svc_clip* clip, uint32_t step); /* returns once nothing reads that step's frames any more */
int svc_clip_flush(svc_clip* clip);           /* enqueue what the pipeline still holds */
int svc_clip_sync(svc_clip* clip);            /* flush + wait for the GPU */

/* HIP-event time of a stage summed over the timed steps, and the launches it covers. */
int svc_clip_stage_time(svc_clip* clip, uint32_t stage, double* total_ms, uint32_t* launches);
/* Frame pairs the timed launches of a stage covered (a step's launches are its chunks): time per step = total_ms x pairs / this. */
int svc_clip_stage_pairs(svc_clip* clip, uint32_t stage, uint64_t* pairs);
int svc_clip_reset_timers(svc_clip* clip);

/* The speculation policy (planes + quant, clip_encoder.hpp): forget what it has measured (syncs first) ... */
int svc_clip_reset_policy(svc_clip* clip);
/* ... and what it did so far: chunk launches that had the choice, those that speculated, the newest foreground share known (-1: none). */
int svc_clip_policy_info(svc_clip* clip, uint64_t* chunks_decided, uint64_t* chunks_speculated, double* foreground_share);

/* The newest finished step's output (syncs first). */
int svc_clip_output(svc_clip* clip, uint32_t buffer, void** d_ptr, uint64_t* bytes);
int svc_clip_read(svc_clip* clip, uint32_t buffer, uint64_t offset, void* dst, uint64_t bytes,
                  int dst_on_device);

#ifdef __cplusplus
}
#endif

#endif /* SVC_CLIP_H */
