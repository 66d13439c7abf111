// svc_common.hpp -- error plumbing shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "svc_hip.h"

namespace svc {

// Per-thread message behind svc_hip_last_error().
char* last_error_buf();
constexpr int kErrBufSize = 512;

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), kErrBufSize, fmt, ap);
  va_end(ap);
  return code;
}

#define SVC_HIP_TRY(expr)                                                       \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess)                                                       \
      return ::svc::fail(SVC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                         hipGetErrorString(e_), __FILE__, __LINE__);            \
  } while (0)

#define SVC_REQUIRE(cond, ...)                                                  \
  do {                                                                          \
    if (!(cond)) return ::svc::fail(SVC_ERR_INVALID_ARG, __VA_ARGS__);          \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(SVC_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return SVC_OK;
}

inline uint64_t pyramid_bytes(uint32_t w, uint32_t h, uint32_t levels) {
  uint64_t n = 0;
  for (uint32_t l = 0; l < levels; ++l) n += (uint64_t)(w >> l) * (h >> l);
  return n;
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Per-thread staging of the host-pointer entry points (*_host): one internal stream, one device buffer and one
// pinned bounce buffer, grown on demand and kept (capi.hip owns the thread_local instance).
struct Staging {
  hipStream_t stream = nullptr;
  uint8_t* dev = nullptr;
  uint8_t* pin = nullptr;
  size_t cap = 0;
  hipEvent_t piece[8] = {};  // D2H of a large result in pieces: piece k's event, so that its copy-out overlaps piece k + 1's transfer
  ~Staging();
  int ensure(size_t bytes);
  // device -> pinned -> caller's memory, `bytes` from d_src (staged at pin_off) to dst[i] (n_dst equal runs, or one): the transfer goes
  // out in pieces and each piece is copied out (by the copy crew) while the next one is still on the link.  Synchronises the stream.
  int download(const uint8_t* d_src, size_t pin_off, size_t bytes, uint8_t* const* dst, uint32_t n_dst);
};
Staging& host_stage();
// pageable <-> pinned copies of the host-pointer entry points: by several threads above 1 MB (csrc/host/copy_crew.hpp)
void host_copy(void* dst, const void* src, size_t bytes);
int require_device();
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so blocks b and
// b + 8 share an L2 but b and b + 1 do not.  This bijective remap gives every XCD one
// CONTIGUOUS range of logical work (rows that overlap stay in one L2).  Speed only: nothing
// depends on the placement (cdna_hip_programming.md T1).
__device__ __forceinline__ uint32_t xcd_contiguous_block(uint32_t bid, uint32_t nblocks) {
  const uint32_t q = nblocks >> 3, r = nblocks & 7u, xcd = bid & 7u, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- kernel launchers (defined in the .hip files) ---------------------------

int launch_stream_probe(const void* d_in, void* d_out, uint64_t bytes, uint32_t reads, uint32_t writes, hipStream_t stream);

int launch_hbma(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride,
                uint32_t n_pairs, uint32_t levels, uint32_t w, uint32_t h, uint32_t range,
                uint32_t bw, uint32_t bh, float* d_mv, float* d_mad, uint32_t flags,
                hipStream_t stream);
int launch_ebma(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride,
                uint32_t n_pairs, uint32_t w, uint32_t h, uint32_t range, uint32_t bw,
                uint32_t bh, float* d_mv, float* d_mad, hipStream_t stream);
int launch_dct(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames, uint32_t w,
               uint32_t h, uint32_t bw, uint32_t bh, const uint32_t* d_types, uint32_t mv_bw,
               uint32_t mv_bh, uint32_t fg_step, uint32_t bg_step, bool quant, float* d_planes,
               hipStream_t stream, uint8_t* d_records = nullptr, uint64_t records_stride = 0,
               uint32_t emit_h = 0, uint8_t* d_luma = nullptr, uint64_t luma_stride = 0);
int launch_dct_quant_speculative(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t block,
                                 uint32_t bg_step, float* d_planes, uint8_t* d_luma, uint64_t luma_stride, hipStream_t stream);
uint64_t dct_redo_workspace_bytes(uint32_t n_frames, uint32_t mv_blocks);
int launch_count_foreground(const uint32_t* d_types, uint64_t n, uint32_t* d_count, hipStream_t stream);
int launch_dct_quant_redo_foreground(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t block,
                                     const uint32_t* d_types, uint32_t mv_bw, uint32_t mv_bh, uint32_t fg_step, float* d_planes,
                                     uint8_t* d_ws, hipStream_t stream);
int launch_wire_patch_types(const uint32_t* d_types, uint32_t n_frames, uint32_t frame_w, uint32_t frame_h, uint32_t emit_h, uint32_t tb,
                            uint32_t mv_bw, uint32_t mv_bh, uint8_t* d_out, uint64_t out_stride, bool force_all, hipStream_t stream);
int launch_dct_tiles(float* d_img, uint32_t w, uint32_t h, uint32_t bw, uint32_t bh, const uint32_t* d_xy, uint32_t n_tiles,
                     hipStream_t stream);
int launch_quant(float* d_coeffs, uint64_t n, uint32_t step, hipStream_t stream);
int launch_quant_frames(float* d_planes, uint32_t n_frames, uint32_t w, uint32_t h,
                        uint32_t mv_bw, uint32_t mv_bh, const uint32_t* d_types,
                        uint32_t fg_step, uint32_t bg_step, hipStream_t stream);
int launch_ransac(const float* d_mv, uint32_t blocks, uint32_t n_frames,
                  svc_ransac_params params, const uint32_t* d_samples, uint32_t iters,
                  float* d_gm, float* d_rmse, uint8_t* d_mask, uint32_t* d_count,
                  uint32_t flags, hipStream_t stream);
int launch_ransac_rmse(const float* d_mv, uint32_t blocks, uint32_t n_frames, svc_ransac_params params, const float* d_gm,
                       const uint8_t* d_mask, const uint32_t* d_count, float* d_rmse, hipStream_t stream);
int launch_block_types(const uint8_t* d_mask, uint64_t n, uint32_t* d_types, hipStream_t stream);
int launch_serialize(const float* d_planes, uint64_t plane_elems, uint32_t n_frames, const uint32_t* d_types,
                     uint32_t mv_blocks, uint32_t frame_w, uint32_t frame_h, uint32_t tbw, uint32_t tbh,
                     uint32_t mfw, uint32_t mv_bw, uint32_t mv_bh, uint8_t* d_out, uint64_t out_stride,
                     hipStream_t stream);
int launch_decode(const float* d_planes, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t block,
                  const uint32_t* d_types, uint32_t mv_bw, uint32_t mv_bh, uint32_t fg_step, uint32_t bg_step,
                  uint32_t gx, uint32_t gy, uint32_t gw, uint32_t gh, float* d_bgr, hipStream_t stream);
int launch_sse(const uint8_t* d_src, uint64_t src_stride, const float* d_rec, uint32_t n_frames, uint32_t w,
               uint32_t h, uint32_t region_w, uint32_t region_h, uint64_t* d_sse, hipStream_t stream);
uint64_t segment_workspace_per_frame(uint32_t n, uint32_t attempts);
int launch_segment(const uint8_t* d_mask, const float* d_mv, uint32_t mfw, uint32_t mfh, uint32_t n_frames,
                   uint32_t mv_bw, uint32_t mv_bh, const svc_segment_params& p, uint64_t seed, uint8_t* d_ws,
                   uint32_t* d_types, uint32_t flags, hipStream_t stream);
uint64_t global_ebma_workspace_bytes(uint32_t range, uint32_t n_pairs);
int launch_global_ebma(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride, uint32_t n_pairs, uint32_t w,
                       uint32_t h, uint32_t range, uint8_t* d_ws, float* d_gm, float* d_min_mad, bool combine,
                       hipStream_t stream);
int launch_global_avg(const float* d_mv, uint32_t blocks, uint32_t n_frames, float* d_out, hipStream_t stream);
int launch_pyr_down_levels(uint8_t* d_pyr, uint64_t pyr_stride, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t levels,
                           uint32_t first_plain_level, hipStream_t stream);
int launch_luma_pyramid(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames,
                        uint32_t w, uint32_t h, uint32_t levels, uint8_t* d_pyr,
                        uint64_t pyr_stride, hipStream_t stream);

}  // namespace svc
