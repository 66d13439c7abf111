// capi.hip -- the extern "C" surface declared in include/svc_hip.h.
//
// Device-pointer entry points validate, pick a kernel and enqueue.  Host-pointer
// entry points add pinned staging, an internal stream and a final synchronise, so
// that the C++ wrappers of include/svc/motion.hpp behave like the reference's
// synchronous calls (libs/encoder.cpp:472-498).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <malloc.h>

#include "host/copy_crew.hpp"

#include <vector>
#include "svc_common.hpp"

namespace svc {

static thread_local char g_err[kErrBufSize] = "";
char* last_error_buf() { return g_err; }

bool fused_supported(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw, uint32_t bh);
bool tiled_supported(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw, uint32_t bh);
bool tiled_is_default();
int launch_hbma_fused(const uint8_t*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t,
                      uint32_t, uint32_t mv_block, float*, float*, int kernel, hipStream_t);
int launch_hbma_wave(const uint8_t*, const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t,
                     uint32_t, uint32_t, uint32_t, float*, float*, hipStream_t);

static inline bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// The reference's asserts (libs/motion.cpp:417-433) as a status, plus what a
// pyramid needs to be well formed (every level an exact halving).
static int validate_hbma(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw, uint32_t bh) {
  SVC_REQUIRE(levels > 0 && levels <= 16, "hbma: level_count %u out of range (motion.cpp:422)", levels);
  SVC_REQUIRE(bw > 0 && bh > 0, "hbma: block %ux%u must be positive (motion.cpp:423-424)", bw, bh);
  SVC_REQUIRE(w > 0 && h > 0, "hbma: frame %ux%u must be positive (motion.cpp:425-426)", w, h);
  SVC_REQUIRE(w % bw == 0 && h % bh == 0, "hbma: frame %ux%u not divisible by block %ux%u (motion.cpp:428-429)", w, h, bw, bh);
  const uint32_t f = 1u << (levels - 1);
  SVC_REQUIRE(range >= f, "hbma: search_range %u < 2^(levels-1) = %u (motion.cpp:433)", range, f);
  SVC_REQUIRE(bw % f == 0 && bh % f == 0 && w % f == 0 && h % f == 0,
              "hbma: block %ux%u and frame %ux%u must be divisible by 2^(levels-1) = %u", bw, bh, w, h, f);
  return SVC_OK;
}

int launch_hbma(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride, uint32_t n_pairs,
                uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw, uint32_t bh,
                float* d_mv, float* d_mad, uint32_t flags, hipStream_t stream) {
  const bool can_fuse = fused_supported(levels, w, h, range, bw, bh) && aligned(d_tracked, 4) &&
                        aligned(d_anchor, 4) && pair_stride % 4 == 0 && aligned(d_mv, 8);
  // which all-level kernel: the shape's default, or the one the caller names (A/B runs and the parity tests)
  const int kernel = (flags & SVC_HBMA_FORCE_TILED) ? 2 : (flags & SVC_HBMA_FORCE_LANE) ? 1 : 0;
  if (flags & (SVC_HBMA_FORCE_FUSED | SVC_HBMA_FORCE_TILED | SVC_HBMA_FORCE_LANE)) {
    if (!can_fuse)
      return fail(SVC_ERR_UNSUPPORTED, "hbma: fused kernel needs square blocks of 8 / 16 / 32, 2 .. log2(block) levels, r_top in 1 .. 4, 4-byte aligned planes");
    return launch_hbma_fused(d_tracked, d_anchor, pair_stride, n_pairs, levels, w, h, range, bw, d_mv, d_mad, kernel, stream);
  }
  if (can_fuse && !(flags & SVC_HBMA_FORCE_WAVE_PER_BLOCK))
    return launch_hbma_fused(d_tracked, d_anchor, pair_stride, n_pairs, levels, w, h, range, bw, d_mv, d_mad, 0, stream);
  return launch_hbma_wave(d_tracked, d_anchor, pair_stride, n_pairs, levels, w, h, range, bw, bh, d_mv, d_mad, stream);
}

// ---- per-thread staging for the host-pointer entry points ----------------------
Staging::~Staging() {
  for (hipEvent_t e : piece)
    if (e) (void)hipEventDestroy(e);
  if (dev) (void)hipFree(dev);
  if (pin) (void)hipHostFree(pin);
  if (stream) (void)hipStreamDestroy(stream);
}
int Staging::ensure(size_t bytes) {
  if (!stream) SVC_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  if (bytes <= cap) return SVC_OK;
  if (dev) { (void)hipFree(dev); dev = nullptr; }
  if (pin) { (void)hipHostFree(pin); pin = nullptr; }
  cap = 0;
  size_t want = (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
  SVC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), want));
  SVC_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pin), want, hipHostMallocDefault));
  cap = want;
  return SVC_OK;
}
static thread_local Staging g_stage;
Staging& host_stage() { return g_stage; }

// One crew for the process, started by the first large copy (four threads with the caller: a 25 MB result leaves the pinned buffer in
// ~0.3 ms instead of ~1 ms).  Never destroyed: a process may exit while another of its threads is still inside a call.
void host_copy(void* dst, const void* src, size_t bytes) {
  if (bytes < (1u << 20)) { std::memcpy(dst, src, bytes); return; }
  static CopyCrew* crew = new CopyCrew(3);
  crew->Copy(dst, src, bytes);
}

int Staging::download(const uint8_t* d_src, size_t pin_off, size_t bytes, uint8_t* const* dst, uint32_t n_dst) {
  const size_t run = bytes / n_dst;  // bytes per destination
  constexpr uint32_t kPieces = 8;
  const uint32_t pieces = bytes >= (4u << 20) ? kPieces : 1;
  // piece boundaries: multiples of 4 KiB, except that a destination's end is always one (so a piece never spans two destinations)
  size_t cut[kPieces * 4 + 2];
  uint32_t n_cut = 0;
  {
    const size_t step = ((bytes / pieces) + 4095) & ~(size_t)4095;
    size_t at = 0;
    cut[n_cut++] = 0;
    while (at < bytes) {
      size_t next = std::min(bytes, at + step);
      const size_t dest_end = (at / run + 1) * run;
      if (next > dest_end) next = dest_end;
      cut[n_cut++] = next;
      at = next;
      if (n_cut + 1 >= sizeof(cut) / sizeof(cut[0])) { cut[n_cut - 1] = bytes; break; }  // cannot happen with n_dst <= 3 * pieces
    }
  }
  const uint32_t n = n_cut - 1;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t slot = k % 8;
    if (!piece[slot]) SVC_HIP_TRY(hipEventCreateWithFlags(&piece[slot], hipEventDisableTiming));
    if (k >= 8) {  // the slot's previous piece: copy it out before its event is reused
      const uint32_t j = k - 8;
      SVC_HIP_TRY(hipEventSynchronize(piece[slot]));
      host_copy(dst[cut[j] / run] + cut[j] % run, pin + pin_off + cut[j], cut[j + 1] - cut[j]);
    }
    SVC_HIP_TRY(hipMemcpyAsync(pin + pin_off + cut[k], d_src + cut[k], cut[k + 1] - cut[k], hipMemcpyDeviceToHost, stream));
    SVC_HIP_TRY(hipEventRecord(piece[slot], stream));
  }
  for (uint32_t j = n > 8 ? n - 8 : 0; j < n; ++j) {
    SVC_HIP_TRY(hipEventSynchronize(piece[j % 8]));
    host_copy(dst[cut[j] / run] + cut[j] % run, pin + pin_off + cut[j], cut[j + 1] - cut[j]);
  }
  SVC_HIP_TRY(hipStreamSynchronize(stream));
  return SVC_OK;
}

int require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(SVC_ERR_NO_DEVICE, "no HIP device visible (%s); this library has no CPU path",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  return SVC_OK;
}

}  // namespace svc

using namespace svc;

extern "C" {

const char* svc_hip_last_error(void) { return g_err; }
int svc_hip_abi_version(void) { return 5; }  // 5: + svc_hip_tune_host_allocator, svc_hip_host_tuning_requested (4: svc_hip_dct_planes_host, imageops.hip)

// Opt-in only (include/svc_hip.h): nothing in this library calls it on its own.
int svc_hip_tune_host_allocator(uint32_t flags) {
  SVC_REQUIRE((flags & ~(SVC_HOST_KEEP_LARGE_BLOCKS | SVC_HOST_ONE_ARENA)) == 0, "tune_host_allocator: unknown flag bits 0x%x", flags);
  if (flags & SVC_HOST_KEEP_LARGE_BLOCKS) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
  }
  if (flags & SVC_HOST_ONE_ARENA) mallopt(M_ARENA_MAX, 1);
  return SVC_OK;
}
int svc_hip_host_tuning_requested(void) {
  const char* e = std::getenv("SVC_KEEP_LARGE_BLOCKS");
  return e && e[0] == '1' && e[1] == 0;
}

int svc_hip_device_count(int* count) {
  SVC_REQUIRE(count, "device_count: null output");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *count = e == hipSuccess ? n : 0;
  return SVC_OK;
}

int svc_hip_probe_stream(const void* d_in, void* d_out, uint64_t bytes, uint32_t reads, uint32_t writes, void* stream) {
  SVC_REQUIRE(d_in && d_out, "probe: null pointer");
  SVC_REQUIRE(aligned(d_in, 16) && aligned(d_out, 16), "probe: buffers must be 16-byte aligned");
  return launch_stream_probe(d_in, d_out, bytes, reads, writes, static_cast<hipStream_t>(stream));
}

uint64_t svc_hip_pyramid_bytes(uint32_t w, uint32_t h, uint32_t levels) { return pyramid_bytes(w, h, levels); }

int svc_hip_hbma_pairs(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride_bytes,
                       uint32_t n_pairs, uint32_t level_count, uint32_t frame_w, uint32_t frame_h,
                       uint32_t search_range, uint32_t block_w, uint32_t block_h, float* d_mv_xy,
                       float* d_min_mad, uint32_t flags, void* stream) {
  if (n_pairs == 0) return SVC_OK;  // empty batch: nothing to enqueue
  SVC_REQUIRE(d_tracked && d_anchor && d_mv_xy && d_min_mad, "hbma: null pointer (motion.cpp:417-420)");
  int rc = validate_hbma(level_count, frame_w, frame_h, search_range, block_w, block_h);
  if (rc) return rc;
  SVC_REQUIRE(n_pairs <= 1 || pair_stride_bytes >= pyramid_bytes(frame_w, frame_h, level_count),
              "hbma: pair stride %llu smaller than one pyramid", (unsigned long long)pair_stride_bytes);
  return launch_hbma(d_tracked, d_anchor, pair_stride_bytes, n_pairs, level_count, frame_w, frame_h,
                     search_range, block_w, block_h, d_mv_xy, d_min_mad, flags, static_cast<hipStream_t>(stream));
}

const char* svc_hip_hbma_kernel_name(uint32_t level_count, uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                                     uint32_t block_w, uint32_t block_h, uint32_t flags) {
  // launch_hbma's and launch_hbma_fused's choice, for aligned pyramids
  if (validate_hbma(level_count, frame_w, frame_h, search_range, block_w, block_h)) return nullptr;
  const bool can_fuse = fused_supported(level_count, frame_w, frame_h, search_range, block_w, block_h);
  const bool can_tile = tiled_supported(level_count, frame_w, frame_h, search_range, block_w, block_h);
  const bool forced = flags & (SVC_HBMA_FORCE_FUSED | SVC_HBMA_FORCE_TILED | SVC_HBMA_FORCE_LANE);
  if ((forced && !can_fuse) || ((flags & SVC_HBMA_FORCE_TILED) && !can_tile)) {
    fail(SVC_ERR_UNSUPPORTED, "hbma: the kernel the flags name does not cover this shape");
    return nullptr;
  }
  if (!forced && (!can_fuse || (flags & SVC_HBMA_FORCE_WAVE_PER_BLOCK))) return "hbma_wave_level_kernel";
  if (flags & SVC_HBMA_FORCE_TILED) return "hbma_tiled16_kernel";
  if (flags & SVC_HBMA_FORCE_LANE) return "hbma_fused_kernel";
  return can_tile && tiled_is_default() ? "hbma_tiled16_kernel" : "hbma_fused_kernel";
}

int svc_hip_ebma_pairs(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride_bytes,
                       uint32_t n_pairs, uint32_t frame_w, uint32_t frame_h, uint32_t search_range,
                       uint32_t block_w, uint32_t block_h, float* d_mv_xy, float* d_min_mad, void* stream) {
  if (n_pairs == 0) return SVC_OK;  // empty batch: nothing to enqueue
  SVC_REQUIRE(d_tracked && d_anchor && d_mv_xy && d_min_mad, "ebma: null pointer (motion.cpp:273-276)");
  SVC_REQUIRE(block_w > 0 && block_h > 0, "ebma: block must be positive (motion.cpp:278-279)");
  SVC_REQUIRE(frame_w > 0 && frame_h > 0 && frame_w % block_w == 0 && frame_h % block_h == 0,
              "ebma: frame %ux%u not divisible by block %ux%u (motion.cpp:281-282)", frame_w, frame_h, block_w, block_h);
  return launch_ebma(d_tracked, d_anchor, pair_stride_bytes, n_pairs, frame_w, frame_h, search_range,
                     block_w, block_h, d_mv_xy, d_min_mad, static_cast<hipStream_t>(stream));
}

uint32_t svc_hip_ransac_iter_count(svc_ransac_params p) {
  // libs/motion.cpp:144-149, f32 throughout
  float num = std::log(1 - p.success_prob);
  float den = std::log(1 - std::pow(p.inlier_ratio, (float)p.subset_sz));
  return (uint32_t)std::ceil(num / den);
}

int svc_hip_ransac_frames(const float* d_mv_xy, uint32_t blocks, uint32_t n_frames, svc_ransac_params params,
                          const uint32_t* d_samples, uint32_t iter_count, float* d_gm_xy, float* d_rmse,
                          uint8_t* d_inlier_mask, uint32_t* d_inlier_count, void* stream) {
  return svc_hip_ransac_frames_ex(d_mv_xy, blocks, n_frames, params, d_samples, iter_count, d_gm_xy, d_rmse, d_inlier_mask,
                                  d_inlier_count, 0, stream);
}

int svc_hip_ransac_frames_ex(const float* d_mv_xy, uint32_t blocks, uint32_t n_frames, svc_ransac_params params,
                             const uint32_t* d_samples, uint32_t iter_count, float* d_gm_xy, float* d_rmse,
                             uint8_t* d_inlier_mask, uint32_t* d_inlier_count, uint32_t flags, void* stream) {
  if (n_frames == 0) return SVC_OK;  // empty batch: nothing to enqueue
  SVC_REQUIRE(d_mv_xy && d_gm_xy && d_rmse && d_inlier_mask && d_inlier_count, "ransac: null pointer (motion.cpp:189-192)");
  SVC_REQUIRE(iter_count == 0 || d_samples, "ransac: null samples");
  SVC_REQUIRE(params.subset_sz > 0 && blocks >= params.subset_sz,
              "ransac: motion field of %u smaller than subset %u (motion.cpp:194)", blocks, params.subset_sz);
  SVC_REQUIRE(aligned(d_mv_xy, 8), "ransac: motion field must be 8-byte aligned");
  return launch_ransac(d_mv_xy, blocks, n_frames, params, d_samples, iter_count, d_gm_xy, d_rmse,
                       d_inlier_mask, d_inlier_count, flags, static_cast<hipStream_t>(stream));
}

int svc_hip_ransac_rmse_frames(const float* d_mv_xy, uint32_t blocks, uint32_t n_frames, svc_ransac_params params,
                               const float* d_gm_xy, const uint8_t* d_inlier_mask, const uint32_t* d_inlier_count,
                               float* d_rmse, void* stream) {
  if (n_frames == 0) return SVC_OK;
  SVC_REQUIRE(d_mv_xy && d_gm_xy && d_rmse && d_inlier_mask && d_inlier_count, "ransac rmse: null pointer");
  SVC_REQUIRE(params.subset_sz > 0 && blocks >= params.subset_sz, "ransac rmse: motion field of %u smaller than subset %u", blocks,
              params.subset_sz);
  SVC_REQUIRE(aligned(d_mv_xy, 8), "ransac rmse: motion field must be 8-byte aligned");
  return launch_ransac_rmse(d_mv_xy, blocks, n_frames, params, d_gm_xy, d_inlier_mask, d_inlier_count, d_rmse,
                            static_cast<hipStream_t>(stream));
}

int svc_hip_block_types_frames(const uint8_t* d_inlier_mask, uint32_t blocks, uint32_t n_frames,
                                uint32_t* d_block_types, void* stream) {
  if (n_frames == 0) return SVC_OK;  // empty batch: nothing to enqueue
  SVC_REQUIRE(d_inlier_mask && d_block_types, "block_types: null pointer");
  return launch_block_types(d_inlier_mask, (uint64_t)blocks * n_frames, d_block_types,
                            static_cast<hipStream_t>(stream));
}

static uint32_t lcm_u32(uint32_t a, uint32_t b) {
  uint32_t x = a, y = b;
  while (y) { const uint32_t t = x % y; x = y; y = t; }
  return a / x * b;
}

int svc_hip_wire_header(uint32_t clip_frame_count, uint32_t frame_w, uint32_t frame_h, uint32_t mv_block_w,
                        uint32_t mv_block_h, uint32_t level_count, uint32_t transform_block_w,
                        uint32_t transform_block_h, svc_wire_header* out) {
  SVC_REQUIRE(out, "wire_header: null output");
  SVC_REQUIRE(mv_block_w > 0 && mv_block_h > 0 && level_count > 0 && level_count <= 16, "wire_header: invalid block / level count");
  const uint32_t f = 1u << (level_count - 1);
  // libs/math.hpp:276-283 (ClosestLargerDivisible) as used at libs/encoder.cpp:164-168
  const uint32_t lw = lcm_u32(mv_block_w, f), lh = lcm_u32(mv_block_h, f);
  const uint32_t pw = (frame_w + lw - 1) / lw * lw, ph = (frame_h + lh - 1) / lh * lh;
  out->frame_count = clip_frame_count > 0 ? clip_frame_count - 1 : 0;  // encoder.cpp:361-367
  out->frame_w = frame_w;
  out->frame_h = frame_h;
  out->frame_excess_w = pw - frame_w;
  out->frame_excess_h = ph - frame_h;
  out->transform_block_w = transform_block_w;
  out->transform_block_h = transform_block_h;
  out->channel_count = 3;
  return SVC_OK;
}

uint64_t svc_hip_serialized_frame_bytes(uint32_t frame_w, uint32_t frame_h, uint32_t tbw, uint32_t tbh) {
  if (!tbw || !tbh) return 0;
  return (uint64_t)div_up(frame_w, tbw) * div_up(frame_h, tbh) * (4ull + 12ull * tbw * tbh);
}

int svc_hip_serialize_frames(const float* d_planes, uint64_t plane_elems, uint32_t n_frames,
                             const uint32_t* d_block_types, uint32_t frame_w, uint32_t frame_h, uint32_t tbw,
                             uint32_t tbh, uint32_t mv_field_w, uint32_t mv_field_h, uint32_t mv_block_w,
                             uint32_t mv_block_h, uint8_t* d_out, uint64_t out_stride_bytes, void* stream) {
  if (n_frames == 0) return SVC_OK;  // empty batch: nothing to enqueue
  SVC_REQUIRE(d_planes && d_block_types && d_out, "serialize: null pointer");
  SVC_REQUIRE(tbw > 0 && tbh > 0, "serialize: transform block must be positive (encoder.cpp:227-228)");
  SVC_REQUIRE(frame_w > 0 && frame_h > 0 && mv_block_w > 0 && mv_block_h > 0, "serialize: empty frame");
  // The reference's asserts (encoder.cpp:230-239, with its swapped w / h) are NOT preconditions here: its documented build (Release,
  // README.md:123) compiles them out, and configurations its own Validate admits trip them -- the encoder hands SerializeEncodedFrame
  // the UNPADDED frame size (:647-650), divisible by the transform block only by accident (a 344-pixel frame with 16 x 16 blocks), and
  // with non-square blocks the swapped comparison fails where the real one holds (MV blocks 32 x 8 with 16 x 8 tiles: "16 <= 8").  The
  // loops run as they run there; what IS required is that every read stays inside the planes and the motion field.
  // every block type and coefficient the loops touch must exist (the reference would read out of bounds)
  const uint32_t last_x = (div_up(frame_w, tbw) - 1) * tbw, last_y = (div_up(frame_h, tbh) - 1) * tbh;
  SVC_REQUIRE(last_x / mv_block_w < mv_field_w && last_y / mv_block_h < mv_field_h,
              "serialize: motion field %ux%u does not cover the frame", mv_field_w, mv_field_h);
  SVC_REQUIRE((uint64_t)(last_y + tbw - 1) * frame_w + last_x + tbh <= plane_elems,
              "serialize: planes of %llu floats are too small for %ux%u", (unsigned long long)plane_elems, frame_w, frame_h);
  SVC_REQUIRE(out_stride_bytes % 4 == 0 && aligned(d_out, 4) &&
                  (n_frames <= 1 || out_stride_bytes >= svc_hip_serialized_frame_bytes(frame_w, frame_h, tbw, tbh)),
              "serialize: output stride %llu too small or unaligned", (unsigned long long)out_stride_bytes);
  return launch_serialize(d_planes, plane_elems, n_frames, d_block_types, mv_field_w * mv_field_h, frame_w, frame_h,
                          tbw, tbh, mv_field_w, mv_block_w, mv_block_h, d_out, out_stride_bytes,
                          static_cast<hipStream_t>(stream));
}

uint64_t svc_hip_segment_workspace_bytes(uint32_t mv_field_w, uint32_t mv_field_h, uint32_t n_frames,
                                         uint32_t attempt_count) {
  return segment_workspace_per_frame(mv_field_w * mv_field_h, attempt_count) * n_frames;
}

int svc_hip_segment_frames(const uint8_t* d_inlier_mask, const float* d_mv_xy, uint32_t mv_field_w,
                           uint32_t mv_field_h, uint32_t n_frames, uint32_t mv_block_w, uint32_t mv_block_h,
                           svc_segment_params params, uint64_t seed, uint8_t* d_workspace,
                           uint64_t workspace_bytes, uint32_t* d_block_types, void* stream) {
  return svc_hip_segment_frames_ex(d_inlier_mask, d_mv_xy, mv_field_w, mv_field_h, n_frames, mv_block_w, mv_block_h, params,
                                   seed, d_workspace, workspace_bytes, d_block_types, 0, stream);
}

// Beyond the fused kernels' table sizes (cluster_count > 64 or attempt_count > 16; the reference's Validate admits any positive count,
// libs/encoder.cpp:39-61): libs/encoder.cpp:507-623 composed the reference's way from the per-call entry points -- mask, close, open,
// index list, BuildMvFeatures (with its m.y overwrite), kmeans, one connectedComponents per cluster, offset numbering -- frame by frame
// through host memory.  Same definitions, same bits as the fused form (tests/test_gpu_segment.py); SYNCHRONOUS: it waits for `stream`
// and returns with the region ids in place.  Up to what the per-call k-means takes (255 clusters, 64 attempts).
static int segment_frames_by_calls(const uint8_t* d_mask, const float* d_mv, uint32_t mfw, uint32_t mfh, uint32_t n_frames, uint32_t bw,
                                   uint32_t bh, const svc_segment_params& p, uint64_t seed, uint32_t* d_types, hipStream_t stream) {
  const size_t n = (size_t)mfw * mfh;
  std::vector<uint8_t> mask(n_frames * n), fg(n), tmp(n), one(n);
  std::vector<float> mv(n_frames * n * 2), feats;
  std::vector<uint32_t> types(n_frames * n, 0u), idx;
  std::vector<int32_t> labels, cc(n);
  SVC_HIP_TRY(hipMemcpyAsync(mask.data(), d_mask, mask.size(), hipMemcpyDeviceToHost, stream));
  SVC_HIP_TRY(hipMemcpyAsync(mv.data(), d_mv, mv.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
  SVC_HIP_TRY(hipStreamSynchronize(stream));
  for (uint32_t f = 0; f < n_frames; ++f) {
    for (size_t i = 0; i < n; ++i) fg[i] = mask[f * n + i] ? 0 : 255;  // libs/encoder.cpp:507-513
    int rc = svc_hip_morph_rect_host(fg.data(), mfw, mfh, p.morph_rect_w, p.morph_rect_h, 3, tmp.data());  // :524-525 close
    if (rc) return rc;
    if ((rc = svc_hip_morph_rect_host(tmp.data(), mfw, mfh, p.morph_rect_w, p.morph_rect_h, 2, fg.data()))) return rc;  // :526-527 open
    idx.clear();
    for (size_t i = 0; i < n; ++i)
      if (fg[i] == 255) idx.push_back((uint32_t)i);  // :538-546
    if (idx.empty()) continue;  // region ids stay 0 (:549-551)
    const uint32_t k = std::min<uint32_t>(p.cluster_count, (uint32_t)idx.size());  // :555
    feats.assign(idx.size() * 4, 0.0f);  // (0, mv.x, x_px, y_px): :300-321 with libs/math.hpp:285-291
    for (size_t j = 0; j < idx.size(); ++j) {
      feats[4 * j + 1] = mv[(f * n + idx[j]) * 2];
      feats[4 * j + 2] = (float)((idx[j] % mfw) * bw);
      feats[4 * j + 3] = (float)((idx[j] / mfw) * bh);
    }
    labels.resize(idx.size());
    double compactness = 0;
    if ((rc = svc_hip_kmeans_host(feats.data(), (uint32_t)idx.size(), 4, k, p.attempt_count, p.max_iter_count, p.epsilon, seed + f,
                                  labels.data(), &compactness)))  // :575-576
      return rc;
    uint32_t* ty = types.data() + f * n;
    uint32_t offset = 0;
    for (uint32_t cid = 0; cid < k; ++cid) {  // :597-623
      std::fill(one.begin(), one.end(), (uint8_t)0);
      for (size_t j = 0; j < idx.size(); ++j)
        if ((uint32_t)labels[j] == cid) one[idx[j]] = 255;
      uint32_t count = 0;
      if ((rc = svc_hip_connected_components_host(one.data(), mfw, mfh, p.connectivity, cc.data(), &count))) return rc;
      for (size_t j = 0; j < idx.size(); ++j)
        if (cc[idx[j]] != 0) ty[idx[j]] = (uint32_t)cc[idx[j]] + offset;
      offset += count;
    }
  }
  SVC_HIP_TRY(hipMemcpyAsync(d_types, types.data(), types.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  SVC_HIP_TRY(hipStreamSynchronize(stream));
  return SVC_OK;
}

int svc_hip_segment_frames_ex(const uint8_t* d_inlier_mask, const float* d_mv_xy, uint32_t mv_field_w,
                              uint32_t mv_field_h, uint32_t n_frames, uint32_t mv_block_w, uint32_t mv_block_h,
                              svc_segment_params params, uint64_t seed, uint8_t* d_workspace,
                              uint64_t workspace_bytes, uint32_t* d_block_types, uint32_t flags, void* stream) {
  if (n_frames == 0) return SVC_OK;  // empty batch: nothing to enqueue
  SVC_REQUIRE(d_inlier_mask && d_mv_xy && d_block_types && d_workspace, "segment: null pointer");
  SVC_REQUIRE(mv_field_w > 0 && mv_field_h > 0 && mv_block_w > 0 && mv_block_h > 0, "segment: empty motion field");
  // libs/encoder.cpp:39-61 (Validate(KMeansParams)), :92-97 (connectivity)
  SVC_REQUIRE(params.cluster_count > 0, "segment: invalid cluster count: must be > 0");
  SVC_REQUIRE(params.attempt_count > 0, "segment: invalid attempt count: must be > 0");
  SVC_REQUIRE(params.max_iter_count > 0, "segment: invalid maximum iteration count: must be > 0");
  SVC_REQUIRE(params.epsilon > 0, "segment: invalid epsilon: must be > 0");
  SVC_REQUIRE(params.connectivity == 4 || params.connectivity == 8,
              "segment: invalid connected components connectivity: must be either 4 or 8");
  SVC_REQUIRE(params.morph_rect_w > 0 && params.morph_rect_h > 0, "segment: morphology rectangle must be positive");
  if ((params.cluster_count > 64 || params.attempt_count > 16) && params.cluster_count <= 255 && params.attempt_count <= 64)
    return segment_frames_by_calls(d_inlier_mask, d_mv_xy, mv_field_w, mv_field_h, n_frames, mv_block_w, mv_block_h, params, seed,
                                   d_block_types, static_cast<hipStream_t>(stream));
  SVC_REQUIRE(workspace_bytes >= svc_hip_segment_workspace_bytes(mv_field_w, mv_field_h, n_frames, params.attempt_count),
              "segment: workspace of %llu B is smaller than the %llu B needed", (unsigned long long)workspace_bytes,
              (unsigned long long)svc_hip_segment_workspace_bytes(mv_field_w, mv_field_h, n_frames, params.attempt_count));
  SVC_REQUIRE(aligned(d_workspace, 16) && aligned(d_mv_xy, 8), "segment: workspace must be 16-byte, motion field 8-byte aligned");
  return launch_segment(d_inlier_mask, d_mv_xy, mv_field_w, mv_field_h, n_frames, mv_block_w, mv_block_h, params,
                        seed, d_workspace, d_block_types, flags, static_cast<hipStream_t>(stream));
}

static int validate_dct(const void* in, const void* out, uint32_t w, uint32_t h, uint32_t bw, uint32_t bh) {
  SVC_REQUIRE(in && out, "dct: null pointer");
  SVC_REQUIRE(bw > 0 && bh > 0, "dct: block must be positive (encoder.cpp:325-326)");
  SVC_REQUIRE(w > 0 && h > 0 && w % bw == 0 && h % bh == 0, "dct: frame %ux%u not divisible by block %ux%u", w, h, bw, bh);
  return SVC_OK;
}

// The 8x8 / 16x16 kernels load 16-byte vectors and store float2; the general kernel has no such needs.
static int validate_dct_alignment(const uint8_t* d_bgr, uint64_t frame_stride, const float* d_planes, uint32_t w, uint32_t bw,
                                  uint32_t bh) {
  if (bw == bh && (bw == 8 || bw == 16) && w % 16 == 0)
    SVC_REQUIRE(aligned(d_bgr, 16) && frame_stride % 16 == 0 && aligned(d_planes, 8),
                "dct: frames must be 16-byte aligned (stride too), planes 8-byte aligned");
  else
    SVC_REQUIRE(aligned(d_planes, 4), "dct: planes must be 4-byte aligned");
  return SVC_OK;
}

int svc_hip_dct_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t frame_w,
                       uint32_t frame_h, uint32_t block_w, uint32_t block_h, float* d_planes, void* stream) {
  if (n_frames == 0) return SVC_OK;  // empty batch: nothing to enqueue
  int rc = validate_dct(d_bgr, d_planes, frame_w, frame_h, block_w, block_h);
  if (rc) return rc;
  rc = validate_dct_alignment(d_bgr, frame_stride_bytes, d_planes, frame_w, block_w, block_h);
  if (rc) return rc;
  return launch_dct(d_bgr, frame_stride_bytes, n_frames, frame_w, frame_h, block_w, block_h, nullptr, 0, 0, 1, 1,
                    false, d_planes, static_cast<hipStream_t>(stream));
}

int svc_hip_dct_quant_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames,
                             uint32_t frame_w, uint32_t frame_h, uint32_t block_w, uint32_t block_h,
                             const uint32_t* d_block_types, uint32_t mv_block_w, uint32_t mv_block_h,
                             uint32_t fg_step, uint32_t bg_step, float* d_planes, void* stream) {
  if (n_frames == 0) return SVC_OK;  // empty batch: nothing to enqueue
  int rc = validate_dct(d_bgr, d_planes, frame_w, frame_h, block_w, block_h);
  if (rc) return rc;
  SVC_REQUIRE(d_block_types, "dct_quant: null block types");
  SVC_REQUIRE(fg_step > 0 && bg_step > 0, "dct_quant: quant steps must be positive (decoder.cpp:35-47)");
  // transform block <= MV block and divides it (libs/encoder.cpp:62-142 Validate)
  SVC_REQUIRE(mv_block_w > 0 && mv_block_h > 0 && mv_block_w % block_w == 0 && mv_block_h % block_h == 0 &&
                  frame_w % mv_block_w == 0 && frame_h % mv_block_h == 0,
              "dct_quant: MV block %ux%u must be a multiple of the transform block %ux%u and divide the frame",
              mv_block_w, mv_block_h, block_w, block_h);
  rc = validate_dct_alignment(d_bgr, frame_stride_bytes, d_planes, frame_w, block_w, block_h);
  if (rc) return rc;
  return launch_dct(d_bgr, frame_stride_bytes, n_frames, frame_w, frame_h, block_w, block_h, d_block_types,
                    mv_block_w, mv_block_h, fg_step, bg_step, true, d_planes, static_cast<hipStream_t>(stream));
}

int svc_hip_dct_records_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames,
                               uint32_t frame_w, uint32_t frame_h, uint32_t block, const uint32_t* d_block_types,
                               uint32_t mv_block_w, uint32_t mv_block_h, uint32_t fg_step, uint32_t bg_step,
                               uint32_t emit_frame_h, uint8_t* d_records, uint64_t records_stride_bytes,
                               void* stream) {
  if (n_frames == 0) return SVC_OK;  // empty batch: nothing to enqueue
  int rc = validate_dct(d_bgr, d_records, frame_w, frame_h, block, block);
  if (rc) return rc;
  SVC_REQUIRE(d_block_types, "dct_records: null block types");
  SVC_REQUIRE((fg_step == 0) == (bg_step == 0), "dct_records: give both quant steps or neither (0, 0 = no quant)");
  SVC_REQUIRE(mv_block_w > 0 && mv_block_h > 0 && mv_block_w % block == 0 && mv_block_h % block == 0 &&
                  frame_w % mv_block_w == 0 && frame_h % mv_block_h == 0,
              "dct_records: MV block %ux%u must be a multiple of the transform block %u and divide the frame",
              mv_block_w, mv_block_h, block);
  SVC_REQUIRE(emit_frame_h > 0 && emit_frame_h <= frame_h, "dct_records: emit_frame_h %u outside (0, %u]", emit_frame_h, frame_h);
  SVC_REQUIRE(aligned(d_bgr, 16) && frame_stride_bytes % 16 == 0 && aligned(d_records, 4) && records_stride_bytes % 4 == 0 &&
                  records_stride_bytes >= svc_hip_serialized_frame_bytes(frame_w, emit_frame_h, block, block),
              "dct_records: frames must be 16-byte aligned; records 4-byte aligned with a stride of at least one frame");
  const bool quant = fg_step != 0;
  return launch_dct(d_bgr, frame_stride_bytes, n_frames, frame_w, frame_h, block, block, d_block_types, mv_block_w,
                    mv_block_h, quant ? fg_step : 1, quant ? bg_step : 1, quant, nullptr,
                    static_cast<hipStream_t>(stream), d_records, records_stride_bytes, emit_frame_h);
}

int svc_hip_decode_frames(const float* d_planes, uint32_t n_frames, uint32_t frame_w, uint32_t frame_h,
                          uint32_t block, const uint32_t* d_block_types, uint32_t mv_block_w, uint32_t mv_block_h,
                          uint32_t fg_step, uint32_t bg_step, uint32_t gaze_x, uint32_t gaze_y, uint32_t gaze_w,
                          uint32_t gaze_h, float* d_bgr_f32, void* stream) {
  if (n_frames == 0) return SVC_OK;
  SVC_REQUIRE(d_planes && d_block_types && d_bgr_f32, "decode: null pointer");
  SVC_REQUIRE(block > 0 && frame_w > 0 && frame_h > 0 && frame_w % block == 0 && frame_h % block == 0,
              "decode: frame %ux%u not divisible by block %u", frame_w, frame_h, block);
  SVC_REQUIRE(fg_step > 0 && bg_step > 0, "decode: quant steps must be positive (libs/decoder.cpp:35-47)");
  SVC_REQUIRE(mv_block_w > 0 && mv_block_h > 0 && mv_block_w % block == 0 && mv_block_h % block == 0 &&
                  frame_w % mv_block_w == 0 && frame_h % mv_block_h == 0,
              "decode: MV block %ux%u must be a multiple of the transform block %u and divide the frame", mv_block_w, mv_block_h, block);
  SVC_REQUIRE(aligned(d_planes, 16) && aligned(d_bgr_f32, 8), "decode: planes must be 16-byte, output 8-byte aligned");
  return launch_decode(d_planes, n_frames, frame_w, frame_h, block, d_block_types, mv_block_w, mv_block_h, fg_step,
                       bg_step, gaze_x, gaze_y, gaze_w, gaze_h, d_bgr_f32, static_cast<hipStream_t>(stream));
}

int svc_hip_sse_frames(const uint8_t* d_src_bgr, uint64_t src_frame_stride_bytes, const float* d_rec_bgr_f32,
                       uint32_t n_frames, uint32_t frame_w, uint32_t frame_h, uint32_t region_w, uint32_t region_h,
                       uint64_t* d_sse, void* stream) {
  if (n_frames == 0) return SVC_OK;
  SVC_REQUIRE(d_src_bgr && d_rec_bgr_f32 && d_sse, "sse: null pointer");
  SVC_REQUIRE(region_w > 0 && region_h > 0 && region_w <= frame_w && region_h <= frame_h,
              "sse: region %ux%u outside the frame %ux%u", region_w, region_h, frame_w, frame_h);
  SVC_REQUIRE(aligned(d_sse, 8), "sse: output must be 8-byte aligned");
  return launch_sse(d_src_bgr, src_frame_stride_bytes, d_rec_bgr_f32, n_frames, frame_w, frame_h, region_w, region_h,
                    d_sse, static_cast<hipStream_t>(stream));
}

int svc_hip_quant(float* d_coeffs, uint64_t n, uint32_t step, void* stream) {
  SVC_REQUIRE(d_coeffs || n == 0, "quant: null pointer");
  SVC_REQUIRE(step > 0, "quant: step must be positive (decoder.cpp:35-47)");
  return launch_quant(d_coeffs, n, step, static_cast<hipStream_t>(stream));
}

int svc_hip_quant_frames(float* d_planes, uint32_t n_frames, uint32_t frame_w, uint32_t frame_h,
                         uint32_t mv_block_w, uint32_t mv_block_h, const uint32_t* d_block_types,
                         uint32_t fg_step, uint32_t bg_step, void* stream) {
  if (n_frames == 0) return SVC_OK;  // empty batch: nothing to enqueue
  SVC_REQUIRE(d_planes && d_block_types, "quant_frames: null pointer");
  SVC_REQUIRE(fg_step > 0 && bg_step > 0, "quant_frames: steps must be positive");
  SVC_REQUIRE(mv_block_w > 0 && mv_block_h > 0 && frame_w % mv_block_w == 0 && frame_h % mv_block_h == 0,
              "quant_frames: frame %ux%u not divisible by MV block %ux%u", frame_w, frame_h, mv_block_w, mv_block_h);
  return launch_quant_frames(d_planes, n_frames, frame_w, frame_h, mv_block_w, mv_block_h, d_block_types,
                             fg_step, bg_step, static_cast<hipStream_t>(stream));
}

int svc_hip_luma_pyramid_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames,
                                uint32_t frame_w, uint32_t frame_h, uint32_t level_count, uint8_t* d_pyr,
                                uint64_t pyr_stride_bytes, void* stream) {
  if (n_frames == 0) return SVC_OK;  // empty batch: nothing to enqueue
  SVC_REQUIRE(d_bgr && d_pyr, "luma_pyramid: null pointer");
  SVC_REQUIRE(level_count > 0 && level_count <= 16, "luma_pyramid: level_count %u out of range", level_count);
  const uint32_t f = 1u << (level_count - 1);
  SVC_REQUIRE(frame_w > 0 && frame_h > 0 && frame_w % f == 0 && frame_h % f == 0,
              "luma_pyramid: frame %ux%u must be divisible by 2^(levels-1) = %u", frame_w, frame_h, f);
  SVC_REQUIRE(aligned(d_bgr, 16) && frame_stride_bytes % 16 == 0 && aligned(d_pyr, 16) && pyr_stride_bytes % 16 == 0,
              "luma_pyramid: frames and pyramids must be 16-byte aligned (strides too)");
  SVC_REQUIRE(n_frames <= 1 || pyr_stride_bytes >= pyramid_bytes(frame_w, frame_h, level_count),
              "luma_pyramid: pyramid stride too small");
  return launch_luma_pyramid(d_bgr, frame_stride_bytes, n_frames, frame_w, frame_h, level_count, d_pyr,
                             pyr_stride_bytes, static_cast<hipStream_t>(stream));
}

int svc_hip_pyramid_levels_frames(uint8_t* d_pyr, uint64_t pyr_stride_bytes, uint32_t n_frames, uint32_t frame_w, uint32_t frame_h,
                                  uint32_t level_count, void* stream) {
  if (n_frames == 0) return SVC_OK;
  SVC_REQUIRE(d_pyr, "pyramid_levels: null pointer");
  SVC_REQUIRE(level_count > 0 && level_count <= 16, "pyramid_levels: level_count %u out of range", level_count);
  const uint32_t f = 1u << (level_count - 1);
  SVC_REQUIRE(frame_w > 0 && frame_h > 0 && frame_w % f == 0 && frame_h % f == 0,
              "pyramid_levels: frame %ux%u must be divisible by 2^(levels-1) = %u", frame_w, frame_h, f);
  SVC_REQUIRE(n_frames <= 1 || pyr_stride_bytes >= pyramid_bytes(frame_w, frame_h, level_count), "pyramid_levels: pyramid stride too small");
  return launch_pyr_down_levels(d_pyr, pyr_stride_bytes, n_frames, frame_w, frame_h, level_count, 0, static_cast<hipStream_t>(stream));
}

int svc_hip_dct_records_luma_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t frame_w,
                                    uint32_t frame_h, uint32_t block, uint32_t emit_frame_h, uint8_t* d_records,
                                    uint64_t records_stride_bytes, uint8_t* d_pyr, uint64_t pyr_stride_bytes, void* stream) {
  if (n_frames == 0) return SVC_OK;
  int rc = validate_dct(d_bgr, d_records, frame_w, frame_h, block, block);
  if (rc) return rc;
  SVC_REQUIRE(d_pyr, "dct_records_luma: null pyramid");
  SVC_REQUIRE(emit_frame_h > 0 && emit_frame_h <= frame_h, "dct_records_luma: emit_frame_h %u outside (0, %u]", emit_frame_h, frame_h);
  SVC_REQUIRE(aligned(d_bgr, 16) && frame_stride_bytes % 16 == 0 && aligned(d_records, 4) && records_stride_bytes % 4 == 0 &&
                  records_stride_bytes >= svc_hip_serialized_frame_bytes(frame_w, emit_frame_h, block, block),
              "dct_records_luma: frames must be 16-byte aligned; records 4-byte aligned with a stride of at least one frame");
  SVC_REQUIRE(aligned(d_pyr, 16) && pyr_stride_bytes % 16 == 0 && (n_frames <= 1 || pyr_stride_bytes >= (uint64_t)frame_w * frame_h),
              "dct_records_luma: pyramids must be 16-byte aligned (stride too), a stride of at least one luma plane");
  return launch_dct(d_bgr, frame_stride_bytes, n_frames, frame_w, frame_h, block, block, nullptr, block, block, 1, 1, false, nullptr,
                    static_cast<hipStream_t>(stream), d_records, records_stride_bytes, emit_frame_h, d_pyr, pyr_stride_bytes);
}

int svc_hip_dct_quant_luma_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t frame_w,
                                  uint32_t frame_h, uint32_t block, uint32_t bg_step, float* d_planes, uint8_t* d_pyr,
                                  uint64_t pyr_stride_bytes, void* stream) {
  if (n_frames == 0) return SVC_OK;
  int rc = validate_dct(d_bgr, d_planes, frame_w, frame_h, block, block);
  if (rc) return rc;
  SVC_REQUIRE(d_pyr, "dct_quant_luma: null pyramid");
  SVC_REQUIRE(bg_step > 0, "dct_quant_luma: quant step must be positive (libs/decoder.cpp:35-47)");
  if ((rc = validate_dct_alignment(d_bgr, frame_stride_bytes, d_planes, frame_w, block, block))) return rc;
  SVC_REQUIRE(aligned(d_pyr, 16) && pyr_stride_bytes % 16 == 0 && (n_frames <= 1 || pyr_stride_bytes >= (uint64_t)frame_w * frame_h),
              "dct_quant_luma: pyramids must be 16-byte aligned (stride too), a stride of at least one luma plane");
  return launch_dct_quant_speculative(d_bgr, frame_stride_bytes, n_frames, frame_w, frame_h, block, bg_step, d_planes, d_pyr,
                                      pyr_stride_bytes, static_cast<hipStream_t>(stream));
}

int svc_hip_count_foreground(const uint32_t* d_block_types, uint64_t n, uint32_t* d_count, void* stream) {
  SVC_REQUIRE(d_count && (d_block_types || n == 0), "count_foreground: null pointer");
  return launch_count_foreground(d_block_types, n, d_count, static_cast<hipStream_t>(stream));
}

uint64_t svc_hip_dct_redo_workspace_bytes(uint32_t n_frames, uint32_t frame_w, uint32_t frame_h, uint32_t mv_block_w, uint32_t mv_block_h) {
  if (!mv_block_w || !mv_block_h) return 0;
  return dct_redo_workspace_bytes(n_frames, (frame_w / mv_block_w) * (frame_h / mv_block_h));
}

int svc_hip_dct_quant_redo_frames(const uint8_t* d_bgr, uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t frame_w, uint32_t frame_h,
                                  uint32_t block, const uint32_t* d_block_types, uint32_t mv_block_w, uint32_t mv_block_h,
                                  uint32_t fg_step, float* d_planes, uint8_t* d_ws, uint64_t ws_bytes, void* stream) {
  if (n_frames == 0) return SVC_OK;
  int rc = validate_dct(d_bgr, d_planes, frame_w, frame_h, block, block);
  if (rc) return rc;
  SVC_REQUIRE(d_block_types && d_ws, "dct_quant_redo: null pointer");
  SVC_REQUIRE(fg_step > 0, "dct_quant_redo: quant step must be positive (libs/decoder.cpp:35-47)");
  SVC_REQUIRE(mv_block_w > 0 && mv_block_h > 0 && mv_block_w % block == 0 && mv_block_h % block == 0 &&
                  frame_w % mv_block_w == 0 && frame_h % mv_block_h == 0,
              "dct_quant_redo: MV block %ux%u must be a multiple of the transform block %u and divide the frame", mv_block_w, mv_block_h, block);
  if ((rc = validate_dct_alignment(d_bgr, frame_stride_bytes, d_planes, frame_w, block, block))) return rc;
  SVC_REQUIRE(aligned(d_ws, 16) && ws_bytes >= svc_hip_dct_redo_workspace_bytes(n_frames, frame_w, frame_h, mv_block_w, mv_block_h),
              "dct_quant_redo: workspace must be 16-byte aligned and hold svc_hip_dct_redo_workspace_bytes()");
  return launch_dct_quant_redo_foreground(d_bgr, frame_stride_bytes, n_frames, frame_w, frame_h, block, d_block_types, mv_block_w, mv_block_h,
                                          fg_step, d_planes, d_ws, static_cast<hipStream_t>(stream));
}

int svc_hip_wire_patch_types_frames(const uint32_t* d_block_types, uint32_t n_frames, uint32_t frame_w, uint32_t frame_h,
                                    uint32_t emit_frame_h, uint32_t block, uint32_t mv_block_w, uint32_t mv_block_h, uint8_t* d_records,
                                    uint64_t records_stride_bytes, int all_tiles, void* stream) {
  if (n_frames == 0) return SVC_OK;
  SVC_REQUIRE(d_block_types && d_records, "wire_patch_types: null pointer");
  SVC_REQUIRE(block > 0 && frame_w > 0 && frame_h > 0 && frame_w % block == 0 && frame_h % block == 0 && block <= 64,
              "wire_patch_types: frame %ux%u not divisible by block %u", frame_w, frame_h, block);
  SVC_REQUIRE(mv_block_w > 0 && mv_block_h > 0 && mv_block_w % block == 0 && mv_block_h % block == 0 &&
                  frame_w % mv_block_w == 0 && frame_h % mv_block_h == 0,
              "wire_patch_types: MV block %ux%u must be a multiple of the transform block %u and divide the frame", mv_block_w, mv_block_h, block);
  SVC_REQUIRE(emit_frame_h > 0 && emit_frame_h <= frame_h, "wire_patch_types: emit_frame_h %u outside (0, %u]", emit_frame_h, frame_h);
  SVC_REQUIRE(aligned(d_records, 4) && records_stride_bytes % 4 == 0 &&
                  records_stride_bytes >= svc_hip_serialized_frame_bytes(frame_w, emit_frame_h, block, block),
              "wire_patch_types: records 4-byte aligned with a stride of at least one frame");
  return launch_wire_patch_types(d_block_types, n_frames, frame_w, frame_h, emit_frame_h, block, mv_block_w, mv_block_h, d_records,
                                 records_stride_bytes, all_tiles != 0, static_cast<hipStream_t>(stream));
}

// ---- host-pointer forms -----------------------------------------------------------

// ---- whole-frame global motion (libs/motion.hpp:38-59), see global_motion.hip ---------------------
static int validate_global_ebma(uint32_t w, uint32_t h, uint32_t range) {
  SVC_REQUIRE(w > 0 && h > 0, "global ebma: frame %ux%u must be positive", w, h);
  // motion.cpp:63-64 asserts range <= frame; at range == frame the overlap is empty (0 / 0 in Mad)
  SVC_REQUIRE(range < w && range < h, "global ebma: search range %u must be smaller than the frame %ux%u (motion.cpp:63-64)", range, w, h);
  SVC_REQUIRE(range <= 1024, "global ebma: search range %u out of range", range);
  return SVC_OK;
}

uint64_t svc_hip_global_ebma_workspace_bytes(uint32_t search_range, uint32_t n_pairs) {
  return global_ebma_workspace_bytes(search_range, n_pairs);
}

int svc_hip_global_ebma_pairs(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride_bytes, uint32_t n_pairs,
                              uint32_t frame_w, uint32_t frame_h, uint32_t search_range, uint8_t* d_workspace,
                              uint64_t workspace_bytes, float* d_gm_xy, float* d_min_mad, void* stream) {
  if (n_pairs == 0) return SVC_OK;
  SVC_REQUIRE(d_tracked && d_anchor && d_gm_xy && d_workspace, "global ebma: null pointer (motion.cpp:58-61)");
  int rc = validate_global_ebma(frame_w, frame_h, search_range);
  if (rc) return rc;
  SVC_REQUIRE(workspace_bytes >= global_ebma_workspace_bytes(search_range, n_pairs) && aligned(d_workspace, 8),
              "global ebma: workspace too small or not 8-byte aligned");
  return launch_global_ebma(d_tracked, d_anchor, pair_stride_bytes, n_pairs, frame_w, frame_h, search_range, d_workspace,
                            d_gm_xy, d_min_mad, false, static_cast<hipStream_t>(stream));
}

int svc_hip_global_avg_frames(const float* d_mv_xy, uint32_t blocks, uint32_t n_frames, float* d_avg_xy, void* stream) {
  if (n_frames == 0) return SVC_OK;
  SVC_REQUIRE(d_mv_xy && d_avg_xy, "global avg: null pointer (motion.cpp:46)");
  return launch_global_avg(d_mv_xy, blocks, n_frames, d_avg_xy, static_cast<hipStream_t>(stream));
}

int svc_hip_hbma_host(const uint8_t* const* tracked_pyr, const uint8_t* const* anchor_pyr, uint32_t level_count,
                      uint32_t frame_w, uint32_t frame_h, uint32_t search_range, uint32_t block_w,
                      uint32_t block_h, float* mv_xy, float* min_mad, uint32_t flags) {
  SVC_REQUIRE(tracked_pyr && anchor_pyr && mv_xy && min_mad, "hbma: null pointer (motion.cpp:417-420)");
  int rc = validate_hbma(level_count, frame_w, frame_h, search_range, block_w, block_h);
  if (rc) return rc;
  for (uint32_t l = 0; l < level_count; ++l)
    SVC_REQUIRE(tracked_pyr[l] && anchor_pyr[l], "hbma: null plane at level %u", l);
  if ((rc = require_device())) return rc;
  const size_t pyr = up256(pyramid_bytes(frame_w, frame_h, level_count));
  const size_t blocks = (size_t)(frame_w / block_w) * (frame_h / block_h);
  const size_t out_off = 2 * pyr, out_bytes = blocks * 12;
  if ((rc = g_stage.ensure(out_off + up256(out_bytes)))) return rc;
  size_t o = 0;
  for (uint32_t l = 0; l < level_count; ++l) {
    const size_t n = (size_t)(frame_w >> l) * (frame_h >> l);
    host_copy(g_stage.pin + o, tracked_pyr[l], n);
    host_copy(g_stage.pin + pyr + o, anchor_pyr[l], n);
    o += n;
  }
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.dev, g_stage.pin, 2 * pyr, hipMemcpyHostToDevice, g_stage.stream));
  float* d_mv = reinterpret_cast<float*>(g_stage.dev + out_off);
  float* d_mad = d_mv + 2 * blocks;
  rc = launch_hbma(g_stage.dev, g_stage.dev + pyr, pyr, 1, level_count, frame_w, frame_h, search_range, block_w,
                   block_h, d_mv, d_mad, flags, g_stage.stream);
  if (rc) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.pin + out_off, d_mv, out_bytes, hipMemcpyDeviceToHost, g_stage.stream));
  SVC_HIP_TRY(hipStreamSynchronize(g_stage.stream));
  std::memcpy(mv_xy, g_stage.pin + out_off, blocks * 8);
  std::memcpy(min_mad, g_stage.pin + out_off + blocks * 8, blocks * 4);
  return SVC_OK;
}

int svc_hip_ebma_host(const uint8_t* tracked, const uint8_t* anchor, uint32_t frame_w, uint32_t frame_h,
                      uint32_t search_range, uint32_t block_w, uint32_t block_h, float* mv_xy, float* min_mad) {
  SVC_REQUIRE(tracked && anchor && mv_xy && min_mad, "ebma: null pointer (motion.cpp:273-276)");
  SVC_REQUIRE(block_w > 0 && block_h > 0, "ebma: block must be positive (motion.cpp:278-279)");
  SVC_REQUIRE(frame_w > 0 && frame_h > 0 && frame_w % block_w == 0 && frame_h % block_h == 0,
              "ebma: frame %ux%u not divisible by block %ux%u (motion.cpp:281-282)", frame_w, frame_h, block_w, block_h);
  int rc = require_device();
  if (rc) return rc;
  const size_t plane = up256((size_t)frame_w * frame_h);
  const size_t blocks = (size_t)(frame_w / block_w) * (frame_h / block_h);
  const size_t out_off = 2 * plane, out_bytes = blocks * 12;
  if ((rc = g_stage.ensure(out_off + up256(out_bytes)))) return rc;
  host_copy(g_stage.pin, tracked, (size_t)frame_w * frame_h);
  host_copy(g_stage.pin + plane, anchor, (size_t)frame_w * frame_h);
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.dev, g_stage.pin, 2 * plane, hipMemcpyHostToDevice, g_stage.stream));
  float* d_mv = reinterpret_cast<float*>(g_stage.dev + out_off);
  float* d_mad = d_mv + 2 * blocks;
  rc = launch_ebma(g_stage.dev, g_stage.dev + plane, plane, 1, frame_w, frame_h, search_range, block_w, block_h,
                   d_mv, d_mad, g_stage.stream);
  if (rc) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.pin + out_off, d_mv, out_bytes, hipMemcpyDeviceToHost, g_stage.stream));
  SVC_HIP_TRY(hipStreamSynchronize(g_stage.stream));
  std::memcpy(mv_xy, g_stage.pin + out_off, blocks * 8);
  std::memcpy(min_mad, g_stage.pin + out_off + blocks * 8, blocks * 4);
  return SVC_OK;
}

int svc_hip_ransac_host(const float* mv_xy, uint32_t blocks, svc_ransac_params params, const uint32_t* samples,
                        uint32_t iter_count, float* gm_xy, float* rmse, uint32_t* inlier_indices,
                        uint32_t* inlier_count) {
  SVC_REQUIRE(mv_xy && gm_xy && rmse && inlier_indices && inlier_count, "ransac: null pointer (motion.cpp:189-192)");
  SVC_REQUIRE(iter_count == 0 || samples, "ransac: null samples");
  SVC_REQUIRE(params.subset_sz > 0 && blocks >= params.subset_sz,
              "ransac: motion field of %u smaller than subset %u (motion.cpp:194)", blocks, params.subset_sz);
  for (size_t i = 0; i < (size_t)iter_count * params.subset_sz; ++i)
    SVC_REQUIRE(samples[i] < blocks, "ransac: sample index %u out of range [0, %u)", samples[i], blocks);
  int rc = require_device();
  if (rc) return rc;
  const size_t mv_b = up256((size_t)blocks * 8), smp_b = up256((size_t)iter_count * params.subset_sz * 4 + 4);
  const size_t mask_b = up256(blocks), misc_b = 256;
  if ((rc = g_stage.ensure(mv_b + smp_b + mask_b + misc_b))) return rc;
  uint8_t* p = g_stage.pin;
  std::memcpy(p, mv_xy, (size_t)blocks * 8);
  std::memcpy(p + mv_b, samples, (size_t)iter_count * params.subset_sz * 4);
  float* misc = reinterpret_cast<float*>(p + mv_b + smp_b + mask_b);
  misc[0] = gm_xy[0];
  misc[1] = gm_xy[1];
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.dev, p, mv_b + smp_b + mask_b + misc_b, hipMemcpyHostToDevice, g_stage.stream));
  uint8_t* d = g_stage.dev;
  float* d_misc = reinterpret_cast<float*>(d + mv_b + smp_b + mask_b);
  rc = launch_ransac(reinterpret_cast<const float*>(d), blocks, 1, params, reinterpret_cast<const uint32_t*>(d + mv_b),
                     iter_count, d_misc, d_misc + 2, d + mv_b + smp_b, reinterpret_cast<uint32_t*>(d_misc + 3), 0,
                     g_stage.stream);
  if (rc) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(p + mv_b + smp_b, d + mv_b + smp_b, mask_b + misc_b, hipMemcpyDeviceToHost, g_stage.stream));
  SVC_HIP_TRY(hipStreamSynchronize(g_stage.stream));
  gm_xy[0] = misc[0];
  gm_xy[1] = misc[1];
  *rmse = misc[2];
  const uint8_t* mask = p + mv_b + smp_b;
  uint32_t k = 0;
  for (uint32_t i = 0; i < blocks; ++i)
    if (mask[i]) inlier_indices[k++] = i;  // ascending, as motion.cpp:244-253 collects them
  *inlier_count = k;
  return SVC_OK;
}

static int dct_host_common(const uint8_t* bgr, uint32_t w, uint32_t h, uint32_t bw, uint32_t bh,
                           const uint32_t* types, uint32_t mv_bw, uint32_t mv_bh, uint32_t fg, uint32_t bg,
                           bool quant, float* planes, float* const* planes3 = nullptr) {
  int rc = validate_dct(bgr, planes3 ? planes3[0] : planes, w, h, bw, bh);
  if (rc) return rc;
  if ((rc = require_device())) return rc;
  const size_t in_b = up256((size_t)w * h * 3), out_b = (size_t)w * h * 12;
  const size_t nt = quant ? (size_t)(w / mv_bw) * (h / mv_bh) : 0, t_b = up256(nt * 4);
  if ((rc = g_stage.ensure(in_b + t_b + up256(out_b)))) return rc;
  host_copy(g_stage.pin, bgr, (size_t)w * h * 3);
  if (quant) std::memcpy(g_stage.pin + in_b, types, nt * 4);
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.dev, g_stage.pin, in_b + t_b, hipMemcpyHostToDevice, g_stage.stream));
  float* d_out = reinterpret_cast<float*>(g_stage.dev + in_b + t_b);
  rc = launch_dct(g_stage.dev, in_b, 1, w, h, bw, bh, reinterpret_cast<const uint32_t*>(g_stage.dev + in_b), mv_bw,
                  mv_bh, fg, bg, quant, d_out, g_stage.stream);
  if (rc) return rc;
  // 25 MB per 1080p frame back: in pieces, each copied out to the caller while the next one crosses the link
  uint8_t* out3[3];
  for (int c = 0; c < 3; ++c) out3[c] = reinterpret_cast<uint8_t*>(planes3 ? planes3[c] : planes + (size_t)c * w * h);
  return g_stage.download(reinterpret_cast<const uint8_t*>(d_out), in_b + t_b, out_b, out3, 3);
}

int svc_hip_dct_planes_host(const uint8_t* bgr, uint32_t frame_w, uint32_t frame_h, uint32_t block_w, uint32_t block_h,
                            float* const planes[3]) {
  SVC_REQUIRE(planes && planes[0] && planes[1] && planes[2], "dct: null plane pointer");
  return dct_host_common(bgr, frame_w, frame_h, block_w, block_h, nullptr, 0, 0, 1, 1, false, nullptr, planes);
}

int svc_hip_dct_host(const uint8_t* bgr, uint32_t frame_w, uint32_t frame_h, uint32_t block_w, uint32_t block_h,
                     float* planes) {
  return dct_host_common(bgr, frame_w, frame_h, block_w, block_h, nullptr, 0, 0, 1, 1, false, planes);
}

int svc_hip_dct_quant_host(const uint8_t* bgr, uint32_t frame_w, uint32_t frame_h, uint32_t block_w,
                           uint32_t block_h, const uint32_t* block_types, uint32_t mv_block_w,
                           uint32_t mv_block_h, uint32_t fg_step, uint32_t bg_step, float* planes) {
  SVC_REQUIRE(block_types, "dct_quant: null block types");
  SVC_REQUIRE(fg_step > 0 && bg_step > 0, "dct_quant: quant steps must be positive");
  SVC_REQUIRE(block_w > 0 && block_h > 0 && mv_block_w > 0 && mv_block_h > 0 && mv_block_w % block_w == 0 &&
                  mv_block_h % block_h == 0 && frame_w % mv_block_w == 0 && frame_h % mv_block_h == 0,
              "dct_quant: MV block %ux%u must be a multiple of the transform block %ux%u and divide the frame",
              mv_block_w, mv_block_h, block_w, block_h);
  return dct_host_common(bgr, frame_w, frame_h, block_w, block_h, block_types, mv_block_w, mv_block_h, fg_step,
                         bg_step, true, planes);
}

int svc_hip_quant_host(float* coeffs, uint64_t n, uint32_t step) {
  SVC_REQUIRE(coeffs || n == 0, "quant: null pointer");
  SVC_REQUIRE(step > 0, "quant: step must be positive");
  if (n == 0) return SVC_OK;
  int rc = require_device();
  if (rc) return rc;
  if ((rc = g_stage.ensure(n * 4))) return rc;
  host_copy(g_stage.pin, coeffs, n * 4);
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.dev, g_stage.pin, n * 4, hipMemcpyHostToDevice, g_stage.stream));
  rc = launch_quant(reinterpret_cast<float*>(g_stage.dev), n, step, g_stage.stream);
  if (rc) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.pin, g_stage.dev, n * 4, hipMemcpyDeviceToHost, g_stage.stream));
  SVC_HIP_TRY(hipStreamSynchronize(g_stage.stream));
  host_copy(coeffs, g_stage.pin, n * 4);
  return SVC_OK;
}

int svc_hip_global_ebma_host(const uint8_t* tracked, const uint8_t* anchor, uint32_t frame_w, uint32_t frame_h,
                             uint32_t search_range, float* gm_xy, float* min_mad) {
  SVC_REQUIRE(tracked && anchor && gm_xy && min_mad, "global ebma: null pointer (motion.cpp:58-61)");
  int rc = validate_global_ebma(frame_w, frame_h, search_range);
  if (rc) return rc;
  if ((rc = require_device())) return rc;
  const size_t plane = up256((size_t)frame_w * frame_h), ws = up256(global_ebma_workspace_bytes(search_range, 1));
  if ((rc = g_stage.ensure(2 * plane + ws + 256))) return rc;
  host_copy(g_stage.pin, tracked, (size_t)frame_w * frame_h);
  host_copy(g_stage.pin + plane, anchor, (size_t)frame_w * frame_h);
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.dev, g_stage.pin, 2 * plane, hipMemcpyHostToDevice, g_stage.stream));
  float* d_out = reinterpret_cast<float*>(g_stage.dev + 2 * plane + ws);
  rc = launch_global_ebma(g_stage.dev, g_stage.dev + plane, plane, 1, frame_w, frame_h, search_range, g_stage.dev + 2 * plane,
                          d_out, d_out + 2, false, g_stage.stream);
  if (rc) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.pin, d_out, 12, hipMemcpyDeviceToHost, g_stage.stream));
  SVC_HIP_TRY(hipStreamSynchronize(g_stage.stream));
  std::memcpy(gm_xy, g_stage.pin, 8);
  std::memcpy(min_mad, g_stage.pin + 8, 4);
  return SVC_OK;
}

// motion.cpp:101-142: exhaustive search on the top level with range / 2^(L-1), then on every finer level a +-1 search
// around ZERO displacement whose result is added to twice the running estimate (:132-140) -- as written there.
int svc_hip_global_hbma_host(const uint8_t* const* tracked_pyr, const uint8_t* const* anchor_pyr, uint32_t level_count,
                             uint32_t frame_w, uint32_t frame_h, uint32_t search_range, float* gm_xy) {
  SVC_REQUIRE(tracked_pyr && anchor_pyr && gm_xy, "global hbma: null pointer (motion.cpp:106-108)");
  SVC_REQUIRE(level_count > 0 && level_count <= 16, "global hbma: level count %u (motion.cpp:110)", level_count);
  const uint32_t f = 1u << (level_count - 1);
  SVC_REQUIRE(frame_w % f == 0 && frame_h % f == 0 && frame_w / f > 0 && frame_h / f > 0,
              "global hbma: frame %ux%u must be divisible by 2^(levels-1) = %u", frame_w, frame_h, f);
  for (uint32_t l = 0; l < level_count; ++l)
    SVC_REQUIRE(tracked_pyr[l] && anchor_pyr[l], "global hbma: null plane at level %u", l);
  const uint32_t top_range = search_range / f;  // :125
  int rc = validate_global_ebma(frame_w / f, frame_h / f, top_range);
  if (rc) return rc;
  if (level_count > 1 && (rc = validate_global_ebma(frame_w >> (level_count - 2), frame_h >> (level_count - 2), 1))) return rc;
  if ((rc = require_device())) return rc;
  const size_t pyr = up256(pyramid_bytes(frame_w, frame_h, level_count));
  const size_t ws = up256(global_ebma_workspace_bytes(top_range > 1 ? top_range : 1, 1));
  if ((rc = g_stage.ensure(2 * pyr + ws + 256))) return rc;
  size_t o = 0;
  for (uint32_t l = 0; l < level_count; ++l) {
    const size_t n = (size_t)(frame_w >> l) * (frame_h >> l);
    host_copy(g_stage.pin + o, tracked_pyr[l], n);
    host_copy(g_stage.pin + pyr + o, anchor_pyr[l], n);
    o += n;
  }
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.dev, g_stage.pin, 2 * pyr, hipMemcpyHostToDevice, g_stage.stream));
  float* d_gm = reinterpret_cast<float*>(g_stage.dev + 2 * pyr + ws);
  for (int l = (int)level_count - 1; l >= 0; --l) {
    o -= (size_t)(frame_w >> l) * (frame_h >> l);  // offset of level l inside a packed pyramid
    const bool top = l == (int)level_count - 1;
    rc = launch_global_ebma(g_stage.dev + o, g_stage.dev + pyr + o, 0, 1, frame_w >> l, frame_h >> l, top ? top_range : 1,
                            g_stage.dev + 2 * pyr, d_gm, nullptr, !top, g_stage.stream);
    if (rc) return rc;
  }
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.pin, d_gm, 8, hipMemcpyDeviceToHost, g_stage.stream));
  SVC_HIP_TRY(hipStreamSynchronize(g_stage.stream));
  std::memcpy(gm_xy, g_stage.pin, 8);
  return SVC_OK;
}

int svc_hip_global_avg_host(const float* mv_xy, uint32_t blocks, float* avg_xy) {
  SVC_REQUIRE(mv_xy && avg_xy, "global avg: null pointer (motion.cpp:46)");
  int rc = require_device();
  if (rc) return rc;
  const size_t in_b = up256((size_t)blocks * 8);
  if ((rc = g_stage.ensure(in_b + 256))) return rc;
  std::memcpy(g_stage.pin, mv_xy, (size_t)blocks * 8);
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.dev, g_stage.pin, in_b, hipMemcpyHostToDevice, g_stage.stream));
  float* d_out = reinterpret_cast<float*>(g_stage.dev + in_b);
  rc = launch_global_avg(reinterpret_cast<const float*>(g_stage.dev), blocks, 1, d_out, g_stage.stream);
  if (rc) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(g_stage.pin, d_out, 8, hipMemcpyDeviceToHost, g_stage.stream));
  SVC_HIP_TRY(hipStreamSynchronize(g_stage.stream));
  std::memcpy(avg_xy, g_stage.pin, 8);
  return SVC_OK;
}

}  // extern "C"
