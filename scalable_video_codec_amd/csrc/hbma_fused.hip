// hbma_fused.hip -- EstimateMotionHierarchical as ONE launch: every pyramid level
// of one 16x16 MV block is searched by one lane.
//
// Why this shape.  A block's MV at level l depends only on the SAME block's MV at
// level l+1 (reference libs/motion.cpp:451-464 walks the levels, but never reads a
// neighbour's vector), so the whole coarse-to-fine chain is private to a block: no
// inter-level grid sync, no MV round trip through HBM, one launch per batch of
// frame pairs.  With <= 25 candidates per level (R_top = R / 2^(L-1) in {1, 2}) a
// wave-per-block search would idle most lanes; a lane-per-block search keeps all 64
// busy and makes the anchor loads of a wave one contiguous run per row (64
// horizontally adjacent blocks x 16 B = 1 KiB), while the tracked-window rows of
// neighbouring lanes overlap in the same L1 lines.
//
// The SAD engine, the window clamps and the selection rules live in hbma_search.hpp.
#include "hbma_search.hpp"

namespace svc {

template <int L, int RT>
__global__ __launch_bounds__(256) void hbma_fused16_kernel(FusedArgs a) {
  // Region-major order.  Workgroups are dealt round-robin over the 8 XCDs, so XCD x gets blockIdx 8k + x: it is
  // given the x-th eighth of the frame (a band of block rows) of EVERY pair, pairs in order.  The pyramid of frame
  // p + 1 is the anchor of pair p and the tracked frame of pair p + 1: the two workgroups that read a band of it are
  // neighbours in one XCD's dispatch sequence, so the second read is served by that XCD's L2 instead of crossing the
  // fabric again.  Speed only: any placement gives the same result.  (The pair-major order of round 1 was measured
  // against it and dropped: profiles/r02_ab_hbma_order.txt.)
  const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
  const uint32_t pair = k / a.wgs_per_region;
  const uint32_t blk = (xcd * a.wgs_per_region + (k - pair * a.wgs_per_region)) * 256u + threadIdx.x;
  if (pair >= a.n_pairs || blk >= a.blocks) return;
  const uint32_t item = pair * a.blocks + blk;
  const int by = (int)(blk / a.mfw), bx = (int)(blk - (uint32_t)by * a.mfw);

  const uint8_t* trk = a.tracked + (size_t)pair * a.pair_stride;
  const uint8_t* anc = a.anchor + (size_t)pair * a.pair_stride;
  const int w = (int)a.w, h = (int)a.h;
  const size_t o1 = (size_t)w * h, o2 = o1 + (o1 >> 2), o3 = o2 + (o1 >> 4);

  int mvx = 0, mvy = 0;
  uint32_t best = 0;
  if (L == 4) {
    TopB2<RT> top;
    load_top_b2<RT>(trk + o3, anc + o3, w >> 3, h >> 3, bx, by, top);
    search_top_b2<RT, 6>(top, bx, by, mvx, mvy, best);
    mvx *= 2; mvy *= 2;  // motion.cpp:458-460
    search_level<4, RT, false, 4>(trk + o2, anc + o2, w >> 2, h >> 2, bx, by, mvx, mvy, best);
  } else {
    search_level<4, RT, true, 4>(trk + o2, anc + o2, w >> 2, h >> 2, bx, by, mvx, mvy, best);
  }
  mvx *= 2; mvy *= 2;
  search_level<8, RT, false, 2>(trk + o1, anc + o1, w >> 1, h >> 1, bx, by, mvx, mvy, best);
  mvx *= 2; mvy *= 2;
  search_level<16, RT, false, 0>(trk, anc, w, h, bx, by, mvx, mvy, best);

  reinterpret_cast<float2*>(a.mv)[item] = make_float2((float)mvx, (float)mvy);
  a.mad[item] = (float)best * (1.0f / 256.0f);  // exact: best < 2^24, power-of-two scale
}

bool fused_supported(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw,
                     uint32_t bh) {
  if (bw != 16 || bh != 16 || (levels != 3 && levels != 4)) return false;
  const uint32_t rt = range >> (levels - 1);
  if (rt != 1 && rt != 2) return false;
  const uint32_t tw = w >> (levels - 1), th = h >> (levels - 1), tb = 16u >> (levels - 1);
  // the top plane must hold a whole candidate grid, and its rows must be dword-aligned: the
  // clamped top-level loads (load_row<.., true>) only leave needed bytes alone when the row
  // width is a multiple of 4 (found by tests/test_gpu_hbma_property.py: 112 x 32, 4 levels)
  return tw >= tb + 8 && tw >= 12 && tw % 4 == 0 && th >= tb + 2 * rt && (w % 16 == 0) && (h % 16 == 0);
}

int launch_hbma_tiled(const FusedArgs& a, uint32_t n_pairs, hipStream_t stream);

// kernel: 0 = the shape's default, 1 = lane-per-block (no LDS), 2 = LDS-tiled (UNSUPPORTED where tiled_supported is false)
int launch_hbma_fused(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride,
                      uint32_t n_pairs, uint32_t levels, uint32_t w, uint32_t h, uint32_t range,
                      float* d_mv, float* d_mad, int kernel, hipStream_t stream) {
  FusedArgs a;
  a.tracked = d_tracked;
  a.anchor = d_anchor;
  a.pair_stride = pair_stride;
  a.mfw = w / 16;
  a.blocks = a.mfw * (h / 16);
  const uint64_t items = (uint64_t)a.blocks * n_pairs;
  if (items == 0) return SVC_OK;
  if (items > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "hbma: %llu work items exceed one launch", (unsigned long long)items);
  a.n_items = (uint32_t)items;
  a.w = w; a.h = h;
  a.mv = d_mv;
  a.mad = d_mad;
  a.n_pairs = n_pairs;
  const bool can_tile = tiled_supported(levels, w, h, range, 16, 16) && ((uintptr_t)d_tracked % 16 == 0) &&
                        ((uintptr_t)d_anchor % 16 == 0) && pair_stride % 16 == 0;
  if (kernel == 2 && !can_tile)
    return fail(SVC_ERR_UNSUPPORTED, "hbma: the LDS-tiled kernel needs 4 levels, r_top 1, a frame width that is a multiple of 64 and 16-byte aligned pyramids");
  if (kernel == 2 || (kernel == 0 && can_tile && kTiledIsDefault)) return launch_hbma_tiled(a, n_pairs, stream);
  a.wgs_per_region = div_up(div_up(a.blocks, 256), 8);
  const uint64_t wgs = (uint64_t)8 * a.wgs_per_region * n_pairs;
  if (wgs > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "hbma: %llu workgroups exceed one launch", (unsigned long long)wgs);
  const dim3 grid((uint32_t)wgs), block(256);
  const uint32_t rt = range >> (levels - 1);
  if (levels == 3 && rt == 2) hipLaunchKernelGGL((hbma_fused16_kernel<3, 2>), grid, block, 0, stream, a);
  else if (levels == 3 && rt == 1) hipLaunchKernelGGL((hbma_fused16_kernel<3, 1>), grid, block, 0, stream, a);
  else if (levels == 4 && rt == 1) hipLaunchKernelGGL((hbma_fused16_kernel<4, 1>), grid, block, 0, stream, a);
  else if (levels == 4 && rt == 2) hipLaunchKernelGGL((hbma_fused16_kernel<4, 2>), grid, block, 0, stream, a);
  else return fail(SVC_ERR_UNSUPPORTED, "hbma fused: levels=%u r_top=%u not instantiated", levels, rt);
  return check_launch("hbma_fused16_kernel");
}

}  // namespace svc
