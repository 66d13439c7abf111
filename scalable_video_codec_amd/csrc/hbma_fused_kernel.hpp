// hbma_fused_kernel.hpp -- the lane-per-block all-level kernel as a template over (MV block, levels, R_top); included by
// one translation unit per MV block size (hbma_fused.hip: 16, hbma_fused8.hip: 8, hbma_fused32.hip: 32) so that the 28
// instantiations compile side by side.  Design notes: hbma_fused.hip.
#pragma once

#include "hbma_search.hpp"

namespace svc {

// Level l of an L-level search of MB x MB blocks: block MB >> l, SAD scale 4^l (areas are powers of two, so the scaled
// integer SAD is the MAD in units of 1 / MB^2).  The top level of 2 x 2 blocks has its own engine.
template <int MB, int L, int RT, int LV>
__device__ __forceinline__ void fused_level(const uint8_t* trk, const uint8_t* anc, int w, int h, size_t off, int bx, int by,
                                            int& mvx, int& mvy, uint32_t& best) {
  constexpr int B = MB >> LV;
  constexpr bool TOP = LV == L - 1;
  static_assert(B >= 2 && (B >= 4 || TOP), "2 x 2 blocks only as the top level");
  if (!TOP) { mvx *= 2; mvy *= 2; }  // motion.cpp:458-460
  if constexpr (B == 2) {
    TopB2<RT> top;
    load_top_b2<RT>(trk + off, anc + off, w >> LV, h >> LV, bx, by, top);
    search_top_b2<RT, 2 * LV>(top, bx, by, mvx, mvy, best);
  } else {
    search_level<B, RT, TOP, 2 * LV>(trk + off, anc + off, w >> LV, h >> LV, bx, by, mvx, mvy, best);
  }
}

template <int MB, int L, int RT>
__global__ __launch_bounds__(256) void hbma_fused_kernel(FusedArgs a) {
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
  const size_t o1 = (size_t)w * h, o2 = o1 + (o1 >> 2), o3 = o2 + (o1 >> 4), o4 = o3 + (o1 >> 6);

  int mvx = 0, mvy = 0;
  uint32_t best = 0;
  if constexpr (L >= 5) fused_level<MB, L, RT, 4>(trk, anc, w, h, o4, bx, by, mvx, mvy, best);
  if constexpr (L >= 4) fused_level<MB, L, RT, 3>(trk, anc, w, h, o3, bx, by, mvx, mvy, best);
  if constexpr (L >= 3) fused_level<MB, L, RT, 2>(trk, anc, w, h, o2, bx, by, mvx, mvy, best);
  if constexpr (L >= 2) fused_level<MB, L, RT, 1>(trk, anc, w, h, o1, bx, by, mvx, mvy, best);
  fused_level<MB, L, RT, 0>(trk, anc, w, h, 0, bx, by, mvx, mvy, best);

  reinterpret_cast<float2*>(a.mv)[item] = make_float2((float)mvx, (float)mvy);
  a.mad[item] = (float)best * (1.0f / (float)(MB * MB));  // exact: best < 2^24, power-of-two scale
}

// The instantiations of one MV block size: L = 2 .. log2(MB) (the top level's blocks are at least 2 x 2), R_top = 1 .. 4
// (1 .. 2 for 32 x 32 blocks: beyond that the 32-bit SAD sums of 49 / 81 candidates no longer fit the register file).
template <int MB, int L>
static int launch_fused_rt(const FusedArgs& a, uint32_t rt, dim3 grid, hipStream_t stream) {
  const dim3 block(256);
  switch (rt) {
    case 1: hipLaunchKernelGGL((hbma_fused_kernel<MB, L, 1>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((hbma_fused_kernel<MB, L, 2>), grid, block, 0, stream, a); break;
    case 3:
      if constexpr (MB <= 16) { hipLaunchKernelGGL((hbma_fused_kernel<MB, L, 3>), grid, block, 0, stream, a); break; }
      [[fallthrough]];
    case 4:
      if constexpr (MB <= 16) { hipLaunchKernelGGL((hbma_fused_kernel<MB, L, 4>), grid, block, 0, stream, a); break; }
      [[fallthrough]];
    default: return fail(SVC_ERR_UNSUPPORTED, "hbma fused: r_top=%u not instantiated for %d x %d blocks", rt, MB, MB);
  }
  return check_launch("hbma_fused_kernel");
}

template <int MB>
static int launch_fused_mb(const FusedArgs& a, uint32_t levels, uint32_t rt, dim3 grid, hipStream_t stream) {
  if (levels == 2) return launch_fused_rt<MB, 2>(a, rt, grid, stream);
  if (levels == 3) return launch_fused_rt<MB, 3>(a, rt, grid, stream);
  if constexpr (MB >= 16)
    if (levels == 4) return launch_fused_rt<MB, 4>(a, rt, grid, stream);
  if constexpr (MB >= 32)
    if (levels == 5) return launch_fused_rt<MB, 5>(a, rt, grid, stream);
  return fail(SVC_ERR_UNSUPPORTED, "hbma fused: levels=%u not instantiated for %d x %d blocks", levels, MB, MB);
}

}  // namespace svc
