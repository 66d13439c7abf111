// hbma_fused.hip -- EstimateMotionHierarchical as ONE launch: every pyramid level
// of one MV block (8x8, 16x16 or 32x32) is searched by one lane.
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
#include "hbma_fused_kernel.hpp"

namespace svc {

int launch_fused_mb8(const FusedArgs& a, uint32_t levels, uint32_t rt, dim3 grid, hipStream_t stream);   // hbma_fused8.hip
int launch_fused_mb32(const FusedArgs& a, uint32_t levels, uint32_t rt, dim3 grid, hipStream_t stream);  // hbma_fused32.hip

// What the kernel is instantiated for (the shapes apps/encoder.cpp:75-104 admits at low cost): square MV blocks of 8, 16
// or 32 pixels, 2 .. log2(block) levels (the top level's blocks are at least 2 x 2), R_top = search_range >> (levels - 1)
// in 1 .. 4 (1 .. 2 for 32 x 32).  Everything else (non-square blocks, one level, wider searches) takes the per-level kernel (hbma_wave.hip).
bool fused_supported(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw,
                     uint32_t bh) {
  if (bw != bh || (bw != 8 && bw != 16 && bw != 32) || levels < 2 || levels > 5 || (bw >> (levels - 1)) < 2) return false;
  const uint32_t rt = range >> (levels - 1);
  if (rt < 1 || rt > (bw == 32 ? 2u : 4u)) return false;  // 32 x 32 blocks at R_top 3 - 4 would spill (hbma_fused_kernel.hpp)
  const uint32_t tw = w >> (levels - 1), th = h >> (levels - 1), tb = bw >> (levels - 1);
  // the top plane must hold a whole candidate grid.  Its rows need not be whole dwords when the top block is 2 x 2
  // (load_top_b2 reads them where they lie: 720 -> 90, 176 -> 22 at 4 levels of 16 x 16); the larger top blocks go through
  // load_row<.., true>, whose in-row clamp only leaves the needed bytes alone when the row width is a multiple of 4 (found
  // by tests/test_gpu_hbma_property.py: 112 x 32, 4 levels) -- and their top widths always are (w is a multiple of the block)
  return tw >= tb + 8 && tw >= 12 && (tw % 4 == 0 || tb == 2) && th >= tb + 2 * rt && (w % bw == 0) && (h % bh == 0);
}

int launch_hbma_tiled(const FusedArgs& a, uint32_t n_pairs, hipStream_t stream);

bool tiled_is_default() { return kTiledIsDefault; }

// kernel: 0 = the shape's default, 1 = lane-per-block (no LDS), 2 = LDS-tiled (UNSUPPORTED where tiled_supported is false)
int launch_hbma_fused(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride,
                      uint32_t n_pairs, uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t mb,
                      float* d_mv, float* d_mad, int kernel, hipStream_t stream) {
  FusedArgs a;
  a.tracked = d_tracked;
  a.anchor = d_anchor;
  a.pair_stride = pair_stride;
  a.mfw = w / mb;
  a.blocks = a.mfw * (h / mb);
  const uint64_t items = (uint64_t)a.blocks * n_pairs;
  if (items == 0) return SVC_OK;
  if (items > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "hbma: %llu work items exceed one launch", (unsigned long long)items);
  a.n_items = (uint32_t)items;
  a.w = w; a.h = h;
  a.mv = d_mv;
  a.mad = d_mad;
  a.n_pairs = n_pairs;
  const bool can_tile = tiled_supported(levels, w, h, range, mb, mb) && ((uintptr_t)d_tracked % 16 == 0) &&
                        ((uintptr_t)d_anchor % 16 == 0) && pair_stride % 16 == 0;
  if (kernel == 2 && !can_tile)
    return fail(SVC_ERR_UNSUPPORTED, "hbma: the LDS-tiled kernel needs 16 x 16 blocks, 4 levels, r_top 1, a frame width that is a multiple of 64 and 16-byte aligned pyramids");
  if (kernel == 2 || (kernel == 0 && can_tile && kTiledIsDefault)) return launch_hbma_tiled(a, n_pairs, stream);
  a.wgs_per_region = div_up(div_up(a.blocks, 256), 8);
  const uint64_t wgs = (uint64_t)8 * a.wgs_per_region * n_pairs;
  if (wgs > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "hbma: %llu workgroups exceed one launch", (unsigned long long)wgs);
  const dim3 grid((uint32_t)wgs);
  const uint32_t rt = range >> (levels - 1);
  if (mb == 8) return launch_fused_mb8(a, levels, rt, grid, stream);
  if (mb == 32) return launch_fused_mb32(a, levels, rt, grid, stream);
  return launch_fused_mb<16>(a, levels, rt, grid, stream);
}

}  // namespace svc
