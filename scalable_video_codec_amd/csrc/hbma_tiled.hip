// hbma_tiled.hip -- the 4-level motion search (the reference's default build, libs/motion.cpp:691-749:
// EstimateMotionHierarchical16x16Sse2, R_top = 1 at every level) with every tracked window read from LDS.
//
// Why.  With R_top = 1 the lane-per-block kernel (hbma_fused.hip) issues two vector loads per tracked row for 12 QSADs,
// and the 4-level search on fine texture returns an incoherent field (a strip of 60 neighbouring blocks spans mv.y over
// +-8 at C5 and C3b alike), so every lane of a wave instruction walks its own row.  Measured on gfx950
// (tools/ubench_tcp.hip): such an instruction holds the vector L1 for ~35 cycles whatever its width (x1, x2, x4 alike,
// ~0.55 cycles per lane address), against 16 cycles per KiB for whole rows -- the vector L1, not HBM, paces that kernel
// (84 % busy at 0.54-0.63 of the HBM roofline).
//
// What.  A workgroup owns a tile of TBX x TBY MV blocks.  Where a block's window can lie at level l is bounded without
// knowing any vector: |mv_in| <= M_l = 2 (M_{l+1} + R_top), M_top = 0 (libs/motion.cpp:458-463), so the union of a
// tile's windows at levels 2, 1 and 0 is three rectangles whose position depends on the tile alone.  They are brought
// into LDS as whole 16-byte chunks of whole rows by LDS-DMA (global_load_lds_dwordx4: no registers, one KiB per wave
// instruction); the anchor blocks go to registers (a wave's anchor row is a contiguous run); the searches then read
// their windows from LDS at per-lane addresses.  The vector L1 sees whole-row traffic only.
//
// Schedule.  What bounds this form is the rate at which ONE CU can pull bytes (~35 GB/s with every CU pulling, set by its
// miss queue), so the pull must never stop: a workgroup is persistent (one per CU), walks the tiles of its XCD's region
// of every pair, and keeps TWO tile buffers in LDS -- while tile i is searched out of one, the DMA and the anchor loads of
// tile i + 1 are in flight into the other (a first version with one buffer and one tile per workgroup ran load phase and
// search phase back to back on every CU and lost to the lane-per-block kernel: profiles/EXPERIMENTS.md).  Two buffers
// leave LDS for 128 blocks per tile, so that all four SIMDs work on a tile TWO lanes share a block: at levels 0 and 1 a
// lane takes the left or right half of the block's columns, at level 2 the upper or lower two rows, the two partial SAD
// sets are added across the lane pair (DPP) and both lanes run the selection.  Arithmetic, candidate order and tie rules
// are hbma_search.hpp's, so results are bit-identical to the other kernels.
//
// The DMA is issued from inline assembly on purpose: the compiler would put `s_waitcnt vmcnt(0)` in front of every LDS
// read that follows an LDS-DMA it can see (it cannot tell the two buffers apart), which would serialise exactly what this
// schedule overlaps.  One explicit vmcnt(0) + barrier per tile closes the hand-over.
#include <algorithm>

#include "hbma_search.hpp"

namespace svc {

template <int B, int M, int RT, int TBX, int TBY>
struct TileGeom {
  static_assert(M + RT <= 16, "the tile starts 16 bytes left of its first anchor column");
  static constexpr int ND = B / 4 + 2;                                       // dwords a whole-block row read spans
  static constexpr int X_LEFT = 16;                                          // bytes left of the first anchor column
  static constexpr int Y_TOP = M + RT;                                       // rows above the first anchor row
  static constexpr int A0_MAX = ((TBX - 1) * B + M - RT) & ~3;               // last dword-aligned window origin
  static constexpr int W = (A0_MAX + 4 * ND + X_LEFT + 15) & ~15;            // bytes per tile row (= LDS pitch)
  static constexpr int CPR = W / 16;                                         // 16-byte chunks per row
  static constexpr int ROWS = Y_TOP + (TBY - 1) * B + (M - RT) + B + 2 * RT;
  static constexpr int CHUNKS = CPR * ROWS;
  // LDS bytes: whole workgroup-wide DMA rounds (nwaves x 1 KiB each), so that the fill is branch-free
  static constexpr int rounds(int nwaves) { return (CHUNKS + 64 * nwaves - 1) / (64 * nwaves); }
  static constexpr int bytes(int nwaves) { return rounds(nwaves) * 1024 * nwaves; }
};

// The tile of one level, global -> LDS, asynchronously.  Chunk i of the tile (row-major) is fetched by lane i % 64 of the
// wave instruction that covers chunks [i & ~63, +64): LDS-DMA writes a wave's 64 x 16 bytes contiguously from the
// wave-uniform LDS address in M0.  (row, c) = this lane's chunk of round 0 (constant over tiles); a round advances every
// lane by 64 * NWAVES chunks.  Chunks outside the plane are never read (a window is always inside it, make_window):
// their source address is clamped into the plane; row widths are multiples of 16 at every staged level (frame width
// multiple of 64), so a chunk is either inside a row or outside.  Lanes past the last chunk fetch the last row again
// into the padding behind the tile.
template <class G, int NWAVES>
__device__ __forceinline__ void stage_tile_async(const uint8_t* plane, int fw, int fh, int x0, int y0,
                                                 uint32_t lds_addr, uint32_t wave, uint32_t row, uint32_t c) {
  constexpr uint32_t DR = (64 * NWAVES) / G::CPR, DC = (64 * NWAVES) % G::CPR;
#pragma unroll
  for (int r = 0; r < G::rounds(NWAVES); ++r) {
    const int gy = min(max(y0 + (int)min(row, (uint32_t)(G::ROWS - 1)), 0), fh - 1);
    const int gx = min(max(x0 + 16 * (int)c, 0), fw - 16);
    const uint32_t voff = (uint32_t)gy * (uint32_t)fw + (uint32_t)gx;
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_addr + ((uint32_t)r * NWAVES + wave) * 1024u);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(plane) : "memory");
    c += DC;
    row += DR;
    if (c >= (uint32_t)G::CPR) { c -= (uint32_t)G::CPR; ++row; }
  }
}

// s_waitcnt vmcnt(0) as the BUILTIN (gfx9 encoding: vmcnt 0, expcnt / lgkmcnt untouched), so that the compiler's own
// bookkeeping also knows that no load is pending behind it -- an asm wait would leave it waiting again, mid-search, for
// registers the prologue loaded.
__device__ __forceinline__ void wait_vm_all() {
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint32_t pair_sum(uint32_t v) {  // v + the other lane of the pair's v
  return v + (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

// One level of one block, by the two lanes that share it: this lane's part is the BW x BH sub-block at (xoff, yoff) of
// the B x B block; its tracked window comes from the level's LDS tile (whose top-left pixel is (x0, y0)), its anchor
// sub-block is in registers.  The window clamps are the whole block's (libs/motion.cpp:375-385).
template <int B, int BW, int BH, int RT, int SHIFT, class G>
__device__ __forceinline__ void search_shared_lds(uint32_t lds_tile, int x0, int y0, const uint32_t (&a)[BH][BW / 4],
                                                  int xoff, int yoff, int fw, int fh, int bx, int by, int& mvx, int& mvy,
                                                  uint32_t& best) {
  constexpr int NW = BW / 4, ND = NW + 2, NDY = 2 * RT + 1, NT = BH + 2 * RT;
  static_assert(RT == 1, "only the 3 x 3 grid is split across lane pairs (candidates 0..2 in the QSAD word)");
  const int ax = bx * B, ay = by * B;
  const Window w = make_window<B, RT>(ax + mvx, ay + mvy, fw, fh);
  const int wxs = w.wx + xoff;
  const int a0 = wxs & ~3;
  const uint32_t sh = (uint32_t)(wxs & 3);
  const uint32_t p = lds_tile + (uint32_t)((w.wy + yoff - y0) * G::W + (a0 - x0));
  const __attribute__((address_space(3))) uint8_t* lp = (const __attribute__((address_space(3))) uint8_t*)(uintptr_t)p;

  // one wave per SIMD: the whole window is requested before the first SAD so that the LDS latency is paid once
  uint32_t m[NT][ND];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int k = 0; k < ND; ++k)
      m[t][k] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(lp + (t * G::W + 4 * k));
  uint64_t acc4[NDY];
#pragma unroll
  for (int d = 0; d < NDY; ++d) acc4[d] = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    uint32_t v[NW + 1];
#pragma unroll
    for (int k = 0; k <= NW; ++k) v[k] = __builtin_amdgcn_alignbyte(m[t][k + 1], m[t][k], sh);
#pragma unroll
    for (int d = 0; d < NDY; ++d) {
      const int r = t - d;  // anchor row of the sub-block that meets tracked row t at vertical offset d
      if (r >= 0 && r < BH) {
#pragma unroll
        for (int k = 0; k < NW; ++k)
          acc4[d] = __builtin_amdgcn_qsad_pk_u16_u8(pack64(v[k], v[k + 1]), a[r >= 0 && r < BH ? r : 0][k], acc4[d]);
      }
    }
  }
  // the block's SADs = this lane's + its partner's; 4 x u16 fields that cannot carry into each other (a block's SAD
  // is at most 65280)
  uint32_t lo[NDY], hi[NDY];
#pragma unroll
  for (int d = 0; d < NDY; ++d) {
    lo[d] = pair_sum((uint32_t)acc4[d]);
    hi[d] = pair_sum((uint32_t)(acc4[d] >> 32));
  }
  select<RT, false, SHIFT>(
      w, ax, ay, [&](int d, int j) { return j == 0 ? lo[d] & 0xFFFFu : j == 1 ? lo[d] >> 16 : hi[d] & 0xFFFFu; }, mvx, mvy,
      best);
}

// What a lane holds in registers for one tile: the top level's rows (both lanes of a pair hold the same) and its
// sub-blocks of the anchor blocks of levels 2, 1, 0.
template <int RT>
struct TileRegs {
  TopB2<RT> top;
  uint32_t a2[2][1], a1[8][1], a0[16][2];
};

struct TilePos {
  uint32_t pair;
  int tx, ty;  // first block column / row of the tile
};

template <int RT, int TBX, int TBY>
__global__ __launch_bounds__(2 * TBX* TBY) void hbma_tiled16_kernel(FusedArgs a, uint32_t tiles_x, uint32_t tiles) {
  constexpr int NWAVES = 2 * TBX * TBY / 64;
  using G0 = TileGeom<16, 14 * RT, RT, TBX, TBY>;  // |mv_in| <= 14 R_top at level 0
  using G1 = TileGeom<8, 6 * RT, RT, TBX, TBY>;    //           6 R_top at level 1
  using G2 = TileGeom<4, 2 * RT, RT, TBX, TBY>;    //           2 R_top at level 2
  constexpr uint32_t OFF1 = G0::bytes(NWAVES), OFF2 = OFF1 + G1::bytes(NWAVES), BUF = OFF2 + G2::bytes(NWAVES);
  static_assert(2 * BUF <= 160 * 1024, "two tile buffers must fit the CU's LDS");
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BUF];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;

  const uint32_t tid = threadIdx.x, wave = tid / 64u;
  const uint32_t blk = tid >> 1, half = tid & 1u;
  const uint32_t lx = blk % TBX, ly = blk / TBX;
  const uint32_t mfh = a.blocks / a.mfw;
  const int w = (int)a.w, h = (int)a.h;
  const size_t o1 = (size_t)w * h, o2 = o1 + (o1 >> 2), o3 = o2 + (o1 >> 4);
  // this lane's chunk of DMA round 0, per level
  const uint32_t r0 = tid / (uint32_t)G0::CPR, c0 = tid - r0 * (uint32_t)G0::CPR;
  const uint32_t r1 = tid / (uint32_t)G1::CPR, c1 = tid - r1 * (uint32_t)G1::CPR;
  const uint32_t r2 = tid / (uint32_t)G2::CPR, c2 = tid - r2 * (uint32_t)G2::CPR;

  // Tile walk.  Workgroups are dealt round-robin over the 8 XCDs: the workgroups of XCD x (blockIdx 8 j + x) walk, side
  // by side, the x-th eighth of the tiles of pair 0, then of pair 1, ...: the pyramid of frame p + 1 is the anchor of pair
  // p and the tracked frame of pair p + 1, so its second read follows the first within about one tile step and finds it in
  // that XCD's L2, as do the margins that neighbouring tiles share.  Speed only: any placement gives the same result.
  const uint32_t xcd = blockIdx.x & 7u, stride = gridDim.x >> 3, tpr = a.wgs_per_region, steps = a.n_pairs * tpr;
  auto tile_at = [&](uint32_t s, TilePos& t) -> bool {
    t.pair = s / tpr;
    const uint32_t tile = xcd * tpr + (s - t.pair * tpr);
    const uint32_t tm = tile / tiles_x;
    t.tx = (int)((tile - tm * tiles_x) * TBX);
    t.ty = (int)(tm * TBY);
    return tile < tiles;
  };
  auto block_of = [&](const TilePos& t, int& bx, int& by) -> bool {
    const uint32_t bxu = (uint32_t)t.tx + lx, byu = (uint32_t)t.ty + ly;
    // a lane beyond the frame searches the last block of its row / column again (in-tile addresses) and stores nothing
    bx = (int)min(bxu, a.mfw - 1);
    by = (int)min(byu, mfh - 1);
    return bxu < a.mfw && byu < mfh;
  };
  // everything tile `t` needs, on its way: DMA of the three tracked tiles into buffer `buf` (coarse to fine), then the
  // lane's registers
  auto fetch = [&](const TilePos& t, uint32_t buf, TileRegs<RT>& g) {
    const uint8_t* trk = a.tracked + (size_t)t.pair * a.pair_stride;
    const uint8_t* anc = a.anchor + (size_t)t.pair * a.pair_stride;
    const uint32_t base = lds0 + buf * BUF;
    stage_tile_async<G2, NWAVES>(trk + o2, w >> 2, h >> 2, t.tx * 4 - G2::X_LEFT, t.ty * 4 - G2::Y_TOP, base + OFF2, wave, r2, c2);
    stage_tile_async<G1, NWAVES>(trk + o1, w >> 1, h >> 1, t.tx * 8 - G1::X_LEFT, t.ty * 8 - G1::Y_TOP, base + OFF1, wave, r1, c1);
    stage_tile_async<G0, NWAVES>(trk, w, h, t.tx * 16 - G0::X_LEFT, t.ty * 16 - G0::Y_TOP, base, wave, r0, c0);
    int bx, by;
    block_of(t, bx, by);
    load_top_b2<RT>(trk + o3, anc + o3, w >> 3, h >> 3, bx, by, g.top);
    const int f2 = w >> 2, f1 = w >> 1;
    const uint8_t* p2 = anc + o2 + (uint32_t)((by * 4 + 2 * (int)half) * f2 + bx * 4);
    const uint8_t* p1 = anc + o1 + (uint32_t)(by * 8 * f1 + bx * 8 + 4 * (int)half);
    const uint8_t* p0 = anc + (uint32_t)(by * 16 * w + bx * 16 + 8 * (int)half);
#pragma unroll
    for (int r = 0; r < 2; ++r) load_anchor_row<1>(p2 + (uint32_t)(r * f2), g.a2[r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) load_anchor_row<1>(p1 + (uint32_t)(r * f1), g.a1[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) load_anchor_row<2>(p0 + (uint32_t)(r * w), g.a0[r]);
    asm volatile("" ::: "memory");  // the loads above stay here: they are the prefetch
  };
  auto search = [&](const TilePos& t, uint32_t buf, const TileRegs<RT>& g) {
    int bx, by;
    const bool live = block_of(t, bx, by);
    const uint32_t base = lds0 + buf * BUF;
    int mvx = 0, mvy = 0;
    uint32_t best = 0;
    search_top_b2<RT, 6>(g.top, bx, by, mvx, mvy, best);
    mvx *= 2; mvy *= 2;  // motion.cpp:458-460
    search_shared_lds<4, 4, 2, RT, 4, G2>(base + OFF2, t.tx * 4 - G2::X_LEFT, t.ty * 4 - G2::Y_TOP, g.a2, 0, 2 * (int)half,
                                          w >> 2, h >> 2, bx, by, mvx, mvy, best);
    mvx *= 2; mvy *= 2;
    search_shared_lds<8, 4, 8, RT, 2, G1>(base + OFF1, t.tx * 8 - G1::X_LEFT, t.ty * 8 - G1::Y_TOP, g.a1, 4 * (int)half, 0,
                                          w >> 1, h >> 1, bx, by, mvx, mvy, best);
    mvx *= 2; mvy *= 2;
    search_shared_lds<16, 8, 16, RT, 0, G0>(base, t.tx * 16 - G0::X_LEFT, t.ty * 16 - G0::Y_TOP, g.a0, 8 * (int)half, 0, w, h,
                                            bx, by, mvx, mvy, best);
    if (live && half == 0) {
      const uint32_t item = t.pair * a.blocks + (uint32_t)by * a.mfw + (uint32_t)bx;
      reinterpret_cast<float2*>(a.mv)[item] = make_float2((float)mvx, (float)mvy);
      a.mad[item] = (float)best * (1.0f / 256.0f);  // exact: best < 2^24, power-of-two scale
    }
  };

  // first tile of this workgroup (a region's last steps may name tiles past the frame: skipped by everyone)
  uint32_t s = blockIdx.x >> 3;
  TilePos cur_t, next_t;
  while (s < steps && !tile_at(s, cur_t)) s += stride;
  if (s >= steps) return;
  TileRegs<RT> cur, next;
  uint32_t buf = 0;
  fetch(cur_t, buf, cur);
  wait_vm_all();
  __syncthreads();
  for (;;) {
    uint32_t sn = s + stride;
    while (sn < steps && !tile_at(sn, next_t)) sn += stride;
    const bool more = sn < steps;  // uniform over the workgroup
#if !defined(SVC_TILED_EXP)
    if (more) fetch(next_t, buf ^ 1u, next);
    search(cur_t, buf, cur);
#elif SVC_TILED_EXP == 1  // loads only
    if (more) fetch(next_t, buf ^ 1u, next);
#elif SVC_TILED_EXP == 2  // search only (stale data)
    search(cur_t, buf, cur);
#elif SVC_TILED_EXP == 3  // search, then fetch: no overlap by construction
    search(cur_t, buf, cur);
    if (more) fetch(next_t, buf ^ 1u, next);
#endif
    // hand-over: every wave's DMA of the next tile has landed, and every wave is done reading this tile's buffer
    // before the fetch of the tile after next overwrites it
    wait_vm_all();
    __syncthreads();
    if (!more) break;
    cur = next;
    cur_t = next_t;
    s = sn;
    buf ^= 1u;
  }
}

// The LDS-tiled kernel serves the 4-level search with R_top = 1 on planes whose rows are whole 16-byte chunks at the
// three levels it stages (so the frame width is a multiple of 64; any height).
bool tiled_supported(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw, uint32_t bh) {
  return fused_supported(levels, w, h, range, bw, bh) && levels == 4 && (range >> 3) == 1 && w % 64 == 0;
}

template <int TBX, int TBY>
static int launch_tiled(FusedArgs a, uint32_t n_pairs, uint32_t cus, hipStream_t stream) {
  const uint32_t mfh = a.blocks / a.mfw;
  const uint32_t tiles_x = div_up(a.mfw, TBX), tiles = tiles_x * div_up(mfh, TBY);
  a.wgs_per_region = div_up(tiles, 8);
  const uint64_t steps = (uint64_t)a.wgs_per_region * n_pairs;
  if (steps > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "hbma: %llu tiles per XCD exceed one launch", (unsigned long long)steps);
  // one persistent workgroup per CU (two tile buffers fill its LDS), fewer when there is less work than that
  const uint32_t per_xcd = (uint32_t)std::min<uint64_t>(std::max(cus / 8u, 1u), steps);
  hipLaunchKernelGGL((hbma_tiled16_kernel<1, TBX, TBY>), dim3(8 * per_xcd), dim3(2 * TBX * TBY), 0, stream, a, tiles_x, tiles);
  return check_launch("hbma_tiled16_kernel");
}

int launch_hbma_tiled(const FusedArgs& a, uint32_t n_pairs, hipStream_t stream) {
  static const uint32_t cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return (uint32_t)n;
  }();
  // tile shape: the one that wastes fewer lanes on the frame's right / bottom edge (16 x 8 also stages 8 % fewer bytes)
  const uint32_t mfh = a.blocks / a.mfw;
  const uint64_t lanes_16x8 = (uint64_t)div_up(a.mfw, 16) * div_up(mfh, 8), lanes_32x4 = (uint64_t)div_up(a.mfw, 32) * div_up(mfh, 4);
  if (lanes_16x8 <= lanes_32x4) return launch_tiled<16, 8>(a, n_pairs, cus, stream);
  return launch_tiled<32, 4>(a, n_pairs, cus, stream);
}

}  // namespace svc
