// hbma_tiled.hip -- the 4-level motion search (the reference's default build, libs/motion.cpp:691-749:
// EstimateMotionHierarchical16x16Sse2, R_top = 1 at every level) with the windows of the two middle levels read from LDS.
//
// Why.  With R_top = 1 the lane-per-block kernel (hbma_fused.hip) issues few SADs per loaded byte, and the 4-level
// search on fine texture returns an incoherent field (a strip of 60 neighbouring blocks spans mv.y over +-8 at C5 and C3b
// alike), so every lane of a wave instruction walks its own row.  Measured on gfx950 (tools/ubench_tcp.hip): such an
// instruction holds the vector L1 for ~35 cycles whatever its width (x1, x2, x4 alike, ~0.55 cycles per lane address;
// dwordx3: 44), against 16 cycles per KiB for whole rows -- the vector L1, not HBM, paces that kernel (84 % busy at
// 0.54-0.63 of the HBM roofline).  Per wave of 64 blocks: level 0 ~1 500 L1 cycles (18 x two gathers + 16 anchor rows),
// level 1 ~470, level 2 ~280, the top level ~50.
//
// What.  A workgroup owns a tile of TBX x TBY MV blocks, one lane per block.  Where a block's window can lie at level l is
// bounded without knowing any vector: |mv_in| <= M_l = 2 (M_{l+1} + R_top), M_top = 0 (libs/motion.cpp:458-463), so the
// union of a tile's windows at a level is a rectangle whose position depends on the tile alone.  For levels 2 and 1 (5 %
// and 19 % of the bytes, a third of the L1 time, and two of the three dependent load -> search steps of a lane) that
// rectangle is brought into LDS as whole 16-byte chunks of whole rows by LDS-DMA (global_load_lds_dwordx4: no registers,
// one KiB per wave instruction) while the 2x2 top level is searched from global memory; levels 2 and 1 then read their
// windows from LDS at per-lane addresses.  Level 0 stays the lane-per-block search of hbma_fused.hip: its superset
// rectangle (+-15 pixels around 16-pixel blocks) is 2.5x the data and would take the whole LDS of a CU for 256 blocks --
// built and measured in two forms (one tile per workgroup; persistent workgroups with two tile buffers and two lanes
// per block), both slower than the lane-per-block kernel because in-order vector-memory issue keeps a wave's tile
// fetch from overlapping its own search and LDS leaves no room for a second wave per SIMD: profiles/EXPERIMENTS.md.
// Arithmetic, candidate order and tie rules are hbma_search.hpp's, so results are bit-identical to the other kernels.
#include <algorithm>

#include "hbma_search.hpp"

#ifndef SVC_HBMA_L0_CEILING
#define SVC_HBMA_L0_CEILING 0
#endif

namespace svc {

template <int B, int M, int RT, int TBX, int TBY>
struct TileGeom {
  static_assert(M + RT <= 16, "the tile starts at most 16 + 15 bytes left of its first anchor column");
  static constexpr int ND = B / 4 + 2;                                       // dwords a whole-block row read spans
  // The tile's first byte is the 16-byte boundary at or below (first anchor column - (M + RT)): whole chunks of whole
  // rows for the DMA.  Tiles whose width in pixels is a multiple of 16 start exactly 16 bytes left of their first anchor
  // column; any other width (15 blocks) up to M + RT + 15.
  static constexpr int X_LEFT_MAX = (TBX * B) % 16 == 0 ? 16 : M + RT + 15;
  static constexpr int Y_TOP = M + RT;                                       // rows above the first anchor row
  static constexpr int A0_MAX = ((TBX - 1) * B + M - RT) & ~3;               // last dword-aligned window origin
  static constexpr int W = (A0_MAX + 4 * ND + X_LEFT_MAX + 15) & ~15;        // bytes per tile row (= LDS pitch)
  static constexpr int CPR = W / 16;                                         // 16-byte chunks per row
  static constexpr int ROWS = Y_TOP + (TBY - 1) * B + (M - RT) + B + 2 * RT;
  static constexpr int CHUNKS = CPR * ROWS;
  // LDS bytes: whole wave instructions (1 KiB each)
  static constexpr int rounds(int nwaves) { return (CHUNKS + 64 * nwaves - 1) / (64 * nwaves); }
  static constexpr int BYTES = ((CHUNKS + 63) / 64) * 1024;
  // x of the tile's first byte, for the tile whose first anchor column is ax0 (level pixels)
  static __device__ __forceinline__ int origin_x(int ax0) { return (ax0 - (M + RT)) & ~15; }
};

struct FakeLevel0Geom { static constexpr int W = 1088, ND = 6; };  // SVC_HBMA_L0_CEILING == 3 (timing experiment)

// The tile of one level, global -> LDS, asynchronously.  Chunk i of the tile (row-major) is fetched by lane i % 64 of the
// wave instruction that covers chunks [i & ~63, +64): LDS-DMA writes a wave's 64 x 16 bytes contiguously from the
// wave-uniform LDS address in M0.  (row, c) = this lane's chunk of round 0 (constant over tiles); a round advances every
// lane by 64 * NWAVES chunks.  Chunks outside the plane are never read (a window is always inside it, make_window):
// their source address is clamped into the plane; row widths are multiples of 16 at every staged level (frame width
// multiple of 64), so a chunk is either inside a row or outside.  Lanes past the last chunk fetch the last row again
// into the padding behind the tile (less than one KiB).
template <class G, int NWAVES>
__device__ __forceinline__ void stage_tile_async(const uint8_t* plane, int fw, int fh, int x0, int y0,
                                                 uint32_t lds_addr, uint32_t wave, uint32_t row, uint32_t c) {
  constexpr uint32_t DR = (64 * NWAVES) / G::CPR, DC = (64 * NWAVES) % G::CPR;
#pragma unroll
  for (int r = 0; r < G::rounds(NWAVES); ++r) {
    const int gy = min(max(y0 + (int)min(row, (uint32_t)(G::ROWS - 1)), 0), fh - 1);
    const int gx = min(max(x0 + 16 * (int)c, 0), fw - 16);
    const uint32_t voff = (uint32_t)gy * (uint32_t)fw + (uint32_t)gx;
    const uint32_t slice = (uint32_t)r * NWAVES + wave;  // this wave instruction's 64 chunks
    if (slice * 64u < (uint32_t)G::CHUNKS) {            // wave-uniform
      const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_addr + slice * 1024u);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(plane) : "memory", "m0");
    }
    c += DC;
    row += DR;
    if (c >= (uint32_t)G::CPR) { c -= (uint32_t)G::CPR; ++row; }
  }
}

// s_waitcnt vmcnt(0) as the BUILTIN (gfx9 encoding: vmcnt 0, expcnt / lgkmcnt untouched): the DMA is issued from inline
// assembly (the compiler would otherwise drain vmcnt in front of every load-use that follows an LDS-DMA it can see, the
// top level's included), so the wait for it is explicit; as a builtin the compiler's own bookkeeping sees it too.
__device__ __forceinline__ void wait_vm_all() {
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
}

// search_level (hbma_search.hpp) with the tracked window read from the level's LDS tile (whose top-left pixel is
// (x0, y0)) and the anchor block already in registers.
template <int B, int RT, int SHIFT, class G>
__device__ __forceinline__ void search_level_lds(uint32_t lds_tile, int x0, int y0, const uint32_t (&a)[B][B / 4], int fw,
                                                 int fh, int bx, int by, int& mvx, int& mvy, uint32_t& best) {
  using P = SadPlan<RT>;
  constexpr int NW = B / 4, NQ = P::NQ, NV = NW + NQ, ND = NV + 1, NDY = 2 * RT + 1, NT = B + 2 * RT;
  static_assert(ND <= G::ND, "the tile's row pitch covers the widest read");
  const int ax = bx * B, ay = by * B;
  const Window w = make_window<B, RT>(ax + mvx, ay + mvy, fw, fh);
  const int a0 = w.wx & ~3;
  const uint32_t sh = (uint32_t)(w.wx & 3);
  const uint32_t p = lds_tile + (uint32_t)((w.wy - y0) * G::W + (a0 - x0));
  const __attribute__((address_space(3))) uint8_t* lp = (const __attribute__((address_space(3))) uint8_t*)(uintptr_t)p;

  uint64_t acc4[NDY][NQ];
  uint32_t acc1[NDY];
#pragma unroll
  for (int d = 0; d < NDY; ++d) {
    acc1[d] = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc4[d][q] = 0;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    uint32_t m[ND], v[NV];
#pragma unroll
    for (int k = 0; k < ND; ++k)
      m[k] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(lp + (t * G::W + 4 * k));
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = __builtin_amdgcn_alignbyte(m[k + 1], m[k], sh);
#pragma unroll
    for (int d = 0; d < NDY; ++d) {
      const int r = t - d;  // anchor row that meets tracked row t at vertical offset d
      if (r >= 0 && r < B) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          const uint32_t av = a[r >= 0 && r < B ? r : 0][k];
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            acc4[d][q] = __builtin_amdgcn_qsad_pk_u16_u8(pack64(v[k + q], v[k + q + 1]), av, acc4[d][q]);
          if (P::kTail) acc1[d] = __builtin_amdgcn_sad_u8(v[k + NQ], av, acc1[d]);
        }
      }
    }
  }
  static_assert(RT == 1, "the tiled kernel is the R_top = 1 search");
  select_refine_packed<RT, SHIFT, NQ>(w, ax, ay, acc4, acc1, mvx, mvy, best);
}

// A workgroup is the tile's TBX x TBY lanes rounded up to whole waves (tile widths that are not powers of two work, lanes
// past the tile idle; none is instantiated: narrow tiles measured slower, see launch_hbma_tiled).
template <int RT, int TBX, int TBY>
__global__ __launch_bounds__((TBX * TBY + 63) / 64 * 64) void hbma_tiled16_kernel(FusedArgs a, uint32_t tiles_x, uint32_t tiles) {
  constexpr int NWAVES = (TBX * TBY + 63) / 64;
  using G1 = TileGeom<8, 6 * RT, RT, TBX, TBY>;  // |mv_in| <= 6 R_top at level 1
  using G2 = TileGeom<4, 2 * RT, RT, TBX, TBY>;  //           2 R_top at level 2
  constexpr uint32_t OFF2 = G1::BYTES;
  __shared__ __attribute__((aligned(16))) uint8_t lds[G1::BYTES + G2::BYTES];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;

  // Region-major order as in hbma_fused_kernel: XCD x (blockIdx 8 k + x) is given the x-th eighth of the tiles of EVERY
  // pair, pairs in order, so that the second read of a pyramid (anchor of pair p, tracked frame of pair p + 1) and the
  // margins neighbouring tiles share are served by that XCD's L2.  Speed only.
  const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
  const uint32_t pair = k / a.wgs_per_region;
  const uint32_t tile = xcd * a.wgs_per_region + (k - pair * a.wgs_per_region);
  if (pair >= a.n_pairs || tile >= tiles) return;  // uniform over the workgroup
  const uint32_t tm = tile / tiles_x, tk = tile - tm * tiles_x;
  const uint32_t mfh = a.blocks / a.mfw;

  const uint32_t tid = threadIdx.x, wave = tid / 64u;
  const uint32_t slot = min(tid, (uint32_t)(TBX * TBY - 1));  // lanes past the tile (the workgroup is whole waves) idle on its last block
  const uint32_t lx = slot % TBX, ly = slot / TBX;
  const uint32_t bxu = tk * TBX + lx, byu = tm * TBY + ly;
  const bool live = tid < (uint32_t)(TBX * TBY) && bxu < a.mfw && byu < mfh;
  // a lane beyond the frame searches the last block of its row / column again (in-tile addresses) and stores nothing
  const int bx = (int)min(bxu, a.mfw - 1), by = (int)min(byu, mfh - 1);

  const uint8_t* trk = a.tracked + (size_t)pair * a.pair_stride;
  const uint8_t* anc = a.anchor + (size_t)pair * a.pair_stride;
  const int w = (int)a.w, h = (int)a.h;
  const size_t o1 = (size_t)w * h, o2 = o1 + (o1 >> 2), o3 = o2 + (o1 >> 4);

  // the tiles of levels 2 and 1, coarse to fine (completion order is issue order) ...
  const int tx = (int)(tk * TBX), ty = (int)(tm * TBY);
  const int x2 = G2::origin_x(tx * 4), y2 = ty * 4 - G2::Y_TOP;
  const int x1 = G1::origin_x(tx * 8), y1 = ty * 8 - G1::Y_TOP;
  stage_tile_async<G2, NWAVES>(trk + o2, w >> 2, h >> 2, x2, y2, lds0 + OFF2, wave, tid / (uint32_t)G2::CPR, tid % (uint32_t)G2::CPR);
  stage_tile_async<G1, NWAVES>(trk + o1, w >> 1, h >> 1, x1, y1, lds0, wave, tid / (uint32_t)G1::CPR, tid % (uint32_t)G1::CPR);
  // ... then what the lane keeps in registers: the top level's rows and the anchor blocks of levels 2 and 1
  TopB2<RT> top;
  load_top_b2<RT>(trk + o3, anc + o3, w >> 3, h >> 3, bx, by, top);
  uint32_t a2r[4][1], a1r[8][2];
  {
    const int f2 = w >> 2, f1 = w >> 1;
    const uint8_t* p2 = anc + o2 + (uint32_t)(by * 4 * f2 + bx * 4);
    const uint8_t* p1 = anc + o1 + (uint32_t)(by * 8 * f1 + bx * 8);
#pragma unroll
    for (int r = 0; r < 4; ++r) load_anchor_row<1>(p2 + (uint32_t)(r * f2), a2r[r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) load_anchor_row<2>(p1 + (uint32_t)(r * f1), a1r[r]);
  }

  int mvx = 0, mvy = 0;
  uint32_t best = 0;
  search_top_b2<RT, 6>(top, bx, by, mvx, mvy, best);

  wait_vm_all();    // this wave's DMA has landed ...
  __syncthreads();  // ... and every other wave's

  mvx *= 2; mvy *= 2;  // motion.cpp:458-460
  search_level_lds<4, RT, 4, G2>(lds0 + OFF2, x2, y2, a2r, w >> 2, h >> 2, bx, by, mvx, mvy, best);
  mvx *= 2; mvy *= 2;
  search_level_lds<8, RT, 2, G1>(lds0, x1, y1, a1r, w >> 1, h >> 1, bx, by, mvx, mvy, best);
  mvx *= 2; mvy *= 2;
#if SVC_HBMA_L0_CEILING == 1 || SVC_HBMA_L0_CEILING == 2
  // TIMING EXPERIMENT ONLY (wrong results; tools/ab_hbma_l0_ceiling.sh): what level 0 would cost if the field were coherent -- every lane's
  // window on its block's own rows (1), and at its block's own columns as well (2).  The kernel's ceiling for any scheme that only reorders or
  // stages level 0's reads: profiles/r06_ab_hbma_tiled_level0.txt.
  mvy = 0;
  if (SVC_HBMA_L0_CEILING >= 2) mvx = 0;
#endif
#if SVC_HBMA_L0_CEILING == 3
  // TIMING EXPERIMENT ONLY (wrong results): level 0's tracked rows read from LDS at per-lane addresses with the REAL vectors' alignments and
  // bank pattern, and nobody putting them there -- a staging scheme whose transfer is free.  What any LDS staging of level 0 can at best
  // reach with this kernel's occupancy: the consumer side alone (anchor rows still from global memory).
  {
    __syncthreads();  // the tiles of levels 2 and 1 are dead: their LDS is read as if it held level 0
    const int fx0 = (int)(tk * TBX) * 16 - 16, fy0 = (by * 16 + mvy - 1) & ~3;  // every lane's window lands inside 22 rows x 1088 bytes
    uint32_t a0r[16][4];
    {
      const uint8_t* p0 = anc + (uint32_t)(by * 16 * w + bx * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) load_anchor_row<4>(p0 + (uint32_t)(r * w), a0r[r]);
    }
    search_level_lds<16, RT, 0, FakeLevel0Geom>(lds0, fx0, fy0, a0r, w, h, bx, by, mvx, mvy, best);
  }
#else
  search_level<16, RT, false, 0>(trk, anc, w, h, bx, by, mvx, mvy, best);
#endif

  if (live) {
    const uint32_t item = pair * a.blocks + byu * a.mfw + bxu;
    reinterpret_cast<float2*>(a.mv)[item] = make_float2((float)mvx, (float)mvy);
    a.mad[item] = (float)best * (1.0f / 256.0f);  // exact: best < 2^24, power-of-two scale
  }
}

// The LDS-tiled kernel serves the 4-level search with R_top = 1 on planes whose rows are whole 16-byte chunks at the
// three levels it stages (so the frame width is a multiple of 64; any height).  (The 3-level search with R_top = 2, level 1
// from LDS, was built the same way and is 4 - 5 % slower than the lane-per-block kernel at C3: the lanes past the frame
// edge and 6 instead of 8 waves per SIMD cost more than the level-1 gathers: profiles/r03_ab_hbma_tiled_3L.txt.)
bool tiled_supported(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw, uint32_t bh) {
  return fused_supported(levels, w, h, range, bw, bh) && bw == 16 && levels == 4 && (range >> 3) == 1 && w % 64 == 0;
}

template <int TBX, int TBY>
static int launch_tiled(FusedArgs a, uint32_t n_pairs, hipStream_t stream) {
  const uint32_t mfh = a.blocks / a.mfw;
  const uint32_t tiles_x = div_up(a.mfw, TBX), tiles = tiles_x * div_up(mfh, TBY);
  a.wgs_per_region = div_up(tiles, 8);
  const uint64_t wgs = (uint64_t)8 * a.wgs_per_region * n_pairs;
  if (wgs > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "hbma: %llu workgroups exceed one launch", (unsigned long long)wgs);
  hipLaunchKernelGGL((hbma_tiled16_kernel<1, TBX, TBY>), dim3((uint32_t)wgs), dim3((TBX * TBY + 63) / 64 * 64), 0, stream, a, tiles_x, tiles);
  return check_launch("hbma_tiled16_kernel");
}

int launch_hbma_tiled(const FusedArgs& a, uint32_t n_pairs, hipStream_t stream) {
  // Tile shape.  Fewer lanes on blocks past the frame's right / bottom edge is better (1080p, 120 x 68 blocks: 64 x 4
  // wastes 6.7 %, 32 x 8 13 %, 16 x 16 25 %; 4K, 240 x 135: 6.7 / 7.4 / 6.7 %), and so is a wider tile: a wave that covers
  // 64 blocks of ONE block row reads its anchor rows as one run and shares the most cache lines among its level-0 windows.
  // Measured at 4K with the same number of lanes: 16 x 16 0.268 ms, 64 x 4 0.233 ms (profiles/r03_ab_hbma_tiled_wide.txt);
  // a 15 x 17 tile, which fits both fields within 1 %, is 7 - 10 % slower than either (r03_ab_hbma_tiled_15x17.txt).  So a
  // tile 16 blocks wide is costed at 1.15 lanes per lane, 32 wide at 1.02, and the cheapest shape wins.
  const uint32_t mfh = a.blocks / a.mfw;
  const uint64_t c16 = (uint64_t)div_up(a.mfw, 16) * div_up(mfh, 16) * 115, c32 = (uint64_t)div_up(a.mfw, 32) * div_up(mfh, 8) * 102,
                 c64 = (uint64_t)div_up(a.mfw, 64) * div_up(mfh, 4) * 100;
  if (c64 <= c32 && c64 <= c16) return launch_tiled<64, 4>(a, n_pairs, stream);
  if (c32 <= c16) return launch_tiled<32, 8>(a, n_pairs, stream);
  return launch_tiled<16, 16>(a, n_pairs, stream);
}

}  // namespace svc
