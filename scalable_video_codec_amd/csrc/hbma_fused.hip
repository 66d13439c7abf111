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
// SAD engine.  v_qsad_pk_u16_u8 returns, for one 4-byte anchor word, the four SADs
// against the tracked bytes at offsets 0..3 of an 8-byte window, accumulated as
// 4 x u16 (a 16x16 block's SAD <= 65280 fits).  Measured on gfx950 (tools/
// ubench_valu.hip): QSAD issues in 16 cycles per wave, v_sad_u8 / v_alignbyte_b32 /
// v_min3 in 4, i.e. 4 cycles per 4-byte SAD either way, and rocprof shows this kernel
// VALU-bound (not HBM-bound), so the instruction count is what is minimised: each
// tracked row is funnel-shifted ONCE to the window origin (shared by every vertical
// offset), then one QSAD covers dx = 0..3 and one v_sad_u8 the fifth column.
// Candidates outside the reference's clamped window (libs/motion.cpp:375-385) are
// masked at selection time.
//
// Arithmetic.  All block areas are powers of two, so MAD = sad / area is an exact
// dyadic rational; the MAD carried across levels (libs/motion.cpp:401 compares a
// level-l MAD with the level-(l+1) minimum) is kept as the integer sad << 2l
// (units of 1/256) and converted once at the end: bit-identical to the float path.
#include "svc_common.hpp"

namespace svc {

typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

struct FusedArgs {
  const uint8_t* tracked;
  const uint8_t* anchor;
  uint64_t pair_stride;
  uint32_t n_items;  // pairs * blocks
  uint32_t n_pairs;
  uint32_t wgs_per_region;  // workgroups of one frame pair per XCD region (see the kernels)
  uint32_t blocks;
  uint32_t mfw;
  uint32_t w, h;     // base-level frame size
  float* mv;
  float* mad;
};

__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) {
  return ((uint64_t)hi << 32) | lo;
}

// Loads N consecutive dwords of a tracked row.  CLAMP (top level only, the last
// plane of a packed pyramid): every dword's column is clamped into the row, so
// nothing past the pyramid is ever touched; a clamped dword only feeds masked
// candidates.
// Addresses are `plane + 32-bit offset`: the plane pointer is uniform over the wavefront in the region-major kernel (the
// pair comes from blockIdx alone), so the loads take the scalar-base + 32-bit-VGPR-offset form and no 64-bit address is
// ever built in vector registers.
template <int N, bool CLAMP>
__device__ __forceinline__ void load_row(const uint8_t* plane, uint32_t row_off, int a0, int fw, uint32_t (&m)[N]) {
  if (CLAMP) {
#pragma unroll
    for (int k = 0; k < N; ++k)
      m[k] = *reinterpret_cast<const uint32_t*>(plane + (row_off + (uint32_t)min(a0 + 4 * k, fw - 4)));
  } else {
    const uint8_t* p = plane + (row_off + (uint32_t)a0);
    if (N == 6) {
      u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
      u32x2_a4 u = *reinterpret_cast<const u32x2_a4*>(p + 16);
      m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w; m[4] = u.x; m[5] = u.y;
    } else if (N == 4) {
      u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
      m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w;
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) m[k] = *reinterpret_cast<const uint32_t*>(p + 4 * k);
    }
  }
}

template <int NW>
__device__ __forceinline__ void load_anchor_row(const uint8_t* p, uint32_t (&a)[NW]) {
  if (NW == 4) {
    u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
  } else if (NW == 2) {
    u32x2_a4 v = *reinterpret_cast<const u32x2_a4*>(p);
    a[0] = v.x; a[1] = v.y;
  } else {
    a[0] = *reinterpret_cast<const uint32_t*>(p);
  }
}

struct Window {
  int wx, wy;              // origin of the (2RT+1) x (2RT+1) candidate grid (always in the plane)
  int jlo, jhi, dlo, dhi;  // the reference's clamped window inside that grid
};

template <int B, int RT>
__device__ __forceinline__ Window make_window(int cx, int cy, int fw, int fh) {
  Window w;
  const int x0 = max(0, cx - RT), x1 = min(fw - B + 1, cx + RT + 1);  // motion.cpp:381-385
  const int y0 = max(0, cy - RT), y1 = min(fh - B + 1, cy + RT + 1);  // :375-379
  w.wx = min(max(cx - RT, 0), fw - (B + 2 * RT));
  w.wy = min(max(cy - RT, 0), fh - (B + 2 * RT));
  w.jlo = x0 - w.wx; w.jhi = x1 - w.wx;
  w.dlo = y0 - w.wy; w.dhi = y1 - w.wy;
  return w;
}

// Picks the winner of the (2RT+1)^2 grid of SADs in the reference's raster order with one
// unsigned min over packed keys  (scaled_sad << 5) | code :
//   refinement (motion.cpp:401, strict `<` against the carried minimum): code = raster
//     index, so equal SADs resolve to the FIRST candidate; the winner replaces the carried
//     value only if its scaled SAD is strictly smaller;
//   top level (motion.cpp:324-337, `<=`): code = 31 - index, so equal SADs resolve to the
//     LAST candidate; and if the valid SADs are non-increasing in raster order every
//     candidate "updated" and the MV is zeroed (the minimum is kept).
// Candidates outside the reference's clamped window get the all-ones key.
template <int RT, bool TOP, int SHIFT, typename GetSad>
__device__ __forceinline__ void select(const Window& w, int ax, int ay, GetSad sad_at, int& mvx,
                                       int& mvy, uint32_t& best) {
  constexpr int N = 2 * RT + 1;
  static_assert(N * N <= 32, "raster index must fit the 5-bit code");
  uint32_t kmin = 0xFFFFFFFFu;
  uint32_t prev = 0xFFFFFFFFu;  // FLT_MAX of motion.cpp:290
  bool mono = true;
#pragma unroll
  for (int d = 0; d < N; ++d) {
    const bool row_ok = d >= w.dlo && d < w.dhi;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const bool valid = row_ok && j >= w.jlo && j < w.jhi;
      const uint32_t s = sad_at(d, j);
      const int idx = d * N + j;
      const uint32_t key = (s << (SHIFT + 5)) | (uint32_t)(TOP ? 31 - idx : idx);
      kmin = min(kmin, valid ? key : 0xFFFFFFFFu);
      if (TOP) {
        mono = mono && (!valid || s <= prev);
        prev = valid ? s : prev;
      }
    }
  }
  const uint32_t smin = kmin >> 5;  // scaled SAD of the winner
  const int idx = TOP ? 31 - (int)(kmin & 31u) : (int)(kmin & 31u);
  const int bd = idx / N, bj = idx - bd * N;
  if (TOP) {
    best = smin;
    mvx = mono ? 0 : w.wx + bj - ax;
    mvy = mono ? 0 : w.wy + bd - ay;
  } else if (smin < best) {
    best = smin;
    mvx = w.wx + bj - ax;
    mvy = w.wy + bd - ay;
  }
}

// One level with block size B >= 4.  Per tracked row: NW + 2 aligned dwords are loaded
// and funnel-shifted once (v_alignbyte_b32) so that word k starts at window byte 4k; then
// for every anchor row that meets it, per anchor word: one v_qsad_pk_u16_u8 (candidates
// dx = 0..3) and, for RT = 2, one v_sad_u8 (dx = 4).
template <int B, int RT, bool TOP, int SHIFT>
__device__ __forceinline__ void search_level(const uint8_t* __restrict__ trk,
                                             const uint8_t* __restrict__ anc, int fw, int fh,
                                             int bx, int by, int& mvx, int& mvy, uint32_t& best) {
  constexpr int NW = B / 4, ND = NW + 2, NDY = 2 * RT + 1, NT = B + 2 * RT;
  const int ax = bx * B, ay = by * B;
  const Window w = make_window<B, RT>(ax + mvx, ay + mvy, fw, fh);
  const int a0 = w.wx & ~3;
  const uint32_t sh = (uint32_t)(w.wx & 3);

  uint64_t acc4[NDY];
  uint32_t acc1[NDY];
#pragma unroll
  for (int d = 0; d < NDY; ++d) { acc4[d] = 0; acc1[d] = 0; }
  uint32_t a[B][NW];
  const uint32_t to = (uint32_t)(w.wy * fw), ao = (uint32_t)(ay * fw + ax);  // a plane is far below 2^32 bytes

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    uint32_t m[ND], v[NW + 1];
    load_row<ND, TOP>(trk, to + (uint32_t)(t * fw), a0, fw, m);
    if (t < B) load_anchor_row<NW>(anc + (ao + (uint32_t)(t * fw)), a[t < B ? t : 0]);
#pragma unroll
    for (int k = 0; k <= NW; ++k) v[k] = __builtin_amdgcn_alignbyte(m[k + 1], m[k], sh);
#pragma unroll
    for (int d = 0; d < NDY; ++d) {
      const int r = t - d;  // anchor row that meets tracked row t at vertical offset d
      if (r >= 0 && r < B) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          const uint32_t av = a[r >= 0 && r < B ? r : 0][k];
          acc4[d] = __builtin_amdgcn_qsad_pk_u16_u8(pack64(v[k], v[k + 1]), av, acc4[d]);
          if (RT == 2) acc1[d] = __builtin_amdgcn_sad_u8(v[k + 1], av, acc1[d]);
        }
      }
    }
  }
  select<RT, TOP, SHIFT>(
      w, ax, ay,
      [&](int d, int j) {
        return j < 4 ? (uint32_t)(acc4[d] >> (16 * (j & 3))) & 0xFFFFu : acc1[d];
      },
      mvx, mvy, best);
}

// Top level of a 4-level pyramid: 2x2 blocks (reference motion.cpp:719-720).  Two
// bytes per anchor row do not fill a QSAD word, so this level uses v_sad_u8 on
// 16-bit slices; it is 1/64 of the pixels of level 0.  Loading and searching are
// separate steps so that a caller can put other loads between them.
template <int RT>
struct TopB2 {
  static constexpr int NT = 2 + 2 * RT;
  Window w;
  uint32_t m[NT][3];
  uint32_t a[2];
};

template <int RT>
__device__ __forceinline__ void load_top_b2(const uint8_t* __restrict__ trk, const uint8_t* __restrict__ anc, int fw,
                                            int fh, int bx, int by, TopB2<RT>& s) {
  constexpr int B = 2, NT = TopB2<RT>::NT;
  const int ax = bx * B, ay = by * B;
  s.w = make_window<B, RT>(ax, ay, fw, fh);
  const int a0 = s.w.wx & ~3;
#pragma unroll
  for (int r = 0; r < B; ++r)
    s.a[r] = *reinterpret_cast<const uint16_t*>(anc + (uint32_t)((ay + r) * fw + ax));
  const uint32_t to = (uint32_t)(s.w.wy * fw);
#pragma unroll
  for (int t = 0; t < NT; ++t) load_row<3, true>(trk, to + (uint32_t)(t * fw), a0, fw, s.m[t]);
}

template <int RT, int SHIFT>
__device__ __forceinline__ void search_top_b2(const TopB2<RT>& s, int bx, int by, int& mvx, int& mvy, uint32_t& best) {
  constexpr int B = 2, NDY = 2 * RT + 1, NT = TopB2<RT>::NT;
  const uint32_t sh = (uint32_t)(s.w.wx & 3);
  uint32_t sad[NDY][NDY];
#pragma unroll
  for (int d = 0; d < NDY; ++d)
#pragma unroll
    for (int j = 0; j < NDY; ++j) sad[d][j] = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    // window bytes 0 .. 2RT+1 (<= 6) as two dwords starting at the window origin
    const uint32_t v0 = __builtin_amdgcn_alignbyte(s.m[t][1], s.m[t][0], sh);
    const uint32_t v1 = __builtin_amdgcn_alignbyte(s.m[t][2], s.m[t][1], sh);
    uint32_t tj[NDY];
#pragma unroll
    for (int j = 0; j < NDY; ++j)
      tj[j] = (j < 4 ? __builtin_amdgcn_alignbyte(v1, v0, j) : v1 >> (8 * (j - 4))) & 0xFFFFu;
#pragma unroll
    for (int d = 0; d < NDY; ++d) {
      const int r = t - d;
      if (r >= 0 && r < B) {
#pragma unroll
        for (int j = 0; j < NDY; ++j)
          sad[d][j] = __builtin_amdgcn_sad_u8(tj[j], s.a[r >= 0 && r < B ? r : 0], sad[d][j]);
      }
    }
  }
  select<RT, true, SHIFT>(s.w, bx * B, by * B, [&](int d, int j) { return sad[d][j]; }, mvx, mvy, best);
}

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

// ---------------------------------------------------------------------------------------------
// LDS-tiled form for the 4-level search (the reference's default build, libs/motion.cpp:691-749).
//
// Why.  With R_top = 1 the per-lane form above issues two vector loads per tracked row for 12 QSADs, and the 4-level
// search on fine texture returns an incoherent field (tools: a strip of 60 neighbouring blocks spans mv.y over +-8 at
// C5 and C3b alike), so every lane of a wave instruction walks its own row: measured on gfx950 (tools/ubench_tcp.hip)
// such an instruction holds the vector L1 for ~35 cycles whatever its width (x1, x2, x4 alike: ~0.55 cycles per lane
// address), against 16 cycles per KiB for whole rows.  The vector L1, not HBM, then paces the kernel (84 % busy).
//
// What.  A workgroup owns a tile of TBX x TBY MV blocks.  Where a block's window can lie at level l is bounded without
// knowing any vector: |mv_in| <= M_l = 2 (M_{l+1} + R_top), M_top = 0 (libs/motion.cpp:458-463), so the union of the
// tile's windows at levels 2, 1 and 0 is three rectangles whose position depends on blockIdx alone.  All three are
// brought into LDS as whole 16-byte chunks of whole rows by LDS-DMA (global_load_lds_dwordx4: no registers, one KiB
// per wave instruction, every load of the workgroup in flight at once), the anchor rows of every level go to registers
// (a wave's anchor row is one contiguous run), the 2x2 top level is searched from global memory while the tiles land,
// then one barrier, then levels 2, 1, 0 read their windows from LDS at per-lane addresses: the vector L1 sees only
// whole-row traffic.  Arithmetic, candidate order and tie rules are search_level's, so results are bit-identical.
// One workgroup per CU (117 KB of LDS): occupancy is not what hides latency here, the depth of the DMA queue is.
template <int B, int M, int RT, int TBX, int TBY>
struct TileGeom {
  static_assert(M + RT <= 16, "the tile starts 16 bytes left of its first anchor column");
  static constexpr int ND = B / 4 + 2;                                       // dwords read per tracked row
  static constexpr int X_LEFT = 16;                                          // bytes left of the first anchor column
  static constexpr int Y_TOP = M + RT;                                       // rows above the first anchor row
  static constexpr int A0_MAX = ((TBX - 1) * B + M - RT) & ~3;               // last dword-aligned window origin
  static constexpr int W = (A0_MAX + 4 * ND + X_LEFT + 15) & ~15;            // bytes per tile row (= LDS pitch)
  static constexpr int CPR = W / 16;                                         // 16-byte chunks per row
  static constexpr int ROWS = Y_TOP + (TBY - 1) * B + (M - RT) + B + 2 * RT;
  static constexpr int CHUNKS = CPR * ROWS;
  // LDS bytes: whole workgroup-wide DMA rounds (NWAVES x 1 KiB each), so that the fill is branch-free
  static constexpr int bytes(int nwaves) { return ((CHUNKS + 64 * nwaves - 1) / (64 * nwaves)) * 1024 * nwaves; }
};

// The tile of one level, global -> LDS.  Chunk i of the tile (row-major) is fetched by lane i % 64 of the wave
// instruction that covers chunks [i & ~63, +64): LDS-DMA writes a wave's 64 x 16 bytes contiguously from the wave-uniform
// base in M0.  Chunks that lie outside the plane are never read (a window is always inside it, make_window): their
// source address is clamped into the plane.  Row widths are multiples of 16 at every level (the frame is a multiple of
// 16 << (L-1)), so a chunk is either inside a row or outside.
template <class G, int NWAVES>
__device__ __forceinline__ void stage_tile(const uint8_t* __restrict__ plane, int fw, int fh, int x0, int y0,
                                           uint8_t* lds_tile, uint32_t wave, uint32_t lane) {
  // Straight-line on purpose (a tile's LDS is sized in whole rounds; lanes past the last chunk fetch it again): with
  // no branch between the loads the compiler's vmcnt bookkeeping stays exact and the top-level search, whose loads
  // were issued first, does not wait for the tiles.
#pragma unroll
  for (int base = 0; base < G::CHUNKS; base += 64 * NWAVES) {
    const uint32_t wbase = (uint32_t)base + wave * 64u;
    const uint32_t i = min(wbase + lane, (uint32_t)(G::CHUNKS - 1));
    const uint32_t row = i / (uint32_t)G::CPR, c = i - row * (uint32_t)G::CPR;
    const int gy = min(max(y0 + (int)row, 0), fh - 1);
    const int gx = min(max(x0 + 16 * (int)c, 0), fw - 16);
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(plane + ((uint32_t)gy * (uint32_t)fw + (uint32_t)gx)),
        (__attribute__((address_space(3))) void*)(lds_tile + wbase * 16u), 16, 0, 0);
  }
}

// search_level with the tracked window read from the level's LDS tile and the anchor block already in registers.
template <int B, int RT, int SHIFT, class G>
__device__ __forceinline__ void search_level_lds(const uint8_t* lds_tile, int x0, int y0, const uint32_t (&a)[B][B / 4],
                                                 int fw, int fh, int bx, int by, int& mvx, int& mvy, uint32_t& best) {
  constexpr int NW = B / 4, ND = NW + 2, NDY = 2 * RT + 1, NT = B + 2 * RT;
  const int ax = bx * B, ay = by * B;
  const Window w = make_window<B, RT>(ax + mvx, ay + mvy, fw, fh);
  const int a0 = w.wx & ~3;
  const uint32_t sh = (uint32_t)(w.wx & 3);
  const uint8_t* p = lds_tile + ((w.wy - y0) * G::W + (a0 - x0));

  uint64_t acc4[NDY];
  uint32_t acc1[NDY];
#pragma unroll
  for (int d = 0; d < NDY; ++d) { acc4[d] = 0; acc1[d] = 0; }
  // one wave per SIMD: the whole window is requested before the first SAD so that the LDS latency is paid once
  uint32_t m[NT][ND];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int k = 0; k < ND; ++k) m[t][k] = *reinterpret_cast<const uint32_t*>(p + (t * G::W + 4 * k));
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    uint32_t v[NW + 1];
#pragma unroll
    for (int k = 0; k <= NW; ++k) v[k] = __builtin_amdgcn_alignbyte(m[t][k + 1], m[t][k], sh);
#pragma unroll
    for (int d = 0; d < NDY; ++d) {
      const int r = t - d;
      if (r >= 0 && r < B) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          const uint32_t av = a[r >= 0 && r < B ? r : 0][k];
          acc4[d] = __builtin_amdgcn_qsad_pk_u16_u8(pack64(v[k], v[k + 1]), av, acc4[d]);
          if (RT == 2) acc1[d] = __builtin_amdgcn_sad_u8(v[k + 1], av, acc1[d]);
        }
      }
    }
  }
  select<RT, false, SHIFT>(
      w, ax, ay,
      [&](int d, int j) {
        return j < 4 ? (uint32_t)(acc4[d] >> (16 * (j & 3))) & 0xFFFFu : acc1[d];
      },
      mvx, mvy, best);
}

template <int B>
__device__ __forceinline__ void load_anchor_block(const uint8_t* __restrict__ anc, int fw, int bx, int by,
                                                  uint32_t (&a)[B][B / 4]) {
  const uint32_t ao = (uint32_t)(by * B * fw + bx * B);
#pragma unroll
  for (int t = 0; t < B; ++t) load_anchor_row<B / 4>(anc + (ao + (uint32_t)(t * fw)), a[t]);
}

template <int RT, int TBX, int TBY>
__global__ __launch_bounds__(TBX* TBY) void hbma_tile16_kernel(FusedArgs a) {
  constexpr int NWAVES = TBX * TBY / 64;
  using G0 = TileGeom<16, 2 * (2 * (2 * RT + RT) + RT), RT, TBX, TBY>;  // M = 14 RT
  using G1 = TileGeom<8, 2 * (2 * RT + RT), RT, TBX, TBY>;             // M = 6 RT
  using G2 = TileGeom<4, 2 * RT, RT, TBX, TBY>;                        // M = 2 RT
  __shared__ __attribute__((aligned(16))) uint8_t lds[G0::bytes(NWAVES) + G1::bytes(NWAVES) + G2::bytes(NWAVES)];
  uint8_t* const t0 = lds;
  uint8_t* const t1 = lds + G0::bytes(NWAVES);
  uint8_t* const t2 = lds + G0::bytes(NWAVES) + G1::bytes(NWAVES);

  // Region-major order as in hbma_fused16_kernel: XCD x is given the x-th eighth of the tiles of EVERY pair.
  const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
  const uint32_t pair = k / a.wgs_per_region;
  const uint32_t tile = xcd * a.wgs_per_region + (k - pair * a.wgs_per_region);
  const uint32_t tiles_x = (a.mfw + TBX - 1) / TBX, mfh = a.blocks / a.mfw;
  const uint32_t tm = tile / tiles_x, tk = tile - tm * tiles_x;
  if (pair >= a.n_pairs || tm * TBY >= mfh) return;  // uniform over the workgroup

  const uint32_t tid = threadIdx.x, wave = tid / 64u, lane = tid & 63u;
  const uint32_t lx = tid % TBX, ly = tid / TBX;
  const uint32_t bxu = tk * TBX + lx, byu = tm * TBY + ly;
  const bool live = bxu < a.mfw && byu < mfh;
  // a lane beyond the frame searches the last block of its row / column again (in-tile addresses) and stores nothing
  const int bx = (int)min(bxu, a.mfw - 1), by = (int)min(byu, mfh - 1);

  const uint8_t* trk = a.tracked + (size_t)pair * a.pair_stride;
  const uint8_t* anc = a.anchor + (size_t)pair * a.pair_stride;
  const int w = (int)a.w, h = (int)a.h;
  const size_t o1 = (size_t)w * h, o2 = o1 + (o1 >> 2), o3 = o2 + (o1 >> 4);

  // the top level's 2x2 blocks and windows, then the anchor blocks of every level: registers (a wave's anchor row is
  // one contiguous run).  The top level is loaded first, so that its search waits for nothing issued behind it.
  TopB2<RT> top;
  load_top_b2<RT>(trk + o3, anc + o3, w >> 3, h >> 3, bx, by, top);
  uint32_t a0r[16][4], a1r[8][2], a2r[4][1];
  load_anchor_block<4>(anc + o2, w >> 2, bx, by, a2r);
  load_anchor_block<8>(anc + o1, w >> 1, bx, by, a1r);
  load_anchor_block<16>(anc, w, bx, by, a0r);

  // tracked tiles, coarse to fine: completion order is issue order
  const int tx = (int)(tk * TBX), ty = (int)(tm * TBY);
  const int x2 = tx * 4 - G2::X_LEFT, y2 = ty * 4 - G2::Y_TOP;
  const int x1 = tx * 8 - G1::X_LEFT, y1 = ty * 8 - G1::Y_TOP;
  const int x0 = tx * 16 - G0::X_LEFT, y0 = ty * 16 - G0::Y_TOP;
  stage_tile<G2, NWAVES>(trk + o2, w >> 2, h >> 2, x2, y2, t2, wave, lane);
  stage_tile<G1, NWAVES>(trk + o1, w >> 1, h >> 1, x1, y1, t1, wave, lane);
  stage_tile<G0, NWAVES>(trk, w, h, x0, y0, t0, wave, lane);

  int mvx = 0, mvy = 0;
  uint32_t best = 0;
  search_top_b2<RT, 6>(top, bx, by, mvx, mvy, best);

  __syncthreads();  // every wave's DMA has landed (the compiler drains vmcnt in front of the barrier)

  mvx *= 2; mvy *= 2;  // motion.cpp:458-460
  search_level_lds<4, RT, 4, G2>(t2, x2, y2, a2r, w >> 2, h >> 2, bx, by, mvx, mvy, best);
  mvx *= 2; mvy *= 2;
  search_level_lds<8, RT, 2, G1>(t1, x1, y1, a1r, w >> 1, h >> 1, bx, by, mvx, mvy, best);
  mvx *= 2; mvy *= 2;
  search_level_lds<16, RT, 0, G0>(t0, x0, y0, a0r, w, h, bx, by, mvx, mvy, best);

  if (live) {
    const uint32_t item = pair * a.blocks + byu * a.mfw + bxu;
    reinterpret_cast<float2*>(a.mv)[item] = make_float2((float)mvx, (float)mvy);
    a.mad[item] = (float)best * (1.0f / 256.0f);
  }
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

// The LDS-tiled kernel serves the 4-level search with R_top = 1 on planes whose rows are whole 16-byte chunks at the
// three levels it stages (so the frame width is a multiple of 64; any height).
bool tile_supported(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw, uint32_t bh) {
  return fused_supported(levels, w, h, range, bw, bh) && levels == 4 && (range >> 3) == 1 && w % 64 == 0;
}

constexpr int kTileBX = 32, kTileBY = 8;

static int launch_hbma_tile(FusedArgs a, uint32_t n_pairs, hipStream_t stream) {
  const uint32_t mfh = a.blocks / a.mfw;
  const uint32_t tiles = div_up(a.mfw, kTileBX) * div_up(mfh, kTileBY);
  a.wgs_per_region = div_up(tiles, 8);
  const uint64_t wgs = (uint64_t)8 * a.wgs_per_region * n_pairs;
  if (wgs > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "hbma: %llu workgroups exceed one launch", (unsigned long long)wgs);
  hipLaunchKernelGGL((hbma_tile16_kernel<1, kTileBX, kTileBY>), dim3((uint32_t)wgs), dim3(kTileBX * kTileBY), 0, stream, a);
  return check_launch("hbma_tile16_kernel");
}

// kernel: 0 = the shape's default, 1 = lane-per-block (no LDS), 2 = LDS-tiled (UNSUPPORTED where tile_supported is false)
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
  const bool can_tile = tile_supported(levels, w, h, range, 16, 16) && ((uintptr_t)d_tracked % 16 == 0) && ((uintptr_t)d_anchor % 16 == 0) &&
                        pair_stride % 16 == 0;
  if (kernel == 2 && !can_tile)
    return fail(SVC_ERR_UNSUPPORTED, "hbma: the LDS-tiled kernel needs 4 levels, r_top 1, a frame width that is a multiple of 64 and 16-byte aligned pyramids");
  if (kernel == 2 || (kernel == 0 && can_tile)) return launch_hbma_tile(a, n_pairs, stream);
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
