// hbma_search.hpp -- device code shared by the all-level motion search kernels (hbma_fused.hip: lane per block;
// hbma_tiled.hip: LDS-tiled): the SAD engine, the reference's window clamps and the candidate selection rules.
//
// SAD engine.  v_qsad_pk_u16_u8 returns, for one 4-byte anchor word, the four SADs
// against the tracked bytes at offsets 0..3 of an 8-byte window, accumulated as
// 4 x u16 (a 16x16 block's SAD <= 65280 fits).  Measured on gfx950 (tools/
// ubench_valu.hip): QSAD issues in 16 cycles per wave, v_sad_u8 / v_alignbyte_b32 /
// v_min3 in 4, i.e. 4 cycles per 4-byte SAD either way: each tracked row is
// funnel-shifted ONCE to the window origin (shared by every vertical offset), then one
// QSAD covers dx = 0..3 and one v_sad_u8 the fifth column.  Candidates outside the
// reference's clamped window (libs/motion.cpp:375-385) are masked at selection time.
//
// Arithmetic.  All block areas are powers of two, so MAD = sad / area is an exact
// dyadic rational; the MAD carried across levels (libs/motion.cpp:401 compares a
// level-l MAD with the level-(l+1) minimum) is kept as the integer sad << 2l
// (units of 1/256) and converted once at the end: bit-identical to the float path.
#pragma once

#include "svc_common.hpp"

namespace svc {

// what the lane-per-block kernel covers (hbma_fused.hip) / the LDS-tiled one (hbma_tiled.hip)
bool fused_supported(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw, uint32_t bh);
bool tiled_supported(uint32_t levels, uint32_t w, uint32_t h, uint32_t range, uint32_t bw, uint32_t bh);
// SVC_HBMA_AUTO takes the LDS-tiled kernel where it applies (16 x 16 blocks, 4 levels, R_top 1: the reference's default
// build): 2 - 4 % faster than the lane-per-block kernel on C5 and C3b once both use the packed refinement select and the
// tile shape is fitted to the frame (profiles/r03_ab_hbma_tiled_final.txt)
constexpr bool kTiledIsDefault = true;

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
  if (NW % 4 == 0) {
#pragma unroll
    for (int q = 0; q < NW / 4; ++q) {
      u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p + 16 * q);
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
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
// unsigned min over packed keys  (scaled_sad << CB) | code  (CB = 5 bits up to 25 candidates, 7 up to 81):
//   refinement (motion.cpp:401, strict `<` against the carried minimum): code = raster
//     index, so equal SADs resolve to the FIRST candidate; the winner replaces the carried
//     value only if its scaled SAD is strictly smaller;
//   top level (motion.cpp:324-337, `<=`): code = 2^CB - 1 - index, so equal SADs resolve to the
//     LAST candidate; and if the valid SADs are non-increasing in raster order every
//     candidate "updated" and the MV is zeroed (the minimum is kept).
// Candidates outside the reference's clamped window get the all-ones key.  A scaled SAD is at most
// 255 * (MV block area) <= 2^18 (32 x 32 blocks), so the key fits 32 bits.
template <int RT, bool TOP, int SHIFT, typename GetSad>
__device__ __forceinline__ void select(const Window& w, int ax, int ay, GetSad sad_at, int& mvx,
                                       int& mvy, uint32_t& best) {
  constexpr int N = 2 * RT + 1;
  constexpr int CB = N * N <= 32 ? 5 : 7;
  constexpr uint32_t CM = (1u << CB) - 1u;
  static_assert(N * N <= 128, "raster index must fit the 7-bit code");
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
      const uint32_t key = (s << (SHIFT + CB)) | (uint32_t)(TOP ? (int)CM - idx : idx);
      kmin = min(kmin, valid ? key : 0xFFFFFFFFu);
      if (TOP) {
        mono = mono && (!valid || s <= prev);
        prev = valid ? s : prev;
      }
    }
  }
  const uint32_t smin = kmin >> CB;  // scaled SAD of the winner
  const int idx = TOP ? (int)CM - (int)(kmin & CM) : (int)(kmin & CM);
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

// The refinement select (strict `<` against the carried minimum, first candidate in raster order on ties) straight on the
// packed QSAD sums, for blocks whose SAD fits 16 bits (B <= 16).  Inside a level the scale of the SADs does not matter,
// so a candidate's key is (sad << 16) | raster index: ONE instruction per candidate -- v_lshl_or_b32 for the sum in the
// low half of its word, v_and_or_b32 for the one in the high half -- instead of extract, shift and or; only the winner
// is scaled (<< SHIFT) for the compare with the carried minimum.  Where every lane of the wave has its whole candidate
// grid inside the reference's clamped window (every block away from the frame border whose vector stays inside: almost
// all of them) the validity masks are skipped too; the other waves take the masked form.  Same order, same winner.
// Used for R_top = 1 (9 candidates); see search_level for why not beyond.
template <int RT, int SHIFT, int NQ>
__device__ __forceinline__ void select_refine_packed(const Window& w, int ax, int ay, const uint64_t (&acc4)[2 * RT + 1][NQ],
                                                     const uint32_t (&acc1)[2 * RT + 1], int& mvx, int& mvy, uint32_t& best) {
  constexpr int N = 2 * RT + 1;
  constexpr uint32_t CM = 0xFFFFu;
  uint32_t key[N][N];
#pragma unroll
  for (int d = 0; d < N; ++d)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const uint32_t code = (uint32_t)(d * N + j);
      if (j >= 4 * NQ) {
        key[d][j] = (acc1[d] << 16) | code;
      } else {
        const uint32_t word = (j & 2) ? (uint32_t)(acc4[d][j >> 2] >> 32) : (uint32_t)acc4[d][j >> 2];
        key[d][j] = (j & 1) ? ((word & 0xFFFF0000u) | code) : ((word << 16) | code);
      }
    }
  const bool whole = w.dlo == 0 && w.dhi == N && w.jlo == 0 && w.jhi == N;
  uint32_t kmin = 0xFFFFFFFFu;
  if (__builtin_amdgcn_ballot_w64(!whole) == 0) {  // wave-uniform: no lane needs a mask
#pragma unroll
    for (int d = 0; d < N; ++d)
#pragma unroll
      for (int j = 0; j < N; ++j) kmin = min(kmin, key[d][j]);
  } else {
#pragma unroll
    for (int d = 0; d < N; ++d) {
      const bool row_ok = d >= w.dlo && d < w.dhi;
#pragma unroll
      for (int j = 0; j < N; ++j) kmin = min(kmin, (row_ok && j >= w.jlo && j < w.jhi) ? key[d][j] : 0xFFFFFFFFu);
    }
  }
  const uint32_t smin = (kmin >> 16) << SHIFT;  // no candidate at all: 0xFFFF << SHIFT, above every carried minimum
  const int idx = (int)(kmin & CM);
  const int bd = idx / N, bj = idx - bd * N;
  if (smin < best && kmin != 0xFFFFFFFFu) {
    best = smin;
    mvx = w.wx + bj - ax;
    mvy = w.wy + bd - ay;
  }
}

// How the 2 RT + 1 horizontal candidates of a level map onto the SAD instructions: NQ v_qsad_pk_u16_u8 per anchor word
// (four candidates each) and, when one candidate is left over (RT = 2, 4), one v_sad_u8.
template <int RT>
struct SadPlan {
  static constexpr int N = 2 * RT + 1;
  static constexpr bool kTail = (N % 4) == 1;          // RT 2, 4: the last column by v_sad_u8
  static constexpr int NQ = kTail ? N / 4 : (N + 3) / 4;  // RT 1: 1, RT 2: 1 (+ tail), RT 3: 2, RT 4: 2 (+ tail)
};

// One level with block size B >= 4.  Per tracked row: NW + NQ + 1 aligned dwords are loaded
// and funnel-shifted once (v_alignbyte_b32) so that word k starts at window byte 4k; then
// for every anchor row that meets it, per anchor word: NQ v_qsad_pk_u16_u8 (candidates
// dx = 4q .. 4q + 3) and, for RT = 2 / 4, one v_sad_u8 (the last column).  The packed 16-bit
// sums of a QSAD hold 256 byte differences: a 32-pixel-wide block spills them to 32 bits
// every 8 anchor rows.
template <int B, int RT, bool TOP, int SHIFT>
__device__ __forceinline__ void search_level(const uint8_t* __restrict__ trk,
                                             const uint8_t* __restrict__ anc, int fw, int fh,
                                             int bx, int by, int& mvx, int& mvy, uint32_t& best) {
  using P = SadPlan<RT>;
  constexpr int NW = B / 4, NQ = P::NQ, NV = NW + NQ, ND = NV + 1, NDY = 2 * RT + 1, NT = B + 2 * RT;
  constexpr bool WIDE = B * B > 256;  // a block's SAD can exceed 16 bits
  constexpr int FLUSH = 256 / B;      // anchor rows per 16-bit accumulation run
  const int ax = bx * B, ay = by * B;
  const Window w = make_window<B, RT>(ax + mvx, ay + mvy, fw, fh);
  const int a0 = w.wx & ~3;
  const uint32_t sh = (uint32_t)(w.wx & 3);

  uint64_t acc4[NDY][NQ];
  uint32_t acc1[NDY];
  uint32_t wide[WIDE ? NDY : 1][WIDE ? 4 * NQ : 1];
#pragma unroll
  for (int d = 0; d < NDY; ++d) {
    acc1[d] = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc4[d][q] = 0;
    if (WIDE)
#pragma unroll
      for (int j = 0; j < 4 * NQ; ++j) wide[d][j] = 0;
  }
  uint32_t a[B][NW];
  const uint32_t to = (uint32_t)(w.wy * fw), ao = (uint32_t)(ay * fw + ax);  // a plane is far below 2^32 bytes

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    uint32_t m[ND], v[NV];
    load_row<ND, TOP>(trk, to + (uint32_t)(t * fw), a0, fw, m);
    if (t < B) load_anchor_row<NW>(anc + (ao + (uint32_t)(t * fw)), a[t < B ? t : 0]);
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
        if (WIDE && (r + 1) % FLUSH == 0) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int f = 0; f < 4; ++f) wide[d][4 * q + f] += (uint32_t)(acc4[d][q] >> (16 * f)) & 0xFFFFu;
            acc4[d][q] = 0;
          }
        }
      }
    }
  }
  // 9 candidates: the packed form wins 3 % on the 4-level search; with 25 it LOSES 1.5 % on the VALU-balanced 3-level
  // kernel (keys held across the branch, or re-formed in both arms: both measured, profiles/r03_ab_select.txt)
  if constexpr (!TOP && !WIDE && RT == 1) {
    select_refine_packed<RT, SHIFT, NQ>(w, ax, ay, acc4, acc1, mvx, mvy, best);
  } else {
    select<RT, TOP, SHIFT>(
        w, ax, ay,
        [&](int d, int j) {
          if (j >= 4 * NQ) return acc1[d];
          return WIDE ? wide[WIDE ? d : 0][WIDE ? j : 0] : (uint32_t)(acc4[d][j >> 2] >> (16 * (j & 3))) & 0xFFFFu;
        },
        mvx, mvy, best);
  }
}

// A top level of 2x2 blocks (reference motion.cpp:719-720: the 4-level search of 16x16 blocks; likewise 3 levels
// of 8x8, 5 of 32x32).  Two bytes per anchor row do not fill a QSAD word, so this level uses v_sad_u8 on
// 16-bit slices; it is 1/64 of the pixels of level 0.  Loading and searching are
// separate steps so that a caller can put other loads between them.
template <int RT>
struct TopB2 {
  static constexpr int NT = 2 + 2 * RT;
  static constexpr int ND = RT <= 2 ? 3 : 4;  // dwords that hold the 2 RT + 2 window bytes at any alignment
  Window w;
  uint32_t m[NT][ND];
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
  if ((fw & 3) == 0) {  // wave-uniform
#pragma unroll
    for (int t = 0; t < NT; ++t) load_row<TopB2<RT>::ND, true>(trk, to + (uint32_t)(t * fw), a0, fw, s.m[t]);
  } else {
    // Rows that are not whole dwords (a frame 16 mod 32 pixels wide at 4 levels: PAL's 720 -> 90, QCIF's 176 -> 22): row
    // starts are not dword-aligned in memory and the in-row clamp above would shift the bytes it is meant to protect.  The
    // dwords are read where they lie (unaligned loads; a dword past the row's end simply continues into the next row), and
    // only the PLANE's end is guarded: a load that would cross it re-reads the plane's last four bytes and shifts them down
    // -- the bytes that fall off are past the plane and belong to no window.
    const uint32_t last = (uint32_t)(fw * fh) - 4u;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < TopB2<RT>::ND; ++k) {
        const uint32_t off = to + (uint32_t)(t * fw) + (uint32_t)(a0 + 4 * k), c = min(off, last), sh = off - c;
        typedef uint32_t u32_a1 __attribute__((aligned(1)));
        const uint32_t raw = *reinterpret_cast<const u32_a1*>(trk + c);
        s.m[t][k] = sh >= 4u ? 0u : raw >> (8u * sh);
      }
  }
}

template <int RT, int SHIFT>
__device__ __forceinline__ void search_top_b2(const TopB2<RT>& s, int bx, int by, int& mvx, int& mvy, uint32_t& best) {
  constexpr int B = 2, NDY = 2 * RT + 1, NT = TopB2<RT>::NT, NV = TopB2<RT>::ND - 1;
  const uint32_t sh = (uint32_t)(s.w.wx & 3);
  uint32_t sad[NDY][NDY];
#pragma unroll
  for (int d = 0; d < NDY; ++d)
#pragma unroll
    for (int j = 0; j < NDY; ++j) sad[d][j] = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    // window bytes 0 .. 2RT+1 as dwords starting at the window origin
    uint32_t v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = __builtin_amdgcn_alignbyte(s.m[t][k + 1], s.m[t][k], sh);
    uint32_t tj[NDY];
#pragma unroll
    for (int j = 0; j < NDY; ++j) {  // the 16-bit slice at window byte j
      const int q = j >> 2, o = j & 3;
      tj[j] = (o < 3 ? v[q] >> (8 * o) : __builtin_amdgcn_alignbyte(v[q + 1 < NV ? q + 1 : q], v[q], 3)) & 0xFFFFu;
    }
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

}  // namespace svc
