// hbma_wave.hip -- one wavefront per MV block, one launch per pyramid level.
//
// The shape-agnostic motion search (any block size, any search range, any level
// count): the kernel behind EstimateMotionExhaustiveSearch (reference
// libs/motion.cpp:268-340) and, level by level, behind EstimateMotionHierarchical
// (:412-465) for shapes the fused kernel (hbma_fused.hip) does not cover.
//
// A workgroup is one 64-lane wave.  Its lanes are cut into groups of 4 .. 64 -- the smallest power
// of two that holds a level's lane tasks (four candidates each on the dword path) -- and every group searches its own MV
// block (25 candidates would otherwise leave 39 of 64 lanes idle, 9 candidates 55); a wave then
// walks several such rounds, because at the coarse levels (4x4, 2x2 blocks) a launch of one tiny
// workgroup per block is bound by the workgroup launch rate, not by its arithmetic.
// Per MV block (one group of lanes):
//   1. the anchor block and the clamped search window of the tracked plane are
//      staged in LDS with coalesced dword loads (window origin aligned down to 4);
//   2. the candidate grid is dealt across lanes four horizontally adjacent candidates at a
//      time, in raster order; a lane walks their rows with v_qsad_pk_u16_u8 (one anchor word
//      against 8 window bytes = the SADs of all four), the window having been funnel-shifted
//      to a dword boundary when it was staged;
//   3. a group-wide min over packed (sad, index) keys (xor-shuffles below the group size never
//      leave the group) gives the argmin with the
//      reference's tie rule: top level `<=` => LAST raster minimum (:324),
//      refinement strict `<` => FIRST raster minimum that beats the MAD carried
//      from the coarser level (:401);
//   4. top level only: if the SADs are non-increasing in raster order every
//      candidate "updated" and the reference zeroes the MV (:333-337).
// Integer SAD order == float MAD order within a level (sad < 2^23, one correctly
// rounded divide), so the argmin is done on integers and the single float compare
// against the carried MAD uses the reference's own expression (float)sad / count.
#include "svc_common.hpp"

namespace svc {

struct WaveLevelArgs {
  const uint8_t* tracked;
  const uint8_t* anchor;
  uint64_t pair_stride;
  uint64_t level_off;   // byte offset of this level's plane inside a packed pyramid
  uint32_t blocks;      // MV blocks per pair
  uint32_t mfw;         // motion-field width
  uint32_t fw, fh;      // plane size at this level
  uint32_t bw, bh;      // block size at this level
  uint32_t range;       // search range at this level
  uint32_t a_pitch;     // LDS row pitch of the anchor block (bytes, multiple of 4)
  uint32_t w_pitch;     // LDS row pitch of the window
  uint32_t sads_off;    // LDS byte offset of the per-candidate SAD array
  uint32_t top;         // 1 = EBMA semantics, 0 = refinement semantics
  uint32_t gs;          // lanes per group: a power of two, 4 .. 64
  uint32_t rounds;      // rounds of 64 / gs blocks a wave walks
  uint32_t lds_group;   // LDS bytes per group
  uint32_t n_items;     // pairs * blocks
  float* mv;            // [pairs][blocks][2]
  float* mad;           // [pairs][blocks]
};

// min over the gs-lane group the lane belongs to (gs a power of two: lane ^ off stays inside it)
__device__ __forceinline__ uint64_t group_min_u64(uint64_t v, uint32_t gs) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    if ((uint32_t)off >= gs) continue;
    uint32_t lo = __shfl_xor((uint32_t)v, off, 64);
    uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
    uint64_t o = ((uint64_t)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}

// NWT: dwords per block row when the block width is 4, 8, 16 or 32 (inner loops unrolled, LDS reads in flight
// together), 0 = any width.
template <bool DW, int NWT>
__global__ __launch_bounds__(64) void hbma_wave_level_kernel(WaveLevelArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
  const uint32_t gs = a.gs, groups = 64u / gs;
  const uint32_t sub = threadIdx.x / gs, lane = threadIdx.x - sub * gs;  // group and lane inside it
  uint8_t* lds = lds_all + (size_t)sub * a.lds_group;
  uint8_t* lds_anchor = lds;
  uint8_t* lds_win = lds + (size_t)a.bh * a.a_pitch;
  uint32_t* lds_sads = reinterpret_cast<uint32_t*>(lds + a.sads_off);
  const unsigned long long gmask = (gs == 64 ? ~0ull : ((1ull << gs) - 1ull)) << (sub * gs);

  for (uint32_t round = 0; round < a.rounds; ++round) {
  const uint32_t item = (blockIdx.x * a.rounds + round) * groups + sub;  // pair * blocks + block
  const bool live = item < a.n_items;  // a dead group keeps pace with the barriers and shuffles, on item 0's data
  const uint32_t it = live ? item : 0u;
  const uint32_t pair = it / a.blocks;
  const uint32_t blk = it - pair * a.blocks;
  const uint32_t by = blk / a.mfw, bx = blk - by * a.mfw;
  const uint32_t ax = bx * a.bw, ay = by * a.bh;

  const uint8_t* trk = a.tracked + pair * a.pair_stride + a.level_off;
  const uint8_t* anc = a.anchor + pair * a.pair_stride + a.level_off;
  float* mv = a.mv + ((size_t)it) * 2;
  float* mad = a.mad + it;

  // centre of the search (libs/motion.cpp:372-373; the MV was doubled at :458-460)
  int mvx = 0, mvy = 0;
  float carried = 0.f;
  if (!a.top && live) {  // a dead group searches around (0, 0): always a valid window
    mvx = 2 * (int)roundf(mv[0]);
    mvy = 2 * (int)roundf(mv[1]);
    carried = *mad;
  }
  const uint32_t cx = (uint32_t)((int)ax + mvx), cy = (uint32_t)((int)ay + mvy);
  // window bounds (:297-310 / :375-385)
  const int lo_x = (int)cx - (int)a.range, lo_y = (int)cy - (int)a.range;
  const uint32_t x0 = lo_x < 0 ? 0u : (uint32_t)lo_x;
  const uint32_t y0 = lo_y < 0 ? 0u : (uint32_t)lo_y;
  const uint32_t x1 = min(a.fw - a.bw + 1u, cx + a.range + 1u);
  const uint32_t y1 = min(a.fh - a.bh + 1u, cy + a.range + 1u);
  const uint32_t nx = x1 - x0, ny = y1 - y0, ncand = nx * ny;
  const uint32_t win_rows = ny + a.bh - 1;

  uint32_t xshift = 0;  // byte offset of column x0 inside a staged window row
  if (DW) {
    const uint32_t x0a = x0 & ~3u;
    xshift = x0 - x0a;
    const uint32_t a_dw = a.a_pitch >> 2, w_dw = a.w_pitch >> 2;
    for (uint32_t i = lane; i < a.bh * a_dw; i += gs) {
      uint32_t r = i / a_dw, c = i - r * a_dw;
      reinterpret_cast<uint32_t*>(lds_anchor)[i] =
          *reinterpret_cast<const uint32_t*>(anc + (size_t)(ay + r) * a.fw + ax + 4 * c);
    }
    for (uint32_t i = lane; i < win_rows * w_dw; i += gs) {
      uint32_t r = i / w_dw, c = i - r * w_dw;
      // funnel-shifted once here so that column x0 sits at byte 0 of the staged row (the QSADs below
      // want their windows dword-aligned); columns past the last needed byte are don't-care
      const uint8_t* row = trk + (size_t)(y0 + r) * a.fw;
      const uint32_t lo = *reinterpret_cast<const uint32_t*>(row + min(x0a + 4 * c, a.fw - 4u));
      const uint32_t hi = *reinterpret_cast<const uint32_t*>(row + min(x0a + 4 * c + 4, a.fw - 4u));
      reinterpret_cast<uint32_t*>(lds_win)[i] = __builtin_amdgcn_alignbyte(hi, lo, xshift);
    }
  } else {
    for (uint32_t i = lane; i < a.bh * a.bw; i += gs) {
      uint32_t r = i / a.bw, c = i - r * a.bw;
      lds_anchor[r * a.a_pitch + c] = anc[(size_t)(ay + r) * a.fw + ax + c];
    }
    const uint32_t wcols = nx + a.bw - 1;
    for (uint32_t i = lane; i < win_rows * wcols; i += gs) {
      uint32_t r = i / wcols, c = i - r * wcols;
      lds_win[r * a.w_pitch + c] = trk[(size_t)(y0 + r) * a.fw + x0 + c];
    }
  }
  __syncthreads();

  uint32_t best_sad = 0xFFFFFFFFu, best_idx = 0;
  if (DW) {
    // a lane takes FOUR horizontally adjacent candidates of one row of the grid: v_qsad_pk_u16_u8 gives, for one
    // anchor word and 8 window bytes, the SADs at byte offsets 0..3 (16 issue cycles for 16 byte-differences:
    // half of align + v_sad_u8 per candidate, and a quarter of the LDS reads)
    const uint32_t ngx = (nx + 3) >> 2, ntasks = ny * ngx;
    const uint32_t a_dw = a.a_pitch >> 2, w_dw = a.w_pitch >> 2, nwords = NWT ? (uint32_t)NWT : a.bw >> 2;
    const uint32_t* a32 = reinterpret_cast<const uint32_t*>(lds_anchor);
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(lds_win);
    const uint32_t flush_rows = max(1u, 256u / a.bw);  // packed u16 accumulators hold 256 byte-differences
    for (uint32_t t = lane; t < ntasks; t += gs) {
      const uint32_t iy = t / ngx, gxi = t - iy * ngx;
      uint32_t s4[4] = {0, 0, 0, 0};
      uint64_t acc = 0;
      uint32_t since = 0;
      for (uint32_t r = 0; r < a.bh; ++r) {
        const uint32_t* wrow = w32 + (iy + r) * w_dw + gxi;
        if constexpr (NWT > 0) {
          uint32_t wv[NWT + 1], av[NWT > 0 ? NWT : 1];
#pragma unroll
          for (int k = 0; k <= NWT; ++k) wv[k] = wrow[k];
#pragma unroll
          for (int k = 0; k < NWT; ++k) av[k] = a32[r * a_dw + k];
#pragma unroll
          for (int k = 0; k < NWT; ++k) acc = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)wv[k + 1] << 32) | wv[k], av[k], acc);
        } else {
          uint32_t lo = wrow[0];
          for (uint32_t k = 0; k < nwords; ++k) {
            const uint32_t hi = wrow[k + 1];
            acc = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)hi << 32) | lo, a32[r * a_dw + k], acc);
            lo = hi;
          }
        }
        if (++since == flush_rows) {
          s4[0] += (uint32_t)acc & 0xFFFFu; s4[1] += (uint32_t)(acc >> 16) & 0xFFFFu;
          s4[2] += (uint32_t)(acc >> 32) & 0xFFFFu; s4[3] += (uint32_t)(acc >> 48);
          acc = 0; since = 0;
        }
      }
      s4[0] += (uint32_t)acc & 0xFFFFu; s4[1] += (uint32_t)(acc >> 16) & 0xFFFFu;
      s4[2] += (uint32_t)(acc >> 32) & 0xFFFFu; s4[3] += (uint32_t)(acc >> 48);
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t ix = 4 * gxi + j;
        if (ix >= nx) break;
        const uint32_t c = iy * nx + ix, sad = s4[j];  // raster index: ascending with t, then j
        lds_sads[c] = sad;
        if (a.top ? (sad <= best_sad) : (sad < best_sad)) {
          best_sad = sad;
          best_idx = c;
        }
      }
    }
  } else {
    // each lane: candidates lane, lane+gs, ... in raster order
    for (uint32_t c = lane; c < ncand; c += gs) {
      const uint32_t iy = c / nx, ix = c - iy * nx;
      uint32_t sad = 0;
      for (uint32_t r = 0; r < a.bh; ++r)
        for (uint32_t k = 0; k < a.bw; ++k) {
          int d = (int)lds_anchor[r * a.a_pitch + k] - (int)lds_win[(iy + r) * a.w_pitch + ix + k];
          sad += (uint32_t)(d < 0 ? -d : d);
        }
      lds_sads[c] = sad;
      if (a.top ? (sad <= best_sad) : (sad < best_sad)) {
        best_sad = sad;
        best_idx = c;
      }
    }
  }
  __syncthreads();

  // group argmin: smaller sad first, then LAST index (top) / FIRST index (refine)
  uint64_t key = ((uint64_t)best_sad << 32) | (a.top ? ~best_idx : best_idx);
  key = group_min_u64(key, gs);
  const uint32_t w_sad = (uint32_t)(key >> 32);
  const uint32_t w_idx = a.top ? ~(uint32_t)key : (uint32_t)key;

  // top level: did every candidate update, i.e. are the SADs non-increasing?
  bool mono = true;
  if (a.top)
    for (uint32_t c = lane; c < ncand; c += gs)
      if (c > 0 && lds_sads[c] > lds_sads[c - 1]) mono = false;
  const bool all_updated = (__ballot(mono) & gmask) == gmask;

  if (lane == 0 && live) {
    const float m = (float)w_sad / (float)(a.bw * a.bh);  // libs/motion.cpp:38-40
    const uint32_t iy = w_idx / nx, ix = w_idx - iy * nx;
    float ox = (float)((int)(x0 + ix) - (int)ax);
    float oy = (float)((int)(y0 + iy) - (int)ay);
    if (a.top) {
      if (all_updated) ox = oy = 0.f;  // :333-337, min_mad is kept
      mv[0] = ox;
      mv[1] = oy;
      *mad = m;
    } else if (m < carried) {  // :401 against the carried value
      mv[0] = ox;
      mv[1] = oy;
      *mad = m;
    } else {  // nothing beat the coarser level: MV stays 2 x coarse (:458-460)
      mv[0] = (float)mvx;
      mv[1] = (float)mvy;
    }
  }
  __syncthreads();  // the LDS areas are restaged by the next round
  }
}

static int launch_wave_level(const uint8_t* d_tracked, const uint8_t* d_anchor,
                             uint64_t pair_stride, uint32_t n_pairs, uint64_t level_off,
                             uint32_t fw, uint32_t fh, uint32_t bw, uint32_t bh, uint32_t range,
                             bool top, float* d_mv, float* d_mad, hipStream_t stream) {
  WaveLevelArgs a;
  a.tracked = d_tracked;
  a.anchor = d_anchor;
  a.pair_stride = pair_stride;
  a.level_off = level_off;
  a.mfw = fw / bw;
  a.blocks = a.mfw * (fh / bh);
  a.fw = fw; a.fh = fh; a.bw = bw; a.bh = bh;
  a.range = range;
  a.top = top ? 1u : 0u;
  a.mv = d_mv;
  a.mad = d_mad;
  a.a_pitch = (bw + 3u) & ~3u;
  a.w_pitch = ((bw + 2u * range + 3u + 3u) & ~3u) + 4u;
  const uint64_t win_bytes = (uint64_t)(bh + 2ull * range) * a.w_pitch;
  uint64_t off = (uint64_t)bh * a.a_pitch + win_bytes;
  off = (off + 15u) & ~15ull;
  const uint64_t ncand_max = (2ull * range + 1) * (2ull * range + 1);
  const uint64_t lds_bytes = off + 4 * ncand_max;
  if (lds_bytes > 64 * 1024)
    return fail(SVC_ERR_UNSUPPORTED,
                "hbma: block %ux%u with search range %u needs %llu B of LDS per block (limit 65536)",
                bw, bh, range, (unsigned long long)lds_bytes);
  a.sads_off = (uint32_t)off;
  const uint64_t items = (uint64_t)a.blocks * n_pairs;
  if (items == 0) return SVC_OK;
  if (items > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "hbma: %llu work items exceed one launch", (unsigned long long)items);
  a.n_items = (uint32_t)items;
  // the dword path keeps four SADs in packed u16 lanes and flushes them per row at the latest: one row of a block
  // wider than 256 would already exceed 65535 in a lane, so such blocks take the byte path
  const bool dw = (bw % 4 == 0) && (fw % 4 == 0) && fw >= 4 && bw <= 256;
  // lane tasks per block: candidates, or groups of four candidates on the dword path
  const uint64_t tasks_max = dw ? (2ull * range + 1) * ((2ull * range + 1 + 3) / 4) : ncand_max;
  a.gs = tasks_max <= 4 ? 4u : tasks_max <= 8 ? 8u : tasks_max <= 16 ? 16u : tasks_max <= 32 ? 32u : 64u;
  const uint32_t groups = 64u / a.gs;
  a.lds_group = (uint32_t)((lds_bytes + 15u) & ~15ull);
  // small blocks: several rounds per wave, so that a launch is not bound by the workgroup launch rate
  a.rounds = bw * bh <= 16 ? 8u : bw * bh <= 64 ? 4u : 2u;
  const uint64_t grid = (items + (uint64_t)groups * a.rounds - 1) / ((uint64_t)groups * a.rounds);
  const dim3 g((uint32_t)grid), b(64);
  const size_t lds_total = (size_t)a.lds_group * groups;
  if (!dw) hipLaunchKernelGGL((hbma_wave_level_kernel<false, 0>), g, b, lds_total, stream, a);
  else if (bw == 4) hipLaunchKernelGGL((hbma_wave_level_kernel<true, 1>), g, b, lds_total, stream, a);
  else if (bw == 8) hipLaunchKernelGGL((hbma_wave_level_kernel<true, 2>), g, b, lds_total, stream, a);
  else if (bw == 16) hipLaunchKernelGGL((hbma_wave_level_kernel<true, 4>), g, b, lds_total, stream, a);
  else if (bw == 32) hipLaunchKernelGGL((hbma_wave_level_kernel<true, 8>), g, b, lds_total, stream, a);
  else hipLaunchKernelGGL((hbma_wave_level_kernel<true, 0>), g, b, lds_total, stream, a);
  return check_launch("hbma_wave_level_kernel");
}

// EstimateMotionHierarchical as L dependent launches (libs/motion.cpp:443-464).
int launch_hbma_wave(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride,
                     uint32_t n_pairs, uint32_t levels, uint32_t w, uint32_t h, uint32_t range,
                     uint32_t bw, uint32_t bh, float* d_mv, float* d_mad, hipStream_t stream) {
  const uint32_t f = 1u << (levels - 1);
  const uint32_t r_top = range / f;
  uint64_t offs[32];
  uint64_t o = 0;
  for (uint32_t l = 0; l < levels; ++l) {
    offs[l] = o;
    o += (uint64_t)(w >> l) * (h >> l);
  }
  for (int l = (int)levels - 1; l >= 0; --l) {
    int rc = launch_wave_level(d_tracked, d_anchor, pair_stride, n_pairs, offs[l], w >> l, h >> l,
                               bw >> l, bh >> l, r_top, l == (int)levels - 1, d_mv, d_mad, stream);
    if (rc) return rc;
  }
  return SVC_OK;
}

int launch_ebma(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride,
                uint32_t n_pairs, uint32_t w, uint32_t h, uint32_t range, uint32_t bw,
                uint32_t bh, float* d_mv, float* d_mad, hipStream_t stream) {
  return launch_wave_level(d_tracked, d_anchor, pair_stride, n_pairs, 0, w, h, bw, bh, range, true,
                           d_mv, d_mad, stream);
}

}  // namespace svc
