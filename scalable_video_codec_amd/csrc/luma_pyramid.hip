// luma_pyramid.hip -- the step in front of the motion search: luma extraction and
// the Gaussian pyramid, on the device, so a frame crosses PCIe once as BGR u8 and
// its pyramid (also the RCCL halo payload) never leaves HBM.
//
// Stands in for cv::cvtColor(BGR2YUV) + cv::extractChannel + cv::buildPyramid
// (reference libs/encoder.cpp:468-470).  OpenCV is not available offline, so these
// are this repo's own fixed-point definitions (scalable_video_codec_amd/synth.py holds
// the same definitions in torch and is what the parity tests compare with):
//   Y      = (1868 B + 9617 G + 4899 R + 8192) >> 14
//   level+1 = 5x5 [1 4 6 4 1]^2 kernel at even samples, BORDER_REFLECT_101,
//            (sum + 128) >> 8
// Both are HBM-bound byte work: 3 B in / 1 B out, then 1 B in / 0.25 B out.
#include "luma16.hpp"
#include "svc_common.hpp"

namespace svc {

struct LumaArgs {
  const uint8_t* bgr;
  uint64_t frame_stride;
  uint8_t* pyr;
  uint64_t pyr_stride;
  uint32_t groups_per_frame;  // W * H / 16
  uint32_t total_groups;
};

// one lane: 16 pixels = 48 B in (3 x dwordx4), 16 B out (1 x dwordx4)
__global__ __launch_bounds__(256) void luma_kernel(LumaArgs a) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= a.total_groups) return;
  const uint32_t frame = g / a.groups_per_frame, gi = g - frame * a.groups_per_frame;
  const uint4* src = reinterpret_cast<const uint4*>(a.bgr + (size_t)frame * a.frame_stride + (size_t)gi * 48);
  const uint4 v0 = src[0], v1 = src[1], v2 = src[2];
  const uint32_t w[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
  uint32_t out[4];
  luma16(w, out);
  uint4* dst = reinterpret_cast<uint4*>(a.pyr + (size_t)frame * a.pyr_stride + (size_t)gi * 16);
  *dst = make_uint4(out[0], out[1], out[2], out[3]);
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

struct PyrDownArgs {
  uint8_t* pyr;
  uint64_t pyr_stride;
  uint64_t src_off, dst_off;  // plane offsets inside a packed pyramid
  uint32_t sw, sh;            // source plane size
  uint32_t dw, dh;            // destination plane size (sw/2, sh/2)
  uint32_t quads_per_row;     // ceil(dw / 4)
  uint32_t total;             // frames * ceil(dh / 2) * quads_per_row
  uint32_t bytewise;          // source or destination rows are not whole dwords (sw % 4 or dw % 4): every quad gathers and
                              // stores byte by byte -- the top level of a 4-level pyramid of a frame 16 mod 32 pixels wide
                              // (PAL's 720 -> 90, 1360 -> 170, QCIF's 176 -> 22), which the reference's default build produces
};

__device__ __forceinline__ int reflect101(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// One lane: 4 destination pixels of TWO consecutive rows (two dword stores) from a 7 x 11 source
// patch -- the rows share 3 of their 5 source rows, so 7 unaligned dwordx4 loads, all in flight
// together, feed 8 outputs.  Taps 1 4 6 4 of an output are one 4 x u8 dot product over the dword that
// starts at its first tap, the fifth tap rides in as the accumulator.  The first and last quad of a row
// reflect (BORDER_REFLECT_101) and gather their bytes one by one -- 2 lanes in 120 at the level this
// runs on, in the same launch.
__global__ __launch_bounds__(256) void pyr_down_kernel(PyrDownArgs a) {
  // consecutive output rows share source rows: keep them on one XCD's L2
  const uint32_t q = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
  if (q >= a.total) return;
  const uint32_t pairs = (a.dh + 1) / 2, per_frame = pairs * a.quads_per_row;
  const uint32_t frame = q / per_frame, rem = q - frame * per_frame;
  const uint32_t dp = rem / a.quads_per_row, dq = rem - dp * a.quads_per_row;
  const uint32_t dy = 2 * dp;
  const uint8_t* src = a.pyr + (size_t)frame * a.pyr_stride + a.src_off;
  const int sx0 = (int)dq * 8;  // source column of the first output's centre
  const bool border = dq == 0 || dq + 1 == a.quads_per_row || a.bytewise != 0;

  uint32_t acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
  const uint32_t taps[5] = {1, 4, 6, 4, 1};
  const uint8_t* rows[7];
#pragma unroll
  for (int r = 0; r < 7; ++r)  // rows past the plane only feed the (discarded) second output of an odd last pair
    rows[r] = src + (size_t)reflect101(min((int)dy * 2 + r - 2, 2 * (int)a.sh - 2), (int)a.sh) * a.sw;
  constexpr uint32_t kTaps = 1u | (4u << 8) | (6u << 16) | (4u << 24);

  uint32_t h[7][4];
  if (!border) {
    u32x4_a4 w[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) w[r] = *reinterpret_cast<const u32x4_a4*>(rows[r] + sx0 - 4);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const uint32_t w0 = w[r].x, w1 = w[r].y, w2 = w[r].z, w3 = w[r].w;  // source columns sx0 - 4 .. sx0 + 11
      h[r][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), kTaps, (w1 >> 16) & 0xFFu, false);
      h[r][1] = __builtin_amdgcn_udot4(w1, kTaps, w2 & 0xFFu, false);
      h[r][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), kTaps, (w2 >> 16) & 0xFFu, false);
      h[r][3] = __builtin_amdgcn_udot4(w2, kTaps, w3 & 0xFFu, false);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      uint32_t px[11];
#pragma unroll
      for (int i = 0; i < 11; ++i)  // (clamped: columns past a partial last quad only feed outputs that are not stored)
        px[i] = rows[r][min(max(reflect101(sx0 - 2 + i, (int)a.sw), 0), (int)a.sw - 1)];
#pragma unroll
      for (int o = 0; o < 4; ++o) h[r][o] = px[2 * o] + 4 * px[2 * o + 1] + 6 * px[2 * o + 2] + 4 * px[2 * o + 3] + px[2 * o + 4];
    }
  }
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (r < 5) acc0[o] += taps[r] * h[r][o];
      if (r >= 2) acc1[o] += taps[r - 2] * h[r][o];
    }
  uint32_t out0 = 0, out1 = 0;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    out0 |= ((acc0[o] + 128u) >> 8) << (8 * o);
    out1 |= ((acc1[o] + 128u) >> 8) << (8 * o);
  }
  uint8_t* dst = a.pyr + (size_t)frame * a.pyr_stride + a.dst_off;
  if (a.bytewise) {  // rows are not whole dwords: the row's last quad may be partial, and no store is dword-aligned
    const uint32_t n = min(4u, a.dw - dq * 4);
    for (uint32_t o = 0; o < n; ++o) {
      dst[(size_t)dy * a.dw + dq * 4 + o] = (uint8_t)(out0 >> (8 * o));
      if (dy + 1 < a.dh) dst[(size_t)(dy + 1) * a.dw + dq * 4 + o] = (uint8_t)(out1 >> (8 * o));
    }
    return;
  }
  *reinterpret_cast<uint32_t*>(dst + (size_t)dy * a.dw + dq * 4) = out0;
  if (dy + 1 < a.dh) *reinterpret_cast<uint32_t*>(dst + (size_t)(dy + 1) * a.dw + dq * 4) = out1;
}

// ---- luma + first pyramid level in one pass over the BGR frame -------------------------
// A workgroup owns a 128 x 32 tile of the luma plane.  (Taller / wider tiles -- 128 x 60, 256 x 60 --
// run no faster and push the tiles in flight per XCD past its L2, so the halo re-reads start to
// miss: FETCH_SIZE 2.37 GB against 1.85 GB with this shape, which is the algorithmic figure.)  Its lanes compute Y for the tile plus
// a 2-pixel halo (reflect-101 at the frame border, exactly what pyr_down_kernel does) into
// LDS, store the interior to the level-0 plane, and then each lane produces 4 level-1 pixels
// from the LDS copy.  Versus luma_kernel + pyr_down_kernel this removes the re-read of the
// whole luma plane and one launch; the halo rows cost 12.5 % more BGR reads, served from L2.
#ifndef SVC_LUMA_HALO_DWORD
#define SVC_LUMA_HALO_DWORD 0
#endif
#ifndef SVC_LUMA_HALO_EDGE
#define SVC_LUMA_HALO_EDGE 0
#endif
#ifndef SVC_LUMA_TW
#define SVC_LUMA_TW 128
#endif
#ifndef SVC_LUMA_TH
#define SVC_LUMA_TH 32
#endif
constexpr int kTWBgr = SVC_LUMA_TW, kTHBgr = SVC_LUMA_TH, kTWPlane = 512, kTHPlane = 32 /* 128 x 64 ... 512 x 32 measured: profiles/r02_ab_pyr_tile.txt */, kOff = 16;  // LDS column c <-> x = x0 - kOff + c

struct LumaPyr1Args {
  const uint8_t* bgr;      // FROM_BGR: interleaved frames
  uint64_t frame_stride;
  uint8_t* pyr;
  uint64_t pyr_stride;
  uint64_t src_off;  // !FROM_BGR: offset of the source plane inside a packed pyramid
  uint64_t dst_off;  // offset of the plane this kernel's 5x5 pass writes (FROM_BGR: level 1 = w * h)
  uint32_t w, h;     // size of the source plane (FROM_BGR: the frame)
  uint32_t tiles_x, tiles_per_frame, total_tiles;
};

__device__ __forceinline__ uint32_t luma_of(uint32_t b, uint32_t g, uint32_t r) {
  return (1868u * b + 9617u * g + 4899u * r + 8192u) >> 14;
}

// The 5x5 pass from a luma tile in LDS (LDS column kOff + c <-> x = x0 + c, row r <-> y = y0 - 2 + r) to the next level's plane.
template <int TW, int TH, int RPT>
__device__ __forceinline__ void next_level_from_tile(const LumaPyr1Args& a, const uint8_t* tile, int x0, int y0, int w, int h, uint8_t* y_plane) {
  constexpr int kTW = TW, kTH = TH, kPitch = TW + 2 * kOff;
  const uint32_t tid = threadIdx.x;
  // (c) next level: a task = a quad of 4 output columns x RPT consecutive output rows.  The output rows of a task share source rows (2 RPT + 3
  // of them instead of 5 RPT) and the four horizontal 5-tap sums of a source row are formed once: per quad of outputs 100 vector
  // instructions and 15 LDS reads at RPT = 1, 62 / 11 at 2, 53 / 8 at 4.  Measured on the plane-to-plane pass (profiles/r05_ab_plane_rpt.txt,
  // C3 wire pyramid stage = Y -> level 1 -> level 2): RPT 1 0.273-0.282 ms, 2 0.266-0.275, 4 0.324-0.326 (one long task per lane hides its
  // LDS latency worse than two short ones): the pass is not bound by its instruction count.  The BGR pass keeps 1, the plane pass takes 2.
  constexpr int kQuads = kTW / 8;  // quads of output columns per tile row
  constexpr int kGroups = kTH / 2 / RPT;
  static_assert(kTH / 2 % RPT == 0, "row groups tile the output rows");
  constexpr uint32_t kTaps = 1u | (4u << 8) | (6u << 16) | (4u << 24);
  constexpr int taps[5] = {1, 4, 6, 4, 1};
  for (int task = (int)tid; task < kQuads * kGroups; task += 256) {
    const int q = task % kQuads, oy0 = (task / kQuads) * RPT;
    const int gx = (x0 >> 1) + 4 * q, gy0 = (y0 >> 1) + oy0;  // output-level coordinates
    if (gx >= (w >> 1) || gy0 >= (h >> 1)) continue;
    uint32_t acc[RPT][4];
#pragma unroll
    for (int o = 0; o < RPT; ++o) acc[o][0] = acc[o][1] = acc[o][2] = acc[o][3] = 0;
#pragma unroll
    for (int r = 0; r < 2 * RPT + 3; ++r) {
      // centre of output column 4q + o is LDS column kOff + 8q + 2o; taps span kOff + 8q - 2 .. + 8
      const uint8_t* rowp = &tile[(2 * oy0 + r) * kPitch + kOff + 8 * q];
      const uint32_t w0 = *reinterpret_cast<const uint32_t*>(rowp - 4);
      const uint2 mid = *reinterpret_cast<const uint2*>(rowp);
      const uint32_t w3 = *reinterpret_cast<const uint32_t*>(rowp + 8);
      // taps 1 4 6 4 of an output are one 4 x u8 dot product over the dword that starts at its first tap;
      // the fifth tap (weight 1) enters as the accumulator
      const uint32_t h0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(mid.x, w0, 2), kTaps, (mid.x >> 16) & 0xFFu, false);
      const uint32_t h1 = __builtin_amdgcn_udot4(mid.x, kTaps, mid.y & 0xFFu, false);
      const uint32_t h2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(mid.y, mid.x, 2), kTaps, (mid.y >> 16) & 0xFFu, false);
      const uint32_t h3 = __builtin_amdgcn_udot4(mid.y, kTaps, w3 & 0xFFu, false);
#pragma unroll
      for (int o = 0; o < RPT; ++o) {
        const int t5 = r - 2 * o;  // which tap of output row o this source row is
        if (t5 >= 0 && t5 < 5) {
          acc[o][0] += (uint32_t)taps[t5] * h0;
          acc[o][1] += (uint32_t)taps[t5] * h1;
          acc[o][2] += (uint32_t)taps[t5] * h2;
          acc[o][3] += (uint32_t)taps[t5] * h3;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < RPT; ++o) {
      if (gy0 + o >= (h >> 1)) break;
      uint32_t out = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) out |= ((acc[o][k] + 128u) >> 8) << (8 * k);
      *reinterpret_cast<uint32_t*>(y_plane + a.dst_off + (size_t)(gy0 + o) * (w >> 1) + gx) = out;
    }
  }
}

// FROM_BGR: luma from the B,G,R frame -> level 0 (stored) -> level 1.  !FROM_BGR: the same tile machinery on an
// existing pyramid plane (level l -> l + 1): aligned 16-byte loads of the source rows into LDS, the 5x5 pass out of
// LDS -- instead of pyr_down_kernel's 7 unaligned dwordx4 loads per 8 outputs straight from L2 (2.3 TB/s).
// TW x TH: the tile.  128 x 32 for the BGR pass (bigger tiles push the halo re-reads out of L2, see above); the
// plane-to-plane pass has a third of the bytes per pixel and takes 512 x 32, so that a lane has four or five loads in flight
// (per launch at C3, 1080p level 1 -> 2: 128x64 75.6 us, 256x64 66.4, 256x32 67.0, 128x128 71.6, 512x32 64.2, 256x128 76.0;
// profiles/r02_ab_pyr_tile.txt).
template <bool FROM_BGR, int TW, int TH, int RPT = 1>
__device__ __forceinline__ void luma_pyr1_tile(const LumaPyr1Args& a, uint32_t t, uint8_t* tile) {
  constexpr int kTW = TW, kTH = TH, kPitch = TW + 2 * kOff;
  static_assert(kPitch % 16 == 0 && kOff % 16 == 0, "LDS rows keep 16-byte alignment for the ds_write_b128");
  const uint32_t tid = threadIdx.x;
  const uint32_t frame = t / a.tiles_per_frame, tr = t - frame * a.tiles_per_frame;
  const uint32_t ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
  const int x0 = (int)tx * kTW, y0 = (int)ty * kTH;
  const int w = (int)a.w, h = (int)a.h;
  const int segs = min(kTW, w - x0) / 16;  // 16-pixel segments of this tile inside the frame
  const int xe = x0 + segs * 16;           // first column right of the tile's valid part
  const uint8_t* src = FROM_BGR ? a.bgr + (size_t)frame * a.frame_stride
                                : a.pyr + (size_t)frame * a.pyr_stride + a.src_off;
  uint8_t* y_plane = a.pyr + (size_t)frame * a.pyr_stride;

  // Rows past y = h are never needed (the last output row is centred on h - 2, its taps end at
  // row h) and must not be touched: reflect101 folds once, so a row further out would index
  // outside the frame (short frames: found by tests/test_gpu_misc_property.py).
  const int rows = min(kTH + 4, h - y0 + 3);  // LDS rows 0 .. rows-1 <-> y = y0 - 2 .. min(y0 + 33, h)
  if constexpr (!FROM_BGR) {
    // plane to plane: a lane's four or five segment loads AND its halo bytes are all issued before the first is waited for -- no branch
    // (clamped addresses; idle tasks store into 16 spare bytes behind the tile): as a loop of load -> LDS store rounds this pass paid the
    // memory latency once per round, 4.5 round trips per tile (profiles/r05_ab_pyr2.txt)
    constexpr int kSegsMax = kTW / 16, kRounds = ((kTH + 4) * kSegsMax + 255) / 256, kSpare = (kTH + 4) * kPitch;
    uint4 v[kRounds];
    int at[kRounds];
#pragma unroll
    for (int it = 0; it < kRounds; ++it) {
      const int task = (int)tid + 256 * it, r = task / kSegsMax, sgm = task - r * kSegsMax;
      const bool ok = r < rows && sgm < segs;
      at[it] = ok ? r * kPitch + kOff + sgm * 16 : kSpare;
      v[it] = *reinterpret_cast<const uint4*>(src + (size_t)reflect101(min(y0 - 2 + r, h), h) * w + min(x0 + sgm * 16, w - 16));
    }
    static_assert((kTH + 4) * 4 <= 256, "one round of halo tasks");
    const int hr = (int)tid >> 2, hk = (int)tid & 3, hx = hk < 2 ? x0 - 2 + hk : xe + (hk - 2);
    const int hat = hr < rows ? hr * kPitch + kOff + (hx - x0) : kSpare;
    const uint8_t hv = src[(size_t)reflect101(min(y0 - 2 + hr, h), h) * w + reflect101(hx, w)];
#pragma unroll
    for (int it = 0; it < kRounds; ++it) *reinterpret_cast<uint4*>(&tile[at[it]]) = v[it];
    tile[hat] = hv;
  } else {
  // (a) segment tasks: 16 pixels of one row -> 4 dwords of LDS (+ the level-0 store)
  for (int task = (int)tid; task < rows * segs; task += 256) {
    const int r = task / segs, sgm = task - r * segs;
    const int y = y0 - 2 + r, yr = reflect101(y, h);
    const int x = x0 + sgm * 16;
    uint4 o4;
    if (FROM_BGR) {
      const uint4* p = reinterpret_cast<const uint4*>(src + ((size_t)yr * w + x) * 3);
      const uint4 v0 = p[0], v1 = p[1], v2 = p[2];
      const uint32_t wd[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
      uint32_t out[4];
      luma16(wd, out);
      o4 = make_uint4(out[0], out[1], out[2], out[3]);
    } else {
      o4 = *reinterpret_cast<const uint4*>(src + (size_t)yr * w + x);
    }
    *reinterpret_cast<uint4*>(&tile[r * kPitch + kOff + sgm * 16]) = o4;  // 16-byte aligned (Guideline 17)
    if (FROM_BGR && r >= 2 && r < kTH + 2 && y < h) *reinterpret_cast<uint4*>(y_plane + (size_t)y * w + x) = o4;
#if SVC_LUMA_HALO_EDGE
    // A/B variant (profiles/r06_ab_luma_halo.txt, NOT shipped): the tile's halo pixels (two columns left, two right) by the lanes that hold the
    // row's first / last segment -- ONE more 16-byte load each (the 16 bytes in front of / behind the segment hold them), none at the frame's
    // border (reflect-101: the pixels are the lane's own) -- instead of the second task loop below (4 lanes per row, three byte loads each: a
    // third of the kernel's lane-loads for 1.6 % of its bytes).  Bit-identical and 3 - 5 % SLOWER (0.571-0.576 against 0.550-0.557 ms at C3):
    // what bounds the plane pass (load instructions by their active lanes) does not bound this kernel; the divergent loads in its main loop
    // hold its level-0 stores back
    if (FROM_BGR && sgm == 0) {
      uint32_t ya, yb;  // Y of columns x0 - 2, x0 - 1
      if (x0 > 0) {
        const uint4 e = *reinterpret_cast<const uint4*>(src + ((size_t)yr * w + x) * 3 - 16);
        ya = luma_of((e.z >> 16) & 0xFFu, e.z >> 24, e.w & 0xFFu);
        yb = luma_of((e.w >> 8) & 0xFFu, (e.w >> 16) & 0xFFu, e.w >> 24);
      } else {
        ya = (o4.x >> 16) & 0xFFu;  // column 2
        yb = (o4.x >> 8) & 0xFFu;   // column 1
      }
      tile[r * kPitch + kOff - 2] = (uint8_t)ya;
      tile[r * kPitch + kOff - 1] = (uint8_t)yb;
    }
    if (FROM_BGR && sgm == segs - 1) {
      uint32_t ya, yb;  // Y of columns xe, xe + 1
      if (xe < w) {
        const uint4 e = *reinterpret_cast<const uint4*>(src + ((size_t)yr * w + x) * 3 + 48);
        ya = luma_of(e.x & 0xFFu, (e.x >> 8) & 0xFFu, (e.x >> 16) & 0xFFu);
        yb = luma_of(e.x >> 24, e.y & 0xFFu, (e.y >> 8) & 0xFFu);
      } else {
        ya = (o4.w >> 16) & 0xFFu;  // column w - 2
        yb = (o4.w >> 8) & 0xFFu;   // column w - 3
      }
      tile[r * kPitch + kOff + segs * 16] = (uint8_t)ya;
      tile[r * kPitch + kOff + segs * 16 + 1] = (uint8_t)yb;
    }
#endif
  }
  // (b) halo pixels: two columns on each side of the valid part, every row
  for (int task = (int)tid; task < ((FROM_BGR && SVC_LUMA_HALO_EDGE) ? 0 : rows * 4); task += 256) {
    const int r = task >> 2, k = task & 3;
    const int yr = reflect101(y0 - 2 + r, h);
    const int x = k < 2 ? x0 - 2 + k : xe + (k - 2);
    if (FROM_BGR) {
#if SVC_LUMA_HALO_DWORD
      // A/B variant (profiles/r06_ab_luma_halo.txt): one unaligned dword that ENDS with the pixel instead of three byte loads -- 2 - 4 % SLOWER
      // (0.564 - 0.579 against 0.550 - 0.558 ms at C3): an unaligned dword is two requests where it straddles.  A halo pixel's column is never
      // 0 -- the reflected columns are 1, 2, w - 3, w - 2 -- so the byte in front of it is in the same row
      typedef uint32_t u32_a1 __attribute__((aligned(1)));
      const uint32_t v = *reinterpret_cast<const u32_a1*>(src + ((size_t)yr * w + reflect101(x, w)) * 3 - 1);
      tile[r * kPitch + kOff + (x - x0)] = (uint8_t)luma_of((v >> 8) & 0xFFu, (v >> 16) & 0xFFu, v >> 24);
#else
      const uint8_t* p = src + ((size_t)yr * w + reflect101(x, w)) * 3;
      tile[r * kPitch + kOff + (x - x0)] = (uint8_t)luma_of(p[0], p[1], p[2]);
#endif
    } else {
      tile[r * kPitch + kOff + (x - x0)] = src[(size_t)yr * w + reflect101(x, w)];
    }
  }
  }  // FROM_BGR
  __syncthreads();

  next_level_from_tile<TW, TH, RPT>(a, tile, x0, y0, w, h, y_plane);
}

#ifndef SVC_PLANE_RPT
#define SVC_PLANE_RPT 2
#endif
// One tile per workgroup (the BGR pass: 76 500 workgroups at C3), or -- PERSIST, the plane-to-plane pass, whose tiles take 1 - 2 us
// each -- a fixed grid whose workgroups walk the tiles of their XCD's share: see launch_pyr_down_levels.
template <bool FROM_BGR, int TW, int TH, bool PERSIST = false>
__global__ __launch_bounds__(256) void luma_pyr1_kernel(LumaPyr1Args a) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[(TH + 4) * (TW + 2 * kOff) + 16];  // + 16 spare bytes (idle load tasks of the plane pass)
  if (!PERSIST) {
    const uint32_t t = xcd_contiguous_block(blockIdx.x, gridDim.x);
    if (t < a.total_tiles) luma_pyr1_tile<FROM_BGR, TW, TH>(a, t, tile);
  } else {
    // XCD x (workgroups x, x + 8, ...) walks the x-th eighth of the tiles, its workgroups interleaved: neighbouring tiles at the same time
    const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3, per = gridDim.x >> 3;
    const uint32_t share = (a.total_tiles + 7u) / 8u, t0 = xcd * share, t1 = min(a.total_tiles, t0 + share);
    for (uint32_t t = t0 + k; t < t1; t += per) {
      luma_pyr1_tile<FROM_BGR, TW, TH, SVC_PLANE_RPT>(a, t, tile);
      __syncthreads();  // the tile buffer is rewritten by the next round
    }
  }
}
// ---- plane -> next level, a wave per column strip, no LDS (round 6) ---------------------------------------------------
// The LDS-tiled plane pass above spends its time on arithmetic and LDS, not on bytes (24 vector instructions per output pixel,
// LDS busy 39 % of a CU's time with 16 M of 41 M cycles bank conflicts: profiles/r05_pmc_C3_sq_tcp_summary.csv).  Here a WAVE owns
// a strip of 512 source columns (lane i: columns 8i .. 8i + 7, one dwordx2 per source row) and walks OB output rows down it:
//   * vertical first, two columns per instruction: a source dword is split once into its even and odd bytes as pairs of u16
//     (E = bytes 0, 2; O = bytes 1, 3) and the 1 4 6 4 1 column sums run on the pairs (<= 4088 each, with the rounding constant
//     folded in as + 8).  Output row oy needs rows 2 oy - 2 .. 2 oy + 2: with S = r[2 oy - 2] + 4 r[2 oy - 1] carried from the
//     previous row, V = S + 6 r[2 oy] + (4 r[2 oy + 1] + r[2 oy + 2]) is four instructions per pair register and no row is kept;
//   * horizontal on the column sums, still in pairs: the two outputs centred on a dword's bytes 0 and 2 are
//     6 E + 4 (O' + O) + (E' + E'') where ' / '' are E, O shifted by one u16 across the neighbouring dword (v_alignbyte) --
//     <= 65 408, so a pair of u16 never carries; the result bytes are bytes 1 and 3 of the sums (the >> 8), picked by one v_perm;
//   * the neighbouring lane's registers arrive by DPP wave shifts; the strip's own neighbours (two column sums left, one right;
//     reflected at the plane's border) come from one more dword per row, loaded by every lane from an address of its own
//     (lane 0: the dword left of the strip, the strip's last lane: the dword right of it, every other lane: a dword it reads anyway).
// No LDS, no barrier, 2 x 8 + 4 bytes in flight per lane and row.
#ifndef SVC_PYR_STRIP
#define SVC_PYR_STRIP 1
#endif
#ifndef SVC_PYR_STRIP_OB
#define SVC_PYR_STRIP_OB 8
#endif
#ifndef SVC_PYR_STRIP_ORDER
#define SVC_PYR_STRIP_ORDER 1
#endif
// timing experiments only (wrong results; tools/ab_pyr_standalone.sh): what the pass costs without its arithmetic / without the strip's
// extra dword per row / with non-temporal loads and stores
#ifndef SVC_PYR_STRIP_NOMATH
#define SVC_PYR_STRIP_NOMATH 0
#endif
#ifndef SVC_PYR_STRIP_NOHALO
#define SVC_PYR_STRIP_NOHALO 0
#endif
#ifndef SVC_PYR_STRIP_NT
#define SVC_PYR_STRIP_NT 0
#endif
#ifndef SVC_PYR_STRIP_HALO
#define SVC_PYR_STRIP_HALO 1
#endif
struct PyrStripArgs {
  uint8_t* pyr;
  uint64_t pyr_stride, src_off, dst_off;
  uint32_t sw, sh;  // source plane; sw % 8 == 0, sh % 2 == 0, sh >= 4
  uint32_t strips, bands, total_waves;
};

typedef uint16_t u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint16_t k, uint32_t c) {  // v_pk_mad_u16
  return __builtin_bit_cast(uint32_t, (u16x2_t)(__builtin_bit_cast(u16x2_t, a) * k + __builtin_bit_cast(u16x2_t, c)));
}

// SVC_PYR_STRIP_HALO: how the strip's own neighbours (two column sums left of it, one right) get into lane 0 / the strip's last lane --
//   1 (as built)  one more vector dword per row under a two-lane exec mask, in one block in front of the rows;
//   0             the same dword loaded by all 64 lanes from per-lane addresses: + 6 % (a load costs the texture addresser by its active lanes);
//   2             two SCALAR loads per row (the addresses are wave-uniform; `src` is a __restrict__ kernel argument, so hipcc emits
//                 s_load_dword) and their column sums on the scalar unit -- 25 % fewer vector instructions, 19 fewer vector loads, and + 6 %:
//                 38 scalar loads per wave that must ALL have returned (lgkmcnt(0)) before the first row is touched.
// alone over a C3-sized clip, one box: 0.2050 / 0.2180 / 0.2168 ms (profiles/r06_ab_pyr_strip.txt, section 9).
template <int OB>
__global__ __launch_bounds__(256) void pyr_strip_kernel(PyrStripArgs a, const uint8_t* __restrict__ pyr_in, uint8_t* __restrict__ pyr_out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * 4u + (threadIdx.x >> 6));
  if (wv >= a.total_waves) return;
#if SVC_PYR_STRIP_ORDER == 0
  // consecutive waves: consecutive bands of one strip (they share two or three source rows)
  const uint32_t band = wv % a.bands, t = wv / a.bands, strip = t % a.strips, frame = t / a.strips;
#else
  // consecutive waves: the strips of one band side by side -- a workgroup's four waves read whole rows of a 1080p plane, one contiguous run
  const uint32_t strip = wv % a.strips, t = wv / a.strips, band = t % a.bands, frame = t / a.bands;
#endif
  const int w = (int)a.sw, h = (int)a.sh, dw = w >> 1, dh = h >> 1;
  const int x0 = (int)strip * 512, valid = min(512, w - x0), la = valid / 8 - 1, xe = x0 + valid;
  const int oy0 = (int)band * OB, rows = min(OB, dh - oy0);
  const uint8_t* src = pyr_in + (size_t)frame * a.pyr_stride + a.src_off;  // the source plane is only read, the destination plane only
  uint8_t* dst = pyr_out + (size_t)frame * a.pyr_stride + a.dst_off;       // written, and they do not overlap: __restrict__ holds

  const uint32_t col = (uint32_t)x0 + min(lane, (uint32_t)la) * 8u;
  const bool last = lane == (uint32_t)la;
  // the strip's neighbours as a pair H = (lo, hi) of u16: left of it the columns (x0 - 2, x0 - 1) -- bytes 2, 3 of the dword in front of the
  // strip, reflected (columns 2, 1: bytes 2, 1 of the strip's first dword) at the plane's left border; right of it column xe in lo -- byte 0 of
  // the dword behind the strip, reflected (column w - 2: byte 2 of the plane's last dword) at the right border
  const uint32_t hcol_l = x0 ? (uint32_t)x0 - 4u : 0u, hcol_r = xe < w ? (uint32_t)xe : (uint32_t)w - 4u;
  const uint32_t hsel_l = x0 ? 0x0c030c02u : 0x0c010c02u, hsel_r = xe < w ? 0x0c0c0c00u : 0x0c0c0c02u;  // v_perm selectors; 0x0c = a zero byte
#if SVC_PYR_STRIP_HALO != 2
  uint32_t hcol = col, hsel = 0x0c0c0c0cu;
  if (lane == 0) { hcol = hcol_l; hsel = hsel_l; }
  if (last) { hcol = hcol_r; hsel = hsel_r; }
#endif

  // a source row -> its pair registers (E = bytes 0, 2; O = bytes 1, 3 of each dword)
  struct Row { uint32_t e0, o0, e1, o1; };
  auto row_offset = [&](int r) { return (uint32_t)reflect101(min(2 * oy0 - 2 + r, h), h) * (uint32_t)w; };  // wave-uniform; a plane is < 4 GB
  auto fetch = [&](int r, uint2& v) {
#if SVC_PYR_STRIP_NT
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t nt = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(src + row_offset(r) + col));
    v = make_uint2(nt.x, nt.y);
#else
    v = *reinterpret_cast<const uint2*>(src + row_offset(r) + col);
#endif
  };
  auto split = [&](const uint2& v) {
    Row r;
    r.e0 = v.x & 0x00FF00FFu; r.o0 = __builtin_amdgcn_perm(0u, v.x, 0x0c030c01u);
    r.e1 = v.y & 0x00FF00FFu; r.o1 = __builtin_amdgcn_perm(0u, v.y, 0x0c030c01u);
    return r;
  };

  // every source row of the band is requested before the first is used (2 OB + 3 rows in flight: a version that walked the band with two
  // rows of prefetch measured level with the LDS-tiled pass it replaces); straight-line code, so the compiler's waits are exact and rows are
  // consumed as they land
  constexpr int kRows = 2 * OB + 3;
  uint2 rv[kRows];
  // the neighbours' pair per row: hl / hr (uniform: scalar registers) with SVC_PYR_STRIP_HALO == 2, else hh (per lane: lane 0 and the last one)
  uint32_t hl[kRows], hr[kRows], hh[kRows];
#if SVC_PYR_STRIP_HALO == 2
  auto pick = [](uint32_t word, uint32_t sel) {  // what v_perm_b32(0, word, sel) gives for the two selectors in use, on the scalar unit
    const uint32_t lo = (word >> (8 * (sel & 3u))) & 0xFFu;
    const uint32_t hi = ((sel >> 16) & 0xFFu) == 0x0cu ? 0u : (word >> (8 * ((sel >> 16) & 3u))) & 0xFFu;
    return lo | (hi << 16);
  };
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const uint32_t ro = row_offset(r);
    hl[r] = pick(*reinterpret_cast<const uint32_t*>(src + ro + hcol_l), hsel_l);
    hr[r] = pick(*reinterpret_cast<const uint32_t*>(src + ro + hcol_r), hsel_r);
    hh[r] = 0;
  }
#elif SVC_PYR_STRIP_HALO == 1
  // the extra dword under the exec mask of the two lanes that need it, in one block in front of the rows (a load instruction costs the texture
  // addresser by its active lanes: with all 64 lanes loading it the pass took 0.215 instead of 0.180 ms without it)
#pragma unroll
  for (int r = 0; r < kRows; ++r) hh[r] = hl[r] = hr[r] = 0;
  if (lane == 0 || last) {
#pragma unroll
    for (int r = 0; r < kRows; ++r) hh[r] = *reinterpret_cast<const uint32_t*>(src + row_offset(r) + hcol);
  }
#pragma unroll
  for (int r = 0; r < kRows; ++r) hh[r] = __builtin_amdgcn_perm(0u, hh[r], hsel);
#else
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    hl[r] = hr[r] = 0;
#if SVC_PYR_STRIP_NOHALO
    hh[r] = 0;
#else
    hh[r] = __builtin_amdgcn_perm(0u, *reinterpret_cast<const uint32_t*>(src + row_offset(r) + hcol), hsel);
#endif
  }
#endif
#pragma unroll
  for (int r = 0; r < kRows; ++r) fetch(r, rv[r]);  // rows past the band's last output (and past the plane: clamped) are loaded and not used

  // column sums, two columns per register: V(oy) = s + 6 e + (4 o + n) with s = r[2 oy - 2] + 4 r[2 oy - 1] + 8 carried, e = r[2 oy],
  // o = r[2 oy + 1], n = r[2 oy + 2] (plain 32-bit adds where no pair can carry: v_add3_u32 / s_add)
  constexpr uint32_t k8 = 0x00080008u;
  Row s, e;
  uint32_t s_l, e_l, s_r, e_r, s_h, e_h;  // the same for the neighbours' pairs
  {
    const Row r0 = split(rv[0]), r1 = split(rv[1]);
    e = split(rv[2]);
    s.e0 = (r1.e0 << 2) + r0.e0 + k8; s.o0 = (r1.o0 << 2) + r0.o0 + k8;
    s.e1 = (r1.e1 << 2) + r0.e1 + k8; s.o1 = (r1.o1 << 2) + r0.o1 + k8;
    s_l = (hl[1] << 2) + hl[0] + k8; e_l = hl[2];
    s_r = (hr[1] << 2) + hr[0] + k8; e_r = hr[2];
    s_h = (hh[1] << 2) + hh[0] + k8; e_h = hh[2];
  }
  uint8_t* out = dst + (size_t)oy0 * dw + (x0 >> 1) + 4 * lane;
  const bool mine = lane <= (uint32_t)la;
#pragma unroll
  for (int i = 0; i < OB; ++i) {
    const Row o = split(rv[3 + 2 * i]), n = split(rv[4 + 2 * i]);
    Row v;
    { const uint32_t q = o.e0 << 2; v.e0 = pk_mad(e.e0, 6, s.e0) + q + n.e0; s.e0 = q + e.e0 + k8; }
    { const uint32_t q = o.o0 << 2; v.o0 = pk_mad(e.o0, 6, s.o0) + q + n.o0; s.o0 = q + e.o0 + k8; }
    { const uint32_t q = o.e1 << 2; v.e1 = pk_mad(e.e1, 6, s.e1) + q + n.e1; s.e1 = q + e.e1 + k8; }
    { const uint32_t q = o.o1 << 2; v.o1 = pk_mad(e.o1, 6, s.o1) + q + n.o1; s.o1 = q + e.o1 + k8; }
    e = n;
    // the neighbours' column sums: (V[x0 - 2], V[x0 - 1]) for lane 0, (V[xe], -) for the last lane
    uint32_t v_l, v_r;
#if SVC_PYR_STRIP_HALO == 2
    { const uint32_t q = hl[3 + 2 * i] << 2; v_l = s_l + 6u * e_l + q + hl[4 + 2 * i]; s_l = q + e_l + k8; e_l = hl[4 + 2 * i]; }
    { const uint32_t q = hr[3 + 2 * i] << 2; v_r = s_r + 6u * e_r + q + hr[4 + 2 * i]; s_r = q + e_r + k8; e_r = hr[4 + 2 * i]; }
#else
    { const uint32_t q = hh[3 + 2 * i] << 2; v_l = pk_mad(e_h, 6, s_h) + q + hh[4 + 2 * i]; s_h = q + e_h + k8; e_h = hh[4 + 2 * i]; }
    v_r = v_l;
#endif
    // neighbours: the previous lane's second dword (lane 0: the strip's left neighbours), the next lane's first dword (last lane: the right one)
    const uint32_t pe = __builtin_amdgcn_update_dpp(v_l << 16, v.e1, 0x138, 0xf, 0xf, false);  // wave_shr:1; lane 0 keeps `old`
    const uint32_t po = __builtin_amdgcn_update_dpp(v_l, v.o1, 0x138, 0xf, 0xf, false);
    uint32_t ne = __builtin_amdgcn_update_dpp(v_r, v.e0, 0x130, 0xf, 0xf, false);  // wave_shl:1
    ne = last ? v_r : ne;
    // outputs centred on columns 0, 2 (dword 0) and 4, 6 (dword 1) of the lane
    const uint32_t l2a = __builtin_amdgcn_alignbyte(v.e0, pe, 2), l1a = __builtin_amdgcn_alignbyte(v.o0, po, 2);
    const uint32_t mid = __builtin_amdgcn_alignbyte(v.e1, v.e0, 2), l1b = __builtin_amdgcn_alignbyte(v.o1, v.o0, 2);
    const uint32_t r2b = __builtin_amdgcn_alignbyte(ne, v.e1, 2);
    const uint32_t qa = pk_mad(v.e0, 6, ((l1a + v.o0) << 2) + l2a + mid);
    const uint32_t qb = pk_mad(v.e1, 6, ((l1b + v.o1) << 2) + mid + r2b);
#if SVC_PYR_STRIP_NOMATH
    const uint32_t px = rv[3 + 2 * i].x ^ rv[4 + 2 * i].y ^ v_l ^ v_r ^ (i == 0 ? rv[0].x ^ rv[1].x ^ rv[2].x : 0u);
    (void)qa; (void)qb;
#else
    const uint32_t px = __builtin_amdgcn_perm(qb, qa, 0x07050301u);  // (>> 8) of the four sums
#endif
#if SVC_PYR_STRIP_NT
    if (mine && i < rows) __builtin_nontemporal_store(px, reinterpret_cast<uint32_t*>(out));
#else
    if (mine && i < rows) *reinterpret_cast<uint32_t*>(out) = px;  // (the plane's last band may be short)
#endif
    out += dw;
  }
  (void)s_l; (void)e_l; (void)s_r; (void)e_r; (void)s_h; (void)e_h;
}

// any frame width: one pixel per lane
__global__ __launch_bounds__(256) void luma_any_kernel(const uint8_t* bgr, uint64_t frame_stride, uint8_t* pyr, uint64_t pyr_stride,
                                                        uint32_t px_per_frame, uint32_t total) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const uint32_t frame = i / px_per_frame, p = i - frame * px_per_frame;
  const uint8_t* s = bgr + (size_t)frame * frame_stride + (size_t)p * 3;
  pyr[(size_t)frame * pyr_stride + p] = (uint8_t)luma_of(s[0], s[1], s[2]);
}

int launch_luma_pyramid(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames, uint32_t w,
                        uint32_t h, uint32_t levels, uint8_t* d_pyr, uint64_t pyr_stride,
                        hipStream_t stream) {
  if (n_frames == 0) return SVC_OK;
  if (w % 16 != 0) {
    // Frames that are not whole 16-pixel segments wide (8 x 8 MV blocks pad to multiples of 8: --mv-block-w 8 on a 360-pixel
    // frame): a pixel per lane for the luma plane, then the plane-to-plane kernels.  Not the roofline path.
    const uint64_t px = (uint64_t)w * h, tot = px * n_frames;
    if (tot > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "luma: too many pixels for one launch");
    hipLaunchKernelGGL(luma_any_kernel, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, stream, d_bgr, frame_stride, d_pyr,
                       pyr_stride, (uint32_t)px, (uint32_t)tot);
    int rc0 = check_launch("luma_any_kernel");
    if (rc0) return rc0;
    return launch_pyr_down_levels(d_pyr, pyr_stride, n_frames, w, h, levels, 0, stream);
  }
  LumaArgs la;
  la.bgr = d_bgr;
  la.frame_stride = frame_stride;
  la.pyr = d_pyr;
  la.pyr_stride = pyr_stride;
  la.groups_per_frame = (uint32_t)(((uint64_t)w * h) / 16);
  const uint64_t tg = (uint64_t)la.groups_per_frame * n_frames;
  if (tg > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "luma: too many pixels for one launch");
  la.total_groups = (uint32_t)tg;
  int rc;
  uint32_t first_plain_level = 0;
  // level-1 width must be a multiple of 4 for the fused kernel's dword stores
  if (levels >= 2 && (w / 2) % 4 == 0 && h % 2 == 0) {
    LumaPyr1Args fa{};
    fa.bgr = d_bgr;
    fa.frame_stride = frame_stride;
    fa.pyr = d_pyr;
    fa.pyr_stride = pyr_stride;
    fa.dst_off = (uint64_t)w * h;
    fa.w = w; fa.h = h;
    fa.tiles_x = div_up(w, kTWBgr);
    fa.tiles_per_frame = fa.tiles_x * div_up(h, kTHBgr);
    const uint64_t tt = (uint64_t)fa.tiles_per_frame * n_frames;
    if (tt > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "luma: too many tiles for one launch");
    fa.total_tiles = (uint32_t)tt;
    // (the same tile with its B,G,R rows brought in by LDS-DMA -- one wait per workgroup instead of three -- measured level: serial 0.551-0.558
    // against 0.557-0.565 ms, pipelined 0.590-0.600 against 0.575-0.582 at C3, level at C5; profiles/r05_ab_luma_dma.txt; removed)
    hipLaunchKernelGGL((luma_pyr1_kernel<true, kTWBgr, kTHBgr>), dim3(fa.total_tiles), dim3(256), 0, stream, fa);
    if ((rc = check_launch("luma_pyr1_kernel"))) return rc;
    first_plain_level = 1;
  } else {
    hipLaunchKernelGGL(luma_kernel, dim3(div_up(la.total_groups, 256)), dim3(256), 0, stream, la);
    if ((rc = check_launch("luma_kernel"))) return rc;
  }

  return launch_pyr_down_levels(d_pyr, pyr_stride, n_frames, w, h, levels, first_plain_level, stream);
}

// Levels first_plain_level + 1 .. levels - 1 of n_frames packed pyramids whose levels 0 .. first_plain_level exist
// (cv::buildPyramid from a given level-0 plane is first_plain_level = 0: svc_hip_build_pyramid_host).
int launch_pyr_down_levels(uint8_t* d_pyr, uint64_t pyr_stride, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t levels,
                           uint32_t first_plain_level, hipStream_t stream) {
  int rc;
  uint64_t off = 0;
  for (uint32_t l = 0; l + 1 < levels; ++l) {
    PyrDownArgs pa;
    pa.pyr = d_pyr;
    pa.pyr_stride = pyr_stride;
    pa.sw = w >> l; pa.sh = h >> l;
    pa.dw = pa.sw / 2; pa.dh = pa.sh / 2;
    pa.src_off = off;
    off += (uint64_t)pa.sw * pa.sh;
    pa.dst_off = off;
    pa.quads_per_row = div_up(pa.dw, 4);
    pa.bytewise = (pa.sw % 4 != 0 || pa.dw % 4 != 0) ? 1u : 0u;
    if (l < first_plain_level) continue;  // produced by luma_pyr1_kernel
    // the LDS-tiled pass: source rows are read as aligned 16-byte segments
    if (pa.sw % 16 == 0 && pa.sh % 2 == 0 && pa.dw % 4 == 0 && pa.src_off % 16 == 0 && pyr_stride % 16 == 0 &&
        (reinterpret_cast<uintptr_t>(d_pyr) & 15) == 0 && pa.sh >= 3) {
      LumaPyr1Args fa{};
      fa.pyr = d_pyr;
      fa.pyr_stride = pyr_stride;
      fa.src_off = pa.src_off;
      fa.dst_off = pa.dst_off;
      fa.w = pa.sw; fa.h = pa.sh;
      fa.tiles_x = div_up(pa.sw, kTWPlane);
      fa.tiles_per_frame = fa.tiles_x * div_up(pa.sh, kTHPlane);
      const uint64_t tt = (uint64_t)fa.tiles_per_frame * n_frames;
      if (tt > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "pyramid: too many tiles for one launch");
      fa.total_tiles = (uint32_t)tt;
#if SVC_PYR_STRIP
      if (pa.sh >= 4) {
        PyrStripArgs sa{};
        sa.pyr = d_pyr; sa.pyr_stride = pyr_stride; sa.src_off = pa.src_off; sa.dst_off = pa.dst_off;
        sa.sw = pa.sw; sa.sh = pa.sh;
        sa.strips = div_up(pa.sw, 512);
        sa.bands = div_up(pa.dh, SVC_PYR_STRIP_OB);
        const uint64_t tw = (uint64_t)sa.strips * sa.bands * n_frames;
        if (tw > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "pyramid: too many strips for one launch");
        sa.total_waves = (uint32_t)tw;
        hipLaunchKernelGGL((pyr_strip_kernel<SVC_PYR_STRIP_OB>), dim3(div_up(sa.total_waves, 4)), dim3(256), 0, stream, sa,
                           static_cast<const uint8_t*>(d_pyr), d_pyr);
        if ((rc = check_launch("pyr_strip_kernel"))) return rc;
        continue;
      }
#endif
      {
        // a fixed grid whose workgroups walk the tiles: 0.062 -> 0.054 ms per launch at C3 (profiles/r04_ab_pyr_persist.txt; 1024 workgroups
        // are too few, 2048 and 4096 level).  A double-buffered LDS-DMA form of the same walk measured no better (r04_ab_pyr_stream.txt).
        const uint32_t grid = std::min<uint32_t>((fa.total_tiles + 7u) / 8u * 8u, 2048u);
        hipLaunchKernelGGL((luma_pyr1_kernel<false, kTWPlane, kTHPlane, true>), dim3(grid), dim3(256), 0, stream, fa);
      }
      if ((rc = check_launch("luma_pyr1_kernel<false>"))) return rc;
      continue;
    }
    if (pa.dw < 1 || pa.dh < 1 || pa.sw < 3 || pa.sh < 3)  // reflect-101 folds once: a source side below 3 cannot be mirrored
      return fail(SVC_ERR_UNSUPPORTED, "pyramid: level %u (%u x %u) is too small to reduce", l, pa.sw, pa.sh);
    const uint64_t tot = (uint64_t)n_frames * ((pa.dh + 1) / 2) * pa.quads_per_row;
    if (tot > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "pyramid: too many pixels for one launch");
    pa.total = (uint32_t)tot;
    hipLaunchKernelGGL(pyr_down_kernel, dim3(div_up(pa.total, 256)), dim3(256), 0, stream, pa);
    rc = check_launch("pyr_down_kernel");
    if (rc) return rc;
  }
  return SVC_OK;
}

}  // namespace svc
