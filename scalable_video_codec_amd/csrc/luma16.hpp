// luma16.hpp -- what the luma + pyramid kernels (luma_pyramid.hip) and the transform that leaves the luma plane -- and, on frames of whole
// 128 x 32 tiles, level 1 -- as a by-product (dct.hip, LUMA) share: Y of 16 interleaved B,G,R pixels in a lane's registers, the one-pixel form,
// BORDER_REFLECT_101, and the 5x5 pass from a luma tile in LDS to the next level.
#ifndef SVC_LUMA16_HPP
#define SVC_LUMA16_HPP

#include <cstdint>

#include <hip/hip_runtime.h>

namespace svc {

// Y of 16 interleaved BGR pixels (12 dwords) -> 16 bytes.  A pixel's three bytes are brought to the
// low end of a dword by one byte-align, and the weighted sum is two 4 x u8 dot products: the weights
// split into high and low bytes (1868 = 7*256 + 76, 9617 = 37*256 + 145, 4899 = 19*256 + 35), the
// fourth byte (the next pixel's B) gets weight 0, and the rounding constant rides in as 32 << 8.
__device__ __forceinline__ void luma16(const uint32_t (&w)[12], uint32_t (&out)[4]) {
  constexpr uint32_t kLo = 76u | (145u << 8) | (35u << 16), kHi = 7u | (37u << 8) | (19u << 16);
  out[0] = out[1] = out[2] = out[3] = 0;
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int d = (3 * p) >> 2, sh = (3 * p) & 3;
    const uint32_t px = sh == 0 ? w[d] : sh == 1 ? w[d] >> 8 : __builtin_amdgcn_alignbyte(w[d + 1 < 12 ? d + 1 : d], w[d], sh);
    const uint32_t hi = __builtin_amdgcn_udot4(px, kHi, 32u, false);
    const uint32_t y = __builtin_amdgcn_udot4(px, kLo, hi << 8, false) >> 14;
    out[p >> 2] |= y << (8 * (p & 3));
  }
}

constexpr int kOff = 16;  // a luma tile in LDS: column kOff + c <-> x = x0 + c (rows keep 16-byte alignment), row r <-> y = y0 - 2 + r

__device__ __forceinline__ int reflect101(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

__device__ __forceinline__ uint32_t luma_of(uint32_t b, uint32_t g, uint32_t r) {
  return (1868u * b + 9617u * g + 4899u * r + 8192u) >> 14;
}

// The 5x5 pass from a luma tile in LDS (LDS column kOff + c <-> x = x0 + c, row r <-> y = y0 - 2 + r) to the next level's plane.
template <int TW, int TH, int RPT>
__device__ __forceinline__ void next_level_from_tile(uint64_t dst_off, const uint8_t* tile, int x0, int y0, int w, int h, uint8_t* y_plane) {
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
      *reinterpret_cast<uint32_t*>(y_plane + dst_off + (size_t)(gy0 + o) * (w >> 1) + gx) = out;
    }
  }
}

}  // namespace svc

#endif  // SVC_LUMA16_HPP
