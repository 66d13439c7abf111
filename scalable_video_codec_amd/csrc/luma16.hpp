// luma16.hpp -- Y of 16 interleaved B,G,R pixels in a lane's registers: shared by the luma + pyramid kernels (luma_pyramid.hip) and by the
// record-emitting transform that also produces the luma plane (dct.hip, LUMA = true).
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

}  // namespace svc

#endif  // SVC_LUMA16_HPP
