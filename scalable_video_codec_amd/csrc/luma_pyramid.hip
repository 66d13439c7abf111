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
  uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int b0 = 3 * p, b1 = 3 * p + 1, b2 = 3 * p + 2;
    const uint32_t b = (w[b0 >> 2] >> (8 * (b0 & 3))) & 0xFFu;
    const uint32_t gch = (w[b1 >> 2] >> (8 * (b1 & 3))) & 0xFFu;
    const uint32_t r = (w[b2 >> 2] >> (8 * (b2 & 3))) & 0xFFu;
    const uint32_t y = (1868u * b + 9617u * gch + 4899u * r + 8192u) >> 14;
    out[p >> 2] |= y << (8 * (p & 3));
  }
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
  uint32_t quads_per_row;     // dw / 4
  uint32_t total;             // frames * dh * quads_per_row
};

__device__ __forceinline__ int reflect101(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// One lane: 4 destination pixels (one dword store) from a 5 x 11 source patch.
// Two launches per level keep every wave convergent: BORDER = false covers the quads
// whose patch lies inside the row (one unaligned dwordx4 per source row, all five in
// flight together); BORDER = true covers the first and last quad of each row with
// reflected byte loads (2 quads per row: noise).
template <bool BORDER>
__global__ __launch_bounds__(256) void pyr_down_kernel(PyrDownArgs a) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= a.total) return;
  const uint32_t lanes_per_row = BORDER ? 2u : a.quads_per_row - 2u;
  const uint32_t per_frame = a.dh * lanes_per_row;
  const uint32_t frame = q / per_frame, rem = q - frame * per_frame;
  const uint32_t dy = rem / lanes_per_row, k = rem - dy * lanes_per_row;
  const uint32_t dq = BORDER ? (k == 0 ? 0u : a.quads_per_row - 1u) : k + 1u;
  const uint8_t* src = a.pyr + (size_t)frame * a.pyr_stride + a.src_off;
  const int sx0 = (int)dq * 8;  // source column of the first output's centre
  constexpr bool interior = !BORDER;

  uint32_t acc[4] = {0, 0, 0, 0};
  const int taps[5] = {1, 4, 6, 4, 1};
  const uint8_t* rows[5];
#pragma unroll
  for (int r = 0; r < 5; ++r)
    rows[r] = src + (size_t)reflect101((int)dy * 2 + r - 2, (int)a.sh) * a.sw;

  auto accumulate = [&](int r, const uint32_t (&px)[11]) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const uint32_t hsum = px[2 * o] + 4 * px[2 * o + 1] + 6 * px[2 * o + 2] + 4 * px[2 * o + 3] + px[2 * o + 4];
      acc[o] += (uint32_t)taps[r] * hsum;
    }
  };

  if (interior) {
    // all five row loads are issued before the first use: one memory round trip, not five
    u32x4_a4 w[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) w[r] = *reinterpret_cast<const u32x4_a4*>(rows[r] + sx0 - 4);
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const uint32_t w0 = w[r].x, w1 = w[r].y, w2 = w[r].z, w3 = w[r].w;
      const uint32_t px[11] = {(w0 >> 16) & 0xFF, w0 >> 24,
                               w1 & 0xFF, (w1 >> 8) & 0xFF, (w1 >> 16) & 0xFF, w1 >> 24,
                               w2 & 0xFF, (w2 >> 8) & 0xFF, (w2 >> 16) & 0xFF, w2 >> 24,
                               w3 & 0xFF};  // source columns sx0 - 2 .. sx0 + 8
      accumulate(r, px);
    }
  } else {
    uint32_t pb[5][11];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int i = 0; i < 11; ++i) pb[r][i] = rows[r][reflect101(sx0 - 2 + i, (int)a.sw)];
#pragma unroll
    for (int r = 0; r < 5; ++r) accumulate(r, pb[r]);
  }
  uint32_t out = 0;
#pragma unroll
  for (int o = 0; o < 4; ++o) out |= ((acc[o] + 128u) >> 8) << (8 * o);
  uint8_t* dst = a.pyr + (size_t)frame * a.pyr_stride + a.dst_off;
  *reinterpret_cast<uint32_t*>(dst + (size_t)dy * a.dw + dq * 4) = out;
}

int launch_luma_pyramid(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames, uint32_t w,
                        uint32_t h, uint32_t levels, uint8_t* d_pyr, uint64_t pyr_stride,
                        hipStream_t stream) {
  if (n_frames == 0) return SVC_OK;
  if (w % 16 != 0 || ((uint64_t)w * h) % 16 != 0)
    return fail(SVC_ERR_UNSUPPORTED, "luma: frame width %u must be a multiple of 16", w);
  if ((w >> (levels - 1)) % 4 != 0)
    return fail(SVC_ERR_UNSUPPORTED, "pyramid: top-level width %u must be a multiple of 4", w >> (levels - 1));
  LumaArgs la;
  la.bgr = d_bgr;
  la.frame_stride = frame_stride;
  la.pyr = d_pyr;
  la.pyr_stride = pyr_stride;
  la.groups_per_frame = (uint32_t)(((uint64_t)w * h) / 16);
  const uint64_t tg = (uint64_t)la.groups_per_frame * n_frames;
  if (tg > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "luma: too many pixels for one launch");
  la.total_groups = (uint32_t)tg;
  hipLaunchKernelGGL(luma_kernel, dim3(div_up(la.total_groups, 256)), dim3(256), 0, stream, la);
  int rc = check_launch("luma_kernel");
  if (rc) return rc;

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
    pa.quads_per_row = pa.dw / 4;
    if (pa.quads_per_row < 2) return fail(SVC_ERR_UNSUPPORTED, "pyramid: level %u is narrower than 16 pixels", l);
    const uint64_t tot = (uint64_t)n_frames * pa.dh * (pa.quads_per_row - 2);
    if (tot > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "pyramid: too many pixels for one launch");
    pa.total = (uint32_t)tot;
    if (pa.total) hipLaunchKernelGGL(pyr_down_kernel<false>, dim3(div_up(pa.total, 256)), dim3(256), 0, stream, pa);
    pa.total = n_frames * pa.dh * 2;
    hipLaunchKernelGGL(pyr_down_kernel<true>, dim3(div_up(pa.total, 256)), dim3(256), 0, stream, pa);
    rc = check_launch("pyr_down_kernel");
    if (rc) return rc;
  }
  return SVC_OK;
}

}  // namespace svc
