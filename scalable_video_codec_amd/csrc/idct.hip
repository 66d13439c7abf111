// idct.hip -- the decoder-side inverse path, headless: what the reference's
// Decoder::operator() does to every tile (libs/decoder.cpp:183-207 -> DecodeBlock :128-149),
// without the GUI: choose the quant step (gazed ? 1 : background ? bg : fg, :130-135; gazed =
// the gaze rectangle contains the tile origin, :202), quantise-round-dequantise (:140-144),
// inverse DCT (cv::idct = inverse of the orthonormal DCT-II, X = C^T Y C), merge to interleaved
// B,G,R f32.  Plus an exact integer SSE against the source frame, for PSNR.
//
// Same shape as dct.hip run backwards: N lanes own a 16-pixel-wide segment column; a lane
// first owns a coefficient ROW (64 contiguous bytes per channel), inverts it in f64, parks it in
// the wave-private LDS slab; the same lanes then take COLUMNS, invert them, keep the three
// channels in registers and store interleaved pixels (192 contiguous bytes per pixel row and
// segment column).  cv::idct is OpenCV (parity unpinned offline); the oracle of record is the
// f64 inverse from the definition.
#include "svc_common.hpp"

namespace svc {

#include "dct_tables.inc"

struct IdctArgs {
  const float* planes;  // [frames][3][H][W]
  float* bgr;           // [frames][H][W][3]
  const uint32_t* types;
  uint32_t w, h, segs_per_band, bands_per_frame, total_segcols;
  uint32_t mv_bw, mv_bh, mfw, mv_blocks;
  uint32_t gaze_x, gaze_y, gaze_w, gaze_h;
  float fg_step, bg_step;
};

template <int N> struct IBasis;
template <> struct IBasis<8> {
  static __device__ __forceinline__ double even(int k, int n) { return kDctEven8[k][n]; }
  static __device__ __forceinline__ double odd(int k, int n) { return kDctOdd8[k][n]; }
};
template <> struct IBasis<16> {
  static __device__ __forceinline__ double even(int k, int n) { return kDctEven16[k][n]; }
  static __device__ __forceinline__ double odd(int k, int n) { return kDctOdd16[k][n]; }
};

// x[n] = sum_k C[k][n] y[k]: even-k terms are symmetric, odd-k terms antisymmetric in n <-> N-1-n
template <int N>
__device__ __forceinline__ void idct1d(const double* __restrict__ y, double* __restrict__ x) {
  constexpr int H = N / 2;
#pragma unroll
  for (int n = 0; n < H; ++n) {
    double e = IBasis<N>::even(0, n) * y[0];
    double o = IBasis<N>::odd(0, n) * y[1];
#pragma unroll
    for (int k = 1; k < H; ++k) {
      e = __builtin_fma(IBasis<N>::even(k, n), y[2 * k], e);
      o = __builtin_fma(IBasis<N>::odd(k, n), y[2 * k + 1], o);
    }
    x[n] = e + o;
    x[N - 1 - n] = e - o;
  }
}

__device__ __forceinline__ float requant(float c, float step) {  // libs/decoder.cpp:141-143, IEEE divide
  float q = c / step;
  q = roundf(q);
  return q * step;
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

constexpr int kPitchI = 144, kSlabI8 = 8 * kPitchI, kSlabI16 = 16 * kPitchI + 128;

template <int N>
__global__ __launch_bounds__(256) void idct_kernel(IdctArgs a) {
  constexpr int kSegPerWg = 256 / N;
  constexpr int kSlab = N == 8 ? kSlabI8 : kSlabI16;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kSegPerWg * kSlab];
  const uint32_t tid = threadIdx.x, sc_local = tid / N, j = tid % N;
  const uint32_t gsc = blockIdx.x * kSegPerWg + sc_local;
  if (gsc >= a.total_segcols) return;
  const uint32_t band_g = gsc / a.segs_per_band, seg = gsc - band_g * a.segs_per_band;
  const uint32_t frame = band_g / a.bands_per_frame, band = band_g - frame * a.bands_per_frame;
  const uint32_t y_pix = band * N, x_pix = seg * 16;

  // quant step of the (up to two) tiles this lane's coefficient row crosses
  float step[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t tx = x_pix + (N == 8 ? 8 * q : 0), ty = y_pix;
    const bool gazed = a.gaze_w && a.gaze_h && tx >= a.gaze_x && tx < a.gaze_x + a.gaze_w && ty >= a.gaze_y &&
                       ty < a.gaze_y + a.gaze_h;
    const uint32_t t = a.types[(size_t)frame * a.mv_blocks + (ty / a.mv_bh) * a.mfw + tx / a.mv_bw];
    step[q] = gazed ? 1.0f : (t == 0 ? a.bg_step : a.fg_step);
  }

  uint8_t* slab = lds + sc_local * kSlab;
  const float* in_frame = a.planes + (size_t)frame * 3 * a.w * a.h;
  float out[3][N == 8 ? 16 : 16];  // N = 8: [c][2 * y + {0,1}] two columns; N = 16: [c][y]

#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float4* src = reinterpret_cast<const float4*>(in_frame + (size_t)c * a.w * a.h + (size_t)(y_pix + j) * a.w + x_pix);
    double yv[16], r[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = src[i];
      const float s = step[N == 8 ? (i >> 1) : 0];
      yv[4 * i + 0] = (double)requant(v.x, s);
      yv[4 * i + 1] = (double)requant(v.y, s);
      yv[4 * i + 2] = (double)requant(v.z, s);
      yv[4 * i + 3] = (double)requant(v.w, s);
    }
    if (N == 8) { idct1d<8>(yv, r); idct1d<8>(yv + 8, r + 8); }
    else idct1d<16>(yv, r);
    double2* row = reinterpret_cast<double2*>(slab + j * kPitchI);
#pragma unroll
    for (int i = 0; i < 8; ++i) row[i] = make_double2(r[2 * i], r[2 * i + 1]);
    wave_sync();
    if (N == 8) {
      double ca[8], cb[8], xa[8], xb[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const double2 t = *reinterpret_cast<const double2*>(slab + v * kPitchI + j * 16);
        ca[v] = t.x; cb[v] = t.y;
      }
      idct1d<8>(ca, xa);
      idct1d<8>(cb, xb);
#pragma unroll
      for (int y = 0; y < 8; ++y) { out[c][2 * y] = (float)xa[y]; out[c][2 * y + 1] = (float)xb[y]; }
    } else {
      double cc[16], xx[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) cc[v] = *reinterpret_cast<const double*>(slab + v * kPitchI + j * 8);
      idct1d<16>(cc, xx);
#pragma unroll
      for (int y = 0; y < 16; ++y) out[c][y] = (float)xx[y];
    }
    wave_sync();
  }

  float* dst = a.bgr + (size_t)frame * a.w * a.h * 3;
  if (N == 8) {
#pragma unroll
    for (int y = 0; y < 8; ++y) {  // two pixels = 6 floats, 8-byte aligned
      float2* p = reinterpret_cast<float2*>(dst + ((size_t)(y_pix + y) * a.w + x_pix + 2 * j) * 3);
      p[0] = make_float2(out[0][2 * y], out[1][2 * y]);
      p[1] = make_float2(out[2][2 * y], out[0][2 * y + 1]);
      p[2] = make_float2(out[1][2 * y + 1], out[2][2 * y + 1]);
    }
  } else {
#pragma unroll
    for (int y = 0; y < 16; ++y) {
      float* p = dst + ((size_t)(y_pix + y) * a.w + x_pix + j) * 3;
      p[0] = out[0][y]; p[1] = out[1][y]; p[2] = out[2][y];
    }
  }
}

struct SseArgs {
  const uint8_t* src;
  uint64_t src_stride;
  const float* rec;
  unsigned long long* sse;  // [frames]
  uint32_t w, h, region_w, region_h;
};

__global__ __launch_bounds__(256) void sse_kernel(SseArgs a) {
  __shared__ unsigned long long s_part[4];
  const uint32_t frame = blockIdx.y, tid = threadIdx.x;
  const uint8_t* src = a.src + (size_t)frame * a.src_stride;
  const float* rec = a.rec + (size_t)frame * a.w * a.h * 3;
  const uint32_t row_elems = a.region_w * 3;
  const uint64_t total = (uint64_t)row_elems * a.region_h;
  unsigned long long acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + tid; i < total; i += (uint64_t)gridDim.x * 256) {
    const uint32_t y = (uint32_t)(i / row_elems), e = (uint32_t)(i - (uint64_t)y * row_elems);
    const size_t k = (size_t)y * a.w * 3 + e;
    const float r = roundf(rec[k]);
    const int v = r < 0.f ? 0 : (r > 255.f ? 255 : (int)r);
    const int d = (int)src[k] - v;
    acc += (unsigned long long)(d * d);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)acc, off, 64), hi = __shfl_xor((uint32_t)(acc >> 32), off, 64);
    acc += ((unsigned long long)hi << 32) | lo;
  }
  if ((tid & 63) == 0) s_part[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) atomicAdd(&a.sse[frame], s_part[0] + s_part[1] + s_part[2] + s_part[3]);  // integers: order-free
}

int launch_decode(const float* d_planes, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t block,
                  const uint32_t* d_types, uint32_t mv_bw, uint32_t mv_bh, uint32_t fg_step, uint32_t bg_step,
                  uint32_t gx, uint32_t gy, uint32_t gw, uint32_t gh, float* d_bgr, hipStream_t stream) {
  if (block != 8 && block != 16) return fail(SVC_ERR_UNSUPPORTED, "decode: transform block %u (supported: 8, 16)", block);
  if (w % 16 != 0 || h % block != 0) return fail(SVC_ERR_UNSUPPORTED, "decode: frame %ux%u must be a multiple of 16 x %u", w, h, block);
  IdctArgs a;
  a.planes = d_planes; a.bgr = d_bgr; a.types = d_types;
  a.w = w; a.h = h;
  a.segs_per_band = w / 16;
  a.bands_per_frame = h / block;
  const uint64_t total = (uint64_t)n_frames * a.segs_per_band * a.bands_per_frame;
  if (total == 0) return SVC_OK;
  if (total > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "decode: too many segment columns for one launch");
  a.total_segcols = (uint32_t)total;
  a.mv_bw = mv_bw; a.mv_bh = mv_bh; a.mfw = w / mv_bw; a.mv_blocks = a.mfw * (h / mv_bh);
  a.gaze_x = gx; a.gaze_y = gy; a.gaze_w = gw; a.gaze_h = gh;
  a.fg_step = (float)fg_step; a.bg_step = (float)bg_step;
  const dim3 grid(div_up(a.total_segcols, 256 / block)), blk(256);
  if (block == 8) hipLaunchKernelGGL(idct_kernel<8>, grid, blk, 0, stream, a);
  else hipLaunchKernelGGL(idct_kernel<16>, grid, blk, 0, stream, a);
  return check_launch("idct_kernel");
}

int launch_sse(const uint8_t* d_src, uint64_t src_stride, const float* d_rec, uint32_t n_frames, uint32_t w,
               uint32_t h, uint32_t region_w, uint32_t region_h, uint64_t* d_sse, hipStream_t stream) {
  if (n_frames == 0) return SVC_OK;
  hipError_t e = hipMemsetAsync(d_sse, 0, sizeof(uint64_t) * n_frames, stream);
  if (e != hipSuccess) return fail(SVC_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
  SseArgs a;
  a.src = d_src; a.src_stride = src_stride; a.rec = d_rec;
  a.sse = reinterpret_cast<unsigned long long*>(d_sse);
  a.w = w; a.h = h; a.region_w = region_w; a.region_h = region_h;
  const uint64_t total = (uint64_t)region_w * region_h * 3;
  const uint32_t gx = (uint32_t)((total + 255) / 256 < 512 ? (total + 255) / 256 : 512);
  hipLaunchKernelGGL(sse_kernel, dim3(gx ? gx : 1, n_frames), dim3(256), 0, stream, a);
  return check_launch("sse_kernel");
}

}  // namespace svc
