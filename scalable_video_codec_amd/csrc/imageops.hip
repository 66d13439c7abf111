// imageops.hip -- one entry point per OpenCV call of the reference's per-frame loop (libs/encoder.cpp:447-640) that does
// arithmetic, host-pointer forms: what the OpenCV-shaped adapter (compat/opencv2/) forwards cv::cvtColor,
// cv::buildPyramid, cv::morphologyEx, cv::kmeans, cv::connectedComponents and the collected cv::dct calls to, so that
// the reference's own Encoder::operator() can drive the GPU unchanged.
//
// These are the PER-CALL forms: a frame's worth of small, latency-bound work behind a PCIe round trip each.  The
// throughput path is the fused, batched device forms (luma_pyramid.hip, segment.hip, dct.hip); the definitions are the
// same (stated in oracle/svc_segment.c and oracle/svc_imageops.c), and tests/test_gpu_imageops.py checks that composing these calls
// the way the reference composes the cv:: ones reproduces svc_hip_segment_frames' region ids.
#include <cstring>

#include "svc_common.hpp"
#include "union_find.hpp"

namespace svc {

// ---- cv::cvtColor(BGR2YUV), 8-bit ---------------------------------------------------------------------------------
// OpenCV 3.4's integer path (14 fractional bits): Y as luma_pyramid.hip; U = descale((B - Y) * 8061 + half),
// V = descale((R - Y) * 14369 + half), half = 128 << 14, descale(x) = (x + 8192) >> 14, saturated.
__global__ __launch_bounds__(256) void bgr2yuv_kernel(const uint8_t* bgr, uint8_t* yuv, uint64_t pixels) {
  // four pixels = three dwords in, three dwords out (the staging buffers are 256-byte aligned)
  const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x, p0 = q * 4;
  if (p0 >= pixels) return;
  uint32_t in[3] = {0, 0, 0}, out[3] = {0, 0, 0};
  const uint32_t np = pixels - p0 < 4 ? (uint32_t)(pixels - p0) : 4u;
  if (np == 4) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(bgr + p0 * 3);
    in[0] = s[0]; in[1] = s[1]; in[2] = s[2];
  } else {
    for (uint32_t i = 0; i < np * 3; ++i) in[i >> 2] |= (uint32_t)bgr[p0 * 3 + i] << (8 * (i & 3));
  }
#pragma unroll
  for (uint32_t p = 0; p < 4; ++p) {
    const uint32_t i0 = 3 * p, i1 = 3 * p + 1, i2 = 3 * p + 2;
    const int b = (int)((in[i0 >> 2] >> (8 * (i0 & 3))) & 255u), g = (int)((in[i1 >> 2] >> (8 * (i1 & 3))) & 255u),
              r = (int)((in[i2 >> 2] >> (8 * (i2 & 3))) & 255u);
    const int y = (1868 * b + 9617 * g + 4899 * r + 8192) >> 14;
    const int u = ((b - y) * 8061 + (128 << 14) + 8192) >> 14;
    const int v = ((r - y) * 14369 + (128 << 14) + 8192) >> 14;
    out[i0 >> 2] |= (uint32_t)y << (8 * (i0 & 3));
    out[i1 >> 2] |= (uint32_t)min(max(u, 0), 255) << (8 * (i1 & 3));
    out[i2 >> 2] |= (uint32_t)min(max(v, 0), 255) << (8 * (i2 & 3));
  }
  if (np == 4) {
    uint32_t* d = reinterpret_cast<uint32_t*>(yuv + p0 * 3);
    d[0] = out[0]; d[1] = out[1]; d[2] = out[2];
  } else {
    for (uint32_t i = 0; i < np * 3; ++i) yuv[p0 * 3 + i] = (uint8_t)(out[i >> 2] >> (8 * (i & 3)));
  }
}

// ---- erode / dilate with a rectangle -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void morph_rect_kernel(const uint8_t* src, uint8_t* dst, int w, int h, int kw, int kh,
                                                        int dilate) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= w * h) return;
  const int y = i / w, x = i - y * w, ax = kw / 2, ay = kh / 2;
  int v = dilate ? 0 : 255;
  for (int ky = 0; ky < kh; ++ky) {
    const int sy = y + ky - ay;
    if (sy < 0 || sy >= h) continue;  // outside the image: ignored
    for (int kx = 0; kx < kw; ++kx) {
      const int sx = x + kx - ax;
      if (sx < 0 || sx >= w) continue;
      const int p = src[sy * w + sx];
      v = dilate ? max(v, p) : min(v, p);
    }
  }
  dst[i] = (uint8_t)v;
}

// ---- k-means (oracle/svc_segment.c's definition, up to 4 integer coordinates) --------------------------------------
constexpr uint32_t kKmT = 1024, kKmMaxK = 255, kKmMaxAttempts = 64;  // labels are bytes; the compactness table is 512 B

struct KmArgs {
  const float* feat;            // [n][dims]
  int* pts;                     // [n][4]
  uint32_t* bad;                // != 0: a coordinate is not an integer of magnitude below 32768
  unsigned long long* dmin;     // [attempts][n] k-means++ running minima
  uint8_t* lab;                 // [attempts][n]
  unsigned long long* compact;  // [attempts]
  int32_t* out;                 // [n] labels of the best attempt
  double* out_compact;
  uint32_t n, dims, k, attempts, max_iter;
  double eps2;
  uint64_t seed;
};

__device__ __forceinline__ uint64_t km_hash(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void km_convert_kernel(KmArgs a) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= a.n) return;
  int p[4] = {0, 0, 0, 0};
  bool bad = false;
  for (uint32_t d = 0; d < a.dims; ++d) {
    const float f = a.feat[(size_t)i * a.dims + d];
    bad = bad || !(f == truncf(f)) || !(fabsf(f) < 32768.0f);
    p[d] = bad ? 0 : (int)f;
  }
  if (bad) atomicOr(a.bad, 1u);
  int4* dst = reinterpret_cast<int4*>(a.pts) + i;
  *dst = make_int4(p[0], p[1], p[2], p[3]);
}

__device__ __forceinline__ uint64_t km_dist_int(const int4& p, const int (&c)[4]) {
  const long long d0 = (long long)p.x - c[0], d1 = (long long)p.y - c[1], d2 = (long long)p.z - c[2], d3 = (long long)p.w - c[3];
  return (uint64_t)(d0 * d0) + (uint64_t)(d1 * d1) + (uint64_t)(d2 * d2) + (uint64_t)(d3 * d3);
}

// s = d0 * d0; s = s + d1 * d1; ... in coordinate order (a constant leading coordinate contributes an exact 0.0, so
// the reference's (0, mv.x, x, y) features give the bits of oracle/svc_segment.c's three-coordinate form)
__device__ __forceinline__ double km_dist_dbl(const int4& p, const double* c, uint32_t dims) {
  const double q[4] = {(double)p.x, (double)p.y, (double)p.z, (double)p.w};
  const double t0 = q[0] - c[0];
  double s = t0 * t0;
  for (uint32_t d = 1; d < dims; ++d) {
    const double t = q[d] - c[d];
    s = s + t * t;
  }
  return s;
}

__device__ __forceinline__ uint64_t km_shfl_up_u64(uint64_t v, int off) {
  const uint32_t lo = __shfl_up((uint32_t)v, off, 64), hi = __shfl_up((uint32_t)(v >> 32), off, 64);
  return ((uint64_t)hi << 32) | lo;
}

// exclusive scan of one u64 per thread over the 1024-lane workgroup (two barriers); *total = block sum
__device__ __forceinline__ uint64_t km_block_excl_scan(uint64_t v, uint64_t* s_scan, uint32_t tid, uint64_t* total) {
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  uint64_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t y = km_shfl_up_u64(x, off);
    if (lane >= (uint32_t)off) x += y;
  }
  __syncthreads();
  if (lane == 63) s_scan[wave] = x;
  __syncthreads();
  uint64_t woff = 0, tot = 0;
  for (uint32_t wv = 0; wv < kKmT / 64; ++wv) {
    const uint64_t t = s_scan[wv];
    woff += wv < wave ? t : 0;
    tot += t;
  }
  *total = tot;
  return woff + x - v;
}

// One workgroup per attempt; thread t owns the contiguous points [t * per, (t + 1) * per).
__global__ __launch_bounds__(kKmT) void km_attempt_kernel(KmArgs a) {
  __shared__ uint64_t s_scan[kKmT / 64];
  __shared__ int s_ci[kKmMaxK][4];
  __shared__ double s_c[kKmMaxK][4];
  __shared__ unsigned long long s_sum[kKmMaxK][4];
  __shared__ uint32_t s_cnt[kKmMaxK];
  __shared__ double s_shift[kKmMaxK];
  __shared__ unsigned long long s_compact;
  __shared__ uint32_t s_pick;
  const uint32_t tid = threadIdx.x, att = blockIdx.x, n = a.n, k = a.k, dims = a.dims;
  const int4* pts = reinterpret_cast<const int4*>(a.pts);
  unsigned long long* dmin = a.dmin + (size_t)att * n;
  uint8_t* lab = a.lab + (size_t)att * n;
  const uint32_t per = (n + kKmT - 1) / kKmT;
  const uint32_t p0 = min(n, tid * per), p1 = min(n, p0 + per);
  const uint64_t aseed = a.seed ^ ((uint64_t)att << 32);

  // k-means++ seeding: exact integer weights, the draw = the first point whose inclusive prefix exceeds r
  if (tid == 0) s_pick = (uint32_t)(km_hash(aseed) % n);
  __syncthreads();
  if (tid < 4) s_ci[0][tid] = a.pts[(size_t)s_pick * 4 + tid];
  __syncthreads();
  for (uint32_t j = 1; j < k; ++j) {
    const int c[4] = {s_ci[j - 1][0], s_ci[j - 1][1], s_ci[j - 1][2], s_ci[j - 1][3]};
    uint64_t lsum = 0;
    for (uint32_t i = p0; i < p1; ++i) {
      const uint64_t d = km_dist_int(pts[i], c);
      const uint64_t m = j == 1 ? d : min((uint64_t)dmin[i], d);
      dmin[i] = m;
      lsum += m;
    }
    uint64_t total;
    const uint64_t excl = km_block_excl_scan(lsum, s_scan, tid, &total);
    if (total == 0) {  // every point coincides with a centre: the first unused index
      if (tid == 0) s_pick = j < n ? j : 0;
    } else {
      const uint64_t r = km_hash(aseed ^ j) % total;
      if (r >= excl && r - excl < lsum) {  // exactly one thread owns the crossing
        uint64_t acc = excl;
        for (uint32_t i = p0; i < p1; ++i) {
          acc += dmin[i];
          if (acc > r) { s_pick = i; break; }
        }
      }
    }
    __syncthreads();
    if (tid < 4) s_ci[j][tid] = a.pts[(size_t)s_pick * 4 + tid];
    __syncthreads();
  }
  if (tid < k)
    for (uint32_t d = 0; d < 4; ++d) s_c[tid][d] = (double)s_ci[tid][d];
  __syncthreads();

  unsigned long long compact = 0;
  for (uint32_t it = 0;; ++it) {
    if (tid < k) { s_cnt[tid] = 0; s_sum[tid][0] = 0; s_sum[tid][1] = 0; s_sum[tid][2] = 0; s_sum[tid][3] = 0; }
    if (tid == 0) s_compact = 0;
    __syncthreads();
    unsigned long long lc = 0;
    for (uint32_t i = p0; i < p1; ++i) {
      const int4 p = pts[i];
      double best = km_dist_dbl(p, s_c[0], dims);
      uint32_t bj = 0;
      for (uint32_t j = 1; j < k; ++j) {
        const double d = km_dist_dbl(p, s_c[j], dims);
        if (d < best) { best = d; bj = j; }  // ties -> the lowest cluster index
      }
      lab[i] = (uint8_t)bj;
      atomicAdd(&s_cnt[bj], 1u);
      atomicAdd(&s_sum[bj][0], (unsigned long long)(long long)p.x);
      atomicAdd(&s_sum[bj][1], (unsigned long long)(long long)p.y);
      atomicAdd(&s_sum[bj][2], (unsigned long long)(long long)p.z);
      atomicAdd(&s_sum[bj][3], (unsigned long long)(long long)p.w);
      lc += (unsigned long long)(best * 256.0);  // fixed point: exact, order-independent
    }
    if (lc) atomicAdd(&s_compact, lc);
    __syncthreads();
    compact = s_compact;
    if (it + 1 >= a.max_iter) break;  // COUNT criterion
    if (tid < k) {
      double s = 0.0;
      if (s_cnt[tid]) {  // an empty cluster keeps its centre
        for (uint32_t d = 0; d < dims; ++d) {
          const double nc = (double)(long long)s_sum[tid][d] / (double)s_cnt[tid];
          const double t = nc - s_c[tid][d];
          s = s + t * t;
          s_c[tid][d] = nc;
        }
      }
      s_shift[tid] = s;
    }
    __syncthreads();
    double shift = 0.0;
    for (uint32_t j = 0; j < k; ++j) shift = s_shift[j] > shift ? s_shift[j] : shift;
    if (shift <= a.eps2) break;  // EPS criterion: the labels of this pass stand (block-uniform)
    __syncthreads();             // s_shift / s_cnt are rewritten by the next pass
  }
  if (tid == 0) a.compact[att] = compact;
}

__global__ __launch_bounds__(256) void km_pick_kernel(KmArgs a) {
  uint32_t best = 0;
  unsigned long long best_c = a.compact[0];
  for (uint32_t t = 1; t < a.attempts; ++t) {
    const unsigned long long c = a.compact[t];
    if (c < best_c) { best_c = c; best = t; }  // ties -> the earlier attempt
  }
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < a.n) a.out[i] = (int32_t)a.lab[(size_t)best * a.n + i];
  if (i == 0) *a.out_compact = (double)best_c / 256.0;
}

// ---- connected components -------------------------------------------------------------------------------------------
constexpr uint32_t kCcT = 1024;

struct CcArgs {
  const uint8_t* img;
  uint32_t* parent;
  int32_t* labels;
  uint32_t* count;
  uint32_t w, h, n, conn;
};

// One workgroup: union-find over the foreground (a set's root is its first pixel in raster order), roots numbered in
// raster order by one block scan, everything else takes its root's number.
__global__ __launch_bounds__(kCcT) void cc_kernel(CcArgs a) {
  __shared__ uint64_t s_scan[kCcT / 64];
  const uint32_t tid = threadIdx.x, n = a.n, w = a.w;
  for (uint32_t i = tid; i < n; i += kCcT) a.parent[i] = i;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kCcT) {
    if (!a.img[i]) continue;
    const uint32_t y = i / w, x = i - y * w;
    if (x > 0 && a.img[i - 1]) uf_unite(a.parent, i, i - 1);
    if (y > 0) {
      if (a.img[i - w]) uf_unite(a.parent, i, i - w);
      if (a.conn == 8) {
        if (x > 0 && a.img[i - w - 1]) uf_unite(a.parent, i, i - w - 1);
        if (x + 1 < w && a.img[i - w + 1]) uf_unite(a.parent, i, i - w + 1);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kCcT) {
    if (!a.img[i]) continue;
    const uint32_t r = uf_find(a.parent, i);
    if (r != i) a.parent[i] = r;  // still an ancestor for any concurrent walker; roots are never rewritten
  }
  __syncthreads();
  const uint32_t per = (n + kCcT - 1) / kCcT;
  const uint32_t p0 = min(n, tid * per), p1 = min(n, p0 + per);
  uint32_t roots = 0;
  for (uint32_t i = p0; i < p1; ++i) roots += (a.img[i] && a.parent[i] == i) ? 1u : 0u;
  uint64_t total;
  uint32_t rank = (uint32_t)km_block_excl_scan(roots, s_scan, tid, &total);
  for (uint32_t i = p0; i < p1; ++i)
    if (a.img[i] && a.parent[i] == i) a.labels[i] = (int32_t)(++rank);
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kCcT) {
    if (!a.img[i]) a.labels[i] = 0;
    else if (a.parent[i] != i) a.labels[i] = a.labels[a.parent[i]];
  }
  if (tid == 0) *a.count = (uint32_t)total + 1u;  // OpenCV counts the background label
}

}  // namespace svc

using namespace svc;

extern "C" {

int svc_hip_bgr2yuv_host(const uint8_t* bgr, uint32_t w, uint32_t h, uint8_t* yuv) {
  SVC_REQUIRE(bgr && yuv, "bgr2yuv: null pointer");
  const uint64_t pixels = (uint64_t)w * h;
  if (pixels == 0) return SVC_OK;
  SVC_REQUIRE(pixels < (1ull << 30), "bgr2yuv: image of %u x %u is too large", w, h);
  int rc = require_device();
  if (rc) return rc;
  Staging& st = host_stage();
  const size_t bytes = up256(pixels * 3);
  if ((rc = st.ensure(2 * bytes))) return rc;
  host_copy(st.pin, bgr, pixels * 3);
  SVC_HIP_TRY(hipMemcpyAsync(st.dev, st.pin, pixels * 3, hipMemcpyHostToDevice, st.stream));
  hipLaunchKernelGGL(bgr2yuv_kernel, dim3((uint32_t)((pixels + 1023) / 1024)), dim3(256), 0, st.stream, st.dev, st.dev + bytes,
                     pixels);
  if ((rc = check_launch("bgr2yuv_kernel"))) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(st.pin + bytes, st.dev + bytes, pixels * 3, hipMemcpyDeviceToHost, st.stream));
  SVC_HIP_TRY(hipStreamSynchronize(st.stream));
  host_copy(yuv, st.pin + bytes, pixels * 3);
  return SVC_OK;
}

int svc_hip_build_pyramid_host(const uint8_t* level0, uint32_t w, uint32_t h, uint32_t level_count,
                               uint8_t* const* out_levels) {
  SVC_REQUIRE(level0 && out_levels, "build_pyramid: null pointer");
  SVC_REQUIRE(level_count >= 1 && level_count <= 16, "build_pyramid: level count %u", level_count);
  const uint32_t f = 1u << (level_count - 1);
  SVC_REQUIRE(w > 0 && h > 0 && w % f == 0 && h % f == 0, "build_pyramid: %ux%u not divisible by 2^(levels - 1) = %u "
              "(libs/encoder.cpp:164-168 pads to that)", w, h, f);
  for (uint32_t l = 1; l < level_count; ++l) SVC_REQUIRE(out_levels[l], "build_pyramid: null plane at level %u", l);
  if (level_count == 1) return SVC_OK;
  int rc = require_device();
  if (rc) return rc;
  Staging& st = host_stage();
  const size_t pyr = up256(pyramid_bytes(w, h, level_count)), n0 = (size_t)w * h;
  if ((rc = st.ensure(pyr))) return rc;
  host_copy(st.pin, level0, n0);
  SVC_HIP_TRY(hipMemcpyAsync(st.dev, st.pin, n0, hipMemcpyHostToDevice, st.stream));
  if ((rc = launch_pyr_down_levels(st.dev, pyr, 1, w, h, level_count, 0, st.stream))) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(st.pin + n0, st.dev + n0, pyramid_bytes(w, h, level_count) - n0, hipMemcpyDeviceToHost, st.stream));
  SVC_HIP_TRY(hipStreamSynchronize(st.stream));
  size_t off = n0;
  for (uint32_t l = 1; l < level_count; ++l) {
    const size_t n = (size_t)(w >> l) * (h >> l);
    host_copy(out_levels[l], st.pin + off, n);
    off += n;
  }
  return SVC_OK;
}

int svc_hip_morph_rect_host(const uint8_t* src, uint32_t w, uint32_t h, uint32_t kernel_w, uint32_t kernel_h, uint32_t op,
                            uint8_t* dst) {
  SVC_REQUIRE(src && dst, "morph: null pointer");
  SVC_REQUIRE(kernel_w > 0 && kernel_h > 0 && kernel_w < 4096 && kernel_h < 4096, "morph: kernel %ux%u", kernel_w, kernel_h);
  SVC_REQUIRE(op <= SVC_MORPH_CLOSE, "morph: unknown operation %u", op);
  const uint64_t n = (uint64_t)w * h;
  if (n == 0) return SVC_OK;
  SVC_REQUIRE(n < (1ull << 30), "morph: image of %u x %u is too large", w, h);
  int rc = require_device();
  if (rc) return rc;
  Staging& st = host_stage();
  const size_t b = up256(n);
  if ((rc = st.ensure(3 * b))) return rc;
  std::memcpy(st.pin, src, n);
  SVC_HIP_TRY(hipMemcpyAsync(st.dev, st.pin, n, hipMemcpyHostToDevice, st.stream));
  const dim3 grid((uint32_t)((n + 255) / 256)), block(256);
  uint8_t *d0 = st.dev, *d1 = st.dev + b, *d2 = st.dev + 2 * b, *res;
  if (op == SVC_MORPH_ERODE || op == SVC_MORPH_DILATE) {
    hipLaunchKernelGGL(morph_rect_kernel, grid, block, 0, st.stream, d0, d1, (int)w, (int)h, (int)kernel_w, (int)kernel_h,
                       op == SVC_MORPH_DILATE ? 1 : 0);
    res = d1;
  } else {  // open = erode, dilate; close = dilate, erode
    const int first = op == SVC_MORPH_CLOSE ? 1 : 0;
    hipLaunchKernelGGL(morph_rect_kernel, grid, block, 0, st.stream, d0, d1, (int)w, (int)h, (int)kernel_w, (int)kernel_h, first);
    hipLaunchKernelGGL(morph_rect_kernel, grid, block, 0, st.stream, d1, d2, (int)w, (int)h, (int)kernel_w, (int)kernel_h, 1 - first);
    res = d2;
  }
  if ((rc = check_launch("morph_rect_kernel"))) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(st.pin + b, res, n, hipMemcpyDeviceToHost, st.stream));
  SVC_HIP_TRY(hipStreamSynchronize(st.stream));
  std::memcpy(dst, st.pin + b, n);
  return SVC_OK;
}

int svc_hip_kmeans_host(const float* features, uint32_t n, uint32_t dims, uint32_t k, uint32_t attempts, uint32_t max_iter,
                        float epsilon, uint64_t seed, int32_t* labels, double* compactness) {
  SVC_REQUIRE(features && labels, "kmeans: null pointer");
  SVC_REQUIRE(dims >= 1 && dims <= 4, "kmeans: %u coordinates per point (1..4 supported)", dims);
  SVC_REQUIRE(k >= 1 && n >= k, "kmeans: %u clusters for %u points (cv::kmeans asserts N >= K, K > 0)", k, n);
  SVC_REQUIRE(attempts >= 1 && max_iter >= 1 && epsilon > 0.0f, "kmeans: attempts, max_iter and epsilon must be positive "
              "(libs/encoder.cpp:39-61)");
  if (k > kKmMaxK) return fail(SVC_ERR_UNSUPPORTED, "kmeans: cluster count %u exceeds %u", k, kKmMaxK);
  if (attempts > kKmMaxAttempts) return fail(SVC_ERR_UNSUPPORTED, "kmeans: attempt count %u exceeds %u", attempts, kKmMaxAttempts);
  if (n > (1u << 22)) return fail(SVC_ERR_UNSUPPORTED, "kmeans: %u points exceed %u", n, 1u << 22);
  int rc = require_device();
  if (rc) return rc;
  Staging& st = host_stage();
  const size_t feat_b = up256((size_t)n * dims * 4), pts_b = up256((size_t)n * 16), dmin_b = up256((size_t)attempts * n * 8);
  const size_t lab_b = up256((size_t)attempts * n), misc_b = 768, out_b = up256((size_t)n * 4);
  if ((rc = st.ensure(feat_b + pts_b + dmin_b + lab_b + misc_b + out_b))) return rc;
  std::memcpy(st.pin, features, (size_t)n * dims * 4);
  SVC_HIP_TRY(hipMemcpyAsync(st.dev, st.pin, (size_t)n * dims * 4, hipMemcpyHostToDevice, st.stream));
  uint8_t* d = st.dev;
  KmArgs a{};
  a.feat = reinterpret_cast<const float*>(d);
  a.pts = reinterpret_cast<int*>(d + feat_b);
  a.dmin = reinterpret_cast<unsigned long long*>(d + feat_b + pts_b);
  a.lab = d + feat_b + pts_b + dmin_b;
  uint8_t* misc = d + feat_b + pts_b + dmin_b + lab_b;  // [0, 512): compactness per attempt; 512: bad flag; 520: result
  a.compact = reinterpret_cast<unsigned long long*>(misc);
  a.bad = reinterpret_cast<uint32_t*>(misc + 512);
  a.out_compact = reinterpret_cast<double*>(misc + 520);
  a.out = reinterpret_cast<int32_t*>(misc + misc_b);
  a.n = n; a.dims = dims; a.k = k; a.attempts = attempts; a.max_iter = max_iter;
  a.eps2 = (double)epsilon * (double)epsilon;
  a.seed = seed;
  SVC_HIP_TRY(hipMemsetAsync(misc, 0, misc_b, st.stream));
  hipLaunchKernelGGL(km_convert_kernel, dim3(div_up(n, 256)), dim3(256), 0, st.stream, a);
  hipLaunchKernelGGL(km_attempt_kernel, dim3(attempts), dim3(kKmT), 0, st.stream, a);
  hipLaunchKernelGGL(km_pick_kernel, dim3(div_up(n, 256)), dim3(256), 0, st.stream, a);
  if ((rc = check_launch("kmeans kernels"))) return rc;
  const size_t back = misc_b + (size_t)n * 4, pin_off = feat_b;
  SVC_HIP_TRY(hipMemcpyAsync(st.pin + pin_off, misc, back, hipMemcpyDeviceToHost, st.stream));
  SVC_HIP_TRY(hipStreamSynchronize(st.stream));
  uint32_t bad;
  std::memcpy(&bad, st.pin + pin_off + 512, 4);
  if (bad)
    return fail(SVC_ERR_UNSUPPORTED, "kmeans: a feature is not an integer of magnitude below 32768 (this definition of "
                                     "cv::kmeans takes block-matching output and pixel positions)");
  std::memcpy(labels, st.pin + pin_off + misc_b, (size_t)n * 4);
  if (compactness) std::memcpy(compactness, st.pin + pin_off + 520, 8);
  return SVC_OK;
}

int svc_hip_connected_components_host(const uint8_t* image, uint32_t w, uint32_t h, uint32_t connectivity, int32_t* labels,
                                      uint32_t* count) {
  SVC_REQUIRE(image && labels && count, "connected_components: null pointer");
  SVC_REQUIRE(connectivity == 4 || connectivity == 8, "connected_components: connectivity %u (libs/encoder.cpp:92-97: 4 or 8)",
              connectivity);
  const uint64_t n = (uint64_t)w * h;
  if (n == 0) { *count = 1; return SVC_OK; }
  if (n > (1ull << 24)) return fail(SVC_ERR_UNSUPPORTED, "connected_components: image of %u x %u exceeds 2^24 pixels", w, h);
  int rc = require_device();
  if (rc) return rc;
  Staging& st = host_stage();
  const size_t img_b = up256(n), par_b = up256(n * 4), lab_b = up256(n * 4 + 4);
  if ((rc = st.ensure(img_b + par_b + lab_b))) return rc;
  std::memcpy(st.pin, image, n);
  SVC_HIP_TRY(hipMemcpyAsync(st.dev, st.pin, n, hipMemcpyHostToDevice, st.stream));
  CcArgs a{};
  a.img = st.dev;
  a.parent = reinterpret_cast<uint32_t*>(st.dev + img_b);
  a.labels = reinterpret_cast<int32_t*>(st.dev + img_b + par_b);
  a.count = reinterpret_cast<uint32_t*>(st.dev + img_b + par_b + n * 4);
  a.w = w; a.h = h; a.n = (uint32_t)n; a.conn = connectivity;
  // ONE workgroup by design (union_find.hpp, SCOPE): this is the per-call form behind compat/'s cv::connectedComponents, fed MV-field-sized
  // masks (8 160 cells at 1080p, 32 400 at 4K: tens of microseconds).  The size cap above admits far larger images, which then take
  // milliseconds on one CU -- correct, slow, and not the throughput path (the batched form is segment.hip's label kernel).
  hipLaunchKernelGGL(cc_kernel, dim3(1), dim3(kCcT), 0, st.stream, a);
  if ((rc = check_launch("cc_kernel"))) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(st.pin + img_b, a.labels, n * 4 + 4, hipMemcpyDeviceToHost, st.stream));
  SVC_HIP_TRY(hipStreamSynchronize(st.stream));
  std::memcpy(labels, st.pin + img_b, n * 4);
  std::memcpy(count, st.pin + img_b + n * 4, 4);
  return SVC_OK;
}

int svc_hip_dct_tiles_host(float* image, uint32_t w, uint32_t h, uint32_t block_w, uint32_t block_h, const uint32_t* tiles_xy,
                           uint32_t n_tiles) {
  SVC_REQUIRE(image, "dct_tiles: null image");
  SVC_REQUIRE(block_w > 0 && block_h > 0, "dct_tiles: block must be positive (libs/encoder.cpp:325-326)");
  const uint64_t n = (uint64_t)w * h;
  SVC_REQUIRE(n > 0 && n < (1ull << 30), "dct_tiles: image of %u x %u", w, h);
  if (!tiles_xy) {
    SVC_REQUIRE(w % block_w == 0 && h % block_h == 0, "dct_tiles: %ux%u not divisible by the block %ux%u", w, h, block_w, block_h);
    n_tiles = (w / block_w) * (h / block_h);
  } else {
    for (uint32_t t = 0; t < n_tiles; ++t)
      SVC_REQUIRE(tiles_xy[2 * t] + (uint64_t)block_w <= w && tiles_xy[2 * t + 1] + (uint64_t)block_h <= h,
                  "dct_tiles: tile %u at (%u, %u) leaves the %u x %u image", t, tiles_xy[2 * t], tiles_xy[2 * t + 1], w, h);
  }
  if (n_tiles == 0) return SVC_OK;
  int rc = require_device();
  if (rc) return rc;
  Staging& st = host_stage();
  const size_t img_b = up256(n * 4), xy_b = tiles_xy ? up256((size_t)n_tiles * 8) : 0;
  if ((rc = st.ensure(img_b + xy_b))) return rc;
  host_copy(st.pin, image, n * 4);
  if (tiles_xy) std::memcpy(st.pin + img_b, tiles_xy, (size_t)n_tiles * 8);
  SVC_HIP_TRY(hipMemcpyAsync(st.dev, st.pin, img_b + xy_b, hipMemcpyHostToDevice, st.stream));
  rc = launch_dct_tiles(reinterpret_cast<float*>(st.dev), w, h, block_w, block_h,
                        tiles_xy ? reinterpret_cast<const uint32_t*>(st.dev + img_b) : nullptr, n_tiles, st.stream);
  if (rc) return rc;
  SVC_HIP_TRY(hipMemcpyAsync(st.pin, st.dev, n * 4, hipMemcpyDeviceToHost, st.stream));
  SVC_HIP_TRY(hipStreamSynchronize(st.stream));
  host_copy(image, st.pin, n * 4);
  return SVC_OK;
}

}  // extern "C"
