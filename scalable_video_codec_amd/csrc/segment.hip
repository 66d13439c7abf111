// segment.hip -- RANSAC inliers + motion field -> region id per MV block, batched:
// one wavefront per frame.  Reference: libs/encoder.cpp:507-623.
//
// The in-repo steps are the reference's (foreground = complement of the inliers :507-513,
// raster-order foreground list :538-546, BuildMvFeatures' (0, mv.x, x_px, y_px) with its
// mv.y overwrite :316-319, cluster_count = min(K, #foreground) :555, per-cluster
// connected components numbered with `offset += count including label 0` :597-623).
// The OpenCV steps (morphologyEx close/open, kmeans, connectedComponents) cannot be pinned
// offline; they follow this repo's deterministic definitions, stated in
// oracle/svc_segment.c, which this kernel reproduces bit for bit.  Everything that could
// depend on a summation order is exact integer arithmetic (k-means++ weights, centre sums,
// fixed-point compactness), so the parallel reductions here are order-free; the remaining
// floating point is per-element f64 in a fixed operation order (FP contraction is off).
//
// The work is latency-bound (hundreds of short barrier-separated phases over a small field), so it
// is cut to shorten the chain: kernel A runs every k-means attempt of every frame as its own
// workgroup (grid = frames x attempts); kernel B picks the best attempt per frame and does the
// connected components (a lock-free union-find: one merge sweep + one flatten sweep).  Byte
// masks, feature points and union-find parents sit in LDS when they fit, otherwise in the
// caller-provided global workspace (L2-resident).
#include "svc_common.hpp"

namespace svc {

struct SegArgs {
  const uint8_t* mask;  // [frames][n], 1 = RANSAC inlier
  const float* mv;      // [frames][n][2]
  uint32_t* types;      // [frames][n]
  uint8_t* ws;
  uint64_t ws_stride;
  uint64_t seed;
  double eps2;
  uint32_t mfw, mfh, n, mv_bw, mv_bh;
  uint32_t morph_w, morph_h, k, attempts, max_iter, conn;
  uint32_t pts_lds_cap;  // feature points an attempt can keep in LDS; more go to the workspace
  uint32_t small_coords; // host check: x_px, y_px < 2^14, so squared distances fit 32 bits
};

constexpr uint32_t kMaxK = 64;
// Lanes per frame are a template parameter T: 256 when the per-block arrays fit in LDS (measured at
// 1080p: 64 lanes 0.86 ms, 256 lanes 0.39 ms per 64 frames -- the field-sized sweeps need the
// lanes), 1024 when they live in global memory (4K), where sweep throughput is what counts.

__device__ __forceinline__ uint64_t seg_hash(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct Pt { int f[3]; };

__device__ __forceinline__ Pt make_pt(const float2* mv, uint32_t b, uint32_t mfw, uint32_t bw, uint32_t bh) {
  Pt p;
  const float mx = mv[b].x;
  p.f[0] = (int)(mx < 0 ? mx - 0.5f : mx + 0.5f);
  p.f[1] = (int)((b % mfw) * bw);
  p.f[2] = (int)((b / mfw) * bh);
  return p;
}

__device__ __forceinline__ uint64_t dist2_int(const Pt& a, const int* c) {
  uint64_t s = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int64_t t = (int64_t)a.f[d] - c[d];
    s += (uint64_t)(t * t);
  }
  return s;
}

// Same value with 32-bit arithmetic: valid when every coordinate difference is < 2^15 in
// magnitude and |mv.x| differences < 2^15 too (three squares < 2^30 each... the host checks the
// frame size, the kernel checks mv.x), which covers every real frame.
__device__ __forceinline__ uint32_t dist2_u32(const Pt& a, const int* c) {
  const int dx = a.f[0] - c[0], dy = a.f[1] - c[1], dz = a.f[2] - c[2];
  return (uint32_t)(dx * dx) + (uint32_t)(dy * dy) + (uint32_t)(dz * dz);
}

__device__ __forceinline__ double dist2_dbl(const Pt& p, const double* c) {
  const double dx = (double)p.f[0] - c[0], dy = (double)p.f[1] - c[1], dz = (double)p.f[2] - c[2];
  double s = dx * dx;
  s = s + dy * dy;
  s = s + dz * dz;
  return s;
}

// Sum of one int per lane over the wavefront, without touching LDS: four DPP steps leave every
// 16-lane row holding its row sum (quad_perm, quad_perm, row_ror:4, row_ror:8), then the four
// row sums are read back as scalars.  Returns the same value in every lane.
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true);  // row_ror:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);  // row_ror:8
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
         __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// exclusive block scan of one u64 per thread (wave scan by shuffles + 4 wave totals in LDS:
// two barriers); returns this thread's prefix, *total = sum over the block
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int off) {
  const uint32_t lo = __shfl_up((uint32_t)v, off, 64), hi = __shfl_up((uint32_t)(v >> 32), off, 64);
  return ((uint64_t)hi << 32) | lo;
}

template <uint32_t T>
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t* s_scan, uint32_t tid, uint64_t* total) {
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  uint64_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t y = shfl_up_u64(x, off);
    if (lane >= (uint32_t)off) x += y;
  }
  __syncthreads();  // s_scan may still be read from the previous scan
  if (lane == 63) s_scan[wave] = x;
  __syncthreads();
  uint64_t woff = 0, tot = 0;
#pragma unroll
  for (uint32_t wv = 0; wv < T / 64; ++wv) {
    const uint64_t t = s_scan[wv];
    woff += wv < wave ? t : 0;
    tot += t;
  }
  *total = tot;
  return woff + x - v;
}

// Lock-free union-find on `parent` (LDS or global): a set's root is its smallest block index
// (= its first block in raster order); links always go from the larger root to the smaller
// with atomicMin, so concurrent unions commute.
__device__ __forceinline__ uint32_t uf_find(const uint32_t* parent, uint32_t x) {
  uint32_t p = parent[x];
  while (p != x) { x = p; p = parent[x]; }
  return x;
}

__device__ __forceinline__ void uf_unite(uint32_t* parent, uint32_t a, uint32_t b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { const uint32_t t = a; a = b; b = t; }
    const uint32_t old = atomicMin(&parent[a], b);
    if (old == a) return;  // a was still a root and now points at b
    a = old;               // someone linked a first: carry on from where it points
  }
}

template <uint32_t T>
__device__ __forceinline__ void morph_pass(const uint8_t* src, uint8_t* dst, const SegArgs& a, bool dilate,
                                           uint32_t tid) {
  const int ax = (int)a.morph_w / 2, ay = (int)a.morph_h / 2;
  if (a.morph_w == 3 && a.morph_h == 3) {  // the default element: branch-free, loads in flight together
    const int W = (int)a.mfw, H = (int)a.mfh;
    const int pad = dilate ? 0 : 255;
    for (uint32_t i = tid; i < a.n; i += T) {
      const int y = (int)(i / a.mfw), x = (int)(i - (uint32_t)y * a.mfw);
      int v = pad;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int sx = x + dx, sy = y + dy;
          const bool in = sx >= 0 && sy >= 0 && sx < W && sy < H;
          const int p = src[in ? sy * W + sx : (int)i];
          const int q = in ? p : pad;
          v = dilate ? max(v, q) : min(v, q);
        }
      dst[i] = (uint8_t)v;
    }
    __syncthreads();
    return;
  }
  for (uint32_t i = tid; i < a.n; i += T) {
    const int y = (int)(i / a.mfw), x = (int)(i - (uint32_t)y * a.mfw);
    int v = dilate ? 0 : 255;
    for (int ky = 0; ky < (int)a.morph_h; ++ky)
      for (int kx = 0; kx < (int)a.morph_w; ++kx) {
        const int sx = x + kx - ax, sy = y + ky - ay;
        if (sx < 0 || sy < 0 || sx >= (int)a.mfw || sy >= (int)a.mfh) continue;
        const int p = src[sy * (int)a.mfw + sx];
        v = dilate ? max(v, p) : min(v, p);
      }
    dst[i] = (uint8_t)v;
  }
  __syncthreads();
}

// ---- workspace layout (per frame) -----------------------------------------------------------
//   [0, 256)            header: u32 nf at 0; u64 compactness[attempt] at 8 + 8 * attempt
//   idx      [n]  u32   foreground list, raster order (written by attempt 0's workgroup)
//   lab      [A][n] u8  labels of each k-means attempt
//   scratch  [A][2n] u8 + [A][n] Pt: masks / feature points of an attempt when they do not fit LDS
//   cl [n] u8, parent [n] u32: connected-components arrays when they do not fit LDS
constexpr uint32_t kMaxAttempts = 16;

struct Workspace {
  uint8_t* base;
  uint32_t n, attempts;
  __host__ __device__ static uint64_t a16(uint64_t v) { return (v + 15) & ~15ull; }
  __host__ __device__ uint64_t off_idx() const { return 256; }
  __host__ __device__ uint64_t off_lab() const { return off_idx() + a16(4ull * n); }
  __host__ __device__ uint64_t off_masks() const { return off_lab() + a16((uint64_t)attempts * n); }
  __host__ __device__ uint64_t off_pts() const { return off_masks() + a16(2ull * attempts * n); }
  __host__ __device__ uint64_t off_cl() const { return off_pts() + a16(12ull * attempts * n); }
  __host__ __device__ uint64_t off_parent() const { return off_cl() + a16(n); }
  __host__ __device__ uint64_t bytes() const { return (off_parent() + 4ull * n + 255) & ~255ull; }
  __device__ uint32_t* nf() const { return reinterpret_cast<uint32_t*>(base); }
  __device__ unsigned long long* compact() const { return reinterpret_cast<unsigned long long*>(base + 8); }
  __device__ uint32_t* idx() const { return reinterpret_cast<uint32_t*>(base + off_idx()); }
  __device__ uint8_t* lab(uint32_t a) const { return base + off_lab() + (uint64_t)a * n; }
  __device__ uint8_t* masks(uint32_t a) const { return base + off_masks() + 2ull * a * n; }
  __device__ Pt* pts(uint32_t a) const { return reinterpret_cast<Pt*>(base + off_pts() + 12ull * a * n); }
  __device__ uint8_t* cl() const { return base + off_cl(); }
  __device__ uint32_t* parent() const { return reinterpret_cast<uint32_t*>(base + off_parent()); }
};

constexpr uint32_t kPtsLds = 1024;  // feature points kept in LDS by an attempt at 1080p (12 KB, 5 workgroups/CU)

// Kernel A: one workgroup per (frame, k-means attempt).  Attempts are independent restarts
// (cv::kmeans' `attempts`), so they run side by side instead of one after the other; each
// rebuilds the (cheap) mask + foreground list for itself.
// LDS_ARRAYS: the two byte masks live in dynamic LDS (2 B per MV block), else in the workspace.
template <bool LDS_ARRAYS, uint32_t T>
__global__ __launch_bounds__(T) void segment_attempt_kernel(SegArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
  __shared__ uint64_t s_scan[T / 64];
  __shared__ int s_cint[kMaxK][3];
  __shared__ double s_c[kMaxK][3];
  __shared__ unsigned long long s_sum[kMaxK][3];
  __shared__ uint32_t s_cnt[kMaxK];
  __shared__ double s_shift[kMaxK];
  __shared__ unsigned long long s_compact;
  __shared__ uint32_t s_pick;

  const uint32_t tid = threadIdx.x, frame = blockIdx.x, att = blockIdx.y, n = a.n;
  const uint8_t* mask = a.mask + (size_t)frame * n;
  const float2* mv = reinterpret_cast<const float2*>(a.mv) + (size_t)frame * n;
  const Workspace ws{a.ws + (size_t)frame * a.ws_stride, n, a.attempts};
  const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
  uint8_t* fg = LDS_ARRAYS ? dyn_lds : ws.masks(att);
  uint8_t* tmp = LDS_ARRAYS ? dyn_lds + n4 : ws.masks(att) + n;
  uint8_t* lab = ws.lab(att);
  uint32_t* idx = ws.idx();
  const uint64_t seed = a.seed + frame;

  // ---- foreground mask, close, open (encoder.cpp:507-527) ---------------------------------
  for (uint32_t i = tid; i < n; i += T) fg[i] = mask[i] ? 0 : 255;
  __syncthreads();
  morph_pass<T>(fg, tmp, a, true, tid);
  morph_pass<T>(tmp, fg, a, false, tid);
  morph_pass<T>(fg, tmp, a, false, tid);
  morph_pass<T>(tmp, fg, a, true, tid);

  // ---- foreground list in raster order (:538-546) -> feature points (:300-321) ------------
  const uint32_t per = (n + T - 1) / T;
  const uint32_t c0 = min(n, tid * per), c1 = min(n, c0 + per);
  uint32_t local = 0;
  for (uint32_t i = c0; i < c1; ++i) local += fg[i] == 255 ? 1u : 0u;
  uint64_t tot64;
  uint32_t pos = (uint32_t)block_excl_scan<T>(local, s_scan, tid, &tot64);
  const uint32_t nf = (uint32_t)tot64;
  Pt* pts_lds = reinterpret_cast<Pt*>(dyn_lds + (LDS_ARRAYS ? 2 * n4 : 0));
  Pt* pts = nf <= a.pts_lds_cap ? pts_lds : ws.pts(att);
  for (uint32_t i = c0; i < c1; ++i)
    if (fg[i] == 255) {
      if (att == 0) idx[pos] = i;
      pts[pos] = make_pt(mv, i, a.mfw, a.mv_bw, a.mv_bh);
      ++pos;
    }
  if (att == 0 && tid == 0) *ws.nf() = nf;
  __syncthreads();
  if (nf == 0) return;
  const uint32_t k = min(a.k, nf);  // :555

  // ---- one k-means attempt on (mv.x, x_px, y_px) (:557-578) --------------------------------
  const uint32_t pper = (nf + T - 1) / T;
  const uint32_t p0 = min(nf, tid * pper), p1 = min(nf, p0 + pper);
  const uint64_t aseed = seed ^ ((uint64_t)att << 32);
  if (tid == 0) {
    const uint32_t first = (uint32_t)(seg_hash(aseed) % nf);
    const Pt p = pts[first];
    s_cint[0][0] = p.f[0]; s_cint[0][1] = p.f[1]; s_cint[0][2] = p.f[2];
  }
  __syncthreads();
  // k-means++: the next centre is drawn with probability ~ (distance to the nearest chosen
  // centre)^2.  Each point's running minimum is kept across steps (in the LDS that held the
  // masks, which are dead now, or in the workspace), so a step costs ONE new distance per point;
  // 32-bit arithmetic when the coordinates allow (same integers either way).
  uint32_t* dmin32 = reinterpret_cast<uint32_t*>(LDS_ARRAYS && 4 * (size_t)nf <= 2 * n4 ? fg : ws.masks(att));
  bool use32 = a.small_coords != 0 && 4 * (size_t)nf <= 2 * (size_t)n;  // workspace slot is 2n bytes too
  {
    bool ok = true;
    for (uint32_t i = tid; i < nf; i += T) ok = ok && pts[i].f[0] > -8192 && pts[i].f[0] < 8192;
    __shared__ uint32_t s_ok;
    if (tid == 0) s_ok = 1;
    __syncthreads();
    if (!ok) s_ok = 0;
    __syncthreads();
    use32 = use32 && s_ok != 0;
  }
  if (use32)
    for (uint32_t i = p0; i < p1; ++i) dmin32[i] = 0xFFFFFFFFu;
  for (uint32_t j = 1; j < k; ++j) {
    uint64_t lsum = 0;
    if (use32) {
      for (uint32_t i = p0; i < p1; ++i) {  // a lane only ever touches its own contiguous range
        const uint32_t m = min(dmin32[i], dist2_u32(pts[i], s_cint[j - 1]));
        dmin32[i] = m;
        lsum += m;
      }
    } else {
      for (uint32_t i = p0; i < p1; ++i) {
        const Pt p = pts[i];
        uint64_t m = ~0ull;
        for (uint32_t q = 0; q < j; ++q) m = min(m, dist2_int(p, s_cint[q]));
        lsum += m;
      }
    }
    uint64_t total;
    const uint64_t excl = block_excl_scan<T>(lsum, s_scan, tid, &total);
    if (total == 0) {
      if (tid == 0) s_pick = j < nf ? j : 0;
    } else {
      const uint64_t r = seg_hash(aseed ^ j) % total;
      if (r >= excl && r < excl + lsum) {  // exactly one lane owns the crossing
        uint64_t acc = excl;
        for (uint32_t i = p0; i < p1; ++i) {
          uint64_t m;
          if (use32) {
            m = dmin32[i];
          } else {
            const Pt p = pts[i];
            m = ~0ull;
            for (uint32_t q = 0; q < j; ++q) m = min(m, dist2_int(p, s_cint[q]));
          }
          acc += m;
          if (acc > r) { s_pick = i; break; }
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      const Pt p = pts[s_pick];
      s_cint[j][0] = p.f[0]; s_cint[j][1] = p.f[1]; s_cint[j][2] = p.f[2];
    }
    __syncthreads();
  }
  if (tid < k) {
    s_c[tid][0] = (double)s_cint[tid][0];
    s_c[tid][1] = (double)s_cint[tid][1];
    s_c[tid][2] = (double)s_cint[tid][2];
  }
  __syncthreads();

  uint64_t compact = 0;
  for (uint32_t it = 0;; ++it) {  // Lloyd
    if (tid < k) { s_sum[tid][0] = 0; s_sum[tid][1] = 0; s_sum[tid][2] = 0; s_cnt[tid] = 0; }
    if (tid == 0) s_compact = 0;
    __syncthreads();
    unsigned long long lc = 0;
    const uint32_t lane = tid & 63u;
    for (uint32_t i0 = 0; i0 < nf; i0 += T) {  // wave-uniform trip count
      const uint32_t i = i0 + tid;
      const bool active = i < nf;
      Pt p = {{0, 0, 0}};
      uint32_t bj = 0xFFFFFFFFu;
      if (active) {
        p = pts[i];
        double best = dist2_dbl(p, s_c[0]);
        bj = 0;
        for (uint32_t j = 1; j < k; ++j) {
          const double d = dist2_dbl(p, s_c[j]);
          if (d < best) { best = d; bj = j; }
        }
        lab[i] = (uint8_t)bj;
        lc += (unsigned long long)(best * 256.0);
      }
      // per-cluster sums: reduce inside the wave first (DPP, no LDS traffic: same-address LDS
      // atomics from 64 lanes serialise, and shuffle reductions load the LDS crossbar), then ONE
      // LDS atomic per wave and cluster; 64 lanes x |coord| fits int32 for |coord| < 2^24
      for (uint32_t j = 0; j < k; ++j) {
        const bool mine = bj == j;
        const unsigned long long bal = __ballot(mine);
        if (bal == 0) continue;
        const int sx = wave_sum_i32(mine ? p.f[0] : 0), sy = wave_sum_i32(mine ? p.f[1] : 0),
                  sz = wave_sum_i32(mine ? p.f[2] : 0);
        if (lane == 0) {
          atomicAdd(&s_cnt[j], (uint32_t)__popcll(bal));
          atomicAdd(&s_sum[j][0], (unsigned long long)(long long)sx);
          atomicAdd(&s_sum[j][1], (unsigned long long)(long long)sy);
          atomicAdd(&s_sum[j][2], (unsigned long long)(long long)sz);
        }
      }
    }
    atomicAdd(&s_compact, lc);
    __syncthreads();
    compact = s_compact;
    if (it + 1 >= a.max_iter) break;
    if (tid < k) {
      double s = 0.0;
      if (s_cnt[tid]) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
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
    if (shift <= a.eps2) break;
  }
  if (tid == 0) ws.compact()[att] = compact;
}

// Kernel B: one workgroup per frame.  Takes the attempt with the smallest compactness (ties ->
// the earlier attempt), then connected components per cluster, numbered as the reference
// numbers them (:597-623).  LDS_PARENT / LDS_CL: union-find parents (4 B per block) and cluster
// ids (1 B per block) in LDS; at 4K only the parents fit (130 KB).
template <bool LDS_PARENT, bool LDS_CL, uint32_t T>
__global__ __launch_bounds__(T) void segment_label_kernel(SegArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
  __shared__ uint64_t s_scan[T / 64];
  const uint32_t tid = threadIdx.x, frame = blockIdx.x, n = a.n;
  uint32_t* types = a.types + (size_t)frame * n;
  const Workspace ws{a.ws + (size_t)frame * a.ws_stride, n, a.attempts};
  const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
  uint32_t* label = LDS_PARENT ? reinterpret_cast<uint32_t*>(dyn_lds) : ws.parent();
  uint8_t* cl = LDS_CL ? dyn_lds + 4 * n4 : ws.cl();
  uint32_t* idx = ws.idx();
  const uint32_t nf = *ws.nf();

  for (uint32_t i = tid; i < n; i += T) { types[i] = 0; cl[i] = 255; label[i] = i; }  // :549-551
  if (nf == 0) return;
  const uint32_t k = min(a.k, nf);
  uint32_t best = 0;
  unsigned long long best_c = ws.compact()[0];
  for (uint32_t t = 1; t < a.attempts; ++t) {
    const unsigned long long c = ws.compact()[t];
    if (c < best_c) { best_c = c; best = t; }
  }
  const uint8_t* best_lab = ws.lab(best);
  __syncthreads();
  for (uint32_t i = tid; i < nf; i += T) cl[idx[i]] = best_lab[i];
  __syncthreads();
  // merge every block with its already-visited neighbours of the same cluster, then flatten
  for (uint32_t i = tid; i < n; i += T) {
    const uint8_t c = cl[i];
    if (c == 255) continue;
    const int y = (int)(i / a.mfw), x = (int)(i - (uint32_t)y * a.mfw);
    if (x > 0 && cl[i - 1] == c) uf_unite(label, i, i - 1);
    if (y > 0) {
      if (cl[i - a.mfw] == c) uf_unite(label, i, i - a.mfw);
      if (a.conn == 8) {
        if (x > 0 && cl[i - a.mfw - 1] == c) uf_unite(label, i, i - a.mfw - 1);
        if (x + 1 < (int)a.mfw && cl[i - a.mfw + 1] == c) uf_unite(label, i, i - a.mfw + 1);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += T)
    if (cl[i] != 255) {
      const uint32_t r = uf_find(label, i);
      if (r != i) label[i] = r;  // still an ancestor for any concurrent walker; roots are never rewritten
    }
  __syncthreads();
  const uint32_t per = (n + T - 1) / T;
  const uint32_t c0 = min(n, tid * per), c1 = min(n, c0 + per);
  uint32_t offset = 0;  // BLOCK_TYPE_BACKGROUND, libs/codec.hpp:6
  for (uint32_t cid = 0; cid < k; ++cid) {
    uint32_t roots = 0;
    for (uint32_t i = c0; i < c1; ++i) roots += (cl[i] == cid && label[i] == i) ? 1u : 0u;
    uint64_t total;
    uint32_t rank = (uint32_t)block_excl_scan<T>(roots, s_scan, tid, &total);
    for (uint32_t i = c0; i < c1; ++i)
      if (cl[i] == cid && label[i] == i) idx[i] = ++rank;  // idx is free now: component number of a root
    __syncthreads();
    for (uint32_t i = tid; i < n; i += T)
      if (cl[i] == cid) types[i] = idx[label[i]] + offset;  // :617
    offset += (uint32_t)total + 1;  // :620, the count includes label 0
    __syncthreads();
  }
}

uint64_t segment_workspace_per_frame(uint32_t n, uint32_t attempts) {
  return Workspace{nullptr, n, attempts ? attempts : 1}.bytes();
}

int launch_segment(const uint8_t* d_mask, const float* d_mv, uint32_t mfw, uint32_t mfh, uint32_t n_frames,
                   uint32_t mv_bw, uint32_t mv_bh, const svc_segment_params& p, uint64_t seed, uint8_t* d_ws,
                   uint32_t* d_types, hipStream_t stream) {
  if (n_frames == 0) return SVC_OK;
  if (p.cluster_count > kMaxK)
    return fail(SVC_ERR_UNSUPPORTED, "segment: cluster_count %u exceeds %u", p.cluster_count, kMaxK);
  if (p.attempt_count > kMaxAttempts)
    return fail(SVC_ERR_UNSUPPORTED, "segment: attempt_count %u exceeds %u", p.attempt_count, kMaxAttempts);
  SegArgs a;
  a.mask = d_mask;
  a.mv = d_mv;
  a.types = d_types;
  a.ws = d_ws;
  a.mfw = mfw; a.mfh = mfh; a.n = mfw * mfh;
  a.ws_stride = segment_workspace_per_frame(a.n, p.attempt_count);
  a.seed = seed;
  a.eps2 = (double)p.epsilon * (double)p.epsilon;
  a.mv_bw = mv_bw; a.mv_bh = mv_bh;
  a.morph_w = p.morph_rect_w; a.morph_h = p.morph_rect_h;
  a.k = p.cluster_count; a.attempts = p.attempt_count; a.max_iter = p.max_iter_count;
  a.conn = p.connectivity;
  const size_t n4 = ((size_t)a.n + 3) & ~(size_t)3;
  const size_t pts_lds = (size_t)kPtsLds * sizeof(Pt);
  const dim3 grid_a(n_frames, p.attempt_count);
  a.small_coords = ((uint64_t)mfw * mv_bw < (1u << 14) && (uint64_t)mfh * mv_bh < (1u << 14)) ? 1u : 0u;
  constexpr size_t kLdsMax = 152 * 1024;  // of the CU's 160 KB; statics take ~4 KB
  a.pts_lds_cap = kPtsLds;
  if (5 * n4 <= 100 * 1024) {  // small fields (1080p: 8 160 blocks): everything in LDS, 256 lanes
    hipLaunchKernelGGL((segment_attempt_kernel<true, 256>), grid_a, dim3(256), 2 * n4 + pts_lds, stream, a);
    hipLaunchKernelGGL((segment_label_kernel<true, true, 256>), dim3(n_frames), dim3(256), 5 * n4, stream, a);
  } else {  // big fields (4K: 32 400 blocks): 1024 lanes, LDS for whatever fits
    if (2 * n4 + pts_lds <= kLdsMax) {  // one workgroup per CU anyway: give the points the rest of the LDS
      a.pts_lds_cap = (uint32_t)((kLdsMax - 2 * n4) / sizeof(Pt));
      hipLaunchKernelGGL((segment_attempt_kernel<true, 1024>), grid_a, dim3(1024), 2 * n4 + a.pts_lds_cap * sizeof(Pt), stream, a);
    } else {
      hipLaunchKernelGGL((segment_attempt_kernel<false, 1024>), grid_a, dim3(1024), pts_lds, stream, a);
    }
    if (4 * n4 <= kLdsMax - 4096)
      hipLaunchKernelGGL((segment_label_kernel<true, false, 1024>), dim3(n_frames), dim3(1024), 4 * n4, stream, a);
    else
      hipLaunchKernelGGL((segment_label_kernel<false, false, 1024>), dim3(n_frames), dim3(1024), 0, stream, a);
  }
  return check_launch("segment kernels");
}

}  // namespace svc
