// segment.hip -- RANSAC inliers + motion field -> region id per MV block, batched over frames.
// Reference: libs/encoder.cpp:507-623.
//
// The in-repo steps are the reference's (foreground = complement of the inliers :507-513,
// raster-order foreground list :538-546, BuildMvFeatures' (0, mv.x, x_px, y_px) with its
// mv.y overwrite :316-319, cluster_count = min(K, #foreground) :555, per-cluster
// connected components numbered with `offset += count including label 0` :597-623).
// The OpenCV steps (morphologyEx close/open, kmeans, connectedComponents) cannot be pinned
// offline; they follow this repo's deterministic definitions, stated in
// oracle/svc_segment.c, which these kernels reproduce bit for bit.  Everything that could
// depend on a summation order is exact integer arithmetic (k-means++ weights, centre sums,
// fixed-point compactness), so the parallel reductions here are order-free; the remaining
// floating point is per-element f64 in a fixed operation order (FP contraction is off).
//
// The work is latency-bound (hundreds of short barrier-separated phases over a small field) and
// very uneven (most frames have a few hundred foreground blocks, a scene cut has most of the
// field), so it is cut into kernels that each get the width they need:
//   P  one workgroup per frame, 1024 lanes: mask -> bit rows, close/open as word operations,
//      foreground list + feature points packed into 32 bits;
//   A  one workgroup per (frame, k-means attempt), launched 256 lanes wide for the light frames
//      and 1024 wide for the heavy ones, the points of a lane in registers throughout;
//   B  one workgroup per frame: best attempt, connected components (horizontal runs by ballot,
//      then a lock-free union-find over the few links the runs do not imply), numbering.
// Lists, labels and whatever does not fit LDS sit in the caller-provided workspace (L2-resident).
#include <algorithm>

#include "svc_common.hpp"
#include "union_find.hpp"

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
  uint32_t lds_bytes;    // dynamic LDS of the attempt kernel
  uint32_t bits_bytes;   // ... of which the bit fields at its start
  uint32_t packable;     // host check: field <= 512 x 512 blocks and x_px, y_px < 2^14 (32-bit distances)
  uint32_t take_all;     // the 256-lane attempt kernel also takes the heavy frames (no 1024-lane launch)
  uint32_t wide_g;       // > 0: the attempts of every frame of packed points run as launch sequences over wide_g workgroups
                         // each (the one attempt launch in front of them then takes the frames of unpacked points only)
  uint32_t wide_step;    // which launch of the sequence this is
};

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

__device__ __forceinline__ uint64_t dist2_int(const Pt& a, const int* c) {
  uint64_t s = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int64_t t = (int64_t)a.f[d] - c[d];
    s += (uint64_t)(t * t);
  }
  return s;
}

// Same value with 32-bit arithmetic, for packed points: every coordinate difference is below 2^15
// in magnitude (x_px, y_px < 2^14 checked on the host, |mv.x| < 2^13 by the kernel that packs), so
// the squares are full-rate 24-bit multiplies and their sum stays below 2^32.
__device__ __forceinline__ uint32_t dist2_u32(const Pt& a, const int* c) {
  const int dx = a.f[0] - c[0], dy = a.f[1] - c[1], dz = a.f[2] - c[2];
  return (uint32_t)__mul24(dx, dx) + (uint32_t)__mul24(dy, dy) + (uint32_t)__mul24(dz, dz);
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

// exclusive block scan of one u64 per thread over the first nw waves of the workgroup (wave scan by
// shuffles + nw wave totals in LDS: two barriers); returns this thread's prefix, *total = block sum
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t* s_scan, uint32_t tid, uint32_t nw,
                                                    uint64_t* total) {
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
  for (uint32_t wv = 0; wv < nw; ++wv) {
    const uint64_t t = s_scan[wv];
    woff += wv < wave ? t : 0;
    tot += t;
  }
  *total = tot;
  return woff + x - v;
}

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, uint32_t lane) {
  uint64_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t y = shfl_up_u64(x, off);
    if (lane >= (uint32_t)off) x += y;
  }
  return x;
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t l) {
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

constexpr uint32_t kTA = 1024;  // lanes both kernels are launched with

// ---- morphology on bit rows ------------------------------------------------------------------
// The field is kept as one bit per MV block, each row starting at a word boundary (W32 words per
// row, bit x & 31 of word x >> 5; bits past the row's end are 0).  A rectangular structuring
// element separates into a horizontal and a vertical pass; positions outside the field are
// ignored, exactly like the byte version this replaces (dilate pads with 0, erode with 1).
struct BitField {
  uint32_t W32, H, NW, last_valid;  // words per row, rows, W32 * H, valid bits of a row's last word
};

__device__ __forceinline__ uint32_t bit_row_word(const uint32_t* row, int w, const BitField& bf, bool erode) {
  if (w < 0 || w >= (int)bf.W32) return erode ? ~0u : 0u;
  uint32_t v = row[w];
  if (erode && w == (int)bf.W32 - 1) v |= ~bf.last_valid;
  return v;
}

// dst = (erode ? AND : OR) over the kw x kh rectangle anchored at (kw / 2, kh / 2); tmp is scratch
__device__ __forceinline__ void bit_morph(const uint32_t* src, uint32_t* tmp, uint32_t* dst, const BitField& bf,
                                          uint32_t kw, uint32_t kh, bool erode, uint32_t tid) {
  const int ax = (int)kw / 2, ay = (int)kh / 2;
  for (uint32_t q = tid; q < bf.NW; q += kTA) {
    const uint32_t y = q / bf.W32;
    const int w = (int)(q - y * bf.W32);
    const uint32_t* row = src + y * bf.W32;
    uint32_t acc = erode ? ~0u : 0u;
    for (int kx = 0; kx < (int)kw; ++kx) {
      const int s = kx - ax, o = s >> 5, b = s & 31;  // out bit x takes in bit x + s = 32 o + b further on
      const uint32_t lo = bit_row_word(row, w + o, bf, erode);
      const uint32_t v = b ? (lo >> b) | (bit_row_word(row, w + o + 1, bf, erode) << (32 - b)) : lo;
      acc = erode ? acc & v : acc | v;
    }
    tmp[q] = w == (int)bf.W32 - 1 ? acc & bf.last_valid : acc;
  }
  __syncthreads();
  for (uint32_t q = tid; q < bf.NW; q += kTA) {
    const int y = (int)(q / bf.W32);
    uint32_t acc = erode ? ~0u : 0u;
    for (int ky = 0; ky < (int)kh; ++ky) {
      const int sy = y + ky - ay;
      if (sy < 0 || sy >= (int)bf.H) continue;
      const uint32_t v = tmp[(int)q + (ky - ay) * (int)bf.W32];
      acc = erode ? acc & v : acc | v;
    }
    const uint32_t w = q - (uint32_t)y * bf.W32;
    dst[q] = w == bf.W32 - 1 ? acc & bf.last_valid : acc;
  }
  __syncthreads();
}

// ---- workspace layout (per frame) -----------------------------------------------------------
//   [0, 256)            header: u32 nf at 0; u32 "points are packed" at 4; u64 compactness[attempt] at 8 + 8 * attempt
//   idx      [n]  u32   foreground list, raster order
//   pk       [n]  u32   packed feature points, list order
//   lab      [A][n] u8  labels of each k-means attempt
//   pts      [n] Pt     unpacked feature points (only for frames whose points cannot be packed)
//   dmin     [A][n] u32 k-means++ running minima when they do not fit registers or LDS
//   cl [n] u8, parent [n] u32: connected-components arrays when they do not fit LDS
//   roots    [n] u32    component roots in raster order
//   dmin2    [A][n] u32 second set of running minima: the multi-launch form reads step j - 1's while it writes step j's
//   wide     [A] WideState   the multi-launch form of an attempt (segment_wide_*_kernel): centres, partial sums per
//                            workgroup, the "attempt is over" flag
constexpr uint32_t kMaxAttempts = 16;
constexpr uint32_t kMaxK = 64;
constexpr uint32_t kWideMaxG = 32;  // workgroups one attempt can be spread over

// State of one attempt that runs as a sequence of launches (see segment_wide_seed_kernel).  Everything a launch reads was
// written by an EARLIER launch of the same stream; within a launch every workgroup recomputes what it needs from it.
struct WideState {
  int cint[kMaxK][3];                               // k-means++ centres as drawn
  double c[2][kMaxK][3];                            // Lloyd centres, double-buffered by iteration parity
  unsigned long long seed_sum[kMaxK][kWideMaxG];    // seeding step j: sum of the running minima per workgroup
  long long part[2][kWideMaxG][kMaxK][4];           // Lloyd: count, sum mv.x, sum column, sum row per workgroup and cluster
  unsigned long long compact[2][kWideMaxG];         // Lloyd: fixed-point compactness per workgroup
  uint32_t done;                                    // 0, or the Lloyd launch step that found the attempt over (>= 1): ws.compact()[att] is
                                                    // final; only launches of a LATER step leave on it (see segment_wide_lloyd_kernel)
  uint32_t pad[3];
};

struct Workspace {
  uint8_t* base;
  uint32_t n, attempts;
  __host__ __device__ static uint64_t a16(uint64_t v) { return (v + 15) & ~15ull; }
  __host__ __device__ uint64_t off_idx() const { return 256; }
  __host__ __device__ uint64_t off_pk() const { return off_idx() + a16(4ull * n); }
  __host__ __device__ uint64_t off_lab() const { return off_pk() + a16(4ull * n); }
  __host__ __device__ uint64_t off_pts() const { return off_lab() + a16((uint64_t)attempts * n); }
  __host__ __device__ uint64_t off_dmin() const { return off_pts() + a16(12ull * n); }
  __host__ __device__ uint64_t off_cl() const { return off_dmin() + a16(4ull * attempts * n); }
  __host__ __device__ uint64_t off_parent() const { return off_cl() + a16(n); }
  __host__ __device__ uint64_t off_roots() const { return off_parent() + a16(4ull * n); }
  __host__ __device__ uint64_t off_dmin2() const { return (off_roots() + 4ull * n + 255) & ~255ull; }
  __host__ __device__ uint64_t off_wide() const { return (off_dmin2() + 4ull * attempts * n + 255) & ~255ull; }
  __host__ __device__ uint64_t bytes() const { return (off_wide() + (uint64_t)attempts * sizeof(WideState) + 255) & ~255ull; }
  __device__ uint32_t* nf() const { return reinterpret_cast<uint32_t*>(base); }
  __device__ uint32_t* packed() const { return reinterpret_cast<uint32_t*>(base + 4); }
  __device__ unsigned long long* compact() const { return reinterpret_cast<unsigned long long*>(base + 8); }
  __device__ uint32_t* idx() const { return reinterpret_cast<uint32_t*>(base + off_idx()); }
  __device__ uint32_t* pk() const { return reinterpret_cast<uint32_t*>(base + off_pk()); }
  __device__ uint8_t* lab(uint32_t a) const { return base + off_lab() + (uint64_t)a * n; }
  __device__ Pt* pts() const { return reinterpret_cast<Pt*>(base + off_pts()); }
  __device__ uint32_t* dmin(uint32_t a) const { return reinterpret_cast<uint32_t*>(base + off_dmin() + 4ull * a * n); }
  __device__ uint8_t* cl() const { return base + off_cl(); }
  __device__ uint32_t* parent() const { return reinterpret_cast<uint32_t*>(base + off_parent()); }
  __device__ uint32_t* roots() const { return reinterpret_cast<uint32_t*>(base + off_roots()); }
  __device__ uint32_t* dmin2(uint32_t a) const { return reinterpret_cast<uint32_t*>(base + off_dmin2() + 4ull * a * n); }
  __device__ WideState* wide(uint32_t a) const { return reinterpret_cast<WideState*>(base + off_wide()) + a; }
};

// The labelling kernel is launched kTA lanes wide, which is what its field-sized sweeps (clears, runs)
// want; once those are done the workgroup keeps only the lanes the frame's foreground count can feed:
// the rest of its waves end there (s_barrier counts surviving waves only).
__device__ __forceinline__ uint32_t lanes_for(uint32_t nf) { return nf <= 64 ? 64u : nf <= 1024 ? 256u : kTA; }


struct KmLds {
  uint64_t scan[kTA / 64];
  int cint[kMaxK][3];
  double c[kMaxK][3];
  unsigned long long sum[kMaxK][3];  // unpacked path
  uint32_t cnt[kMaxK];
  unsigned long long acc[kMaxK][4];  // packed paths: see lloyd_quad
  double shift[kMaxK];
  unsigned long long compact;      // unpacked path
  unsigned long long compact2[2];  // packed paths: iteration it adds into [it & 1]
  uint64_t draw;
  uint32_t pick, bad;
};

// A feature point in 32 bits: MV-block column (9 bits), row (9 bits), rounded mv.x (14 bits, signed).
// Used when the field is at most 512 x 512 blocks and |mv.x| < 8192 -- every real frame.
__device__ __forceinline__ uint32_t pack_pt(int mvx, uint32_t bx, uint32_t by) {
  return bx | (by << 9) | ((uint32_t)mvx << 18);
}
__device__ __forceinline__ Pt unpack_pt(uint32_t v, uint32_t bw, uint32_t bh) {
  Pt p;
  p.f[0] = (int)v >> 18;
  p.f[1] = (int)((v & 511u) * bw);
  p.f[2] = (int)(((v >> 9) & 511u) * bh);
  return p;
}
__device__ __forceinline__ void set_centre(KmLds& L, uint32_t j, uint32_t v, uint32_t bw, uint32_t bh) {
  const Pt p = unpack_pt(v, bw, bh);
  L.cint[j][0] = p.f[0]; L.cint[j][1] = p.f[1]; L.cint[j][2] = p.f[2];
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, off, 64), hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}

// ---- Lloyd iteration pieces shared by the packed paths -----------------------------------------
// Four points of a lane against all centres: nearest centre (strict <: the lowest index wins ties),
// fixed-point compactness, and the correction of the per-cluster sums.  The sums of (1, mv.x + 8192,
// column, row) are all integers, so they are kept across iterations and only corrected for the
// points whose label changed: after the first couple of iterations few lanes have anything to add.
// A cluster has four 64-bit accumulators, two fields each: everything that ever entered it
// (count | mv.x sum << 32, column sum | row sum << 32) and everything that ever left it; the fields
// only grow between two centre updates (which re-base them), so the low ones stay far below 2^32; the
// high ones may wrap, and entered - left per field mod 2^32 is the exact current sum.  oldj = 0xFF: no
// label yet.
__device__ __forceinline__ void lloyd_quad(const uint32_t (&v)[4], const bool (&act)[4], const uint32_t (&oldj)[4],
                                           uint32_t (&bj)[4], uint64_t& lc, KmLds& L, uint32_t k, uint32_t bw,
                                           uint32_t bh, uint32_t lane) {
  double px[4], py[4], pz[4], best[4];
  {
    const double c0 = L.c[0][0], c1 = L.c[0][1], c2 = L.c[0][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const Pt p = unpack_pt(v[u], bw, bh);
      px[u] = (double)p.f[0]; py[u] = (double)p.f[1]; pz[u] = (double)p.f[2];
      const double dx = px[u] - c0, dy = py[u] - c1, dz = pz[u] - c2;
      double d = dx * dx;
      d = d + dy * dy;
      d = d + dz * dz;
      best[u] = d;
      bj[u] = 0;
    }
  }
  // four independent chains per centre; the next centre is fetched from LDS while this one is used
  double n0 = L.c[k > 1 ? 1 : 0][0], n1 = L.c[k > 1 ? 1 : 0][1], n2 = L.c[k > 1 ? 1 : 0][2];
  for (uint32_t j = 1; j < k; ++j) {
    const double a0 = n0, a1 = n1, a2 = n2;
    const uint32_t jn = j + 1 < k ? j + 1 : j;
    n0 = L.c[jn][0]; n1 = L.c[jn][1]; n2 = L.c[jn][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double ax = px[u] - a0, ay = py[u] - a1, az = pz[u] - a2;
      double da = ax * ax;
      da = da + ay * ay;
      da = da + az * az;
      bj[u] = da < best[u] ? j : bj[u];        // strict <: the lowest index keeps a tie
      best[u] = __builtin_fmin(best[u], da);  // one v_min_f64 instead of two selects (no NaNs here)
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (act[u]) lc += (unsigned long long)(best[u] * 256.0);
    if (act[u] && bj[u] != oldj[u]) {
      const unsigned long long wa = 1ull | ((unsigned long long)(uint32_t)(((int)v[u] >> 18) + 8192) << 32);
      const unsigned long long wb = (unsigned long long)(v[u] & 511u) | ((unsigned long long)((v[u] >> 9) & 511u) << 32);
      atomicAdd(&L.acc[bj[u]][0], wa);
      atomicAdd(&L.acc[bj[u]][1], wb);
      if (oldj[u] != 0xFFu) {
        atomicAdd(&L.acc[oldj[u]][2], wa);
        atomicAdd(&L.acc[oldj[u]][3], wb);
      }
    }
  }
}

__device__ __forceinline__ void lloyd_begin(KmLds& L, uint32_t k, uint32_t tid) {
  if (tid < k) {
    L.c[tid][0] = (double)L.cint[tid][0];
    L.c[tid][1] = (double)L.cint[tid][1];
    L.c[tid][2] = (double)L.cint[tid][2];
    L.acc[tid][0] = 0; L.acc[tid][1] = 0; L.acc[tid][2] = 0; L.acc[tid][3] = 0;
  }
  if (tid == 0) { L.compact2[0] = 0; L.compact2[1] = 0; }
  __syncthreads();
}

// Closes iteration `it`: publishes the compactness, moves the centres to the means of their clusters;
// true when the attempt is over (iteration cap, or no centre moved further than epsilon).
__device__ __forceinline__ bool lloyd_end_iter(uint64_t lc, uint64_t& compact, uint32_t it, KmLds& L, const SegArgs& a,
                                               uint32_t k, uint32_t tid, uint32_t lane) {
  lc = wave_sum_u64(lc);
  if (lane == 0) atomicAdd(&L.compact2[it & 1], (unsigned long long)lc);
  __syncthreads();
  compact = L.compact2[it & 1];
  if (it + 1 >= a.max_iter) return true;
  if (tid == 0) L.compact2[(it + 1) & 1] = 0;  // last read one iteration ago
  if (tid < k) {
    double s = 0.0;
    const unsigned long long ia = L.acc[tid][0], ib = L.acc[tid][1], oa = L.acc[tid][2], ob = L.acc[tid][3];
    const uint32_t cnt = (uint32_t)ia - (uint32_t)oa;
    const uint32_t smv = (uint32_t)(ia >> 32) - (uint32_t)(oa >> 32), sbx = (uint32_t)ib - (uint32_t)ob,
                   sby = (uint32_t)(ib >> 32) - (uint32_t)(ob >> 32);
    // re-base, so that the low fields never grow past two iterations' worth whatever max_iter is
    L.acc[tid][0] = (unsigned long long)cnt | ((unsigned long long)smv << 32);
    L.acc[tid][1] = (unsigned long long)sbx | ((unsigned long long)sby << 32);
    L.acc[tid][2] = 0;
    L.acc[tid][3] = 0;
    if (cnt) {
      const long long sums[3] = {(long long)smv - 8192ll * cnt, (long long)sbx * a.mv_bw, (long long)sby * a.mv_bh};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double nc = (double)sums[d] / (double)cnt;
        const double t = nc - L.c[tid][d];
        s = s + t * t;
        L.c[tid][d] = nc;
      }
    }
    L.shift[tid] = s;
  }
  __syncthreads();
  double shift = 0.0;
  for (uint32_t j = 0; j < k; ++j) shift = L.shift[j] > shift ? L.shift[j] : shift;
  return shift <= a.eps2;
}

// One k-means attempt with the lane's points in registers: lane t owns the list entries
// [t * pper, (t + 1) * pper), pper <= P.  `pk` is only read once, and for the drawn centres.
// The k-means++ seeding of an attempt with the lane's points in registers (v: filled here and kept for the caller): on
// return L.cint[0 .. k - 1] are the drawn centres (a barrier has passed).
template <uint32_t P>
__device__ __forceinline__ void seed_regs(const uint32_t* pk, uint32_t (&v)[P], KmLds& L, const SegArgs& a, uint32_t nf,
                                          uint32_t k, uint64_t aseed, uint32_t tid, uint32_t te) {
  const uint32_t nw = te >> 6, wave = tid >> 6, lane = tid & 63u;
  const uint32_t bw = a.mv_bw, bh = a.mv_bh;
  const uint32_t pper = (nf + te - 1) / te;  // block-uniform, <= P
  const uint32_t p0 = tid * pper;
  uint32_t dm[P];
#pragma unroll
  for (uint32_t t = 0; t < P; ++t) {
    v[t] = (t < pper && p0 + t < nf) ? pk[p0 + t] : 0u;
    dm[t] = 0xFFFFFFFFu;
  }
  if (tid == 0) set_centre(L, 0, pk[(uint32_t)(seg_hash(aseed) % nf)], bw, bh);
  __syncthreads();
  // k-means++: the next centre is drawn with probability ~ (distance to the nearest chosen centre)^2;
  // each point's running minimum stays in its register, so a step costs ONE new distance per point.
  // The draw is "the first point whose inclusive prefix exceeds r", found by one scan over the lanes'
  // sums and a walk through the one lane whose range holds it.
  for (uint32_t j = 1; j < k; ++j) {
    const int c[3] = {L.cint[j - 1][0], L.cint[j - 1][1], L.cint[j - 1][2]};
    uint64_t lsum = 0;
#pragma unroll
    for (uint32_t t = 0; t < P; ++t)
      if (t < pper) {
        // many points per lane: keep them packed across the rounds (unpacked once outside the loop they would be 3 P
        // registers and spill)
        if constexpr (P > 8) asm volatile("" : "+v"(v[t]));
        const uint32_t m = p0 + t < nf ? min(dm[t], dist2_u32(unpack_pt(v[t], bw, bh), c)) : 0u;
        dm[t] = m;
        lsum += m;
      }
    // wave totals by DPP (two 32-bit halves: a lane's sum is below 2^33), the draw by wave 0, the
    // scan over the lanes only inside the wave that owns the draw
    const uint64_t wtot = (uint64_t)(uint32_t)wave_sum_i32((int)(lsum & 0xFFFFFFu)) +
                          ((uint64_t)(uint32_t)wave_sum_i32((int)(lsum >> 24)) << 24);
    if (lane == 0) L.scan[wave] = wtot;
    __syncthreads();
    if (wave == 0) {
      uint64_t total = 0;
      for (uint32_t wv = 0; wv < nw; ++wv) total += L.scan[wv];
      if (total == 0) {
        if (tid == 0) {
          set_centre(L, j, pk[j < nf ? j : 0], bw, bh);
          L.pick = 0xFFFFFFFFu;
        }
      } else {
        uint64_t r = seg_hash(aseed ^ j) % total;
        uint32_t owner = 0;
        for (; owner + 1 < nw && r >= L.scan[owner]; ++owner) r -= L.scan[owner];
        if (tid == 0) { L.pick = owner; L.draw = r; }  // r is now relative to the owner's first point
      }
    }
    __syncthreads();
    if (L.pick == wave) {
      const uint64_t r = L.draw;
      const uint64_t excl = wave_incl_scan_u64(lsum, lane) - lsum;
      if (r >= excl && r - excl < lsum) {  // exactly one lane owns the crossing
        uint64_t acc = excl;
        bool found = false;
#pragma unroll
        for (uint32_t t = 0; t < P; ++t)
          if (t < pper) {
            acc += dm[t];
            if (!found && acc > r) { found = true; set_centre(L, j, v[t], bw, bh); }
          }
      }
    }
    __syncthreads();
  }
}

template <uint32_t P>
__device__ __forceinline__ uint64_t kmeans_regs(const uint32_t* pk, uint8_t* lab, KmLds& L, const SegArgs& a,
                                                uint32_t nf, uint32_t k, uint64_t aseed, uint32_t tid, uint32_t te) {
  const uint32_t lane = tid & 63u;
  const uint32_t bw = a.mv_bw, bh = a.mv_bh;
  const uint32_t pper = (nf + te - 1) / te;  // block-uniform, <= P
  const uint32_t p0 = tid * pper;
  uint32_t v[P];
  seed_regs<P>(pk, v, L, a, nf, k, aseed, tid, te);
  lloyd_begin(L, k, tid);

  uint32_t oldpack[P / 4];  // the points' current labels, a byte each
#pragma unroll
  for (uint32_t g = 0; g < P / 4; ++g) oldpack[g] = 0xFFFFFFFFu;
  uint64_t compact = 0;
  for (uint32_t it = 0;; ++it) {
    uint64_t lc = 0;
#pragma unroll
    for (uint32_t g = 0; g < P / 4; ++g)
      if (4 * g < pper) {
        uint32_t vq[4], oldj[4], bj[4];
        bool act[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          const uint32_t t = 4 * g + u;
          vq[u] = v[t];
          act[u] = t < pper && p0 + t < nf;
          oldj[u] = (oldpack[g] >> (8 * u)) & 0xFFu;
        }
        lloyd_quad(vq, act, oldj, bj, lc, L, k, bw, bh, lane);
        uint32_t np = 0;
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) np |= (act[u] ? bj[u] : 0xFFu) << (8 * u);
        oldpack[g] = np;
      }
    const bool done = lloyd_end_iter(lc, compact, it, L, a, k, tid, lane);
    if (done) break;
  }
#pragma unroll
  for (uint32_t t = 0; t < P; ++t)
    if (t < pper && p0 + t < nf) lab[p0 + t] = (uint8_t)((oldpack[t / 4] >> (8 * (t & 3))) & 0xFFu);
  return compact;
}

// The same attempt for more than kRegPts points per lane (4K / 8K scene cuts): points and running
// minima are read from `pk` / `dmin` (LDS when they fit, else the workspace) each time.  A wave owns a
// contiguous chunk of the list and its lanes interleave inside it (conflict-free LDS, coalesced
// global); the draw is located by wave totals first, then by a scan inside the one wave that holds it.
__device__ __forceinline__ uint64_t kmeans_packed(const uint32_t* pk, uint32_t* dmin, uint8_t* lab, KmLds& L,
                                                  const SegArgs& a, uint32_t nf, uint32_t k, uint64_t aseed,
                                                  uint32_t tid, uint32_t te) {
  const uint32_t nw = te >> 6, wave = tid >> 6, lane = tid & 63u;
  const uint32_t bw = a.mv_bw, bh = a.mv_bh;
  if (tid == 0) set_centre(L, 0, pk[(uint32_t)(seg_hash(aseed) % nf)], bw, bh);
  const uint32_t chunk = (((nf + nw - 1) / nw) + 63u) & ~63u;
  const uint32_t w0 = min(nf, wave * chunk), w1 = min(nf, w0 + chunk);
  for (uint32_t i = w0 + lane; i < w1; i += 64) dmin[i] = 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t j = 1; j < k; ++j) {
    const int c[3] = {L.cint[j - 1][0], L.cint[j - 1][1], L.cint[j - 1][2]};
    uint64_t lsum = 0;
    for (uint32_t i = w0 + lane; i < w1; i += 64) {  // a lane only ever touches its own entries of dmin
      const uint32_t m = min(dmin[i], dist2_u32(unpack_pt(pk[i], bw, bh), c));
      dmin[i] = m;
      lsum += m;
    }
    const uint64_t wtot = wave_sum_u64(lsum);
    if (lane == 0) L.scan[wave] = wtot;
    __syncthreads();
    uint64_t woff = 0, total = 0;
    for (uint32_t wv = 0; wv < nw; ++wv) {
      const uint64_t t = L.scan[wv];
      woff += wv < wave ? t : 0;
      total += t;
    }
    if (total == 0) {
      if (tid == 0) set_centre(L, j, pk[j < nf ? j : 0], bw, bh);
    } else {
      const uint64_t r = seg_hash(aseed ^ j) % total;
      if (r >= woff && r - woff < wtot) {  // wave-uniform: exactly one wave owns the crossing
        uint64_t acc = woff;
        for (uint32_t base = w0; base < w1; base += 64) {
          const uint32_t i = base + lane;
          const uint64_t incl = wave_incl_scan_u64(i < w1 ? dmin[i] : 0u, lane);
          const unsigned long long bal = __ballot(acc + incl > r);
          if (bal) {
            if (lane == (uint32_t)__builtin_ctzll(bal)) set_centre(L, j, pk[i], bw, bh);
            break;
          }
          acc += readlane_u64(incl, 63);
        }
      }
    }
    __syncthreads();
  }
  lloyd_begin(L, k, tid);

  uint64_t compact = 0;
  for (uint32_t it = 0;; ++it) {
    uint64_t lc = 0;
    for (uint32_t i0 = 0; i0 < nf; i0 += 4 * te) {  // wave-uniform trip count
      uint32_t vq[4], oldj[4], bj[4];
      bool act[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t i = i0 + u * te + tid;
        act[u] = i < nf;
        vq[u] = act[u] ? pk[i] : 0u;
        oldj[u] = (it && act[u]) ? lab[i] : 0xFFu;
      }
      lloyd_quad(vq, act, oldj, bj, lc, L, k, bw, bh, lane);
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        if (act[u] && bj[u] != oldj[u]) lab[i0 + u * te + tid] = (uint8_t)bj[u];
    }
    if (lloyd_end_iter(lc, compact, it, L, a, k, tid, lane)) break;
  }
  return compact;
}

// The same attempt over unpacked 12-byte points in the workspace: fields larger than 512 x 512
// blocks or |mv.x| >= 8192, which block matching never produces.  Kept simple: 64-bit distances,
// every step recomputes the minimum over the chosen centres.
__device__ __forceinline__ uint64_t kmeans_generic(const Pt* pts, uint8_t* lab, KmLds& L, const SegArgs& a,
                                                   uint32_t nf, uint32_t k, uint64_t aseed, uint32_t tid,
                                                   uint32_t te) {
  const uint32_t nw = te >> 6, lane = tid & 63u;
  const uint32_t pper = (nf + te - 1) / te;
  const uint32_t p0 = min(nf, tid * pper), p1 = min(nf, p0 + pper);
  if (tid == 0) {
    const Pt p = pts[(uint32_t)(seg_hash(aseed) % nf)];
    L.cint[0][0] = p.f[0]; L.cint[0][1] = p.f[1]; L.cint[0][2] = p.f[2];
  }
  __syncthreads();
  for (uint32_t j = 1; j < k; ++j) {
    uint64_t lsum = 0;
    for (uint32_t i = p0; i < p1; ++i) {
      const Pt p = pts[i];
      uint64_t m = ~0ull;
      for (uint32_t q = 0; q < j; ++q) m = min(m, dist2_int(p, L.cint[q]));
      lsum += m;
    }
    uint64_t total;
    const uint64_t excl = block_excl_scan(lsum, L.scan, tid, nw, &total);
    if (total == 0) {
      if (tid == 0) L.pick = j < nf ? j : 0;
    } else {
      const uint64_t r = seg_hash(aseed ^ j) % total;
      if (r >= excl && r < excl + lsum) {  // exactly one lane owns the crossing
        uint64_t acc = excl;
        for (uint32_t i = p0; i < p1; ++i) {
          const Pt p = pts[i];
          uint64_t m = ~0ull;
          for (uint32_t q = 0; q < j; ++q) m = min(m, dist2_int(p, L.cint[q]));
          acc += m;
          if (acc > r) { L.pick = i; break; }
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      const Pt p = pts[L.pick];
      L.cint[j][0] = p.f[0]; L.cint[j][1] = p.f[1]; L.cint[j][2] = p.f[2];
    }
    __syncthreads();
  }
  if (tid < k) {
    L.c[tid][0] = (double)L.cint[tid][0];
    L.c[tid][1] = (double)L.cint[tid][1];
    L.c[tid][2] = (double)L.cint[tid][2];
  }
  __syncthreads();

  uint64_t compact = 0;
  for (uint32_t it = 0;; ++it) {  // Lloyd
    if (tid < k) { L.sum[tid][0] = 0; L.sum[tid][1] = 0; L.sum[tid][2] = 0; L.cnt[tid] = 0; }
    if (tid == 0) L.compact = 0;
    __syncthreads();
    unsigned long long lc = 0;
    for (uint32_t i0 = 0; i0 < nf; i0 += te) {  // wave-uniform trip count
      const uint32_t i = i0 + tid;
      const bool active = i < nf;
      Pt p = {{0, 0, 0}};
      uint32_t bj = 0xFFFFFFFFu;
      if (active) {
        p = pts[i];
        double best = dist2_dbl(p, L.c[0]);
        bj = 0;
        for (uint32_t j = 1; j < k; ++j) {
          const double d = dist2_dbl(p, L.c[j]);
          if (d < best) { best = d; bj = j; }
        }
        lab[i] = (uint8_t)bj;
        lc += (unsigned long long)(best * 256.0);
      }
      for (uint32_t j = 0; j < k; ++j) {  // coordinates can be anything here: 64-bit sums per lane
        const bool mine = bj == j;
        const unsigned long long bal = __ballot(mine);
        if (bal == 0) continue;
        const uint64_t sx = wave_sum_u64(mine ? (uint64_t)(int64_t)p.f[0] : 0),
                       sy = wave_sum_u64(mine ? (uint64_t)(int64_t)p.f[1] : 0),
                       sz = wave_sum_u64(mine ? (uint64_t)(int64_t)p.f[2] : 0);
        if (lane == 0) {
          atomicAdd(&L.cnt[j], (uint32_t)__popcll(bal));
          atomicAdd(&L.sum[j][0], (unsigned long long)sx);
          atomicAdd(&L.sum[j][1], (unsigned long long)sy);
          atomicAdd(&L.sum[j][2], (unsigned long long)sz);
        }
      }
    }
    atomicAdd(&L.compact, lc);
    __syncthreads();
    compact = L.compact;
    if (it + 1 >= a.max_iter) break;
    if (tid < k) {
      double s = 0.0;
      if (L.cnt[tid]) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const double nc = (double)(long long)L.sum[tid][d] / (double)L.cnt[tid];
          const double t = nc - L.c[tid][d];
          s = s + t * t;
          L.c[tid][d] = nc;
        }
      }
      L.shift[tid] = s;
    }
    __syncthreads();
    double shift = 0.0;
    for (uint32_t j = 0; j < k; ++j) shift = L.shift[j] > shift ? L.shift[j] : shift;
    if (shift <= a.eps2) break;
  }
  return compact;
}

// Kernel P: one workgroup per frame.  Foreground mask, close, open, the foreground list in raster
// order and the feature points (encoder.cpp:507-546, :300-321), shared by the frame's k-means attempts.
// Dynamic LDS: two bit fields + the flat bitmap they are built from (a.bits_bytes), then the list.
__global__ __launch_bounds__(kTA) void segment_prepare_kernel(SegArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
  __shared__ uint64_t s_scan[kTA / 64];
  __shared__ uint32_t s_bad;

  const uint32_t tid = threadIdx.x, frame = blockIdx.x, n = a.n;
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  const uint8_t* mask = a.mask + (size_t)frame * n;
  const float2* mv = reinterpret_cast<const float2*>(a.mv) + (size_t)frame * n;
  const Workspace ws{a.ws + (size_t)frame * a.ws_stride, n, a.attempts};
  uint32_t* idx = ws.idx();

  BitField bf;
  bf.W32 = (a.mfw + 31) / 32; bf.H = a.mfh; bf.NW = bf.W32 * bf.H;
  bf.last_valid = (a.mfw & 31u) ? ((1u << (a.mfw & 31u)) - 1u) : ~0u;
  uint32_t* bitA = reinterpret_cast<uint32_t*>(dyn_lds);
  uint32_t* bitB = bitA + bf.NW;
  uint32_t* flat = bitB + bf.NW;
  uint32_t* lst_lds = reinterpret_cast<uint32_t*>(dyn_lds + a.bits_bytes);
  const size_t lds_cap = a.lds_bytes - a.bits_bytes;

  // ---- foreground = complement of the inliers (:507-513), one bit per block: 64 blocks per ballot,
  // eight loads in flight per lane
  const uint32_t trips = (n + kTA - 1) / kTA;
  for (uint32_t t0 = 0; t0 < trips; t0 += 8) {
    uint8_t m[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint32_t i = (t0 + u) * kTA + tid;
      m[u] = i < n ? mask[i] : 1;
    }
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const unsigned long long bal = __ballot(m[u] == 0);
      if (lane == 0 && t0 + u < trips) {
        flat[2 * ((t0 + u) * (kTA / 64) + wave)] = (uint32_t)bal;
        flat[2 * ((t0 + u) * (kTA / 64) + wave) + 1] = (uint32_t)(bal >> 32);
      }
    }
  }
  if (tid < 2) flat[trips * (kTA / 32) + tid] = 0;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  for (uint32_t q = tid; q < bf.NW; q += kTA) {
    const uint32_t y = q / bf.W32, w = q - y * bf.W32;
    const uint32_t o = y * a.mfw + 32 * w, fw = o >> 5, fb = o & 31u;
    const uint32_t v = fb ? (flat[fw] >> fb) | (flat[fw + 1] << (32 - fb)) : flat[fw];
    bitA[q] = w == bf.W32 - 1 ? v & bf.last_valid : v;
  }
  __syncthreads();
  // ---- close, open (:515-527) --------------------------------------------------------------------
  bit_morph(bitA, bitB, bitA, bf, a.morph_w, a.morph_h, false, tid);
  bit_morph(bitA, bitB, bitA, bf, a.morph_w, a.morph_h, true, tid);
  bit_morph(bitA, bitB, bitA, bf, a.morph_w, a.morph_h, true, tid);
  bit_morph(bitA, bitB, bitA, bf, a.morph_w, a.morph_h, false, tid);

  // ---- foreground list in raster order (:538-546) -> feature points (:300-321) ------------
  const uint32_t wper = (bf.NW + kTA - 1) / kTA;
  const uint32_t q0 = min(bf.NW, tid * wper), q1 = min(bf.NW, q0 + wper);
  uint32_t local = 0;
  for (uint32_t q = q0; q < q1; ++q) local += __popc(bitA[q]);
  uint64_t tot64;
  const uint32_t pos0 = (uint32_t)block_excl_scan(local, s_scan, tid, kTA / 64, &tot64);
  const uint32_t nf = (uint32_t)tot64;
  if (tid == 0) *ws.nf() = nf;
  if (nf == 0) return;
  uint32_t* pk = ws.pk();
  if (a.packable) {
    uint32_t* lst = 4 * (size_t)nf <= lds_cap ? lst_lds : idx;  // else expand straight into the workspace
    uint32_t pos = pos0;
    for (uint32_t q = q0; q < q1; ++q) {
      const uint32_t y = q / bf.W32, base = y * a.mfw + 32 * (q - y * bf.W32);
      for (uint32_t bits = bitA[q]; bits; bits &= bits - 1) lst[pos++] = base + (uint32_t)__builtin_ctz(bits);
    }
    __syncthreads();
    bool bad = false;
    for (uint32_t q0b = 0; q0b < nf; q0b += 4 * kTA) {  // four independent index -> mv chains per lane
      uint32_t bi[4];
      float mx[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t q = q0b + u * kTA + tid;
        bi[u] = q < nf ? lst[q] : 0u;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) mx[u] = mv[bi[u]].x;
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t q = q0b + u * kTA + tid;
        if (q < nf) {
          const uint32_t y = bi[u] / a.mfw, x = bi[u] - y * a.mfw;
          bad = bad || !(mx[u] > -8191.0f && mx[u] < 8191.0f);
          pk[q] = pack_pt((int)(mx[u] < 0 ? mx[u] - 0.5f : mx[u] + 0.5f), x, y);
          if (lst != idx) idx[q] = bi[u];
        }
      }
    }
    if (bad) s_bad = 1;
    __syncthreads();
  }
  const bool packed = a.packable != 0 && s_bad == 0;
  if (tid == 0) *ws.packed() = packed ? 1u : 0u;
  if (!packed) {  // never for block-matching output: serial per word, unpacked points
    Pt* pts = ws.pts();
    uint32_t pos = pos0;
    for (uint32_t q = q0; q < q1; ++q) {
      const uint32_t y = q / bf.W32, x0 = 32 * (q - y * bf.W32);
      for (uint32_t bits = bitA[q]; bits; bits &= bits - 1) {
        const uint32_t x = x0 + (uint32_t)__builtin_ctz(bits), i = y * a.mfw + x;
        const float mx = mv[i].x;
        Pt p;
        p.f[0] = (int)(mx < 0 ? mx - 0.5f : mx + 0.5f); p.f[1] = (int)(x * a.mv_bw); p.f[2] = (int)(y * a.mv_bh);
        pts[pos] = p;
        idx[pos] = i;
        ++pos;
      }
    }
  }
}

// Kernel A: one workgroup per (frame, k-means attempt).  Attempts are independent restarts
// (cv::kmeans' `attempts`), so they run side by side instead of one after the other.  Launched twice:
// T = 256 lanes takes the frames of at most kLightMax foreground blocks (nearly all of them: four
// waves per workgroup, several workgroups per CU, and the hundreds of short barrier-separated phases
// of a light frame stay cheap; a single wave up to 256 blocks), T = 1024 the heavy ones (a scene cut: most of the field is
// foreground); a workgroup whose frame belongs to the other launch ends at once.
// Dynamic LDS (fields that can have more than kRegPts points per lane only): the packed points and,
// if they fit too, the running minima of the path that does not keep them in registers.
constexpr uint32_t kLightMax = 2048;
constexpr uint32_t kRegPts = 8;  // points a lane keeps in registers through an attempt; frames with more per lane
                                 // (> 8 192 blocks at 1 024 lanes) go through LDS / the workspace

template <uint32_t T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) void segment_attempt_kernel(SegArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
  __shared__ KmLds L;

  const uint32_t tid = threadIdx.x, frame = blockIdx.x, att = blockIdx.y, n = a.n;
  const Workspace ws{a.ws + (size_t)frame * a.ws_stride, n, a.attempts};
  const uint32_t nf = *ws.nf();
  if (nf == 0) return;
  if (a.wide_g) {  // the launch sequence of segment_wide_*_kernel has every frame of packed points, light ones included;
    if (*ws.packed() != 0) return;  // this (1024-lane, the only attempt launch then) takes the frames of unpacked points
  } else if (!a.take_all && (nf > kLightMax) != (T == kTA)) {
    return;
  }
  const uint32_t te = T == kTA ? kTA : nf <= 256 ? 64u : 256u;  // one wave: no barrier ever waits
  if (tid >= te) return;  // whole waves; the barriers below count the surviving ones only
  uint8_t* lab = ws.lab(att);
  const uint32_t k = min(a.k, nf);  // :555
  const bool packed = *ws.packed() != 0;

  // ---- one k-means attempt on (mv.x, x_px, y_px) (:557-578) --------------------------------
  const uint64_t aseed = (a.seed + frame) ^ ((uint64_t)att << 32);
  uint64_t compact;
  if (packed && nf <= kRegPts * te) {
    compact = kmeans_regs<kRegPts>(ws.pk(), lab, L, a, nf, k, aseed, tid, te);
  } else if (packed) {
    const size_t lds_cap = T == kTA ? a.lds_bytes : 0;  // the 256-lane launch has no dynamic LDS: workspace
    uint32_t* pk = ws.pk();
    uint32_t* dmin = ws.dmin(att);
    if (4 * (size_t)nf <= lds_cap) {
      uint32_t* pk_lds = reinterpret_cast<uint32_t*>(dyn_lds);
      for (uint32_t i = tid; i < nf; i += te) pk_lds[i] = pk[i];
      pk = pk_lds;
      if (8 * (size_t)nf <= lds_cap) dmin = pk_lds + nf;
      __syncthreads();
    }
    compact = kmeans_packed(pk, dmin, lab, L, a, nf, k, aseed, tid, te);
  } else {
    compact = kmeans_generic(ws.pts(), lab, L, a, nf, k, aseed, tid, te);
  }
  if (tid == 0) ws.compact()[att] = compact;
}

// ---- an attempt as a SEQUENCE OF LAUNCHES over G workgroups (few frames, large fields) ---------------------------
// One (frame, attempt) of the kernel above is bound to one CU by its ~40 workgroup barriers: 0.57 ms for a 4K scene cut
// (29 600 foreground blocks), which is what an 8-frame shard of a 4K clip then waits for while 230 CUs idle.  When
// frames x attempts is small the host instead runs the attempt as a sequence of launches of G workgroups per (frame,
// attempt): every grid-wide reduction of the algorithm (the k-means++ draw, the centre update) sits on a kernel boundary,
// so there is no spin-wait and no residency assumption.  A launch reads only what EARLIER launches wrote (WideState);
// inside a launch every workgroup recomputes the shared quantities (draw, centres, convergence) redundantly from the
// per-workgroup partial sums, which are exact integers -- so the result is the single-kernel path's, and
// oracle/svc_segment.c's, bit for bit.  Workgroup g owns the g-th contiguous chunk of the foreground list.  The light
// frames of the batch ride along (most of their workgroups own nothing): the sequence is as long with them as without.
constexpr uint32_t kTW = 256;  // lanes of a wide workgroup

__device__ __forceinline__ void wide_chunk(uint32_t nf, uint32_t G, uint32_t g, uint32_t& w0, uint32_t& w1) {
  const uint32_t chunk = (((nf + G - 1) / G) + 63u) & ~63u;
  w0 = min(nf, g * chunk);
  w1 = min(nf, w0 + chunk);
}

// The head of the launch sequence, one workgroup per (frame, attempt): the whole k-means++ seeding with the points in
// registers (k - 1 rounds of one distance per point and a block-wide draw: 0.02 ms for a 4K scene cut, against k launches
// of segment_wide_seed_kernel at 8 us each), for fields of at most kWideRegPts x 1024 blocks; frames of unpacked points
// (none after block matching) get their whole attempt here, on the generic path.
constexpr uint32_t kWideRegPts = 32;
__global__ __launch_bounds__(kTA) void segment_wide_head_kernel(SegArgs a) {
  __shared__ KmLds L;
  const uint32_t tid = threadIdx.x, frame = blockIdx.x, att = blockIdx.y, n = a.n;
  const Workspace ws{a.ws + (size_t)frame * a.ws_stride, n, a.attempts};
  const uint32_t nf = *ws.nf();
  if (nf == 0) return;
  const uint32_t k = min(a.k, nf);
  const uint64_t aseed = (a.seed + frame) ^ ((uint64_t)att << 32);
  if (*ws.packed() == 0) {
    const uint64_t compact = kmeans_generic(ws.pts(), ws.lab(att), L, a, nf, k, aseed, tid, kTA);
    if (tid == 0) ws.compact()[att] = compact;
    return;
  }
  const uint32_t te = nf <= 256 ? 64u : nf <= 256 * kRegPts ? 256u : kTA;  // as the single-kernel attempts
  if (tid >= te) return;  // whole waves; the barriers below count the surviving ones only
  uint32_t v[kWideRegPts];
  if (te == kTA) {
    seed_regs<kWideRegPts>(ws.pk(), v, L, a, nf, k, aseed, tid, te);
  } else {
    uint32_t v8[kRegPts];
    seed_regs<kRegPts>(ws.pk(), v8, L, a, nf, k, aseed, tid, te);
  }
  WideState& W = *ws.wide(att);
  if (tid < k) { W.cint[tid][0] = L.cint[tid][0]; W.cint[tid][1] = L.cint[tid][1]; W.cint[tid][2] = L.cint[tid][2]; }
  if (tid == 0) W.done = 0;
}

// Launch j = a.wide_step of the seeding (j = 0 .. k - 1): fixes centre j, then folds it into the running minima of this
// workgroup's chunk and publishes their sum for launch j + 1's draw (libs/encoder.cpp:557-578 via cv::kmeans' k-means++,
// as oracle/svc_segment.c states it).
__global__ __launch_bounds__(kTW) void segment_wide_seed_kernel(SegArgs a) {
  __shared__ uint64_t s_red[kTW / 64];
  __shared__ uint64_t s_sums[kWideMaxG];
  __shared__ uint32_t s_pick;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t g = blockIdx.x, G = a.wide_g, frame = blockIdx.y, att = blockIdx.z, n = a.n, j = a.wide_step;
  const Workspace ws{a.ws + (size_t)frame * a.ws_stride, n, a.attempts};
  const uint32_t nf = *ws.nf();
  if (nf == 0 || *ws.packed() == 0) return;  // frames of unpacked points stay on the single-kernel path
  const uint32_t k = min(a.k, nf);
  if (j >= k) return;
  WideState& W = *ws.wide(att);
  const uint32_t* pk = ws.pk();
  // step j reads the minima of step j - 1 (another workgroup may still be scanning this workgroup's chunk for the draw)
  // and writes its own into the other set
  const uint32_t* dprev = (j & 1u) ? ws.dmin(att) : ws.dmin2(att);
  uint32_t* dmin = (j & 1u) ? ws.dmin2(att) : ws.dmin(att);
  const uint64_t aseed = (a.seed + frame) ^ ((uint64_t)att << 32);
  const uint32_t bw = a.mv_bw, bh = a.mv_bh;
  uint32_t w0, w1;
  wide_chunk(nf, G, g, w0, w1);

  uint32_t cv;  // centre j as a packed point
  if (j == 0) {
    cv = pk[(uint32_t)(seg_hash(aseed) % nf)];
    if (g == 0 && tid == 0) W.done = 0;
  } else {
    // the draw of step j over the minima that launch j - 1 left: total, owner workgroup, offset inside its chunk
    if (tid < G) s_sums[tid] = W.seed_sum[j - 1][tid];  // G loads in flight together, then LDS
    __syncthreads();
    uint64_t total = 0;
    for (uint32_t q = 0; q < G; ++q) total += s_sums[q];
    if (total == 0) {
      cv = pk[j < nf ? j : 0];
    } else {
      uint64_t r = seg_hash(aseed ^ j) % total;
      uint32_t owner = 0;
      for (; owner + 1 < G && r >= s_sums[owner]; ++owner) r -= s_sums[owner];
      uint32_t o0, o1;
      wide_chunk(nf, G, owner, o0, o1);
      // "the first point whose inclusive prefix exceeds r", in list order: every lane sums a contiguous run of the
      // owner's chunk, one block scan over the 256 run sums finds the run that holds the crossing, its lane walks it
      const uint32_t run = (o1 - o0 + kTW - 1) / kTW;
      const uint32_t b = min(o1, o0 + tid * run), e = min(o1, b + run);
      uint64_t seg = 0;
      for (uint32_t i = b; i < e; ++i) seg += dprev[i];
      uint64_t blk_total;
      const uint64_t excl = block_excl_scan(seg, s_red, tid, kTW / 64, &blk_total);
      if (r >= excl && r - excl < seg) {  // exactly one lane (blk_total = the owner's published sum > r)
        uint64_t acc = excl;
        for (uint32_t i = b; i < e; ++i) {
          acc += dprev[i];
          if (acc > r) { s_pick = i; break; }
        }
      }
      __syncthreads();
      cv = pk[s_pick];
      __syncthreads();  // s_pick is read before anyone could get to overwrite it (single use: kept for symmetry)
    }
  }
  if (g == 0 && tid == 0) {
    const Pt p = unpack_pt(cv, bw, bh);
    W.cint[j][0] = p.f[0]; W.cint[j][1] = p.f[1]; W.cint[j][2] = p.f[2];
  }
  if (j + 1 >= k) return;  // the last centre: nothing draws after it
  const Pt cp = unpack_pt(cv, bw, bh);
  const int c[3] = {cp.f[0], cp.f[1], cp.f[2]};
  uint64_t lsum = 0;
  for (uint32_t i = w0 + tid; i < w1; i += kTW) {
    const uint32_t d = dist2_u32(unpack_pt(pk[i], bw, bh), c);
    const uint32_t m = j == 0 ? d : min(dprev[i], d);
    dmin[i] = m;
    lsum += m;
  }
  lsum = wave_sum_u64(lsum);
  if (lane == 0) s_red[wave] = lsum;
  __syncthreads();
  if (tid == 0) {
    uint64_t t = 0;
    for (uint32_t q = 0; q < kTW / 64; ++q) t += s_red[q];
    W.seed_sum[j][g] = t;
  }
}

// Launch it = a.wide_step of the Lloyd iterations (it = 0 .. max_iter): closes iteration it - 1 (compactness, centre
// update, convergence: lloyd_end_iter's arithmetic in lloyd_end_iter's order), then assigns this workgroup's chunk to
// the new centres and publishes its per-cluster sums.  The close of the last iteration (it = max_iter: nothing but the sum of
// the workgroups' compactness) is done by the labelling kernel, which saves a launch in a latency-bound chain.
__global__ __launch_bounds__(kTW) void segment_wide_lloyd_kernel(SegArgs a) {
  __shared__ double s_c[kMaxK][3];
  __shared__ double s_shift[kMaxK];
  __shared__ unsigned long long s_acc[kMaxK][2];
  __shared__ unsigned long long s_sum[kMaxK][4];
  __shared__ unsigned long long s_lc;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t g = blockIdx.x, G = a.wide_g, frame = blockIdx.y, att = blockIdx.z, n = a.n, it = a.wide_step;
  const Workspace ws{a.ws + (size_t)frame * a.ws_stride, n, a.attempts};
  const uint32_t nf = *ws.nf();
  if (nf == 0 || *ws.packed() == 0) return;
  const uint32_t k = min(a.k, nf);
  WideState& W = *ws.wide(att);
  // The verdict of an EARLIER launch only: the launch that finds the attempt over writes its own step number (below), which
  // no workgroup of that same launch honours whenever it is dispatched -- a late workgroup whose waves loaded the flag on
  // either side of the store would otherwise split (some waves leave, the barriers count the rest, the partial sums lose
  // lanes).  Every workgroup of the writing launch derives the same `over` from the partial sums and leaves by itself.
  {
    const uint32_t d = W.done;
    if (d != 0 && d < it) return;
  }
  const uint32_t bw = a.mv_bw, bh = a.mv_bh;
  const uint32_t par = it & 1u, prev = par ^ 1u;

  if (it == 0) {
    if (tid < k) { s_c[tid][0] = (double)W.cint[tid][0]; s_c[tid][1] = (double)W.cint[tid][1]; s_c[tid][2] = (double)W.cint[tid][2]; }
    __syncthreads();
  } else {
    // iteration it - 1 is complete in W.part[prev] / W.compact[prev]: summed over the G workgroups with every load in flight
    // at once (one lane per (workgroup, cluster) record), not G round trips in a row
    if (tid < k) { s_sum[tid][0] = 0; s_sum[tid][1] = 0; s_sum[tid][2] = 0; s_sum[tid][3] = 0; }
    if (tid == 0) s_lc = 0;
    __syncthreads();
    for (uint32_t t = tid; t < G * k; t += kTW) {
      const uint32_t q = t / k, j = t - q * k;
      const long long* r = W.part[prev][q][j];
      const long long r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
      atomicAdd(&s_sum[j][0], (unsigned long long)r0); atomicAdd(&s_sum[j][1], (unsigned long long)r1);
      atomicAdd(&s_sum[j][2], (unsigned long long)r2); atomicAdd(&s_sum[j][3], (unsigned long long)r3);
    }
    if (tid < G) atomicAdd(&s_lc, W.compact[prev][tid]);
    __syncthreads();
    const unsigned long long compact = s_lc;
    const bool last = it >= a.max_iter;  // lloyd_end_iter: `if (it + 1 >= max_iter) return true` before any update
    if (!last && tid < k) {
      const long long cnt = (long long)s_sum[tid][0], smv = (long long)s_sum[tid][1], sbx = (long long)s_sum[tid][2],
                      sby = (long long)s_sum[tid][3];
      double s = 0.0;
      double c0 = W.c[prev][tid][0], c1 = W.c[prev][tid][1], c2 = W.c[prev][tid][2];
      if (cnt) {
        const long long sums[3] = {smv, sbx * (long long)bw, sby * (long long)bh};
        double cc[3] = {c0, c1, c2};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const double nc = (double)sums[d] / (double)cnt;
          const double t = nc - cc[d];
          s = s + t * t;
          cc[d] = nc;
        }
        c0 = cc[0]; c1 = cc[1]; c2 = cc[2];
      }
      s_c[tid][0] = c0; s_c[tid][1] = c1; s_c[tid][2] = c2;
      s_shift[tid] = s;
    }
    __syncthreads();
    bool over = last;
    if (!last) {
      double shift = 0.0;
      for (uint32_t q = 0; q < k; ++q) shift = s_shift[q] > shift ? s_shift[q] : shift;
      over = shift <= a.eps2;
    }
    if (over) {  // the labels of iteration it - 1 stand; every workgroup sees the same verdict
      if (g == 0 && tid == 0) { ws.compact()[att] = compact; W.done = it; }  // it >= 1 here
      return;
    }
  }
  if (g == 0 && tid < k) { W.c[par][tid][0] = s_c[tid][0]; W.c[par][tid][1] = s_c[tid][1]; W.c[par][tid][2] = s_c[tid][2]; }
  if (tid < k) { s_acc[tid][0] = 0; s_acc[tid][1] = 0; }
  if (tid == 0) s_lc = 0;
  __syncthreads();

  // assign: nearest centre (strict <: the lowest index keeps a tie), f64 in lloyd_quad's operation order
  const uint32_t* pk = ws.pk();
  uint8_t* lab = ws.lab(att);
  uint32_t w0, w1;
  wide_chunk(nf, G, g, w0, w1);
  unsigned long long lc = 0;
  for (uint32_t i0 = w0; i0 < w1; i0 += 4 * kTW) {  // four points per lane at a time: four independent f64 chains
    uint32_t v[4], bj[4];
    bool act[4];
    double px[4], py[4], pz[4], best[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t i = i0 + (uint32_t)u * kTW + tid;
      act[u] = i < w1;
      v[u] = act[u] ? pk[i] : 0u;
      const Pt p = unpack_pt(v[u], bw, bh);
      px[u] = (double)p.f[0]; py[u] = (double)p.f[1]; pz[u] = (double)p.f[2];
      const double dx = px[u] - s_c[0][0], dy = py[u] - s_c[0][1], dz = pz[u] - s_c[0][2];
      double d = dx * dx;
      d = d + dy * dy;
      d = d + dz * dz;
      best[u] = d;
      bj[u] = 0;
    }
    for (uint32_t q = 1; q < k; ++q) {
      const double c0 = s_c[q][0], c1 = s_c[q][1], c2 = s_c[q][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double ax = px[u] - c0, ay = py[u] - c1, az = pz[u] - c2;
        double da = ax * ax;
        da = da + ay * ay;
        da = da + az * az;
        bj[u] = da < best[u] ? q : bj[u];          // strict <: the lowest index keeps a tie
        best[u] = __builtin_fmin(best[u], da);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!act[u]) continue;
      lab[i0 + (uint32_t)u * kTW + tid] = (uint8_t)bj[u];
      lc += (unsigned long long)(best[u] * 256.0);
      // two fields per 64-bit LDS atomic, as lloyd_quad packs them: count | (mv.x + 8192) << 32 and column | row << 32; a
      // chunk holds fewer than 2^17 points, so no field reaches 2^32
      atomicAdd(&s_acc[bj[u]][0], 1ull | ((unsigned long long)(uint32_t)(((int)v[u] >> 18) + 8192) << 32));
      atomicAdd(&s_acc[bj[u]][1], (unsigned long long)(v[u] & 511u) | ((unsigned long long)((v[u] >> 9) & 511u) << 32));
    }
  }
  lc = wave_sum_u64(lc);
  if (lane == 0) atomicAdd(&s_lc, lc);
  __syncthreads();
  if (tid < k) {
    const long long cnt = (long long)(uint32_t)s_acc[tid][0];
    W.part[par][g][tid][0] = cnt;
    W.part[par][g][tid][1] = (long long)(s_acc[tid][0] >> 32) - 8192ll * cnt;  // sum of mv.x
    W.part[par][g][tid][2] = (long long)(uint32_t)s_acc[tid][1];
    W.part[par][g][tid][3] = (long long)(s_acc[tid][1] >> 32);
  }
  if (tid == 0) W.compact[par][g] = s_lc;
}

// Kernel B: one workgroup per frame.  Takes the attempt with the smallest compactness (ties ->
// the earlier attempt), then connected components per cluster, numbered as the reference
// numbers them (:597-623): inside a cluster by the raster position of a component's first block,
// clusters stacked with `offset += count including label 0`.  Everything after the clears walks the
// foreground list, not the field.  LDS_PARENT / LDS_CL: union-find parents (4 B per block) and
// cluster ids (1 B per block) in LDS; at 4K only the parents fit (130 KB).
template <bool LDS_PARENT, bool LDS_CL>
__global__ __launch_bounds__(kTA) void segment_label_kernel(SegArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
  __shared__ uint64_t s_scan[kTA / 64];
  __shared__ uint32_t s_wcnt[kTA / 64][kMaxK];  // roots per (wave, cluster) in the current round
  __shared__ uint32_t s_base[kMaxK];            // roots per cluster in earlier rounds; then the cluster's offset
  const uint32_t tid = threadIdx.x, frame = blockIdx.x, n = a.n;
  uint32_t* types = a.types + (size_t)frame * n;
  const Workspace ws{a.ws + (size_t)frame * a.ws_stride, n, a.attempts};
  const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
  uint32_t* parent = LDS_PARENT ? reinterpret_cast<uint32_t*>(dyn_lds) : ws.parent();
  uint8_t* cl = LDS_CL ? dyn_lds + 4 * n4 : ws.cl();
  const uint32_t* idx = ws.idx();
  uint32_t* roots = ws.roots();
  const uint32_t nf = *ws.nf();

  for (uint32_t i = tid; i < n; i += kTA) { types[i] = 0; cl[i] = 255; }  // :549-551
  if (nf == 0) return;
  const uint32_t k = min(a.k, nf);
  __shared__ unsigned long long s_compact[kMaxAttempts];
  if (tid < a.attempts) {
    unsigned long long c = ws.compact()[tid];
    if (a.wide_g && *ws.packed() != 0) {
      // an attempt of the launch sequence that ran to the iteration cap: its last Lloyd launch left the per-workgroup
      // compactness of iteration max_iter - 1, and lloyd_end_iter's "it + 1 >= max_iter" close is this sum
      const WideState& W = *ws.wide(tid);
      if (!W.done) {
        c = 0;
        for (uint32_t g = 0; g < a.wide_g; ++g) c += W.compact[(a.max_iter - 1) & 1u][g];
      }
    }
    s_compact[tid] = c;
  }
  __syncthreads();
  uint32_t best = 0;
  unsigned long long best_c = s_compact[0];
  for (uint32_t t = 1; t < a.attempts; ++t) {
    const unsigned long long c = s_compact[t];
    if (c < best_c) { best_c = c; best = t; }
  }
  const uint8_t* best_lab = ws.lab(best);
  for (uint32_t j = tid; j < (kTA / 64) * kMaxK; j += kTA) (&s_wcnt[0][0])[j] = 0;
  if (tid < kMaxK) s_base[tid] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < nf; i += kTA) cl[idx[i]] = best_lab[i];
  __syncthreads();
  // Horizontal runs first, without a single union: a wave looks at 64 consecutive blocks, one ballot
  // says which of them continue the run of their left neighbour, and every block of a run is pointed
  // straight at the run's first block inside the window (the smallest index, as the forest wants).
  {
    const uint32_t step_x = kTA % a.mfw, step_y = kTA / a.mfw;
    uint32_t y = tid / a.mfw, x = tid - y * a.mfw;
    const uint32_t lane = tid & 63u;
    for (uint32_t i0 = 0; i0 < n; i0 += kTA) {
      const uint32_t i = i0 + tid;
      const uint32_t c = i < n ? cl[i] : 255u;
      const bool same_left = c != 255u && x > 0 && cl[i - 1] == c;
      const unsigned long long sbits = __ballot(same_left);
      if (c != 255u) {
        const unsigned long long z = ~sbits & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        parent[i] = i - lane + (z ? 63u - (uint32_t)__builtin_clzll(z) : 0u);
      }
      x += step_x; y += step_y;
      if (x >= a.mfw) { x -= a.mfw; ++y; }
    }
  }
  const uint32_t te = lanes_for(nf);
  __syncthreads();
  if (tid >= te) return;  // whole waves; the barriers below count the surviving ones only
  const uint32_t nw = te >> 6, wave = tid >> 6, lane = tid & 63u;
  // Then only the unions the runs do not already imply: a run that crosses a window boundary, and a
  // block with the row above -- skipped where the left neighbour provably made the same connection
  // (its own vertical link plus the run in the row above), which is everywhere inside a blob.
  for (uint32_t q = tid; q < nf; q += te) {
    const uint32_t i = idx[q];
    const uint32_t c = cl[i];
    const uint32_t y = i / a.mfw, x = i - y * a.mfw;
    const bool left = x > 0 && cl[i - 1] == c;
    if (left && (i & 63u) == 0) uf_unite(parent, i, i - 1);
    if (y > 0) {
      const bool up = cl[i - a.mfw] == c;
      const bool upleft = x > 0 && cl[i - a.mfw - 1] == c;
      if (a.conn == 4) {
        if (up && !(left && upleft)) uf_unite(parent, i, i - a.mfw);
      } else if (up) {  // up-left and up-right sit in up's run
        if (!(left && upleft)) uf_unite(parent, i, i - a.mfw);
      } else {
        if (upleft && !left) uf_unite(parent, i, i - a.mfw - 1);  // else the left neighbour is linked to it
        const bool upright = x + 1 < a.mfw && cl[i - a.mfw + 1] == c;
        const bool right = x + 1 < a.mfw && cl[i + 1] == c;
        if (upright && !right) uf_unite(parent, i, i - a.mfw + 1);  // else the right neighbour links to it
      }
    }
  }
  __syncthreads();
  for (uint32_t q = tid; q < nf; q += te) {
    const uint32_t i = idx[q];
    const uint32_t r = uf_find(parent, i);
    if (r != i) parent[i] = r;  // still an ancestor for any concurrent walker; roots are never rewritten
  }
  __syncthreads();
  // component roots, compacted in raster order (lane-contiguous ranges of the list + one block scan)
  const uint32_t pper = (nf + te - 1) / te;
  const uint32_t p0 = min(nf, tid * pper), p1 = min(nf, p0 + pper);
  uint32_t nroots = 0;
  for (uint32_t q = p0; q < p1; ++q) {
    const uint32_t i = idx[q];
    nroots += parent[i] == i ? 1u : 0u;
  }
  uint64_t total_roots;
  uint32_t rpos = (uint32_t)block_excl_scan(nroots, s_scan, tid, nw, &total_roots);
  for (uint32_t q = p0; q < p1; ++q) {
    const uint32_t i = idx[q];
    if (parent[i] == i) roots[rpos++] = i;
  }
  __syncthreads();
  // number the roots inside their cluster, te roots per round: rank inside the wave by ballots over
  // the clusters present, across waves and rounds by the small per-(wave, cluster) table.  A root's
  // parent entry becomes 0x80000000 | its 1-based number (nobody walks the forest any more).
  const uint32_t R = (uint32_t)total_roots;
  for (uint32_t r0 = 0; r0 < R; r0 += te) {
    const uint32_t q = r0 + tid;
    const bool act = q < R;
    const uint32_t b = act ? roots[q] : 0u;
    const uint32_t cid = act ? cl[b] : 0xFFFFFFFFu;
    uint32_t in_wave = 0;
    unsigned long long rem = __ballot(act);
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    while (rem) {
      const uint32_t j = __builtin_amdgcn_readlane(cid, __builtin_ctzll(rem));
      const unsigned long long bal = __ballot(cid == j);
      rem &= ~bal;
      if (cid == j) in_wave = (uint32_t)__popcll(bal & lt);
      if (lane == 0) s_wcnt[wave][j] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (act) {
      uint32_t before = s_base[cid];
      for (uint32_t wv = 0; wv < wave; ++wv) before += s_wcnt[wv][cid];
      parent[b] = 0x80000000u | (before + in_wave + 1);
    }
    __syncthreads();
    if (tid < k) {
      uint32_t s = 0;
      for (uint32_t wv = 0; wv < nw; ++wv) { s += s_wcnt[wv][tid]; s_wcnt[wv][tid] = 0; }
      s_base[tid] += s;
    }
    __syncthreads();
  }
  if (tid == 0) {  // BLOCK_TYPE_BACKGROUND = 0 (libs/codec.hpp:6); :620, the count includes label 0
    uint32_t offset = 0;
    for (uint32_t cid = 0; cid < k; ++cid) {
      const uint32_t cnt = s_base[cid];
      s_base[cid] = offset;
      offset += cnt + 1;
    }
  }
  __syncthreads();
  for (uint32_t q = tid; q < nf; q += te) {
    const uint32_t i = idx[q];
    uint32_t pv = parent[i];
    if (!(pv & 0x80000000u)) pv = parent[pv];
    types[i] = (pv & 0x7FFFFFFFu) + s_base[cl[i]];  // :617
  }
}

// One side stream + fork/join events per host thread and device, created on first use and kept.
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};

// A few per device, one per caller stream (least recently used slot for a new caller): launches enqueued
// on different streams (a pipelined encoder runs the segmentation of consecutive steps on streams of their own) must
// not queue their heavy attempts behind each other on ONE side stream.
static SideStream* side_stream(hipStream_t caller) {
  constexpr int kMaxDevices = 64, kPool = 4;
  struct Slot { hipStream_t caller = nullptr; uint64_t last_use = 0; SideStream s; };
  static thread_local Slot per_device[kMaxDevices][kPool];
  static thread_local uint64_t tick = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  Slot* slot = nullptr;
  for (int i = 0; i < kPool; ++i) {  // this caller's slot, else the least recently used one (stream order makes reuse safe)
    Slot& c = per_device[dev][i];
    if (c.last_use && c.caller == caller) { slot = &c; break; }
    if (!slot || c.last_use < slot->last_use) slot = &c;
  }
  slot->caller = caller;
  slot->last_use = ++tick;
  SideStream& s = slot->s;
  if (!s.stream) {
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) { s.stream = nullptr; return nullptr; }
    if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess)
      return nullptr;
  }
  return &s;
}

uint64_t segment_workspace_per_frame(uint32_t n, uint32_t attempts) {
  return Workspace{nullptr, n, attempts ? attempts : 1}.bytes();
}

int launch_segment(const uint8_t* d_mask, const float* d_mv, uint32_t mfw, uint32_t mfh, uint32_t n_frames,
                   uint32_t mv_bw, uint32_t mv_bh, const svc_segment_params& p, uint64_t seed, uint8_t* d_ws,
                   uint32_t* d_types, uint32_t flags, hipStream_t stream) {
  if (n_frames == 0) return SVC_OK;
  if (p.cluster_count > kMaxK)
    return fail(SVC_ERR_UNSUPPORTED, "segment: cluster_count %u exceeds %u", p.cluster_count, kMaxK);
  if (p.attempt_count > kMaxAttempts)
    return fail(SVC_ERR_UNSUPPORTED, "segment: attempt_count %u exceeds %u", p.attempt_count, kMaxAttempts);
  if ((uint64_t)mfw * mfh >= (1ull << 31))
    return fail(SVC_ERR_UNSUPPORTED, "segment: motion field of %u x %u blocks is too large", mfw, mfh);
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
  const dim3 grid_a(n_frames, p.attempt_count);
  const bool small_coords = (uint64_t)mfw * mv_bw < (1u << 14) && (uint64_t)mfh * mv_bh < (1u << 14);
  a.packable = (small_coords && mfw <= 512 && mfh <= 512) ? 1u : 0u;
  constexpr size_t kLdsBig = 144 * 1024;  // of the CU's 160 KB; the static part takes ~5 KB
  // two bit fields (rows padded to words) + the flat bitmap (whole 1024-bit trips + 2 words)
  const size_t bit_words = 2 * (size_t)((mfw + 31) / 32) * mfh + (size_t)((a.n + kTA - 1) / kTA) * (kTA / 32) + 2;
  a.bits_bytes = (uint32_t)((4 * bit_words + 15) & ~(size_t)15);
  if (a.bits_bytes + 4096 > kLdsBig)
    return fail(SVC_ERR_UNSUPPORTED, "segment: motion field of %u x %u blocks is too large", mfw, mfh);
  a.lds_bytes = (uint32_t)(a.bits_bytes + 4 * n4 <= kLdsBig ? a.bits_bytes + 4 * n4 : kLdsBig);
  // SVC_LAUNCH_BESIDE (fields of at most 8 192 blocks): shapes that fit on a CU NEXT TO the bandwidth kernels of a
  // pipelined schedule -- the transform kernel leaves 4 wave slots and 256 VGPRs per SIMD and 12 KB of LDS.  The
  // foreground list and the labelling arrays go to the workspace instead of 33-41 KB of LDS, and ONE 256-lane attempt
  // launch takes every frame: the 1 024-lane launch needs an EMPTY CU for each of its frames x attempts workgroups (4
  // waves x 128 VGPRs on every SIMD), even those that exit at once, so beside a kernel that keeps every CU partly
  // occupied it does not start until that kernel drains.  Scene cuts then run on the 256-lane workspace path (slower
  // alone, hidden beside the transform).
  const bool small = (flags & SVC_LAUNCH_BESIDE) != 0 && a.n <= kRegPts * kTA;
  a.take_all = small ? 1u : 0u;
  if (small) a.lds_bytes = a.bits_bytes + 16;  // the foreground list goes to the workspace
  // Few frames of a large field (the shard of a multi-GPU 4K run): a heavy frame's attempts run as launch sequences over
  // wide_g workgroups each instead of one workgroup (segment_wide_*_kernel) -- when one workgroup per (frame, attempt) would
  // not even fill the CUs.  (Up to frames x attempts = CUs: C5's whole 64-frame clip, 192 of them, runs 0.85 -> 0.56 ms alone and
  // level beside the main stream's kernels, profiles/r03_ab_wide64.txt.)
  a.wide_g = 0;
  a.wide_step = 0;
  if (!small && a.packable && a.n > kRegPts * kTA && !(flags & SVC_LAUNCH_NO_WIDE)) {
    static const uint32_t cus = [] {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
      return (uint32_t)v;
    }();
    const uint64_t work = (uint64_t)n_frames * p.attempt_count;
    if ((flags & SVC_LAUNCH_WIDE) && (n_frames > 65535u || p.attempt_count > 65535u))
      return fail(SVC_ERR_UNSUPPORTED, "segment: SVC_LAUNCH_WIDE takes at most 65535 frames per call (grid.y), got %u", n_frames);
    // The sequence is one launch per Lloyd iteration (plus one per centre above kWideRegPts x 1024 blocks) whether or not the
    // attempts have converged: 5-10 us each, so beyond ~32 launches it is no shorter than the one-workgroup form it replaces.
    const uint64_t seq = (uint64_t)p.max_iter_count + (a.n > kWideRegPts * kTA ? p.cluster_count : 1u);
    if ((flags & SVC_LAUNCH_WIDE) || (work <= cus && seq <= 32))
      a.wide_g = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4 * (uint64_t)cus / std::max<uint64_t>(work, 1), 2), kWideMaxG);  // ~4 workgroups per CU
  }
  hipLaunchKernelGGL(segment_prepare_kernel, dim3(n_frames), dim3(kTA), a.lds_bytes, stream, a);
  a.lds_bytes = 0;
  auto launch_wide = [&](hipStream_t st, bool seeds) {
    SegArgs w = a;
    const dim3 grid_w(a.wide_g, n_frames, p.attempt_count);
    for (uint32_t j = 0; seeds && j < a.k; ++j) {
      w.wide_step = j;
      hipLaunchKernelGGL(segment_wide_seed_kernel, grid_w, dim3(kTW), 0, st, w);
    }
    for (uint32_t it = 0; it < a.max_iter; ++it) {  // the close of iteration max_iter - 1 is the labelling kernel's first step
      w.wide_step = it;
      hipLaunchKernelGGL(segment_wide_lloyd_kernel, grid_w, dim3(kTW), 0, st, w);
    }
  };
  if (small) {
    hipLaunchKernelGGL((segment_attempt_kernel<256>), grid_a, dim3(256), 0, stream, a);
  } else if (a.wide_g) {
    // every frame of packed points, light or heavy, rides the launch sequence (its length is set by k and max_iter, not by
    // the frames in it; a separate launch for the light frames would only add its 0.1 ms to the chain); the one attempt
    // launch in front of it takes the frames of unpacked points (none after block matching) and exits otherwise
    if (a.n <= kWideRegPts * kTA) {  // seeding in one launch (points in registers), then the Lloyd launches
      hipLaunchKernelGGL(segment_wide_head_kernel, grid_a, dim3(kTA), 0, stream, a);
      launch_wide(stream, false);
    } else {
      SegArgs heavy = a;
      heavy.lds_bytes = (uint32_t)(8 * n4 <= kLdsBig ? 8 * n4 : kLdsBig);
      hipLaunchKernelGGL((segment_attempt_kernel<kTA>), grid_a, dim3(kTA), heavy.lds_bytes, stream, heavy);
      launch_wide(stream, true);
    }
  } else if (a.n > kLightMax) {
    // The two attempt launches are independent (each frame belongs to exactly one): the heavy one goes to a
    // side stream, forked after the prepare kernel and joined before the labelling, so a scene cut's long
    // workgroups run beside the light frames instead of after them.  Not while `stream` is being captured
    // into a hipGraph: a fork out of a stream that is itself a fork of the capture's origin sends ROCm 7.2's
    // hipStreamEndCapture into unbounded recursion, so a capture gets the two launches in stream order.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    SVC_HIP_TRY(hipStreamIsCapturing(stream, &capturing));
    SegArgs heavy = a;
    // more than kRegPts points per lane are possible: LDS for the large-frame path
    if (a.n > kRegPts * kTA) heavy.lds_bytes = (uint32_t)(8 * n4 <= kLdsBig ? 8 * n4 : kLdsBig);
    if (capturing == hipStreamCaptureStatusNone && !(flags & SVC_LAUNCH_NO_FORK)) {
      SideStream* side = side_stream(stream);
      if (!side) return fail(SVC_ERR_HIP, "segment: cannot create the side stream");
      SVC_HIP_TRY(hipEventRecord(side->fork, stream));
      SVC_HIP_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
      hipLaunchKernelGGL((segment_attempt_kernel<kTA>), grid_a, dim3(kTA), heavy.lds_bytes, side->stream, heavy);
      SVC_HIP_TRY(hipEventRecord(side->join, side->stream));
      hipLaunchKernelGGL((segment_attempt_kernel<256>), grid_a, dim3(256), 0, stream, a);
      SVC_HIP_TRY(hipStreamWaitEvent(stream, side->join, 0));
    } else {
      hipLaunchKernelGGL((segment_attempt_kernel<kTA>), grid_a, dim3(kTA), heavy.lds_bytes, stream, heavy);
      hipLaunchKernelGGL((segment_attempt_kernel<256>), grid_a, dim3(256), 0, stream, a);
    }
  } else {
    hipLaunchKernelGGL((segment_attempt_kernel<256>), grid_a, dim3(256), 0, stream, a);
  }
  if (small)
    hipLaunchKernelGGL((segment_label_kernel<false, false>), dim3(n_frames), dim3(kTA), 0, stream, a);
  else if (5 * n4 <= kLdsBig)
    hipLaunchKernelGGL((segment_label_kernel<true, true>), dim3(n_frames), dim3(kTA), 5 * n4, stream, a);
  else if (4 * n4 <= kLdsBig)
    hipLaunchKernelGGL((segment_label_kernel<true, false>), dim3(n_frames), dim3(kTA), 4 * n4, stream, a);
  else
    hipLaunchKernelGGL((segment_label_kernel<false, false>), dim3(n_frames), dim3(kTA), 0, stream, a);
  return check_launch("segment kernels");
}

}  // namespace svc
