// global_motion.hip -- the three whole-frame global-motion estimators of libs/motion.hpp:38-59.
//
// None of them has a caller in the reference (the encoder uses RANSAC, libs/encoder.cpp:491-498);
// they are provided so that libsvc_motion.so replaces the reference's `motion` library symbol for
// symbol.  Semantics follow libs/motion.cpp:45-142 with ONE documented deviation:
//   EstimateGlobalMotionExhaustiveSearch loops `for (int dy = -R; dy <= search_range /*uint*/; ...)`
//   (motion.cpp:72, :81): the comparison converts dy to unsigned, so for R > 0 the body never runs
//   and the reference returns {0, 0}, FLT_MAX.  Here the loop runs over dy, dx in [-R, R] as
//   written: MAD of the overlap of the two frames shifted by (dx, dy), strict `<`, raster order (dy
//   outer), i.e. the first minimum wins (:90-93).  For R = 0 both agree (one candidate).
//
// Kernels.  global_sad_kernel: every candidate's SAD over the overlap.  A workgroup owns a band of
// anchor rows for one dy and a chunk of 32 consecutive dx; a lane walks anchor dwords, keeps the two
// tracked dwords that straddle the shifted position and forms the four byte shifts with
// v_alignbyte_b32 + v_sad_u8 (interior dwords), bytes near the row ends one at a time; 32 register
// accumulators, reduced in the wave, across waves in LDS, one 64-bit atomic per candidate and
// workgroup.  global_pick_kernel: MAD = (float)(u32)sad / (float)(u32)(bw * bh) exactly as Mad
// computes it (motion.cpp:27-40: 32-bit unsigned sum and count, both wrap like the reference's), first
// strict minimum by an unsigned min over (mad bits, raster index).  global_avg_kernel: the running
// mean of motion.cpp:49-51, an order-dependent f32 recurrence, walked by one lane.
#include "svc_common.hpp"

namespace svc {

struct GlobalSadArgs {
  const uint8_t* tracked;
  const uint8_t* anchor;
  uint64_t pair_stride;
  uint32_t w, h;
  int32_t range;        // R
  uint32_t rows_per_wg;
  uint32_t dx_chunks;   // ceil((2R + 1) / 32)
  uint64_t* sad;        // [pairs][2R + 1][2R + 1]
  uint32_t aligned4;    // planes, stride and width allow dword loads
};

constexpr int kDxChunk = 32;

__global__ __launch_bounds__(256) void global_sad_kernel(GlobalSadArgs a) {
  __shared__ unsigned long long s_acc[kDxChunk];
  const uint32_t tid = threadIdx.x;
  const int32_t R = a.range, side = 2 * R + 1;
  const uint32_t pair = blockIdx.z / a.dx_chunks, chunk = blockIdx.z - pair * a.dx_chunks;
  const int32_t dy = (int32_t)blockIdx.y - R;
  const int32_t dx0 = -R + (int32_t)chunk * kDxChunk;  // first dx of this chunk
  const int32_t ndx = side - (int32_t)chunk * kDxChunk < kDxChunk ? side - (int32_t)chunk * kDxChunk : kDxChunk;
  const uint8_t* T = a.tracked + (size_t)pair * a.pair_stride;
  const uint8_t* A = a.anchor + (size_t)pair * a.pair_stride;
  const int32_t w = (int32_t)a.w, h = (int32_t)a.h;
  if (tid < kDxChunk) s_acc[tid] = 0;
  __syncthreads();

  uint32_t acc[kDxChunk];
#pragma unroll
  for (int i = 0; i < kDxChunk; ++i) acc[i] = 0;

  const int32_t row0 = (int32_t)(blockIdx.x * a.rows_per_wg);
  const int32_t row1 = row0 + (int32_t)a.rows_per_wg < h ? row0 + (int32_t)a.rows_per_wg : h;
  const int32_t w4 = (w + 3) >> 2;
  for (int32_t ay = row0; ay < row1; ++ay) {
    const int32_t ty = ay + dy;
    if (ty < 0 || ty >= h) continue;  // anchor row without a partner under this dy (uniform)
    const uint8_t* arow = A + (size_t)ay * w;
    const uint8_t* trow = T + (size_t)ty * w;
    for (int32_t x4 = (int32_t)tid; x4 < w4; x4 += 256) {
      const int32_t ax = 4 * x4;
      // interior: all four anchor bytes exist and every shift of the chunk stays inside the row (w is a multiple
      // of 4 on this path, so the second dword of a non-zero funnel shift is inside the row too; with a zero shift
      // it is not needed and not loaded past the row)
      const bool interior = a.aligned4 && ax + 3 < w && ax + dx0 >= 0 && ax + dx0 + ndx - 1 + 3 < w;
      if (interior) {
        const uint32_t av = *reinterpret_cast<const uint32_t*>(arow + ax);
        // tracked bytes at ax + dx, dx = dx0 + i: dword index q = (ax + dx) >> 2, shift s = (ax + dx) & 3
        int32_t pos = ax + dx0;
        int32_t q = pos >> 2;  // pos >= 0 here
        uint32_t lo = *reinterpret_cast<const uint32_t*>(trow + 4 * q);
        uint32_t hi = 4 * q + 7 < w ? *reinterpret_cast<const uint32_t*>(trow + 4 * q + 4) : 0u;
#pragma unroll
        for (int i = 0; i < kDxChunk; ++i) {
          if (i < ndx) {
            const int32_t p = pos + i;
            if ((p >> 2) != q) {
              q = p >> 2;
              lo = hi;
              hi = 4 * q + 7 < w ? *reinterpret_cast<const uint32_t*>(trow + 4 * q + 4) : 0u;
            }
            const uint32_t tv = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(p & 3));
            acc[i] = __builtin_amdgcn_sad_u8(av, tv, acc[i]);
          }
        }
      } else {
#pragma unroll 1
        for (int32_t b = 0; b < 4; ++b) {
          const int32_t x = ax + b;
          if (x >= w) break;
          const int32_t av = arow[x];
#pragma unroll
          for (int i = 0; i < kDxChunk; ++i) {
            const int32_t tx = x + dx0 + i;
            if (i < ndx && tx >= 0 && tx < w) {
              const int32_t d = av - (int32_t)trow[tx];
              acc[i] += (uint32_t)(d < 0 ? -d : d);
            }
          }
        }
      }
    }
  }
  // wave reduce, then one LDS atomic per wave and candidate, one global atomic per workgroup and candidate
#pragma unroll
  for (int i = 0; i < kDxChunk; ++i) {
    uint32_t v = acc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0 && i < ndx && v) atomicAdd(&s_acc[i], (unsigned long long)v);
  }
  __syncthreads();
  if ((int32_t)tid < ndx && s_acc[tid])
    atomicAdd(reinterpret_cast<unsigned long long*>(a.sad) + ((size_t)pair * side + (size_t)(dy + R)) * side + (dx0 + R) + tid,
              s_acc[tid]);
}

struct GlobalPickArgs {
  const uint64_t* sad;
  uint32_t w, h;
  int32_t range;
  float* gm;       // [pairs][2]
  float* min_mad;  // [pairs] or null
  uint32_t combine;  // 1: gm = 2 * gm + pick (EstimateGlobalMotionHierarchical, motion.cpp:140)
};

__global__ __launch_bounds__(256) void global_pick_kernel(GlobalPickArgs a) {
  __shared__ unsigned long long s_best[4];
  const uint32_t tid = threadIdx.x, pair = blockIdx.x;
  const int32_t R = a.range, side = 2 * R + 1, n = side * side;
  unsigned long long best = ~0ull;
  for (int32_t i = (int32_t)tid; i < n; i += 256) {
    const int32_t dy = i / side - R, dx = i % side - R;
    const uint32_t bw = a.w - (uint32_t)(dx < 0 ? -dx : dx), bh = a.h - (uint32_t)(dy < 0 ? -dy : dy);
    const uint32_t sad = (uint32_t)a.sad[(size_t)pair * n + i];  // the reference sums in a 32-bit unsigned (motion.cpp:27)
    const uint32_t count = bw * bh;                              // :36
    const float mad = (float)sad / (float)count;                 // :38
    if (mad == mad) {  // a NaN (count = 0) never passes `mad < *min_mad`
      const unsigned long long key = ((unsigned long long)__float_as_uint(mad) << 32) | (uint32_t)i;  // mad >= 0: bits order like values
      best = key < best ? key : best;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(best, off, 64);
    best = o < best ? o : best;
  }
  if ((tid & 63) == 0) s_best[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < 4; ++k) best = s_best[k] < best ? s_best[k] : best;
    float mx = 0.f, my = 0.f, mm = 3.402823466e+38f;  // *global_motion = {}, *min_mad = FLT_MAX (:66-67)
    if (best != ~0ull && __uint_as_float((uint32_t)(best >> 32)) < mm) {
      const int32_t i = (int32_t)(uint32_t)best;
      mx = (float)(i % side - R);
      my = (float)(i / side - R);
      mm = __uint_as_float((uint32_t)(best >> 32));
    }
    float* g = a.gm + 2 * (size_t)pair;
    if (a.combine) {
      g[0] = 2.0f * g[0] + mx;
      g[1] = 2.0f * g[1] + my;
    } else {
      g[0] = mx;
      g[1] = my;
    }
    if (a.min_mad) a.min_mad[pair] = mm;
  }
}

// libs/motion.cpp:45-53: avg += (mv[i] - avg) * (1.0f / (i + 1)), i ascending -- an f32 recurrence.  The
// reciprocals do not depend on the chain: the other lanes prepare them (and stage the vectors) in LDS, lane 0
// walks the chain.
__global__ __launch_bounds__(256) void global_avg_kernel(const float* mv, uint32_t n, float* out) {
  __shared__ float s_x[1024], s_y[1024], s_r[1024];
  const uint32_t tid = threadIdx.x, frame = blockIdx.x;
  const float* m = mv + 2 * (size_t)frame * n;
  float ax = 0.f, ay = 0.f;
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t cnt = n - base < 1024 ? n - base : 1024;
    for (uint32_t i = tid; i < cnt; i += 256) {
      s_x[i] = m[2 * (size_t)(base + i)];
      s_y[i] = m[2 * (size_t)(base + i) + 1];
      s_r[i] = 1.0f / (float)(base + i + 1);  // correctly rounded divide, as the CPU's divss
    }
    __syncthreads();
    if (tid == 0) {
      for (uint32_t i = 0; i < cnt; ++i) {
        const float r = s_r[i];
        ax = ax + (s_x[i] - ax) * r;  // no contraction: the build has -ffp-contract=off
        ay = ay + (s_y[i] - ay) * r;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    out[2 * (size_t)frame] = ax;
    out[2 * (size_t)frame + 1] = ay;
  }
}

uint64_t global_ebma_workspace_bytes(uint32_t range, uint32_t n_pairs) {
  const uint64_t side = 2ull * range + 1;
  return side * side * 8 * n_pairs;
}

int launch_global_ebma(const uint8_t* d_tracked, const uint8_t* d_anchor, uint64_t pair_stride, uint32_t n_pairs, uint32_t w,
                       uint32_t h, uint32_t range, uint8_t* d_ws, float* d_gm, float* d_min_mad, bool combine,
                       hipStream_t stream) {
  if (n_pairs == 0) return SVC_OK;
  const uint32_t side = 2 * range + 1;
  SVC_HIP_TRY(hipMemsetAsync(d_ws, 0, global_ebma_workspace_bytes(range, n_pairs), stream));
  GlobalSadArgs s{};
  s.tracked = d_tracked; s.anchor = d_anchor; s.pair_stride = pair_stride;
  s.w = w; s.h = h; s.range = (int32_t)range;
  s.rows_per_wg = 16;
  s.dx_chunks = div_up(side, kDxChunk);
  s.sad = reinterpret_cast<uint64_t*>(d_ws);
  s.aligned4 = (w % 4 == 0 && pair_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(d_tracked) & 3) == 0 &&
                (reinterpret_cast<uintptr_t>(d_anchor) & 3) == 0) ? 1u : 0u;
  const uint64_t gz = (uint64_t)n_pairs * s.dx_chunks;
  if (gz > 65535 || side > 65535)
    return fail(SVC_ERR_UNSUPPORTED, "global ebma: %u pairs x search range %u exceed one launch", n_pairs, range);
  hipLaunchKernelGGL(global_sad_kernel, dim3(div_up(h, s.rows_per_wg), side, (uint32_t)gz), dim3(256), 0, stream, s);
  GlobalPickArgs p{};
  p.sad = s.sad; p.w = w; p.h = h; p.range = (int32_t)range;
  p.gm = d_gm; p.min_mad = d_min_mad; p.combine = combine ? 1u : 0u;
  hipLaunchKernelGGL(global_pick_kernel, dim3(n_pairs), dim3(256), 0, stream, p);
  return check_launch("global ebma kernels");
}

int launch_global_avg(const float* d_mv, uint32_t blocks, uint32_t n_frames, float* d_out, hipStream_t stream) {
  if (n_frames == 0) return SVC_OK;
  hipLaunchKernelGGL(global_avg_kernel, dim3(n_frames), dim3(256), 0, stream, d_mv, blocks, d_out);
  return check_launch("global_avg_kernel");
}

}  // namespace svc
