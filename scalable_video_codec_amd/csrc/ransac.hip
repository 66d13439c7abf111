// ransac.hip -- EstimateGlobalMotionRansac (reference libs/motion.cpp:182-266),
// batched: one workgroup per frame's motion field.
//
// The reference seeds a function-local static engine from std::random_device
// (:186-187) and draws from [0, N] inclusive (:208, an out-of-bounds read).  Here the
// accepted draws are an explicit input (svc_hip.h), so a run is reproducible and
// checkable; everything downstream of the draws is bit-identical:
//   - the model of an iteration is the SEQUENTIAL f32 sum of the sampled MVs times
//     1/n (:151-163) -- one lane does it, n is tiny;
//   - the inlier test is elementwise f32, (gm.x-m.x)^2 + (gm.y-m.y)^2 < thresh^2
//     (:228), evaluated by all lanes with FP contraction off (the reference is built
//     for baseline x86-64: no FMA);
//   - `>=` keeps the LATER of two equally good iterations (:233);
//   - the final mean and RMSE over the inliers are sequential f32 sums in index order
//     (:255-259), which are order-dependent, so one lane walks the mask.  That serial
//     tail is ~N dependent adds per frame; frames run in parallel on separate CUs and
//     the whole launch sits beside the DCT on another stream.
#include "svc_common.hpp"

#pragma clang fp contract(off)

namespace svc {

struct RansacArgs {
  const float* mv;       // [frames][blocks][2]
  const uint32_t* samples;  // [frames][iters][subset]
  uint32_t blocks, iters, subset;
  float thresh;
  float* gm;             // [frames][2] in/out
  float* rmse;           // [frames]
  uint8_t* mask;         // [frames][blocks]
  uint32_t* count;       // [frames]
};

__device__ __forceinline__ bool is_inlier(float gx, float gy, float mx, float my, float t2) {
  const float dx = gx - mx, dy = gy - my;
  return dx * dx + dy * dy < t2;
}

constexpr uint32_t kChunk = 4096;  // LDS staging: 4096 float2 = 32 KiB

// Sequential f32 sums over an LDS-staged chunk, in index order, by ONE lane: the
// reference's accumulation order (motion.cpp:156-159, :172-175) is part of the result.
// Entries that are not inliers were staged as +0.0f, which leaves a sum unchanged.
__device__ __forceinline__ void serial_sum2(const float2* st, uint32_t n, float& sx, float& sy) {
  uint32_t i = 0;
  for (; i + 4 <= n; i += 4) {
    const float4 a = *reinterpret_cast<const float4*>(st + i);
    const float4 b = *reinterpret_cast<const float4*>(st + i + 2);
    sx = sx + a.x; sy = sy + a.y;
    sx = sx + a.z; sy = sy + a.w;
    sx = sx + b.x; sy = sy + b.y;
    sx = sx + b.z; sy = sy + b.w;
  }
  for (; i < n; ++i) { sx = sx + st[i].x; sy = sy + st[i].y; }
}

__device__ __forceinline__ void serial_sum1(const float* st, uint32_t n, float& acc) {
  // One lane, strictly in index order.  Two register blocks of 32 values alternate: while one is
  // being added the other is in flight from LDS, and no value is ever copied between registers, so
  // the chain runs at the dependent-add issue rate (4 cycles per term) instead of at LDS latency.
  const uint32_t nb = n / 32;
  if (nb) {
    float4 A[8], B[8];
#define SVC_LOAD32(X, off) _Pragma("unroll") for (int k = 0; k < 8; ++k) X[k] = *reinterpret_cast<const float4*>(st + (off) + 4 * k)
#define SVC_ADD32(X) _Pragma("unroll") for (int k = 0; k < 8; ++k) { acc += X[k].x; acc += X[k].y; acc += X[k].z; acc += X[k].w; }
    SVC_LOAD32(A, 0);
    uint32_t b = 0;
    for (; b + 2 <= nb; b += 2) {
      SVC_LOAD32(B, (b + 1) * 32);
      SVC_ADD32(A);
      if (b + 2 < nb) SVC_LOAD32(A, (b + 2) * 32);
      SVC_ADD32(B);
    }
    if (b < nb) SVC_ADD32(A);
#undef SVC_LOAD32
#undef SVC_ADD32
  }
  for (uint32_t i = nb * 32; i < n; ++i) acc += st[i];
}

constexpr uint32_t kModelsPerPass = 8;  // iteration models scored per pass over the field

// block-wide sum of one counter per thread; result valid in every thread
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* s_red, uint32_t tid) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();  // s_red may still be read from the previous reduction
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ __launch_bounds__(256) void ransac_kernel(RansacArgs a) {
  __shared__ __attribute__((aligned(16))) float2 s_stage[kChunk];
  __shared__ float s_model[kModelsPerPass][2];
  __shared__ uint32_t s_red[4];
  __shared__ int s_isum[2];
  __shared__ uint32_t s_flag, s_mag;
  __shared__ float s_gm[2];
  const uint32_t tid = threadIdx.x, frame = blockIdx.x;
  const float2* mv = reinterpret_cast<const float2*>(a.mv) + (size_t)frame * a.blocks;
  const uint32_t* samples = a.samples + (size_t)frame * a.iters * a.subset;
  uint8_t* mask = a.mask + (size_t)frame * a.blocks;
  const float t2 = a.thresh * a.thresh;

  // ---- iterations (motion.cpp:210-238), kModelsPerPass at a time: the models of a group
  // are built by separate lanes, then ONE pass over the field counts the inliers of all.
  uint32_t best_n = 0, best_it = 0;
  float bgx = 0.f, bgy = 0.f;
  for (uint32_t it0 = 0; it0 < a.iters; it0 += kModelsPerPass) {
    const uint32_t nm = min(kModelsPerPass, a.iters - it0);
    if (tid < nm) {
      float sx = 0.f, sy = 0.f;  // sequential f32 sum of the subset (motion.cpp:156-160)
      for (uint32_t i = 0; i < a.subset; ++i) {
        const float2 m = mv[min(samples[(size_t)(it0 + tid) * a.subset + i], a.blocks - 1)];  // a draw past the field (the reference's [0, N], motion.cpp:208) never leaves it: svc_hip.h
        sx = sx + m.x;
        sy = sy + m.y;
      }
      const float inv = 1.0f / (float)a.subset;
      s_model[tid][0] = sx * inv;
      s_model[tid][1] = sy * inv;
    }
    __syncthreads();
    float gx[kModelsPerPass], gy[kModelsPerPass];
    uint32_t cnt[kModelsPerPass];
#pragma unroll
    for (uint32_t k = 0; k < kModelsPerPass; ++k) {
      gx[k] = s_model[k < nm ? k : 0][0];
      gy[k] = s_model[k < nm ? k : 0][1];
      cnt[k] = 0;
    }
    for (uint32_t i = tid; i < a.blocks; i += 256) {
      const float2 m = mv[i];
#pragma unroll
      for (uint32_t k = 0; k < kModelsPerPass; ++k) cnt[k] += is_inlier(gx[k], gy[k], m.x, m.y, t2) ? 1u : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < kModelsPerPass; ++k) {
      const uint32_t total = block_sum(cnt[k], s_red, tid);
      if (k < nm && total >= best_n) {  // motion.cpp:233, in iteration order: ties -> later
        best_n = total;
        best_it = it0 + k;
        bgx = gx[k];
        bgy = gy[k];
      }
    }
    __syncthreads();  // s_model is rewritten by the next group
  }
  const bool any_iter = a.iters > 0;

  if (best_n < a.subset) {
    for (uint32_t i = tid; i < a.blocks; i += 256) {
      const float2 m = mv[i];
      mask[i] = (any_iter && is_inlier(bgx, bgy, m.x, m.y, t2)) ? 1 : 0;
    }
    // motion.cpp:240-242: RMSE of the best subset against the INCOMING global motion
    if (tid == 0) {
      const float ix = a.gm[2 * frame], iy = a.gm[2 * frame + 1];
      float acc = 0.f;
      for (uint32_t i = 0; i < a.subset; ++i) {
        const float2 m = mv[min(samples[(size_t)best_it * a.subset + i], a.blocks - 1)];
        const float dx = m.x - ix, dy = m.y - iy;
        acc += dx * dx + dy * dy;
      }
      a.gm[2 * frame] = bgx;
      a.gm[2 * frame + 1] = bgy;
      a.rmse[frame] = sqrtf(acc / (float)a.subset);
      a.count[frame] = best_n;
    }
    return;
  }

  // ---- inlier mask (== best_inliers, motion.cpp:244-253) and, in the same pass, the
  // final model = mean of the inliers (motion.cpp:255-256).
  // Fast path: when every inlier MV component is an integer and the sum of magnitudes stays
  // below 2^24 (always true for block-matching output), every partial sum of the
  // reference's sequential f32 accumulation is exactly representable, so an integer
  // reduction in any order gives the identical float.  Otherwise: the serial walk.
  if (tid == 0) { s_flag = 1u; s_isum[0] = 0; s_isum[1] = 0; s_mag = 0; }
  __syncthreads();
  {
    bool ok = true;
    int ix = 0, iy = 0;
    uint32_t mag = 0;
    for (uint32_t i = tid; i < a.blocks; i += 256) {
      const float2 m = mv[i];
      const bool in = is_inlier(bgx, bgy, m.x, m.y, t2);
      mask[i] = in ? 1 : 0;
      if (!in) continue;
      ok = ok && m.x == truncf(m.x) && m.y == truncf(m.y) && fabsf(m.x) <= 32768.f && fabsf(m.y) <= 32768.f;
      if (ok) {
        ix += (int)m.x; iy += (int)m.y;
        mag += (uint32_t)fabsf(m.x) + (uint32_t)fabsf(m.y);
      }
    }
    if (!ok || mag >= (1u << 24)) s_flag = 0u;  // 256 addends below 2^24 cannot wrap s_mag
    atomicAdd(&s_isum[0], ix);
    atomicAdd(&s_isum[1], iy);
    atomicAdd(&s_mag, mag & 0xFFFFFFu);
  }
  __syncthreads();
  const bool exact_int = s_flag != 0u && s_mag < (1u << 24);
  float sx = 0.f, sy = 0.f;
  if (exact_int) {
    sx = (float)s_isum[0];
    sy = (float)s_isum[1];
  } else {
    for (uint32_t base = 0; base < a.blocks; base += kChunk) {
      const uint32_t n = min(kChunk, a.blocks - base);
      for (uint32_t i = tid; i < n; i += 256) {
        const float2 m = mv[base + i];
        s_stage[i] = is_inlier(bgx, bgy, m.x, m.y, t2) ? m : make_float2(0.f, 0.f);
      }
      __syncthreads();
      if (tid == 0) serial_sum2(s_stage, n, sx, sy);
      __syncthreads();
    }
    if (tid == 0) { s_gm[0] = sx; s_gm[1] = sy; }
    __syncthreads();
    sx = s_gm[0];
    sy = s_gm[1];
  }
  const float inv = 1.0f / (float)best_n;
  const float out_gx = sx * inv, out_gy = sy * inv;

  // ---- RMSE (motion.cpp:258-259, :165-180): terms in parallel, the sum in order -----
  float acc = 0.f;
  float* terms = reinterpret_cast<float*>(s_stage);
  for (uint32_t base = 0; base < a.blocks; base += 2 * kChunk) {
    const uint32_t n = min(2 * kChunk, a.blocks - base);
    for (uint32_t i = tid; i < n; i += 256) {
      const float2 m = mv[base + i];
      const float dx = m.x - out_gx, dy = m.y - out_gy;
      terms[i] = is_inlier(bgx, bgy, m.x, m.y, t2) ? dx * dx + dy * dy : 0.f;
    }
    __syncthreads();
    if (tid == 0) serial_sum1(terms, n, acc);
    __syncthreads();
  }
  if (tid == 0) {
    a.gm[2 * frame] = out_gx;
    a.gm[2 * frame + 1] = out_gy;
    a.rmse[frame] = sqrtf(acc / (float)best_n);
    a.count[frame] = best_n;
  }
}

// ---- the same, with the motion field in registers ---------------------------------------------
// The kernel above walks the field in L2 half a dozen times and pays a block-wide reduction (two
// barriers) per model.  For fields of up to PER x G blocks every lane keeps ITS blocks (i = u * G + gt)
// in registers for the whole kernel: the models of a pass are scored against registers, their
// counts reduced inside the wave and summed across waves by one lane after ONE barrier, and the
// mask / mean / RMSE passes never touch memory for the field again.  Out-of-range slots hold
// +inf, which no model accepts.  Same arithmetic, same order: bit-identical results.
//
// A workgroup of T lanes takes F frames, G = T / F lanes each.  What that buys is the serial tail:
// the in-order f32 sums are one dependent add per term (~8.7 cycles each, 8 160 terms at 1080p)
// whatever the lane count, and two workgroups sharing a CU were measured to stretch each other's
// chain by 1.7x -- a 300-frame clip on 256 CUs doubles up on 43 of them.  With F = 2 the clip is 150
// workgroups, one per CU, and the chains of a workgroup's frames run as LANES of one wave, in the
// same instructions.  Every barrier is reached by all groups: nothing below returns early or loops
// on a per-frame condition.
constexpr uint32_t kPassModels = 16;

// DEFER: the in-order RMSE sum over the inliers (the kernel's serial tail, needed by nothing downstream) is left to
// ransac_rmse_kernel on another stream; gm, mask and count are final when this kernel ends either way.
template <uint32_t T, uint32_t PER, uint32_t F, bool DEFER>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 8))) void ransac_reg_kernel(RansacArgs a, uint32_t n_frames) {
  constexpr uint32_t G = T / F, GW = G / 64;  // lanes and waves per frame
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];  // F staging areas of kChunk float2
  __shared__ float s_model[F][kPassModels][2];
  __shared__ uint32_t s_cnt[2][F][GW][kPassModels];
  __shared__ int s_isum[F][2];
  __shared__ uint32_t s_flag[F], s_mag[F], s_bestn[F], s_bestit[F], s_serial;
  __shared__ float s_gm[F][2], s_out[F][2];
  const uint32_t tid = threadIdx.x, g = tid / G, gt = tid - g * G, lane = tid & 63u, gwave = gt >> 6;
  const uint32_t frame_raw = blockIdx.x * F + g;
  const bool live = frame_raw < n_frames;
  const uint32_t frame = live ? frame_raw : 0u;  // a dead group recomputes frame 0 and writes nothing
  const float2* mv = reinterpret_cast<const float2*>(a.mv) + (size_t)frame * a.blocks;
  const uint32_t* samples = a.samples + (size_t)frame * a.iters * a.subset;
  uint8_t* mask = a.mask + (size_t)frame * a.blocks;
  float2* s_stage = reinterpret_cast<float2*>(dyn_lds) + (size_t)g * kChunk;
  const float t2 = a.thresh * a.thresh;

  float2 m[PER];
#pragma unroll
  for (uint32_t u = 0; u < PER; ++u) {
    const uint32_t i = u * G + gt;
    m[u] = i < a.blocks ? mv[i] : make_float2(__builtin_inff(), __builtin_inff());
  }

  // ---- iterations (motion.cpp:210-238), kPassModels at a time ---------------------------------
  uint32_t best_n = 0, best_it = 0;  // tracked by the group's first lane, published below
  float bgx = 0.f, bgy = 0.f;
  uint32_t par = 0;
  for (uint32_t it0 = 0; it0 < a.iters; it0 += kPassModels, par ^= 1u) {
    const uint32_t nm = min(kPassModels, a.iters - it0);
    if (gt < nm) {
      float sx = 0.f, sy = 0.f;  // sequential f32 sum of the subset (motion.cpp:156-160)
      for (uint32_t i = 0; i < a.subset; ++i) {
        const float2 s = mv[min(samples[(size_t)(it0 + gt) * a.subset + i], a.blocks - 1)];  // a draw past the field (the reference's [0, N], motion.cpp:208) never leaves it: svc_hip.h
        sx = sx + s.x;
        sy = sy + s.y;
      }
      const float inv = 1.0f / (float)a.subset;
      s_model[g][gt][0] = sx * inv;
      s_model[g][gt][1] = sy * inv;
    }
    __syncthreads();
    float gx[kPassModels], gy[kPassModels];
#pragma unroll
    for (uint32_t k = 0; k < kPassModels; ++k) {
      gx[k] = s_model[g][k < nm ? k : 0][0];
      gy[k] = s_model[g][k < nm ? k : 0][1];
    }
#pragma unroll
    for (uint32_t k = 0; k < kPassModels; ++k) {
      if (k >= nm) break;  // block-uniform
      uint32_t c = 0;
#pragma unroll
      for (uint32_t u = 0; u < PER; ++u) c += is_inlier(gx[k], gy[k], m[u].x, m[u].y, t2) ? 1u : 0u;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
      if (lane == 0) s_cnt[par][g][gwave][k] = c;
    }
    __syncthreads();
    if (gt == 0) {
      for (uint32_t k = 0; k < nm; ++k) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < GW; ++w) total += s_cnt[par][g][w][k];
        if (total >= best_n) {  // motion.cpp:233, in iteration order: ties -> later
          best_n = total;
          best_it = it0 + k;
          bgx = gx[k];
          bgy = gy[k];
        }
      }
    }
    // s_model is rewritten by the next pass only after every lane has read it (the reads sit before
    // the barrier above); s_cnt alternates between two copies
  }
  if (gt == 0) {
    s_bestn[g] = best_n; s_bestit[g] = best_it; s_gm[g][0] = bgx; s_gm[g][1] = bgy;
    s_flag[g] = 1u; s_isum[g][0] = 0; s_isum[g][1] = 0; s_mag[g] = 0;
  }
  if (tid == 0) s_serial = 0;
  __syncthreads();
  best_n = s_bestn[g]; best_it = s_bestit[g]; bgx = s_gm[g][0]; bgy = s_gm[g][1];
  const bool any_iter = a.iters > 0;
  const bool few = best_n < a.subset;  // motion.cpp:240-242: keep the best subset's model, RMSE against the incoming one

  // ---- inlier mask (== best_inliers, motion.cpp:244-253) and the final model = mean of the
  // inliers (motion.cpp:255-256); integer fast path as in ransac_kernel
  uint64_t inl = 0;  // bit u: block u * G + gt is an inlier
  {
    bool ok = true;
    int ix = 0, iy = 0;
    uint32_t mag = 0;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
      const uint32_t i = u * G + gt;
      const bool in = (!few || any_iter) && is_inlier(bgx, bgy, m[u].x, m[u].y, t2);
      if (live && i < a.blocks) mask[i] = in ? 1 : 0;
      if (!in) continue;
      inl |= 1ull << u;
      ok = ok && m[u].x == truncf(m[u].x) && m[u].y == truncf(m[u].y) && fabsf(m[u].x) <= 32768.f && fabsf(m[u].y) <= 32768.f;
      if (ok) {
        ix += (int)m[u].x; iy += (int)m[u].y;
        mag += (uint32_t)fabsf(m[u].x) + (uint32_t)fabsf(m[u].y);
      }
    }
    if (!ok || mag >= (1u << 24)) s_flag[g] = 0u;
    mag &= 0xFFFFFFu;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      ix += __shfl_xor(ix, off, 64);
      iy += __shfl_xor(iy, off, 64);
      mag += __shfl_xor(mag, off, 64);
    }
    if (lane == 0) {
      if (mag >= (1u << 24)) { s_flag[g] = 0u; mag = 0; }  // a wave's 64 addends fit; GW clamped ones cannot wrap s_mag
      atomicAdd(&s_isum[g][0], ix);
      atomicAdd(&s_isum[g][1], iy);
      atomicAdd(&s_mag[g], mag);
    }
  }
  __syncthreads();
  const bool exact_int = s_flag[g] != 0u && s_mag[g] < (1u << 24);
  if (gt == 0 && !few && !exact_int) s_serial = 1;  // some frame of this workgroup needs the in-order walk
  __syncthreads();
  float sx = 0.f, sy = 0.f;
  if (exact_int) {
    sx = (float)s_isum[g][0];
    sy = (float)s_isum[g][1];
  }
  if (s_serial) {  // workgroup-uniform
    for (uint32_t base = 0; base < a.blocks; base += kChunk) {
      const uint32_t n = min(kChunk, a.blocks - base);
#pragma unroll
      for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = u * G + gt;
        if (i >= base && i < base + n) s_stage[i - base] = ((inl >> u) & 1ull) ? m[u] : make_float2(0.f, 0.f);
      }
      __syncthreads();
      if (gt == 0 && !exact_int) serial_sum2(s_stage, n, sx, sy);
      __syncthreads();
    }
    if (gt == 0 && !exact_int) { s_gm[g][0] = sx; s_gm[g][1] = sy; }
    __syncthreads();
    if (!exact_int) { sx = s_gm[g][0]; sy = s_gm[g][1]; }
  }
  const float inv = 1.0f / (float)best_n;
  const float out_gx = few ? bgx : sx * inv, out_gy = few ? bgy : sy * inv;
  if (gt == 0) { s_out[g][0] = out_gx; s_out[g][1] = out_gy; }  // read after the barriers of the RMSE loop

  // ---- RMSE (motion.cpp:258-259, :165-180): terms in parallel, the sums in order -- lane f of the
  // first wave walks frame f's terms, all F chains in the same instructions
  float acc = 0.f;
  if (DEFER) __syncthreads();  // s_out is read below
  for (uint32_t base = 0; !DEFER && base < a.blocks; base += 2 * kChunk) {
    const uint32_t n = min(2 * kChunk, a.blocks - base);
    float* terms = reinterpret_cast<float*>(s_stage);
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
      const uint32_t i = u * G + gt;
      if (i >= base && i < base + n) {
        const float dx = m[u].x - out_gx, dy = m[u].y - out_gy;
        terms[i - base] = ((inl >> u) & 1ull) ? dx * dx + dy * dy : 0.f;
      }
    }
    __syncthreads();
    if (tid < F) serial_sum1(reinterpret_cast<const float*>(reinterpret_cast<const float2*>(dyn_lds) + (size_t)tid * kChunk), n, acc);
    __syncthreads();
  }
  if (tid < F && blockIdx.x * F + tid < n_frames) {
    const uint32_t f = blockIdx.x * F + tid;  // lane f of the first wave finishes frame f of the workgroup
    const uint32_t bn = s_bestn[tid];
    const bool fw = bn < a.subset;
    float r;
    if (fw) {  // motion.cpp:240-242: RMSE of the best subset against the INCOMING global motion
      const float2* fmv = reinterpret_cast<const float2*>(a.mv) + (size_t)f * a.blocks;
      const uint32_t* fs = a.samples + (size_t)f * a.iters * a.subset;
      const float ix = a.gm[2 * f], iy = a.gm[2 * f + 1];
      float e = 0.f;
      for (uint32_t i = 0; i < a.subset; ++i) {
        const float2 s = fmv[min(fs[(size_t)s_bestit[tid] * a.subset + i], a.blocks - 1)];
        const float dx = s.x - ix, dy = s.y - iy;
        e += dx * dx + dy * dy;
      }
      r = sqrtf(e / (float)a.subset);
    } else {
      r = sqrtf(acc / (float)bn);
    }
    a.gm[2 * f] = s_out[tid][0];  // published by the frame's own group
    a.gm[2 * f + 1] = s_out[tid][1];
    if (!DEFER || fw) a.rmse[f] = r;
    a.count[f] = bn;
  }
}

// The RMSE of frames whose model is the mean of their inliers (motion.cpp:258-259, :165-180), from what the RANSAC
// kernel left: final gm, inlier mask, inlier count.  Terms in parallel into LDS, the f32 sum strictly in index order by
// one lane per frame (non-inliers were staged as +0.0f, which leaves the sum unchanged) -- the same arithmetic as the tail
// of ransac_reg_kernel, so the same bits.  Frames that kept the best subset's model (count < subset, :240-242) already
// have their RMSE and are left alone.  F frames per workgroup: their chains run as lanes of one wave (see above).
template <uint32_t F>
__global__ __launch_bounds__(256) void ransac_rmse_kernel(RansacArgs a, uint32_t n_frames) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];  // F staging areas of 2 * kChunk floats
  const uint32_t tid = threadIdx.x;
  float acc = 0.f;
  for (uint32_t base = 0; base < a.blocks; base += 2 * kChunk) {
    const uint32_t n = min(2 * kChunk, a.blocks - base);
    for (uint32_t g = 0; g < F; ++g) {
      const uint32_t f = min(blockIdx.x * F + g, n_frames - 1);
      const float2* mv = reinterpret_cast<const float2*>(a.mv) + (size_t)f * a.blocks + base;
      const uint8_t* mask = a.mask + (size_t)f * a.blocks + base;
      const float gx = a.gm[2 * f], gy = a.gm[2 * f + 1];
      float* terms = reinterpret_cast<float*>(dyn_lds) + (size_t)g * 2 * kChunk;
      for (uint32_t i = tid; i < n; i += 256) {
        const float2 m = mv[i];
        const float dx = m.x - gx, dy = m.y - gy;
        terms[i] = mask[i] ? dx * dx + dy * dy : 0.f;
      }
    }
    __syncthreads();
    if (tid < F) serial_sum1(reinterpret_cast<const float*>(dyn_lds) + (size_t)tid * 2 * kChunk, n, acc);
    __syncthreads();
  }
  if (tid < F && blockIdx.x * F + tid < n_frames) {
    const uint32_t f = blockIdx.x * F + tid, bn = a.count[f];
    if (bn >= a.subset) a.rmse[f] = sqrtf(acc / (float)bn);
  }
}

// libs/encoder.cpp:507-513 + :549-551 (the in-repo part of the segmentation glue):
// the foreground mask is the complement of the RANSAC inliers and every block starts
// as BLOCK_TYPE_BACKGROUND (0, libs/codec.hpp:6).  Until the OpenCV-side clustering
//  exists, all foreground blocks form one region with id 1.
__global__ __launch_bounds__(256) void block_types_kernel(const uint8_t* mask, uint32_t* types, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    types[i] = mask[i] ? 0u : 1u;
}

int launch_block_types(const uint8_t* d_mask, uint64_t n, uint32_t* d_types, hipStream_t stream) {
  if (n == 0) return SVC_OK;
  const uint64_t want = (n + 255) / 256;
  hipLaunchKernelGGL(block_types_kernel, dim3((uint32_t)(want < 2048 ? want : 2048)), dim3(256), 0, stream,
                     d_mask, d_types, n);
  return check_launch("block_types_kernel");
}

int launch_ransac(const float* d_mv, uint32_t blocks, uint32_t n_frames, svc_ransac_params params,
                  const uint32_t* d_samples, uint32_t iters, float* d_gm, float* d_rmse,
                  uint8_t* d_mask, uint32_t* d_count, uint32_t flags, hipStream_t stream) {
  if (n_frames == 0) return SVC_OK;
  RansacArgs a;
  a.mv = d_mv;
  a.samples = d_samples;
  a.blocks = blocks;
  a.iters = iters;
  a.subset = params.subset_sz;
  a.thresh = params.inlier_thresh;
  a.gm = d_gm;
  a.rmse = d_rmse;
  a.mask = d_mask;
  a.count = d_count;
  constexpr size_t kStage = kChunk * sizeof(float2);
  // SVC_LAUNCH_BESIDE: 256 lanes x 32 blocks in registers -- one wave per SIMD, which fits next to the transform
  // kernel's waves; a 1 024-lane workgroup of this kernel (4 waves x 114-128 VGPRs on every SIMD) needs an empty CU and
  // would only start once the kernel it was meant to run beside has drained.  Alone it is slower (0.10 vs 0.08 ms for
  // 300 frames at 1080p: a quarter of the lanes for the parallel phases, the same serial RMSE chain).
  // SVC_LAUNCH_DEFER_RMSE: the caller runs launch_ransac_rmse() itself (on another stream); only the register kernels
  // have the form without the tail, the fallback for fields above 32 768 blocks computes the RMSE regardless
  const bool defer = (flags & SVC_LAUNCH_DEFER_RMSE) != 0;
#define SVC_RANSAC_LAUNCH(T, PER, F, GRID, LDS)                                                                   \
  do {                                                                                                            \
    if (defer) hipLaunchKernelGGL((ransac_reg_kernel<T, PER, F, true>), dim3(GRID), dim3(T), LDS, stream, a, n_frames);  \
    else hipLaunchKernelGGL((ransac_reg_kernel<T, PER, F, false>), dim3(GRID), dim3(T), LDS, stream, a, n_frames);       \
  } while (0)
  if ((flags & SVC_LAUNCH_BESIDE) && blocks > 8 * 256 && blocks <= 32 * 256)
    SVC_RANSAC_LAUNCH(256, 32, 1, n_frames, kStage);
  else if (blocks <= 8 * 256)
    SVC_RANSAC_LAUNCH(256, 8, 1, n_frames, kStage);
  else if (blocks <= 8 * 1024 && n_frames <= 256)  // one workgroup per CU as it is
    SVC_RANSAC_LAUNCH(1024, 8, 1, n_frames, kStage);
  else if (blocks <= 16 * 512)  // more frames than CUs: two frames per workgroup rather than two workgroups per CU
    SVC_RANSAC_LAUNCH(1024, 16, 2, (n_frames + 1) / 2, 2 * kStage);
  else if (blocks <= 32 * 1024)
    SVC_RANSAC_LAUNCH(1024, 32, 1, n_frames, kStage);
  else
    hipLaunchKernelGGL(ransac_kernel, dim3(n_frames), dim3(256), 0, stream, a);
#undef SVC_RANSAC_LAUNCH
  return check_launch("ransac_kernel");
}

int launch_ransac_rmse(const float* d_mv, uint32_t blocks, uint32_t n_frames, svc_ransac_params params, const float* d_gm,
                       const uint8_t* d_mask, const uint32_t* d_count, float* d_rmse, hipStream_t stream) {
  if (n_frames == 0) return SVC_OK;
  RansacArgs a;
  a.mv = d_mv;
  a.samples = nullptr;
  a.blocks = blocks;
  a.iters = 0;
  a.subset = params.subset_sz;
  a.thresh = params.inlier_thresh;
  a.gm = const_cast<float*>(d_gm);
  a.rmse = d_rmse;
  a.mask = const_cast<uint8_t*>(d_mask);
  a.count = const_cast<uint32_t*>(d_count);
  constexpr size_t kTerms = 2 * kChunk * sizeof(float);
  // more frames than CUs: two chains per workgroup, as lanes of one wave, rather than two workgroups per CU
  if (n_frames > 256) hipLaunchKernelGGL((ransac_rmse_kernel<2>), dim3((n_frames + 1) / 2), dim3(256), 2 * kTerms, stream, a, n_frames);
  else hipLaunchKernelGGL((ransac_rmse_kernel<1>), dim3(n_frames), dim3(256), kTerms, stream, a, n_frames);
  return check_launch("ransac_rmse_kernel");
}

}  // namespace svc
