// ubench_fill.hip -- what a store-only kernel reaches on this GPU, by store shape: the ceiling of the transform kernel (80 % of its
// bytes are stores; csrc/dct.hip writes 512 contiguous bytes per wave instruction, 2 KiB row pieces per workgroup, rows 7 680 bytes apart).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench_fill.hip -o /tmp/ubench_fill && /tmp/ubench_fill
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: dwordx4 per lane, a wave instruction = 1 KiB contiguous; 1: the same, nontemporal; 2: dwordx2 per lane (512 B per instruction);
// 3: dword per lane (256 B per instruction)
template <int MODE>
__global__ __launch_bounds__(256) void fill_flat(float* out, size_t n_floats, float v) {
  constexpr int PER = MODE <= 1 ? 4 : MODE == 2 ? 2 : 1;
  const size_t n = n_floats / PER;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (MODE == 0) reinterpret_cast<f32x4*>(out)[i] = f32x4{v, v, v, v};
    if (MODE == 1) __builtin_nontemporal_store(f32x4{v, v, v, v}, reinterpret_cast<f32x4*>(out) + i);
    if (MODE == 2) reinterpret_cast<f32x2*>(out)[i] = f32x2{v, v};
    if (MODE == 3) out[i] = v;
  }
}

// the transform kernel's store shape: a workgroup owns 32 segment columns (512 pixels) of an 8-row band; per channel and row v its
// 256 lanes write float2 each = 2 KiB contiguous; rows are w floats apart; planes w * h floats apart.  ROWS_FIRST: all 8 rows of a
// channel back to back (as the kernel does); else one instruction per (channel, row) in the same order.
template <bool NT>
__global__ __launch_bounds__(256) void fill_tiles(float* out, uint32_t w, uint32_t h, uint32_t frames, float v) {
  const uint32_t segs = w / 16, bands = h / 8, per_frame = segs * bands / 32;
  const uint32_t total = per_frame * frames;
  for (uint32_t g = blockIdx.x; g < total; g += gridDim.x) {
    const uint32_t frame = g / per_frame, r = g - frame * per_frame;
    const uint32_t gsc = r * 32 + threadIdx.x / 8, band = gsc / segs, seg = gsc - band * segs;
    float* base = out + (size_t)frame * 3 * w * h + (size_t)band * 8 * w + seg * 16 + 2 * (threadIdx.x % 8);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        f32x2* p = reinterpret_cast<f32x2*>(base + (size_t)c * w * h + (size_t)y * w);
        if (NT) __builtin_nontemporal_store(f32x2{v, v}, p);
        else *p = f32x2{v, v};
      }
  }
}

// UNROLL consecutive KiB per wave per iteration, all stores issued back to back (more stores in flight per wave)
template <int UNROLL>
__global__ __launch_bounds__(256) void fill_unrolled(float* out, size_t n_floats, float v) {
  const size_t n = n_floats / 4 / UNROLL;  // groups of UNROLL x 16 B per lane
  f32x4* o = reinterpret_cast<f32x4*>(out);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t wg = i / 256, lane = i % 256;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) o[(wg * UNROLL + u) * 256 + lane] = f32x4{v, v, v, v};
  }
}

// write-through stores (sc0 sc1): the line does not stay in the L2
__global__ __launch_bounds__(256) void fill_sc1(float* out, size_t n_floats, float v) {
  const size_t n = n_floats / 4;
  const f32x4 val = {v, v, v, v};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    f32x4* p = reinterpret_cast<f32x4*>(out) + i;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(val) : "memory");
  }
}

__device__ __forceinline__ uint32_t xcd_contiguous_block(uint32_t bid, uint32_t nblocks) {
  const uint32_t q = nblocks >> 3, r = nblocks & 7u, xcd = bid & 7u, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// one 4 KiB chunk per workgroup, no loop; XCDC: chunk = xcd_contiguous_block(blockIdx) (each XCD fills one contiguous eighth)
template <bool XCDC>
__global__ __launch_bounds__(256) void fill_oneshot(float* out, float v) {
  const size_t b = XCDC ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
  reinterpret_cast<f32x4*>(out)[b * 256 + threadIdx.x] = f32x4{v, v, v, v};
}

// the transform kernel's shape, one workgroup per unit, by mapping and by workgroup size:
//   ORDER 0: linear (unit = blockIdx), 1: XCD-contiguous (as csrc/dct.hip), 2: XCD-contiguous inside each FRAME (frames in order)
//   LANES 256: 32 segment columns (2 KiB row pieces); 1024: 128 segment columns (a whole 1080p band row and a bit: 8 KiB pieces)
//   PLANE_MAJOR: all 8 rows of a plane back to back (as the kernel); else row by row, the three planes of a row back to back
template <int ORDER, int LANES, bool PLANE_MAJOR>
__global__ __launch_bounds__(LANES) void fill_tiles2(float* out, uint32_t w, uint32_t h, uint32_t frames, float v) {
  constexpr uint32_t kSeg = LANES / 8;
  const uint32_t segs = w / 16, bands = h / 8, segcols = segs * bands, per_frame = (segcols + kSeg - 1) / kSeg;
  uint32_t g = blockIdx.x;
  if (ORDER == 1) g = xcd_contiguous_block(blockIdx.x, gridDim.x);
  if (ORDER == 2) { const uint32_t f = blockIdx.x / per_frame; g = f * per_frame + xcd_contiguous_block(blockIdx.x - f * per_frame, per_frame); }
  const uint32_t frame = g / per_frame, r = g - frame * per_frame;
  const uint32_t gsc = r * kSeg + threadIdx.x / 8;
  if (frame >= frames || gsc >= segcols) return;
  const uint32_t band = gsc / segs, seg = gsc - band * segs;
  float* base = out + (size_t)frame * 3 * w * h + (size_t)band * 8 * w + seg * 16 + 2 * (threadIdx.x % 8);
  if (PLANE_MAJOR) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int y = 0; y < 8; ++y) *reinterpret_cast<f32x2*>(base + (size_t)c * w * h + (size_t)y * w) = f32x2{v, v};
  } else {
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int c = 0; c < 3; ++c) *reinterpret_cast<f32x2*>(base + (size_t)c * w * h + (size_t)y * w) = f32x2{v, v};
  }
}

// one whole band (8 rows x the frame's width) per workgroup: 8 * w * 4 contiguous bytes per plane (rows of a band are adjacent in memory)
template <bool XCDC, int SPLIT>
__global__ __launch_bounds__(1024) void fill_bands(float* out, uint32_t w, uint32_t h, uint32_t frames, float v) {
  const uint32_t bands = h / 8, total = bands * frames * SPLIT;
  const uint32_t u = XCDC ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
  if (u >= total) return;
  const uint32_t g = u / SPLIT, part = u % SPLIT;  // SPLIT workgroups share a band: each takes w / SPLIT columns
  const uint32_t frame = g / bands, band = g - frame * bands;
  const uint32_t lane_px = 2 * threadIdx.x + part * (w / SPLIT);  // float2 per lane: w / 2 / SPLIT lanes per row
  if (2 * threadIdx.x >= w / SPLIT) return;
  float* base = out + (size_t)frame * 3 * w * h + (size_t)band * 8 * w + lane_px;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int y = 0; y < 8; ++y) *reinterpret_cast<f32x2*>(base + (size_t)c * w * h + (size_t)y * w) = f32x2{v, v};
}

// factors: STREAMS regions (n / STREAMS floats apart) x PER consecutive 4 KiB (X4) or 2 KiB (!X4) chunks per workgroup, one store each
template <int STREAMS, int PER, bool X4>
__global__ __launch_bounds__(256) void fill_factors(float* out, size_t n_floats, float v) {
  const size_t u = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const size_t region = n_floats / STREAMS;
#pragma unroll
  for (int s = 0; s < STREAMS; ++s)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (X4) reinterpret_cast<f32x4*>(out + s * region)[(u * PER + k) * 256 + threadIdx.x] = f32x4{v, v, v, v};
      else reinterpret_cast<f32x2*>(out + s * region)[(u * PER + k) * 256 + threadIdx.x] = f32x2{v, v};
    }
}

template <typename F>
static void timeit(const char* name, double bytes, F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("%-64s %7.3f ms  %5.2f TB/s\n", name, ms, bytes / ms / 1e9);
}

int main() {
  const uint32_t w = 1920, h = 1088, frames = 299;
  const size_t n = (size_t)w * h * 3 * frames;  // 7.5 GB of f32: the transform's output at C3
  float* out;
  if (hipMalloc(&out, n * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(out, 0, n * 4);
  const double bytes = (double)n * 4;
  for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
    char nm[128];
    snprintf(nm, sizeof nm, "flat dwordx4 (1 KiB per wave instruction), grid %d", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL(fill_flat<0>, dim3(grid), dim3(256), 0, 0, out, n, 1.f); });
  }
  for (int grid : {256 * 256, 256 * 1024, (int)(n / 4 / 256)}) {
    char nm[128];
    snprintf(nm, sizeof nm, "flat dwordx4, grid %d", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL(fill_flat<0>, dim3(grid), dim3(256), 0, 0, out, n, 1.f); });
  }
  for (int grid : {2048, 16384}) {
    char nm[128];
    snprintf(nm, sizeof nm, "flat dwordx4 x 4 per iteration (4 KiB per wave), grid %d", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL(fill_unrolled<4>, dim3(grid), dim3(256), 0, 0, out, n, 1.f); });
    snprintf(nm, sizeof nm, "flat dwordx4 x 8 per iteration, grid %d", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL(fill_unrolled<8>, dim3(grid), dim3(256), 0, 0, out, n, 1.f); });
    snprintf(nm, sizeof nm, "flat dwordx4 write-through (sc0 sc1), grid %d", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL(fill_sc1, dim3(grid), dim3(256), 0, 0, out, n, 1.f); });
  }
  timeit("flat dwordx4 nontemporal, grid 4096", bytes, [&] { hipLaunchKernelGGL(fill_flat<1>, dim3(4096), dim3(256), 0, 0, out, n, 1.f); });
  timeit("flat dwordx2 (512 B per wave instruction), grid 4096", bytes, [&] { hipLaunchKernelGGL(fill_flat<2>, dim3(4096), dim3(256), 0, 0, out, n, 1.f); });
  timeit("flat dword (256 B per wave instruction), grid 4096", bytes, [&] { hipLaunchKernelGGL(fill_flat<3>, dim3(4096), dim3(256), 0, 0, out, n, 1.f); });
  for (int grid : {256 * 4, 256 * 8, 256 * 16}) {
    char nm[128];
    snprintf(nm, sizeof nm, "transform-shaped (2 KiB row pieces, 3 planes x 8 rows), grid %d", grid);
    timeit(nm, bytes, [&] { hipLaunchKernelGGL(fill_tiles<false>, dim3(grid), dim3(256), 0, 0, out, w, h, frames, 1.f); });
  }
  timeit("transform-shaped, one workgroup per tile group (as the kernel)", bytes,
         [&] { hipLaunchKernelGGL(fill_tiles<false>, dim3(w / 16 * (h / 8) / 32 * frames), dim3(256), 0, 0, out, w, h, frames, 1.f); });
  {
    const uint32_t chunks = (uint32_t)(n / 4 / 256);
    timeit("one 4 KiB chunk per workgroup, linear", bytes, [&] { hipLaunchKernelGGL(fill_oneshot<false>, dim3(chunks), dim3(256), 0, 0, out, 1.f); });
    timeit("one 4 KiB chunk per workgroup, XCD-contiguous", bytes, [&] { hipLaunchKernelGGL(fill_oneshot<true>, dim3(chunks), dim3(256), 0, 0, out, 1.f); });
    const uint32_t segcols = w / 16 * (h / 8);
    const uint32_t g256 = (segcols + 31) / 32 * frames, g1024 = (segcols + 127) / 128 * frames;
#define TILE_VARIANT(ORDER, LANES, PM, grid, label) \
    timeit(label, bytes, [&] { hipLaunchKernelGGL((fill_tiles2<ORDER, LANES, PM>), dim3(grid), dim3(LANES), 0, 0, out, w, h, frames, 1.f); })
    TILE_VARIANT(0, 256, true, g256, "tiles 256 lanes, linear, plane-major");
    TILE_VARIANT(1, 256, true, g256, "tiles 256 lanes, XCD-contiguous over the clip (the kernel), plane-major");
    TILE_VARIANT(2, 256, true, g256, "tiles 256 lanes, XCD-contiguous inside a frame, plane-major");
    TILE_VARIANT(0, 256, false, g256, "tiles 256 lanes, linear, row-major");
    TILE_VARIANT(1, 256, false, g256, "tiles 256 lanes, XCD-contiguous over the clip, row-major");
    TILE_VARIANT(0, 1024, true, g1024, "tiles 1024 lanes, linear, plane-major");
    TILE_VARIANT(1, 1024, true, g1024, "tiles 1024 lanes, XCD-contiguous over the clip, plane-major");
    TILE_VARIANT(2, 1024, true, g1024, "tiles 1024 lanes, XCD-contiguous inside a frame, plane-major");
    TILE_VARIANT(0, 1024, false, g1024, "tiles 1024 lanes, linear, row-major");
  }
  {
    const uint32_t nb = h / 8 * frames;
    timeit("a whole band per workgroup (960 lanes, 7 680 B per store instruction), linear", bytes,
           [&] { hipLaunchKernelGGL((fill_bands<false, 1>), dim3(nb), dim3(960), 0, 0, out, w, h, frames, 1.f); });
    timeit("a whole band per workgroup, XCD-contiguous", bytes,
           [&] { hipLaunchKernelGGL((fill_bands<true, 1>), dim3(nb), dim3(960), 0, 0, out, w, h, frames, 1.f); });
    timeit("half a band per workgroup (480 lanes), linear", bytes,
           [&] { hipLaunchKernelGGL((fill_bands<false, 2>), dim3(nb * 2), dim3(480), 0, 0, out, w, h, frames, 1.f); });
    timeit("half a band per workgroup (480 lanes), XCD-contiguous", bytes,
           [&] { hipLaunchKernelGGL((fill_bands<true, 2>), dim3(nb * 2), dim3(480), 0, 0, out, w, h, frames, 1.f); });
  }
  {
#define FACTOR(ST, PER, X4, label) \
    timeit(label, bytes, [&] { hipLaunchKernelGGL((fill_factors<ST, PER, X4>), dim3((uint32_t)(n / ST / PER / (X4 ? 4 : 2) / 256)), dim3(256), 0, 0, out, n, 1.f); })
    FACTOR(1, 1, true, "factors: 1 stream x 1 chunk x dwordx4 (= one 4 KiB chunk per workgroup)");
    FACTOR(1, 1, false, "factors: 1 stream x 1 chunk x dwordx2 (2 KiB per workgroup)");
    FACTOR(1, 8, true, "factors: 1 stream x 8 consecutive chunks x dwordx4 (32 KiB per workgroup)");
    FACTOR(1, 8, false, "factors: 1 stream x 8 consecutive chunks x dwordx2");
    FACTOR(1, 24, false, "factors: 1 stream x 24 consecutive chunks x dwordx2 (48 KiB per workgroup)");
    FACTOR(3, 1, true, "factors: 3 streams x 1 chunk x dwordx4");
    FACTOR(3, 8, false, "factors: 3 streams x 8 consecutive chunks x dwordx2 (the kernel's count, contiguous per plane)");
    FACTOR(3, 8, true, "factors: 3 streams x 8 consecutive chunks x dwordx4");
  }
  timeit("transform-shaped nontemporal, grid 4096", bytes, [&] { hipLaunchKernelGGL(fill_tiles<true>, dim3(4096), dim3(256), 0, 0, out, w, h, frames, 1.f); });
  timeit("hipMemsetD32Async", bytes, [&] { hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(out), 0x3f800000, n, 0); });
  hipFree(out);
  return 0;
}
