// ubench_stream_oneshot.hip -- read / write mixes with ONE unit of work per short-lived workgroup, in XCD-contiguous order (what
// tools/ubench_fill.hip found fastest for stores), against workgroups that loop over their slice (the bench line's svc_hip_probe_stream):
// the ceiling of the luma + pyramid kernel (3 bytes read per 1.3 written) and of the transform (1 read : 4 written).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench_stream_oneshot.hip -o /tmp/ubench_oneshot && /tmp/ubench_oneshot
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t xcd_contiguous_block(uint32_t bid, uint32_t nblocks) {
  const uint32_t q = nblocks >> 3, r = nblocks & 7u, xcd = bid & 7u, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// a workgroup's unit: 256 lanes x R adjacent dwordx4 loads (a lane's loads are adjacent: the BGR pattern for R = 3) and W dwordx4 stores
// (store k of all lanes contiguous: 4 KiB pieces).  LOOP: 4 096 workgroups walk the units; else one unit per workgroup.
template <int R, int W, bool LOOP>
__global__ __launch_bounds__(256) void stream(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t units, uint32_t magic) {
  uint4 acc = make_uint4(magic, 0, 0, 0);
  for (uint32_t u0 = blockIdx.x; u0 < units; u0 += gridDim.x) {
    const size_t u = LOOP ? u0 : xcd_contiguous_block(u0, units);
    const size_t i = u * 256 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint4 v = in[i * R + r];
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) out[(u * W + w) * 256 + threadIdx.x] = acc;
    if (!LOOP) break;
  }
  if (W == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;
}

// the 3 : 1 unit with WORK dependent-free integer dot products per lane and an LDS round trip per 16 of them between its loads and its
// store: does the chip stream slower when VALU and LDS are busy beside the memory pipeline (the luma kernel: ~620 vector instructions per wave)?
template <int WORK>
__global__ __launch_bounds__(256) void stream31_with_work(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t units) {
  __shared__ uint32_t lds[256 * 4];
  const size_t u = xcd_contiguous_block(blockIdx.x, units);
  const size_t i = u * 256 + threadIdx.x;
  uint4 v0 = in[i * 3], v1 = in[i * 3 + 1], v2 = in[i * 3 + 2];
  uint32_t a0 = v0.x ^ v1.y, a1 = v0.y ^ v2.z, a2 = v1.x ^ v2.w, a3 = v0.z ^ v1.w;
#pragma unroll 16
  for (int k = 0; k < WORK; k += 4) {
    a0 = __builtin_amdgcn_udot4(v0.x + k, v1.x, a0, false);
    a1 = __builtin_amdgcn_udot4(v0.y, v1.y + k, a1, false);
    a2 = __builtin_amdgcn_udot4(v0.z + k, v2.x, a2, false);
    a3 = __builtin_amdgcn_udot4(v0.w, v2.y + k, a3, false);
    if ((k & 63) == 60) {
      lds[threadIdx.x * 4 + (k >> 6 & 3)] = a0 ^ a2;
      a1 ^= lds[(threadIdx.x ^ 1) * 4 + (k >> 6 & 3)];
    }
  }
  out[u * 256 + threadIdx.x] = make_uint4(a0, a1, a2, a3);
}

template <int WORK>
static void run_work(const uint4* in, uint4* out, size_t bytes) {
  const uint32_t units = (uint32_t)(bytes / 16 / 256 / 3);
  const double moved = (double)units * 256 * 16 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto launch = [&] { hipLaunchKernelGGL((stream31_with_work<WORK>), dim3(units), dim3(256), 0, 0, in, out, units); };
  launch(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("3 read : 1 written + %4d dot products per lane (+ LDS round trips)            %7.3f ms  %5.2f TB/s\n", WORK, ms, moved / ms / 1e9);
}

template <int R, int W>
static void run(const char* name, const uint4* in, uint4* out, size_t bytes) {
  const uint32_t units = (uint32_t)(bytes / 16 / 256 / (R > W ? R : W));
  const double moved = (double)units * 256 * 16 * (R + W);
  for (int loop = 0; loop < 2; ++loop) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto launch = [&] {
      if (loop) hipLaunchKernelGGL((stream<R, W, true>), dim3(4096), dim3(256), 0, 0, in, out, units, 1u);
      else hipLaunchKernelGGL((stream<R, W, false>), dim3(units), dim3(256), 0, 0, in, out, units, 1u);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-34s %-44s %7.3f ms  %5.2f TB/s\n", name, loop ? "4 096 workgroups looping over the units" : "one unit per workgroup, XCD-contiguous", ms, moved / ms / 1e9);
  }
}

int main() {
  const size_t bytes = 6ull << 30;
  uint4 *in, *out;
  if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
  run<1, 0>("read only (16 B per lane)", in, out, bytes);
  run<3, 0>("read only (3 x 16 B per lane)", in, out, bytes);
  run<0, 1>("write only", in, out, bytes);
  run<1, 1>("copy 1 : 1", in, out, bytes);
  run<3, 1>("3 read : 1 written (luma)", in, out, bytes);
  run<1, 4>("1 read : 4 written (transform)", in, out, bytes);
  run_work<0>(in, out, bytes);
  run_work<128>(in, out, bytes);
  run_work<512>(in, out, bytes);
  run_work<1024>(in, out, bytes);
  run_work<2048>(in, out, bytes);
  return 0;
}
