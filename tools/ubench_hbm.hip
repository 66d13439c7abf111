// ubench_hbm.hip -- what HBM gives a plain streaming kernel on this GPU: read-only, write-only, and mixes.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o /tmp/ubench_hbm && /tmp/ubench_hbm
// Puts the per-kernel roofline fractions in context (MI355X_MICROARCH.md quotes ~8 TB/s peak).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// each lane: R dwordx4 loads and W dwordx4 stores per iteration, contiguous across the workgroup
template <int R, int W>
__global__ __launch_bounds__(256) void stream(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16, uint32_t magic) {
  // a workgroup walks its own contiguous slice, as the library's svc_hip_probe_stream does
  const size_t iters = n16 / (R > W ? R : W), per_wg = (iters + gridDim.x - 1) / gridDim.x;
  const size_t i0 = (size_t)blockIdx.x * per_wg, i1 = i0 + per_wg < iters ? i0 + per_wg : iters;
  uint4 acc = make_uint4(magic, 0, 0, 0);
  for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint4 v = in[i * R + r];  // a lane's R loads are adjacent; lanes are R * 16 B apart (the BGR pattern for R = 3)
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) out[i * W + w] = acc;
  }
  if (W == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;
}

template <int R, int W>
static void run(const char* name, const uint4* in, uint4* out, size_t bytes) {
  const size_t n16 = bytes / 16;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 256 * 16;
  hipLaunchKernelGGL((stream<R, W>), dim3(grid), dim3(256), 0, 0, in, out, n16, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream<R, W>), dim3(grid), dim3(256), 0, 0, in, out, n16, 1u);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double iters = (double)(n16 / (R > W ? R : W));
  const double moved = iters * 16.0 * (R + W);
  printf("%-28s %7.3f ms  %6.2f TB/s (%.2f GB read, %.2f GB written)\n", name, ms, moved / ms / 1e9, iters * 16 * R / 1e9, iters * 16 * W / 1e9);
}

int main() {
  const size_t bytes = 6ull << 30;
  uint4 *in, *out;
  hipMalloc(&in, bytes); hipMalloc(&out, bytes);
  hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
  run<1, 0>("read only (16 B/lane)", in, out, bytes);
  run<3, 0>("read only (3 x 16 B/lane)", in, out, bytes);
  run<0, 1>("write only", in, out, bytes);
  run<1, 1>("copy 1:1", in, out, bytes);
  run<3, 1>("3 read : 1 written (luma)", in, out, bytes);
  run<1, 4>("1 read : 4 written (dct)", in, out, bytes);
  return 0;
}
