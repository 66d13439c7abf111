// ubench_pcie.hip -- what the device -> host direction of this box's PCIe link gives, by who moves the bytes (round 6: svc::StreamEncoder's
// figure IS the D2H rate of its box, 33 - 55 GB/s across the pool while H2D stays at 54 - 55: DESIGN.md section 5).
//   (a) hipMemcpyAsync into pinned memory (the runtime's choice: an SDMA engine), one call of the whole buffer and in 25 MB pieces;
//   (b) a copy KERNEL storing to the pinned buffer's device address, by workgroup count and store form;
//   (c) the same two for host -> device (loads from the pinned buffer), for the asymmetry.
// HSA_ENABLE_SDMA=0 in the environment makes (a) use the runtime's blit kernels instead.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_pcie.hip -o tools/_bin/ubench_pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// grid-stride copy of n 16-byte units; NT: nontemporal stores
template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const u32x4 v = src[i];
    if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}

// each workgroup owns a contiguous slice and walks it 4 KiB at a time, four loads in flight per lane
__global__ __launch_bounds__(256) void copy_slices_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + 256 * k < hi) v[k] = src[i + 256 * k];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + 256 * k < hi) dst[i + 256 * k] = v[k];
  }
}

template <class F>
static double time_ms(hipStream_t s, int reps, F&& f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 400) << 20, n = bytes / 16;
  const int reps = 5;
  uint8_t *dev = nullptr, *pin = nullptr;
  CK(hipMalloc(&dev, bytes));
  CK(hipHostMalloc(reinterpret_cast<void**>(&pin), bytes, hipHostMallocDefault));
  CK(hipMemset(dev, 1, bytes));
  for (size_t i = 0; i < bytes; i += 4096) pin[i] = 2;  // pages touched
  void* pin_dev = nullptr;
  CK(hipHostGetDevicePointer(&pin_dev, pin, 0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const char* sdma = std::getenv("HSA_ENABLE_SDMA");
  std::printf("buffer %zu MiB, HSA_ENABLE_SDMA=%s\n", bytes >> 20, sdma ? sdma : "(unset)");
  auto line = [&](const char* what, double ms) { std::printf("%-86s %8.3f ms  %6.1f GB/s\n", what, ms, bytes / ms * 1e-6); std::fflush(stdout); };

  line("D2H hipMemcpyAsync, one call", time_ms(s, reps, [&] { CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s)); }));
  line("D2H hipMemcpyAsync, 25 MiB pieces", time_ms(s, reps, [&] {
         for (size_t o = 0; o < bytes; o += (25u << 20)) CK(hipMemcpyAsync(pin + o, dev + o, bytes - o < (25u << 20) ? bytes - o : (25u << 20), hipMemcpyDeviceToHost, s)); }));
  for (int g : {4, 8, 16, 32, 64, 128, 256, 1024}) {
    char t[128];
    std::snprintf(t, sizeof t, "D2H copy kernel, grid-stride, %d workgroups", g);
    line(t, time_ms(s, reps, [&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(g), dim3(256), 0, s, (u32x4*)pin_dev, (const u32x4*)dev, n); }));
  }
  for (int g : {16, 64, 256}) {
    char t[128];
    std::snprintf(t, sizeof t, "D2H copy kernel, grid-stride, nontemporal stores, %d workgroups", g);
    line(t, time_ms(s, reps, [&] { hipLaunchKernelGGL(copy_kernel<true>, dim3(g), dim3(256), 0, s, (u32x4*)pin_dev, (const u32x4*)dev, n); }));
  }
  for (int g : {16, 64, 256}) {
    char t[128];
    std::snprintf(t, sizeof t, "D2H copy kernel, a contiguous slice per workgroup, %d workgroups", g);
    line(t, time_ms(s, reps, [&] { hipLaunchKernelGGL(copy_slices_kernel, dim3(g), dim3(256), 0, s, (u32x4*)pin_dev, (const u32x4*)dev, n); }));
  }
  line("H2D hipMemcpyAsync, one call", time_ms(s, reps, [&] { CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s)); }));
  for (int g : {16, 64, 256}) {
    char t[128];
    std::snprintf(t, sizeof t, "H2D copy kernel, grid-stride, %d workgroups", g);
    line(t, time_ms(s, reps, [&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(g), dim3(256), 0, s, (u32x4*)dev, (const u32x4*)pin_dev, n); }));
  }
  // a single D2H copy after the link and the copy engines have been idle for a while (a short clip's batches come in bursts)
  for (int gap_ms : {0, 1, 5, 20, 100}) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double sum = 0;
    for (int i = 0; i < 4; ++i) {
      CK(hipStreamSynchronize(s));
      const auto t0 = std::chrono::steady_clock::now();
      while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < gap_ms) {}
      CK(hipEventRecord(e0, s));
      CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      sum += ms;
    }
    char t[128];
    std::snprintf(t, sizeof t, "D2H hipMemcpyAsync, one call, %d ms of idle in front of each", gap_ms);
    line(t, sum / 4);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
  // both directions at once on two streams (what a pipelined encoder does): SDMA both ways, then the kernel for D2H beside an SDMA H2D
  {
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    uint8_t *dev2 = nullptr, *pin2 = nullptr;
    const size_t b2 = bytes / 4;  // H2D is a quarter of D2H in the encoder (6.3 MB in, 25 MB out per frame)
    CK(hipMalloc(&dev2, b2));
    CK(hipHostMalloc(reinterpret_cast<void**>(&pin2), b2, hipHostMallocDefault));
    for (size_t i = 0; i < b2; i += 4096) pin2[i] = 3;
    auto both = [&](bool kernel) {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < reps; ++i) {
        CK(hipMemcpyAsync(dev2, pin2, b2, hipMemcpyHostToDevice, s2));
        if (kernel) hipLaunchKernelGGL(copy_kernel<false>, dim3(64), dim3(256), 0, s, (u32x4*)pin_dev, (const u32x4*)dev, n);
        else CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s));
      }
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
      return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    both(false);
    line("D2H hipMemcpyAsync beside an H2D hipMemcpyAsync of a quarter the size (D2H bytes / wall)", both(false));
    both(true);
    line("D2H copy kernel (64 workgroups) beside the same H2D", both(true));
  }
  // the result is the source
  CK(hipMemset(dev, 7, bytes));
  CK(hipDeviceSynchronize());  // s is a non-blocking stream: it does not wait for the null stream's memset
  hipLaunchKernelGGL(copy_kernel<false>, dim3(64), dim3(256), 0, s, (u32x4*)pin_dev, (const u32x4*)dev, n);
  CK(hipStreamSynchronize(s));
  size_t bad = 0;
  for (size_t i = 0; i < bytes; i += 4099) bad += pin[i] != 7;
  std::printf("copy kernel result check: %zu mismatches\n", bad);
  return bad != 0;
}
