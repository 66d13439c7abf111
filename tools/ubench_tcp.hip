// ubench_tcp.hip -- what a vector-memory wave instruction costs the vector L1 (TCP) on gfx950, by access shape.
// Every kernel reads an L1-resident region (18 KB, the same for every workgroup) so that the time per wave
// instruction is the TCP's processing rate, not HBM's; run it under
//   rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_WAVES
// for tag lookups per wave instruction.  Shapes are the ones the fused motion search issues (per-lane windows, 16 B
// lane pitch + a per-lane motion vector) and the ones an LDS-staging load would issue (whole rows).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_tcp.hip -o tools/_bin/ubench_tcp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define ITERS 512
constexpr int PITCH = 1152;  // bytes per row of the region (18 x 64)
constexpr int ROWS = 16;

typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// SHAPE:
//  0 x4  lane*16, all lanes one row, 64-B aligned run             (anchor rows of level 0)
//  1 x4  lane*16 + 4, one row                                     (coherent tracked rows, misaligned)
//  2 x4  lane*16, every lane its own row                          (incoherent, 16-B aligned)
//  3 x4  lane*16 + 4*(0..3) + own row                             (incoherent tracked rows: what <4,1> issues)
//  4 x2  the 8 bytes behind shape 3's 16                          (second half of the 24-byte window)
//  5 x4  lane pairs: [a, a+16) and [a+8, a+24), pair's own row    (two lanes per block)
//  6 x1  lane*4 + 4*(0..3) + own row                              (level 2: three single dwords ...)
//  7 x3  same addresses as 6                                      (... or one dwordx3)
//  8 x4  tile rows of 19 chunks, row start 16-B aligned only      (staging load, tight)
//  9 x4  tile rows of 24 chunks, row start 64-B aligned           (staging load, padded to sectors)
// 10 x2  lane*8 one row                                           (anchor rows of level 1)
// 11 x1  lane*4 one row                                           (anchor rows of level 2)
// 12 x4  lane*16 + 4*(0..3), rows own with 40 % of the lanes on the common row (measured field statistics)
// 13 x4  even lanes only (32 of 64 active), (lane/2)*16, one row   (anchor rows when two lanes share a block and one of them loads)
// 14 x4  lanes 0..31 only, lane*16, one row                        (the same with the active lanes packed)
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ buf, uint32_t* out, uint32_t seed) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t acc = 0;
  uint32_t h = hash(seed + threadIdx.x * 977u + blockIdx.x * 7919u);
  for (int it = 0; it < ITERS; ++it) {
    h = h * 1664525u + 1013904223u;
    const uint32_t row = (h >> 8) % ROWS, sh = ((h >> 16) & 3u) * 4u;
    const uint32_t urow = (uint32_t)it % ROWS;  // uniform row
    uint32_t off;
    if (SHAPE == 0) off = urow * PITCH + lane * 16;
    if (SHAPE == 1) off = urow * PITCH + lane * 16 + 4;
    if (SHAPE == 2) off = row * PITCH + lane * 16;
    if (SHAPE == 3) off = row * PITCH + lane * 16 + sh;
    if (SHAPE == 4) off = row * PITCH + lane * 16 + sh + 16;
    if (SHAPE == 5) {
      const uint32_t hp = __shfl(h, (int)(lane & ~1u));
      off = ((hp >> 8) % ROWS) * PITCH + (lane >> 1) * 16 + ((hp >> 16) & 3u) * 4u + (lane & 1u) * 8;
    }
    if (SHAPE == 6 || SHAPE == 7) off = row * PITCH + lane * 4 + sh;
    if (SHAPE == 8) { const uint32_t t = lane + 64u * (it & 3); off = ((t / 19u) % ROWS) * PITCH + 48 + (t % 19u) * 16; }
    if (SHAPE == 9) { const uint32_t t = lane + 64u * (it & 3); off = ((t / 24u) % ROWS) * PITCH + (t % 24u) * 16; }
    if (SHAPE == 10) off = urow * PITCH + lane * 8;
    if (SHAPE == 11) off = urow * PITCH + lane * 4;
    if (SHAPE == 12) off = (((h >> 20) % 10u) < 4u ? urow : row) * PITCH + lane * 16 + sh;
    if (SHAPE == 13) { if (lane & 1u) continue; off = urow * PITCH + (lane >> 1) * 16; }
    if (SHAPE == 14) { if (lane >= 32u) continue; off = urow * PITCH + lane * 16; }
    const uint8_t* p = buf + off;
    if (SHAPE == 4 || SHAPE == 10) { u32x2 v = *reinterpret_cast<const u32x2*>(p); acc ^= v.x ^ v.y; }
    else if (SHAPE == 6 || SHAPE == 11) { acc ^= *reinterpret_cast<const uint32_t*>(p); }
    else if (SHAPE == 7) { u32x3 v = *reinterpret_cast<const u32x3*>(p); acc ^= v.x ^ v.y ^ v.z; }
    else { u32x4 v = *reinterpret_cast<const u32x4*>(p); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int SHAPE> void run(const char* name, const uint8_t* buf, uint32_t* out) {
  const int blocks = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, buf, out, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, buf, out, 2u + r);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  // 16 waves per CU share one TCP: ns per wave instruction as the TCP sees them back to back
  const double inst_per_cu = 16.0 * ITERS;
  printf("%-2d %-58s %8.3f ms  %7.1f ns per wave instruction per CU (%.0f clk at 2.1 GHz)\n", SHAPE, name, ms,
         ms * 1e6 / inst_per_cu, ms * 1e6 / inst_per_cu * 2.1);
}

int main() {
  uint8_t* buf; uint32_t* out;
  hipMalloc(&buf, PITCH * (ROWS + 2));
  hipMalloc(&out, 256 * 4 * 256 * 4);
  std::vector<uint8_t> h(PITCH * (ROWS + 2));
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 31 + 7);
  hipMemcpy(buf, h.data(), h.size(), hipMemcpyHostToDevice);
  run<0>("x4 lane*16, one row, aligned", buf, out);
  run<1>("x4 lane*16+4, one row", buf, out);
  run<2>("x4 lane*16, own rows", buf, out);
  run<3>("x4 lane*16+4s, own rows (tracked row, first 16 B)", buf, out);
  run<4>("x2 behind it (tracked row, last 8 B)", buf, out);
  run<5>("x4 lane pairs over 24 B, pair's own row", buf, out);
  run<6>("x1 lane*4+4s, own rows", buf, out);
  run<7>("x3 lane*4+4s, own rows", buf, out);
  run<8>("x4 tile rows of 19 chunks (16-B aligned start)", buf, out);
  run<9>("x4 tile rows of 24 chunks (64-B aligned start)", buf, out);
  run<10>("x2 lane*8, one row", buf, out);
  run<11>("x1 lane*4, one row", buf, out);
  run<12>("x4 lane*16+4s, 40 % of lanes on a common row", buf, out);
  run<13>("x4 even lanes only, (lane/2)*16, one row (512 B)", buf, out);
  run<14>("x4 lanes 0..31 only, lane*16, one row (512 B)", buf, out);
  return 0;
}
