// probe.hip -- what HBM gives a plain streaming kernel on THIS GPU, for the bench's roofline context.
// Not part of the hot path: bench.py times these launches next to the kernels it reports, because the
// rate a memory-bound kernel can reach differs from box to box (DCT+quant: 1.52 - 1.66 ms on nominally
// identical MI355X) and with the read/write mix (tools/ubench_hbm.hip is the stand-alone version).
#include "svc_common.hpp"

namespace svc {

// each lane: R dwordx4 loads and W dwordx4 stores per iteration, contiguous across the workgroup
template <int R, int W>
__global__ __launch_bounds__(256) void stream_probe_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t iters) {
  // a workgroup walks its own contiguous slice (what a fill does): scattering consecutive 4 KiB pieces over the
  // whole grid, as a grid-stride loop does, costs a third of the write rate on this HBM
  const uint64_t per_wg = (iters + gridDim.x - 1) / gridDim.x;
  const uint64_t i0 = (uint64_t)blockIdx.x * per_wg, i1 = i0 + per_wg < iters ? i0 + per_wg : iters;
  uint4 acc = make_uint4(1, 0, 0, 0);
  for (uint64_t i = i0 + threadIdx.x; i < i1; i += 256) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint4 v = in[i * R + r];
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) out[i * W + w] = acc;
  }
  if (W == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;  // keeps the loads alive
}

int launch_stream_probe(const void* d_in, void* d_out, uint64_t bytes, uint32_t reads, uint32_t writes, hipStream_t stream) {
  const uint64_t per_iter = 16ull * (reads > writes ? reads : writes);
  const uint64_t iters = bytes / per_iter;
  if (iters == 0) return SVC_OK;
  const dim3 grid(256 * 16), block(256);
  const uint4* in = static_cast<const uint4*>(d_in);
  uint4* out = static_cast<uint4*>(d_out);
#define SVC_PROBE(R_, W_) hipLaunchKernelGGL((stream_probe_kernel<R_, W_>), grid, block, 0, stream, in, out, iters)
  if (reads == 1 && writes == 0) SVC_PROBE(1, 0);
  else if (reads == 3 && writes == 0) SVC_PROBE(3, 0);
  else if (reads == 0 && writes == 1) SVC_PROBE(0, 1);
  else if (reads == 1 && writes == 1) SVC_PROBE(1, 1);
  else if (reads == 3 && writes == 1) SVC_PROBE(3, 1);
  else if (reads == 1 && writes == 4) SVC_PROBE(1, 4);
  else return fail(SVC_ERR_UNSUPPORTED, "probe: read:write mix %u:%u is not one of 1:0, 3:0, 0:1, 1:1, 3:1, 1:4", reads, writes);
#undef SVC_PROBE
  return check_launch("stream_probe_kernel");
}

}  // namespace svc
