// probe.hip -- what HBM gives a plain streaming kernel on THIS GPU, for the bench's roofline context.
// Not part of the hot path: bench.py times these launches next to the kernels it reports, because the
// rate a memory-bound kernel can reach differs from box to box (DCT+quant: 1.52 - 1.66 ms on nominally
// identical MI355X) and with the read/write mix (tools/ubench_hbm.hip is the stand-alone version).
#include "svc_common.hpp"

namespace svc {

// A workgroup's unit: 256 lanes x R adjacent dwordx4 loads (a lane's loads are adjacent: the BGR pattern for R = 3) and W dwordx4 stores
// (store k of all lanes contiguous: 4 KiB pieces).  ONE unit per short-lived workgroup, units dealt XCD-contiguously: the form that gets
// the most out of this HBM -- workgroups that LOOP over a slice of the buffer (what this probe did until round 4) lose 10 - 35 %
// (tools/ubench_stream_oneshot.hip, profiles/r04_ubench_oneshot.txt: write-only 7.1 against 4.6 TB/s, 3 : 1 5.9 against 4.9).
template <int R, int W>
__global__ __launch_bounds__(256) void stream_probe_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t units) {
  const size_t u = xcd_contiguous_block(blockIdx.x, units);
  const size_t i = u * 256 + threadIdx.x;
  uint4 acc = make_uint4(1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint4 v = in[i * R + r];
    acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
  }
#pragma unroll
  for (int w = 0; w < W; ++w) out[(u * W + w) * 256 + threadIdx.x] = acc;
  if (W == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;  // keeps the loads alive
}

int launch_stream_probe(const void* d_in, void* d_out, uint64_t bytes, uint32_t reads, uint32_t writes, hipStream_t stream) {
  const uint64_t per_unit = 256ull * 16ull * (reads > writes ? reads : writes);  // the larger side of a unit
  const uint64_t n_units = bytes / per_unit;
  if (n_units == 0) return SVC_OK;
  if (n_units > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "probe: %llu units exceed one launch", (unsigned long long)n_units);
  const uint32_t units = (uint32_t)n_units;
  const dim3 grid(units), block(256);
  const uint4* in = static_cast<const uint4*>(d_in);
  uint4* out = static_cast<uint4*>(d_out);
#define SVC_PROBE(R_, W_) hipLaunchKernelGGL((stream_probe_kernel<R_, W_>), grid, block, 0, stream, in, out, units)
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
