// union_find.hpp -- lock-free union-find shared by the labelling kernels (segment.hip, imageops.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace svc {

// Lock-free union-find on `parent` (LDS or global): a set's root is its smallest index (= its first element in raster
// order); links always go from the larger root to the smaller with atomicMin, so concurrent unions commute.
// SCOPE: uf_find reads `parent` with plain loads while other waves atomicMin it.  That is sound for waves of ONE workgroup (one CU: LDS, or
// global memory seen through that CU's one vector L1, where a stale parent is still an ancestor and the loop in uf_unite retries).  Every
// user in this repo is a single workgroup per union-find domain (segment.hip's label kernel: one workgroup per frame; imageops.hip's
// cc_kernel: one workgroup per image).  A grid that unites across workgroups would need device-scope atomic loads here (L1s of
// different CUs are not coherent) and a grid-wide phase split between uniting and numbering, as segment.hip's launch sequences have.
__device__ __forceinline__ uint32_t uf_find(const uint32_t* parent, uint32_t x) {
  uint32_t p = parent[x];
  while (p != x) { x = p; p = parent[x]; }
  return x;
}

__device__ __forceinline__ void uf_unite(uint32_t* parent, uint32_t a, uint32_t b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { const uint32_t t = a; a = b; b = t; }
    const uint32_t old = atomicMin(&parent[a], b);
    if (old == a) return;  // a was still a root and now points at b
    a = old;               // someone linked a first: carry on from where it points
  }
}

}  // namespace svc
