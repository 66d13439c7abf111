// hbma_fused32.hip -- the lane-per-block all-level kernel (hbma_fused_kernel.hpp) instantiated for 32 x 32 MV blocks.
#include "hbma_fused_kernel.hpp"

namespace svc {

int launch_fused_mb32(const FusedArgs& a, uint32_t levels, uint32_t rt, dim3 grid, hipStream_t stream) {
  return launch_fused_mb<32>(a, levels, rt, grid, stream);
}

}  // namespace svc
