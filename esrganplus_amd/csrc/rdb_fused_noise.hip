// rdb_fused_noise.hip — the inference chain with the fused Philox noise layers (a train-mode module under no_grad:
// block.py:117-123 is active whenever `self.training`), csrc/rdb_chain_kernel.h: NZ = 1.  The common inference
// instantiations (rdb_fused.hip) are built without this block tail.
#include "rdb_chain_kernel.h"

int esr_rdb_launch_noisy(const esr_rdb_chain& p, int grid, int ntiles, int tiles_x, int tiles_y, unsigned* host_abort, hipStream_t st) {
  if (p.dtype == ESR_F16)
    hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 0, false, 1>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  else if (p.dtype == ESR_F32)
    hipLaunchKernelGGL((rdb_chain_kernel<float, 0, false, 1>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  else { esr_set_error("esr_rdb_forward: bad dtype %d", p.dtype); return ESR_ERR_INVALID; }
  return esr_check_launch("rdb_chain_kernel<noise>");
}
