// rdb_fused_train.hip — the TRAINING-forward instantiation of the fused dense-block chain (esr_rdb_chain.mode 1,
// csrc/rdb_chain_kernel.h): block.py:260-268,287-291 with every activation the backward needs left in memory.
#include "rdb_chain_kernel.h"

int esr_rdb_launch_train(const esr_rdb_chain& p, int grid, int ntiles, int tiles_x, int tiles_y, unsigned* host_abort, hipStream_t st) {
  // one instantiation per block-tail form (rdb_chain_kernel.h: NZ)
  if (p.noise_mode != ESR_NOISE_OFF)
    hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 1, false, 1>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  else
    hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 1, false, 0>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  return esr_check_launch("rdb_chain_kernel<train>");
}
