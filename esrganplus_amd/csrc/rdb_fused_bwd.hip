// rdb_fused_bwd.hip — the BACKWARD instantiation of the fused dense-block chain (esr_rdb_backward, esr_rdb_chain.mode 2,
// csrc/rdb_chain_kernel.h): the input gradients of block.py:260-268,287-291 (autograd backward at SRRaGAN_model.py:140).
#include "rdb_chain_kernel.h"

int esr_rdb_launch_bwd(const esr_rdb_chain& p, int grid, int ntiles, int tiles_x, int tiles_y, unsigned* host_abort, hipStream_t st) {
  hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 2>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  return esr_check_launch("rdb_chain_kernel<bwd>");
}
