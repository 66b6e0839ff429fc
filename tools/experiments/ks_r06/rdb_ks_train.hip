// rdb_ks_train.hip — the training-forward chain (esr_rdb_chain.mode 1) on 4x32 tiles in the K-split form: two row pairs
// x two halves of the contraction per workgroup (csrc/rdb_chain_kernel.h: ESR_KS; rdb_fused.hip picks it for the
// launches the one-row build used to take).
#define ESR_R 2
#define ESR_KS 2
#include "rdb_chain_kernel.h"

int esr_rdb_launch_train_ks(const esr_rdb_chain& p, int grid, int ntiles, int tiles_x, int tiles_y, unsigned* host_abort, hipStream_t st) {
  if (p.noise_mode != ESR_NOISE_OFF)
    hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 1, false, 1>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  else
    hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 1, false, 0>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  return esr_check_launch("rdb_chain_kernel<train, K-split>");
}
