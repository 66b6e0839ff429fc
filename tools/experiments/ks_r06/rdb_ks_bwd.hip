// rdb_ks_bwd.hip — the backward chain (esr_rdb_backward) on 4x32 tiles in the K-split form (csrc/rdb_chain_kernel.h:
// ESR_KS; rdb_fused.hip picks it for the launches the one-row build used to take).
#define ESR_R 2
#define ESR_KS 2
#include "rdb_chain_kernel.h"

int esr_rdb_launch_bwd_ks(const esr_rdb_chain& p, int grid, int ntiles, int tiles_x, int tiles_y, unsigned* host_abort, hipStream_t st) {
  hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 2>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  return esr_check_launch("rdb_chain_kernel<backward, K-split>");
}
