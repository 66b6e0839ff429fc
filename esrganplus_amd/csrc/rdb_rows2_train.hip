// rdb_rows2_train.hip — the training-forward chain (esr_rdb_chain.mode 1) built for 2 row(s) per wave: 8x32 tiles
// for launches whose 16x32 tiles would leave most CUs idle (csrc/rdb_chain_kernel.h: ESR_R; rdb_fused.hip: pick_rows).
#define ESR_R 2
#include "rdb_chain_kernel.h"

int esr_rdb_launch_train_r2(const esr_rdb_chain& p, int grid, int ntiles, int tiles_x, int tiles_y, unsigned* host_abort, hipStream_t st) {
  if (p.noise_mode != ESR_NOISE_OFF)
    hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 1, false, 1>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  else
    hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 1, false, 0>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  return esr_check_launch("rdb_chain_kernel<train, 2 rows>");
}
