// rdb_rows1_bwd.hip — the backward chain (esr_rdb_backward) built for 1 row(s) per wave: 4x32 tiles
// (csrc/rdb_chain_kernel.h: ESR_R; rdb_fused.hip: pick_rows).
#define ESR_R 1
#include "rdb_chain_kernel.h"

int esr_rdb_launch_bwd_r1(const esr_rdb_chain& p, int grid, int ntiles, int tiles_x, int tiles_y, unsigned* host_abort, hipStream_t st) {
  hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 2>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  return esr_check_launch("rdb_chain_kernel<backward, 1 rows>");
}
