// rdb_rows1_fwd.hip — the inference chain (fp16, no noise layers) built for 1 row(s) per wave: 4x32 tiles for
// launches whose 16x32 tiles would leave most CUs idle — a single 128x128 LR tile (BASELINE configs[0],
// test_image/test.py) is 32 such tiles on 256 CUs (csrc/rdb_chain_kernel.h: ESR_R; rdb_fused.hip: rows_per_wave).
#define ESR_R 1
#include "rdb_chain_kernel.h"

int esr_rdb_launch_fwd_r1(const esr_rdb_chain& p, int grid, int ntiles, int tiles_x, int tiles_y, unsigned* host_abort, hipStream_t st) {
  hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 0, false, 0>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  return esr_check_launch("rdb_chain_kernel<forward, 1 rows>");
}
