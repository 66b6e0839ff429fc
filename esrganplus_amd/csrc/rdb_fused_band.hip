// rdb_fused_band.hip — the row-band instantiations of the fused dense-block chain (esr_rdb_chain.band_rows,
// csrc/rdb_chain_kernel.h): inference forward of images with more 16x32 tiles than the GPU has CUs (a DIV2K-sized
// LR image, test_image/test.py:26-40), cut into bands that each recompute a margin of their neighbours' rows.
#include "rdb_chain_kernel.h"

int esr_rdb_launch_band(const esr_rdb_chain& p, int grid, int ntiles, int tiles_x, int tiles_y, unsigned* host_abort, hipStream_t st) {
  if (p.dtype == ESR_F16)
    hipLaunchKernelGGL((rdb_chain_kernel<_Float16, 0, true, 0>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  else
    hipLaunchKernelGGL((rdb_chain_kernel<float, 0, true, 0>), dim3(grid), dim3(NT), 0, st, p, ntiles, tiles_x, tiles_y, host_abort);
  return esr_check_launch("rdb_chain_kernel<band>");
}
