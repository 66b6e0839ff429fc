// Internal helpers shared by the gfx950 kernels of libesrgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/esrgan_hip.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define ESR_LRELU_SLOPE 0.2f
#define ESR_NO_LAYER 0xFFFFFFFFu

void esr_set_error(const char* fmt, ...);
int esr_check_launch(const char* what);

template <typename T> struct DT;
template <> struct DT<_Float16> {
  static constexpr int CPG = 16;   // channels per 32-byte group
  static constexpr int GPB = 2;    // groups per 32-cout block
  static constexpr int DTYPE = ESR_F16;
};
template <> struct DT<float> {
  static constexpr int CPG = 8;
  static constexpr int GPB = 4;
  static constexpr int DTYPE = ESR_F32;
};

// MFMA accumulator row -> packed cout row permutation.  With the weights' 32 cout rows stored in
// the order pi(i), lane (pixel j, half h) ends up holding the 16 CONSECUTIVE couts 16h..16h+15 in
// its 16 accumulator registers (C/D map of v_mfma_*_32x32: row = (r&3) + 8*(r>>2) + 4*h), so the
// epilogue reads/writes whole 32-byte channel groups per lane.
__host__ __device__ inline int esr_pi(int i) {   // packed A-row i -> cout within the 32-block
  const int h = (i >> 2) & 1;
  const int r = (i & 3) + 4 * (i >> 3);
  return 16 * h + r;
}

// ------------------------------------------------------------------------------------------
// GaussianNoise's z when no explicit z tensor is given: a counter-based stream of this library's own definition
// (block.py:120 draws from torch's global generator, which no fused kernel can reproduce; the explicit-z mode is
// the bit-parity path).  Counter = (pixel, channel / 8, layer, 0), key = seed, so that the training forward, the
// backward (which needs the same z again) and esr_fill_noise all compute a value from its coordinates alone.
//   * Philox-4x32 with 7 rounds: the smallest round count that passes BigCrush (Salmon, Moraes, Dror, Shaw,
//     "Parallel random numbers: as easy as 1, 2, 3", SC'11, table 2); the customary 10 only adds margin, and
//     the 32x32->64 multiplies are quarter-rate VALU work executed in the chain's epilogues, where the MFMA
//     pipe idles (10 rounds + 4 normals per call cost 2.9 ms per fwd+bwd at the bench shape);
//   * 8 normals per call: word k feeds one Box-Muller pair — radius from its high 16 bits (u in (0, 1] in steps
//     of 2^-16, |z| <= 4.71), angle from its low 16 bits.
// ------------------------------------------------------------------------------------------
#define ESR_PHILOX_ROUNDS 7
__device__ inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                  uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int i = 0; i < ESR_PHILOX_ROUNDS; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// z[0..7] = N(0,1) for channels 8 * oct .. 8 * oct + 7 of pixel `pix` in noise layer `layer`
__device__ inline void philox_normal8(uint32_t pix, uint32_t oct, uint32_t layer, uint64_t seed, float z[8]) {
  uint32_t r[4];
  philox4x32(pix, oct, layer, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float u0 = (float)((r[k] >> 16) + 1u) * 1.52587890625e-05f;          // (0, 1]
    const float u1 = (float)(r[k] & 0xFFFFu) * 1.52587890625e-05f;            // [0, 1): the angle in turns
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));   // sqrt(-2 ln u0), v_log_f32 = log2
    z[2 * k] = rad * __builtin_amdgcn_cosf(u1);                                // v_cos_f32 / v_sin_f32 take turns
    z[2 * k + 1] = rad * __builtin_amdgcn_sinf(u1);
  }
}
