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
// Philox-4x32-10 + Box-Muller: 4 N(0,1) per call.  Counter = (pixel, channel/4, layer, 0),
// key = seed.  The stream is this library's own definition of GaussianNoise's z (block.py:120
// draws from torch's global generator, which no fused kernel can reproduce); the explicit-z mode
// is the bit-parity path.
// ------------------------------------------------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
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

__device__ inline void philox_normal4(uint32_t pix, uint32_t cq, uint32_t layer, uint64_t seed,
                                      float z[4]) {
  uint32_t r[4];
  philox4x32_10(pix, cq, layer, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  // u in (0,1]: (r + 1) * 2^-32 ; Box-Muller on two pairs
  const float u0 = ((float)r[0] + 1.0f) * 2.3283064365386963e-10f;
  const float u1 = (float)r[1] * 2.3283064365386963e-10f;
  const float u2 = ((float)r[2] + 1.0f) * 2.3283064365386963e-10f;
  const float u3 = (float)r[3] * 2.3283064365386963e-10f;
  const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
  float s, c;
  __sincosf(6.283185307179586f * u1, &s, &c);
  z[0] = ra * c; z[1] = ra * s;
  __sincosf(6.283185307179586f * u3, &s, &c);
  z[2] = rb * c; z[3] = rb * s;
}
