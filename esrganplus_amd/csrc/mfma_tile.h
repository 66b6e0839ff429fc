// mfma_tile.h — device helpers shared by the MFMA kernels (conv_mfma.hip, rdb_fused.hip):
// MFMA wrappers, 16-channel pixel <-> register conversion for the G32 layout, LDS-DMA, compile-time loops.
#pragma once
#include <type_traits>
#include <utility>
#include "common.h"

#ifndef ESR_ACT_AUX
#define ESR_ACT_AUX 0   // cache-policy bits of the ACTIVATION DMAs (A/B knob: 2 = nt, 1 = sc0, 16 = sc1)
#endif

namespace {


template <typename T> __device__ __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b);

template <> __device__ __forceinline__ void mma<_Float16>(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a),
                                              __builtin_bit_cast(half8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<float>(f32x16& acc, const u32x4& a, const u32x4& b) {
  const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
  for (int t = 0; t < 4; ++t)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
}

// 16 consecutive channels of one pixel <-> float[16]
template <typename T> struct Px16;
template <> struct Px16<_Float16> {
  // lane half h owns group (2*cb + h): one 32-byte group
  static __device__ __forceinline__ void load(const esr_g32& t, int b, int cb, int h, int64_t pix, float v[16]) {
    if (2 * cb + h >= t.ngroups) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = 0.f;
      return;
    }
    const char* p = (const char*)t.ptr + b * t.batch_stride + (int64_t)(2 * cb + h) * t.group_stride + pix * 32;
    const u32x4 a = *(const u32x4*)p, c = *(const u32x4*)(p + 16);
    const half8 x = __builtin_bit_cast(half8, a), y = __builtin_bit_cast(half8, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (float)x[i]; v[8 + i] = (float)y[i]; }
  }
  static __device__ __forceinline__ void store(const esr_g32& t, int b, int cb, int h, int64_t pix, const float v[16], int flavour = 0) {
    if (2 * cb + h >= t.ngroups) return;
    char* p = (char*)t.ptr + b * t.batch_stride + (int64_t)(2 * cb + h) * t.group_stride + pix * 32;
    half8 x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)v[i]; y[i] = (_Float16)v[8 + i]; }
    const u32x4 a = __builtin_bit_cast(u32x4, x), c = __builtin_bit_cast(u32x4, y);
    if (flavour == 1) {          // non-temporal (measured +1 % on the forward bench, profiles/)
      __builtin_nontemporal_store(a, (u32x4*)p);
      __builtin_nontemporal_store(c, (u32x4*)(p + 16));
    } else if (flavour == 2) {   // measurement-only (NO output): the epilogue's arithmetic without its store instructions
      if (a[0] == 0x7fc07fc1u && c[3] == 0x12345678u) *(u32x4*)p = a;      // (keeps the values alive)
    } else if (flavour == 3) {   // measurement-only (WRONG layout): each store instruction of a half-wave
                                 // covers 512 contiguous bytes instead of every other 16 bytes of 1 KB
      const int j = __lane_id() & 31;
      char* row = p - j * 32;
      *(u32x4*)(row + j * 16) = a;
      *(u32x4*)(row + 512 + j * 16) = c;
    } else {
      *(u32x4*)p = a;
      *(u32x4*)(p + 16) = c;
    }
  }
};
template <> struct Px16<float> {
  // lane half h owns groups (4*cb + 2h) and (4*cb + 2h + 1)
  static __device__ __forceinline__ void load(const esr_g32& t, int b, int cb, int h, int64_t pix, float v[16]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
      if (4 * cb + 2 * h + g < t.ngroups) {
        const char* p = (const char*)t.ptr + b * t.batch_stride + (int64_t)(4 * cb + 2 * h + g) * t.group_stride + pix * 32;
        a = *(const f32x4*)p; c = *(const f32x4*)(p + 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[8 * g + i] = a[i]; v[8 * g + 4 + i] = c[i]; }
    }
  }
  static __device__ __forceinline__ void store(const esr_g32& t, int b, int cb, int h, int64_t pix, const float v[16], int flavour = 0) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (4 * cb + 2 * h + g >= t.ngroups) continue;
      char* p = (char*)t.ptr + b * t.batch_stride + (int64_t)(4 * cb + 2 * h + g) * t.group_stride + pix * 32;
      f32x4 a, c;
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = v[8 * g + i]; c[i] = v[8 * g + 4 + i]; }
      *(f32x4*)p = a;
      *(f32x4*)(p + 16) = c;
    }
  }
};

// async global -> LDS copy of 16 bytes per lane (LDS-DMA): destination = wave-uniform LDS base
// + lane*16, source = per-lane global address.  No VGPR round trip, no staging registers.
__device__ __forceinline__ void dma16(const char* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void dma16_act(const char* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, ESR_ACT_AUX);
}

// Philox seed: by value, or through device memory so a captured graph replays with a fresh seed
__device__ __forceinline__ uint64_t noise_seed(const esr_conv& p) {
  return p.seed_dev ? __builtin_nontemporal_load(p.seed_dev) : p.seed;
}

// Accumulators are 8 NAMED vector members (never an indexable array): every access is resolved at
// compile time, so the register allocator keeps them in AGPRs for the whole kernel.  (An
// `f32x16 acc[8]` that is indexed by a not-fully-unrolled loop anywhere — e.g. the epilogue — is
// demoted to scratch and re-stored after every K step.)
struct Acc8 { f32x16 a0, a1, a2, a3, a4, a5, a6, a7; };

template <int R> __device__ __forceinline__ f32x16& accsel(Acc8& s) {
  static_assert(R >= 0 && R < 8, "row");
  if constexpr (R == 0) return s.a0;
  else if constexpr (R == 1) return s.a1;
  else if constexpr (R == 2) return s.a2;
  else if constexpr (R == 3) return s.a3;
  else if constexpr (R == 4) return s.a4;
  else if constexpr (R == 5) return s.a5;
  else if constexpr (R == 6) return s.a6;
  else return s.a7;
}

__device__ __forceinline__ void acc_zero(Acc8& s) {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  s.a0 = z; s.a1 = z; s.a2 = z; s.a3 = z; s.a4 = z; s.a5 = z; s.a6 = z; s.a7 = z;
}

__device__ __forceinline__ f32x16 pick8(const Acc8& s, int r) {   // r is wave-uniform
  switch (r) {
    case 0: return s.a0;
    case 1: return s.a1;
    case 2: return s.a2;
    case 3: return s.a3;
    case 4: return s.a4;
    case 5: return s.a5;
    case 6: return s.a6;
    default: return s.a7;
  }
}

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(f, std::make_integer_sequence<int, N>{});
}

// Raw (storage-typed) 16-channel pixel: 32 bytes for fp16, 64 for fp32.
template <typename T> struct Raw16;
template <> struct Raw16<_Float16> {
  u32x4 q[2];
  __device__ __forceinline__ void load(const esr_g32& t, int b, int cb, int h, int64_t pix) {
    if (2 * cb + h >= t.ngroups) { q[0] = u32x4{0, 0, 0, 0}; q[1] = u32x4{0, 0, 0, 0}; return; }
    const char* p = (const char*)t.ptr + b * t.batch_stride + (int64_t)(2 * cb + h) * t.group_stride + pix * 32;
    q[0] = *(const u32x4*)p; q[1] = *(const u32x4*)(p + 16);
  }
  __device__ __forceinline__ void get(float v[16]) const {
    const half8 x = __builtin_bit_cast(half8, q[0]), y = __builtin_bit_cast(half8, q[1]);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (float)x[i]; v[8 + i] = (float)y[i]; }
  }
};
template <> struct Raw16<float> {
  f32x4 q[4];
  __device__ __forceinline__ void load(const esr_g32& t, int b, int cb, int h, int64_t pix) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      q[2 * g] = f32x4{0.f, 0.f, 0.f, 0.f}; q[2 * g + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (4 * cb + 2 * h + g < t.ngroups) {
        const char* p = (const char*)t.ptr + b * t.batch_stride + (int64_t)(4 * cb + 2 * h + g) * t.group_stride + pix * 32;
        q[2 * g] = *(const f32x4*)p; q[2 * g + 1] = *(const f32x4*)(p + 16);
      }
    }
  }
  __device__ __forceinline__ void get(float v[16]) const {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[4 * g + i] = q[g][i];
  }
};

}  // namespace
