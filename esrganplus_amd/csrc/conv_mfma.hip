// conv_mfma.hip — im2col-free implicit-GEMM convolution on the gfx950 matrix cores.
//
// Replaces, for the ESRGAN+ hot path, what the reference runs as nn.Conv2d (cuDNN) + separate
// elementwise kernels: conv_block (block.py:125-151), the ResidualDenseBlock_5C tails
// (block.py:263,266,268), the RRDB tail (block.py:291; test_image/block.py:256), the trunk
// ShortcutBlock add (block.py:84-86) and nn.Upsample(x2, nearest) (block.py:315-322).
//
// GEMM view:  D[cout][pixel] += W[cout][tap, cin] * X[tap, cin][pixel]
//   A operand = packed weights (32 couts x K-slice), streamed from L2 straight into VGPRs;
//   B operand = activations: a (rows x cols) halo tile of ONE 32-byte channel group is staged in
//               LDS per K step; the 3x3 (or 4x4/s2, or upsampled) taps are just shifted
//               ds_read_b128 addresses into that tile — nothing is ever im2col'ed.
//   One wave owns 8 output rows x 32 output pixels x 32 couts = 8 MFMA 32x32 accumulators
//   (v_mfma_f32_32x32x16_f16, or 4x v_mfma_f32_32x32x2_f32 for the exact-fp32 path).  Each
//   B fragment read from LDS feeds up to KS MFMAs (the kh taps of different output rows), each
//   A fragment up to 8 (the rows).
// A workgroup is 4 waves arranged WR x WC spatially x NCG cout-blocks.
#include <type_traits>
#include <utility>

#include "common.h"

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
    const char* p = (const char*)t.ptr + b * t.batch_stride + (int64_t)(2 * cb + h) * t.group_stride + pix * 32;
    const u32x4 a = *(const u32x4*)p, c = *(const u32x4*)(p + 16);
    const half8 x = __builtin_bit_cast(half8, a), y = __builtin_bit_cast(half8, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (float)x[i]; v[8 + i] = (float)y[i]; }
  }
  static __device__ __forceinline__ void store(const esr_g32& t, int b, int cb, int h, int64_t pix, const float v[16]) {
    if (2 * cb + h >= t.ngroups) return;
    char* p = (char*)t.ptr + b * t.batch_stride + (int64_t)(2 * cb + h) * t.group_stride + pix * 32;
    half8 x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)v[i]; y[i] = (_Float16)v[8 + i]; }
    *(u32x4*)p = __builtin_bit_cast(u32x4, x);
    *(u32x4*)(p + 16) = __builtin_bit_cast(u32x4, y);
  }
};
template <> struct Px16<float> {
  // lane half h owns groups (4*cb + 2h) and (4*cb + 2h + 1)
  static __device__ __forceinline__ void load(const esr_g32& t, int b, int cb, int h, int64_t pix, float v[16]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const char* p = (const char*)t.ptr + b * t.batch_stride + (int64_t)(4 * cb + 2 * h + g) * t.group_stride + pix * 32;
      const f32x4 a = *(const f32x4*)p, c = *(const f32x4*)(p + 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[8 * g + i] = a[i]; v[8 * g + 4 + i] = c[i]; }
    }
  }
  static __device__ __forceinline__ void store(const esr_g32& t, int b, int cb, int h, int64_t pix, const float v[16]) {
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

template <int KS, int S, bool UPS, int WR, int WC>
struct Geo {
  static constexpr int TH = 8 * WR, TW = 32 * WC;                       // output tile
  static constexpr int IH = UPS ? TH / 2 + 2 : (TH - 1) * S + KS;      // staged input tile
  static constexpr int IW = UPS ? TW / 2 + 2 : (TW - 1) * S + KS;
  static constexpr int WIH = UPS ? 6 : 7 * S + KS;                      // input rows one wave touches
  static constexpr int STAGE = ((IH * IW * 32 + 1023) / 1024) * 1024;   // bytes per LDS stage
  static constexpr int NSLOT = IH * IW * 2;                             // 16-byte slots per stage
  static constexpr int NLD = (NSLOT + 255) / 256;
  static constexpr int PAD = (KS - 1) / 2;
};

// async global -> LDS copy of 16 bytes per lane (LDS-DMA): destination = wave-uniform LDS base
// + lane*16, source = per-lane global address.  No VGPR round trip, no staging registers.
__device__ __forceinline__ void dma16(const char* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
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

template <typename T>
__device__ __forceinline__ void noise16(int explicit_z, const esr_g32 zt, uint32_t layer, uint64_t seed, float sigma,
                                     int b, int cb, int h, int64_t pixoff_z, uint32_t pix, float v[16]) {
  float z[16];
  if (explicit_z) {
    Px16<T>::load(zt, b, cb, h, pixoff_z, z);
  } else {
#pragma unroll 1
    for (int q = 0; q < 4; ++q) philox_normal4(pix, (uint32_t)(cb * 8 + h * 4 + q), layer, seed, &z[4 * q]);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = v[e] + z[e] * (sigma * v[e]);   // block.py:119-121
}

// Per-row epilogue (see esr_conv in esrgan_hip.h for the operation order).
template <typename T>
__device__ __forceinline__ void epilogue_row(const esr_conv& p, const f32x16& a, const f32x16* a1, const float bias[16],
                                    int b, int cb, int h, int oy, int ox) {
  float v[16], tmp[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float x = a[e] + bias[e];
    if (p.act == ESR_ACT_LRELU) x = x > 0.f ? x : x * ESR_LRELU_SLOPE;
    else if (p.act == ESR_ACT_RELU) x = x > 0.f ? x : 0.f;
    v[e] = x;
  }
  if (p.aux_out.ptr) Px16<T>::store(p.aux_out, b, cb, h, (int64_t)(oy + 1) * p.aux_out.wp + ox + 1, v);
  if (a1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += (*a1)[e];
  }
  if (p.res1.ptr) {
    Px16<T>::load(p.res1, b, cb, h, (int64_t)(oy + 1) * p.res1.wp + ox + 1, tmp);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = v[e] * p.alpha + tmp[e];
  }
  const uint32_t pix = (uint32_t)((b * p.H + oy) * p.W + ox);
  if ((p.noise_mode == ESR_NOISE_PHILOX && p.layer1 != ESR_NO_LAYER) || (p.noise_mode == ESR_NOISE_EXPLICIT && p.z1.ptr))
    noise16<T>(p.noise_mode == ESR_NOISE_EXPLICIT, p.z1, p.layer1, p.seed, p.sigma, b, cb, h, (int64_t)(oy + 1) * p.z1.wp + ox + 1, pix, v);
  if (p.res2.ptr) {
    Px16<T>::load(p.res2, b, cb, h, (int64_t)(oy + 1) * p.res2.wp + ox + 1, tmp);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = v[e] * p.beta + tmp[e];
  }
  if ((p.noise_mode == ESR_NOISE_PHILOX && p.layer2 != ESR_NO_LAYER) || (p.noise_mode == ESR_NOISE_EXPLICIT && p.z2.ptr))
    noise16<T>(p.noise_mode == ESR_NOISE_EXPLICIT, p.z2, p.layer2, p.seed, p.sigma, b, cb, h, (int64_t)(oy + 1) * p.z2.wp + ox + 1, pix, v);
  if (p.out.ptr) Px16<T>::store(p.out, b, cb, h, (int64_t)(oy + 1) * p.out.wp + ox + 1, v);
  if (p.mask.ptr) {
    Px16<T>::load(p.mask, b, cb, h, (int64_t)(oy + 1) * p.mask.wp + ox + 1, tmp);
    const float neg = p.act == ESR_ACT_RELU ? 0.f : ESR_LRELU_SLOPE;
#pragma unroll
    for (int e = 0; e < 16; ++e) tmp[e] = tmp[e] > 0.f ? v[e] : v[e] * neg;
    Px16<T>::store(p.out2, b, cb, h, (int64_t)(oy + 1) * p.out2.wp + ox + 1, tmp);
  }
  if (p.nchw_out_c > 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ch = cb * 32 + 16 * h + e;
      if (ch < p.nchw_out_c) p.nchw_out[(((int64_t)b * p.nchw_out_c + ch) * p.H + oy) * p.W + ox] = v[e];
    }
  }
}

template <typename T, int KS, int S, bool UPS, int WR, int WC, int NCG, bool HAS1X1>
__global__ __launch_bounds__(256, 1) void conv_kernel(const esr_conv p) {
  using G = Geo<KS, S, UPS, WR, WC>;
  static_assert(WR * WC * NCG == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) char smem[2 * G::STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % NCG;
  const int wc = (wave / NCG) % WC;
  const int wr = (wave / NCG) / WC;
  const int j = lane & 31, h = lane >> 5;

  // ---- XCD-aware tile mapping: block b runs on XCD b%8; give every XCD a contiguous run of
  // tiles (neighbouring tiles share halo rows -> they hit the same private L2).
  const int tiles_x = (p.W + G::TW - 1) / G::TW, tiles_y = (p.H + G::TH - 1) / G::TH;
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
  const int cb = blockIdx.y * NCG + cg;             // 32-cout block of this wave
  const bool cb_ok = cb < p.cout_blocks;
  const int oy0 = ty * G::TH, ox0 = tx * G::TW;     // output tile origin (logical)

  // ---- staging map: LDS slot s = tid + 256*i (16 bytes) <- input tile, by LDS-DMA.
  // LDS image: [row][col][2 halves]; the two 16-byte halves of pixel `col` are swapped when
  // (col>>3)&1 so that the 16 lanes of a ds_read_b128 group cover 16 distinct bank slots.  The
  // DMA destination is lane-linear, so the swizzle is applied to the SOURCE address.
  const int iy0 = UPS ? oy0 / 2 : oy0 * S + 1 - G::PAD;   // padded coords of tile origin
  const int ix0 = UPS ? ox0 / 2 : ox0 * S + 1 - G::PAD;
  int goff[G::NLD];
#pragma unroll
  for (int i = 0; i < G::NLD; ++i) {
    int s = tid + 256 * i;
    if (s >= G::NSLOT) s = G::NSLOT - 1;            // tail lanes land in the stage's padding
    const int row = s / (2 * G::IW), rem = s - row * 2 * G::IW;
    const int col = rem >> 1, hs = rem & 1, half = hs ^ ((col >> 3) & 1);
    goff[i] = ((iy0 + row) * p.in.wp + ix0 + col) * 32 + half * 16;
  }
  const char* in_b = (const char*)p.in.ptr + b * p.in.batch_stride;
  char* const lds_w = smem + wave * 1024;           // this wave's 64-slot window inside a stage

  // ---- B-fragment (activation) read offsets, one per kw tap
  int colofs[KS];
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    const int col = UPS ? (((wc * 32 + j + kw - 1) >> 1) + 1) : ((wc * 32 + j) * S + kw);
    colofs[kw] = col * 32 + ((h ^ ((col >> 3) & 1)) << 4) + (UPS ? wr * 4 : wr * 8 * S) * G::IW * 32;
  }

  // ---- A-fragment (weight) pointer: [cb][chunk][tap][lane][16B]
  const int nchunks = p.cin_groups;
  const char* wp = (const char*)p.w + ((int64_t)(cb_ok ? cb : 0) * nchunks * (KS * KS) * 64 + lane) * 16;
  const char* w1p = HAS1X1 ? (const char*)p.w1x1 + ((int64_t)(cb_ok ? cb : 0) * p.n1x1_groups * 64 + lane) * 16 : nullptr;

  Acc8 acc, acc1;
  acc_zero(acc);
  if constexpr (HAS1X1) acc_zero(acc1);

  u32x4 wf[KS * KS], wn[KS * KS];
  u32x4 w1f = {0, 0, 0, 0}, w1n = {0, 0, 0, 0};

  const int64_t in_gs = p.in.group_stride;
  const int n1x1 = HAS1X1 ? p.n1x1_groups : 0;
  auto stage_in = [&](int chunk, int st) {
    const char* src = in_b + (int64_t)chunk * in_gs;
    char* dst = lds_w + st * G::STAGE;
#pragma unroll
    for (int i = 0; i < G::NLD; ++i)   // only the last round can run past the tile
      if (i < G::NLD - 1 || wave * 64 + 256 * i < ((G::NSLOT + 63) / 64) * 64) dma16(src + goff[i], dst + 4096 * i);
  };

  // prologue: chunk 0
  stage_in(0, 0);
#pragma unroll
  for (int tp = 0; tp < KS * KS; ++tp) wf[tp] = *(const u32x4*)(wp + tp * 1024);
  if (HAS1X1) w1f = *(const u32x4*)w1p;
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) {   // prefetch chunk c+1: activations by LDS-DMA, weights into wn
      stage_in(c + 1, (c + 1) & 1);
      const char* wsrc = wp + (int64_t)(c + 1) * (KS * KS) * 1024;
#pragma unroll
      for (int tp = 0; tp < KS * KS; ++tp) wn[tp] = *(const u32x4*)(wsrc + tp * 1024);
      if (HAS1X1 && c + 1 < n1x1) w1n = *(const u32x4*)(w1p + (int64_t)(c + 1) * 1024);
    }

    const char* lds = smem + (c & 1) * G::STAGE;
    const bool do1x1 = HAS1X1 && c < n1x1;
    sfor<G::WIH>([&](auto IR) __attribute__((always_inline)) {
      constexpr int ir = decltype(IR)::value;
      u32x4 bf[KS];
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) bf[kw] = *(const u32x4*)(lds + colofs[kw] + ir * G::IW * 32);
      sfor<KS>([&](auto KW) __attribute__((always_inline)) {
        constexpr int kw = decltype(KW)::value;
        sfor<KS>([&](auto KH) __attribute__((always_inline)) {
          constexpr int kh = decltype(KH)::value;
          if constexpr (UPS) {
            sfor<8>([&](auto R) __attribute__((always_inline)) {
              constexpr int r = decltype(R)::value;
              if constexpr ((((r + kh - 1) >> 1) + 1) == ir) mma<T>(accsel<r>(acc), wf[kh * KS + kw], bf[kw]);
            });
          } else {
            constexpr int tt = ir - kh;
            if constexpr (tt >= 0 && tt % S == 0 && tt / S < 8) mma<T>(accsel<tt / S>(acc), wf[kh * KS + kw], bf[kw]);
          }
        });
        if constexpr (HAS1X1 && !UPS && S == 1 && kw == G::PAD) {   // centre tap feeds the 1x1 residual conv
          constexpr int r = ir - G::PAD;
          if constexpr (r >= 0 && r < 8) {
            if (do1x1) mma<T>(accsel<r>(acc1), w1f, bf[kw]);
          }
        }
      });
    });

    if (more) {
#pragma unroll
      for (int tp = 0; tp < KS * KS; ++tp) wf[tp] = wn[tp];
      if (HAS1X1) w1f = w1n;
    }
    __syncthreads();   // drains the DMA (vmcnt(0)) and fences LDS for the next stage
  }

  if (!cb_ok) return;

  // ---------------------------------------------------------------- epilogue
  float bias[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bias[e] = p.bias ? p.bias[cb * 32 + 16 * h + e] : 0.f;
  const int ox = ox0 + wc * 32 + j;
  if (ox >= p.W) return;
#pragma unroll 1
  for (int r = 0; r < 8; ++r) {
    const int oy = oy0 + wr * 8 + r;
    if (oy >= p.H) break;
    const f32x16 a = pick8(acc, r);
    if constexpr (HAS1X1) {
      const f32x16 a1 = pick8(acc1, r);
      epilogue_row<T>(p, a, &a1, bias, b, cb, h, oy, ox);
    } else {
      epilogue_row<T>(p, a, nullptr, bias, b, cb, h, oy, ox);
    }
  }
}

template <typename T, int KS, int S, bool UPS, int WR, int WC, int NCG, bool HAS1X1>
int launch(const esr_conv& p, hipStream_t st) {
  using G = Geo<KS, S, UPS, WR, WC>;
  const int tiles = ((p.W + G::TW - 1) / G::TW) * ((p.H + G::TH - 1) / G::TH) * p.B;
  dim3 grid(tiles, (p.cout_blocks + NCG - 1) / NCG);
  hipLaunchKernelGGL((conv_kernel<T, KS, S, UPS, WR, WC, NCG, HAS1X1>), grid, dim3(256), 0, st, p);
  return esr_check_launch("conv_kernel");
}

template <typename T>
int dispatch(const esr_conv& p, hipStream_t st) {
  const bool has1 = p.w1x1 != nullptr;
  const int cbk = p.cout_blocks;
  if (p.ks == 3 && p.stride == 1 && !p.upsample) {
    if (has1) {
      if (cbk != 1) { esr_set_error("conv: fused 1x1 needs cout_blocks==1"); return ESR_ERR_UNSUPPORTED; }
      return p.W <= 32 ? launch<T, 3, 1, false, 4, 1, 1, true>(p, st) : launch<T, 3, 1, false, 2, 2, 1, true>(p, st);
    }
    if (cbk == 1) return p.W <= 32 ? launch<T, 3, 1, false, 4, 1, 1, false>(p, st) : launch<T, 3, 1, false, 2, 2, 1, false>(p, st);
    if (cbk < 4 || cbk % 4) return launch<T, 3, 1, false, 2, 1, 2, false>(p, st);
    return launch<T, 3, 1, false, 1, 1, 4, false>(p, st);
  }
  if (has1) { esr_set_error("conv: fused 1x1 only with 3x3/s1"); return ESR_ERR_UNSUPPORTED; }
  if (p.ks == 3 && p.stride == 1 && p.upsample) {
    if ((p.H | p.W) & 1) { esr_set_error("conv: upsample needs even output size"); return ESR_ERR_INVALID; }
    return cbk == 1 ? launch<T, 3, 1, true, 2, 2, 1, false>(p, st) : launch<T, 3, 1, true, 2, 1, 2, false>(p, st);
  }
  if (p.ks == 4 && p.stride == 2 && !p.upsample) {
    return (cbk < 4 || cbk % 4) ? launch<T, 4, 2, false, 1, 2, 2, false>(p, st) : launch<T, 4, 2, false, 1, 1, 4, false>(p, st);
  }
  if (p.ks == 1 && p.stride == 1 && !p.upsample) {
    return cbk == 1 ? launch<T, 1, 1, false, 2, 2, 1, false>(p, st) : launch<T, 1, 1, false, 2, 1, 2, false>(p, st);
  }
  esr_set_error("conv: unsupported ks=%d stride=%d upsample=%d", p.ks, p.stride, p.upsample);
  return ESR_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int esr_conv_forward(const esr_conv* p, esr_stream_t stream) {
  if (!p || !p->in.ptr || !p->w || p->cin_groups <= 0 || p->cout_blocks <= 0 || p->B <= 0 || p->H <= 0 || p->W <= 0) {
    esr_set_error("esr_conv_forward: invalid arguments");
    return ESR_ERR_INVALID;
  }
  if (p->noise_mode == ESR_NOISE_EXPLICIT && !p->z1.ptr && !p->z2.ptr) {
    esr_set_error("esr_conv_forward: explicit noise without z tensors");
    return ESR_ERR_INVALID;
  }
  if (p->mask.ptr && !p->out2.ptr) { esr_set_error("esr_conv_forward: mask without out2"); return ESR_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (p->dtype == ESR_F16) return dispatch<_Float16>(*p, st);
  if (p->dtype == ESR_F32) return dispatch<float>(*p, st);
  esr_set_error("esr_conv_forward: bad dtype %d", p->dtype);
  return ESR_ERR_INVALID;
}
