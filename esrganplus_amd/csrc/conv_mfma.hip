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
//   One wave owns 4 output rows x 32 output pixels x 32*NCW couts = 4*NCW MFMA 32x32 accumulators
//   (v_mfma_f32_32x32x16_f16, or 4x v_mfma_f32_32x32x2_f32 for the exact-fp32 path).  Each
//   B fragment read from LDS feeds up to KS MFMAs (the kh taps of different output rows), each
//   A fragment up to 8 (the rows).
// A workgroup is 8 waves arranged WR x WC spatially x NCG cout-groups (2 waves per SIMD).
#include <type_traits>
#include <utility>

#include <cstdlib>
#include "common.h"

#ifndef ESR_PROBES
#define ESR_PROBES 0   // 1 = extra measurement-only debug flags (tools/mma_probe.py); adds branches to the K loop
#endif
#ifndef ESR_NSA_MAX
#define ESR_NSA_MAX 2   // activation ring depth cap. A/B (profiles/r01_experiments.md): 3 (two K steps ahead) is 1.5 % SLOWER
#endif

#include "mfma_tile.h"

namespace {

// Epilogue of one 32-cout block of a wave (R rows x 1 pixel x 16 couts per lane); operation order
// as documented on esr_conv in esrgan_hip.h.  Two phases: (1) issue EVERY global load of all R rows
// (bias, residuals, explicit z, mask) back to back, (2) compute and store.  With one or two waves
// per SIMD a load->use->store chain per row would expose R full memory latencies.
template <typename T, int R, int NCW, int CW, bool HAS1X1, bool BWD, int RS = 1>
__device__ __forceinline__ void epilogue_block(const esr_conv& p, Acc8& acc, Acc8& acc1, int b, int cb, int h,
                                               int oyb, int ox) {
  const bool n1 = (p.noise_mode == ESR_NOISE_PHILOX && p.layer1 != ESR_NO_LAYER) || (p.noise_mode == ESR_NOISE_EXPLICIT && p.z1.ptr);
  const bool n2 = (p.noise_mode == ESR_NOISE_PHILOX && p.layer2 != ESR_NO_LAYER) || (p.noise_mode == ESR_NOISE_EXPLICIT && p.z2.ptr);
  const bool n3 = (p.noise_mode == ESR_NOISE_PHILOX && p.layer3 != ESR_NO_LAYER) || (p.noise_mode == ESR_NOISE_EXPLICIT && p.z3.ptr);
  const bool xz = p.noise_mode == ESR_NOISE_EXPLICIT;
  f32x4 bq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bq[i] = p.bias ? *(const f32x4*)(p.bias + cb * 32 + 16 * h + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (!BWD && !HAS1X1) {
    // Plain layers (bias + activation, ONE output: the up-convs, HR_conv0, the head conv, HR_conv1's NCHW result, every
    // D / VGG conv in inference): straight-line form without the residual / noise / second-output stages of the general
    // epilogue below (measured on the 512x512 tail: 435 -> 394 us for the 64 -> 64 conv, tools/tail_conv_probe.py).
    // max(x, x * slope) is LeakyReLU (slope 0.2) and the identity (slope 1).
    const bool g32_only = p.out.ptr && p.nchw_out_c <= 0, nchw_only = !p.out.ptr && p.nchw_out_c > 0;
    if (!p.res1.ptr && !p.res2.ptr && !p.aux_out.ptr && p.noise_mode == ESR_NOISE_OFF && (g32_only || nchw_only) &&
        !(p.debug_flags & 0x18)) {
      const bool relu = p.act == ESR_ACT_RELU;
      const float slope = p.act == ESR_ACT_LRELU ? ESR_LRELU_SLOPE : 1.f;
      sfor<R>([&](auto RR) __attribute__((always_inline)) {
        constexpr int r = decltype(RR)::value;
        const int oy = oyb + RS * r;
        if (oy >= p.H) return;
        const f32x16 a = accsel<r * NCW + CW>(acc);
        float v[16];
        if (relu) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = fmaxf(a[e] + bq[e >> 2][e & 3], 0.f);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float x = a[e] + bq[e >> 2][e & 3];
            v[e] = fmaxf(x, x * slope);
          }
        }
        if (g32_only) Px16<T>::store(p.out, b, cb, h, (int64_t)(oy + 1) * p.out.wp + ox + 1, v);
        else {
          const int c0 = cb * 32 + 16 * h;
          float* const o = p.nchw_out + (((int64_t)b * p.nchw_out_c + c0) * p.H + oy) * p.W + ox;
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (c0 + e < p.nchw_out_c) o[(int64_t)e * p.H * p.W] = v[e];
        }
      });
      return;
    }
  }
  if constexpr (BWD && !HAS1X1) {
    // dgrad with nothing but the activation mask (the tail of the generator, every D / VGG dgrad): masked gradient to
    // out2, mask rows fetched up front
    if (p.mask.ptr && p.mask_cb_begin == 0 && !p.out.ptr && !p.res1.ptr && !p.res2.ptr && !p.out3.ptr && !p.aux_out.ptr &&
        !p.bias && p.noise_mode == ESR_NOISE_OFF && p.alpha == 1.0f && p.act == ESR_ACT_NONE && p.nchw_out_c <= 0) {
      Raw16<T> mk[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int oy = oyb + RS * r < p.H ? oyb + RS * r : p.H - 1;
        mk[r].load(p.mask, b, cb, h, (int64_t)(oy + 1) * p.mask.wp + ox + 1);
      }
      const float neg = p.mask_act == ESR_ACT_RELU ? 0.f : ESR_LRELU_SLOPE;
      sfor<R>([&](auto RR) __attribute__((always_inline)) {
        constexpr int r = decltype(RR)::value;
        const int oy = oyb + RS * r;
        if (oy >= p.H) return;
        const f32x16 a = accsel<r * NCW + CW>(acc);
        float m[16], v[16];
        mk[r].get(m);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = m[e] > 0.f ? a[e] : a[e] * neg;
        Px16<T>::store(p.out2, b, cb, h, (int64_t)(oy + 1) * p.out2.wp + ox + 1, v);
      });
      return;
    }
  }
  Raw16<T> r1[R], r2[R];   // prefetched; explicit-z operands (test mode) load in phase 2
  // dgrad: the activation mask rides in r2's registers when there is no second residual (every dgrad conv of the
  // networks here), so its R loads are in flight with the rest instead of one load -> use -> store chain per row
  const bool mask_on = BWD && p.mask.ptr && cb >= p.mask_cb_begin;
  const bool mask_pre = mask_on && !p.res2.ptr;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int oy = oyb + RS * r < p.H ? oyb + RS * r : p.H - 1;     // clamp: rows past the image are not stored
    if (p.res1.ptr) r1[r].load(p.res1, b, cb, h, (int64_t)(oy + 1) * p.res1.wp + ox + 1);
    if (p.res2.ptr) r2[r].load(p.res2, b, cb, h, (int64_t)(oy + 1) * p.res2.wp + ox + 1);
    else if (mask_pre) r2[r].load(p.mask, b, cb - p.mask_cb_begin, h, (int64_t)(oy + 1) * p.mask.wp + ox + 1);
  }
  sfor<R>([&](auto RR) __attribute__((always_inline)) {
    constexpr int r = decltype(RR)::value;
    const int oy = oyb + RS * r;
    if (oy >= p.H) return;
    const f32x16 a = accsel<r * NCW + CW>(acc);
    float v[16], tmp[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float x = a[e] + bq[e >> 2][e & 3];
      if (p.act == ESR_ACT_LRELU) x = x > 0.f ? x : x * ESR_LRELU_SLOPE;
      else if (p.act == ESR_ACT_RELU) x = x > 0.f ? x : 0.f;
      v[e] = x;
    }
    if (p.aux_out.ptr) Px16<T>::store(p.aux_out, b, cb, h, (int64_t)(oy + 1) * p.aux_out.wp + ox + 1, v);
    if constexpr (HAS1X1) {
      const f32x16 a1 = accsel<r>(acc1);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += a1[e];
    }
    if constexpr (BWD) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] *= p.alpha;
      if (p.res1.ptr) {
        r1[r].get(tmp);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += tmp[e];
      }
    } else if (p.res1.ptr) {
      r1[r].get(tmp);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = v[e] * p.alpha + tmp[e];
    }
    const uint32_t pix = (uint32_t)((b * p.H + oy) * p.W + ox);
    if (n1) {
      if (xz) Px16<T>::load(p.z1, b, cb, h, (int64_t)(oy + 1) * p.z1.wp + ox + 1, tmp);
      else {
#pragma unroll
        for (int o = 0; o < 2; ++o) philox_normal8(pix, (uint32_t)(cb * 4 + h * 2 + o), p.layer1, noise_seed(p), &tmp[8 * o]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = v[e] + tmp[e] * (p.sigma * v[e]);   // block.py:119-121
    }
    if (p.res2.ptr) {
      r2[r].get(tmp);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = v[e] * p.beta + tmp[e];
    }
    if (n2) {
      if (xz) Px16<T>::load(p.z2, b, cb, h, (int64_t)(oy + 1) * p.z2.wp + ox + 1, tmp);
      else {
#pragma unroll
        for (int o = 0; o < 2; ++o) philox_normal8(pix, (uint32_t)(cb * 4 + h * 2 + o), p.layer2, noise_seed(p), &tmp[8 * o]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = v[e] + tmp[e] * (p.sigma * v[e]);
    }
    if (p.out.ptr) Px16<T>::store(p.out, b, cb, h, (int64_t)(oy + 1) * p.out.wp + ox + 1, v, (p.debug_flags >> 3) & 3);
    if constexpr (BWD) {
    if (mask_on) {
      const int mb = cb - p.mask_cb_begin;
      if (mask_pre) r2[r].get(tmp);
      else Px16<T>::load(p.mask, b, mb, h, (int64_t)(oy + 1) * p.mask.wp + ox + 1, tmp);
      const float neg = p.mask_act == ESR_ACT_RELU ? 0.f : ESR_LRELU_SLOPE;
#pragma unroll
      for (int e = 0; e < 16; ++e) tmp[e] = tmp[e] > 0.f ? v[e] : v[e] * neg;
      Px16<T>::store(p.out2, b, mb, h, (int64_t)(oy + 1) * p.out2.wp + ox + 1, tmp);
    }
    if (p.out3.ptr) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] *= p.gamma;
      if (n3) {
        if (xz) Px16<T>::load(p.z3, b, cb, h, (int64_t)(oy + 1) * p.z3.wp + ox + 1, tmp);
        else {
#pragma unroll
          for (int o = 0; o < 2; ++o) philox_normal8(pix, (uint32_t)(cb * 4 + h * 2 + o), p.layer3, noise_seed(p), &tmp[8 * o]);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] + tmp[e] * (p.sigma * v[e]);
      }
      Px16<T>::store(p.out3, b, cb, h, (int64_t)(oy + 1) * p.out3.wp + ox + 1, v);
    }
    }
    if (p.nchw_out_c > 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ch = cb * 32 + 16 * h + e;
        if (ch < p.nchw_out_c) p.nchw_out[(((int64_t)b * p.nchw_out_c + ch) * p.H + oy) * p.W + ox] = v[e];
      }
    }
  });
}

// ------------------------------------------------------------------------------------------------
// Tile geometry.  A workgroup = 8 waves = WR x WC spatial waves x NCG cout-groups; every wave owns
// R=4 output rows x 32 output pixels x NCW blocks of 32 couts (R*NCW <= 8 MFMA accumulators).
// LDS holds a ring of activation stages (the halo tile of ONE 32-byte channel group per K step) and
// a 2-deep ring of weight stages (the A fragments of that K step for every cout block of the
// workgroup).  Every wave issues exactly NLD activation DMAs and WLD weight DMAs per K step (tail
// rounds re-copy an earlier window), weights before activations, so a counted
// `s_waitcnt vmcnt(NLD)` leaves exactly the newest activation stage in flight.
// ------------------------------------------------------------------------------------------------
template <int KS, int S, int UPS, int WR, int WC, int NCG, int NCW, bool WLDS, bool HAS1X1, bool PIPE_ = false, int RW = 4, bool PACK_ = false>
struct Geo {
  static constexpr int R = RW;      // output rows per wave (2: half-height tiles, twice the workgroups, for grids
                                    // far below one workgroup per CU — the LR-size training launches)
  static constexpr int NW = WR * WC * NCG;                               // waves per workgroup
  static constexpr int NT = NW * 64;
  static constexpr int TH = R * WR, TW = 32 * WC;                        // output tile
  // PACK (4x4/s2 forward conv on maps of at most 16 output columns — the discriminator behind its third stride-2
  // stage): the tile's 32 output columns are 2 / 4 / 8 IMAGES side by side (and, for 4-row maps, its two row groups
  // two images on top of each other), each with its own halo in the staged tile: [pr][rows][pc][cols].  Sized for the
  // worst case: 2 x 10 rows, 8 x 10 columns.  Always run with the K loop SPLIT over blockIdx.z (see launch_s2_packed).
  static constexpr bool PACK = PACK_;
  // UPS == 3 (sub-pixel up-conv, KS == 2): the tile is counted in INPUT pixels; the 4 cout groups are the 4
  // output phases (dy, dx), each a 2x2 conv on the input shifted by (dy, dx)
  // (the transposed conv — UPS == 2, the input gradient of those layers — packs likewise: 8 / 4 / 2 images across the
  //  tile's 64 output columns, two on top of each other for 8-row maps: 2 x 6 rows, 8 x 6 columns of input)
  static constexpr int IH = PACK_ ? (UPS == 2 ? 12 : 20) : UPS == 3 ? TH + 2 : UPS ? TH / 2 + 2 : (TH - 1) * S + KS;       // staged input tile
  static constexpr int IW = PACK_ ? (UPS == 2 ? 48 : 80) : UPS == 3 ? TW + 2 : UPS ? TW / 2 + 2 : (TW - 1) * S + KS;
  static constexpr int WIH = UPS == 3 ? R + 1 : UPS ? R / 2 + 2 : (R - 1) * S + KS;        // input rows one wave reads
  static constexpr int NSLOT = IH * IW * 2;                              // 16-byte slots (activations)
  static constexpr int NLD = (NSLOT + NT - 1) / NT;                      // activation DMA rounds
  static constexpr int ACT = NLD * NT * 16;                              // bytes (tail lanes -> padding)
  static constexpr int NTAP = KS * KS;
  static constexpr int WWIN = WLDS ? NCG * NCW * NTAP + (HAS1X1 ? 1 : 0) : 0;   // 1 KB weight windows
  static constexpr int WLD = (WWIN + NW - 1) / NW;                        // weight DMA rounds
  static constexpr int WBYTES = WWIN * 1024;                              // weight stage
  // 8-wave workgroups own the CU's LDS; 4-wave workgroups are sized so TWO fit a CU (their barrier
  // and DMA-wait phases then interleave instead of idling the matrix pipe)
  static constexpr int LDS_BUDGET = (NW == 8 ? 160 : 80) * 1024;
  // Two rings: activations (the bulk, MALL/HBM latency ~2-3 us under load) are prefetched TWO K
  // steps ahead when three stages fit; weights (L2-resident) one step ahead in a 2-deep ring.
  static constexpr int NSW = WLDS ? 2 : 0;
  // PIPE: the hand-pipelined K loop (small grids, below) prefetches activations two K steps ahead
  static constexpr bool PIPE = PIPE_;
#ifndef ESR_PIPE_NSA
#define ESR_PIPE_NSA 3   // activation ring depth of the pipelined instantiation (A/B knob)
#endif
  static constexpr int NSA = (((PIPE_ && ESR_PIPE_NSA >= 3) || ESR_NSA_MAX >= 3) && 3 * ACT + NSW * WBYTES <= LDS_BUDGET) ? 3 : 2;
  static constexpr int WOFF = NSA * ACT;                                  // weight ring base
  static constexpr int LDS_BYTES = NSA * ACT + NSW * WBYTES;
  static constexpr int PAD = (KS - 1) / 2;
  static_assert(NW == 8 || NW == 4, "4 or 8 waves per workgroup");
  static_assert(UPS != 3 || (KS == 2 && S == 1 && NCG == 4 && WLDS && !HAS1X1), "sub-pixel up-conv: 2x2 taps, one cout group per phase");
  static_assert(!PACK_ || (KS == 4 && S == 2 && UPS == 0 && WR == 2 && WC == 1 && !WLDS && !HAS1X1 && RW == 4) ||
                    (KS == 4 && S == 1 && UPS == 2 && WR == 4 && WC == 2 && NCG == 1 && WLDS && !HAS1X1 && RW == 4),
                "packed small maps: the 4x4/s2 conv and its transpose");
  static_assert(R * NCW <= 8, "accumulator budget");
  static_assert(LDS_BYTES <= LDS_BUDGET, "LDS budget");
};

// s_waitcnt through the builtin: the compiler's own waitcnt pass SEES these (it does not parse inline
// asm and would re-wait, conservatively with lgkmcnt(0), before the next use).  gfx9 encoding:
// vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
template <int N> __device__ __forceinline__ void wait_vmcnt_b() {
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | ((N >> 4) << 14));
}
__device__ __forceinline__ void wait_lgkm0_b() { __builtin_amdgcn_s_waitcnt(0xC07F); }


template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int KS, int S, int UPS, int WR, int WC, int NCG, int NCW, bool WLDS, bool HAS1X1, bool BWD, bool PIPE = false, int RW = 4, bool PACK = false>
__device__ __forceinline__ void conv_body(const esr_conv& p, char* const smem, const int block_x, const int grid_x,
                                          const int block_y, const int block_z = 0) {
  using G = Geo<KS, S, UPS, WR, WC, NCG, NCW, WLDS, HAS1X1, PIPE, RW, PACK>;
  static_assert(RW == 4 || (KS == 3 && S == 1 && UPS == 0), "half-height tiles: plain 3x3");
  static_assert(!PIPE || (KS == 3 && S == 1 && UPS == 0 && WLDS && NCG == 1), "pipelined K loop: 3x3/s1, LDS weights");
  constexpr int R = G::R;
  constexpr int NSA = G::NSA;
  static_assert(!HAS1X1 || (NCW == 1 && NCG == 1 && KS == 3 && S == 1 && !UPS && WLDS), "fused 1x1 only on the N=32 3x3 conv");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % NCG;
  const int wc = (wave / NCG) % WC;
  const int wr = (wave / NCG) / WC;
  const int j = lane & 31, h = lane >> 5;

  // ---- XCD-aware tile mapping: block b runs on XCD b%8; give every XCD a contiguous run of tiles
  // (= whole images) so the activations it wrote in the previous launch are in ITS L2 / nearby MALL.
  const int tw_ = UPS == 3 ? p.W / 2 : p.W, th_ = UPS == 3 ? p.H / 2 : p.H;   // extent the tiles cover
  const int tiles_x = (tw_ + G::TW - 1) / G::TW, tiles_y = (th_ + G::TH - 1) / G::TH;
  int t;
  {
    const int nwg = grid_x, bid = block_x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
  // PACK: images per tile row (pc_n) / column (pr_n), this lane's image and column inside it
  int pc_n = 1, pr_n = 1, ihs = G::IH, iws = G::IW, lane_pc = 0, lane_x = j, b_lane = 0;
  if constexpr (PACK) {
    t = block_x;                                       // (no XCD renumbering: a handful of tiles)
    if constexpr (UPS == 2) {
      // transposed conv: the tile is 16 rows x 64 columns of OUTPUT (p.H x p.W = 8 / 16 / 32 square); lane j is input
      // column j of the tile's 32, its wave (wc) the output-column parity
      const int wi = p.W >> 1;                         // input columns per image
      pc_n = 32 / wi;
      pr_n = p.H <= 8 ? 2 : 1;
      const int tyn = pr_n == 2 ? 1 : (p.H + G::TH - 1) / G::TH;
      b = (t / tyn) * (pc_n * pr_n);
      tx = 0; ty = t % tyn;
      ihs = pr_n == 2 ? (p.H >> 1) + 2 : G::TH / 2 + 2;
      iws = wi + 2;
      lane_pc = j / wi;
      lane_x = j - lane_pc * wi;
      b_lane = b + (pr_n == 2 ? (wr >> 1) * pc_n : 0) + lane_pc;
    } else {
      pc_n = 32 / p.W;                                   // p.W in {4, 8, 16}
      pr_n = p.H <= 4 ? 2 : 1;
      const int tyn = pr_n == 2 ? 1 : (p.H + G::TH - 1) / G::TH;     // tile rows per image
      b = (t / tyn) * (pc_n * pr_n);                     // first image of this tile
      tx = 0; ty = t % tyn;
      ihs = pr_n == 2 ? (R - 1) * S + KS : G::IH;        // rows of one image's patch in the staged tile (10 | all)
      iws = (p.W - 1) * S + KS;                          // columns of one image's patch
      lane_pc = j / p.W;
      lane_x = j - lane_pc * p.W;
      b_lane = b + (pr_n == 2 ? wr * pc_n : 0) + lane_pc;
    }
  }
  const int cb0 = UPS == 3 ? block_y * NCW : (block_y * NCG + cg) * NCW;     // first 32-cout block of this wave
  const int pdy = UPS == 3 ? cg >> 1 : 0, pdx = UPS == 3 ? cg & 1 : 0;           // sub-pixel phase of this wave
  const int oy0 = ty * G::TH, ox0 = tx * G::TW;      // output tile origin (logical)

  // ---- activation staging map: LDS slot s = tid + NT*i (16 bytes) <- input tile, by LDS-DMA.
  // LDS image [row][col][2 halves]; the two 16-byte halves of pixel `col` are swapped when
  // (col>>3)&1 so the 16 lanes of a ds_read_b128 group cover 16 distinct bank slots.  The DMA
  // destination is lane-linear, so the swizzle is applied to the SOURCE address.
  const int iy0 = UPS == 3 ? oy0 : UPS ? oy0 / 2 : oy0 * S + 1 - G::PAD;   // padded coords of the tile origin
  const int ix0 = UPS == 3 ? ox0 : UPS ? ox0 / 2 : ox0 * S + 1 - G::PAD;
  int goff[G::NLD];
#pragma unroll
  for (int i = 0; i < G::NLD; ++i) {
    int s = tid + G::NT * i;
    if (s >= G::NSLOT) s = G::NSLOT - 1;             // tail lanes: harmless re-copy into the padding
    const int row = s / (2 * G::IW), rem = s - row * 2 * G::IW;
    const int col = rem >> 1, hs = rem & 1, half = hs ^ ((col >> 3) & 1);
    if constexpr (PACK) {
      // LDS (row, col) -> image (pr, pc) of the tile and the pixel inside its patch; slots outside every patch (and
      // images past the batch) re-copy a valid pixel nobody reads
      int pr = row / ihs, ry = row - pr * ihs, pc = col / iws, cx = col - pc * iws;
      if (pr >= pr_n || pc >= pc_n) { pr = 0; pc = 0; ry = 0; cx = 0; }
      int bi = b + pr * pc_n + pc;
      if (bi >= p.B) bi = p.B - 1;
      goff[i] = (bi - b) * (int)p.in.batch_stride + ((iy0 + ry) * p.in.wp + ix0 + cx) * 32 + half * 16;
    } else {
      goff[i] = ((iy0 + row) * p.in.wp + ix0 + col) * 32 + half * 16;
    }
  }
  const char* in_b = (const char*)p.in.ptr + b * p.in.batch_stride;
  const int64_t in_gs = p.in.group_stride;
  char* const lds_wv = smem + wave * 1024;           // this wave's 64-slot DMA window

  // ---- B-fragment (activation) read offsets, one per kw tap
  int colofs[KS];
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    // UPS==1: nearest-x2 gather; UPS==2 (transposed stride-2): this wave owns the 32 output columns
    // of parity wc, tap kw contributes iff (wc+1-kw) is even and then reads g column j+(wc+1-kw)/2
    const int col = UPS == 1 ? (((wc * 32 + j + kw - 1) >> 1) + 1)
                  : UPS == 2 ? (j + ((wc + 1 - kw) >> 1) + 1)
                  : UPS == 3 ? (wc * 32 + j + pdx + kw)
                             : ((wc * 32 + j) * S + kw);
    colofs[kw] = col * 32 + ((h ^ ((col >> 3) & 1)) << 4) + (UPS == 3 ? wr * R + pdy : UPS ? wr * (R / 2) : wr * R * S) * G::IW * 32;
    if constexpr (PACK) {
      if constexpr (UPS == 2) {
        const int pcol = lane_pc * iws + lane_x + ((wc + 1 - kw) >> 1) + 1;
        const int prow = pr_n == 2 ? (wr >> 1) * ihs + (wr & 1) * (R / 2) : wr * (R / 2);
        colofs[kw] = pcol * 32 + ((h ^ ((pcol >> 3) & 1)) << 4) + prow * G::IW * 32;
      } else {
        const int pcol = lane_pc * iws + lane_x * S + kw;            // this lane's image patch, column x * S + kw
        colofs[kw] = pcol * 32 + ((h ^ ((pcol >> 3) & 1)) << 4) + (pr_n == 2 ? wr * ihs : wr * R * S) * G::IW * 32;
      }
    }
  }

  // ---- weights: packed [cout_block][chunk][tap][lane][16 B]
  int nchunks = p.cin_groups;
  const int64_t w_cb_stride = (int64_t)nchunks * G::NTAP * 1024;
  const char* wbase = (const char*)p.w;
  if constexpr (PACK) {
    // split K: this workgroup contracts input chunks [c0, c1) only (esr_conv.ksplit; partial sums to split_ws)
    const int per = (nchunks + p.ksplit - 1) / p.ksplit, c0 = block_z * per;
    const int c1 = c0 + per < nchunks ? c0 + per : nchunks;
    in_b += (int64_t)c0 * p.in.group_stride;
    wbase += (int64_t)c0 * G::NTAP * 1024;
    nchunks = c1 > c0 ? c1 - c0 : 0;
  }
  const int n1x1 = HAS1X1 ? p.n1x1_groups : 0;
  // LDS weight windows this wave copies each K step: window q = wave + 8*i (wrapping -> duplicate)
  const char* wsrc[G::WLD > 0 ? G::WLD : 1];
  int wdst[G::WLD > 0 ? G::WLD : 1];
  int64_t wstep[G::WLD > 0 ? G::WLD : 1];
  if constexpr (WLDS) {
#pragma unroll
    for (int i = 0; i < G::WLD; ++i) {
      const int q = (wave + G::NW * i) % G::WWIN;
      wdst[i] = q * 1024;
      if (HAS1X1 && q == G::WWIN - 1) {              // the fused 1x1's A fragment: [cb][chunk][lane]
        wsrc[i] = (const char*)p.w1x1 + ((int64_t)cb0 * n1x1 * 64 + lane) * 16;
        wstep[i] = 1024;
      } else {
        const int blk = q / G::NTAP;                  // cout block within the workgroup
        int cb = block_y * NCG * NCW + blk;
        if constexpr (UPS == 3) {                     // packed phase-major: [phase][cout_block]
          int rcb = block_y * NCW + blk % NCW;
          if (rcb >= p.cout_blocks) rcb = 0;
          cb = (blk / NCW) * p.cout_blocks + rcb;
        } else if (cb >= p.cout_blocks) cb = 0;
        wsrc[i] = wbase + cb * w_cb_stride + ((q - blk * G::NTAP) * 64 + lane) * 16;
        wstep[i] = (int64_t)G::NTAP * 1024;
      }
    }
  }
  // register-resident weight path (WLDS == false)
  const char* wreg[NCW];
#pragma unroll
  for (int cw = 0; cw < NCW; ++cw) {
    const int cb = cb0 + cw < p.cout_blocks ? cb0 + cw : 0;
    wreg[cw] = wbase + cb * w_cb_stride + lane * 16;
  }

  const int dbg = p.debug_flags;
  Acc8 acc, acc1;
  acc_zero(acc);
  if constexpr (HAS1X1) acc_zero(acc1);

  u32x4 wf[WLDS ? 1 : NCW * G::NTAP], wn[WLDS ? 1 : NCW * G::NTAP];

  auto stage_acts = [&](int chunk, int sa) __attribute__((always_inline)) {
    const char* src = in_b + (int64_t)chunk * in_gs;
    char* dst = lds_wv + sa * G::ACT;
    if (PIPE || !(dbg & 4)) {
#pragma unroll
      for (int i = 0; i < G::NLD; ++i) dma16_act(src + goff[i], dst + G::NT * 16 * i);
    }
  };
  auto stage_wts = [&](int chunk, int sw) __attribute__((always_inline)) {
    if constexpr (WLDS) {
#pragma unroll
      for (int i = 0; i < G::WLD; ++i) {
        const int cc = (HAS1X1 && wstep[i] == 1024 && chunk >= n1x1) ? 0 : chunk;   // 1x1 has fewer K steps
        dma16(wsrc[i] + cc * wstep[i], smem + G::WOFF + sw * G::WBYTES + wdst[i]);
      }
    }
  };

  if constexpr (PIPE) {
    // ------------------------------------------------------------------------------------------
    // Hand-pipelined K loop.  Measured on the compiler-scheduled loop (tools/mma_probe.py): per K
    // step the MFMAs (0.95 us/CU), the LDS fragment reads (+0.5), the DMA issue (+0.3) and the
    // barrier (+0.2) ADD UP — nothing overlaps, because every wave runs the phases
    //   barrier -> issue DMAs -> read fragments -> wait -> MFMAs
    // strictly in that order.  Here each K step c is three MFMA groups (one per column tap kw) and
    // everything else rides BETWEEN the MFMAs of a group:
    //   kw=0: MFMAs(0,c)  || ds_read fragments (1,c)
    //   kw=1: MFMAs(1,c)  || ds_read fragments (2,c)
    //   kw=2: retire step c+1 (own DMAs landed, own reads of step c done), s_barrier,
    //         ds_read fragments (0,c+1), then MFMAs(2,c) || DMA issue of weights c+2, acts c+NSA
    // so the matrix pipe never waits for a barrier, an LDS round trip or a DMA issue.  Fragment
    // registers are double-buffered; with 3 groups per step the buffer parity flips every step,
    // hence the two instantiations P=0/1.  STEADY bodies (all refills in range) are branch-free.
    // ------------------------------------------------------------------------------------------
    u32x4 af[2][KS * NCW], bf[2][G::WIH], a1f;
    auto rd = [&](auto KW, auto BUF, const char* lds, const char* ldw) __attribute__((always_inline)) {
      constexpr int kw = decltype(KW)::value, buf = decltype(BUF)::value;
#pragma unroll
      for (int kh = 0; kh < KS; ++kh)
#pragma unroll
        for (int cw = 0; cw < NCW; ++cw)
          af[buf][kh * NCW + cw] = *(const u32x4*)(ldw + (cw * G::NTAP + kh * KS + kw) * 1024);
#pragma unroll
      for (int ir = 0; ir < G::WIH; ++ir) bf[buf][ir] = *(const u32x4*)(lds + colofs[kw] + ir * G::IW * 32);
    };
    auto mm = [&](auto KW, auto BUF) __attribute__((always_inline)) {
      constexpr int kw = decltype(KW)::value, buf = decltype(BUF)::value;
      sfor<G::WIH>([&](auto IR) __attribute__((always_inline)) {
        constexpr int ir = decltype(IR)::value;
        sfor<KS>([&](auto KH) __attribute__((always_inline)) {
          constexpr int kh = decltype(KH)::value;
          constexpr int r = ir - kh;
          if constexpr (r >= 0 && r < R) {
            sfor<NCW>([&](auto CW) __attribute__((always_inline)) {
              constexpr int cw = decltype(CW)::value;
              mma<T>(accsel<r * NCW + cw>(acc), af[buf][kh * NCW + cw], bf[buf][ir]);
            });
          }
        });
        if constexpr (HAS1X1 && kw == G::PAD) {   // centre tap also feeds the fused 1x1 conv
          constexpr int r = ir - G::PAD;          // (zero A fragment on K steps past its 64 inputs)
          if constexpr (r >= 0 && r < R) mma<T>(accsel<r>(acc1), a1f, bf[buf][ir]);
        }
      });
    };
    constexpr int NMM = R * KS * NCW;                       // MFMAs of one kw group (per T=f16)
    constexpr int NRD = KS * NCW + G::WIH;                  // fragment reads of one kw group
    constexpr int NDM = G::NLD + G::WLD;                    // DMA issues of one K step
    constexpr bool F16 = true;                              // group-interleave both dtypes
    constexpr int MPER = sizeof(T) == 2 ? 1 : 4;            // MFMA instructions per mma<T>()
    const char* const ldw0 = smem + G::WOFF + lane * 16;
    auto load_a1f = [&](int c, const char* ldwb) __attribute__((always_inline)) {
      if constexpr (HAS1X1) {
        a1f = *(const u32x4*)(ldwb + (G::WWIN - 1) * 1024);
        if (c >= n1x1) a1f = u32x4{0, 0, 0, 0};
      }
    };

    // prologue: step 0 (and 1) in flight, retire step 0, refill, first fragments
    stage_wts(0, 0);
    stage_acts(0, 0);
    if constexpr (NSA == 3) { if (nchunks > 1) stage_acts(1, 1); }
    if (NSA == 3 && nchunks > 1) wait_vmcnt<G::NLD>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (nchunks > 1) stage_wts(1, 1);
    if (NSA - 1 < nchunks) stage_acts(NSA - 1, NSA - 1);
    rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, smem, ldw0);
    load_a1f(0, ldw0);

    auto kstep = [&](auto PAR, auto STEADY, int c, int st) __attribute__((always_inline)) {
      constexpr int par = decltype(PAR)::value;
      constexpr bool steady = decltype(STEADY)::value;
      const char* lds = smem + st * G::ACT;
      const char* ldw = ldw0 + (c & 1) * G::WBYTES;
      sfor<KS>([&](auto KW) __attribute__((always_inline)) {
        constexpr int kw = decltype(KW)::value;
        constexpr int buf = (kw + par) & 1;
        using BUF = std::integral_constant<int, buf>;
        using NBUF = std::integral_constant<int, buf ^ 1>;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kw + 1 < KS) {
          rd(std::integral_constant<int, kw + 1>{}, NBUF{}, lds, ldw);
          mm(KW, BUF{});
          if constexpr (F16) {
            constexpr int NP = NRD < NMM ? NRD : NMM;
            sfor<NP>([&](auto) __attribute__((always_inline)) {
              __builtin_amdgcn_sched_group_barrier(0x008, MPER, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            });
            if constexpr (NRD > NP) __builtin_amdgcn_sched_group_barrier(0x100, NRD - NP, 0);
            if constexpr (NMM > NP) __builtin_amdgcn_sched_group_barrier(0x008, (NMM - NP) * MPER, 0);
          }
        } else {
          int sn = st + 1; if (sn == NSA) sn = 0;               // ring slot of step c+1
          const char* ldsn = smem + sn * G::ACT;
          const char* ldwn = ldw0 + ((c + 1) & 1) * G::WBYTES;
          if (steady || c + 1 < nchunks) {
            // retire step c+1: my fragment reads of step c are done (their slots get refilled
            // below), my DMAs of step c+1 have landed; only acts c+2 may still be in flight
            wait_lgkm0_b();
            if (NSA == 3 && (steady || c + 2 < nchunks)) wait_vmcnt_b<G::NLD>(); else wait_vmcnt_b<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            rd(std::integral_constant<int, 0>{}, NBUF{}, ldsn, ldwn);
            load_a1f(c + 1, ldwn);
            if (steady || c + 2 < nchunks) stage_wts(c + 2, c & 1);
            if (steady || c + NSA < nchunks) stage_acts(c + NSA, st);
          }
          mm(KW, BUF{});
          if constexpr (F16 && steady) {
            __builtin_amdgcn_sched_group_barrier(0x100, NRD + (HAS1X1 ? 1 : 0), 0);
            constexpr int NP = NDM < NMM ? NDM : NMM;
            sfor<NP>([&](auto) __attribute__((always_inline)) {
              __builtin_amdgcn_sched_group_barrier(0x008, MPER, 0);
              __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            });
            if constexpr (NDM > NP) __builtin_amdgcn_sched_group_barrier(0x010, NDM - NP, 0);
            if constexpr (NMM > NP) __builtin_amdgcn_sched_group_barrier(0x008, (NMM - NP) * MPER, 0);
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    };

    int c = 0, st = 0;
    const int nsteady = nchunks - NSA;          // steps whose refills (w c+2, acts c+NSA) are all in range
    for (; c + 1 < nsteady; c += 2) {
      kstep(std::integral_constant<int, 0>{}, std::true_type{}, c, st);
      if (++st == NSA) st = 0;
      kstep(std::integral_constant<int, 1>{}, std::true_type{}, c + 1, st);
      if (++st == NSA) st = 0;
    }
    for (; c < nchunks; c += 2) {
      kstep(std::integral_constant<int, 0>{}, std::false_type{}, c, st);
      if (++st == NSA) st = 0;
      if (c + 1 < nchunks) {
        kstep(std::integral_constant<int, 1>{}, std::false_type{}, c + 1, st);
        if (++st == NSA) st = 0;
      }
    }
  } else {
  // prologue: weights of step 0, activations of steps 0 .. NSA-2
    stage_wts(0, 0);
    stage_acts(0, 0);
    if constexpr (NSA == 3) { if (nchunks > 1) stage_acts(1, 1); }
    if constexpr (!WLDS) {
  #pragma unroll
      for (int cw = 0; cw < NCW; ++cw)
  #pragma unroll
        for (int tp = 0; tp < G::NTAP; ++tp) wf[cw * G::NTAP + tp] = *(const u32x4*)(wreg[cw] + tp * 1024);
    }
  
    int st = 0;   // activation ring slot of K step c
    for (int c = 0; c < nchunks; ++c) {
      // ---- retire K step c.  In-order return: everything older than the NLD activation DMAs of
      // step c+1 (issued last) has landed once vmcnt <= NLD.
      if constexpr (WLDS) {
        if (NSA == 3 && c + 1 < nchunks) wait_vmcnt<G::NLD>(); else wait_vmcnt<0>();
      } else {
        wait_vmcnt<0>();   // mixed VGPR loads + DMA: plain drain
      }
      __builtin_amdgcn_s_barrier();   // step c visible to all waves; all waves done reading step c-1
      // ---- refill: weights of step c+1 first, then activations of step c+NSA-1
      if (c + 1 < nchunks) stage_wts(c + 1, (c + 1) & 1);
      const int cn = c + NSA - 1;
      if (cn < nchunks) {
        int sn = st + NSA - 1; if (sn >= NSA) sn -= NSA;
        stage_acts(cn, sn);
      }
      if constexpr (!WLDS) {
        if (c + 1 < nchunks) {
  #pragma unroll
          for (int cw = 0; cw < NCW; ++cw)
  #pragma unroll
            for (int tp = 0; tp < G::NTAP; ++tp)
              wn[cw * G::NTAP + tp] = *(const u32x4*)(wreg[cw] + (int64_t)(c + 1) * G::NTAP * 1024 + tp * 1024);
        }
      }
  
      const char* lds = smem + st * G::ACT;
      const char* ldwb = smem + G::WOFF + (c & 1) * G::WBYTES;
      const char* ldw = ldwb + ((cg * NCW) * G::NTAP * 64 + lane) * 16;   // this wave's A fragments
      const bool do1x1 = HAS1X1 && c < n1x1;
  
      // kw-major: the KS*NCW A fragments of column tap kw stay in registers while the wave walks its
      // WIH input rows; each B fragment read feeds up to KS*NCW MFMAs.  Fragments of tap kw+1 are
      // read from LDS while the MFMAs of tap kw run.
      u32x4 af[2][KS * NCW], bf[2][G::WIH], a1f;
      auto read_step = [&](auto KW, u32x4 (&a)[KS * NCW], u32x4 (&bq)[G::WIH]) __attribute__((always_inline)) {
        constexpr int kw = decltype(KW)::value;
  #if ESR_PROBES
        if (dbg & 32) return;   // measurement-only: MFMAs on stale fragments, no LDS reads
  #endif
  #pragma unroll
        for (int kh = 0; kh < KS; ++kh)
  #pragma unroll
          for (int cw = 0; cw < NCW; ++cw) {
            if constexpr (WLDS) a[kh * NCW + cw] = *(const u32x4*)(ldw + (cw * G::NTAP + kh * KS + kw) * 1024);
            else a[kh * NCW + cw] = wf[cw * G::NTAP + kh * KS + kw];
          }
  #pragma unroll
        for (int ir = 0; ir < G::WIH; ++ir) bq[ir] = *(const u32x4*)(lds + colofs[kw] + ir * G::IW * 32);
      };
      if (!(dbg & 2)) {
        read_step(std::integral_constant<int, 0>{}, af[0], bf[0]);
        if constexpr (HAS1X1) a1f = *(const u32x4*)(ldwb + ((G::WWIN - 1) * 64 + lane) * 16);
        sfor<KS>([&](auto KW) __attribute__((always_inline)) {
          constexpr int kw = decltype(KW)::value;
          constexpr int cur = kw & 1;
          if constexpr (kw + 1 < KS) read_step(std::integral_constant<int, kw + 1>{}, af[cur ^ 1], bf[cur ^ 1]);
          sfor<G::WIH>([&](auto IR) __attribute__((always_inline)) {
            constexpr int ir = decltype(IR)::value;
            sfor<KS>([&](auto KH) __attribute__((always_inline)) {
              constexpr int kh = decltype(KH)::value;
              if constexpr (UPS == 1) {
                sfor<R>([&](auto RR) __attribute__((always_inline)) {
                  constexpr int r = decltype(RR)::value;
                  if constexpr ((((r + kh - 1) >> 1) + 1) == ir) {
                    sfor<NCW>([&](auto CW) __attribute__((always_inline)) {
                      constexpr int cw = decltype(CW)::value;
                      mma<T>(accsel<r * NCW + cw>(acc), af[cur][kh * NCW + cw], bf[cur][ir]);
                    });
                  }
                });
              } else if constexpr (UPS == 2) {
                // adjoint of the 4x4/s2/p1 conv: output row r takes tap kh from g row (r+1-kh)/2
                constexpr int r = 2 * (ir - 1) + kh - 1;
                if constexpr (r >= 0 && r < R) {
                  if (((wc + 1 - kw) & 1) == 0) {
                    sfor<NCW>([&](auto CW) __attribute__((always_inline)) {
                      constexpr int cw = decltype(CW)::value;
                      mma<T>(accsel<r * NCW + cw>(acc), af[cur][kh * NCW + cw], bf[cur][ir]);
                    });
                  }
                }
              } else {
                constexpr int tt = ir - kh;
                if constexpr (tt >= 0 && tt % S == 0 && tt / S < R) {
                  sfor<NCW>([&](auto CW) __attribute__((always_inline)) {
                    constexpr int cw = decltype(CW)::value;
                    mma<T>(accsel<(tt / S) * NCW + cw>(acc), af[cur][kh * NCW + cw], bf[cur][ir]);
                  });
                }
              }
            });
            if constexpr (HAS1X1 && kw == G::PAD) {   // centre tap also feeds the fused 1x1 conv
              constexpr int r = ir - G::PAD;
              if constexpr (r >= 0 && r < R) {
                if (do1x1) mma<T>(accsel<r>(acc1), a1f, bf[cur][ir]);
              }
            }
          });
        });
      }
  
      if constexpr (!WLDS) {
        if (c + 1 < nchunks) {
  #pragma unroll
          for (int tp = 0; tp < NCW * G::NTAP; ++tp) wf[tp] = wn[tp];
        }
      }
      if (++st == NSA) st = 0;
    }
  
  }

  // ---------------------------------------------------------------- epilogue
  if constexpr (PACK) {
    // partial sums of this K range, fp32: split_ws[split][image][oy][ox][cout] (cout padded to whole 32-blocks) — plain
    // stores, one slab per split; the finishing launch adds the slabs up in split order (deterministic)
    if (lane_pc >= pc_n || b_lane >= p.B) return;
    const int CP = p.cout_blocks * 32;
    const int oyb_ = UPS == 2 ? (pr_n == 2 ? (wr & 1) * R : oy0 + wr * R) : (pr_n == 2 ? 0 : oy0 + wr * R);
    const int ox_ = UPS == 2 ? 2 * lane_x + wc : lane_x;
    sfor<NCW>([&](auto CW) __attribute__((always_inline)) {
      constexpr int cw = decltype(CW)::value;
      if (cb0 + cw >= p.cout_blocks) return;
      float* const base = p.split_ws + ((int64_t)block_z * p.B + b_lane) * p.H * p.W * CP + (cb0 + cw) * 32 + 16 * h;
      sfor<R>([&](auto RR) __attribute__((always_inline)) {
        constexpr int r = decltype(RR)::value;
        const int oy = oyb_ + r;
        if (oy >= p.H) return;
        const f32x16 a = accsel<r * NCW + cw>(acc);
        float* const o = base + ((int64_t)oy * p.W + ox_) * CP;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)(o + 4 * q) = f32x4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
      });
    });
    return;
  }
  const int ox = UPS == 2 ? ox0 + 2 * j + wc : UPS == 3 ? 2 * (ox0 + wc * 32 + j) + pdx : ox0 + wc * 32 + j;
  if (ox >= p.W || (dbg & 1)) return;
  const int oyb = UPS == 3 ? 2 * (oy0 + wr * R) + pdy : oy0 + wr * R;
  sfor<NCW>([&](auto CW) __attribute__((always_inline)) {
    constexpr int cw = decltype(CW)::value;
    const int cb = cb0 + cw;
    if (cb >= p.cout_blocks) return;
    epilogue_block<T, R, NCW, cw, HAS1X1, BWD, (UPS == 3 ? 2 : 1)>(p, acc, acc1, b, cb, h, oyb, ox);
  });
}

template <typename T, int KS, int S, int UPS, int WR, int WC, int NCG, int NCW, bool WLDS, bool HAS1X1, bool BWD, bool PIPE = false, int RW = 4, bool PACK = false>
__global__ __launch_bounds__(WR * WC * NCG * 64, 2) void conv_kernel(const esr_conv p) {
  using G = Geo<KS, S, UPS, WR, WC, NCG, NCW, WLDS, HAS1X1, PIPE, RW, PACK>;
  __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];
  conv_body<T, KS, S, UPS, WR, WC, NCG, NCW, WLDS, HAS1X1, BWD, PIPE, RW, PACK>(p, smem, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z);
}

// ---- the deep 4x4/s2 convs of the discriminators (maps of 4 / 8 / 16 output columns, 256-512 channels) ----------------
// One image per 8x32 tile leaves these launches 50-94 % padding AND makes every tile's workgroup stream the layer's
// whole weight tensor (128 workgroups x 2.1 MB = 268 MB of L2 -> register traffic for a 4 GFLOP layer: 137 us).  Packing
// 2-16 images into a tile removes both, but leaves 8-32 workgroups with a serial K loop of 32 steps; so the packed
// launch also SPLITS K over blockIdx.z — every workgroup contracts a few input chunks into an fp32 slab — and a
// finishing launch adds the slabs up in split order, applies bias / activation, stores the fp16 G32 output and, when
// asked, accumulates the BatchNorm statistics of that output (replacing the ESR_BN_STATS pass that would follow).
template <typename T>
__global__ __launch_bounds__(256) void s2_finish_kernel(const esr_conv p) {
  // thread = (pixel, 16-channel group); grid (ceil(H W / 256), groups, B)
  const int g = blockIdx.y, b = blockIdx.z;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const bool ok = pix < p.H * p.W;
  const int CP = p.cout_blocks * 32, c0 = g * 16;
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = 0.f;
  if (ok) {
    for (int s = 0; s < p.ksplit; ++s) {
      const float* q = p.split_ws + (((int64_t)s * p.B + b) * p.H * p.W + pix) * CP + c0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 a = *(const f32x4*)(q + 4 * k);
        v[4 * k] += a[0]; v[4 * k + 1] += a[1]; v[4 * k + 2] += a[2]; v[4 * k + 3] += a[3];
      }
    }
    half8 x, y;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float t = v[e] + (p.bias ? p.bias[c0 + e] : 0.f);
      if (p.act == ESR_ACT_LRELU) t = fmaxf(t, t * ESR_LRELU_SLOPE);
      else if (p.act == ESR_ACT_RELU) t = fmaxf(t, 0.f);
      const _Float16 hq = (_Float16)t;
      if (e < 8) x[e] = hq; else y[e - 8] = hq;
      v[e] = (float)hq;                                  // the statistics see what the apply pass will read
    }
    if (g < p.out.ngroups) {
      const int oy = pix / p.W, ox = pix - oy * p.W;
      char* o = (char*)p.out.ptr + b * p.out.batch_stride + (int64_t)g * p.out.group_stride + ((int64_t)(oy + 1) * p.out.wp + ox + 1) * 32;
      *(u32x4*)o = __builtin_bit_cast(u32x4, x);
      *(u32x4*)(o + 16) = __builtin_bit_cast(u32x4, y);
    }
  }
  if (!p.stat_sums) return;
  // BatchNorm statistics (ESR_BN_STATS): per channel sum / sum of squares over the group's images -> fp64 atomics
  float s0[16], s1[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { s0[e] = ok ? v[e] : 0.f; s1[e] = ok ? v[e] * v[e] : 0.f; }
  __shared__ float red[4][2][16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0[e] += __shfl_xor(s0[e], o); s1[e] += __shfl_xor(s1[e], o); }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) { red[wave][0][e] = s0[e]; red[wave][1][e] = s1[e]; }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int k = threadIdx.x >> 4, e = threadIdx.x & 15, c = c0 + e;
    if (c < p.stat_C) {
      const int ngrp = p.stat_groups > 1 ? p.stat_groups : 1, grp = b / (p.B / ngrp);
      atomicAdd(p.stat_sums + (int64_t)grp * 2 * p.stat_C + k * p.stat_C + c,
                (double)red[0][k][e] + red[1][k][e] + red[2][k][e] + red[3][k][e]);
    }
  }
}

// the transposed 4x4/s2 conv (esr_conv.upsample == 2: input gradient of those layers) on output maps of 8 / 16 / 32 columns
template <typename T, int NCW>
int launch_ts2_packed(const esr_conv& p, hipStream_t st) {
  using G = Geo<4, 1, 2, 4, 2, 1, NCW, true, false, false, 4, true>;
  const int per_tile = (64 / p.W) * (p.H <= 8 ? 2 : 1);
  const int tyn = p.H <= 8 ? 1 : (p.H + G::TH - 1) / G::TH;
  dim3 grid(((p.B + per_tile - 1) / per_tile) * tyn, (p.cout_blocks + NCW - 1) / NCW, p.ksplit);
  hipLaunchKernelGGL((conv_kernel<T, 4, 1, 2, 4, 2, 1, NCW, true, false, false, false, 4, true>), grid, dim3(G::NT), 0, st, p);
  int rc = esr_check_launch("conv_kernel (packed, split-K transposed 4x4/s2)");
  if (rc) return rc;
  dim3 fgrid((p.H * p.W + 255) / 256, p.cout_blocks * 2, p.B);
  hipLaunchKernelGGL(s2_finish_kernel<T>, fgrid, dim3(256), 0, st, p);
  return esr_check_launch("s2_finish_kernel");
}

template <typename T>
int launch_s2_packed(const esr_conv& p, hipStream_t st) {
  using G = Geo<4, 2, 0, 2, 1, 4, 1, false, false, false, 4, true>;
  const int per_tile = (32 / p.W) * (p.H <= 4 ? 2 : 1);
  const int tyn = p.H <= 4 ? 1 : (p.H + G::TH - 1) / G::TH;
  dim3 grid(((p.B + per_tile - 1) / per_tile) * tyn, (p.cout_blocks + 3) / 4, p.ksplit);
  hipLaunchKernelGGL((conv_kernel<T, 4, 2, 0, 2, 1, 4, 1, false, false, false, false, 4, true>), grid, dim3(G::NT), 0, st, p);
  int rc = esr_check_launch("conv_kernel (packed, split-K 4x4/s2)");
  if (rc) return rc;
  dim3 fgrid((p.H * p.W + 255) / 256, p.cout_blocks * 2, p.B);
  hipLaunchKernelGGL(s2_finish_kernel<T>, fgrid, dim3(256), 0, st, p);
  return esr_check_launch("s2_finish_kernel");
}

template <typename T, int KS, int S, int UPS, int WR, int WC, int NCG, int NCW, bool WLDS, bool HAS1X1, bool PIPE = false, int RW = 4>
int launch(const esr_conv& p, hipStream_t st) {
  using G = Geo<KS, S, UPS, WR, WC, NCG, NCW, WLDS, HAS1X1, PIPE, RW>;
  const int tw_ = UPS == 3 ? p.W / 2 : p.W, th_ = UPS == 3 ? p.H / 2 : p.H;
  const int tiles = ((tw_ + G::TW - 1) / G::TW) * ((th_ + G::TH - 1) / G::TH) * p.B;
  dim3 grid(tiles, UPS == 3 ? (p.cout_blocks + NCW - 1) / NCW : (p.cout_blocks + NCG * NCW - 1) / (NCG * NCW));
  // the backward-chain epilogue stages (always-on alpha, partial residual views, mask/out2, out3)
  // live in their own instantiation so the forward kernels stay lean
  constexpr int GPB = DT<T>::GPB;
  const bool bwd = p.mask.ptr || p.out3.ptr || (!p.res1.ptr && p.alpha != 1.0f) ||
                   (p.res1.ptr && p.res1.ngroups < p.cout_blocks * GPB && p.res1.ngroups < p.out.ngroups) ||
                   (p.res2.ptr && p.res2.ngroups < p.cout_blocks * GPB && p.res2.ngroups < p.out.ngroups);
  if (bwd) hipLaunchKernelGGL((conv_kernel<T, KS, S, UPS, WR, WC, NCG, NCW, WLDS, HAS1X1, true, PIPE, RW>), grid, dim3(G::NT), 0, st, p);
  else hipLaunchKernelGGL((conv_kernel<T, KS, S, UPS, WR, WC, NCG, NCW, WLDS, HAS1X1, false, PIPE, RW>), grid, dim3(G::NT), 0, st, p);
  return esr_check_launch("conv_kernel");
}

template <typename T>
int dispatch(const esr_conv& p, hipStream_t st) {
  const bool has1 = p.w1x1 != nullptr;
  const int cbk = p.cout_blocks;
  // Tile shapes.  Up to 96 couts: 4-wave workgroups on a 16x32 tile, TWO resident per CU so their
  // barrier / DMA-wait phases interleave (measured +3.6 % over one 8-wave 16x64 workgroup, and twice
  // the workgroup count on small images).  Wider convs (discriminator, VGG): 8 waves = 2 spatial x 4
  // cout groups with register-resident weights.
  if (p.ks == 3 && p.stride == 1 && !p.upsample) {
    const int64_t tiles = (int64_t)((p.W + 31) / 32) * ((p.H + 15) / 16) * p.B;
    // Grids far below one workgroup per CU (16 LR crops of 32x32 = 32 tiles of 16x32: the training launches):
    // tiles of 4 or 8 rows instead of 16 — 4x / 2x the workgroups, a quarter / half of the MFMA chain per wave and
    // K step.  Thresholds on the 16-row grid size; ESR_TILE_ROWS=4 (or debug_flags bit 8) keeps 16-row tiles.
    static const int q1 = [] { const char* e = getenv("ESR_TILE_Q1"); return e ? atoi(e) : 128; }();
    static const int q2 = [] { const char* e = getenv("ESR_TILE_Q2"); return e ? atoi(e) : 384; }();
    static const int rows_forced = [] { const char* e = getenv("ESR_TILE_ROWS"); return e ? atoi(e) : 0; }();
    int rw = tiles * cbk <= q1 ? 1 : (tiles * cbk <= q2 ? 2 : 4);
    // maps of at most 8 (4) rows: the lower half (three quarters) of a 16-row tile is padding whatever the grid
    // size — the VGG / discriminator layers behind four or five poolings (ESR_TILE_HCLAMP=0 switches it off)
    static const bool hclamp = [] { const char* e = getenv("ESR_TILE_HCLAMP"); return !e || atoi(e) != 0; }();
    if (hclamp && p.H <= 4) rw = 1;
    else if (hclamp && p.H <= 8 && rw > 2) rw = 2;
    if (rows_forced) rw = rows_forced;
    if ((p.debug_flags & 256) || sizeof(T) != 2) rw = 4;
    if (has1 && cbk != 1) { esr_set_error("conv: fused 1x1 needs cout_blocks==1"); return ESR_ERR_UNSUPPORTED; }
    if constexpr (sizeof(T) == 2) {
      if (rw == 1) {
        if (has1) return launch<T, 3, 1, 0, 4, 1, 1, 1, true, true, false, 1>(p, st);
        if (cbk == 1) return launch<T, 3, 1, 0, 4, 1, 1, 1, true, false, true, 1>(p, st);
        return launch<T, 3, 1, 0, 4, 1, 1, 1, true, false, false, 1>(p, st);
      }
      if (rw == 2) {
        if (has1) return launch<T, 3, 1, 0, 4, 1, 1, 1, true, true, false, 2>(p, st);
        if (cbk == 1) return launch<T, 3, 1, 0, 4, 1, 1, 1, true, false, true, 2>(p, st);
        return launch<T, 3, 1, 0, 4, 1, 1, 1, true, false, false, 2>(p, st);
      }
    }
    if (has1) return launch<T, 3, 1, 0, 4, 1, 1, 1, true, true>(p, st);
    if (cbk == 1) {
      // Few workgroups (training tiles: at most one per CU): nothing else hides a wave's barrier / LDS
      // round trip / DMA issue, so the hand-pipelined K loop pays (0.87 -> ~0.55 us per K step);
      // with two lock-stepped workgroups per CU streaming at the fabric limit it does not (r01_experiments.md)
      // debug_flags bit 6 / bit 7 select the instantiation per call (plain loop / pipelined), so that every
      // golden test can run both; default: pipelined up to 256 tiles
      const bool pipe = (p.debug_flags & 128) ? true : ((p.debug_flags & 64) ? false : tiles <= 256);
      if (pipe) return launch<T, 3, 1, 0, 4, 1, 1, 1, true, false, true>(p, st);
      return launch<T, 3, 1, 0, 4, 1, 1, 1, true, false>(p, st);
    }
    // Several cout blocks: grid.y walks them, ONE per workgroup on small grids (training tiles: twice
    // the workgroups, half the MFMA chain each), TWO per wave otherwise (each B fragment feeds two
    // MFMAs).  The 8-wave 2x4 register-weight kernel this used to switch to at >= 4 blocks lost on
    // every shape measured (tools/wide_probe.py: 32->128 ... 256->256: 426-928 vs 474-1192 TF/s).
    if (tiles * ((cbk + 1) / 2) < 256) return launch<T, 3, 1, 0, 4, 1, 1, 1, true, false>(p, st);
    return launch<T, 3, 1, 0, 4, 1, 1, 2, true, false>(p, st);
  }
  if (has1) { esr_set_error("conv: fused 1x1 only with 3x3/s1"); return ESR_ERR_UNSUPPORTED; }
  if (p.ks == 4 && p.stride == 1 && p.upsample == 2) {
    if ((p.H | p.W) & 1) { esr_set_error("conv: transposed stride-2 needs even output size"); return ESR_ERR_INVALID; }
    if (p.ksplit > 1) {
      if (sizeof(T) != 2 || !(p.W == 8 || p.W == 16 || p.W == 32) || p.H != p.W || !p.split_ws || !p.out.ptr || p.res1.ptr ||
          p.res2.ptr || p.aux_out.ptr || p.mask.ptr || p.out3.ptr || p.noise_mode != ESR_NOISE_OFF || p.nchw_out_c > 0 ||
          p.alpha != 1.0f || p.ksplit > p.cin_groups || p.bias || p.act != ESR_ACT_NONE || p.stat_sums) {
        esr_set_error("conv: ksplit (transposed) needs a plain fp16 layer on an 8 / 16 / 32-column square output map");
        return ESR_ERR_UNSUPPORTED;
      }
      return cbk == 1 ? launch_ts2_packed<T, 1>(p, st) : launch_ts2_packed<T, 2>(p, st);
    }
    // several cout blocks: ONE per workgroup while that still leaves the grid below one workgroup per CU (the input
    // gradients of the discriminators' deep stride-2 layers at training batches: 128 workgroups streamed 512 x 512 x 16
    // weights for two blocks each — 75 us; as the 3x3 dispatch above does), two per wave otherwise
    const int64_t tiles_t = (int64_t)((p.W + Geo<4, 1, 2, 4, 2, 1, 2, true, false, false, 4>::TW - 1) / Geo<4, 1, 2, 4, 2, 1, 2, true, false, false, 4>::TW) *
                            ((p.H + Geo<4, 1, 2, 4, 2, 1, 2, true, false, false, 4>::TH - 1) / Geo<4, 1, 2, 4, 2, 1, 2, true, false, false, 4>::TH) * p.B;
    static const bool ts2_one = [] { const char* e = getenv("ESR_TS2_ONE"); return !e || atoi(e) != 0; }();
    if (cbk == 1 || (ts2_one && tiles_t * ((cbk + 1) / 2) < 256)) return launch<T, 4, 1, 2, 4, 2, 1, 1, true, false>(p, st);
    return launch<T, 4, 1, 2, 4, 2, 1, 2, true, false>(p, st);
  }
  if (p.ks == 3 && p.stride == 1 && p.upsample == 1) {
    if ((p.H | p.W) & 1) { esr_set_error("conv: upsample needs even output size"); return ESR_ERR_INVALID; }
    return cbk == 1 ? launch<T, 3, 1, 1, 4, 1, 1, 1, true, false>(p, st) : launch<T, 3, 1, 1, 4, 1, 1, 2, true, false>(p, st);
  }
  if (p.ks == 2 && p.stride == 1 && p.upsample == 3) {
    // sub-pixel form of nearest-x2 + 3x3 (weights packed with esr_pack.ups_fwd): 8 waves = 2 row groups x 4 phases
    if ((p.H | p.W) & 1) { esr_set_error("conv: upsample needs even output size"); return ESR_ERR_INVALID; }
    // (one cout block per wave with two workgroups per CU measured slower: 0.49 vs 0.43 ms for both up-convs)
    return cbk == 1 ? launch<T, 2, 1, 3, 2, 1, 4, 1, true, false>(p, st) : launch<T, 2, 1, 3, 2, 1, 4, 2, true, false>(p, st);
  }
  if (p.ks == 4 && p.stride == 2 && !p.upsample) {
    // 8 waves = 2 row groups x 4 cout groups; with at most two cout blocks (the up-convs' adjoint, the
    // discriminator's first stride-2 conv) half of those would idle: 2 row x 2 column x 2 cout groups on an 8x64 tile
    static const bool wide = [] { const char* e = getenv("ESR_S2_WIDE"); return !e || atoi(e) != 0; }();
    if (wide && cbk <= 2 && p.W > 32) return launch<T, 4, 2, 0, 2, 2, 2, 1, false, false>(p, st);
    if (p.ksplit > 1) {
      // packed small maps + split K (launch_s2_packed): the caller sized split_ws for it (esr_conv_split_ws_floats)
      if (sizeof(T) != 2 || !(p.W == 4 || p.W == 8 || p.W == 16) || p.H != p.W || !p.split_ws || !p.out.ptr || p.res1.ptr ||
          p.res2.ptr || p.aux_out.ptr || p.mask.ptr || p.out3.ptr || p.noise_mode != ESR_NOISE_OFF || p.nchw_out_c > 0 ||
          p.alpha != 1.0f || p.ksplit > p.cin_groups) {
        esr_set_error("conv: ksplit needs a plain fp16 4x4/s2 layer on a 4 / 8 / 16-column square map");
        return ESR_ERR_UNSUPPORTED;
      }
      return launch_s2_packed<T>(p, st);
    }
    return launch<T, 4, 2, 0, 2, 1, 4, 1, false, false>(p, st);
  }
  if (p.ks == 1 && p.stride == 1 && !p.upsample) {
    return cbk == 1 ? launch<T, 1, 1, 0, 4, 1, 1, 1, true, false>(p, st) : launch<T, 1, 1, 0, 4, 1, 1, 2, true, false>(p, st);
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
  if (p->mask.ptr && !p->out2.ptr) { esr_set_error("esr_conv_forward: mask without out2"); return ESR_ERR_INVALID; }
  if (p->stat_sums && p->ksplit <= 1) { esr_set_error("esr_conv_forward: stat_sums rides on the split-K finishing pass (ksplit > 1)"); return ESR_ERR_INVALID; }

  hipStream_t st = (hipStream_t)stream;
  if (p->dtype == ESR_F16) return dispatch<_Float16>(*p, st);
  if (p->dtype == ESR_F32) return dispatch<float>(*p, st);
  esr_set_error("esr_conv_forward: bad dtype %d", p->dtype);
  return ESR_ERR_INVALID;
}
