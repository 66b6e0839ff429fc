// wgrad.hip — weight / bias gradients of the fused convolutions (autograd backward of nn.Conv2d:
// the reference gets it implicitly from `l_g_total.backward()`, SRRaGAN_model.py:140,167).
//
//   dW[co][ci][kh][kw] += scale * sum_{b,y,x} g[b][co][y][x] * in[b][ci][y*S + kh - PAD][x*S + kw - PAD]
//   db[co]             += scale * sum_{b,y,x} g[b][co][y][x]
//
// GEMM view: M = 32 couts, N = (tap, ci) columns of one 32-byte input channel group (+1 all-ones
// column that yields the bias gradient for free), K = pixels.  The contraction runs on the exact
// fp32 matrix pipe (v_mfma_f32_32x32x2_f32; fp16 operands are widened on the LDS read) so the result
// is an fp32 fma chain per (row-strip, image) followed by fp32 atomic adds into the OIHW fp32
// gradient — the layout torch.optim.Adam consumes directly (SRRaGAN_model.py:82-89).
// Round-1 version: correctness first (plain staging, scalar LDS operand reads); the fp16
// transposed-read (ds_read_b64_tr_b16) MFMA-f16 version is the planned upgrade.
#include "common.h"

namespace {

template <typename T, int KS, int S, bool UPS>
struct WG {
  static constexpr int CPG = DT<T>::CPG;
  static constexpr int ESZ = 32 / CPG;                        // bytes per element
  static constexpr int TR = 8, TC = 32;                        // g tile: 8 rows x 32 px
  static constexpr int PAD = (KS - 1) / 2;
  static constexpr int IH = UPS ? TR / 2 + 2 : (TR - 1) * S + KS;
  static constexpr int IW = UPS ? TC / 2 + 2 : (TC - 1) * S + KS;
  static constexpr int NCOL = KS * KS * CPG + 1;               // +1: bias column
  static constexpr int NTILE = (NCOL + 31) / 32;
  static constexpr int TPW = (NTILE + 3) / 4;                  // N-tiles per wave (4 waves)
  static constexpr int G_BYTES = TR * TC * 32 * ESZ;           // [pixel][32 couts]
  static constexpr int IN_BYTES = IH * IW * 32;                // [pixel][CPG ch]
};

template <typename T, int KS, int S, bool UPS>
__global__ __launch_bounds__(256) void wgrad_kernel(const esr_wgrad p) {
  using W = WG<T, KS, S, UPS>;
  __shared__ __attribute__((aligned(16))) char smem[W::G_BYTES + W::IN_BYTES];
  char* const lg = smem;
  char* const li = smem + W::G_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kk = lane >> 5;
  const int strips = (p.W + W::TC - 1) / W::TC;
  const int b = blockIdx.x / strips, sx = blockIdx.x % strips;
  const int cg = blockIdx.y;               // input channel group
  const int cb = blockIdx.z;               // 32-cout block
  const int ox0 = sx * W::TC;

  // per-lane B column -> LDS byte offset inside the input tile (for pixel (0,0) of the g tile)
  int boff[W::TPW];
  int bkind[W::TPW];                       // 0 = data column, 1 = bias (ones) column, 2 = padding
  int kh_[W::TPW], kw_[W::TPW];
#pragma unroll
  for (int t = 0; t < W::TPW; ++t) {
    const int n = (wave + 4 * t) * 32 + i;
    const int tap = n / W::CPG, ci = n - tap * W::CPG;
    kh_[t] = tap / KS; kw_[t] = tap - kh_[t] * KS;
    bkind[t] = n < KS * KS * W::CPG ? 0 : (n == KS * KS * W::CPG ? 1 : 2);
    boff[t] = ci * W::ESZ;
  }

  f32x16 acc[W::TPW];
#pragma unroll
  for (int t = 0; t < W::TPW; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const char* gbase = (const char*)p.g.ptr + b * p.g.batch_stride + (int64_t)(cb * DT<T>::GPB) * p.g.group_stride;
  const char* ibase = (const char*)p.in.ptr + b * p.in.batch_stride + (int64_t)cg * p.in.group_stride;
  constexpr int GPB = DT<T>::GPB;

  for (int oy0 = 0; oy0 < p.H; oy0 += W::TR) {
    __syncthreads();
    // ---- stage g tile: [TR*TC pixels][32 couts] (GPB groups of 32 bytes each); rows/cols past the
    // image read the zero halo or are zero-filled
    for (int s = tid; s < W::TR * W::TC * GPB * 2; s += 256) {
      const int half = s & 1, g = (s >> 1) % GPB, px = (s >> 1) / GPB;
      const int r = px / W::TC, c = px % W::TC;
      u32x4 v = {0, 0, 0, 0};
      if (oy0 + r < p.H && ox0 + c < p.W && cb * GPB + g < p.g.ngroups)
        v = *(const u32x4*)(gbase + (int64_t)g * p.g.group_stride + ((int64_t)(oy0 + r + 1) * p.g.wp + ox0 + c + 1) * 32 + half * 16);
      *(u32x4*)(lg + (px * GPB + g) * 32 + half * 16) = v;
    }
    // ---- stage input tile (one channel group) with its halo
    const int iy0 = UPS ? oy0 / 2 : oy0 * S + 1 - W::PAD;
    const int ix0 = UPS ? ox0 / 2 : ox0 * S + 1 - W::PAD;
    for (int s = tid; s < W::IH * W::IW * 2; s += 256) {
      const int half = s & 1, px = s >> 1;
      const int r = px / W::IW, c = px % W::IW;
      *(u32x4*)(li + px * 32 + half * 16) =
          *(const u32x4*)(ibase + ((int64_t)(iy0 + r) * p.in.wp + ix0 + c) * 32 + half * 16);
    }
    __syncthreads();

    // ---- K loop: 2 pixels per MFMA (lane half kk picks the pixel)
    for (int s = 0; s < W::TR * W::TC / 2; ++s) {
      const int k = 2 * s + kk;
      const int r = k / W::TC, c = k % W::TC;
      const float a = (float)*(const T*)(lg + k * GPB * 32 + i * W::ESZ);
#pragma unroll
      for (int t = 0; t < W::TPW; ++t) {
        if ((wave + 4 * t) >= W::NTILE) continue;
        float bv;
        if (bkind[t] == 0) {
          int rr, cc;
          if (UPS) { rr = ((r + kh_[t] - 1) >> 1) + 1; cc = ((c + kw_[t] - 1) >> 1) + 1; }
          else { rr = r * S + kh_[t]; cc = c * S + kw_[t]; }
          bv = (float)*(const T*)(li + (rr * W::IW + cc) * 32 + boff[t]);
        } else {
          bv = bkind[t] == 1 ? 1.f : 0.f;
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
      }
    }
  }

  // ---- accumulate into the fp32 OIHW gradient (and bias gradient) with atomics
#pragma unroll
  for (int t = 0; t < W::TPW; ++t) {
    if ((wave + 4 * t) >= W::NTILE || bkind[t] == 2) continue;
    const int n = (wave + 4 * t) * 32 + i;
    const int tap = n / W::CPG, ci = cg * W::CPG + (n - tap * W::CPG);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;     // MFMA C/D row map
      if (co >= p.cout) continue;
      const float v = acc[t][e] * p.scale;
      if (bkind[t] == 1) {
        if (p.dbias && cg == 0) atomicAdd(p.dbias + co, v);
      } else if (ci < p.cin) {
        atomicAdd(p.dw + ((int64_t)co * p.cin + ci) * (KS * KS) + tap, v);
      }
    }
  }
}

template <typename T, int KS, int S, bool UPS>
int launch_wgrad(const esr_wgrad& p, hipStream_t st) {
  const int strips = (p.W + 31) / 32;
  dim3 grid(p.B * strips, p.in.ngroups, (p.cout + 31) / 32);
  hipLaunchKernelGGL((wgrad_kernel<T, KS, S, UPS>), grid, dim3(256), 0, st, p);
  return esr_check_launch("wgrad_kernel");
}

template <typename T>
int dispatch_wgrad(const esr_wgrad& p, hipStream_t st) {
  if (p.ks == 3 && p.stride == 1 && !p.upsample) return launch_wgrad<T, 3, 1, false>(p, st);
  if (p.ks == 3 && p.stride == 1 && p.upsample) return launch_wgrad<T, 3, 1, true>(p, st);
  if (p.ks == 1 && p.stride == 1 && !p.upsample) return launch_wgrad<T, 1, 1, false>(p, st);
  if (p.ks == 4 && p.stride == 2 && !p.upsample) return launch_wgrad<T, 4, 2, false>(p, st);
  esr_set_error("wgrad: unsupported ks=%d stride=%d upsample=%d", p.ks, p.stride, p.upsample);
  return ESR_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int esr_conv_wgrad(const esr_wgrad* p, esr_stream_t stream) {
  if (!p || !p->g.ptr || !p->in.ptr || !p->dw || p->B <= 0 || p->H <= 0 || p->W <= 0 || p->cout <= 0 || p->cin <= 0) {
    esr_set_error("esr_conv_wgrad: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (p->dtype == ESR_F16) return dispatch_wgrad<_Float16>(*p, st);
  if (p->dtype == ESR_F32) return dispatch_wgrad<float>(*p, st);
  esr_set_error("esr_conv_wgrad: bad dtype %d", p->dtype);
  return ESR_ERR_INVALID;
}
