// nn_kernels.hip — the non-conv layers of Discriminator_VGG_128 (architecture.py:87-129) and
// VGG19 features (architecture.py:279-307): BatchNorm2d, MaxPool2d(2,2), Linear.  All HBM-bound,
// one pass per phase over G32 tensors (row-major fp32 for the classifier).
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ void ld16(const char* p, float v[DT<T>::CPG]);
template <> __device__ __forceinline__ void ld16<_Float16>(const char* p, float v[16]) {
  const half8 a = __builtin_bit_cast(half8, *(const u32x4*)p), b = __builtin_bit_cast(half8, *(const u32x4*)(p + 16));
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = (float)a[i]; v[8 + i] = (float)b[i]; }
}
template <> __device__ __forceinline__ void ld16<float>(const char* p, float v[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <typename T> __device__ __forceinline__ void st16(char* p, const float v[DT<T>::CPG]);
template <> __device__ __forceinline__ void st16<_Float16>(char* p, const float v[16]) {
  half8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)v[i]; b[i] = (_Float16)v[8 + i]; }
  *(u32x4*)p = __builtin_bit_cast(u32x4, a);
  *(u32x4*)(p + 16) = __builtin_bit_cast(u32x4, b);
}
template <> __device__ __forceinline__ void st16<float>(char* p, const float v[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *(f32x4*)p = a;
  *(f32x4*)(p + 16) = b;
}

__device__ __forceinline__ int64_t pix_off(const esr_g32& t, int b, int g, int y, int x) {
  return b * t.batch_stride + (int64_t)g * t.group_stride + ((int64_t)(y + 1) * t.wp + x + 1) * 32;
}

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ESR_ACT_LRELU) return v > 0.f ? v : v * ESR_LRELU_SLOPE;
  if (act == ESR_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
__device__ __forceinline__ float act_bwd(float y, int act) {   // derivative selected by saved OUTPUT
  if (act == ESR_ACT_LRELU) return y > 0.f ? 1.f : ESR_LRELU_SLOPE;
  if (act == ESR_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}

// grid: (ceil(H*W/256), groups, B); block 256.  Per-channel reductions: wave shuffle + LDS + fp64 atomics.
// `ppt` = pixels per thread: the reduction modes walk several 256-pixel chunks per workgroup so the
// fp64 atomics (2*CPG per workgroup, all workgroups of a channel group hitting the same addresses) are
// amortised — one chunk per workgroup made the backward reduce atomic-bound (928 us on 256x64x128^2).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_pass_kernel(const esr_bn p, const int ppt) {
  constexpr int CPG = DT<T>::CPG;
  const int g = blockIdx.y, b = blockIdx.z;
  float xv[CPG], yv[CPG], gv[CPG];
  float s0[CPG], s1[CPG];
#pragma unroll
  for (int e = 0; e < CPG; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  // statistics groups: images [grp*B/groups, (grp+1)*B/groups) are one BatchNorm batch (two forward calls of the
  // reference run as one launch); every per-channel array carries one [C] row per group
  const int ngrp = p.groups > 1 ? p.groups : 1, bpg = p.B / ngrp, grp = b / bpg;
  const double N = (double)bpg * p.H * p.W;
  const float* mean = p.mean + grp * p.C;
  const float* invstd = p.invstd + grp * p.C;
  double* const sums = p.sums + (int64_t)grp * 2 * p.C;
  if constexpr (MODE == ESR_BN_FIN_APPLY) {
    // FINALIZE folded into the APPLY pass (training): every workgroup forms the statistics of its own 16 channels
    // from the sums (the FINALIZE arithmetic, so the numbers are the same); the workgroups at pixel chunk 0 of a
    // group's first image also leave them in p.mean / p.invstd for the backward, and the one at image 0 applies the
    // running-statistics updates of all groups in order
    __shared__ float st_m[CPG], st_i[CPG];
    if (threadIdx.x < CPG) {
      const int c = g * CPG + threadIdx.x;
      float m_ = 0.f, i_ = 0.f;
      if (c < p.C) {
        const double m = sums[c] / N;
        double var = sums[p.C + c] / N - m * m;
        if (var < 0) var = 0;
        m_ = (float)m;
        i_ = (float)(1.0 / sqrt(var + (double)p.eps));
        if (blockIdx.x == 0 && b == grp * bpg) {
          p.mean[grp * p.C + c] = m_;
          p.invstd[grp * p.C + c] = i_;
        }
        if (blockIdx.x == 0 && b == 0) {
          if (p.running_mean) {
            float rm = p.running_mean[c], rv = p.running_var[c];
            for (int q = 0; q < ngrp; ++q) {
              const double* s = p.sums + (int64_t)q * 2 * p.C;
              const double mq = s[c] / N;
              double vq = s[p.C + c] / N - mq * mq;
              if (vq < 0) vq = 0;
              rm = (float)((1.0 - p.momentum) * rm + p.momentum * mq);
              rv = (float)((1.0 - p.momentum) * rv + p.momentum * vq * (N / (N - 1.0)));
            }
            p.running_mean[c] = rm;
            p.running_var[c] = rv;
          }
          if (p.num_batches_tracked && c == 0) *p.num_batches_tracked += ngrp;
        }
      }
      st_m[threadIdx.x] = m_;
      st_i[threadIdx.x] = i_;
    }
    __syncthreads();
    mean = st_m - g * CPG;        // indexed by channel below
    invstd = st_i - g * CPG;
  }
  if constexpr (MODE == ESR_BN_BWD_APPLY) {
    // BWD_FINAL folded in: dgamma / dbeta from the sums of all groups, once per channel
    if (p.dgamma && blockIdx.x == 0 && b == 0 && threadIdx.x < CPG) {
      const int c = g * CPG + threadIdx.x;
      if (c < p.C) {
        double sg = 0, sb = 0;
        for (int q = 0; q < ngrp; ++q) {
          const double* s = p.sums + (int64_t)q * 2 * p.C;
          sg += s[p.C + c]; sb += s[c];
        }
        p.dgamma[c] += (float)sg;
        if (p.dbeta) p.dbeta[c] += (float)sb;
      }
    }
  }
  for (int k = 0; k < ppt; ++k) {
  const int pix = (blockIdx.x * ppt + k) * 256 + threadIdx.x;
  const bool ok = pix < p.H * p.W;
  const int y = ok ? pix / p.W : 0, x = ok ? pix % p.W : 0;
  if (ok) {
    ld16<T>((const char*)p.x.ptr + pix_off(p.x, b, g, y, x), xv);
    if (MODE == ESR_BN_STATS) {
#pragma unroll
      for (int e = 0; e < CPG; ++e) { s0[e] += xv[e]; s1[e] += xv[e] * xv[e]; }
    } else if (MODE == ESR_BN_APPLY || MODE == ESR_BN_FIN_APPLY) {
#pragma unroll
      for (int e = 0; e < CPG; ++e) {
        const int c = g * CPG + e;
        float v = 0.f;
        if (c < p.C) v = act_fwd((xv[e] - mean[c]) * invstd[c] * p.gamma[c] + p.beta[c], p.act);
        yv[e] = v;
      }
      st16<T>((char*)p.y.ptr + pix_off(p.y, b, g, y, x), yv);
    } else {   // BWD_REDUCE / BWD_APPLY
      ld16<T>((const char*)p.y.ptr + pix_off(p.y, b, g, y, x), yv);
      ld16<T>((const char*)p.g.ptr + pix_off(p.g, b, g, y, x), gv);
#pragma unroll
      for (int e = 0; e < CPG; ++e) {
        const int c = g * CPG + e;
        if (c >= p.C) { gv[e] = 0.f; continue; }
        const float gp = gv[e] * act_bwd(yv[e], p.act);
        const float xh = (xv[e] - mean[c]) * invstd[c];
        if (MODE == ESR_BN_BWD_REDUCE) { s0[e] += gp; s1[e] += gp * xh; }
        else {
          float r = gp;
          if (p.training) r = gp - (float)(sums[c] / N) - xh * (float)(sums[p.C + c] / N);
          gv[e] = r * p.gamma[c] * invstd[c];
        }
      }
      if (MODE == ESR_BN_BWD_APPLY) st16<T>((char*)p.gx.ptr + pix_off(p.gx, b, g, y, x), gv);
    }
  }
  }   // pixel chunks
  if (MODE == ESR_BN_STATS || MODE == ESR_BN_BWD_REDUCE) {
    __shared__ float red[4][2][CPG];
#pragma unroll
    for (int e = 0; e < CPG; ++e) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { s0[e] += __shfl_xor(s0[e], o); s1[e] += __shfl_xor(s1[e], o); }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < CPG; ++e) { red[wave][0][e] = s0[e]; red[wave][1][e] = s1[e]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * CPG) {
      const int k = threadIdx.x / CPG, e = threadIdx.x % CPG, c = g * CPG + e;
      if (c < p.C) {
        const double v = (double)red[0][k][e] + red[1][k][e] + red[2][k][e] + red[3][k][e];
        atomicAdd(sums + k * p.C + c, v);
      }
    }
  }
}

__global__ void bn_small_kernel(const esr_bn p) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.C) return;
  const int ngrp = p.groups > 1 ? p.groups : 1;
  const double N = (double)(p.B / ngrp) * p.H * p.W;
  if (p.mode == ESR_BN_FINALIZE) {
    if (p.training) {
      // groups in order: the running statistics see the same sequence of momentum updates as one forward call
      // per group would give them
      float rm = p.running_mean ? p.running_mean[c] : 0.f, rv = p.running_mean ? p.running_var[c] : 0.f;
      for (int q = 0; q < ngrp; ++q) {
        const double* s = p.sums + (int64_t)q * 2 * p.C;
        const double m = s[c] / N;
        double var = s[p.C + c] / N - m * m;
        if (var < 0) var = 0;
        p.mean[q * p.C + c] = (float)m;
        p.invstd[q * p.C + c] = (float)(1.0 / sqrt(var + (double)p.eps));
        rm = (float)((1.0 - p.momentum) * rm + p.momentum * m);
        rv = (float)((1.0 - p.momentum) * rv + p.momentum * var * (N / (N - 1.0)));
      }
      if (p.running_mean) { p.running_mean[c] = rm; p.running_var[c] = rv; }
      if (p.num_batches_tracked && c == 0) *p.num_batches_tracked += ngrp;
    } else {
      for (int q = 0; q < ngrp; ++q) {
        p.mean[q * p.C + c] = p.running_mean[c];
        p.invstd[q * p.C + c] = 1.0f / sqrtf(p.running_var[c] + p.eps);
      }
    }
  } else if (p.mode == ESR_BN_RESTAT) {
    // the running-statistics side of ANOTHER training forward over the same batch with the groups in reverse order
    // (the sums of the forward that ran are still in place): what the reference's second pair of netD calls leaves
    if (!p.training || !p.running_mean) return;
    float rm = p.running_mean[c], rv = p.running_var[c];
    for (int q = ngrp - 1; q >= 0; --q) {
      const double* s = p.sums + (int64_t)q * 2 * p.C;
      const double m = s[c] / N;
      double var = s[p.C + c] / N - m * m;
      if (var < 0) var = 0;
      rm = (float)((1.0 - p.momentum) * rm + p.momentum * m);
      rv = (float)((1.0 - p.momentum) * rv + p.momentum * var * (N / (N - 1.0)));
    }
    p.running_mean[c] = rm;
    p.running_var[c] = rv;
    if (p.num_batches_tracked && c == 0) *p.num_batches_tracked += ngrp;
  } else {   // BWD_FINAL
    double sg = 0, sb = 0;
    for (int q = 0; q < ngrp; ++q) {
      const double* s = p.sums + (int64_t)q * 2 * p.C;
      sg += s[p.C + c]; sb += s[c];
    }
    if (p.dgamma) p.dgamma[c] += (float)sg;
    if (p.dbeta) p.dbeta[c] += (float)sb;
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void pool_kernel(const esr_pool p) {
  constexpr int CPG = DT<T>::CPG;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y, b = blockIdx.z;
  if (pix >= p.H * p.W) return;
  const int y = pix / p.W, x = pix % p.W;
  float v[4][CPG], m[CPG];
#pragma unroll
  for (int q = 0; q < 4; ++q) ld16<T>((const char*)p.x.ptr + pix_off(p.x, b, g, 2 * y + (q >> 1), 2 * x + (q & 1)), v[q]);
  int am[CPG];
#pragma unroll
  for (int e = 0; e < CPG; ++e) {
    m[e] = v[0][e]; am[e] = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) if (v[q][e] > m[e]) { m[e] = v[q][e]; am[e] = q; }   // first max wins
  }
  if (MODE == 0) {
    st16<T>((char*)p.y.ptr + pix_off(p.y, b, g, y, x), m);
  } else {
    float gv[CPG], o[CPG];
    ld16<T>((const char*)p.g.ptr + pix_off(p.g, b, g, y, x), gv);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < CPG; ++e) o[e] = (am[e] == q && (!p.relu_mask || v[q][e] > 0.f)) ? gv[e] : 0.f;
      st16<T>((char*)p.gx.ptr + pix_off(p.gx, b, g, 2 * y + (q >> 1), 2 * x + (q & 1)), o);
    }
  }
}

// nn.PixelShuffle(2) (block.py:299-312) on G32 tensors: out[b][c][2y+i][2x+j] = in[b][4c + 2i + j][y][x].  One thread per
// (low-res pixel, high-res channel group): the group's CPG channels at the four output pixels are the 4 CPG channels of
// four consecutive low-res groups.  INV: the gradient's way back (gx[4c + 2i + j][y][x] = g[c][2y+i][2x+j]), optionally
// masked by the ReLU of the conv that produced the shuffled tensor (p.x = that conv's stored output).
template <typename T, bool INV>
__global__ __launch_bounds__(256) void shuffle_kernel(const esr_pool p) {
  constexpr int CPG = DT<T>::CPG;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y, b = blockIdx.z;
  if (pix >= p.H * p.W) return;
  const int y = pix / p.W, x = pix % p.W;
  float lo[4][CPG], hi[4][CPG];
  if (!INV) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ld16<T>((const char*)p.x.ptr + pix_off(p.x, b, 4 * g + q, y, x), lo[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < CPG; ++e) { const int cl = q * CPG + e; hi[cl & 3][cl >> 2] = lo[q][e]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) st16<T>((char*)p.y.ptr + pix_off(p.y, b, g, 2 * y + (q >> 1), 2 * x + (q & 1)), hi[q]);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) ld16<T>((const char*)p.g.ptr + pix_off(p.g, b, g, 2 * y + (q >> 1), 2 * x + (q & 1)), hi[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (p.relu_mask) ld16<T>((const char*)p.x.ptr + pix_off(p.x, b, 4 * g + q, y, x), lo[q]);
      float o[CPG];
#pragma unroll
      for (int e = 0; e < CPG; ++e) {
        const int cl = q * CPG + e;
        o[e] = (!p.relu_mask || lo[q][e] > 0.f) ? hi[cl & 3][cl >> 2] : 0.f;
      }
      st16<T>((char*)p.gx.ptr + pix_off(p.gx, b, 4 * g + q, y, x), o);
    }
  }
}

// y[b][o] = act(sum_i x[b][i] w[o][i] + bias[o]) — one block per (o, b)
__global__ __launch_bounds__(256) void linear_fwd_kernel(const esr_linear p) {
  const int o = blockIdx.x, b = blockIdx.y;
  float s = 0.f;
  for (int i = threadIdx.x; i < p.I; i += 256) s += p.x[(int64_t)b * p.I + i] * p.w[(int64_t)o * p.I + i];
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = red[0] + red[1] + red[2] + red[3] + (p.b ? p.b[o] : 0.f);
    p.y[(int64_t)b * p.O + o] = act_fwd(v, p.act);
  }
}
// gx[b][i] = sum_o (g[b][o] * act'(ysaved[b][o])) w[o][i]   [* in_act'(x[b][i]): the activation that produced this input]
__global__ __launch_bounds__(256) void linear_bwdx_kernel(const esr_linear p) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= p.I) return;
  float s = 0.f;
  // ten rows of w in flight per round (the sum keeps its order): one dependent load per iteration left the 100-row
  // classifier layer of the discriminators at 57 us for 3 MB of weights — on the train step's critical path
  int o = 0;
  for (; o + 10 <= p.O; o += 10) {
    float wv[10], gv[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) wv[u] = p.w[(int64_t)(o + u) * p.I + i];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      gv[u] = p.g[(int64_t)b * p.O + o + u];
      if (p.ysaved) gv[u] *= act_bwd(p.ysaved[(int64_t)b * p.O + o + u], p.act);
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) s += gv[u] * wv[u];
  }
  for (; o < p.O; ++o) {
    float gg = p.g[(int64_t)b * p.O + o];
    if (p.ysaved) gg *= act_bwd(p.ysaved[(int64_t)b * p.O + o], p.act);
    s += gg * p.w[(int64_t)o * p.I + i];
  }
  if (p.in_act != ESR_ACT_NONE && p.x) s *= act_bwd(p.x[(int64_t)b * p.I + i], p.in_act);
  p.gx[(int64_t)b * p.I + i] = s;
}
// dw[o][i] += sum_b g'[b][o] x[b][i];  db[o] += sum_b g'[b][o]
__global__ __launch_bounds__(256) void linear_bwdw_kernel(const esr_linear p) {
  const int i = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
  if (i >= p.I) return;
  float s = 0.f, sb = 0.f;
  int b = 0;
  for (; b + 8 <= p.B; b += 8) {            // eight rows of x in flight per round, sums in the same order
    float xv[8], gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = p.x[(int64_t)(b + u) * p.I + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      gv[u] = p.g[(int64_t)(b + u) * p.O + o];
      if (p.ysaved) gv[u] *= act_bwd(p.ysaved[(int64_t)(b + u) * p.O + o], p.act);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += gv[u] * xv[u]; sb += gv[u]; }
  }
  for (; b < p.B; ++b) {
    float gg = p.g[(int64_t)b * p.O + o];
    if (p.ysaved) gg *= act_bwd(p.ysaved[(int64_t)b * p.O + o], p.act);
    s += gg * p.x[(int64_t)b * p.I + i];
    sb += gg;
  }
  p.dw[(int64_t)o * p.I + i] += s;
  if (i == 0 && p.db) p.db[o] += sb;
}

template <typename T>
int bn_dispatch(const esr_bn& p, hipStream_t st) {
  constexpr int CPG = DT<T>::CPG;
  const int chunks = (p.H * p.W + 255) / 256;
  dim3 grid(chunks, (p.C + CPG - 1) / CPG, p.B), block(256);
  // reductions: up to 16 chunks per workgroup, but keep >= ~1024 workgroups in flight
  int ppt = 1;
  while (ppt < 16 && (int64_t)((chunks + 2 * ppt - 1) / (2 * ppt)) * grid.y * grid.z >= 1024) ppt *= 2;
  dim3 rgrid((chunks + ppt - 1) / ppt, grid.y, grid.z);
  switch (p.mode) {
    case ESR_BN_STATS: hipLaunchKernelGGL((bn_pass_kernel<T, ESR_BN_STATS>), rgrid, block, 0, st, p, ppt); break;
    case ESR_BN_APPLY: hipLaunchKernelGGL((bn_pass_kernel<T, ESR_BN_APPLY>), grid, block, 0, st, p, 1); break;
    case ESR_BN_FIN_APPLY:
      if (!p.training) { esr_set_error("esr_batchnorm: FIN_APPLY is a training-mode pass"); return ESR_ERR_INVALID; }
      hipLaunchKernelGGL((bn_pass_kernel<T, ESR_BN_FIN_APPLY>), grid, block, 0, st, p, 1);
      break;
    case ESR_BN_BWD_REDUCE: hipLaunchKernelGGL((bn_pass_kernel<T, ESR_BN_BWD_REDUCE>), rgrid, block, 0, st, p, ppt); break;
    case ESR_BN_BWD_APPLY: hipLaunchKernelGGL((bn_pass_kernel<T, ESR_BN_BWD_APPLY>), grid, block, 0, st, p, 1); break;
    case ESR_BN_FINALIZE:
    case ESR_BN_RESTAT:
    case ESR_BN_BWD_FINAL: hipLaunchKernelGGL(bn_small_kernel, dim3((p.C + 63) / 64), dim3(64), 0, st, p); break;
    default: esr_set_error("esr_batchnorm: bad mode %d", p.mode); return ESR_ERR_INVALID;
  }
  return esr_check_launch("bn_kernel");
}

}  // namespace

extern "C" int esr_batchnorm(const esr_bn* p, esr_stream_t stream) {
  if (!p || p->B <= 0 || p->C <= 0 || p->H <= 0 || p->W <= 0 || !p->mean || !p->invstd) {
    esr_set_error("esr_batchnorm: invalid arguments");
    return ESR_ERR_INVALID;
  }
  if (p->groups > 1 && p->B % p->groups) {
    esr_set_error("esr_batchnorm: batch %d does not split into %d statistics groups", p->B, p->groups);
    return ESR_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (p->dtype == ESR_F16) return bn_dispatch<_Float16>(*p, st);
  if (p->dtype == ESR_F32) return bn_dispatch<float>(*p, st);
  esr_set_error("esr_batchnorm: bad dtype");
  return ESR_ERR_INVALID;
}

extern "C" int esr_maxpool2(const esr_pool* p, esr_stream_t stream) {
  if (!p || p->B <= 0 || p->C <= 0 || p->H <= 0 || p->W <= 0 || !p->x.ptr) {
    esr_set_error("esr_maxpool2: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const int cpg = p->dtype == ESR_F16 ? 16 : 8;
  dim3 grid((p->H * p->W + 255) / 256, (p->C + cpg - 1) / cpg, p->B), block(256);
  if (p->mode == 2 || p->mode == 3) {
    // nn.PixelShuffle(2) / its adjoint: C = channels of the HIGH-resolution tensor, H x W = the LOW-resolution size
    if (p->C % cpg != 0 || (p->mode == 2 ? !p->y.ptr : (!p->g.ptr || !p->gx.ptr))) {
      esr_set_error("esr_maxpool2 (pixel shuffle): C must be a multiple of the channel group, operands must be set");
      return ESR_ERR_INVALID;
    }
    if (p->dtype == ESR_F16) {
      if (p->mode == 2) hipLaunchKernelGGL((shuffle_kernel<_Float16, false>), grid, block, 0, st, *p);
      else hipLaunchKernelGGL((shuffle_kernel<_Float16, true>), grid, block, 0, st, *p);
    } else if (p->dtype == ESR_F32) {
      if (p->mode == 2) hipLaunchKernelGGL((shuffle_kernel<float, false>), grid, block, 0, st, *p);
      else hipLaunchKernelGGL((shuffle_kernel<float, true>), grid, block, 0, st, *p);
    } else { esr_set_error("esr_maxpool2: bad dtype"); return ESR_ERR_INVALID; }
    return esr_check_launch("shuffle_kernel");
  }
  if (p->dtype == ESR_F16) {
    if (p->mode == 0) hipLaunchKernelGGL((pool_kernel<_Float16, 0>), grid, block, 0, st, *p);
    else hipLaunchKernelGGL((pool_kernel<_Float16, 1>), grid, block, 0, st, *p);
  } else if (p->dtype == ESR_F32) {
    if (p->mode == 0) hipLaunchKernelGGL((pool_kernel<float, 0>), grid, block, 0, st, *p);
    else hipLaunchKernelGGL((pool_kernel<float, 1>), grid, block, 0, st, *p);
  } else { esr_set_error("esr_maxpool2: bad dtype"); return ESR_ERR_INVALID; }
  return esr_check_launch("pool_kernel");
}

extern "C" int esr_linear_op(const esr_linear* p, esr_stream_t stream) {
  if (!p || p->B <= 0 || p->I <= 0 || p->O <= 0 || !p->w) {
    esr_set_error("esr_linear_op: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (p->mode == 0) hipLaunchKernelGGL(linear_fwd_kernel, dim3(p->O, p->B), dim3(256), 0, st, *p);
  else if (p->mode == 1) hipLaunchKernelGGL(linear_bwdx_kernel, dim3((p->I + 255) / 256, p->B), dim3(256), 0, st, *p);
  else if (p->mode == 2) hipLaunchKernelGGL(linear_bwdw_kernel, dim3((p->I + 255) / 256, p->O), dim3(256), 0, st, *p);
  else { esr_set_error("esr_linear_op: bad mode"); return ESR_ERR_INVALID; }
  return esr_check_launch("linear_kernel");
}

// ------------------------------------------------------------------------------------------------
// Fused multi-tensor Adam: HBM-bound (4 reads + 3 writes of 4 bytes per element), one launch per
// network instead of ~8 foreach launches over 770 tensors.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void adam_kernel(const esr_adam a) {
  const esr_adam_block blk = a.blocks[blockIdx.x];
  const esr_adam_entry e = a.entries[blk.entry];
  const int64_t end = min((int64_t)blk.first + ESR_ADAM_BLOCK_ELEMS, e.n);
  float bc1 = a.bc1, bc2 = a.bc2;
  if (a.step_count) {                      // device step counter: only APPLIED steps age the bias correction
    const double t = (double)a.step_count[0];
    bc1 = (float)(1.0 - pow(a.beta1_d, t));
    bc2 = (float)(1.0 - pow(a.beta2_d, t));
  }
  const float step = a.lr / bc1, rs2 = rsqrtf(bc2), omb1 = 1.f - a.beta1, omb2 = 1.f - a.beta2;
  float gs = a.grad_scale;
  if (a.amp_state) {                       // dynamic loss scaling: skip the step on overflow, else un-scale
    if (a.amp_state[4 + a.amp_slot] != 0.f) return;
    gs /= a.amp_state[0];
  }
  for (int64_t i = blk.first + threadIdx.x; i < end; i += 256) {
    float p = e.p[i];
    float g = a.grad[e.goff + i] * gs;
    if (a.weight_decay != 0.f) g += a.weight_decay * p;
    const float m = a.beta1 * a.exp_avg[e.goff + i] + omb1 * g;
    const float v = a.beta2 * a.exp_avg_sq[e.goff + i] + omb2 * g * g;
    a.exp_avg[e.goff + i] = m;
    a.exp_avg_sq[e.goff + i] = v;
    e.p[i] = p - step * m / (sqrtf(v) * rs2 + a.eps);
  }
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void amp_check_kernel(const esr_amp a) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
    const float g = a.grad[i];
    bad |= !(fabsf(g) <= 3.4028234e38f);        // inf or nan
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) a.state[4 + a.slot] = 1.f;   // benign race: every writer stores the same value
}
__global__ void amp_count_kernel(const esr_amp a) {
  if (a.state[4 + a.slot] == 0.f) a.step_count[0] += 1.f;
}
__global__ void amp_update_kernel(const esr_amp a) {
  float s = a.state[0], good = a.state[2];
  const bool found = a.state[4] != 0.f || a.state[5] != 0.f || a.state[6] != 0.f || a.state[7] != 0.f;
  if (found) { s *= a.backoff; good = 0.f; }
  else if (++good >= (float)a.interval) { s *= a.growth; good = 0.f; }
  a.state[0] = s; a.state[1] = 0.f; a.state[2] = good;
  a.state[4] = a.state[5] = a.state[6] = a.state[7] = 0.f;
}
}  // namespace

extern "C" int esr_amp_step(const esr_amp* p, esr_stream_t stream) {
  if (!p || !p->state || (p->mode == ESR_AMP_CHECK && (!p->grad || p->n <= 0)) ||
      ((p->mode == ESR_AMP_CHECK || p->mode == ESR_AMP_COUNT) && (p->slot < 0 || p->slot > 3)) ||
      (p->mode == ESR_AMP_COUNT && !p->step_count) ||
      (p->mode == ESR_AMP_UPDATE && (p->interval <= 0 || !(p->growth >= 1.f) || !(p->backoff > 0.f && p->backoff <= 1.f)))) {
    esr_set_error("esr_amp_step: invalid arguments");
    return ESR_ERR_INVALID;
  }
  if (p->mode == ESR_AMP_CHECK) {
    const int64_t blocks = (p->n + 256 * 16 - 1) / (256 * 16);
    hipLaunchKernelGGL(amp_check_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, *p);
  } else if (p->mode == ESR_AMP_UPDATE) {
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *p);
  } else if (p->mode == ESR_AMP_COUNT) {
    hipLaunchKernelGGL(amp_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *p);
  } else { esr_set_error("esr_amp_step: bad mode %d", p->mode); return ESR_ERR_INVALID; }
  return esr_check_launch("amp_kernel");
}

extern "C" int esr_adam_step(const esr_adam* p, esr_stream_t stream) {
  if (!p || !p->entries || !p->blocks || p->nblocks <= 0 || !p->grad || !p->exp_avg || !p->exp_avg_sq ||
      (!p->step_count && (!(p->bc1 > 0.f) || !(p->bc2 > 0.f))) || (p->amp_state && (p->amp_slot < 0 || p->amp_slot > 3))) {
    esr_set_error("esr_adam_step: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipLaunchKernelGGL(adam_kernel, dim3(p->nblocks), dim3(256), 0, (hipStream_t)stream, *p);
  return esr_check_launch("adam_kernel");
}

// ------------------------------------------------------------------------------------------------
// Separable resampling (data path: MATLAB-style bicubic imresize).  HBM-bound gather of <= ~18 taps.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void resample_kernel(const esr_resample a) {
  const int oh = a.axis == 0 ? a.out_len : a.in_h, ow = a.axis == 0 ? a.in_w : a.out_len;
  const int64_t n = (int64_t)a.planes * oh * ow;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int x = (int)(e % ow), y = (int)((e / ow) % oh);
  const int64_t pl = e / ((int64_t)ow * oh);
  const float* src = a.in + pl * a.in_h * a.in_w;
  const int o = a.axis == 0 ? y : x;
  const float* w = a.w + (int64_t)o * a.taps;
  const int32_t* ix = a.idx + (int64_t)o * a.taps;
  float acc = 0.f;
  if (a.axis == 0) {
    for (int t = 0; t < a.taps; ++t) acc += w[t] * src[(int64_t)ix[t] * a.in_w + x];
  } else {
    for (int t = 0; t < a.taps; ++t) acc += w[t] * src[(int64_t)y * a.in_w + ix[t]];
  }
  a.out[e] = acc;
}
}  // namespace

extern "C" int esr_resample_axis(const esr_resample* p, esr_stream_t stream) {
  if (!p || !p->in || !p->out || !p->w || !p->idx || p->planes <= 0 || p->in_h <= 0 || p->in_w <= 0 ||
      p->out_len <= 0 || p->taps <= 0 || (p->axis != 0 && p->axis != 1)) {
    esr_set_error("esr_resample_axis: invalid arguments");
    return ESR_ERR_INVALID;
  }
  const int oh = p->axis == 0 ? p->out_len : p->in_h, ow = p->axis == 0 ? p->in_w : p->out_len;
  const int64_t n = (int64_t)p->planes * oh * ow;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
  return esr_check_launch("resample_kernel");
}

