// aux_kernels.hip — weight packing, NCHW<->G32 layout conversion, noise fill.
#include "common.h"

namespace {

// One thread per 16-byte fragment piece: [cb][chunk][kh][kw][lane] -> 8 halves / 4 floats.
template <typename T>
__device__ __forceinline__ void pack_piece(const esr_pack& p, int64_t idx) {
  constexpr int CPG = DT<T>::CPG;
  constexpr int EPL = CPG / 2;   // elements per lane (8 halves / 4 floats)
  if (p.one_t) {
    // transposed 1x1 of a dense block for the backward chain (esrgan_hip.h: esr_pack.one_t): fragment (cb, c), lane
    // (row i, k half h), element e  <-  W1x1[g_x2 channel 16 h + 8 c + e][x channel cb * 32 + pi(i)]
    if constexpr (CPG == 16) {
      const int lane = idx & 63, c = (idx >> 6) & 1, cb = (int)(idx >> 7);
      const int i = lane & 31, h = lane >> 5;
      const int ci_f = cb * 32 + esr_pi(i);
      T* dst = (T*)p.dst + idx * EPL;
#pragma unroll
      for (int e = 0; e < EPL; ++e) dst[e] = (T)(p.src[(int64_t)(16 * h + 8 * c + e) * p.cin + ci_f] * (p.scale == 0.f ? 1.f : p.scale));
    }
    return;
  }
  if (p.gather) {
    // one K range of a gather-form dgrad operand (esrgan_hip.h): rows = slice channels, K = fwd couts
    const int nchunks = (p.cout + CPG - 1) / CPG;
    const int lane = idx & 63;
    int64_t rest = idx >> 6;
    const int tap = rest % 9; rest /= 9;
    const int chunk = rest % nchunks;
    const int cb = rest / nchunks;
    const int kh = tap / 3, kw = tap % 3;
    const int i = lane & 31, h = lane >> 5;
    const int co = cb * 32 + esr_pi(i);
    T v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int ci = chunk * CPG + EPL * h + e;
      float x = 0.f;
      if (co < p.dst_cout && ci < p.cout) {
        const int64_t base = (int64_t)ci * p.cin + p.src_co0 + co;
        if (p.src_ks == 3) x = p.src[(base * 3 + (2 - kh)) * 3 + (2 - kw)];
        else if (kh == 1 && kw == 1) x = p.src[base];
        // identity path folded in (x4 = lrelu(a4) + x2: conv's x4 columns added to its x2 columns)
        if (p.fold_co0 > 0 && p.src_ks == 3)
          x += p.src[(((int64_t)ci * p.cin + p.fold_co0 + co) * 3 + (2 - kh)) * 3 + (2 - kw)];
      }
      v[e] = (T)(x * p.scale);
    }
    T* dst = (T*)p.dst + ((((int64_t)cb * p.dst_nchunks + p.dst_chunk0 + chunk) * 9 + tap) * 64 + lane) * EPL;
#pragma unroll
    for (int e = 0; e < EPL; ++e) dst[e] = v[e];
    return;
  }
  if (p.ups_fwd) {
    // sub-pixel form of nearest-x2 + conv3x3: phase (dy, dx), 2x2 taps (a, b); tap a of phase dy collects the
    // forward rows {0} / {1,2} (dy = 0) or {0,1} / {2} (dy = 1), same for columns
    const int nchunks = (p.cin + CPG - 1) / CPG, cbs = (p.cout + 31) / 32;
    const int lane = idx & 63;
    int64_t rest = idx >> 6;
    const int tap = rest % 4; rest /= 4;
    const int chunk = rest % nchunks; rest /= nchunks;
    const int cb = rest % cbs, phase = rest / cbs;
    const int dy = phase >> 1, dx = phase & 1, ta = tap >> 1, tb = tap & 1;
    const int r0 = dy == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2), r1 = dy == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
    const int c0 = dx == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2), c1 = dx == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
    const int i = lane & 31, h = lane >> 5;
    const int co = cb * 32 + esr_pi(i);
    T v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int ci = chunk * CPG + EPL * h + e;
      float x = 0.f;
      if (co < p.cout && ci < p.cin)
        for (int r = r0; r <= r1; ++r)
          for (int c = c0; c <= c1; ++c) x += p.src[(((int64_t)co * p.cin + ci) * 3 + r) * 3 + c];
      v[e] = (T)x;
    }
    T* dst = (T*)p.dst + idx * EPL;
#pragma unroll
    for (int e = 0; e < EPL; ++e) dst[e] = v[e];
    return;
  }
  const int kdim = p.transpose_flip ? p.cout : p.cin;
  const int nchunks = (kdim + CPG - 1) / CPG;
  const int lane = idx & 63;
  int64_t rest = idx >> 6;
  const int taps = p.ks * p.ks;
  const int tap = rest % taps; rest /= taps;
  const int chunk = rest % nchunks;
  const int cb = rest / nchunks;
  const int kh = tap / p.ks, kw = tap % p.ks;
  const int i = lane & 31, h = lane >> 5;
  const int co = cb * 32 + esr_pi(i);
  T v[EPL];
  // NB: in transpose_flip mode p.ks is the OUTPUT kernel size (4 when ups_dgrad).
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int ci = chunk * CPG + EPL * h + e;
    float x = 0.f;
    if (!p.transpose_flip) {
      // src is OIHW [cout][cin][ks][ks]
      if (co < p.cout && ci < p.cin) x = p.src[(((int64_t)co * p.cin + ci) * p.ks + kh) * p.ks + kw];
    } else {
      // dgrad operand.  Here `co` runs over the FORWARD conv's input channels (p.cin of them) and
      // `ci` over its output channels (p.cout); src is the forward OIHW [p.cout][p.cin][3|ks][..].
      if (co < p.cin && ci < p.cout) {
        if (!p.ups_dgrad) {
          // taps rotated 180 degrees (mode 1) or kept (mode 2: transposed stride-2 operand)
          const int fk = p.transpose_flip == 2 ? kh : p.ks - 1 - kh, fw = p.transpose_flip == 2 ? kw : p.ks - 1 - kw;
          x = p.src[(((int64_t)ci * p.cin + co) * p.ks + fk) * p.ks + fw];
          if (p.sum_count > 0 && co >= p.sum_dst && co < p.sum_dst + p.sum_count)
            x += p.src[(((int64_t)ci * p.cin + (co - p.sum_dst + p.sum_src)) * p.ks + fk) * p.ks + fw];
        } else {
          // adjoint of nearest-x2 + conv3x3 as a 4x4/s2/p1 conv over g: tap ky collects the forward
          // rows {2}, {1,2}, {0,1}, {0} for ky = 0..3 (same for kx) — see DESIGN.md
          const int r0 = kh == 0 ? 2 : (kh == 1 ? 1 : 0), r1 = kh == 0 ? 2 : (kh == 1 ? 2 : (kh == 2 ? 1 : 0));
          const int c0 = kw == 0 ? 2 : (kw == 1 ? 1 : 0), c1 = kw == 0 ? 2 : (kw == 1 ? 2 : (kw == 2 ? 1 : 0));
          for (int r = r0; r <= r1; ++r)
            for (int c = c0; c <= c1; ++c) x += p.src[(((int64_t)ci * p.cin + co) * 3 + r) * 3 + c];
        }
      }
    }
    v[e] = (T)x;
  }
  T* dst = (T*)p.dst + idx * EPL;
#pragma unroll
  for (int e = 0; e < EPL; ++e) dst[e] = v[e];
}

template <typename T>
__global__ void pack_kernel(const esr_pack p, int nchunks, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) pack_piece<T>(p, idx);
}

__global__ void pack_batch_kernel(const esr_pack_batch pb) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pb.total_pieces) return;
  int lo = 0, hi = pb.n - 1;                      // last entry with piece_begin <= idx
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pb.piece_begin[mid] <= idx) lo = mid; else hi = mid - 1;
  }
  const esr_pack p = pb.table[lo];
  const int64_t li = idx - pb.piece_begin[lo];
  #ifndef ESR_PACK_SLOW
  if (p.dtype == ESR_F16 && p.ks == 3 && !p.transpose_flip && !p.gather && !p.ups_fwd && !p.one_t && (p.cin & 7) == 0 &&
      ((uintptr_t)p.src & 15) == 0) {
    // Plain forward 3x3 operand, fp16 (every training step re-packs the generator's 16.8 M weights on the step's
    // critical path): the 9 taps of (cout row, 8 input channels) are 72 CONTIGUOUS floats of the OIHW tensor — the
    // thread that owns tap 0 reads them as 18 16-byte loads and writes all nine 16-byte pieces (tap stride 1 KB), the
    // other eight threads of the group retire at once; the per-piece path gathers every value with its own 4-byte
    // load at a 36-byte stride.  Same values, same places.
    const int tap = (int)((li >> 6) % 9);
    if (tap != 0) return;
    const int lane = (int)(li & 63), i = lane & 31, h = lane >> 5;
    const int64_t rest = (li >> 6) / 9;
    const int nchunks = (p.cin + 15) / 16;
    const int chunk = (int)(rest % nchunks), cb = (int)(rest / nchunks);
    const int co = cb * 32 + esr_pi(i), ci0 = chunk * 16 + 8 * h;
    _Float16* const dst = (_Float16*)p.dst + li * 8;
    if (co < p.cout && ci0 < p.cin) {
      const f32x4* const src = (const f32x4*)(p.src + ((int64_t)co * p.cin + ci0) * 9);
      float f[72];
#pragma unroll
      for (int q = 0; q < 18; ++q) { const f32x4 v = src[q]; f[4 * q] = v[0]; f[4 * q + 1] = v[1]; f[4 * q + 2] = v[2]; f[4 * q + 3] = v[3]; }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)f[e * 9 + t];
        *(half8*)(dst + (int64_t)t * 64 * 8) = o;
      }
    } else {
      const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 9; ++t) *(half8*)(dst + (int64_t)t * 64 * 8) = z;
    }
    return;
  }
#endif
  if (p.dtype == ESR_F16) pack_piece<_Float16>(p, li);
  else pack_piece<float>(p, li);
}

template <typename T>
__global__ void to_g32_kernel(const esr_layout p) {
  constexpr int CPG = DT<T>::CPG;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int g = blockIdx.z % p.g32.ngroups, b = blockIdx.z / p.g32.ngroups;
  if (x >= p.W) return;
  T v[CPG];
#pragma unroll
  for (int e = 0; e < CPG; ++e) {
    const int c = g * CPG + e;
    float f = 0.f;
    if (c < p.C) {
      f = p.nchw[(((int64_t)b * p.C + c) * p.H + y) * p.W + x];
      if (p.use_affine && c < 4) f = (f - p.mean_c[c]) * p.inv_std_c[c];
    }
    v[e] = (T)f;
  }
  char* dst = (char*)p.g32.ptr + b * p.g32.batch_stride + (int64_t)g * p.g32.group_stride +
              ((int64_t)(y + 1) * p.g32.wp + x + 1) * 32;
  const u32x4* s = (const u32x4*)v;
  ((u32x4*)dst)[0] = s[0];
  ((u32x4*)dst)[1] = s[1];
}

template <typename T>
__global__ void from_g32_kernel(const esr_layout p) {
  constexpr int CPG = DT<T>::CPG;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int g = blockIdx.z % p.g32.ngroups, b = blockIdx.z / p.g32.ngroups;
  if (x >= p.W) return;
  const char* src = (const char*)p.g32.ptr + b * p.g32.batch_stride + (int64_t)g * p.g32.group_stride +
                    ((int64_t)(y + 1) * p.g32.wp + x + 1) * 32;
  T v[CPG];
  ((u32x4*)v)[0] = ((const u32x4*)src)[0];
  ((u32x4*)v)[1] = ((const u32x4*)src)[1];
#pragma unroll
  for (int e = 0; e < CPG; ++e) {
    const int c = g * CPG + e;
    if (c < p.C) {
      float f = (float)v[e];
      if (p.use_affine && c < 4) f *= p.inv_std_c[c];   // adjoint of (x - mean) * inv_std
      float* const o = p.nchw + (((int64_t)b * p.C + c) * p.H + y) * p.W + x;
      *o = p.accumulate ? *o + f : f;
    }
  }
}

__global__ void noise_fill_kernel(const esr_noise_fill p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int co = blockIdx.z % ((p.C + 7) / 8), b = blockIdx.z / ((p.C + 7) / 8);
  if (x >= p.W) return;
  float z[8];
  philox_normal8((uint32_t)((b * p.H + y) * p.W + x), (uint32_t)co, p.layer, p.seed, z);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = co * 8 + e;
    if (c < p.C) p.dst[(((int64_t)b * p.C + c) * p.H + y) * p.W + x] = z[e];
  }
}

}  // namespace

extern "C" size_t esr_packed_weight_bytes(int32_t cout, int32_t cin, int32_t ks, int32_t dtype) {
  const int cpg = dtype == ESR_F16 ? 16 : 8;
  const size_t cbs = (cout + 31) / 32, chunks = (cin + cpg - 1) / cpg;
  return cbs * chunks * ks * ks * 1024;
}

extern "C" int esr_pack_conv_weights(const esr_pack* p, esr_stream_t stream) {
  if (!p || !p->src || !p->dst || p->cout <= 0 || p->cin <= 0 || (p->ks != 1 && p->ks != 3 && p->ks != 4) ||
      (p->ups_dgrad && !(p->transpose_flip && p->ks == 4)) || (p->ups_fwd && (p->transpose_flip || p->gather || p->ks != 3))) {
    esr_set_error("esr_pack_conv_weights: invalid arguments");
    return ESR_ERR_INVALID;
  }
  const int cpg = p->dtype == ESR_F16 ? 16 : 8;
  // packed geometry: rows = the packed conv's couts, K = its cins
  const int rows = p->transpose_flip ? p->cin : p->cout;
  const int kdim = p->transpose_flip ? p->cout : p->cin;
  const int nchunks = (kdim + cpg - 1) / cpg;
  const int64_t total = p->ups_fwd ? (int64_t)((rows + 31) / 32) * 4 * nchunks * 4 * 64 : (int64_t)((rows + 31) / 32) * nchunks * p->ks * p->ks * 64;
  const int blocks = (int)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (p->dtype == ESR_F16) hipLaunchKernelGGL(pack_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, *p, nchunks, total);
  else if (p->dtype == ESR_F32) hipLaunchKernelGGL(pack_kernel<float>, dim3(blocks), dim3(256), 0, st, *p, nchunks, total);
  else { esr_set_error("esr_pack_conv_weights: bad dtype"); return ESR_ERR_INVALID; }
  return esr_check_launch("pack_kernel");
}

extern "C" int64_t esr_pack_pieces(const esr_pack* p) {
  const int cpg = p->dtype == ESR_F16 ? 16 : 8;
  if (p->one_t) return 4 * 64;
  if (p->gather) return (int64_t)((p->dst_cout + 31) / 32) * ((p->cout + cpg - 1) / cpg) * 9 * 64;
  if (p->ups_fwd) return (int64_t)((p->cout + 31) / 32) * 4 * ((p->cin + cpg - 1) / cpg) * 4 * 64;
  const int rows = p->transpose_flip ? p->cin : p->cout;
  const int kdim = p->transpose_flip ? p->cout : p->cin;
  return (int64_t)((rows + 31) / 32) * ((kdim + cpg - 1) / cpg) * p->ks * p->ks * 64;
}

extern "C" int esr_pack_conv_weights_batch(const esr_pack_batch* p, esr_stream_t stream) {
  if (!p || !p->table || !p->piece_begin || p->n <= 0 || p->total_pieces <= 0) {
    esr_set_error("esr_pack_conv_weights_batch: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)((p->total_pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
  return esr_check_launch("pack_batch_kernel");
}

extern "C" int esr_convert_layout(const esr_layout* p, esr_stream_t stream) {
  if (!p || !p->nchw || !p->g32.ptr || p->B <= 0 || p->C <= 0 || p->H <= 0 || p->W <= 0 || p->g32.ngroups <= 0) {
    esr_set_error("esr_convert_layout: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((p->W + 63) / 64, p->H, p->B * p->g32.ngroups), block(64);
  if (p->dtype == ESR_F16) {
    if (p->to_g32) hipLaunchKernelGGL(to_g32_kernel<_Float16>, grid, block, 0, st, *p);
    else hipLaunchKernelGGL(from_g32_kernel<_Float16>, grid, block, 0, st, *p);
  } else if (p->dtype == ESR_F32) {
    if (p->to_g32) hipLaunchKernelGGL(to_g32_kernel<float>, grid, block, 0, st, *p);
    else hipLaunchKernelGGL(from_g32_kernel<float>, grid, block, 0, st, *p);
  } else { esr_set_error("esr_convert_layout: bad dtype"); return ESR_ERR_INVALID; }
  return esr_check_launch("layout_kernel");
}

extern "C" int esr_fill_noise(const esr_noise_fill* p, esr_stream_t stream) {
  if (!p || !p->dst || p->B <= 0 || p->C <= 0 || p->H <= 0 || p->W <= 0) {
    esr_set_error("esr_fill_noise: invalid arguments");
    return ESR_ERR_INVALID;
  }
  dim3 grid((p->W + 63) / 64, p->H, p->B * ((p->C + 7) / 8));
  hipLaunchKernelGGL(noise_fill_kernel, grid, dim3(64), 0, (hipStream_t)stream, *p);
  return esr_check_launch("noise_fill_kernel");
}

// ---- diagnostic: what the matrix pipe sustains under the package power limit (bench.py's `mfma_sustained_probe`) ----
// Every SIMD runs ONE wave issuing independent v_mfma_f32_32x32x16_f16 back to back from registers (no LDS, no memory
// in the loop); the operands are the caller's (random fp16 = the toggle rate of real data; zeros reach the quoted peak).
// clk2[0] / clk2[1]: shader-clock ticks / 100 MHz ticks the loop took on workgroup 0 (-> the clock it ran at).
namespace {
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(const half8* __restrict__ ops, float* sink, uint64_t* clk2, int iters) {
  __shared__ char pad[120 * 1024];                        // one workgroup per CU
  pad[threadIdx.x] = 0;
  const int lane = threadIdx.x;
  half8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = ops[i * 256 + lane]; b[i] = ops[(4 + i) * 256 + lane]; }
  probe_f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  uint64_t t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk2[0] = __builtin_amdgcn_s_memtime() - t0; clk2[1] = __builtin_amdgcn_s_memrealtime() - r0; }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][lane & 15];
  if (s == 123.456f) sink[0] = s + pad[5];
}
}  // namespace

extern "C" int esr_debug_mfma_probe(const void* operands, int32_t iters, uint64_t* clk2, float* sink, int32_t n_workgroups, esr_stream_t stream) {
  if (!operands || !clk2 || !sink || iters <= 0 || n_workgroups <= 0) {
    esr_set_error("esr_debug_mfma_probe: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)n_workgroups), dim3(256), 0, (hipStream_t)stream, (const half8*)operands, sink, clk2, (int)iters);
  return esr_check_launch("mfma_probe_kernel");
}
