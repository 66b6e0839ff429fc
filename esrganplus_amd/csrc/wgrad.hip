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
#include <cstdlib>
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

  // ---- deterministic form (esr_wgrad.partial): this workgroup's (image, column strip) slot of the partial arena,
  // tap-major like the fp16 kernels', plain stores; wgrad_reduce_kernel adds the slots up in slot order.
  // Otherwise: fp32 atomics into the OIHW gradient (and bias gradient).
  float* const slot = p.partial ? p.partial + (int64_t)blockIdx.x * p.partial_elems : nullptr;
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
      if (slot) {
        if (bkind[t] == 1) { if (p.dbias && cg == 0) slot[(int64_t)KS * KS * p.cout * p.cin + co] = v; }
        else if (ci < p.cin) slot[((int64_t)tap * p.cout + co) * p.cin + ci] = v;
        continue;
      }
      if (bkind[t] == 1) {
        if (p.dbias && cg == 0) atomicAdd(p.dbias + co, v);
      } else if (ci < p.cin) {
        atomicAdd(p.dw + ((int64_t)co * p.cin + ci) * (KS * KS) + tap, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fp16 path: the contraction over pixels runs on v_mfma_f32_32x32x16_f16.  Both operands need 8
// consecutive PIXELS of one channel per lane, while LDS holds [pixel][16 channels] (G32 order), so
// the fragments are fetched with the gfx950 transposing LDS read ds_read_b64_tr_b16: in a 16-lane
// group lane i supplies the address of 4 consecutive channels of pixel (i>>2) (quarter i&3 of that
// pixel's 32 bytes) and receives channel i of those 4 pixels.  Two reads = one 8-pixel fragment.
// Workgroup = 8 waves = NCO cout blocks x (8/NCO) blocks of 32 input channels; a wave keeps the 9
// (or 1) per-tap 32x32 accumulators of its (cout block, cin block) plus one bias accumulator, so a
// g fragment read feeds up to 10 MFMAs.
// ------------------------------------------------------------------------------------------------
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));

__device__ __forceinline__ u32x4 tr_frag(const char* lds_base, int off0, int off1) {
  // off0/off1: per-lane byte offsets of the two 4-pixel reads
  const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(lds_base + off0));
  const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(lds_base + off1));
  u32x4 r;
  r[0] = ((const uint32_t*)&a)[0]; r[1] = ((const uint32_t*)&a)[1];
  r[2] = ((const uint32_t*)&b)[0]; r[3] = ((const uint32_t*)&b)[1];
  return r;
}

struct Acc9 { f32x16 a[9]; };

// async global -> LDS copy of 16 bytes per lane (LDS-DMA): wave-uniform LDS base + lane*16
__device__ __forceinline__ void dma16(const char* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Every variant runs as 4-wave workgroups staged by LDS-DMA, two resident per CU (an earlier 8-wave
// form staged through registers: 112 KB of LDS per workgroup, spills, no overlap — 2.3x slower).
template <int S, bool UPS> struct Wg16Mode {
  static constexpr int NWV = 4;                        // waves per workgroup
};

// log2 of the column count one image gets in a packed tile (stride 2: at least 8, the 64 KB static LDS bound)
template <int S> __host__ __device__ constexpr int pack_wl(int W) { return (W <= 4 && S == 1) ? 2 : (W <= 8 ? 3 : 4); }
// PACK (feature maps at most 16 pixels wide — the discriminator's deep layers): the 32 tile columns are 2, 4 or 8
// IMAGES of width Wt = 16, 8, 4 side by side, each with its own halo columns in the input tile, instead of one
// image's mostly empty row; IW is then the widest such pitch (8 images of width 4; stride 2: 4 of width 8).
template <int KS, int S, bool UPS, int NCO, bool PACK = false>
struct Wg16Geo {
  static constexpr int TR = S == 2 ? 2 : 4, TC = 32;
  static constexpr int NCI = Wg16Mode<S, UPS>::NWV / NCO;
  static constexpr int IH = UPS ? TR / 2 + 2 : (TR - 1) * S + KS;
  static constexpr int IW = PACK ? (S == 2 ? 4 * (7 * S + KS) : 8 * (3 * S + KS)) : UPS ? TC / 2 + 2 : (TC - 1) * S + KS;
  static constexpr int G_BYTES = NCO * 2 * TR * TC * 32;
  static constexpr int IN_GROUP = IH * IW * 32;
  static constexpr int LDS_BYTES = G_BYTES + NCI * 2 * IN_GROUP;
};

// (bx, by, bz) = the workgroup's coordinates in THIS conv's grid (the batched launch packs the grids
// of several convs into one 1-D grid)
template <int KS, int S, bool UPS, int NCO, int T0, int NT, bool PACK = false>
__device__ __forceinline__ void wgrad16_body(const esr_wgrad& p, const int rows_arg, char* const smem,
                                             const int bx, const int by, const int bz) {
  static_assert(!PACK || !UPS, "packed tiles: plain and stride-2 convs");
  // taps [T0, T0+NT) of the KSxKS kernel are accumulated by this launch (4x4 kernels: two launches
  // of 8 taps, keeping the accumulators within the register file)
  constexpr int TR = S == 2 ? 2 : 4, TC = 32, NTAP = KS * KS, PAD = (KS - 1) / 2;
  constexpr int NWV = Wg16Mode<S, UPS>::NWV, NTH = NWV * 64;
  constexpr int NCI = NWV / NCO;                      // cin blocks (of 32 channels) per workgroup
  constexpr int IH = UPS ? TR / 2 + 2 : (TR - 1) * S + KS, IWMAX = Wg16Geo<KS, S, UPS, NCO, PACK>::IW;
  static_assert(NT <= 9 && T0 + NT <= NTAP, "tap range");
  constexpr int G_BYTES = NCO * 2 * TR * TC * 32;     // [cout group][row][col][32 B]
  // packed tiles: wl = log2 of the per-image column count, ipt images per tile, iwimg input columns per image;
  // the input tile's pitch IW is then a run-time (wave-uniform) number, at most IWMAX
  const int wl = !PACK ? 5 : pack_wl<S>(p.W);
  const int ipt = 32 >> wl, iwimg = ((1 << wl) - 1) * S + KS;
  const int IW = PACK ? ipt * iwimg : IWMAX;
  const int IN_GROUP = IH * IW * 32;
  char* const lg = smem;
  char* const li = smem + G_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave % NCO, wci = wave / NCO;       // this wave's cout block / cin block in the WG
  const int strips = (p.W + TC - 1) / TC;
  // rows_arg = rows per workgroup | images per workgroup << 16 (many small images: one workgroup walks
  // several of them, so the number of dW atomic rounds stays ~64 per conv instead of one per image)
  const int rows_per_wg = rows_arg & 0xFFFF, ipw = (rows_arg >> 16) > 0 ? (rows_arg >> 16) : 1;
  const int rchunks = (p.H + rows_per_wg - 1) / rows_per_wg;
  const int sx = bx % strips, rc = (bx / strips) % rchunks, b0 = (bx / (strips * rchunks)) * ipw;
  const int bend = min(p.B, b0 + ipw);
  const int ox0 = sx * TC;
  const int cb = bz * NCO + wco;              // 32-cout block
  const int cib = by * NCI + wci;             // 32-cin block  (= G32 groups 2*cib, 2*cib+1)
  const int ngin = p.in.ngroups;
  const bool active = cb * 32 < p.cout && 2 * cib < ngin;

  // per-lane transposed-read geometry: lane -> (pixel-in-run, 8-byte quarter, group half)
  const int i16 = lane & 15, jrow = i16 >> 2, q = i16 & 3, ghalf = (lane >> 4) & 1, kg = lane >> 5;
  const int pk = 8 * kg + jrow;                       // pixel of the 16-pixel run for read 0 (+4 for read 1)

  Acc9 acc;
  float bsum = 0.f;     // bias gradient: this lane's share of sum_px g[co = cb*32 + lane%32][px]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc.a[t][e] = 0.f;

  const char* gbase = (const char*)p.g.ptr + b0 * p.g.batch_stride;
  const char* ibase = (const char*)p.in.ptr + b0 * p.in.batch_stride;
  const int y_begin = rc * rows_per_wg, y_end = min(p.H, y_begin + rows_per_wg);

  constexpr int GS = NCO * 2 * TR * TC * 2;                 // 16-byte slots of the g tile
  // input groups past the tensor's last one are never read by an active wave: not staged
  const int in_groups = min(NCI * 2, ngin - (int)by * NCI * 2);
  const int IS = in_groups > 0 ? in_groups * IH * IW * 2 : 0;
  // LDS-DMA staging: slot s (16 bytes) of the linear [g tile | input tile] image comes straight from
  // global memory; no bounds tests — the G32 invariant (zero ring, nothing ever written outside the
  // image) supplies the padding, lanes past the last existing group are masked off.
  constexpr int NP = (GS + NCI * 2 * IH * IWMAX * 2 + NTH - 1) / NTH;
  // packed tiles: the slot -> source map (image of the tile, row, column) costs integer divisions by run-time
  // numbers, so it is built once: poff = byte offset from the first image of the tile at tile row 0 (-1: no
  // slot), pzero = offset of a zero pixel of the same plane (for images past the batch) | image-in-tile
  int poff[PACK ? NP : 1], pzero[PACK ? NP : 1];
  if constexpr (PACK) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int s = tid + NTH * k;
      poff[k] = -1; pzero[k] = 0;
      if (s < GS) {
        const int half = s & 1, px = (s >> 1) % (TR * TC), g = (s >> 1) / (TR * TC);
        const int gg = (bz * NCO) * 2 + g, c = px % TC, bi = c >> wl, x = c & ((1 << wl) - 1);
        if (gg < p.g.ngroups) {
          pzero[k] = (int)(gg * p.g.group_stride) + half * 16 + bi;       // padded (0, 0): ring corner
          poff[k] = (int)(bi * p.g.batch_stride + gg * p.g.group_stride) + ((px / TC + 1) * p.g.wp + x + 1) * 32 + half * 16;
        }
      } else if (s - GS < IS) {
        const int t = s - GS;
        const int half = t & 1, px = (t >> 1) % (IH * IW), g = (t >> 1) / (IH * IW);
        const int gg = by * NCI * 2 + g, colp = px % IW, bi = colp / iwimg, xi = colp - bi * iwimg;
        pzero[k] = (int)(gg * p.in.group_stride) + half * 16 + bi;
        poff[k] = (int)(bi * p.in.batch_stride + gg * p.in.group_stride) + ((px / IW + 1 - PAD) * p.in.wp + 1 - PAD + xi) * 32 + half * 16;
      }
    }
  }
  int soff[PACK ? 1 : NP];
  if constexpr (!PACK) {
    const int ix0 = UPS ? ox0 / 2 : ox0 * S + 1 - PAD;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int s = tid + NTH * k;
      soff[k] = -1;
      if (s < GS) {
        const int half = s & 1, px = (s >> 1) % (TR * TC), g = (s >> 1) / (TR * TC);
        const int gg = (bz * NCO) * 2 + g;
        if (gg < p.g.ngroups)
          soff[k] = (int)(gg * p.g.group_stride) + ((px / TC + 1) * p.g.wp + ox0 + px % TC + 1) * 32 + half * 16;
      } else if (s - GS < IS) {
        const int t = s - GS;
        const int half = t & 1, px = (t >> 1) % (IH * IW), g = (t >> 1) / (IH * IW);
        const int gg = by * NCI * 2 + g;
        soff[k] = (int)(gg * p.in.group_stride) + ((px / IW + (UPS ? 0 : 1 - PAD)) * p.in.wp + ix0 + px % IW) * 32 + half * 16;
      }
    }
  }
  auto dma_tile = [&](int oy0, int b) __attribute__((always_inline)) {
    if constexpr (PACK) {
      const int grow = oy0 * p.g.wp * 32, irow = oy0 * S * p.in.wp * 32;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (poff[k] < 0) continue;
        char* const dst = smem + (wave * 64 + NTH * k) * 16;
        const bool isg = tid + NTH * k < GS;
        const bool inb = b + (pzero[k] & 15) < bend;
        const char* src = (isg ? gbase : ibase) + (inb ? poff[k] + (isg ? grow : irow) : (pzero[k] & ~15));
        dma16(src, dst);
      }
      return;
    }
    // the slot -> source map is the same for every tile of the workgroup (soff, built once below): per tile only
    // the row base moves — the per-slot divisions and 64-bit multiplies used to cost as many issue cycles per
    // tile as its MFMAs
    const char* const grow = gbase + (int64_t)oy0 * p.g.wp * 32;
    const char* const irow = ibase + (int64_t)(UPS ? oy0 / 2 : oy0 * S) * p.in.wp * 32;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      char* const dst = smem + (wave * 64 + NTH * k) * 16;         // wave-uniform; lane*16 is implicit
      if (soff[k] >= 0) dma16((tid + NTH * k < GS ? grow : irow) + soff[k], dst);
    }
  };
  // packed: per-lane column terms of the two transposed reads of each 16-pixel half run
  int cm[PACK ? 4 : 1];
  if constexpr (PACK) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = 16 * (u >> 1) + pk + 4 * (u & 1);
      cm[u] = ((c >> wl) * iwimg + (c & ((1 << wl) - 1)) * S) * 32;
    }
  }
  const int bstep = PACK ? ipt : 1;
  for (int b = b0; b < bend; b += bstep, gbase += bstep * p.g.batch_stride, ibase += bstep * p.in.batch_stride)
  for (int oy0 = y_begin; oy0 < y_end; oy0 += TR) {
    __syncthreads();                 // every wave is done reading the previous tile
    dma_tile(oy0, b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                 // (the CU's other workgroup computes meanwhile)
    if (!active) continue;
    const char* lgw = lg + (wco * 2 + ghalf) * TR * TC * 32 + q * 8;
    const char* liw = li + (wci * 2 + ghalf) * IN_GROUP + q * 8;
#pragma unroll
    for (int r = 0; r < TR; ++r) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int c0 = 16 * hf + pk;                    // column of this lane's first pixel (read 0)
        const u32x4 af = tr_frag(lgw, (r * TC + c0) * 32, (r * TC + c0 + 4) * 32);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int kh = (T0 + tt) / KS, kw = (T0 + tt) % KS;
          int o0, o1;
          if (UPS) {
            const int rr = ((r + kh - 1) >> 1) + 1;
            o0 = (rr * IW + ((c0 + kw - 1) >> 1) + 1) * 32;
            o1 = (rr * IW + ((c0 + 4 + kw - 1) >> 1) + 1) * 32;
          } else if (PACK) {
            o0 = ((r * S + kh) * IW + kw) * 32 + cm[2 * hf];
            o1 = ((r * S + kh) * IW + kw) * 32 + cm[2 * hf + 1];
          } else {
            o0 = ((r * S + kh) * IW + c0 * S + kw) * 32;
            o1 = o0 + 4 * S * 32;
          }
          const u32x4 bf = tr_frag(liw, o0, o1);
          acc.a[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, af), __builtin_bit_cast(half8, bf), acc.a[tt], 0, 0, 0);
        }
        if (T0 == 0 && cib == 0 && p.dbias)
        {
          // A-fragment layout: lane holds 8 consecutive pixels (k) of cout row lane%32 — summing them on
          // the VALU costs one register instead of a 16-register all-ones-column accumulator
          const half8 hv = __builtin_bit_cast(half8, af);
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum += (float)hv[e];
        }
      }
    }
  }
  if (!active) return;
  const int n = lane & 31, ci = cib * 32 + n;
  if (p.partial) {
    // ---- deterministic form: this workgroup's own slot (spatial split bx) of the partial arena, tap-major,
    // plain stores; wgrad_reduce_kernel adds the slots up in a fixed order.  `partial_elems` carries the slot
    // stride here (set by the launcher): [NTAP][cout][cin] floats + [cout] bias sums.
    float* const slot = p.partial + (int64_t)bx * p.partial_elems;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int t = T0 + tt;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
        if (co < p.cout && ci < p.cin) slot[((int64_t)t * p.cout + co) * p.cin + ci] = acc.a[tt][e] * p.scale;
      }
    }
    if (T0 == 0 && cib == 0 && p.dbias) {
      // lanes l and l+32 hold the two k halves of row l%32: fold them here, one store per row
      const float other = __shfl_xor(bsum, 32);
      const int co = cb * 32 + (lane & 31);
      if (lane < 32 && co < p.cout) slot[(int64_t)NTAP * p.cout * p.cin + co] = (bsum + other) * p.scale;
    }
    return;
  }
  // ---- fp32 atomics into dW[co][ci][kh][kw] / dbias[co]
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const int t = T0 + tt;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
      if (co < p.cout && ci < p.cin) {
        float* d = p.tap_major ? p.dw + ((int64_t)t * p.cout + co) * p.cin + ci
                               : p.dw + ((int64_t)co * p.cin + ci) * NTAP + t;
        atomicAdd(d, acc.a[tt][e] * p.scale);
      }
    }
  }
  if (T0 == 0 && cib == 0 && p.dbias) {
    const int co = cb * 32 + (lane & 31);          // lanes l and l+32 hold the two k halves of row l%32
    if (co < p.cout) atomicAdd(p.dbias + co, bsum * p.scale);
  }
}


template <int KS, int S, bool UPS, int NCO, int T0, int NT, bool PACK = false>
__global__ __launch_bounds__((Wg16Mode<S, UPS>::NWV * 64), 2) void wgrad16_kernel(const esr_wgrad p, int rows_per_wg) {
  __shared__ __attribute__((aligned(16))) char smem[Wg16Geo<KS, S, UPS, NCO, PACK>::LDS_BYTES];
  wgrad16_body<KS, S, UPS, NCO, T0, NT, PACK>(p, rows_per_wg, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

constexpr int WG_BATCH_MAX = 18;   // an RRDB: 3 x 6 (kernel arguments: 3 KB of 4)
// ---- several independent 3x3/s1 and 1x1 wgrads in ONE launch.  At training sizes (16 x 32x32) one
// conv's wgrad is ~64 workgroups of mostly idle waves and ~25 us of pure latency; the six convs of a
// residual dense block (same saved input, six gradient slices) fill the chip together.
struct WgradBatch {
  int32_t n;
  int32_t start[WG_BATCH_MAX + 1];   // first linear workgroup of entry i
  int32_t gx[WG_BATCH_MAX], gy[WG_BATCH_MAX];
  int32_t rows[WG_BATCH_MAX];
  int32_t kind[WG_BATCH_MAX];        // 0: 3x3 NCO=1   1: 3x3 NCO=2   2: 1x1 NCO=1   3: 1x1 NCO=2
  esr_wgrad w[WG_BATCH_MAX];
};

__global__ __launch_bounds__(256, 2) void wgrad16_batch_kernel(const WgradBatch pb) {
  constexpr int L0 = Wg16Geo<3, 1, false, 1>::LDS_BYTES, L1 = Wg16Geo<3, 1, false, 2>::LDS_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[L0 > L1 ? L0 : L1];
  int i = 0;
#pragma unroll
  for (int k = 1; k < WG_BATCH_MAX; ++k)
    if (k < pb.n && (int)blockIdx.x >= pb.start[k]) i = k;
  const int l = blockIdx.x - pb.start[i];
  const int bx = l % pb.gx[i], by = (l / pb.gx[i]) % pb.gy[i], bz = l / (pb.gx[i] * pb.gy[i]);
  const esr_wgrad& p = pb.w[i];
  switch (pb.kind[i]) {
    case 0: wgrad16_body<3, 1, false, 1, 0, 9>(p, pb.rows[i], smem, bx, by, bz); break;
    case 1: wgrad16_body<3, 1, false, 2, 0, 9>(p, pb.rows[i], smem, bx, by, bz); break;
    case 2: wgrad16_body<1, 1, false, 1, 0, 1>(p, pb.rows[i], smem, bx, by, bz); break;
    default: wgrad16_body<1, 1, false, 2, 0, 1>(p, pb.rows[i], smem, bx, by, bz); break;
  }
}

// ---- stage 2 of the deterministic form: dw[elem] += sum over the spatial splits, in split order
struct WgradReduce {
  int32_t n;
  int64_t begin[WG_BATCH_MAX + 1];     // first linear element of entry i
  const float* partial[WG_BATCH_MAX];
  int64_t stride[WG_BATCH_MAX];
  int32_t nsplit[WG_BATCH_MAX], cout[WG_BATCH_MAX], cin[WG_BATCH_MAX], ntap[WG_BATCH_MAX], tap_major[WG_BATCH_MAX];
  float* dw[WG_BATCH_MAX];
  float* dbias[WG_BATCH_MAX];
};
__global__ void wgrad_reduce_kernel(const WgradReduce r) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= r.begin[r.n]) return;
  int i = 0;
#pragma unroll
  for (int k = 1; k < WG_BATCH_MAX; ++k)
    if (k < r.n && idx >= r.begin[k]) i = k;
  const int64_t k = idx - r.begin[i];                 // element of the slot: tap-major weights, then bias sums
  const float* src = r.partial[i] + k;
  float s = 0.f;
  // slot order, eight loads in flight (a conv with few weights and many splits — the 64 -> 3 HR conv: 256 slots of
  // 1.7 K floats — is a handful of workgroups walking the slots: one dependent load at a time took 146 us)
  const int ns = r.nsplit[i];
  const int64_t st = r.stride[i];
  int sp = 0;
  for (; sp + 8 <= ns; sp += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sp + u) * st];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sp < ns; ++sp) s += src[(int64_t)sp * st];
  const int64_t nw = (int64_t)r.ntap[i] * r.cout[i] * r.cin[i];
  if (k >= nw) { r.dbias[i][k - nw] += s; return; }
  if (r.tap_major[i]) { r.dw[i][k] += s; return; }
  const int ci = (int)(k % r.cin[i]);
  const int64_t q = k / r.cin[i];
  const int co = (int)(q % r.cout[i]), t = (int)(q / r.cout[i]);
  r.dw[i][((int64_t)co * r.cin[i] + ci) * r.ntap[i] + t] += s;
}
static int64_t slot_elems(const esr_wgrad& p) {
  const int64_t n = (int64_t)p.ks * p.ks * p.cout * p.cin + (p.dbias ? p.cout : 0);
  return (n + 3) & ~(int64_t)3;
}
static void reduce_entry(WgradReduce& r, int i, const esr_wgrad& p, const float* partial, int nsplit) {
  r.partial[i] = partial; r.stride[i] = slot_elems(p); r.nsplit[i] = nsplit;
  r.cout[i] = p.cout; r.cin[i] = p.cin; r.ntap[i] = p.ks * p.ks; r.tap_major[i] = p.tap_major;
  r.dw[i] = p.dw; r.dbias[i] = p.dbias;
  r.begin[i + 1] = r.begin[i] + (int64_t)p.ks * p.ks * p.cout * p.cin + (p.dbias ? p.cout : 0);
}
static int launch_reduce(WgradReduce& r, int n, hipStream_t st) {
  r.n = n;
  for (int k = n; k < WG_BATCH_MAX; ++k) {
    r.begin[k + 1] = r.begin[n]; r.partial[k] = r.partial[0]; r.stride[k] = 0; r.nsplit[k] = 0; r.cout[k] = r.cin[k] = r.ntap[k] = 1;
    r.tap_major[k] = 1; r.dw[k] = r.dw[0]; r.dbias[k] = r.dbias[0];
  }
  const int64_t total = r.begin[n];
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r);
  return esr_check_launch("wgrad_reduce_kernel");
}

// grid shape of one conv's fp16 wgrad (shared by the single and the batched launch)
static int max_rows() {
  static const int v = [] { const char* e = getenv("ESR_WGRAD_MAX_ROWS"); return e ? atoi(e) : 32; }();
  return v;
}
struct Wg16Grid { int nco, gx, gy, gz, rows; };
// packed tiles (Wg16Geo): maps of at most 16 columns, more than one image (ESR_WGRAD_PACK=0 switches it off)
template <int S, bool UPS> static bool wgrad16_packed(const esr_wgrad& p) {
  static const bool on = [] { const char* e = getenv("ESR_WGRAD_PACK"); return !e || atoi(e) != 0; }();
  return on && !UPS && p.W <= 16 && p.B > 1 && (S == 2 || (p.ks == 3 && p.cout > 32));   // (two cout blocks per workgroup: LDS)
}   // rows = rows per workgroup | images per workgroup << 16
template <int S, bool UPS>
Wg16Grid wgrad16_grid(const esr_wgrad& p, int64_t min_wgs, int min_rows, int cap_rows) {
  const int strips = (p.W + 31) / 32;
  const int coblocks = (p.cout + 31) / 32;
  const int ciblocks = (p.in.ngroups + 1) / 2;
  Wg16Grid g;
  g.nco = (coblocks >= 2 || S == 2) ? 2 : 1;           // stride 2: LDS only fits NCO=2
  const int nci = Wg16Mode<S, UPS>::NWV / g.nco;
  g.gy = (ciblocks + nci - 1) / nci;
  g.gz = (coblocks + g.nco - 1) / g.nco;
  // Every workgroup ends with 8 waves x 9 taps x 1024 fp32 atomics, so use as FEW spatial splits as
  // still give ~min_wgs workgroups: start from whole column strips and halve only while the grid is tiny.
  int rows = ((p.H + 3) / 4) * 4;
  if (rows > cap_rows) rows = cap_rows;       // bound the serial load->LDS->MFMA iterations of one workgroup
  // With the two-stage reduction a spatial split costs one partial tile, not a set of atomics: the kernel is fastest
  // with ~32 workgroups per CU to hide each other's load -> wait -> MFMA latency (tools/train_op_times.py, 16 images,
  // cap 32 / 16 / 8 / 4 rows: 512^2 64->64: 0.46 / 0.36 / 0.41 / 0.51 ms; 256^2 up-conv: 0.35 / 0.21 / 0.15 / 0.14 ms)
  {
    const int64_t base = (int64_t)p.B * strips * g.gy * g.gz;
    int want = (int)(((int64_t)p.H * base / 8192) / 4 * 4);
    if (want < 4) want = 4;
    if (rows > want) rows = want;
  }
  while (rows > min_rows && (int64_t)p.B * strips * ((p.H + rows - 1) / rows) * g.gy * g.gz < min_wgs) rows = ((rows / 2 + 3) / 4) * 4;
  const int rchunks = (p.H + rows - 1) / rows;
  int ipw = wgrad16_packed<S, UPS>(p) ? 32 >> pack_wl<S>(p.W) : 1;   // images per workgroup (packed: >= one tile of them)
  if (ipw > p.B) ipw = p.B;
  while (ipw < p.B && ipw < 0x7FFF &&
         (int64_t)((p.B + 2 * ipw - 1) / (2 * ipw)) * strips * rchunks * g.gy * g.gz >= min_wgs) ipw *= 2;
  g.rows = rows | (ipw << 16);
  g.gx = ((p.B + ipw - 1) / ipw) * strips * rchunks;
  return g;
}

template <int KS, int S, bool UPS, int T0, int NT>
int launch_wgrad16(const esr_wgrad& p_in, hipStream_t st, bool reduce = true) {
  const Wg16Grid g = wgrad16_grid<S, UPS>(p_in, 64, 8, max_rows());
  dim3 grid(g.gx, g.gy, g.gz);
  esr_wgrad p = p_in;
  if (p.partial) {
    if ((int64_t)g.gx * slot_elems(p) > p.partial_elems) {
      esr_set_error("wgrad: partial arena too small (%lld floats, need %lld: esr_wgrad_workspace_elems)",
                    (long long)p.partial_elems, (long long)((int64_t)g.gx * slot_elems(p)));
      return ESR_ERR_INVALID;
    }
    p.partial_elems = slot_elems(p);              // the kernel reads the slot stride here
  }
  if constexpr (!UPS && (S == 2 || KS == 3)) {
    if (wgrad16_packed<S, UPS>(p_in)) {
      hipLaunchKernelGGL((wgrad16_kernel<KS, S, UPS, 2, T0, NT, true>), grid, dim3(Wg16Mode<S, UPS>::NWV * 64), 0, st, p, g.rows);
      const int rc = esr_check_launch("wgrad16_kernel");
      if (rc || !p.partial || !reduce) return rc;
      WgradReduce r;
      r.begin[0] = 0;
      reduce_entry(r, 0, p_in, p.partial, g.gx);
      return launch_reduce(r, 1, st);
    }
  }
  if constexpr (S == 2) {
    hipLaunchKernelGGL((wgrad16_kernel<KS, S, UPS, 2, T0, NT>), grid, dim3(Wg16Mode<S, UPS>::NWV * 64), 0, st, p, g.rows);
  } else {
    if (g.nco == 2) hipLaunchKernelGGL((wgrad16_kernel<KS, S, UPS, 2, T0, NT>), grid, dim3(Wg16Mode<S, UPS>::NWV * 64), 0, st, p, g.rows);
    else hipLaunchKernelGGL((wgrad16_kernel<KS, S, UPS, 1, T0, NT>), grid, dim3(Wg16Mode<S, UPS>::NWV * 64), 0, st, p, g.rows);
  }
  const int rc = esr_check_launch("wgrad16_kernel");
  if (rc || !p.partial || !reduce) return rc;
  WgradReduce r;
  r.begin[0] = 0;
  reduce_entry(r, 0, p_in, p.partial, g.gx);
  return launch_reduce(r, 1, st);
}
// spatial splits (= partial slots) the single launch of this conv uses
template <int S, bool UPS> static int single_splits(const esr_wgrad& p) { return wgrad16_grid<S, UPS>(p, 64, 8, max_rows()).gx; }
static int64_t single_partial_elems(const esr_wgrad& p) {
  if (p.dtype == ESR_F32) return (int64_t)p.B * ((p.W + 31) / 32) * slot_elems(p);   // wgrad_kernel: one slot per (image, strip)
  if (p.dtype != ESR_F16) return 0;
  int gx = 0;
  if (p.ks == 4 && p.stride == 2 && !p.upsample) gx = single_splits<2, false>(p);
  else if (p.stride == 1 && (p.ks == 3 || p.ks == 1)) gx = p.upsample ? single_splits<1, true>(p) : single_splits<1, false>(p);
  return (int64_t)gx * slot_elems(p);
}

template <typename T, int KS, int S, bool UPS>
int launch_wgrad(const esr_wgrad& p, hipStream_t st) {
  const int strips = (p.W + 31) / 32;
  dim3 grid(p.B * strips, p.in.ngroups, (p.cout + 31) / 32);
  esr_wgrad q = p;
  // two-stage reduction when the caller gave a partial arena that holds one slot per (image, strip); else atomics
  const bool det = q.partial && (int64_t)grid.x * slot_elems(p) <= q.partial_elems && !q.tap_major;
  if (det) q.partial_elems = slot_elems(p); else q.partial = nullptr;
  hipLaunchKernelGGL((wgrad_kernel<T, KS, S, UPS>), grid, dim3(256), 0, st, q);
  const int rc = esr_check_launch("wgrad_kernel");
  if (rc || !det) return rc;
  WgradReduce r;
  r.begin[0] = 0;
  reduce_entry(r, 0, p, q.partial, (int)grid.x);
  return launch_reduce(r, 1, st);
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

namespace {
__global__ void unpermute_kernel(const esr_unpermute p) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.total) return;
  int lo = 0, hi = p.n - 1;                      // last entry with elem_begin <= idx
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (p.table[mid].elem_begin <= idx) lo = mid; else hi = mid - 1;
  }
  const esr_unperm_entry e = p.table[lo];
  const int64_t k = idx - e.elem_begin;          // OIHW-linear index inside this conv
  const int t = (int)(k % e.ntap);
  const int64_t r = k / e.ntap;
  const int ci = (int)(r % e.cin), co = (int)(r / e.cin);
  p.dst[e.dst_off + k] = p.src[e.src_off + ((int64_t)t * e.cout + co) * e.cin + ci];
}
// one thread per (cout, cin) pair: for every tap the lanes of a wave read consecutive cin (coalesced), each lane writes
// its ntap adjacent OIHW elements; one table search per pair instead of per element
__global__ void unpermute_pairs_kernel(const esr_unpermute p) {
  const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (idx >= p.n_pairs) return;
  int lo = 0, hi = p.n - 1;                      // last entry with pair_begin <= idx
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (p.table[mid].pair_begin <= idx) lo = mid; else hi = mid - 1;
  }
  const esr_unperm_entry e = p.table[lo];
  const int k = idx - e.pair_begin;              // co * cin + ci
  const float* src = p.src + e.src_off + k;
  float* dst = p.dst + e.dst_off + (int64_t)k * e.ntap;
  const int64_t plane = (int64_t)e.cout * e.cin;
  if (e.ntap == 9) {
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = src[t * plane];
#pragma unroll
    for (int t = 0; t < 9; ++t) dst[t] = v[t];
  } else {
    for (int t = 0; t < e.ntap; ++t) dst[t] = src[t * plane];
  }
}
}  // namespace

extern "C" int esr_grad_unpermute(const esr_unpermute* p, esr_stream_t stream) {
  if (!p || !p->table || p->n <= 0 || p->total <= 0 || !p->src || !p->dst) {
    esr_set_error("esr_grad_unpermute: invalid arguments");
    return ESR_ERR_INVALID;
  }
  if (p->n_pairs > 0) {
    hipLaunchKernelGGL(unpermute_pairs_kernel, dim3((unsigned)((p->n_pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
    return esr_check_launch("unpermute_pairs_kernel");
  }
  hipLaunchKernelGGL(unpermute_kernel, dim3((unsigned)((p->total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p);
  return esr_check_launch("unpermute_kernel");
}

extern "C" int esr_conv_wgrad(const esr_wgrad* p, esr_stream_t stream) {
  if (!p || !p->g.ptr || !p->in.ptr || !p->dw || p->B <= 0 || p->H <= 0 || p->W <= 0 || p->cout <= 0 || p->cin <= 0) {
    esr_set_error("esr_conv_wgrad: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (p->tap_major && p->dtype != ESR_F16) {
    esr_set_error("esr_conv_wgrad: tap_major only with the fp16 kernels");
    return ESR_ERR_INVALID;
  }
  if (p->dtype == ESR_F16) {
    if (p->ks == 3 && p->stride == 1 && !p->upsample) return launch_wgrad16<3, 1, false, 0, 9>(*p, st);
    if (p->ks == 3 && p->stride == 1 && p->upsample) return launch_wgrad16<3, 1, true, 0, 9>(*p, st);
    if (p->ks == 1 && p->stride == 1 && !p->upsample) return launch_wgrad16<1, 1, false, 0, 1>(*p, st);
    if (p->ks == 4 && p->stride == 2 && !p->upsample) {   // discriminator: two launches of 8 taps (one reduce)
      const int rc = launch_wgrad16<4, 2, false, 0, 8>(*p, st, false);
      return rc ? rc : launch_wgrad16<4, 2, false, 8, 8>(*p, st, true);
    }
    return dispatch_wgrad<_Float16>(*p, st);
  }
  if (p->dtype == ESR_F32) return dispatch_wgrad<float>(*p, st);
  esr_set_error("esr_conv_wgrad: bad dtype %d", p->dtype);
  return ESR_ERR_INVALID;
}

// spatial splits per conv inside a batched launch: every split costs a full set of dW atomics, and the
// batch as a whole (not each conv) has to fill the chip
static int batch_min_wgs() {
  static const int v = [] { const char* e = getenv("ESR_WGRAD_MIN_WGS"); return e ? atoi(e) : 64; }();
  return v;
}

static bool wgrad_batchable(const esr_wgrad& p) {
  return p.dtype == ESR_F16 && p.stride == 1 && !p.upsample && (p.ks == 3 || p.ks == 1) && p.g.ptr && p.in.ptr &&
         p.dw && p.B > 0 && p.H > 0 && p.W > 0 && p.cout > 0 && p.cin > 0;
}

extern "C" int esr_conv_wgrad_multi(const esr_wgrad* items, int32_t n, esr_stream_t stream) {
  if (!items || n <= 0) { esr_set_error("esr_conv_wgrad_multi: invalid arguments"); return ESR_ERR_INVALID; }
  int i = 0;
  while (i < n) {
    // greedily pack consecutive batchable entries; anything else goes through the single launch
    if (!wgrad_batchable(items[i])) {
      const int rc = esr_conv_wgrad(&items[i], stream);
      if (rc) return rc;
      ++i;
      continue;
    }
    int m = 0;
    while (i + m < n && m < WG_BATCH_MAX && wgrad_batchable(items[i + m])) ++m;
    if (m == 1) {
      const int rc = esr_conv_wgrad(&items[i], stream);
      if (rc) return rc;
      ++i;
      continue;
    }
    WgradBatch pb;
    WgradReduce red;
    red.begin[0] = 0;
    const bool det = items[i].partial != nullptr;
    int64_t used = 0;                                  // floats of the partial arena handed out
    pb.n = m;
    int total = 0;
    for (int k = 0; k < m; ++k) {
      const esr_wgrad& p = items[i + k];
      if ((p.partial != nullptr) != det || (det && p.partial != items[i].partial)) {
        esr_set_error("esr_conv_wgrad_multi: the wgrads of one run must share one partial arena (or none)");
        return ESR_ERR_INVALID;
      }
      // batched launch: every spatial split costs a full set of dW atomics (~5 us per million), so take
      // the LARGEST row chunk that still gives each conv ~64 workgroups (6 convs fill the chip once),
      // and never less than 32 rows
      const Wg16Grid g = wgrad16_grid<1, false>(p, batch_min_wgs(), 32, 1 << 30);
      pb.start[k] = total;
      pb.gx[k] = g.gx; pb.gy[k] = g.gy; pb.rows[k] = g.rows;
      pb.kind[k] = (p.ks == 3 ? 0 : 2) + (g.nco == 2 ? 1 : 0);
      pb.w[k] = p;
      if (det) {
        pb.w[k].partial = p.partial + used;
        pb.w[k].partial_elems = slot_elems(p);
        reduce_entry(red, k, p, pb.w[k].partial, g.gx);
        used += (int64_t)g.gx * slot_elems(p);
        if (used > p.partial_elems) {
          esr_set_error("wgrad: partial arena too small (%lld floats, need >= %lld: esr_wgrad_workspace_elems)",
                        (long long)p.partial_elems, (long long)used);
          return ESR_ERR_INVALID;
        }
      }
      total += g.gx * g.gy * g.gz;
    }
    for (int k = m; k <= WG_BATCH_MAX; ++k) pb.start[k] = total;
    for (int k = m; k < WG_BATCH_MAX; ++k) { pb.gx[k] = pb.gy[k] = 1; pb.rows[k] = 8; pb.kind[k] = 0; pb.w[k] = items[i]; }
    hipLaunchKernelGGL(wgrad16_batch_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, pb);
    int rc = esr_check_launch("wgrad16_batch_kernel");
    if (rc == ESR_OK && det) rc = launch_reduce(red, m, (hipStream_t)stream);
    if (rc) return rc;
    i += m;
  }
  return ESR_OK;
}

// floats of partial arena one esr_conv_wgrad_multi call on these items needs (same grouping as above)
extern "C" int64_t esr_wgrad_run_partial_elems(const esr_wgrad* items, int32_t n) {
  int64_t need = 0;
  int i = 0;
  while (items && i < n) {
    int m = 0;
    if (wgrad_batchable(items[i]))
      while (i + m < n && m < WG_BATCH_MAX && wgrad_batchable(items[i + m])) ++m;
    if (m <= 1) {
      const int64_t e = single_partial_elems(items[i]);
      if (e > need) need = e;
      ++i;
      continue;
    }
    int64_t used = 0;
    for (int k = 0; k < m; ++k) used += (int64_t)wgrad16_grid<1, false>(items[i + k], batch_min_wgs(), 32, 1 << 30).gx * slot_elems(items[i + k]);
    if (used > need) need = used;
    i += m;
  }
  return need;
}
