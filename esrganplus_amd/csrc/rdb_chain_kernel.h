// rdb_fused.hip — input-stationary, persistent ResidualDenseBlock_5C chain on the gfx950 matrix cores.
//
// Replaces, for a CHAIN of dense blocks (the RRDB trunk of RRDBNet, architecture.py:57-59, i.e.
// 3 x nb ResidualDenseBlock_5C, block.py:232-268, with the RRDB tails of block.py:287-291 /
// test_image/block.py:252-256), what conv_mfma.hip runs as 5 launches per block.
//
// Why another kernel.  Layer by layer every conv of a block re-reads its whole concat prefix:
// (64+96+128+160+192) = 640 channel-reads per pixel for 192 channels of new data, all of it from beyond
// the 4 MB L2 (a block's working set is 12.6 MB per XCD).  profiles/r01_*: conv time = (time of the
// memory side alone) + (MFMA time), i.e. the launch is paced by the bytes it moves.  Here the roles are
// swapped: the OUTPUTS stay put.  A workgroup (4 waves, ONE per SIMD, 512 registers each) owns a
// 16x32-pixel tile and keeps the fp32 accumulators of ALL 192 output channels of the block in registers:
// a wave owns 4 rows x 32 pixels x 6 cout blocks = 24 MFMA accumulators = 384 registers.  hipcc only ever
// emits the AGPR form of v_mfma (256 accumulator registers at most, anything beyond is shuttled through
// v_accvgpr copies), so the MFMAs are issued through inline asm with explicit register classes: conv3,
// conv4, conv5 accumulate in the 256 AGPRs ("+a"), conv1 / conv2 (the first to retire) in 128 VGPRs
// ("+v"), which leaves 128 VGPRs for fragments and addresses.  The block then runs as five PHASES, one
// per newly available input slice:
//     phase 1  stage x   (64 ch)  -> accumulate into conv1..conv5        (6 cout blocks)
//     phase 2  stage x1  (32 ch)  -> conv2..conv5                        (5)
//     phase 3  stage x2           -> conv3..conv5                        (4)
//     phase 4  stage x3           -> conv4, conv5                        (3)
//     phase 5  stage x4           -> conv5                               (2)
// so every input channel is staged ONCE (192 channel-reads per pixel instead of 640) and each staged
// B fragment feeds 3 kh x NB MFMAs instead of 3.  After phase p the finished conv_p leaves through the
// usual fused epilogue (bias, LeakyReLU, + conv1x1(x) for x2, + x2 for x4, *0.2 + x, noise, RRDB tail).
// The bias-free 1x1 (block.py:263) is computed between phases 1 and 2 from the tile's own x pixels
// (no halo, no neighbour needed) into the registers conv1 just vacated — it fills the wait for the
// neighbours' x1.
//
// The 3x3 taps of phase p+1 need x_p on a 1-pixel halo, i.e. from the 8 neighbouring tiles.  All tiles
// of an image are co-resident (one workgroup per CU, <= 256 tiles per image) and run in lock step; each
// publishes "phase e done" through a per-tile flag and polls its neighbours' flags before staging the
// next slice.  The hand-off is placement independent (agent scope; cdna guide, Guideline 16 R1): payload
// = write-through (sc1) 16-byte stores, every storing wave drains vmcnt, one lane stores the flag (relaxed,
// agent scope); the consumer polls relaxed and then reads the payload with sc1 loads (LDS-DMA, L1
// bypassed).  Spins are bounded; a time-out raises the abort word of the workspace and every workgroup
// leaves.  Overwrite hazards: a slot written in phase e is read by the neighbours in phase e+1 only, and
// is next overwritten in phase e+5, which the owner cannot reach before every neighbour published e+1.
//
// GEMM view per unit (K step c, column tap kw):  D[cout][pixel] += W[cout][(kh, cin16)] X[(kh, cin16)][pixel]
//   A fragments: NB x 3 (kh) x 1 KB per unit, streamed (LDS-DMA, L2 resident) through a 4-slot ring;
//   B fragments: one 18x34 halo tile of a 32-byte channel group per K step — fp32: 3-slot ring (2 steps ahead);
//   fp16: resident in the LDS (the epilogues write the tile's own pixels, only the halo ring is fetched).
// The fp16 path runs every phase as crit_p (conv_p alone) -> epilogue -> bulk_p (the remaining convs, with the
// halo hand-off hidden under them): see `Sched` below.  The accumulation order per output element is (chunk, kw,
// kh) as in conv_mfma.hip; the fp16 path folds the block residual (conv5's accumulators start at 5 x) and keeps
// x1..x4 as fp16 in the LDS.  tests/test_gpu_rdb_chain.py holds it to 1e-5 (fp32) / 2e-3 (fp16) of the per-conv
// launches and requires bit-equal results run to run.
#pragma once
#include <cstdlib>
#include <mutex>

#include "mfma_tile.h"

#ifndef ESR_ABL
#define ESR_ABL 0   // timing ablations (development only; results are wrong when non-zero)
#endif

namespace {

// ESR_R: output rows per wave.  4 = the 16x32 tile every launch with enough tiles uses; the 2- and 1-row builds
// (8x32 / 4x32 tiles, rdb_fused_rows.hip) are for launches whose 16-row tiles would leave most CUs idle — the
// reference's training crops (batch 16 of 32x32 LR = 32 tiles of 16x32 on 256 CUs, SRRaGAN train configs).
#ifndef ESR_R
#define ESR_R 4
#endif
constexpr int R = ESR_R;                    // output rows per wave
static_assert(R == 1 || R == 2 || R == 4, "rows per wave");
constexpr int TH = 4 * R, TW = 32;          // tile of a workgroup (4 waves stacked vertically)
constexpr int IH = TH + 2, IW = TW + 2;     // staged halo tile
constexpr int NT = 256;                     // 4 waves, one per SIMD
constexpr int NSLOT = IH * IW * 2;          // 16-byte slots of one activation stage
constexpr int NLD = (NSLOT + NT - 1) / NT;  // DMA rounds per stage (5)
constexpr int ASLOT = NLD * NT * 16;        // 20480
constexpr int AR = 4;                       // activation slots: all stages of a 64-channel fp16 input are resident
// ESR_FAT: the unit runner of the small-tile builds (run_units_fat below).  With one row per wave a unit is 6..15
// MFMAs (0.08..0.2 us): a fragment read covers ONE MFMA and the unit boundary (barrier, DMA set-up) comes every few
// hundred cycles, so the fragments of unit i+1 are read while unit i's MFMAs issue, and the weight ring is deeper
// (the 4-row tile leaves the LDS room: 8 slots of 15 fragments, requests 4 units ahead).
#ifndef ESR_FAT
#define ESR_FAT (ESR_R == 1)
#endif
constexpr bool FAT = ESR_FAT != 0;
constexpr int WSLOT = FAT ? 15 * 1024 : 18 * 1024;   // weight unit (fp16 schedule: <= 15 fragments; fp32 phases: 6 blocks x 3 kh)
constexpr int WR = FAT ? 8 : 4;             // weight ring depth
#ifndef ESR_AH
#define ESR_AH (ESR_FAT ? 4 : 3)     // (block at 16 x 32^2 with a lead of 3 / 4 / 5 / 6 units: 22.6 / 20.8 / 21.1 / 21.7 us)
#endif
constexpr int AH = ESR_AH;                  // a unit's weights are requested AH units ahead (<= WR - 1; FAT: <= WR - 2)
static_assert(AH >= 3 && AH <= (FAT ? WR - 2 : WR - 1), "weight ring lead");
constexpr int WOFF = AR * ASLOT;
constexpr int LDS_CTRL = WOFF + WR * WSLOT;   // two control words behind the rings
constexpr int LDS_BIAS = LDS_CTRL + 64;        // fp16 path: the block's 192 biases (fp32)
constexpr int LDS_FLAGS = LDS_BIAS + 192 * 4;  // fp16 path: the neighbours' flags as wave 0 last fetched them (64 words)
constexpr int LDS_HALO = LDS_FLAGS + 256;      // fp16 path: per thread {source, destination} offset of its halo slot
constexpr int LDS_NBR = LDS_HALO + NT * 8;     // fp16 path: per lane of wave 0, the byte offset (in the workspace) of the flag it fetches
constexpr int LDS_BYTES = LDS_NBR + 256;       // 159040
constexpr int MASK_SLICE = 512 * R;            // bytes of one slice's masks of a tile: [wave][lane][R rows x u16]
constexpr int MASK_TILE = 4 * MASK_SLICE;
constexpr int LDS_MASK = LDS_BYTES;            // backward: two buffers of LeakyReLU masks (one dense slice each)
constexpr int LDS_BYTES_BWD = LDS_MASK + 2 * MASK_SLICE;
static_assert(LDS_BYTES_BWD <= 160 * 1024, "LDS budget");
constexpr int NHALO = 2 * 2 * IW + 2 * 2 * TH;  // 16-byte slots of the 1-pixel halo ring of one stage (200)
constexpr int WS_HDR = 16;                  // workspace words before the per-tile flags
enum { WS_TICKET = 0, WS_ABORT = 1 };

// ---- 24 named accumulators per wave: [cout block 0..5][row 0..3] --------------------------------------
// cout blocks of a dense block: 0..3 = conv1..conv4, 4/5 = conv5[0:32]/[32:64].
struct Acc24 {
  f32x16 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, v16, v17, v18, v19, v20, v21, v22, v23;
};
template <int I> __device__ __forceinline__ f32x16& acc_at(Acc24& s) {
  static_assert(I >= 0 && I < 24, "acc index");
#define ESR_ACC_CASE(n) if constexpr (I == n) return s.v##n; else
  ESR_ACC_CASE(0) ESR_ACC_CASE(1) ESR_ACC_CASE(2) ESR_ACC_CASE(3) ESR_ACC_CASE(4) ESR_ACC_CASE(5)
  ESR_ACC_CASE(6) ESR_ACC_CASE(7) ESR_ACC_CASE(8) ESR_ACC_CASE(9) ESR_ACC_CASE(10) ESR_ACC_CASE(11)
  ESR_ACC_CASE(12) ESR_ACC_CASE(13) ESR_ACC_CASE(14) ESR_ACC_CASE(15) ESR_ACC_CASE(16) ESR_ACC_CASE(17)
  ESR_ACC_CASE(18) ESR_ACC_CASE(19) ESR_ACC_CASE(20) ESR_ACC_CASE(21) ESR_ACC_CASE(22)
  return s.v23;
#undef ESR_ACC_CASE
}
template <int BLK, int ROW> __device__ __forceinline__ f32x16& acc_br(Acc24& s) { return acc_at<BLK * R + ROW>(s); }

// MFMA through inline asm with an explicit accumulator register class: AGPR (blocks 2..5) or VGPR
// (blocks 0,1).  hipcc does not pad hazards of asm statements (cdna guide 5.7): accumulate chains
// (same D as C) need none; before anything else reads or overwrites an accumulator the callers run
// mfma_drain().  A/B come from ds_read (s_waitcnt is placed by the compiler through the operands).
constexpr bool acc_in_agpr(int blk) { return blk >= 2; }
// FIRST: the accumulator's first MFMA of a block takes the inline constant 0 as SrcC and a write-only
// output, so accumulators are never zeroed by VALU code (hipcc materialises such zeros lazily with
// v_mov copies BETWEEN the asm MFMAs, where nothing pads the VALU-write -> MFMA-read hazard).
template <typename T, bool AGPR, bool FIRST = false>
__device__ __forceinline__ void mma_cls(f32x16& acc, const u32x4& a, const u32x4& b) {
  if constexpr (sizeof(T) == 2) {
    if constexpr (FIRST) {
      if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
    } else {
      if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
  } else {
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    if constexpr (FIRST) {
      if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(fa[0]), "v"(fb[0]));
      else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(fa[0]), "v"(fb[0]));
    }
#pragma unroll
    for (int t = FIRST ? 1 : 0; t < 4; ++t) {
      if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(fa[t]), "v"(fb[t]));
      else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fa[t]), "v"(fb[t]));
    }
  }
}
// fp16 MFMA with the counted LDS wait for its fragments INSIDE the statement (WAITC < 0: none).  hipcc pads every
// asm statement that follows another one defining VGPRs (a wait or tie statement, a fragment read) with `s_nop 0`; one
// wave per SIMD pays an issue slot for each — 370 per block of the 4-row build before this.
template <bool AGPR, bool FIRST, int WAITC>
__device__ __forceinline__ void mma_w(f32x16& acc, const u32x4& a, const u32x4& b) {
  if constexpr (WAITC >= 0) {
    if constexpr (FIRST) {
      if constexpr (AGPR) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b), "n"(WAITC));
      else asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b), "n"(WAITC));
    } else {
      if constexpr (AGPR) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b), "n"(WAITC));
      else asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b), "n"(WAITC));
    }
  } else {
    mma_cls<_Float16, AGPR, FIRST>(acc, a, b);
  }
}
// >= 18 wait states: covers "XDL write VGPR -> VALU / VMEM read or write" for 8- and 16-pass MFMAs
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// The carried accumulators of cout blocks 4 / 5 (written by VALU code: the block tail, the tile set-up) are pinned
// into their AGPRs HERE, with the VALU-write -> MFMA-SrcC wait states inside the statement: left alone, hipcc keeps
// them in VGPRs and copies them over (v_accvgpr_write) lazily BETWEEN the asm MFMAs of the next block's first bulk
// unit, where nothing pads that hazard (cdna guide 5.7 item 2) — seen as 4 stale accumulator registers of one row.
#if ESR_R == 4
#define ESR_ACC45_OPS(a) "+a"(a.v16), "+a"(a.v17), "+a"(a.v18), "+a"(a.v19), "+a"(a.v20), "+a"(a.v21), "+a"(a.v22), "+a"(a.v23)
#elif ESR_R == 2
#define ESR_ACC45_OPS(a) "+a"(a.v8), "+a"(a.v9), "+a"(a.v10), "+a"(a.v11)
#else
#define ESR_ACC45_OPS(a) "+a"(a.v4), "+a"(a.v5)
#endif
__device__ __forceinline__ void pin_acc45(Acc24& acc) { asm volatile("s_nop 7" : ESR_ACC45_OPS(acc)); }
// Fences of an MFMA segment.  With all 256 AGPRs holding accumulators hipcc
// sometimes parks one accumulator tuple in VGPRs around the boundary code (v_accvgpr_read right behind a segment's
// last MFMA, v_accvgpr_write in front of the next segment's first use).  It cannot know that the asm statements are
// MFMAs, so it pads neither "XDL write -> v_accvgpr_read" (18 wait states) nor "VALU write -> MFMA SrcC" — seen as a
// few accumulator registers that miss the last MFMA's contribution.  These statements take every AGPR accumulator
// as an operand: the copies can only sit outside [seg_open, seg_close], and the wait states are inside the strings.
#if ESR_R == 4
#define ESR_ACC_B4(a) "+a"(a.v16), "+a"(a.v17), "+a"(a.v18), "+a"(a.v19), "+a"(a.v20), "+a"(a.v21), "+a"(a.v22), "+a"(a.v23)
#define ESR_ACC_B3(a) "+a"(a.v12), "+a"(a.v13), "+a"(a.v14), "+a"(a.v15), ESR_ACC_B4(a)
#define ESR_ACC_AGPR_OPS(a) "+a"(a.v8), "+a"(a.v9), "+a"(a.v10), "+a"(a.v11), ESR_ACC_B3(a)
#elif ESR_R == 2
#define ESR_ACC_B4(a) "+a"(a.v8), "+a"(a.v9), "+a"(a.v10), "+a"(a.v11)
#define ESR_ACC_B3(a) "+a"(a.v6), "+a"(a.v7), ESR_ACC_B4(a)
#define ESR_ACC_AGPR_OPS(a) "+a"(a.v4), "+a"(a.v5), ESR_ACC_B3(a)
#else
#define ESR_ACC_B4(a) "+a"(a.v4), "+a"(a.v5)
#define ESR_ACC_B3(a) "+a"(a.v3), ESR_ACC_B4(a)
#define ESR_ACC_AGPR_OPS(a) "+a"(a.v2), ESR_ACC_B3(a)
#endif
// register lists of the B-row fragments (R + 2 staged rows feed R output rows) for the hand-placed LDS waits
#if ESR_R == 4
#define ESR_BF_ALL(b) "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5])
#define ESR_BF_HI(b) "+v"(b[3]), "+v"(b[4]), "+v"(b[5])
#define ESR_ROWS(b) "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
#elif ESR_R == 2
#define ESR_BF_ALL(b) "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
#define ESR_BF_HI(b) "+v"(b[3])
#define ESR_ROWS(b) "+v"(b[0]), "+v"(b[1])
#else
#define ESR_BF_ALL(b) "+v"(b[0]), "+v"(b[1]), "+v"(b[2])
#define ESR_ROWS(b) "+v"(b[0])
#endif
#ifndef ESR_FENCE_OPEN_NOP
#define ESR_FENCE_OPEN_NOP "s_nop 4"
#endif
#ifndef ESR_FENCE_CLOSE_NOP
#define ESR_FENCE_CLOSE_NOP "s_nop 15\n\ts_nop 15"
#endif
// B0: the first cout block that is still live (<= 2: all AGPR accumulators).  A fence that names a finished block's
// registers keeps them occupied, and what is fetched ahead of conv5 for the block tail (the RRDB residual rows: 64
// registers) then has nowhere to live but scratch.
template <int B0 = 2> __device__ __forceinline__ void seg_open(Acc24& acc) {
  if constexpr (B0 <= 2) asm volatile(ESR_FENCE_OPEN_NOP : ESR_ACC_AGPR_OPS(acc));
  else if constexpr (B0 == 3) asm volatile(ESR_FENCE_OPEN_NOP : ESR_ACC_B3(acc));
  else asm volatile(ESR_FENCE_OPEN_NOP : ESR_ACC_B4(acc));
}
template <int B0 = 2> __device__ __forceinline__ void seg_close(Acc24& acc) {
  if constexpr (B0 <= 2) asm volatile(ESR_FENCE_CLOSE_NOP : ESR_ACC_AGPR_OPS(acc));
  else if constexpr (B0 == 3) asm volatile(ESR_FENCE_CLOSE_NOP : ESR_ACC_B3(acc));
  else asm volatile(ESR_FENCE_CLOSE_NOP : ESR_ACC_B4(acc));
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_dyn(int n) {   // n is wave-uniform; conservative above 47
  if (n >= 16) {
    if (n >= 32) { if (n >= 40) wait_vm<40>(); else wait_vm<32>(); }
    else { if (n >= 24) wait_vm<24>(); else wait_vm<16>(); }
    return;
  }
  switch (n) {
    case 0: wait_vm<0>(); break;   case 1: wait_vm<1>(); break;   case 2: wait_vm<2>(); break;
    case 3: wait_vm<3>(); break;   case 4: wait_vm<4>(); break;   case 5: wait_vm<5>(); break;
    case 6: wait_vm<6>(); break;   case 7: wait_vm<7>(); break;   case 8: wait_vm<8>(); break;
    case 9: wait_vm<9>(); break;   case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
    case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break;
    default: wait_vm<15>(); break;
  }
}

// agent-scope (L1-bypassing) LDS-DMA: the activations another workgroup just published
__device__ __forceinline__ void dma16_sc1(const char* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 16);
}

// One G32 view of ONE image as a buffer resource (wave-uniform): 16-byte sc1 loads / stores.
struct ImgView {
  __amdgpu_buffer_rsrc_t r;
  int gs;        // group stride (bytes)
  int ng;        // groups addressable
};
// hipcc wraps every buffer access whose descriptor it cannot PROVE wave-uniform in a waterfall loop
// (readfirstlane x4 + compare + saveexec, cdna guide T20) — and anything that went through the LDS or a
// block-table load counts as divergent.  Descriptor inputs therefore pass through readfirstlane once.
__device__ __forceinline__ char* uniform_ptr(const void* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (char*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ ImgView img_view(const esr_g32& v, int b, int g0 = 0) {
  ImgView o;
  char* base = uniform_ptr((char*)v.ptr + (int64_t)b * v.batch_stride + (int64_t)g0 * v.group_stride);
  o.r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  o.gs = __builtin_amdgcn_readfirstlane((int)v.group_stride);
  o.ng = v.ngroups - g0;
  return o;
}

// 16 consecutive channels (cout block cb, lane half h) of one pixel <-> float[16], through sc1 accesses
template <typename T> struct Ch16;
template <> struct Ch16<_Float16> {
  static __device__ __forceinline__ void store(const ImgView& t, int cb, int h, int pixoff, const float v[16]) {
    half8 x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)v[i]; y[i] = (_Float16)v[8 + i]; }
    const int off = (2 * cb + h) * t.gs + pixoff;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), t.r, off, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), t.r, off + 16, 0, 16);
  }
  // packed row: the lane's 16 channels as stored (2 x 16 bytes)
  static __device__ __forceinline__ void pack(const float v[16], u32x4 (&q)[2]) {
    half8 x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)v[i]; y[i] = (_Float16)v[8 + i]; }
    q[0] = __builtin_bit_cast(u32x4, x); q[1] = __builtin_bit_cast(u32x4, y);
  }
  static __device__ __forceinline__ void store_packed(const ImgView& t, int cb, int h, int pixoff, const u32x4 (&q)[2]) {
    const int off = (2 * cb + h) * t.gs + pixoff;
    __builtin_amdgcn_raw_buffer_store_b128(q[0], t.r, off, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(q[1], t.r, off + 16, 0, 16);
  }
  // The same 64 bytes per pixel pair of lanes, ROW-CONTIGUOUS: lane (pixel j, group h) holds the 32 bytes of its pixel
  // in group h, so store_packed's two instructions each fill half of every 32-byte sector (16 bytes per lane, 32 apart)
  // — and these are write-through stores: 13.5 GB reached memory per training-forward launch for 6.9 GB of slices
  // (PMC WRITE_SIZE, profiles/r04_fwdbwd_pmc_summary.txt).  One v_permlane32_swap per dword exchanges the upper
  // half-wave's first 16 bytes with the lower half-wave's second 16: afterwards lanes 0..31 / 32..63 hold bytes 0..15 /
  // 16..31 of the 32 pixels of ONE group, and each instruction stores a whole 1 KB group row.  pixoff: the offset of
  // the lane's own pixel (both lanes of a pair: the same pixel j), or the out-of-range offset for a dropped pixel.
  static __device__ __forceinline__ void store_packed_rows(const ImgView& t, int cb, int h, int pixoff, const u32x4 (&q)[2]) {
    u32x4 a = q[0], b = q[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const auto r = __builtin_amdgcn_permlane32_swap(a[i], b[i], false, false);
      a[i] = r[0]; b[i] = r[1];
    }
    const int off = pixoff + h * 16;           // (an out-of-range pixoff stays out of range: + 16 at most)
    __builtin_amdgcn_raw_buffer_store_b128(a, t.r, (2 * cb) * t.gs + off, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(b, t.r, (2 * cb + 1) * t.gs + off, 0, 16);
  }
  struct Raw { u32x4 q[2]; };
  static __device__ __forceinline__ void load(const ImgView& t, int cb, int h, int pixoff, Raw& r) {
    const int off = (2 * cb + h) * t.gs + pixoff;
    r.q[0] = __builtin_amdgcn_raw_buffer_load_b128(t.r, off, 0, 16);
    r.q[1] = __builtin_amdgcn_raw_buffer_load_b128(t.r, off + 16, 0, 16);
  }
  static __device__ __forceinline__ void get(const Raw& r, float v[16]) {
    const half8 x = __builtin_bit_cast(half8, r.q[0]), y = __builtin_bit_cast(half8, r.q[1]);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (float)x[i]; v[8 + i] = (float)y[i]; }
  }
};
template <> struct Ch16<float> {
  static __device__ __forceinline__ void store(const ImgView& t, int cb, int h, int pixoff, const float v[16]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int off = (4 * cb + 2 * h + g) * t.gs + pixoff;
      f32x4 a, c;
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = v[8 * g + i]; c[i] = v[8 * g + 4 + i]; }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), t.r, off, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, c), t.r, off + 16, 0, 16);
    }
  }
  struct Raw { u32x4 q[4]; };
  static __device__ __forceinline__ void load(const ImgView& t, int cb, int h, int pixoff, Raw& r) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int off = (4 * cb + 2 * h + g) * t.gs + pixoff;
      r.q[2 * g] = __builtin_amdgcn_raw_buffer_load_b128(t.r, off, 0, 16);
      r.q[2 * g + 1] = __builtin_amdgcn_raw_buffer_load_b128(t.r, off + 16, 0, 16);
    }
  }
  static __device__ __forceinline__ void get(const Raw& r, float v[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 a = __builtin_bit_cast(f32x4, r.q[g]);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[4 * g + i] = a[i];
    }
  }
};

template <typename T> struct Cfg {
  static constexpr int CPG = DT<T>::CPG, GPB = DT<T>::GPB;
  static constexpr int KX = 64 / CPG;      // K steps of the 64-channel block input
  static constexpr int KD = 32 / CPG;      // K steps of one 32-channel dense slice
  static constexpr int ksteps(int p) { return p == 1 ? KX : KD; }
  static constexpr int nblk(int p) { return 7 - p; }
  // byte offset of phase p (1..5) in the block's fused weight stream; phase_off(6) = the 1x1 fragments
  static constexpr int phase_off(int p) {
    int o = 0;
    for (int q = 1; q < p; ++q) o += ksteps(q) * 3 * nblk(q) * 3 * 1024;
    return o;
  }
  static constexpr int STREAM_BYTES = phase_off(6) + KX * 1024;
};

// The tile in flight.  Only wave-uniform values live here (SGPRs).  Everything per lane is RE-DERIVED from the
// lane id where it is used: a per-lane constant computed once per tile is a VGPR that lives across the whole
// block loop, i.e. across the MFMA segments where all registers are taken — hipcc spills it and reloads it at
// every use, and with weight DMAs in flight each scratch reload is a full `vmcnt(0)` drain.  The lane id itself
// costs nothing to keep: v_mbcnt derives it from EXEC (volatile asm, so that the values derived from it are
// not hoisted back out of the loops into long-lived registers).
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
struct Tile {
  int b, oy0, ox0;
  int ty, tx, tiles_y, tiles_x;
  int wave;
  int wp;            // row pitch (pixels) of every view
  int hlim, wlim;    // H + 1 / W + 1: the zero halo row / column behind the image — staging reads of a tile that sticks
                     // out of the image are clamped to them (beyond lies another plane, or the end of the allocation)
  __device__ __forceinline__ int lane() const { return fresh_lane(); }
  __device__ __forceinline__ int tid() const { return wave * 64 + fresh_lane(); }
  // per-lane B-fragment offset of column tap kw (pixel j + kw, half h; halves swapped by (col >> 3) & 1)
  static __device__ __forceinline__ int colofs(int lane, int kw) {
    const int col = (lane & 31) + kw;
    return col * 32 + (((lane >> 5) ^ ((col >> 3) & 1)) << 4);
  }
  // this lane's own pixel (row 4*wave, column j) inside an activation slot, half 0; swz = 16 if the two halves
  // of that pixel are swapped in the LDS image
  __device__ __forceinline__ void own(int lane, int& px, int& swz) const {
    const int j = lane & 31;
    px = ((wave * R + 1) * IW + j + 1) * 32;
    swz = (((j + 1) >> 3) & 1) << 4;
  }
  // LDS-resident path (fp16): the 1-pixel halo ring of a stage = 200 16-byte slots, one per thread: rows 0 / 17
  // (34 pixels each), then columns 0 / 33 of rows 1..16.  src: byte offset inside a group plane (-1: this
  // thread has no slot); dst: byte offset inside an activation slot
  __device__ __forceinline__ void halo(int& src, int& dst) const {
    const int i = tid();
    int row, col, hs;
    if (i < 4 * IW) { const int s = i % (2 * IW); row = i < 2 * IW ? 0 : IH - 1; col = s >> 1; hs = s & 1; }
    else { const int s = (i - 4 * IW) % (2 * TH); row = 1 + (s >> 1); col = i < 4 * IW + 2 * TH ? 0 : IW - 1; hs = s & 1; }
    const int sy = oy0 + row < hlim ? oy0 + row : hlim, sx = ox0 + col < wlim ? ox0 + col : wlim;
    src = i < NHALO ? (sy * wp + sx) * 32 + ((hs ^ ((col >> 3) & 1)) << 4) : -1;
    dst = (row * IW + col) * 32 + hs * 16;
  }
};

// ---- weights of unit u of a phase -> ring slot (u & 3) ---------------------------------------------
template <int NF> __device__ __forceinline__ void issue_w(const char* wsrc, int u, char* smem, const Tile& t) {
  const char* src = wsrc + ((int64_t)u * NF) * 1024 + t.lane() * 16;
  char* dst = smem + WOFF + (u & (WR - 1)) * WSLOT;
#pragma unroll
  for (int i = 0; i < (NF + 3) / 4; ++i) {
    const int q = t.wave + 4 * i;
    if (q < NF) dma16(src + q * 1024, dst + q * 1024);
  }
}
// ---- activation stage (one 32-byte channel group, 18x34 halo tile) -> ring slot sa -----------------
__device__ __forceinline__ void issue_a(const char* plane, int sa, char* smem, const Tile& t) {
  // LDS slot s = tid + 256*i  <-  (row, col, 16-byte half), halves swapped by (col>>3)&1 on the SOURCE address
  // so that ds_read_b128 B-fragment reads are bank-conflict free.  Offsets are recomputed per call: this
  // path only stages a chain's first input (and the fp32 reference path), and 5 live registers cost more.
  char* dst = smem + sa * ASLOT + t.wave * 1024;
  const int tid = t.tid();
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int s = tid + NT * i;
    if (s >= NSLOT) s = NSLOT - 1;            // tail lanes: harmless re-copy into the padding
    const int row = s / (2 * IW), r2 = s - row * 2 * IW;
    const int col = r2 >> 1, hs = r2 & 1, half = hs ^ ((col >> 3) & 1);
    const int sy = t.oy0 + row < t.hlim ? t.oy0 + row : t.hlim, sx = t.ox0 + col < t.wlim ? t.ox0 + col : t.wlim;
    dma16_sc1(plane + (sy * t.wp + sx) * 32 + half * 16, dst + NT * 16 * i);
  }
}

// ---- LDS-resident activations (fp16 path) -----------------------------------------------------------
// A tile's own 16x32 pixels of a slice never come back from memory: the epilogue that produces them also
// writes them into the stage slots the next phase reads (slot = channel group of the slice).  Only the
// 1-pixel ring around the tile is fetched (sc1 loads, after the neighbours published): one 16-byte load
// + one ds_write per thread and stage.
// one LDS word through inline asm: a compiler-visible access of the generic `smem` pointer becomes a FLAT load that is
// waited for with vmcnt(0) — a full drain of the weight DMAs in flight, at every hand-off
__device__ __forceinline__ int lds_word(const char* smem, int off) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)off) : "memory");
  return v;
}
template <int K> struct HaloRegs { u32x4 q[K]; };
template <int K> __device__ __forceinline__ void halo_issue(const ImgView& v, int g0, int hsrc, HaloRegs<K>& h) {
  // threads without a slot read (and later drop) the view's first bytes: a branch around the loads would
  // also fence them off from the MFMAs they are meant to hide under
#pragma unroll
  for (int c = 0; c < K; ++c) h.q[c] = __builtin_amdgcn_raw_buffer_load_b128(v.r, hsrc >= 0 ? (g0 + c) * v.gs + hsrc : 0, 0, 16);
}
template <int K> __device__ __forceinline__ void halo_put(char* smem, int slot0, int hsrc, int hdst, const HaloRegs<K>& h) {
  if (hsrc >= 0) {
#pragma unroll
    for (int c = 0; c < K; ++c) *(u32x4*)(smem + (slot0 + c) * ASLOT + hdst) = h.q[c];
  }
}
template <int K> __device__ __forceinline__ void halo_fetch(const ImgView& v, int g0, char* smem, const Tile& t, int slot0 = 0) {
  HaloRegs<K> h;
  // the thread's slot as the tile set-up cached it (LDS_HALO)
  const int hsrc = lds_word(smem, LDS_HALO + t.tid() * 8), hdst = lds_word(smem, LDS_HALO + t.tid() * 8 + 4);
  halo_issue<K>(v, g0, hsrc, h);
  halo_put<K>(smem, slot0, hsrc, hdst, h);
}
// the lane's packed 16 channels of row r (own pixel) -> stage slot `slot`
__device__ __forceinline__ void lds_put_row(char* smem, int slot, int r, const u32x4 (&q)[2], int own_px, int own_swz) {
  char* px = smem + slot * ASLOT + own_px + r * (IW * 32);
  *(u32x4*)(px + own_swz) = q[0];
  *(u32x4*)(px + (own_swz ^ 16)) = q[1];
}

// .. and back (the lane's own pixel as the epilogue stored it)
template <typename RAW> __device__ __forceinline__ void lds_get_rows(const char* smem, int slot0, RAW (&q)[R], const Tile& t) {
  const int lane = t.lane();
  int own_px, own_swz;
  t.own(lane, own_px, own_swz);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const char* px = smem + (slot0 + (lane >> 5)) * ASLOT + own_px + r * (IW * 32);
    q[r].q[0] = *(const u32x4*)(px + own_swz);
    q[r].q[1] = *(const u32x4*)(px + (own_swz ^ 16));
  }
}

// ---- the MFMAs of one unit of phase P: cout blocks P-1..5 x 3 kh taps x 4 rows ----------------------
// Every LDS fragment read is inline asm as well: with a compiler-tracked ds_read outstanding hipcc puts
// `s_waitcnt lgkmcnt(0)` in front of every asm MFMA.  A fragments are double buffered; block bi+1's
// three fragments are requested after the 4th of block bi's 12 MFMAs (the buffer's previous readers,
// block bi-1, are >= 4 MFMAs back; 8 MFMAs = 256 cycles cover the LDS latency) and waited for with
// lgkmcnt(0) in front of block bi+1.
template <int OFF> __device__ __forceinline__ void lds_read16(u32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_wait3(u32x4& a, u32x4& b, u32x4& c) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
}
struct NoIssue { __device__ __forceinline__ void operator()(int) const {} };
// `issue(i)`, i = 0..4: the wave's i-th weight DMA of the unit it prefetches; called BETWEEN the cout
// blocks' MFMA groups so that the DMA issue time hides under the matrix pipe (one wave per SIMD: any
// instruction that is not in an MFMA's shadow is lost time).
template <typename T, int P, bool FIRST = false, bool CARRY45 = false, typename ISSUE = NoIssue>
__device__ __forceinline__ void unit_mma(Acc24& acc, const uint32_t lds_b, const uint32_t lds_w, ISSUE&& issue = NoIssue{}) {
  constexpr int NB = 7 - P;
  u32x4 bf[R + 2], af[2][3];
  sfor<R + 2>([&](auto IR) __attribute__((always_inline)) { lds_read16<decltype(IR)::value * IW * 32>(bf[decltype(IR)::value], lds_b); });
  sfor<3>([&](auto KH) __attribute__((always_inline)) { lds_read16<decltype(KH)::value * 1024>(af[0][decltype(KH)::value], lds_w); });
  asm volatile("s_waitcnt lgkmcnt(0)" : ESR_BF_ALL(bf), "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]));
  sfor<NB>([&](auto BI) __attribute__((always_inline)) {
    constexpr int bi = decltype(BI)::value;       // position in the unit's fragment list
    constexpr int blk = P - 1 + bi;
    constexpr int cur = bi & 1, nxt = cur ^ 1;
    // the 12 (input row, kh) pairs in issue order: ir-major, rows r = ir - kh
    sfor<R + 2>([&](auto IR) __attribute__((always_inline)) {
      constexpr int ir = decltype(IR)::value;
      sfor<3>([&](auto KH) __attribute__((always_inline)) {
        constexpr int kh = decltype(KH)::value;
        constexpr int r = ir - kh;
        if constexpr (r >= 0 && r < R) mma_cls<T, acc_in_agpr(blk), FIRST && kh == 0 && !(CARRY45 && blk >= 4)>(acc_br<blk, r>(acc), af[cur][kh], bf[ir]);
      });
      if constexpr (ir == 2 && bi + 1 < NB) {     // after MFMA 6 of 12
        sfor<3>([&](auto KH) __attribute__((always_inline)) {
          lds_read16<((bi + 1) * 3 + decltype(KH)::value) * 1024>(af[nxt][decltype(KH)::value], lds_w);
        });
      }
    });
    if constexpr (bi + 1 < NB) lds_wait3(af[nxt][0], af[nxt][1], af[nxt][2]);
    __builtin_amdgcn_sched_barrier(0);
    issue(bi);
    if constexpr (bi + 1 == NB) {
#pragma unroll
      for (int i = NB; i < 5; ++i) issue(i);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// ---- resident path (fp16): the unit schedule of a dense block ----------------------------------------
// The five convs of a block form a dependent chain: conv_p -> epilogue (x_p) -> halo exchange with the 8
// neighbouring tiles -> conv_{p+1}.  Of a phase's MFMAs (stage x_{p-1} into conv_p..conv5) only conv_p's
// are on that chain, so every phase is split:
//     crit_p   stage x_{p-1} -> conv_p only                 (cout block p-1; conv5: blocks 4, 5)
//     epilogue x_p: border pixels to memory, own pixels to the LDS
//     bulk_p   stage x_{p-1} -> conv_{p+1}..conv5           (blocks p..5) — nothing waits for these, so the
//              hand-off hides under them: the stores drain while the first bulk units run, the flag goes out
//              at the second unit's barrier, and the neighbours' flags are long up when the bulk ends
//     poll, fetch x_p's halo ring, crit_{p+1} ...
// Units (one weight-ring slot each, in stream order; K = 4 K-steps for x, 2 for x1..x4):
//     crit_p (p<5): K units (c)      = 3 kw x 1 block x 3 kh  =  9 fragments, 36 MFMAs
//     bulk_p (p<5): 3K units (c, kw) = (5-p)+1 blocks x 3 kh  = 15/12/9/6 fragments
//     1x1         : 1 unit after bulk_1 (K fragments)
//     crit_5      : 3K units (c, kw) = 2 blocks x 3 kh        =  6 fragments
// Stage slots: x -> 0..3; x1 -> 0,1 (after the 1x1 has read x); x2 -> 2,3; x3 -> 0,1; x4 -> 2,3; the block
// output -> 0..3.  A stage is overwritten only after the bulk that read its predecessor in those slots.
enum { U_CRIT = 0, U_BULK = 1, U_ONE = 2 };
struct UDesc { int kind, P, c, kw, nf, off; };        // off: fragments from the start of the block's stream
// BW: the schedule of the BACKWARD chain (esr_rdb_backward): the same five phases over the gradient slices; the
// 1x1 unit (here the transposed 1x1: g_x += W1x1^T g_x2, 2 cout blocks x KD K steps) sits behind crit_3 — g_x2 is
// what crit_3 finishes — instead of behind bulk_1.
template <typename T, bool BW = false> struct Sched {
  using CF = Cfg<T>;
  using Elem = T;
  static constexpr int ONE_P = BW ? 3 : 1;               // phase that carries the 1x1 unit
  static constexpr int ONE_NF = BW ? 2 * CF::KD : CF::KX;
  // idx < 0: {.., nf = number of units, off = fragments of the whole stream}
  static constexpr UDesc at(int idx) {
    int i = 0, off = 0;
    for (int P = 1; P <= 5; ++P) {
      const int K = CF::ksteps(P);
      if (P < 5) {
        for (int c = 0; c < K; ++c) { if (i == idx) return {U_CRIT, P, c, 0, 9, off}; ++i; off += 9; }
        if (BW && P == 3) { if (i == idx) return {U_ONE, 3, 0, 0, ONE_NF, off}; ++i; off += ONE_NF; }
        const int nf = (6 - P) * 3;
        for (int c = 0; c < K; ++c)
          for (int kw = 0; kw < 3; ++kw) { if (i == idx) return {U_BULK, P, c, kw, nf, off}; ++i; off += nf; }
        if (!BW && P == 1) { if (i == idx) return {U_ONE, 1, 0, 0, ONE_NF, off}; ++i; off += ONE_NF; }
      } else {
        for (int c = 0; c < K; ++c)
          for (int kw = 0; kw < 3; ++kw) { if (i == idx) return {U_CRIT, 5, c, kw, 6, off}; ++i; off += 6; }
      }
    }
    return {-1, 0, 0, 0, i, off};
  }
  static constexpr int N = at(-1).nf;
  static constexpr int first(int kind, int P) {
    for (int i = 0; i < N; ++i) if (at(i).kind == kind && at(i).P == P) return i;
    return -1;
  }
  static constexpr int end(int kind, int P) {
    int e = -1;
    for (int i = 0; i < N; ++i) if (at(i).kind == kind && at(i).P == P) e = i + 1;
    return e;
  }
  static constexpr int nkw(const UDesc d) { return (d.kind == U_CRIT && d.P < 5) ? 3 : 1; }     // B-fragment sets
  static constexpr int nblk(const UDesc d) { return d.kind == U_BULK ? 6 - d.P : (d.P < 5 ? 1 : 2); }
  static constexpr int blk0(const UDesc d) { return d.kind == U_BULK ? d.P : d.P - 1; }
  static constexpr int slot(const UDesc d) { return ((d.P == 3 || d.P == 5) ? 2 : 0) + d.c; }  // stage slot
  // fragments of unit j, continuing into the next block's stream (none: that block does not exist)
  static constexpr int nf_at(int j, bool has_next) { return j < N ? at(j).nf : (has_next ? at(j - N).nf : 0); }
  // Counted waits.  Unit i's weights are requested during unit i-3 (wave w copies fragments w, w+4, ..: one
  // per step, the rest after the last step), so every wave has requested AT LEAST nf >> 2 fragments of a unit.
  // Vector memory operations retire in order: `vmcnt(n)` with n = the requests certainly issued SINCE the
  // wanted ones proves those landed whatever else (epilogue stores, bias / halo loads) is in flight as well.
  //   top(i): before unit i, units i+1 and i+2 were requested since;
  //   mid(i): opening unit i+1 inside unit i's last step: unit i+2, and unit i+3's first steps(i)-1 requests.
  static constexpr int steps(int i) { return nkw(at(i)) * nblk(at(i)); }
  static constexpr int wait_top(int i, bool has_next) {
    int n = 0;
    for (int k = 1; k < AH; ++k) n += nf_at(i + k, has_next) >> 2;
    return n;
  }
  // FAT runner: at the top of unit i the NEXT unit's weights must have landed (they are read while unit i's MFMAs
  // issue); requested since: units i+2 .. i+AH-1 (during units i+2-AH .. i-1)
  static constexpr int wait_fat(int i, bool has_next) {
    int n = 0;
    for (int k = 2; k < AH; ++k) n += nf_at(i + k, has_next) >> 2;
    return n;
  }
  // .. strict form: nothing but the requests issued during unit i-1 may still be in flight (the epilogue's stores in
  // front of unit i-1 have landed)
  static constexpr int wait_fat_strict(int i, bool has_next) { return nf_at(i - 1 + AH, has_next) >> 2; }
  static constexpr int wait_mid(int i, bool has_next) {
    static_assert(FAT || AH == 3, "wait_mid counts a lead of 3 units");
    const int part = nf_at(i + 3, has_next) >> 2, done = steps(i) - 1;
    return (nf_at(i + 2, has_next) >> 2) + (part < done ? part : done);
  }
  // mid(i) of the FIRST unit after an epilogue, strict form: nothing but this unit's own requests may still be
  // in flight, i.e. the epilogue's stores (older than those, younger than unit i+2's requests) have landed
  static constexpr int wait_mid_strict(int i, bool has_next) {
    const int part = nf_at(i + 3, has_next) >> 2, done = steps(i) - 1;
    return part < done ? part : done;
  }
  // steps of units [i0, i): which of the two A-fragment register sets unit i starts on
  static constexpr int parity(int i0, int i) {
    int p = 0;
    for (int u = i0; u < i; ++u) p += steps(u);
    return p & 1;
  }
};

// One unit = NKW x NBLK steps of 12 MFMAs (3 kh x 4 rows against one set of 6 B fragments).  Fragments sit
// in ONE set of B registers and two of A (48 registers), refilled in place as their last reader has issued:
// a step's MFMAs 1..6 read B rows 0..2, MFMAs 7..12 rows 3..5, so when the next step (or unit) reads another
// column tap its rows 0..2 are requested after MFMA 6 — together with its A fragments, into the other A set
// — and its rows 3..5 after MFMA 12; the wait in front of a step leaves those last three reads in flight
// (`lgkmcnt(3)`: LDS operations return in order), they are first needed 6 MFMAs later.
// Units run back to back: the next unit's first fragments are requested the same way during this unit's
// last step — after `mid()`, the next unit's DMA wait + barrier, which therefore hides under the remaining
// MFMAs — so a unit opens straight with its MFMAs.  PAR = the A set the unit starts on.
struct UFrags { u32x4 bf[R + 2]; u32x4 a[2][3]; };
template <typename T, int BLK0, int NBLK, int NKW, bool FIRST, bool PRE, bool NXT, int PAR, typename ISSUE, typename MID>
__device__ __forceinline__ void unit_steps(Acc24& acc, UFrags& f, const uint32_t (&lb)[3], const uint32_t lw,
                                           const uint32_t lbn, const uint32_t lwn, ISSUE&& issue, MID&& mid) {
  constexpr int NS = NKW * NBLK;
  u32x4 (&bf)[R + 2] = f.bf;
  if constexpr (!PRE) {
    sfor<R + 2>([&](auto IR) __attribute__((always_inline)) { lds_read16<decltype(IR)::value * IW * 32>(bf[decltype(IR)::value], lb[0]); });
    sfor<3>([&](auto KH) __attribute__((always_inline)) { lds_read16<decltype(KH)::value * 1024>(f.a[PAR][decltype(KH)::value], lw); });
  }
  sfor<NS>([&](auto SI) __attribute__((always_inline)) {
    constexpr int s = decltype(SI)::value;
    constexpr int kwi = s / NBLK, bi = s % NBLK, blk = BLK0 + bi;
    constexpr bool fresh = s == 0 ? PRE : (s / NBLK != (s - 1) / NBLK);   // B rows 3..5 of this step still in flight
    constexpr bool newset = s + 1 < NS && (s + 1) / NBLK != kwi;          // the next step reads another column tap
    u32x4 (&af)[3] = f.a[(PAR + s) & 1];
    u32x4 (&an)[3] = f.a[(PAR + s + 1) & 1];
    // this step's A fragments and B rows 0..2
    if constexpr (fresh) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]) : "n"(R - 1));
    else if constexpr (s == 0)
      asm volatile("s_waitcnt lgkmcnt(0)" : ESR_BF_ALL(bf), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]));
    else lds_wait3(af[0], af[1], af[2]);
    sfor<R + 2>([&](auto IR) __attribute__((always_inline)) {
      constexpr int ir = decltype(IR)::value;
      sfor<3>([&](auto KH) __attribute__((always_inline)) {
        constexpr int kh = decltype(KH)::value;
        constexpr int r = ir - kh;
        if constexpr (r >= 0 && r < R) mma_cls<T, acc_in_agpr(blk), FIRST && kwi == 0 && kh == 0 && blk < 4>(acc_br<blk, r>(acc), af[kh], bf[ir]);
      });
      if constexpr (ir == 2) {                    // after MFMA 6 of 12
#if ESR_R > 1
        if constexpr (fresh) asm volatile("s_waitcnt lgkmcnt(0)" : ESR_BF_HI(bf));
#endif
        if constexpr (s + 1 < NS) {
          if constexpr (newset && !(ESR_ABL & 4))
            sfor<3>([&](auto IR2) __attribute__((always_inline)) { lds_read16<decltype(IR2)::value * IW * 32>(bf[decltype(IR2)::value], lb[newset ? kwi + 1 : 0]); });
          if constexpr (!(ESR_ABL & 2))
            sfor<3>([&](auto KH) __attribute__((always_inline)) { lds_read16<((s + 1) * 3 + decltype(KH)::value) * 1024>(an[decltype(KH)::value], lw); });
        } else if constexpr (NXT) {
          if constexpr (!(ESR_ABL & 8)) mid();
          if constexpr (!(ESR_ABL & 4))
            sfor<3>([&](auto IR2) __attribute__((always_inline)) { lds_read16<decltype(IR2)::value * IW * 32>(bf[decltype(IR2)::value], lbn); });
          sfor<3>([&](auto KH) __attribute__((always_inline)) { lds_read16<decltype(KH)::value * 1024>(an[decltype(KH)::value], lwn); });
        }
      }
    });
    // after MFMA 12: rows 3..5 of the next column tap
    if constexpr (newset && !(ESR_ABL & 4))
      sfor<R - 1>([&](auto IR2) __attribute__((always_inline)) { lds_read16<(3 + decltype(IR2)::value) * IW * 32>(bf[3 + decltype(IR2)::value], lb[newset ? kwi + 1 : 0]); });
    if constexpr (s + 1 == NS && NXT && !(ESR_ABL & 4))
      sfor<R - 1>([&](auto IR2) __attribute__((always_inline)) { lds_read16<(3 + decltype(IR2)::value) * IW * 32>(bf[3 + decltype(IR2)::value], lbn); });
    __builtin_amdgcn_sched_barrier(0);
    issue(std::integral_constant<int, s>{});
    if constexpr (s + 1 == NS && NS < 4)
      sfor<4 - NS>([&](auto X) __attribute__((always_inline)) { issue(std::integral_constant<int, NS + decltype(X)::value>{}); });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// ---- one phase: K steps x 3 column taps over NB cout blocks ----------------------------------------
// The first three weight units are already in flight (issue_w_head, before the neighbour poll).
template <int NB> __device__ __forceinline__ void issue_w_head(const char* wsrc, int K, char* smem, const Tile& t) {
  issue_w<NB * 3>(wsrc, 0, smem, t);
  issue_w<NB * 3>(wsrc, 1, smem, t);
  issue_w<NB * 3>(wsrc, 2, smem, t);      // every phase has >= 6 units
}

template <typename T, int P>
__device__ __forceinline__ void run_phase(Acc24& acc, const char* wsrc, const char* aplane, const int64_t a_gs,
                                          const int K, char* smem, const Tile& t) {
  constexpr int NB = 7 - P, NF = NB * 3;
  const int NU = 3 * K;
  const int nW = (NF + 3 - t.wave) >> 2;                 // weight DMAs this wave issues per unit
  constexpr int nA = NLD;                                // activation DMAs per stage
  issue_a(aplane, 0, smem, t);
  if (K > 1) issue_a(aplane + a_gs, 1, smem, t);
  int g1 = K > 1 ? nA : 0, g2 = 0;                       // DMAs issued in the previous two units
  int sa = 0;
  const int lane = t.lane();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t lds_rows = lds0 + t.wave * (R * IW * 32);
  // one K step = 3 units.  The block's very first K step (phase 1, c = 0) is peeled: its first unit writes
  // the accumulators (SrcC = 0) and must not share a control-flow join with the accumulating form (the phi
  // of a fully allocated AGPR file is resolved through scratch).
  auto kstep = [&](const int c, auto FIRSTC) __attribute__((always_inline)) {
    constexpr bool firstc = decltype(FIRSTC)::value;
    sfor<3>([&](auto KW) __attribute__((always_inline)) {
      constexpr int kw = decltype(KW)::value;
      const int u = 3 * c + kw;
      // in-order return: unit u's weights (issued 3 units ago) and K step c's activations (6 units ago)
      // have landed once only what was issued after them is outstanding
      int n = g1 + g2;
      if (kw == 0) n = firstc ? g1 : n + (c + 1 < K ? nA : 0);
      wait_vm_dyn(n);
      __builtin_amdgcn_s_barrier();       // unit u visible to all waves; all waves done with unit u-1
      int cnt = 0;
      if (u + 3 < NU) { issue_w<NF>(wsrc, u + 3, smem, t); cnt += nW; }
      if (kw == 0 && c + 2 < K) {
        int sn = sa + 2; if (sn >= AR) sn -= AR;
        issue_a(aplane + (int64_t)(c + 2) * a_gs, sn, smem, t);
        cnt += nA;
      }
      g2 = g1; g1 = cnt;
      const uint32_t lb = lds_rows + sa * ASLOT + Tile::colofs(lane, kw);
      const uint32_t lw = lds0 + WOFF + (u & (WR - 1)) * WSLOT + lane * 16;
      unit_mma<T, P, P == 1 && firstc && kw == 0>(acc, lb, lw);
    });
    if (++sa == AR) sa = 0;
  };
  kstep(0, std::true_type{});
#pragma unroll 1
  for (int c = 1; c < K; ++c) kstep(c, std::false_type{});
}

// Resident form: the stages sit in activation slots (Sched::slot); only weights stream, and they stream
// CONTINUOUSLY across units, phases and blocks: unit i of a block sits in ring slot (ring + i) & 3, and while
// it runs the wave issues (between its steps) its share of the unit 3 places further down the schedule (the
// next block's stream after this block's last units).  Everything about a unit except the ring position and
// "is there a next block" is a compile-time constant: the units execute once per block out of a cold
// instruction cache, where every data-dependent branch costs a fetch round trip.
struct WStream {
  const char* w;       // this block's fused weight stream
  const char* wnext;   // the next block's (nullptr: none)
  int ring;            // ring slot of this block's unit 0
  uint64_t* dbg;       // ESR_ABL & 32: time stamps of the traced segment's unit boundaries (measurement builds)
};
#ifndef ESR_DBG_SEG
#define ESR_DBG_SEG 4        // first unit of the traced segment (forward schedule: 4 = bulk_1)
#endif
template <int I0> __device__ __forceinline__ void dbg_stamp(const WStream& s, int k) {
  if constexpr ((ESR_ABL & 32) && I0 == ESR_DBG_SEG) {
    if (s.dbg && threadIdx.x == 0) s.dbg[k] = __builtin_amdgcn_s_memtime();
  }
}
template <typename T, int A, int B> __device__ __forceinline__ void wait_units(const WStream& s) {
  if constexpr (A == B) wait_vm<A>();
  else { if (s.wnext) wait_vm<A>(); else wait_vm<B>(); }
}
// This wave's share of the unit 3 places after unit I: the nf fragments of a unit are split into four
// contiguous runs, wave w copies fragments [w nf / 4, (w+1) nf / 4) — at least nf >> 2, at most 4 — one per
// step.  Source and LDS address of a run are set up ONCE per unit; the requests themselves differ only in the
// instruction's immediate offset, which the LDS-DMA adds to the global AND the LDS address (i * 1024 here): a
// request is one instruction in the shadow of the MFMA issued before it, not an address computation.
struct Ahead { const char* src; char* dst; int cnt; };
template <typename S, int I>
__device__ __forceinline__ Ahead ahead_of(const WStream& s, const Tile& t, char* smem, uint32_t lane16) {
  constexpr int J = I + AH;
  constexpr bool wrap = J >= S::N;
  constexpr UDesc dj = S::at(wrap ? J - S::N : J);
  const int start = (t.wave * dj.nf) >> 2;
  Ahead a;
  a.cnt = (((t.wave + 1) * dj.nf) >> 2) - start;
  if ((wrap && !s.wnext) || (ESR_ABL & 1)) a.cnt = 0;
  a.src = (wrap ? s.wnext : s.w) + (dj.off + start) * 1024 + (size_t)lane16;
  a.dst = smem + WOFF + ((s.ring + J) & (WR - 1)) * WSLOT + start * 1024;
  return a;
}
// SURE = requests every wave issues unconditionally (nf >> 2 of a unit of this block; none when the unit may
// belong to a next block that does not exist)
// MAXC = the most any wave issues (ceil(nf / 4)): the slots behind it cost no test
template <int I_, int SURE = 0, int MAXC = 4> __device__ __forceinline__ void issue_one(const Ahead& a) {
  if constexpr (I_ < 4 && I_ < MAXC) {
    if (I_ < SURE || I_ < a.cnt)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a.src,
                                       (__attribute__((address_space(3))) void*)a.dst, 16, I_ * 1024, 0);
  }
}
template <typename S, int I> constexpr int sure_ahead() {
  return (I + AH >= S::N || (ESR_ABL & 1)) ? 0 : (S::at(I + AH).nf >> 2);
}
template <typename S, int I> constexpr int most_ahead() {
  return (ESR_ABL & 1) ? 0 : (S::at(I + AH >= S::N ? I + AH - S::N : I + AH).nf + 3) >> 2;
}
// units [I0, I1) of the schedule, back to back.  `hook(I)` runs inside unit I after the barrier that opens
// unit I+1 (every wave has waited for everything older than unit I+2's requests by then).
struct NoHook { template <typename X, typename P> __device__ __forceinline__ void operator()(X, P) const {} };
// Phases of hook(I, PH) — the hand-off work that rides on unit I of a bulk: PH_START in front of unit I's MFMAs,
// PH_END in front of the DMA wait + barrier that opens unit I+1, PH_TOP behind that barrier, PH_IN a few MFMAs into
// unit I+1 (FAT runner: behind its MFMA 2; unit_steps runner: in front of unit I+1, i.e. behind the six MFMAs that
// follow the barrier).  The LDS words a phase needs are requested one phase earlier, so that nothing waits for an LDS
// round trip with the matrix pipe idle (round 4: the wave-0 hooks cost 3.5 % of a block on 16-row tiles, 15 % on 4-row
// tiles — address arithmetic from spilled SGPRs and ds_read + lgkmcnt(0) pairs behind the barrier).
enum { PH_START = 0, PH_END = 1, PH_TOP = 2, PH_IN = 3 };
template <int P> using PhC = std::integral_constant<int, P>;
// TRAIL: barrier after the last unit.  The slot of a segment's last unit is next written by the request of a unit
// that runs behind the NEXT segment's opening barrier, so the barrier is only needed where the code that follows
// writes LDS the last unit reads (the block tail's own-pixel writes over x4's slots).
// ---- FAT runner (ESR_FAT) -----------------------------------------------------------------------------
// A unit's MFMAs in issue order: (column tap, input row, kh, cout block) — per accumulator the same (kw, kh) order as
// unit_steps, so the builds agree bit for bit; consecutive MFMAs go to different accumulators wherever the unit has
// more than one (a dependent accumulate issues later than an independent one).
template <int NKW, int NBLK> struct FatShape {
  static constexpr int NA = NKW * NBLK * 3, NB = NKW * (R + 2), NR = NA + NB, NM = NA * R;
  struct Tab {
    int a[NM], b[NM], kwi[NM], ir[NM], kh[NM], bi[NM];   // per MFMA: its fragments and coordinates
    int ra[NM], rb[NM];                                  // refill issued behind MFMA m: A fragment / B fragment index (-1: none)
    int qa[NA], qb[NB];                                  // position of a fragment's refill in the unit's refill order
    int before[NM + 1];                                  // refills issued behind MFMAs 0 .. m-1
    int allow[2][NM];                                    // lgkmcnt in front of MFMA m ([1]: this unit refills too)
    bool wait[2][NM];                                    // .. and whether a wait is needed at all
    int ord0[NR], own0[NM][2], allow0[2][NM];            // the segment's first unit (reads its own fragments): see make()
    bool wait0[2][NM], bad;
  };
#ifndef ESR_P0
#define ESR_P0 8
#endif
  static constexpr int P0 = ESR_P0;
  static constexpr Tab make() {
    Tab t{};
    int m = 0;
    for (int kwi = 0; kwi < NKW; ++kwi)
      for (int ir = 0; ir < R + 2; ++ir)
        for (int kh = 0; kh < 3; ++kh) {
          const int r = ir - kh;
          if (r < 0 || r >= R) continue;
          for (int bi = 0; bi < NBLK; ++bi) {
            t.a[m] = (kwi * NBLK + bi) * 3 + kh; t.b[m] = kwi * (R + 2) + ir;
            t.kwi[m] = kwi; t.ir[m] = ir; t.kh[m] = kh; t.bi[m] = bi;
            ++m;
          }
        }
    int q = 0;
    for (int i = 0; i < NM; ++i) {
      t.before[i] = q;
      bool la = true, lb = true;                         // last reader of its fragments?
      for (int j = i + 1; j < NM; ++j) { if (t.a[j] == t.a[i]) la = false; if (t.b[j] == t.b[i]) lb = false; }
      t.ra[i] = la ? t.a[i] : -1; t.rb[i] = lb ? t.b[i] : -1;
      if (la) t.qa[t.a[i]] = q++;
      if (lb) t.qb[t.b[i]] = q++;
    }
    t.before[NM] = q;
    // In-order return: `lgkmcnt(c)` proves every read but the last c issued has landed.  In front of MFMA i of a unit
    // whose fragments were refilled during the previous unit (NR reads, positions qa / qb), issued since the later of
    // its two fragments: the rest of that unit's refills and, when this unit refills for the next one, before[i].
    for (int nx = 0; nx < 2; ++nx) {
      int proven = -1;                                   // refill position known to have landed
      for (int i = 0; i < NM; ++i) {
        const int pa = t.qa[t.a[i]], pb = t.qb[t.b[i]], pos = pa > pb ? pa : pb;
        int c = (NR - 1 - pos) + (nx ? t.before[i] : 0);
        if (c > 15) c = 15;
        t.allow[nx][i] = c;
        t.wait[nx][i] = pos > proven;
        if (pos > proven) proven = NR + (nx ? t.before[i] : 0) - c - 1;
      }
    }
    // The segment's FIRST unit reads its own fragments, in consumption order (ord0; P0 of them ahead of the first MFMA,
    // two more behind every MFMA — the LDS takes 4 cycles per wave and read, all at once they would keep the matrix
    // pipe waiting for ~300 cycles) and starts its MFMAs as they land.  The issue sequence is simulated here: in
    // front of MFMA i, issued since the later of its two reads = everything behind it.
    {
      int q0 = 0, seen_a[NA] = {}, seen_b[NB] = {};
      for (int i = 0; i < NM; ++i) {
        if (!seen_b[t.b[i]]) { seen_b[t.b[i]] = 1; t.ord0[q0++] = NA + t.b[i]; }
        if (!seen_a[t.a[i]]) { seen_a[t.a[i]] = 1; t.ord0[q0++] = t.a[i]; }
      }
      for (int nx = 0; nx < 2; ++nx) {
        int pos[NR] = {}, issued = 0, next = 0, proven = -1;
        for (int f = 0; f < NR; ++f) pos[f] = -1;
        for (; next < P0 && next < NR; ++next) pos[t.ord0[next]] = issued++;
        for (int i = 0; i < NM; ++i) {
          const int pa = pos[t.a[i]], pb = pos[NA + t.b[i]], need = pa > pb ? pa : pb;
          if (pa < 0 || pb < 0) t.bad = true;
          int c = issued - 1 - need;
          if (c > 15) c = 15;
          t.allow0[nx][i] = c;
          t.wait0[nx][i] = need > proven;
          if (need > proven) proven = issued - c - 1;
          for (int k = 0; k < 2; ++k) {
            t.own0[i][k] = next < NR ? t.ord0[next] : -1;
            if (next < NR) pos[t.ord0[next++]] = issued++;
          }
          if (nx) issued += (t.ra[i] >= 0) + (t.rb[i] >= 0);
        }
        if (next < NR) t.bad = true;
      }
    }
    return t;
  }
  static constexpr Tab tab = make();
};
// the registers of a fragment pass through a statement in front of their MFMA: the wait itself, or nothing
template <int C> __device__ __forceinline__ void lds_wait2(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(C));
}
__device__ __forceinline__ void lds_tie2(u32x4& a, u32x4& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// units [I0, I1) — one segment, all of one shape — back to back.  The fragments of a unit sit in registers before
// its first MFMA: the first unit reads its own (one exposed LDS round trip per segment), every later one was read
// during its predecessor, each register refilled right behind its last reader.  Top of unit i: its DMA wait (unit
// i+1's weights) + barrier + hook(i-1); the ring slot a request of unit i overwrites was last READ during unit
// i+AH-WR-1 <= i-3 and consumed by every wave before the barrier that opened unit i-1.
template <typename S, int I0, int I1, bool STRICT0, bool TRAIL, bool PREW, typename HOOK>
__device__ __forceinline__ void run_units_fat(Acc24& acc, const WStream& s, char* smem, const Tile& t, HOOK&& hook) {
  using T = typename S::Elem;
  constexpr UDesc d0 = S::at(I0);
  constexpr int NKW = S::nkw(d0), NBLK = S::nblk(d0), BLK0 = S::blk0(d0);
  using F = FatShape<NKW, NBLK>;
  static_assert(F::NM >= 4, "a unit carries up to four weight requests per wave");
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t lds_rows = lds0 + t.wave * (R * IW * 32);
  const int lane = t.lane();
  const uint32_t lane16 = (uint32_t)lane * 16u;
  const int colofs[3] = {Tile::colofs(lane, 0), Tile::colofs(lane, 1), Tile::colofs(lane, 2)};
  u32x4 fa[F::NA], fb[F::NB];
  dbg_stamp<I0>(s, 0);
  if constexpr (!PREW) wait_units<T, S::wait_fat(I0, true), S::wait_fat(I0, false)>(s);      // the first two units' weights
  __builtin_amdgcn_s_barrier();
  dbg_stamp<I0>(s, 1);
  const uint32_t lrow0 = lds_rows + S::slot(d0) * ASLOT;
  const uint32_t lw0 = lds0 + WOFF + ((s.ring + I0) & (WR - 1)) * WSLOT + lane16;
  {
    static_assert(!F::tab.bad, "first-unit read schedule");
    sfor<(F::P0 < F::NR ? F::P0 : F::NR)>([&](auto QI) __attribute__((always_inline)) {
      constexpr int f = F::tab.ord0[decltype(QI)::value];
      if constexpr (f >= F::NA) lds_read16<((f - F::NA) % (R + 2)) * IW * 32>(fb[f - F::NA], lrow0 + colofs[NKW == 3 ? (f - F::NA) / (R + 2) : d0.kw]);
      else lds_read16<f * 1024>(fa[f], lw0);
    });
  }
  dbg_stamp<I0>(s, 2);
  sfor<I1 - I0>([&](auto II) __attribute__((always_inline)) {
    constexpr int I = I0 + decltype(II)::value;
    constexpr UDesc d = S::at(I);
    constexpr bool NXT = I + 1 < I1;
    constexpr UDesc dn = S::at(NXT ? I + 1 : I);
    static_assert(S::nkw(d) == NKW && S::nblk(d) == NBLK && S::blk0(d) == BLK0, "one shape per segment");
    constexpr bool FIRST = d.P == 1 && d.c == 0 && d.kw == 0;
    const uint32_t lrn = lds_rows + S::slot(dn) * ASLOT;
    const uint32_t lbn[3] = {lrn + colofs[NKW == 3 ? 0 : dn.kw], lrn + colofs[1], lrn + colofs[2]};
    const uint32_t lwn = lds0 + WOFF + ((s.ring + I + 1) & (WR - 1)) * WSLOT + lane16;
    const Ahead ah = ahead_of<S, I>(s, t, smem, lane16);
    if constexpr (NXT) hook(std::integral_constant<int, I>{}, PhC<PH_START>{});
    sfor<F::NM>([&](auto MI) __attribute__((always_inline)) {
      constexpr int m = decltype(MI)::value;
      constexpr int a = F::tab.a[m], b = F::tab.b[m], blk = BLK0 + F::tab.bi[m], r = F::tab.ir[m] - F::tab.kh[m];
      constexpr int WC = (I > I0 && F::tab.wait[NXT][m] && !(ESR_ABL & 6)) ? F::tab.allow[NXT][m]
                         : (I == I0 && F::tab.wait0[NXT][m]) ? F::tab.allow0[NXT][m] : -1;
      mma_w<acc_in_agpr(blk), FIRST && F::tab.kwi[m] == 0 && F::tab.kh[m] == 0 && blk < 4, WC>(acc_br<blk, r>(acc), fa[a], fb[b]);
      if constexpr (I == I0) {
        sfor<2>([&](auto KI) __attribute__((always_inline)) {
          constexpr int f = F::tab.own0[m][decltype(KI)::value];
          if constexpr (f >= F::NA) lds_read16<((f - F::NA) % (R + 2)) * IW * 32>(fb[f - F::NA], lrow0 + colofs[NKW == 3 ? (f - F::NA) / (R + 2) : d0.kw]);
          else if constexpr (f >= 0) lds_read16<f * 1024>(fa[f], lw0);
        });
      }
      if constexpr (NXT) {
        if constexpr (F::tab.ra[m] >= 0 && !(ESR_ABL & 2)) lds_read16<F::tab.ra[m] * 1024>(fa[F::tab.ra[m]], lwn);
        if constexpr (F::tab.rb[m] >= 0 && !(ESR_ABL & 4))
          lds_read16<(F::tab.rb[m] % (R + 2)) * IW * 32>(fb[F::tab.rb[m]], lbn[F::tab.rb[m] / (R + 2)]);
      }
      // the wave's (up to four) weight requests, spread over the unit: the LDS-DMA path takes 64 bytes per cycle and
      // CU (tools/experiments/issue_probe.hip: one 1 KB request per wave and MFMA doubles the MFMA's time, one per four
      // costs 7 %) — four waves requesting behind four consecutive MFMAs stall on it
      if constexpr (m == 2 && I > I0) hook(std::integral_constant<int, I - 1>{}, PhC<PH_IN>{});
      {
        constexpr int MC = most_ahead<S, I>();
        constexpr int k = MC ? (m * MC + F::NM - 1) / F::NM : 0;       // the k-th request sits behind MFMA k NM / MC
        if constexpr (MC > 0 && k < MC && m == k * F::NM / MC) {
          __builtin_amdgcn_sched_barrier(0);
          issue_one<k, sure_ahead<S, I>(), MC>(ah);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
    if constexpr (NXT) {
      // top of unit I+1: its successor's weights, the barrier, hook(I)
      hook(std::integral_constant<int, I>{}, PhC<PH_END>{});
      if constexpr (!(ESR_ABL & 256)) {
        if constexpr (STRICT0 && I == I0 && !(ESR_ABL & 2048)) wait_units<T, S::wait_fat_strict(I + 1, true), S::wait_fat_strict(I + 1, false)>(s);
        else wait_units<T, S::wait_fat(I + 1, true), S::wait_fat(I + 1, false)>(s);
      }
      dbg_stamp<I0>(s, 3 + 3 * (I - I0));
      if constexpr (!(ESR_ABL & 128)) __builtin_amdgcn_s_barrier();
      dbg_stamp<I0>(s, 4 + 3 * (I - I0));
      if constexpr (!(ESR_ABL & 512)) hook(std::integral_constant<int, I>{}, PhC<PH_TOP>{});
      dbg_stamp<I0>(s, 5 + 3 * (I - I0));
    } else {
      dbg_stamp<I0>(s, 3 + 3 * (I - I0));
    }
  });
  if constexpr (TRAIL) __builtin_amdgcn_s_barrier();
}

// PREW: the wait for the segment's first weights already happened (prewait_units, in front of the epilogue whose stores
// would otherwise stand between those requests and the allowed count: vmcnt retires in order)
template <typename S, int I0> __device__ __forceinline__ void prewait_units(const WStream& s) {
  using T = typename S::Elem;
  if constexpr (FAT) wait_units<T, S::wait_fat(I0, true), S::wait_fat(I0, false)>(s);
  else wait_units<T, S::wait_top(I0, true), S::wait_top(I0, false)>(s);
}
template <typename S, int I0, int I1, bool STRICT0 = false, bool TRAIL = true, bool PREW = false, typename HOOK = NoHook>
__device__ __forceinline__ void run_units_s(Acc24& acc, const WStream& s, char* smem, const Tile& t, HOOK&& hook = NoHook{}) {
  using T = typename S::Elem;
  if constexpr (FAT) { run_units_fat<S, I0, I1, STRICT0, TRAIL, PREW>(acc, s, smem, t, hook); return; }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t lds_rows = lds0 + t.wave * (R * IW * 32);
  UFrags f;
  const int lane = t.lane();
  const uint32_t lane16 = (uint32_t)lane * 16u;
  const int colofs[3] = {Tile::colofs(lane, 0), Tile::colofs(lane, 1), Tile::colofs(lane, 2)};
  dbg_stamp<I0>(s, 0);
  if constexpr (!PREW) wait_units<T, S::wait_top(I0, true), S::wait_top(I0, false)>(s);      // the first unit's weights
  __builtin_amdgcn_s_barrier();
  dbg_stamp<I0>(s, 1);
  dbg_stamp<I0>(s, 2);
  sfor<I1 - I0>([&](auto II) __attribute__((always_inline)) {
    constexpr int I = I0 + decltype(II)::value;
    constexpr UDesc d = S::at(I);
    constexpr UDesc dn = S::at(I + 1 < I1 ? I + 1 : I);
    constexpr int NKW = S::nkw(d);
    const uint32_t lrow = lds_rows + S::slot(d) * ASLOT;
    const uint32_t lb[3] = {lrow + colofs[NKW == 3 ? 0 : d.kw], lrow + colofs[1], lrow + colofs[2]};
    const uint32_t lw = lds0 + WOFF + ((s.ring + I) & (WR - 1)) * WSLOT + lane16;
    const uint32_t lbn = lds_rows + S::slot(dn) * ASLOT + colofs[S::nkw(dn) == 3 ? 0 : dn.kw];
    const uint32_t lwn = lds0 + WOFF + ((s.ring + I + 1) & (WR - 1)) * WSLOT + lane16;
    const Ahead ah = ahead_of<S, I>(s, t, smem, lane16);
    auto issue = [&](auto SI) __attribute__((always_inline)) { issue_one<decltype(SI)::value, sure_ahead<S, I>()>(ah); };
    if constexpr (I > I0) hook(std::integral_constant<int, I - 1>{}, PhC<PH_IN>{});
    if constexpr (I + 1 < I1) hook(std::integral_constant<int, I>{}, PhC<PH_START>{});
    auto mid = [&]() __attribute__((always_inline)) {
      hook(std::integral_constant<int, I>{}, PhC<PH_END>{});
      if constexpr (!(ESR_ABL & 256)) {
      if constexpr (STRICT0 && I == I0 && !(ESR_ABL & 2048)) wait_units<T, S::wait_mid_strict(I, true), S::wait_mid_strict(I, false)>(s);
      else wait_units<T, S::wait_mid(I, true), S::wait_mid(I, false)>(s);   // the next unit's weights
      }
      dbg_stamp<I0>(s, 3 + 3 * (I - I0));
      if constexpr (!(ESR_ABL & 128)) __builtin_amdgcn_s_barrier();        // next unit visible to all waves; all waves past this unit's LDS reads
      dbg_stamp<I0>(s, 4 + 3 * (I - I0));
      if constexpr (!(ESR_ABL & 512)) hook(std::integral_constant<int, I>{}, PhC<PH_TOP>{});
      dbg_stamp<I0>(s, 5 + 3 * (I - I0));
    };
    unit_steps<T, S::blk0(d), S::nblk(d), NKW, (d.P == 1 && d.c == 0 && d.kw == 0), (I > I0), (I + 1 < I1), S::parity(I0, I)>(
        acc, f, lb, lw, lbn, lwn, issue, mid);
  });
  if constexpr (TRAIL) __builtin_amdgcn_s_barrier();            // every wave done with the last unit's slots
}
template <typename T, int I0, int I1, bool STRICT0 = false, bool TRAIL = true, bool PREW = false, typename HOOK = NoHook>
__device__ __forceinline__ void run_units(Acc24& acc, const WStream& s, char* smem, const Tile& t, HOOK&& hook = NoHook{}) {
  run_units_s<Sched<T>, I0, I1, STRICT0, TRAIL, PREW>(acc, s, smem, t, hook);
}

// P = conv1x1(x) from the resident x stages (slots 0..KX-1); its fragments are one unit of the weight
// stream (after bulk_1).
template <typename T>
__device__ __forceinline__ void run_1x1_res(Acc24& acc, const WStream& s, char* smem, const Tile& t) {
  using CF = Cfg<T>;
  using S = Sched<T>;
  constexpr int K = CF::KX;
  constexpr int I = S::first(U_ONE, 1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = t.lane();
  const uint32_t lb = lds0 + t.wave * (R * IW * 32) + IW * 32 + Tile::colofs(lane, 1);   // centre tap: rows 1..4, col j+1
  const uint32_t lw = lds0 + WOFF + ((s.ring + I) & (WR - 1)) * WSLOT + lane * 16;
  wait_units<T, S::wait_top(I, true), S::wait_top(I, false)>(s);
  __builtin_amdgcn_s_barrier();
  { const Ahead ah = ahead_of<S, I>(s, t, smem, (uint32_t)lane * 16u);
    sfor<4>([&](auto X) __attribute__((always_inline)) { issue_one<decltype(X)::value>(ah); }); }
  // two fragment sets: K step c+1 is requested before the MFMAs of step c (one exposed LDS round trip, not K)
  u32x4 fa[2], fb[2][R];
  auto rd = [&](auto CI, auto SET) __attribute__((always_inline)) {
    constexpr int c = decltype(CI)::value, st = decltype(SET)::value;
    lds_read16<c * 1024>(fa[st], lw);
    sfor<R>([&](auto RR) __attribute__((always_inline)) { lds_read16<c * ASLOT + decltype(RR)::value * IW * 32>(fb[st][decltype(RR)::value], lb); });
  };
  rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  sfor<K>([&](auto CI) __attribute__((always_inline)) {
    constexpr int c = decltype(CI)::value, st = c & 1;
    if constexpr (c + 1 < K) {
      rd(std::integral_constant<int, c + 1>{}, std::integral_constant<int, st ^ 1>{});
      asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(fa[st]), ESR_ROWS(fb[st]) : [n] "n"(R + 1));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[st]), ESR_ROWS(fb[st]));
    }
    sfor<R>([&](auto RR) __attribute__((always_inline)) {
      mma_cls<T, acc_in_agpr(0), c == 0>(acc_br<0, decltype(RR)::value>(acc), fa[st], fb[st][decltype(RR)::value]);
    });
  });
}

// ---- P = conv1x1(x) on the tile's own pixels (block.py:263), into cout block 0's registers ----------
template <typename T>
__device__ __forceinline__ void run_1x1(Acc24& acc, const char* w1, const char* aplane, const int64_t a_gs,
                                        char* smem, const Tile& t) {
  constexpr int K = Cfg<T>::KX;
  const int lane = t.lane();
  // all K fragments -> weight slot 0 (wave w copies fragments w, w+4, ...)
#pragma unroll
  for (int i = 0; i < (K + 3) / 4; ++i) {
    const int q = t.wave + 4 * i;
    if (q < K) dma16(w1 + q * 1024 + lane * 16, smem + WOFF + q * 1024);
  }
  issue_a(aplane, 0, smem, t);
  issue_a(aplane + a_gs, 1, smem, t);
  int sa = 0;
  const char* lds_rows = smem + t.wave * (R * IW * 32) + IW * 32 + Tile::colofs(lane, 1);   // centre tap: rows 1..4, col j+1
#pragma unroll 1
  for (int c = 0; c < K; ++c) {
    if (c + 1 < K) wait_vm<NLD>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (c + 2 < K) {
      int sn = sa + 2; if (sn >= AR) sn -= AR;
      issue_a(aplane + (int64_t)(c + 2) * a_gs, sn, smem, t);
    }
    {                                       // the 1x1 lands in conv1's vacated registers (block 0)
      const u32x4 a = *(const u32x4*)(smem + WOFF + c * 1024 + lane * 16);
      const char* lb = lds_rows + sa * ASLOT;
      u32x4 bq[R];
#pragma unroll
      for (int r = 0; r < R; ++r) bq[r] = *(const u32x4*)(lb + r * IW * 32);
      if (c == 0) {
        sfor<R>([&](auto RR) __attribute__((always_inline)) { mma_cls<T, acc_in_agpr(0), true>(acc_br<0, decltype(RR)::value>(acc), a, bq[decltype(RR)::value]); });
      } else {
        sfor<R>([&](auto RR) __attribute__((always_inline)) { mma_cls<T, acc_in_agpr(0)>(acc_br<0, decltype(RR)::value>(acc), a, bq[decltype(RR)::value]); });
      }
    }
    if (++sa == AR) sa = 0;
  }
}

// measurement only: time stamps (100 MHz) of the tile's SECOND block (the first one stages x differently)
// The scalars of the launch the boundary code reads, as plain values (training / backward instantiations): passing
// the kernel-argument struct itself by reference makes hipcc keep a scratch copy of it, and every `p.H` in an
// epilogue becomes a scratch load whose wait drains the weight stream (cdna guide: vmcnt counts everything)
struct PS {
  int H, W, noise_mode, save_dense, _pad2;
  float sigma;
  uint64_t seed;
  const uint64_t* seed_dev;
  struct { int wp; } dense;
  uint64_t* trace;
};
// banded launch (esr_rdb_chain.band_rows): the scalars + the band geometry
struct PSB : PS {
  int band_rows, band_margin, img_H;
};
template <typename PT>
__device__ __forceinline__ void trace_ev(const PT& p, int tile, int& ev) {
  if (p.trace && ev >= 0 && ev < 64 && threadIdx.x == 0) p.trace[(int64_t)tile * 64 + ev] = (ESR_ABL & 16) ? __builtin_amdgcn_s_memtime() : __builtin_amdgcn_s_memrealtime();
  if (ev >= 0) ++ev;
}

// ---- publish / consume ------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned gu32;

template <typename PT = esr_rdb_chain>
__device__ __forceinline__ void publish(unsigned* flags, int tile, unsigned epoch, const Tile& t,
                                        const PT* tp = nullptr, int* ev = nullptr) {
  if (tp) trace_ev(*tp, tile, *ev);                      // epilogue done (stores issued)
  if constexpr (!(ESR_ABL & 1024)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains its sc1 stores
  if (tp) trace_ev(*tp, tile, *ev);                      // own stores drained
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store((gu32*)(flags + tile), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0, lanes 0..7 poll one neighbour each (relaxed, agent scope) until all reached `epoch`.
// Returns false (whole workgroup) on abort / time-out.
template <typename PT = esr_rdb_chain>
__device__ __forceinline__ bool wait_neighbours(unsigned* ws, unsigned epoch, char* smem, const Tile& t,
                                                const PT* tp = nullptr, int* ev = nullptr, int tile_ = 0) {
  if constexpr ((ESR_ABL & 1024) != 0) { if (tp) trace_ev(*tp, tile_, *ev); __syncthreads(); return true; }   // "free hand-off" bound
  if (t.wave == 0) {
    // the neighbour this lane polls (lanes 0..7)
    const int lane = t.lane();
    int my_nbr_tile = -1;
    if (lane < 8) {
      const int k = lane < 4 ? lane : lane + 1;
      const int ny = t.ty + k / 3 - 1, nx = t.tx + k % 3 - 1;
      if (ny >= 0 && ny < t.tiles_y && nx >= 0 && nx < t.tiles_x) my_nbr_tile = (t.b * t.tiles_y + ny) * t.tiles_x + nx;
    }
    bool ok = my_nbr_tile < 0;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    bool dead = false;
    for (unsigned it = 1;; ++it) {
      if (!ok) ok = __hip_atomic_load((gu32*)(ws + WS_HDR + my_nbr_tile), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
      if (__all(ok)) break;
      // the abort word is a second dependent round trip: look at it (and at the clock) every 16th turn only
      if ((it & 15u) == 0u && ((__builtin_amdgcn_s_memrealtime() - t0) > 100000000ull ||      // 1 s of the 100 MHz counter
                               __hip_atomic_load((gu32*)(ws + WS_ABORT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        dead = true;
        break;
      }
    }
    if (lane == 0) {
      if (dead) {
        __hip_atomic_store((gu32*)(ws + WS_ABORT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // sticky report for the host (pinned memory, read without a synchronisation at the library's next entry)
        unsigned* const ha = *(unsigned* volatile*)(smem + LDS_CTRL + 48);
        if (ha) __hip_atomic_store(ha, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      *(volatile int*)(smem + LDS_CTRL + 16) = dead ? 1 : 0;
    }
  }
  if (tp) trace_ev(*tp, tile_, *ev);                     // neighbours' flags seen (wave 0)
  __syncthreads();
  const int dead = *(volatile int*)(smem + LDS_CTRL + 16);
  return dead == 0;
}

// Non-blocking form for the bulks (wave 0 only): the 8 flags are fetched by LDS-DMA into LDS_FLAGS — no
// register result, hence nothing to wait for — and looked at some units later with plain LDS reads: a flag that
// has not landed yet simply still shows its older (smaller) value and sends the tile through the blocking
// poll after the bulk.  LDS accesses here are inline asm: hipcc orders a visible LDS access after every
// LDS-DMA in flight with `vmcnt(0)`, which would drain the weight stream.
// The hand-off in the phases of the unit runners (PH_*): every LDS word is requested one phase before it is used.
// State that crosses phases (VGPRs that live for one unit).
struct PollRegs { unsigned nbr, flag, tag, hsrc; };
__device__ __forceinline__ void lds_req32(unsigned& d, uint32_t addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr)); }
// a word requested one phase ago.  FAT runner: at least six fragment reads were issued behind it (in-order return);
// unit_steps runner: nothing is in flight at the places that ask (its steps drain their reads)
__device__ __forceinline__ void lds_old(unsigned& d) {
  if constexpr (FAT) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(d));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d));
}
// the two words of PH_TOP, asked for in PH_IN.  unit_steps runner: the last step requested the next unit's first six
// fragments behind them (and R - 1 more): leave those in flight
__device__ __forceinline__ void lds_old2(unsigned& a, unsigned& b) {
  if constexpr (FAT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
  else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a), "+v"(b));
}
// the byte offset of the flag a lane of wave 0 fetches (lanes 0..7: its neighbour; the others and absent neighbours:
// the tile's own flag, which is up) — once per tile, by wave 0
__device__ __forceinline__ void stage_nbr(int tile, char* smem, const Tile& t) {
  const int lane = t.lane();
  int nbr = tile;
  if (lane < 8) {
    const int k = lane < 4 ? lane : lane + 1;
    const int ny = t.ty + k / 3 - 1, nx = t.tx + k % 3 - 1;
    if (ny >= 0 && ny < t.tiles_y && nx >= 0 && nx < t.tiles_x) nbr = (t.b * t.tiles_y + ny) * t.tiles_x + nx;
  }
  *(volatile int*)(smem + LDS_NBR + lane * 4) = (WS_HDR + nbr) * 4;
}
__device__ __forceinline__ void poll_issue_at(unsigned* ws, unsigned off, char* smem) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)ws + off),
                                   (__attribute__((address_space(3))) void*)(smem + LDS_FLAGS), 4, 0, 16);
}
// wave 0, in front of the barrier: all neighbours up -> the tag every wave reads behind it
__device__ __forceinline__ void poll_tag(unsigned flag, unsigned epoch, uint32_t lds0, const Tile& t) {
  const bool ok = __all(flag >= epoch);
  const unsigned tag = ok ? epoch : 0u;
  if (t.lane() == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + LDS_CTRL + 32), "v"(tag) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ---- epilogue of one finished 32-cout block ---------------------------------------------------------
// MODE 0: v = lrelu(acc + bias)                               (x1, x3)
// MODE 1: v = lrelu(acc + bias) + P (block 0's accumulators)  (x2, block.py:263)
// MODE 2: v = lrelu(acc + bias) + ex[own pixel]               (x4 = lrelu(conv4) + x2, block.py:266)
// MODE 3: v = (acc + bias)*0.2 + ex; noise1; [v = v*0.2 + r2; noise2]   (block.py:267-268, 291)
// ex / r2 = the lane's own pixels of 4 rows in storage form, fetched by load_rows() well ahead of use
// (a conditional load inside the row loop makes hipcc wait for every load separately) or kept from the
// epilogue that produced them.
// LW (fp16 resident path), bit 0: also write the lane's pixels into activation slots slot0 + h of the LDS
// (the next phase's stage); bit 1: hand them back in `keep` (conv1: the 1x1 still reads x in those slots;
// conv2: x2 is the residual of x4).
template <typename T> struct RowsRaw { typename Ch16<T>::Raw q[R]; };

// `live` false: the same loads at an offset past num_records — the buffer range check answers 0 without touching
// memory.  The RRDB residual rows are fetched this way in EVERY block: a load under `if (has_res2)` whose result is
// used under another `if (has_res2)` makes hipcc send the rows through scratch one by one, each behind a vmcnt(0).
template <typename T, typename PT>
__device__ __forceinline__ void load_rows(const ImgView& v, int cb, const PT& p, const Tile& t, RowsRaw<T>& o, bool live = true) {
  const int lane = t.lane();
  const int ox = t.ox0 + (lane & 31), oyb = t.oy0 + t.wave * R, wp32 = p.dense.wp * 32;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int oy = oyb + r < p.H ? oyb + r : p.H - 1;            // clamped: rows / columns past the image are not used
    const int off = (oy + 1) * wp32 + (ox < p.W ? ox + 1 : 1) * 32;
    Ch16<T>::load(v, live ? cb : 0, lane >> 5, live ? off : (int)0x80000000u, o.q[r]);
  }
}

struct Bias16 { f32x4 q[4]; };
// the lane's 16 biases of a cout block (bias = wave-uniform pointer): requested at the START of the phase
// whose epilogue adds them — a load placed in the epilogue itself exposes a memory round trip per phase
__device__ __forceinline__ void load_bias(const float* bias, const Tile& t, Bias16& b) {
  const f32x4* bp = (const f32x4*)bias + 4 * (t.lane() >> 5);
#pragma unroll
  for (int i = 0; i < 4; ++i) b.q[i] = bp[i];
}
// fp16 path: the block's biases sit in the LDS (copied by DMA at the top of the block): a register copy would
// be a vector-memory load whose first use makes hipcc drain every weight DMA in flight (vmcnt(0))
__device__ __forceinline__ void stage_bias(const float* bias, char* smem, const Tile& t) {
  if (t.wave < 3)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bias + t.wave * 64 + t.lane()),
                                     (__attribute__((address_space(3))) void*)(smem + LDS_BIAS + t.wave * 256), 4, 0, 0);
}
__device__ __forceinline__ void lds_bias(const char* smem, int first, const Tile& t, Bias16& b) {
  const f32x4* bp = (const f32x4*)(smem + LDS_BIAS + first * 4) + 4 * (t.lane() >> 5);
#pragma unroll
  for (int i = 0; i < 4; ++i) b.q[i] = bp[i];
}
// the block-table entry's scalars, read ONCE per block into SGPRs (a field referenced through the table is
// re-loaded with a vector load + full wait wherever it is used)
struct BlkS {
  const float* bias;       // [192]: conv1..conv4 (32 each), conv5 (64)
  uint32_t layer1, layer2;
  bool has_res2, full_out;
  bool band_own;           // banded launch: x_out takes only the band's own rows
};
// NOISE (MODE 3): false = the instantiation without the Philox layers and the explicit `+ x` (the fp16 path's
// common case takes it through one uniform branch: the tail is executed once per block out of a cold
// instruction cache, so what is not needed should not be in the way)
// TRAIN (training forward, esr_rdb_chain.mode 1): every slice reaches memory in full (the backward and the weight
// gradients read it), and MODE 0..2 also write the slice's LeakyReLU masks — bit 15 - e of row r = sign bit of
// lrelu(acc + bias) for the lane's channel e (set: slope 0.2) — as 8 bytes per lane into the tile's mask record (mask_base: this tile's record, MASK_TILE bytes;
// slice 0..3 = a1..a4): x2 and x4 carry residuals, so the sign of their pre-activation cannot be read off the
// stored slice.
template <typename T, int BLK, int MODE, int LW = 0, bool NOISE = true, bool TRAIN = false, typename PT = esr_rdb_chain>
__device__ __forceinline__ void epilogue(Acc24& acc, const PT& p, const BlkS& blk, const Bias16& bias,
                                         const ImgView& out, int out_cb, int ch_cb, const RowsRaw<T>* ex,
                                         const RowsRaw<T>* r2, bool has_res2, const Tile& t, char* smem = nullptr,
                                         int slot0 = 0, RowsRaw<T>* keep = nullptr, float carry_scale = 0.f,
                                         bool full_store = true, char* mask_base = nullptr, int mask_slice = 0) {
  using C16 = Ch16<T>;
  const int lane = t.lane(), tj = lane & 31, th = lane >> 5;
  int own_px = 0, own_swz = 0;
  if constexpr ((LW & 1) != 0) t.own(lane, own_px, own_swz);
  const int ox = t.ox0 + tj;
  const int oyb = t.oy0 + t.wave * R;
  const f32x4 (&bq)[4] = bias.q;
  const int wp32 = p.dense.wp * 32;
  constexpr bool BAND = std::is_same_v<PT, PSB>;
  const bool ragged = BAND || t.oy0 + TH > p.H || t.ox0 + TW > p.W;      // wave-uniform
  const uint32_t layer1 = blk.layer1, layer2 = blk.layer2;
  const bool n1 = MODE == 3 && NOISE && p.noise_mode == ESR_NOISE_PHILOX && layer1 != ESR_NO_LAYER;
  const bool n2 = MODE == 3 && NOISE && p.noise_mode == ESR_NOISE_PHILOX && layer2 != ESR_NO_LAYER && has_res2;
  uint64_t seed = p.seed;
  if ((n1 || n2) && p.seed_dev) seed = __builtin_nontemporal_load(p.seed_dev);
  uint32_t mbits[4] = {0u, 0u, 0u, 0u};
  sfor<R>([&](auto RR) __attribute__((always_inline)) {
    constexpr int r = decltype(RR)::value;
    const int oy = oyb + r;
    const f32x16 a = acc_br<BLK, r>(acc);
    float v[16], tmp[16];
    // pairs: v_pk_add_f32 / v_pk_mul_f32 do two elements per instruction (same IEEE results as the scalar forms)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const f32x2 av = {a[e], a[e + 1]}, bv = {bq[e >> 2][e & 3], bq[e >> 2][(e & 3) + 1]};
      const f32x2 x = av + bv;
      if constexpr (MODE != 3) {
        const f32x2 y = x * ESR_LRELU_SLOPE;                                     // LeakyReLU(0.2) = max(x, 0.2 x)
        v[e] = __builtin_fmaxf(x[0], y[0]);
        v[e + 1] = __builtin_fmaxf(x[1], y[1]);
        if constexpr (TRAIN) {
          // one v_alignbit per element: shift the sign bit of lrelu(a) (= the sign of a) into the row's mask word
          // (channel e ends up at bit 15 - e; 1 = negative -> slope 0.2)
          mbits[r] = __builtin_amdgcn_alignbit(mbits[r], __builtin_bit_cast(uint32_t, v[e]), 31);
          mbits[r] = __builtin_amdgcn_alignbit(mbits[r], __builtin_bit_cast(uint32_t, v[e + 1]), 31);
        }
      } else {
        v[e] = x[0];
        v[e + 1] = x[1];
      }
    }
    if constexpr (MODE == 1) {
      const f32x16 a1 = acc_br<0, r>(acc);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += a1[e];
    }
    if constexpr (MODE == 2) {
      C16::get(ex->q[r], tmp);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = v[e] * 1.0f + tmp[e];
    }
    if constexpr (MODE == 3) {
      if (NOISE && ex) {           // explicit residual (fp32 path, noise): out = conv5*0.2 + x
        C16::get(ex->q[r], tmp);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] * 0.2f + tmp[e];
      } else {                     // folded: the accumulators started at 5 x
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] *= 0.2f;
      }
      const uint32_t pix = (uint32_t)((t.b * p.H + oy) * p.W + ox);
      if (n1) {
#pragma unroll
        for (int o = 0; o < 2; ++o) philox_normal8(pix, (uint32_t)(ch_cb * 4 + th * 2 + o), layer1, seed, &tmp[8 * o]);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] + tmp[e] * (p.sigma * v[e]);     // block.py:119-121
      }
      if (has_res2) {
        C16::get(r2->q[r], tmp);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] * 0.2f + tmp[e];
        if (n2) {
#pragma unroll
          for (int o = 0; o < 2; ++o) philox_normal8(pix, (uint32_t)(ch_cb * 4 + th * 2 + o), layer2, seed, &tmp[8 * o]);
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = v[e] + tmp[e] * (p.sigma * v[e]);
        }
      }
    }
    // pixels beyond the image: offset 2^31 lies past num_records (2^31 - 1), the buffer range check drops the store
    bool live = oy < p.H && ox < p.W;          // a pixel of the image
    bool inside = live;                        // ... that this epilogue stores
    if constexpr (BAND) {
      // a band's view covers band_margin rows of its neighbours (recomputed here, owned there) and, at the ends of
      // the image, rows that do not exist: those stay zero everywhere (the padding the convs must see)
      const int g = t.b * p.band_rows - p.band_margin + oy;
      live = live && g >= 0 && g < p.img_H;
      inside = live && (!(MODE == 3 && blk.band_own) || (oy >= p.band_margin && oy < p.band_margin + p.band_rows));
    }
    const int po = (oy + 1) * wp32 + (ox + 1) * 32;
    if constexpr (LW == 0) {
      C16::store(out, inside ? out_cb : 0, th, inside ? po : (int)0x80000000u, v);
    } else {
      typename C16::Raw q;
      C16::pack(v, q.q);
      // x1..x4 are only ever read back as HALO pixels by the neighbouring tiles (the tile's own pixels stay
      // in the LDS / registers): unless the caller wants the dense slices in memory (save_dense), only the
      // tile's border pixels are stored — 82 % fewer bytes through the lock-stepped store bursts
      const bool edge = TRAIN || (MODE == 3 && full_store) || p.save_dense || tj == 0 || tj == TW - 1 || (t.wave == 0 && r == 0) ||
                        (t.wave == NT / 64 - 1 && r == R - 1);
      C16::store_packed_rows(out, (inside && edge) ? out_cb : 0, th, (inside && edge) ? po : (int)0x80000000u, q.q);
      // beyond the image: the zero padding (only tiles that stick out of the image have such pixels: one scalar test)
      if (ragged && !live) { q.q[0] = u32x4{0, 0, 0, 0}; q.q[1] = u32x4{0, 0, 0, 0}; }
      if constexpr (LW & 1) lds_put_row(smem, slot0 + th, r, q.q, own_px, own_swz);
      if constexpr (LW & 2) keep->q[r] = q;
      if constexpr (MODE == 3) {
        // carry into the next block: its conv5 accumulators start at 5 x (x = this output AS STORED), so
        // that block's tail `conv5*0.2 + x` needs no residual read (zero when it adds x explicitly)
        float xs[16];
        C16::get(q, xs);
        f32x16 nx;
#pragma unroll
        for (int e = 0; e < 16; ++e) nx[e] = carry_scale * xs[e];
        acc_br<BLK, r>(acc) = nx;
      }
    }
  });
  if constexpr (TRAIN && MODE != 3) {
    // 2 R bytes per lane: [wave][lane][row] u16
    char* const mp = mask_base + mask_slice * MASK_SLICE + (t.wave * 64 + lane) * (2 * R);
    if constexpr (R == 4) *(u32x2*)mp = u32x2{(mbits[0] & 0xFFFFu) | (mbits[1] << 16), (mbits[2] & 0xFFFFu) | (mbits[3] << 16)};
    else if constexpr (R == 2) *(uint32_t*)mp = (mbits[0] & 0xFFFFu) | (mbits[1] << 16);
    else *(uint16_t*)mp = (uint16_t)mbits[0];
  }
}

// ======================================= backward chain (esr_rdb_backward) =======================================
// The input gradients of a dense block have the forward's shape with the roles mirrored ("gather
// form"): with  Q = [g_t (64) | g_a4 | g_a3 | g_a2 | g_a1]  the gradient of channel slice x4, x3, x2, x1, x is ONE conv
// over a growing prefix of Q whose operand is gathered from the transposed / rotated forward weights, so the backward
// of a block runs the forward's five phases over Q:
//   phase 1  stage g_t  -> accumulate into g_x4, g_x3, g_x2, g_x1, g_x     (cout blocks 0..3, 4/5)
//   phase 2  stage g_a4 -> g_x3, g_x2, g_x1, g_x        ...        phase 5  stage g_a1 -> g_x
// and after phase p the finished slice leaves through an epilogue that applies the LeakyReLU mask of the forward's
// pre-activation (saved by the training forward as one bit per element):  g_a = g_x * (a > 0 ? 1 : 0.2).
// Two things differ from the forward's dataflow:
//   * x4 = lrelu(a4) + x2 (block.py:266): the identity path is folded into the operand (conv5's x4 columns are added
//     to its x2 columns at pack time), so g_x2 needs no residual;
//   * x2 = lrelu(a2) + conv1x1(x) (block.py:263): g_x += W1x1^T g_x2 with the UNMASKED g_x2.  Its 16 MFMAs per wave
//     run inside the epilogue of g_x2 straight from registers: the lane's 16 packed channels of a pixel ARE the two
//     B fragments of the 32-channel contraction under the K order  k = 8 h + i  <->  channel 16 h + 8 c + i  (chunk
//     c), which the 1x1's A fragments are packed for (esr_pack.one_t).  The unmasked g_x2 is also stored (32 channels,
//     `aux`): the 1x1's weight gradient needs it.
// The block tail mirrors block.py:267-268,291 backwards:  v = acc (started at g_t: d(0.2 conv5 + x)/dx) [+ A];
// [A' = v n2' -> out_a;  v = 0.2 A'];  t = v n1' -> x_out = the next block's g_t.

// lane's mask words of a slice (R x u16: row r = bits 16 (r & 1).. of word r >> 1): LDS read in asm (a compiler-visible
// LDS read next to DMAs in flight costs a vmcnt(0))
__device__ __forceinline__ u32x2 lds_mask(uint32_t addr) {
  u32x2 v = {0u, 0u};
  if constexpr (R == 4) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  else if constexpr (R == 2) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v[0]) : "v"(addr) : "memory");
  else asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v[0]) : "v"(addr) : "memory");
  return v;
}
// MFMA whose B operand was just written by VALU code (the packed gradient): the VALU-write -> MFMA-read wait states
// go inside the string (cdna guide 5.7 item 2)
template <bool AGPR>
__device__ __forceinline__ void mma_f16_after_valu(f32x16& acc, const u32x4& a, const u32x4& b) {
  if constexpr (AGPR) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

struct OneT { u32x4 a[4]; };      // A fragments of the transposed 1x1: [cout block 4/5][chunk 0/1]

// the 1x1 unit of the backward schedule (behind crit_3): wait for its four fragments, keep the stream going, read them
template <typename S>
__device__ __forceinline__ void load_1x1t(OneT& f, const WStream& s, char* smem, const Tile& t) {
  constexpr int I = S::first(U_ONE, S::ONE_P);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = t.lane();
  const uint32_t lw = lds0 + WOFF + ((s.ring + I) & (WR - 1)) * WSLOT + lane * 16;
  wait_units<typename S::Elem, S::wait_top(I, true), S::wait_top(I, false)>(s);
  __builtin_amdgcn_s_barrier();
  { const Ahead ah = ahead_of<S, I>(s, t, smem, (uint32_t)lane * 16u);
    sfor<4>([&](auto X) __attribute__((always_inline)) { issue_one<decltype(X)::value>(ah); }); }
  sfor<4>([&](auto F) __attribute__((always_inline)) { lds_read16<decltype(F)::value * 1024>(f.a[decltype(F)::value], lw); });
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]));
}

// epilogue of one finished 32-channel gradient slice (cout block BLK = 0..3 <-> g_x4, g_x3, g_x2, g_x1):
//   g_a = g_x * mask  -> `out` (dense slice out_cb, in full: the weight gradients read it) + LDS stage / registers (LW)
// WITH1X1 (g_x2): the unmasked slice -> `aux`, and g_x (blocks 4, 5) += W1x1^T g_x2.
template <typename T, int BLK, int LW, bool WITH1X1 = false, typename PT = esr_rdb_chain>
__device__ __forceinline__ void epilogue_bwd(Acc24& acc, const PT& p, const ImgView& out, int out_cb, const Tile& t,
                                             char* smem, int slot0, RowsRaw<T>* keep, int mask_buf, const ImgView* aux = nullptr,
                                             const OneT* one = nullptr) {
  using C16 = Ch16<T>;
  const int lane = t.lane(), tj = lane & 31, th = lane >> 5;
  int own_px = 0, own_swz = 0;
  if constexpr ((LW & 1) != 0) t.own(lane, own_px, own_swz);
  const int ox = t.ox0 + tj, oyb = t.oy0 + t.wave * R;
  const int wp32 = p.dense.wp * 32;
  const bool ragged = t.oy0 + TH > p.H || t.ox0 + TW > p.W;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const u32x2 mw = lds_mask(lds0 + LDS_MASK + mask_buf * MASK_SLICE + (t.wave * 64 + lane) * (2 * R));
  sfor<R>([&](auto RR) __attribute__((always_inline)) {
    constexpr int r = decltype(RR)::value;
    const int oy = oyb + r;
    const f32x16 a = acc_br<BLK, r>(acc);
    const bool inside = oy < p.H && ox < p.W;
    const int po = (oy + 1) * wp32 + (ox + 1) * 32;
    if constexpr (WITH1X1) {
      float raw[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) raw[e] = a[e];
      typename C16::Raw q;
      C16::pack(raw, q.q);
      if (ragged && !inside) { q.q[0] = u32x4{0, 0, 0, 0}; q.q[1] = u32x4{0, 0, 0, 0}; }
      C16::store_packed_rows(*aux, 0, th, inside ? po : (int)0x80000000u, q.q);
      mma_f16_after_valu<acc_in_agpr(4)>(acc_br<4, r>(acc), one->a[0], q.q[0]);
      mma_f16_after_valu<acc_in_agpr(4)>(acc_br<4, r>(acc), one->a[1], q.q[1]);
      mma_f16_after_valu<acc_in_agpr(5)>(acc_br<5, r>(acc), one->a[2], q.q[0]);
      mma_f16_after_valu<acc_in_agpr(5)>(acc_br<5, r>(acc), one->a[3], q.q[1]);
    }
    const uint32_t bits = (r < 2 ? mw[0] : mw[1]) >> ((r & 1) * 16);
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = a[e] * (((bits >> (15 - e)) & 1u) ? ESR_LRELU_SLOPE : 1.0f);   // lrelu'(a): 0.2 where a < 0 (block.py:12)
    typename C16::Raw q;
    C16::pack(v, q.q);
    C16::store_packed_rows(out, inside ? out_cb : 0, th, inside ? po : (int)0x80000000u, q.q);
    if (ragged && !inside) { q.q[0] = u32x4{0, 0, 0, 0}; q.q[1] = u32x4{0, 0, 0, 0}; }
    if constexpr (LW & 1) lds_put_row(smem, slot0 + th, r, q.q, own_px, own_swz);
    if constexpr (LW & 2) keep->q[r] = q;
  });
}

// block tail of the backward chain for g_x's cout block BLK (4 / 5; ch_cb = 0 / 1):
//   v = acc [+ r2];  [a = v (1 + sigma z2) -> out_a;  v = 0.2 a];  t = v (1 + sigma z1) -> out, LDS stage, next block's carry
template <typename T, int BLK, typename PT = esr_rdb_chain>
__device__ __forceinline__ void tail_bwd(Acc24& acc, const PT& p, const BlkS& blk, const ImgView& out, int ch_cb,
                                         const RowsRaw<T>* r2, bool has_res2, const ImgView& out_a, const bool has_out_a,
                                         const Tile& t, char* smem, int slot0) {
  using C16 = Ch16<T>;
  const int lane = t.lane(), tj = lane & 31, th = lane >> 5;
  int own_px = 0, own_swz = 0;
  t.own(lane, own_px, own_swz);
  const int ox = t.ox0 + tj, oyb = t.oy0 + t.wave * R;
  const int wp32 = p.dense.wp * 32;
  const bool ragged = t.oy0 + TH > p.H || t.ox0 + TW > p.W;
  const uint32_t layer1 = blk.layer1, layer2 = blk.layer2;
  const bool n1 = p.noise_mode == ESR_NOISE_PHILOX && layer1 != ESR_NO_LAYER;
  const bool n2 = p.noise_mode == ESR_NOISE_PHILOX && layer2 != ESR_NO_LAYER && has_out_a;
  uint64_t seed = p.seed;
  if ((n1 || n2) && p.seed_dev) seed = __builtin_nontemporal_load(p.seed_dev);
  sfor<R>([&](auto RR) __attribute__((always_inline)) {
    constexpr int r = decltype(RR)::value;
    const int oy = oyb + r;
    const f32x16 a = acc_br<BLK, r>(acc);
    float v[16], tmp[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = a[e];
    if (has_res2) {
      C16::get(r2->q[r], tmp);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += tmp[e];
    }
    const bool inside = oy < p.H && ox < p.W;
    const int po = (oy + 1) * wp32 + (ox + 1) * 32;
    const uint32_t pix = (uint32_t)((t.b * p.H + oy) * p.W + ox);
    if (has_out_a) {      // (a view by reference + a flag: a pointer selected at run time sends the descriptor through scratch)
      if (n2) {
#pragma unroll
        for (int o = 0; o < 2; ++o) philox_normal8(pix, (uint32_t)(ch_cb * 4 + th * 2 + o), layer2, seed, &tmp[8 * o]);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] + tmp[e] * (p.sigma * v[e]);
      }
      typename C16::Raw qa;
      C16::pack(v, qa.q);
      C16::store_packed_rows(out_a, inside ? ch_cb : 0, th, inside ? po : (int)0x80000000u, qa.q);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] *= 0.2f;
    }
    if (n1) {
#pragma unroll
      for (int o = 0; o < 2; ++o) philox_normal8(pix, (uint32_t)(ch_cb * 4 + th * 2 + o), layer1, seed, &tmp[8 * o]);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = v[e] + tmp[e] * (p.sigma * v[e]);
    }
    typename C16::Raw q;
    C16::pack(v, q.q);
    C16::store_packed_rows(out, inside ? ch_cb : 0, th, inside ? po : (int)0x80000000u, q.q);
    if (ragged && !inside) { q.q[0] = u32x4{0, 0, 0, 0}; q.q[1] = u32x4{0, 0, 0, 0}; }
    lds_put_row(smem, slot0 + th, r, q.q, own_px, own_swz);
    // the next block's g_x accumulators start at its g_t (= this output as stored): d(0.2 conv5 + x)/dx = 1
    float xs[16];
    C16::get(q, xs);
    f32x16 nx;
#pragma unroll
    for (int e = 0; e < 16; ++e) nx[e] = xs[e];
    acc_br<BLK, r>(acc) = nx;
  });
}

// DIR (esr_rdb_chain.mode): 0 = inference forward (fp16 / fp32), 1 = training forward (fp16: every slice, block output
// and the LeakyReLU masks reach memory), 2 = backward (fp16, esr_rdb_backward).  One instantiation per translation
// unit (rdb_fused.hip / rdb_fused_train.hip / rdb_fused_bwd.hip: they compile in parallel).
// host_abort: a pinned HOST word (may be null): set when a bounded spin timed out, so that the library can report
// the aborted launch at its next entry without synchronising (esr_rdb_check_abort).
// BAND (DIR 0): the row-band form for images with more tiles than CUs (esr_rdb_chain.band_rows).
// NZ: 1 = the launch has noise layers, 0 = it has none, -1 = decided at run time (DIR 2, whose tail handles both in one
// body).  One instantiation per form: with both block tails in one kernel hipcc parks a live accumulator tuple in
// scratch around every hand-off of the training forward (a vmcnt(0) per reload with weight DMAs in flight, +0.9 us
// per phase), and the inference kernel spilled 64 KB per tile and block on behalf of a noisy tail it never ran
// (2.5 GB of HBM writes per launch, profiles/r03_experiments.md).
template <typename T, int DIR = 0, bool BAND = false, int NZ = -1>
__global__ __launch_bounds__(NT, 1) void rdb_chain_kernel(const esr_rdb_chain p, const int ntiles, const int tiles_x,
                                                          const int tiles_y, unsigned* const host_abort) {
  using CF = Cfg<T>;
  static_assert(DIR == 0 || sizeof(T) == 2, "training forward / backward chains: fp16");
  static_assert(DIR == 0 || !BAND, "row bands: inference forward only");
  __shared__ __attribute__((aligned(16))) char smem[DIR == 2 ? LDS_BYTES_BWD : LDS_BYTES];
  unsigned* const ws = (unsigned*)p.workspace;
  unsigned* const flags = ws + WS_HDR;
  Tile t;
  t.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tpi = tiles_x * tiles_y;
  const int wp = p.dense.wp;
  if (threadIdx.x == 0) *(unsigned* volatile*)(smem + LDS_CTRL + 48) = host_abort;   // (visible behind the ticket barrier)
  // what the boundary code reads of the launch: the argument struct itself (inference: unchanged code) or a copy of
  // its scalars in registers (training / backward)
  using PT = std::conditional_t<BAND, PSB, std::conditional_t<DIR == 0, esr_rdb_chain, PS>>;
  PSB ps;
  if constexpr (DIR != 0 || BAND) {
    ps.H = __builtin_amdgcn_readfirstlane(p.H); ps.W = __builtin_amdgcn_readfirstlane(p.W);
    ps.noise_mode = __builtin_amdgcn_readfirstlane(p.noise_mode); ps.save_dense = BAND ? __builtin_amdgcn_readfirstlane(p.save_dense) : 0; ps._pad2 = 0;
    // the Philox key is resolved here, once (graph replay reads it from device memory)
    ps.sigma = p.sigma; ps.seed = p.seed_dev ? __builtin_nontemporal_load(p.seed_dev) : p.seed; ps.seed_dev = nullptr;
    ps.dense.wp = __builtin_amdgcn_readfirstlane(p.dense.wp);
    ps.trace = p.trace;
    ps.band_rows = __builtin_amdgcn_readfirstlane(p.band_rows); ps.band_margin = __builtin_amdgcn_readfirstlane(p.band_margin);
    ps.img_H = __builtin_amdgcn_readfirstlane(p.img_H);
  }
  const PT& q = *[&]() { if constexpr (DIR == 0 && !BAND) return &p; else return (const PT*)&ps; }();

  for (;;) {
    // ---- claim the next tile (tickets go out in order, so an image's tiles are co-resident)
    if (threadIdx.x < 64) ((volatile unsigned*)(smem + LDS_FLAGS))[threadIdx.x] = 0u;   // flags restart at 0 with the tile
    __syncthreads();
    if (threadIdx.x == 0)
      *(volatile int*)(smem + LDS_CTRL) = (int)__hip_atomic_fetch_add((gu32*)(ws + WS_TICKET), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int tile = __builtin_amdgcn_readfirstlane(*(volatile int*)(smem + LDS_CTRL));
    if (tile >= ntiles) break;
    t.b = tile / tpi;
    const int rem = tile - t.b * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    t.oy0 = ty * TH;
    t.ox0 = tx * TW;
    t.ty = ty; t.tx = tx; t.tiles_y = tiles_y; t.tiles_x = tiles_x;
    t.wp = wp;
    t.hlim = p.H + 1; t.wlim = p.W + 1;
    if constexpr (sizeof(T) == 2) {   // each thread's halo source offset, for the requests issued from inside the bulks
      int hsrc, hdst;
      t.halo(hsrc, hdst);
      *(volatile int*)(smem + LDS_HALO + t.tid() * 8) = hsrc;
      *(volatile int*)(smem + LDS_HALO + t.tid() * 8 + 4) = hdst;
      if (t.wave == 0) stage_nbr(tile, smem, t);
    }
    const ImgView dense = img_view(p.dense, t.b);
    const char* const dense_b = (const char*)p.dense.ptr + (int64_t)t.b * p.dense.batch_stride;
    const int64_t d_gs = p.dense.group_stride;


    unsigned epoch = 0;      // phases this tile has published
    WStream ws_{};
    Acc24 acc;               // never zeroed: an accumulator's first MFMA of a block takes SrcC = 0 — except conv5's
                             // (fp16 path), which carry 5 x in from the previous block's epilogue
    int ev = -1;
    if constexpr (sizeof(T) == 2) {
      // conv5's accumulators of the FIRST block start at 5 x (zero when the tail adds x explicitly, i.e. with
      // noise); later blocks get theirs from the previous epilogue.  Done ahead of the block loop: a second
      // definition inside it would join the carried one through scratch copies.
      const ImgView xin0 = img_view(p.blocks[0].x_in, t.b);
      const bool noisy = NZ != 0 && p.noise_mode != ESR_NOISE_OFF;
      RowsRaw<T> c0, c1;
      load_rows<T>(xin0, 0, q, t, c0); load_rows<T>(xin0, 1, q, t, c1);
      // (backward: g_x's accumulators start at g_t itself; training forward: always the folded form)
      const float cs = DIR == 2 ? 1.f : ((noisy && DIR == 0) ? 0.f : 5.f);
      sfor<R>([&](auto RR) __attribute__((always_inline)) {
        constexpr int r = decltype(RR)::value;
        float xs[16];
        f32x16 nx;
        Ch16<T>::get(c0.q[r], xs);
#pragma unroll
        for (int e = 0; e < 16; ++e) nx[e] = cs * xs[e];
        acc_br<4, r>(acc) = nx;
        Ch16<T>::get(c1.q[r], xs);
#pragma unroll
        for (int e = 0; e < 16; ++e) nx[e] = cs * xs[e];
        acc_br<5, r>(acc) = nx;
      });
      pin_acc45(acc);
    }
    for (int rb = 0; rb < p.n_blocks; ++rb) {
      const esr_rdb_block& blk = p.blocks[rb];
      ev = rb == 1 ? 0 : -1;
      ws_.dbg = ((ESR_ABL & 32) && rb == 1 && q.trace) ? q.trace + (int64_t)tile * 64 + 32 : nullptr;
      trace_ev(q, tile, ev);
      const char* const w = uniform_ptr(blk.w);
      const char* const xin_b = (const char*)blk.x_in.ptr + (int64_t)t.b * blk.x_in.batch_stride;
      const ImgView xin = img_view(blk.x_in, t.b), xout = img_view(blk.x_out, t.b);
      // block-table fields are wave-uniform, but only readfirstlane makes that provable: without it every
      // test on them becomes an exec-masked region and every use a fresh vector load
      const bool noisy = NZ != 0 && p.noise_mode != ESR_NOISE_OFF;
      BlkS bs;
      bs.bias = (const float*)uniform_ptr(blk.bias);
      bs.layer1 = __builtin_amdgcn_readfirstlane(blk.layer1);
      bs.layer2 = __builtin_amdgcn_readfirstlane(blk.layer2);
      bs.has_res2 = __builtin_amdgcn_readfirstlane((int)(blk.res2.ptr != nullptr)) != 0;
      bs.full_out = __builtin_amdgcn_readfirstlane((int)((blk.flags & ESR_RDB_FULL_OUT) != 0)) != 0 || noisy || p.save_dense;
      bs.band_own = BAND && __builtin_amdgcn_readfirstlane((int)((blk.flags & ESR_RDB_BAND_OWN) != 0)) != 0;
      const bool has_res2 = bs.has_res2, full_out = bs.full_out;
      const ImgView res2 = img_view(has_res2 ? blk.res2 : blk.x_in, t.b);
      constexpr bool RES = sizeof(T) == 2;      // fp16: LDS-resident slices (fp32 stages by DMA, 8 K steps of x)
      const char* const wnext = rb + 1 < p.n_blocks ? (const char*)p.blocks[rb + 1].w : nullptr;

      if constexpr (RES) {
if constexpr (DIR == 2) {
        // =========================== fp16 backward: the same phases over the gradient slices ===========================
        using S = Sched<T, true>;
        ws_.w = w;
        ws_.wnext = wnext;
        const ImgView dblk = img_view(blk.dense, t.b);                 // g_a4 | g_a3 | g_a2 | g_a1 of this block
        const ImgView aux = img_view(blk.aux, t.b);                    // unmasked g_x2 (32 channels)
        const bool has_out_a = __builtin_amdgcn_readfirstlane((int)(blk.out_a.ptr != nullptr)) != 0;
        const ImgView out_a = img_view(has_out_a ? blk.out_a : blk.x_out, t.b);
        const char* const mbase = uniform_ptr(blk.mask) + (int64_t)tile * MASK_TILE;
        // masks of slice s (0..3 = a1..a4 of the forward block; the backward consumes a4, a3, a2, a1) -> LDS buffer
        // `buf`: MASK_SLICE bytes, 16 per lane (2 KB by waves 0 and 1 for 4 rows per wave).  Landing: every consumer sits behind weight requests that were issued after
        // this one and have been waited for (in-order retirement) + a barrier.
        auto mask_dma = [&](int slice, int buf) __attribute__((always_inline)) {
          if (R >= 2 ? t.wave < MASK_SLICE / 1024 : (t.wave == 0 && t.lane() < MASK_SLICE / 16))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mbase + slice * MASK_SLICE + (t.wave * 64 + t.lane()) * 16),
                                             (__attribute__((address_space(3))) void*)(smem + LDS_MASK + buf * MASK_SLICE + t.wave * 1024), 16, 0, 0);
        };
        int early = 0;
        HaloRegs<CF::KD> hq;
        PollRegs pr;
        auto bulk_hook = [&](auto IDX, auto PH, auto FIRST_, auto END_, int g0, int next_slice, int next_buf) __attribute__((always_inline)) {
          constexpr int I = decltype(IDX)::value, rel = I - decltype(FIRST_)::value, left = decltype(END_)::value - 1 - I;
          constexpr int ph = decltype(PH)::value;
          const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
          if constexpr (rel == 0 && ph == PH_TOP) {
            if (threadIdx.x == 0) __hip_atomic_store((gu32*)(flags + tile), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (next_slice >= 0) mask_dma(next_slice, next_buf);       // (every wave is past the epilogue that read this buffer)
          }
          if constexpr (left == 4) {
            if constexpr (ph == PH_START) { if (t.wave == 0) lds_req32(pr.nbr, lds0 + LDS_NBR + t.lane() * 4); }
            if constexpr (ph == PH_TOP) { if (t.wave == 0) { lds_old(pr.nbr); poll_issue_at(ws, pr.nbr, smem); } }
          }
          if constexpr (left == 1) {
            if constexpr (ph == PH_START) { if (t.wave == 0) lds_req32(pr.flag, lds0 + LDS_FLAGS + t.lane() * 4); }
            if constexpr (ph == PH_END) { if (t.wave == 0) { lds_old(pr.flag); poll_tag(pr.flag, epoch, lds0, t); } }
            if constexpr (ph == PH_TOP) {
              lds_req32(pr.tag, lds0 + LDS_CTRL + 32);
              lds_req32(pr.hsrc, lds0 + LDS_HALO + t.tid() * 8);
            }
            if constexpr (ph == PH_IN) {
              lds_old2(pr.tag, pr.hsrc);
              early = (ESR_ABL & 1024) ? 1 : __builtin_amdgcn_readfirstlane((int)(pr.tag == epoch));
              if (early) halo_issue<CF::KD>(dblk, g0, (int)pr.hsrc, hq);
            }
          }
        };
        auto finish_halo = [&](int g0, int slot0) __attribute__((always_inline)) -> bool {
          const int hsrc = lds_word(smem, LDS_HALO + t.tid() * 8), hdst = lds_word(smem, LDS_HALO + t.tid() * 8 + 4);
          if (!early) {
            if (!wait_neighbours(ws, epoch, smem, t, &q, &ev, tile)) return false;
            halo_issue<CF::KD>(dblk, g0, hsrc, hq);
          } else {
            trace_ev(q, tile, ev);
          }
          halo_put<CF::KD>(smem, slot0, hsrc, hdst, hq);
          __syncthreads();
          return true;
        };
        mask_dma(3, 0);                      // a4's masks (first epilogue), a3's (second)
        mask_dma(2, 1);
        if (rb == 0) {
          sfor<AH>([&](auto UI) __attribute__((always_inline)) {
            const Ahead ah = ahead_of<S, decltype(UI)::value - AH>(ws_, t, smem, (uint32_t)t.lane() * 16u);
            sfor<4>([&](auto X) __attribute__((always_inline)) { issue_one<decltype(X)::value>(ah); });
          });
          sfor<CF::KX>([&](auto CI) __attribute__((always_inline)) {
            issue_a(xin_b + decltype(CI)::value * blk.x_in.group_stride, decltype(CI)::value, smem, t);
          });
          wait_vm<0>();
        } else {
          if (!wait_neighbours(ws, epoch, smem, t, &q, &ev, tile)) return;
          halo_fetch<CF::KX>(xin, 0, smem, t);
        }
        __syncthreads();
        trace_ev(q, tile, ev);
        // ---------------- g_x4 -> g_a4 (kept in registers: g_t still occupies the slots)
        seg_open(acc);
        run_units_s<S, S::first(U_CRIT, 1), S::end(U_CRIT, 1), false, false>(acc, ws_, smem, t);
        seg_close(acc);
        trace_ev(q, tile, ev);
        prewait_units<S, S::first(U_BULK, 1)>(ws_);     // bulk_1's first weights, in front of the epilogue's stores
        mfma_drain();
        RowsRaw<T> s1;
        epilogue_bwd<T, 0, 2>(acc, q, dblk, 0, t, smem, 0, &s1, 0);
        ++epoch;
        trace_ev(q, tile, ev);
        seg_open(acc);
        run_units_s<S, S::first(U_BULK, 1), S::end(U_BULK, 1), true, false, true>(acc, ws_, smem, t, [&](auto IDX, auto PH) __attribute__((always_inline)) { bulk_hook(IDX, PH, std::integral_constant<int, S::first(U_BULK, 1)>{}, std::integral_constant<int, S::end(U_BULK, 1)>{}, 0, 1, 0); });
        seg_close(acc);
        trace_ev(q, tile, ev);
        __builtin_amdgcn_s_barrier();          // every wave done reading g_t in slots 0, 1
        { const int lane = t.lane(); int opx, osw; t.own(lane, opx, osw);
#pragma unroll
          for (int r = 0; r < R; ++r) lds_put_row(smem, lane >> 5, r, s1.q[r].q, opx, osw); }
        trace_ev(q, tile, ev);
        if (!finish_halo(0, 0)) return;
        trace_ev(q, tile, ev);
        // ---------------- g_x3 -> g_a3
        seg_open(acc);
        run_units_s<S, S::first(U_CRIT, 2), S::end(U_CRIT, 2), false, false>(acc, ws_, smem, t);
        seg_close(acc);
        trace_ev(q, tile, ev);
        prewait_units<S, S::first(U_BULK, 2)>(ws_);     // bulk_2's first weights, in front of the epilogue's stores
        mfma_drain();
        epilogue_bwd<T, 1, 1>(acc, q, dblk, 1, t, smem, 2, nullptr, 1);
        ++epoch;
        trace_ev(q, tile, ev);
        seg_open(acc);
        run_units_s<S, S::first(U_BULK, 2), S::end(U_BULK, 2), true, false, true>(acc, ws_, smem, t, [&](auto IDX, auto PH) __attribute__((always_inline)) { bulk_hook(IDX, PH, std::integral_constant<int, S::first(U_BULK, 2)>{}, std::integral_constant<int, S::end(U_BULK, 2)>{}, CF::KD, 0, 1); });
        seg_close(acc);
        trace_ev(q, tile, ev);
        if (!finish_halo(CF::KD, 2)) return;
        trace_ev(q, tile, ev);
        // ---------------- g_x2: unmasked -> aux and, through the transposed 1x1, into g_x; masked -> g_a2
        seg_open(acc);
        run_units_s<S, S::first(U_CRIT, 3), S::end(U_CRIT, 3), false, false>(acc, ws_, smem, t);
        seg_close(acc);
        trace_ev(q, tile, ev);
        { OneT one;
          load_1x1t<S>(one, ws_, smem, t);
          prewait_units<S, S::first(U_BULK, 3)>(ws_);     // bulk_3's first weights, in front of the epilogue's stores
          mfma_drain();
          epilogue_bwd<T, 2, 1, true>(acc, q, dblk, 2, t, smem, 0, nullptr, 0, &aux, &one);
          seg_close(acc); }                 // (the epilogue's own MFMAs into g_x)
        ++epoch;
        trace_ev(q, tile, ev);
        seg_open<3>(acc);
        run_units_s<S, S::first(U_BULK, 3), S::end(U_BULK, 3), true, false, true>(acc, ws_, smem, t, [&](auto IDX, auto PH) __attribute__((always_inline)) { bulk_hook(IDX, PH, std::integral_constant<int, S::first(U_BULK, 3)>{}, std::integral_constant<int, S::end(U_BULK, 3)>{}, 2 * CF::KD, -1, 0); });
        seg_close<3>(acc);
        trace_ev(q, tile, ev);
        if (!finish_halo(2 * CF::KD, 0)) return;
        trace_ev(q, tile, ev);
        // ---------------- g_x1 -> g_a1
        seg_open<3>(acc);
        run_units_s<S, S::first(U_CRIT, 4), S::end(U_CRIT, 4), false, false>(acc, ws_, smem, t);
        seg_close<3>(acc);
        trace_ev(q, tile, ev);
        prewait_units<S, S::first(U_BULK, 4)>(ws_);     // bulk_4's first weights, in front of the epilogue's stores
        mfma_drain();
        epilogue_bwd<T, 3, 1>(acc, q, dblk, 3, t, smem, 2, nullptr, 1);
        ++epoch;
        trace_ev(q, tile, ev);
        seg_open<4>(acc);
        run_units_s<S, S::first(U_BULK, 4), S::end(U_BULK, 4), true, false, true>(acc, ws_, smem, t, [&](auto IDX, auto PH) __attribute__((always_inline)) { bulk_hook(IDX, PH, std::integral_constant<int, S::first(U_BULK, 4)>{}, std::integral_constant<int, S::end(U_BULK, 4)>{}, 3 * CF::KD, -1, 0); });
        seg_close<4>(acc);
        trace_ev(q, tile, ev);
        if (!finish_halo(3 * CF::KD, 2)) return;
        trace_ev(q, tile, ev);
        RowsRaw<T> tr0, tr1;
        // ---------------- g_x; block tail
        seg_open<4>(acc);
        run_units_s<S, S::first(U_CRIT, 5), S::end(U_CRIT, 5)>(acc, ws_, smem, t);
        seg_close<4>(acc);
        load_rows<T>(res2, 0, q, t, tr0, has_res2); load_rows<T>(res2, 1, q, t, tr1, has_res2);   // (see the forward)
        trace_ev(q, tile, ev);
        mfma_drain();
        tail_bwd<T, 4>(acc, q, bs, xout, 0, &tr0, has_res2, out_a, has_out_a, t, smem, 0);
        tail_bwd<T, 5>(acc, q, bs, xout, 1, &tr1, has_res2, out_a, has_out_a, t, smem, 2);
        pin_acc45(acc);
        publish(flags, tile, ++epoch, t, &q, &ev);
        ws_.ring = (ws_.ring + S::N) & (WR - 1);
        trace_ev(q, tile, ev);
        } else {
        // =========================== fp16: own pixels stay in the LDS ===========================
        using S = Sched<T>;
        constexpr bool TR = DIR == 1;
        ws_.w = w;
        ws_.wnext = wnext;
        // training forward: the block's own x1..x4 buffer and its tile record of LeakyReLU masks
        ImgView dblk = dense;
        if constexpr (TR) dblk = img_view(blk.dense, t.b);
        char* const mbase = TR ? uniform_ptr(blk.mask) + (int64_t)tile * MASK_TILE : nullptr;
        // The hand-off of x_p runs INSIDE bulk_p (hooks at the barriers that open the bulk's next units):
        //   unit 0: every wave has waited for its epilogue stores (strict wait) -> thread 0 raises the flag;
        //   5th unit from the end: wave 0 requests the 8 neighbours' flags;  3rd: it looks at them -> LDS word;
        //   2nd: all up -> every halo thread requests its 16 bytes per stage; they land under the last unit.
        // A neighbour that is late (flag not up yet at unit 3) sends the tile through the blocking poll after
        // the bulk instead.
        int early = 0;
        HaloRegs<CF::KD> hq;
        PollRegs pr;
        auto bulk_hook = [&](auto IDX, auto PH, auto FIRST_, auto END_, int g0) __attribute__((always_inline)) {
          constexpr int I = decltype(IDX)::value, rel = I - decltype(FIRST_)::value, left = decltype(END_)::value - 1 - I;
          constexpr int ph = decltype(PH)::value;
          const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
          if constexpr (rel == 0 && ph == PH_TOP) {
            if (threadIdx.x == 0) __hip_atomic_store((gu32*)(flags + tile), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if constexpr (left == 4 && !(ESR_ABL & 64)) {
            if constexpr (ph == PH_START) { if (t.wave == 0) lds_req32(pr.nbr, lds0 + LDS_NBR + t.lane() * 4); }
            if constexpr (ph == PH_TOP) { if (t.wave == 0) { lds_old(pr.nbr); poll_issue_at(ws, pr.nbr, smem); } }
          }
          if constexpr (left == 1 && !(ESR_ABL & 64)) {
            if constexpr (ph == PH_START) { if (t.wave == 0) lds_req32(pr.flag, lds0 + LDS_FLAGS + t.lane() * 4); }
            if constexpr (ph == PH_END) { if (t.wave == 0) { lds_old(pr.flag); poll_tag(pr.flag, epoch, lds0, t); } }
            if constexpr (ph == PH_TOP) {
              lds_req32(pr.tag, lds0 + LDS_CTRL + 32);
              lds_req32(pr.hsrc, lds0 + LDS_HALO + t.tid() * 8);
            }
            if constexpr (ph == PH_IN) {
              lds_old2(pr.tag, pr.hsrc);
              early = (ESR_ABL & 1024) ? 1 : __builtin_amdgcn_readfirstlane((int)(pr.tag == epoch));
              if (early) halo_issue<CF::KD>(dblk, g0, (int)pr.hsrc, hq);
            }
          }
        };
        // after the bulk: the halo of stage g0 into slots slot0..
        auto finish_halo = [&](int g0, int slot0) __attribute__((always_inline)) -> bool {
          const int hsrc = lds_word(smem, LDS_HALO + t.tid() * 8), hdst = lds_word(smem, LDS_HALO + t.tid() * 8 + 4);
          if (!early) {
            if (!wait_neighbours(ws, epoch, smem, t, &q, &ev, tile)) return false;
            halo_issue<CF::KD>(dblk, g0, hsrc, hq);
          } else {
            trace_ev(q, tile, ev);
          }
          halo_put<CF::KD>(smem, slot0, hsrc, hdst, hq);
          __syncthreads();
          return true;
        };
        stage_bias(bs.bias, smem, t);          // every wave is past the previous block's tail (publish)
        if (rb == 0) {
          // the chain's input comes from another launch: stage all of x (with halo) by DMA, and start
          // the weight stream (its first three units)
          sfor<AH>([&](auto UI) __attribute__((always_inline)) {
            const Ahead ah = ahead_of<S, decltype(UI)::value - AH>(ws_, t, smem, (uint32_t)t.lane() * 16u);
            sfor<4>([&](auto X) __attribute__((always_inline)) { issue_one<decltype(X)::value>(ah); });
          });
          sfor<CF::KX>([&](auto CI) __attribute__((always_inline)) {
            issue_a(xin_b + decltype(CI)::value * blk.x_in.group_stride, decltype(CI)::value, smem, t);
          });
          wait_vm<0>();
        } else {
          // own pixels were written by the previous block's epilogue; weights are in flight already
          if (!wait_neighbours(ws, epoch, smem, t, &q, &ev, tile)) return;
          halo_fetch<CF::KX>(xin, 0, smem, t);
        }
        __syncthreads();
        Bias16 bb;
        trace_ev(q, tile, ev);
        // ---------------- conv1
        seg_open(acc);
        run_units<T, S::first(U_CRIT, 1), S::end(U_CRIT, 1), false, false>(acc, ws_, smem, t);
        seg_close(acc);
        trace_ev(q, tile, ev);
        lds_bias(smem, 0, t, bb);
        prewait_units<S, S::first(U_BULK, 1)>(ws_);     // bulk_1's first weights, in front of the epilogue's stores
        mfma_drain();
        RowsRaw<T> x1, x2;
        epilogue<T, 0, 0, 2, true, TR>(acc, q, bs, bb, dblk, 0, 0, nullptr, nullptr, false, t, smem, 0, &x1, 0.f, true, mbase, 0);     // x1 (kept: x still occupies its slots)
        ++epoch;
        trace_ev(q, tile, ev);
        seg_open(acc);
        run_units<T, S::first(U_BULK, 1), S::end(U_BULK, 1), true, false, true>(acc, ws_, smem, t, [&](auto IDX, auto PH) __attribute__((always_inline)) { bulk_hook(IDX, PH, std::integral_constant<int, S::first(U_BULK, 1)>{}, std::integral_constant<int, S::end(U_BULK, 1)>{}, 0); });
        seg_close(acc);
        trace_ev(q, tile, ev);
        // ---------------- P = conv1x1(x) from the resident x; then x1 may take x's slots
        seg_open(acc);
        run_1x1_res<T>(acc, ws_, smem, t);
        seg_close(acc);
        __builtin_amdgcn_s_barrier();          // every wave done reading x
        { const int lane = t.lane(); int opx, osw; t.own(lane, opx, osw);
#pragma unroll
          for (int r = 0; r < R; ++r) lds_put_row(smem, lane >> 5, r, x1.q[r].q, opx, osw); }
        trace_ev(q, tile, ev);
        if (!finish_halo(0, 0)) return;
        trace_ev(q, tile, ev);
        // ---------------- conv2
        seg_open(acc);
        run_units<T, S::first(U_CRIT, 2), S::end(U_CRIT, 2), false, false>(acc, ws_, smem, t);
        seg_close(acc);
        trace_ev(q, tile, ev);
        lds_bias(smem, 32, t, bb);
        prewait_units<S, S::first(U_BULK, 2)>(ws_);     // bulk_2's first weights, in front of the epilogue's stores
        mfma_drain();
        epilogue<T, 1, 1, 1, true, TR>(acc, q, bs, bb, dblk, 1, 0, nullptr, nullptr, false, t, smem, 2, nullptr, 0.f, true, mbase, 1);       // x2
        ++epoch;
        trace_ev(q, tile, ev);
        seg_open(acc);
        run_units<T, S::first(U_BULK, 2), S::end(U_BULK, 2), true, false, true>(acc, ws_, smem, t, [&](auto IDX, auto PH) __attribute__((always_inline)) { bulk_hook(IDX, PH, std::integral_constant<int, S::first(U_BULK, 2)>{}, std::integral_constant<int, S::end(U_BULK, 2)>{}, CF::KD); });
        seg_close(acc);
        trace_ev(q, tile, ev);
        if (!finish_halo(CF::KD, 2)) return;
        trace_ev(q, tile, ev);
        // ---------------- conv3
        seg_open(acc);
        run_units<T, S::first(U_CRIT, 3), S::end(U_CRIT, 3), false, false>(acc, ws_, smem, t);
        seg_close(acc);
        trace_ev(q, tile, ev);
        lds_bias(smem, 64, t, bb);
        prewait_units<S, S::first(U_BULK, 3)>(ws_);     // bulk_3's first weights, in front of the epilogue's stores
        mfma_drain();
        epilogue<T, 2, 0, 1, true, TR>(acc, q, bs, bb, dblk, 2, 0, nullptr, nullptr, false, t, smem, 0, nullptr, 0.f, true, mbase, 2);       // x3
        ++epoch;
        trace_ev(q, tile, ev);
        seg_open<3>(acc);
        run_units<T, S::first(U_BULK, 3), S::end(U_BULK, 3), true, false, true>(acc, ws_, smem, t, [&](auto IDX, auto PH) __attribute__((always_inline)) { bulk_hook(IDX, PH, std::integral_constant<int, S::first(U_BULK, 3)>{}, std::integral_constant<int, S::end(U_BULK, 3)>{}, 2 * CF::KD); });
        seg_close<3>(acc);
        trace_ev(q, tile, ev);
        if (!finish_halo(2 * CF::KD, 0)) return;
        trace_ev(q, tile, ev);
        // ---------------- conv4
        seg_open<3>(acc);
        run_units<T, S::first(U_CRIT, 4), S::end(U_CRIT, 4), false, false>(acc, ws_, smem, t);
        seg_close<3>(acc);
        trace_ev(q, tile, ev);
        lds_bias(smem, 96, t, bb);
        lds_get_rows(smem, 2, x2.q, t);   // x2's own pixels still sit in the slots x4 is about to take
        prewait_units<S, S::first(U_BULK, 4)>(ws_);     // bulk_4's first weights, in front of the epilogue's stores
        mfma_drain();
        epilogue<T, 3, 2, 1, true, TR>(acc, q, bs, bb, dblk, 3, 0, &x2, nullptr, false, t, smem, 2, nullptr, 0.f, true, mbase, 3);           // x4 (+ x2)
        RowsRaw<T> tx0, tx1, tr0, tr1;
        ++epoch;
        trace_ev(q, tile, ev);
        seg_open<4>(acc);
        run_units<T, S::first(U_BULK, 4), S::end(U_BULK, 4), true, false, true>(acc, ws_, smem, t, [&](auto IDX, auto PH) __attribute__((always_inline)) { bulk_hook(IDX, PH, std::integral_constant<int, S::first(U_BULK, 4)>{}, std::integral_constant<int, S::end(U_BULK, 4)>{}, 3 * CF::KD); });
        seg_close<4>(acc);
        trace_ev(q, tile, ev);
        if (!finish_halo(3 * CF::KD, 2)) return;
        trace_ev(q, tile, ev);
        // ---------------- conv5; block tail (+ RRDB tail)
        seg_open<4>(acc);
        run_units<T, S::first(U_CRIT, 5), S::end(U_CRIT, 5)>(acc, ws_, smem, t);
        seg_close<4>(acc);
        // the block tail's residual (every third block).  Requested only now: ahead of conv5 hipcc has no registers for
        // the 16 rows and sends every one through scratch behind its own vmcnt(0)
        load_rows<T>(res2, 0, q, t, tr0, has_res2); load_rows<T>(res2, 1, q, t, tr1, has_res2);
        if constexpr (!TR) { if (noisy) { load_rows<T>(xin, 0, q, t, tx0); load_rows<T>(xin, 1, q, t, tx1); } }   // rare path: latency exposed
        trace_ev(q, tile, ev);
        mfma_drain();
        Bias16 bb2;
        lds_bias(smem, 128, t, bb);
        lds_bias(smem, 160, t, bb2);
        if constexpr (TR) {
          // training: the folded form with the noise layers (the carried 5 x makes `conv5 * 0.2 + x` one multiply);
          // every block output is kept for the backward
          if (NZ == 1 || (NZ < 0 && noisy)) {
            epilogue<T, 4, 3, 1, true, true>(acc, q, bs, bb, xout, 0, 0, nullptr, &tr0, has_res2, t, smem, 0, nullptr, 5.f, true);
            epilogue<T, 5, 3, 1, true, true>(acc, q, bs, bb2, xout, 1, 1, nullptr, &tr1, has_res2, t, smem, 2, nullptr, 5.f, true);
          } else {
            epilogue<T, 4, 3, 1, false, true>(acc, q, bs, bb, xout, 0, 0, nullptr, &tr0, has_res2, t, smem, 0, nullptr, 5.f, true);
            epilogue<T, 5, 3, 1, false, true>(acc, q, bs, bb2, xout, 1, 1, nullptr, &tr1, has_res2, t, smem, 2, nullptr, 5.f, true);
          }
        } else if (noisy) {
          epilogue<T, 4, 3, 1, true>(acc, q, bs, bb, xout, 0, 0, &tx0, &tr0, has_res2, t, smem, 0, nullptr, 0.f, full_out);
          epilogue<T, 5, 3, 1, true>(acc, q, bs, bb2, xout, 1, 1, &tx1, &tr1, has_res2, t, smem, 2, nullptr, 0.f, full_out);
        } else {
          epilogue<T, 4, 3, 1, false>(acc, q, bs, bb, xout, 0, 0, nullptr, &tr0, has_res2, t, smem, 0, nullptr, 5.f, full_out);
          epilogue<T, 5, 3, 1, false>(acc, q, bs, bb2, xout, 1, 1, nullptr, &tr1, has_res2, t, smem, 2, nullptr, 5.f, full_out);
        }
        pin_acc45(acc);
        publish(flags, tile, ++epoch, t, &q, &ev);
        ws_.ring = (ws_.ring + S::N) & (WR - 1);
        trace_ev(q, tile, ev);
        }
      } else {
      // =========================== fp32: every stage by DMA ===========================
      Bias16 fb[6];
      for (int i = 0; i < 5; ++i) load_bias(bs.bias + 32 * i, t, fb[i]);
      load_bias(bs.bias + 160, t, fb[5]);
      // ---------------- phase 1: x -> conv1..conv5
      issue_w_head<6>(w + CF::phase_off(1), CF::KX, smem, t);
      if (epoch > 0 && !wait_neighbours(ws, epoch, smem, t)) return;
      trace_ev(p, tile, ev);
      run_phase<T, 1>(acc, w + CF::phase_off(1), xin_b, blk.x_in.group_stride, CF::KX, smem, t);
      trace_ev(p, tile, ev);
      mfma_drain();
      epilogue<T, 0, 0>(acc, q, bs, fb[0], dense, 0, 0, nullptr, nullptr, false, t);       // x1
      publish(flags, tile, ++epoch, t);
      trace_ev(p, tile, ev);
      // ---------------- P = conv1x1(x) on own pixels, then phase 2: x1 -> conv2..conv5
      run_1x1<T>(acc, w + CF::phase_off(6), xin_b, blk.x_in.group_stride, smem, t);
      trace_ev(p, tile, ev);
      __syncthreads();                      // every wave done with the 1x1's LDS before phase 2 refills it
      issue_w_head<5>(w + CF::phase_off(2), CF::KD, smem, t);
      if (!wait_neighbours(ws, epoch, smem, t)) return;
      trace_ev(p, tile, ev);
      run_phase<T, 2>(acc, w + CF::phase_off(2), dense_b, d_gs, CF::KD, smem, t);
      trace_ev(p, tile, ev);
      mfma_drain();
      epilogue<T, 1, 1>(acc, q, bs, fb[1], dense, 1, 0, nullptr, nullptr, false, t);     // x2
      publish(flags, tile, ++epoch, t);
      trace_ev(p, tile, ev);
      // ---------------- phase 3: x2 -> conv3..conv5
      issue_w_head<4>(w + CF::phase_off(3), CF::KD, smem, t);
      if (!wait_neighbours(ws, epoch, smem, t)) return;
      trace_ev(p, tile, ev);
      run_phase<T, 3>(acc, w + CF::phase_off(3), dense_b + CF::KD * d_gs, d_gs, CF::KD, smem, t);
      trace_ev(p, tile, ev);
      mfma_drain();
      epilogue<T, 2, 0>(acc, q, bs, fb[2], dense, 2, 0, nullptr, nullptr, false, t);     // x3
      publish(flags, tile, ++epoch, t);
      trace_ev(p, tile, ev);
      // ---------------- phase 4: x3 -> conv4, conv5
      issue_w_head<3>(w + CF::phase_off(4), CF::KD, smem, t);
      if (!wait_neighbours(ws, epoch, smem, t)) return;
      trace_ev(p, tile, ev);
      run_phase<T, 4>(acc, w + CF::phase_off(4), dense_b + 2 * CF::KD * d_gs, d_gs, CF::KD, smem, t);
      trace_ev(p, tile, ev);
      mfma_drain();
      { RowsRaw<T> x2r; load_rows<T>(dense, 1, p, t, x2r);
        epilogue<T, 3, 2>(acc, q, bs, fb[3], dense, 3, 0, &x2r, nullptr, false, t); }     // x4 (+ x2)
      publish(flags, tile, ++epoch, t);
      trace_ev(p, tile, ev);
      // ---------------- phase 5: x4 -> conv5; block tail (+ RRDB tail)
      issue_w_head<2>(w + CF::phase_off(5), CF::KD, smem, t);
      if (!wait_neighbours(ws, epoch, smem, t)) return;
      trace_ev(p, tile, ev);
      run_phase<T, 5>(acc, w + CF::phase_off(5), dense_b + 3 * CF::KD * d_gs, d_gs, CF::KD, smem, t);
      trace_ev(p, tile, ev);
      mfma_drain();
      { RowsRaw<T> tx0, tx1, tr0, tr1;
        load_rows<T>(xin, 0, p, t, tx0); load_rows<T>(xin, 1, p, t, tx1);
        load_rows<T>(res2, 0, p, t, tr0); load_rows<T>(res2, 1, p, t, tr1);
        epilogue<T, 4, 3>(acc, q, bs, fb[4], xout, 0, 0, &tx0, &tr0, has_res2, t);
        epilogue<T, 5, 3>(acc, q, bs, fb[5], xout, 1, 1, &tx1, &tr1, has_res2, t); }
      publish(flags, tile, ++epoch, t);
      trace_ev(p, tile, ev);
      }
    }
  }
}

// ---- host-side helpers shared by the three translation units
int g_num_cus = 0;
std::once_flag g_cu_once;
int num_cus() {
  std::call_once(g_cu_once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
    if (g_num_cus <= 0) g_num_cus = 256;
  });
  return g_num_cus;
}

}  // namespace
