// rdb_wgrad.hip — weight / bias gradients of a whole ResidualDenseBlock_5C in one pass over its saved
// activations and gradients (autograd's conv backward-weight for block.py:239-268, triggered at
// SRRaGAN_model.py:140; replaces the six per-conv esr_conv_wgrad problems of a block).
//
//   dW_k[co][ci][kh][kw] += sum_{b,y,x} g_k[b][co][y][x] * in[b][ci][y+kh-1][x+kw-1]      k = conv1..conv5
//   dW_1x1[co][ci]       += sum g_x2[b][co][y][x] * x[b][ci][y][x]                         (block.py:263)
//   db_k[co]             += sum g_k[b][co][y][x]
// with in = [x | x1 | x2 | x3 | x4] (192 channels, conv_k reads the first 32(k+1)) and the gradients
// g = [g_t (64: conv5, x 0.2) | g_a4 | g_a3 | g_a2 | g_a1 | g_x2] (224 channels, the "Q" of the gather-form
// backward).  GEMM view per (cout block, cin block) PAIR: D[32 couts][32 cins] per tap, K = pixels.
//
// Why another kernel (wgrad.hip's fp16 kernel stays for every other conv).  Per conv, a workgroup of
// wgrad16_kernel stages a 4-row tile, waits for it, multiplies, and starts over: single-buffered, 72 MFMAs per
// ~2 us of exposed L2 -> LDS latency (MfmaUtil 0.24, profiles/r02), and the six convs of a block re-read the
// same concat prefix 3.2x over (700 MB per block at 16 x 128^2 against 217 MB of operands).  Here
//   * a workgroup = 8 waves = 8 PAIRS that share what is staged: the block's 26 3x3 pairs + 2 1x1 pairs are cut
//     into four SETS by input-channel range (conv5 x in[0:96) + the 1x1 | conv5 x in[96:192) | conv4 + conv1 |
//     conv3 + conv2), so a set stages only the input and gradient channel blocks it multiplies: 1.2x the operand
//     bytes instead of 3.2x;
//   * rows STREAM through a ring: a step = 4 output rows = in-rows 4t..4t+5; the ring holds five row PAIRS of
//     the set's input blocks and two quads of gradient rows; while step t multiplies, the LDS-DMA of the two
//     row pairs + gradient quad of step t+1 is in flight, waited for (vmcnt(0): it had a whole step to land)
//     right before the next batch is issued.  No halo re-reads between steps: every input row is staged once
//     per set;
//   * a B (input) fragment of in-row i feeds the three output rows i-2..i (kh = 2..0) against three resident A
//     (gradient) fragments: 88 transposed LDS reads per 72 MFMAs instead of 160;
//   * the two 16-channel groups of a 32-channel block are interleaved per pixel ([px][g0 32 B | g1 32 B]), so
//     the 32 lanes of a ds_read_b64_tr_b16 half cover 256 contiguous bytes: bank-conflict free, and the image is
//     still linear in DMA slot order (the permutation lives in the per-lane SOURCE address);
//   * a task walks several images of one column strip, so the per-task partial (the deterministic two-stage
//     reduction of wgrad.hip, same slot layout per conv) is written once per 4 images.
// Numerics: fp16 operands, fp32 MFMA accumulation over a task's pixels, fp32 sum over the tasks in a fixed order
// (bit-identical run to run).
#include <cstdlib>
#include "common.h"

namespace {

#ifndef ESR_WGRAD_NBF
#define ESR_WGRAD_NBF 3
#endif
#ifndef ESR_WGRAD_ABL
#define ESR_WGRAD_ABL 0      // timing ablations (development only; results are wrong when non-zero): 1 = no staging DMA after a
#endif                       // task's first batch, 2 = no fragment reads, 4 = no MFMAs, 8 = no step barrier / DMA wait
constexpr int NWV = 8, NTH = NWV * 64;
constexpr int RS = 4;                       // output rows per step
constexpr int INB = 34 * 64;                // one in-block row: [34 px][2 groups][32 B]
constexpr int GBB = 32 * 64;                // one g-block row:  [32 px][2 groups][32 B]
constexpr int MAX_NIB = 5, MAX_NGB = 3;
constexpr int NPAIR = 5;                    // ring of in-row pairs
constexpr int pair_bytes(int nib) { return ((2 * nib * INB + 1023) / 1024) * 1024; }
constexpr int gquad_bytes(int ngb) { return RS * ngb * GBB; }
constexpr int lds_need(int nib, int ngb) { return NPAIR * pair_bytes(nib) + 2 * gquad_bytes(ngb); }
constexpr int LDS_BYTES = lds_need(5, 2) > lds_need(3, 3) ? lds_need(5, 2) : lds_need(3, 3);

// conv ids: 0..4 = conv1..conv5, 5 = conv1x1;   g-block ids: 0,1 = g_t[0:32],[32:64] (conv5); 2 = g_a4; 3 = g_a3;
// 4 = g_a2; 5 = g_a1; 6 = g_x2 (1x1);   in-block ids: 0,1 = x[0:32],[32:64]; 2..5 = x1..x4
__device__ __host__ constexpr int conv_of_gblock(int gb) { return gb < 2 ? 4 : (gb == 6 ? 5 : 5 - gb); }   // 2->3 (conv4), 3->2, 4->1, 5->0
__device__ __host__ constexpr int conv_cin(int k) { return k == 5 ? 64 : 64 + 32 * k; }
__device__ __host__ constexpr int conv_cout(int k) { return k == 4 ? 64 : 32; }
__device__ __host__ constexpr int conv_ntap(int k) { return k == 5 ? 1 : 9; }
// slot layout (floats): per conv [ntap][cout][cin] then [cout] bias sums (none for the 1x1), conv1..conv5, 1x1
__device__ __host__ constexpr int conv_slot_elems(int k) { return conv_ntap(k) * conv_cout(k) * conv_cin(k) + (k == 5 ? 0 : conv_cout(k)); }
__device__ __host__ constexpr int conv_slot_off(int k) {
  int o = 0;
  for (int q = 0; q < k; ++q) o += conv_slot_elems(q);
  return o;
}
constexpr int SLOT_ELEMS = conv_slot_off(5) + conv_slot_elems(5);      // 241 664 weights + 192 bias sums
static_assert(SLOT_ELEMS == 241664 + 192, "slot");

// staged g-block index, staged in-block index; kind 0 = 3x3, 1 = 1x1, 2 = idle; bias: mask of the staged g-blocks whose
// bias gradient (the sum of the gradient over the pixels) this wave accumulates.  The sums ride on the waves with the
// least matrix work — the 1x1 pairs of set A (8 MFMAs per step against 72), the idle wave of sets C and D — not on the
// 3x3 pairs that happen to read the block: 128 dot-product instructions per step on two of eight waves kept the other six
// waiting at the step barrier (round 5).  Every conv's bias comes from exactly one set (conv5: A; conv4, conv1: C;
// conv3, conv2: D).
struct WaveJob { int g, i, kind, bias; };
struct SetDesc {
  int nib, ngb;
  int inb[MAX_NIB];                         // in-block ids staged
  int gbk[MAX_NGB];                         // g-block ids staged
  WaveJob wave[NWV];
};
constexpr SetDesc kSets[4] = {
    // A: conv5 x in blocks 0..2, + the 1x1 (g_x2 x in blocks 0,1)
    {3, 3, {0, 1, 2, 0, 0}, {0, 1, 6}, {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 2, 0, 0}, {1, 0, 0, 0}, {1, 1, 0, 0}, {1, 2, 0, 0}, {2, 0, 1, 1}, {2, 1, 1, 2}}},
    // B: conv5 x in blocks 3..5
    {3, 2, {3, 4, 5, 0, 0}, {0, 1, 0}, {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 2, 0, 0}, {1, 0, 0, 0}, {1, 1, 0, 0}, {1, 2, 0, 0}, {0, 0, 2, 0}, {0, 0, 2, 0}}},
    // C: conv4 x in blocks 0..4, conv1 x in blocks 0,1
    {5, 2, {0, 1, 2, 3, 4}, {2, 5, 0}, {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 2, 0, 0}, {0, 3, 0, 0}, {0, 4, 0, 0}, {1, 0, 0, 0}, {1, 1, 0, 0}, {0, 0, 2, 3}}},
    // D: conv3 x in blocks 0..3, conv2 x in blocks 0..2
    {4, 2, {0, 1, 2, 3, 0}, {3, 4, 0}, {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 2, 0, 0}, {0, 3, 0, 0}, {1, 0, 0, 0}, {1, 1, 0, 0}, {1, 2, 0, 0}, {0, 0, 2, 3}}},
};

// Fragment reads are inline asm: with a compiler-visible LDS read after an LDS-DMA in flight hipcc waits `vmcnt(0)`
// first (it cannot prove that the read does not alias the pending LDS write) — which would serialise the batch
// that is meant to land UNDER this step's MFMAs.  One 8-pixel MFMA fragment = two transposed reads (4 pixels each);
// OFF goes into the instructions' offset fields; completion is counted by hand (lgkmcnt, in-order) through wait
// statements that name the registers they release (cdna guide 5.7, form ii).
struct Frag { u32x2 lo, hi; };
template <int OFF> __device__ __forceinline__ void tr_issue(Frag& f, uint32_t a) {
  if constexpr ((ESR_WGRAD_ABL & 2) != 0) { asm volatile("" : "=v"(f.lo), "=v"(f.hi) : "v"(a)); return; }
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(f.lo), "=&v"(f.hi) : "v"(a), "n"(OFF), "n"(OFF + 4 * 64));
}
__device__ __forceinline__ half8 frag_val(const Frag& f) {
  const u32x4 r = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
  return __builtin_bit_cast(half8, r);
}
template <int N> __device__ __forceinline__ void wait_frag(Frag& f) {       // all but the newest N DS operations have returned
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.lo), "+v"(f.hi) : "n"(N));
}
template <int N> __device__ __forceinline__ void wait_frag5(Frag& a, Frag& b, Frag& c, Frag& d, Frag& e) {
  asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi),
               "+v"(e.lo), "+v"(e.hi) : "n"(N));
}

// descriptor inputs must be PROVABLY wave-uniform (cdna guide T20): through readfirstlane
__device__ __forceinline__ char* uptr(const void* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (char*)(((uint64_t)hi << 32) | lo);
}

struct Acc9 { f32x16 a0, a1, a2, a3, a4, a5, a6, a7, a8; };
template <int T> __device__ __forceinline__ f32x16& acc_t(Acc9& s) {
  if constexpr (T == 0) return s.a0; else if constexpr (T == 1) return s.a1; else if constexpr (T == 2) return s.a2;
  else if constexpr (T == 3) return s.a3; else if constexpr (T == 4) return s.a4; else if constexpr (T == 5) return s.a5;
  else if constexpr (T == 6) return s.a6; else if constexpr (T == 7) return s.a7; else return s.a8;
}
template <typename F, int... I> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

struct KArgs {
  esr_rdb_wgrad p;
  int strips, igroups, ipw;       // column strips per image, image groups, images per task
  int ntasks;                     // per block: 4 sets x strips x igroups
  int64_t slot_stride;            // floats per slot (SLOT_ELEMS rounded up to 4)
  int total;                      // tasks of the launch (n_blocks x ntasks): workgroup w runs w, w + grid, w + 2 grid, ...
  uint32_t* pace;                 // NULL, or one step counter per workgroup (zero at launch): see Pace
  // ---- follower form (esr_rdb_wgrad_run_follow): the pass runs NEXT TO the backward-chain launch that produces its
  // gradient slices, one block behind it
  const uint32_t* follow;         // NULL, or the chain's per-tile flags (its workspace behind the header words)
  int f_tx, f_ty;                 // the chain launch's tiles per image row / column (tile width = a column strip: 32)
  uint32_t* done;                 // follower: compute tasks finished per block (zero at launch); done[n_blocks] = "a wait timed
                                  // out": every later wait of the launch returns at once
  unsigned* host_abort;           // pinned host word raised when a bounded wait timed out (may be NULL)
  int rdelay;                     // follower: a block's reduce tasks stand this many blocks further down the task list
};

// ---- follower (round 6) -------------------------------------------------------------------------------------------
// Training crops leave the backward chain half of the chip (16 x 32^2 LR = 128 four-row tiles on 256 CUs).  Up to round
// 5 the chain ran as two launches so that the first run's weight gradients could execute under the second run — and the
// second run's were left behind the chain on the step's critical path (0.4 ms + a 0.08 ms reduction).  A follower pass
// is launched TOGETHER with ONE chain launch over all blocks, on the side stream, as a persistent grid on the CUs the
// chain leaves free: a task of block k (the chain's k-th block; the pass lists its blocks in the chain's order) starts
// when every tile it reads has published the end of block k — the chain's own hand-off flags, value 5 (k + 1), read
// with agent-scope loads; the gradient slices were written through to the L2 in front of the flag (rdb_chain_kernel.h:
// publish) and are staged here with sc1 loads.  The reduction of a block's task slots rides on the compute tasks one
// grid round later (rdb_wgrad_follow_kernel: arrival counter per block, write-through slot stores in front of it,
// L1-bypassing loads behind it — no fences) in the same slot order as the reduce kernel: results are bit-identical to
// the two-launch form, and what is left behind the chain is the last block's tasks.  Waits are bounded (a chain that aborted never raises its flags): a time-out raises the library's
// abort word and lets the pass run to its end on whatever is there — the next library call reports it.
__device__ __forceinline__ void follow_wait(const KArgs& ka, const int blk, const int chunk) {
  // wave 0: one flag per lane, 64 at a time
  const int sx = chunk % ka.strips, ig = chunk / ka.strips;
  const int b_begin = ig * ka.ipw, b_end = min(ka.p.B, b_begin + ka.ipw);
  const int nflag = (b_end - b_begin) * ka.f_ty;
  const uint32_t need = 5u * (uint32_t)(blk + 1);
  const int lane = threadIdx.x & 63;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  const uint32_t* const dead = ka.done + ka.p.n_blocks;
  for (int base = 0; base < nflag; base += 64) {
    const int i = base + lane;
    const bool mine = i < nflag;
    const int b = b_begin + (mine ? i / ka.f_ty : 0), ty = mine ? i % ka.f_ty : 0;
    const uint32_t* const f = ka.follow + ((int64_t)b * ka.f_ty + ty) * ka.f_tx + sx;
    bool ok = !mine;
    for (;;) {
      if (!ok) ok = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
      if (__all(ok)) break;
      const bool late = __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull;          // 2 s of the 100 MHz counter
      if (late || __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        if (late && lane == 0) {
          __hip_atomic_store(ka.done + ka.p.n_blocks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ka.host_abort) __hip_atomic_store(ka.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  }
}

// ---- lock step of the four sets of a chunk (round 5; an EXPERIMENT, off by default: ESR_RDB_WGRAD_PACE=1) -----------
// The four workgroups that run the four channel sets of one (block, image group, column strip) read the same rows of
// S and Q: set A, C, D all stage x and x1, B, C, D stage x2, A and B stage g_t ... 24 staged 32-channel blocks for 13
// distinct ones.  They sit on ONE XCD (workgroup ids 8 apart) and start together, but nothing keeps them together: PMC
// shows 22.9 GB fetched per launch at 16 x 128^2 for 15 GB of operands, L2 hit rate 0.30.  With the lock step the grid is
// persistent (every workgroup resident for the whole launch, striding over the tasks) and a workgroup starts row step s
// only when its three siblings have started step s - 1: the later readers of a row find it in the XCD's L2 (FETCH 16.4
// GB, hit rate 0.53) — and the launch takes 17 % LONGER (see esr_rdb_wgrad_run): the bytes were not the bound.
// Protocol, one lane of wave 0: publish the own step count (relaxed, agent scope), look at the siblings' counts that
// were LOADED DURING THE PREVIOUS STEP (those loads retire under the step's vmcnt(0): no extra latency on the common
// path), spin on fresh loads only when one is behind — bounded: a sibling that does not show up within ~60 us (another
// kernel holds its CU) switches the lock step off for the rest of this workgroup's launch.  Speed only: every
// workgroup computes the same sums whatever the others do.
struct Pace {
  uint32_t* mine;                 // this workgroup's counter (NULL: lock step off)
  const uint32_t* sib[3];
  uint32_t count;                 // steps this workgroup has started
  uint32_t seen[3];               // the siblings' counts as loaded during the previous step
};
__device__ __forceinline__ uint32_t pace_load(const uint32_t* q) {
  return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pace_step(Pace& pc) {
  // called by wave 0 only, in front of the step's barrier
  if (!pc.mine) return;
  const uint32_t c = ++pc.count;
  if (threadIdx.x == 0) __hip_atomic_store(pc.mine, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool behind = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) behind |= pc.seen[k] + 1u < c;
  if (__builtin_amdgcn_readfirstlane((int)behind)) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      uint32_t lo = 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < 3; ++k) { const uint32_t v = pace_load(pc.sib[k]); lo = v < lo ? v : lo; }
      if (lo + 1u >= c) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 6000ull) { pc.mine = nullptr; return; }     // 100 MHz: 60 us
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
__device__ __forceinline__ void pace_prefetch(Pace& pc) {
  // behind the barrier: the loads land under this step's MFMAs and are covered by the next step's vmcnt(0)
  if (!pc.mine) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) pc.seen[k] = pace_load(pc.sib[k]);
}

// one step of one wave: output rows 0..3 of the step against in-rows 0..5 (ring pairs p0, p1, p2).
// NIB_ROW / NGB_ROW: bytes of one in-row / g-row image of the set (compile time: every fragment address is
// base + immediate).  Per 16-pixel half run: 4 A (gradient) fragments stay resident; the 18 B (input) fragments
// (in-row ir, column tap kw) stream through three registers sets, two reads ahead of the MFMAs that consume them —
// B(ir, kw) feeds the output rows ir - kh, kh = 0..2.
template <int KIND, int NIB_ROW, int NGB_ROW>
__device__ __forceinline__ void step_mma(Acc9& acc, const uint32_t lg, const uint32_t lp0, const uint32_t lp1, const uint32_t lp2) {
  sfor<2>([&](auto HF) __attribute__((always_inline)) {
    constexpr int hf = decltype(HF)::value;
    Frag A[RS];
    sfor<RS>([&](auto RR) __attribute__((always_inline)) {
      constexpr int r = decltype(RR)::value;
      tr_issue<r * NGB_ROW + hf * (16 * 64)>(A[r], lg);
    });
    if constexpr (KIND == 1) {
      // 1x1: centre tap only — output row r against in-row r + 1, column + 1
      Frag B[RS];
      sfor<RS>([&](auto RR) __attribute__((always_inline)) {
        constexpr int r = decltype(RR)::value, ir = r + 1;
        tr_issue<(ir & 1) * NIB_ROW + (hf * 16 + 1) * 64>(B[r], ir < 2 ? lp0 : (ir < 4 ? lp1 : lp2));
      });
      wait_frag5<0>(A[0], A[1], A[2], A[3], B[0]);
      wait_frag5<0>(B[1], B[2], B[3], B[0], A[0]);
      sfor<RS>([&](auto RR) __attribute__((always_inline)) {
        constexpr int r = decltype(RR)::value;
        acc.a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_val(A[r]), frag_val(B[r]), acc.a0, 0, 0, 0);
      });
    } else {
      // NBF register sets, NBF - 1 fragments ahead (ESR_WGRAD_NBF: 3 = two ahead, the round-2 form)
      constexpr int NBF = ESR_WGRAD_NBF, AHD = NBF - 1;
      Frag B[NBF];
      auto issue_b = [&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value, ir = j / 3, kw = j % 3;
        tr_issue<(ir & 1) * NIB_ROW + hf * (16 * 64) + kw * 64>(B[j % NBF], ir < 2 ? lp0 : (ir < 4 ? lp1 : lp2));
      };
      sfor<AHD>([&](auto J) __attribute__((always_inline)) { issue_b(J); });
      wait_frag5<2 * (AHD - 1)>(A[0], A[1], A[2], A[3], B[0]);   // everything but the reads of B1 .. B(AHD - 1)
      sfor<18>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value, ir = j / 3, kw = j % 3;
        if constexpr (j + AHD < 18) issue_b(std::integral_constant<int, j + AHD>{});
        sfor<3>([&](auto KH) __attribute__((always_inline)) {
          constexpr int kh = decltype(KH)::value;
          constexpr int r = ir - kh;
          if constexpr (r >= 0 && r < RS) {
            f32x16& d = acc_t<kh * 3 + kw>(acc);
            if constexpr (!(ESR_WGRAD_ABL & 4)) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_val(A[r]), frag_val(B[j % NBF]), d, 0, 0, 0);
            else d[0] += (float)frag_val(A[r])[0] * (float)frag_val(B[j % NBF])[7];      // (keeps the reads alive)
          }
        });
        // B(j + 1): everything issued behind it may stay in flight (fragments j + 2 .. min(j + AHD, 17), two reads each)
        if constexpr (j + 1 < 18) wait_frag<2 * ((j + AHD < 17 ? j + AHD : 17) - (j + 1))>(B[(j + 1) % NBF]);
      });
    }
  });
}

// bias sums of one staged gradient block over a step: the lane's 8 pixels of cout row lane % 32, four row fragments per
// 16-pixel half run: v_dot2_f32_f16 against (1, 1) — exact products, fp32 accumulation, one instruction per two pixels.
// Order per lane: (half run, row, pixel pair) — the order the 3x3 pairs used to sum in, so the results are unchanged.
template <int NGB_ROW>
__device__ __forceinline__ void step_bias(float& bsum, const uint32_t lg) {
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  const half2_t ones = {(_Float16)1.0f, (_Float16)1.0f};
  Frag A[2][RS];
  sfor<2>([&](auto HF) __attribute__((always_inline)) {
    constexpr int hf = decltype(HF)::value;
    sfor<RS>([&](auto RR) __attribute__((always_inline)) {
      constexpr int r = decltype(RR)::value;
      tr_issue<r * NGB_ROW + hf * (16 * 64)>(A[hf][r], lg);
    });
  });
  wait_frag5<0>(A[0][0], A[0][1], A[0][2], A[0][3], A[1][0]);
  wait_frag5<0>(A[1][1], A[1][2], A[1][3], A[1][0], A[0][0]);
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const half8 hv = frag_val(A[hf][r]);
#pragma unroll
      for (int e = 0; e < 8; e += 2) bsum = __builtin_amdgcn_fdot2(half2_t{hv[e], hv[e + 1]}, ones, bsum, false);
    }
}

template <int SET, bool PACED, bool FOLLOW = false>
__device__ __forceinline__ void wgrad_set(const KArgs& ka, const int blk, const int chunk, char* const smem, Pace& pc) {
  constexpr SetDesc sd = kSets[SET];
  constexpr int nib = sd.nib, ngb = sd.ngb;
  constexpr int PAIRB = pair_bytes(nib), GQB = gquad_bytes(ngb);
  constexpr int ps_slots = PAIRB / 16, gq_slots = GQB / 16;
  constexpr int pair_used = 2 * nib * 34 * 4;                       // slots of a pair that hold data (the rest pads it to 1 KB)
  constexpr int NIB_ROW = nib * INB, NGB_ROW = ngb * GBB;
  const esr_rdb_wgrad& p = ka.p;
  const esr_rdb_wgrad_block& bd = p.blocks[blk];
  const int nchunk = ka.strips * ka.igroups;
  const int sx = chunk % ka.strips, ig = chunk / ka.strips;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = __builtin_amdgcn_readfirstlane(bd.in.wp);
  const int ox0 = sx * 32;

  // ---- per-thread DMA tables: a PAIR piece (two in-rows of the set's input blocks, the same table for every pair)
  // and a G QUAD piece (four gradient rows), each filled by rounds of 512 slots (slot s = tid + 512 k: 16 bytes;
  // a round of a wave = 1 KB of the piece's linear LDS image).  Entry = byte offset of the slot's 16 bytes inside
  // ONE IMAGE of its tensor, relative to the piece's first row; padding slots (the tail that rounds a pair up to
  // 1 KB) point past the image — the buffer range check returns zeros for them: no predication.
  constexpr int NRP = (ps_slots + NTH - 1) / NTH, NRG = (gq_slots + NTH - 1) / NTH;
  uint32_t entp[NRP], entg[NRG];
#pragma unroll
  for (int k = 0; k < NRP; ++k) {
    const int ls = tid + NTH * k;
    entp[k] = 0xFFFFFFF0u;
    if (ls < pair_used) {
      const int h = ls & 1, g = (ls >> 1) & 1, q = ls >> 2;
      const int px = q % 34, ri = q / 34, i = ri % nib, r2 = ri / nib;
      int bid = sd.inb[0];
#pragma unroll
      for (int u = 1; u < nib; ++u) if (i == u) bid = sd.inb[u];
      entp[k] = (uint32_t)((2 * bid + g) * (int)bd.in.group_stride + (r2 * wp + ox0 + px) * 32 + h * 16);
    }
  }
#pragma unroll
  for (int k = 0; k < NRG; ++k) {
    const int ls = tid + NTH * k;
    entg[k] = 0xFFFFFFF0u;
    if (ls < gq_slots) {
      const int h = ls & 1, g = (ls >> 1) & 1, q = ls >> 2;
      const int px = q & 31, rj = q >> 5, j = rj % ngb, r = rj / ngb;
      int gb = sd.gbk[0];
#pragma unroll
      for (int u = 1; u < ngb; ++u) if (j == u) gb = sd.gbk[u];
      entg[k] = (uint32_t)((2 * gb + g) * (int)bd.q.group_stride + ((r + 1) * wp + ox0 + px + 1) * 32 + h * 16);
    }
  }

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t lds_g = lds0 + NPAIR * PAIRB;
  // one image of each tensor as a buffer resource (wave-uniform), rebuilt per image
  __amdgpu_buffer_rsrc_t r_in, r_q;
  const int row_bytes = wp * 32;
  // pair `pi` (in-rows 2 pi, 2 pi + 1) -> ring slot pi % NPAIR; g quad t -> buffer t & 1.  One
  // `buffer_load_dwordx4 ... lds` per round: per-lane offset from the table, the row advance in the scalar offset.
  auto issue_pair = [&](int pi) __attribute__((always_inline)) {
    const int rowadv = 2 * pi * row_bytes;
    char* const dst0 = smem + (pi % NPAIR) * PAIRB + wave * 1024;
#pragma unroll
    for (int k = 0; k < NRP; ++k) {
      if ((k + 1) * NTH > ps_slots && wave * 64 + NTH * k >= ps_slots) continue;     // (last round: wave-uniform tail test)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (__attribute__((address_space(3))) void*)(dst0 + k * (NTH * 16)), 16,
                                               (int)entp[k], rowadv, 0, 0);
    }
  };
  auto issue_gquad = [&](int t) __attribute__((always_inline)) {
    const int rowadv = RS * t * row_bytes;
    char* const dst0 = smem + NPAIR * PAIRB + (t & 1) * GQB + wave * 1024;
#pragma unroll
    for (int k = 0; k < NRG; ++k) {
      if ((k + 1) * NTH > gq_slots && wave * 64 + NTH * k >= gq_slots) continue;
      // (follower: the slices were written by the chain launch that is still running — L1 bypassed, sc1)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_q, (__attribute__((address_space(3))) void*)(dst0 + k * (NTH * 16)), 16,
                                               (int)entg[k], rowadv, 0, FOLLOW ? 16 : 0);
    }
  };

  // ---- this wave's pair
  int jg = 0, ji = 0, kind = 2, bmask = 0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) if (wave == w) { jg = sd.wave[w].g; ji = sd.wave[w].i; kind = sd.wave[w].kind; bmask = sd.wave[w].bias; }
  int gbid = sd.gbk[0], ibid = sd.inb[0];
#pragma unroll
  for (int u = 1; u < ngb; ++u) if (jg == u) gbid = sd.gbk[u];
#pragma unroll
  for (int u = 1; u < nib; ++u) if (ji == u) ibid = sd.inb[u];
  const int conv = conv_of_gblock(gbid);
  const int cb = gbid == 1 ? 1 : 0;                     // cout block inside the conv (conv5's second half)
  // bias sums: staged g-blocks 0 / 1 of the set (a wave takes one or both), skipped where the caller wants no db
  const bool bias0 = (bmask & 1) && bd.db[conv_of_gblock(sd.gbk[0])] != nullptr;
  const bool bias1 = (bmask & 2) && bd.db[conv_of_gblock(sd.gbk[1])] != nullptr;
  // per-lane transposed-read geometry (wgrad.hip): lane -> pixel-in-run, 8-byte quarter, 16-channel half
  const int i16 = lane & 15, jrow = i16 >> 2, q = i16 & 3, ghalf = (lane >> 4) & 1, kg = lane >> 5;
  const uint32_t lane_off = (uint32_t)((8 * kg + jrow) * 64 + ghalf * 32 + q * 8);

  Acc9 acc;
  {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    acc.a0 = z; acc.a1 = z; acc.a2 = z; acc.a3 = z; acc.a4 = z; acc.a5 = z; acc.a6 = z; acc.a7 = z; acc.a8 = z;
  }
  float bsum0 = 0.f, bsum1 = 0.f;

  const int T = (p.H + RS - 1) / RS;
  const int b_begin = ig * ka.ipw, b_end = min(p.B, b_begin + ka.ipw);
  for (int b = b_begin; b < b_end; ++b) {
    r_in = __builtin_amdgcn_make_buffer_rsrc(uptr((char*)bd.in.ptr + (int64_t)b * bd.in.batch_stride), 0,
                                             __builtin_amdgcn_readfirstlane((int)bd.in.batch_stride), 0x00020000);
    r_q = __builtin_amdgcn_make_buffer_rsrc(uptr((char*)bd.q.ptr + (int64_t)b * bd.q.batch_stride), 0,
                                            __builtin_amdgcn_readfirstlane((int)bd.q.batch_stride), 0x00020000);
    __builtin_amdgcn_s_barrier();                  // every wave is done with the previous image's ring
    // batch 0 = pairs 0, 1, 2 + g quad 0
    issue_pair(0);
    issue_pair(1);
    issue_pair(2);
    issue_gquad(0);
    for (int t = 0; t < T; ++t) {
      if constexpr (!(ESR_WGRAD_ABL & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // batch t (issued a whole step ago) has landed
      if constexpr (PACED) { if (wave == 0) pace_step(pc); }   // lock step with the chunk's other sets (a late sibling: bounded spin)
      if constexpr (!(ESR_WGRAD_ABL & 8)) __builtin_amdgcn_s_barrier();                // ... for every wave; every wave is done reading step t - 1
      if constexpr (PACED) { if (wave == 0) pace_prefetch(pc); }
      if (t + 1 < T && !(ESR_WGRAD_ABL & 1)) {
        issue_pair(2 * t + 3);
        issue_pair(2 * t + 4);
        issue_gquad(t + 1);
      }
      if (bias0) step_bias<NGB_ROW>(bsum0, lds_g + (t & 1) * GQB + lane_off);
      if (bias1) step_bias<NGB_ROW>(bsum1, lds_g + (t & 1) * GQB + GBB + lane_off);
      if (kind == 2) continue;
      const uint32_t lg = lds_g + (t & 1) * GQB + jg * GBB + lane_off;
      const uint32_t lp0 = lds0 + ((2 * t) % NPAIR) * PAIRB + ji * INB + lane_off;
      const uint32_t lp1 = lds0 + ((2 * t + 1) % NPAIR) * PAIRB + ji * INB + lane_off;
      const uint32_t lp2 = lds0 + ((2 * t + 2) % NPAIR) * PAIRB + ji * INB + lane_off;
      if (kind == 0) step_mma<0, NIB_ROW, NGB_ROW>(acc, lg, lp0, lp1, lp2);
      else step_mma<1, NIB_ROW, NGB_ROW>(acc, lg, lp0, lp1, lp2);
    }
  }
  // a slot element.  Follower: read by ANOTHER workgroup of the same launch (possibly on another XCD) — a write-through
  // (sc1) store, as the chains publish their slices; no fence: an agent-scope release / acquire pair writes back and
  // invalidates the XCD's whole L2, i.e. the weight stream of the chain that runs next to this pass
  auto put_f = [](float* q, const float v) __attribute__((always_inline)) {
    if constexpr (FOLLOW) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *q = v;
  };
  // ---- bias sums -> behind the weights of their conv's slot ([9][cout][cin] | [cout])
  {
    float* const slot0 = p.partial + ((int64_t)blk * nchunk + chunk) * ka.slot_stride;
    auto put_bias = [&](const float bsum, const int gb) __attribute__((always_inline)) {
      const int cv = conv_of_gblock(gb), half = gb == 1 ? 1 : 0;
      const float scb = p.scale * (cv == 4 ? p.scale5 : 1.f);
      const float other = __shfl_xor(bsum, 32);          // lanes l and l + 32 hold the two k halves of cout row l % 32
      if (lane < 32) put_f(&slot0[conv_slot_off(cv) + (int64_t)9 * conv_cout(cv) * conv_cin(cv) + half * 32 + lane], (bsum + other) * scb);
    };
    if (bias0) put_bias(bsum0, sd.gbk[0]);
    if (bias1) put_bias(bsum1, sd.gbk[1]);
  }
  if (kind == 2) return;

  // ---- this task's slot of the partial arena: [block][chunk] x SLOT; per conv tap-major [tap][cout][cin] + bias sums
  float* const slot = p.partial + ((int64_t)blk * nchunk + chunk) * ka.slot_stride + conv_slot_off(conv);
  const int cin = conv_cin(conv), cout = conv_cout(conv);
  const int ci = ibid * 32 + (lane & 31);                 // 32-channel input block == cin block of the conv
  const float sc = p.scale * (conv == 4 ? p.scale5 : 1.f);
  auto put = [&](const f32x16& a, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
      put_f(&slot[((int64_t)t * cout + co) * cin + ci], a[e] * sc);
    }
  };
  if (kind == 1) {
    put(acc.a0, 0);
  } else {
    put(acc.a0, 0); put(acc.a1, 1); put(acc.a2, 2); put(acc.a3, 3); put(acc.a4, 4);
    put(acc.a5, 5); put(acc.a6, 6); put(acc.a7, 7); put(acc.a8, 8);
  }
}

// PACED: the lock-step experiment's instantiation.  The default one carries no Pace state at all: the struct is passed by
// reference through the set switch, lives in scratch, and a test of `pc.mine` in front of every step's barrier was a
// scratch load on wave 0 — a memory round trip per row step that every wave then waited for at the barrier.
template <bool PACED>
__global__ __launch_bounds__(NTH, 2) void rdb_wgrad_kernel(const KArgs ka) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  // ---- task decode: linear id -> (block, chunk, set) with a chunk's four sets on ONE XCD (ids 8 apart: the
  // dispatcher places workgroup b on XCD b % 8) and in lock step (Pace), so that what two sets both stage (conv5's two
  // halves share g_t; sets A, C, D share x) is an L2 hit for the later ones.  Speed only.
  // Persistent grid: workgroup w runs tasks w, w + grid, ... — with grid a multiple of 32 a workgroup keeps its XCD, its
  // set and its three siblings (w with bits 3, 4 changed) for the whole launch.
  const int per_block = ka.ntasks;
  const int nchunk = ka.strips * ka.igroups;
  const int full = (nchunk / 8) * 32;                    // tasks in complete groups of 8 chunks
  Pace pc;
  pc.mine = nullptr;
  pc.count = 0;
  pc.seen[0] = pc.seen[1] = pc.seen[2] = 0;
  if (PACED && ka.pace) {                                 // (host: only when every task sits in a complete group)
    const int w = blockIdx.x, me = (w >> 3) & 3;
    pc.mine = ka.pace + w;
    int k = 0;
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
      if (sidx != me) pc.sib[k++] = ka.pace + ((w & ~24) | (sidx << 3));
  }
  for (int task = blockIdx.x; task < ka.total; task += gridDim.x) {
    const int blk = task / per_block;
    const int l = task - blk * per_block;
    int chunk, set;
    if (l < full) { const int grp = l >> 5, w = l & 31; chunk = grp * 8 + (w & 7); set = w >> 3; }
    else { const int rem = nchunk & 7, w = l - full; chunk = (nchunk / 8) * 8 + w % rem; set = w / rem; }
    switch (set) {
      case 0: wgrad_set<0, PACED>(ka, blk, chunk, smem, pc); break;
      case 1: wgrad_set<1, PACED>(ka, blk, chunk, smem, pc); break;
      case 2: wgrad_set<2, PACED>(ka, blk, chunk, smem, pc); break;
      default: wgrad_set<3, PACED>(ka, blk, chunk, smem, pc); break;
    }
  }
}

// ---- stage 2: dw / db += sum over a block's task slots, in slot order (deterministic)
// element k of block blk's slot.  NT: the slots were written by other workgroups of the SAME launch (follower form):
// loads that bypass the L1
__device__ __forceinline__ float reduce_sum(const KArgs& ka, const int blk, const int k) {
  const int nchunk = ka.strips * ka.igroups;
  const float* src = ka.p.partial + (int64_t)blk * nchunk * ka.slot_stride + k;
  float s = 0.f;
  int sp = 0;
  for (; sp + 8 <= nchunk; sp += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sp + u) * ka.slot_stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sp < nchunk; ++sp) s += src[(int64_t)sp * ka.slot_stride];
  return s;
}
__device__ __forceinline__ void reduce_apply(const KArgs& ka, const int blk, const int k, const float s) {
  const esr_rdb_wgrad& p = ka.p;
  int conv = 0;
#pragma unroll
  for (int c = 1; c < 6; ++c) if (k >= conv_slot_off(c)) conv = c;
  const int e = k - conv_slot_off(conv);
  const esr_rdb_wgrad_block& bd = p.blocks[blk];
  const int ntap = conv_ntap(conv), cout = conv_cout(conv), cin = conv_cin(conv);
  const int nw = ntap * cout * cin;
  if (e >= nw) { if (bd.db[conv]) bd.db[conv][e - nw] += s; return; }
  float* const dw = bd.dw[conv];
  if (!dw) return;
  if (p.tap_major || ntap == 1) { dw[e] += s; return; }
  const int ci = e % cin, r = e / cin, co = r % cout, t = r / cout;
  dw[((int64_t)co * cin + ci) * ntap + t] += s;
}
// four consecutive elements (k a multiple of 4: a group never straddles a conv's weights / bias sums — every region of
// the slot is a multiple of 4 long): the same sums in the same slot order, 16-byte accesses.  Follower form only (the
// slots come from other workgroups of the same launch: `nt` loads go past the L1 like sc1 ones, profiles/r02_experiments)
__device__ __forceinline__ f32x4 reduce_sum4(const KArgs& ka, const int blk, const int k) {
  const int nchunk = ka.strips * ka.igroups;
  const f32x4* src = (const f32x4*)(ka.p.partial + (int64_t)blk * nchunk * ka.slot_stride + k);
  const int64_t st4 = ka.slot_stride / 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int sp = 0;
  for (; sp + 8 <= nchunk; sp += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (int64_t)(sp + u) * st4);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sp + 4 <= nchunk; sp += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + (int64_t)(sp + u) * st4);
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u];
  }
  for (; sp < nchunk; ++sp) s += __builtin_nontemporal_load(src + (int64_t)sp * st4);
  return s;
}
__device__ __forceinline__ void reduce_apply4(const KArgs& ka, const int blk, const int k, const f32x4 s) {
  const esr_rdb_wgrad& p = ka.p;
  int conv = 0;
#pragma unroll
  for (int c = 1; c < 6; ++c) if (k >= conv_slot_off(c)) conv = c;
  const int e = k - conv_slot_off(conv);
  const esr_rdb_wgrad_block& bd = p.blocks[blk];
  const int ntap = conv_ntap(conv), nw = ntap * conv_cout(conv) * conv_cin(conv);
  if (e >= nw) { if (bd.db[conv]) { f32x4* q = (f32x4*)(bd.db[conv] + (e - nw)); *q = *q + s; } return; }
  if (!bd.dw[conv]) return;
  if (p.tap_major || ntap == 1) { f32x4* q = (f32x4*)(bd.dw[conv] + e); *q = *q + s; return; }
#pragma unroll
  for (int j = 0; j < 4; ++j) reduce_apply(ka, blk, k + j, s[j]);
}
__global__ __launch_bounds__(256) void rdb_wgrad_reduce_kernel(const KArgs ka) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int blk = (int)(idx / SLOT_ELEMS);
  if (blk >= ka.p.n_blocks) return;
  const int k = (int)(idx - (int64_t)blk * SLOT_ELEMS);
  reduce_apply(ka, blk, k, reduce_sum(ka, blk, k));
}

// ---- follower form: persistent grid next to the chain launch (see follow_wait).  Task (vb, l) of the list = compute
// task l of block vb, then the l-th slice of the REDUCTION of block vb - rdelay (a whole round of the grid earlier: its
// compute tasks are done, or all but, when a workgroup gets there; arrival counter per block, write-through stores in
// front of it, L1-bypassing loads behind it) — the reduce kernel's arithmetic, slot order and all, so the results are
// bit-identical to the pass that runs behind the chain.  rdelay extra list positions at the end carry the last
// blocks' reductions.  (Tried first: the block's last task reduces the whole block — 472 dependent rounds of loads per
// thread, ~1 ms per block; reduce tasks of their own right behind the block's compute tasks — a third of the
// workgroups idle for a task's length, every block.)  Every workgroup walks the list in order and a compute task never
// waits for a reduction: no cycle of waits.
__global__ __launch_bounds__(NTH, 2) void rdb_wgrad_follow_kernel(const KArgs ka) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int per_block = ka.ntasks;
  const int nchunk = ka.strips * ka.igroups;
  const int full = (nchunk / 8) * 32;
  const int rspan = ((SLOT_ELEMS + per_block - 1) / per_block + 8 * NTH - 1) / (8 * NTH) * (8 * NTH);   // elements per reduction slice
  Pace pc;
  pc.mine = nullptr;
  pc.count = 0;
  pc.seen[0] = pc.seen[1] = pc.seen[2] = 0;
  for (int task = blockIdx.x; task < ka.total; task += gridDim.x) {
    const int vb = task / per_block;
    const int l = task - vb * per_block;
    if (vb < ka.p.n_blocks) {
      const int blk = vb;
      int chunk, set;
      if (l < full) { const int grp = l >> 5, w = l & 31; chunk = grp * 8 + (w & 7); set = w >> 3; }
      else { const int rem = nchunk & 7, w = l - full; chunk = (nchunk / 8) * 8 + w % rem; set = w / rem; }
      if (threadIdx.x < 64) follow_wait(ka, blk, chunk);
      __syncthreads();
      switch (set) {
        case 0: wgrad_set<0, false, true>(ka, blk, chunk, smem, pc); break;
        case 1: wgrad_set<1, false, true>(ka, blk, chunk, smem, pc); break;
        case 2: wgrad_set<2, false, true>(ka, blk, chunk, smem, pc); break;
        default: wgrad_set<3, false, true>(ka, blk, chunk, smem, pc); break;
      }
      // arrival: every storing wave has drained its write-through stores (the slot is complete at the coherent level)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(ka.done + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int rblk = vb - ka.rdelay;
    if (rblk >= 0) {
      // ---- slice l of block rblk's reduction
      if (threadIdx.x == 0) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(ka.done + rblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)per_block) {
          const bool late = __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull;      // 2 s: the chain never got there
          if (late || __hip_atomic_load(ka.done + ka.p.n_blocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            if (late) {
              __hip_atomic_store(ka.done + ka.p.n_blocks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (ka.host_abort) __hip_atomic_store(ka.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            break;
          }
          __builtin_amdgcn_s_sleep(32);
        }
      }
      __syncthreads();
      const int k0 = l * rspan, k1 = min(SLOT_ELEMS, k0 + rspan);
      for (int k = k0 + 4 * (int)threadIdx.x; k < k1; k += 8 * NTH) {
        // two groups of four elements per round: their loads are in flight together, the read-modify-writes follow
        const int kb = k + 4 * NTH;
        const f32x4 sa = reduce_sum4(ka, rblk, k);
        f32x4 sb = {0.f, 0.f, 0.f, 0.f};
        if (kb < k1) sb = reduce_sum4(ka, rblk, kb);
        reduce_apply4(ka, rblk, k, sa);
        if (kb < k1) reduce_apply4(ka, rblk, kb, sb);
      }
    }
  }
}

int images_per_task(int B, int strips, int n_blocks) {
  // as many images per task as still leave the chip ~4 tasks per CU (fewer slots to write and reduce); env override
  static const int forced = [] { const char* e = getenv("ESR_RDB_WGRAD_IPW"); return e ? atoi(e) : 0; }();
  if (forced > 0) return forced < B ? forced : B;
  int ipw = 1;
  while (ipw * 2 <= B && (int64_t)n_blocks * 4 * strips * ((B + 2 * ipw - 1) / (2 * ipw)) >= 1024 && ipw < 8) ipw *= 2;
  return ipw;
}

}  // namespace

constexpr int PACE_WORDS = 1024;            // step counters of the persistent grid (one per workgroup, <= CUs), behind the slots

static int64_t slot_floats(int32_t B, int32_t W, int32_t n_blocks) {
  const int strips = (W + 31) / 32;
  const int ipw = images_per_task(B, strips, n_blocks);
  const int igroups = (B + ipw - 1) / ipw;
  return (int64_t)n_blocks * strips * igroups * ((SLOT_ELEMS + 3) & ~3);
}

extern "C" int64_t esr_rdb_wgrad_workspace_elems(int32_t B, int32_t H, int32_t W, int32_t n_blocks) {
  if (B <= 0 || H <= 0 || W <= 0 || n_blocks <= 0) return 0;
  return slot_floats(B, W, n_blocks) + PACE_WORDS;
}

static int wgrad_cus() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
  if (!cached[dev]) {
    hipDeviceProp_t pr;
    cached[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  return cached[dev];
}

static int rdb_wgrad_launch(const esr_rdb_wgrad* p, esr_stream_t stream, const uint32_t* follow, int f_tx, int f_ty, unsigned* host_abort);

extern "C" int esr_rdb_wgrad_run(const esr_rdb_wgrad* p, esr_stream_t stream) { return rdb_wgrad_launch(p, stream, nullptr, 0, 0, nullptr); }

// Follower form (api.hip: an ESR_OP_RDB_WGRAD op flagged ESR_OPF_FOLLOW right behind the ESR_OP_RDB_CHAIN_BWD op whose
// blocks it lists, in the chain's order): `flags` = the chain launch's per-tile flags (cleared, on a stream this one
// waits for), tiles_x / tiles_y = that launch's tile grid.  p->max_workgroups = the CUs the chain leaves free.
int esr_rdb_wgrad_run_follow(const esr_rdb_wgrad* p, esr_stream_t stream, const uint32_t* flags, int tiles_x, int tiles_y, unsigned* host_abort) {
  if (!flags || tiles_x <= 0 || tiles_y <= 0 || !p || p->max_workgroups <= 0 || tiles_x != (p->W + 31) / 32) {
    esr_set_error("esr_rdb_wgrad_run_follow: invalid arguments (needs the chain's flags, its tile grid and max_workgroups > 0)");
    return ESR_ERR_INVALID;
  }
  return rdb_wgrad_launch(p, stream, flags, tiles_x, tiles_y, host_abort);
}

static int rdb_wgrad_launch(const esr_rdb_wgrad* p, esr_stream_t stream, const uint32_t* follow, int f_tx, int f_ty, unsigned* host_abort) {
  if (!p || !p->blocks || p->n_blocks <= 0 || p->B <= 0 || p->H <= 0 || p->W <= 0 || !p->partial) {
    esr_set_error("esr_rdb_wgrad_run: invalid arguments");
    return ESR_ERR_INVALID;
  }
  if (p->dtype != ESR_F16) { esr_set_error("esr_rdb_wgrad_run: fp16 only (the fp32 parity path uses esr_conv_wgrad)"); return ESR_ERR_UNSUPPORTED; }
  KArgs ka;
  ka.p = *p;
  ka.strips = (p->W + 31) / 32;
  ka.ipw = images_per_task(p->B, ka.strips, p->n_blocks);
  ka.igroups = (p->B + ka.ipw - 1) / ka.ipw;
  ka.ntasks = 4 * ka.strips * ka.igroups;
  ka.slot_stride = (SLOT_ELEMS + 3) & ~3;
  if (p->partial_elems < esr_rdb_wgrad_workspace_elems(p->B, p->H, p->W, p->n_blocks)) {
    esr_set_error("esr_rdb_wgrad_run: partial arena too small (%lld floats, need %lld: esr_rdb_wgrad_workspace_elems)",
                  (long long)p->partial_elems, (long long)esr_rdb_wgrad_workspace_elems(p->B, p->H, p->W, p->n_blocks));
    return ESR_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  // Grid.  Default: one workgroup per task (the hardware deals tasks to CUs as they free up).  Persistent forms — every
  // workgroup strides over the tasks — when the caller caps the workgroups (p->max_workgroups: the train plan runs a pass
  // NEXT TO the following backward-chain launch and leaves that launch its CUs), with ESR_RDB_WGRAD_GRID=n, and for the
  // lock step of a chunk's four sets (ESR_RDB_WGRAD_PACE=1, see Pace).  Measured at 69 blocks x 16 x 128^2 (round 5,
  // tools/pmc_wgrad.sh): one workgroup per task 9.32 ms, FETCH 22.9 GB, L2 hit 0.30; persistent 9.38 ms, 25.9 GB, 0.20;
  // persistent + lock step 11.05 ms, FETCH 16.4 GB, L2 hit 0.53 — the lock step removes the re-reads and the kernel gets
  // SLOWER: it is bound by its L2 -> LDS staging and the step barrier, not by HBM bytes, and in lock step every chunk
  // runs at the pace of its slowest set (C stages 7 channel blocks per row step, B 5).  Hence off by default.
  ka.total = p->n_blocks * ka.ntasks;
  static const int env_grid = [] { const char* e = getenv("ESR_RDB_WGRAD_GRID"); return e ? atoi(e) : 0; }();
  static const bool pace_on = [] { const char* e = getenv("ESR_RDB_WGRAD_PACE"); return e && atoi(e) != 0; }();
  int grid = ka.total;
  if (p->max_workgroups > 0 || env_grid > 0 || pace_on) {
    grid = wgrad_cus();
    if (p->max_workgroups > 0 && p->max_workgroups < grid) grid = p->max_workgroups;
    if (env_grid > 0) grid = env_grid;
    if (follow) { if (grid >= 8) grid &= ~7; }   // (a follower's tasks are bound to the chain's progress, not to each other: whole XCD rounds)
    else if (grid >= 32) grid &= ~31;            // a multiple of 32 keeps every workgroup on its XCD and with its set
    if (grid > ka.total) grid = ka.total;
  }
  const int nchunk = ka.strips * ka.igroups;
  ka.pace = nullptr;
  ka.follow = follow; ka.f_tx = f_tx; ka.f_ty = f_ty; ka.done = nullptr; ka.host_abort = host_abort; ka.rdelay = 0;
  if (follow) {
    if (p->n_blocks >= PACE_WORDS) { esr_set_error("esr_rdb_wgrad_run_follow: at most %d blocks per pass", PACE_WORDS); return ESR_ERR_UNSUPPORTED; }
    ka.done = (uint32_t*)(p->partial + slot_floats(p->B, p->W, p->n_blocks));
    if (hipMemsetAsync(ka.done, 0, (size_t)(p->n_blocks + 1) * sizeof(uint32_t), st) != hipSuccess) {
      esr_set_error("esr_rdb_wgrad_run_follow: hipMemsetAsync failed");
      return ESR_ERR_LAUNCH;
    }
    ka.rdelay = (grid + ka.ntasks - 1) / ka.ntasks;
    ka.total = (p->n_blocks + ka.rdelay) * ka.ntasks;
    hipLaunchKernelGGL(rdb_wgrad_follow_kernel, dim3((unsigned)grid), dim3(NTH), 0, st, ka);
    return esr_check_launch("rdb_wgrad_follow_kernel");
  }
  if (pace_on && nchunk % 8 == 0 && grid % 32 == 0 && grid <= PACE_WORDS) {
    ka.pace = (uint32_t*)(p->partial + slot_floats(p->B, p->W, p->n_blocks));
    if (hipMemsetAsync(ka.pace, 0, (size_t)grid * sizeof(uint32_t), st) != hipSuccess) {
      esr_set_error("esr_rdb_wgrad_run: hipMemsetAsync failed");
      return ESR_ERR_LAUNCH;
    }
  }
  if (ka.pace) hipLaunchKernelGGL(rdb_wgrad_kernel<true>, dim3((unsigned)grid), dim3(NTH), 0, st, ka);
  else hipLaunchKernelGGL(rdb_wgrad_kernel<false>, dim3((unsigned)grid), dim3(NTH), 0, st, ka);
  int rc = esr_check_launch("rdb_wgrad_kernel");
  if (rc) return rc;
  const int64_t total = (int64_t)p->n_blocks * SLOT_ELEMS;
  hipLaunchKernelGGL(rdb_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ka);
  return esr_check_launch("rdb_wgrad_reduce_kernel");
}
