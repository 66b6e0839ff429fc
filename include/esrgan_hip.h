/* esrgan_hip.h — C ABI of libesrgan_hip.so (MI355X / gfx950 only).
 *
 * The reference (ncarraz/ESRGANplus) has no FFI layer: its hot path is expressed as stock
 * torch.nn modules that dispatch to cuDNN/cuBLAS.  This header is the boundary a maintainer binds
 * instead (INTEGRATION.md shows the ctypes stub): every entry point replaces the library call made
 * by the cited reference line.  Plain C types only; all buffers are caller-owned device memory;
 * every launch goes to the caller's hipStream_t; the library never allocates, never synchronises
 * and never changes the current device (graph-capture safe).  Every entry returns 0 on success
 * or a negative esr_status; esr_last_error() gives a per-thread message.
 *
 * ---- activation layout "G32" -----------------------------------------------------------------
 * A tensor of C channels is stored as ngroups = ceil(C / cpg) planes of 32-byte channel groups
 * (cpg = 16 for fp16, 8 for fp32):   [B][ngroups][Hp][Wp][32 bytes]
 * Logical pixel (y,x) lives at padded coordinates (y+1, x+1); row 0 / column 0 and everything at
 * or beyond (H+1, W+1) is ZERO and is never written by any kernel (this is the conv zero padding
 * of block.py:55-58,137 — the halo is physical, so kernels need no bounds checks on loads).
 * Hp >= roundup(H,32)+6, Wp >= roundup(W,32)+2 (see esr_g32_dims).
 */
#ifndef ESRGAN_HIP_H
#define ESRGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* esr_stream_t; /* hipStream_t */

enum esr_status {
  ESR_OK = 0,
  ESR_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
  ESR_ERR_LAUNCH = -2,      /* HIP launch error (message in esr_last_error) */
  ESR_ERR_UNSUPPORTED = -3
};

enum esr_dtype { ESR_F16 = 0, ESR_F32 = 1 };
enum esr_act { ESR_ACT_NONE = 0, ESR_ACT_LRELU = 1 /* slope 0.2, block.py:12 */, ESR_ACT_RELU = 2 };
enum esr_noise { ESR_NOISE_OFF = 0, ESR_NOISE_PHILOX = 1, ESR_NOISE_EXPLICIT = 2 };

/* One G32 tensor view (pointer already offset to the first channel group of interest). */
typedef struct esr_g32 {
  void* ptr;
  int64_t batch_stride; /* bytes */
  int64_t group_stride; /* bytes */
  int32_t wp;           /* row pitch in pixels (32-byte units) */
  int32_t ngroups;      /* groups addressable through this view */
} esr_g32;

/* Fused convolution.  Replaces nn.Conv2d + LeakyReLU/ReLU + the elementwise tail the reference
 * runs as separate kernels (conv_block block.py:125-151; RDB tail block.py:263,266,268; RRDB
 * tail block.py:291 / test_image/block.py:256; ShortcutBlock block.py:84-86; nearest upsample
 * block.py:315-322 folded into the load).  Epilogue, per output element, in this order:
 *     v = acc + bias;  v = act(v);  [aux_out = v];  v += acc_1x1;
 *     v = v*alpha + res1;  v *= (1 + sigma*z1);  v = v*beta + res2;  v *= (1 + sigma*z2);  out = v
 * (each step skipped when its operand is absent; a residual / z view with fewer channel groups
 * than the output contributes zero to the channels it does not cover).  Backward chains add
 *     out2 = v * act'(mask)      for cout blocks >= mask_cb_begin (act' = 1 if mask>0 else 0.2 / 0)
 *     out3 = v * gamma * (1 + sigma*z3)
 * which is how dgrad launches also apply the LeakyReLU / GaussianNoise / residual-scale backward
 * of block.py:262-268,291 without extra elementwise passes.                                    */
typedef struct esr_conv {
  int32_t dtype;       /* esr_dtype: storage type of every G32 tensor here (accumulate fp32) */
  int32_t ks;          /* 1, 3 or 4 (2: upsample == 3) */
  int32_t stride;      /* 1, or 2 (ks==4) */
  int32_t upsample;    /* 1: input is nearest-x2 upsampled on load (ks==3 only);
                          2: TRANSPOSED 4x4/stride-2 conv (adjoint of ks==4,stride==2): the input is at
                             half the output resolution; weights packed with transpose_flip=2;
                          3: the same function as 1 in its 4-phase 2x2 ("sub-pixel") form, ks == 2, weights
                             packed with esr_pack.ups_fwd (2.25x fewer MACs; results differ from 1 only by the
                             rounding of the pre-summed taps) */
  int32_t B, H, W;     /* OUTPUT logical size */
  int32_t cin_groups;  /* K loop length: input channel groups read from `in` */
  int32_t cout_blocks; /* ceil(Cout/32) */
  esr_g32 in;
  esr_g32 out;         /* out.ngroups bounds the groups actually stored */
  const void* w;       /* packed by esr_pack_conv_weights */
  const float* bias;   /* fp32 [cout_blocks*32] or NULL */
  const void* w1x1;    /* packed 1x1 (bias-free, block.py:153-154,263) or NULL */
  int32_t n1x1_groups; /* input groups the 1x1 reads (prefix of `in`) */
  int32_t act;         /* esr_act */
  esr_g32 aux_out;     /* ptr NULL = off */
  float alpha; esr_g32 res1;
  float beta;  esr_g32 res2;
  int32_t noise_mode;  /* esr_noise */
  float sigma;         /* 0.1, block.py:111 */
  uint64_t seed;       /* philox key */
  uint32_t layer1, layer2; /* philox stream ids of z1 / z2; layer==0xFFFFFFFF = that noise off */
  esr_g32 z1, z2;      /* explicit z (G32, same dtype); ptr NULL = off */
  esr_g32 mask;        /* dgrad: saved activation whose sign gives act' (ptr NULL = off) */
  esr_g32 out2;        /* dgrad: masked output */
  int32_t nchw_out_c;  /* >0: ALSO store the first nchw_out_c channels as fp32 NCHW */
  float* nchw_out;     /* [B][nchw_out_c][H][W] */
  int32_t debug_flags; /* measurement only: 1 = skip epilogue, 2 = skip MFMAs, 4 = skip activation DMA;
                          test hooks (results valid): 64 = force the plain K loop, 128 = force the hand-pipelined
                          K loop of the 32-cout 3x3 conv (default: pipelined up to 256 tiles), 256 = keep 16-row
                          tiles (default: fp16 3x3/s1 convs whose 16-row grid x cout blocks is <= 128 / <= 384
                          workgroups, or whose map has <= 4 / <= 8 rows, run on 4- / 8-row tiles — same results) */
  int32_t mask_cb_begin; /* mask/out2 apply to cout blocks >= this one, indexed from it */
  float gamma;          /* third stage (backward chains): out3 = v * gamma * (1 + sigma*z3) */
  uint32_t layer3;      /* philox stream id of z3 (0xFFFFFFFF = no noise on out3) */
  esr_g32 z3;           /* explicit z3 */
  esr_g32 out3;         /* ptr NULL = off */
  int32_t mask_act;     /* esr_act whose derivative the mask selects (LRELU: 1 / 0.2, RELU: 1 / 0) */
  int32_t _pad2;
  const uint64_t* seed_dev; /* non-NULL: the Philox seed is read from DEVICE memory at run time (so a
                               captured graph can be replayed with a fresh seed); overrides `seed` */
  /* ---- round 4: the discriminators' deep 4x4/s2 convs (architecture.py:87-129: features.14 / .20 / .26 — fp16, square
   * maps of 4 / 8 / 16 output columns, plain layer: bias [+ act], one G32 output).  ksplit > 1: several images share a
   * tile and the K loop is split over ksplit workgroups per tile; each writes an fp32 slab of split_ws
   * ([ksplit][B][H][W][cout_blocks * 32] floats) and a finishing launch of the same call adds the slabs up in split
   * order (deterministic), applies bias / act and stores `out`.  stat_sums (optional): that pass also accumulates the
   * BatchNorm statistics of the stored output exactly as ESR_BN_STATS would (sums[grp][c] += x, sums[grp][C + c] +=
   * x^2, groups = stat_groups, C = stat_C; the caller zeroes stat_sums), so the BatchNorm that follows starts at
   * ESR_BN_FIN_APPLY.  ksplit <= 1: off. */
  int32_t ksplit, stat_groups, stat_C, _pad3;
  float* split_ws;
  double* stat_sums;
} esr_conv;

/* Weight packing: OIHW fp32 master (the nn.Parameter the reference keeps, e.g. state-dict key
 * model.1.sub.0.RDB1.conv1.0.weight) -> MFMA A-fragment order
 *   [cout_block][cin_group][kh][kw][lane 0..63][16 bytes].
 * transpose_flip=1 packs the dgrad operand (Cin<->Cout swapped, taps rotated 180 degrees). */
typedef struct esr_pack {
  const float* src;    /* OIHW fp32 */
  void* dst;
  int32_t cout, cin, ks;
  int32_t dtype;
  int32_t transpose_flip; /* 1: dgrad operand (Cin<->Cout, taps rotated 180 deg); 2: Cin<->Cout only
                             (operand of the transposed stride-2 conv, upsample==2);
                             `cout`/`cin` below are still the FORWARD conv's */
  int32_t sum_dst;     /* transpose_flip only: dgrad output channels [sum_dst, sum_dst+sum_count) */
  int32_t sum_src;     /*   additionally receive the weights of forward input channels             */
  int32_t sum_count;   /*   [sum_src, sum_src+sum_count) (x4 = lrelu(a4) + x2, block.py:266)      */
  int32_t ups_dgrad;   /* 1 (with transpose_flip, ks==3): emit the 4x4/stride-2 kernel that is the
                          exact adjoint of nearest-x2-upsample + 3x3 conv (block.py:315-322)       */
  /* gather == 1 (with transpose_flip == 1, ks == 3): this entry fills ONE K range of a larger packed
   * operand — the input-gradient conv of a dense-block channel slice in "gather" form:
   *   g_slice = sum over the later convs k of  conv_k^T[slice] (g_ak)
   * is ONE conv whose K dimension concatenates the gradients g_ak (dense connectivity mirrored), so
   * the slice's gradient is written once instead of being accumulated conv by conv.  The piece
   * maps packed row r -> forward input channel src_co0 + r (r < dst_cout) and packed K index c
   * (c < cout, placed at chunks [dst_chunk0, ...) of dst_nchunks) -> forward output channel c;
   * weights are multiplied by `scale`; src_ks == 1 embeds a 1x1 kernel as the centre tap. */
  int32_t gather, dst_cout, dst_chunk0, dst_nchunks, src_co0, src_ks;
  float scale;
  int32_t ups_fwd;     /* 1 (ks == 3, no transpose_flip): emit the 4-phase 2x2 ("sub-pixel") form of
                          nearest-x2-upsample + 3x3 conv (block.py:315-322), the operand of esr_conv.upsample == 3:
                          output pixel (2y+dy, 2x+dx) only ever sees the 2x2 input pixels (y-1+dy .., x-1+dx ..),
                          each weighted by the SUM of the 3x3 taps that land on it — 16 instead of 36 MACs per
                          4 outputs.  Packed as a 2x2 conv with 4*cout_blocks blocks, phase-major:
                          [phase = 2 dy + dx][cout_block][chunk][2 a + b] */
  int32_t fold_co0;    /* gather pieces (3x3): when > 0, the weights of forward input channels fold_co0 + r are ADDED to
                          those of src_co0 + r — the identity path x4 = lrelu(a4) + x2 (block.py:266) folded into conv5's
                          x2 columns, so that the backward chain needs no residual for g_x2.  0: off */
  int32_t one_t;       /* 1: the transposed 1x1 of a dense block (src = conv1x1.weight [32][64]) as the backward chain
                          consumes it: g_x[64] += W^T g_x2[32]; 2 cout blocks x 2 K chunks of 1 KB, chunk c holding the
                          K order k = 8 h + i <-> g_x2 channel 16 h + 8 c + i (the lane's packed epilogue registers
                          are the B fragments).  fp16 only */
} esr_pack;

/* All weight packs of a network in ONE launch: `table` is a DEVICE array of n esr_pack entries,
 * `piece_begin` a DEVICE int64[n+1] prefix sum of their 16-byte piece counts. */
typedef struct esr_pack_batch {
  const esr_pack* table;
  const int64_t* piece_begin;
  int32_t n, _pad;
  int64_t total_pieces;
} esr_pack_batch;
int64_t esr_pack_pieces(const esr_pack* p);   /* host helper: 16-byte pieces one entry produces */

size_t esr_packed_weight_bytes(int32_t cout, int32_t cin, int32_t ks, int32_t dtype);

/* NCHW fp32 <-> G32 (the tensors crossing the nn.Module boundary are NCHW fp32:
 * test_image/test.py:31-37, SRRaGAN_model.py:103-111). */
typedef struct esr_layout {
  int32_t dtype;
  int32_t to_g32;      /* 1: NCHW -> G32 (pads channels with zeros), 0: G32 -> NCHW */
  int32_t B, C, H, W;
  float* nchw;
  esr_g32 g32;
  int32_t use_affine;  /* to_g32 only, C <= 4: v = (v - mean_c[c]) * inv_std_c[c]  (VGG input
                          norm, architecture.py:304-305) */
  float mean_c[4];
  float inv_std_c[4];
  int32_t accumulate;  /* to_g32 == 0 only: nchw += value instead of nchw = value (round 4: the input gradients of netF
                          and netD land in the buffer that already holds the pixel loss's, SRRaGAN_model.py:124-140:
                          dL/d fake_H = d l_pix + d l_fea + d l_gan without add launches) */
  int32_t _pad;
} esr_layout;

/* Philox-4x32-7 + Box-Muller N(0,1) fill (csrc/common.h), NCHW fp32 — the exact z the fused noise epilogue uses for
 * (seed, layer); lets tests feed the same z to the oracle (GaussianNoise, block.py:117-122). */
typedef struct esr_noise_fill {
  float* dst; int32_t B, C, H, W; uint64_t seed; uint32_t layer;
} esr_noise_fill;

/* Weight + bias gradient of a fused conv (autograd's conv backward-weight; SRRaGAN_model.py:140).
 * Accumulates with fp32 atomics: the caller zeroes dw / dbias first. */
typedef struct esr_wgrad {
  int32_t dtype, ks, stride, upsample;
  int32_t B, H, W;     /* size of g (= conv OUTPUT) */
  int32_t cout, cin;   /* true channel counts of dw [cout][cin][ks][ks] */
  esr_g32 g;           /* upstream gradient wrt the conv's pre-activation output */
  esr_g32 in;          /* the conv's saved input */
  float* dw;
  float* dbias;        /* may be NULL */
  float scale;
  int32_t tap_major;   /* 1: dw is laid out [tap][cout][cin] (lane-contiguous atomics: 2 cache lines per
                          wave instruction instead of ~36); esr_grad_unpermute restores OIHW */
  /* Deterministic two-stage reduction (ESR_F16 kernels): with `partial` set every workgroup stores its
   * 32x32xtaps partial with plain stores into its own slot of this arena and a second launch of the same
   * call adds the slots of a conv up in a fixed order into dw / dbias (+=, no atomics): bit-identical
   * gradients run to run.  `partial_elems` = floats available; esr_wgrad_workspace_elems() returns what an
   * op list needs (the arena is shared by all wgrad ops of a stream: a slot lives only inside one call).
   * NULL: fp32 atomicAdd (results vary in the last bits with the arrival order). */
  float* partial;
  int64_t partial_elems;
} esr_wgrad;

/* One launch that rewrites every tap-major gradient block into its OIHW slot.  `table` is a DEVICE
 * array of n entries {src_off, dst_off, elem_begin, cout, cin, ntap} (int64 x3, int32 x3, pad);
 * src/dst are fp32 arenas. */
typedef struct esr_unperm_entry {
  int64_t src_off, dst_off, elem_begin;
  int32_t cout, cin, ntap;
  int32_t pair_begin;       /* number of (cout, cin) pairs of the entries before this one (used when n_pairs > 0) */
} esr_unperm_entry;
typedef struct esr_unpermute {
  const esr_unperm_entry* table;
  int32_t n;
  int32_t n_pairs;          /* > 0: the sum of cout * cin over the table; the kernel then runs one thread per pair
                               (ntap coalesced reads, ntap adjacent writes, one table search per pair); 0: per element */
  int64_t total;
  const float* src;
  float* dst;
} esr_unpermute;

/* BatchNorm2d(affine) over a G32 tensor (block.py:28-32; Discriminator_VGG_128,
 * architecture.py:93-118), split in phases so each is one memory pass:
 *   STATS      sums[c] += sum x, sums[C+c] += sum x^2            (fp64 atomics; caller zeroes sums)
 *   FINALIZE   mean/invstd from sums (training: biased var; running stats updated with momentum and
 *              the UNBIASED var) or from running stats (eval)
 *   APPLY      y = act((x-mean)*invstd*gamma + beta)
 *   BWD_REDUCE g' = g*act'(y);  sums[c] += sum g',  sums[C+c] += sum g'*xhat
 *   BWD_FINAL  dgamma += sums[C+c]; dbeta += sums[c]
 *   BWD_APPLY  gx = gamma*invstd*(g' - sum g'/N - xhat*sum(g' xhat)/N)   (eval: gamma*invstd*g') */
enum esr_bn_mode { ESR_BN_STATS = 0, ESR_BN_FINALIZE = 1, ESR_BN_APPLY = 2, ESR_BN_BWD_REDUCE = 3,
                   ESR_BN_BWD_FINAL = 4, ESR_BN_BWD_APPLY = 5,
                   ESR_BN_FIN_APPLY = 7,  /* training: FINALIZE and APPLY as ONE pass (every workgroup forms its channels'
                                          statistics from the sums; mean / invstd / running statistics / num_batches_tracked
                                          are written as FINALIZE would).  BWD_APPLY likewise performs BWD_FINAL when
                                          dgamma is set */
                   ESR_BN_RESTAT = 6   /* training: apply the running-statistics (and num_batches_tracked) updates of
                                          ANOTHER forward call over the same batch from the sums still in place, groups in
                                          REVERSE order — the train step calls netD on (fake, real) and then, weights
                                          unchanged, on (real, fake) (SRRaGAN_model.py:133-134,150-151): the second pair's
                                          activations equal the first's, only the BatchNorm buffers move */ };
typedef struct esr_bn {
  int32_t dtype, mode;
  int32_t B, C, H, W;
  int32_t training, act;
  float momentum, eps;
  esr_g32 x, y, g, gx;
  double* sums;                 /* [2*C] */
  float* mean; float* invstd;   /* [C] batch statistics kept for backward */
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;
  float* dgamma; float* dbeta;
  /* groups > 1: the batch is `groups` independent BatchNorm batches of B/groups images each (the reference's
   * netD(real) and netD(fake) calls, SRRaGAN_model.py:133-134,150-151, as ONE launch per layer): sums is
   * [groups][2*C], mean / invstd are [groups][C]; FINALIZE applies the running-statistics update once per group,
   * in group order; BWD_FINAL adds the groups' sums into dgamma / dbeta.  0 or 1 = one batch. */
  int32_t groups, _pad;
  int64_t* num_batches_tracked; /* FINALIZE (training): += groups (NULL: caller counts) */
} esr_bn;

/* MaxPool2d(2,2) (torchvision VGG19 cfg 'E', architecture.py:287-298).  mode 0: y = pool(x);
 * mode 1: gx = route g to the FIRST maximum of each 2x2 window (torch semantics).
 * Round 5 — the same struct carries nn.PixelShuffle(2) (pixelshuffle_block, block.py:299-312; SRResNet's upsampler):
 * mode 2: y[b][c][2h+i][2w+j] = x[b][4c + 2i + j][h][w]   (C = channels of y, H x W = size of x; C % channel group == 0)
 * mode 3: its adjoint, gx[b][4c + 2i + j][h][w] = g[b][c][2h+i][2w+j], with relu_mask: times ReLU'(x) of the conv that
 *         produced x (the activation behind the shuffle commutes with it, so it rides in the conv's epilogue). */
typedef struct esr_pool {
  int32_t dtype, mode;
  int32_t B, C, H, W;           /* OUTPUT (pooled) size */
  int32_t relu_mask;            /* mode 1: also apply ReLU' of the pooled tensor's producer (x > 0) */
  int32_t _pad;
  esr_g32 x, y, g, gx;
} esr_pool;

/* nn.Linear on row-major fp32 (Discriminator classifier, architecture.py:121-123).
 * mode 0: y = act(x W^T + b);  mode 1: gx = (g * act'(y_saved)) W ... see fields;  mode 2: dw, db. */
typedef struct esr_linear {
  int32_t mode, B, I, O, act;
  int32_t in_act;               /* mode 1 (input gradient): the activation that PRODUCED x (a conv + LeakyReLU feeding the
                                   head without a norm layer, architecture.py:163-170): gx *= in_act'(x); 0 = none */
  const float* x; const float* w; const float* b; float* y;
  const float* g;               /* dL/dy (already multiplied by act' by the caller kernel: see gmask) */
  const float* ysaved;          /* bwd: saved activation output of THIS layer for act' (may be NULL) */
  float* gx; float* dw; float* db;
} esr_linear;

/* One axis of a separable resampling on fp32 planes (MATLAB-style bicubic `imresize`,
 * codes/data/util.py:221-343: the H pass then the W pass).  out[p][i][j] = sum_t w[o][t] * in[p][..]
 * where o is the output coordinate along `axis` (0 = rows/H, 1 = columns/W) and idx[o][t] the source
 * coordinate of tap t with the symmetric border padding already resolved.  Tables are DEVICE arrays. */
typedef struct esr_resample {
  const float* in; float* out;
  int32_t planes, in_h, in_w, out_len, axis, taps;
  const float* w;        /* [out_len][taps] */
  const int32_t* idx;    /* [out_len][taps] */
} esr_resample;

/* Adam over a whole network in ONE launch (torch.optim.Adam semantics, SRRaGAN_model.py:77-91:
 * amsgrad off; the reference steps ~770 parameter tensors per optimizer).  `entries` / `blocks` are
 * DEVICE tables built once per parameter set: entry e = {param pointer, offset of its gradient /
 * moments in the flat fp32 buffers, element count}; block b = {entry, first element}: each
 * workgroup updates up to ESR_ADAM_BLOCK_ELEMS consecutive elements of one tensor.
 *   g = grad[goff+i]*grad_scale (+ weight_decay*p);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)          bc1 = 1-b1^t, bc2 = 1-b2^t (host). */
#define ESR_ADAM_BLOCK_ELEMS 4096
typedef struct esr_adam_entry { float* p; int64_t goff; int64_t n; } esr_adam_entry;
typedef struct esr_adam_block { int32_t entry; int32_t first; } esr_adam_block;
typedef struct esr_adam {
  const esr_adam_entry* entries;
  const esr_adam_block* blocks;
  int32_t nblocks, _pad;
  const float* grad;      /* flat gradients (loss-scaled; see grad_scale) */
  float* exp_avg;         /* flat first / second moments, same offsets as grad */
  float* exp_avg_sq;
  float lr, beta1, beta2, eps, bc1, bc2, grad_scale, weight_decay;
  const float* amp_state; /* NULL, or the DEVICE state of dynamic loss scaling (esr_amp): gradients are additionally
                             divided by amp_state[0] (the loss scale), and the whole update is skipped — weights and
                             moments untouched — while amp_state[4 + amp_slot] != 0 (a non-finite gradient was found in
                             THIS optimizer's gradients).  No host synchronisation anywhere: the fp16 train path's
                             overflow handling */
  int32_t amp_slot, _pad2;
  const float* step_count; /* NULL: bc1 / bc2 above are used.  Else a DEVICE counter of the steps this optimizer has
                             APPLIED including the current one (ESR_AMP_COUNT advances it only when the step is not
                             skipped): bc = 1 - beta^step_count[0] is formed in the kernel from beta1_d / beta2_d, so a
                             skipped step does not age Adam's bias correction (torch.amp.GradScaler semantics) */
  double beta1_d, beta2_d;
} esr_adam;

/* Dynamic loss scaling for the fp16 training path (new capability; the reference trains in fp32 only).
 * state = DEVICE float[8] {scale, -, good_steps, -, found[0..3]}: one found-a-non-finite-gradient flag per optimizer
 * (slot), as torch.amp.GradScaler tracks found_inf per optimizer — an overflow in G's gradients does not skip D's step.
 *   ESR_AMP_CHECK   found[slot] |= any(!isfinite(grad[0..n)))       (after the backward / gradient exchange)
 *   ESR_AMP_COUNT   if (!found[slot]) step_count[0] += 1            (before esr_adam_step: the step it will apply)
 *   ESR_AMP_UPDATE  any(found) ? (scale *= backoff, good = 0) : (++good == interval ? (scale *= growth, good = 0) : -);
 *                   found[*] = 0                                      (once per iteration, after the optimizers)  */
enum esr_amp_mode { ESR_AMP_CHECK = 0, ESR_AMP_UPDATE = 1, ESR_AMP_COUNT = 2 };
typedef struct esr_amp {
  int32_t mode, interval;
  float* state;
  const float* grad;
  int64_t n;
  float growth, backoff;
  int32_t slot, _pad;         /* 0..3 (CHECK, COUNT) */
  float* step_count;          /* COUNT */
} esr_amp;

/* ---- validation metrics on the device (metrics.hip) -------------------------------------------------
 * tensor2img (codes/utils/util.py:71-95: clamp, scale, x255, round, RGB -> BGR, HWC uint8) of one image or of
 * an (sr, hr) pair, and on the pair, cropped by `crop` pixels per side (codes/train.py:131-148):
 *   out[0] = sum of squared uint8 differences, out[1] unused, out[2] = sum of the SSIM map (util.py:117-158;
 *   over all compared planes), out[3] unused;   PSNR = 20 log10(255 / sqrt(out[0] / n)),  SSIM = out[2] / n_ssim
 * with n = (H-2crop)(W-2crop) planes, n_ssim = (H-2crop-10)(W-2crop-10) planes (the caller divides).
 * y_only (C == 3): compare the MATLAB-style Y planes of the BGR images as codes/test.py:81-90 forms them — bgr2ycbcr on
 * the float images, data/util.py:150-168, i.e. luma NOT rounded to uint8. */
typedef struct esr_l1_loss {
  const float* a; const float* b;
  float* grad_a;               /* may be NULL */
  float* loss;                 /* 1 float */
  double* scratch;             /* 2 doubles, zero */
  int64_t n;
  float weight;
  float grad_scale;            /* round 4: grad_a is additionally multiplied by this (the loss scale of the fp16 path: the
                                  loss itself stays unscaled); 0 = 1 */
  const float* grad_scale_dev; /* non-NULL: multiplied in as well, read on the device (dynamic loss scaling, esr_amp state[0]) */
} esr_l1_loss;

typedef struct esr_ragan_loss {
  const float* x; const float* y;      /* n logits each */
  float* grad_x; float* grad_y;        /* may be NULL */
  float* loss; float* mean_x; float* mean_y;
  float* bce_x; float* bce_y;          /* optional: the two BCE terms, unweighted (the reference logs them) */
  int32_t n;
  float tx, ty, weight;                /* targets (1 = real, 0 = fake) of the x / y terms */
  /* Data-parallel form (round 3): the batch means stay GLOBAL (SRRaGAN_model.py:136-137,151-152) while every
   * launch only sees the rank's own n logits — the two scalar sums cross the ranks (RCCL all-reduce) between launches:
   *   mode 0  everything from the local batch (one GPU)
   *   mode 1  sums[0..1] = {sum x, sum y}
   *   mode 2  means = ext[0]/ext[2], ext[1]/ext[2] (global sums, global count): loss, mean_x/y, bce_x/y and
   *           sums[0..1] = {sum_i sigmoid(x_i - mean y) - tx,  sum_i sigmoid(y_i - mean x) - ty}
   *   mode 3  grad_x / grad_y from the global means and ext[3..4] = those two sums over all ranks         */
  int32_t mode;
  float grad_scale;                    /* round 4: grad_x / grad_y times this (loss scale); 0 = 1 */
  float* sums;                         /* 2 floats (modes 1, 2) */
  const float* ext;                    /* 5 floats (modes 2, 3) */
  const float* grad_scale_dev;         /* non-NULL: multiplied in as well, read on the device */
} esr_ragan_loss;

typedef struct esr_img_metrics {
  const float* sr;           /* [C][H][W] fp32 */
  const float* hr;           /* same shape, or NULL: conversion only */
  int32_t C, H, W, crop;
  int32_t y_only, _pad;
  float lo, hi;              /* tensor2img's min_max */
  uint8_t* img_sr;           /* [H][W][C] uint8 BGR */
  uint8_t* img_hr;
  double* y_sr;              /* [H][W] fp64 (y_only): UNROUNDED luma, codes/test.py:81-86 converts the float images */
  double* y_hr;
  double* out;               /* 4 doubles (device) */
  double win[11];            /* 1-D Gaussian window (cv2.getGaussianKernel(11, 1.5)); the 2-D one is its outer product */
} esr_img_metrics;

/* ---- fused ResidualDenseBlock_5C chain (rdb_fused.hip) -------------------------------------------
 * ONE persistent launch runs n_blocks dense blocks (block.py:260-268) back to back on every 16x32
 * tile, the fp32 accumulators of all 192 output channels of the block in flight held in registers
 * ("input stationary": each input slice x, x1..x4 is staged once and feeds every later conv), with
 * the 1-pixel halos of x1..x4 / the block output exchanged between neighbouring tiles inside the
 * launch.  The RRDB tail (block.py:291: out*0.2 + x_rrdb; test_image/block.py:256 adds a noise
 * layer) is the second residual stage of a block whose res2 is set.
 *   x1 = lrelu(conv1(x));  x2 = lrelu(conv2(x,x1)) + conv1x1(x);  x3 = lrelu(conv3(x..x2));
 *   x4 = lrelu(conv4(x..x3)) + x2;  y = noise1(conv5(x..x4)*0.2 + x);  [y = noise2(y*0.2 + res2)]
 * Weights: one fused stream per block (esr_rdb_weight_stream_bytes), 1 KB MFMA A fragments
 * F(blk, c, kh, kw) = packed(conv)[cout_block][chunk c][kh][kw] (blk 0..3 = conv1..conv4, 4/5 = conv5's
 * two cout blocks) in the order the kernel consumes them.  With c running over the K steps (chunks) of
 * the input slice of phase p = 1..5 (x, x1, x2, x3, x4):
 *   ESR_F32:  for p, c, kw, blk = p-1..5, kh: F;  then the 1x1's chunks.
 *   ESR_F16:  for p = 1..4: [for c, kw, kh: F(p-1)]  [for c, kw, blk = p..5, kh: F]  (after p = 1: the
 *             1x1's chunks);  p = 5: for c, kw, blk = 4..5, kh: F.
 *             (conv_p alone first, so that its epilogue and halo hand-off overlap the remaining convs.)
 * Build it from the per-conv packed weights with esr_gather_fragments.
 * All views share one geometry (same wp / H / W); x_out may alias x_in or res2 (pixel-local). */
typedef struct esr_rdb_block {
  const void* w;            /* fused weight stream of this block */
  const float* bias;        /* fp32 [192]: biases of conv1..conv4 (32 each) then conv5 (64), gathered by the caller
                               (esr_gather_fragments with piece_bytes = 128 does it in one launch for a whole net) */
  esr_g32 x_in;             /* 64-channel block input */
  esr_g32 x_out;            /* 64-channel block output */
  esr_g32 res2;             /* RRDB input of the fused RRDB tail; ptr NULL = plain dense block */
  uint32_t layer1, layer2;  /* Philox stream ids of the block noise / the RRDB-tail noise; 0xFFFFFFFF = off */
  uint32_t flags;           /* ESR_RDB_FULL_OUT: every pixel of x_out must reach memory — the chain's last block, or an
                               output that is read again as a later block's res2.  Otherwise (fp16, no noise) only the
                               tiles' border pixels are stored: the next block takes its input from the LDS and its
                               `+ x` residual from the accumulators this block's epilogue primes with 5 x */
  uint32_t _pad;
  /* ---- training chains (esr_rdb_chain.mode 1 / 2, fp16) ---- */
  esr_g32 dense;            /* this block's own 128 channels: mode 1: x1..x4 (kept for the backward);
                               mode 2: g_a4 | g_a3 | g_a2 | g_a1 (what the weight gradients read) */
  void* mask;               /* LeakyReLU masks of a1..a4, one bit per element: B * tiles records of 8 KB in the chain's
                               16x32 tile order, [slice a1..a4][wave][lane][4 rows x u16] (esr_rdb_mask_bytes); written
                               by mode 1, read by mode 2 of the same geometry */
  esr_g32 aux;              /* mode 2: unmasked g_x2 (32 channels), input of the 1x1's weight gradient */
  esr_g32 out_a;            /* mode 2, blocks with res2 (the RDB1 of an RRDB): A' = (acc + res2) (1 + sigma z[layer2]) is
                               stored here (the skip gradient of the previous RRDB) and x_out = 0.2 A' (1 + sigma z[layer1]);
                               ptr NULL: x_out = (acc [+ res2]) (1 + sigma z[layer1]) */
} esr_rdb_block;
#define ESR_RDB_FULL_OUT 1u
#define ESR_RDB_BAND_OWN 2u  /* banded launch (esr_rdb_chain.band_rows): x_out receives only the band's OWN rows */
/* Backward weight stream of a block (mode 2): the forward's layout over the gather-form operands
 *   blk 0..3 = the x4, x3, x2, x1 slice convs (esr_pack.gather; the x2 one with fold_co0 = the x4 columns of conv5),
 *   blk 4/5 = the x slice conv's two cout blocks (K = 192: g_t, g_a4..g_a1),
 * crit_p / bulk_p as in the forward; the 1x1 unit (esr_pack.one_t: 2 cout blocks x 2 chunks) sits behind crit_3. */

typedef struct esr_rdb_chain {
  int32_t dtype, B, H, W;
  int32_t n_blocks;
  int32_t noise_mode;       /* ESR_NOISE_OFF or ESR_NOISE_PHILOX (explicit z: per-conv path) */
  float sigma;
  int32_t save_dense;       /* 0: fp16 path stores only each tile's border pixels of x1..x4 (all the launch itself
                               re-reads); 1: the whole slices reach `dense` (e.g. to inspect / save activations) */
  uint64_t seed;
  const uint64_t* seed_dev; /* as esr_conv.seed_dev */
  esr_g32 dense;            /* 128-channel scratch: x1..x4 of the block in flight */
  const esr_rdb_block* blocks;  /* DEVICE array of n_blocks entries */
  void* workspace;          /* esr_rdb_workspace_bytes(B,H,W) bytes of device memory; word 1 != 0 after the
                               launch = a bounded spin timed out (results invalid) */
  size_t workspace_bytes;
  uint64_t* trace;          /* measurement only (NULL = off): per tile 64 x uint64 time stamps (100 MHz) */
  int32_t mode;             /* 0: inference forward (esr_rdb_forward);  1: TRAINING forward (esr_rdb_forward, fp16): every
                               block writes x1..x4 to its own `dense`, its output in full and its LeakyReLU masks;
                               2: BACKWARD (esr_rdb_backward, fp16): blocks in backward order, x_in = the block's g_t
                               (dL/d(conv5 * 0.2 + x)), weights = the gather-form streams (below), x_out = the next
                               block's g_t */
  int32_t _pad2;
  /* Row bands (mode 0, noise off): an image with more 16x32 tiles than CUs cannot run as one chain (a tile spins on
   * its neighbours, so all tiles of an image must be resident).  It is cut into bands of `band_rows` rows and the
   * bands take the place of the batch: B = number of bands, H = band_rows + 2 * band_margin, every view's
   * batch_stride = band_rows rows, and view row 0 of band b is image row b * band_rows - band_margin (the caller's
   * buffers carry band_margin rows of zero padding above and below the image).  A band recomputes band_margin rows
   * of its neighbours on either side — band_margin >= 5 * n_blocks, the chain's dependency radius — and the kernel
   * writes only rows that exist in the image (0 <= row < img_H), and into an ESR_RDB_BAND_OWN block's x_out only the
   * band's own rows.  band_rows = 0: off. */
  int32_t band_rows, band_margin, img_H, _pad3;
} esr_rdb_chain;

/* Piece gather: dst[f * piece_bytes ..] = src_base[src_off[f] ..] for f < n (src_off: DEVICE int64 byte offsets;
 * src_base may be NULL, then they are absolute addresses).  piece_bytes: power of two in 16..4096, 0 = 1024 (one MFMA
 * A fragment). */
typedef struct esr_frag_gather {
  const int64_t* src_off;
  const void* src_base;
  void* dst;
  int64_t n;
  int32_t piece_bytes, _pad;
} esr_frag_gather;

/* ---- weight / bias gradients of a whole dense block in one pass (rdb_wgrad.hip) --------------------------------
 * Replaces the six esr_conv_wgrad problems of a ResidualDenseBlock_5C (block.py:239-268; autograd's conv
 * backward-weight, SRRaGAN_model.py:140) for n_blocks blocks in ONE launch (+ one deterministic reduce launch):
 *   in = [x (64) | x1 | x2 | x3 | x4] (192 channels),   q = [g_t (64) | g_a4 | g_a3 | g_a2 | g_a1 | g_x2] (224 channels)
 *   dW_convk += g_ak (x) in[0 : 32 (k + 1))  (3x3; conv5: g_a5 = scale5 * g_t),  dW_1x1 += g_x2 (x) x,  db_k += sum g_ak
 * where g_x2 is the gradient of x2 BEFORE its LeakyReLU mask (it feeds the bias-free 1x1, block.py:263).
 * Both views share one geometry (wp, H, W); one image of either must be < 4 GB.  dw[k]: k = 0..4 conv1..conv5 ([cout][cin][3][3] fp32, or tap-major
 * [tap][cout][cin] when tap_major; accumulated: +=), k = 5 the 1x1 ([32][64]); db[k] k = 0..4 (NULL: skipped). */
typedef struct esr_rdb_wgrad_block {
  esr_g32 in;               /* the block's saved concat buffer (forward) */
  esr_g32 q;                /* the block's gradient concat (backward chain) */
  float* dw[6];
  float* db[6];             /* db[5] unused (the 1x1 has no bias) */
} esr_rdb_wgrad_block;
typedef struct esr_rdb_wgrad {
  int32_t dtype;            /* ESR_F16 */
  int32_t B, H, W;
  int32_t n_blocks, tap_major;
  float scale5;             /* 0.2: g_a5 = 0.2 g_t (block.py:267) */
  float scale;              /* applied to every gradient (1.0) */
  const esr_rdb_wgrad_block* blocks;   /* DEVICE array */
  float* partial;           /* esr_rdb_wgrad_workspace_elems() floats: per-task partial sums, reduced in a fixed order
                               (+ the step counters that keep the four channel sets of a row band in lock step) */
  int64_t partial_elems;
  int32_t max_workgroups;   /* 0 (default): one workgroup per task, a plain non-persistent grid; n > 0: a persistent
                               grid of at most n workgroups striding over the tasks — the caller runs the pass NEXT TO
                               other work (the train plan: a run of RRDBs' weight gradients on the side stream under the
                               next run's backward chain) */
  int32_t _pad;
} esr_rdb_wgrad;

enum esr_op_kind { ESR_OP_CONV = 1, ESR_OP_PACK = 2, ESR_OP_LAYOUT = 3, ESR_OP_NOISE_FILL = 4,
                   ESR_OP_WGRAD = 5, ESR_OP_BN = 6, ESR_OP_POOL = 7, ESR_OP_LINEAR = 8,
                   ESR_OP_UNPERMUTE = 9, ESR_OP_PACK_BATCH = 10,
                   ESR_OP_RDB_CHAIN = 11, ESR_OP_FRAG_GATHER = 12, ESR_OP_RDB_WGRAD = 13,
                   ESR_OP_RDB_CHAIN_BWD = 14 /* u.rdb_chain with mode 2 */ };

/* esr_op.flags */
#define ESR_OPF_SIDE 1   /* on a run of consecutive ESR_OP_WGRAD ops: launch the run on the library's side
                            stream, ordered after every earlier op.  It is complete (a) before any op that
                            follows the NEXT side run, (b) before any ESR_OP_UNPERMUTE op, (c) when the work
                            esr_run_ops enqueued on `stream` completes.  The caller guarantees that nothing
                            up to the end of the next side run writes the run's inputs or reads its outputs
                            (the train plan double-buffers the gradient slices it reads). */
#define ESR_OPF_SIDE_FREE 2   /* with ESR_OPF_SIDE: the run's inputs are never overwritten and its outputs never read
                            inside this list, and its esr_wgrad.partial region belongs to this run alone — it is
                            launched without waiting for earlier side runs, on one of three library-owned streams
                            (round robin), complete before any ESR_OP_UNPERMUTE and when the list's work on `stream`
                            completes.  (The discriminator's backward: ten independent, latency-bound weight
                            gradients next to the dgrad chain.) */

#define ESR_OPF_FOLLOW 4   /* on an ESR_OP_RDB_WGRAD op that stands right behind the ESR_OP_RDB_CHAIN_BWD op whose blocks it
                            lists (same blocks, the chain's order): the pass is launched TOGETHER with the chain, on the
                            library's side stream, as a persistent grid on the CUs the chain's grid leaves free (at least
                            32; esr_rdb_wgrad.max_workgroups caps it further) — a task of block k starts when the chain's
                            tiles it reads have published the end of block k, a block's partial sums are reduced by
                            its last task.  Results are bit-identical to the pass run behind the chain; completion as for
                            ESR_OPF_SIDE.  For launches that leave CUs free (training crops); refused otherwise. */

typedef struct esr_op {
  int32_t kind;
  int32_t flags;
  union {
    esr_conv conv;
    esr_pack pack;
    esr_layout layout;
    esr_noise_fill noise_fill;
    esr_wgrad wgrad;
    esr_bn bn;
    esr_pool pool;
    esr_linear linear;
    esr_unpermute unpermute;
    esr_pack_batch pack_batch;
    esr_rdb_chain rdb_chain;
    esr_frag_gather frag_gather;
    esr_rdb_wgrad rdb_wgrad;
  } u;
} esr_op;

/* Geometry helper: padded plane size for a logical HxW image. */
void esr_g32_dims(int32_t H, int32_t W, int32_t* Hp, int32_t* Wp);

/* Single-op entry points. */
int esr_conv_forward(const esr_conv* p, esr_stream_t stream);
int esr_pack_conv_weights(const esr_pack* p, esr_stream_t stream);
int esr_convert_layout(const esr_layout* p, esr_stream_t stream);
int esr_fill_noise(const esr_noise_fill* p, esr_stream_t stream);
int esr_conv_wgrad(const esr_wgrad* p, esr_stream_t stream);
/* n independent weight-gradient problems (disjoint dw/dbias blocks).  fp16 3x3/s1 and 1x1 entries are
 * packed, up to 8 at a time, into ONE launch (at training sizes a single conv's wgrad is ~64
 * mostly-idle workgroups); anything else falls back to esr_conv_wgrad.  esr_run_ops routes every run
 * of consecutive ESR_OP_WGRAD ops through this entry. */
int esr_conv_wgrad_multi(const esr_wgrad* items, int32_t n, esr_stream_t stream);
int esr_batchnorm(const esr_bn* p, esr_stream_t stream);
int esr_maxpool2(const esr_pool* p, esr_stream_t stream);
int esr_linear_op(const esr_linear* p, esr_stream_t stream);
int esr_grad_unpermute(const esr_unpermute* p, esr_stream_t stream);
int esr_adam_step(const esr_adam* p, esr_stream_t stream);
int esr_amp_step(const esr_amp* p, esr_stream_t stream);
int esr_resample_axis(const esr_resample* p, esr_stream_t stream);
int esr_pack_conv_weights_batch(const esr_pack_batch* p, esr_stream_t stream);
/* Fused dense-block chain (replaces 5 x n_blocks esr_conv_forward launches; block.py:260-268,287-291).
 * Every launch assumes its whole grid resident.  Launches on one stream are ordered by the stream; a launch whose
 * grid does not fit next to the chain launches still in flight on OTHER streams is made to wait for them on the
 * device (event wait; the call itself never blocks).  Streams under graph capture are not tracked. */
int esr_rdb_forward(const esr_rdb_chain* p, esr_stream_t stream);
/* Fused dense-block BACKWARD chain (mode 2): the input gradients of n_blocks dense blocks (autograd backward of
 * block.py:260-268,287-291, triggered at SRRaGAN_model.py:140) in one persistent launch — replaces 5 x n_blocks dgrad
 * launches; the weight gradients follow with esr_rdb_wgrad_run over the Q buffers this launch leaves complete. */
int esr_rdb_backward(const esr_rdb_chain* p, esr_stream_t stream);
size_t esr_rdb_mask_bytes(int32_t B, int32_t H, int32_t W);   /* esr_rdb_block.mask of one block */
/* Sticky abort report: 1 if a chain launch — or a follower pass of weight gradients (ESR_OPF_FOLLOW) waiting for one —
 * since the last call gave up on a bounded spin (its results are invalid; e.g. another kernel held CUs for > 1 s), else 0.  Reads a pinned host word: no synchronisation, but only meaningful
 * for launches that have completed.  esr_run_ops / esr_rdb_forward / esr_rdb_backward check it on entry and fail with
 * ESR_ERR_LAUNCH, so an aborted launch cannot go unnoticed past the next call. */
int esr_rdb_check_abort(void);
/* Diagnostic (tests/test_gpu_rdb_chain.py: starved launch): n_workgroups workgroups that each take a CU's LDS and
 * sleep until *release (device-visible memory) becomes non-zero or max_ms (<= 10000) have passed; every workgroup
 * that has started adds 1 to *started (optional). */
int esr_debug_hold_cus(int32_t n_workgroups, const uint32_t* release, uint32_t max_ms, uint32_t* started, esr_stream_t stream);
/* Diagnostic (bench.py `mfma_sustained_probe`; tools/experiments/mfma_power_probe.hip is the stand-alone form): what the
 * matrix pipe sustains under the package power limit.  n_workgroups workgroups of four waves (one per SIMD; one workgroup
 * per CU) each issue iters x 16 independent v_mfma_f32_32x32x16_f16 from registers; operands: 8 x 256 x 16 bytes of fp16
 * (four A and four B fragments per lane: zeros reach the quoted 2.5 PFLOP/s, random values ~1.65 on MI355X).  clk2[0] /
 * clk2[1] = shader-clock / 100 MHz ticks the loop took on workgroup 0; sink: 4 bytes nothing is written to. */
int esr_debug_mfma_probe(const void* operands, int32_t iters, uint64_t* clk2, float* sink, int32_t n_workgroups, esr_stream_t stream);
/* Diagnostic (tests/test_gpu_train_chain.py: a follower without its chain): the follower form of esr_rdb_wgrad_run
 * (ESR_OPF_FOLLOW) against `flags` — tiles_x * tiles_y words per image that the caller controls instead of a chain
 * launch.  A follower whose flags never rise gives up after 2 s, raises the abort word (esr_rdb_check_abort) and runs
 * to its end on whatever is in its inputs. */
int esr_debug_rdb_wgrad_follow(const struct esr_rdb_wgrad* p, const uint32_t* flags, int32_t tiles_x, int32_t tiles_y, esr_stream_t stream);
/* Library state and devices.  What the library keeps between calls — the chain launches in flight (the ordering above),
 * the abort word, the side streams of ESR_OPF_SIDE runs — is keyed by the CURRENT DEVICE of the calling thread:
 * nn.DataParallel's one-thread-per-device replicas (networks.py:105-107) never wait for, or report the aborts of, each
 * other's devices.  Callers on several threads / streams of ONE device share that device's side streams; their fork
 * (event record + wait) is atomic under a per-device mutex, so the sharing is safe (ESR_SIDE_PER_STREAM=1: one set of
 * side streams per caller stream instead — measured slower).
 * Diagnostics for tests on a one-GPU box: esr_debug_device_alias makes the CALLING THREAD's bookkeeping use `alias` in
 * place of hipGetDevice() (alias < 0: back to the real device; returns the previous alias); launches still go to the
 * real device.  esr_debug_chain_order_waits: cross-stream event waits the chain ordering has inserted so far. */
int esr_debug_device_alias(int32_t alias);
uint64_t esr_debug_chain_order_waits(void);
size_t esr_rdb_workspace_bytes(int32_t B, int32_t H, int32_t W);
size_t esr_rdb_weight_stream_bytes(int32_t dtype);
int esr_rdb_max_tiles_per_image(void);   /* 16x32 tiles of ONE image must not exceed this (= CUs) */
int esr_gather_fragments(const esr_frag_gather* g, esr_stream_t stream);
/* Dense-block weight gradients (replaces 6 x n_blocks esr_conv_wgrad calls). */
int esr_rdb_wgrad_run(const esr_rdb_wgrad* p, esr_stream_t stream);
int64_t esr_rdb_wgrad_workspace_elems(int32_t B, int32_t H, int32_t W, int32_t n_blocks);
int esr_image_metrics(const esr_img_metrics* p, esr_stream_t stream);   /* replaces util.py:71-158 on the device */

/* The train step's losses with their gradients, one launch each (SRRaGAN_model.py:124-137,150-156).
 * esr_l1_loss_forward: *loss = weight * mean|a - b| (nn.L1Loss, loss.py cri_pix / cri_fea);
 *   grad_a (optional) = weight * sign(a - b) / n.  scratch: 2 doubles of device memory, zero before the first
 *   call (the kernel leaves them zero).
 * esr_ragan_loss_forward: *loss = weight/2 * ( BCEWithLogits(x - mean(y), tx) + BCEWithLogits(y - mean(x), ty) )
 *   (GANLoss 'vanilla' on the relativistic-average logits, loss.py:6-38); grad_x / grad_y optional (the means'
 *   dependence on the other side included); mean_x / mean_y optional outputs (the D_real / D_fake log values). */
int esr_l1_loss_forward(const esr_l1_loss* p, esr_stream_t stream);
int esr_ragan_loss_forward(const esr_ragan_loss* p, esr_stream_t stream);

/* Run a recorded list of ops back to back on `stream` (one host call per network pass; this is
 * what RRDBNet.forward — architecture.py:76-78 — becomes). */
int esr_run_ops(const esr_op* ops, int32_t n, esr_stream_t stream);
/* floats of esr_wgrad.partial arena the wgrad ops of an op list need (max over the launches esr_run_ops forms) */
int64_t esr_wgrad_workspace_elems(const esr_op* ops, int32_t n);

/* hipGraph replay of an op list (training plans are ~1 000 launches of 10-100 us: per-launch host cost
 * is what bounds the step).  esr_graph_create captures `esr_run_ops(ops, n)` — including its
 * ESR_OPF_SIDE fork/joins — on a library-owned capture stream and instantiates it; every pointer and
 * scalar inside the ops is baked in, so callers keep their I/O in fixed buffers and pass per-step
 * scalars through device memory (esr_conv.seed_dev).  esr_graph_launch enqueues the whole list on
 * `stream` with one host call. */
typedef struct esr_graph_s* esr_graph_t;
int esr_graph_create(const esr_op* ops, int32_t n, esr_graph_t* out);
int esr_graph_launch(esr_graph_t g, esr_stream_t stream);
int esr_graph_destroy(esr_graph_t g);

/* Measurement-only variant (bench.py): brackets every op with hipEvents on `stream`, waits for the
 * stream, and writes each op's elapsed milliseconds to ms_out[n].  This is the one entry point that
 * creates events and synchronises; never call it under graph capture. */
int esr_run_ops_timed(const esr_op* ops, int32_t n, esr_stream_t stream, float* ms_out);

const char* esr_last_error(void);
int esr_abi_version(void);   /* 6 (round 6: ESR_OPF_FOLLOW, esr_debug_rdb_wgrad_follow); 5 (round 5: esr_rdb_wgrad.max_workgroups, esr_debug_device_alias / esr_debug_chain_order_waits); 4 (round 4: esr_conv.ksplit / split_ws / stat_sums, ESR_BN_FIN_APPLY / ESR_BN_RESTAT); 3 (round 3: esr_ragan_loss.mode / sums / ext, ...; 2 = round 2: esr_bn.groups / num_batches_tracked,
                                esr_l1_loss, esr_ragan_loss, ESR_OPF_SIDE_FREE) */
size_t esr_sizeof_op(void);

#ifdef __cplusplus
}
#endif
#endif /* ESRGAN_HIP_H */
