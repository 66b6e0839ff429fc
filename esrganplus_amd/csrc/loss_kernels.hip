// loss_kernels.hip — the three losses of the ESRGAN+ train step (SRRaGAN_model.py:124-137,150-156) with their
// gradients, one launch each:  L1Loss (loss.py / nn.L1Loss: cri_pix, cri_fea) and the relativistic-average GAN
// term  ( BCEWithLogits(x - mean(y), tx) + BCEWithLogits(y - mean(x), ty) ) / 2  (GANLoss 'vanilla',
// loss.py:6-38).  As torch ops these are ~70 tiny dependent launches per step (sub, abs, mean, sigmoid, ...,
// and their backward) — more chip time in launch gaps than in arithmetic.  HBM-bound, tiny.
#include "common.h"

namespace {

// loss = weight * mean|a - b| ; grad_a = weight * sign(a - b) / n.  One pass; the last workgroup to arrive turns the
// fp64 sum into the loss and clears the scratch for the next call.
__global__ __launch_bounds__(256) void l1_loss_kernel(const esr_l1_loss p) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  double s = 0.0;
  const float gw = p.weight / (float)p.n * (p.grad_scale != 0.f ? p.grad_scale : 1.f) * (p.grad_scale_dev ? *p.grad_scale_dev : 1.f);
  // 16-byte vector accesses only when all three pointers allow them (a view with an odd storage offset takes the
  // scalar path: same sums, same order per thread)
  const bool vec = (((uintptr_t)p.a | (uintptr_t)p.b | (uintptr_t)p.grad_a) & 15) == 0;
  for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < p.n; i0 += stride) {
    if (vec && i0 + 4 <= p.n) {
      const f32x4 a = *(const f32x4*)(p.a + i0), b = *(const f32x4*)(p.b + i0);
      f32x4 g;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = a[e] - b[e];
        s += fabsf(d);
        g[e] = d > 0.f ? gw : (d < 0.f ? -gw : 0.f);
      }
      if (p.grad_a) *(f32x4*)(p.grad_a + i0) = g;
    } else {
      for (int64_t i = i0; i < p.n && i < i0 + 4; ++i) {
        const float d = p.a[i] - p.b[i];
        s += fabsf(d);
        if (p.grad_a) p.grad_a[i] = d > 0.f ? gw : (d < 0.f ? -gw : 0.f);
      }
    }
  }
  __shared__ double red[4];
  __shared__ bool last;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(p.scratch, red[0] + red[1] + red[2] + red[3]);
    __threadfence();
    const unsigned long long t = atomicAdd((unsigned long long*)(p.scratch + 1), 1ull);
    last = t == (unsigned long long)gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    const double tot = atomicAdd(p.scratch, 0.0);
    *p.loss = (float)(tot / (double)p.n) * p.weight;
    p.scratch[0] = 0.0;
    *(unsigned long long*)(p.scratch + 1) = 0ull;
  }
}

__device__ __forceinline__ float softplus(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }

// one workgroup; n logits per side (the discriminator's outputs: 16 per GPU).
// p.mode (data-parallel runs keep the batch means GLOBAL, SRRaGAN_model.py:136-137,151-152 — the two tiny sums
// cross the ranks between launches of this kernel, esrganplus_amd/losses.py):
//   0  everything from the local batch: means, loss, gradients                         (one GPU)
//   1  sums[0..1] = {sum x, sum y}                                                      (-> all-reduce with n)
//   2  means = ext[0] / ext[2], ext[1] / ext[2]: loss, mean_x/y, bce_x/y, sums[0..1] = {sum (sigmoid(z1) - tx),
//      sum (sigmoid(z2) - ty)} over the local batch                                     (-> all-reduce)
//   3  gradients from the global means and ext[3..4] = those two sums over ALL ranks, ext[2] = global n
__global__ __launch_bounds__(256) void ragan_loss_kernel(const esr_ragan_loss p) {
  __shared__ float red[4][4];
  auto block_sum4 = [&](float v0, float v1, float v2, float v3, float* out) {
    float v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
      for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    for (int k = 0; k < 4; ++k) out[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
  };
  const float inv = 1.f / (float)p.n;
  float mx, my, ninv_glob = inv;
  if (p.mode == 0 || p.mode == 1) {
    float sx = 0.f, sy = 0.f;
    for (int i = threadIdx.x; i < p.n; i += 256) { sx += p.x[i]; sy += p.y[i]; }
    float m[4];
    block_sum4(sx, sy, 0.f, 0.f, m);
    if (p.mode == 1) {
      if (threadIdx.x == 0) { p.sums[0] = m[0]; p.sums[1] = m[1]; }
      return;
    }
    mx = m[0] * inv; my = m[1] * inv;
  } else {
    ninv_glob = 1.f / p.ext[2];
    mx = p.ext[0] * ninv_glob; my = p.ext[1] * ninv_glob;
  }
  // z1 = x - mean(y) against tx, z2 = y - mean(x) against ty
  const float hw = 0.5f * p.weight;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.mode != 3) {
    float l1 = 0.f, l2 = 0.f, d1 = 0.f, d2 = 0.f;
    for (int i = threadIdx.x; i < p.n; i += 256) {
      const float z1 = p.x[i] - my, z2 = p.y[i] - mx;
      l1 += softplus(z1) - p.tx * z1;
      l2 += softplus(z2) - p.ty * z2;
      d1 += sigmoidf(z1) - p.tx;
      d2 += sigmoidf(z2) - p.ty;
    }
    block_sum4(l1, l2, d1, d2, t);
    if (threadIdx.x == 0) {
      *p.loss = hw * (t[0] + t[1]) * inv;
      if (p.mean_x) *p.mean_x = mx;
      if (p.mean_y) *p.mean_y = my;
      if (p.bce_x) *p.bce_x = t[0] * inv;
      if (p.bce_y) *p.bce_y = t[1] * inv;
      if (p.mode == 2) { p.sums[0] = t[2]; p.sums[1] = t[3]; }
    }
    if (p.mode == 2) return;
  } else {
    t[2] = p.ext[3]; t[3] = p.ext[4];
  }
  // d loss / d x_i = hw/n [ (sigmoid(z1_i) - tx) - (sum_j (sigmoid(z2_j) - ty)) / N ]: the second term is the mean's share
  // (N and the sum run over all ranks in mode 3 — every rank's loss sees the mean, gradients are averaged over ranks)
  const float gs = hw * inv * (p.grad_scale != 0.f ? p.grad_scale : 1.f) * (p.grad_scale_dev ? *p.grad_scale_dev : 1.f);
  for (int i = threadIdx.x; i < p.n; i += 256) {
    const float z1 = p.x[i] - my, z2 = p.y[i] - mx;
    if (p.grad_x) p.grad_x[i] = gs * ((sigmoidf(z1) - p.tx) - t[3] * ninv_glob);
    if (p.grad_y) p.grad_y[i] = gs * ((sigmoidf(z2) - p.ty) - t[2] * ninv_glob);
  }
}

}  // namespace

extern "C" int esr_l1_loss_forward(const esr_l1_loss* p, esr_stream_t stream) {
  if (!p || !p->a || !p->b || !p->loss || !p->scratch || p->n <= 0) {
    esr_set_error("esr_l1_loss_forward: invalid arguments");
    return ESR_ERR_INVALID;
  }
  int64_t wgs = (p->n + 4095) / 4096;           // 16 elements per thread
  if (wgs > 1024) wgs = 1024;
  hipLaunchKernelGGL(l1_loss_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, *p);
  return esr_check_launch("l1_loss_kernel");
}

extern "C" int esr_ragan_loss_forward(const esr_ragan_loss* p, esr_stream_t stream) {
  if (!p || !p->x || !p->y || p->n <= 0 || p->mode < 0 || p->mode > 3 || ((p->mode == 0 || p->mode == 2) && !p->loss) ||
      ((p->mode == 1 || p->mode == 2) && !p->sums) || (p->mode >= 2 && !p->ext)) {
    esr_set_error("esr_ragan_loss_forward: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipLaunchKernelGGL(ragan_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *p);
  return esr_check_launch("ragan_loss_kernel");
}
