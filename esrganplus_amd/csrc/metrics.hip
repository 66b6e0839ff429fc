// metrics.hip — validation metrics on the device: tensor2img, PSNR, SSIM, Y channel
// (codes/utils/util.py:71-95,107-158, codes/data/util.py:123-168 as the validation loops of codes/train.py:131-148
// and test.py use them).  HBM-bound, tiny: the point is that a validation pass over many images never leaves
// the device (one 4-double read-back per image pair instead of two full-image copies + numpy).
#include "common.h"

namespace {

// NCHW fp32 pixel -> the uint8 the reference's tensor2img stores (clamp to [lo, hi], scale, x255, round half
// to even in float32 like numpy)
__device__ __forceinline__ int quant8(float v, float lo, float hi) {
  v = fminf(fmaxf(v, lo), hi);
  v = (v - lo) / (hi - lo);
  return (int)rintf(v * 255.0f);
}
// MATLAB-style Y of a uint8 BGR pixel, NOT rounded: the validation script converts the FLOAT images
// (codes/test.py:81-86: bgr2ycbcr(sr_img / 255., only_y=True) -> data/util.py:150-168, float branch), so PSNR_Y / SSIM_Y
// compare unrounded luma values  (B 24.966 + G 128.553 + R 65.481) / 255 + 16  in [16, 235]
__device__ __forceinline__ double y_of_bgr(int b, int g, int r) {
  return ((double)b * 24.966 + (double)g * 128.553 + (double)r * 65.481) / 255.0 + 16.0;
}

// one thread per pixel: writes the HWC BGR uint8 image(s) and, when y_only, the Y planes
__global__ void tensor2img_kernel(const esr_img_metrics p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= p.W) return;
  const int64_t plane = (int64_t)p.H * p.W, pix = (int64_t)y * p.W + x;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* src = s == 0 ? p.sr : p.hr;
    uint8_t* img = s == 0 ? p.img_sr : p.img_hr;
    if (!src) continue;
    int q[3] = {0, 0, 0};
    for (int c = 0; c < p.C; ++c) q[c] = quant8(src[c * plane + pix], p.lo, p.hi);
    if (p.C == 3) {
      img[pix * 3 + 0] = (uint8_t)q[2];   // BGR (util.py:86: img_np[[2, 1, 0], :, :])
      img[pix * 3 + 1] = (uint8_t)q[1];
      img[pix * 3 + 2] = (uint8_t)q[0];
      if (p.y_only) (s == 0 ? p.y_sr : p.y_hr)[pix] = y_of_bgr(q[2], q[1], q[0]);
    } else {
      img[pix] = (uint8_t)q[0];
    }
  }
}

__device__ __forceinline__ void block_add(double v, double* dst, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    atomicAdd(dst, s);
  }
  __syncthreads();
}

// planes compared: the NP channels of the HWC images, or the single Y plane
template <typename E> __device__ __forceinline__ double px(const E* img, int np, int c, int64_t pix) { return (double)img[pix * np + c]; }

// sum of squared differences over the cropped region (util.py:107-114 on the cropped uint8 images)
template <typename E> __global__ void sse_kernel(const esr_img_metrics p, const E* a, const E* b, int np) {
  __shared__ double red[8];
  const int x = blockIdx.x * blockDim.x + threadIdx.x + p.crop, y = blockIdx.y + p.crop;
  double s = 0.0;
  if (x < p.W - p.crop) {
    const int64_t pix = (int64_t)y * p.W + x;
    for (int c = 0; c < np; ++c) {
      const double d = px(a, np, c, pix) - px(b, np, c, pix);
      s += d * d;
    }
  }
  block_add(s, p.out + 0, red);
}

// SSIM (util.py:117-158): 11x11 Gaussian window (sigma 1.5), valid region of the cropped planes, fp64
template <typename E> __global__ void ssim_kernel(const esr_img_metrics p, const E* a, const E* b, int np) {
  __shared__ double red[8];
  const int ow = p.W - 2 * p.crop - 10;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
  double v = 0.0;
  if (x < ow) {
    double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
    for (int i = 0; i < 11; ++i) {
      const int64_t row = (int64_t)(y + p.crop + i) * p.W + x + p.crop;
      for (int j = 0; j < 11; ++j) {
        const double w = p.win[i] * p.win[j];
        const double u = px(a, np, c, row + j), t = px(b, np, c, row + j);
        m1 += w * u; m2 += w * t; s11 += w * u * u; s22 += w * t * t; s12 += w * u * t;
      }
    }
    const double c1 = (0.01 * 255) * (0.01 * 255), c2 = (0.03 * 255) * (0.03 * 255);
    const double v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, cv = s12 - m1 * m2;
    v = ((2 * m1 * m2 + c1) * (2 * cv + c2)) / ((m1 * m1 + m2 * m2 + c1) * (v1 + v2 + c2));
  }
  block_add(v, p.out + 2, red);
}

}  // namespace

extern "C" int esr_image_metrics(const esr_img_metrics* p, esr_stream_t stream) {
  if (!p || !p->sr || !p->img_sr || p->H <= 0 || p->W <= 0 || (p->C != 1 && p->C != 3) || p->crop < 0 || !(p->hi > p->lo) ||
      (p->hr && (!p->img_hr || !p->out)) || (p->y_only && (p->C != 3 || !p->y_sr || (p->hr && !p->y_hr)))) {
    esr_set_error("esr_image_metrics: invalid arguments");
    return ESR_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(tensor2img_kernel, dim3((p->W + 255) / 256, p->H), dim3(256), 0, st, *p);
  if (p->hr) {
    const int ch = p->H - 2 * p->crop, cw = p->W - 2 * p->crop;
    if (ch <= 0 || cw <= 0) { esr_set_error("esr_image_metrics: crop leaves no pixels"); return ESR_ERR_INVALID; }
    if (hipMemsetAsync(p->out, 0, 4 * sizeof(double), st) != hipSuccess) { esr_set_error("esr_image_metrics: memset failed"); return ESR_ERR_LAUNCH; }
    const bool ssim = ch > 10 && cw > 10;
    if (p->y_only) {
      hipLaunchKernelGGL(sse_kernel<double>, dim3((cw + 255) / 256, ch), dim3(256), 0, st, *p, (const double*)p->y_sr, (const double*)p->y_hr, 1);
      if (ssim) hipLaunchKernelGGL(ssim_kernel<double>, dim3((cw - 10 + 255) / 256, ch - 10, 1), dim3(256), 0, st, *p, (const double*)p->y_sr, (const double*)p->y_hr, 1);
    } else {
      hipLaunchKernelGGL(sse_kernel<uint8_t>, dim3((cw + 255) / 256, ch), dim3(256), 0, st, *p, (const uint8_t*)p->img_sr, (const uint8_t*)p->img_hr, p->C);
      if (ssim) hipLaunchKernelGGL(ssim_kernel<uint8_t>, dim3((cw - 10 + 255) / 256, ch - 10, p->C), dim3(256), 0, st, *p, (const uint8_t*)p->img_sr, (const uint8_t*)p->img_hr, p->C);
    }
  }
  return esr_check_launch("esr_image_metrics");
}
