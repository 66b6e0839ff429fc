/* ORACLE (test infrastructure, NOT product code) — plain-C restatement of the tensor primitives
 * the reference's hot path is made of.  The reference calls them through stock torch.nn modules
 * (third-party: PyTorch >= 1.0, README.md:20; no lock file), so this file restates the
 * *published* semantics of those modules, NCHW fp32:
 *   conv2d       nn.Conv2d(k, stride, padding=(k-1)//2 zero pad)   block.py:125-142, 55-58
 *   leaky_relu   nn.LeakyReLU(0.2)                                 block.py:12-21
 *   upsample2    nn.Upsample(scale_factor=2, mode='nearest')       block.py:315-322
 *   noise        x + z*(sigma*x)                                   block.py:117-122
 *   batchnorm    nn.BatchNorm2d(affine) train/eval                 block.py:28-32
 *   maxpool2     nn.MaxPool2d(2,2)  (torchvision vgg19 cfg 'E')    architecture.py:287-298
 *   linear       nn.Linear                                         architecture.py:122-123
 * Accumulation is in double so the result is order-independent to ~1e-7 relative; it is the
 * independent cross-check for oracle/ref_torch.py (tests/test_oracle.py) and is never linked or
 * imported by the product.  Build: make -C oracle   (gcc -O3 -fopenmp -shared).
 */
#include <math.h>
#include <stddef.h>

#define EXPORT __attribute__((visibility("default")))

EXPORT void ref_conv2d(const float *x, const float *w, const float *b, float *y, int N, int C,
                       int H, int W, int K, int ks, int stride, int pad) {
  const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k)
      for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox) {
          double acc = b ? (double)b[k] : 0.0;
          for (int c = 0; c < C; ++c)
            for (int r = 0; r < ks; ++r) {
              const int iy = oy * stride - pad + r;
              if (iy < 0 || iy >= H) continue;
              for (int s = 0; s < ks; ++s) {
                const int ix = ox * stride - pad + s;
                if (ix < 0 || ix >= W) continue;
                acc += (double)x[(((size_t)n * C + c) * H + iy) * W + ix] *
                       (double)w[(((size_t)k * C + c) * ks + r) * ks + s];
              }
            }
          y[(((size_t)n * K + k) * Ho + oy) * Wo + ox] = (float)acc;
        }
}

EXPORT void ref_leaky_relu(float *x, size_t n, float slope) {
  for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : x[i] * slope;
}

EXPORT void ref_upsample2(const float *x, float *y, int NC, int H, int W) {
  for (int p = 0; p < NC; ++p)
    for (int oy = 0; oy < 2 * H; ++oy)
      for (int ox = 0; ox < 2 * W; ++ox)
        y[((size_t)p * 2 * H + oy) * 2 * W + ox] = x[((size_t)p * H + oy / 2) * W + ox / 2];
}

/* y = a*alpha + b (elementwise), e.g. x5*0.2 + x (block.py:268) */
EXPORT void ref_axpb(const float *a, float alpha, const float *b, float *y, size_t n) {
  for (size_t i = 0; i < n; ++i) y[i] = a[i] * alpha + b[i];
}

/* GaussianNoise in train mode: x + z*(sigma*x) (block.py:117-122) */
EXPORT void ref_noise(float *x, const float *z, float sigma, size_t n) {
  for (size_t i = 0; i < n; ++i) x[i] = x[i] + z[i] * (sigma * x[i]);
}

/* BatchNorm2d, affine.  training: normalise with biased batch variance; running stats updated
 * with momentum using the UNBIASED variance.  eval: use running stats. */
EXPORT void ref_batchnorm(float *x, int N, int C, int HW, const float *gamma, const float *beta,
                          float *rmean, float *rvar, int training, float momentum, float eps) {
  for (int c = 0; c < C; ++c) {
    double mean, var;
    if (training) {
      double s = 0, ss = 0;
      const double cnt = (double)N * HW;
      for (int n = 0; n < N; ++n)
        for (int i = 0; i < HW; ++i) s += x[((size_t)n * C + c) * HW + i];
      mean = s / cnt;
      for (int n = 0; n < N; ++n)
        for (int i = 0; i < HW; ++i) {
          const double d = x[((size_t)n * C + c) * HW + i] - mean;
          ss += d * d;
        }
      var = ss / cnt;
      rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
      rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * (ss / (cnt - 1.0)));
    } else {
      mean = rmean[c];
      var = rvar[c];
    }
    const double inv = 1.0 / sqrt(var + (double)eps);
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < HW; ++i) {
        float *p = &x[((size_t)n * C + c) * HW + i];
        *p = (float)((*p - mean) * inv * gamma[c] + beta[c]);
      }
  }
}

EXPORT void ref_maxpool2(const float *x, float *y, int NC, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  for (int p = 0; p < NC; ++p)
    for (int oy = 0; oy < Ho; ++oy)
      for (int ox = 0; ox < Wo; ++ox) {
        const float *q = &x[((size_t)p * H + 2 * oy) * W + 2 * ox];
        float m = q[0];
        if (q[1] > m) m = q[1];
        if (q[W] > m) m = q[W];
        if (q[W + 1] > m) m = q[W + 1];
        y[((size_t)p * Ho + oy) * Wo + ox] = m;
      }
}

EXPORT void ref_linear(const float *x, const float *w, const float *b, float *y, int N, int I,
                       int O) {
  for (int n = 0; n < N; ++n)
    for (int o = 0; o < O; ++o) {
      double acc = b ? (double)b[o] : 0.0;
      for (int i = 0; i < I; ++i) acc += (double)x[(size_t)n * I + i] * (double)w[(size_t)o * I + i];
      y[(size_t)n * O + o] = (float)acc;
    }
}
