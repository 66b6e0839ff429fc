// EXPERIMENT — not product code, not part of libesrgan_hip.so (see tools/experiments/README.md).
// mfma_power_probe.hip — what the matrix pipe sustains under the package power limit (round 5).
// Every SIMD of the chip runs ONE wave that issues independent v_mfma_f32_32x32x16_f16 back to back from registers
// (no LDS, no memory in the loop): the only limits are the MFMA issue rate and the clock the power management allows.
// Operands: zeros, or random fp16 (the toggle rate of real data).  The kernel also reads the shader clock counter
// (s_memtime) against the 100 MHz real-time counter, so the line says at which clock the rate was reached.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power_probe tools/experiments/mfma_power_probe.hip && /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void probe(const half8* __restrict__ ops, float* sink, uint64_t* clk, int iters) {
  __shared__ char pad[120 * 1024];                        // one workgroup per CU
  pad[threadIdx.x] = 0;
  const int lane = threadIdx.x;
  half8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = ops[(i * 256 + lane)]; b[i] = ops[((4 + i) * 256 + lane)]; }
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  uint64_t t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][lane & 15];
  if (s == 123.456f) sink[0] = s + pad[5];
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  half8* ops; float* sink; uint64_t* clk;
  hipMalloc(&ops, 8 * 256 * sizeof(half8)); hipMalloc(&sink, 64); hipMalloc(&clk, 16);
  const char* names[4] = {"zero", "N(0, 0.05)", "A N(0,.02) B N(0,1)", "random bits"};
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<_Float16> h(8 * 256 * 8);
    srand(7);
    for (size_t i = 0; i < h.size(); ++i) {
      float u = 0.f;
      if (mode == 1 || mode == 2) {
        for (int k = 0; k < 12; ++k) u += rand() / (float)RAND_MAX;      // ~N(0, 1)
        const bool is_a = i < h.size() / 2;
        u = (u - 6.f) * (mode == 1 ? 0.05f : (is_a ? 0.02f : 1.0f));     // (weights-like A, activation-like B)
      }
      h[i] = (_Float16)u;
      if (mode == 3) {                                                   // uniform bit patterns of finite values (|x| < 2)
        uint16_t bits = (uint16_t)(rand() & 0xBFFF);
        __builtin_memcpy(&h[i], &bits, 2);
      }
    }
    hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    constexpr int NACC = 16;
    const int iters = 200000;                      // 3.2 M MFMAs per wave and launch (~50 ms)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<NACC>, dim3(cus), dim3(256), 0, 0, ops, sink, clk, 1000);
    hipDeviceSynchronize();
    double total_ms = 0; int launches = 0; float ms = 0; uint64_t c[2] = {0, 0};
    while (total_ms < secs * 1000) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe<NACC>, dim3(cus), dim3(256), 0, 0, ops, sink, clk, iters);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); total_ms += ms; ++launches;
    }
    hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)cus * 4 * iters * NACC * 32768.0;
    printf("%-20s operands: %.1f TFLOP/s (last launch %.2f ms after %.1f s of load), shader clock %.3f GHz, MFMA issue %.1f cycles each\n",
           names[mode], flops / ms / 1e9, ms, total_ms / 1000, (double)c[0] / c[1] * 0.1,
           (double)c[0] / ((double)iters * NACC));
  }
  return 0;
}
