// EXPERIMENT — not product code, not part of libesrgan_hip.so (see tools/experiments/README.md).
// issue_probe.hip — what one wave per SIMD pays per MFMA for the instructions between its MFMAs (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/issue_probe.hip -o gpurun_out/issue_probe
// Every variant: 128 workgroups x 4 waves, N MFMAs (v_mfma_f32_32x32x16_f16) per wave; prints ns and shader cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int I> struct IC { static constexpr int value = I; };
template <int N, typename F, int... Is> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl<N>(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// MODE: 0 = NACC independent accumulators round robin, nothing else
//       1 = + one ds_read_b128 behind every MFMA           2 = + two
//       3 = + one LDS-DMA (1 KB) behind every DMAEVERY-th MFMA (global_load_lds_dwordx4)
//       4 = + s_barrier every 8 MFMAs                       5 = 1 + 3 together
//       9 = every MFMA reads its own A fragment register (6 of them), refilled by a ds_read_b128 issued DMAEVERY MFMAs
//           behind its reader (0 = right behind it: write-after-read on an MFMA source that is still being read?)
//       7 = one LDS-DMA per DMAEVERY MFMAs STREAMING through a 32 MB region (every workgroup the same addresses, like the
//           chain's weight stream), 6 KB of LDS ring per wave, vmcnt(12) every 16 MFMAs;  8 = 7 + one ds_read per MFMA
template <int MODE, int NACC, int DMAEVERY>
__global__ __launch_bounds__(256) void probe(const char* w, uint64_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  u32x4 a = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a, r0 = a, r1 = a;
  u32x4 fa[6] = {a, a, a, a, a, a};
  const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + threadIdx.x * 16;
  const char* src = w + (size_t)blockIdx.x * 4096 + threadIdx.x * 16;
  char* dst = smem + 32 * 1024 + wave * 1024;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  const char* stream = w + wave * 1024 + lane * 16;
  for (int it = 0; it < iters; ++it) {
    if ((MODE == 7 || MODE == 8) && (it & 63) == 0) stream = w + wave * 1024 + lane * 16;     // 64 x 48 / DMAEVERY x 4 KB
    sfor<48>([&](auto MI) __attribute__((always_inline)) {
      constexpr int m = decltype(MI)::value;
      if constexpr (MODE == 9) {
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[m % 6]));
        mma(acc[m % NACC], fa[m % 6], b);
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[(m + 6 - DMAEVERY) % 6]) : "v"(lds));
      } else
      mma(acc[m % NACC], a, b);
      if constexpr (MODE == 1 || MODE == 2 || MODE == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(lds));
      if constexpr (MODE == 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(r1) : "v"(lds));
      if constexpr ((MODE == 3 || MODE == 5) && m % DMAEVERY == 0) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, (m % 4) * 1024, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (MODE == 7 || MODE == 8) {
        if constexpr (MODE == 8) asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(lds));
        if constexpr (m % DMAEVERY == 0) {
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)stream, (__attribute__((address_space(3))) void*)dst, 16, (m % 4) * 1024, 0);
          __builtin_amdgcn_sched_barrier(0);
          stream += 4096;
        }
        if constexpr (m % 16 == 15) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        if constexpr (MODE == 8 && m % 12 == 11) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1));
      }
      if constexpr (MODE == 4 && m % 8 == 7) __builtin_amdgcn_s_barrier();
      if constexpr ((MODE == 1 || MODE == 2 || MODE == 5) && m % 12 == 11) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1));
      if constexpr ((MODE == 3 || MODE == 5) && m % 24 == 23) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    });
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 123.f && r0[0] == 7 && r1[0] == 9 && fa[0][0] + fa[1][0] + fa[2][0] + fa[3][0] + fa[4][0] + fa[5][0] == 1) out[blockIdx.x] = 0;
}

template <int MODE, int NACC, int DMAEVERY = 1> void run(const char* name, const char* w, uint64_t* out) {
  const int iters = 200, grid = 128;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, NACC, DMAEVERY>), dim3(grid), dim3(256), 0, 0, w, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, NACC, DMAEVERY>), dim3(grid), dim3(256), 0, 0, w, out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(grid);
  hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= grid;
  const double n = 48.0 * iters;
  printf("%-58s %7.2f ns / MFMA   %7.1f s_memtime ticks / MFMA\n", name, ms * 1e6 / n, cyc / n);
}


// MODE 10: the 4-row chain's inner loop in miniature — per MFMA: its own A fragment (refilled in place), every 4th
// MFMA a streaming 1 KB LDS-DMA, a counted wait and EXTRA vector instructions (address code, spill reloads) — run by 4
// waves (one per SIMD) or by 8 waves (two per SIMD, each half the MFMAs): what a second wave per SIMD would hide.
template <int NW, int EXTRA>
__global__ __launch_bounds__(NW * 64) void probe2(const char* w, uint64_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[96 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  u32x4 b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  u32x4 fa[6] = {b, b, b, b, b, b};
  const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 255) * 16;
  char* dst = smem + 64 * 1024 + wave * 1024;
  const char* stream = w + wave * 1024 + lane * 16;
  int x = lane;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    if ((it & 63) == 0) stream = w + wave * 1024 + lane * 16;
    sfor<48>([&](auto MI) __attribute__((always_inline)) {
      constexpr int m = decltype(MI)::value;
      asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[m % 6]));
      mma(acc[m % 3], fa[m % 6], b);
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[m % 6]) : "v"(lds), "n"((m % 8) * 4096));
      if constexpr (m % 4 == 0) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)stream, (__attribute__((address_space(3))) void*)dst, 16, (m % 4) * 1024, 0);
        __builtin_amdgcn_sched_barrier(0);
        stream += NW * 1024;
      }
#pragma unroll
      for (int e = 0; e < EXTRA; ++e) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x));
      if constexpr (m % 16 == 15) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      if constexpr (m % 12 == 11) __builtin_amdgcn_s_barrier();
    });
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = acc[0][0] + acc[1][0] + acc[2][0];
  if (s == 123.f && x == 7 && fa[0][0] + fa[1][0] + fa[2][0] + fa[3][0] + fa[4][0] + fa[5][0] == 1) out[blockIdx.x] = 0;
}
template <int NW, int EXTRA> void run2(const char* name, const char* w, uint64_t* out) {
  const int iters = 200 * 4 / NW, grid = 128;          // the same MFMAs per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe2<NW, EXTRA>), dim3(grid), dim3(NW * 64), 0, 0, w, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe2<NW, EXTRA>), dim3(grid), dim3(NW * 64), 0, 0, w, out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double n = 48.0 * 200;                          // MFMAs per SIMD
  printf("%-58s %7.2f ns / MFMA of the SIMD\n", name, ms * 1e6 / n);
}

int main() {
  char* w; uint64_t* out;
  hipMalloc(&w, 64 << 20); hipMemset(w, 0, 64 << 20); hipMalloc(&out, 8 * 1024);
  run<0, 1>("1 accumulator (dependent chain)", w, out);
  run<0, 2>("2 accumulators", w, out);
  run<0, 3>("3 accumulators", w, out);
  run<0, 4>("4 accumulators", w, out);
  run<0, 6>("6 accumulators", w, out);
  run<1, 6>("6 acc + 1 ds_read_b128 / MFMA", w, out);
  run<2, 6>("6 acc + 2 ds_read_b128 / MFMA", w, out);
  run<3, 6, 1>("6 acc + 1 LDS-DMA / MFMA", w, out);
  run<3, 6, 2>("6 acc + 1 LDS-DMA / 2 MFMA", w, out);
  run<3, 6, 4>("6 acc + 1 LDS-DMA / 4 MFMA", w, out);
  run<4, 6>("6 acc + s_barrier / 8 MFMA", w, out);
  run<5, 6, 4>("6 acc + 1 ds_read / MFMA + 1 LDS-DMA / 4 MFMA", w, out);
  run<5, 6, 1>("6 acc + 1 ds_read / MFMA + 1 LDS-DMA / MFMA", w, out);
  run<5, 1, 1>("1 acc + 1 ds_read / MFMA + 1 LDS-DMA / MFMA", w, out);
  run<7, 6, 4>("6 acc + streaming LDS-DMA / 4 MFMA (3 MB per 64 iterations)", w, out);
  run<7, 6, 2>("6 acc + streaming LDS-DMA / 2 MFMA", w, out);
  run<8, 6, 4>("6 acc + ds_read / MFMA + streaming LDS-DMA / 4 MFMA", w, out);
  run<9, 6, 0>("6 acc, A refilled right behind its reader", w, out);
  run<9, 6, 1>("6 acc, A refilled 1 MFMA behind its reader", w, out);
  run<9, 6, 2>("6 acc, A refilled 2 MFMAs behind its reader", w, out);
  run<9, 6, 3>("6 acc, A refilled 3 MFMAs behind its reader", w, out);
  run2<4, 0>("inner loop, 4 waves, no extra instructions", w, out);
  run2<8, 0>("inner loop, 8 waves (2 per SIMD), no extra", w, out);
  run2<4, 3>("inner loop, 4 waves, 3 extra VALU per MFMA", w, out);
  run2<8, 3>("inner loop, 8 waves, 3 extra VALU per MFMA", w, out);
  run2<4, 6>("inner loop, 4 waves, 6 extra VALU per MFMA", w, out);
  run2<8, 6>("inner loop, 8 waves, 6 extra VALU per MFMA", w, out);
  return 0;
}
