// EXPERIMENT — not product code, not part of libesrgan_hip.so (see tools/experiments/README.md).
// Flag ping-pong between two workgroups of one launch: same XCD vs different XCDs, with the load policies a
// cross-workgroup hand-off could use on gfx950 (sc1 = device scope, sc0 = group scope, none = wave scope).
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_pingpong xcd_pingpong.hip ;  run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int POL> __device__ __forceinline__ unsigned ld(const unsigned* p) {
  unsigned v;
  if (POL == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (POL == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (POL == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (POL == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (POL == 4) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int SPOL> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
  if (SPOL == 0) asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
  if (SPOL == 1) asm volatile("global_store_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
  if (SPOL == 2) asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
  if (SPOL == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}

// flags[0] written by A, flags[64] by B (separate lines); payload[] written by A before each flag (checks that a
// same-policy load of DATA written just before the flag is fresh)
template <int POL, int SPOL>
__global__ void pingpong(unsigned* flags, unsigned* payload, int a, int b, int n, unsigned long long* out, unsigned* xcc) {
  const int id = blockIdx.x;
  if (threadIdx.x == 0) xcc[id] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
  if (threadIdx.x != 0 || (id != a && id != b)) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned bad = 0;
  for (int i = 1; i <= n; ++i) {
    if (id == a) {
      st<SPOL>(payload + (i & 15) * 64, (unsigned)i * 7u);
      st<SPOL>(flags, (unsigned)i);
      unsigned spins = 0;
      while (ld<POL>(flags + 64) < (unsigned)i) if (++spins > 20000000u) { out[2] = 1; return; }
    } else {
      unsigned spins = 0;
      while (ld<POL>(flags) < (unsigned)i) if (++spins > 20000000u) { out[2] = 1; return; }
      if (ld<POL>(payload + (i & 15) * 64) != (unsigned)i * 7u) ++bad;
      st<SPOL>(flags + 64, (unsigned)i);
    }
  }
  if (id == a) out[0] = __builtin_amdgcn_s_memrealtime() - t0;
  else out[1] = bad;
}

template <int POL, int SPOL> void run(const char* name, int a, int b, unsigned* flags, unsigned* payload, unsigned long long* out, unsigned* xcc) {
  hipMemset(flags, 0, 4096); hipMemset(payload, 0, 16 * 256); hipMemset(out, 0, 32);
  const int n = 2000;
  hipLaunchKernelGGL((pingpong<POL, SPOL>), dim3(16), dim3(64), 0, 0, flags, payload, a, b, n, out, xcc);
  hipDeviceSynchronize();
  unsigned long long h[4]; unsigned hx[16];
  hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost);
  printf("%-26s blocks %2d(xcc %u) <-> %2d(xcc %u): %7.3f us per round trip, stale payloads %llu%s\n", name, a, hx[a], b, hx[b],
         h[0] / 100.0 / n, h[1], h[2] ? "  TIMEOUT" : "");
}

int main() {
  unsigned *flags, *payload, *xcc; unsigned long long* out;
  hipMalloc(&flags, 4096); hipMalloc(&payload, 16 * 256); hipMalloc(&out, 32); hipMalloc(&xcc, 64);
  for (int pair = 0; pair < 2; ++pair) {
    const int a = 0, b = pair == 0 ? 8 : 1;       // round-robin dispatch: block 8 shares block 0's XCD, block 1 does not
    run<2, 2>("ld sc1 / st sc1", a, b, flags, payload, out, xcc);
    run<3, 3>("ld sc0sc1 / st sc0sc1", a, b, flags, payload, out, xcc);
    run<1, 2>("ld sc0 / st sc1", a, b, flags, payload, out, xcc);
    run<1, 1>("ld sc0 / st sc0", a, b, flags, payload, out, xcc);
    run<1, 0>("ld sc0 / st plain", a, b, flags, payload, out, xcc);
    run<4, 2>("ld nt / st sc1", a, b, flags, payload, out, xcc);
    run<0, 2>("ld plain / st sc1", a, b, flags, payload, out, xcc);
  }
  return 0;
}
