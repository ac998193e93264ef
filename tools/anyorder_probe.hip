// Probe: can two kernels of ONE HIP stream run side by side on this stack?
//  (1) hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch): a waiter launched first, the setter launched behind it in the same
//      stream with the any-order flag — does the waiter see the flag before its bounded wait gives up?
//  (2) the same with two streams (control: must work).
//  (3) cost of a dependent boundary with and without the flag (chain of trivial kernels).
// Build: hipcc --offload-arch=gfx950 -O2 -o anyorder_probe tools/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(1))) unsigned int gu32;
__global__ void k_wait(unsigned int* flag, unsigned int* res, unsigned long long* clk) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();
  unsigned int spins = 0, seen = 2u;
  while (spins++ < 60000u) {      // ~10 ms
    if (__hip_atomic_load((gu32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { seen = 1u; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  res[0] = seen;
  clk[0] = wall_clock64() - t0;
}
__global__ void k_set(unsigned int* flag) { __hip_atomic_store((gu32*)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_touch(int* p) { if (threadIdx.x == 9999) p[0] = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  unsigned int* d; CK(hipMalloc(&d, 4096));
  unsigned long long* clk; CK(hipMalloc(&clk, 4096));
  hipStream_t st, st2; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  // warm-up (code upload)
  hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, d + 8); hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st, d + 8, d + 9, clk);
  hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st2, d + 8);
  CK(hipDeviceSynchronize());
  for (int mode = 0; mode < 4; mode++) {
    CK(hipMemset(d, 0, 64)); CK(hipDeviceSynchronize());
    unsigned int res = 0; unsigned long long c = 0;
    if (mode == 0) {            // same stream, plain launches: must time out
      hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st, d, d + 1, clk);
      hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, d);
    } else if (mode == 1) {     // same stream, setter any-order
      hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st, d, d + 1, clk);
      hipExtLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d);
    } else if (mode == 2) {     // same stream, both any-order
      hipExtLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d, d + 1, clk);
      hipExtLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d);
    } else {                    // two streams
      hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st, d, d + 1, clk);
      hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st2, d);
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&res, d + 1, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const char* names[4] = {"same stream, plain", "same stream, setter any-order", "same stream, both any-order", "two streams"};
    printf("mode %d (%s): waiter %s after %.1f us\n", mode, names[mode], res == 1 ? "SAW the flag" : "gave up", (double)c / 100.0);
  }
  // boundary cost: chain of N trivial kernels, plain vs any-order
  int* p = reinterpret_cast<int*>(d + 64);
  for (int grid : {1, 256}) {
    for (int mode = 0; mode < 2; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipStreamSynchronize(st));
        const int N = 2000;
        const double t0 = now();
        for (int i = 0; i < N; i++) {
          if (mode == 0) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, st, p);
          else hipExtLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, p);
        }
        CK(hipStreamSynchronize(st));
        const double t1 = now();
        if (rep) printf("chain of %d trivial kernels, grid %d, %s: %.2f us per kernel\n", N, grid, mode ? "any-order" : "plain", (t1 - t0) / N * 1e6);
      }
    }
  }
  return 0;
}
