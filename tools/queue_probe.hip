// Probe: which HIP streams of one process share a hardware queue on this stack?  For every ordered pair (A, B) of N streams a
// one-wave waiter is launched on A and a setter on B: if B's launch is queued behind A's in a shared hardware queue the waiter
// gives up (bounded wait).  Stream set: as liodom_create makes them (two at the highest priority, one a level below), then
// further streams with default priority / flags.  Run with GPU_MAX_HW_QUEUES unset and set.
// Build: hipcc --offload-arch=gfx950 -O2 -o queue_probe tools/queue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(1))) unsigned int gu32;
__global__ void k_wait(unsigned int* flag, unsigned int* res) {
  if (threadIdx.x != 0) return;
  unsigned int spins = 0, seen = 2u;
  while (spins++ < 6000u) {      // ~1 ms
    if (__hip_atomic_load((gu32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { seen = 1u; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  res[0] = seen;
}
__global__ void k_set(unsigned int* flag) { __hip_atomic_store((gu32*)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 6;
  const int scheme = argc > 2 ? atoi(argv[2]) : 0;      // 0: liodom priorities then default; 1: all default priority; 2: all highest priority
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  printf("streams %d, scheme %d, GPU_MAX_HW_QUEUES=%s\n", N, scheme, e ? e : "(unset)");
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("priority range: least %d greatest %d\n", lo, hi);
  std::vector<hipStream_t> st(N);
  for (int i = 0; i < N; i++) {
    int prio = 0; bool with_prio = false;
    if (scheme == 0) { if (i == 0 || i == 2) { prio = hi; with_prio = true; } else if (i == 1) { prio = (lo - hi >= 2) ? hi + 1 : lo; with_prio = true; } }
    if (scheme == 2) { prio = hi; with_prio = true; }
    if (with_prio) CK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, prio));
    else CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  }
  unsigned int* d; CK(hipMalloc(&d, 4096));
  for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st[i], d + 512); }      // warm every queue
  CK(hipDeviceSynchronize());
  printf("rows: waiter stream, columns: setter stream; '.' = ran side by side, 'X' = setter stuck behind the waiter (shared queue)\n");
  for (int a = 0; a < N; a++) {
    printf("  %d: ", a);
    for (int b = 0; b < N; b++) {
      if (a == b) { printf("- "); continue; }
      CK(hipMemset(d, 0, 64)); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st[a], d, d + 1);
      hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st[b], d);
      CK(hipDeviceSynchronize());
      unsigned int res = 0; CK(hipMemcpy(&res, d + 1, 4, hipMemcpyDeviceToHost));
      printf("%c ", res == 1 ? '.' : 'X');
    }
    printf("\n");
  }
  // the null stream and a copy: does a hipMemcpyAsync on stream b run beside a waiter on stream a?
  return 0;
}
