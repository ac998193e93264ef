// Microbenchmark: cost of dependent kernel boundaries and of host wait mechanisms on this box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_empty(int* p) { if (threadIdx.x == 9999) p[0] = 1; }
__global__ void k_dep(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= p[0]) return; p[1 + (i % n)] += 1; }
struct Big { double a[40]; void* p[30]; int x[20]; };
__global__ void k_bigarg(Big b, int* p) { if ((int)(blockIdx.x * blockDim.x + threadIdx.x) >= p[0]) return; p[1] = (int)b.a[3]; }
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipStream_t st2; CK(hipStreamCreate(&st2));
  const int N = 2000;
  for (int grid : {1, 64, 256, 440, 2048}) {
    for (int which = 0; which < 3; which++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipStreamSynchronize(st));
        double t0 = now();
        Big b{}; 
        for (int i = 0; i < N; i++) {
          if (which == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, d);
          else if (which == 1) hipLaunchKernelGGL(k_dep, dim3(grid), dim3(256), 0, st, d, 1000);
          else hipLaunchKernelGGL(k_bigarg, dim3(grid), dim3(256), 0, st, b, d);
        }
        double t1 = now();
        CK(hipStreamSynchronize(st));
        double t2 = now();
        if (rep == 1) printf("grid %4d kernel %d: enqueue %.2f us/launch, total %.2f us/launch\n", grid, which, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
      }
    }
  }
  // graph of 13 kernels
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 13; i++) hipLaunchKernelGGL(k_dep, dim3(440), dim3(256), 0, st, d, 1000);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipStreamSynchronize(st));
      double t0 = now();
      for (int i = 0; i < 200; i++) CK(hipGraphLaunch(ge, st));
      double t1 = now();
      CK(hipStreamSynchronize(st));
      double t2 = now();
      if (rep) printf("graph(13 x k_dep grid 440): enqueue %.2f us/graph, total %.2f us/graph = %.2f us/kernel\n", (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6, (t2 - t0) / 200 / 13 * 1e6);
    }
  }
  // host wait mechanisms: 1 kernel + D2H 512 B to pinned + wait
  void* pin; CK(hipHostMalloc(&pin, 4096, hipHostMallocDefault));
  hipEvent_t ev, evb; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evb, hipEventDisableTiming | hipEventBlockingSync));
  for (int mode = 0; mode < 5; mode++) {
    double t0 = now();
    const int M = 300;
    for (int i = 0; i < M; i++) {
      hipLaunchKernelGGL(k_dep, dim3(64), dim3(256), 0, st, d, 1000);
      if (mode < 4) CK(hipMemcpyAsync(pin, d, 512, hipMemcpyDeviceToHost, st));
      if (mode == 0) { CK(hipEventRecord(ev, st)); CK(hipEventSynchronize(ev)); }
      else if (mode == 1) { CK(hipEventRecord(ev, st)); while (hipEventQuery(ev) == hipErrorNotReady) {} }
      else if (mode == 2) { CK(hipStreamSynchronize(st)); }
      else if (mode == 3) { CK(hipEventRecord(evb, st)); CK(hipEventSynchronize(evb)); }
      else { // mode 4: kernel writes directly to host-mapped memory, host polls a flag (no memcpy)
        CK(hipStreamSynchronize(st));
      }
    }
    double t1 = now();
    const char* names[] = {"memcpyAsync+eventSynchronize", "memcpyAsync+eventQuery spin", "memcpyAsync+streamSynchronize", "memcpyAsync+blocking event", "kernel only+streamSynchronize"};
    printf("wait mode %d (%s): %.2f us/iter\n", mode, names[mode], (t1 - t0) / M * 1e6);
  }
  // zero-copy: kernel writes into pinned host memory, host spins on it
  {
    volatile int* hp = (volatile int*)pin; int* dp; CK(hipHostGetDevicePointer((void**)&dp, pin, 0));
    hp[0] = 0x7fffffff; hp[1] = 0;
    double t0 = now();
    const int M = 300;
    for (int i = 0; i < M; i++) {
      int before = hp[1];
      hipLaunchKernelGGL(k_dep, dim3(1), dim3(64), 0, st, dp, 1);
      while (hp[1] == before) {}
    }
    double t1 = now();
    printf("zero-copy host poll: %.2f us/iter\n", (t1 - t0) / M * 1e6);
  }
  return 0;
}
