// The "second process" of tools/soak_two_process.py: keeps every CU of the GPU busy until a stop file appears.
//   spin   : bandwidth-bound elementwise kernels over 256 MiB (as the soak's earlier torch `x * a + b` loop)
//   matmul : compute-bound kernels (FP32 FMA chains, ~10 ms each: what a GEMM loop does to the CUs)
// A HIP program instead of a torch script: the first `import torch` on a fresh box takes up to two minutes.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/gpu_hog tools/gpu_hog.hip     usage: gpu_hog spin|matmul <stop file>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <sys/stat.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "gpu_hog: %s (line %d)\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_axpb(float* x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = x[i] * 1.0001f + 0.5f;
}
__global__ void k_fma(float* sink, int iters) {
  float a = (float)threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; i++) { a = a * b + c; d = d * b + a; c = c * 0.9999f + 1e-4f; b = b * 0.99999f + 1e-5f; }
  if (a + d == 12345.678f) sink[0] = a;
}
static bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: gpu_hog spin|matmul <stop file>\n"); return 2; }
  const bool spin = strcmp(argv[1], "spin") == 0;
  const std::string stop = argv[2];
  const size_t n = 64ull * 1024 * 1024;
  float* x; CK(hipMalloc(&x, n * sizeof(float))); CK(hipMemset(x, 0, n * sizeof(float)));
  CK(hipDeviceSynchronize());
  FILE* f = fopen((stop + ".ready").c_str(), "w"); if (f) { fputs("1", f); fclose(f); }
  long iters = 0;
  while (!exists(stop)) {
    if (spin) { for (int i = 0; i < 32; i++) hipLaunchKernelGGL(k_axpb, dim3(4096), dim3(256), 0, 0, x, n); }
    else { for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k_fma, dim3(8192), dim3(256), 0, 0, x, 60000); }
    CK(hipDeviceSynchronize());
    iters++;
  }
  printf("hog iterations %ld\n", iters);
  return 0;
}
