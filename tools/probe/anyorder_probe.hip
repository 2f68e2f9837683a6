// Does hipExtAnyOrderLaunch remove the barrier between two independent kernels of one stream on gfx950?  (hip_ext.h says the flag is
// not supported on GFX9xx for the module-launch form.)  Two kernels that each keep HALF of the CUs busy for ~T: serial 2T, overlapped ~T.
// build: hipcc --offload-arch=gfx950 -O2 -o anyorder_probe anyorder_probe.hip ; run: ./anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(float* p, int iters) {
  float a = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0000001f, 1e-9f);
  p[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
  float *a, *b;
  hipMalloc(&a, 1 << 22); hipMalloc(&b, 1 << 22);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, st);
      for (int k = 0; k < 8; ++k) {
        hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, a, iters);
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, b, iters);
        else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, b, iters);
        else hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, nullptr, nullptr, 0, b, iters);
      }
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s): 16 launches %.3f ms\n", mode, mode == 0 ? "plain" : (mode == 1 ? "any-order flag on every second" : "ext launch, no flag"), ms);
    }
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
