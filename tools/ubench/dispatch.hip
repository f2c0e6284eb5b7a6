// Micro-benchmark: workgroup dispatch cost of a 512-thread / 130 KiB-LDS workgroup (the 256x256 S16 GEMM configuration):
// how much of the ~18 us per-tile overhead is just getting a workgroup onto a CU?   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(512) k(float* out, int spin) {
  __shared__ float lds[130 * 256];
  lds[threadIdx.x] = (float)blockIdx.x;
  __syncthreads();
  float v = lds[(threadIdx.x + 1) & 511];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) out[blockIdx.x] = v;
}
int main() {
  float* o;
  (void)hipMalloc(&o, 1 << 20);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int spin : {0, 2000, 20000}) {
    for (int wg : {256, 3760, 15040}) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wg), dim3(512), 0, 0, o, spin);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("spin %6d  workgroups %6d: %8.3f ms  = %6.2f us per workgroup-slot round (256 CUs)\n", spin, wg, ms,
                        ms * 1e3 / ((wg + 255) / 256));
      }
    }
  }
  return 0;
}
