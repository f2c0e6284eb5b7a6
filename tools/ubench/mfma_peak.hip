// Micro-benchmark: what does the f16 MFMA pipe sustain on random data, with and without a workgroup barrier every 24
// MFMAs (the structure of k_nt_s16's 128x128 configuration), at 1 or 2 workgroups per CU?   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BAR>
__global__ void __launch_bounds__(256, 2) k(const f16x8* __restrict__ in, float* out, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(threadIdx.x + i * 256) & 4095];
    b[i] = in[(threadIdx.x * 3 + i * 131 + blockIdx.x) & 4095];
  }
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + p + s) & 3], b[(j + 2 * p + s) & 3], acc[i][j], 0, 0, 0);
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int n = 4096;
  _Float16* h = (_Float16*)malloc(n * 16);
  for (int i = 0; i < n * 8; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  f16x8* d;
  float* o;
  hipMalloc(&d, n * 16);
  hipMalloc(&o, 4096 * 256 * 4);
  hipMemcpy(d, h, n * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int bar = 0; bar < 2; ++bar)
    for (int wg = 256; wg <= 512; wg += 256) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (bar) hipLaunchKernelGGL(k<1>, dim3(wg), dim3(256), 0, 0, d, o, iters);
        else hipLaunchKernelGGL(k<0>, dim3(wg), dim3(256), 0, 0, d, o, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("barrier=%d workgroups=%d: %.3f ms  %.0f TFLOP/s executed (%.0f fp32-equivalent at 3 MFMA/MAC)\n", bar, wg, ms,
                        (double)wg * 4 * iters * 24 * 32768.0 / ms / 1e9, (double)wg * 4 * iters * 24 * 32768.0 / ms / 1e9 / 3);
      }
    }
  return 0;
}
