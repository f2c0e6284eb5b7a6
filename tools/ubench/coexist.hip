// Micro-benchmark: can an HBM-bound streaming kernel (<= 64 VGPRs per wave) run BESIDE a register-hungry MFMA kernel of another
// stream when that kernel leaves room in the register file?  The MFMA kernel holds 8 waves per CU (2 per SIMD) with NACC
// 16-register accumulators each (NACC = 13: ~216 VGPRs -> 2 x 216 = 432 of a SIMD's 512 registers, a 64-register wave fits;
// NACC = 15: ~248 VGPRs, nothing fits) and LDS_KB of LDS; one workgroup per CU, 256 workgroups, each running `iters` rounds
// (a long-lived workgroup, like a 256x256 weight-gradient tile).  The copy kernel streams `mb` MB once.
//   hipcc --offload-arch=gfx950 -O3 -o coexist tools/ubench/coexist.hip && ./coexist
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int LDS_KB>
__global__ void __launch_bounds__(512, 2) hog(const f16x8* __restrict__ in, float* out, int iters) {
  __shared__ char lds[LDS_KB * 1024];
  f16x8 a[2], b[2];
  for (int i = 0; i < 2; ++i) {
    a[i] = in[(threadIdx.x + i * 512) & 4095];
    b[i] = in[(threadIdx.x * 3 + i * 131 + blockIdx.x) & 4095];
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
    if ((it & 7) == 7) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) lds[threadIdx.x] = 1;      // (keeps the LDS allocation)
  out[blockIdx.x * 512 + threadIdx.x] = s + (float)lds[(threadIdx.x * 7) & (LDS_KB * 1024 - 1)];
}

__global__ void __launch_bounds__(256) copy4(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i] * 1.0001f;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC, int LDS_KB>
void run(const char* name, const f16x8* in, float* out, const f32x4* src, f32x4* dst, long n4, hipStream_t s1, hipStream_t s2) {
  hipEvent_t e0, e1, c0, c1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
  const int iters = 60000;                      // ~ 1.3 ms of MFMAs per workgroup
  float t_h = 0, t_c = 0, t_hb = 0, t_cb = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL((hog<NACC, LDS_KB>), dim3(256), dim3(512), 0, s1, in, out, iters);
    CK(hipEventRecord(e1, s1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&t_h, e0, e1));
    CK(hipEventRecord(c0, s2));
    hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, s2, src, dst, n4);
    CK(hipEventRecord(c1, s2));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&t_c, c0, c1));
    // both: the hog first (it owns every CU), the copy 100 us later on the other stream
    CK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL((hog<NACC, LDS_KB>), dim3(256), dim3(512), 0, s1, in, out, iters);
    CK(hipEventRecord(e1, s1));
    CK(hipEventRecord(c0, s2));
    hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, s2, src, dst, n4);
    CK(hipEventRecord(c1, s2));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&t_hb, e0, e1));
    CK(hipEventElapsedTime(&t_cb, c0, c1));
  }
  int nv = 0;
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)hog<NACC, LDS_KB>));
  nv = fa.numRegs;
  printf("%-28s %3d VGPRs  alone: MFMA kernel %7.1f us, copy %6.1f us (%.2f TB/s)   together: MFMA kernel %7.1f us, copy %7.1f us\n",
         name, nv, t_h * 1e3, t_c * 1e3, n4 * 32.0 / (t_c * 1e-3) / 1e12, t_hb * 1e3, t_cb * 1e3);
}

int main() {
  f16x8* in; float* out; f32x4 *src, *dst;
  const long n4 = 170l * 1024 * 1024 / 16;       // 170 MB read + 170 MB written
  CK(hipMalloc(&in, 4096 * sizeof(f16x8))); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16));
  CK(hipMemset(in, 0x3c, 4096 * sizeof(f16x8))); CK(hipMemset(src, 0, n4 * 16));
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  run<15, 130>("2 x 248 VGPRs, 130 KB LDS", in, out, src, dst, n4, s1, s2);
  run<13, 130>("2 x 216 VGPRs, 130 KB LDS", in, out, src, dst, n4, s1, s2);
  run<11, 130>("2 x 184 VGPRs, 130 KB LDS", in, out, src, dst, n4, s1, s2);
  run<11, 64>("2 x 184 VGPRs,  64 KB LDS", in, out, src, dst, n4, s1, s2);
  return 0;
}
