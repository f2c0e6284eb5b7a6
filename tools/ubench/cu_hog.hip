// "CU hog": N persistent workgroups on a queue of their own -- a stand-in for RCCL's channel kernels (one workgroup per
// channel, resident for the duration of a collective, copying at a few tens of GB/s each) -- so that the effect of such
// residents on the training step can be measured on ONE GPU (tools/cu_hog_ab.py).  Each workgroup (256 threads,
// 16 KiB of LDS like an RCCL channel's staging, few registers) records where it runs (XCC / SE / CU ids), then alternates a
// 64-KiB copy inside its own slice of `buf` with a short sleep until the host raises *stop (pinned, host-coherent) or
// max_ms of wall clock have passed -- it can never hang the GPU.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC cu_hog.hip -o libcuhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) k_cu_hog(const volatile int* stop, float* buf, int64_t slice_floats, int* where,
                                                unsigned long long max_ticks, int copy) {
  __shared__ float stage[4096];
  const unsigned long long t0 = wall_clock64();                 // 100 MHz
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf;  // HW_REG_XCC_ID
    where[blockIdx.x * 2 + 0] = (int)xcc;
    where[blockIdx.x * 2 + 1] = (int)hw;
  }
  float* mine = buf + (int64_t)blockIdx.x * slice_floats;
  int64_t off = 0;
  for (;;) {
    if (copy) {
#pragma unroll
      for (int i = 0; i < 16; ++i) stage[i * 256 + threadIdx.x] = mine[off + i * 256 + threadIdx.x];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[slice_floats / 2 + off + i * 256 + threadIdx.x] = stage[i * 256 + threadIdx.x] + 1.f;
      off += 4096;
      if (off + 4096 > slice_floats / 2) off = 0;
    }
    __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    if (*stop != 0 || wall_clock64() - t0 > max_ticks) break;
  }
}

extern "C" int cu_hog_launch(void* stream, int n_wg, const int* stop, float* buf, int64_t slice_floats, int* where, double max_ms,
                             int copy) {
  if (n_wg <= 0) return 0;
  hipLaunchKernelGGL(k_cu_hog, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, (const volatile int*)stop, buf, slice_floats, where,
                     (unsigned long long)(max_ms * 1e5), copy);
  return (int)hipGetLastError();
}
