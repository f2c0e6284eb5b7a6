// Probe: which LDS elements does ds_read_b64_tr_b16 (gfx950 transpose read) deliver to which lane?  Groundwork for a
// weight-gradient GEMM that reads the S16 ROWS of dy / x and transposes on the LDS read instead of consuming
// producer-written transposed copies.   hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
// LDS holds 16-bit values equal to their element index.  Experiment A: lane l reads at byte address 8*l (its own 4
// consecutive elements 4l..4l+3).  Experiment B: lane l reads row (l & 15) of a [16][pitch] image at column block
// (l >> 4): byte address (l & 15) * pitch_bytes + (l >> 4) * 8.  Printed: the 4 element indices every lane received.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) k(int mode, int pitch_bytes, int* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  const int byte = mode == 0 ? 8 * l : (l & 15) * pitch_bytes + (l >> 4) * 8;
  const uint32_t addr = (uint32_t)(uintptr_t)lds + byte;      // LDS aperture offset (low 32 bits of the shared pointer)
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (int)(uint16_t)v[j];
}

int main() {
  int* d;
  (void)hipMalloc(&d, 256 * sizeof(int));
  int h[256];
  const int modes[3][2] = {{0, 0}, {1, 32}, {1, 64}};
  for (auto& m : modes) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, m[0], m[1], d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d pitch %d B (lane: 4 element indices; address = element %s)\n", m[0], m[1],
           m[0] == 0 ? "4*lane" : "(lane&15)*pitch/2 + (lane>>4)*4");
    for (int l = 0; l < 64; ++l) printf("  l%02d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
  }
  return 0;
}
