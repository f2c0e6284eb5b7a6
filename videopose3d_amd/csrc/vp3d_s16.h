// "S16" split-fp16 operand format shared by the split-precision GEMM kernels and the streaming kernels that
// produce their operands (not part of the C ABI; the format itself is documented in include/vp3d.h).
//
// An fp32 value v of a tensor with per-tensor exponent e is kept as two fp16 numbers
//     hi = fp16(v * 2^-e),   lo = fp16(v * 2^-e - hi)          v ~= 2^e * (hi + lo)
// i.e. 22-23 significant bits.  Eight consecutive elements of a row are stored as 16 B of hi followed by 16 B
// of lo, so an S16 row has exactly the byte geometry of the fp32 row it replaces (4 B per element, 32-B groups)
// and every 16-B chunk is one MFMA operand fragment (8 consecutive k of one row).
// The GEMMs evaluate a*b ~= ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_f16 with fp32 accumulation: every
// fp16 product is exact in fp32 and the dropped al*bl term is <= 2^-22 |a*b|: fp32-class results at the fp16
// matrix rate (3 MFMAs per 16-deep step instead of 8 fp32 MFMAs).
// The exponent keeps the tensor inside fp16's range: producers choose it from a guaranteed bound of max|v| so
// that |v * 2^-e| < 2^15; elements below 2^-14 relative to that carry an absolute error <= 2^-25 (scaled units).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vp3d {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// exponent e such that |v| <= bound  =>  |v * 2^-e| < 2^15   (bound = 0 or non-finite -> 0)
__device__ __forceinline__ int s16_exp_for_bound(float bound) {
  if (!(bound > 0.f) || !(bound < 3.0e38f)) return 0;
  int ex;
  frexpf(bound, &ex);            // bound = m * 2^ex, m in [0.5, 1)  ->  bound < 2^ex
  return ex - 15;
}

// A "bound" is VP3D_BOUND_SLOTS consecutive floats whose maximum is the bound: kernels that measure a maximum spread
// their atomicMax over the slots (one per workgroup, slot = workgroup id % slots) instead of hammering one address;
// computed bounds are written to slot 0 of a zeroed array.  Every lane of the calling wave gets the result.
constexpr int kBoundSlots = 32;
__device__ __forceinline__ float s16_load_bound(const float* __restrict__ b) {
  float m = b[threadIdx.x & (kBoundSlots - 1)];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}
__device__ __forceinline__ int s16_exp_of(const float* __restrict__ b) { return s16_exp_for_bound(s16_load_bound(b)); }
__device__ __forceinline__ void s16_atomic_bound(float* b, float v) {
  atomicMax(reinterpret_cast<int*>(b + (blockIdx.x & (kBoundSlots - 1))), __float_as_int(v));
}

typedef float f32x8_t __attribute__((ext_vector_type(8)));
// (as vector conversions: hipcc selects v_cvt_pk_f16_f32 -- two elements per instruction, round to nearest even like the
//  scalar conversion -- and needs no v_pack: 24 instead of 40 VALU operations per 8 elements)
__device__ __forceinline__ void s16_split8(const float (&v)[8], float inv_scale, f16x8& hi, f16x8& lo) {
  f32x8_t x;
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = v[j] * inv_scale;
  hi = __builtin_convertvector(x, f16x8);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x8_t), f16x8);
}

__device__ __forceinline__ void s16_join8(const f16x8& hi, const f16x8& lo, float scale, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = ((float)hi[j] + (float)lo[j]) * scale;
}

// n / d for 0 <= n < 2^31, 1 <= d < 2^31, as one 32 x 32 -> 64 multiply and a shift (the row -> (sample, frame) split of the
// streaming kernels: a 64-bit integer division is ~120 VALU instructions, as much as the rest of a row's work).
// s = ceil(log2 d), mul = ceil(2^(31+s) / d) <= 2^32 - 1:  n * mul / 2^(31+s) = n/d + n*e/2^(31+s) with 0 <= e < 1, and
// n*e / 2^(31+s) < 2^-s <= 1/d, so the floor is exact.
struct FastDiv {
  uint32_t mul, shift, d;
};
inline FastDiv make_fastdiv(int64_t d_) {
  FastDiv f;
  const uint64_t d = d_ > 0 ? (uint64_t)d_ : 1;
  uint32_t s = 0;
  while (((uint64_t)1 << s) < d) ++s;
  f.shift = 31 + s;
  f.mul = (uint32_t)((((uint64_t)1 << f.shift) + d - 1) / d);
  f.d = (uint32_t)d;
  return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv& f) { return (uint32_t)(((uint64_t)n * f.mul) >> f.shift); }

__device__ __forceinline__ float s16_pow2(int e) { return ldexpf(1.0f, e); }

// "activation bits": one byte per (row, 8 consecutive channels), bit e = [bn(y) > 0 and the element was kept by the
// dropout]: written by the forward producers (vp3d_bn_act_fwd_s16, or the fused-activation epilogue of the GEMM), read
// by the backward passes instead of regenerating the Philox mask (10 integer multiplies per element and pass).  Stored
// per 64-channel tile so that a block's bytes are contiguous: byte of (row m, channels c..c+7) at
// ((c / 64) * M + m) * 8 + (c % 64) / 8.
__device__ __forceinline__ int64_t act_bits_index(int c, int64_t m, int M) {
  return ((int64_t)(c >> 6) * M + m) * 8 + ((c & 63) >> 3);
}

// Ticket of a block that has finished writing its contribution: true for the block that arrives last.  Publish / consume
// follow the agent-scope release -> relaxed atomic -> acquire hand-off (all stores of the block drained and released
// before the ticket; the last arriver acquires before any thread of it reads the other blocks' rows).
__device__ __forceinline__ bool last_arriver(int* counter, int expected, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == expected - 1) ? 1 : 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// s16_load_bound of a bound whose slots other workgroups of the SAME launch have just atomicMax-ed (read behind a
// last_arriver hand-over): agent-scope loads, so that no stale line of this XCD's L2 is consulted
__device__ __forceinline__ float s16_load_bound_agent(const float* b) {
  float m = __hip_atomic_load(b + (threadIdx.x & (kBoundSlots - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}

}  // namespace vp3d
