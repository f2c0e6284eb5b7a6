// Counter-based dropout shared by the streaming kernels and the fused GEMM epilogue (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vp3d.h"

namespace vp3d {

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: the dropout mask is a pure function of
// (seed, offset, layer, element index) and is regenerated in backward instead of being stored.
// ---------------------------------------------------------------------------------------------------------
struct DropP {
  float p, inv_keep;
  uint32_t k0, k1, off_lo, layer;
  int on;
  const uint64_t* off_ptr;   // optional device-side step counter added to the offset (hipGraph replays: the launch
                             // arguments are frozen, the counter is bumped inside the graph)
};

// call once at the top of a kernel (the by-value kernel argument is a private copy)
__device__ __forceinline__ void drop_resolve(DropP& d) {
  if (d.on && d.off_ptr != nullptr) d.off_lo += (uint32_t)*d.off_ptr;
}

__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                        uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // (one 32 x 32 -> 64 multiply each: v_mad_u64_u32; as __umulhi + a 32-bit product hipcc issues two quarter-rate multiplies)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep*scale factors of the 4 elements 4*q .. 4*q+3
__device__ __forceinline__ void drop4(const DropP& d, uint64_t q, float (&mk)[4]) {
  uint32_t r[4];
  philox4((uint32_t)q, (uint32_t)(q >> 32), d.layer, d.off_lo, d.k0, d.k1, r);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float u = (float)(r[e] >> 8) * (1.0f / 16777216.0f);
    mk[e] = (u >= d.p) ? d.inv_keep : 0.f;
  }
}


inline DropP make_drop(const vp3d_dropout* d) {
  DropP r;
  r.on = (d != nullptr && d->p > 0.f) ? 1 : 0;
  r.p = r.on ? d->p : 0.f;
  r.inv_keep = r.on ? 1.0f / (1.0f - d->p) : 1.f;
  r.k0 = r.on ? (uint32_t)d->seed : 0u;
  r.k1 = r.on ? (uint32_t)(d->seed >> 32) : 0u;
  r.off_lo = r.on ? (uint32_t)d->offset : 0u;
  r.layer = r.on ? d->layer : 0u;
  r.off_ptr = r.on ? d->offset_ptr : nullptr;
  return r;
}


}  // namespace vp3d
