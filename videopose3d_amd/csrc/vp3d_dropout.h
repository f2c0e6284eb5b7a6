// Counter-based dropout shared by the streaming kernels and the fused GEMM epilogue (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vp3d.h"

namespace vp3d {

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: the dropout mask is a pure function of
// (seed, offset, layer, element index): backward regenerates it (fp32 engine) or reads the forward's activation bits
// (split-fp16 engine); it is never stored as a tensor.
// ---------------------------------------------------------------------------------------------------------
struct DropP {
  float p, inv_keep;
  uint32_t thr;              // an element is dropped when its 16 random bits are < thr = round(p * 65536)
  uint32_t thr2;             // "two" mode: dropped when its 2 random bits are < thr2 = thr / 16384
  uint32_t k0, k1, off_lo, layer;
  int on;
  int two;                   // thr is a multiple of 16384 (p = 0.25 -- run.py's default --, 0.5, 0.75): 2 random bits per element
                             // decide exactly, so ONE Philox block serves 64 elements instead of 8
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

// The mask of 8 elements from their 16 random bits ("two" mode: element e takes bits 2e, 2e+1).
__device__ __forceinline__ void drop8_from_bits16(const DropP& d, uint32_t h16, float (&mk)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) mk[e] = (((h16 >> (2 * e)) & 3u) >= d.thr2) ? d.inv_keep : 0.f;
}

// One Philox block (128 bits) serves
//   * 16-bit mode (any p): the 8 elements 8*q8 .. 8*q8+7, 16 random bits each -- element e takes the low (e even) or high (e odd)
//     half of word e / 2.  The keep probability is 1 - round(p * 65536) / 65536, within 7.6e-6 of 1 - p.  (Round 1 spent a
//     block per 4 elements, 24 bits each: the 20 multiplies of a block were half of the VALU work of the activation epilogues.)
//   * "two" mode (round(p * 65536) a multiple of 16384: p = 0.25 / 0.5 / 0.75, exact): the 64 elements 64*q64 .. 64*q64+63, 2 bits
//     each -- the 8-element group q8 takes the 16 bits (q8 & 7) of block q8 >> 3: low / high half of word (q8 & 7) >> 1.  A kernel
//     whose wave covers whole 64-element blocks evaluates each block once and hands the 16-bit pieces round with shuffles
//     (k_expand_fwd_s16: a quarter of the Philox work of its VALU-bound activation pass); everybody else evaluates the block
//     of its group -- the same mask either way, one definition for both engines and for vp3d_dropout_mask.
__device__ __forceinline__ void drop8(const DropP& d, uint64_t q8, float (&mk)[8]) {
  uint32_t r[4];
  if (d.two) {
    const uint64_t q64 = q8 >> 3;
    philox4((uint32_t)q64, (uint32_t)(q64 >> 32), d.layer, d.off_lo, d.k0, d.k1, r);
    const uint32_t sub = (uint32_t)q8 & 7u;
    const uint32_t w = (sub & 4u) ? ((sub & 2u) ? r[3] : r[2]) : ((sub & 2u) ? r[1] : r[0]);
    drop8_from_bits16(d, (sub & 1u) ? (w >> 16) : (w & 0xffffu), mk);
    return;
  }
  philox4((uint32_t)q8, (uint32_t)(q8 >> 32), d.layer, d.off_lo, d.k0, d.k1, r);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t u = (e & 1) ? (r[e >> 1] >> 16) : (r[e >> 1] & 0xffffu);
    mk[e] = (u >= d.thr) ? d.inv_keep : 0.f;
  }
}

// keep*scale factors of the 4 elements 4*q .. 4*q+3 (the same mask: half of the 8-element group q / 2)
__device__ __forceinline__ void drop4(const DropP& d, uint64_t q, float (&mk)[4]) {
  uint32_t r[4];
  const uint64_t q8 = q >> 1;
  const bool up = (q & 1) != 0;
  if (d.two) {
    const uint64_t q64 = q8 >> 3;
    philox4((uint32_t)q64, (uint32_t)(q64 >> 32), d.layer, d.off_lo, d.k0, d.k1, r);
    const uint32_t sub = (uint32_t)q8 & 7u;
    const uint32_t w = (sub & 4u) ? ((sub & 2u) ? r[3] : r[2]) : ((sub & 2u) ? r[1] : r[0]);
    const uint32_t h16 = (sub & 1u) ? (w >> 16) : (w & 0xffffu);
    const uint32_t b8 = up ? (h16 >> 8) : h16;
#pragma unroll
    for (int e = 0; e < 4; ++e) mk[e] = (((b8 >> (2 * e)) & 3u) >= d.thr2) ? d.inv_keep : 0.f;
    return;
  }
  philox4((uint32_t)q8, (uint32_t)(q8 >> 32), d.layer, d.off_lo, d.k0, d.k1, r);
  const uint32_t ra = up ? r[2] : r[0], rb = up ? r[3] : r[1];
  const uint32_t u[4] = {ra & 0xffffu, ra >> 16, rb & 0xffffu, rb >> 16};
#pragma unroll
  for (int e = 0; e < 4; ++e) mk[e] = (u[e] >= d.thr) ? d.inv_keep : 0.f;
}


inline DropP make_drop(const vp3d_dropout* d) {
  DropP r;
  r.on = (d != nullptr && d->p > 0.f) ? 1 : 0;
  r.p = r.on ? d->p : 0.f;
  r.inv_keep = r.on ? 1.0f / (1.0f - d->p) : 1.f;
  r.thr = r.on ? (uint32_t)(d->p * 65536.0f + 0.5f) : 0u;
  r.two = (r.on && r.thr > 0u && r.thr < 65536u && (r.thr & 16383u) == 0u) ? 1 : 0;
  r.thr2 = r.thr >> 14;
  r.k0 = r.on ? (uint32_t)d->seed : 0u;
  r.k1 = r.on ? (uint32_t)(d->seed >> 32) : 0u;
  r.off_lo = r.on ? (uint32_t)d->offset : 0u;
  r.layer = r.on ? d->layer : 0u;
  r.off_ptr = r.on ? d->offset_ptr : nullptr;
  return r;
}


}  // namespace vp3d
