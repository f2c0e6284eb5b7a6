// Bodies of the HBM-bound S16 producers (activation forward, BatchNorm-backward apply) as device functions with an explicit
// block index, shared by the per-launch kernels (vp3d_s16_stream.hip) and the persistent tail kernel (vp3d_tail_s16.hip).
// Not part of the C ABI.
#pragma once
#include "vp3d_internal.h"
#include "vp3d_s16.h"

namespace vp3d {
namespace s16b {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TCH = 64;       // channels per block
constexpr int TPITCH = 65;    // LDS tile row pitch (floats)

struct TOut {                 // transposed output (ptr == nullptr: off)
  float* ptr;                 // S16, rows of ld 4-byte units
  int64_t ld;                 // >= roundup(M / taps, 64)
  int taps;                   // 1 or 3 (M % taps == 0)
};

// phase 2: tile[rows = taps*64][TPITCH] (scaled fp32, zero beyond M) -> T[(tap*C + c0 + ch)][col0 + 0..63]
__device__ __forceinline__ void tile_store_t(const float* tile, const TOut& t, int C, int c0, int64_t col0) {
  const int items = t.taps * 512;
  for (int idx = threadIdx.x; idx < items; idx += 256) {
    const int cg = idx & 7, ch = (idx >> 3) & 63, tap = idx >> 9;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[((cg * 8 + j) * t.taps + tap) * TPITCH + ch];
    f16x8 hi, lo;
    s16_split8(v, 1.f, hi, lo);
    f16x8* d = reinterpret_cast<f16x8*>(t.ptr + ((int64_t)tap * C + c0 + ch) * t.ld + col0 + cg * 8);
    d[0] = hi;
    d[1] = lo;
  }
}

// 8 consecutive per-channel parameters (two 16-byte loads when the array allows it)
__device__ __forceinline__ void load8(const float* __restrict__ p, float (&o)[8]) {
  if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = a[e];
      o[4 + e] = b[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = p[e];
  }
}

struct ResS16 {
  const float* res;           // S16 rows
  const float* bound;
  int t_dst, r_t, r_stride, r_off, r_ld;
  FastDiv div_t;              // row m -> sample m / t_dst
};

// ---------------------------------------------------------------------------------------------------------
// a = [res +] dropout(relu(y*scale + shift))  ->  S16 rows out[m][c] (+ transposed)
// grid.x = C/64, grid.y = row tiles of taps*64 rows; thread (r = tid>>3, g8 = tid&7) owns 8 channels
// ---------------------------------------------------------------------------------------------------------
// Body of one (64-channel strip bx, row tile by) block: `d` resolved, inv = 2^-e(out), rscale = 2^e(residual); `tile` = the
// block's LDS tile [taps*64][TPITCH] (only touched when a transposed copy is written).  256 threads.
__device__ __forceinline__ void bn_act_fwd_s16_body(int M, int C, const float* __restrict__ y, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const DropP& d, const ResS16& rm,
                                                    float inv, float rscale, float* __restrict__ out,
                                                    float* __restrict__ out_f32, const TOut& t,
                                                    uint8_t* __restrict__ act_bits, float* tile, int bx, int by) {
  const int g8 = threadIdx.x & 7, rsub = threadIdx.x >> 3;
  const int c0 = bx * TCH, c = c0 + g8 * 8;
  const int taps = t.ptr != nullptr ? t.taps : 1;
  const int R = taps * 64;
  const int64_t m0 = (int64_t)by * R;
  float sc[8], sh[8];
  load8(scale + c, sc);
  load8(shift + c, sh);
  for (int r = rsub; r < R; r += 32) {
    const int64_t m = m0 + r;
    float v[8];
    if (m < M) {
      const int64_t e0 = m * C + c;
      const f32x4 y0 = *reinterpret_cast<const f32x4*>(y + e0);
      const f32x4 y1 = *reinterpret_cast<const f32x4*>(y + e0 + 4);
      float mk[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
      if (d.on) {
        drop8(d, (uint64_t)(e0 >> 3), mk);               // (e0 % 8 == 0: C % 64 == 0, c % 8 == 0)
      }
      float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (rm.res != nullptr) {
        const int b = (int)fastdiv((uint32_t)m, rm.div_t);
        const int tt = (int)m - b * rm.t_dst;
        const f16x8* rp = reinterpret_cast<const f16x8*>(
            rm.res + ((int64_t)b * rm.r_t + (int64_t)tt * rm.r_stride + rm.r_off) * rm.r_ld + c);
        s16_join8(rp[0], rp[1], rscale, rv);
      }
      uint32_t bits = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yy = e < 4 ? y0[e] : y1[e - 4];
        const float z = fmaf(yy, sc[e], sh[e]);
        v[e] = rv[e] + (z > 0.f ? z * mk[e] : (z != z ? z : 0.f));
        bits |= (z > 0.f && mk[e] != 0.f) ? (1u << e) : 0u;
      }
      if (act_bits != nullptr) act_bits[act_bits_index(c, m, M)] = (uint8_t)bits;
      if (out_f32 != nullptr) {
        *reinterpret_cast<f32x4*>(out_f32 + e0) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(out_f32 + e0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= inv;
      f16x8 hi, lo;
      s16_split8(v, 1.f, hi, lo);
      f16x8* o = reinterpret_cast<f16x8*>(out + e0);
      o[0] = hi;
      o[1] = lo;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    if (t.ptr != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) tile[r * TPITCH + g8 * 8 + e] = v[e];
    }
  }
  if (t.ptr == nullptr) return;
  __syncthreads();
  tile_store_t(tile, t, C, c0, (int64_t)by * 64);
}

// ---------------------------------------------------------------------------------------------------------
// dy = scale*(g - dbeta/M - xhat*dgamma/M),  g = go*keep*[z>0]   ->  S16 rows (+ transposed, taps = 1)
// BITS: keep*[z>0] comes from the forward producer's activation bits (no Philox, no z); otherwise it is regenerated.
// A block walks the 64-row tiles ty = blockIdx.y, blockIdx.y + gridDim.y, ... of its 64-channel strip, so the
// per-channel constants are loaded once per block.
// ---------------------------------------------------------------------------------------------------------
// MASK (with BITS): only g = go * keep * [z>0] is produced (no y, no per-channel constants), scaled by the exponent of
// go_bound * keep_scale, which block (0,0) also publishes in mask_bound -- the operand of the expand layer's backward
// GEMM (vp3d_expand_bwd_s16), which needs no dy at all.
// Body of one block (64-channel strip bx; row tiles by, by + gy, ...): `d` resolved, inv = 2^-e(dy); `tile` = LDS tile
// [64][TPITCH] (only touched when a transposed copy is written).  256 threads.
template <bool BITS, bool MASK = false>
__device__ __forceinline__ void bn_bwd_apply_s16_body(int M, int C, const float* __restrict__ go, const float* __restrict__ y,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const DropP& d, const uint8_t* __restrict__ act_bits, float keep_scale,
                                                      const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                      float inv, float* __restrict__ dy, const TOut& t, float* tile, int bx,
                                                      int by, int gy) {
  const int g8 = threadIdx.x & 7, rsub = threadIdx.x >> 3;
  const int c0 = bx * TCH, c = c0 + g8 * 8;
  const float inv_m = 1.0f / (float)M;
  // v = A*g + B + Cx*(y - mu):   A = scale*inv, B = -A*dbeta/M, Cx = -A*invstd*dgamma/M
  float ka[8], kb[8], kc[8], mu[8], sc[8], sh[8];
  if (MASK) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ka[e] = inv;
      kb[e] = kc[e] = mu[e] = sc[e] = sh[e] = 0.f;
    }
  } else {
    float is[8], dg[8], db[8];
    load8(scale + c, sc);
    load8(mean + c, mu);
    load8(invstd + c, is);
    load8(dgamma + c, dg);
    load8(dbeta + c, db);
    if (!BITS) load8(shift + c, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ka[e] = sc[e] * inv;
      kb[e] = -ka[e] * (db[e] * inv_m);
      kc[e] = -ka[e] * (is[e] * (dg[e] * inv_m));
    }
  }
  const int ntiles = (M + 63) >> 6;
  for (int ty = by; ty < ntiles; ty += gy) {
    const int64_t m0 = (int64_t)ty * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = rsub + it * 32;
      const int64_t m = m0 + r;
      float v[8];
      if (m < M) {
        const int64_t e0 = m * C + c;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(go + e0);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(go + e0 + 4);
        f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
        if (!MASK) {
          y0 = *reinterpret_cast<const f32x4*>(y + e0);
          y1 = *reinterpret_cast<const f32x4*>(y + e0 + 4);
        }
        float mk[8];
        if (BITS) {
          const uint32_t bits = act_bits[act_bits_index(c, m, M)];
#pragma unroll
          for (int e = 0; e < 8; ++e) mk[e] = ((bits >> e) & 1u) ? keep_scale : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) mk[e] = 1.f;
          if (d.on) {
            drop8(d, (uint64_t)(e0 >> 3), mk);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yy = e < 4 ? y0[e] : y1[e - 4];
          const float gg = e < 4 ? g0[e] : g1[e - 4];
          float g = gg * mk[e];
          if (!BITS) g = fmaf(yy, sc[e], sh[e]) > 0.f ? g : 0.f;
          v[e] = MASK ? g * ka[e] : fmaf(kc[e], yy - mu[e], fmaf(ka[e], g, kb[e]));
        }
        if (dy != nullptr) {                        // (the expand conv needs no dgrad: only the transposed copy is written)
          f16x8 hi, lo;
          s16_split8(v, 1.f, hi, lo);
          f16x8* o = reinterpret_cast<f16x8*>(dy + e0);
          o[0] = hi;
          o[1] = lo;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
      if (t.ptr != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r * TPITCH + g8 * 8 + e] = v[e];
      }
    }
    if (t.ptr != nullptr) {
      __syncthreads();
      tile_store_t(tile, t, C, c0, m0);
      __syncthreads();
    }
  }
}

}  // namespace s16b
}  // namespace vp3d
