// Streaming (HBM-bound) producers of the S16 operands of the split-fp16 GEMMs (format: vp3d_s16.h):
//   k_bn_act_fwd_s16    y (fp32 conv output) -> a = [res +] dropout(relu(bn(y))) as S16 rows (+ transposed copy)
//   k_bn_bwd_apply_s16  (go, y) -> dy = BN/ReLU/dropout backward as S16 rows (+ transposed copy)
//   k_split_t           fp32 rows -> S16 rows and/or transposed S16 (expand-conv input staging)
//   k_pack_weight_s16   reference Conv1d.weight -> forward pack Wt[co][k*C_in+ci] and dgrad pack Wd[(k,ci)][co] in S16
//   bound kernels       guaranteed magnitude bounds (Samuelson: |x - mean| <= std*sqrt(M-1)) -> exponents, on the device
// "Transposed copy": the weight-gradient GEMM reduces over the rows m, and the fp16 MFMA wants 8 consecutive reduction
// indices per lane, so the producers of dy and of the layer inputs also emit  T[(tap*C + c)][m / taps]  (taps = the
// stride of the consuming strided conv: its input rows 3t..3t+2 are the taps of output row t), as S16 rows along m.
// A block owns taps*64 consecutive rows x 64 channels: phase 1 computes the values (16/32-B accesses along the
// channels), stores the S16 rows and parks the scaled fp32 values in an LDS tile [rows][65]; phase 2 walks the tile
// column-wise (conflict-free with the 65-float pitch) and writes 256-B runs of the transposed rows.
#include "vp3d_internal.h"
#include "vp3d_s16.h"
#include "vp3d_s16_stream_bodies.h"

namespace vp3d {
namespace {

using namespace s16b;
__global__ void __launch_bounds__(256) k_bn_act_fwd_s16(int M, int C, const float* __restrict__ y,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        DropP d, ResS16 rm, const float* __restrict__ out_bound,
                                                        float* __restrict__ out, float* __restrict__ out_f32, TOut t,
                                                        uint8_t* __restrict__ act_bits) {
  drop_resolve(d);
  extern __shared__ float tile[];
  const float inv = s16_pow2(-s16_exp_of(out_bound));
  const float rscale = rm.res != nullptr ? s16_pow2(s16_exp_of(rm.bound)) : 0.f;
  bn_act_fwd_s16_body(M, C, y, scale, shift, d, rm, inv, rscale, out, out_f32, t, act_bits, tile, blockIdx.x, blockIdx.y);
}

template <bool BITS, bool MASK = false>
__global__ void __launch_bounds__(256) k_bn_bwd_apply_s16(int M, int C, const float* __restrict__ go,
                                                          const float* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, DropP d,
                                                          const uint8_t* __restrict__ act_bits, float keep_scale,
                                                          const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                          const float* __restrict__ out_bound, float* __restrict__ dy,
                                                          TOut t, float* __restrict__ mask_bound) {
  if (!BITS) drop_resolve(d);
  extern __shared__ float tile[];
  float inv;
  if (MASK) {                                          // out_bound = the bound of go
    const float gb = s16_load_bound(out_bound) * keep_scale;
    inv = s16_pow2(-s16_exp_for_bound(gb));
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) mask_bound[0] = gb;   // (slots 1.. stay zero)
  } else {
    inv = s16_pow2(-s16_exp_of(out_bound));
  }
  bn_bwd_apply_s16_body<BITS, MASK>(M, C, go, y, scale, shift, mean, invstd, d, act_bits, keep_scale, dgamma, dbeta, inv, dy, t,
                                    tile, blockIdx.x, blockIdx.y, gridDim.y);
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm-backward column sums with the activation bits:  per channel  sum(g), sum(g * xhat),  g = go*keep*[z>0].
// Thread = 4 channels (one 16-byte load per array and row: a wave reads 1 KB runs) x the rows rsub, rsub + step, ...;
// the block folds its row sub-groups through LDS (fixed order) and writes ONE partial row:
// partials[(blockIdx.y*2 + {0,1})*C + c].
// ---------------------------------------------------------------------------------------------------------
// Finalize fused into the reduction (k_bn_bwd_reduce_strips below): what the strip's last block needs.
struct BwdFin {
  int* cnt;                    // [strips] tickets, zero on entry, zero again on exit
  float* dgamma;
  float* dbeta;
  const float* scale;
  const float* go_bound;
  float* dy_bound;
  float inv_keep, inv_m, sqrt_m1;
};

__global__ void __launch_bounds__(256) k_bn_bwd_reduce_bits(int M, int C, const float* __restrict__ go,
                                                            const float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const uint8_t* __restrict__ act_bits, float keep_scale,
                                                            float* __restrict__ partials, int lanes_per_row,
                                                            int rows_per_block) {
  __shared__ float red[256 * 8];
  const int lr = threadIdx.x % lanes_per_row, rsub = threadIdx.x / lanes_per_row;
  const int c = (blockIdx.x * lanes_per_row + lr) * 4;
  const bool live = rsub < rows_per_block && c < C;
  float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    float mu[4], is[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mu[e] = mean[c + e];
      is[e] = invstd[c + e];
    }
    const uint8_t* bp = act_bits + act_bits_index(c, 0, M);
    const int sh = c & 4;
    const int row_step = gridDim.y * rows_per_block;
#pragma unroll 8
    for (int m = blockIdx.y * rows_per_block + rsub; m < M; m += row_step) {
      const int64_t e0 = (int64_t)m * C + c;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(go + e0);
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + e0);
      const uint32_t bits = (uint32_t)bp[(int64_t)m * 8] >> sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = ((bits >> e) & 1u) ? gv[e] * keep_scale : 0.f;
        sg[e] += g;
        sgx[e] = fmaf(g, (yv[e] - mu[e]) * is[e], sgx[e]);
      }
    }
  }
  if (rows_per_block > 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[(e * 2 + 0) * 256 + threadIdx.x] = sg[e];
      red[(e * 2 + 1) * 256 + threadIdx.x] = sgx[e];
    }
    __syncthreads();
    if (rsub == 0 && c < C) {
      for (int r = 1; r < rows_per_block; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sg[e] += red[(e * 2 + 0) * 256 + r * lanes_per_row + lr];
          sgx[e] += red[(e * 2 + 1) * 256 + r * lanes_per_row + lr];
        }
      }
    }
  }
  if (rsub == 0 && c < C) {
    *reinterpret_cast<f32x4*>(partials + ((int64_t)blockIdx.y * 2 + 0) * C + c) = f32x4{sg[0], sg[1], sg[2], sg[3]};
    *reinterpret_cast<f32x4*>(partials + ((int64_t)blockIdx.y * 2 + 1) * C + c) = f32x4{sgx[0], sgx[1], sgx[2], sgx[3]};
  }
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm-backward column sums + finalize in ONE launch, strip-owned (vp3d_bn_bwd_reduce_fin_s16):
//   block (strip s of 64 channels, row block rb of R): thread (rsub = tid / 16, q = tid % 16) walks the rows
//   rb*rows_per + rsub + 16 i of its 4 channels (a wave reads four 256-byte runs per array and step), the block folds its
//   16 row lanes through LDS in lane order and writes ONE partial row of 128 floats: partials[(s*R + rb)*128 + {0,1}*64 + ch].
//   The LAST block of a strip to finish (one ticket per strip) loads the strip's R partial rows in a single batch (R <= 32:
//   <= 16 floats per thread), sums them in fp64 in row order (deterministic) and writes dgamma / dbeta of its 64 channels and
//   the strip's contribution to the bound of dy.  One fence / ticket / load round trip after the main loop: the two-stage
//   full-row version this replaces spent 18-30 us in its tail (4-6 dependent round trips; tools/reduce_bench.py), the
//   separate [C]-sized finalize launch before it waited 80-140 us for a CU slot behind the second stream's GEMM.
// ---------------------------------------------------------------------------------------------------------
constexpr int RED_MAX_R = 32;
__global__ void __launch_bounds__(256) k_bn_bwd_reduce_strips(int M, int C, int rows_per, const float* __restrict__ go,
                                                              const float* __restrict__ y, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const uint8_t* __restrict__ act_bits,
                                                              float* __restrict__ partials, BwdFin fin) {
  __shared__ float red[RED_MAX_R * 128];               // [16 row lanes][128] for the block fold, [R][128] for the strip fold
  __shared__ int flag;
  const int q = threadIdx.x & 15, rsub = threadIdx.x >> 4;
  const int s = blockIdx.x, rb = blockIdx.y, R = gridDim.y;
  const int c = s * 64 + q * 4;
  float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
  {
    float mu[4], is[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mu[e] = mean[c + e];
      is[e] = invstd[c + e];
    }
    const uint8_t* bp = act_bits + act_bits_index(c, 0, M);
    const int sh = c & 4;
    const int m_end = min(M, (rb + 1) * rows_per);
#pragma unroll 8
    for (int m = rb * rows_per + rsub; m < m_end; m += 16) {
      const int64_t e0 = (int64_t)m * C + c;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(go + e0);
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + e0);
      const uint32_t bits = (uint32_t)bp[(int64_t)m * 8] >> sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = ((bits >> e) & 1u) ? gv[e] * fin.inv_keep : 0.f;
        sg[e] += g;
        sgx[e] = fmaf(g, (yv[e] - mu[e]) * is[e], sgx[e]);
      }
    }
  }
  // block fold over the 16 row lanes (lane order)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[rsub * 128 + q * 4 + e] = sg[e];
    red[rsub * 128 + 64 + q * 4 + e] = sgx[e];
  }
  __syncthreads();
  float* prow = partials + ((int64_t)s * R + rb) * 128;
  if (threadIdx.x < 128) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r * 128 + threadIdx.x];
    prow[threadIdx.x] = t;
  }
  if (!last_arriver(fin.cnt + s, R, &flag)) return;   // (its barriers also fence red[])
  // ---- strip fold + finalize by the strip's last block ------------------------------------------------------------
  const float* srows = partials + (int64_t)s * R * 128;
  for (int i = threadIdx.x; i < R * 32; i += 256)
    *reinterpret_cast<f32x4*>(red + i * 4) = *reinterpret_cast<const f32x4*>(srows + i * 4);
  __syncthreads();
  float bmax = 0.f;
  const float gmax = s16_load_bound(fin.go_bound) * fin.inv_keep;      // (whole waves take part in the bound's shuffle)
  if (threadIdx.x < 64) {
    double a = 0.0, b = 0.0;
    for (int r = 0; r < R; ++r) {
      a += (double)red[r * 128 + threadIdx.x];
      b += (double)red[r * 128 + 64 + threadIdx.x];
    }
    const float db = (float)a, dg = (float)b;
    const int cc = s * 64 + threadIdx.x;
    fin.dbeta[cc] = db;
    fin.dgamma[cc] = dg;
    bmax = fabsf(fin.scale[cc]) * (gmax + fabsf(db) * fin.inv_m + fin.sqrt_m1 * fabsf(dg) * fin.inv_m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, o));
    if (threadIdx.x == 0) {
      s16_atomic_bound(fin.dy_bound, bmax);
      fin.cnt[s] = 0;                                  // every block of this strip has drawn its ticket
    }
  }
}

// im2row source of k_split_t<true>: row m = (b, t) is the k_valid contiguous floats at x[(b*t_src + t*t_stride)*ldx], zeros
// behind them, 1 in column one_col (vp3d_im2row's rows, never materialised)
struct Im2Row {
  int t_dst, t_src, t_stride, ldx, k_valid, one_col;
  FastDiv div_t;
};

// fp32 rows [M][C] (pitch ld_src; IM2ROW: gathered, see above) -> S16 rows (optional) and transposed S16 (optional, taps = 1)
template <bool IM2ROW>
__device__ __forceinline__ void split_t_body(int M, int C, const float* __restrict__ src, int64_t ld_src,
                                             const float* __restrict__ bound, float* __restrict__ out, int64_t ld_out,
                                             const TOut& t, const Im2Row& g, float* tile, int bx, int by) {
  const int g8 = threadIdx.x & 7, rsub = threadIdx.x >> 3;
  const int c0 = bx * TCH, c = c0 + g8 * 8;
  const int64_t m0 = (int64_t)by * 64;
  const float inv = bound != nullptr ? s16_pow2(-s16_exp_of(bound)) : 1.f;
  for (int r = rsub; r < 64; r += 32) {
    const int64_t m = m0 + r;
    float v[8];
    if (m < M) {
      if (IM2ROW) {
        const int b = (int)fastdiv((uint32_t)m, g.div_t), tt = (int)m - b * g.t_dst;
        const float* row = src + ((int64_t)b * g.t_src + (int64_t)tt * g.t_stride) * g.ldx;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (c + e < g.k_valid ? row[c + e] : (c + e == g.one_col ? 1.f : 0.f)) * inv;
      } else {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + m * ld_src + c);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + m * ld_src + c + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (e < 4 ? a0[e] : a1[e - 4]) * inv;
      }
      if (out != nullptr) {
        f16x8 hi, lo;
        s16_split8(v, 1.f, hi, lo);
        f16x8* o = reinterpret_cast<f16x8*>(out + m * ld_out + c);
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
  if (t.ptr == nullptr) return;
  __syncthreads();
  tile_store_t(tile, t, C, c0, (int64_t)by * 64);
}

template <bool IM2ROW>
__global__ void __launch_bounds__(256) k_split_t(int M, int C, const float* __restrict__ src, int64_t ld_src,
                                                 const float* __restrict__ bound, float* __restrict__ out, int64_t ld_out,
                                                 TOut t, Im2Row g) {
  extern __shared__ float tile[];
  split_t_body<IM2ROW>(M, C, src, ld_src, bound, out, ld_out, t, g, tile, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------
// weights: W[co][ci][k] (reference layout) -> S16 packs.  Block = 64 co x 64 ci (all taps) through LDS.
//   fwd  : Wt[co][k*C_in + ci]                    (rows of ld_f 4-byte units; columns beyond taps*C_in untouched)
//   dgr  : strided: Wd[(k*C_in + ci)][co]         (rows of ld_d)
//          dilated: Wd[ci][k*C_out + co]
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_weight_s16(int c_out, int c_in, int taps, const float* __restrict__ w,
                                                         const float* __restrict__ bound, float* __restrict__ wf,
                                                         int64_t ld_f, float* __restrict__ wd, int64_t ld_d,
                                                         int dilated_form) {
  extern __shared__ float tile[];                 // [taps][64 co][TPITCH] (ci fastest)
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
  const float inv = s16_pow2(-s16_exp_of(bound));
  const int row_f = 64 * taps;                    // contiguous floats per co row of the tile
  for (int idx = threadIdx.x; idx < 64 * row_f; idx += 256) {
    const int co = idx / row_f, q = idx - co * row_f;     // q = ci_local*taps + k
    const int ci = q / taps, k = q - ci * taps;
    tile[(k * 64 + co) * TPITCH + ci] = w[((int64_t)(co0 + co) * c_in + ci0) * taps + q] * inv;
  }
  __syncthreads();
  if (wf != nullptr) {
    for (int idx = threadIdx.x; idx < taps * 512; idx += 256) {      // (k, co, ci-group)
      const int g = idx & 7, co = (idx >> 3) & 63, k = idx >> 9;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[(k * 64 + co) * TPITCH + g * 8 + j];
      f16x8 hi, lo;
      s16_split8(v, 1.f, hi, lo);
      f16x8* d = reinterpret_cast<f16x8*>(wf + (int64_t)(co0 + co) * ld_f + (int64_t)k * c_in + ci0 + g * 8);
      d[0] = hi;
      d[1] = lo;
    }
  }
  if (wd != nullptr) {
    for (int idx = threadIdx.x; idx < taps * 512; idx += 256) {      // (k, ci, co-group)
      const int g = idx & 7, ci = (idx >> 3) & 63, k = idx >> 9;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[(k * 64 + g * 8 + j) * TPITCH + ci];
      f16x8 hi, lo;
      s16_split8(v, 1.f, hi, lo);
      float* row = dilated_form ? wd + (int64_t)(ci0 + ci) * ld_d + (int64_t)k * c_out
                                : wd + ((int64_t)k * c_in + ci0 + ci) * ld_d;
      f16x8* d = reinterpret_cast<f16x8*>(row + co0 + g * 8);
      d[0] = hi;
      d[1] = lo;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// bounds
// ---------------------------------------------------------------------------------------------------------
// maximum over a 1024-thread block, returned to every thread (xor butterfly per wave + the 16 wave maxima through LDS:
// two barriers instead of the eleven of an LDS tree -- these kernels are pure latency)
__device__ __forceinline__ float block_max_1024(float m, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __syncthreads();                                   // readers of a previous call are done with red[]
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) r = fmaxf(r, red[i]);
  return r;
}

// |dropout(relu(bn(y)))| <= (|gamma|*sqrt(M-1) + |beta|) / (1-p)   (Samuelson), plus the residual's bound
__global__ void __launch_bounds__(1024) k_act_bound(int C, float sqrt_m1, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float inv_keep,
                                                    const float* __restrict__ res_bound, float* __restrict__ out) {
  __shared__ float red[1024];
  const float rb = res_bound != nullptr ? s16_load_bound(res_bound) : 0.f;
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 1024) m = fmaxf(m, fabsf(gamma[c]) * sqrt_m1 + fabsf(beta[c]));
  m = block_max_1024(m, red);
  if (threadIdx.x == 0) out[0] = m * inv_keep + rb;
}

// |dy_c| <= |scale_c| * (gmax + |dbeta_c|/M + sqrt(M-1)*|dgamma_c|/M),  gmax = go_bound / (1-p)
__global__ void __launch_bounds__(1024) k_dy_bound(int C, float inv_m, float sqrt_m1, const float* __restrict__ scale,
                                                   const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                   const float* __restrict__ go_bound, float inv_keep,
                                                   float* __restrict__ out) {
  __shared__ float red[1024];
  const float gmax = s16_load_bound(go_bound) * inv_keep;
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 1024)
    m = fmaxf(m, fabsf(scale[c]) * (gmax + fabsf(dbeta[c]) * inv_m + sqrt_m1 * fabsf(dgamma[c]) * inv_m));
  m = block_max_1024(m, red);
  if (threadIdx.x == 0) out[0] = m;
}

// ---------------------------------------------------------------------------------------------------------
// per-step prologue of the training forward, one launch each for ALL layers (the per-layer launches of a few
// microseconds each added up to ~0.3 ms of a 5.8 ms step)
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxLayers = 16;

struct AmaxMulti {
  const float* src[kMaxLayers];
  int64_t n[kMaxLayers];
  float* bounds;               // bound of tensor i at bounds + i * kBoundSlots
};

__global__ void __launch_bounds__(256) k_amax_multi(AmaxMulti a) {
  const int ti = blockIdx.y;
  const float* src = a.src[ti];
  const int64_t n = a.n[ti];
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) ? n >> 2 : 0;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
#pragma unroll 4
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = s4[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(src[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) s16_atomic_bound(a.bounds + ti * kBoundSlots, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

struct PackMulti {
  const float* w[kMaxLayers];
  float* wf[kMaxLayers];
  float* wd[kMaxLayers];
  int taps[kMaxLayers];
  const float* bounds;         // bound of layer i at bounds + i * kBoundSlots
  int c_out, c_in;
};

// load phase of the pack: the block's 64 co rows of 64 * TAPS contiguous floats each, as 16-byte loads (a row starts at a
// multiple of 256 * TAPS bytes: c_in % 64 == 0), FOUR in flight per thread -- the scalar form (48 dependent-issue 4-byte loads per
// thread for 3 taps) left the kernel at 2.3 TB/s with every CU slot taken: latency-bound, and a bad neighbour for whatever the
// other stream runs beside it (round 6: profiles/r06_prologue_overlap.txt)
template <int TAPS>
__device__ __forceinline__ void pack_load_tile(const float* __restrict__ w, int c_in, int co0, int ci0, float inv, float* tile) {
  constexpr int ROW4 = 16 * TAPS;                   // float4 per co row
  constexpr int N4 = 64 * ROW4;                     // float4 per block: 1024 * TAPS = 4 * TAPS per thread
  typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int b0 = 0; b0 < N4; b0 += 4 * 256) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = b0 + u * 256 + threadIdx.x;
      const int co = f / ROW4, q4 = f - co * ROW4;
      v[u] = *reinterpret_cast<const f4*>(w + ((int64_t)(co0 + co) * c_in + ci0) * TAPS + q4 * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = b0 + u * 256 + threadIdx.x;
      const int co = f / ROW4, q4 = f - co * ROW4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = q4 * 4 + e;
        const int ci = q / TAPS, k = q - ci * TAPS;
        tile[(k * 64 + co) * TPITCH + ci] = v[u][e] * inv;
      }
    }
  }
}

__device__ __forceinline__ void pack_weight_s16_multi_body(const PackMulti& a, float* tile, int bx, int by, int li) {
  const int taps = a.taps[li], c_in = a.c_in, c_out = a.c_out;
  const float* __restrict__ w = a.w[li];
  float* __restrict__ wf = a.wf[li];
  float* __restrict__ wd = a.wd[li];
  const int co0 = by * 64, ci0 = bx * 64;
  const float inv = s16_pow2(-s16_exp_of(a.bounds + li * kBoundSlots));
  if ((reinterpret_cast<uintptr_t>(w) & 15u) == 0 && taps == 3) {
    pack_load_tile<3>(w, c_in, co0, ci0, inv, tile);
  } else if ((reinterpret_cast<uintptr_t>(w) & 15u) == 0 && taps == 1) {
    pack_load_tile<1>(w, c_in, co0, ci0, inv, tile);
  } else if ((reinterpret_cast<uintptr_t>(w) & 15u) == 0 && taps == 2) {
    pack_load_tile<2>(w, c_in, co0, ci0, inv, tile);
  } else {
    const int row_f = 64 * taps;
    for (int idx = threadIdx.x; idx < 64 * row_f; idx += 256) {
      const int co = idx / row_f, q = idx - co * row_f;
      const int ci = q / taps, k = q - ci * taps;
      tile[(k * 64 + co) * TPITCH + ci] = w[((int64_t)(co0 + co) * c_in + ci0) * taps + q] * inv;
    }
  }
  __syncthreads();
  const int64_t ld_f = (int64_t)taps * c_in, ld_d = c_out;
  if (wf != nullptr) {
    for (int idx = threadIdx.x; idx < taps * 512; idx += 256) {
      const int g = idx & 7, co = (idx >> 3) & 63, k = idx >> 9;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[(k * 64 + co) * TPITCH + g * 8 + j];
      f16x8 hi, lo;
      s16_split8(v, 1.f, hi, lo);
      f16x8* d = reinterpret_cast<f16x8*>(wf + (int64_t)(co0 + co) * ld_f + (int64_t)k * c_in + ci0 + g * 8);
      d[0] = hi;
      d[1] = lo;
    }
  }
  if (wd != nullptr) {
    for (int idx = threadIdx.x; idx < taps * 512; idx += 256) {
      const int g = idx & 7, ci = (idx >> 3) & 63, k = idx >> 9;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[(k * 64 + g * 8 + j) * TPITCH + ci];
      f16x8 hi, lo;
      s16_split8(v, 1.f, hi, lo);
      f16x8* d = reinterpret_cast<f16x8*>(wd + ((int64_t)k * c_in + ci0 + ci) * ld_d + co0 + g * 8);
      d[0] = hi;
      d[1] = lo;
    }
  }
}

__global__ void __launch_bounds__(256) k_pack_weight_s16_multi(PackMulti a) {
  extern __shared__ float tile[];                 // [taps][64 co][TPITCH]
  pack_weight_s16_multi_body(a, tile, blockIdx.x, blockIdx.y, blockIdx.z);
}

struct ActBoundsMulti {
  const float* gamma[kMaxLayers];
  const float* beta[kMaxLayers];
  float sqrt_m1[kMaxLayers];
  int res_from[kMaxLayers];    // index of the layer whose activation is the residual source, -1: none
  float* bounds;               // bound of layer i at bounds + i * kBoundSlots (slot 0 written, the rest zero)
  int n_layers, C;
  float inv_keep;
};

__global__ void __launch_bounds__(1024) k_act_bounds_multi(ActBoundsMulti a) {
  __shared__ float red[1024];
  __shared__ float done[kMaxLayers];
  for (int li = 0; li < a.n_layers; ++li) {
    float m = 0.f;
    for (int c = threadIdx.x; c < a.C; c += 1024)
      m = fmaxf(m, fabsf(a.gamma[li][c]) * a.sqrt_m1[li] + fabsf(a.beta[li][c]));
    m = block_max_1024(m, red);
    if (threadIdx.x == 0) {
      const float b = m * a.inv_keep + (a.res_from[li] >= 0 ? done[a.res_from[li]] : 0.f);
      done[li] = b;
      a.bounds[li * kBoundSlots] = b;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// The prologue of a training forward as TWO launches (vp3d_prologue_a_s16 / vp3d_prologue_b_s16) instead of seven: untraced
// the dependent chain amax(x) -> im2row + split | amax(weights) -> expand-weight pack -> its S16 split -> weight packs |
// activation bounds takes 125 us before the first GEMM (tools/region_time.py) -- launch A holds everything that depends on
// nothing (all maxima, the activation bounds), launch B everything that needs only those.
// ---------------------------------------------------------------------------------------------------------
struct PrologueA {
  const float* src[kMaxLayers];
  int64_t n[kMaxLayers];
  float* bound[kMaxLayers];    // 32-slot bound of tensor i (zeroed by the caller)
  float floor_[kMaxLayers];
  int n_tensors;
  ActBoundsMulti ab;           // ab.n_layers == 0: none
};

__global__ void __launch_bounds__(256) k_prologue_a(PrologueA a) {
  __shared__ float red[4];
  __shared__ float done[kMaxLayers];
  const int ti = blockIdx.y;
  if (ti < a.n_tensors) {
    const float* src = a.src[ti];
    const int64_t n = a.n[ti];
    float m = a.floor_[ti];
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) ? n >> 2 : 0;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {       // four 16-byte loads in flight per thread
      const f32x4 v0 = s4[i], v1 = s4[i + stride], v2 = s4[i + 2 * stride], v3 = s4[i + 3 * stride];
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(fmaxf(m, fmaxf(fabsf(v0[e]), fabsf(v1[e]))), fmaxf(fabsf(v2[e]), fabsf(v3[e])));
    }
    for (; i < n4; i += stride) {
      const f32x4 v = s4[i];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    for (int64_t j = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) m = fmaxf(m, fabsf(src[j]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) s16_atomic_bound(a.bound[ti], fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    return;
  }
  if (blockIdx.x != 0) return;                           // one block: the activation bounds of all layers (k_act_bounds_multi)
  const ActBoundsMulti& b = a.ab;
  for (int li = 0; li < b.n_layers; ++li) {
    float m = 0.f;
    for (int c = threadIdx.x; c < b.C; c += 256) m = fmaxf(m, fabsf(b.gamma[li][c]) * b.sqrt_m1[li] + fabsf(b.beta[li][c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * b.inv_keep + (b.res_from[li] >= 0 ? done[b.res_from[li]] : 0.f);
      done[li] = v;
      b.bounds[li * kBoundSlots] = v;
    }
    __syncthreads();
  }
}

struct PrologueB {
  // blocks [0, n_in): im2row + S16 split of the raw input (k_split_t<true>): (kpad / 64) x row tiles
  int M, kpad, n_in;
  const float* x;
  const float* x_bound;
  float* x_rows;
  TOut x_t;
  Im2Row g;
  // blocks [n_in, n_in + n_w0): the expand conv's weight W0 [c0][cin0][taps0] -> fp32 pack [c0][kpad] (zero padded) and its
  // S16 rows (vp3d_pack_weight + vp3d_split_rows): 16 rows x (kpad / 8) groups per block
  int n_w0, c0, cin0, taps0;
  const float* w0;
  const float* w0_bound;
  float* w0_packed;
  float* w0_s16;
  // blocks behind: the C x C weight packs (k_pack_weight_s16_multi): (c_in / 64) x (c_out / 64) x layers
  PackMulti pk;
  int pk_layers;
};

__global__ void __launch_bounds__(256) k_prologue_b(PrologueB a) {
  extern __shared__ float tile[];
  int b = blockIdx.x;
  if (b < a.n_in) {
    const int gx = a.kpad / 64;
    split_t_body<true>(a.M, a.kpad, a.x, (int64_t)0, a.x_bound, a.x_rows, (int64_t)a.kpad, a.x_t, a.g, tile, b % gx, b / gx);
    return;
  }
  b -= a.n_in;
  if (b < a.n_w0) {
    const int groups = a.kpad / 8, rows_per = 256 / groups;
    const int r = threadIdx.x / groups, g8 = threadIdx.x % groups;
    const int co = b * rows_per + r;
    if (r >= rows_per || co >= a.c0) return;
    const int kv = a.cin0 * a.taps0;
    const float inv = s16_pow2(-s16_exp_of(a.w0_bound));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = g8 * 8 + e;
      const int k = j / a.cin0, ci = j - k * a.cin0;
      v[e] = j < kv ? a.w0[((int64_t)co * a.cin0 + ci) * a.taps0 + k] : 0.f;
    }
    float* dst = a.w0_packed + (int64_t)co * a.kpad + g8 * 8;
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    f16x8 hi, lo;
    s16_split8(v, inv, hi, lo);
    f16x8* o = reinterpret_cast<f16x8*>(a.w0_s16 + (int64_t)co * a.kpad + g8 * 8);
    o[0] = hi;
    o[1] = lo;
    return;
  }
  b -= a.n_w0;
  const int gx = a.pk.c_in / 64, gy = a.pk.c_out / 64;
  const int li = b / (gx * gy), q = b - li * (gx * gy);
  if (li < a.pk_layers) pack_weight_s16_multi_body(a.pk, tile, q % gx, q / gx, li);
}

// BN backward finalize (dgamma, dbeta from the partial sums, fp64) + the bound of dy in the same launch:
// |dy_c| <= |scale_c| * (g + |dbeta_c|/M + sqrt(M-1)*|dgamma_c|/M),  g = go_bound / (1-p)
// (256-thread blocks of <= 32 VGPRs: they fit beside the two 240-VGPR waves per SIMD of a 256x256 wgrad GEMM running on
// the second stream -- 1024-thread blocks waited 80-140 us for whole workgroups of that GEMM to retire)
constexpr int FIN_CH = 16, FIN_GROUPS = 16;
__global__ void __launch_bounds__(FIN_CH * FIN_GROUPS) k_bn_bwd_finalize_bound(
    int C, const float* __restrict__ partials, int nparts, float* dgamma, float* dbeta, const float* __restrict__ scale,
    const float* __restrict__ go_bound, float inv_keep, float inv_m, float sqrt_m1, float* __restrict__ dy_bound) {
  __shared__ double s1[FIN_GROUPS][FIN_CH], s2[FIN_GROUPS][FIN_CH];
  __shared__ float bmax[FIN_CH];
  const float gmax = s16_load_bound(go_bound) * inv_keep;
  const int cl = threadIdx.x % FIN_CH, g = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  double a1 = 0.0, a2 = 0.0;
  if (c < C) {
#pragma unroll 8
    for (int p = g; p < nparts; p += FIN_GROUPS) {
      a1 += (double)partials[((int64_t)p * 2 + 0) * C + c];
      a2 += (double)partials[((int64_t)p * 2 + 1) * C + c];
    }
  }
  s1[g][cl] = a1;
  s2[g][cl] = a2;
  __syncthreads();
  for (int o = FIN_GROUPS / 2; o >= 1; o >>= 1) {
    if (g < o) {
      s1[g][cl] += s1[g + o][cl];
      s2[g][cl] += s2[g + o][cl];
    }
    __syncthreads();
  }
  if (g == 0) {
    float b = 0.f;
    if (c < C) {
      const float db = (float)s1[0][cl], dg = (float)s2[0][cl];
      dbeta[c] = db;
      dgamma[c] = dg;
      b = fabsf(scale[c]) * (gmax + fabsf(db) * inv_m + sqrt_m1 * fabsf(dg) * inv_m);
    }
    bmax[cl] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = bmax[0];
    for (int i = 1; i < FIN_CH; ++i) m = fmaxf(m, bmax[i]);
    s16_atomic_bound(dy_bound, m);
  }
}

// ---------------------------------------------------------------------------------------------------------
// S16 rows x[B][t_src][C] -> the transposed operand of a conv's weight gradient for ANY (stride, dilation, taps):
//     T[(k*C + c)][m] = x[b][t*t_stride + t_off + k*tap_step][c],   m = b*t_dst + t      (zero beyond M, out of range)
// i.e. the layout the forward producers write for strided convs whose windows tile the input, built on demand (in
// backward, from the saved rows) for the dilated class, for windows that do not tile and for wider filters.
// grid = (C/64, row tiles of 64, taps); values keep their exponent (decode hi + lo, re-split: value-preserving).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_t_s16(int M, int C, int t_dst, int t_src, int t_stride, int tap_step,
                                                      int t_off, const float* __restrict__ x, TOut t) {
  extern __shared__ float tile[];
  const int g8 = threadIdx.x & 7, rsub = threadIdx.x >> 3;
  const int c0 = blockIdx.x * TCH, c = c0 + g8 * 8;
  const int k = blockIdx.z;
  const int64_t m0 = (int64_t)blockIdx.y * 64;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = rsub + it * 32;
    const int64_t m = m0 + r;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      const int b = (int)(m / t_dst);
      const int tt = (int)(m - (int64_t)b * t_dst) * t_stride + t_off + k * tap_step;
      if ((unsigned)tt < (unsigned)t_src) {
        const f16x8* rp = reinterpret_cast<const f16x8*>(x + ((int64_t)b * t_src + tt) * C + c);
        s16_join8(rp[0], rp[1], 1.f, v);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r * TPITCH + g8 * 8 + e] = v[e];
  }
  __syncthreads();
  TOut tk{t.ptr + (int64_t)k * C * t.ld, t.ld, 1};
  tile_store_t(tile, tk, C, c0, m0);
}

// ---------------------------------------------------------------------------------------------------------
// Backward of the expand layer (conv -> BatchNorm -> ReLU -> dropout) WITHOUT materialising dy.  The layer has no data
// gradient to produce and its conv is linear in a 128-column operand X (im2row rows with a bias column of ones), so with
// G = go * keep * [z>0] (vp3d_act_mask_t_s16) everything follows from two small reductions over the rows,
//     P = G^T X  [C][kpad]      and      S = X^T X  [kpad][kpad]      (S16 GEMMs, K = rows):
//   dbeta_c  = sum g              = P[c][one]                        (the bias column)
//   sum g*y  = <W_c, P[c]>        (y = X W^T)           ->  dgamma_c = invstd_c (sum g*y - mean_c dbeta_c)
//   dy       = A g + B + Cx (y - mean),  A = scale, B = -A dbeta/M, Cx = -A invstd dgamma/M   (k_bn_bwd_apply_s16)
//   dW[c][j] = sum_m dy[m][c] X[m][j] = A P[c][j] + B sX[j] + Cx ((W S)[c][j] - mean_c sX[j]),   sX[j] = S[j][one]
// Replaces, for that layer, vp3d_bn_bwd_reduce_bits + finalize + vp3d_bn_bwd_apply_s16 (two passes over go and y,
// one dy write: 2.4 GB for the cfg3 step) by one pass over go, and the weight-gradient un-pack.
// ---------------------------------------------------------------------------------------------------------
// out[i] = sum_s ws[s][i] (fp64 accumulation, fixed order)
__global__ void __launch_bounds__(256) k_sum_slices(int64_t n, int splits, const float* __restrict__ ws, double* __restrict__ out) {
  // 64 elements per block x 4 slice lanes: slice lane l adds the slices l, l + 4, ... (8 loads in flight), the lanes are folded
  // in lane order through LDS -- a fixed tree, and 4x the blocks / a quarter of the dependent chain of the one-thread-per-
  // element loop this replaces (64 slices of X^T X: 30 -> 8 us)
  __shared__ double red[4][64];
  const int e = threadIdx.x & 63, l = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  double a = 0.0;
  if (i < n) {
    for (int s0 = l; s0 < splits; s0 += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s0 + 4 * u < splits ? ws[(int64_t)(s0 + 4 * u) * n + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) a += (double)v[u];
    }
  }
  red[l][e] = a;
  __syncthreads();
  if (l == 0 && i < n) out[i] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
}

// block = 8 channels x 32 lanes, lane = 4 consecutive columns of the kpad <= 128 wide rows
__global__ void __launch_bounds__(256) k_expand_bwd_post(int C, int kpad, int kv, int one, int c_in, int taps, int splits,
                                                         const float* __restrict__ pws, const double* __restrict__ S,
                                                         const float* __restrict__ wp, const float* __restrict__ scale,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         double inv_m, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         float* __restrict__ dw, const float* __restrict__ xt, int64_t ld_t,
                                                         const float* __restrict__ x_bound) {
  extern __shared__ double s_lds[];                    // S rows 0 .. kv-1 ([kv][kpad] doubles), then the block's 8 weight rows
  float* w_s = reinterpret_cast<float*>(s_lds + (size_t)kv * kpad);
  const int cl = threadIdx.x >> 5, jq = threadIdx.x & 31, j0 = jq * 4;
  const int c = blockIdx.x * 8 + cl;
  const bool live = c < C && j0 < kpad;
  // S = X^T X into LDS (the loop below reads every row of it for every channel: from L2 that was 17 dependent batches)
  if (xt == nullptr) {
    const int n2 = kv * kpad / 2;                      // double2 units
    const double2* src = reinterpret_cast<const double2*>(S);
    double2* dst = reinterpret_cast<double2*>(s_lds);
    for (int i = threadIdx.x; i < n2; i += 256) dst[i] = src[i];
  } else {
    // S is the FORWARD's centred second-moment matrix G (vp3d_expand_stats_gram_s16: G_ij = sum_m (x_i - o_i)(x_j - o_j) with
    // o_k = X[0][k], the constant-1 column unshifted, so row `one` holds the column sums of the shifted data and G[one][one] = M):
    //   (X^T X)_ij = G_ij + o_i G[one][j] + o_j G[one][i] + M o_i o_j        (fp64)
    // -- the backward forms no second-moment matrix of its own (no ride-along MFMAs in the P launch, no vp3d_sum_slices).
    __shared__ double o_s[128], g1_s[128];
    if ((int)threadIdx.x < kpad) {
      const int k = threadIdx.x;
      const _Float16* g = reinterpret_cast<const _Float16*>(xt + (int64_t)k * ld_t);
      o_s[k] = k == one ? 0.0 : (double)(((float)g[0] + (float)g[8]) * s16_pow2(s16_exp_of(x_bound)));
      g1_s[k] = S[(int64_t)one * kpad + k];
    }
    __syncthreads();
    const double m_rows = g1_s[one];
    for (int i = threadIdx.x; i < kv * kpad; i += 256) {
      const int r = i / kpad, cc = i - r * kpad;
      s_lds[i] = S[i] + o_s[r] * g1_s[cc] + o_s[cc] * g1_s[r] + m_rows * o_s[r] * o_s[cc];
    }
  }
  double P[4] = {0.0, 0.0, 0.0, 0.0};
  float w4[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    // the slices in slice order, 16 loads in flight
    for (int s0 = 0; s0 < splits; s0 += 16) {
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        v[u] = s0 + u < splits ? *reinterpret_cast<const f32x4*>(pws + ((int64_t)(s0 + u) * C + c) * kpad + j0)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) P[e] += (double)v[u][e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) w4[e] = j0 + e < kv ? wp[(int64_t)c * kpad + j0 + e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) w_s[cl * 128 + ((j0 + e) & 127)] = w4[e];
  // per-channel sums over the 32 lanes of a channel (half a wave): sum g*y and the bias column
  double sgy = 0.0, db = 0.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sgy += (double)w4[e] * P[e];
    db += (live && j0 + e == one) ? P[e] : 0.0;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    sgy += __shfl_xor(sgy, o);
    db += __shfl_xor(db, o);
  }
  __syncthreads();
  if (!live) return;
  const double mu = (double)mean[c], is = (double)invstd[c], A = (double)scale[c];
  const double dg = is * (sgy - mu * db);
  if (jq == 0) {
    dgamma[c] = (float)dg;
    dbeta[c] = (float)db;
  }
  const double B = -A * db * inv_m, Cx = -A * is * dg * inv_m;
  double T[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 6
  for (int i = 0; i < kv; ++i) {
    const double wi = (double)w_s[cl * 128 + i];
    const double2 s01 = *reinterpret_cast<const double2*>(s_lds + (size_t)i * kpad + j0);
    const double2 s23 = *reinterpret_cast<const double2*>(s_lds + (size_t)i * kpad + j0 + 2);
    T[0] += wi * s01.x;
    T[1] += wi * s01.y;
    T[2] += wi * s23.x;
    T[3] += wi * s23.y;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = j0 + e;
    if (j >= kv) continue;
    const double sx = s_lds[(size_t)j * kpad + one];
    const double v = A * P[e] + B * sx + Cx * (T[e] - mu * sx);
    const int k = j / c_in, ci = j - k * c_in;
    dw[((int64_t)c * c_in + ci) * taps + k] = (float)v;
  }
}

}  // namespace
}  // namespace vp3d

using namespace vp3d;

static int check_t(const char* who, void* t_out, int64_t ld_t, int32_t taps, int64_t M) {
  if (t_out == nullptr) return VP3D_OK;
  VP3D_REQUIRE(taps >= 1 && M % taps == 0 && ld_t % 8 == 0 && ld_t >= (M / taps + 63) / 64 * 64 && aligned16(t_out),
               "%s: transposed output needs M %% taps == 0 and ld_t >= roundup(M/taps, 64) (M=%lld taps=%d ld_t=%lld)", who,
               (long long)M, taps, (long long)ld_t);
  return VP3D_OK;
}

extern "C" {

int vp3d_bn_act_fwd_s16(vp3d_stream_t stream, int64_t M, int32_t C, const float* y, const float* scale,
                        const float* shift, const vp3d_dropout* drop, const void* res, const float* res_bound,
                        int32_t t_dst, int32_t r_t, int32_t r_stride, int32_t r_off, int32_t r_ld,
                        const float* out_bound, void* out, float* out_f32, void* t_out, int64_t ld_t, int32_t taps,
                        uint8_t* act_bits) {
  VP3D_REQUIRE(M > 0 && M < ((int64_t)1 << 31) && C > 0 && C % 64 == 0 && y && scale && shift && out && out_bound &&
                   (out_f32 == nullptr || aligned16(out_f32)),
               "bn_act_fwd_s16: bad argument (needs C %% 64 == 0)");
  VP3D_REQUIRE(aligned16(y) && aligned16(out) && (res == nullptr || (aligned16(res) && res_bound && r_ld % 8 == 0)),
               "bn_act_fwd_s16: 16-byte aligned buffers required");
  int rc = check_t("bn_act_fwd_s16", t_out, ld_t, taps, M);
  if (rc) return rc;
  if (drop) VP3D_REQUIRE(drop->p >= 0.f && drop->p < 1.f, "bn_act_fwd_s16: dropout p=%f", drop->p);
  const DropP d = make_drop(drop);
  ResS16 rm{(const float*)res, res_bound, t_dst > 0 ? t_dst : 1, r_t, r_stride, r_off, r_ld, make_fastdiv(t_dst)};
  TOut t{(float*)t_out, ld_t, t_out ? taps : 1};
  const int R = 64 * t.taps;
  VP3D_REQUIRE((M + R - 1) / R <= 65535, "bn_act_fwd_s16: more than 65535 row tiles (M=%lld)", (long long)M);
  const size_t lds = t_out ? (size_t)R * TPITCH * 4 : 0;
  VP3D_LAUNCH(k_bn_act_fwd_s16, dim3(C / 64, (unsigned)((M + R - 1) / R)), dim3(256), lds, (hipStream_t)stream,
                     (int)M, C, y, scale, shift, d, rm, out_bound, (float*)out, out_f32, t, act_bits);
  return check_launch("bn_act_fwd_s16");
}

int vp3d_bn_bwd_apply_s16(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          const vp3d_dropout* drop, const uint8_t* act_bits, const float* dgamma, const float* dbeta,
                          const float* out_bound, void* dy, void* t_out, int64_t ld_t) {
  VP3D_REQUIRE(M > 0 && M < ((int64_t)1 << 31) && C > 0 && C % 64 == 0 && go && y && scale && shift && mean && invstd &&
                   dgamma && dbeta && out_bound && (dy || t_out),
               "bn_bwd_apply_s16: bad argument (needs C %% 64 == 0)");
  VP3D_REQUIRE(aligned16(go) && aligned16(y) && aligned16(dy), "bn_bwd_apply_s16: 16-byte aligned buffers required");
  int rc = check_t("bn_bwd_apply_s16", t_out, ld_t, 1, M);
  if (rc) return rc;
  if (drop) VP3D_REQUIRE(drop->p >= 0.f && drop->p < 1.f, "bn_bwd_apply_s16: dropout p=%f", drop->p);
  const DropP d = make_drop(drop);
  TOut t{(float*)t_out, ld_t, 1};
  const size_t lds = t_out ? (size_t)64 * TPITCH * 4 : 0;
  const int64_t ntiles = (M + 63) / 64, gx = C / 64;
  int64_t per_block = gx * ntiles / 2048;       // tiles per block: amortise the per-channel constants, keep >= 2048 blocks
  per_block = per_block < 1 ? 1 : (per_block > 8 ? 8 : per_block);
  int64_t gy = (ntiles + per_block - 1) / per_block;
  gy = gy > 65535 ? 65535 : gy;                 // (the kernel strides over the tiles)
  const dim3 grid((unsigned)gx, (unsigned)gy);
  if (act_bits != nullptr)
    VP3D_LAUNCH((k_bn_bwd_apply_s16<true>), grid, dim3(256), lds, (hipStream_t)stream, (int)M, C, go, y, scale, shift,
                       mean, invstd, d, act_bits, d.inv_keep, dgamma, dbeta, out_bound, (float*)dy, t, (float*)nullptr);
  else
    VP3D_LAUNCH((k_bn_bwd_apply_s16<false>), grid, dim3(256), lds, (hipStream_t)stream, (int)M, C, go, y, scale,
                       shift, mean, invstd, d, act_bits, d.inv_keep, dgamma, dbeta, out_bound, (float*)dy, t, (float*)nullptr);
  return check_launch("bn_bwd_apply_s16");
}

int vp3d_act_mask_s16(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* go_bound,
                      const uint8_t* act_bits, float p, float* g_bound, void* rows_out, void* t_out, int64_t ld_t) {
  VP3D_REQUIRE(M > 0 && M < ((int64_t)1 << 31) && C > 0 && C % 64 == 0 && go && go_bound && act_bits && g_bound &&
                   (rows_out || t_out) && p >= 0.f && p < 1.f && aligned16(go) && aligned16(rows_out),
               "act_mask_s16: bad argument (needs C %% 64 == 0, 16-byte aligned rows)");
  int rc = check_t("act_mask_s16", t_out, ld_t, 1, M);
  if (rc) return rc;
  TOut t{(float*)t_out, ld_t, 1};
  const size_t lds = t_out ? (size_t)64 * TPITCH * 4 : 0;
  const int64_t ntiles = (M + 63) / 64, gx = C / 64;
  int64_t per_block = gx * ntiles / 2048;
  per_block = per_block < 1 ? 1 : (per_block > 8 ? 8 : per_block);
  int64_t gy = (ntiles + per_block - 1) / per_block;
  gy = gy > 65535 ? 65535 : gy;
  DropP d{};
  VP3D_LAUNCH((k_bn_bwd_apply_s16<true, true>), dim3((unsigned)gx, (unsigned)gy), dim3(256), lds, (hipStream_t)stream, (int)M,
                     C, go, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, d, act_bits, 1.0f / (1.0f - p), (const float*)nullptr, (const float*)nullptr, go_bound,
                     (float*)rows_out, t, g_bound);
  return check_launch("act_mask_s16");
}

int vp3d_gather_t_s16(vp3d_stream_t stream, const vp3d_rowmap* map, const void* x, int32_t C, void* t_out, int64_t ld_t) {
  VP3D_REQUIRE(map && x && t_out && C > 0 && C % 64 == 0 && map->batch > 0 && map->t_dst > 0 && map->t_src > 0 && map->taps >= 1 &&
                   aligned16(x),
               "gather_t_s16: bad argument (needs C %% 64 == 0, 16-byte aligned S16 rows)");
  const int64_t M = (int64_t)map->batch * map->t_dst;
  VP3D_REQUIRE(M < ((int64_t)1 << 31) && (M + 63) / 64 <= 65535 * (int64_t)64 && map->taps <= 65535,
               "gather_t_s16: too many rows (M=%lld)", (long long)M);
  int rc = check_t("gather_t_s16", t_out, ld_t, 1, M);
  if (rc) return rc;
  VP3D_REQUIRE((M + 63) / 64 <= 2147483647 / 1, "gather_t_s16: grid");
  TOut t{(float*)t_out, ld_t, 1};
  const int64_t tiles = (M + 63) / 64;
  VP3D_REQUIRE(tiles <= 65535, "gather_t_s16: more than 65535 row tiles (M=%lld)", (long long)M);
  VP3D_LAUNCH(k_gather_t_s16, dim3(C / 64, (unsigned)tiles, map->taps), dim3(256), (size_t)64 * TPITCH * 4,
                     (hipStream_t)stream, (int)M, C, map->t_dst, map->t_src, map->t_stride, map->tap_step, map->t_off,
                     (const float*)x, t);
  return check_launch("gather_t_s16");
}

int vp3d_sum_slices(vp3d_stream_t stream, int64_t n, int32_t splits, const float* ws, double* out) {
  VP3D_REQUIRE(n > 0 && splits > 0 && ws && out, "sum_slices: bad argument");
  VP3D_LAUNCH(k_sum_slices, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, n, splits, ws, out);
  return check_launch("sum_slices");
}

// ---- expand layer: BatchNorm coefficients from the centred second-moment matrix of X (k_expand_gram_s16) ------------------------
// block = 8 channels x 32 threads.  G [kpad][kpad] doubles: G[i][j] = sum_m (x_i - o_i)(x_j - o_j), row `one` = column sums of the
// shifted data, G[one][one] = M (vp3d_expand_s16.hip).  Cov_ij = G_ij / M - s_i s_j / M^2 goes to LDS once per block; channel n:
// mean = W[n] . (o + s / M),  var = W[n]^T Cov W[n]  (fp64), then exactly k_bn_finalize's outputs.
static __global__ void __launch_bounds__(256) k_expand_stats_fin(int C, int kpad, int kv, int one, double M, const double* __restrict__ G,
                                                                 const float* __restrict__ xt, int64_t ld_t,
                                                                 const float* __restrict__ x_bound, const float* __restrict__ wp,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, float momentum_arg, const float* __restrict__ momentum_dev,
                                                                 float* running_mean, float* running_var, int64_t* nbt, float* scale,
                                                                 float* shift, float* save_mean, float* save_invstd, int32_t* illcond) {
  // LDS: S [kv][kpad] doubles (rows 0 .. kv-1 of G, then Cov in place), sv [kpad] (row `one` of G: the column sums), mx [kpad]
  // (mean of x), then the block's 8 weight rows [8][kpad] floats.
  // The kernel is 128 blocks of dependent latency chains, not bandwidth: G and the weights were written by kernels on other XCDs,
  // so every batch of loads pays a trip past this XCD's L2 (~2 us).  Everything is therefore fetched in TWO bursts of 16-byte
  // loads per thread (the first versions fetched entry by entry, 41 dependent batches: 40 us; then 6 + 7 batches: 32 us).
  extern __shared__ double S[];
  double* sv = S + (size_t)kv * kpad;
  double* mx = sv + kpad;
  float* w_s = reinterpret_cast<float*>(mx + kpad);
  const float momentum = momentum_dev != nullptr ? momentum_dev[0] : momentum_arg;
  const double inv_m = 1.0 / M;
  const float xs = s16_pow2(s16_exp_of(x_bound));        // X^T holds x * 2^-e_x
  const int tid = threadIdx.x;
  {
    const int n2 = kv * kpad / 2;                        // double2 units of the first kv rows (kpad is even)
    const double2* src = reinterpret_cast<const double2*>(G);
    double2* dst = reinterpret_cast<double2*>(S);
    for (int i0 = tid; i0 < n2; i0 += 256 * 16) {
      double2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = i0 + u * 256 < n2 ? src[i0 + u * 256] : double2{0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (i0 + u * 256 < n2) dst[i0 + u * 256] = v[u];
    }
    if (tid < kpad / 2) reinterpret_cast<double2*>(sv)[tid] = reinterpret_cast<const double2*>(G + (int64_t)one * kpad)[tid];
    // the block's weight rows: 8 x kpad floats = 2 * kpad float4 (one per thread)
    const int c0 = blockIdx.x * 8;
    for (int i = tid; i < 2 * kpad; i += 256) {
      const int rr = i / (kpad / 4), q4 = i - rr * (kpad / 4);
      const float4 wv = c0 + rr < C ? reinterpret_cast<const float4*>(wp + (int64_t)(c0 + rr) * kpad)[q4] : float4{0.f, 0.f, 0.f, 0.f};
      reinterpret_cast<float4*>(w_s + rr * kpad)[q4] = wv;
    }
  }
  double o_k = 0.0;
  if (tid < kv) {
    const _Float16* g = reinterpret_cast<const _Float16*>(xt + (int64_t)tid * ld_t);
    o_k = (double)(((float)g[0] + (float)g[8]) * xs);    // the offset k_expand_gram_s16 used (real units)
  }
  __syncthreads();
  if (tid < kv) mx[tid] = o_k + sv[tid] * inv_m;
  for (int i = tid; i < kv * kv; i += 256) {             // Cov in place (entries j >= kv of a row stay raw and unused)
    const int r = i / kv, cc = i - r * kv;
    S[r * kpad + cc] = (S[r * kpad + cc] - sv[r] * sv[cc] * inv_m) * inv_m;
  }
  __syncthreads();
  // channel cl of the block, 32 threads per channel: thread t takes the columns j = t, t + 32, ... (adjacent threads read
  // adjacent doubles of a Cov row: conflict-free; the 2 channels of a wave read the same address: broadcast)
  const int cl = tid >> 5, t = tid & 31;
  const int c = blockIdx.x * 8 + cl;
  const float* wr = w_s + cl * kpad;
  // d_acc = sum_ij |w_i| |Cov_ij| |w_j|: what the relative error of the matrix (~4e-9, k_expand_gram_s16) is multiplied by on its
  // way into var; kappa = d_acc / (var + eps) is this channel's conditioning (1 for a one-tap filter, ~1e4 for a temporal
  // difference over frame-to-frame correlated keypoints).  illcond <- max over channels of floor(log2 kappa): the host
  // (range_guard) moves the layer back to the statistics pass over the conv output when that reaches GRAM_KAPPA_LOG2_MAX.
  double m_acc = 0.0, v_acc = 0.0, d_acc = 0.0;
  for (int j = t; j < kv; j += 32) {
    const double wj = (double)wr[j];
    m_acc += wj * mx[j];
    double r0 = 0.0, r1 = 0.0, a0 = 0.0, a1 = 0.0;
    int i = 0;
    for (; i + 1 < kv; i += 2) {
      const double s0 = S[i * kpad + j], s1 = S[(i + 1) * kpad + j], w0 = (double)wr[i], w1 = (double)wr[i + 1];
      r0 += s0 * w0;
      r1 += s1 * w1;
      a0 += fabs(s0) * fabs(w0);
      a1 += fabs(s1) * fabs(w1);
    }
    if (i < kv) {
      r0 += S[i * kpad + j] * (double)wr[i];
      a0 += fabs(S[i * kpad + j]) * fabs((double)wr[i]);
    }
    v_acc += wj * (r0 + r1);
    d_acc += fabs(wj) * (a0 + a1);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    m_acc += __shfl_xor(m_acc, o);
    v_acc += __shfl_xor(v_acc, o);
    d_acc += __shfl_xor(d_acc, o);
  }
  if (t == 0 && c < C) {
    const double mean = m_acc;
    double var = v_acc;
    if (var < 0.0) var = 0.0;
    if (illcond != nullptr) {
      const double kappa = d_acc / (var + (double)eps);
      if (kappa >= 2.0) atomicMax(illcond, kappa < 1e300 ? ilogb(kappa) : 1000);
    }
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * invstd);
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var != nullptr) {
      const double unbiased = var * (M / (M > 1.0 ? M - 1.0 : 1.0));
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) nbt[0] += 1;
}

// sum of the partial second-moment matrices: 16 elements x 16 slice lanes per block (k_sum_slices' 4 lanes leave each thread a
// chain of 54 loads for 216 slices: 15 us for 14 MB)
static __global__ void __launch_bounds__(256) k_gram_reduce(int n, int splits, const float* __restrict__ ws, double* __restrict__ out) {
  __shared__ double red[16][17];
  const int e = threadIdx.x & 15, l = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + e;
  double a = 0.0;
  if (i < n) {
    for (int s0 = l; s0 < splits; s0 += 64) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = s0 + 16 * u < splits ? ws[(int64_t)(s0 + 16 * u) * n + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) a += (double)v[u];
    }
  }
  red[l][e] = a;
  __syncthreads();
  if (l == 0 && i < n) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][e];        // fixed order
    out[i] = t;
  }
}

int vp3d_expand_stats_gram_groups(int64_t M) { return M > 0 ? expand_gram_groups(M) : 0; }

int vp3d_expand_stats_gram_s16(vp3d_stream_t stream, int64_t M, int32_t C, int32_t kpad, int32_t kv, int32_t one_col, const void* xt,
                               int64_t ld_t, const float* x_bound, const float* w_packed, float* part, double* gram,
                               const float* gamma, const float* beta, float eps, float momentum, const float* momentum_dev,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                               float* save_mean, float* save_invstd, int32_t* illcond) {
  VP3D_REQUIRE(M > 1 && M < ((int64_t)1 << 31) && C > 0 && kpad >= 32 && kpad <= 128 && kpad % 32 == 0 && kv > 0 && kv < kpad &&
                   one_col >= kv && one_col < kpad && xt && x_bound && w_packed && part && gram && gamma && beta && scale && shift &&
                   save_mean && save_invstd && aligned16(xt) && aligned16(part) && ld_t >= M && ld_t % 8 == 0 &&
                   (int64_t)kpad * ld_t * 4 < ((int64_t)1 << 31),
               "expand_stats_gram_s16: bad argument (kpad <= 128 with a constant-1 padding column, transposed X below 2 GiB)");
  const int groups = expand_gram_groups(M);
  int rc = launch_expand_gram_s16((hipStream_t)stream, M, kpad, (const float*)xt, ld_t, x_bound, one_col, groups, part);
  if (rc != VP3D_OK) return rc;
  const int64_t n = (int64_t)kpad * kpad;
  VP3D_LAUNCH(k_gram_reduce, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, (int)n, groups, part, gram);
  rc = check_launch("expand_stats_gram(sum)");
  if (rc != VP3D_OK) return rc;
  const size_t lds = ((size_t)kv * kpad + 2 * kpad) * sizeof(double) + (size_t)8 * kpad * sizeof(float);
  static bool attr_set[64] = {};
  int dev_id = 0;
  VP3D_REQUIRE(hipGetDevice(&dev_id) == hipSuccess && dev_id >= 0 && dev_id < 64, "expand_stats_gram_s16: hipGetDevice failed");
  if (!attr_set[dev_id]) {
    const hipError_t st = hipFuncSetAttribute((const void*)k_expand_stats_fin, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              160 * 1024 - 4096);
    if (st != hipSuccess) (void)hipGetLastError();
    VP3D_REQUIRE(st == hipSuccess, "expand_stats_gram_s16: cannot opt in to %zu B of dynamic LDS (%s)", lds, hipGetErrorString(st));
    attr_set[dev_id] = true;
  }
  VP3D_LAUNCH(k_expand_stats_fin, dim3((C + 7) / 8), dim3(256), lds, (hipStream_t)stream, C, kpad, kv, one_col, (double)M,
                     gram, (const float*)xt, ld_t, x_bound, w_packed, gamma, beta, eps, momentum, momentum_dev, running_mean,
                     running_var, num_batches_tracked, scale, shift, save_mean, save_invstd, illcond);
  return check_launch("expand_stats_gram(fin)");
}

int vp3d_expand_bwd_gram_s16(vp3d_stream_t stream, int32_t C, int32_t c_in, int32_t taps, int32_t kpad, int32_t one_col, int64_t M,
                             int32_t splits, const float* p_partials, const double* gram, const void* x_t, int64_t ld_t,
                             const float* x_bound, const float* w_packed, const float* scale, const float* mean,
                             const float* invstd, float* dgamma, float* dbeta, float* dw) {
  const int kv = c_in * taps;
  VP3D_REQUIRE(C > 0 && c_in > 0 && taps > 0 && kpad % 4 == 0 && kpad <= 128 && kv < kpad && one_col >= kv && one_col < kpad &&
                   M > 0 && splits > 0 && p_partials && gram && w_packed && scale && mean && invstd && dgamma && dbeta && dw &&
                   aligned16(p_partials),
               "expand_bwd_s16: bad argument (kpad <= 128, a spare padding column for the bias)");
  VP3D_REQUIRE(x_t == nullptr || (x_bound != nullptr && ld_t >= 16 && aligned16(x_t)),
               "expand_bwd_gram_s16: the centred form needs the transposed S16 X (its first column holds the offsets) and its bound");
  const size_t lds = (size_t)kv * kpad * sizeof(double) + 8 * 128 * sizeof(float);
  // > 64 KiB of dynamic LDS needs the opt-in, and the attribute is PER DEVICE: one flag per device ordinal (a process that
  // drives several GPUs launches this on each of them); a failed opt-in is reported, not swallowed
  static bool attr_set[64] = {};
  int dev_id = 0;
  VP3D_REQUIRE(hipGetDevice(&dev_id) == hipSuccess && dev_id >= 0 && dev_id < 64, "expand_bwd_s16: hipGetDevice failed");
  if (!attr_set[dev_id]) {
    const hipError_t st = hipFuncSetAttribute((const void*)k_expand_bwd_post, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              160 * 1024 - 4096);
    if (st != hipSuccess) (void)hipGetLastError();
    VP3D_REQUIRE(st == hipSuccess, "expand_bwd_s16: cannot opt in to %zu B of dynamic LDS (%s)", lds, hipGetErrorString(st));
    attr_set[dev_id] = true;
  }
  VP3D_LAUNCH(k_expand_bwd_post, dim3((C + 7) / 8), dim3(256), lds, (hipStream_t)stream, C, kpad, kv, one_col, c_in, taps,
                     splits, p_partials, gram, w_packed, scale, mean, invstd, 1.0 / (double)M, dgamma, dbeta, dw,
                     (const float*)x_t, ld_t, x_bound);
  return check_launch("expand_bwd_s16");
}

int vp3d_expand_bwd_s16(vp3d_stream_t stream, int32_t C, int32_t c_in, int32_t taps, int32_t kpad, int32_t one_col, int64_t M,
                        int32_t splits, const float* p_partials, const double* gram, const float* w_packed,
                        const float* scale, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* dw) {
  return vp3d_expand_bwd_gram_s16(stream, C, c_in, taps, kpad, one_col, M, splits, p_partials, gram, nullptr, 0, nullptr, w_packed,
                                  scale, mean, invstd, dgamma, dbeta, dw);
}

int vp3d_bn_bwd_reduce_bits(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                            const float* mean, const float* invstd, const uint8_t* act_bits, float keep_scale,
                            float* partials, int32_t* nparts) {
  VP3D_REQUIRE(M > 0 && M < ((int64_t)1 << 31) && C > 0 && C % 64 == 0 && nparts,
               "bn_bwd_reduce_bits: bad argument (needs C %% 64 == 0)");
  const int c4 = C / 4;
  const int lpr = c4 < 256 ? c4 : 256;
  const int rpb = 256 / lpr;
  const int gx = (c4 + lpr - 1) / lpr;
  int64_t gy = (M + (int64_t)rpb * 16 - 1) / ((int64_t)rpb * 16);     // >= 16 rows per thread ...
  const int64_t cap = 512 / gx > 0 ? 512 / gx : 1;                     // ... and at most ~512 blocks (2 per CU)
  gy = gy > cap ? cap : gy;
  *nparts = (int32_t)gy;
  if (partials == nullptr) return VP3D_OK;      // size query
  VP3D_REQUIRE(go && y && mean && invstd && act_bits && aligned16(go) && aligned16(y) && aligned16(partials),
               "bn_bwd_reduce_bits: null or unaligned pointer");
  VP3D_LAUNCH(k_bn_bwd_reduce_bits, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (int)M, C, go, y, mean,
                     invstd, act_bits, keep_scale, partials, lpr, rpb);
  return check_launch("bn_bwd_reduce_bits");
}

int vp3d_bn_bwd_reduce_fin_s16(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                               const float* mean, const float* invstd, const uint8_t* act_bits, float p,
                               const float* scale, const float* go_bound, float* partials, double* group_partials,
                               int32_t* tickets, float* dgamma, float* dbeta, float* dy_bound, int32_t* nparts,
                               int32_t* ngroups, int32_t* ntickets) {
  VP3D_REQUIRE(M > 0 && M < ((int64_t)1 << 31) && C > 0 && C % 64 == 0 && nparts && ngroups && ntickets && p >= 0.f && p < 1.f,
               "bn_bwd_reduce_fin_s16: bad argument (needs C %% 64 == 0)");
  (void)group_partials;                                // (single-stage fold: no second workspace any more)
  const int strips = C / 64;
  // row blocks: >= 4 steps of 16 rows per block, ~512 blocks at most (2 per CU), <= RED_MAX_R partial rows per strip
  int64_t R = (M + 63) / 64;
  const int64_t cap = 512 / strips > 0 ? 512 / strips : 1;
  R = R > cap ? cap : R;
  R = R > RED_MAX_R ? RED_MAX_R : R;
  const int rows_per = (int)((M + R - 1) / R);
  R = (M + rows_per - 1) / rows_per;
  *nparts = (int32_t)R;
  *ngroups = 0;
  *ntickets = strips;
  if (partials == nullptr) return VP3D_OK;             // size query
  VP3D_REQUIRE(go && y && mean && invstd && act_bits && scale && go_bound && tickets && dgamma && dbeta && dy_bound &&
                   aligned16(go) && aligned16(y) && aligned16(partials),
               "bn_bwd_reduce_fin_s16: null or unaligned pointer");
  BwdFin fin;
  fin.cnt = tickets;
  fin.dgamma = dgamma;
  fin.dbeta = dbeta;
  fin.scale = scale;
  fin.go_bound = go_bound;
  fin.dy_bound = dy_bound;
  fin.inv_keep = 1.0f / (1.0f - p);
  fin.inv_m = 1.0f / (float)M;
  fin.sqrt_m1 = sqrtf((float)(M > 1 ? M - 1 : 1));
  VP3D_LAUNCH(k_bn_bwd_reduce_strips, dim3(strips, (unsigned)R), dim3(256), 0, (hipStream_t)stream, (int)M, C, rows_per,
                     go, y, mean, invstd, act_bits, partials, fin);
  return check_launch("bn_bwd_reduce_fin_s16");
}

int vp3d_split_t(vp3d_stream_t stream, int64_t M, int32_t C, const float* src, int64_t ld_src, const float* bound,
                 void* out, int64_t ld_out, void* t_out, int64_t ld_t) {
  VP3D_REQUIRE(M > 0 && M < ((int64_t)1 << 31) && C > 0 && C % 64 == 0 && src && (out || t_out) && ld_src % 4 == 0 &&
                   aligned16(src) && (out == nullptr || (aligned16(out) && ld_out % 8 == 0)),
               "split_t: bad argument (needs C %% 64 == 0, 16-byte aligned rows)");
  int rc = check_t("split_t", t_out, ld_t, 1, M);
  if (rc) return rc;
  TOut t{(float*)t_out, ld_t, 1};
  VP3D_REQUIRE((M + 63) / 64 <= 65535, "split_t: more than 65535 row tiles (M=%lld)", (long long)M);
  const size_t lds = t_out ? (size_t)64 * TPITCH * 4 : 0;
  VP3D_LAUNCH(k_split_t<false>, dim3(C / 64, (unsigned)((M + 63) / 64)), dim3(256), lds, (hipStream_t)stream, (int)M, C,
                     src, ld_src, bound, (float*)out, ld_out, t, Im2Row{});
  return check_launch("split_t");
}

int vp3d_im2row_split_s16(vp3d_stream_t stream, const vp3d_rowmap* map, const float* x, int32_t ldx, int32_t k_valid, int32_t kpad,
                          int32_t one_col, const float* bound, void* out, void* t_out, int64_t ld_t) {
  VP3D_REQUIRE(map && x && bound && (out || t_out), "im2row_split_s16: null pointer");
  VP3D_REQUIRE(one_col < 0 || (one_col >= k_valid && one_col < kpad), "im2row_split_s16: the bias column must be a padding column");
  VP3D_REQUIRE(map->batch > 0 && map->t_dst > 0 && map->t_src > 0 && k_valid > 0 && kpad >= k_valid && kpad % 64 == 0 &&
                   (out == nullptr || aligned16(out)),
               "im2row_split_s16: bad sizes (k_valid=%d kpad=%d: the S16 split works on 64-column tiles)", k_valid, kpad);
  VP3D_REQUIRE((int64_t)(map->t_dst - 1) * map->t_stride * ldx + k_valid <= (int64_t)map->t_src * ldx,
               "im2row_split_s16: rows run past the end of a sample");
  const int64_t M = (int64_t)map->batch * map->t_dst;
  VP3D_REQUIRE(M < ((int64_t)1 << 31) && (M + 63) / 64 <= 65535, "im2row_split_s16: too many rows (M=%lld)", (long long)M);
  int rc = check_t("im2row_split_s16", t_out, ld_t, 1, M);
  if (rc) return rc;
  TOut t{(float*)t_out, ld_t, 1};
  const size_t lds = t_out ? (size_t)64 * TPITCH * 4 : 0;
  const Im2Row g{map->t_dst, map->t_src, map->t_stride, ldx, k_valid, one_col < 0 ? -1 : one_col, make_fastdiv(map->t_dst)};
  VP3D_LAUNCH(k_split_t<true>, dim3(kpad / 64, (unsigned)((M + 63) / 64)), dim3(256), lds, (hipStream_t)stream, (int)M, kpad,
                     x, (int64_t)0, bound, (float*)out, (int64_t)kpad, t, g);
  return check_launch("im2row_split_s16");
}

int vp3d_pack_weight_s16(vp3d_stream_t stream, const float* w, int32_t c_out, int32_t c_in, int32_t taps,
                         const float* bound, void* wf, int64_t ld_f, void* wd, int64_t ld_d, int32_t dilated_form) {
  VP3D_REQUIRE(w && bound && (wf || wd) && c_out > 0 && c_in > 0 && taps >= 1 && taps <= 8 && c_out % 64 == 0 && c_in % 64 == 0,
               "pack_weight_s16: bad argument (needs c_out, c_in %% 64 == 0)");
  VP3D_REQUIRE((wf == nullptr || (aligned16(wf) && ld_f % 8 == 0 && ld_f >= (int64_t)taps * c_in)) &&
                   (wd == nullptr || (aligned16(wd) && ld_d % 8 == 0 && ld_d >= (dilated_form ? (int64_t)taps * c_out : c_out))),
               "pack_weight_s16: bad pitches");
  const size_t lds = (size_t)taps * 64 * TPITCH * 4;
  VP3D_LAUNCH(k_pack_weight_s16, dim3(c_in / 64, c_out / 64), dim3(256), lds, (hipStream_t)stream, c_out, c_in, taps,
                     w, bound, (float*)wf, ld_f, (float*)wd, ld_d, dilated_form);
  return check_launch("pack_weight_s16");
}

int vp3d_act_bound(vp3d_stream_t stream, int32_t C, int64_t M, const float* gamma, const float* beta, float p,
                   const float* res_bound, float* out) {
  VP3D_REQUIRE(C > 0 && M > 0 && gamma && beta && out && p >= 0.f && p < 1.f, "act_bound: bad argument");
  VP3D_LAUNCH(k_act_bound, dim3(1), dim3(1024), 0, (hipStream_t)stream, C, sqrtf((float)(M > 1 ? M - 1 : 1)), gamma,
                     beta, 1.0f / (1.0f - p), res_bound, out);
  return check_launch("act_bound");
}

int vp3d_dy_bound(vp3d_stream_t stream, int32_t C, int64_t M, const float* scale, const float* dgamma,
                  const float* dbeta, const float* go_bound, float p, float* out) {
  VP3D_REQUIRE(C > 0 && M > 0 && scale && dgamma && dbeta && go_bound && out && p >= 0.f && p < 1.f, "dy_bound: bad argument");
  VP3D_LAUNCH(k_dy_bound, dim3(1), dim3(1024), 0, (hipStream_t)stream, C, 1.0f / (float)M,
                     sqrtf((float)(M > 1 ? M - 1 : 1)), scale, dgamma, dbeta, go_bound, 1.0f / (1.0f - p), out);
  return check_launch("dy_bound");
}

int vp3d_amax_multi(vp3d_stream_t stream, int32_t n_tensors, const float* const* src, const int64_t* n, float* bounds) {
  VP3D_REQUIRE(n_tensors > 0 && n_tensors <= kMaxLayers && src && n && bounds, "amax_multi: bad argument (at most %d tensors)",
               kMaxLayers);
  AmaxMulti a;
  int64_t nmax = 0;
  for (int i = 0; i < kMaxLayers; ++i) {
    a.src[i] = i < n_tensors ? src[i] : nullptr;
    a.n[i] = i < n_tensors ? n[i] : 0;
    if (i < n_tensors) {
      VP3D_REQUIRE(src[i] != nullptr && n[i] > 0, "amax_multi: tensor %d is empty", i);
      nmax = n[i] > nmax ? n[i] : nmax;
    }
  }
  a.bounds = bounds;
  int64_t blocks = (nmax + 256 * 32 - 1) / (256 * 32);           // >= 8 float4 per thread of the largest tensor
  blocks = blocks < 512 ? blocks : 512;
  VP3D_LAUNCH(k_amax_multi, dim3((unsigned)blocks, n_tensors), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("amax_multi");
}

int vp3d_pack_weight_s16_multi(vp3d_stream_t stream, int32_t n_layers, const float* const* w, const int32_t* taps,
                               int32_t c_out, int32_t c_in, const float* bounds, void* const* wf, void* const* wd) {
  VP3D_REQUIRE(n_layers > 0 && n_layers <= kMaxLayers && w && taps && bounds && wf && wd && c_out > 0 && c_in > 0 &&
                   c_out % 64 == 0 && c_in % 64 == 0,
               "pack_weight_s16_multi: bad argument (at most %d layers, channels %% 64 == 0)", kMaxLayers);
  PackMulti a;
  int tmax = 1;
  for (int i = 0; i < kMaxLayers; ++i) {
    a.w[i] = i < n_layers ? w[i] : nullptr;
    a.wf[i] = i < n_layers ? (float*)wf[i] : nullptr;
    a.wd[i] = i < n_layers ? (float*)wd[i] : nullptr;
    a.taps[i] = i < n_layers ? taps[i] : 1;
    if (i < n_layers) {
      VP3D_REQUIRE(w[i] && taps[i] >= 1 && taps[i] <= 3 && (wf[i] || wd[i]) && aligned16(wf[i]) && aligned16(wd[i]),
                   "pack_weight_s16_multi: layer %d (taps 1..3, 16-byte aligned outputs)", i);
      tmax = taps[i] > tmax ? taps[i] : tmax;
    }
  }
  a.bounds = bounds;
  a.c_out = c_out;
  a.c_in = c_in;
  VP3D_LAUNCH(k_pack_weight_s16_multi, dim3(c_in / 64, c_out / 64, n_layers), dim3(256), (size_t)tmax * 64 * TPITCH * 4,
                     (hipStream_t)stream, a);
  return check_launch("pack_weight_s16_multi");
}

int vp3d_act_bounds_multi(vp3d_stream_t stream, int32_t n_layers, int32_t C, const float* const* gamma,
                          const float* const* beta, const int64_t* M, const int32_t* res_from, float p, float* bounds) {
  VP3D_REQUIRE(n_layers > 0 && n_layers <= kMaxLayers && C > 0 && gamma && beta && M && res_from && bounds && p >= 0.f && p < 1.f,
               "act_bounds_multi: bad argument (at most %d layers)", kMaxLayers);
  ActBoundsMulti a;
  for (int i = 0; i < kMaxLayers; ++i) {
    a.gamma[i] = i < n_layers ? gamma[i] : nullptr;
    a.beta[i] = i < n_layers ? beta[i] : nullptr;
    a.sqrt_m1[i] = i < n_layers ? sqrtf((float)(M[i] > 1 ? M[i] - 1 : 1)) : 0.f;
    a.res_from[i] = i < n_layers ? res_from[i] : -1;
    if (i < n_layers) VP3D_REQUIRE(gamma[i] && beta[i] && res_from[i] < i, "act_bounds_multi: layer %d", i);
  }
  a.bounds = bounds;
  a.n_layers = n_layers;
  a.C = C;
  a.inv_keep = 1.0f / (1.0f - p);
  VP3D_LAUNCH(k_act_bounds_multi, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return check_launch("act_bounds_multi");
}

int vp3d_prologue_a_s16(vp3d_stream_t stream, int32_t n_tensors, const float* const* src, const int64_t* n, float* const* bound,
                        const float* floor_, int32_t n_layers, int32_t C, const float* const* gamma, const float* const* beta,
                        const int64_t* M, const int32_t* res_from, float p, float* act_bounds) {
  VP3D_REQUIRE(n_tensors > 0 && n_tensors <= kMaxLayers && src && n && bound && n_layers >= 0 && n_layers <= kMaxLayers &&
                   (n_layers == 0 || (C > 0 && gamma && beta && M && res_from && act_bounds)) && p >= 0.f && p < 1.f,
               "prologue_a_s16: bad argument (at most %d tensors / layers)", kMaxLayers);
  PrologueA a{};
  int64_t nmax = 0;
  for (int i = 0; i < n_tensors; ++i) {
    VP3D_REQUIRE(src[i] && n[i] > 0 && bound[i], "prologue_a_s16: tensor %d", i);
    a.src[i] = src[i];
    a.n[i] = n[i];
    a.bound[i] = bound[i];
    a.floor_[i] = floor_ ? floor_[i] : 0.f;
    nmax = n[i] > nmax ? n[i] : nmax;
  }
  a.n_tensors = n_tensors;
  for (int i = 0; i < n_layers; ++i) {
    VP3D_REQUIRE(gamma[i] && beta[i] && res_from[i] < i, "prologue_a_s16: layer %d", i);
    a.ab.gamma[i] = gamma[i];
    a.ab.beta[i] = beta[i];
    a.ab.sqrt_m1[i] = sqrtf((float)(M[i] > 1 ? M[i] - 1 : 1));
    a.ab.res_from[i] = res_from[i];
  }
  a.ab.bounds = act_bounds;
  a.ab.n_layers = n_layers;
  a.ab.C = C;
  a.ab.inv_keep = 1.0f / (1.0f - p);
  int64_t blocks = (nmax + 256 * 32 - 1) / (256 * 32);
  blocks = blocks < 512 ? blocks : 512;
  VP3D_LAUNCH(k_prologue_a, dim3((unsigned)blocks, n_tensors + 1), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("prologue_a_s16");
}

int vp3d_prologue_b_s16(vp3d_stream_t stream, const vp3d_rowmap* map, const float* x, int32_t ldx, int32_t k_valid, int32_t kpad,
                        int32_t one_col, const float* x_bound, void* x_rows, void* x_t, int64_t ld_t, const float* w0, int32_t c0,
                        int32_t cin0, int32_t taps0, const float* w0_bound, float* w0_packed, void* w0_s16, int32_t n_layers,
                        const float* const* w, const int32_t* taps, int32_t c_out, int32_t c_in, const float* w_bounds,
                        void* const* wf, void* const* wd) {
  VP3D_REQUIRE(map && x && x_bound && x_rows && w0 && w0_bound && w0_packed && w0_s16, "prologue_b_s16: null pointer");
  VP3D_REQUIRE(one_col < 0 || (one_col >= k_valid && one_col < kpad), "prologue_b_s16: the bias column must be a padding column");
  VP3D_REQUIRE(map->batch > 0 && map->t_dst > 0 && map->t_src > 0 && k_valid > 0 && kpad >= k_valid && kpad % 64 == 0 && kpad <= 128 &&
                   aligned16(x_rows) && aligned16(w0_packed) && aligned16(w0_s16) && c0 > 0 && cin0 > 0 && taps0 > 0 &&
                   cin0 * taps0 == k_valid,
               "prologue_b_s16: bad sizes (k_valid=%d kpad=%d)", k_valid, kpad);
  VP3D_REQUIRE((int64_t)(map->t_dst - 1) * map->t_stride * ldx + k_valid <= (int64_t)map->t_src * ldx,
               "prologue_b_s16: rows run past the end of a sample");
  const int64_t M = (int64_t)map->batch * map->t_dst;
  VP3D_REQUIRE(M < ((int64_t)1 << 31) && (M + 63) / 64 <= 65535 * (int64_t)16, "prologue_b_s16: too many rows (M=%lld)", (long long)M);
  int rc = check_t("prologue_b_s16", x_t, ld_t, 1, M);
  if (rc) return rc;
  VP3D_REQUIRE(n_layers >= 0 && n_layers <= kMaxLayers && (n_layers == 0 || (w && taps && w_bounds && wf && wd && c_out > 0 && c_in > 0 &&
                                                                              c_out % 64 == 0 && c_in % 64 == 0)),
               "prologue_b_s16: bad weight-pack argument");
  PrologueB a{};
  a.M = (int)M; a.kpad = kpad; a.n_in = (kpad / 64) * (int)((M + 63) / 64);
  a.x = x; a.x_bound = x_bound; a.x_rows = (float*)x_rows;
  a.x_t = TOut{(float*)x_t, ld_t, 1};
  a.g = Im2Row{map->t_dst, map->t_src, map->t_stride, ldx, k_valid, one_col < 0 ? -1 : one_col, make_fastdiv(map->t_dst)};
  const int rows_per = 256 / (kpad / 8);
  a.n_w0 = (c0 + rows_per - 1) / rows_per; a.c0 = c0; a.cin0 = cin0; a.taps0 = taps0;
  a.w0 = w0; a.w0_bound = w0_bound; a.w0_packed = w0_packed; a.w0_s16 = (float*)w0_s16;
  int tmax = 1;
  for (int i = 0; i < n_layers; ++i) {
    VP3D_REQUIRE(w[i] && taps[i] >= 1 && taps[i] <= 3 && (wf[i] || wd[i]) && aligned16(wf[i]) && aligned16(wd[i]),
                 "prologue_b_s16: layer %d (taps 1..3, 16-byte aligned outputs)", i);
    a.pk.w[i] = w[i];
    a.pk.wf[i] = (float*)wf[i];
    a.pk.wd[i] = (float*)wd[i];
    a.pk.taps[i] = taps[i];
    tmax = taps[i] > tmax ? taps[i] : tmax;
  }
  a.pk.bounds = w_bounds;
  a.pk.c_out = c_out;
  a.pk.c_in = c_in;
  a.pk_layers = n_layers;
  const int n_pk = n_layers ? (c_in / 64) * (c_out / 64) * n_layers : 0;
  const size_t lds = (size_t)(n_layers ? tmax : 1) * 64 * TPITCH * 4;
  VP3D_LAUNCH(k_prologue_b, dim3((unsigned)(a.n_in + a.n_w0 + n_pk)), dim3(256), lds, (hipStream_t)stream, a);
  return check_launch("prologue_b_s16");
}

int vp3d_bn_bwd_finalize_s16(vp3d_stream_t stream, int32_t C, int64_t M, const float* partials, int32_t nparts,
                             float* dgamma, float* dbeta, const float* scale, const float* go_bound, float p,
                             float* dy_bound) {
  VP3D_REQUIRE(C > 0 && M > 0 && nparts > 0 && partials && dgamma && dbeta && scale && go_bound && dy_bound && p >= 0.f && p < 1.f,
               "bn_bwd_finalize_s16: bad argument");
  VP3D_LAUNCH(k_bn_bwd_finalize_bound, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_GROUPS), 0, (hipStream_t)stream, C,
                     partials, nparts, dgamma, dbeta, scale, go_bound, 1.0f / (1.0f - p), 1.0f / (float)M,
                     sqrtf((float)(M > 1 ? M - 1 : 1)), dy_bound);
  return check_launch("bn_bwd_finalize_s16");
}

}  // extern "C"
