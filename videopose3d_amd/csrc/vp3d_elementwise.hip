// HBM-bound kernels around the GEMMs: BatchNorm statistics / apply / backward, dropout (Philox, never stored),
// weight packing / BN folding, split-K reduction of wgrad, column sums, camera projection.
// All streaming kernels move 16 B per lane (float4) when the channel count allows it (C % 4 == 0).
#include <cstdlib>

#include "vp3d_internal.h"
#include "vp3d_dropout.h"

namespace vp3d {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// BatchNorm statistics finalize (training): merge the GEMM's 64-row slab partials in fp64.
// ---------------------------------------------------------------------------------------------------------
// Geometry of the two finalize kernels: a block owns FIN_CH channels and splits the partial rows over FIN_GROUPS
// thread groups (1024 threads), so C/16 = 64 blocks x 1024 threads keep enough loads in flight to stream the
// [parts][C] arrays (up to ~10 MB for the M = 82,944 expand layer) in a few microseconds instead of the ~40 us a
// 32-block x 256-thread version needed (latency-bound).
constexpr int FIN_CH = 16, FIN_GROUPS = 64;

__global__ void __launch_bounds__(FIN_CH * FIN_GROUPS) k_bn_finalize(
    int C, int64_t M, int nslab, int slab_rows, const float* __restrict__ psum, const float* __restrict__ pm2,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum_arg,
    const float* __restrict__ momentum_dev, float* running_mean, float* running_var, int64_t* nbt, float* scale, float* shift,
    float* save_mean, float* save_invstd) {
  __shared__ double s1[FIN_GROUPS][FIN_CH], s2[FIN_GROUPS][FIN_CH];
  // momentum from device memory when given: a captured launch (hipGraph replay) then follows run.py's per-epoch
  // set_bn_momentum without a re-capture
  const float momentum = momentum_dev != nullptr ? momentum_dev[0] : momentum_arg;
  const int cl = threadIdx.x % FIN_CH, g = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  double a1 = 0.0, a2 = 0.0;
  if (c < C) {
    // 8 slabs = 16 loads in flight per thread (round 4: the partial rows were written by GEMM workgroups on other XCDs, so a
    // batch of loads is a ~2-us trip past this XCD's L2 and the kernel is the number of dependent batches: `#pragma unroll 4`
    // gave 4 of them for the 864 32-row slabs of a 27,648-row layer = 17 us, 2 for 432 slabs = 7 us).  Same summation order.
    for (int s0 = g; s0 < nslab; s0 += FIN_GROUPS * 8) {
      float ps[8], pq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u * FIN_GROUPS;
        ps[u] = s < nslab ? psum[(int64_t)s * C + c] : 0.f;
        pq[u] = s < nslab ? pm2[(int64_t)s * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u * FIN_GROUPS;
        if (s < nslab) {
          const int64_t left = M - (int64_t)s * slab_rows;
          const double cnt = (double)(left < slab_rows ? left : slab_rows);
          const double sum = (double)ps[u];
          a1 += sum;
          a2 += (double)pq[u] + sum * sum / cnt;
        }
      }
    }
  }
  s1[g][cl] = a1;
  s2[g][cl] = a2;
  __syncthreads();
  for (int o = FIN_GROUPS / 2; o >= 1; o >>= 1) {      // fixed-shape tree: deterministic
    if (g < o) {
      s1[g][cl] += s1[g + o][cl];
      s2[g][cl] += s2[g + o][cl];
    }
    __syncthreads();
  }
  if (g == 0 && c < C) {
    const double S1 = s1[0][cl], S2 = s2[0][cl];
    const double mean = S1 / (double)M;
    double var = (S2 - S1 * mean) / (double)M;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * invstd);
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var != nullptr) {
      const double unbiased = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) nbt[0] += 1;
}

// The same with 16-byte loads: a block owns 16 channels as 4 lanes x float4 and splits the partial rows over 128 groups (512
// threads).  The scalar kernel above issues one 4-byte load per (slab, channel): 27 load instructions per thread for the 864
// 32-row slabs of a 27,648-row layer, each touching four half-used cache lines per wave -- 17-22 us on the forward's dependent
// chain, proportional to the slab count (7 us for 432 slabs), whatever the number of loads in flight.  Summation order: slab
// s = g, g + 128, ... per group, then a fixed tree over the groups (deterministic; not the scalar kernel's order).
constexpr int FIN4_GROUPS = 128;
__global__ void __launch_bounds__(4 * FIN4_GROUPS) k_bn_finalize_v4(
    int C, int64_t M, int nslab, int slab_rows, const float* __restrict__ psum, const float* __restrict__ pm2,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum_arg,
    const float* __restrict__ momentum_dev, float* running_mean, float* running_var, int64_t* nbt, float* scale, float* shift,
    float* save_mean, float* save_invstd) {
  __shared__ double s1[FIN4_GROUPS][16], s2[FIN4_GROUPS][16];
  const float momentum = momentum_dev != nullptr ? momentum_dev[0] : momentum_arg;
  const int q = threadIdx.x & 3, g = threadIdx.x >> 2;
  const int c4 = blockIdx.x * 16 + q * 4;                  // (C % 16 == 0: every quad exists)
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
  for (int s0 = g; s0 < nslab; s0 += FIN4_GROUPS * 4) {
    f32x4 ps[4], pq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + u * FIN4_GROUPS;
      ps[u] = s < nslab ? *reinterpret_cast<const f32x4*>(psum + (int64_t)s * C + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      pq[u] = s < nslab ? *reinterpret_cast<const f32x4*>(pm2 + (int64_t)s * C + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + u * FIN4_GROUPS;
      if (s < nslab) {
        const int64_t left = M - (int64_t)s * slab_rows;
        const double inv_cnt = 1.0 / (double)(left < slab_rows ? left : slab_rows);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double sum = (double)ps[u][e];
          a1[e] += sum;
          a2[e] += (double)pq[u][e] + sum * sum * inv_cnt;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s1[g][q * 4 + e] = a1[e];
    s2[g][q * 4 + e] = a2[e];
  }
  __syncthreads();
  // fixed-shape tree over the groups: thread (g, q) folds its 4 channels
  for (int o = FIN4_GROUPS / 2; o >= 1; o >>= 1) {
    if (g < o) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[g][q * 4 + e] += s1[g + o][q * 4 + e];
        s2[g][q * 4 + e] += s2[g + o][q * 4 + e];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 16) {
    const int cl = threadIdx.x, c = blockIdx.x * 16 + cl;
    const double S1 = s1[0][cl], S2 = s2[0][cl];
    const double mean = S1 / (double)M;
    double var = (S2 - S1 * mean) / (double)M;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * invstd);
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var != nullptr) {
      const double unbiased = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) nbt[0] += 1;
}

__global__ void k_bn_fold(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                          float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float s = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - rm[c] * s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// out = [res +] dropout(relu(y*scale + shift))
// ---------------------------------------------------------------------------------------------------------
struct ResMap {
  const float* res;
  int t_dst, r_t, r_stride, r_off, r_ld;
};

// Thread layout of the streaming BN kernels ("column-owned"): a thread owns VEC consecutive channels (its
// per-channel coefficients stay in registers) and walks rows m = blockIdx.y*rows_per_block + rsub, += gridDim.y *
// rows_per_block; a wave covers 64*VEC consecutive channels of one row (1 KiB, coalesced).  No divisions per element.
template <int VEC>
__global__ void __launch_bounds__(256) k_bn_act_fwd(int M, int C, const float* __restrict__ y,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    DropP d, ResMap rm, float* __restrict__ out, int lanes_per_row,
                                                    int rows_per_block) {
  drop_resolve(d);
  const int lr = threadIdx.x % lanes_per_row;
  const int rsub = threadIdx.x / lanes_per_row;
  const int c = (blockIdx.x * lanes_per_row + lr) * VEC;
  if (rsub >= rows_per_block || c >= C) return;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sc[e] = scale[c + e];
    sh[e] = shift[c + e];
  }
  const int row_step = gridDim.y * rows_per_block;
#pragma unroll 2
  for (int m = blockIdx.y * rows_per_block + rsub; m < M; m += row_step) {
    const int64_t e0 = (int64_t)m * C + c;
    const float* rrow = nullptr;
    if (rm.res != nullptr) {
      const int b = m / rm.t_dst;
      const int t = m - b * rm.t_dst;
      rrow = rm.res + ((int64_t)b * rm.r_t + (int64_t)t * rm.r_stride + rm.r_off) * rm.r_ld + c;
    }
    float mk[4] = {1.f, 1.f, 1.f, 1.f};
    if (d.on) drop4(d, (uint64_t)(e0 >> 2), mk);
    if (VEC == 4) {
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + e0);
      f32x4 rv = {0.f, 0.f, 0.f, 0.f};
      if (rrow != nullptr) rv = *reinterpret_cast<const f32x4*>(rrow);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaf(yv[e], sc[e], sh[e]);
        o[e] = rv[e] + (z > 0.f ? z * mk[e] : (z != z ? z : 0.f));
      }
      *reinterpret_cast<f32x4*>(out + e0) = o;
    } else {
      const float z = fmaf(y[e0], sc[0], sh[0]);
      out[e0] = (rrow != nullptr ? rrow[0] : 0.f) + (z > 0.f ? z * mk[e0 & 3] : (z != z ? z : 0.f));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// BN backward, pass 1: per-channel partial sums of g and g*xhat.   Thread owns VEC channels and strides rows;
// partials[part][0][c] = sum g, partials[part][1][c] = sum g*xhat.
// Launch: grid.x = channel strips of 256*VEC, grid.y = row groups; part = blockIdx.y * rows_per_block + rsub.
// ---------------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_bn_bwd_reduce(int M, int C, const float* __restrict__ go,
                                                       const float* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, DropP d,
                                                       float* __restrict__ partials, int lanes_per_row,
                                                       int rows_per_block) {
  drop_resolve(d);
  const int lr = threadIdx.x % lanes_per_row;
  const int rsub = threadIdx.x / lanes_per_row;
  const int c = (blockIdx.x * lanes_per_row + lr) * VEC;
  if (rsub >= rows_per_block || c >= C) return;
  float sg[VEC], sgx[VEC], sc[VEC], sh[VEC], mu[VEC], is[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sg[e] = sgx[e] = 0.f;
    sc[e] = scale[c + e]; sh[e] = shift[c + e]; mu[e] = mean[c + e]; is[e] = invstd[c + e];
  }
  const int row_step = gridDim.y * rows_per_block;
#pragma unroll 2
  for (int m = blockIdx.y * rows_per_block + rsub; m < M; m += row_step) {
    const int64_t e0 = (int64_t)m * C + c;
    float mk[4] = {1.f, 1.f, 1.f, 1.f};
    if (VEC == 4) {
      if (d.on) drop4(d, (uint64_t)(e0 >> 2), mk);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(go + e0);
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + e0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaf(yv[e], sc[e], sh[e]);
        const float g = z > 0.f ? gv[e] * mk[e] : 0.f;
        sg[e] += g;
        sgx[e] += g * ((yv[e] - mu[e]) * is[e]);
      }
    } else {
      float mkv = 1.f;
      if (d.on) {
        drop4(d, (uint64_t)(e0 >> 2), mk);
        mkv = mk[e0 & 3];
      }
      const float yv = y[e0];
      const float z = fmaf(yv, sc[0], sh[0]);
      const float g = z > 0.f ? go[e0] * mkv : 0.f;
      sg[0] += g;
      sgx[0] += g * ((yv - mu[0]) * is[0]);
    }
  }
  const int64_t part = (int64_t)blockIdx.y * rows_per_block + rsub;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    partials[(part * 2 + 0) * C + c + e] = sg[e];
    partials[(part * 2 + 1) * C + c + e] = sgx[e];
  }
}

__global__ void __launch_bounds__(FIN_CH * FIN_GROUPS) k_bn_bwd_finalize(int C, const float* __restrict__ partials,
                                                                         int nparts, float* dgamma, float* dbeta) {
  __shared__ double s1[FIN_GROUPS][FIN_CH], s2[FIN_GROUPS][FIN_CH];
  const int cl = threadIdx.x % FIN_CH, g = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  double a1 = 0.0, a2 = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int p = g; p < nparts; p += FIN_GROUPS) {       // fp32 partials carried in fp64
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
  if (g == 0 && c < C) {
    dbeta[c] = (float)s1[0][cl];
    dgamma[c] = (float)s2[0][cl];
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) k_bn_bwd_apply(int M, int C, const float* __restrict__ go,
                                                      const float* __restrict__ y, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, DropP d,
                                                      const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                      float* __restrict__ dy, int lanes_per_row, int rows_per_block) {
  drop_resolve(d);
  const int lr = threadIdx.x % lanes_per_row;
  const int rsub = threadIdx.x / lanes_per_row;
  const int c = (blockIdx.x * lanes_per_row + lr) * VEC;
  if (rsub >= rows_per_block || c >= C) return;
  const float inv_m = 1.0f / (float)M;
  float sc[VEC], sh[VEC], mu[VEC], is[VEC], kb[VEC], kg[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sc[e] = scale[c + e]; sh[e] = shift[c + e]; mu[e] = mean[c + e]; is[e] = invstd[c + e];
    kb[e] = dbeta[c + e] * inv_m;
    kg[e] = dgamma[c + e] * inv_m;
  }
  const int row_step = gridDim.y * rows_per_block;
#pragma unroll 2
  for (int m = blockIdx.y * rows_per_block + rsub; m < M; m += row_step) {
    const int64_t e0 = (int64_t)m * C + c;
    float mk[4] = {1.f, 1.f, 1.f, 1.f};
    if (d.on) drop4(d, (uint64_t)(e0 >> 2), mk);
    if (VEC == 4) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(go + e0);
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + e0);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaf(yv[e], sc[e], sh[e]);
        const float g = z > 0.f ? gv[e] * mk[e] : 0.f;
        const float xh = (yv[e] - mu[e]) * is[e];
        o[e] = sc[e] * (g - kb[e] - xh * kg[e]);
      }
      *reinterpret_cast<f32x4*>(dy + e0) = o;
    } else {
      const float yv = y[e0];
      const float z = fmaf(yv, sc[0], sh[0]);
      const float g = z > 0.f ? go[e0] * mk[e0 & 3] : 0.f;
      const float xh = (yv - mu[0]) * is[0];
      dy[e0] = sc[0] * (g - kb[0] - xh * kg[0]);
    }
  }
}

// dy = scale*(g - dbeta/M - xhat*dgamma/M) with g already formed by the fused dgrad epilogue (no mask / ReLU work)
template <int VEC>
__global__ void __launch_bounds__(256) k_bn_bwd_apply_g(int M, int C, const float* __restrict__ g,
                                                        const float* __restrict__ y, const float* __restrict__ scale,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                        float* __restrict__ dy, int lanes_per_row, int rows_per_block) {
  const int lr = threadIdx.x % lanes_per_row;
  const int rsub = threadIdx.x / lanes_per_row;
  const int c = (blockIdx.x * lanes_per_row + lr) * VEC;
  if (rsub >= rows_per_block || c >= C) return;
  const float inv_m = 1.0f / (float)M;
  float sc[VEC], mu[VEC], is[VEC], kb[VEC], kg[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sc[e] = scale[c + e]; mu[e] = mean[c + e]; is[e] = invstd[c + e];
    kb[e] = dbeta[c + e] * inv_m;
    kg[e] = dgamma[c + e] * inv_m;
  }
  const int row_step = gridDim.y * rows_per_block;
#pragma unroll 2
  for (int m = blockIdx.y * rows_per_block + rsub; m < M; m += row_step) {
    const int64_t e0 = (int64_t)m * C + c;
    if (VEC == 4) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + e0);
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + e0);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (yv[e] - mu[e]) * is[e];
        o[e] = sc[e] * (gv[e] - kb[e] - xh * kg[e]);
      }
      *reinterpret_cast<f32x4*>(dy + e0) = o;
    } else {
      const float xh = (y[e0] - mu[0]) * is[0];
      dy[e0] = sc[0] * (g[e0] - kb[0] - xh * kg[0]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------------------------------
// out[co*ld_out + k*c_in + ci] = w[co][ci][k] * scale[co];  columns [taps*c_in, ld_out) are zero-filled
__global__ void __launch_bounds__(256) k_pack_weight(int c_out, int c_in, int taps, const float* __restrict__ w,
                                                     const float* __restrict__ scale, float* __restrict__ out,
                                                     int ld_out) {
  const int64_t total = (int64_t)c_out * c_in;
  const int pad = ld_out - taps * c_in;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t co = i / c_in;
    const int ci = (int)(i - co * c_in);
    const float s = scale != nullptr ? scale[co] : 1.f;
    const float* src = w + i * taps;
    float* dst = out + co * (int64_t)ld_out + ci;
    for (int k = 0; k < taps; ++k) dst[(int64_t)k * c_in] = src[k] * s;
    for (int z = ci; z < pad; z += c_in) out[co * (int64_t)ld_out + taps * c_in + z] = 0.f;
  }
}

// dw[co][ci][k] = sum_s partials[s][co*ld_part + k*c_in + ci]
__global__ void __launch_bounds__(256) k_wgrad_reduce(int splits, int c_out, int c_in, int taps,
                                                      const float* __restrict__ partials, int ld_part,
                                                      float* __restrict__ dw) {
  const int64_t total = (int64_t)c_out * c_in;
  const int64_t mat = (int64_t)c_out * ld_part;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t co = i / c_in;
    const int ci = (int)(i - co * c_in);
    for (int k = 0; k < taps; ++k) {
      const float* src = partials + co * (int64_t)ld_part + (int64_t)k * c_in + ci;
      float acc = 0.f;
      for (int s = 0; s < splits; ++s) acc += src[(int64_t)s * mat];
      dw[i * taps + k] = acc;
    }
  }
}

// the same with 16-byte accesses: thread = (co, 4 consecutive ci); per tap and split one 16-byte load along ci (the loads
// of all taps and up to 4 splits are independent), then the interleaved [ci][k] run of 4*TAPS floats as 16-byte stores
template <int TAPS>
__global__ void __launch_bounds__(256) k_wgrad_reduce_v4(int splits, int c_out, int c_in, const float* __restrict__ partials,
                                                         int ld_part, float* __restrict__ dw) {
  const int cq = c_in >> 2;
  const int64_t total = (int64_t)c_out * cq;
  const int64_t mat = (int64_t)c_out * ld_part;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t co = i / cq;
    const int ci = (int)(i - co * cq) * 4;
    const float* src = partials + co * (int64_t)ld_part + ci;
    f32x4 acc[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
      f32x4 v[4][TAPS];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < TAPS; ++k) v[u][k] = *reinterpret_cast<const f32x4*>(src + (int64_t)(s + u) * mat + (int64_t)k * c_in);
#pragma unroll
      for (int u = 0; u < 4; ++u)               // same summation order as the scalar kernel
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc[k] += v[u][k];
    }
    for (; s < splits; ++s)
#pragma unroll
      for (int k = 0; k < TAPS; ++k) acc[k] += *reinterpret_cast<const f32x4*>(src + (int64_t)s * mat + (int64_t)k * c_in);
    float o[4 * TAPS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < TAPS; ++k) o[j * TAPS + k] = acc[k][j];
    f32x4* dst = reinterpret_cast<f32x4*>(dw + (co * c_in + ci) * TAPS);
#pragma unroll
    for (int q = 0; q < TAPS; ++q) dst[q] = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
  }
}

// out[n] = sum_m g[m*ld + n]
__global__ void __launch_bounds__(256) k_colsum(int64_t M, int N, const float* __restrict__ g, int ld, float* out) {
  __shared__ float red[256];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int64_t m = threadIdx.x; m < M; m += 256) s += g[m * ld + n];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = red[0];
}

// out[m][0..kpad) = the k_valid contiguous floats starting at x[(b*t_src + t*t_stride)*ldx], zero padded.
// (expand conv: taps*C_in = 102 contiguous floats per output row -> 128-wide rows the fast GEMM path can DMA)
// one_col >= k_valid: that padding column holds 1 (a "bias column": a GEMM that reduces over the rows then also yields
// the column sums of its other operand -- engine_s16's expand-layer backward reads sum(g) and colsum(x) from it)
__global__ void __launch_bounds__(256) k_im2row(int M, int t_dst, int t_src, int t_stride, int ldx, int k_valid,
                                                int kpad, int one_col, const float* __restrict__ x, float* __restrict__ out) {
  const int qpr = kpad >> 2;                          // float4 per output row
  const int64_t total = (int64_t)M * qpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / qpr);
    const int k0 = (int)(i - (int64_t)m * qpr) * 4;
    const int b = m / t_dst;
    const int t = m - b * t_dst;
    const float* src = x + ((int64_t)b * t_src + (int64_t)t * t_stride) * ldx;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k0 + e < k_valid) ? src[k0 + e] : (k0 + e == one_col ? 1.f : 0.f);
    *reinterpret_cast<f32x4*>(out + (int64_t)m * kpad + k0) = v;
  }
}

__global__ void __launch_bounds__(256) k_dropout_mask(int64_t n, DropP d, float* out) {
  drop_resolve(d);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    float mk[4] = {1.f, 1.f, 1.f, 1.f};
    if (d.on) drop4(d, (uint64_t)(e >> 2), mk);
    out[e] = mk[e & 3];
  }
}

// ---------------------------------------------------------------------------------------------------------
// camera projection (reference common/camera.py:37-67, 69-90) forward / backward wrt X
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_project_fwd(int64_t n_cam, int64_t ppc, const float* __restrict__ X,
                                                     const float* __restrict__ cam, int linear, float* __restrict__ out) {
  const int64_t total = n_cam * ppc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float* cp = cam + (i / ppc) * 9;
    const float x = X[i * 3], yv = X[i * 3 + 1], z = X[i * 3 + 2];
    const float u = x / z, v = yv / z;
    const float a = fminf(fmaxf(u, -1.f), 1.f), b = fminf(fmaxf(v, -1.f), 1.f);
    float ox, oy;
    if (linear) {
      ox = cp[0] * a + cp[2];
      oy = cp[1] * b + cp[3];
    } else {
      const float r2 = a * a + b * b;
      const float s = 1.f + (cp[4] * r2 + cp[5] * (r2 * r2) + cp[6] * (r2 * r2 * r2)) + (cp[7] * a + cp[8] * b);
      ox = cp[0] * (a * s + cp[7] * r2) + cp[2];
      oy = cp[1] * (b * s + cp[8] * r2) + cp[3];
    }
    out[i * 2] = ox;
    out[i * 2 + 1] = oy;
  }
}

__global__ void __launch_bounds__(256) k_project_bwd(int64_t n_cam, int64_t ppc, const float* __restrict__ X,
                                                     const float* __restrict__ cam, const float* __restrict__ gout,
                                                     int linear, float* __restrict__ dX) {
  const int64_t total = n_cam * ppc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float* cp = cam + (i / ppc) * 9;
    const float k1 = linear ? 0.f : cp[4], k2 = linear ? 0.f : cp[5], k3 = linear ? 0.f : cp[6];
    const float p1 = linear ? 0.f : cp[7], p2 = linear ? 0.f : cp[8];
    const float x = X[i * 3], yv = X[i * 3 + 1], z = X[i * 3 + 2];
    const float u = x / z, v = yv / z;
    const float a = fminf(fmaxf(u, -1.f), 1.f), b = fminf(fmaxf(v, -1.f), 1.f);
    const float r2 = a * a + b * b;
    const float s = 1.f + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2 + p1 * a + p2 * b;
    const float hx = cp[0] * gout[i * 2], hy = cp[1] * gout[i * 2 + 1];
    const float ds = hx * a + hy * b;
    const float dr2 = hx * p1 + hy * p2 + ds * (k1 + 2.f * k2 * r2 + 3.f * k3 * r2 * r2);
    const float da = hx * s + ds * p1 + 2.f * a * dr2;
    const float db = hy * s + ds * p2 + 2.f * b * dr2;
    const float du = (u >= -1.f && u <= 1.f) ? da : 0.f;
    const float dv = (v >= -1.f && v <= 1.f) ? db : 0.f;
    dX[i * 3] = du / z;
    dX[i * 3 + 1] = dv / z;
    dX[i * 3 + 2] = -(du * u + dv * v) / z;
  }
}

inline int stream_grid(int64_t work_items, int threads = 256) {
  int64_t blocks = (work_items + threads - 1) / threads;
  const int64_t cap = 256 * 8;   // 256 CUs x 8 blocks, grid-stride beyond
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace
}  // namespace vp3d

using namespace vp3d;

extern "C" {

// scalar or 16-byte-load flavour of the statistics finalize
static void launch_bn_finalize(hipStream_t st, int C, int64_t M, int nslab, int slab_rows, const float* stat_sum, const float* stat_m2,
                               const float* gamma, const float* beta, float eps, float momentum, const float* momentum_dev,
                               float* running_mean, float* running_var, int64_t* nbt, float* scale, float* shift, float* save_mean,
                               float* save_invstd) {
  if (C % 16 == 0 && aligned16(stat_sum) && aligned16(stat_m2) && nslab >= 64)
    VP3D_LAUNCH(k_bn_finalize_v4, dim3(C / 16), dim3(4 * FIN4_GROUPS), 0, st, C, M, nslab, slab_rows, stat_sum, stat_m2, gamma,
                       beta, eps, momentum, momentum_dev, running_mean, running_var, nbt, scale, shift, save_mean, save_invstd);
  else
    VP3D_LAUNCH(k_bn_finalize, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_GROUPS), 0, st, C, M, nslab, slab_rows,
                       stat_sum, stat_m2, gamma, beta, eps, momentum, momentum_dev, running_mean, running_var, nbt, scale, shift,
                       save_mean, save_invstd);
}

int vp3d_bn_finalize(vp3d_stream_t stream, int32_t C, int64_t M, const float* stat_sum, const float* stat_m2,
                     const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                     float* save_mean, float* save_invstd) {
  VP3D_REQUIRE(C > 0 && M > 0, "bn_finalize: C=%d M=%lld", C, (long long)M);
  VP3D_REQUIRE(stat_sum && stat_m2 && gamma && beta && scale && shift && save_mean && save_invstd,
               "bn_finalize: null pointer");
  const int nslab = (int)vp3d_stat_slabs(M);
  launch_bn_finalize((hipStream_t)stream, C, M, nslab, 64, stat_sum, stat_m2, gamma, beta, eps, momentum, nullptr, running_mean,
                     running_var, num_batches_tracked, scale, shift, save_mean, save_invstd);
  return check_launch("bn_finalize");
}

int vp3d_bn_finalize_dm(vp3d_stream_t stream, int32_t C, int64_t M, const float* stat_sum, const float* stat_m2,
                        const float* gamma, const float* beta, float eps, const float* momentum_dev, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                        float* save_mean, float* save_invstd) {
  VP3D_REQUIRE(C > 0 && M > 0, "bn_finalize_dm: C=%d M=%lld", C, (long long)M);
  VP3D_REQUIRE(stat_sum && stat_m2 && gamma && beta && scale && shift && save_mean && save_invstd && momentum_dev,
               "bn_finalize_dm: null pointer");
  const int nslab = (int)vp3d_stat_slabs(M);
  launch_bn_finalize((hipStream_t)stream, C, M, nslab, 64, stat_sum, stat_m2, gamma, beta, eps, 0.f, momentum_dev, running_mean,
                     running_var, num_batches_tracked, scale, shift, save_mean, save_invstd);
  return check_launch("bn_finalize_dm");
}

int vp3d_bn_finalize_slab(vp3d_stream_t stream, int32_t C, int64_t M, int32_t slab_rows, const float* stat_sum, const float* stat_m2,
                          const float* gamma, const float* beta, float eps, float momentum, const float* momentum_dev,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                          float* save_mean, float* save_invstd) {
  VP3D_REQUIRE(C > 0 && M > 0 && (slab_rows == 32 || slab_rows == 64), "bn_finalize_slab: C=%d M=%lld slab_rows=%d", C, (long long)M,
               slab_rows);
  VP3D_REQUIRE(stat_sum && stat_m2 && gamma && beta && scale && shift && save_mean && save_invstd, "bn_finalize_slab: null pointer");
  const int nslab = (int)((M + slab_rows - 1) / slab_rows);
  launch_bn_finalize((hipStream_t)stream, C, M, nslab, slab_rows, stat_sum, stat_m2, gamma, beta, eps, momentum, momentum_dev,
                     running_mean, running_var, num_batches_tracked, scale, shift, save_mean, save_invstd);
  return check_launch("bn_finalize_slab");
}

int vp3d_bn_fold(vp3d_stream_t stream, int32_t C, const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float eps, float* scale, float* shift) {
  VP3D_REQUIRE(C > 0 && gamma && beta && running_mean && running_var && scale && shift, "bn_fold: bad argument");
  VP3D_LAUNCH(k_bn_fold, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  return check_launch("bn_fold");
}

// Launch geometry of the column-owned streaming kernels: lanes_per_row threads cover a row strip of
// lanes_per_row*VEC channels, rows_per_block rows per block pass; grid.y row groups up to ~max_blocks blocks.
static void col_geometry(int64_t M, int C, bool vec, int64_t max_blocks, int min_rows_per_thread, int* lanes_per_row,
                         int* rows_per_block, int* gx, int* gy) {
  const int v = vec ? 4 : 1;
  const int cq = (C + v - 1) / v;
  *lanes_per_row = cq < 256 ? cq : 256;
  *rows_per_block = 256 / *lanes_per_row;
  *gx = (cq + *lanes_per_row - 1) / *lanes_per_row;
  const int64_t want = (M + *rows_per_block - 1) / *rows_per_block;
  int64_t cap = max_blocks / (*gx);
  if (cap < 1) cap = 1;
  int64_t gy64 = (want + min_rows_per_thread - 1) / min_rows_per_thread;
  if (gy64 > cap) gy64 = cap;
  if (gy64 < 1) gy64 = 1;
  *gy = (int)gy64;
}

int vp3d_bn_act_fwd(vp3d_stream_t stream, int64_t M, int32_t C, const float* y, const float* scale,
                    const float* shift, const vp3d_dropout* drop, const float* res, int32_t t_dst, int32_t r_t,
                    int32_t r_stride, int32_t r_off, int32_t r_ld, float* out) {
  VP3D_REQUIRE(M > 0 && C > 0 && y && scale && shift && out, "bn_act_fwd: bad argument");
  if (drop) VP3D_REQUIRE(drop->p >= 0.f && drop->p < 1.f, "bn_act_fwd: dropout p=%f", drop->p);
  ResMap rm{res, t_dst, r_t, r_stride, r_off, r_ld};
  if (res) VP3D_REQUIRE(t_dst > 0 && r_t > 0 && r_ld >= C, "bn_act_fwd: bad residual map");
  const DropP d = make_drop(drop);
  VP3D_REQUIRE(M < ((int64_t)1 << 31), "bn_act_fwd: more than 2^31 rows");
  const bool vec = (C % 4 == 0) && aligned16(y) && aligned16(out) && (!res || (aligned16(res) && r_ld % 4 == 0));
  int lpr, rpb, gx, gy;
  col_geometry(M, C, vec, 2048, 1, &lpr, &rpb, &gx, &gy);
  if (vec)
    VP3D_LAUNCH((k_bn_act_fwd<4>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (int)M, C, y, scale, shift, d,
                       rm, out, lpr, rpb);
  else
    VP3D_LAUNCH((k_bn_act_fwd<1>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (int)M, C, y, scale, shift, d,
                       rm, out, lpr, rpb);
  return check_launch("bn_act_fwd");
}

int vp3d_bn_bwd_reduce(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                       const float* scale, const float* shift, const float* mean, const float* invstd,
                       const vp3d_dropout* drop, float* partials, int32_t* nparts) {
  VP3D_REQUIRE(M > 0 && C > 0 && nparts, "bn_bwd_reduce: bad argument");
  VP3D_REQUIRE(M < ((int64_t)1 << 31), "bn_bwd_reduce: more than 2^31 rows");
  const bool vec = (C % 4 == 0);
  int lpr, rpb, gx, gy;
  col_geometry(M, C, vec, 1024, 8, &lpr, &rpb, &gx, &gy);
  *nparts = gy * rpb;
  if (partials == nullptr) return VP3D_OK;   // size query
  VP3D_REQUIRE(go && y && scale && shift && mean && invstd, "bn_bwd_reduce: null pointer");
  VP3D_REQUIRE(!vec || (aligned16(go) && aligned16(y)), "bn_bwd_reduce: go / y must be 16-byte aligned");
  const DropP d = make_drop(drop);
  if (vec)
    VP3D_LAUNCH((k_bn_bwd_reduce<4>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (int)M, C, go, y, scale,
                       shift, mean, invstd, d, partials, lpr, rpb);
  else
    VP3D_LAUNCH((k_bn_bwd_reduce<1>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (int)M, C, go, y, scale,
                       shift, mean, invstd, d, partials, lpr, rpb);
  return check_launch("bn_bwd_reduce");
}

int vp3d_bn_bwd_finalize(vp3d_stream_t stream, int32_t C, const float* partials, int32_t nparts, float* dgamma,
                         float* dbeta) {
  VP3D_REQUIRE(C > 0 && nparts > 0 && partials && dgamma && dbeta, "bn_bwd_finalize: bad argument");
  VP3D_LAUNCH(k_bn_bwd_finalize, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_GROUPS), 0, (hipStream_t)stream, C, partials, nparts,
                     dgamma, dbeta);
  return check_launch("bn_bwd_finalize");
}

int vp3d_bn_bwd_apply(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                      const float* scale, const float* shift, const float* mean, const float* invstd,
                      const vp3d_dropout* drop, const float* dgamma, const float* dbeta, float* dy) {
  VP3D_REQUIRE(M > 0 && C > 0 && go && y && scale && shift && mean && invstd && dgamma && dbeta && dy,
               "bn_bwd_apply: bad argument");
  const DropP d = make_drop(drop);
  VP3D_REQUIRE(M < ((int64_t)1 << 31), "bn_bwd_apply: more than 2^31 rows");
  const bool vec = (C % 4 == 0) && aligned16(go) && aligned16(y) && aligned16(dy);
  int lpr, rpb, gx, gy;
  col_geometry(M, C, vec, 2048, 1, &lpr, &rpb, &gx, &gy);
  if (vec)
    VP3D_LAUNCH((k_bn_bwd_apply<4>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (int)M, C, go, y, scale,
                       shift, mean, invstd, d, dgamma, dbeta, dy, lpr, rpb);
  else
    VP3D_LAUNCH((k_bn_bwd_apply<1>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (int)M, C, go, y, scale,
                       shift, mean, invstd, d, dgamma, dbeta, dy, lpr, rpb);
  return check_launch("bn_bwd_apply");
}

int vp3d_bn_bwd_apply_g(vp3d_stream_t stream, int64_t M, int32_t C, const float* g, const float* y,
                        const float* scale, const float* mean, const float* invstd, const float* dgamma,
                        const float* dbeta, float* dy) {
  VP3D_REQUIRE(M > 0 && C > 0 && g && y && scale && mean && invstd && dgamma && dbeta && dy, "bn_bwd_apply_g: bad argument");
  VP3D_REQUIRE(M < ((int64_t)1 << 31), "bn_bwd_apply_g: more than 2^31 rows");
  const bool vec = (C % 4 == 0) && aligned16(g) && aligned16(y) && aligned16(dy);
  int lpr, rpb, gx, gy;
  col_geometry(M, C, vec, 2048, 1, &lpr, &rpb, &gx, &gy);
  if (vec)
    VP3D_LAUNCH((k_bn_bwd_apply_g<4>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (int)M, C, g, y, scale, mean,
                       invstd, dgamma, dbeta, dy, lpr, rpb);
  else
    VP3D_LAUNCH((k_bn_bwd_apply_g<1>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (int)M, C, g, y, scale, mean,
                       invstd, dgamma, dbeta, dy, lpr, rpb);
  return check_launch("bn_bwd_apply_g");
}

int vp3d_pack_weight(vp3d_stream_t stream, const float* w, int32_t c_out, int32_t c_in, int32_t taps,
                     const float* scale, float* out, int32_t ld_out) {
  VP3D_REQUIRE(w && out && c_out > 0 && c_in > 0 && taps > 0 && ld_out >= taps * c_in, "pack_weight: bad argument");
  VP3D_LAUNCH(k_pack_weight, dim3(stream_grid((int64_t)c_out * c_in)), dim3(256), 0, (hipStream_t)stream, c_out,
                     c_in, taps, w, scale, out, ld_out);
  return check_launch("pack_weight");
}

int vp3d_wgrad_reduce(vp3d_stream_t stream, const float* partials, int32_t ld_part, int32_t splits, int32_t c_out,
                      int32_t c_in, int32_t taps, float* dw) {
  VP3D_REQUIRE(partials && dw && splits > 0 && c_out > 0 && c_in > 0 && taps > 0 && ld_part >= taps * c_in,
               "wgrad_reduce: bad argument");
  const bool vec = c_in % 4 == 0 && ld_part % 4 == 0 && aligned16(partials) && aligned16(dw) && (taps == 1 || taps == 3);
  if (vec && taps == 3)
    VP3D_LAUNCH((k_wgrad_reduce_v4<3>), dim3(stream_grid((int64_t)c_out * c_in / 4)), dim3(256), 0, (hipStream_t)stream,
                       splits, c_out, c_in, partials, ld_part, dw);
  else if (vec)
    VP3D_LAUNCH((k_wgrad_reduce_v4<1>), dim3(stream_grid((int64_t)c_out * c_in / 4)), dim3(256), 0, (hipStream_t)stream,
                       splits, c_out, c_in, partials, ld_part, dw);
  else
    VP3D_LAUNCH(k_wgrad_reduce, dim3(stream_grid((int64_t)c_out * c_in)), dim3(256), 0, (hipStream_t)stream,
                       splits, c_out, c_in, taps, partials, ld_part, dw);
  return check_launch("wgrad_reduce");
}

int vp3d_colsum(vp3d_stream_t stream, int64_t M, int32_t N, const float* g, int32_t ld, float* out) {
  VP3D_REQUIRE(M > 0 && N > 0 && g && out && ld >= N, "colsum: bad argument");
  VP3D_LAUNCH(k_colsum, dim3(N), dim3(256), 0, (hipStream_t)stream, M, N, g, ld, out);
  return check_launch("colsum");
}

int vp3d_im2row(vp3d_stream_t stream, const vp3d_rowmap* map, const float* x, int32_t ldx, int32_t k_valid,
                int32_t kpad, int32_t one_col, float* out) {
  VP3D_REQUIRE(map && x && out, "im2row: null pointer");
  VP3D_REQUIRE(one_col < 0 || (one_col >= k_valid && one_col < kpad), "im2row: the bias column must be a padding column");
  VP3D_REQUIRE(map->batch > 0 && map->t_dst > 0 && map->t_src > 0 && k_valid > 0 && kpad >= k_valid && kpad % 4 == 0 &&
                   aligned16(out), "im2row: bad sizes (k_valid=%d kpad=%d)", k_valid, kpad);
  VP3D_REQUIRE((int64_t)(map->t_dst - 1) * map->t_stride * ldx + k_valid <= (int64_t)map->t_src * ldx,
               "im2row: rows run past the end of a sample");
  const int64_t M = (int64_t)map->batch * map->t_dst;
  VP3D_REQUIRE(M < ((int64_t)1 << 31), "im2row: more than 2^31 rows");
  VP3D_LAUNCH(k_im2row, dim3(stream_grid(M * (kpad / 4))), dim3(256), 0, (hipStream_t)stream, (int)M, map->t_dst,
                     map->t_src, map->t_stride, ldx, k_valid, kpad, one_col < 0 ? -1 : one_col, x, out);
  return check_launch("im2row");
}

int vp3d_dropout_mask(vp3d_stream_t stream, int64_t n, const vp3d_dropout* drop, float* out) {
  VP3D_REQUIRE(n > 0 && out, "dropout_mask: bad argument");
  VP3D_LAUNCH(k_dropout_mask, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n, make_drop(drop), out);
  return check_launch("dropout_mask");
}

int vp3d_project_to_2d_fwd(vp3d_stream_t stream, int64_t n_cam, int64_t pts_per_cam, const float* X,
                           const float* cam, int32_t linear, float* out) {
  VP3D_REQUIRE(n_cam > 0 && pts_per_cam > 0 && X && cam && out, "project_to_2d_fwd: bad argument");
  VP3D_LAUNCH(k_project_fwd, dim3(stream_grid(n_cam * pts_per_cam)), dim3(256), 0, (hipStream_t)stream, n_cam,
                     pts_per_cam, X, cam, linear, out);
  return check_launch("project_to_2d_fwd");
}

int vp3d_project_to_2d_bwd(vp3d_stream_t stream, int64_t n_cam, int64_t pts_per_cam, const float* X,
                           const float* cam, const float* gout, int32_t linear, float* dX) {
  VP3D_REQUIRE(n_cam > 0 && pts_per_cam > 0 && X && cam && gout && dX, "project_to_2d_bwd: bad argument");
  VP3D_LAUNCH(k_project_bwd, dim3(stream_grid(n_cam * pts_per_cam)), dim3(256), 0, (hipStream_t)stream, n_cam,
                     pts_per_cam, X, cam, gout, linear, dX);
  return check_launch("project_to_2d_bwd");
}

}  // extern "C"
