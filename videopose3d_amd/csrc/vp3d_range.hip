// Intra-tensor dynamic-range statistics for the guard of the split-fp16 arithmetic (videopose3d_amd/range_guard.py).
//
// The S16 format (vp3d_s16.h) keeps ONE exponent per tensor: elements below 2^-17 of the tensor's bound lose bits.  The
// reference's BatchNorm affine is unconstrained (common/model.py:32,117-119: nn.BatchNorm1d(channels)), so a single hot
// gamma_c / beta_c -- or one hot weight row -- can push every other channel of a tensor down the format's range.  These
// kernels measure how far the parameters are from that regime, on the device, without any host synchronisation:
//   "spread" of a set of per-group magnitudes g_i > 0  :=  E(max_i g_i) - E(median_i g_i),  E = binary exponent (frexp)
// i.e. log2 of how far the hottest group sits above the typical one (zero groups -- dead channels -- are exact in any
// exponent and are left out).  The median comes from a 256-bin histogram of exponents: O(n), one block.
//   out[0]: max over BatchNorm layers of the spread of the per-channel activation bound  |gamma_c| * k_l + |beta_c|
//           (k_l = sqrt(M_l - 1) in training -- the Samuelson factor of vp3d_act_bounds_multi --, a small constant in eval)
//   out[1]: max over weight tensors of the spread of the per-output-row maxima max_k |W[n][k]|
// Both are atomicMax-ed into `out` (int32, zeroed by the caller when a fresh measurement is wanted).
#include "vp3d_internal.h"
#include "vp3d_s16.h"

namespace vp3d {
namespace {

constexpr int kMaxRangeTensors = 24;
constexpr int kNoExp = -1000;          // marker of an all-zero group

__device__ __forceinline__ int exp_of(float v) {
  if (!(v > 0.f)) return kNoExp;
  if (!(v < 3.0e38f)) return 128;      // inf / nan: as hot as it gets
  int e;
  frexpf(v, &e);
  return e;                            // -148 .. 128
}

// histogram of exponents (block of 256 threads; hist[0..511] in LDS, bin = e + 200) -> spread of the groups seen so far
__device__ __forceinline__ int spread_from_hist(int* hist, int* sh) {
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0, emax = kNoExp;
    for (int b = 0; b < 512; ++b)
      if (hist[b] > 0) {
        total += hist[b];
        emax = b - 200;
      }
    int spread = 0;
    if (total > 0) {
      const int want = (total + 1) / 2;            // lower median
      int seen = 0, emed = emax;
      for (int b = 0; b < 512; ++b) {
        seen += hist[b];
        if (seen >= want) {
          emed = b - 200;
          break;
        }
      }
      spread = emax - emed;
    }
    *sh = spread;
  }
  __syncthreads();
  return *sh;
}

struct AffineArgs {
  const float* gamma[kMaxRangeTensors];
  const float* beta[kMaxRangeTensors];
  float kfac[kMaxRangeTensors];
  int C;
  int32_t* out;
};

__global__ void __launch_bounds__(256) k_range_affine(AffineArgs a) {
  __shared__ int hist[512];
  __shared__ int sh;
  const int li = blockIdx.x;
  for (int b = threadIdx.x; b < 512; b += 256) hist[b] = 0;
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += 256) {
    const float g = a.gamma[li] != nullptr ? fabsf(a.gamma[li][c]) : 1.f;
    const float bt = a.beta[li] != nullptr ? fabsf(a.beta[li][c]) : 0.f;
    const int e = exp_of(g * a.kfac[li] + bt);
    if (e != kNoExp) atomicAdd(&hist[e + 200], 1);
  }
  const int s = spread_from_hist(hist, &sh);
  if (threadIdx.x == 0) atomicMax(a.out, s);
}

struct RowsArgs {
  const float* t[kMaxRangeTensors];
  int64_t rows[kMaxRangeTensors], row_len[kMaxRangeTensors], ws_off[kMaxRangeTensors];
  int32_t* ws;              // per-row exponents
  int32_t* out;
};

// one wave per row: max |.| over the row's contiguous row_len floats -> its exponent
__global__ void __launch_bounds__(256) k_range_rows(RowsArgs a) {
  const int ti = blockIdx.y;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows[ti]) return;
  const int lane = threadIdx.x & 63;
  const float* src = a.t[ti] + row * a.row_len[ti];
  const int64_t n = a.row_len[ti];
  float m = 0.f;
  if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0 && (n & 3) == 0) {
    const f32x4_t* s4 = reinterpret_cast<const f32x4_t*>(src);
    for (int64_t i = lane; i < (n >> 2); i += 64) {
      const f32x4_t v = s4[i];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
  } else {
    for (int64_t i = lane; i < n; i += 64) m = fmaxf(m, fabsf(src[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) a.ws[a.ws_off[ti] + row] = exp_of(m);
}

__global__ void __launch_bounds__(256) k_range_rows_finish(RowsArgs a) {
  __shared__ int hist[512];
  __shared__ int sh;
  const int ti = blockIdx.x;
  for (int b = threadIdx.x; b < 512; b += 256) hist[b] = 0;
  __syncthreads();
  for (int64_t r = threadIdx.x; r < a.rows[ti]; r += 256) {
    const int e = a.ws[a.ws_off[ti] + r];
    if (e != kNoExp) atomicAdd(&hist[e + 200], 1);
  }
  const int s = spread_from_hist(hist, &sh);
  if (threadIdx.x == 0) atomicMax(a.out + 1, s);
}

// ---- per-COLUMN maxima of a row-major [M][C] tensor (the input batch: C = joints x 2 keypoint columns; the loss gradient at the
// head: C = joints x 3) -> spread of the columns.  One hot column raises the tensor's single S16 exponent exactly as one hot
// BatchNorm channel does.  Non-negative floats order as their bit patterns: LDS atomicMax per element, one global atomicMax
// per (block, column); ws [C] ints are zero on entry and zeroed again by the finishing block.
constexpr int kMaxRangeCols = 1024;

__global__ void __launch_bounds__(256) k_range_cols(const float* __restrict__ x, int64_t M, int C, int64_t ld, int rows_per_block,
                                                    int32_t* ws) {
  __shared__ int cmax[kMaxRangeCols];
  for (int c = threadIdx.x; c < C; c += 256) cmax[c] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  const int64_t n = (r1 - r0) * C;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    float v = fabsf(x[(r0 + r) * ld + c]);
    if (v != v) v = __int_as_float(0x7f800000);          // nan counts as inf
    atomicMax(&cmax[c], __float_as_int(v));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256)
    if (cmax[c] > 0) atomicMax(ws + c, cmax[c]);
}

__global__ void __launch_bounds__(256) k_range_cols_finish(int C, int32_t* ws, int32_t* out) {
  __shared__ int hist[512];
  __shared__ int sh;
  for (int b = threadIdx.x; b < 512; b += 256) hist[b] = 0;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int e = exp_of(__int_as_float(ws[c]));
    ws[c] = 0;
    if (e != kNoExp) atomicAdd(&hist[e + 200], 1);
  }
  const int s = spread_from_hist(hist, &sh);
  if (threadIdx.x == 0) atomicMax(out, s);
}

}  // namespace
}  // namespace vp3d

using namespace vp3d;

extern "C" {

int vp3d_range_max_tensors(void) { return kMaxRangeTensors; }

int vp3d_range_cols(vp3d_stream_t stream, int64_t M, int32_t C, const float* x, int64_t ld, int32_t* ws, int32_t* out) {
  VP3D_REQUIRE(M > 0 && C > 0 && C <= kMaxRangeCols && x && ld >= C && ws && out, "range_cols: bad argument (at most %d columns)",
               kMaxRangeCols);
  int64_t blocks = (M + 63) / 64;
  if (blocks > 2048) blocks = 2048;
  const int rows_per_block = (int)((M + blocks - 1) / blocks);
  blocks = (M + rows_per_block - 1) / rows_per_block;
  VP3D_LAUNCH(k_range_cols, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, M, (int)C, ld, rows_per_block, ws);
  if (int rc = check_launch("range_cols")) return rc;
  VP3D_LAUNCH(k_range_cols_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, (int)C, ws, out);
  return check_launch("range_cols_finish");
}

int vp3d_range_stats(vp3d_stream_t stream, int32_t n_layers, int32_t C, const float* const* gamma, const float* const* beta,
                     const float* kfac, int32_t n_tensors, const float* const* w, const int64_t* rows, const int64_t* row_len,
                     int32_t* ws, int64_t ws_ints, int32_t* out) {
  VP3D_REQUIRE(n_layers >= 0 && n_layers <= kMaxRangeTensors && n_tensors >= 0 && n_tensors <= kMaxRangeTensors && out,
               "range_stats: at most %d layers / tensors", kMaxRangeTensors);
  VP3D_REQUIRE(n_layers == 0 || (C > 0 && gamma && beta && kfac), "range_stats: bad BatchNorm arguments");
  VP3D_REQUIRE(n_tensors == 0 || (w && rows && row_len && ws), "range_stats: bad tensor arguments");
  if (n_layers > 0) {
    AffineArgs a;
    for (int i = 0; i < n_layers; ++i) {
      a.gamma[i] = gamma[i];
      a.beta[i] = beta[i];
      a.kfac[i] = kfac[i];
    }
    a.C = C;
    a.out = out;
    VP3D_LAUNCH(k_range_affine, dim3(n_layers), dim3(256), 0, (hipStream_t)stream, a);
    if (int rc = check_launch("range_affine")) return rc;
  }
  if (n_tensors > 0) {
    RowsArgs r;
    int64_t off = 0, max_rows = 0;
    for (int i = 0; i < n_tensors; ++i) {
      VP3D_REQUIRE(w[i] && rows[i] > 0 && row_len[i] > 0, "range_stats: tensor %d", i);
      r.t[i] = w[i];
      r.rows[i] = rows[i];
      r.row_len[i] = row_len[i];
      r.ws_off[i] = off;
      off += rows[i];
      if (rows[i] > max_rows) max_rows = rows[i];
    }
    VP3D_REQUIRE(off <= ws_ints, "range_stats: workspace of %lld ints, need %lld", (long long)ws_ints, (long long)off);
    VP3D_REQUIRE((max_rows + 3) / 4 <= 0x7fffffff, "range_stats: too many rows");
    r.ws = ws;
    r.out = out;
    VP3D_LAUNCH(k_range_rows, dim3((unsigned)((max_rows + 3) / 4), n_tensors), dim3(256), 0, (hipStream_t)stream, r);
    if (int rc = check_launch("range_rows")) return rc;
    VP3D_LAUNCH(k_range_rows_finish, dim3(n_tensors), dim3(256), 0, (hipStream_t)stream, r);
    if (int rc = check_launch("range_rows_finish")) return rc;
  }
  return 0;
}

}  // extern "C"
