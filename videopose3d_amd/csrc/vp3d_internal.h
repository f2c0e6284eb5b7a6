// Internal declarations shared by the libvp3d translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vp3d.h"
#include "vp3d_dropout.h"

namespace vp3d {

// thread-local last-error string
void set_error(const char* fmt, ...);
int check_launch(const char* what);
void count_launch();      // vp3d_launch_count(): one per kernel enqueued
// every kernel launch of the library goes through this (bench.py reports launches per step: the step is a dependent chain of
// ~200 launches and their count is a lever of its own)
#define VP3D_LAUNCH(...)               \
  do {                                 \
    ::vp3d::count_launch();            \
    hipLaunchKernelGGL(__VA_ARGS__);   \
  } while (0)

#define VP3D_REQUIRE(cond, ...)              \
  do {                                       \
    if (!(cond)) {                           \
      ::vp3d::set_error(__VA_ARGS__);        \
      return VP3D_E_INVALID;                 \
    }                                        \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Epilogue as the kernels see it (vp3d_epilogue + output addressing).
struct Epi {
  float* C;
  int64_t c_bpitch;  // floats per sample in C
  int32_t ldc;       // floats per row in C
  const float* bias;
  int32_t relu;
  const float* R;
  int64_t r_bpitch;
  int32_t r_ld, r_t, r_stride, r_off, r_col0, r_cols;
  float* stat_sum;
  float* stat_m2;
  int32_t vec;       // set by the launcher: float4 epilogue legal (sizes / pitches % 4, 16-B aligned bases)
  // fused backward of the upstream activation (vp3d_act_bwd; ab_y == nullptr: off).  Addressing of ab_y / ab_g is
  // that of C.  Partials per 64-row slab: ab_part[((slab * (N / ab_c) + n / ab_c) * 2 + which) * ab_c + n % ab_c].
  const float* ab_y;
  const float* ab_scale;
  const float* ab_shift;
  const float* ab_mean;
  const float* ab_invstd;
  float* ab_g;
  float* ab_part;
  int32_t ab_c, ab_store_v;
  DropP ab_drop;
  // split-fp16 GEMMs (vp3d_gemm_s16.hip): per-tensor magnitude bounds of the two S16 operands (device floats, NULL =
  // exponent 0); the accumulator is scaled by 2^(exp(*bound_a) + exp(*bound_b)) before anything else.
  // amax_out: atomicMax of |stored value| over the launch (the bound of the result; zeroed by the caller), or NULL
  const float* bound_a;
  const float* bound_b;
  float* amax_out;
  // S16 residual: R holds S16 rows with the exponent of *r_bound (r_s16 != 0)
  int32_t r_s16;
  const float* r_bound;
  // S16 output (eval forward, c_s16 != 0): C receives S16 rows whose exponent comes from the bound
  //   l1[0] * amax(in_amax) + l1[1] + (res_amax ? amax(res_amax) : 0)      (l1 = {max_n sum_k |W[n][k]|, max_n |bias[n]|})
  // which every workgroup evaluates identically and workgroup 0 publishes in out_wbound[0] for the consumers
  int32_t c_s16;
  const float* in_amax;
  const float* l1;
  const float* res_amax;
  float* out_wbound;
  // statistics-only launch (no_out != 0): nothing but the BatchNorm slab statistics is written (C may be NULL)
  int32_t no_out;
  // fused activation (act_scale != nullptr): C receives the S16 rows of dropout(relu(acc * act_scale[n] + act_shift[n]))
  // (mask: ab_drop, element index m * N + n) under the exponent of *act_bound, act_bits the [z > 0 and kept] bits --
  // exactly what vp3d_bn_act_fwd_s16 writes from a stored conv output
  const float* act_scale;
  const float* act_shift;
  const float* act_bound;
  uint8_t* act_bits;
  // fused BatchNorm-backward column sums of the upstream activation (vp3d_s16_red; red != 0): uses ab_y / ab_mean / ab_invstd /
  // ab_scale / ab_part / ab_c from above; upstream row of output element (b, t, n) = b * red_row_b + t * red_row_t + n / ab_c
  int32_t red;
  const uint8_t* red_bits;
  int32_t red_m, red_row_b, red_row_t;
  float red_inv_keep, red_inv_m, red_sqrt_m1;
  int32_t* red_cnt;
  float* red_dgamma;
  float* red_dbeta;
  float* red_dy_bound;
  // BatchNorm finalize inside the split-K finishing pass (vp3d_s16_fin; fin_tickets == nullptr: off)
  int32_t* fin_tickets;
  const float* fin_gamma;
  const float* fin_beta;
  const float* fin_momentum_dev;
  float fin_eps, fin_momentum;
  float* fin_running_mean;
  float* fin_running_var;
  int64_t* fin_nbt;
  float* fin_scale;
  float* fin_shift;
  float* fin_save_mean;
  float* fin_save_invstd;
};

// GEMM over gathered rows: C[m][n] = sum_k A[row(m,k)][.] * B   (forward conv: B k-contiguous "NT";
// dgrad: B n-contiguous "NN").
struct RowsGemmArgs {
  const float* A;
  const float* B;
  const float* zeros;
  int32_t M, N, K;
  int32_t lda, c_src;          // row pitch of A, channels per tap (K = taps*c_src)
  int32_t ldb, b_tap_stride;   // NT: B[n*ldb + k];  NN: B[(k % c_src)*ldb + (k / c_src)*b_tap_stride + n]
  int32_t t_dst, t_src, t_stride, tap_step, t_off, taps;
  int32_t m_tiles, n_tiles;
  // K-sliced tail (set by launch_rows_gemm from plan_rows_gemm): tile positions >= pos_full are cut into `splits`
  // slices whose raw partial tiles go to `part` ([split][tail position][128][128]); k_splitk_finish sums them
  int32_t pos_full, tail_pos, splits, kt_per_split;
  int32_t stat_slab_rows = 64;   // rows per statistics slab the caller sized epi.stat_sum / stat_m2 for (S16 GEMM: vp3d_s16.stat_slab_rows)
  float* part;
  int64_t part_floats;
  uint32_t a_bytes, b_bytes;   // S16 kernel, buffer-descriptor DMA: byte extents of the two operands (set by the launcher)
  int32_t m_begin, m_end;      // S16 kernel: the rows this launch covers (set by the launcher; [0, M) normally)
  // S16 kernel, stream-K instances (set by the launcher): blocks [0, sk_dp_blocks) run the first tiles of the linear order
  // whole, the sk_blocks behind share the K-tiles of the last sk_tiles tiles; workspace sk_ws: sk_max_seg slots of one tile
  // of floats per shared tile, sk_cnt: one ticket per shared tile (zero on entry, zero again on exit)
  int32_t sk_dp_blocks, sk_tiles, sk_blocks, sk_max_seg;
  float* sk_ws;
  int32_t* sk_cnt;
  Epi epi;
};

// wgrad: C[i][n] = sum_m G[m][i] * X[row(m, tap(n))][ci(n)],   n = tap*c_x + ci.
struct RedGemmArgs {
  const float* G;
  const float* X;
  const float* zeros;
  float* C;                 // partials [splits][Mo][N]
  int32_t Mred;             // reduction length (rows)
  int32_t Mo, N;            // output rows (c_out) and columns (taps*c_x)
  int32_t ldg, ldx, c_x;
  int32_t t_dst, t_src, t_stride, tap_step, t_off;
  int32_t m_tiles, n_tiles, splits, kt_per_split;
};

struct RowsPlan {
  int positions;   // tile positions of the launch order (valid tiles + the padding of ragged m-/n-groups)
  int pos_full;    // positions [0, pos_full) run as whole tiles
  int splits;      // K-slices of the positions behind (1 = none)
};
RowsPlan plan_rows_gemm(int M, int N, int K);
int64_t rows_gemm_ws_floats(int M, int N, int K);
int rows_gemm_splits(int M, int N, int K);
int red_gemm_splits(int Mred, int Mo, int N);
int launch_rows_gemm(hipStream_t s, const RowsGemmArgs& a, bool b_kcontig);
int launch_red_gemm(hipStream_t s, const RedGemmArgs& a);
// split-fp16 NT GEMM (vp3d_gemm_s16.hip); cfg selects the tile configuration
void plan_nt_s16(int M, int N, int K, int allow_split, int raw, int* cfg_out, int* splits_out, int allow_mix = 0);
int nt_s16_stat_slab_rows(int cfg);
int launch_nt_s16(hipStream_t s, const RowsGemmArgs& a, int cfg, int splits, float* ws, int64_t ws_floats,
                  bool raw_partials, int32_t* tickets = nullptr);
void nt_s16_workspace(int M, int N, int K, int cfg, int splits, int raw, int64_t* ws_floats, int32_t* tickets);
int nt_s16_has_experiments();
// expand layer, forward (vp3d_expand_s16.hip): statistics pass (stat_sum != nullptr) or activation pass (out != nullptr)
int launch_expand_fwd_s16(hipStream_t s, int64_t M, int32_t N, int32_t kpad, const float* x, const float* x_bound,
                          const float* w, const float* w_bound, float* stat_sum, float* stat_m2, const float* scale,
                          const float* shift, const DropP& drop, const float* out_bound, float* out, uint8_t* bits);
// expand layer, backward: partial P = G^T X per row group from go + activation bits (vp3d_expand_s16.hip)
int expand_bwd_groups(int64_t M, int32_t C);
int expand_gram_groups(int64_t M);
int launch_expand_gram_s16(hipStream_t s, int64_t M, int32_t kpad, const float* xt, int64_t ld_t, const float* x_bound,
                           int32_t one_col, int32_t groups, float* part);
int launch_expand_bwd_p_s16(hipStream_t s, int64_t M, int32_t C, int32_t kpad, const float* go, const float* go_bound,
                            const uint8_t* bits, float p, const float* xt, int64_t ld_t, const float* x_bound, int32_t groups,
                            float* part, float* gram_part);
int launch_split_rows(hipStream_t s, int64_t M, int32_t C, const float* src, int64_t ld_src, float* dst, int64_t ld_dst,
                      const float* bound);
int launch_amax(hipStream_t s, int64_t n, const float* src, float* bound, float floor_ = 0.f);
int launch_wgrad_rows_s16(hipStream_t s, int64_t Mk, const float* dy, int64_t ld_dy, int32_t c_out, const float* dy_bound,
                          const float* x, int64_t ld_x, int32_t taps, int32_t c_in, const float* x_bound, int32_t splits,
                          float* part);

}  // namespace vp3d
