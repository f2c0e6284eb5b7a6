/*
 * vp3d.h -- C ABI of libvp3d.so: the MI355X (gfx950) implementation of the VideoPose3D
 * temporal-model hot path (reference common/model.py).  Plain C: raw device pointers, sizes and a
 * hipStream_t passed as void*.  No torch types, no C++ types.
 *
 * What each entry point replaces in the reference (all arithmetic there is delegated to torch.nn):
 *   vp3d_tconv_fwd        nn.Conv1d forward         model.py:102,113-118,167,178-180,33 (+ folded BN/ReLU/residual
 *                                                   of model.py:127,134-135 in eval mode)
 *   vp3d_tconv_dgrad      nn.Conv1d backward-data   autograd of model.py:134-135,193-194 (+ residual scatter of :132/:191)
 *   vp3d_tconv_wgrad      nn.Conv1d backward-weight autograd of the same calls
 *   vp3d_bn_finalize      nn.BatchNorm1d (train)    model.py:32,117,119,179,181: batch stats, running update
 *   vp3d_bn_act_fwd       bn -> ReLU -> Dropout (+ residual add)          model.py:127,134-135 / 188,193-194
 *   vp3d_bn_bwd_reduce / vp3d_bn_bwd_finalize / vp3d_bn_bwd_apply         autograd of the above
 *   vp3d_pack_weight / vp3d_bn_fold                                       eval-mode BN folding (model.eval(), run.py:427)
 *   vp3d_project_to_2d_fwd / _bwd                                         common/camera.py:37-67, 69-90
 *   vp3d_gather_chunks    ChunkedGenerator / UnchunkedGenerator batch assembly   common/generators.py:105-149, 216-239
 *   vp3d_mpjpe            mpjpe / weighted_mpjpe (+ gradient)                    common/loss.py:11-25
 *   vp3d_tta_fold         test-time-augmentation un-flip + average                run.py:677-680
 *   vp3d_adam_step        optim.Adam(amsgrad=True).step() on flat buffers         run.py:252,264,420
 *   vp3d_expand_stats_gram_s16   expand_bn's batch statistics (model.py:32,74,127,188) from the second-moment matrix of the
 *                                layer's <= 128-column input instead of a pass over expand_conv's output
 *   vp3d_range_stats      nothing in the reference: the dynamic-range statistic behind the split-fp16 arithmetic's guard (the
 *                         reference's BatchNorm affine and conv weights are unconstrained, model.py:32,102,113-119)
 *
 * Conventions
 *   - Layout: activations are channels-last rows, x[b][t][c] ("NLC"); this IS the reference's module boundary
 *     layout ([B,T,J*F] in, [B,T_out,J_out*3] out, model.py:68-75), so no transposes exist anywhere.
 *   - Ownership: the library never allocates, frees or retains user-visible memory.  Every buffer (inputs,
 *     outputs, workspaces) is a device pointer owned by the caller and must stay alive until the stream has
 *     passed the call.
 *   - Ordering: every call only enqueues kernels on `stream`; no implicit synchronisation, no allocation,
 *     hipGraph-capture safe.
 *   - Errors: every function returns 0 on success or a negative VP3D_E_* code; vp3d_last_error() returns a
 *     thread-local message.  Arguments are validated before anything is launched.  Nothing throws.
 *   - Threading: stateless; concurrent calls on distinct streams are safe.
 */
#ifndef VP3D_H_
#define VP3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VP3D_VERSION 110
#define VP3D_BOUND_SLOTS 32

#define VP3D_OK 0
#define VP3D_E_INVALID (-1)   /* bad argument (null pointer, size, alignment) */
#define VP3D_E_LAUNCH (-2)    /* HIP reported a launch error */
#define VP3D_E_UNSUPPORTED (-3)

typedef void* vp3d_stream_t; /* hipStream_t */

/* Row gather of a temporal convolution.  Output row m = b*t_dst + t reads, for tap k, the source row
 *   b*t_src + (t*t_stride + k*tap_step + t_off)
 * and contributes zero when that time index falls outside [0, t_src).
 *   forward conv (taps,dil,stride):    t_stride=stride, tap_step=dil,  t_off=0
 *   dgrad, gather form (dilated):      t_stride=1,      tap_step=-dil, t_off=0   (rows of dx gather rows of dy)
 *   dgrad, strided (stride==taps):     plain GEMM: taps=1, output written with ldc = taps*C_in
 */
typedef struct vp3d_rowmap {
  int32_t batch;     /* B */
  int32_t t_dst;     /* rows per sample on the output side; M = batch*t_dst */
  int32_t t_src;     /* rows per sample of the gathered tensor */
  int32_t t_stride;
  int32_t tap_step;  /* signed */
  int32_t t_off;
  int32_t taps;
} vp3d_rowmap;

/* Counter-based dropout (Philox4x32-10): element e of layer `layer` is kept iff
 * u16(e) >= round(p * 65536), where u16(e) is 16-bit half (e & 1) of word ((e & 7) >> 1) of
 * Philox(key=seed, counter=(e>>3, layer, offset)): one block serves 8 elements, the keep probability is exact for p = 0.25
 * and within 7.6e-6 of 1 - p otherwise; kept values are scaled by 1/(1-p).
 * The mask is never stored as a tensor: backward regenerates it from the same (seed, offset, layer), or reads the
 * forward's activation bits (split-fp16 engine). */
typedef struct vp3d_dropout {
  float p;
  uint64_t seed;
  uint64_t offset;
  uint32_t layer;
  const uint64_t* offset_ptr; /* optional device uint64 added to `offset` when the kernel runs (NULL: none): lets a
                                 captured hipGraph draw a fresh mask on every replay (bump the counter inside the graph) */
} vp3d_dropout;

/* Backward of the UPSTREAM layer's activation a = dropout(relu(bn(y_up))) fused into the dgrad epilogue (vp3d_tconv_dgrad
 * only; N % 128 == 0, c_stat % 4 == 0, 16-byte aligned float4 epilogue).  The value the epilogue would store,
 * v = acc (+ residual scatter), IS that activation's incoming gradient, so the kernel also produces what
 * vp3d_bn_bwd_reduce would compute in a separate pass over (v, y_up):
 *   g = (y_up*scale + shift > 0) ? v * keep_scale : 0           -> g_out   (same addressing as the dgrad output)
 *   per 64-row slab s of the output rows and column n:  sum g,  sum g*xhat,  xhat = (y_up - mean)*invstd
 *     -> partials[((s * (N / c_stat) + n / c_stat) * 2 + {0,1}) * c_stat + n % c_stat]
 *        = vp3d_bn_bwd_finalize's [nparts][2][C] layout with nparts = ceil(M/64) * (N / c_stat)
 * (c_stat = channels of the upstream BatchNorm; N = taps*c_stat for the strided dgrad whose rows are 3 frames wide).
 * store_v = 0: nobody else needs the raw gradient, it is not written (dx may then be NULL). */
typedef struct vp3d_act_bwd {
  const float* y_up;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  const vp3d_dropout* drop; /* NULL: no dropout */
  float* g_out;
  float* partials;
  int32_t c_stat;
  int32_t store_v;
} vp3d_act_bwd;

/* Fused epilogue of the GEMM kernels (all parts optional; NULL / 0 = off).
 *   v = acc (+ bias[n]) ; if relu: v = max(v,0) ; v += residual ; C[b*c_bpitch + t*ldc + n] = v
 * residual row for output row (b,t): tr = t*r_stride + r_off, used iff 0 <= tr < r_t and
 *   r_col0 <= n < r_col0 + r_cols, read from R[b*r_bpitch + tr*r_ld + (n - r_col0)].
 * stats (training-mode BatchNorm): per 64-row slab s and column n, over the valid rows of the slab of the RAW
 *   accumulator: stat_sum[s*N + n] = sum, stat_m2[s*N + n] = sum (v - slab_mean)^2.  Deterministic (no atomics). */
typedef struct vp3d_epilogue {
  const float* bias;
  int32_t relu;
  const float* residual;
  int64_t r_bpitch;
  int32_t r_ld;
  int32_t r_t;
  int32_t r_stride;
  int32_t r_off;
  int32_t r_col0;
  int32_t r_cols;
  float* stat_sum;
  float* stat_m2;
  const vp3d_act_bwd* act_bwd; /* dgrad only; NULL = off */
} vp3d_epilogue;

int vp3d_version(void);
/* kernel launches this library has issued in this process so far (every entry point counts the kernels it enqueues; launches
 * inside a hipGraph capture count once, at capture): what bench.py reports as launches per step */
int64_t vp3d_launch_count(void);
const char* vp3d_last_error(void);

/* number of 64-row statistic slabs a [M, *] output produces (size stat_sum / stat_m2 as slabs*N floats) */
int64_t vp3d_stat_slabs(int64_t M);
/* K-slicing vp3d_tconv_fwd / _dgrad apply to an [M,N,K] problem when given a workspace: the 128x128 tiles that do
 * not fill a whole round of the 256 CUs (for the small-M layers of the T_out = 1..9 tail: every tile) are cut into
 * K-slices that are dispatched last; vp3d_rows_gemm_splits returns the slice count (1 = none) and
 * vp3d_rows_gemm_ws_floats the workspace size in floats (0 = none needed) */
int vp3d_rows_gemm_splits(int64_t M, int32_t N, int32_t K);
int64_t vp3d_rows_gemm_ws_floats(int64_t M, int32_t N, int32_t K);
/* recommended `splits` for vp3d_tconv_wgrad (reduction over M rows into a [c_out, n_cols] matrix) */
int vp3d_wgrad_splits(int64_t M, int32_t c_out, int32_t n_cols);

/* y[b,t,:] = sum_k x[b, map(t,k), :] @ W_k  (+ epilogue).   M = B*t_dst, N = c_out, K = taps*c_in.
 *   x  : gathered activations, row pitch ldx floats, c_in channels used per tap
 *   wt : packed weights  wt[n*ldw + k*c_in + ci] == W[n][ci][k]   (vp3d_pack_weight, mode 0)
 *   y  : output rows at y[b*y_bpitch + t*ldy + n]
 *   zeros: >= 1024 B of device zeros (source for out-of-range taps / ragged tiles)
 *   splitk_ws: optional workspace of vp3d_rows_gemm_ws_floats(M, N, K) floats (NULL = never slice K) */
int vp3d_tconv_fwd(vp3d_stream_t stream, const vp3d_rowmap* map, const float* x, int32_t ldx, int32_t c_in,
                   const float* wt, int32_t ldw, int32_t c_out, float* y, int64_t y_bpitch, int32_t ldy,
                   const vp3d_epilogue* epi, const float* zeros, float* splitk_ws, int64_t splitk_ws_floats);

/* ------------------------------------------------------------------------------------------------------------
 * Split-fp16 ("S16") GEMM path: fp32-class results on the fp16 matrix cores.
 * An S16 tensor keeps an fp32 value v as two fp16 numbers hi = fp16(v*2^-e), lo = fp16(v*2^-e - hi) (22+ significant
 * bits; e = per-tensor exponent held in a device int32, NULL = 0).  8 consecutive elements of a row are stored as 16 B
 * of hi followed by 16 B of lo: an S16 row has the byte geometry of the fp32 row it replaces (ld* stay in 4-byte
 * units).  a*b is evaluated as ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_f16 with fp32 accumulation.
 * ------------------------------------------------------------------------------------------------------------ */
/* BatchNorm-backward column sums of the UPSTREAM activation inside the dgrad launch that produces its incoming gradient
 * (training backward; autograd of model.py:134 / :193 drop(relu(bn(conv(x)))) for the layer whose output this dgrad's
 * result is the gradient of).  The launch's result go (y, fp32, possibly with the residual gradient added) is still
 * stored; while a tile of it sits in registers the epilogue reads the same tile of y_up and its activation bits and forms
 *   g = bit ? go / (1 - p) : 0,   sum g,  sum g * (y_up - mean) * invstd      per column over the tile's rows,
 * writes them as partial rows [m_tile][2][c_out], and the LAST workgroup of every column strip (one ticket per tile-wide
 * strip of the c_up channels) folds the strip's rows (all taps of a strided dgrad, c_out = taps * c_up) in fp64 into
 * dbeta / dgamma and max-es the strip's share of the bound of dy,
 *   max_{c in strip} |scale[c]| * (max|go over the strip's columns| / (1 - p) + |dbeta[c]| / rows_up
 *                                  + sqrt(rows_up - 1) * |dgamma[c]| / rows_up),
 * into dy_bound (32 zeroed slots): dbeta / dgamma are those of vp3d_bn_bwd_reduce_fin_s16's pass over (go, y_up), the bound
 * is a (tighter or equal) guaranteed bound of the same dy -- g of a channel is bounded by max|go| over that channel's column.
 * Deterministic (fixed summation order).  Needs: amax_out (the bound of go), splits == 1, tile configuration 20 / 22 (operands
 * below 2 GiB), fp32 output with y_bpitch % c_up == 0 and ldy % c_up == 0 (go addresses the upstream activation
 * [rows_up][c_up] densely), c_up % 256 == 0, no bias / relu / statistics; an fp32 residual is fine.
 *   partials : >= ceil(M / 128) * 2 * c_out floats;  tickets : 2 * (c_up / 128) zeroed int32 (zero again on exit). */
typedef struct vp3d_s16_red {
  const float* y_up;
  const float* mean;
  const float* invstd;
  const float* scale;
  const uint8_t* act_bits;
  int64_t rows_up;
  int32_t c_up;
  float p;
  float* partials;
  int64_t partials_floats;
  int32_t* tickets;
  float* dgamma;
  float* dbeta;
  float* dy_bound;
} vp3d_s16_red;
/* Extra arguments of the S16 GEMM.
 *   x_bound / w_bound : device "bounds" of max|.| of the two operand tensors.  A bound is VP3D_BOUND_SLOTS (32)
 *                       consecutive floats whose maximum is a guaranteed bound (measuring kernels spread their atomics
 *                       over the slots; computed bounds sit in slot 0 of a zeroed array).  The S16 exponent is
 *                       e = frexp-exponent(bound) - 15 (so |v * 2^-e| < 2^15); NULL = exponent 0.  The accumulator is
 *                       scaled by 2^(e_x + e_w) before the epilogue.
 *   amax_out          : optional bound, atomically max-ed with |stored value| (the bound of the result for a
 *                       consumer that re-splits it; zero its 32 slots before the launch)
 *   cfg               : tile configuration, -1: planned (vp3d_nt_s16_plan).  20 / 22: 128x128 tiles of 4 waves / 256x256
 *                       tiles of 8 waves with buffer-descriptor LDS-DMA (operands < 2 GiB, else their flat-address forms
 *                       0 / 4 are used); 30: 22 on the rows that fill whole rounds of 256 tiles + 20 on the rest (two
 *                       launches, no split-K); 10, 13, 21, 23: register-pipelined variants kept for measurements
 *   splits            : K slices (1 = none, 0 = planned: vp3d_nt_s16_plan); > 1 needs ws of splits*M*N floats.
 *                       The slices write raw scaled partial matrices [splits][M][N]; a finishing pass sums them and
 *                       applies the epilogue -- unless raw_partials, where the partials ARE the result (wgrad:
 *                       vp3d_wgrad_reduce sums them; y may be NULL). */
/* BatchNorm finalize inside a K-SLICED launch's finishing pass (training forward of the small-M layers; model.py:32,117-119 in
 * training mode): the pass already writes the 64-row-slab statistics; the LAST of its workgroups to finish a 64-column strip
 * (one ticket per strip) merges the strip's slabs in fp64 -- the summation order of vp3d_bn_finalize, bit-identical outputs --
 * and writes scale / shift / save_mean / save_invstd and the running statistics: no separate vp3d_bn_finalize launch between the
 * GEMM and the activation pass.  tickets: (c_out + 63) / 64 zeroed int32 (zero again on exit).  momentum_dev != NULL is read at
 * execution time instead of `momentum`.  Ignored (the caller must finalise itself) unless the launch runs with splits > 1:
 * vp3d_nt_s16_plan says so beforehand. */
typedef struct vp3d_s16_fin {
  const float* gamma;
  const float* beta;
  float eps;
  float momentum;
  const float* momentum_dev;
  float* running_mean;
  float* running_var;
  int64_t* num_batches_tracked;
  float* scale;
  float* shift;
  float* save_mean;
  float* save_invstd;
  int32_t* tickets;
} vp3d_s16_fin;
typedef struct vp3d_s16 {
  const float* x_bound;
  const float* w_bound;
  float* amax_out;
  int32_t cfg;
  int32_t splits;
  float* ws;
  int64_t ws_floats;
  int32_t raw_partials;
  /* eval-mode chaining without an fp32 round trip (all optional, 0 / NULL = off):
   *   res_s16 / res_bound : the epilogue's residual tensor holds S16 rows with the exponent of *res_bound
   *   out_s16             : y receives S16 rows instead of fp32 (same addressing, same bytes).  Their exponent comes
   *                         from the bound  l1[0]*max(in_amax) + l1[1] + (res_amax ? max(res_amax) : 0)  with
   *                         l1 = { max_n sum_k |wt[n][k]|, max_n |bias[n]| } (device floats, weight-only) and in_amax /
   *                         res_amax the MEASURED maxima of the input / residual tensors (bounds written by the
   *                         amax_out of the launches that produced them): a guaranteed bound that is loose by one
   *                         layer only.  It is published in out_wbound[0] (32 zeroed floats) = the bound consumers
   *                         decode y with; amax_out still measures the true values.  No statistics, no split-K. */
  int32_t res_s16;
  const float* res_bound;
  int32_t out_s16;
  const float* in_amax;
  const float* l1;
  const float* res_amax;
  float* out_wbound;
  /* Fused conv + BatchNorm + ReLU + dropout for a conv whose GEMM is cheap to run twice (the expand conv: K = 128; replaces
   * model.py:74 / :127 expand_conv -> expand_bn -> relu -> drop without storing the conv output):
   *   no_output            : pass 1 -- only the epilogue's BatchNorm slab statistics are written (y may be NULL)
   *   act_scale, act_shift : pass 2 -- y receives the S16 rows of dropout(relu(acc*act_scale[n] + act_shift[n])) under the
   *                          exponent of *act_bound (mask: act_drop, element index m*c_out + n; NULL = no dropout) and
   *                          act_bits (may be NULL) the [z > 0 and kept] bits, bit for bit what vp3d_bn_act_fwd_s16 produces
   *                          from a stored conv output.  y must be contiguous [M][c_out], c_out % 64 == 0; no other epilogue. */
  int32_t no_output;
  const float* act_scale;
  const float* act_shift;
  const vp3d_dropout* act_drop;
  const float* act_bound;
  uint8_t* act_bits;
  /* Stream-K configurations (cfg 120 / 122 = the tilings 20 / 22 with the K-tiles of the last, partly filled round of tiles
   * shared equally by a full round of workgroups; the last contributor of a tile sums the partial accumulators in a fixed
   * order and runs the epilogue -- one launch, no finishing pass, bit-reproducible): ws / ws_floats is their workspace and
   * tickets the zeroed int32 counters (zero again on exit; keep one buffer per stream), sizes from vp3d_nt_s16_workspace. */
  int32_t* tickets;
  /* fused BatchNorm-backward column sums of the upstream activation (dgrad launches of the training backward), or NULL */
  const vp3d_s16_red* red;
  /* Rows per BatchNorm statistics slab that epi->stat_sum / stat_m2 were sized for: 0 or 64 = vp3d_stat_slabs(M) rows of 64
   * (every configuration but 28); 32 = (M + 31) / 32 rows -- what tile configuration 28 writes (its 224-row tiles are not
   * multiples of 64 rows) and vp3d_bn_finalize_slab merges.  A launch whose configuration writes another slab size than the
   * caller announced is refused, never silently mis-indexed. */
  int32_t stat_slab_rows;
  /* BatchNorm finalize in the finishing pass of a K-sliced launch, or NULL (vp3d_s16_fin above) */
  const vp3d_s16_fin* fin;
} vp3d_s16;
/* Configuration 29: the same with wave rows of 3 + 2 row blocks = 160 x 256 tiles (the 9,216-row launches: 58 x 4 = 232 tiles = 91 %
 * of one round); both take K slices like 20 / 22 (their statistics then come from the finishing pass: 64-row slabs); the fused
 * BatchNorm-backward sums exist for 28, not for 29.
 * Configuration 28: 224 x 256 tiles (the 8 waves of configuration 22; wave row 0 owns 4 row blocks of 32, wave row 1 owns 3):
 * B * T_out = 27,648 rows of the benchmark step are 124 x 4 = 496 such tiles = 1.94 rounds of the 256 CUs, where the 432
 * tiles of 256 x 256 = 1.69 rounds cost 2.  fp32 or S16 output, bias / ReLU / residual / statistics (32-row slabs in one K
 * slice) / amax epilogues; no fused activation, operands below 2 GiB.
 * vp3d_nt_s16_plan flags: bit 0 = raw partial output (weight gradients), bit 1 = configuration 28 may be chosen (the caller
 * sizes its statistics for vp3d_nt_s16_stat_slab_rows(cfg) and does not attach act / red epilogues). */
int vp3d_nt_s16_plan(int64_t M, int32_t N, int32_t K, int32_t flags, int32_t* cfg, int32_t* splits);
int vp3d_nt_s16_stat_slab_rows(int32_t cfg);
/* vp3d_bn_finalize / vp3d_bn_finalize_dm for statistics in slabs of slab_rows (32 or 64) rows: momentum_dev != NULL is read
 * at execution time instead of `momentum`. */
int vp3d_bn_finalize_slab(vp3d_stream_t stream, int32_t C, int64_t M, int32_t slab_rows, const float* stat_sum, const float* stat_m2,
                          const float* gamma, const float* beta, float eps, float momentum, const float* momentum_dev,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                          float* save_mean, float* save_invstd);
/* The expand layer's forward, dedicated kernel (replaces model.py:74 / :127 / :176
 *   x = self.drop(self.relu(self.expand_bn(self.expand_conv(x))))
 * in training mode without ever storing the conv output): x = S16 im2row rows [M][kpad] (kpad = 32..128, e.g. 3 taps x 34
 * channels = 102 -> 128), w = S16 weight rows [N][kpad], both with their bounds.  Two launches around vp3d_bn_finalize:
 *   statistics pass (stat_sum, stat_m2 != NULL, out == NULL): the 64-row-slab BatchNorm statistics of X W^T, the layout
 *       vp3d_tconv_nt_s16's statistics epilogue writes ([M/64 slabs][N]);
 *   activation pass (out != NULL): out = S16 rows [M][N] of dropout(relu(X W^T * scale[n] + shift[n])) under the exponent of
 *       *out_bound, act_bits (may be NULL) the [z > 0 and kept] bits -- bit for bit what vp3d_tconv_nt_s16 +
 *       vp3d_bn_act_fwd_s16 produce.
 * A workgroup keeps the W fragments of its 256 columns in registers and streams 64-row tiles of X through LDS. */
int vp3d_expand_fwd_s16(vp3d_stream_t stream, int64_t M, int32_t N, int32_t kpad, const void* x, const float* x_bound,
                        const void* w, const float* w_bound, float* stat_sum, float* stat_m2, const float* scale,
                        const float* shift, const vp3d_dropout* drop, const float* out_bound, void* out, uint8_t* act_bits);
/* The per-step prologue of the split-fp16 training forward as TWO launches (untraced, the seven dependent launches they
 * replace -- vp3d_amax_floor, vp3d_im2row_split_s16, vp3d_amax_multi, vp3d_pack_weight, vp3d_split_rows,
 * vp3d_pack_weight_s16_multi, vp3d_act_bounds_multi -- take 125 us before the first GEMM of the cfg3 step):
 *   A: every maximum (tensor i: bound[i] = max(bound[i], floor[i], max|src[i]|), 32-slot bounds zeroed by the caller) and the
 *      activation bounds of vp3d_act_bounds_multi (n_layers == 0: none) -- nothing here depends on anything else;
 *   B: what needs only those: the raw input's im2row + S16 split (vp3d_im2row_split_s16's arguments), the expand conv's weight
 *      W0 [c0][cin0][taps0] -> fp32 pack [c0][kpad] + its S16 rows under *w0_bound, and the C x C weight packs of
 *      vp3d_pack_weight_s16_multi (n_layers == 0: none). */
int vp3d_prologue_a_s16(vp3d_stream_t stream, int32_t n_tensors, const float* const* src, const int64_t* n, float* const* bound,
                        const float* floor_, int32_t n_layers, int32_t C, const float* const* gamma, const float* const* beta,
                        const int64_t* M, const int32_t* res_from, float p, float* act_bounds);
int vp3d_prologue_b_s16(vp3d_stream_t stream, const vp3d_rowmap* map, const float* x, int32_t ldx, int32_t k_valid, int32_t kpad,
                        int32_t one_col, const float* x_bound, void* x_rows, void* x_t, int64_t ld_t, const float* w0, int32_t c0,
                        int32_t cin0, int32_t taps0, const float* w0_bound, float* w0_packed, void* w0_s16, int32_t n_layers,
                        const float* const* w, const int32_t* taps, int32_t c_out, int32_t c_in, const float* w_bounds,
                        void* const* wf, void* const* wd);

/* Workspace of a vp3d_tconv_nt_s16 launch in configuration (cfg, splits): floats (0: none) and int32 tickets (0: none). */
/* 1 when the library was built with -DVP3D_BUILD_EXPERIMENTS (the measured-and-not-adopted S16 GEMM instances: stream-K
 * 120 / 122, register-pipelined 10 / 13 / 21 / 23, 256x128 / 128x256 pairs 24 / 25, four-wave 256x256 26, hybrid pair 30);
 * the default build has only what the planner uses (20 / 22 and their flat-address forms 0 / 4) and rejects the others. */
int vp3d_has_experiments(void);
int vp3d_nt_s16_workspace(int64_t M, int32_t N, int32_t K, int32_t cfg, int32_t splits, int32_t raw_partials, int64_t* ws_floats,
                          int32_t* tickets);
/* y = conv(x; wt) exactly as vp3d_tconv_fwd (same row gather, same epilogue), with x and wt in S16 form
 * (c_in % 32 == 0, 16-byte aligned rows, ldx / ldw in 4-byte units).  Every GEMM of the model runs through it:
 *   forward : wt = S16 of the packed rows Wt[co][k*c_in+ci]
 *   dgrad   : x = dy, wt = S16 of the transposed pack Wd[(k,ci)][co] (strided: plain GEMM; dilated: gather with
 *             tap_step = -dil and Wd[ci][k*c_out+co])
 *   wgrad   : x = dy^T [c_out][M], wt = x^T [(k,ci)][M] (written by the producers of dy / x), K = M, raw_partials. */
int vp3d_tconv_nt_s16(vp3d_stream_t stream, const vp3d_rowmap* map, const void* x, int32_t ldx, int32_t c_in,
                      const void* wt, int32_t ldw, int32_t c_out, float* y, int64_t y_bpitch, int32_t ldy,
                      const vp3d_epilogue* epi, const float* zeros, const vp3d_s16* opts);
/* fp32 rows [M][C] (pitch ld_src floats) -> S16 rows (pitch ld_dst 4-byte units) with the exponent of *bound */
int vp3d_split_rows(vp3d_stream_t stream, int64_t M, int32_t C, const float* src, int64_t ld_src, void* dst,
                    int64_t ld_dst, const float* bound);
/* bound (32 slots) = max(bound, max|src[0..n)|)   (atomic; zero the slots first) */
int vp3d_amax(vp3d_stream_t stream, int64_t n, const float* src, float* bound);

/* Producers of S16 operands (streaming kernels; C % 64 == 0).  "t_out" is the optional TRANSPOSED copy the weight-
 * gradient GEMM reduces over: t_out[(tap*C + c)*ld_t + m/taps] = value(m, c), tap = m % taps (taps = stride of the
 * strided conv that consumes the tensor, 1 otherwise; M % taps == 0), S16 rows along m, ld_t >= roundup(M/taps, 64)
 * 4-byte units, columns [M/taps, roundup(M/taps, 64)) zero-filled.  Bounds are device floats (see vp3d_s16). */
/* out = [res +] dropout(relu(y*scale + shift)) as S16 with the exponent of *out_bound; res (S16, exponent of *res_bound)
 * is addressed as in vp3d_bn_act_fwd; out_f32 (may be NULL) additionally receives the plain fp32 values.
 * act_bits (may be NULL; M*C/8 bytes): the "activation bits" the backward passes read instead of regenerating the
 * dropout mask (model.py:28's mask, which autograd would have saved): bit e of the byte at
 * ((c/64)*M + m)*8 + (c%64)/8  =  [y*scale+shift > 0 and element (m, c - c%8 + e) kept],  c % 8 == 0. */
int vp3d_bn_act_fwd_s16(vp3d_stream_t stream, int64_t M, int32_t C, const float* y, const float* scale,
                        const float* shift, const vp3d_dropout* drop, const void* res, const float* res_bound,
                        int32_t t_dst, int32_t r_t, int32_t r_stride, int32_t r_off, int32_t r_ld,
                        const float* out_bound, void* out, float* out_f32, void* t_out, int64_t ld_t, int32_t taps,
                        uint8_t* act_bits);
/* dy of vp3d_bn_bwd_apply as S16 rows (dy, may be NULL when only the transposed copy is wanted) + transposed copy
 * (taps = 1).  act_bits != NULL: the mask and the ReLU predicate come from the forward's activation bits (drop only
 * supplies p); NULL: they are regenerated from (drop, y, scale, shift) */
int vp3d_bn_bwd_apply_s16(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          const vp3d_dropout* drop, const uint8_t* act_bits, const float* dgamma, const float* dbeta,
                          const float* out_bound, void* dy, void* t_out, int64_t ld_t);
/* vp3d_bn_bwd_reduce on the activation bits: partials[(i*2 + {0,1})*C + c] = partial sums over the rows of block i of
 * g and g*xhat, g = go * keep_scale * bit;  *nparts rows (call with partials == NULL to query), then
 * vp3d_bn_bwd_finalize / vp3d_bn_bwd_finalize_s16.  keep_scale = 1/(1-p) (1 without dropout).  C % 64 == 0. */
int vp3d_bn_bwd_reduce_bits(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                            const float* mean, const float* invstd, const uint8_t* act_bits, float keep_scale,
                            float* partials, int32_t* nparts);
/* vp3d_bn_bwd_reduce_bits + vp3d_bn_bwd_finalize_s16 in ONE launch (replaces the reduction half of autograd's
 * batch_norm backward, model.py:127,134 backward).  Strip-owned: a block walks a row range of one 64-channel strip and
 * writes one partial row; the LAST block of a strip to finish (one ticket per strip) sums the strip's *nparts (<= 32) partial
 * rows (fp64, row order => deterministic), writes dgamma / dbeta of its channels and maxes the bound of dy into dy_bound
 * (zeroed by the caller).  Workspaces: partials [C/64][*nparts][2][64] floats (= *nparts * 2 * C), tickets [*ntickets] int32
 * that must be ZERO on entry and are zero again on exit (keep one buffer per stream); group_partials is no longer used
 * (*ngroups = 0, may be NULL).  Call with partials == NULL to query the sizes.  keep scale = 1/(1-p). */
int vp3d_bn_bwd_reduce_fin_s16(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                               const float* mean, const float* invstd, const uint8_t* act_bits, float p,
                               const float* scale, const float* go_bound, float* partials, double* group_partials,
                               int32_t* tickets, float* dgamma, float* dbeta, float* dy_bound, int32_t* nparts,
                               int32_t* ngroups, int32_t* ntickets);
/* Transposed operand of a conv's weight gradient from the S16 ROWS of the conv's input, for any (stride, dilation, taps):
 *   t_out[(k*C + c)*ld_t + m] = x[b][t*t_stride + t_off + k*tap_step][c],  m = b*t_dst + t  (zero for rows out of range and
 *   for m >= M up to the next multiple of 64).  The strided forward producers write this layout directly when the windows
 *   tile the input; the dilated class / ragged windows build it in backward.  The values keep their exponent. */
int vp3d_gather_t_s16(vp3d_stream_t stream, const vp3d_rowmap* map, const void* x, int32_t C, void* t_out, int64_t ld_t);
/* Backward of the expand layer without materialising dy (videopose3d_amd/engine_s16.py; replaces autograd's backward of
 * model.py:74,127 for expand_conv / expand_bn when no input gradient is wanted).  With X = the im2row rows of the layer
 * input incl. a bias column of ones (vp3d_im2row one_col) and G = go * keep * [bn(y) > 0]:
 *   vp3d_act_mask_s16     G as S16 rows [M][C] (rows_out) and / or as a transposed S16 operand [C][ld_t] (t_out) under
 *                         the bound go_bound/(1-p), published in g_bound (32 floats, zeroed by the caller)
 *   vp3d_wgrad_rows_s16 / vp3d_tconv_nt_s16   raw partials of  P = G^T X [C][kpad]  (from the rows when kpad == 128,
 *                         else from the transposed copies) and of  S = X^T X [kpad][kpad]  (K = rows)
 *   vp3d_sum_slices       S as doubles (fixed summation order)
 *   vp3d_expand_bwd_s16   dbeta = P[:, one_col], dgamma = invstd (<W, P> - mean dbeta), and
 *                         dW = A P + B sX + Cx (W S - mean sX)  (A = scale, B = -A dbeta/M, Cx = -A invstd dgamma/M,
 *                         sX = S[:, one_col]) un-packed to Conv1d.weight layout [C][c_in][taps]. */
int vp3d_act_mask_s16(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* go_bound,
                      const uint8_t* act_bits, float p, float* g_bound, void* rows_out, void* t_out, int64_t ld_t);
int vp3d_sum_slices(vp3d_stream_t stream, int64_t n, int32_t splits, const float* ws, double* out);
/* P = G^T X without G: the raw partials [*nparts][C][kpad] of P straight from the incoming gradient go (fp32 rows [M][C]), the
 * forward's activation bits and the transposed S16 copy x_t [kpad][ld_t] of X (zero for columns >= M).  Replaces
 * vp3d_act_mask_s16 + the split-K GEMM over G (go is read once, G never exists); vp3d_expand_bwd_s16 takes the partials
 * (splits = *nparts).  gram_partials != NULL: the workgroups of the first column slice also accumulate the raw partials
 * [*nparts][kpad][kpad] of S = X^T X (the X^T fragments they hold are both of its operands: no extra loads; vp3d_sum_slices
 * folds them) -- the separate split-K GEMM for S is not needed then.  Call with partials == NULL to query *nparts. */
int vp3d_expand_bwd_p_s16(vp3d_stream_t stream, int64_t M, int32_t C, int32_t kpad, const float* go, const float* go_bound,
                          const uint8_t* act_bits, float p, const void* x_t, int64_t ld_t, const float* x_bound, float* partials,
                          float* gram_partials, int32_t* nparts);
int vp3d_expand_bwd_s16(vp3d_stream_t stream, int32_t C, int32_t c_in, int32_t taps, int32_t kpad, int32_t one_col, int64_t M,
                        int32_t splits, const float* p_partials, const double* gram, const float* w_packed,
                        const float* scale, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* dw);
/* The same from the FORWARD's centred second-moment matrix (vp3d_expand_stats_gram_s16's `gram` output, kept until backward):
 * x_t / ld_t / x_bound name the transposed S16 X whose first column holds the offsets o_k the forward centred with;
 *   (X^T X)_ij = G_ij + o_i G[one][j] + o_j G[one][i] + M o_i o_j   in fp64 inside the launch
 * -- the backward then needs no second-moment matrix of its own: vp3d_expand_bwd_p_s16 runs with gram_partials == NULL and
 * vp3d_sum_slices is not called (x_t == NULL: `gram` is X^T X itself, i.e. vp3d_expand_bwd_s16). */
int vp3d_expand_bwd_gram_s16(vp3d_stream_t stream, int32_t C, int32_t c_in, int32_t taps, int32_t kpad, int32_t one_col, int64_t M,
                             int32_t splits, const float* p_partials, const double* gram, const void* x_t, int64_t ld_t,
                             const float* x_bound, const float* w_packed, const float* scale, const float* mean,
                             const float* invstd, float* dgamma, float* dbeta, float* dw);
/* Weight gradient of a (strided) conv straight from S16 ROWS -- no transposed copies: the kernel transposes on the LDS
 * read (ds_read_b64_tr_b16).  Replaces autograd's conv weight gradient (model.py:178-180 backward) like
 * vp3d_tconv_nt_s16 in raw-partials mode does:
 *   partials[s][co][tap*c_in + ci] = sum over the rows m of K-slice s of  dy[m][co] * x[m*taps + tap][ci]
 * dy: [M][ld_dy] S16 (exponent of *dy_bound), x: [M*taps][ld_x] S16 (exponent of *x_bound; for a conv of stride == taps
 * these are simply the rows of its input), partials: splits * c_out * taps*c_in floats, summed and un-packed by
 * vp3d_wgrad_reduce(partials, taps*c_in, splits, c_out, c_in, taps, dw).  c_out % 256 == 0, c_in % 256 == 0 or
 * c_in == 128 (one narrow column tile per tap: the expand conv's 128-wide im2row rows), operands < 2 GiB. */
int vp3d_wgrad_rows_s16(vp3d_stream_t stream, int64_t M, const void* dy, int64_t ld_dy, int32_t c_out,
                        const float* dy_bound, const void* x, int64_t ld_x, int32_t taps, int32_t c_in,
                        const float* x_bound, int32_t splits, float* partials);
/* fp32 rows -> S16 rows (out, may be NULL) and / or the transposed S16 copy (t_out, may be NULL; taps = 1) */
int vp3d_split_t(vp3d_stream_t stream, int64_t M, int32_t C, const float* src, int64_t ld_src, const float* bound,
                 void* out, int64_t ld_out, void* t_out, int64_t ld_t);
/* vp3d_im2row + vp3d_split_t in one pass (the expand conv's operand straight from the [B, T, J*F] input: the 128-wide fp32
 * staging rows are never written): out = S16 rows [M][kpad], t_out = their transposed S16 copy [kpad][ld_t] (either may be
 * NULL), both under the exponent of *bound, which must cover the input AND the bias column's 1 (vp3d_amax_floor).
 * kpad % 64 == 0.  Row map and one_col as vp3d_im2row. */
int vp3d_im2row_split_s16(vp3d_stream_t stream, const vp3d_rowmap* map, const float* x, int32_t ldx, int32_t k_valid, int32_t kpad,
                          int32_t one_col, const float* bound, void* out, void* t_out, int64_t ld_t);
/* vp3d_amax with a floor: *bound = max(*bound, floor, max|src|) */
int vp3d_amax_floor(vp3d_stream_t stream, int64_t n, const float* src, float floor, float* bound);
/* reference Conv1d.weight [c_out][c_in][taps] -> S16 packs with the exponent of *bound (c_out, c_in % 64 == 0, taps <= 3):
 *   wf (may be NULL): Wt[co*ld_f + k*c_in + ci]                    (forward)
 *   wd (may be NULL): strided form  Wd[(k*c_in + ci)*ld_d + co]     (dgrad of a stride == taps conv, plain GEMM)
 *                     dilated form  Wd[ci*ld_d + k*c_out + co]      (dgrad of a dilated conv, gather with -dil) */
int vp3d_pack_weight_s16(vp3d_stream_t stream, const float* w, int32_t c_out, int32_t c_in, int32_t taps,
                         const float* bound, void* wf, int64_t ld_f, void* wd, int64_t ld_d, int32_t dilated_form);
/* *out = max_c(|gamma_c|*sqrt(M-1) + |beta_c|)/(1-p) + (res_bound ? *res_bound : 0): a guaranteed bound of
 * |[res +] dropout(relu(bn(y)))| for batch statistics over M rows (Samuelson's inequality) */
int vp3d_act_bound(vp3d_stream_t stream, int32_t C, int64_t M, const float* gamma, const float* beta, float p,
                   const float* res_bound, float* out);
/* *out = max_c |scale_c|*(g + |dbeta_c|/M + sqrt(M-1)*|dgamma_c|/M), g = *go_bound/(1-p): bound of vp3d_bn_bwd_apply's dy */
int vp3d_dy_bound(vp3d_stream_t stream, int32_t C, int64_t M, const float* scale, const float* dgamma,
                  const float* dbeta, const float* go_bound, float p, float* out);

/* One-launch forms of the per-layer prologue of a training step (at most 16 tensors / layers; the pointer tables are
 * HOST arrays, copied into the kernel arguments).  bounds: n consecutive bounds (32 floats each), zeroed by the caller. */
int vp3d_amax_multi(vp3d_stream_t stream, int32_t n_tensors, const float* const* src, const int64_t* n, float* bounds);
/* vp3d_pack_weight_s16 (strided dgrad form) for n_layers [c_out][c_in][taps_i] weights; layer i uses bounds + 32*i */
int vp3d_pack_weight_s16_multi(vp3d_stream_t stream, int32_t n_layers, const float* const* w, const int32_t* taps,
                               int32_t c_out, int32_t c_in, const float* bounds, void* const* wf, void* const* wd);
/* vp3d_act_bound for every layer of the stack: bounds[i] = max_c(|gamma_i|*sqrt(M_i-1) + |beta_i|)/(1-p) +
 * (res_from[i] >= 0 ? bounds[res_from[i]] : 0)   (res_from[i] < i) */
int vp3d_act_bounds_multi(vp3d_stream_t stream, int32_t n_layers, int32_t C, const float* const* gamma,
                          const float* const* beta, const int64_t* M, const int32_t* res_from, float p, float* bounds);
/* vp3d_bn_bwd_finalize + vp3d_dy_bound in one launch (dy_bound: zeroed bound, atomically max-ed) */
int vp3d_bn_bwd_finalize_s16(vp3d_stream_t stream, int32_t C, int64_t M, const float* partials, int32_t nparts,
                             float* dgamma, float* dbeta, const float* scale, const float* go_bound, float p,
                             float* dy_bound);

/* dx[b,s,:] = sum_k dy[b, map(s,k), :] @ W_k^T (+ epilogue: the residual-gradient scatter).
 *   M = B*t_dst rows of dx, N = n_out columns, K = taps*c_out.
 *   wt: the SAME forward-packed weights; column n of tap k is read at wt[co*ldw + k*w_tap_stride + n]. */
int vp3d_tconv_dgrad(vp3d_stream_t stream, const vp3d_rowmap* map, const float* dy, int32_t lddy, int32_t c_out,
                     const float* wt, int32_t ldw, int32_t w_tap_stride, int32_t n_out, float* dx,
                     int64_t dx_bpitch, int32_t lddx, const vp3d_epilogue* epi, const float* zeros, float* splitk_ws,
                     int64_t splitk_ws_floats);

/* dWt[co][k*c_in + ci] = sum_m dy[m][co] * x[map(m,k)][ci].   Reduction over M = B*t_dst rows, split in
 * `splits` slices whose partial [c_out, taps*c_in] matrices go to `partials` (splits*c_out*taps*c_in floats);
 * vp3d_wgrad_reduce sums them.  With splits == 1 `partials` is the result itself. */
int vp3d_tconv_wgrad(vp3d_stream_t stream, const vp3d_rowmap* map, const float* dy, int32_t lddy, int32_t c_out,
                     const float* x, int32_t ldx, int32_t c_in, float* partials, int32_t splits,
                     const float* zeros);
/* dW[co][ci][k] (reference Conv1d.weight layout) = sum_s partials[s][co*ld_part + k*c_in + ci]
 * (partial matrices are c_out*ld_part floats apart; ld_part >= taps*c_in) */
int vp3d_wgrad_reduce(vp3d_stream_t stream, const float* partials, int32_t ld_part, int32_t splits, int32_t c_out,
                      int32_t c_in, int32_t taps, float* dw);

/* out[co*ld_out + k*c_in + ci] = w[co][ci][k] * (scale ? scale[co] : 1)   (reference layout -> packed rows);
 * columns [taps*c_in, ld_out) are zero-filled (K padding for the expand conv) */
int vp3d_pack_weight(vp3d_stream_t stream, const float* w, int32_t c_out, int32_t c_in, int32_t taps,
                     const float* scale, float* out, int32_t ld_out);

/* Row staging for convs whose taps*C_in is not a multiple of 32 (expand_conv: 3*34 = 102): row m = (b,t) of `out`
 * receives the k_valid contiguous floats at x[(b*t_src + t*t_stride)*ldx] followed by zeros up to kpad, so that the
 * conv becomes a 1-tap GEMM over 16-byte-aligned kpad-wide rows (dil == 1 only: taps are adjacent rows).
 * one_col in [k_valid, kpad): that padding column holds 1 instead of 0 (a bias column -- the matching weight column
 * is zero, so the forward is unchanged; vp3d_expand_bwd_s16 reads column sums from it); -1: none. */
int vp3d_im2row(vp3d_stream_t stream, const vp3d_rowmap* map, const float* x, int32_t ldx, int32_t k_valid,
                int32_t kpad, int32_t one_col, float* out);

/* eval-mode BN folding: scale[c] = gamma/sqrt(running_var+eps), shift[c] = beta - running_mean*scale */
int vp3d_bn_fold(vp3d_stream_t stream, int32_t C, const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float eps, float* scale, float* shift);

/* Training BatchNorm statistics from the GEMM's slab partials (Chan merge in fp64):
 * save_mean, save_invstd, scale = gamma*invstd, shift = beta - mean*scale; running_mean/var updated in place
 * (unbiased variance), num_batches_tracked += 1 (any of the running pointers may be NULL). */
int vp3d_bn_finalize(vp3d_stream_t stream, int32_t C, int64_t M, const float* stat_sum, const float* stat_m2,
                     const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                     float* save_mean, float* save_invstd);
/* The same with the momentum read from device memory (momentum_dev[0]) at execution time: a launch captured into a
 * hipGraph follows run.py's per-epoch set_bn_momentum (run.py:590-593, model.py:36-39) without a re-capture. */
int vp3d_bn_finalize_dm(vp3d_stream_t stream, int32_t C, int64_t M, const float* stat_sum, const float* stat_m2,
                        const float* gamma, const float* beta, float eps, const float* momentum_dev, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                        float* save_mean, float* save_invstd);

/* out[m][c] = (res ? res[resrow(m)][c] : 0) + dropout(relu(y[m][c]*scale[c] + shift[c])).
 * resrow(m = b*t_dst + t) = b*r_t + t*r_stride + r_off (row pitch r_ld).  drop may be NULL (p = 0). */
int vp3d_bn_act_fwd(vp3d_stream_t stream, int64_t M, int32_t C, const float* y, const float* scale,
                    const float* shift, const vp3d_dropout* drop, const float* res, int32_t t_dst, int32_t r_t,
                    int32_t r_stride, int32_t r_off, int32_t r_ld, float* out);

/* Backward of a = dropout(relu(bn(y))):  g = go*keep*[z>0];  partial sums of g and g*xhat per channel into
 * partials[nparts][2][C]; returns the number of parts through *nparts (query with partials == NULL). */
int vp3d_bn_bwd_reduce(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                       const float* scale, const float* shift, const float* mean, const float* invstd,
                       const vp3d_dropout* drop, float* partials, int32_t* nparts);
/* dgamma[c] = sum g*xhat, dbeta[c] = sum g   (fp64 accumulation over the parts) */
int vp3d_bn_bwd_finalize(vp3d_stream_t stream, int32_t C, const float* partials, int32_t nparts, float* dgamma,
                         float* dbeta);
/* dy = scale*(g - dbeta/M - xhat*dgamma/M) */
int vp3d_bn_bwd_apply(vp3d_stream_t stream, int64_t M, int32_t C, const float* go, const float* y,
                      const float* scale, const float* shift, const float* mean, const float* invstd,
                      const vp3d_dropout* drop, const float* dgamma, const float* dbeta, float* dy);

/* the same with g = go*keep*[z>0] already formed (by the fused dgrad epilogue, vp3d_act_bwd):
 * dy = scale*(g - dbeta/M - xhat*dgamma/M) */
int vp3d_bn_bwd_apply_g(vp3d_stream_t stream, int64_t M, int32_t C, const float* g, const float* y,
                        const float* scale, const float* mean, const float* invstd, const float* dgamma,
                        const float* dbeta, float* dy);
/* number of [2][c_stat] partial rows a fused vp3d_act_bwd epilogue writes for an [M, N] dgrad output */
int64_t vp3d_act_bwd_parts(int64_t M, int32_t N, int32_t c_stat);

/* out[n] = sum_m g[m*ld + n]   (bias gradient of the shrink conv) */
int vp3d_colsum(vp3d_stream_t stream, int64_t M, int32_t N, const float* g, int32_t ld, float* out);

/* The head of the model at small row counts: the 3*J_out-column shrink conv (reference common/model.py:33, applied at :137 /
 * :196) and its whole backward, as dedicated kernels (csrc/vp3d_head.hip) -- fp32 FMAs, the reference's own weight layout
 * w[N][K] (a 1-tap Conv1d weight is its own pack), deterministic sums.  M = B * T_out rows, K = channels, N = 3 * J_out.
 *   vp3d_head_supported: 1 when (M, K, N) is served (M <= VP3D_HEAD_MAX_ROWS, K % 4 == 0, K <= 4096, N <= 128); callers keep the
 *                        general GEMM entry points (vp3d_tconv_fwd / _dgrad / _wgrad + vp3d_colsum) for everything else
 *   vp3d_head_fwd : out[m][n] = bias[n] + sum_k h[m][k] * w[n][k]                         (bias may be NULL)
 *   vp3d_head_bwd : dh[m][k] = sum_n dy[m][n] * w[n][k];  dh_bound (32 zeroed slots, or NULL) receives max|dh| (the split-fp16
 *                   engine's bound of its first backward operand);  ws (vp3d_head_bwd_ws_floats floats, or NULL: no weight /
 *                   bias gradient) receives the per-32-row-slice partials of dw / db in the SAME launch
 *   vp3d_head_fold: dw[n][k] = sum_slices, db[n] = sum_slices (slice order; db may be NULL) -- nothing in backward reads them,
 *                   so the caller may enqueue this on another stream behind vp3d_head_bwd */
#define VP3D_HEAD_MAX_ROWS 4096
int vp3d_head_supported(int64_t M, int32_t K, int32_t N);
int64_t vp3d_head_bwd_ws_floats(int64_t M, int32_t K, int32_t N);
int vp3d_head_fwd(vp3d_stream_t stream, int64_t M, int32_t K, int32_t N, const float* h, const float* w, const float* bias,
                  float* out);
int vp3d_head_bwd(vp3d_stream_t stream, int64_t M, int32_t K, int32_t N, const float* dy, const float* h, const float* w,
                  float* dh, float* dh_bound, float* ws);
int vp3d_head_fold(vp3d_stream_t stream, int64_t M, int32_t K, int32_t N, const float* ws, float* dw, float* db);

/* materialise the dropout keep*scale mask of a layer as floats (tests / debugging only) */
int vp3d_dropout_mask(vp3d_stream_t stream, int64_t n, const vp3d_dropout* drop, float* out);

/* camera.py:37-67 / 69-90: X [n_cam, pts_per_cam, 3], cam [n_cam, 9] -> out [n_cam, pts_per_cam, 2] */
int vp3d_project_to_2d_fwd(vp3d_stream_t stream, int64_t n_cam, int64_t pts_per_cam, const float* X,
                           const float* cam, int32_t linear, float* out);
int vp3d_project_to_2d_bwd(vp3d_stream_t stream, int64_t n_cam, int64_t pts_per_cam, const float* X,
                           const float* cam, const float* gout, int32_t linear, float* dX);

/* ------------------------------------------------------------------------------------------------------------
 * The callers either side of the temporal stack inside one training / evaluation step.
 * ------------------------------------------------------------------------------------------------------------ */

/* Batch assembly on the device: replaces the per-sample numpy loop of reference common/generators.py:105-149
 * (ChunkedGenerator.next_epoch) and the padding + mirrored copy of :216-239 (UnchunkedGenerator.next_epoch).
 * The dataset stays resident in HBM as concatenated fp32 frames; the host keeps the (seq, start, flip) pair list
 * and its permutation (generators.py:39-48, 89-97) and uploads only the chunk table of one batch.
 *   chunks   : device int32 [n_chunks][3] = (seq index, start_3d, flip)
 *   seq_off  : device int64 [n_seq + 1]: sequence s owns frames [seq_off[s], seq_off[s+1]) of poses_2d / poses_3d
 *   out_2d[i][f] = poses_2d[seq][clamp(start - pad - causal_shift + f, 0, len-1)],  f in [0, chunk_length + 2*pad)
 *   out_3d[i][f] = poses_3d[seq][clamp(start + f, 0, len-1)],                        f in [0, chunk_length)
 *   out_cam[i]   = cameras[seq]
 *   mirrored chunks (flip != 0): coordinate 0 negated, out joint j reads joint kps_perm[j] / joints_perm[j]
 *   (perm = the reference's kps_left+kps_right <- kps_right+kps_left assignment as a gather map), camera
 *   entries 2 and 7 negated.  poses_3d / cameras (and their outputs) may be NULL.  Bit-exact: pure copies/negations. */
typedef struct vp3d_gather {
  int32_t n_chunks;
  const int32_t* chunks;
  const int64_t* seq_off;
  const float* poses_2d;
  int32_t j2, f2;
  const int32_t* kps_perm;
  const float* poses_3d;
  int32_t j3, f3;
  const int32_t* joints_perm;
  const float* cameras;
  int32_t cam_dim;
  int32_t chunk_length, pad, causal_shift;
  float* out_2d;
  float* out_3d;
  float* out_cam;
} vp3d_gather;
int vp3d_gather_chunks(vp3d_stream_t stream, const vp3d_gather* g);

/* Test-time augmentation fold (reference run.py:677-680): pred [2][n_frames][n_joints][dim], copy 1 is the
 * prediction for the mirrored input:  out = (pred[0] + unflip(pred[1])) / 2,  unflip = negate coordinate 0 and
 * read joint joints_perm[j] (NULL: no joint swap, as for the trajectory model). */
int vp3d_tta_fold(vp3d_stream_t stream, int64_t n_frames, int32_t n_joints, int32_t dim, const float* pred,
                  const int32_t* joints_perm, float* out);

/* reference common/loss.py:11-17 mpjpe and :19-25 weighted_mpjpe, forward and gradient in one pass:
 *   loss[0]  = (1/n_pts) * sum_i w_i * || pred_i - target_i ||_2          (rows of `dim` floats; w NULL = 1)
 *   grad[i]  = w_i * (pred_i - target_i) / (||.||_2 * n_pts)               (d loss / d pred; 0 where the norm is 0;
 *                                                                           grad may be NULL: loss only)
 * ws: workspace of vp3d_mpjpe_ws_bytes(n_pts) bytes, 8-byte aligned (0 bytes / NULL up to 256 points); its FIRST 8 BYTES hold
 * the ticket of the one-launch block fold and must be zero on entry -- the call leaves them zero, so a workspace that was
 * zeroed once serves every later call on the same stream.
 * Deterministic (the blocks' fp64 partials are folded in block order by the block that arrives last; no float atomics). */
int64_t vp3d_mpjpe_ws_bytes(int64_t n_pts);
int vp3d_mpjpe(vp3d_stream_t stream, int64_t n_pts, int32_t dim, const float* pred, const float* target,
               const float* w, float* loss, float* grad, void* ws);

/* reference run.py:252,264,420: torch.optim.Adam(params, lr, amsgrad=True).step(), as ONE pass over flat fp32
 * buffers (parameters, gradients and optimizer state of all tensors concatenated), torch/optim/adam.py arithmetic:
 *   g += weight_decay*p ; m += (1-beta1)*(g-m) ; v = v*beta2 + (1-beta2)*g*g ; vmax = max(vmax, v)
 *   p -= lr/(1-beta1^step) * m / (sqrt(amsgrad ? vmax : v)/sqrt(1-beta2^step) + eps)
 * `step` is the 1-based index of THIS update.  max_exp_avg_sq may be NULL unless amsgrad. */
typedef struct vp3d_adam {
  double lr, beta1, beta2, eps, weight_decay; /* Python-float (double) hyper-parameters, as torch keeps them: 1-beta2
                                                 must be formed in double before it is rounded to fp32 */
  int64_t step;
  int32_t amsgrad;
} vp3d_adam;
int vp3d_adam_step(vp3d_stream_t stream, int64_t n, float* param, const float* grad, float* exp_avg,
                   float* exp_avg_sq, float* max_exp_avg_sq, const vp3d_adam* h);

/* BatchNorm coefficients of the expand layer (model.py:32,74 / :127 / :188: expand_bn(expand_conv(x)) in training mode) WITHOUT a pass
 * over the conv output: y = X W^T with only kpad <= 128 input columns, so mean_n = W[n] . mean(x) and var_n = W[n]^T Cov(x) W[n].
 * One MFMA pass over the transposed S16 copy of the im2row rows (xt [kpad][ld_t], the copy the no-dy backward reads) forms the
 * second-moment matrix of X CENTRED at its first row (no E[x^2] - E[x]^2 cancellation; the constant-1 column `one_col` of the
 * rows is not shifted and yields the column sums), `part` = vp3d_expand_stats_gram_groups(M) partial [kpad][kpad] matrices, `gram`
 * = their sum as kpad * kpad doubles; then C quadratic forms in fp64 give exactly vp3d_bn_finalize's outputs (momentum_dev != NULL
 * is read at execution time instead of `momentum`).  w_packed: the fp32 weight rows [C][kpad] (columns >= kv are padding).
 * The matrix carries a relative error of ~4e-9 (exact products, one fp32 accumulator per 64-row slab, fp64 across slabs), which
 * var_n sees multiplied by kappa_n = sum_ij |w_i Cov_ij w_j| / (var_n + eps) -- large for temporal-difference filters over
 * correlated keypoint columns.  illcond (device int32, may be NULL; the caller zeroes it when it starts a measurement) receives
 * atomicMax of floor(log2 kappa_n) over the channels: above VP3D_GRAM_KAPPA_LOG2_MAX the statistics are no longer inside what
 * the reference's own fp32 conv + BatchNorm leaves (1e-6 sqrt(kappa)) and the caller should use the pass over the conv output
 * (vp3d_expand_fwd_s16 statistics + vp3d_bn_finalize), which has no such term. */
#define VP3D_GRAM_KAPPA_LOG2_MAX 16
int vp3d_expand_stats_gram_groups(int64_t M);
int vp3d_expand_stats_gram_s16(vp3d_stream_t stream, int64_t M, int32_t C, int32_t kpad, int32_t kv, int32_t one_col, const void* xt,
                               int64_t ld_t, const float* x_bound, const float* w_packed, float* part, double* gram,
                               const float* gamma, const float* beta, float eps, float momentum, const float* momentum_dev,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                               float* save_mean, float* save_invstd, int32_t* illcond);

/* Guard of the split-fp16 arithmetic (videopose3d_amd/range_guard.py).  The S16 operand format keeps one exponent per
 * tensor, while the reference's BatchNorm affine and conv weights are unconstrained (common/model.py:32,102,113-119): this
 * measures, on the device and without a host synchronisation, how far the parameters are from the regime in which a hot
 * channel pushes the others down the format's range.  "spread" of per-group magnitudes g_i := E(max g) - E(median of the
 * non-zero g), E = binary exponent.  atomicMax-ed into out (int32[2], zeroed by the caller for a fresh measurement):
 *   out[0]  BatchNorm layers l < n_layers (host pointer tables): groups = channels, g_c = |gamma_l[c]| * kfac[l] + |beta_l[c]|
 *           (kfac = sqrt(M_l - 1) in training: the activation bound of vp3d_act_bounds_multi per channel; ~4 in eval);
 *   out[1]  tensors i < n_tensors: groups = rows of row_len[i] contiguous floats (Conv1d.weight [C_out][C_in*taps]: output rows).
 * ws: sum(rows) ints of workspace.  At most vp3d_range_max_tensors() layers / tensors. */
/* The same statistic over the COLUMNS of a row-major [M][C] tensor (C <= 1024): the input batch of model.py:63-77 (C = joints x
 * features: one hot keypoint column) and the loss gradient at the head (C = joints x 3).  atomicMax-ed into *out; ws = C ints,
 * zero on entry, left zero. */
int vp3d_range_cols(vp3d_stream_t stream, int64_t M, int32_t C, const float* x, int64_t ld, int32_t* ws, int32_t* out);
int vp3d_range_max_tensors(void);
int vp3d_range_stats(vp3d_stream_t stream, int32_t n_layers, int32_t C, const float* const* gamma, const float* const* beta,
                     const float* kfac, int32_t n_tensors, const float* const* w, const int64_t* rows, const int64_t* row_len,
                     int32_t* ws, int64_t ws_ints, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* VP3D_H_ */
