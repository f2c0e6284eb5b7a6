// C ABI entry points for the temporal-convolution GEMMs + error plumbing (see include/vp3d.h).
#include <atomic>
#include <stdarg.h>
#include <stdio.h>

#include "vp3d_internal.h"

namespace vp3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::atomic<int64_t> g_launches{0};     // kernel launches issued by this library in this process (vp3d_launch_count)

void count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return VP3D_E_LAUNCH;
  }
  return VP3D_OK;
}

static int fill_epi(Epi* e, const vp3d_epilogue* u, float* C, int64_t c_bpitch, int32_t ldc, int32_t n_cols) {
  e->C = C;
  e->c_bpitch = c_bpitch;
  e->ldc = ldc;
  e->bias = nullptr;
  e->relu = 0;
  e->R = nullptr;
  e->r_bpitch = 0;
  e->r_ld = e->r_t = e->r_stride = e->r_off = e->r_col0 = e->r_cols = 0;
  e->stat_sum = e->stat_m2 = nullptr;
  e->vec = 0;
  e->ab_y = e->ab_scale = e->ab_shift = e->ab_mean = e->ab_invstd = nullptr;
  e->ab_g = e->ab_part = nullptr;
  e->ab_c = 0;
  e->ab_store_v = 1;
  e->ab_drop = make_drop(nullptr);
  e->bound_a = e->bound_b = nullptr;
  e->amax_out = nullptr;
  e->r_s16 = e->c_s16 = 0;
  e->r_bound = e->in_amax = e->l1 = e->res_amax = nullptr;
  e->out_wbound = nullptr;
  e->no_out = 0;
  e->act_scale = e->act_shift = e->act_bound = nullptr;
  e->act_bits = nullptr;
  e->red = 0;
  e->red_bits = nullptr;
  e->red_m = e->red_row_b = e->red_row_t = 0;
  e->red_inv_keep = e->red_inv_m = e->red_sqrt_m1 = 0.f;
  e->red_cnt = nullptr;
  e->red_dgamma = e->red_dbeta = e->red_dy_bound = nullptr;
  e->fin_tickets = nullptr;
  e->fin_gamma = e->fin_beta = e->fin_momentum_dev = nullptr;
  e->fin_eps = e->fin_momentum = 0.f;
  e->fin_running_mean = e->fin_running_var = nullptr;
  e->fin_nbt = nullptr;
  e->fin_scale = e->fin_shift = e->fin_save_mean = e->fin_save_invstd = nullptr;
  if (u == nullptr) return VP3D_OK;
  e->bias = u->bias;
  e->relu = u->relu;
  if (u->residual != nullptr) {
    VP3D_REQUIRE(u->r_t > 0 && u->r_ld > 0 && u->r_cols > 0 && u->r_col0 >= 0 && u->r_col0 + u->r_cols <= n_cols,
                 "epilogue: bad residual map (r_t=%d r_ld=%d r_col0=%d r_cols=%d n=%d)", u->r_t, u->r_ld, u->r_col0,
                 u->r_cols, n_cols);
    e->R = u->residual;
    e->r_bpitch = u->r_bpitch;
    e->r_ld = u->r_ld;
    e->r_t = u->r_t;
    e->r_stride = u->r_stride;
    e->r_off = u->r_off;
    e->r_col0 = u->r_col0;
    e->r_cols = u->r_cols;
  }
  VP3D_REQUIRE((u->stat_sum == nullptr) == (u->stat_m2 == nullptr), "epilogue: stat_sum and stat_m2 go together");
  e->stat_sum = u->stat_sum;
  e->stat_m2 = u->stat_m2;
  if (u->act_bwd != nullptr) {
    const vp3d_act_bwd* ab = u->act_bwd;
    VP3D_REQUIRE(ab->y_up && ab->scale && ab->shift && ab->mean && ab->invstd && ab->g_out && ab->partials,
                 "epilogue: act_bwd has a null pointer");
    VP3D_REQUIRE(ab->c_stat > 0 && ab->c_stat % 4 == 0 && n_cols % ab->c_stat == 0 && n_cols % 128 == 0,
                 "epilogue: act_bwd needs c_stat %% 4 == 0, N %% c_stat == 0 and N %% 128 == 0 (c_stat=%d N=%d)",
                 ab->c_stat, n_cols);
    VP3D_REQUIRE(aligned16(ab->y_up) && aligned16(ab->g_out) && aligned16(ab->scale) && aligned16(ab->shift) &&
                     aligned16(ab->mean) && aligned16(ab->invstd), "epilogue: act_bwd buffers must be 16-byte aligned");
    VP3D_REQUIRE(ab->store_v == 0 || C != nullptr, "epilogue: act_bwd.store_v needs an output buffer");
    if (ab->drop) VP3D_REQUIRE(ab->drop->p >= 0.f && ab->drop->p < 1.f, "epilogue: act_bwd dropout p=%f", ab->drop->p);
    e->ab_y = ab->y_up;
    e->ab_scale = ab->scale;
    e->ab_shift = ab->shift;
    e->ab_mean = ab->mean;
    e->ab_invstd = ab->invstd;
    e->ab_g = ab->g_out;
    e->ab_part = ab->partials;
    e->ab_c = ab->c_stat;
    e->ab_store_v = ab->store_v;
    e->ab_drop = make_drop(ab->drop);
  }
  return VP3D_OK;
}

// K-slicing only when the caller handed a workspace of vp3d_rows_gemm_ws_floats() (checked by launch_rows_gemm)
static void set_splits(RowsGemmArgs* a, float* ws, int64_t ws_floats) {
  a->splits = 1;
  a->kt_per_split = 0;
  a->pos_full = a->tail_pos = 0;
  a->part = ws;
  a->part_floats = ws != nullptr ? ws_floats : 0;
}

static int check_map(const vp3d_rowmap* m, const char* who) {
  VP3D_REQUIRE(m != nullptr, "%s: null rowmap", who);
  VP3D_REQUIRE(m->batch > 0 && m->t_dst > 0 && m->t_src > 0 && m->taps > 0, "%s: bad rowmap (B=%d t_dst=%d t_src=%d taps=%d)",
               who, m->batch, m->t_dst, m->t_src, m->taps);
  VP3D_REQUIRE((int64_t)m->batch * m->t_dst < (int64_t)1 << 31 && (int64_t)m->batch * m->t_src < (int64_t)1 << 31,
               "%s: more than 2^31 rows", who);
  return VP3D_OK;
}

}  // namespace vp3d

using namespace vp3d;

extern "C" {

int vp3d_version(void) { return VP3D_VERSION; }
int64_t vp3d_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
const char* vp3d_last_error(void) { return g_err; }
int64_t vp3d_stat_slabs(int64_t M) { return (M + 63) / 64; }
int vp3d_rows_gemm_splits(int64_t M, int32_t N, int32_t K) {
  if (M <= 0 || M >= ((int64_t)1 << 31) || N <= 0 || K <= 0) return 1;
  return rows_gemm_splits((int)M, N, K);
}

int64_t vp3d_rows_gemm_ws_floats(int64_t M, int32_t N, int32_t K) {
  if (M <= 0 || M >= ((int64_t)1 << 31) || N <= 0 || K <= 0) return 0;
  return rows_gemm_ws_floats((int)M, N, K);
}

int64_t vp3d_act_bwd_parts(int64_t M, int32_t N, int32_t c_stat) {
  if (M <= 0 || N <= 0 || c_stat <= 0 || N % c_stat != 0) return 0;
  return ((M + 63) / 64) * (N / c_stat);
}

int vp3d_wgrad_splits(int64_t M, int32_t c_out, int32_t n_cols) {
  if (M <= 0 || M >= ((int64_t)1 << 31) || c_out <= 0 || n_cols <= 0) return 1;
  return red_gemm_splits((int)M, c_out, n_cols);
}

int vp3d_tconv_fwd(vp3d_stream_t stream, const vp3d_rowmap* map, const float* x, int32_t ldx, int32_t c_in,
                   const float* wt, int32_t ldw, int32_t c_out, float* y, int64_t y_bpitch, int32_t ldy,
                   const vp3d_epilogue* epi, const float* zeros, float* splitk_ws, int64_t splitk_ws_floats) {
  int rc = check_map(map, "tconv_fwd");
  if (rc) return rc;
  VP3D_REQUIRE(x && wt && y && zeros, "tconv_fwd: null pointer");
  VP3D_REQUIRE(c_in > 0 && c_out > 0 && ldx >= 1 && ldw >= map->taps * c_in && ldy >= c_out,
               "tconv_fwd: bad sizes (c_in=%d c_out=%d ldx=%d ldw=%d ldy=%d)", c_in, c_out, ldx, ldw, ldy);
  RowsGemmArgs a;
  a.A = x;
  a.B = wt;
  a.zeros = zeros;
  a.M = map->batch * map->t_dst;
  a.N = c_out;
  a.K = map->taps * c_in;
  a.lda = ldx;
  a.c_src = c_in;
  a.ldb = ldw;
  a.b_tap_stride = 0;
  a.t_dst = map->t_dst;
  a.t_src = map->t_src;
  a.t_stride = map->t_stride;
  a.tap_step = map->tap_step;
  a.t_off = map->t_off;
  a.taps = map->taps;
  a.m_tiles = (a.M + 127) / 128;
  a.n_tiles = (a.N + 127) / 128;
  VP3D_REQUIRE(epi == nullptr || epi->act_bwd == nullptr, "tconv_fwd: act_bwd is a dgrad-only epilogue");
  rc = fill_epi(&a.epi, epi, y, y_bpitch, ldy, c_out);
  if (rc) return rc;
  set_splits(&a, splitk_ws, splitk_ws_floats);
  return launch_rows_gemm((hipStream_t)stream, a, /*b_kcontig=*/true);
}

int vp3d_nt_s16_stat_slab_rows(int32_t cfg) { return nt_s16_stat_slab_rows(cfg); }

int vp3d_nt_s16_plan(int64_t M, int32_t N, int32_t K, int32_t flags, int32_t* cfg, int32_t* splits) {
  VP3D_REQUIRE(M > 0 && M < ((int64_t)1 << 31) && N > 0 && K > 0 && cfg && splits && (flags & ~3) == 0, "nt_s16_plan: bad argument");
  int c, sp;
  plan_nt_s16((int)M, N, K, 1, flags & 1, &c, &sp, (flags >> 1) & 1);
  *cfg = c;
  *splits = sp;
  return VP3D_OK;
}

int vp3d_tconv_nt_s16(vp3d_stream_t stream, const vp3d_rowmap* map, const void* x, int32_t ldx, int32_t c_in,
                      const void* wt, int32_t ldw, int32_t c_out, float* y, int64_t y_bpitch, int32_t ldy,
                      const vp3d_epilogue* epi, const float* zeros, const vp3d_s16* o) {
  int rc = check_map(map, "tconv_nt_s16");
  if (rc) return rc;
  VP3D_REQUIRE(x && wt && zeros && o, "tconv_nt_s16: null pointer");
  VP3D_REQUIRE(y || (o->raw_partials && o->ws) || (o->no_output && epi && epi->stat_sum), "tconv_nt_s16: no output buffer");
  VP3D_REQUIRE(c_in > 0 && c_out > 0 && ldx >= 1 && ldw >= map->taps * c_in && (o->raw_partials || ldy >= c_out),
               "tconv_nt_s16: bad sizes (c_in=%d c_out=%d ldx=%d ldw=%d ldy=%d)", c_in, c_out, ldx, ldw, ldy);
  RowsGemmArgs a;
  a.A = (const float*)x;
  a.B = (const float*)wt;
  a.zeros = zeros;
  a.M = map->batch * map->t_dst;
  a.N = c_out;
  a.K = map->taps * c_in;
  a.lda = ldx;
  a.c_src = c_in;
  a.ldb = ldw;
  a.b_tap_stride = 0;
  a.t_dst = map->t_dst;
  a.t_src = map->t_src;
  a.t_stride = map->t_stride;
  a.tap_step = map->tap_step;
  a.t_off = map->t_off;
  a.taps = map->taps;
  a.m_tiles = a.n_tiles = 0;
  VP3D_REQUIRE(epi == nullptr || epi->act_bwd == nullptr, "tconv_nt_s16: act_bwd is not supported by the split-fp16 GEMM");
  VP3D_REQUIRE(!o->raw_partials || epi == nullptr, "tconv_nt_s16: raw partial output takes no epilogue");
  if (o->raw_partials) {
    // the partial matrices [splits][M][N] in ws are the result (splits == 1: a single plain matrix)
    VP3D_REQUIRE(o->ws && o->splits >= 1 && o->ws_floats >= (int64_t)o->splits * a.M * a.N,
                 "tconv_nt_s16: raw partial output needs ws of splits*M*N floats and an explicit split count");
    rc = fill_epi(&a.epi, nullptr, o->ws, 0, a.N, c_out);
  } else {
    rc = fill_epi(&a.epi, epi, y, y_bpitch, ldy, c_out);
  }
  if (rc) return rc;
  a.epi.bound_a = o->x_bound;
  a.epi.bound_b = o->w_bound;
  a.epi.amax_out = o->amax_out;
  VP3D_REQUIRE(o->stat_slab_rows == 0 || o->stat_slab_rows == 32 || o->stat_slab_rows == 64, "tconv_nt_s16: stat_slab_rows=%d",
               o->stat_slab_rows);
  a.stat_slab_rows = o->stat_slab_rows ? o->stat_slab_rows : 64;
  if (o->res_s16) {
    VP3D_REQUIRE(a.epi.R != nullptr && o->res_bound != nullptr && a.epi.r_ld % 8 == 0 && a.epi.r_col0 % 8 == 0 && c_out % 8 == 0,
                 "tconv_nt_s16: an S16 residual needs its bound and 8-element aligned rows");
    a.epi.r_s16 = 1;
    a.epi.r_bound = o->res_bound;
  }
  if (o->out_s16) {
    VP3D_REQUIRE(!o->raw_partials && o->in_amax && o->l1 && o->out_wbound && c_out % 8 == 0 && ldy % 8 == 0 && y_bpitch % 8 == 0 &&
                     aligned16(y) && (epi == nullptr || epi->stat_sum == nullptr) && (o->splits == 1 || o->splits == 0),
                 "tconv_nt_s16: S16 output needs in_amax / l1 / out_wbound, 8-element aligned rows, no statistics, no split-K");
    VP3D_REQUIRE(a.epi.R == nullptr || o->res_s16, "tconv_nt_s16: S16 output takes an S16 residual");
    a.epi.c_s16 = 1;
    a.epi.in_amax = o->in_amax;
    a.epi.l1 = o->l1;
    a.epi.res_amax = o->res_amax;
    a.epi.out_wbound = o->out_wbound;
  }
  if (o->no_output) {
    VP3D_REQUIRE(!o->raw_partials && !o->out_s16 && a.epi.stat_sum != nullptr && o->act_scale == nullptr,
                 "tconv_nt_s16: a statistics-only launch needs the statistics epilogue and nothing else");
    a.epi.no_out = 1;
  }
  if (o->act_scale != nullptr) {
    VP3D_REQUIRE(!o->raw_partials && !o->out_s16 && o->act_shift && o->act_bound && y && c_out % 64 == 0 && ldy == c_out &&
                     y_bpitch == (int64_t)map->t_dst * c_out && aligned16(y) && a.epi.R == nullptr && a.epi.bias == nullptr &&
                     a.epi.stat_sum == nullptr && !a.epi.relu,
                 "tconv_nt_s16: the fused activation writes contiguous S16 rows [M][c_out] (c_out %% 64 == 0) and takes no "
                 "other epilogue");
    if (o->act_drop) VP3D_REQUIRE(o->act_drop->p >= 0.f && o->act_drop->p < 1.f, "tconv_nt_s16: dropout p=%f", o->act_drop->p);
    a.epi.act_scale = o->act_scale;
    a.epi.act_shift = o->act_shift;
    a.epi.act_bound = o->act_bound;
    a.epi.act_bits = o->act_bits;
    a.epi.ab_drop = make_drop(o->act_drop);
  }
  if (o->red != nullptr) {
    const vp3d_s16_red* r = o->red;
    VP3D_REQUIRE(!o->raw_partials && !o->out_s16 && !o->no_output && o->act_scale == nullptr && y && o->amax_out &&
                     a.epi.bias == nullptr && !a.epi.relu && a.epi.stat_sum == nullptr && !a.epi.r_s16,
                 "tconv_nt_s16: the fused BatchNorm-backward sums ride on a plain fp32 dgrad launch (amax_out set, no bias / "
                 "relu / statistics / S16 output)");
    VP3D_REQUIRE(r->y_up && r->mean && r->invstd && r->scale && r->act_bits && r->partials && r->tickets && r->dgamma && r->dbeta &&
                     r->dy_bound && aligned16(r->y_up) && aligned16(r->mean) && aligned16(r->invstd) && aligned16(r->partials),
                 "tconv_nt_s16: red has a null or unaligned pointer");
    VP3D_REQUIRE(r->c_up > 0 && r->c_up % 256 == 0 && c_out % r->c_up == 0 && y_bpitch % r->c_up == 0 && ldy % r->c_up == 0 &&
                     r->rows_up > 0 && r->rows_up < ((int64_t)1 << 31) && r->p >= 0.f && r->p < 1.f,
                 "tconv_nt_s16: red needs c_up %% 256 == 0, c_out / y_bpitch / ldy multiples of c_up, rows_up < 2^31 (c_up=%d "
                 "c_out=%d ldy=%d)", r->c_up, c_out, ldy);
    VP3D_REQUIRE((int64_t)(map->batch - 1) * (y_bpitch / r->c_up) + (int64_t)(map->t_dst - 1) * (ldy / r->c_up) + c_out / r->c_up <=
                     r->rows_up, "tconv_nt_s16: red: the output rows do not fit the upstream activation's %lld rows",
                 (long long)r->rows_up);
    VP3D_REQUIRE(r->partials_floats >= ((a.M + 127) / 128) * 2 * (int64_t)c_out,
                 "tconv_nt_s16: red partials need ceil(M/128)*2*c_out floats");
    VP3D_REQUIRE(r->rows_up * (r->c_up / 8) < ((int64_t)1 << 31), "tconv_nt_s16: red: more than 2 GiB of activation bits");
    a.epi.red = 1;
    a.epi.ab_y = r->y_up;
    a.epi.ab_mean = r->mean;
    a.epi.ab_invstd = r->invstd;
    a.epi.ab_scale = r->scale;
    a.epi.ab_part = r->partials;
    a.epi.ab_c = r->c_up;
    a.epi.red_bits = r->act_bits;
    a.epi.red_m = (int32_t)r->rows_up;
    a.epi.red_row_b = (int32_t)(y_bpitch / r->c_up);
    a.epi.red_row_t = ldy / r->c_up;
    a.epi.red_inv_keep = 1.0f / (1.0f - r->p);
    a.epi.red_inv_m = 1.0f / (float)r->rows_up;
    a.epi.red_sqrt_m1 = sqrtf((float)(r->rows_up > 1 ? r->rows_up - 1 : 1));
    a.epi.red_cnt = r->tickets;
    a.epi.red_dgamma = r->dgamma;
    a.epi.red_dbeta = r->dbeta;
    a.epi.red_dy_bound = r->dy_bound;
  }
  if (o->fin != nullptr) {
    const vp3d_s16_fin* f = o->fin;
    VP3D_REQUIRE(a.epi.stat_sum != nullptr && !o->raw_partials && !o->out_s16 && !o->no_output && o->act_scale == nullptr &&
                     o->red == nullptr, "tconv_nt_s16: fin rides on a launch that writes BatchNorm statistics and an fp32 output");
    VP3D_REQUIRE(f->gamma && f->beta && f->scale && f->shift && f->save_mean && f->save_invstd && f->tickets,
                 "tconv_nt_s16: fin has a null pointer");
    a.epi.fin_tickets = f->tickets;
    a.epi.fin_gamma = f->gamma;
    a.epi.fin_beta = f->beta;
    a.epi.fin_eps = f->eps;
    a.epi.fin_momentum = f->momentum;
    a.epi.fin_momentum_dev = f->momentum_dev;
    a.epi.fin_running_mean = f->running_mean;
    a.epi.fin_running_var = f->running_var;
    a.epi.fin_nbt = f->num_batches_tracked;
    a.epi.fin_scale = f->scale;
    a.epi.fin_shift = f->shift;
    a.epi.fin_save_mean = f->save_mean;
    a.epi.fin_save_invstd = f->save_invstd;
  }
  set_splits(&a, nullptr, 0);
  const bool single = o->out_s16 || o->res_s16 || o->no_output || o->act_scale != nullptr || o->red != nullptr;
  return launch_nt_s16((hipStream_t)stream, a, o->cfg, single ? 1 : o->splits, o->ws, o->ws_floats,
                       o->raw_partials != 0,    // the finishing pass of a split launch knows neither S16 residuals nor S16 output
                       o->tickets);
}

int vp3d_expand_fwd_s16(vp3d_stream_t stream, int64_t M, int32_t N, int32_t kpad, const void* x, const float* x_bound,
                        const void* w, const float* w_bound, float* stat_sum, float* stat_m2, const float* scale,
                        const float* shift, const vp3d_dropout* drop, const float* out_bound, void* out, uint8_t* act_bits) {
  VP3D_REQUIRE(M > 0 && M < ((int64_t)1 << 31) && N > 0 && N % 8 == 0 && kpad >= 32 && kpad <= 128 && kpad % 32 == 0 && x && w &&
                   x_bound && w_bound && aligned16(x) && aligned16(w) && M * kpad * 4 < ((int64_t)1 << 31),
               "expand_fwd_s16: bad argument (16-byte aligned S16 rows of 32..128 columns, N %% 8 == 0, X below 2 GiB)");
  const bool stats = stat_sum != nullptr, act = out != nullptr;
  VP3D_REQUIRE(stats != act, "expand_fwd_s16: exactly one of the statistics pass (stat_sum, stat_m2) and the activation pass (out)");
  if (stats) VP3D_REQUIRE(stat_m2 != nullptr, "expand_fwd_s16: stat_m2 is NULL");
  if (act) {
    VP3D_REQUIRE(scale && shift && out_bound && aligned16(out), "expand_fwd_s16: the activation pass needs scale, shift, out_bound");
    if (drop) VP3D_REQUIRE(drop->p >= 0.f && drop->p < 1.f, "expand_fwd_s16: dropout p=%f", drop->p);
  }
  return launch_expand_fwd_s16((hipStream_t)stream, M, N, kpad, (const float*)x, x_bound, (const float*)w, w_bound, stat_sum,
                               stat_m2, scale, shift, make_drop(act ? drop : nullptr), out_bound, (float*)out, act_bits);
}

int vp3d_expand_bwd_p_s16(vp3d_stream_t stream, int64_t M, int32_t C, int32_t kpad, const float* go, const float* go_bound,
                          const uint8_t* act_bits, float p, const void* x_t, int64_t ld_t, const float* x_bound, float* partials,
                          float* gram_partials, int32_t* nparts) {
  VP3D_REQUIRE(M > 0 && C > 0 && C % 64 == 0 && kpad >= 32 && kpad <= 128 && kpad % 32 == 0 && nparts && p >= 0.f && p < 1.f &&
                   M * C * 4 < ((int64_t)1 << 31) && ld_t >= M && ld_t % 64 == 0 && kpad * ld_t * 4 < ((int64_t)1 << 31),
               "expand_bwd_p_s16: bad argument (C %% 64 == 0, kpad 32..128, ld_t %% 64 == 0, go below 2 GiB)");
  *nparts = expand_bwd_groups(M, C);
  if (partials == nullptr) return VP3D_OK;              // size query: partials = [*nparts][C][kpad] floats
  VP3D_REQUIRE(go && go_bound && act_bits && x_t && x_bound && aligned16(x_t) && aligned16(partials),
               "expand_bwd_p_s16: null or unaligned pointer");
  return launch_expand_bwd_p_s16((hipStream_t)stream, M, C, kpad, go, go_bound, act_bits, p, (const float*)x_t, ld_t, x_bound,
                                 *nparts, partials, gram_partials);
}



int vp3d_has_experiments(void) { return nt_s16_has_experiments(); }

int vp3d_nt_s16_workspace(int64_t M, int32_t N, int32_t K, int32_t cfg, int32_t splits, int32_t raw_partials, int64_t* ws_floats,
                          int32_t* tickets) {
  VP3D_REQUIRE(M > 0 && M < ((int64_t)1 << 31) && N > 0 && K > 0 && ws_floats && tickets, "nt_s16_workspace: bad argument");
  nt_s16_workspace((int)M, N, K, cfg, splits, raw_partials, ws_floats, tickets);
  return VP3D_OK;
}

int vp3d_split_rows(vp3d_stream_t stream, int64_t M, int32_t C, const float* src, int64_t ld_src, void* dst,
                    int64_t ld_dst, const float* bound) {
  VP3D_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && src && dst && ld_src >= C && ld_dst >= C && ld_src % 4 == 0 && ld_dst % 8 == 0 &&
                   aligned16(src) && aligned16(dst),
               "split_rows: needs C %% 8 == 0 and 16-byte aligned rows");
  return launch_split_rows((hipStream_t)stream, M, C, src, ld_src, (float*)dst, ld_dst, bound);
}

int vp3d_amax(vp3d_stream_t stream, int64_t n, const float* src, float* bound) {
  VP3D_REQUIRE(n > 0 && src && bound, "amax: bad argument");
  return launch_amax((hipStream_t)stream, n, src, bound);
}

int vp3d_amax_floor(vp3d_stream_t stream, int64_t n, const float* src, float floor, float* bound) {
  VP3D_REQUIRE(n > 0 && src && bound && floor >= 0.f, "amax_floor: bad argument");
  return launch_amax((hipStream_t)stream, n, src, bound, floor);
}

int vp3d_wgrad_rows_s16(vp3d_stream_t stream, int64_t M, const void* dy, int64_t ld_dy, int32_t c_out,
                        const float* dy_bound, const void* x, int64_t ld_x, int32_t taps, int32_t c_in,
                        const float* x_bound, int32_t splits, float* partials) {
  return launch_wgrad_rows_s16((hipStream_t)stream, M, (const float*)dy, ld_dy, c_out, dy_bound, (const float*)x, ld_x, taps,
                               c_in, x_bound, splits, partials);
}

int vp3d_tconv_dgrad(vp3d_stream_t stream, const vp3d_rowmap* map, const float* dy, int32_t lddy, int32_t c_out,
                     const float* wt, int32_t ldw, int32_t w_tap_stride, int32_t n_out, float* dx,
                     int64_t dx_bpitch, int32_t lddx, const vp3d_epilogue* epi, const float* zeros, float* splitk_ws,
                     int64_t splitk_ws_floats) {
  int rc = check_map(map, "tconv_dgrad");
  if (rc) return rc;
  const bool g_only = epi != nullptr && epi->act_bwd != nullptr && epi->act_bwd->store_v == 0;
  VP3D_REQUIRE(dy && wt && (dx || g_only) && zeros, "tconv_dgrad: null pointer");
  VP3D_REQUIRE(c_out > 0 && n_out > 0 && lddy >= c_out && ldw >= n_out && lddx >= n_out,
               "tconv_dgrad: bad sizes (c_out=%d n_out=%d lddy=%d ldw=%d lddx=%d)", c_out, n_out, lddy, ldw, lddx);
  RowsGemmArgs a;
  a.A = dy;
  a.B = wt;
  a.zeros = zeros;
  a.M = map->batch * map->t_dst;
  a.N = n_out;
  a.K = map->taps * c_out;
  a.lda = lddy;
  a.c_src = c_out;
  a.ldb = ldw;
  a.b_tap_stride = w_tap_stride;
  a.t_dst = map->t_dst;
  a.t_src = map->t_src;
  a.t_stride = map->t_stride;
  a.tap_step = map->tap_step;
  a.t_off = map->t_off;
  a.taps = map->taps;
  a.m_tiles = (a.M + 127) / 128;
  a.n_tiles = (a.N + 127) / 128;
  rc = fill_epi(&a.epi, epi, dx, dx_bpitch, lddx, n_out);
  if (rc) return rc;
  set_splits(&a, splitk_ws, splitk_ws_floats);
  return launch_rows_gemm((hipStream_t)stream, a, /*b_kcontig=*/false);
}

int vp3d_tconv_wgrad(vp3d_stream_t stream, const vp3d_rowmap* map, const float* dy, int32_t lddy, int32_t c_out,
                     const float* x, int32_t ldx, int32_t c_in, float* partials, int32_t splits,
                     const float* zeros) {
  int rc = check_map(map, "tconv_wgrad");
  if (rc) return rc;
  VP3D_REQUIRE(dy && x && partials && zeros, "tconv_wgrad: null pointer");
  VP3D_REQUIRE(c_out > 0 && c_in > 0 && lddy >= c_out && ldx >= 1 && splits >= 1, "tconv_wgrad: bad sizes");
  RedGemmArgs a;
  a.G = dy;
  a.X = x;
  a.zeros = zeros;
  a.C = partials;
  a.Mred = map->batch * map->t_dst;
  a.Mo = c_out;
  a.N = map->taps * c_in;
  a.ldg = lddy;
  a.ldx = ldx;
  a.c_x = c_in;
  a.t_dst = map->t_dst;
  a.t_src = map->t_src;
  a.t_stride = map->t_stride;
  a.tap_step = map->tap_step;
  a.t_off = map->t_off;
  a.m_tiles = (a.Mo + 127) / 128;
  a.n_tiles = (a.N + 127) / 128;
  const int nkt = (a.Mred + 31) / 32;
  VP3D_REQUIRE(splits <= nkt, "tconv_wgrad: splits=%d exceeds the %d reduction tiles", splits, nkt);
  a.splits = splits;
  a.kt_per_split = (nkt + splits - 1) / splits;
  return launch_red_gemm((hipStream_t)stream, a);
}

}  // extern "C"
