// The callers either side of the temporal stack inside one training / evaluation step:
//   batch assembly   reference common/generators.py:105-149 (ChunkedGenerator.next_epoch body) and :216-239
//                    (UnchunkedGenerator: edge padding + mirrored copy)
//   loss             reference common/loss.py:11-17 (mpjpe), :19-25 (weighted_mpjpe): forward + gradient in one pass
//   TTA fold         reference run.py:677-680 (un-flip the mirrored prediction and average)
//   optimizer        reference run.py:252,264,420 (torch.optim.Adam(..., amsgrad=True).step()) on flat buffers
// All of it is HBM / latency bound index-and-byte work: coalesced rows, no MFMA.
#include "vp3d_internal.h"

namespace vp3d {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// Batch assembly.  One workgroup row per (chunk, frame): out[i][f][j][c] = src[seq_off + clamp(start + f)][perm j][c]
// with x (c == 0) negated for mirrored chunks.  `edge` padding of np.pad == clamping the frame index.
// ---------------------------------------------------------------------------------------------------------
struct GatherArgs {
  const int32_t* chunks;      // [n][3] = (seq, start_3d, flip)
  const int64_t* seq_off;     // [n_seq + 1] frame offsets into the concatenated pose arrays
  const float* p2;
  const float* p3;
  const float* cams;
  const int32_t* perm2;       // mirrored copy: out joint j reads joint perm2[j]   (NULL: identity)
  const int32_t* perm3;
  float* o2;
  float* o3;
  float* ocam;
  int32_t n, j2, f2, j3, f3, cam_dim, chunk_len, pad, shift, len2;   // len2 = chunk_len + 2*pad
};

__global__ void __launch_bounds__(256) k_gather_chunks(const GatherArgs a) {
  const int row2 = a.j2 * a.f2;                       // floats per 2D frame (34)
  const int row3 = a.j3 * a.f3;                       // floats per 3D frame (51)
  const int64_t n2 = (int64_t)a.n * a.len2 * row2;
  const int64_t n3 = a.p3 != nullptr ? (int64_t)a.n * a.chunk_len * row3 : 0;
  const int64_t nc = a.cams != nullptr ? (int64_t)a.n * a.cam_dim : 0;
  const int64_t total = n2 + n3 + nc;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < n2) {
      const int64_t fr = e / row2;                    // (chunk, frame)
      const int r = (int)(e - fr * row2);
      const int i = (int)(fr / a.len2), f = (int)(fr - (int64_t)i * a.len2);
      const int seq = a.chunks[i * 3], start = a.chunks[i * 3 + 1], flip = a.chunks[i * 3 + 2];
      const int64_t s0 = a.seq_off[seq];
      const int len = (int)(a.seq_off[seq + 1] - s0);
      int t = start - a.pad - a.shift + f;            // generators.py:107-108
      t = t < 0 ? 0 : (t >= len ? len - 1 : t);       // :113-118 'edge'
      const int j = r / a.f2, c = r - j * a.f2;
      const int js = (flip && a.perm2 != nullptr) ? a.perm2[j] : j;          // :122-123
      float v = a.p2[(s0 + t) * row2 + js * a.f2 + c];
      if (flip && c == 0) v = -v;                     // :121
      a.o2[e] = v;
    } else if (e < n2 + n3) {
      const int64_t e3 = e - n2;
      const int64_t fr = e3 / row3;
      const int r = (int)(e3 - fr * row3);
      const int i = (int)(fr / a.chunk_len), f = (int)(fr - (int64_t)i * a.chunk_len);
      const int seq = a.chunks[i * 3], start = a.chunks[i * 3 + 1], flip = a.chunks[i * 3 + 2];
      const int64_t s0 = a.seq_off[seq];
      const int len = (int)(a.seq_off[seq + 1] - s0);
      int t = start + f;                              // :126-135
      t = t < 0 ? 0 : (t >= len ? len - 1 : t);
      const int j = r / a.f3, c = r - j * a.f3;
      const int js = (flip && a.perm3 != nullptr) ? a.perm3[j] : j;          // :139-141
      float v = a.p3[(s0 + t) * row3 + js * a.f3 + c];
      if (flip && c == 0) v = -v;                     // :138
      a.o3[e3] = v;
    } else {
      const int64_t ec = e - n2 - n3;
      const int i = (int)(ec / a.cam_dim), c = (int)(ec - (int64_t)i * a.cam_dim);
      const int seq = a.chunks[i * 3], flip = a.chunks[i * 3 + 2];
      float v = a.cams[(int64_t)seq * a.cam_dim + c];
      if (flip && (c == 2 || c == 7)) v = -v;         // :146-149
      a.ocam[ec] = v;
    }
  }
}

// out[t][j][c] = 0.5 * (pred[0][t][j][c] + s * pred[1][t][perm j][c]),  s = -1 for c == 0     (run.py:677-680)
__global__ void __launch_bounds__(256) k_tta_fold(int64_t T, int J, int D, const float* __restrict__ pred,
                                                  const int32_t* __restrict__ perm, float* __restrict__ out) {
  const int64_t total = T * J * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tj = e / D;
    const int c = (int)(e - tj * D);
    const int64_t t = tj / J;
    const int j = (int)(tj - t * J);
    const int js = perm != nullptr ? perm[j] : j;
    float m = pred[total + (t * J + js) * D + c];
    if (c == 0) m = -m;
    // torch.mean over the 2 copies = (a + b) / 2
    out[e] = (pred[e] + m) / 2.0f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// mpjpe / weighted_mpjpe: loss = mean_i w_i * ||p_i - t_i||_2 ; grad_i = w_i * (p_i - t_i) / (||.|| * n)
// Deterministic: per-thread serial sums, fixed-shape LDS tree, fp64 combine of the block partials.
// ---------------------------------------------------------------------------------------------------------
constexpr int LOSS_THREADS = 256;
constexpr int LOSS_MAX_BLOCKS = 1024;

// One point per thread and pass (a block of 256 threads, up to LOSS_MAX_BLOCKS blocks): the gradient of a point needs nothing
// but the point, so it is written at once; the value is the sum of the blocks' fp64 partials, folded IN BLOCK ORDER by the block
// that draws the last ticket (round 6: the single 1024-thread block of rounds 2-5 walked 17 points per thread through dependent
// scalar loads -- 15 us on the dependent chain between the head's forward and backward; this form takes ~5).
// ws: [0] = ticket (int32 in the first 8 bytes: zero on entry, zero again on exit), [1 ..] = the partials.
template <int DIM>
__global__ void __launch_bounds__(LOSS_THREADS) k_mpjpe(int64_t n, int dim, const float* __restrict__ p,
                                                        const float* __restrict__ t, const float* __restrict__ w,
                                                        float inv_n, float* __restrict__ grad, double* __restrict__ ws,
                                                        float* __restrict__ loss) {
  __shared__ double red[LOSS_THREADS / 64];
  __shared__ int last_flag;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = DIM > 0 ? DIM : dim;
    float df[DIM > 0 ? DIM : 8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < d; ++c) {
      df[c] = p[i * d + c] - t[i * d + c];
      s += df[c] * df[c];
    }
    const float nrm = sqrtf(s);
    const float wi = w != nullptr ? w[i] : 1.f;
    acc += (double)(wi * nrm);
    if (grad != nullptr) {
      const float k = nrm > 0.f ? wi * inv_n / nrm : 0.f;      // torch: zero sub-gradient at the origin
#pragma unroll
      for (int c = 0; c < d; ++c) grad[i * d + c] = df[c] * k;
    }
  }
  // fixed-shape reduction: xor butterfly inside each wave, then the wave sums in order
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < LOSS_THREADS / 64; ++i) tot += red[i];
    if (gridDim.x == 1) {
      loss[0] = (float)(tot * (double)inv_n);
      last_flag = 0;
    } else {
      // publish the partial past this XCD's L2, draw the ticket; the last arriver acquires and folds in block order
      __hip_atomic_store(ws + 1 + blockIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int* ticket = reinterpret_cast<int*>(ws);
      const int tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      last_flag = tk == (int)gridDim.x - 1 ? 1 : 0;
      if (last_flag) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double sum = 0.0;
        for (int i = 0; i < (int)gridDim.x; ++i)
          sum += __hip_atomic_load(ws + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        loss[0] = (float)(sum * (double)inv_n);
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero for the next launch
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Adam / AMSGrad on flat fp32 buffers (torch.optim.Adam single-tensor arithmetic, torch/optim/adam.py):
//   g += wd * p ; m = lerp(m, g, 1-b1) ; v = v*b2 + (1-b2)*g*g ; vmax = max(vmax, v)
//   p -= step_size * m / (sqrt(vmax or v) / sqrt(bc2) + eps)
// One pass: 16 (20 with amsgrad) B read + 12 (16) B written per parameter.
// ---------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)   // no FMA contraction below: the update is reproducible op-for-op against a CPU restatement
struct AdamK {
  float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, wd;
};

template <bool AMS>
__global__ void __launch_bounds__(256) k_adam(int64_t n4, int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, float* __restrict__ vmax,
                                              const AdamK k) {
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = q * 4;
    if (e + 4 <= n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + e), gv = *reinterpret_cast<const f32x4*>(g + e);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + e), vv = *reinterpret_cast<f32x4*>(v + e), xv;
      if (AMS) xv = *reinterpret_cast<f32x4*>(vmax + e);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float gi = gv[c];
        if (k.wd != 0.f) gi = gi + k.wd * pv[c];
        mv[c] = mv[c] + k.one_minus_b1 * (gi - mv[c]);
        vv[c] = vv[c] * k.b2 + k.one_minus_b2 * gi * gi;
        float d;
        if (AMS) {
          xv[c] = fmaxf(xv[c], vv[c]);
          d = sqrtf(xv[c]) / k.bc2_sqrt + k.eps;
        } else {
          d = sqrtf(vv[c]) / k.bc2_sqrt + k.eps;
        }
        pv[c] = pv[c] - k.step_size * (mv[c] / d);
      }
      *reinterpret_cast<f32x4*>(p + e) = pv;
      *reinterpret_cast<f32x4*>(m + e) = mv;
      *reinterpret_cast<f32x4*>(v + e) = vv;
      if (AMS) *reinterpret_cast<f32x4*>(vmax + e) = xv;
    } else {
      for (int64_t i = e; i < n; ++i) {
        float gi = g[i];
        if (k.wd != 0.f) gi = gi + k.wd * p[i];
        const float mi = m[i] + k.one_minus_b1 * (gi - m[i]);
        const float vi = v[i] * k.b2 + k.one_minus_b2 * gi * gi;
        float d;
        if (AMS) {
          const float xi = fmaxf(vmax[i], vi);
          vmax[i] = xi;
          d = sqrtf(xi) / k.bc2_sqrt + k.eps;
        } else {
          d = sqrtf(vi) / k.bc2_sqrt + k.eps;
        }
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - k.step_size * (mi / d);
      }
    }
  }
}

inline int grid_for(int64_t items, int threads, int cap) {
  int64_t b = (items + threads - 1) / threads;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace
}  // namespace vp3d

using namespace vp3d;

extern "C" {

int vp3d_gather_chunks(vp3d_stream_t stream, const vp3d_gather* g) {
  VP3D_REQUIRE(g != nullptr, "gather_chunks: null descriptor");
  VP3D_REQUIRE(g->n_chunks > 0 && g->chunks && g->seq_off && g->poses_2d && g->out_2d,
               "gather_chunks: null pointer / no chunks");
  VP3D_REQUIRE(g->j2 > 0 && g->f2 > 0 && g->chunk_length > 0 && g->pad >= 0, "gather_chunks: bad 2D shape");
  VP3D_REQUIRE((g->poses_3d == nullptr) == (g->out_3d == nullptr), "gather_chunks: poses_3d and out_3d go together");
  VP3D_REQUIRE((g->cameras == nullptr) == (g->out_cam == nullptr), "gather_chunks: cameras and out_cam go together");
  if (g->poses_3d) VP3D_REQUIRE(g->j3 > 0 && g->f3 > 0, "gather_chunks: bad 3D shape");
  if (g->cameras) VP3D_REQUIRE(g->cam_dim > 0, "gather_chunks: bad camera width");
  GatherArgs a;
  a.chunks = g->chunks;
  a.seq_off = g->seq_off;
  a.p2 = g->poses_2d;
  a.p3 = g->poses_3d;
  a.cams = g->cameras;
  a.perm2 = g->kps_perm;
  a.perm3 = g->joints_perm;
  a.o2 = g->out_2d;
  a.o3 = g->out_3d;
  a.ocam = g->out_cam;
  a.n = g->n_chunks;
  a.j2 = g->j2;
  a.f2 = g->f2;
  a.j3 = g->j3;
  a.f3 = g->f3;
  a.cam_dim = g->cam_dim;
  a.chunk_len = g->chunk_length;
  a.pad = g->pad;
  a.shift = g->causal_shift;
  a.len2 = g->chunk_length + 2 * g->pad;
  const int64_t total = (int64_t)a.n * a.len2 * a.j2 * a.f2 + (a.p3 ? (int64_t)a.n * a.chunk_len * a.j3 * a.f3 : 0) +
                        (a.cams ? (int64_t)a.n * a.cam_dim : 0);
  VP3D_LAUNCH(k_gather_chunks, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("gather_chunks");
}

int vp3d_tta_fold(vp3d_stream_t stream, int64_t n_frames, int32_t n_joints, int32_t dim, const float* pred,
                  const int32_t* joints_perm, float* out) {
  VP3D_REQUIRE(n_frames > 0 && n_joints > 0 && dim > 0 && pred && out, "tta_fold: bad argument");
  VP3D_LAUNCH(k_tta_fold, dim3(grid_for(n_frames * n_joints * dim, 256, 256 * 8)), dim3(256), 0,
                     (hipStream_t)stream, n_frames, n_joints, dim, pred, joints_perm, out);
  return check_launch("tta_fold");
}

static int mpjpe_blocks(int64_t n_pts) { return grid_for(n_pts, LOSS_THREADS, LOSS_MAX_BLOCKS); }

int64_t vp3d_mpjpe_ws_bytes(int64_t n_pts) {
  const int blocks = mpjpe_blocks(n_pts);
  return blocks <= 1 ? 0 : (int64_t)(1 + blocks) * (int64_t)sizeof(double);
}

int vp3d_mpjpe(vp3d_stream_t stream, int64_t n_pts, int32_t dim, const float* pred, const float* target,
               const float* w, float* loss, float* grad, void* ws) {
  VP3D_REQUIRE(n_pts > 0 && dim > 0 && dim <= 8 && pred && target && loss, "mpjpe: bad argument (n=%lld dim=%d)",
               (long long)n_pts, dim);
  const int blocks = mpjpe_blocks(n_pts);
  VP3D_REQUIRE(blocks == 1 || (ws != nullptr && (reinterpret_cast<uintptr_t>(ws) & 7u) == 0),
               "mpjpe: %lld points need an 8-byte aligned workspace of vp3d_mpjpe_ws_bytes() whose first 8 bytes are zero",
               (long long)n_pts);
  const float inv_n = (float)(1.0 / (double)n_pts);
  double* part = reinterpret_cast<double*>(ws);
  if (dim == 3)
    VP3D_LAUNCH((k_mpjpe<3>), dim3(blocks), dim3(LOSS_THREADS), 0, (hipStream_t)stream, n_pts, dim, pred, target,
                       w, inv_n, grad, part, loss);
  else if (dim == 2)
    VP3D_LAUNCH((k_mpjpe<2>), dim3(blocks), dim3(LOSS_THREADS), 0, (hipStream_t)stream, n_pts, dim, pred, target,
                       w, inv_n, grad, part, loss);
  else
    VP3D_LAUNCH((k_mpjpe<0>), dim3(blocks), dim3(LOSS_THREADS), 0, (hipStream_t)stream, n_pts, dim, pred, target,
                       w, inv_n, grad, part, loss);
  return check_launch("mpjpe");
}

int vp3d_adam_step(vp3d_stream_t stream, int64_t n, float* param, const float* grad, float* exp_avg,
                   float* exp_avg_sq, float* max_exp_avg_sq, const vp3d_adam* h) {
  VP3D_REQUIRE(n > 0 && param && grad && exp_avg && exp_avg_sq && h, "adam_step: bad argument");
  VP3D_REQUIRE(h->step >= 1, "adam_step: step must be the 1-based count of this update (got %lld)", (long long)h->step);
  VP3D_REQUIRE(h->beta1 >= 0.0 && h->beta1 < 1.0 && h->beta2 >= 0.0 && h->beta2 < 1.0 && h->eps >= 0.0,
               "adam_step: bad hyper-parameters");
  VP3D_REQUIRE(!h->amsgrad || max_exp_avg_sq, "adam_step: amsgrad needs max_exp_avg_sq");
  VP3D_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq) &&
                   (!h->amsgrad || aligned16(max_exp_avg_sq)), "adam_step: buffers must be 16-byte aligned");
  // torch/optim/adam.py (_single_tensor_adam): python-float (double) scalars, applied to fp32 tensors
  const double bc1 = 1.0 - pow(h->beta1, (double)h->step);
  const double bc2 = 1.0 - pow(h->beta2, (double)h->step);
  AdamK k;
  k.one_minus_b1 = (float)(1.0 - h->beta1);
  k.b2 = (float)h->beta2;
  k.one_minus_b2 = (float)(1.0 - h->beta2);
  k.step_size = (float)(h->lr / bc1);
  k.bc2_sqrt = (float)sqrt(bc2);
  k.eps = (float)h->eps;
  k.wd = (float)h->weight_decay;
  const int64_t n4 = (n + 3) / 4;
  const int blocks = grid_for(n4, 256, 256 * 8);
  if (h->amsgrad)
    VP3D_LAUNCH((k_adam<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, n4, n, param, grad, exp_avg,
                       exp_avg_sq, max_exp_avg_sq, k);
  else
    VP3D_LAUNCH((k_adam<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, n4, n, param, grad, exp_avg,
                       exp_avg_sq, max_exp_avg_sq, k);
  return check_launch("adam_step");
}

}  // extern "C"
