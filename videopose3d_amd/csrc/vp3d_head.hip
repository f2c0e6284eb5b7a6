// The head of the temporal model at small row counts: the 3*J_out-column shrink conv (reference common/model.py:33
// `self.shrink = nn.Conv1d(channels, num_joints_out*3, 1)`, applied at model.py:137 / :196) and its whole backward.
//
// M = B * T_out rows (1024 in training: T_out = 1), K = channels (1024), N = 3 * J_out (51): 0.03 % of the step's FLOPs.  On
// the general GEMM kernels this was a dependent chain of small launches -- forward: 128 x 128 MFMA tiles of which 60 % are
// padding, K-sliced to fill the GPU, + the finishing pass (15.7 + 10.9 us); backward: dgrad (20.0) + amax of its result (5.0) on
// the critical path, column sums + weight gradient + slice reduction (5.0 + 25.9 + 4.7) beside it (profiles/r05_step_timeline.txt).
// Here: plain fp32 FMAs (exact products, fp32 accumulation like the fp32-MFMA path; the work is bound by launch latency
// and the L2, not by arithmetic), the reference's own weight layout W[n][k] (a 1-tap conv weight needs no pack), ONE launch
// forward (bias included) and ONE launch backward (dh + its maximum + the row-sliced partials of dW / dbias) + a fold launch
// that nothing on the critical path waits for.  Deterministic: every sum has a fixed order (no float atomics).
#include "vp3d_internal.h"
#include "vp3d_s16.h"

namespace vp3d {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HEAD_RT = 4;        // rows per block: forward and dh (M = 1024 -> 256 blocks; every block streams W once from the L2)
constexpr int HEAD_RS = 32;       // rows per slice of the weight-gradient partials (one h value per row in registers)
constexpr int HEAD_MAX_N = 128;   // columns (3 * J_out) at most: LDS staging of a slice's dy rows
constexpr int HEAD_MAX_K = 4096;  // channels at most: LDS staging of a block's h rows (64 KiB)

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b, float acc) {
  acc = fmaf(a[0], b[0], acc);
  acc = fmaf(a[1], b[1], acc);
  acc = fmaf(a[2], b[2], acc);
  return fmaf(a[3], b[3], acc);
}

// out[m][n] = bias[n] + sum_k h[m][k] * w[n][k].  Block = HEAD_RT rows staged in LDS; wave q of 4 owns columns n = q, q + 4, ...:
// lanes split K in float4 pieces (coalesced 1-KiB reads of a weight row), HEAD_RT partial sums per lane, butterfly over the wave.
__global__ void __launch_bounds__(256) k_head_fwd(int M, int K, int N, const float* __restrict__ h, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float hs[];       // [HEAD_RT][K]
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int m0 = blockIdx.x * HEAD_RT, rows = min(HEAD_RT, M - m0);
  const int k4n = K >> 2;
  const f32x4* h4 = reinterpret_cast<const f32x4*>(h);
  f32x4* hs4 = reinterpret_cast<f32x4*>(hs);
  for (int i = tid; i < HEAD_RT * k4n; i += 256) {
    const int r = i / k4n, c = i - r * k4n;
    hs4[i] = r < rows ? h4[(int64_t)(m0 + r) * k4n + c] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
  // ALL columns of the wave (HEAD_NC = 13 for N <= 52; a second round beyond) advance together along K: per 256-float4 chunk of K
  // one batch of HEAD_NC weight loads, the next chunk's batch already in flight (register double buffer) -- one column or four
  // at a time left 13 / 4 dependent L2 round trips + as many butterfly phases in a row (18-20 us for the launch); one butterfly
  // phase over the HEAD_NC x HEAD_RT partial sums at the end
  constexpr int HEAD_NC = 13;
  for (int nbase = 0; nbase < N; nbase += 4 * HEAD_NC) {
    float acc[HEAD_NC][HEAD_RT];
#pragma unroll
    for (int j = 0; j < HEAD_NC; ++j)
#pragma unroll
      for (int r = 0; r < HEAD_RT; ++r) acc[j][r] = 0.f;
    f32x4 wv[2][HEAD_NC];
    auto fetch = [&](int c, int buf) {
#pragma unroll
      for (int j = 0; j < HEAD_NC; ++j) {
        const int n = nbase + q + 4 * j;
        wv[buf][j] = (n < N && c < k4n) ? w4[(int64_t)n * k4n + c] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    fetch(lane, 0);
    int it = 0;
    for (int c = lane; c < k4n; c += 64, ++it) {
      // (two explicit copies of the body: the buffer index must be a compile-time constant for wv to stay in registers)
      if ((it & 1) == 0) {
        fetch(c + 64, 1);
        f32x4 hv[HEAD_RT];
#pragma unroll
        for (int r = 0; r < HEAD_RT; ++r) hv[r] = hs4[r * k4n + c];
#pragma unroll
        for (int j = 0; j < HEAD_NC; ++j)
#pragma unroll
          for (int r = 0; r < HEAD_RT; ++r) acc[j][r] = dot4(wv[0][j], hv[r], acc[j][r]);
      } else {
        fetch(c + 64, 0);
        f32x4 hv[HEAD_RT];
#pragma unroll
        for (int r = 0; r < HEAD_RT; ++r) hv[r] = hs4[r * k4n + c];
#pragma unroll
        for (int j = 0; j < HEAD_NC; ++j)
#pragma unroll
          for (int r = 0; r < HEAD_RT; ++r) acc[j][r] = dot4(wv[1][j], hv[r], acc[j][r]);
      }
    }
#pragma unroll
    for (int j = 0; j < HEAD_NC; ++j)
#pragma unroll
      for (int r = 0; r < HEAD_RT; ++r)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j][r] += __shfl_xor(acc[j][r], o);
#pragma unroll
    for (int j = 0; j < HEAD_NC; ++j) {
      const int n = nbase + q + 4 * j;
      if (lane < rows && n < N) {
        float v = acc[j][0];
#pragma unroll
        for (int r = 1; r < HEAD_RT; ++r) v = lane == r ? acc[j][r] : v;
        out[(int64_t)(m0 + lane) * N + n] = v + (bias != nullptr ? bias[n] : 0.f);
      }
    }
  }
}

struct HeadBwd {
  int M, K, N;
  const float* dy;        // [M][N]
  const float* h;         // [M][K]
  const float* w;         // [N][K]
  float* dh;              // [M][K]
  float* dh_bound;        // 32-slot bound (atomicMax of |dh|; zeroed by the caller) or nullptr
  float* part;            // [slices][N][K] partial weight gradients, then [slices][N] partial bias gradients (nullptr: no dW / db)
  int n_dh;               // blocks [0, n_dh): dh rows;  blocks behind: (slice, 256-channel chunk) of the partials
  int kchunks, slices;
};

// blocks [0, n_dh):   dh[m][k] = sum_n dy[m][n] * w[n][k] for HEAD_RT rows (thread = one float4 of k; w rows stream from the L2),
//                     + the maximum of |dh| (the bound of the split-fp16 engine's first backward operand)
// blocks behind:      part[s][n][k] = sum_{m in slice s} dy[m][n] * h[m][k] (thread = one k; the slice's h column in registers),
//                     chunk 0 also the slice's column sums of dy
__global__ void __launch_bounds__(256) k_head_bwd(const HeadBwd a) {
  __shared__ __attribute__((aligned(16))) float dys[HEAD_RS * HEAD_MAX_N];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int N = a.N, K = a.K;
  int b = blockIdx.x;
  if (b < a.n_dh) {
    const int m0 = b * HEAD_RT, rows = min(HEAD_RT, a.M - m0);
    for (int i = tid; i < HEAD_RT * N; i += 256) {
      const int r = i / N;
      dys[i] = r < rows ? a.dy[(int64_t)m0 * N + i] : 0.f;
    }
    __syncthreads();
    const int k4n = K >> 2;
    const f32x4* w4 = reinterpret_cast<const f32x4*>(a.w);
    float amax = 0.f;
    for (int c = tid; c < k4n; c += 256) {
      f32x4 acc[HEAD_RT];
#pragma unroll
      for (int r = 0; r < HEAD_RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      // (16 weight rows in flight per thread: with 4 the 51 dependent-issue L2 round trips were the launch's duration)
      for (int nb = 0; nb < N; nb += 16) {
        f32x4 wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = nb + u < N ? w4[(int64_t)(nb + u) * k4n + c] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (nb + u < N) {
#pragma unroll
            for (int r = 0; r < HEAD_RT; ++r) {
              const float g = dys[r * N + nb + u];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[r][e] = fmaf(g, wv[u][e], acc[r][e]);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < HEAD_RT; ++r) {
        if (r < rows) {
          reinterpret_cast<f32x4*>(a.dh)[(int64_t)(m0 + r) * k4n + c] = acc[r];
          amax = fmaxf(amax, fmaxf(fmaxf(fabsf(acc[r][0]), fabsf(acc[r][1])), fmaxf(fabsf(acc[r][2]), fabsf(acc[r][3]))));
        }
      }
    }
    if (a.dh_bound != nullptr) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
      if ((tid & 63) == 0) red[tid >> 6] = amax;
      __syncthreads();
      if (tid == 0) s16_atomic_bound(a.dh_bound, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    }
    return;
  }
  b -= a.n_dh;
  const int s = b / a.kchunks, kc = b - s * a.kchunks;
  const int m0 = s * HEAD_RS, rows = min(HEAD_RS, a.M - m0);
  // the slice's dy rows at a pitch of NP = N rounded up to 4 floats (zero padded): every thread reads the SAME address (an LDS
  // broadcast), so four columns per ds_read_b128 instead of one per ds_read_b32 -- the scalar form issued 32 x N LDS reads per
  // thread and was the launch's duration
  const int NP = (N + 3) & ~3;
  for (int i = tid; i < HEAD_RS * NP; i += 256) {
    const int r = i / NP, n = i - r * NP;
    dys[i] = (r < rows && n < N) ? a.dy[(int64_t)(m0 + r) * N + n] : 0.f;
  }
  const int k = kc * 256 + tid;
  float hv[HEAD_RS];
#pragma unroll
  for (int r = 0; r < HEAD_RS; ++r) hv[r] = (r < rows && k < K) ? a.h[(int64_t)(m0 + r) * K + k] : 0.f;
  __syncthreads();
  if (k < K) {
    float* prow = a.part + (int64_t)s * N * K + k;
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dys);
    const int np4 = NP >> 2;
    for (int n4 = 0; n4 < np4; ++n4) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < HEAD_RS; ++r) {
        const f32x4 g = d4[r * np4 + n4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(g[e], hv[r], acc[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n4 * 4 + e < N) prow[(int64_t)(n4 * 4 + e) * K] = acc[e];
    }
  }
  if (kc == 0 && tid < N) {
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < HEAD_RS; ++r) acc += dys[r * NP + tid];
    a.part[(int64_t)a.slices * N * K + (int64_t)s * N + tid] = acc;
  }
}

// dw[n][k] = sum_s part[s][n][k] (slice order), db[n] = sum_s pdb[s][n]: thread = one float4 of dw; the last block folds db
__global__ void __launch_bounds__(256) k_head_fold(int K, int N, int slices, const float* __restrict__ part, float* __restrict__ dw,
                                                   float* __restrict__ db) {
  const int64_t nk4 = (int64_t)N * (K >> 2);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x + 1 == gridDim.x) {
    if (db != nullptr && (int)threadIdx.x < N) {
      const float* pdb = part + (int64_t)slices * N * K;
      float acc = 0.f;
      for (int s = 0; s < slices; ++s) acc += pdb[(int64_t)s * N + threadIdx.x];
      db[threadIdx.x] = acc;
    }
    return;
  }
  if (i >= nk4) return;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 4 <= slices; s += 4) {                // four independent loads in flight, summed in slice order
    const f32x4 v0 = p4[(int64_t)s * nk4 + i], v1 = p4[(int64_t)(s + 1) * nk4 + i], v2 = p4[(int64_t)(s + 2) * nk4 + i],
                v3 = p4[(int64_t)(s + 3) * nk4 + i];
    acc += v0;
    acc += v1;
    acc += v2;
    acc += v3;
  }
  for (; s < slices; ++s) acc += p4[(int64_t)s * nk4 + i];
  reinterpret_cast<f32x4*>(dw)[i] = acc;
}

bool head_shape_ok(int64_t M, int32_t K, int32_t N) {
  return M > 0 && M <= VP3D_HEAD_MAX_ROWS && K > 0 && K % 4 == 0 && K <= HEAD_MAX_K && N > 0 && N <= HEAD_MAX_N;
}

}  // namespace
}  // namespace vp3d

using namespace vp3d;

extern "C" {

int vp3d_head_supported(int64_t M, int32_t K, int32_t N) { return head_shape_ok(M, K, N) ? 1 : 0; }

int64_t vp3d_head_bwd_ws_floats(int64_t M, int32_t K, int32_t N) {
  if (!head_shape_ok(M, K, N)) return 0;
  const int64_t slices = (M + HEAD_RS - 1) / HEAD_RS;
  return slices * N * K + slices * N;
}

int vp3d_head_fwd(vp3d_stream_t stream, int64_t M, int32_t K, int32_t N, const float* h, const float* w, const float* bias,
                  float* out) {
  VP3D_REQUIRE(head_shape_ok(M, K, N), "head_fwd: unsupported shape (M=%lld K=%d N=%d; vp3d_head_supported)", (long long)M, K, N);
  VP3D_REQUIRE(h && w && out && aligned16(h) && aligned16(w), "head_fwd: null or misaligned pointer");
  VP3D_LAUNCH(k_head_fwd, dim3((unsigned)((M + HEAD_RT - 1) / HEAD_RT)), dim3(256), (size_t)HEAD_RT * K * 4,
                     (hipStream_t)stream, (int)M, K, N, h, w, bias, out);
  return check_launch("head_fwd");
}

int vp3d_head_bwd(vp3d_stream_t stream, int64_t M, int32_t K, int32_t N, const float* dy, const float* h, const float* w,
                  float* dh, float* dh_bound, float* ws) {
  VP3D_REQUIRE(head_shape_ok(M, K, N), "head_bwd: unsupported shape (M=%lld K=%d N=%d; vp3d_head_supported)", (long long)M, K, N);
  VP3D_REQUIRE(dy && h && w && dh && aligned16(w) && aligned16(dh) && (ws == nullptr || aligned16(ws)),
               "head_bwd: null or misaligned pointer");
  HeadBwd a;
  a.M = (int)M; a.K = K; a.N = N;
  a.dy = dy; a.h = h; a.w = w; a.dh = dh; a.dh_bound = dh_bound; a.part = ws;
  a.n_dh = (int)((M + HEAD_RT - 1) / HEAD_RT);
  a.kchunks = (K + 255) / 256;
  a.slices = (int)((M + HEAD_RS - 1) / HEAD_RS);
  const int n_dw = ws != nullptr ? a.slices * a.kchunks : 0;
  VP3D_LAUNCH(k_head_bwd, dim3((unsigned)(a.n_dh + n_dw)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("head_bwd");
}

int vp3d_head_fold(vp3d_stream_t stream, int64_t M, int32_t K, int32_t N, const float* ws, float* dw, float* db) {
  VP3D_REQUIRE(head_shape_ok(M, K, N), "head_fold: unsupported shape (M=%lld K=%d N=%d)", (long long)M, K, N);
  VP3D_REQUIRE(ws && dw && aligned16(ws) && aligned16(dw), "head_fold: null or misaligned pointer");
  const int slices = (int)((M + HEAD_RS - 1) / HEAD_RS);
  const int64_t nk4 = (int64_t)N * (K / 4);
  VP3D_LAUNCH(k_head_fold, dim3((unsigned)((nk4 + 255) / 256 + 1)), dim3(256), 0, (hipStream_t)stream, K, N, slices, ws, dw,
                     db);
  return check_launch("head_fold");
}

}  // extern "C"
