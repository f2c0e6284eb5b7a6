// The small-M tail of the strided training stack as ONE persistent kernel per direction (gfx950 / CDNA4).
//
// Reference semantics: common/model.py:190-196 (TemporalModelOptimized1f._forward_blocks)
//     res = x[:, :, shift + fw//2 :: fw];  x = drop(relu(bn(conv_k3s3(x))));  x = res + drop(relu(bn(conv_1x1(x))))
// and its autograd backward, for the blocks whose convs produce few rows (B * T_out <= a few thousand: T_out = 3 and 1 of
// arc 3,3,3,3,3 at B = 1024).  As separate launches those layers are ~45 dependent kernels of 5-90 us per direction -- GEMMs
// that fill a fraction of the chip, each followed by a split-K finish, a statistics finalize and an activation pass, with
// the host's launch latency between them: 1.2 ms of a 4.4 ms step for 10 % of its FLOPs (profiles/r02_step_timeline.txt).
//
// Here a grid of co-resident workgroups (2 per CU) walks the layers itself:
//   forward, per layer    GEMM (128x128 S16 MFMA tiles x K-slices, raw partial tiles)        | grid barrier
//                         statistics: a workgroup owns 8 channels x ALL rows: sums the K-slices into the conv output y,
//                           exact two-pass BatchNorm statistics in fp64, scale/shift, running statistics  | grid barrier
//                         activation: [res +] dropout(relu(bn(y))) -> S16 rows, activation bits, transposed copy for the
//                           next conv's weight gradient (the per-launch kernel's body)       | grid barrier
//   backward, per layer   go = sum of the next conv's dgrad K-slices (+ residual gradient); sum g, sum g*xhat per channel
//                           (8 channels x all rows, fp64) -> dgamma, dbeta; max|go|  [+ the previous layer's dW un-pack]
//                                                                                             | grid barrier
//                         dy = BN/ReLU/dropout backward -> S16 rows + transposed copy (per-launch kernel's body) | barrier
//                         GEMMs: dgrad tiles (dy x Wd) and wgrad tiles (dy^T x x^T) in ONE phase -- together they fill the
//                           grid where each alone does not                                    | grid barrier
// so that "each conv is a fused conv + BN + ReLU + dropout kernel" (north star) holds for these layers in the literal sense:
// conv, batch statistics, normalisation, activation and dropout of four layers are one launch.
//
// Grid barrier: monotonic arrival counter in device memory (agent-scope release -> relaxed atomic -> acquire, the hand-off
// k_bn_bwd_reduce_strips uses between workgroups of different XCDs), zeroed by the launcher; every workgroup must be
// resident, so the launcher sizes the grid from the occupancy query and refuses to launch otherwise.  A spinning wave gives
// up after ~seconds and raises an error flag instead of hanging the GPU.
#include "vp3d_internal.h"
#include "vp3d_s16.h"
#include "vp3d_s16_mma.h"
#include "vp3d_s16_stream_bodies.h"

#include <cstdlib>
#include <mutex>
#include <vector>

namespace vp3d {
namespace {

using namespace mma;
using namespace s16b;

typedef Cfg<2, 2, 2, 2, 2, 32, 0, 1> TC;      // 128x128 tile, 4 waves of 64x64, two 32-element K stages, buffer-descriptor DMA
constexpr int T_NT = 256;                     // threads per workgroup
constexpr int T_SMEM = 2 * TC::STAGE_B;       // 64 KiB: the operand ring; statistics / transposition scratch aliases it
constexpr int kMaxTail = 8;                   // layers (4 blocks)
constexpr unsigned kSpinLimit = 1u << 22;     // polls of ~0.5 us before a barrier gives up

// ---------------------------------------------------------------------------------------------------------
// grid barrier
// ---------------------------------------------------------------------------------------------------------
// Two-level arrival with ONE cache write-back / invalidate per XCD.  Workgroup b runs on XCD b % 8 (round-robin dispatch;
// verified once per device by k_xcc_probe, else the flat fallback below is used), and an XCD's 64 workgroups share its L2:
//   every workgroup   drains its stores into the L2 (vmcnt(0)) and arrives on its XCD's counter;
//   the XCD's last    writes the L2 back (agent-scope release), arrives on the top counter, waits for the top flag (the last
//                     XCD publishes it), invalidates L1 + L2 (agent-scope acquire) and releases its XCD through the XCD's flag;
//   the others        poll their XCD's flag and invalidate (agent scope: their CU's L1; the L2 is already clean).
// First versions: 512 workgroups on ONE counter + flag, each with its own L2 write-back / invalidate: ~45 us per barrier
// (device-scope atomics and polls serialise per line at the memory side); 32 groups with per-group flags, still one
// write-back / invalidate PER WORKGROUP: ~18 us.  Every counter / flag sits on a 128-byte line of its own.
constexpr int kXcds = 8;
constexpr int kLineWords = 32;
// sync words: [0] error flag | line 1: top counter | line 2: top flag | lines 3..10: XCD counters | lines 11..18: XCD flags
constexpr int kSyncWords = (3 + 2 * kXcds) * kLineWords;

struct GridSync {
  unsigned* base;    // kSyncWords zeroed words
  unsigned nwg;
  unsigned k;        // barriers passed so far
  int grouped;       // 1: workgroup b runs on XCD b % 8 (probed): per-XCD cache maintenance; 0: every workgroup fences
  unsigned long long* trace;   // optional: workgroup 0 stamps the 100 MHz wall clock at kernel start and after every barrier
  int n_stamp;
};

__device__ __forceinline__ void stamp(GridSync& g) {
  if (g.trace != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && g.n_stamp < 126) g.trace[1 + g.n_stamp] = wall_clock64();
  ++g.n_stamp;
  if (g.trace != nullptr && blockIdx.x == 0 && threadIdx.x == 0) g.trace[0] = (unsigned long long)g.n_stamp;
}

__device__ __forceinline__ bool spin_until(unsigned* flag, unsigned k, unsigned* err) {
  unsigned spins = 0;
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - k) < 0) {
    __builtin_amdgcn_s_sleep(4);
    if ((++spins & 1023u) == 0u && (spins > kSpinLimit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // results are garbage from here on, but
      return false;                                                                 // the GPU does not hang
    }
  }
  return true;
}

__device__ __forceinline__ void grid_barrier(GridSync& g) {
  g.k += 1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are in the L2
  __syncthreads();                                   // ... and so are the workgroup's
  if (threadIdx.x == 0) {
    unsigned* err = g.base;
    unsigned* top = g.base + kLineWords;
    unsigned* topflag = g.base + 2 * kLineWords;
    const unsigned x = blockIdx.x % kXcds;
    unsigned* cnt = g.base + (3 + x) * kLineWords;
    unsigned* flag = g.base + (3 + kXcds + x) * kLineWords;
    const unsigned members = g.nwg / kXcds;            // (the grid is a multiple of 8)
    if (!g.grouped) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned v = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v + 1u == g.k * members) {                     // the XCD's last arriver
      if (g.grouped) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // the XCD's dirty lines -> memory
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1u == g.k * kXcds) __hip_atomic_store(topflag, g.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else spin_until(topflag, g.k, err);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                     // L1 + the XCD's L2 dropped
      __hip_atomic_store(flag, g.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      spin_until(flag, g.k, err);
      // (this CU's L1 must go as well; `buffer_inv sc0` alone -- the workgroup-scope invalidate -- left stale lines behind:
      // parity tests failed with it, so the members issue the agent-scope invalidate too.  What the grouped flavour saves is
      // the 63 redundant L2 WRITE-BACKS per XCD and barrier.)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  stamp(g);
}

// the XCD every workgroup of a launch really runs on (HW_REG_XCC_ID, bits 3:0): the grouped barrier above is only used
// when workgroup b ran on XCD b % 8 in this probe (same grid, same round-robin dispatcher)
__global__ void k_xcc_probe(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf);
}

// The probe certifies the mapping for ITS launch; every persistent launch re-checks its own placement (one s_getreg per
// workgroup): a workgroup of the grouped flavour that finds itself on another XCD than b % 8 (partition mode changed, a
// dispatcher that places this grid differently) raises the error flag -- the host reports the launch as invalid
// (ops_s16._tail_watch) instead of consuming lines that a foreign XCD's write-back never covered.
__device__ __forceinline__ void check_xcc_placement(const GridSync& g) {
  if (g.grouped && threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu;
    if (xcc != (blockIdx.x % kXcds)) __hip_atomic_store(g.base, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// items of a phase: workgroup b runs on XCD b % 8 and takes a CONTIGUOUS eighth of the item order, so that the tiles an XCD
// works on at a time are neighbours (shared operand panels in its L2); returns false past the end
__device__ __forceinline__ bool next_item(int k, int items, int& L) {
  const int per = (items + 7) >> 3;
  const int q = ((int)blockIdx.x >> 3) + k * ((int)gridDim.x >> 3);
  L = ((int)blockIdx.x & 7) * per + q;
  return q < per && L < items;
}
__device__ __forceinline__ int max_rounds(int items) { return (((items + 7) >> 3) + ((int)gridDim.x >> 3) - 1) / ((int)gridDim.x >> 3); }

// ---------------------------------------------------------------------------------------------------------
// plain S16 "NT" GEMM tile:  part[split][m][n] = 2^(ea+eb) * sum_{k in slice} A[m][k] * B[n][k]      (raw partial tiles)
// ---------------------------------------------------------------------------------------------------------
struct NtOp {
  const float* A;            // S16 rows [M][lda]
  const float* B;            // S16 rows [N][ldb]
  float* part;               // [splits][M][N] fp32
  const float* bound_a;      // 32-slot bounds (exponents of the two operands)
  const float* bound_b;
  int M, N, K;               // K % 32 == 0, N % 4 == 0
  int lda, ldb;              // row pitches in 4-byte units
  int splits, kt_per_split;  // K-slices, K-tiles per slice
  int mt, nt;                // tile grid
};
__device__ __forceinline__ int nt_items(const NtOp& g) { return g.mt * g.nt * g.splits; }

__device__ __forceinline__ void nt_tile(const NtOp& g, char* smem, int L) {
  constexpr int RB = TC::RB, CB = TC::CB, BM = TC::BM, BN = TC::BN, PA = TC::PA, PB = TC::PB, BK = TC::BKE, ROWB = TC::ROWB;
  constexpr int RPP = TC::RPP, CPR = ROWB / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / TC::WN, wn = w % TC::WN;
  const int h = lane >> 5, cl = lane & 31;
  const int tiles = g.mt * g.nt;
  const int split = L / tiles, t = L - split * tiles;
  const int tile_m = t / g.nt, tile_n = t - tile_m * g.nt;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt_all = g.K / BK;
  const int kt_begin = split * g.kt_per_split;
  const int nkt = max(0, min(nkt_all, kt_begin + g.kt_per_split) - kt_begin);

  f32x16 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LDS-DMA through buffer descriptors: one 32-bit byte offset per 1-KiB piece, out-of-range rows deliver zeros
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((int64_t)g.M * g.lda * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)((int64_t)g.N * g.ldb * 4), 0x00020000);
  int a_cur[PA], b_cur[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (w * PA + i) * RPP + lane / CPR;
    const int chunk = (lane & (CPR - 1)) ^ TC::swz(r);
    const int row = m0 + r;
    a_cur[i] = row < g.M ? (row * g.lda + kt_begin * BK + chunk * 4) * 4 : kOob;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int r = (w * PB + i) * RPP + lane / CPR;
    const int chunk = (lane & (CPR - 1)) ^ TC::swz(r);
    const int row = n0 + r;
    b_cur[i] = row < g.N ? (row * g.ldb + kt_begin * BK + chunk * 4) * 4 : kOob;
  }
  auto issue = [&](int stage, bool live) {           // `live` = false: all-zero DMA (keeps vmcnt uniform)
    char* sA = smem + stage * TC::STAGE_B;
    char* sB = sA + TC::A_B;
#pragma unroll
    for (int i = 0; i < PA; ++i) blds16(rsA, live ? a_cur[i] : kOob, sA + (w * PA + i) * 1024);
#pragma unroll
    for (int i = 0; i < PB; ++i) blds16(rsB, live ? b_cur[i] : kOob, sB + (w * PB + i) * 1024);
#pragma unroll
    for (int i = 0; i < PA; ++i) a_cur[i] += BK * 4;    // (kOob + a few K-tiles stays out of range)
#pragma unroll
    for (int i = 0; i < PB; ++i) b_cur[i] += BK * 4;
  };
  const int sw = TC::swz(cl);
  const int off0 = ((2 * (0 + h)) ^ sw) * 16;
  const int off1 = ((2 * (2 + h)) ^ sw) * 16;
  const int a_row = (wm * RB * 32 + cl) * ROWB;
  const int b_row = (wn * CB * 32 + cl) * ROWB;

  __syncthreads();                                   // the previous item of this workgroup is done with the ring / staging
  if (nkt > 0) {
    issue(0, true);
    int st_c = 0, st_i = 1;
    for (int it = 0; it < nkt; ++it) {
      wait_vmcnt<0>();
      __syncthreads();
      issue(st_i, it + 1 < nkt);
      const char* sA = smem + st_c * TC::STAGE_B;
      compute_tile<RB, CB, BK / 16, ROWB>(sA + a_row, sA + TC::A_B + b_row, acc, off0, off1);
      st_c ^= 1;
      st_i ^= 1;
    }
  }
  wait_vmcnt<0>();                                   // the trailing zero DMAs must not land in the staging below
  {
    const int ex = s16_exp_of(g.bound_a) + s16_exp_of(g.bound_b);
    const float scale = s16_pow2(ex);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= scale;
  }
  __syncthreads();                                   // every wave is done reading the operand ring
  constexpr int WCOLS = CB * 32, LPR = WCOLS / 4, ERPP = 64 / LPR;
  float* wreg = reinterpret_cast<float*>(smem) + w * (32 * WCOLS);
  const int rr = lane / LPR, c4 = (lane % LPR) * 4;
  float* out = g.part + (int64_t)split * g.M * g.N;
  const int n = n0 + wn * WCOLS + c4;
#pragma unroll
  for (int i = 0; i < RB; ++i) {
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        wreg[r * WCOLS + j * 32 + cl] = acc[i][j][reg];
      }
    epi_stage_sync();
#pragma unroll 4
    for (int ps = 0; ps < 32 / ERPP; ++ps) {
      const int r = ps * ERPP + rr;
      const int m = m0 + (wm * RB + i) * 32 + r;
      if (m >= g.M || n >= g.N) continue;
      *reinterpret_cast<f32x4*>(out + (int64_t)m * g.N + n) = *reinterpret_cast<const f32x4*>(wreg + r * WCOLS + c4);
    }
    epi_stage_sync();
  }
}

// ---------------------------------------------------------------------------------------------------------
// block reductions over 256 threads x 8 channels in fp64, fixed tree (deterministic); result in red[0..7]
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_sum8(double* red, const double (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = v[e];
  __syncthreads();
  for (int o = T_NT / 2; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] += red[(threadIdx.x + o) * 8 + e];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
struct FwdLayer {
  const float* wf;            // S16 forward pack [C][taps*C]
  const float* w_bound;
  const float* gamma;
  const float* beta;
  float* run_mean;            // (nullable)
  float* run_var;
  int64_t* nbt;
  float* y;                   // fp32 conv output [M][C] (saved for backward)
  float* coef;                // [4][C]: scale, shift, mean, invstd
  float* a;                   // S16 activation rows [M][C]
  const float* a_bound;       // 32-slot bound of a (guaranteed: Samuelson, computed by the step's prologue)
  float* a_f32;               // optional fp32 copy of the activation (the stack output: input of the fp32 shrink conv)
  float* a_t;                 // optional transposed S16 copy [(tap*C + c)][ld_at] for the NEXT conv's weight gradient
  uint8_t* bits;              // optional activation bits
  DropP drop;
  int M, taps, splits, kt_per_split;
  int res_start;              // >= 0: second conv of a block (residual = block input rows taps_prev*m + res_start); -1: none
  int taps_at, ld_at;
};

struct TailFwdArgs {
  FwdLayer L[kMaxTail];
  int n_layers, C;
  const float* x0;            // S16 input rows of the first layer [M0*taps0][C]
  const float* x0_bound;
  float* part;                // workspace: max over layers of splits*M*C floats
  float eps, momentum;
  const float* momentum_dev;  // (nullable) device float read instead of `momentum`
  unsigned* sync;             // kSyncWords words of barrier state ([0] = error flag)
  unsigned long long* trace;  // (nullable) phase time stamps of workgroup 0
  int grouped;                // barrier flavour (tail_grid)
};

// statistics of layer l for the 8 channels of strip s: y = sum of the K-slices, exact two-pass mean / variance (fp64),
// scale / shift / running statistics (the arithmetic of k_bn_finalize)
__device__ __forceinline__ void fwd_stats_strip(const TailFwdArgs& a, const FwdLayer& F, int s, double* red) {
  const int M = F.M, C = a.C, c = s * 8, tid = threadIdx.x;
  double sum[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const int64_t slice = (int64_t)M * C;
  for (int m = tid; m < M; m += T_NT) {
    const float* p = a.part + (int64_t)m * C + c;
    f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
    for (int sp = 1; sp < F.splits; ++sp) {          // slice order: deterministic
      v0 += *reinterpret_cast<const f32x4*>(p + sp * slice);
      v1 += *reinterpret_cast<const f32x4*>(p + sp * slice + 4);
    }
    *reinterpret_cast<f32x4*>(F.y + (int64_t)m * C + c) = v0;
    *reinterpret_cast<f32x4*>(F.y + (int64_t)m * C + c + 4) = v1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sum[e] += (double)v0[e];
      sum[4 + e] += (double)v1[e];
    }
  }
  block_sum8(red, sum);
  double mean[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) mean[e] = red[e] / (double)M;
  __syncthreads();
  double q[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int m = tid; m < M; m += T_NT) {              // (the thread re-reads the rows it has just written)
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(F.y + (int64_t)m * C + c);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(F.y + (int64_t)m * C + c + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double d0 = (double)v0[e] - mean[e], d1 = (double)v1[e] - mean[4 + e];
      q[e] += d0 * d0;
      q[4 + e] += d1 * d1;
    }
  }
  block_sum8(red, q);
  if (tid < 8) {
    const int cc = c + tid;
    const double mu = mean[tid];                     // (mean[] is identical in every thread)
    double var = red[tid] / (double)M;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)a.eps);
    const float sc = (float)((double)F.gamma[cc] * invstd);
    F.coef[cc] = sc;
    F.coef[C + cc] = F.beta[cc] - (float)mu * sc;
    F.coef[2 * C + cc] = (float)mu;
    F.coef[3 * C + cc] = (float)invstd;
    const float mom = a.momentum_dev != nullptr ? a.momentum_dev[0] : a.momentum;
    if (F.run_mean != nullptr) F.run_mean[cc] = (1.f - mom) * F.run_mean[cc] + mom * (float)mu;
    if (F.run_var != nullptr) {
      const double unbiased = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
      F.run_var[cc] = (1.f - mom) * F.run_var[cc] + mom * (float)unbiased;
    }
  }
  if (s == 0 && tid == 0 && F.nbt != nullptr) F.nbt[0] += 1;
  __syncthreads();                                   // red[] is free for the next strip
}

__global__ void __launch_bounds__(T_NT, 2) k_tail_fwd(const TailFwdArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[T_SMEM];
  GridSync gs{a.sync, gridDim.x, 0u, a.grouped, a.trace, 0};
  check_xcc_placement(gs);
  stamp(gs);
  const int C = a.C;
  for (int l = 0; l < a.n_layers; ++l) {
    const FwdLayer& F = a.L[l];
    // ---- conv: raw K-slice tiles of  in [M][taps*C]  x  Wf [C][taps*C]^T --------------------------------------------
    {
      NtOp g;
      g.A = l == 0 ? a.x0 : a.L[l - 1].a;
      g.bound_a = l == 0 ? a.x0_bound : a.L[l - 1].a_bound;
      g.B = F.wf;
      g.bound_b = F.w_bound;
      g.part = a.part;
      g.M = F.M; g.N = C; g.K = F.taps * C;
      g.lda = g.K; g.ldb = g.K;
      g.splits = F.splits; g.kt_per_split = F.kt_per_split;
      g.mt = (F.M + TC::BM - 1) / TC::BM; g.nt = (C + TC::BN - 1) / TC::BN;
      const int items = nt_items(g), rounds = max_rounds(items);
      for (int k = 0; k < rounds; ++k) {
        int L;
        if (next_item(k, items, L)) nt_tile(g, smem, L);
      }
    }
    grid_barrier(gs);
    // ---- BatchNorm statistics (8 channels x all rows per workgroup) -------------------------------------------------
    for (int s = blockIdx.x; s < C / 8; s += gridDim.x) fwd_stats_strip(a, F, s, reinterpret_cast<double*>(smem));
    grid_barrier(gs);
    // ---- activation: [res +] dropout(relu(bn(y))) -> S16 rows (+ bits, transposed copy, fp32 copy) -----------------
    {
      DropP d = F.drop;
      drop_resolve(d);
      const float inv = s16_pow2(-s16_exp_of(F.a_bound));
      ResS16 rm;
      rm.res = nullptr;
      rm.bound = nullptr;
      float rscale = 0.f;
      if (F.res_start >= 0) {                         // block input: the rows taps_prev * m + res_start of layer l-1's input
        rm.res = l == 1 ? a.x0 : a.L[l - 2].a;
        rm.bound = l == 1 ? a.x0_bound : a.L[l - 2].a_bound;
        rscale = s16_pow2(s16_exp_of(rm.bound));
      }
      rm.t_dst = 1; rm.r_t = l > 0 ? a.L[l - 1].taps : 1; rm.r_stride = 0; rm.r_off = F.res_start; rm.r_ld = C;
      rm.div_t.mul = 0x80000000u; rm.div_t.shift = 31; rm.div_t.d = 1;       // n / 1
      TOut t{F.a_t, (int64_t)F.ld_at, F.taps_at > 0 ? F.taps_at : 1};
      const int R = (F.a_t != nullptr ? t.taps : 1) * 64;
      const int gx = C / 64, gy = (F.M + R - 1) / R;
      for (int it = blockIdx.x; it < gx * gy; it += gridDim.x) {
        bn_act_fwd_s16_body(F.M, C, F.y, F.coef, F.coef + C, d, rm, inv, rscale, F.a, F.a_f32, t, F.bits,
                            reinterpret_cast<float*>(smem), it % gx, it / gx);
        __syncthreads();                             // the LDS tile is free for the next item
      }
    }
    if (l + 1 < a.n_layers) grid_barrier(gs);
  }
  stamp(gs);
}

// ---------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------
struct BwdLayer {
  const float* wd;            // S16 dgrad pack [taps*C][C]
  const float* w_bound;
  const float* y;             // fp32 conv output [M][C]
  const float* coef;          // [4][C]
  const uint8_t* bits;
  const float* x_t;           // transposed S16 input of this conv [(taps*C)][ld_xt]
  const float* x_bound;
  float* go;                  // fp32 [M][C]: gradient wrt this layer's activation (last layer: input; else assembled here)
  float* go_bound;            // 32 slots (last layer: input; else zeroed, measured here)
  float* dy;                  // S16 rows [M][C]
  float* dy_t;                // S16 transposed [C][ld_dyt]
  float* dy_bound;            // 32 slots, zeroed; slot 0 receives the guaranteed bound of dy
  float* dgamma;
  float* dbeta;
  float* dw;                  // [C][C][taps] (reference layout)
  int M, taps, res_start;
  int ld_xt, ld_dyt;
  int splits_d, ktps_d, splits_w, ktps_w;
};

struct TailBwdArgs {
  BwdLayer L[kMaxTail];
  int n_layers, C;
  float inv_keep;
  float* dpart;               // dgrad K-slices: max over layers of splits_d * M * taps * C floats
  float* wpart;               // wgrad K-slices: max over layers of splits_w * C * taps * C floats
  float* dx0;                 // fp32 [M0*taps0][C]: gradient wrt the tail's input rows
  float* dx0_bound;           // 32 slots (zeroed): max|dx0|
  unsigned* sync;
  unsigned long long* trace;
  int grouped;
};

// gradient wrt the rows of layer l's activation = the dgrad of conv l+1 (its K-slices, viewed [M_l][C]) plus, when conv l+1
// opens a block, the residual gradient of that block's second conv:  go_l[r] += go_{l+2}[r / taps] for r % taps == start
__device__ __forceinline__ void assemble_go8(const TailBwdArgs& a, int l_next, int r, int c, float (&v)[8]) {
  const BwdLayer& N = a.L[l_next];
  const int C = a.C, taps = N.taps;
  const int m = taps == 1 ? r : r / taps, tap = r - m * taps;
  const int64_t slice = (int64_t)N.M * taps * C;
  const float* p = a.dpart + ((int64_t)m * taps + tap) * C + c;
  f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
  for (int sp = 1; sp < N.splits_d; ++sp) {
    v0 += *reinterpret_cast<const f32x4*>(p + sp * slice);
    v1 += *reinterpret_cast<const f32x4*>(p + sp * slice + 4);
  }
  if ((l_next & 1) == 0 && l_next + 1 < a.n_layers) {      // conv l_next opens a block: the block's residual gradient
    const BwdLayer& N2 = a.L[l_next + 1];
    if (N2.res_start >= 0 && tap == N2.res_start) {
      const float* q = N2.go + (int64_t)m * C + c;
      v0 += *reinterpret_cast<const f32x4*>(q);
      v1 += *reinterpret_cast<const f32x4*>(q + 4);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = v0[e];
    v[4 + e] = v1[e];
  }
}

// strip s (8 channels x all rows) of layer l: go (assembled, stored, measured), dgamma / dbeta
__device__ __forceinline__ void bwd_reduce_strip(const TailBwdArgs& a, int l, int s, double* red) {
  const BwdLayer& B = a.L[l];
  const int M = B.M, C = a.C, c = s * 8, tid = threadIdx.x;
  const bool given = l + 1 == a.n_layers;            // the last layer's go is the kernel's input
  float mu[8], is[8];
  load8(B.coef + 2 * C + c, mu);
  load8(B.coef + 3 * C + c, is);
  double sg[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, sgx[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  float gmax = 0.f;
  for (int m = tid; m < M; m += T_NT) {
    float gv[8];
    if (given) {
      load8(B.go + (int64_t)m * C + c, gv);
    } else {
      assemble_go8(a, l + 1, m, c, gv);
      *reinterpret_cast<f32x4*>(B.go + (int64_t)m * C + c) = f32x4{gv[0], gv[1], gv[2], gv[3]};
      *reinterpret_cast<f32x4*>(B.go + (int64_t)m * C + c + 4) = f32x4{gv[4], gv[5], gv[6], gv[7]};
#pragma unroll
      for (int e = 0; e < 8; ++e) gmax = fmaxf(gmax, fabsf(gv[e]));
    }
    float yv[8];
    load8(B.y + (int64_t)m * C + c, yv);
    const uint32_t bits = B.bits[act_bits_index(c, m, M)];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = ((bits >> e) & 1u) ? gv[e] * a.inv_keep : 0.f;
      sg[e] += (double)g;
      sgx[e] += (double)(g * ((yv[e] - mu[e]) * is[e]));
    }
  }
  block_sum8(red, sg);
  if (tid < 8) B.dbeta[c + tid] = (float)red[tid];
  __syncthreads();
  block_sum8(red, sgx);
  if (tid < 8) B.dgamma[c + tid] = (float)red[tid];
  if (!given) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
    if ((tid & 63) == 0) atomicMax(reinterpret_cast<int*>(B.go_bound + ((s * 4 + (tid >> 6)) & (kBoundSlots - 1))), __float_as_int(gmax));
  }
  __syncthreads();
}

// dw[co][ci][k] = sum_s wpart[s][co][k*C + ci]   (virtual block vb of nvb; the arithmetic of k_wgrad_reduce_v4)
template <int TAPS>
__device__ __forceinline__ void wgrad_unpack(const float* __restrict__ partials, int splits, int C, float* __restrict__ dw, int vb,
                                             int nvb) {
  const int cq = C >> 2, ld = TAPS * C;
  const int64_t total = (int64_t)C * cq, mat = (int64_t)C * ld;
  for (int64_t i = (int64_t)vb * T_NT + threadIdx.x; i < total; i += (int64_t)nvb * T_NT) {
    const int64_t co = i / cq;
    const int ci = (int)(i - co * cq) * 4;
    const float* src = partials + co * (int64_t)ld + ci;
    f32x4 acc[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s)
#pragma unroll
      for (int k = 0; k < TAPS; ++k) acc[k] += *reinterpret_cast<const f32x4*>(src + (int64_t)s * mat + (int64_t)k * C);
    float o[4 * TAPS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < TAPS; ++k) o[j * TAPS + k] = acc[k][j];
    f32x4* dst = reinterpret_cast<f32x4*>(dw + (co * C + ci) * TAPS);
#pragma unroll
    for (int q = 0; q < TAPS; ++q) dst[q] = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
  }
}

__global__ void __launch_bounds__(T_NT, 2) k_tail_bwd(const TailBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[T_SMEM];
  __shared__ float bred[4];
  GridSync gs{a.sync, gridDim.x, 0u, a.grouped, a.trace, 0};
  check_xcc_placement(gs);
  stamp(gs);
  const int C = a.C;
  constexpr int kUnpackBlocks = 128;                 // virtual blocks of a weight-gradient un-pack riding in a reduce phase
  for (int l = a.n_layers - 1; l >= 0; --l) {
    const BwdLayer& B = a.L[l];
    // ---- column sums of this layer (strip-owned) + un-pack of the previous (= next higher) layer's weight gradient ----
    {
      const int strips = C / 8;
      const int extra = l + 1 < a.n_layers ? kUnpackBlocks : 0;
      for (int it = blockIdx.x; it < strips + extra; it += gridDim.x) {
        if (it < strips) {
          bwd_reduce_strip(a, l, it, reinterpret_cast<double*>(smem));
        } else {
          const BwdLayer& P = a.L[l + 1];
          if (P.taps == 3) wgrad_unpack<3>(a.wpart, P.splits_w, C, P.dw, it - strips, extra);
          else wgrad_unpack<1>(a.wpart, P.splits_w, C, P.dw, it - strips, extra);
        }
      }
    }
    grid_barrier(gs);
    // ---- dy = BatchNorm / ReLU / dropout backward -> S16 rows + transposed copy -------------------------------------
    {
      // guaranteed bound of dy (k_bn_bwd_finalize_bound's formula), evaluated identically by every workgroup
      const float gmax = s16_load_bound(B.go_bound) * a.inv_keep;
      const float inv_m = 1.0f / (float)B.M, sqrt_m1 = sqrtf((float)(B.M > 1 ? B.M - 1 : 1));
      float bm = 0.f;
      for (int c = threadIdx.x; c < C; c += T_NT)
        bm = fmaxf(bm, fabsf(B.coef[c]) * (gmax + fabsf(B.dbeta[c]) * inv_m + sqrt_m1 * fabsf(B.dgamma[c]) * inv_m));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o));
      __syncthreads();
      if ((threadIdx.x & 63) == 0) bred[threadIdx.x >> 6] = bm;
      __syncthreads();
      bm = fmaxf(fmaxf(bred[0], bred[1]), fmaxf(bred[2], bred[3]));
      if (blockIdx.x == 0 && threadIdx.x == 0) B.dy_bound[0] = bm;       // (slots 1.. stay zero) for the GEMM phase below
      const float inv = s16_pow2(-s16_exp_for_bound(bm));
      DropP d{};
      TOut t{B.dy_t, (int64_t)B.ld_dyt, 1};
      const int gx = C / 64, gy = (B.M + 63) / 64;
      for (int it = blockIdx.x; it < gx * gy; it += gridDim.x) {
        bn_bwd_apply_s16_body<true, false>(B.M, C, B.go, B.y, B.coef, B.coef + C, B.coef + 2 * C, B.coef + 3 * C, d, B.bits,
                                           a.inv_keep, B.dgamma, B.dbeta, inv, B.dy, t, reinterpret_cast<float*>(smem), it % gx,
                                           it / gx, gy);
      }
    }
    grid_barrier(gs);
    // ---- data gradient and weight gradient of conv l, one phase ---------------------------------------------------------
    {
      NtOp gd, gw;
      gd.A = B.dy; gd.bound_a = B.dy_bound; gd.B = B.wd; gd.bound_b = B.w_bound; gd.part = a.dpart;
      gd.M = B.M; gd.N = B.taps * C; gd.K = C; gd.lda = C; gd.ldb = C;
      gd.splits = B.splits_d; gd.kt_per_split = B.ktps_d;
      gd.mt = (gd.M + TC::BM - 1) / TC::BM; gd.nt = (gd.N + TC::BN - 1) / TC::BN;
      gw.A = B.dy_t; gw.bound_a = B.dy_bound; gw.B = B.x_t; gw.bound_b = B.x_bound; gw.part = a.wpart;
      gw.M = C; gw.N = B.taps * C; gw.K = B.ld_dyt; gw.lda = B.ld_dyt; gw.ldb = B.ld_xt;
      gw.splits = B.splits_w; gw.kt_per_split = B.ktps_w;
      gw.mt = (C + TC::BM - 1) / TC::BM; gw.nt = (gw.N + TC::BN - 1) / TC::BN;
      const int nd = nt_items(gd), nw = nt_items(gw);
      // two item lists, each spread over the XCDs on its own: the heavier list first
      const int r1 = max_rounds(nw);
      for (int k = 0; k < r1; ++k) {
        int L;
        if (next_item(k, nw, L)) nt_tile(gw, smem, L);
      }
      const int r2 = max_rounds(nd);
      for (int k = r2 - 1; k >= 0; --k) {             // (reverse round order: the workgroups with a spare slot in the last
        int L;                                        //  wgrad round start on the dgrad's partly filled round)
        if (next_item(k, nd, L)) nt_tile(gd, smem, L);
      }
    }
    grid_barrier(gs);
  }
  // ---- gradient wrt the tail's input rows + the last weight-gradient un-pack -----------------------------------------------
  {
    const BwdLayer& B0 = a.L[0];
    const int rows = B0.M * B0.taps, groups = C / 8;
    const int nblk = max(1, (int)gridDim.x - kUnpackBlocks);
    if ((int)blockIdx.x < nblk) {
      float amax = 0.f;
      for (int64_t i = (int64_t)blockIdx.x * T_NT + threadIdx.x; i < (int64_t)rows * groups; i += (int64_t)nblk * T_NT) {
        const int r = (int)(i / groups), c = (int)(i - (int64_t)r * groups) * 8;
        float v[8];
        assemble_go8(a, 0, r, c, v);
        *reinterpret_cast<f32x4*>(a.dx0 + (int64_t)r * C + c) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(a.dx0 + (int64_t)r * C + c + 4) = f32x4{v[4], v[5], v[6], v[7]};
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
      if ((threadIdx.x & 63) == 0)
        atomicMax(reinterpret_cast<int*>(a.dx0_bound + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (kBoundSlots - 1))), __float_as_int(amax));
    } else {
      const int vb = (int)blockIdx.x - nblk, nvb = (int)gridDim.x - nblk;
      if (B0.taps == 3) wgrad_unpack<3>(a.wpart, B0.splits_w, C, B0.dw, vb, nvb);
      else wgrad_unpack<1>(a.wpart, B0.splits_w, C, B0.dw, vb, nvb);
    }
  }
  stamp(gs);
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct TailGrid {
  int wgs = 0;      // co-resident workgroups the two kernels are launched with (0: not yet queried)
  int grouped = 0;  // 1: a probe launch of that grid put workgroup b on XCD b % 8
};
TailGrid g_grid[64];
std::mutex g_grid_mutex;                               // (first use per device: occupancy query + probe launch)

int tail_grid(bool fwd, int* wgs) {
  std::lock_guard<std::mutex> lock(g_grid_mutex);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    set_error("tail: hipGetDevice failed");
    return VP3D_E_INVALID;
  }
  if (g_grid[dev].wgs == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      (void)hipGetLastError();
      set_error("tail: hipGetDeviceProperties failed");
      return VP3D_E_INVALID;
    }
    int occ_f = 0, occ_b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_f, k_tail_fwd, T_NT, 0) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, k_tail_bwd, T_NT, 0) != hipSuccess) {
      (void)hipGetLastError();
      set_error("tail: occupancy query failed");
      return VP3D_E_INVALID;
    }
    int occ = occ_f < occ_b ? occ_f : occ_b;
    if (occ > 2) occ = 2;
    if (occ < 1) {
      set_error("tail: the persistent kernels do not fit on a compute unit");
      return VP3D_E_INVALID;
    }
    int wg = prop.multiProcessorCount * occ;
    wg -= wg % 8;                                      // whole XCD shares (next_item)
    // one-time probe (synchronous; the first tail launch of a process is an eager one -- graph.py warms up before it
    // captures): where does the dispatcher put workgroup b of a grid of this size?
    int grouped = 0;
    int* d_ids = nullptr;
    if (hipMalloc((void**)&d_ids, wg * sizeof(int)) == hipSuccess) {
      std::vector<int> ids(wg, -1);
      VP3D_LAUNCH(k_xcc_probe, dim3(wg), dim3(T_NT), 0, 0, d_ids);
      if (hipMemcpy(ids.data(), d_ids, wg * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
        grouped = 1;
        for (int b = 0; b < wg; ++b)
          if (ids[b] != b % kXcds) grouped = 0;
      }
      (void)hipFree(d_ids);
    }
    (void)hipGetLastError();
    if (const char* e = getenv("VP3D_TAIL_FLAT_BARRIER"))
      if (e[0] == '1') grouped = 0;
    g_grid[dev].grouped = grouped;
    g_grid[dev].wgs = wg;
  }
  (void)fwd;
  *wgs = g_grid[dev].wgs;
  return VP3D_OK;
}
int tail_grouped() {
  int dev = 0;
  return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) ? g_grid[dev].grouped : 0;
}

// K-slices of an [M, N, K] GEMM on `wgs` resident 128x128-tile workgroups: the fewest rounds first, then the least partial
// traffic.  Cost in microseconds: a K-tile of a 128x128 tile takes ~0.66 us of a workgroup slot when two workgroups share
// a CU (0.86 alone), a tile's prologue + epilogue ~2.5 us, partial tiles are written and read back once (~4 TB/s each way).
int tail_splits(int64_t M, int64_t N, int64_t K, int wgs) {
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const int nkt = (int)(K / 32);
  double best = 1e30;
  int best_s = 1;
  for (int s = 1; s <= 16; ++s) {
    if (s > 1 && nkt / s < 4) break;
    const int64_t items = tiles * s;
    const int64_t rounds = (items + wgs - 1) / wgs;
    const double nk = (double)((nkt + s - 1) / s);
    const double per = items * 2 > wgs ? 0.66 : 0.86;
    double cost = (double)rounds * (nk * per + 2.5);
    if (s > 1) cost += (double)s * (double)M * (double)N * 8.0 / 4.0e6;
    if (cost < best * 0.97) {
      best = cost;
      best_s = s;
    }
  }
  return best_s;
}

}  // namespace

int tail_max_layers() { return kMaxTail; }
int tail_sync_bytes() { return kSyncWords * (int)sizeof(unsigned); }
int tail_barrier_grouped() { return tail_grouped(); }

int launch_tail_fwd(hipStream_t s, const vp3d_tail_fwd* d) {
  VP3D_REQUIRE(d && d->layers && d->n_layers >= 2 && d->n_layers <= kMaxTail && d->n_layers % 2 == 0,
               "tail_fwd: 2..%d layers, whole blocks (strided conv + 1x1 conv)", kMaxTail);
  VP3D_REQUIRE(d->C > 0 && d->C % 64 == 0 && d->x0 && d->x0_bound && d->part && d->sync && aligned16(d->x0) && aligned16(d->part),
               "tail_fwd: bad argument (C %% 64 == 0, 16-byte aligned buffers)");
  int wgs = 0;
  int rc = tail_grid(true, &wgs);
  if (rc != VP3D_OK) return rc;
  TailFwdArgs a{};
  a.n_layers = d->n_layers;
  a.C = d->C;
  a.x0 = (const float*)d->x0;
  a.x0_bound = d->x0_bound;
  a.part = d->part;
  a.eps = d->eps;
  a.momentum = d->momentum;
  a.momentum_dev = d->momentum_dev;
  a.sync = (unsigned*)d->sync;
  a.trace = (unsigned long long*)d->trace;
  a.grouped = tail_grouped();
  int64_t need = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    const vp3d_tail_fwd_layer& u = d->layers[l];
    FwdLayer& F = a.L[l];
    VP3D_REQUIRE(u.M > 0 && u.M <= 65536 && (u.taps == 1 || u.taps == 3) && u.wf && u.w_bound && u.gamma && u.beta && u.y && u.coef &&
                     u.a && u.a_bound && aligned16(u.wf) && aligned16(u.y) && aligned16(u.a) && aligned16(u.coef),
                 "tail_fwd: layer %d: bad argument", l);
    VP3D_REQUIRE((l % 2 == 0) == (u.res_start < 0) && (l % 2 == 0 || (u.taps == 1 && u.res_start < d->layers[l - 1].taps)),
                 "tail_fwd: layer %d: blocks are (strided conv, 1x1 conv + residual)", l);
    VP3D_REQUIRE(l == 0 || (int64_t)d->layers[l - 1].M == u.M * u.taps, "tail_fwd: layer %d: rows do not chain", l);
    VP3D_REQUIRE(u.M * (int64_t)u.taps * d->C * 4 < ((int64_t)1 << 31), "tail_fwd: layer %d: operand of 2 GiB or more", l);
    VP3D_REQUIRE(u.a_t == nullptr || ((u.taps_at == 1 || u.taps_at == 3) && u.M % u.taps_at == 0 &&
                                      u.ld_at >= (u.M / u.taps_at + 63) / 64 * 64 && u.ld_at % 8 == 0 && aligned16(u.a_t)),
                 "tail_fwd: layer %d: transposed copy geometry", l);
    F.wf = (const float*)u.wf; F.w_bound = u.w_bound; F.gamma = u.gamma; F.beta = u.beta;
    F.run_mean = u.running_mean; F.run_var = u.running_var; F.nbt = u.num_batches_tracked;
    F.y = u.y; F.coef = u.coef; F.a = (float*)u.a; F.a_bound = u.a_bound; F.a_f32 = u.a_f32; F.a_t = (float*)u.a_t;
    F.bits = u.act_bits;
    F.drop = make_drop(u.drop);
    F.M = (int)u.M; F.taps = u.taps; F.res_start = u.res_start;
    F.taps_at = u.a_t ? u.taps_at : 1; F.ld_at = (int)u.ld_at;
    const int K = u.taps * d->C, nkt = K / 32;
    F.splits = u.splits > 0 ? u.splits : tail_splits(u.M, d->C, K, wgs);
    if (F.splits > nkt) F.splits = nkt;
    F.kt_per_split = (nkt + F.splits - 1) / F.splits;
    const int64_t w = (int64_t)F.splits * u.M * d->C;
    if (w > need) need = w;
  }
  VP3D_REQUIRE(d->part_floats >= need, "tail_fwd: the workspace needs %lld floats (vp3d_tail_workspace)", (long long)need);
  if (hipMemsetAsync(d->sync, 0, kSyncWords * sizeof(unsigned), s) != hipSuccess) {
    (void)hipGetLastError();
    set_error("tail_fwd: hipMemsetAsync failed");
    return VP3D_E_INVALID;
  }
  VP3D_LAUNCH(k_tail_fwd, dim3(wgs), dim3(T_NT), 0, s, a);
  return check_launch("tail_fwd");
}

int launch_tail_bwd(hipStream_t s, const vp3d_tail_bwd* d) {
  VP3D_REQUIRE(d && d->layers && d->n_layers >= 2 && d->n_layers <= kMaxTail && d->n_layers % 2 == 0,
               "tail_bwd: 2..%d layers, whole blocks", kMaxTail);
  VP3D_REQUIRE(d->C > 0 && d->C % 64 == 0 && d->dpart && d->wpart && d->dx0 && d->dx0_bound && d->sync && d->p >= 0.f && d->p < 1.f &&
                   aligned16(d->dpart) && aligned16(d->wpart) && aligned16(d->dx0),
               "tail_bwd: bad argument");
  int wgs = 0;
  int rc = tail_grid(false, &wgs);
  if (rc != VP3D_OK) return rc;
  TailBwdArgs a{};
  a.n_layers = d->n_layers;
  a.C = d->C;
  a.inv_keep = 1.0f / (1.0f - d->p);
  a.dpart = d->dpart;
  a.wpart = d->wpart;
  a.dx0 = d->dx0;
  a.dx0_bound = d->dx0_bound;
  a.sync = (unsigned*)d->sync;
  a.trace = (unsigned long long*)d->trace;
  a.grouped = tail_grouped();
  int64_t need_d = 0, need_w = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    const vp3d_tail_bwd_layer& u = d->layers[l];
    BwdLayer& B = a.L[l];
    VP3D_REQUIRE(u.M > 0 && u.M <= 65536 && (u.taps == 1 || u.taps == 3) && u.wd && u.w_bound && u.y && u.coef && u.act_bits && u.x_t &&
                     u.x_bound && u.go && u.go_bound && u.dy && u.dy_t && u.dy_bound && u.dgamma && u.dbeta && u.dw &&
                     aligned16(u.wd) && aligned16(u.y) && aligned16(u.x_t) && aligned16(u.go) && aligned16(u.dy) && aligned16(u.dy_t) &&
                     aligned16(u.dw) && aligned16(u.coef),
                 "tail_bwd: layer %d: null or unaligned pointer", l);
    VP3D_REQUIRE((l % 2 == 0) == (u.res_start < 0) && (l % 2 == 0 || u.taps == 1), "tail_bwd: layer %d: block structure", l);
    VP3D_REQUIRE(l == 0 || (int64_t)d->layers[l - 1].M == u.M * u.taps, "tail_bwd: layer %d: rows do not chain", l);
    const int64_t ldt = (u.M + 63) / 64 * 64;
    VP3D_REQUIRE(u.ld_dyt == ldt && u.ld_xt == ldt && u.M * (int64_t)u.taps * d->C * 4 < ((int64_t)1 << 31) &&
                     (int64_t)u.taps * d->C * ldt * 4 < ((int64_t)1 << 31),
                 "tail_bwd: layer %d: transposed operands must have the pitch roundup(M, 64) = %lld and stay below 2 GiB", l, (long long)ldt);
    B.wd = (const float*)u.wd; B.w_bound = u.w_bound; B.y = u.y; B.coef = u.coef; B.bits = u.act_bits;
    B.x_t = (const float*)u.x_t; B.x_bound = u.x_bound; B.go = u.go; B.go_bound = u.go_bound;
    B.dy = (float*)u.dy; B.dy_t = (float*)u.dy_t; B.dy_bound = u.dy_bound; B.dgamma = u.dgamma; B.dbeta = u.dbeta; B.dw = u.dw;
    B.M = (int)u.M; B.taps = u.taps; B.res_start = u.res_start; B.ld_xt = (int)u.ld_xt; B.ld_dyt = (int)u.ld_dyt;
    const int nkt_d = d->C / 32, nkt_w = (int)(ldt / 32);
    B.splits_d = u.splits_d > 0 ? u.splits_d : tail_splits(u.M, (int64_t)u.taps * d->C, d->C, wgs);
    if (B.splits_d > nkt_d) B.splits_d = nkt_d;
    B.ktps_d = (nkt_d + B.splits_d - 1) / B.splits_d;
    B.splits_w = u.splits_w > 0 ? u.splits_w : tail_splits(d->C, (int64_t)u.taps * d->C, ldt, wgs);
    if (B.splits_w > nkt_w) B.splits_w = nkt_w;
    B.ktps_w = (nkt_w + B.splits_w - 1) / B.splits_w;
    const int64_t wd = (int64_t)B.splits_d * u.M * u.taps * d->C, ww = (int64_t)B.splits_w * d->C * u.taps * d->C;
    if (wd > need_d) need_d = wd;
    if (ww > need_w) need_w = ww;
  }
  VP3D_REQUIRE(d->dpart_floats >= need_d && d->wpart_floats >= need_w,
               "tail_bwd: the workspaces need %lld + %lld floats (vp3d_tail_workspace)", (long long)need_d, (long long)need_w);
  if (hipMemsetAsync(d->sync, 0, kSyncWords * sizeof(unsigned), s) != hipSuccess) {
    (void)hipGetLastError();
    set_error("tail_bwd: hipMemsetAsync failed");
    return VP3D_E_INVALID;
  }
  VP3D_LAUNCH(k_tail_bwd, dim3(wgs), dim3(T_NT), 0, s, a);
  return check_launch("tail_bwd");
}

// workspace sizes (floats) for a tail of n layers with the given (M, taps) per layer, as the launchers will plan it
int tail_workspace(int32_t C, int32_t n_layers, const int64_t* M, const int32_t* taps, int64_t* fwd_floats, int64_t* dpart_floats,
                   int64_t* wpart_floats) {
  int wgs = 0;
  int rc = tail_grid(true, &wgs);
  if (rc != VP3D_OK) return rc;
  int64_t f = 0, dd = 0, ww = 0;
  for (int l = 0; l < n_layers; ++l) {
    const int64_t K = (int64_t)taps[l] * C, ldt = (M[l] + 63) / 64 * 64;
    int s = tail_splits(M[l], C, K, wgs);
    if (s > K / 32) s = (int)(K / 32);
    if ((int64_t)s * M[l] * C > f) f = (int64_t)s * M[l] * C;
    int sd = tail_splits(M[l], K, C, wgs);
    if (sd > C / 32) sd = C / 32;
    if ((int64_t)sd * M[l] * K > dd) dd = (int64_t)sd * M[l] * K;
    int sw = tail_splits(C, K, ldt, wgs);
    if (sw > ldt / 32) sw = (int)(ldt / 32);
    if ((int64_t)sw * C * K > ww) ww = (int64_t)sw * C * K;
  }
  *fwd_floats = f;
  *dpart_floats = dd;
  *wpart_floats = ww;
  return VP3D_OK;
}

}  // namespace vp3d
