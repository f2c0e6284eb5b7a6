// fp32-MFMA implicit-GEMM kernels for the temporal convolutions (gfx950 / CDNA4).
//
// One 128x128 output tile per 256-thread workgroup (4 waves, each a 64x64 sub-tile = 2x2 blocks of
// v_mfma_f32_32x32x2_f32, 64 accumulator VGPRs), K consumed in 32-wide tiles that are DMA'd straight from
// HBM/L2 into LDS with global_load_lds (16 B per lane, no VGPR round trip) into a 2-stage ring, so the next
// K-tile streams in underneath the 64 MFMAs (4096 matrix-pipe cycles) of the current one.  Two workgroups are
// resident per CU (2 x 64 KiB LDS), so one wave's barrier wait is covered by the other workgroup's MFMAs.
//
// Operand images in LDS
//   "KC" (k contiguous: gathered activation rows, forward-packed weight rows): [128 rows][32 k] floats, 128 B
//        rows, 16-B chunks XOR-swizzled by ((row>>1)&7) so the ds_read_b128 fragment reads are conflict-free.
//        Because global_load_lds writes LDS lane-linearly, the swizzle is applied to the per-lane SOURCE
//        address (which chunk of the row a lane fetches) and again on the read.
//   "MC" (m/n contiguous: operands whose reduction index is the row): [32 k][128 cols] floats, read with
//        conflict-free ds_read_b32.
// K-permutation: a lane in wave half h supplies k = 8*kg + 4*h + j at MFMA step j of k-group kg, for BOTH
// operands, so one ds_read_b128 feeds four MFMAs.  The sum over k is the same set of products.
//
// Rows of the A operand are gathered per output row (temporal taps / dilation / stride), out-of-range taps and
// ragged tiles read a page of zeros, so no im2col buffer exists.
#include "vp3d_internal.h"

namespace vp3d {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NTHREADS = 256;
constexpr int TILE_B = BM * BK * 4;       // 16 KiB per operand tile
constexpr int STAGE_B = 2 * TILE_B;       // A + B
constexpr int TAB_OFF = 2 * STAGE_B;      // two stages
constexpr int SMEM_B = TAB_OFF + 2 * BM * 4;

__device__ __forceinline__ void glds16(const float* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------
// MFMA core: one 32-deep K tile for this wave's 64x64 sub-tile.
// ---------------------------------------------------------------------------------------------------------
template <bool KC>
__device__ __forceinline__ void load_frag(const char* __restrict__ sT, int w_half, int kg, int h, int cl,
                                          float (&f)[2][4]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (KC) {
      const int row = w_half * 64 + i * 32 + cl;
      const int chunk = (kg * 2 + h) ^ ((row >> 1) & 7);
      const f32x4 v = *reinterpret_cast<const f32x4*>(sT + row * 128 + chunk * 16);
      f[i][0] = v[0]; f[i][1] = v[1]; f[i][2] = v[2]; f[i][3] = v[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        f[i][j] = *reinterpret_cast<const float*>(sT + (kg * 8 + 4 * h + j) * 512 + (w_half * 64 + i * 32 + cl) * 4);
    }
  }
}

// Software-pipelined over the four 8-deep k-groups: the LDS reads of group kg+1 are issued before the 16 MFMAs
// (1024 matrix-pipe cycles) of group kg, so the pipe never waits on LDS latency inside a K-tile.
template <bool A_KC, bool B_KC, bool WITH_DMA>
__device__ __forceinline__ void compute_tile(const char* __restrict__ sA, const char* __restrict__ sB,
                                             f32x16 (&acc)[2][2], int wm, int wn, int lane) {
  const int h = lane >> 5, cl = lane & 31;
  float a[2][2][4], b[2][2][4];
  load_frag<A_KC>(sA, wm, 0, h, cl, a[0]);
  load_frag<B_KC>(sB, wn, 0, h, cl, b[0]);
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) {
    const int cur = kg & 1, nxt = cur ^ 1;
    if (kg < 3) {
      load_frag<A_KC>(sA, wm, kg + 1, h, cl, a[nxt]);
      load_frag<B_KC>(sB, wn, kg + 1, h, cl, b[nxt]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][j], b[cur][jn][j], acc[i][jn], 0, 0, 0);
  }
  // Pin the instruction interleave of the whole K-tile (hipcc otherwise sinks every ds_read to just before its
  // first use -- read -> wait -> 4 MFMA -> read ... -- and emits the next tile's LDS-DMA issue, ~60 VALU/SALU of
  // address arithmetic + 8 global_load_lds, as one block in front of the first MFMA).  Masks: 0x008 MFMA,
  // 0x100 DS read, 0x010 VMEM, 0x002 VALU.  Each fp32 32x32x2 MFMA occupies the matrix pipe for 64 cycles, so
  // ~10 other instructions issue for free behind it.
  constexpr int R = (A_KC ? 2 : 4) + (B_KC ? 2 : 4);   // LDS reads per k-group: b128 per 32 rows / merged read2_b32 per j
  __builtin_amdgcn_sched_group_barrier(0x100, R, 0);    // k-group 0 operands
  constexpr int kUsed0 = 0;
  // (Interleaving the next tile's 8 LDS-DMA issues into the first MFMAs with VALU/VMEM groups was tried and made
  // hipcc's solver scramble the whole tile; the DMA block stays in front of the first MFMA, with its address
  // arithmetic reduced to running pointers.)
  constexpr int kRoom0 = 16 - kUsed0;                   // MFMAs of group 0 still unscheduled
  constexpr int kPer0 = R <= kRoom0 ? 1 : 2;            // group-1 reads per MFMA slot
  constexpr int kSlots0 = (R + kPer0 - 1) / kPer0;
#pragma unroll
  for (int q = 0; q < kSlots0; ++q) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, kPer0, 0);
  }
  if constexpr (kRoom0 - kSlots0 > 0) __builtin_amdgcn_sched_group_barrier(0x008, kRoom0 - kSlots0, 0);
#pragma unroll
  for (int kg = 1; kg < 3; ++kg) {                      // next group's reads ride the first MFMAs of this one
#pragma unroll
    for (int q = 0; q < R; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if constexpr (16 - R > 0) __builtin_amdgcn_sched_group_barrier(0x008, 16 - R, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
}

// ---------------------------------------------------------------------------------------------------------
// Epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// ---------------------------------------------------------------------------------------------------------
// The accumulators are first turned into a row-major [128][128] fp32 tile in LDS (the 64 KiB of the two operand
// stages are free after the last K-tile), then written out by a coalesced pass: a thread handles one float4 of a
// 512-B row segment per step -> 16-B global stores (and 16-B residual / bias loads) instead of 64 scattered dword
// accesses per lane.  BatchNorm slab statistics are taken from the registers before the staging.
template <bool HAS_TAB>
__device__ __forceinline__ void epilogue(const Epi& e, f32x16 (&acc)[2][2], char* smem, int m0, int n0, int wm,
                                         int wn, int tid, int lane, int M, int N, const int* tab_b,
                                         const int* tab_t, int slab_index) {
  const int h = lane >> 5, cl = lane & 31;

  if (e.stat_sum != nullptr) {
    const int cnt = min(64, M - (m0 + wm * 64));   // wave-uniform number of valid rows in this 64-row slab
    if (cnt > 0) {
#pragma unroll
      for (int jn = 0; jn < 2; ++jn) {
        const int n = n0 + wn * 64 + jn * 32 + cl;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int r = i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            s += (r < cnt) ? acc[i][jn][reg] : 0.f;
          }
        s += __shfl_xor(s, 32);
        const float mean = s / (float)cnt;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int r = i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const float d = acc[i][jn][reg] - mean;
            q += (r < cnt) ? d * d : 0.f;
          }
        q += __shfl_xor(q, 32);
        if (h == 0 && n < N) {
          e.stat_sum[(int64_t)slab_index * N + n] = s;
          e.stat_m2[(int64_t)slab_index * N + n] = q;
        }
      }
    }
  }

  __syncthreads();                               // every wave is done reading the operand stages
  float* ct = reinterpret_cast<float*>(smem);    // [128][128]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        ct[r * BN + wn * 64 + jn * 32 + cl] = acc[i][jn][reg];
      }
  __syncthreads();

  const int col = (tid & 31) * 4;
  const int n = n0 + col;
  const bool fused = e.ab_y != nullptr;          // launcher guarantees: vec, N % 128 == 0 (no ragged columns, so
  if (n >= N) return;                            // nobody leaves before the barriers of the fused reduction)
  const int rc = n - e.r_col0;
  if (e.vec) {       // N, ldc, pitches, r_col0/r_cols multiples of 4 and 16-B aligned bases (checked by the launcher)
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (e.bias != nullptr) bias = *reinterpret_cast<const f32x4*>(e.bias + n);
    const bool rcol_ok = e.R != nullptr && rc >= 0 && rc < e.r_cols;
    // fused activation backward of the upstream layer: per-column BN coefficients, per-half-tile partial sums
    f32x4 a_sc, a_sh, a_mu, a_is;
    f32x4 sg[2], sgx[2];
    DropP ab_drop = e.ab_drop;
    if (fused) drop_resolve(ab_drop);       // (partial-tile launches carry an uninitialised ab_drop)
    if (fused) {
      const int ch = n % e.ab_c;
      a_sc = *reinterpret_cast<const f32x4*>(e.ab_scale + ch);
      a_sh = *reinterpret_cast<const f32x4*>(e.ab_shift + ch);
      a_mu = *reinterpret_cast<const f32x4*>(e.ab_mean + ch);
      a_is = *reinterpret_cast<const f32x4*>(e.ab_invstd + ch);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        sg[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        sgx[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (fused) {
      // The 8 upstream-activation loads (and residual loads) of each 64-row half are issued up front: the fused
      // epilogue is latency-bound otherwise (16 dependent global loads per thread), and an epilogue is only hidden
      // while the co-resident workgroup is still in its main loop.  (8, not 16: 16 in flight spill registers.)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x4 yv[8], rv[8];
        int64_t offs[8];
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) {
          const int r = (hh * 8 + i8) * 8 + (tid >> 5);
          const bool live = m0 + r < M;
          const int b = HAS_TAB ? tab_b[r] : 0;
          const int t = HAS_TAB ? tab_t[r] : m0 + r;
          offs[i8] = live ? (int64_t)b * e.c_bpitch + (int64_t)t * e.ldc + n : (int64_t)-1;
          yv[i8] = f32x4{0.f, 0.f, 0.f, 0.f};
          rv[i8] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (live) {
            yv[i8] = *reinterpret_cast<const f32x4*>(e.ab_y + offs[i8]);
            if (rcol_ok) {
              const int tr = t * e.r_stride + e.r_off;
              if ((unsigned)tr < (unsigned)e.r_t)
                rv[i8] = *reinterpret_cast<const f32x4*>(e.R + (int64_t)b * e.r_bpitch + (int64_t)tr * e.r_ld + rc);
            }
          }
        }
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) {
          if (offs[i8] < 0) continue;
          const int r = (hh * 8 + i8) * 8 + (tid >> 5);
          f32x4 v = *reinterpret_cast<const f32x4*>(ct + r * BN + col) + bias;
          if (e.relu) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = v[c] < 0.f ? 0.f : v[c];
          }
          v += rv[i8];
          float mk[4] = {1.f, 1.f, 1.f, 1.f};
          if (ab_drop.on) drop4(ab_drop, (uint64_t)(offs[i8] >> 2), mk);
          f32x4 g;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float z = fmaf(yv[i8][c], a_sc[c], a_sh[c]);
            g[c] = z > 0.f ? v[c] * mk[c] : 0.f;
            sg[hh][c] += g[c];
            sgx[hh][c] += g[c] * ((yv[i8][c] - a_mu[c]) * a_is[c]);
          }
          *reinterpret_cast<f32x4*>(e.ab_g + offs[i8]) = g;
          if (e.ab_store_v) *reinterpret_cast<f32x4*>(e.C + offs[i8]) = v;
        }
      }
    } else {
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int r = it * 8 + (tid >> 5);
        const int m = m0 + r;
        if (m >= M) continue;
        int b, t;
        if (HAS_TAB) {
          b = tab_b[r];
          t = tab_t[r];
        } else {
          b = 0;
          t = m;
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(ct + r * BN + col) + bias;
        if (e.relu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = v[c] < 0.f ? 0.f : v[c];
        }
        if (rcol_ok) {
          const int tr = t * e.r_stride + e.r_off;
          if ((unsigned)tr < (unsigned)e.r_t)
            v += *reinterpret_cast<const f32x4*>(e.R + (int64_t)b * e.r_bpitch + (int64_t)tr * e.r_ld + rc);
        }
        *reinterpret_cast<f32x4*>(e.C + (int64_t)b * e.c_bpitch + (int64_t)t * e.ldc + n) = v;
      }
    }
    if (fused) {
      // column sums over the 8 row groups of each 64-row slab: LDS tree in the (now consumed) staging tile
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);   // [2 slabs][2 sums][8 row groups][128 cols]
      const int rgp = tid >> 5;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          red[((hh * 2 + 0) * 8 + rgp) * BN + col + c] = sg[hh][c];
          red[((hh * 2 + 1) * 8 + rgp) * BN + col + c] = sgx[hh][c];
        }
      __syncthreads();
      const int hh = tid >> 7, cc = tid & (BN - 1);
      if (m0 + hh * 64 < M) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) {
          t0 += red[((hh * 2 + 0) * 8 + g8) * BN + cc];
          t1 += red[((hh * 2 + 1) * 8 + g8) * BN + cc];
        }
        const int nn = n0 + cc;
        const int64_t slab = (int64_t)(m0 >> 6) + hh;
        const int64_t p = slab * (N / e.ab_c) + nn / e.ab_c;
        e.ab_part[(p * 2 + 0) * e.ab_c + nn % e.ab_c] = t0;
        e.ab_part[(p * 2 + 1) * e.ab_c + nn % e.ab_c] = t1;
      }
    }
  } else {
    for (int it = 0; it < 16; ++it) {
      const int r = it * 8 + (tid >> 5);
      const int m = m0 + r;
      if (m >= M) continue;
      int b, t;
      if (HAS_TAB) {
        b = tab_b[r];
        t = tab_t[r];
      } else {
        b = 0;
        t = m;
      }
      const int tr = t * e.r_stride + e.r_off;
      const bool r_row_ok = e.R != nullptr && (unsigned)tr < (unsigned)e.r_t;
      const float* rrow = e.R + (int64_t)b * e.r_bpitch + (int64_t)tr * e.r_ld - e.r_col0;
      float* crow = e.C + (int64_t)b * e.c_bpitch + (int64_t)t * e.ldc;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int nn = n + c;
        if (nn >= N) continue;
        float v = ct[r * BN + col + c] + (e.bias != nullptr ? e.bias[nn] : 0.f);
        if (e.relu) v = v < 0.f ? 0.f : v;
        const int rcc = nn - e.r_col0;
        if (r_row_ok && rcc >= 0 && rcc < e.r_cols) v += rrow[nn];
        crow[nn] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Rows GEMM:  C[m][n] = sum_k A[gather(m,k)] * B     (forward: B_KC = true; dgrad: B_KC = false)
// FAST: global_load_lds path (c_src % 32 == 0, N % 128 == 0, 16-B aligned rows).  Otherwise a bounds-checked
// register-staged loader builds the identical LDS image (any shape; used for expand / shrink / odd channels).
// ---------------------------------------------------------------------------------------------------------
template <bool B_KC, bool FAST>
__global__ void __launch_bounds__(NTHREADS, 2) k_rows_gemm(const RowsGemmArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[SMEM_B];
  int* tab_b = reinterpret_cast<int*>(smem + TAB_OFF);
  int* tab_t = tab_b + BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;

  // XCD-aware tile order: workgroup id b runs on XCD b%8; give every XCD whole m-tiles (all their n-tiles
  // back to back) so the 8 column tiles of one activation row-panel share that XCD's L2.
  // Tail split: tile positions [0, pos_full) are whole tiles; the positions behind them (the last m-groups, i.e.
  // the part of the tile count that does not fill a round of 2 workgroups x 256 CUs -- or, for the small-M layers,
  // every tile) are cut into `splits` K-slices that are dispatched LAST and fill the CUs the whole tiles leave idle;
  // their raw partial tiles go to the workspace and k_splitk_finish applies the epilogue.
  // Wide outputs (dgrad of a strided conv: N = 3C = 24 column tiles) are walked in groups of 8 column tiles so
  // that the ~64 workgroups an XCD runs concurrently form an 8x8 patch (8 A panels + 8 B panels in its 4 MiB L2).
  int bid = blockIdx.x;
  int split = 0;
  const bool tail = bid >= p.pos_full;     // block-uniform
  if (tail) {
    const int j = bid - p.pos_full;
    split = j / p.tail_pos;                // slices of one K range are adjacent: they share operand panels in L2
    bid = p.pos_full + (j - split * p.tail_pos);
  }
  const int xcd = bid & 7, q = bid >> 3;
  const int gn = min(p.n_tiles, 8);
  const int m_groups = (p.m_tiles + 7) >> 3;
  const int inner = q % gn, rest = q / gn;
  const int tile_n = (rest / m_groups) * gn + inner;
  const int tile_m = (rest % m_groups) * 8 + xcd;
  if (tile_m >= p.m_tiles || tile_n >= p.n_tiles) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  if (tid < BM) {
    const int m = min(m0 + tid, p.M - 1);
    const int b = m / p.t_dst;
    tab_b[tid] = b;
    tab_t[tid] = m - b * p.t_dst;
  }
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

  const int nkt_all = (p.K + BK - 1) / BK;
  const int kt_begin = tail ? split * p.kt_per_split : 0;
  const int kt_end = tail ? min(nkt_all, kt_begin + p.kt_per_split) : nkt_all;
  const int nkt = max(0, kt_end - kt_begin);

  if (FAST) {
    // per-thread staging assignments: KC tile piece (w,i) = rows (w*4+i)*8 .. +8, lane -> (row, 16-B chunk).
    // Running pointers: every K-tile advances them by a wave-uniform increment (plus a jump at tap boundaries),
    // so the per-tile address arithmetic is one 64-bit add (+ a validity select) per LDS-DMA piece.
    const int tap0 = (kt_begin * BK) / p.c_src;
    int c0 = kt_begin * BK - tap0 * p.c_src;
    int a_t[4];
    const float* a_ptr[4];
    int zoff[4];
    const float* b_ptr[4];
    int b_inc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (w * 4 + i) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((r >> 1) & 7);
      const int b = tab_b[r], t = tab_t[r];
      a_t[i] = t * p.t_stride + p.t_off + tap0 * p.tap_step;
      a_ptr[i] = p.A + ((int64_t)b * p.t_src + a_t[i]) * p.lda + c0 + chunk * 4;
      zoff[i] = chunk * 4;
      if (B_KC) {
        const bool ok = (n0 + r) < p.N;    // weight rows beyond N (e.g. shrink: N = 51) come from the zero page
        b_ptr[i] = ok ? p.B + (int64_t)(n0 + r) * p.ldb + (int64_t)kt_begin * BK + chunk * 4 : p.zeros + chunk * 4;
        b_inc[i] = ok ? BK : 0;
      } else {
        const int kr = (w * 4 + i) * 2 + (lane >> 5);
        b_ptr[i] = p.B + (int64_t)(c0 + kr) * p.ldb + (int64_t)tap0 * p.b_tap_stride + n0 + (lane & 31) * 4;
        b_inc[i] = 0;
      }
    }
    const int64_t a_jump = (int64_t)p.tap_step * p.lda - p.c_src + BK;            // at a tap boundary
    const int64_t b_step = (int64_t)BK * p.ldb;                                    // NN: next 32 k-rows
    const int64_t b_jump = (int64_t)p.b_tap_stride - (int64_t)(p.c_src - BK) * p.ldb;

    auto issue = [&](int stage, bool live) {          // `live` = false: harmless zero-page DMA (keeps it branch-free)
      char* sA = smem + stage * STAGE_B;
      char* sB = sA + TILE_B;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = live && (unsigned)a_t[i] < (unsigned)p.t_src;
        const float* g = ok ? a_ptr[i] : (p.zeros + zoff[i]);
        glds16(g, sA + (w * 4 + i) * 1024);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* g = live ? b_ptr[i] : (p.zeros + zoff[i]);
        glds16(g, sB + (w * 4 + i) * 1024);
      }
      c0 += BK;
      const bool wrap = c0 >= p.c_src;                // wave-uniform
      if (wrap) c0 = 0;
      const int64_t a_inc = wrap ? a_jump : (int64_t)BK;
      const int t_inc = wrap ? p.tap_step : 0;
      const int64_t bb = B_KC ? 0 : (wrap ? b_jump : b_step);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a_ptr[i] += a_inc;
        a_t[i] += t_inc;
        b_ptr[i] += bb + b_inc[i];
      }
    };

    if (nkt > 0) issue(0, true);
    for (int it = 0; it < nkt; ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces of tile `it` have landed
      __syncthreads();   // ... and everyone's have, and everyone left stage (it+1)&1
      issue((it + 1) & 1, it + 1 < nkt);
      const char* sA = smem + (it & 1) * STAGE_B;
      compute_tile<true, B_KC, true>(sA, sA + TILE_B, acc, wm, wn, lane);
    }
  } else {
    for (int it = 0; it < nkt; ++it) {
      char* sA = smem + (it & 1) * STAGE_B;
      char* sB = sA + TILE_B;
      const int k0 = (kt_begin + it) * BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // A (KC image)
        const int r = (w * 4 + i) * 8 + (lane >> 3);
        const int pchunk = lane & 7;
        const int chunk = pchunk ^ ((r >> 1) & 7);
        const int b = tab_b[r], t = tab_t[r];
        const int t0 = t * p.t_stride + p.t_off;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = k0 + chunk * 4 + e;
          float x = 0.f;
          if (k < p.K) {
            const int tp = k / p.c_src;
            const int c = k - tp * p.c_src;
            const int st = t0 + tp * p.tap_step;
            if ((unsigned)st < (unsigned)p.t_src) x = p.A[((int64_t)b * p.t_src + st) * p.lda + c];
          }
          v[e] = x;
        }
        *reinterpret_cast<f32x4*>(sA + r * 128 + pchunk * 16) = v;
        // B
        if (B_KC) {
          const int nn = n0 + r;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = k0 + chunk * 4 + e;
            v[e] = (nn < p.N && k < p.K) ? p.B[(int64_t)nn * p.ldb + k] : 0.f;
          }
          *reinterpret_cast<f32x4*>(sB + r * 128 + pchunk * 16) = v;
        } else {
          const int kr = (w * 4 + i) * 2 + (lane >> 5);
          const int k = k0 + kr;
          const int tp = k / p.c_src;
          const int co = k - tp * p.c_src;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int nn = n0 + (lane & 31) * 4 + e;
            v[e] = (k < p.K && nn < p.N) ? p.B[(int64_t)co * p.ldb + (int64_t)tp * p.b_tap_stride + nn] : 0.f;
          }
          *reinterpret_cast<f32x4*>(sB + kr * 512 + (lane & 31) * 16) = v;
        }
      }
      __syncthreads();
      compute_tile<true, B_KC, false>(sA, sB, acc, wm, wn, lane);
      // the next iteration writes the other stage; two stages + one barrier per tile is race-free because a
      // wave can only be one tile ahead of the slowest wave (it must pass the barrier above).
    }
  }

  if (tail && p.splits > 1) {
    // K-slice: raw 128x128 partial tile to the workspace [split][tail position][128][128]
    Epi e;
    e.C = p.part + ((int64_t)split * p.tail_pos + (bid - p.pos_full)) * (BM * BN);
    e.c_bpitch = 0;
    e.ldc = BN;
    e.bias = nullptr;
    e.relu = 0;
    e.R = nullptr;
    e.r_bpitch = 0;
    e.r_ld = e.r_t = e.r_stride = e.r_off = e.r_col0 = e.r_cols = 0;
    e.stat_sum = e.stat_m2 = nullptr;
    e.ab_y = nullptr;
    e.ab_drop.on = 0;
    e.ab_drop.off_ptr = nullptr;
    e.vec = 1;                                   // workspace tiles are 16-B aligned
    epilogue<false>(e, acc, smem, 0, 0, wm, wn, tid, lane, BM, BN, nullptr, nullptr, 0);
    return;
  }
  epilogue<true>(p.epi, acc, smem, m0, n0, wm, wn, tid, lane, p.M, p.N, tab_b, tab_t, tile_m * 2 + wm);
}

// Finish of the K-sliced tail tiles: sum the partial tiles and apply the fused epilogue (bias / ReLU / residual /
// 64-row-slab BatchNorm statistics).  blockIdx.x = tail position, blockIdx.y = (64-row half, 64-column half) of its
// tile: thread (rg, cq) owns rows rg, rg+16, rg+32, rg+48 of the slab and the float4 column group cq; 16-B loads,
// LDS reduction for the statistics.
__global__ void __launch_bounds__(256) k_splitk_finish(const float* __restrict__ part, int splits, int pos_full,
                                                       int tail_pos, int m_tiles, int n_tiles, int M, int N, int vec,
                                                       int t_dst, const Epi e) {
  __shared__ float red[16][64];
  __shared__ float mean_s[64];
  // same position -> tile map as k_rows_gemm
  const int bid = pos_full + blockIdx.x;
  const int xcd = bid & 7, q = bid >> 3;
  const int gn = min(n_tiles, 8);
  const int m_groups = (m_tiles + 7) >> 3;
  const int inner = q % gn, rest = q / gn;
  const int tile_n = (rest / m_groups) * gn + inner;
  const int tile_m = (rest % m_groups) * 8 + xcd;
  if (tile_m >= m_tiles || tile_n >= n_tiles) return;
  const int sh = blockIdx.y >> 1, ch = blockIdx.y & 1;
  const int rg = threadIdx.x >> 4, cq = threadIdx.x & 15;
  const int n = tile_n * BN + ch * 64 + cq * 4;
  const int m_base = tile_m * BM + sh * 64;
  const int slab = m_base >> 6;
  const int cnt = min(64, M - m_base);
  if (cnt <= 0) return;
  const int64_t mat = (int64_t)tail_pos * (BM * BN);        // floats between two K-slices
  const float* tile = part + (int64_t)blockIdx.x * (BM * BN) + (sh * 64) * BN + ch * 64 + cq * 4;
  const bool nok = n < N;
  f32x4 raw[4];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) raw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the K-slices are summed in slice order; the loads of 4 slices x 4 rows are issued together (one slice per
  // iteration left every add waiting for its own load: ~1 us per slice)
  for (int sp0 = 0; sp0 < splits; sp0 += 4) {
    f32x4 v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = nok && rg + 16 * i < cnt && sp0 + u < splits;
        v[u][i] = ok ? *reinterpret_cast<const f32x4*>(tile + (rg + 16 * i) * BN + (int64_t)(sp0 + u) * mat)
                     : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) raw[i] += v[u][i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) s += raw[i];
  if (e.stat_sum != nullptr) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[rg][cq * 4 + c] = s[c];
    __syncthreads();
    if (threadIdx.x < 64) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
      mean_s[threadIdx.x] = t / (float)cnt;
      const int nn = tile_n * BN + ch * 64 + threadIdx.x;
      if (nn < N) e.stat_sum[(int64_t)slab * N + nn] = t;
    }
    __syncthreads();
    f32x4 q2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (rg + 16 * i < cnt) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float d = raw[i][c] - mean_s[cq * 4 + c];
          q2[c] += d * d;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) red[rg][cq * 4 + c] = q2[c];
    __syncthreads();
    if (threadIdx.x < 64) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
      const int nn = tile_n * BN + ch * 64 + threadIdx.x;
      if (nn < N) e.stat_m2[(int64_t)slab * N + nn] = t;
    }
  }
  const bool fused = e.ab_y != nullptr;          // launcher guarantees vec and N % 128 == 0 -> nok for everyone
  if (!nok) return;
  f32x4 a_sc, a_sh, a_mu, a_is;
  f32x4 sgv = {0.f, 0.f, 0.f, 0.f}, sgxv = {0.f, 0.f, 0.f, 0.f};
  DropP ab_drop2 = e.ab_drop;
  if (fused) drop_resolve(ab_drop2);
  if (fused) {
    const int chn = n % e.ab_c;
    a_sc = *reinterpret_cast<const f32x4*>(e.ab_scale + chn);
    a_sh = *reinterpret_cast<const f32x4*>(e.ab_shift + chn);
    a_mu = *reinterpret_cast<const f32x4*>(e.ab_mean + chn);
    a_is = *reinterpret_cast<const f32x4*>(e.ab_invstd + chn);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = rg + 16 * i;
    if (r >= cnt) continue;
    const int m = m_base + r;
    const int b = m / t_dst;
    const int t = m - b * t_dst;
    const int tr = t * e.r_stride + e.r_off;
    const bool r_row_ok = e.R != nullptr && (unsigned)tr < (unsigned)e.r_t;
    float* crow = e.C + (int64_t)b * e.c_bpitch + (int64_t)t * e.ldc;
    const float* rrow = e.R + (int64_t)b * e.r_bpitch + (int64_t)tr * e.r_ld - e.r_col0;
    if (vec) {      // N, ldc, r_ld, r_col0 multiples of 4 and 16-B aligned bases: whole float4 in or out of range
      f32x4 v = raw[i];
      if (e.bias != nullptr) v += *reinterpret_cast<const f32x4*>(e.bias + n);
      if (e.relu) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = v[c] < 0.f ? 0.f : v[c];
      }
      const int rc = n - e.r_col0;
      if (r_row_ok && rc >= 0 && rc < e.r_cols) v += *reinterpret_cast<const f32x4*>(rrow + n);
      if (fused) {
        const int64_t off = (int64_t)b * e.c_bpitch + (int64_t)t * e.ldc + n;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(e.ab_y + off);
        float mk[4] = {1.f, 1.f, 1.f, 1.f};
        if (ab_drop2.on) drop4(ab_drop2, (uint64_t)(off >> 2), mk);
        f32x4 g;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float z = fmaf(yv[c], a_sc[c], a_sh[c]);
          g[c] = z > 0.f ? v[c] * mk[c] : 0.f;
          sgv[c] += g[c];
          sgxv[c] += g[c] * ((yv[c] - a_mu[c]) * a_is[c]);
        }
        *reinterpret_cast<f32x4*>(e.ab_g + off) = g;
        if (e.ab_store_v) *reinterpret_cast<f32x4*>(crow + n) = v;
      } else {
        *reinterpret_cast<f32x4*>(crow + n) = v;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int nn = n + c;
        if (nn >= N) continue;
        float v = raw[i][c] + (e.bias != nullptr ? e.bias[nn] : 0.f);
        if (e.relu) v = v < 0.f ? 0.f : v;
        const int rc = nn - e.r_col0;
        if (r_row_ok && rc >= 0 && rc < e.r_cols) v += rrow[nn];
        crow[nn] = v;
      }
    }
  }
  if (fused) {                                   // per-slab column sums over the 16 row groups (fixed order)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) red[rg][cq * 4 + c] = sgv[c];
    __syncthreads();
    float t0 = 0.f, t1 = 0.f;
    if (threadIdx.x < 64) {
#pragma unroll
      for (int g = 0; g < 16; ++g) t0 += red[g][threadIdx.x];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) red[rg][cq * 4 + c] = sgxv[c];
    __syncthreads();
    if (threadIdx.x < 64) {
#pragma unroll
      for (int g = 0; g < 16; ++g) t1 += red[g][threadIdx.x];
      const int nn = tile_n * BN + ch * 64 + threadIdx.x;
      const int64_t p = (int64_t)slab * (N / e.ab_c) + nn / e.ab_c;
      e.ab_part[(p * 2 + 0) * e.ab_c + nn % e.ab_c] = t0;
      e.ab_part[(p * 2 + 1) * e.ab_c + nn % e.ab_c] = t1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Reduction GEMM (wgrad):  C[i][n] = sum_m G[m][i] * X[gather(m, tap(n))][ci(n)]   both operands "MC".
// ---------------------------------------------------------------------------------------------------------
template <bool FAST>
__global__ void __launch_bounds__(NTHREADS, 2) k_red_gemm(const RedGemmArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[TAB_OFF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;

  const int bid = blockIdx.x;
  const int split = bid % p.splits;
  const int q = bid / p.splits;
  const int tile_n = q % p.n_tiles;
  const int tile_m = q / p.n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int nkt_all = (p.Mred + BK - 1) / BK;
  const int kt_begin = split * p.kt_per_split;
  const int kt_end = min(nkt_all, kt_begin + p.kt_per_split);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

  const int colchunk = lane & 31;
  if (FAST) {
    const int tap = n0 / p.c_x;
    const int ci0 = n0 - tap * p.c_x;
    const int tap_t = tap * p.tap_step + p.t_off;
    const float* zsrc = p.zeros + colchunk * 4;
    // running (m, t, pointers) per staged row: each K-tile advances m by 32 rows; (b,t) and the gathered x-row
    // pointer advance by per-launch constants plus a per-lane wrap correction -> no division / 64-bit multiply
    const int adv_b = BK / p.t_dst, adv_t = BK - adv_b * p.t_dst;
    const int64_t x_inc = ((int64_t)adv_b * p.t_src + (int64_t)adv_t * p.t_stride) * p.ldx;
    const int64_t x_wrap = ((int64_t)p.t_src - (int64_t)p.t_dst * p.t_stride) * p.ldx;   // extra when t wraps
    int rm[4], rt[4];
    const float* g_ptr[4];
    const float* x_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = (w * 4 + i) * 2 + (lane >> 5);
      rm[i] = kt_begin * BK + kr;
      const int rb = rm[i] / p.t_dst;
      rt[i] = rm[i] - rb * p.t_dst;
      g_ptr[i] = p.G + (int64_t)rm[i] * p.ldg + m0 + colchunk * 4;
      x_ptr[i] = p.X + ((int64_t)rb * p.t_src + (int64_t)rt[i] * p.t_stride + tap_t) * p.ldx + ci0 + colchunk * 4;
    }
    const int64_t g_step = (int64_t)BK * p.ldg;

    auto issue = [&](int stage, bool live) {
      char* sA = smem + stage * STAGE_B;
      char* sB = sA + TILE_B;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool mok = live && rm[i] < p.Mred;
        const int st = rt[i] * p.t_stride + tap_t;
        const bool xok = mok && (unsigned)st < (unsigned)p.t_src;
        const float* ga = mok ? g_ptr[i] : zsrc;
        const float* gb = xok ? x_ptr[i] : zsrc;
        glds16(ga, sA + (w * 4 + i) * 1024);
        glds16(gb, sB + (w * 4 + i) * 1024);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rm[i] += BK;
        g_ptr[i] += g_step;
        rt[i] += adv_t;
        const bool wrapped = rt[i] >= p.t_dst;
        rt[i] -= wrapped ? p.t_dst : 0;
        x_ptr[i] += x_inc + (wrapped ? x_wrap : (int64_t)0);
      }
    };

    if (kt_begin < kt_end) {
      issue(0, true);
      const int n_it = kt_end - kt_begin;
      for (int it = 0; it < n_it; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue((it + 1) & 1, it + 1 < n_it);
        const char* sA = smem + (it & 1) * STAGE_B;
        compute_tile<false, false, true>(sA, sA + TILE_B, acc, wm, wn, lane);
      }
    }
  } else {
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      char* sA = smem + ((kt - kt_begin) & 1) * STAGE_B;
      char* sB = sA + TILE_B;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kr = (w * 4 + i) * 2 + (lane >> 5);
        const int m = kt * BK + kr;
        const bool mok = m < p.Mred;
        const int b = m / p.t_dst;
        const int t = m - b * p.t_dst;
        f32x4 va, vb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int io = m0 + colchunk * 4 + e;
          va[e] = (mok && io < p.Mo) ? p.G[(int64_t)m * p.ldg + io] : 0.f;
          const int nn = n0 + colchunk * 4 + e;
          float x = 0.f;
          if (mok && nn < p.N) {
            const int tp = nn / p.c_x;
            const int ci = nn - tp * p.c_x;
            const int st = t * p.t_stride + tp * p.tap_step + p.t_off;
            if ((unsigned)st < (unsigned)p.t_src) x = p.X[((int64_t)b * p.t_src + st) * p.ldx + ci];
          }
          vb[e] = x;
        }
        *reinterpret_cast<f32x4*>(sA + kr * 512 + colchunk * 16) = va;
        *reinterpret_cast<f32x4*>(sB + kr * 512 + colchunk * 16) = vb;
      }
      __syncthreads();
      compute_tile<false, false, false>(sA, sB, acc, wm, wn, lane);
    }
  }

  Epi e;
  e.C = p.C + (int64_t)split * p.Mo * p.N;
  e.c_bpitch = 0;
  e.ldc = p.N;
  e.bias = nullptr;
  e.relu = 0;
  e.R = nullptr;
  e.r_bpitch = 0;
  e.r_ld = e.r_t = e.r_stride = e.r_off = e.r_col0 = e.r_cols = 0;
  e.stat_sum = e.stat_m2 = nullptr;
  e.ab_y = nullptr;
  e.ab_drop.on = 0;
  e.ab_drop.off_ptr = nullptr;
  e.vec = (p.N % 4 == 0) ? 1 : 0;               // partial matrices are 16-B aligned allocations
  epilogue<false>(e, acc, smem, m0, n0, wm, wn, tid, lane, p.Mo, p.N, nullptr, nullptr, 0);
}

}  // namespace

// Tile positions of a rows GEMM (the k_rows_gemm order) and the K-sliced tail.
//   slots = 512: two workgroups per CU share the matrix pipes; a tile that runs alone on its CU is ~1.6x faster
//   than a co-resident one, so an under-filled last round costs less than its slot count suggests.
RowsPlan plan_rows_gemm(int M, int N, int K) {
  RowsPlan r;
  const int m_tiles = (M + BM - 1) / BM, n_tiles = (N + BN - 1) / BN;
  const int gn = n_tiles < 8 ? n_tiles : 8;
  const int m_groups = (m_tiles + 7) / 8, n_groups = (n_tiles + gn - 1) / gn;
  const int chunk = 8 * gn;                                  // positions per m-group
  r.positions = chunk * m_groups * n_groups;
  r.pos_full = r.positions;
  r.splits = 1;
  const int64_t tiles = (int64_t)m_tiles * n_tiles;
  const int nkt = (K + BK - 1) / BK;
  if (nkt < 8) return r;
  if (tiles <= 512) {
    // the whole problem is at most one round: slice every tile (cost model in units of one K-tile of matrix-pipe
    // time, ~1.7 us: a CU retires its workgroups one block-time each, every workgroup pays ~3 units of
    // prologue/epilogue, the finishing pass streams splits*tiles partial tiles at ~4 TB/s)
    double best = 1e30;
    int best_s = 1;
    for (int s = 1; s <= 8; ++s) {
      if (s > 1 && nkt / s < 4) break;
      const double rounds = (double)((tiles * s + 255) / 256);
      double cost = rounds * ((double)((nkt + s - 1) / s) + 3.0);
      if (s > 1) cost += 3.0 + (double)s * (double)M * (double)N * 4.0 / (4.0e6 * 1.7);
      if (cost < best * 0.97) {           // prefer fewer slices unless clearly better
        best = cost;
        best_s = s;
      }
    }
    if (best_s > 1) {
      r.pos_full = 0;
      r.splits = best_s;
    }
    return r;
  }
  const int rem = (int)(tiles % 512);
  if (rem == 0) return r;
  // tail = the last m-groups of the last column group, enough to hold the remainder
  const int gn_last = n_tiles - (n_groups - 1) * gn;         // valid column tiles in the last column group
  int tail_mg = 0, tail_tiles = 0;
  for (int g = m_groups - 1; g >= 0 && tail_tiles < rem; --g) {
    const int rows = (g == m_groups - 1) ? (m_tiles - 8 * (m_groups - 1)) : 8;
    tail_tiles += rows * gn_last;
    ++tail_mg;
  }
  int s = (512 + tail_tiles / 2) / tail_tiles;               // ~one round of slices
  if (s > nkt / 4) s = nkt / 4;
  if (s > 16) s = 16;
  if (s < 2) return r;
  // benefit: the idle share of the last round (a lone workgroup runs ~1.6x faster than a co-resident one);
  // cost: the finishing pass (write + read of the partial tiles at ~4 TB/s) + one launch
  const double t_tile_us = (double)nkt * 1.78 * 2.0;
  const double benefit = 0.6 * (1.0 - (double)rem / 512.0) * t_tile_us;
  const double cost = 8.0 + (double)tail_tiles * s * (BM * BN * 4.0) * 2.0 / 4.0e6;
  if (benefit < 2.0 * cost) return r;
  r.pos_full = r.positions - tail_mg * chunk;
  r.splits = s;
  return r;
}

int64_t rows_gemm_ws_floats(int M, int N, int K) {
  const RowsPlan r = plan_rows_gemm(M, N, K);
  if (r.splits <= 1) return 0;
  return (int64_t)r.splits * (r.positions - r.pos_full) * (BM * BN);
}

int rows_gemm_splits(int M, int N, int K) { return plan_rows_gemm(M, N, K).splits; }

int red_gemm_splits(int Mred, int Mo, int N) {
  // wgrad reduces over M rows into a [Mo, N] matrix (192 tiles for a 3-tap 1024x1024 conv): slice the reduction so
  // that the workgroup count is a whole number of 256-CU rounds; same cost model, the finish streams s*Mo*N floats.
  const int64_t tiles = (int64_t)((Mo + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int nkt = (Mred + BK - 1) / BK;
  double best = 1e30;
  int best_s = 1;
  for (int s = 1; s <= 64; ++s) {
    if (s > 1 && nkt / s < 4) break;
    const double rounds = (double)((tiles * s + 255) / 256);
    double cost = rounds * ((double)((nkt + s - 1) / s) + 3.0);
    if (s > 1) cost += 3.0 + (double)s * (double)Mo * (double)N * 4.0 / (4.0e6 * 1.7);
    if (cost < best * 0.97) {
      best = cost;
      best_s = s;
    }
  }
  return best_s;
}

static int epi_vec_ok(const Epi& e, int N) {
  return (N % 4 == 0) && (e.ldc % 4 == 0) && (e.c_bpitch % 4 == 0) && aligned16(e.C) &&
         (e.bias == nullptr || aligned16(e.bias)) &&
         (e.R == nullptr || (aligned16(e.R) && e.r_ld % 4 == 0 && e.r_bpitch % 4 == 0 && e.r_col0 % 4 == 0 &&
                             e.r_cols % 4 == 0));
}

int launch_rows_gemm(hipStream_t s, const RowsGemmArgs& a_in, bool b_kcontig) {
  RowsGemmArgs a = a_in;
  a.epi.vec = epi_vec_ok(a.epi, a.N);
  if (a.epi.ab_y != nullptr) {
    // with store_v == 0 the output pointer is unused (may be NULL): alignment is then judged on g_out
    const bool ok = (a.epi.ab_store_v ? a.epi.vec : ((a.N % 4 == 0) && (a.epi.ldc % 4 == 0) && (a.epi.c_bpitch % 4 == 0) &&
                                                     (a.epi.R == nullptr || (aligned16(a.epi.R) && a.epi.r_ld % 4 == 0 &&
                                                                             a.epi.r_bpitch % 4 == 0 && a.epi.r_col0 % 4 == 0 &&
                                                                             a.epi.r_cols % 4 == 0))));
    if (!ok || a.N % BN != 0) {
      set_error("rows_gemm: the fused act_bwd epilogue needs the float4 epilogue (N, pitches %% 4, 16-B aligned) and N %% 128 == 0");
      return VP3D_E_UNSUPPORTED;
    }
    a.epi.vec = 1;
  }
  const int nkt = (a.K + BK - 1) / BK;
  const RowsPlan plan = plan_rows_gemm(a.M, a.N, a.K);
  a.pos_full = plan.positions;
  a.tail_pos = 0;
  a.splits = 1;
  a.kt_per_split = nkt;
  if (plan.splits > 1 && a.part != nullptr && aligned16(a.part) &&
      a.part_floats >= (int64_t)plan.splits * (plan.positions - plan.pos_full) * (BM * BN)) {
    a.pos_full = plan.pos_full;
    a.tail_pos = plan.positions - plan.pos_full;
    a.splits = plan.splits;
    a.kt_per_split = (nkt + plan.splits - 1) / plan.splits;
  }
  const dim3 grid(a.pos_full + a.tail_pos * a.splits), block(NTHREADS);
  bool fast = (a.c_src % BK == 0) && (a.lda % 4 == 0) && (a.ldb % 4 == 0) && aligned16(a.A) && aligned16(a.B) &&
              aligned16(a.zeros);
  if (!b_kcontig) fast = fast && (a.N % BN == 0) && (a.b_tap_stride % 4 == 0);
  if (b_kcontig) {
    if (fast) VP3D_LAUNCH((k_rows_gemm<true, true>), grid, block, 0, s, a);
    else VP3D_LAUNCH((k_rows_gemm<true, false>), grid, block, 0, s, a);
  } else {
    if (fast) VP3D_LAUNCH((k_rows_gemm<false, true>), grid, block, 0, s, a);
    else VP3D_LAUNCH((k_rows_gemm<false, false>), grid, block, 0, s, a);
  }
  int rc = check_launch("rows_gemm");
  if (rc != VP3D_OK || a.splits == 1) return rc;
  VP3D_LAUNCH(k_splitk_finish, dim3(a.tail_pos, 4), dim3(256), 0, s, a.part, a.splits, a.pos_full, a.tail_pos,
                     a.m_tiles, a.n_tiles, a.M, a.N, a.epi.vec, a.t_dst, a.epi);
  return check_launch("splitk_finish");
}

int launch_red_gemm(hipStream_t s, const RedGemmArgs& a) {
  const dim3 grid(a.m_tiles * a.n_tiles * a.splits), block(NTHREADS);
  const bool fast = (a.Mo % BM == 0) && (a.c_x % BN == 0) && (a.ldg % 4 == 0) && (a.ldx % 4 == 0) &&
                    aligned16(a.G) && aligned16(a.X) && aligned16(a.zeros);
  if (fast) VP3D_LAUNCH((k_red_gemm<true>), grid, block, 0, s, a);
  else VP3D_LAUNCH((k_red_gemm<false>), grid, block, 0, s, a);
  return check_launch("red_gemm");
}

}  // namespace vp3d
