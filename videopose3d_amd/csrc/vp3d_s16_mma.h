// Building blocks of the split-fp16 ("S16", vp3d_s16.h) MFMA GEMM kernels, shared by vp3d_gemm_s16.hip (the per-launch
// kernels) and vp3d_tail_s16.hip (the persistent kernel of the small-M tail): tile configuration, LDS-DMA issue helpers and
// the pinned LDS-read / MFMA schedule of one K-tile.  Not part of the C ABI.
#pragma once
#include "vp3d_internal.h"
#include "vp3d_s16.h"

namespace vp3d {
namespace mma {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SK_AUX = 16;  // cache policy of the stream-K partial-tile traffic (16 = sc1: past the L2)
constexpr int KQ = 32;   // K and channels-per-tap are multiples of this many elements (launcher check)

// BKE: elements per K-tile (16 or 32 -> 64- or 128-byte rows in the LDS image, 1 or 2 MFMA k-steps per tile).
// With 16-element tiles a 4-deep ring fits twice in a CU's LDS (2 workgroups x 4 x 16 KiB for 128x128), keeping three
// tiles in flight per workgroup: the MFMA time of a K-tile (768 pipe cycles for a 64x64 sub-tile) is far below the
// LDS-DMA latency under load (~1.5 us), so the depth of the ring, not its width, is what feeds the matrix pipe.
// RBL_: row blocks of the LAST wave row (default: RB_ like the others).  RBL_ < RB_ gives tile heights that are not multiples
// of WM * 64 rows -- the 224 x 256 tile (Cfg<2, 4, 4, 2, 2, 32, 0, 1, 3>: wave row 0 owns rows 0..127, wave row 1 rows
// 128..223): 27,648 rows are 124 x 4 = 496 such tiles = 1.94 rounds of 256 CUs where 432 tiles of 256 x 256 = 1.69 rounds cost
// 2.  Wave w runs on SIMD w % 4 and WN == 4, so every SIMD hosts one wave of each row: the matrix pipes stay balanced.  Such a
// configuration ("MIX") writes its BatchNorm statistics per 32-row slab (a 64-row slab would straddle two tiles).
template <int WM_, int WN_, int RB_, int CB_, int NSTAGE_, int BKE_ = 32, int PIPE_ = 0, int BUF_ = 0, int RBL_ = RB_>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, RB = RB_, CB = CB_, NSTAGE = NSTAGE_, BKE = BKE_, PIPE = PIPE_, BUF = BUF_, RBL = RBL_;
  static constexpr bool MIX = RBL_ != RB_;
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int BM = ((WM - 1) * RB + RBL) * 32, BN = WN * CB * 32;
  static constexpr int SLAB = MIX ? 32 : 64;                 // rows per BatchNorm statistics slab of the epilogue
  static constexpr int ROWB = BKE * 4;                       // bytes per row per K-tile
  static constexpr int RPP = 1024 / ROWB;                    // rows per 1-KiB LDS-DMA piece (8 or 16)
  static constexpr int PA_ALL = BM / RPP;                    // A pieces per K-tile
  static constexpr int PA = (PA_ALL + NW - 1) / NW, PB = BN / RPP / NW;   // (most) pieces per wave per K-tile
  // A pieces of wave w: the first PA_BIG waves issue PA pieces each, the others PA - 1 (uniform configurations: all PA)
  static constexpr int PA_BIG = PA_ALL - NW * (PA - 1);
  __device__ static __forceinline__ int pa_count(int w) { return (!MIX || w < PA_BIG) ? PA : PA - 1; }
  __device__ static __forceinline__ int pa_first(int w) { return (!MIX || w < PA_BIG) ? w * PA : PA_BIG * PA + (w - PA_BIG) * (PA - 1); }
  // row blocks of wave row wm
  __device__ static __forceinline__ int rb_of(int wm) { return (MIX && wm == WM - 1) ? RBL : RB; }
  static constexpr int A_B = BM * ROWB, B_B = BN * ROWB, STAGE_B = A_B + B_B;
  static constexpr int TAB_OFF = NSTAGE * STAGE_B;
  static constexpr int SMEM_B = TAB_OFF + 2 * BM * 4;
  static constexpr int OCC = (SMEM_B * 2 <= 160 * 1024 && NT * 2 <= 1024) ? 2 : 1;   // workgroups per CU aimed at
  static_assert(BKE == 16 || BKE == 32, "K-tile of 16 or 32 elements");
  static_assert(BM % RPP == 0 && (MIX || BM % (RPP * NW) == 0) && BN % (RPP * NW) == 0, "DMA pieces must divide evenly over the waves");
  static_assert(NW * 32 * CB * 32 * 4 <= NSTAGE * STAGE_B, "epilogue staging must fit in the operand ring");
  static_assert(MIX || RB % 2 == 0, "64-row statistic slabs need an even number of 32-row blocks per wave");
  static_assert(!MIX || (RBL < RB && RBL >= 1 && NSTAGE == 2 && !PIPE && BUF), "MIX: 2-stage buffer-descriptor ring only");
  // 16-B chunk c of tile row r sits at chunk position c ^ swz(r): conflict-free ds_read_b128 for both row widths
  __device__ static __forceinline__ int swz(int r) { return BKE == 32 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
};

__device__ __forceinline__ void glds16(const float* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: 32-bit per-lane byte offset, out-of-range offsets deliver zeros (no zero page,
// no 64-bit pointer arithmetic or validity select per piece per K-tile)
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, int voff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}
constexpr int kOob = (int)0x80000000u;     // >= num_records of any tensor the launcher lets onto this path (< 2 GiB)

// Between a wave's writes to and reads from ITS OWN epilogue staging region: the region is private to the wave and a
// wave's LDS operations execute in order, so draining its LDS counter (plus a compiler barrier) is enough -- a workgroup
// barrier here would only make the 4-8 waves of a tile store in lockstep (measured: -0.3 % step time without it).
__device__ __forceinline__ void epi_stage_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Register-resident fragments of one K-tile of a wave's sub-tile, and the two halves of compute_tile as separate
// functions: the pipelined main loop (Cfg::PIPE) reads the fragments of tile it+1 while the MFMAs of tile it run.
template <int RB, int CB, int STEPS>
struct Frags {
  f16x8 ah[STEPS][RB], al[STEPS][RB], bh[STEPS][CB], bl[STEPS][CB];
};

template <int RB, int CB, int STEPS, int ROWB>
__device__ __forceinline__ void load_frags(const char* __restrict__ sA, const char* __restrict__ sB, int off0, int off1,
                                           Frags<RB, CB, STEPS>& f) {
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int off = s == 0 ? off0 : off1;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      f.ah[s][i] = *reinterpret_cast<const f16x8*>(sA + i * (32 * ROWB) + off);
      f.al[s][i] = *reinterpret_cast<const f16x8*>(sA + i * (32 * ROWB) + (off ^ 16));
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      f.bh[s][j] = *reinterpret_cast<const f16x8*>(sB + j * (32 * ROWB) + off);
      f.bl[s][j] = *reinterpret_cast<const f16x8*>(sB + j * (32 * ROWB) + (off ^ 16));
    }
  }
}

template <int RB, int CB, int STEPS>
__device__ __forceinline__ void mma_frags(const Frags<RB, CB, STEPS>& f, f32x16 (&acc)[RB][CB]) {
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[s][i], f.bh[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[s][i], f.bl[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[s][i], f.bh[s][j], acc[i][j], 0, 0, 0);
  }
}

// One K-tile of this wave's sub-tile: STEPS MFMA k-steps (16 elements each) x 3 products.
// RBW <= RB: the wave only owns the first RBW row blocks of its accumulator array (the 224-row tiling: wave row 1 has 3 of 4).
template <int RB, int CB, int STEPS, int ROWB, int RBW = RB>
__device__ __forceinline__ void compute_tile(const char* __restrict__ sA, const char* __restrict__ sB,
                                             f32x16 (&acc)[RB][CB], int off0, int off1) {
  f16x8 ah[STEPS][RBW], al[STEPS][RBW], bh[STEPS][CB], bl[STEPS][CB];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int off = s == 0 ? off0 : off1;
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
      ah[s][i] = *reinterpret_cast<const f16x8*>(sA + i * (32 * ROWB) + off);
      al[s][i] = *reinterpret_cast<const f16x8*>(sA + i * (32 * ROWB) + (off ^ 16));
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      bh[s][j] = *reinterpret_cast<const f16x8*>(sB + j * (32 * ROWB) + off);
      bl[s][j] = *reinterpret_cast<const f16x8*>(sB + j * (32 * ROWB) + (off ^ 16));
    }
  }
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][i], bh[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bl[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bh[s][j], acc[i][j], 0, 0, 0);
  }
  // Pin the interleave (hipcc otherwise sinks every ds_read to just before its first use: read -> wait -> 2 MFMA -> read
  // ..., exposing the LDS latency a dozen times per K-tile): all fragment reads of k-step 0 up front, the reads of k-step
  // 1 one at a time behind the first MFMAs of step 0.  Masks: 0x008 MFMA, 0x100 DS read.
  constexpr int R = 2 * (RBW + CB), MQ = 3 * RBW * CB;       // fragment reads / MFMAs per k-step
  __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
  if constexpr (STEPS == 2) {
    constexpr int NI = R < MQ ? R : MQ;
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if constexpr (R > NI) __builtin_amdgcn_sched_group_barrier(0x100, R - NI, 0);
    if constexpr (MQ > NI) __builtin_amdgcn_sched_group_barrier(0x008, MQ - NI, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MQ, 0);
  } else {
    __builtin_amdgcn_sched_group_barrier(0x008, MQ, 0);
  }
}

}  // namespace mma
}  // namespace vp3d
