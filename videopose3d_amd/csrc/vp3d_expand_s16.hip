// The expand layer of the temporal model in split-fp16 arithmetic (gfx950 / CDNA4), forward:
//     a0 = dropout(relu(BatchNorm(X W^T)))          reference model.py:74 (TemporalModelBase._forward_blocks callers :127/:176)
// X = the im2row rows of the 2-D keypoints [M][kpad] (kpad <= 128: 3 taps x 34 channels = 102 -> 128), W [N][kpad].
//
// The generic NT GEMM (vp3d_gemm_s16.hip) is built around a deep K loop; with K = 128 a 256x256 tile is four K-tiles of main
// loop and then an epilogue that writes 256 KB with the matrix pipe idle -- the layer took 105 us (statistics pass) + 247 us
// (BatchNorm + ReLU + dropout pass) for 4 % of the step's FLOPs.  Here the shape is used instead of fought:
//   * W is tiny: a wave keeps the fragments of ITS 32 output columns in registers for the whole launch (64 VGPRs for
//     K = 128) -- no B operand traffic, no B tiles in LDS;
//   * a workgroup = 8 compute waves (256 columns) + 1 loader wave walks the 64-row tiles of its row group: the loader
//     streams the X rows HBM -> LDS (LDS-DMA through a buffer descriptor, 3-deep ring of 32-KiB tiles, rows XOR-swizzled
//     against ds_read_b128 bank conflicts); only the loader has loads in flight, so its counted vmcnt waits stay exact
//     while the compute waves issue stores (mixed loads and stores retire out of order with respect to each other);
//   * a 64-row tile IS one BatchNorm statistics slab: pass 1 reduces the accumulators in registers (wave shuffles) and
//     writes the slab's (sum, M2) per column -- nothing else; pass 2 recomputes the tile (48 MFMAs per wave), applies
//     BatchNorm + ReLU + dropout and writes S16 rows + activation bits.  The conv output never exists in memory.
// Both passes are bit-identical to the generic path (same k-step order, same epilogue arithmetic): tests/test_gpu_s16.py.
#include "vp3d_internal.h"
#include "vp3d_s16.h"

namespace vp3d {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int EX_CW = 8;                        // compute waves (32 output columns each)
constexpr int EX_NT = (EX_CW + 1) * 64;         // + the loader wave
constexpr int EX_ROWS = 64;                     // rows per tile = one statistics slab
constexpr int EX_ROWB = 512;                    // bytes per staged row (kpad <= 128 S16 elements)
constexpr int EX_STAGES = 3;
constexpr int EX_STAGE_B = EX_ROWS * EX_ROWB;   // 32 KiB
constexpr int EX_PIECES = EX_STAGE_B / 1024;    // 1-KiB LDS-DMA pieces per tile (2 rows each)
constexpr int EX_EPI_PITCH = 36;                // floats per row of a wave's [32][32] transposition region
constexpr int EX_EPI_B = 32 * EX_EPI_PITCH * 4;
constexpr int EX_SMEM = EX_STAGES * EX_STAGE_B + EX_CW * EX_EPI_B;
constexpr int EX_KS = 8;                        // 16-element k-steps (zero fragments beyond kpad)
constexpr int kOobOff = (int)0x80000000u;

struct ExpandArgs {
  const float* X;              // S16 rows [M][kpad]
  const float* W;              // S16 rows [N][kpad]
  const float* x_bound;
  const float* w_bound;
  int32_t M, N, kpad;
  uint32_t x_bytes;
  int32_t n_slices, row_groups;   // grid = n_slices * row_groups workgroups
  float* stat_sum;             // pass 1
  float* stat_m2;
  const float* scale;          // pass 2
  const float* shift;
  const float* out_bound;
  float* out;
  uint8_t* bits;
  DropP drop;
};

template <int N>
__device__ __forceinline__ void ex_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void ex_stage_sync() {           // a wave's own LDS writes -> its own LDS reads
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <bool ACT>
__global__ void __launch_bounds__(EX_NT, 3) k_expand_fwd_s16(const ExpandArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[EX_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup b runs on XCD b % 8: the n_slices workgroups that read the same rows sit on the same XCD (one L2 fetch)
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;
  const int group = (b & 7) + 8 * (b / (8 * p.n_slices));
  const int n_tiles = (p.M + EX_ROWS - 1) / EX_ROWS;
  const int row_pitch_b = p.kpad * 4;

  if (w == EX_CW) {
    // ---- loader wave: tile it of this row group -> stage it % 3, two tiles ahead of the compute waves -----------------
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
    const int chunks = p.kpad / 4;                    // 16-byte chunks per row
    auto issue = [&](int tile, int stage) {
      char* sA = smem + stage * EX_STAGE_B;
#pragma unroll 8
      for (int pc = 0; pc < EX_PIECES; ++pc) {
        const int r = 2 * pc + (lane >> 5), cp = lane & 31;          // LDS row / chunk position of this lane's 16 bytes
        const int c = cp ^ (r & 15);                                 // source chunk (swizzle on the source side)
        const int64_t row = (int64_t)tile * EX_ROWS + r;
        const int off = (c < chunks && row < p.M) ? (int)(row * row_pitch_b + c * 16) : kOobOff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sA + pc * 1024), 16, off, 0, 0, 0);
      }
    };
    int t = group;
    if (t < n_tiles) issue(t, 0);
    if (t + p.row_groups < n_tiles) issue(t + p.row_groups, 1);
    int it = 0;
    for (; t < n_tiles; t += p.row_groups, ++it) {
      // tile `it` has landed when at most the pieces of tile it+1 are outstanding (loads retire in order)
      if (t + p.row_groups < n_tiles) ex_wait_vmcnt<EX_PIECES>();
      else ex_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                   // B_it: tile `it` visible; everybody is done with tile it-1
      const int t2 = t + 2 * p.row_groups;
      if (t2 < n_tiles) issue(t2, (it + 2) % EX_STAGES);
    }
    return;
  }

  // ---- compute waves ---------------------------------------------------------------------------------------------------
  const int h = lane >> 5, cl = lane & 31;
  const int n_col = slice * (EX_CW * 32) + w * 32 + cl;          // this lane's output column (B fragment / accumulator column)
  // W fragments of the wave's 32 columns: k-step s, lane half h -> elements 16 s + 8 h .. + 7 (hi chunk 4 s + 2 h, lo the next)
  f16x8 bh[EX_KS], bl[EX_KS];
  {
    const bool nok = n_col < p.N;
    const char* wrow = reinterpret_cast<const char*>(p.W) + (int64_t)(nok ? n_col : 0) * row_pitch_b;
#pragma unroll
    for (int s = 0; s < EX_KS; ++s) {
      const int c = 4 * s + 2 * h;
      const bool ok = nok && c * 4 < p.kpad;
      f16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
      bh[s] = ok ? *reinterpret_cast<const f16x8*>(wrow + c * 16) : z;
      bl[s] = ok ? *reinterpret_cast<const f16x8*>(wrow + c * 16 + 16) : z;
    }
  }
  const float acc_scale = s16_pow2(s16_exp_of(p.x_bound) + s16_exp_of(p.w_bound));
  // fragment addressing in a staged tile: row (blk*32 + cl) * 512 B, chunk (4 s + 2 h) ^ (row & 15)
  const int a_row = cl * EX_ROWB, swz = cl & 15;
  float* wreg = reinterpret_cast<float*>(smem + EX_STAGES * EX_STAGE_B + w * EX_EPI_B);

  // pass-2 constants: lane = 8 consecutive columns of a row of the wave's 32-column strip
  const int q4 = lane & 3, rsub = lane >> 2;                     // column group, row within a 16-row pass
  const int n8 = slice * (EX_CW * 32) + w * 32 + q4 * 8;
  float sc8[8], sh8[8];
  DropP d = p.drop;
  float inv_out = 1.f;
  if (ACT) {
    drop_resolve(d);
    inv_out = s16_pow2(-s16_exp_of(p.out_bound));
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      sc8[c] = n8 + c < p.N ? p.scale[n8 + c] : 0.f;
      sh8[c] = n8 + c < p.N ? p.shift[n8 + c] : 0.f;
    }
  }

  int it = 0;
  for (int t = group; t < n_tiles; t += p.row_groups, ++it) {
    __builtin_amdgcn_s_barrier();                     // B_it (see the loader)
    const char* sA = smem + (it % EX_STAGES) * EX_STAGE_B;
    // NB 32-row blocks per compute phase: both for the statistics pass (a slab's M2 needs the slab mean first), one at a time
    // for the activation pass (its epilogue constants and Philox temporaries need the registers)
    constexpr int NB = ACT ? 1 : 2;
    f32x16 acc[NB];
    auto compute = [&](int blk0) {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;
      // fragments of k-step s+1 are read while the MFMAs of k-step s run (register double buffer; pinned below: hipcc's own
      // schedule is read -> wait -> 1-2 MFMAs -> read ..., every LDS latency exposed)
      f16x8 fa[2][NB][2];                              // [buffer][block][hi / lo]
      auto load_frags = [&](int s_, int buf) {
        const int co = ((4 * s_ + 2 * h) ^ swz) * 16;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          fa[buf][blk][0] = *reinterpret_cast<const f16x8*>(sA + (blk0 + blk) * 32 * EX_ROWB + a_row + co);
          fa[buf][blk][1] = *reinterpret_cast<const f16x8*>(sA + (blk0 + blk) * 32 * EX_ROWB + a_row + (co ^ 16));
        }
      };
      load_frags(0, 0);
#pragma unroll
      for (int s = 0; s < EX_KS; ++s) {
        if (s + 1 < EX_KS) load_frags(s + 1, (s + 1) & 1);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][blk][1], bh[s], acc[blk], 0, 0, 0);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][blk][0], bl[s], acc[blk], 0, 0, 0);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][blk][0], bh[s], acc[blk], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NB, 0);        // masks: 0x100 DS read, 0x008 MFMA
#pragma unroll
      for (int s = 0; s + 1 < EX_KS; ++s) {
#pragma unroll
        for (int q = 0; q < 2 * NB; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * NB, 0);
      if (acc_scale != 1.f) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[blk][r] *= acc_scale;
      }
    };
    const int row0 = t * EX_ROWS;
    const int cnt = min(EX_ROWS, p.M - row0);         // rows of this slab (wave-uniform)
    if constexpr (!ACT) {
      // ---- BatchNorm slab statistics (sum, M2 around the slab mean): accumulator row = (reg&3) + 8 (reg>>2) + 4 h ------
      compute(0);
      float s = 0.f;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = blk * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          s += (r < cnt) ? acc[blk][reg] : 0.f;
        }
      s += __shfl_xor(s, 32);
      const float mean = s / (float)cnt;
      float q = 0.f;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = blk * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          const float dlt = acc[blk][reg] - mean;
          q += (r < cnt) ? dlt * dlt : 0.f;
        }
      q += __shfl_xor(q, 32);
      if (h == 0 && n_col < p.N) {
        p.stat_sum[(int64_t)t * p.N + n_col] = s;
        p.stat_m2[(int64_t)t * p.N + n_col] = q;
      }
    } else {
      // ---- BatchNorm + ReLU + dropout -> S16 rows + activation bits, one 32-row block at a time through the wave's region --
      // "two" mode of the mask (vp3d_dropout.h: 2 random bits per element, one Philox block per 64 elements): the wave's 32
      // columns are half of ONE block per row, so lane L evaluates the block of tile row L once per tile and keeps the two words
      // of this wave's half; the four passes below fetch their 16 bits with shuffles -- 1 Philox evaluation per lane and tile
      // instead of 4 (the pass is VALU-bound and the block's 20 multiplies were ~40 % of it)
      uint32_t pw0 = 0u, pw1 = 0u;
      const bool shared_mask = d.on && d.two && (p.N & 63) == 0;
      if (shared_mask) {
        const uint64_t q64 = (uint64_t)(row0 + lane) * (uint64_t)(p.N >> 6) + (uint64_t)((n8 - q4 * 8) >> 6);
        uint32_t r4[4];
        philox4((uint32_t)q64, (uint32_t)(q64 >> 32), d.layer, d.off_lo, d.k0, d.k1, r4);
        const bool upper = (((n8 - q4 * 8) >> 5) & 1) != 0;          // this wave's strip = columns 32..63 of the 64-column block
        pw0 = upper ? r4[2] : r4[0];
        pw1 = upper ? r4[3] : r4[1];
      }
#pragma unroll 1
      for (int blk = 0; blk < 2; ++blk) {
        compute(blk);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          wreg[r * EX_EPI_PITCH + cl] = acc[0][reg];
        }
        ex_stage_sync();
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int r = ps * 16 + rsub;
          const int m = row0 + blk * 32 + r;
          // (before the row predicate: every lane takes part in the shuffles)
          const uint32_t sw0 = __shfl(pw0, blk * 32 + r), sw1 = __shfl(pw1, blk * 32 + r);
          if (m >= p.M || n8 >= p.N) continue;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(wreg + r * EX_EPI_PITCH + q4 * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(wreg + r * EX_EPI_PITCH + q4 * 8 + 4);
          const float y8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          const int64_t e0 = (int64_t)m * p.N + n8;   // element index in the [M][N] activation (the mask's counter)
          float mk[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
          if (shared_mask) {
            const uint32_t wsel = (q4 & 2) ? sw1 : sw0;
            drop8_from_bits16(d, (q4 & 1) ? (wsel >> 16) : (wsel & 0xffffu), mk);
          } else if (d.on) {
            drop8(d, (uint64_t)(e0 >> 3), mk);
          }
          float v[8];
          uint32_t bits = 0;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float z = fmaf(y8[c], sc8[c], sh8[c]);
            v[c] = (z > 0.f ? z * mk[c] : (z != z ? z : 0.f)) * inv_out;
            bits |= (z > 0.f && mk[c] != 0.f) ? (1u << c) : 0u;
          }
          if (p.bits != nullptr) p.bits[act_bits_index(n8, m, p.M)] = (uint8_t)bits;
          f16x8 hi, lo;
          s16_split8(v, 1.f, hi, lo);
          f16x8* o = reinterpret_cast<f16x8*>(p.out + e0);
          o[0] = hi;
          o[1] = lo;
        }
        ex_stage_sync();
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the expand layer: P = G^T X  [C][kpad],  G = go * keep * [bn(y) > 0]  (include/vp3d.h, "Backward of the expand
// layer without materialising dy"), straight from the incoming gradient go (fp32 rows) and the forward's activation bits.
// Before: vp3d_act_mask_s16 wrote G as S16 (340 MB in, 340 MB out at the benchmark size) and a split-K GEMM read it again
// (136 + 118 us).  Here a wave owns 32 channels: it loads ITS column of go for 16 rows at a time (8 rows per lane half: exactly
// the MFMA A fragment, 128-byte runs per load), masks, scales and splits in registers, and multiplies with the B fragments
// of X^T (the transposed S16 copy the forward already keeps for X^T X; 64-column slabs through a 2-deep LDS ring).  go is
// read once; the kernel is bound by that read.  Register ring of 4 k-steps of loads in flight per wave (9 loads each: 8 rows of go, one bit word that the lanes share by ds_bpermute) (all VMEM operations
// of a wave are loads, so the counted vmcnt waits are exact).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int EB_NW = 8, EB_NT = EB_NW * 64;
constexpr int EB_SLAB = 64;                     // rows (reduction index) per LDS stage = 4 k-steps
constexpr int EB_ROWB = EB_SLAB * 4;            // bytes per staged X^T row
constexpr int EB_STAGE_B = 128 * EB_ROWB;       // kpad <= 128 rows
constexpr int EB_SMEM = 2 * EB_STAGE_B;

struct ExpandBwdArgs {
  const float* go;             // [M][C] fp32
  const float* go_bound;
  const uint8_t* bits;
  const float* xt;             // S16 transposed X: [kpad][ld_t]
  const float* x_bound;
  float* part;                 // [groups][C][kpad]
  float* gram_part;            // [groups][kpad][kpad] partials of X^T X (slice 0 only), or NULL
  int32_t M, C, kpad, ld_t;
  uint32_t go_bytes, bits_bytes, xt_bytes;
  int32_t n_slices, groups, rows_per;     // rows_per % 64 == 0
  float inv_keep;
};

template <int NJ>
__global__ void __launch_bounds__(EB_NT, 2) k_expand_bwd_p_s16(const ExpandBwdArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[EB_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, cl = lane & 31;
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;               // the slices of one row group share an XCD (X^T slabs: one L2 fetch)
  const int group = (b & 7) + 8 * (b / (8 * p.n_slices));
  const int c0w = slice * (EB_NW * 32) + w * 32;         // first channel of this wave
  const int row_begin = group * p.rows_per, row_end = min(p.M, row_begin + p.rows_per);
  const int n_stage = (max(0, row_end - row_begin) + EB_SLAB - 1) / EB_SLAB;

  __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)p.go, 0, p.go_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bits, 0, p.bits_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.xt, 0, p.xt_bytes, 0x00020000);

  // ---- register ring: k-step q -> slot q % 4: 8 rows of go for this lane's channel + their activation-bit words ----------
  float gv[4][8];
  uint32_t bw[4];                                        // the bit word of row 8 h + (cl & 7): lanes 0..7 of a half hold its 8 rows
  const bool c_ok = c0w + cl < p.C;
  const int g_col = (c0w + cl) * 4;
  const int64_t bits_base = (int64_t)(c0w >> 6) * p.M * 8 + ((c0w & 63) >> 3);
  auto issue_loads = [&](int q, int slot) {              // rows row_begin + 16 q + 8 h + i
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = row_begin + q * 16 + 8 * h + i;
      const bool ok = m < row_end && c_ok;
      gv[slot][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsG, ok ? (int)((int64_t)m * p.C * 4 + g_col) : kOobOff, 0, 0));
    }
    const int mb = row_begin + q * 16 + 8 * h + (cl & 7);
    bw[slot] = __builtin_amdgcn_raw_buffer_load_b32(rsB, (mb < row_end && c0w < p.C) ? (int)(bits_base + (int64_t)mb * 8) : kOobOff, 0, 0);
  };
  // X^T slab S -> buffer S & 1: piece pc = 4 rows x 256 B; lane's 16 bytes: row 4 pc + (lane >> 4), chunk position lane & 15
  auto issue_slab = [&](int S) {
    char* sB = smem + (S & 1) * EB_STAGE_B;
    const int m0 = row_begin + S * EB_SLAB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = w * 4 + i;
      const int r = 4 * pc + (lane >> 4), cp = lane & 15;
      const int c = cp ^ (r & 15);
      const int off = r < p.kpad ? (int)(((int64_t)r * p.ld_t + m0) * 4 + c * 16) : kOobOff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(sB + pc * 1024), 16, off, 0, 0, 0);
    }
  };

  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // X^T X rides along on the column slice 0 workgroups: the B fragments of X^T ARE its operands (both of them: a fragment of
  // rows 32 i .. 32 i + 31 is the same registers as A or as B), so the NJ x NJ blocks cost MFMAs only -- wave w takes blocks
  // (i, j) = (b / NJ, b % NJ) for b = w, w + 8, ...
  constexpr int NG = (NJ * NJ + EB_NW - 1) / EB_NW;
  const bool do_gram = p.gram_part != nullptr && slice == 0;      // workgroup-uniform
  f32x16 accg[NG];
#pragma unroll
  for (int u = 0; u < NG; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) accg[u][r] = 0.f;

  const float g_bound = s16_load_bound(p.go_bound) * p.inv_keep;
  const int e_g = s16_exp_for_bound(g_bound);
  const float g_scale = p.inv_keep * s16_pow2(-e_g);
  const int b_row = cl * EB_ROWB, swz = cl & 15;

  auto k_step = [&](int slot, int ks, const char* sB) {
    // B fragments of all NJ column blocks first (the LDS latency runs under the A-fragment arithmetic below)
    f16x8 bh[NJ], bl[NJ];
    const int co = ((4 * ks + 2 * h) ^ swz) * 16;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      bh[j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * EB_ROWB + b_row + co);
      bl[j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * EB_ROWB + b_row + (co ^ 16));
    }
    // A fragment: G[rows 8 h + i][channel cl] = go * keep-scale * bit, split into hi + lo halves
    f16x8 ah, al;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute((h * 32 + i) * 4, (int)bw[slot]);      // row 8 h + i
      const float g = ((word >> cl) & 1u) ? gv[slot][i] * g_scale : 0.f;
      const _Float16 hh = (_Float16)g;
      ah[i] = hh;
      al[i] = (_Float16)(g - (float)hh);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[j], 0, 0, 0);
    if (do_gram) {
      // (one case per wave: the fragment indices are compile-time constants -- selecting registers by a run-time index costs
      //  a v_cndmask per register and candidate: +50 us on the launch)
#define VP3D_GRAM_BLK(U, BLK)                                                                               \
      if constexpr ((BLK) < NJ * NJ) {                                                                       \
        constexpr int gi = (BLK) / NJ, gj = (BLK) % NJ;                                                      \
        accg[U] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[gi], bh[gj], accg[U], 0, 0, 0);                  \
        accg[U] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[gi], bl[gj], accg[U], 0, 0, 0);                  \
        accg[U] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[gi], bh[gj], accg[U], 0, 0, 0);                  \
      }
#define VP3D_GRAM_CASE(W) case W: VP3D_GRAM_BLK(0, W) if constexpr (NG > 1) { VP3D_GRAM_BLK(NG - 1, W + 8) } break;
      switch (w) {
        VP3D_GRAM_CASE(0) VP3D_GRAM_CASE(1) VP3D_GRAM_CASE(2) VP3D_GRAM_CASE(3)
        VP3D_GRAM_CASE(4) VP3D_GRAM_CASE(5) VP3D_GRAM_CASE(6) VP3D_GRAM_CASE(7)
      }
#undef VP3D_GRAM_CASE
#undef VP3D_GRAM_BLK
    }
  };

  if (n_stage > 0) {
    issue_slab(0);
    issue_loads(0, 0);
    issue_loads(1, 1);
    issue_loads(2, 2);
    for (int S = 0; S < n_stage; ++S) {
      // oldest outstanding first: [slab S+... issued after the previous barrier], loads q+1, q+2: the loads of k-step 4 S (and
      // the slab S, older) have landed when at most 32 newer loads are outstanding
      ex_wait_vmcnt<18>();
      __builtin_amdgcn_s_barrier();                  // slab S visible; everybody is done with slab S-1
      if (S + 1 < n_stage) issue_slab(S + 1);
      const char* sB = smem + (S & 1) * EB_STAGE_B;
      const int q = 4 * S;
      issue_loads(q + 3, 3);                         // (hipcc tracks these register loads itself: its counted waits before the
      k_step(0, 0, sB);                              //  uses below leave the newer k-steps -- and the next slab -- in flight)
      issue_loads(q + 4, 0);
      k_step(1, 1, sB);
      issue_loads(q + 5, 1);
      k_step(2, 2, sB);
      issue_loads(q + 6, 2);
      k_step(3, 3, sB);
    }
  }
  ex_wait_vmcnt<0>();

  // ---- partial P of this row group: part[group][c][j], scaled ------------------------------------------------------------
  const int e_x = s16_exp_of(p.x_bound);
  const float scale = s16_pow2(e_g + e_x);
  float* out = p.part + (int64_t)group * p.C * p.kpad;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int c = c0w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (c < p.C) out[(int64_t)c * p.kpad + j * 32 + cl] = acc[j][reg] * scale;
    }
  }
  if (do_gram) {
    const float sg = s16_pow2(2 * e_x);
    float* gout = p.gram_part + (int64_t)group * p.kpad * p.kpad;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int blk = w + u * EB_NW;
      if (blk < NJ * NJ) {
        const int gi = blk / NJ, gj = blk % NJ;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = gi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          gout[(int64_t)r * p.kpad + gj * 32 + cl] = accg[u][reg] * sg;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// BatchNorm statistics of the expand layer WITHOUT a pass over its conv output (round 4): y = X W^T with only kpad <= 128 input
// columns, so   mean_n = W[n] . mean(x),   var_n = W[n]^T Cov(x) W[n]   -- a [kpad x kpad] second-moment matrix of X (one MFMA pass
// over the transposed S16 copy the forward keeps anyway, 42 MB at the benchmark size) + 1024 small quadratic forms in fp64
// replace the statistics-only GEMM pass (k_expand_fwd_s16<false>, 67-78 us + vp3d_bn_finalize).
// CENTRED: the slab statistics this replaces subtract a slab mean before they square (no E[x^2] - E[x]^2 cancellation,
// test_bn_statistics_are_robust_to_large_mean); here every column k is shifted by o_k = X[0][k] -- its first row, the same
// offset in every workgroup, so the partial matrices simply add -- before the products are formed: Cov is shift-invariant and
// the shifted data has |mean| ~ std.  The constant-1 column the im2row rows carry for the no-dy backward (one_col) is NOT
// shifted: row `one_col` of the result holds the column sums  sum_m (x_k - o_k)  and its diagonal entry the row count.
//   part[group][i][j] = 2^(2 e_x) * sum_{m in group} (x'_i[m] - o'_i) (x'_j[m] - o'_j)        x' = the S16-scaled values
// Structure of k_expand_bwd_p_s16's X^T X ride-along: X^T slabs of 64 rows through a 2-deep LDS ring, wave w owns the blocks
// (i, j) = (b / NJ, b % NJ) for b = w, w + 8; the fragments are re-centred and re-split in registers (3 fragments per k-step).
// ---------------------------------------------------------------------------------------------------------------------------
struct ExpandGramArgs {
  const float* xt;             // S16 transposed X: [kpad][ld_t]
  float* part;                 // [groups][kpad][kpad]
  const float* x_bound;
  int32_t M, kpad, ld_t, one_col;
  uint32_t xt_bytes;
  int32_t groups, rows_per;    // rows_per % 64 == 0
};

// scaled offset of row r of X^T (= column r of X): its first element, 0 for the constant-1 column
__device__ __forceinline__ float gram_offset(const ExpandGramArgs& p, int r) {
  if (r >= p.kpad || r == p.one_col) return 0.f;
  const _Float16* g = reinterpret_cast<const _Float16*>(p.xt + (int64_t)r * p.ld_t);
  return (float)g[0] + (float)g[8];                  // element 0 of the first S16 group: hi half, lo half 16 B further
}

constexpr int EG_STAGES = 4;                    // X^T slabs in flight + in use: the kernel is a 42-MB read, 6 slabs per workgroup --
                                                // with a 2-deep ring every iteration waited out one DMA latency (20 us, 2.1 TB/s)
template <int NJ>
__global__ void __launch_bounds__(EB_NT, 1) k_expand_gram_s16(const ExpandGramArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[EG_STAGES * EB_STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, cl = lane & 31;
  const int group = blockIdx.x;
  const int row_begin = group * p.rows_per, row_end = min(p.M, row_begin + p.rows_per);
  const int n_stage = (max(0, row_end - row_begin) + EB_SLAB - 1) / EB_SLAB;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.xt, 0, p.xt_bytes, 0x00020000);
  auto issue_slab = [&](int S) {                    // (S >= n_stage: an all-zero DMA, so that the counted waits stay uniform)
    char* sB = smem + (S % EG_STAGES) * EB_STAGE_B;
    const int m0 = row_begin + S * EB_SLAB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = w * 4 + i;
      const int r = 4 * pc + (lane >> 4), cp = lane & 15;
      const int c = cp ^ (r & 15);
      const int off = (r < p.kpad && S < n_stage) ? (int)(((int64_t)r * p.ld_t + m0) * 4 + c * 16) : kOobOff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(sB + pc * 1024), 16, off, 0, 0, 0);
    }
  };
  // this wave's blocks: b0 = w (rows block ia, columns block jb) and b1 = w + 8 (rows block ib, the same columns)
  constexpr int NB = NJ * NJ;
  const bool has0 = w < NB, has1 = w + EB_NW < NB;
  const int ia = has0 ? w / NJ : 0, jb = has0 ? w % NJ : 0, ib = has1 ? (w + EB_NW) / NJ : 0;
  // ACCUMULATION (round 5): var_n = W[n]^T Cov W[n] cancels by kappa_n = sum |w_i w_j Cov_ij| / var_n (temporal-difference filters
  // over frame-to-frame correlated keypoints: 1e3 .. 1e5), so every relative error of G arrives in var_n times kappa_n.  The products
  // are exact (all four hi/lo terms; the lo*lo term used to be dropped: a one-sided 2^-22 / 3 on the diagonal); what rounds is the
  // fp32 accumulator.  It therefore only ever holds ONE 64-row slab -- the twelve small terms first, the four hi*hi terms last --
  // and is added to fp64 registers after every slab: ~2 * 2^-24 of a slab's sum per slab, independent between the ~M / 64 slabs
  // -> ~4e-9 of G at the benchmark size (was ~1e-7: 72 fp32 accumulations per group + the dropped term).
  f32x16 acc0, acc1;
  double dacc0[16], dacc1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[r] = acc1[r] = 0.f;
    dacc0[r] = dacc1[r] = 0.0;
  }
  const int b_row = cl * EB_ROWB, swz = cl & 15;
  // Centring happens ONCE per element, in place in LDS, before the k-steps read the slab (the first version re-centred every
  // fragment in registers: 24 fragment passes per k-step and workgroup for 4 distinct fragments, 30 us of VALU): thread ->
  // (row r, group g of 8 consecutive m) items, 128 rows x 8 groups = 1024 items over 512 threads
  float o_it[2];
  int r_it[2], g_it[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int item = tid + u * EB_NT;
    r_it[u] = item >> 3;
    g_it[u] = item & 7;
    o_it[u] = gram_offset(p, r_it[u]);
  }
  auto centre_slab = [&](char* sB, int m0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = r_it[u], g = g_it[u];
      if (r >= NJ * 32) continue;
      char* row = sB + r * EB_ROWB;
      f16x8* ph = reinterpret_cast<f16x8*>(row + (((2 * g) ^ (r & 15)) * 16));
      f16x8* pl = reinterpret_cast<f16x8*>(row + (((2 * g + 1) ^ (r & 15)) * 16));
      const f16x8 hi = *ph, lo = *pl;
      const int n_valid = row_end - (m0 + 8 * g);            // elements of this group inside the row range
      f16x8 fh, fl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = ((float)hi[e] + (float)lo[e]) - o_it[u];
        if (e >= n_valid) v = 0.f;
        const _Float16 hh = (_Float16)v;
        fh[e] = hh;
        fl[e] = (_Float16)(v - (float)hh);
      }
      *ph = fh;
      *pl = fl;
    }
  };
  auto frag = [&](const char* sB, int blk, int ks, f16x8& fh, f16x8& fl) {
    const int co = ((4 * ks + 2 * h) ^ swz) * 16;
    fh = *reinterpret_cast<const f16x8*>(sB + blk * 32 * EB_ROWB + b_row + co);
    fl = *reinterpret_cast<const f16x8*>(sB + blk * 32 * EB_ROWB + b_row + (co ^ 16));
  };

  if (n_stage > 0) {
#pragma unroll
    for (int S = 0; S < EG_STAGES - 1; ++S) issue_slab(S);
    for (int S = 0; S < n_stage; ++S) {
      ex_wait_vmcnt<4 * (EG_STAGES - 2)>();          // slab S has landed (this wave's 4 pieces; the newer slabs stay in flight)
      __builtin_amdgcn_s_barrier();                  // ... everybody's pieces; and everybody is done reading slab S-1
      issue_slab(S + EG_STAGES - 1);                 // into the buffer slab S-1 has just left
      char* sB = smem + (S % EG_STAGES) * EB_STAGE_B;
      centre_slab(sB, row_begin + S * EB_SLAB);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS writes are done (NOT __syncthreads(): it also waits for
      __builtin_amdgcn_s_barrier();                        // the slabs in flight, which is what the ring is there to avoid)
      if (has0) {
        // all fragments of the slab first (12 + 12 reads), then the MFMAs with the two blocks' chains interleaved
        f16x8 ah[4], al[4], bh[4], bl[4], jh[4], jl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          frag(sB, jb, ks, jh[ks], jl[ks]);
          frag(sB, ia, ks, ah[ks], al[ks]);
          frag(sB, ib, ks, bh[ks], bl[ks]);            // (block 0 again when this wave has no second block: unused)
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                 // small terms first: lo*lo, then the two cross terms
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], jl[ks], acc0, 0, 0, 0);
          if (has1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ks], jl[ks], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], jh[ks], acc0, 0, 0, 0);
          if (has1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ks], jh[ks], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], jl[ks], acc0, 0, 0, 0);
          if (has1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks], jl[ks], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], jh[ks], acc0, 0, 0, 0);
          if (has1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks], jh[ks], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {                   // this slab's sums leave the fp32 accumulator
          dacc0[r] += (double)acc0[r];
          acc0[r] = 0.f;
          if (has1) {
            dacc1[r] += (double)acc1[r];
            acc1[r] = 0.f;
          }
        }
      }
    }
  }
  ex_wait_vmcnt<0>();
  const double sg = (double)s16_pow2(2 * s16_exp_of(p.x_bound));
  float* gout = p.part + (int64_t)group * p.kpad * p.kpad;
  if (has0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      gout[(int64_t)(ia * 32 + r) * p.kpad + jb * 32 + cl] = (float)(dacc0[reg] * sg);
      if (has1) gout[(int64_t)(ib * 32 + r) * p.kpad + jb * 32 + cl] = (float)(dacc1[reg] * sg);
    }
  }
}

}  // namespace

int launch_expand_fwd_s16(hipStream_t s, int64_t M, int32_t N, int32_t kpad, const float* x, const float* x_bound,
                          const float* w, const float* w_bound, float* stat_sum, float* stat_m2, const float* scale,
                          const float* shift, const DropP& drop, const float* out_bound, float* out, uint8_t* bits) {
  ExpandArgs a;
  a.X = x;
  a.W = w;
  a.x_bound = x_bound;
  a.w_bound = w_bound;
  a.M = (int32_t)M;
  a.N = N;
  a.kpad = kpad;
  a.x_bytes = (uint32_t)(M * kpad * 4);
  a.n_slices = (N + EX_CW * 32 - 1) / (EX_CW * 32);
  const int n_tiles = (int)((M + EX_ROWS - 1) / EX_ROWS);
  // one workgroup per CU (132 KiB of LDS); every row group at least ~4 tiles; multiples of 8 groups
  int groups = 256 / a.n_slices;
  while (groups > 8 && groups * 4 > n_tiles) groups >>= 1;
  groups = groups < 8 ? 8 : (groups / 8) * 8;
  a.row_groups = groups;
  a.stat_sum = stat_sum;
  a.stat_m2 = stat_m2;
  a.scale = scale;
  a.shift = shift;
  a.out_bound = out_bound;
  a.out = out;
  a.bits = bits;
  a.drop = drop;
  const dim3 grid(a.n_slices * a.row_groups), block(EX_NT);
  if (out != nullptr) VP3D_LAUNCH(k_expand_fwd_s16<true>, grid, block, 0, s, a);
  else VP3D_LAUNCH(k_expand_fwd_s16<false>, grid, block, 0, s, a);
  return check_launch("expand_fwd_s16");
}

}  // namespace vp3d

namespace vp3d {
int launch_expand_bwd_p_s16(hipStream_t s, int64_t M, int32_t C, int32_t kpad, const float* go, const float* go_bound,
                            const uint8_t* bits, float p, const float* xt, int64_t ld_t, const float* x_bound, int32_t groups,
                            float* part, float* gram_part) {
  ExpandBwdArgs a;
  a.gram_part = gram_part;
  a.go = go;
  a.go_bound = go_bound;
  a.bits = bits;
  a.xt = xt;
  a.x_bound = x_bound;
  a.part = part;
  a.M = (int32_t)M;
  a.C = C;
  a.kpad = kpad;
  a.ld_t = (int32_t)ld_t;
  a.go_bytes = (uint32_t)(M * C * 4);
  a.bits_bytes = (uint32_t)(M * C / 8);
  a.xt_bytes = (uint32_t)((int64_t)kpad * ld_t * 4);
  a.n_slices = (C + EB_NW * 32 - 1) / (EB_NW * 32);
  a.groups = groups;
  const int64_t slabs = (M + EB_SLAB - 1) / EB_SLAB;
  a.rows_per = (int32_t)((slabs + groups - 1) / groups) * EB_SLAB;
  a.inv_keep = 1.0f / (1.0f - p);
  const dim3 grid(a.n_slices * groups), block(EB_NT);
  switch (kpad / 32) {
    case 1: VP3D_LAUNCH(k_expand_bwd_p_s16<1>, grid, block, 0, s, a); break;
    case 2: VP3D_LAUNCH(k_expand_bwd_p_s16<2>, grid, block, 0, s, a); break;
    case 3: VP3D_LAUNCH(k_expand_bwd_p_s16<3>, grid, block, 0, s, a); break;
    default: VP3D_LAUNCH(k_expand_bwd_p_s16<4>, grid, block, 0, s, a); break;
  }
  return check_launch("expand_bwd_p_s16");
}

// row groups of vp3d_expand_bwd_p_s16: one workgroup per CU over all column slices, >= 4 slabs per group, a multiple of 8
int expand_bwd_groups(int64_t M, int32_t C) {
  const int n_slices = (C + EB_NW * 32 - 1) / (EB_NW * 32);
  const int64_t slabs = (M + EB_SLAB - 1) / EB_SLAB;
  int groups = 256 / n_slices;
  while (groups > 8 && (int64_t)groups * 4 > slabs) groups >>= 1;
  return groups < 8 ? 8 : (groups / 8) * 8;
}

// row groups (= partial matrices) of vp3d_expand_stats_gram_s16: ~one workgroup per CU, whole 64-row slabs, at least 2 per group
int expand_gram_groups(int64_t M) {
  const int64_t slabs = (M + EB_SLAB - 1) / EB_SLAB;
  int64_t g = slabs / 2;
  if (g > 240) g = 240;
  if (g < 1) g = 1;
  const int64_t per = (slabs + g - 1) / g;           // slabs per group
  return (int)((slabs + per - 1) / per);             // no empty group
}

int launch_expand_gram_s16(hipStream_t s, int64_t M, int32_t kpad, const float* xt, int64_t ld_t, const float* x_bound,
                           int32_t one_col, int32_t groups, float* part) {
  ExpandGramArgs a;
  a.xt = xt;
  a.part = part;
  a.x_bound = x_bound;
  a.M = (int32_t)M;
  a.kpad = kpad;
  a.ld_t = (int32_t)ld_t;
  a.one_col = one_col;
  a.xt_bytes = (uint32_t)((int64_t)kpad * ld_t * 4);
  a.groups = groups;
  const int64_t slabs = (M + EB_SLAB - 1) / EB_SLAB;
  a.rows_per = (int32_t)((slabs + groups - 1) / groups) * EB_SLAB;
  const dim3 grid(groups), block(EB_NT);
  switch (kpad / 32) {
    case 1: VP3D_LAUNCH(k_expand_gram_s16<1>, grid, block, 0, s, a); break;
    case 2: VP3D_LAUNCH(k_expand_gram_s16<2>, grid, block, 0, s, a); break;
    case 3: VP3D_LAUNCH(k_expand_gram_s16<3>, grid, block, 0, s, a); break;
    default: VP3D_LAUNCH(k_expand_gram_s16<4>, grid, block, 0, s, a); break;
  }
  return check_launch("expand_gram_s16");
}
}  // namespace vp3d
