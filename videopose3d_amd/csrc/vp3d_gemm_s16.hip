// Split-fp16 ("S16", see vp3d_s16.h) implicit-GEMM kernel for the temporal convolutions (gfx950 / CDNA4):
//   C[m][n] = 2^(ea+eb) * sum_k (Ah + Al)[gather(m,k)] * (Bh + Bl)[n][k]      "NT": both operands k-contiguous
// with a*b evaluated as ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_f16 (fp32 accumulate): fp32-class results
// (22+ significant operand bits, exact fp16 products) at 3/16 of the fp32-MFMA matrix-pipe time.
//
// Because an S16 row has the byte geometry of the fp32 row it replaces, the staging is the fp32 kernel's: K in
// 32-element (128-B) tiles (or 16-element / 64-B ones), DMA'd HBM/L2 -> LDS 16 B per lane into an NSTAGE ring of
// [rows][128 B] images whose 16-B chunks are XOR-swizzled by ((row>>1)&7) on the SOURCE address; chunk 2g / 2g+1
// of a row = hi / lo halves of elements 8g..8g+7 = one MFMA A/B fragment each (ds_read_b128, conflict-free).
// Per 32-element K-tile a wave with an (RB x CB)-block sub-tile issues 4(RB+CB) fragment reads and 6*RB*CB
// MFMAs of 32 matrix-pipe cycles: the MFMA time of a K-tile is ~5x shorter than the fp32 kernel's while the bytes per
// tile are the same, so the kernel lives or dies by operand traffic per FLOP (hence the 256x256 configuration) and by
// the cost of issuing the LDS-DMA (hence the buffer-descriptor form: one 32-bit offset per 1-KiB piece).  What was
// measured on the way (ring depth, K-tile width, register-pipelined fragments, other tile shapes) is in DESIGN.md 4.6;
// the template keeps those knobs.
//
// All GEMM forms of the model are expressed as NT: forward (B = packed weight rows), dgrad (B = the transposed
// pack [(tap,ci)][co]), wgrad (both operands pre-transposed by their producers).
#include <cstdlib>
#include <type_traits>

#include "vp3d_internal.h"
#include "vp3d_s16.h"
#include "vp3d_s16_mma.h"

namespace vp3d {
namespace {

using namespace mma;

// -DVP3D_TRACE (tools/gemm_trace.py, never in the shipped library): thread 0 of every workgroup of k_nt_s16 leaves the 100-MHz
// wall clock at its phase boundaries in g_trace[workgroup][8] -- where a tile's time goes between workgroup start and exit
#ifdef VP3D_TRACE
__device__ unsigned long long* g_trace = nullptr;
#define VP3D_TR(k)                                                                                         \
  do {                                                                                                     \
    if (g_trace != nullptr && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 8 + (k)] = wall_clock64();    \
  } while (0)
#else
#define VP3D_TR(k) do { } while (0)
#endif

// position in the launch order -> tile.  Workgroup id b runs on XCD b % 8 (round-robin dispatch), and every XCD has its
// own L2, so XCD x gets a CONTIGUOUS eighth [x*per_xcd, (x+1)*per_xcd) of a linear tile order in which consecutive tiles
// form compact patches: column blocks of up to 8 tiles, inside a block m-tile by m-tile (the ~32-64 tiles an XCD runs at
// a time are a 4..8 x 8 patch: ~12 operand panels for 32 tiles).  Equal shares matter: handing out whole m-tiles per XCD
// (the fp32 kernel's order) leaves e.g. 108 m-tiles as 14/14/14/14/13/13/13/13 -- a sixth round on four XCDs.
__device__ __forceinline__ void tile_from_linear(int L, int m_tiles, int n_tiles, int& tile_m, int& tile_n) {
  const int gn = min(n_tiles, 8);
  const int full = (n_tiles / gn) * gn;              // columns in full blocks
  const int in_full = m_tiles * full;
  if (L < in_full) {
    const int blk = L / (m_tiles * gn), r = L - blk * (m_tiles * gn);
    tile_m = r / gn;
    tile_n = blk * gn + (r - tile_m * gn);
  } else {
    const int gl = n_tiles - full, r = L - in_full;
    tile_m = r / gl;
    tile_n = full + (r - tile_m * gl);
  }
}

// ---- stream-K (SK instances) -------------------------------------------------------------------------------------------
// A launch of T tiles on P resident workgroups leaves the last ceil(T/P)-th round partly empty (432 tiles of 256x256 on
// 256 CUs: 1.69 rounds cost 2).  The SK instances run the first sk_dp_blocks tiles of the linear order whole, one per
// workgroup as always, and hand the K-tiles ("units") of the remaining sk_tiles tiles to sk_blocks further workgroups in
// EQUAL contiguous shares: workgroup rank r owns units [r U / G, (r + 1) U / G) of the U = sk_tiles * K/BK units, i.e. the
// tail of one tile, possibly whole tiles, and the head of another.  A workgroup that holds only part of a tile's K range
// stores its scaled accumulators in the tile's workspace slot of its segment and draws the tile's ticket; the LAST
// contributor to arrive sums all segments in segment order (deterministic) and runs the normal epilogue.  Nobody waits:
// no co-residency requirement.  Partials cross XCDs (separate L2s), so they are stored and loaded with sc1 (past the L2)
// instead of fencing -- an agent-scope acquire would invalidate the L2 that holds everybody's operand panels.
__device__ __forceinline__ int sk_rank_of(int64_t x, int64_t U, int G) {       // largest r with floor(r U / G) <= x
  return (int)(((x + 1) * G + U - 1) / U) - 1;
}

// ACT: the instance that carries the fused BatchNorm + ReLU + dropout epilogue (Epi::act_scale) -- kept out of the
// general instances so that its Philox temporaries do not enter their register allocation.
// RED: the dgrad instance that also forms the BatchNorm-backward column sums of the upstream activation (Epi::red).
template <class C, bool ACT = false, bool SK = false, bool RED = false>
__global__ void __launch_bounds__(C::NT, (C::NW * C::OCC) / 4) k_nt_s16(const RowsGemmArgs p) {
  constexpr int RB = C::RB, CB = C::CB, BM = C::BM, BN = C::BN, NSTAGE = C::NSTAGE, PA = C::PA, PB = C::PB;
  constexpr int BK = C::BKE, ROWB = C::ROWB, RPP = C::RPP, CPR = ROWB / 16;   // chunks per row
  __shared__ __attribute__((aligned(16))) char smem[C::SMEM_B];
  int* tab_b = reinterpret_cast<int*>(smem + C::TAB_OFF);
  int* tab_t = tab_b + BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / C::WN, wn = w % C::WN;
  const int h = lane >> 5, cl = lane & 31;
  VP3D_TR(0);

  const int nkt_all = p.K / BK;
  // ---- work of this workgroup: one (tile, K range) segment, or (stream-K workgroups) a run of them ------------------------
  // uniform split-K: the (K range, tile) units in range-major order, XCD x = blockIdx % 8 takes the contiguous share
  // [x * per_xcd, (x + 1) * per_xcd) -- the tiles of ONE K range run side by side on an XCD and share its operand panels in that
  // XCD's L2 (tile_from_linear orders the tiles of a range in compact patches); every split writes its raw, scaled partial matrix [M][N] at
  // part + split*part_stride
  const int bid = blockIdx.x;
  const int nt_tiles = p.m_tiles * p.n_tiles;
  const int nt_unit = (bid & 7) * (p.pos_full >> 3) + (bid >> 3);
  const int split = SK ? 0 : nt_unit / nt_tiles;
  const int sk_T_dp = p.m_tiles * p.n_tiles - p.sk_tiles;           // stream-K: tiles [0, sk_T_dp) run whole
  const int64_t sk_U = (int64_t)p.sk_tiles * nkt_all;
  int sk_rank = 0;
  __shared__ int sk_flag;
  // One (tile, K range) segment: main loop, then the tile's epilogue -- or, for part of a shared tile, the fix-up.
  auto run_segment = [&](const int tile_m, const int tile_n, const int kt_begin, const int kt_end, const int sk_ts) {
  const int m0 = p.m_begin + tile_m * BM, n0 = tile_n * BN;      // this launch covers rows [m_begin, m_end)

  for (int r = tid; r < BM; r += C::NT) {
    const int m = min(m0 + r, p.m_end - 1);
    const int b = m / p.t_dst;
    tab_b[r] = b;
    tab_t[r] = m - b * p.t_dst;
  }
  __syncthreads();

  f32x16 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = max(0, kt_end - kt_begin);

  // ---- LDS-DMA staging: running pointers, one 64-bit add per piece per K-tile (as the fp32 kernel) ----
  const int tap0 = (kt_begin * BK) / p.c_src;
  int c0 = kt_begin * BK - tap0 * p.c_src;
  int a_t[PA];
  const float* a_ptr[PA];
  const float* b_ptr[PB];
  int b_inc[PB];
  const int zoff = (lane & (CPR - 1)) * 4;
  const int pa_cnt = C::pa_count(w), pa_first = C::pa_first(w);   // (uniform configurations: PA and w * PA)
  const int rbw = C::rb_of(wm);                                   // row blocks of this wave (MIX: the last wave row has fewer)
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (i < pa_cnt ? pa_first + i : 0) * RPP + lane / CPR;
    const int chunk = (lane & (CPR - 1)) ^ C::swz(r);
    const int b = tab_b[r], t = tab_t[r];
    a_t[i] = t * p.t_stride + p.t_off + tap0 * p.tap_step;
    a_ptr[i] = p.A + ((int64_t)b * p.t_src + a_t[i]) * p.lda + c0 + chunk * 4;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int r = (w * PB + i) * RPP + lane / CPR;
    const int chunk = (lane & (CPR - 1)) ^ C::swz(r);
    const bool ok = (n0 + r) < p.N;
    b_ptr[i] = ok ? p.B + (int64_t)(n0 + r) * p.ldb + (int64_t)kt_begin * BK + chunk * 4 : p.zeros + chunk * 4;
    b_inc[i] = ok ? BK : 0;
  }
  const int64_t a_jump = (int64_t)p.tap_step * p.lda - p.c_src + BK;   // at a tap boundary

  // buffer-descriptor path (C::BUF): per piece one 32-bit byte offset that advances by a constant per K-tile
  int a_vo[PA], a_cur[PA], b_cur[PB];
  __amdgpu_buffer_rsrc_t rsA, rsB;
  if (C::BUF) {
    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      a_vo[i] = (int)((a_ptr[i] - p.A) * 4);
      a_cur[i] = (unsigned)a_t[i] < (unsigned)p.t_src ? a_vo[i] : kOob;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) b_cur[i] = b_inc[i] != 0 ? (int)((b_ptr[i] - p.B) * 4) : kOob;
  }
  auto issue = [&](int stage, bool live) {          // `live` = false: harmless all-zero DMA (keeps vmcnt uniform)
    char* sA = smem + stage * C::STAGE_B;
    char* sB = sA + C::A_B;
    if (C::BUF) {
      if (live) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
          if (!C::MIX || i < pa_cnt) blds16(rsA, a_cur[i], sA + (pa_first + i) * 1024);
#pragma unroll
        for (int i = 0; i < PB; ++i) blds16(rsB, b_cur[i], sB + (w * PB + i) * 1024);
      } else {
#pragma unroll
        for (int i = 0; i < PA + PB; ++i) blds16(rsA, kOob, sA + (w * (PA + PB) + i) * 1024 % C::STAGE_B);
      }
      c0 += BK;
      if (c0 >= p.c_src) {                            // wave-uniform, once per tap
        c0 = 0;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          a_vo[i] += (int)(a_jump * 4);
          a_t[i] += p.tap_step;
          a_cur[i] = (unsigned)a_t[i] < (unsigned)p.t_src ? a_vo[i] : kOob;
        }
      } else {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          a_vo[i] += BK * 4;
          a_cur[i] += BK * 4;                         // kOob + a few K-tiles stays out of range
        }
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) b_cur[i] += BK * 4;
      return;
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const bool ok = live && (unsigned)a_t[i] < (unsigned)p.t_src;
      const float* g = ok ? a_ptr[i] : (p.zeros + zoff);
      glds16(g, sA + (w * PA + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const float* g = live ? b_ptr[i] : (p.zeros + zoff);
      glds16(g, sB + (w * PB + i) * 1024);
    }
    c0 += BK;
    const bool wrap = c0 >= p.c_src;                // wave-uniform
    if (wrap) c0 = 0;
    const int64_t a_inc = wrap ? a_jump : (int64_t)BK;
    const int t_inc = wrap ? p.tap_step : 0;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      a_ptr[i] += a_inc;
      a_t[i] += t_inc;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) b_ptr[i] += b_inc[i];
  };

  // fragment addressing: row = block base + cl, the swizzle term depends on cl only (block bases are multiples of 32);
  // lane half h supplies elements 16 s + 8 h .. +7 of k-step s: chunk 2 (2 s + h) holds their hi parts, the next their lo
  const int sw = C::swz(cl);
  const int off0 = ((2 * (0 + h)) ^ sw) * 16;
  const int off1 = BK == 32 ? ((2 * (2 + h)) ^ sw) * 16 : 0;
  const int a_row = (wm * RB * 32 + cl) * ROWB;
  const int b_row = (wn * CB * 32 + cl) * ROWB;

  if (nkt > 0 && !C::PIPE) {
    // NSTAGE-deep ring, NSTAGE-1 tiles in flight: iteration `it` waits for ITS tile only (counted vmcnt: the newer
    // tiles stay in flight across the barrier), then refills the stage everyone left at the end of iteration it-1
#pragma unroll
    VP3D_TR(1);
    for (int ps = 0; ps < NSTAGE - 1; ++ps) issue(ps, ps < nkt);
    // (MIX: one specialised loop per wave row -- both pass the same barriers; a branch INSIDE the loop made hipcc spill)
    auto main_loop = [&](auto rbw_c) {
      constexpr int RBW = decltype(rbw_c)::value;
      int st_c = 0, st_i = NSTAGE - 1;               // stage computed / issued this iteration
      for (int it = 0; it < nkt; ++it) {
        wait_vmcnt<(PA + PB) * (NSTAGE - 2)>();
        if (NSTAGE == 2) __syncthreads();
        else __builtin_amdgcn_s_barrier();
        issue(st_i, it + NSTAGE - 1 < nkt);
        const char* sA = smem + st_c * C::STAGE_B;
        compute_tile<RB, CB, BK / 16, ROWB, RBW>(sA + a_row, sA + C::A_B + b_row, acc, off0, off1);
        st_c = st_c + 1 == NSTAGE ? 0 : st_c + 1;
        st_i = st_i + 1 == NSTAGE ? 0 : st_i + 1;
      }
    };
    if (!C::MIX || wm != C::WM - 1) main_loop(std::integral_constant<int, RB>());
    else main_loop(std::integral_constant<int, C::RBL>());
  }
  if (nkt > 0 && C::PIPE) {
    // Register double-buffered variant: the fragments of tile it+1 are read from LDS while the MFMAs of tile `it`
    // (already in registers) run, so a wave's matrix work starts right behind the barrier instead of behind its DMA
    // issue + LDS read latency.  All NSTAGE stages are kept loaded: the stage of tile `it` is free once everybody has
    // its fragments in registers (lgkmcnt(0) + barrier) and is refilled with tile it+NSTAGE.
    Frags<RB, CB, BK / 16> fr[2];
#pragma unroll
    for (int ps = 0; ps < NSTAGE; ++ps) issue(ps, ps < nkt);
    wait_vmcnt<(PA + PB) * (NSTAGE - 1)>();          // tile 0 landed
    __builtin_amdgcn_s_barrier();
    load_frags<RB, CB, BK / 16, ROWB>(smem + a_row, smem + C::A_B + b_row, off0, off1, fr[0]);
    int st = 0;                                      // stage of tile `it`
    auto body = [&](int it, Frags<RB, CB, BK / 16>& cur, Frags<RB, CB, BK / 16>& nxt) {
      wait_vmcnt<(PA + PB) * (NSTAGE - 2)>();        // tile it+1 landed (own pieces)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my fragment reads of tile `it` are in registers
      __builtin_amdgcn_s_barrier();
      issue(st, it + NSTAGE < nkt);                  // refill the stage of tile `it`
      st = st + 1 == NSTAGE ? 0 : st + 1;
      const char* sN = smem + st * C::STAGE_B;       // tile it+1 (a harmless zero-page tile past the end)
      load_frags<RB, CB, BK / 16, ROWB>(sN + a_row, sN + C::A_B + b_row, off0, off1, nxt);
      mma_frags<RB, CB, BK / 16>(cur, acc);
      // pinned interleave: the next tile's fragment reads ride one by one behind this tile's first MFMAs
      constexpr int NR = 2 * (RB + CB) * (BK / 16), NM = 3 * RB * CB * (BK / 16), NI = NR < NM ? NR : NM;
#pragma unroll
      for (int q = 0; q < NI; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if constexpr (NR > NI) __builtin_amdgcn_sched_group_barrier(0x100, NR - NI, 0);
      if constexpr (NM > NI) __builtin_amdgcn_sched_group_barrier(0x008, NM - NI, 0);
    };
    int it = 0;
    for (; it + 1 < nkt; it += 2) {
      body(it, fr[0], fr[1]);
      body(it + 1, fr[1], fr[0]);
    }
    if (it < nkt) body(it, fr[0], fr[1]);
  }
  VP3D_TR(2);
  wait_vmcnt<0>();                                   // the trailing zero-page DMAs must not land in the staging below
  VP3D_TR(3);

  // ---- epilogue -------------------------------------------------------------------------------------------
  const bool partial = p.splits > 1;
  const Epi& e = p.epi;
  {
    // (round 6: these bound loads -- ~1 us of dependent L2 round trips per tile with the matrix pipe idle, tools/gemm_trace.py --
    //  were hoisted in front of the K loop and carried in scalar registers: no change of the step or of the eval forward in
    //  three alternating-process pairs each, one spilled VGPR: reverted, profiles/r06_gemm_tile_trace.txt)
    int ex = 0;
    if (e.bound_a != nullptr) ex += s16_exp_of(e.bound_a);
    if (e.bound_b != nullptr) ex += s16_exp_of(e.bound_b);
    if (ex != 0) {
      const float scale = s16_pow2(ex);
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= scale;
    }
  }
  if (SK && (kt_begin != 0 || kt_end != nkt_all)) {
    // ---- stream-K fix-up: this workgroup holds part of the tile's K range ------------------------------------------------
    const int r_first = sk_rank_of((int64_t)sk_ts * nkt_all, sk_U, p.sk_blocks);
    const int r_last = sk_rank_of((int64_t)sk_ts * nkt_all + nkt_all - 1, sk_U, p.sk_blocks);
    const int nseg = r_last - r_first + 1, my_seg = sk_rank - r_first;
    constexpr int NV = RB * CB * 4;                    // 16-byte vectors of accumulators per lane
    float* tile_ws = p.sk_ws + (int64_t)sk_ts * p.sk_max_seg * (BM * BN);
    {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(tile_ws + (int64_t)my_seg * (BM * BN)), 0,
                                                                    BM * BN * 4, 0x00020000);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x16& a16 = acc[v / (CB * 4)][(v / 4) % CB];
        const int q4 = (v & 3) * 4;
        u32x4 d;
        d[0] = __float_as_uint(a16[q4 + 0]);
        d[1] = __float_as_uint(a16[q4 + 1]);
        d[2] = __float_as_uint(a16[q4 + 2]);
        d[3] = __float_as_uint(a16[q4 + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, (v * C::NT + tid) * 16, 0, SK_AUX);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores have reached memory ...
    __syncthreads();                                   // ... and so have everybody's
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(p.sk_cnt + sk_ts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = t == nseg - 1;
      if (last) __hip_atomic_store(p.sk_cnt + sk_ts, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero for the next launch
      sk_flag = last;
    }
    __syncthreads();
    if (!sk_flag) return;                              // (workgroup-uniform) somebody else finishes this tile
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int sg = 0; sg < nseg; ++sg) {                // segment order: the sum does not depend on who arrived last
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(tile_ws + (int64_t)sg * (BM * BN)), 0,
                                                                    BM * BN * 4, 0x00020000);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (v * C::NT + tid) * 16, 0, SK_AUX);
        f32x16& a16 = acc[v / (CB * 4)][(v / 4) % CB];
        const int q4 = (v & 3) * 4;
        a16[q4 + 0] += __uint_as_float(d[0]);
        a16[q4 + 1] += __uint_as_float(d[1]);
        a16[q4 + 2] += __uint_as_float(d[2]);
        a16[q4 + 3] += __uint_as_float(d[3]);
      }
    }
  }
  if (!partial) {
    if (C::MIX && e.stat_sum != nullptr) {           // ... per 32-row slab (a 64-row slab would straddle two 224-row tiles)
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        if (i >= rbw) break;
        const int row0 = m0 + (wm * RB + i) * 32;
        const int cnt = min(32, p.m_end - row0);     // wave-uniform
        if (cnt > 0) {
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            const int n = n0 + (wn * CB + j) * 32 + cl;
            float s = 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
              s += (r < cnt) ? acc[i][j][reg] : 0.f;
            }
            s += __shfl_xor(s, 32);
            const float mean = s / (float)cnt;
            float q = 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
              const float d = acc[i][j][reg] - mean;
              q += (r < cnt) ? d * d : 0.f;
            }
            q += __shfl_xor(q, 32);
            if (h == 0 && n < p.N) {
              e.stat_sum[(int64_t)(row0 >> 5) * p.N + n] = s;
              e.stat_m2[(int64_t)(row0 >> 5) * p.N + n] = q;
            }
          }
        }
      }
    }
    if (!C::MIX && e.stat_sum != nullptr) {          // BatchNorm slab statistics of the raw conv output
#pragma unroll
      for (int sb = 0; sb < RB / 2; ++sb) {
        const int row0 = m0 + (wm * RB + sb * 2) * 32;
        const int cnt = min(64, p.m_end - row0);     // wave-uniform
        if (cnt > 0) {
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            const int n = n0 + (wn * CB + j) * 32 + cl;
            float s = 0.f;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) {
                const int r = i2 * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                s += (r < cnt) ? acc[sb * 2 + i2][j][reg] : 0.f;
              }
            s += __shfl_xor(s, 32);
            const float mean = s / (float)cnt;
            float q = 0.f;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) {
                const int r = i2 * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const float d = acc[sb * 2 + i2][j][reg] - mean;
                q += (r < cnt) ? d * d : 0.f;
              }
            q += __shfl_xor(q, 32);
            if (h == 0 && n < p.N) {
              e.stat_sum[(int64_t)(row0 >> 6) * p.N + n] = s;
              e.stat_m2[(int64_t)(row0 >> 6) * p.N + n] = q;
            }
          }
        }
      }
    }
  }

  VP3D_TR(4);
  if (!partial && e.no_out) return;                  // statistics-only launch (workgroup-uniform)
  __syncthreads();                                   // every wave is done reading the operand ring
  // Each wave stages one 32-row block of its sub-tile at a time through its own [32][CB*32] fp32 LDS region and
  // writes it out with 16-B stores (a row of the region = CB*128 B contiguous in C).
  constexpr int WCOLS = CB * 32, LPR = WCOLS / 4, ERPP = 64 / LPR;   // lanes per row, rows per pass
  float* wreg = reinterpret_cast<float*>(smem) + w * (32 * WCOLS);
  const int rr = lane / LPR, c4 = (lane % LPR) * 4;
  float* Cbase = partial ? p.part + (int64_t)split * p.part_floats : e.C;
  const int Nlim = p.N;
  const bool vec = partial ? (p.N % 4 == 0) : (e.vec != 0);
  float amax = 0.f;
  const float rscale = (!partial && e.r_s16) ? s16_pow2(s16_exp_of(e.r_bound)) : 1.f;
  if (ACT) {
    // ---- fused BatchNorm + ReLU + dropout -> S16 rows + activation bits: lane = 8 consecutive columns of a row -----
    DropP d = e.ab_drop;
    drop_resolve(d);
    const float inv = s16_pow2(-s16_exp_of(e.act_bound));
    constexpr int LPR8 = WCOLS / 8, ERPP8 = 64 / LPR8;
    const int rr8 = lane / LPR8, c8 = (lane % LPR8) * 8;
    const int n8 = n0 + wn * WCOLS + c8;
    float sc8[8], sh8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      sc8[c] = n8 < Nlim ? e.act_scale[n8 + c] : 0.f;
      sh8[c] = n8 < Nlim ? e.act_shift[n8 + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if (C::MIX && i >= rbw) break;
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          wreg[r * WCOLS + j * 32 + cl] = acc[i][j][reg];
        }
      epi_stage_sync();
#pragma unroll 2
      for (int ps = 0; ps < 32 / ERPP8; ++ps) {
        const int r = ps * ERPP8 + rr8;
        const int lr = (wm * RB + i) * 32 + r;
        const int m = m0 + lr;
        if (m >= p.m_end || n8 >= Nlim) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(wreg + r * WCOLS + c8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(wreg + r * WCOLS + c8 + 4);
        const float y8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const int64_t e0 = (int64_t)m * p.N + n8;      // element index in the [M][N] activation (the mask's counter)
        float mk[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        if (d.on) {
          drop8(d, (uint64_t)(e0 >> 3), mk);             // (e0 % 8 == 0: N % 8 == 0, n8 % 8 == 0)
        }
        float v[8];
        uint32_t bits = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float z = fmaf(y8[c], sc8[c], sh8[c]);
          v[c] = (z > 0.f ? z * mk[c] : (z != z ? z : 0.f)) * inv;
          bits |= (z > 0.f && mk[c] != 0.f) ? (1u << c) : 0u;
        }
        if (e.act_bits != nullptr) e.act_bits[act_bits_index(n8, m, p.M)] = (uint8_t)bits;
        f16x8 hi, lo;
        s16_split8(v, 1.f, hi, lo);
        f16x8* o = reinterpret_cast<f16x8*>(e.C + (int64_t)m * p.N + n8);
        o[0] = hi;
        o[1] = lo;
      }
      epi_stage_sync();
    }
    return;
  }
  if (!partial && e.c_s16) {
    // ---- S16 output (eval chaining): lane = 8 consecutive columns = one S16 group --------------------------
    const float wb = e.l1[0] * s16_load_bound(e.in_amax) + e.l1[1] + (e.res_amax != nullptr ? s16_load_bound(e.res_amax) : 0.f);
    if (blockIdx.x == 0 && tid == 0) e.out_wbound[0] = wb;
    const float inv = s16_pow2(-s16_exp_for_bound(wb));
    constexpr int LPR8 = WCOLS / 8, ERPP8 = 64 / LPR8;
    const int rr8 = lane / LPR8, c8 = (lane % LPR8) * 8;
    const int n8 = n0 + wn * WCOLS + c8;
    float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e.bias != nullptr && n8 < Nlim) {
#pragma unroll
      for (int c = 0; c < 8; ++c) b8[c] = e.bias[n8 + c];
    }
    const int rc8 = n8 - e.r_col0;
    const bool rcol_ok8 = e.R != nullptr && rc8 >= 0 && rc8 < e.r_cols;
    // Residual rows are fetched TWO 32-row blocks at a time, before the blocks' stores: a wave that mixes loads and stores
    // only ever gets s_waitcnt vmcnt(0) from hipcc, so a load issued between stores exposes a full load + store latency
    // (8 times per tile with the loads inside the row loop: ~22 us per round of 256x256 tiles).  The K loop's
    // operand registers are dead here: 2 x NPS x 8 VGPRs hold the prefetched S16 groups.
    constexpr int NPS = 32 / ERPP8;
    constexpr bool PRE = !SK && NPS <= 4 && RB % 2 == 0 && !C::MIX;      // (the stream-K instances have no registers to spare)
    f16x8 rh[PRE ? 2 * NPS : 1], rl[PRE ? 2 * NPS : 1];
    auto fetch_res = [&](int i0) {
#pragma unroll
      for (int q = 0; q < 2 * NPS; ++q) {
        const int lr = (wm * RB + i0 + q / NPS) * 32 + (q % NPS) * ERPP8 + rr8;
        const int b = tab_b[lr], t = tab_t[lr];
        const int tr = t * e.r_stride + e.r_off;
        const bool ok = rcol_ok8 && m0 + lr < p.m_end && n8 < Nlim && (unsigned)tr < (unsigned)e.r_t;
        const f16x8* rp = reinterpret_cast<const f16x8*>(e.R + (int64_t)b * e.r_bpitch + (int64_t)tr * e.r_ld + rc8);
        f16x8 z;
#pragma unroll
        for (int c = 0; c < 8; ++c) z[c] = (_Float16)0.f;
        rh[q] = ok ? rp[0] : z;
        rl[q] = ok ? rp[1] : z;
      }
    };
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if (C::MIX && i >= rbw) break;
      if constexpr (PRE) {
        if (i % 2 == 0 && e.R != nullptr) fetch_res(i);
      }
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          wreg[r * WCOLS + j * 32 + cl] = acc[i][j][reg];
        }
      epi_stage_sync();
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int r = ps * ERPP8 + rr8;
        const int lr = (wm * RB + i) * 32 + r;
        if (m0 + lr >= p.m_end || n8 >= Nlim) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(wreg + r * WCOLS + c8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(wreg + r * WCOLS + c8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const int b = tab_b[lr], t = tab_t[lr];
        float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) {
          if (e.R != nullptr) s16_join8(rh[(i & 1) * NPS + ps], rl[(i & 1) * NPS + ps], rscale, rv);   // (zeros where there is none)
        } else {
          const int tr = t * e.r_stride + e.r_off;
          if (rcol_ok8 && (unsigned)tr < (unsigned)e.r_t) {
            const f16x8* rp = reinterpret_cast<const f16x8*>(e.R + (int64_t)b * e.r_bpitch + (int64_t)tr * e.r_ld + rc8);
            s16_join8(rp[0], rp[1], rscale, rv);
          }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float x = v[c] + b8[c];
          if (e.relu) x = x < 0.f ? 0.f : x;
          x += rv[c];
          amax = fmaxf(amax, fabsf(x));
          v[c] = x * inv;
        }
        f16x8 hi, lo;
        s16_split8(v, 1.f, hi, lo);
        f16x8* o = reinterpret_cast<f16x8*>(e.C + (int64_t)b * e.c_bpitch + (int64_t)t * e.ldc + n8);
        o[0] = hi;
        o[1] = lo;
      }
      epi_stage_sync();
    }
  } else {
  const int n = n0 + wn * WCOLS + c4;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (!partial && e.bias != nullptr && vec && n < Nlim) bias = *reinterpret_cast<const f32x4*>(e.bias + n);
  const int rc = n - e.r_col0;
  const bool rcol_ok = !partial && e.R != nullptr && rc >= 0 && rc < e.r_cols;
  if constexpr (RED) {
    // ---- go = acc (+ residual gradient) stored as fp32 AND reduced: g = bit ? go / keep : 0, column sums of g and g * xhat
    // of the upstream activation whose gradient this tile is.  The y_up / bit loads of 32-row block i + 1 are issued before
    // block i's stores (a wave that mixes loads and stores only gets whole-counter waits: a load issued between the stores
    // of a block would expose its full latency per pass).
    constexpr int NPS = 32 / ERPP;
    constexpr int LW = BN / 4, RG = C::NT / LW;       // strip fold: lanes per partial row (4 columns each), row groups
    // The y_up rows and bits of a 32-row block are fetched in one batch at the top of the block (before its accumulators are
    // staged) and nothing is loaded between the block's stores: the first pass waits for the whole batch (one exposed load
    // latency per block), the other passes wait for nothing.  Prefetching block i + 1 during block i was built first: hipcc
    // cannot count vmcnt through the row predicates' branches and falls back to "all but the newest operation", so every
    // pass then waits for the previous pass's store (+35 us per 256x256 tile); the branch-free forms (buffer-descriptor
    // predication, clamped addresses, one register slot refilled per pass) all spill 100-200 registers beside the 128
    // accumulators of the 256x256 tiling.
    const int tap_n = n / e.ab_c, ch = n - tap_n * e.ab_c;      // (N % BN == 0: every column of the tile exists)
    const f32x4 mu = *reinterpret_cast<const f32x4*>(e.ab_mean + ch);
    const f32x4 is = *reinterpret_cast<const f32x4*>(e.ab_invstd + ch);
    const uint8_t* bp = e.red_bits + act_bits_index(ch, tap_n, e.red_m);
    const float* yp = e.ab_y + n;
    const int sh = ch & 4;
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
    // y_up rows: two register slots (block i + 1 is fetched while block i is stored); the bits: two slots where the
    // accumulators leave room (128x128 tiling), one slot fetched at the top of its own block otherwise
    constexpr int BSLOT = RB > 2 ? 1 : 2;
    f32x4 yv[2][NPS];
    uint32_t bb[BSLOT][NPS];
    auto fetch_y = [&](const int i) {
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int lr = (wm * RB + i) * 32 + ps * ERPP + rr;       // (the row table is clamped to the launch's last row:
        const int b = tab_b[lr], t = tab_t[lr];                   //  the address is valid, the row is masked below)
        yv[i & 1][ps] = *reinterpret_cast<const f32x4*>(yp + (int64_t)b * e.c_bpitch + (int64_t)t * e.ldc);
      }
    };
    auto fetch_bits = [&](const int i) {
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int lr = (wm * RB + i) * 32 + ps * ERPP + rr;
        const int b = tab_b[lr], t = tab_t[lr];
        bb[i % BSLOT][ps] = bp[((int64_t)b * e.red_row_b + (int64_t)t * e.red_row_t) * 8];
      }
    };
    fetch_y(0);
    if (BSLOT == 2) fetch_bits(0);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if (C::MIX && i >= rbw) break;                   // (the last wave row of a mixed tiling has fewer row blocks)
      if (BSLOT == 1) fetch_bits(i);
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          wreg[r * WCOLS + j * 32 + cl] = acc[i][j][reg];
        }
      if (i + 1 < RB && (!C::MIX || i + 1 < rbw)) {    // (after the staging: this block's accumulator registers are free)
        fetch_y(i + 1);
        if (BSLOT == 2) fetch_bits(i + 1);
      }
      epi_stage_sync();
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int r = ps * ERPP + rr;
        const int lr = (wm * RB + i) * 32 + r;
        if (m0 + lr >= p.m_end) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(wreg + r * WCOLS + c4);
        const int b = tab_b[lr], t = tab_t[lr];
        const int tr = t * e.r_stride + e.r_off;
        if (rcol_ok && (unsigned)tr < (unsigned)e.r_t)
          v += *reinterpret_cast<const f32x4*>(e.R + (int64_t)b * e.r_bpitch + (int64_t)tr * e.r_ld - e.r_col0 + n);
        *reinterpret_cast<f32x4*>(e.C + (int64_t)b * e.c_bpitch + (int64_t)t * e.ldc + n) = v;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        const uint32_t bits = bb[i % BSLOT][ps] >> sh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float g = ((bits >> c) & 1u) ? v[c] * e.red_inv_keep : 0.f;
          sg[c] += g;
          sgx[c] = fmaf(g, (yv[i & 1][ps][c] - mu[c]) * is[c], sgx[c]);
        }
      }
      epi_stage_sync();
    }
    VP3D_TR(5);
    // ---- tile column sums: row lanes of the wave (shuffles), the WM waves of a column (LDS), one partial row per tile ----
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sg[c] += __shfl_xor(sg[c], o);
        sgx[c] += __shfl_xor(sgx[c], o);
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    float* red2 = reinterpret_cast<float*>(smem);     // [WM][2][BN] column sums, then [NW] maxima
    __syncthreads();                                   // every wave is done with its staging region
    if (lane < LPR) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        red2[(wm * 2 + 0) * BN + wn * WCOLS + c4 + c] = sg[c];
        red2[(wm * 2 + 1) * BN + wn * WCOLS + c4 + c] = sgx[c];
      }
    }
    if (lane == 0) red2[C::WM * 2 * BN + w] = amax;
    __syncthreads();
    static_assert(C::NT == 2 * BN, "one thread per (sum, column) of the tile");
    {
      const int q = tid / BN, col = tid - q * BN;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < C::WM; ++k) t += red2[(k * 2 + q) * BN + col];
      // (agent-scope store: straight past the L2.  A release fence here would write back the XCD's whole L2, which holds
      //  this round's go tiles -- measured 25-35 us per tile)
      if (n0 + col < Nlim)
        __hip_atomic_store(e.ab_part + ((int64_t)tile_m * 2 + q) * p.N + n0 + col, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int strips = e.ab_c / BN, taps_n = p.n_tiles / strips;
    const int strip = tile_n % strips;
    if (tid == 0) {
      float m = red2[C::WM * 2 * BN];
      for (int i = 1; i < C::NW; ++i) m = fmaxf(m, red2[C::WM * 2 * BN + i]);
      s16_atomic_bound(e.amax_out, m);
      // max|go| over the STRIP's columns (what bounds g for its channels): non-negative floats order as integers
      __hip_atomic_fetch_max(e.red_cnt + strips + strip, __float_as_int(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the strip's last tile folds the strip's partial rows (all row tiles x taps) in fp64: dbeta / dgamma, and the strip's
    // share of the bound of dy.  One hand-over: what this workgroup published went out as agent-scope stores / atomics,
    // acknowledged = visible, so the ticket needs no release fence; the last arriver acquires.
    __shared__ int red_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int tk = __hip_atomic_fetch_add(e.red_cnt + strip, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = tk == p.m_tiles * taps_n - 1 ? 1 : 0;
      if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      red_flag = last;
    }
    __syncthreads();
    VP3D_TR(6);
    if (!red_flag) return;
    {
      const int col4 = (tid % LW) * 4, rg = tid / LW;
      const int R = p.m_tiles * taps_n;
      double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
      for (int r = rg; r < R; r += RG) {
        const int tm = r / taps_n, tp = r - tm * taps_n;
        const float* pr = e.ab_part + ((int64_t)tm * 2) * p.N + tp * e.ab_c + strip * BN + col4;
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(pr);
        const f32x4 u1 = *reinterpret_cast<const f32x4*>(pr + p.N);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a0[c] += (double)u0[c];
          a1[c] += (double)u1[c];
        }
      }
      double* red3 = reinterpret_cast<double*>(smem);  // [RG][2][BN]
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        red3[(rg * 2 + 0) * BN + col4 + c] = a0[c];
        red3[(rg * 2 + 1) * BN + col4 + c] = a1[c];
      }
      __syncthreads();
      const int q = tid / BN, col = tid - q * BN;
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < RG; ++k) t += red3[(k * 2 + q) * BN + col];
      (q == 0 ? e.red_dbeta : e.red_dgamma)[strip * BN + col] = (float)t;
      float* fin = reinterpret_cast<float*>(smem) + RG * 4 * BN;      // [2][BN] behind red3, then [NW] maxima
      fin[q * BN + col] = fabsf((float)t);
      __syncthreads();
      const float gmax = __int_as_float(__hip_atomic_load(e.red_cnt + strips + strip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) *
                         e.red_inv_keep;
      float bmax = 0.f;
      if (tid < BN)
        bmax = fabsf(e.ab_scale[strip * BN + tid]) * (gmax + fin[tid] * e.red_inv_m + e.red_sqrt_m1 * fin[BN + tid] * e.red_inv_m);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, o));
      if (lane == 0) fin[2 * BN + w] = bmax;
      __syncthreads();
      if (tid == 0) {
        float m = fin[2 * BN];
        for (int i = 1; i < C::NW; ++i) m = fmaxf(m, fin[2 * BN + i]);
        s16_atomic_bound(e.red_dy_bound, m);
        e.red_cnt[strip] = 0;                          // every tile of the strip has drawn its ticket: zero for the next launch
        e.red_cnt[strips + strip] = 0;
      }
    }
    return;
  }
  // fp32 residual rows of a 32-row block are fetched BEFORE the block is staged and stored: a wave that mixes loads and stores
  // only gets whole-counter waits from hipcc, so a residual load between the stores of a pass exposes a load + store latency
  // per pass.  Left to itself hipcc hoists the loads in some instances and not in others (27,648 x 3072 x 1024 dgrad launch,
  // residual on a third of the tiles: + 7 us on 256-row tiles, + 55 us on 224-row ones; residual on every tile, N = 1024:
  // + 51 us on 256-row tiles); with the explicit batch + 9 / + 9 / + 18 us (tools/dgrad_epi_bench.py,
  // profiles/r04_dgrad_residual_prefetch_ab.txt).  8 x 4 VGPRs, the K loop's operand registers are dead here.  Its own copy of
  // the row loop (the vector path only), so that the general loop below keeps its size.
  bool rows_done = false;
  if constexpr (!SK) {
    if (!partial && vec && !e.r_s16) {                            // (uniform)
      constexpr int NPSG = 32 / ERPP;
      const bool has_res = e.R != nullptr;                        // (uniform)
      f32x4 rpre[NPSG];
      int bq[NPSG], tq[NPSG];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        if (C::MIX && i >= rbw) break;
        // the block's row-table entries in one batch (the general loop reads them pass by pass: two dependent LDS round trips
        // in front of every store)
#pragma unroll
        for (int ps = 0; ps < NPSG; ++ps) {
          const int lr = (wm * RB + i) * 32 + ps * ERPP + rr;
          bq[ps] = tab_b[lr];
          tq[ps] = tab_t[lr];
        }
        // (branch-free: a lane without a residual element reads the first 16 bytes of R and discards them -- with the loads
        //  under exec branches every one of them waited for its own two row-table reads)
        if (has_res) {
#pragma unroll
          for (int ps = 0; ps < NPSG; ++ps) {
            const int lr = (wm * RB + i) * 32 + ps * ERPP + rr;
            const int tr = tq[ps] * e.r_stride + e.r_off;
            const bool ok = rcol_ok & (m0 + lr < p.m_end) & (n < Nlim) & ((unsigned)tr < (unsigned)e.r_t);
            const int64_t off = ok ? (int64_t)bq[ps] * e.r_bpitch + (int64_t)tr * e.r_ld - e.r_col0 + n : (int64_t)0;
            const f32x4 x = *reinterpret_cast<const f32x4*>(e.R + off);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rpre[ps] = ok ? x : z;
          }
        }
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            wreg[r * WCOLS + j * 32 + cl] = acc[i][j][reg];
          }
        epi_stage_sync();
#pragma unroll
        for (int ps = 0; ps < NPSG; ++ps) {
          const int r = ps * ERPP + rr;
          const int lr = (wm * RB + i) * 32 + r;
          const bool ok = (m0 + lr < p.m_end) & (n < Nlim);
          f32x4 v = *reinterpret_cast<const f32x4*>(wreg + r * WCOLS + c4);
          v += bias;
          if (e.relu) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = v[c] < 0.f ? 0.f : v[c];
          }
          if (has_res) v += rpre[ps];                   // (zeros where there is no residual)
          if (ok) {
            *reinterpret_cast<f32x4*>(e.C + (int64_t)bq[ps] * e.c_bpitch + (int64_t)tq[ps] * e.ldc + n) = v;
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          }
        }
        epi_stage_sync();
      }
      rows_done = true;
    }
  }
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    if (rows_done || (C::MIX && i >= rbw)) break;
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
      const int lr = (wm * RB + i) * 32 + r;         // row inside the tile
      if (m0 + lr >= p.m_end || n >= Nlim) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(wreg + r * WCOLS + c4);
      if (partial) {
        float* prow = Cbase + (int64_t)(m0 + lr) * p.N + n;
        if (vec) {
          *reinterpret_cast<f32x4*>(prow) = v;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (n + c < p.N) prow[c] = v[c];
        }
        continue;
      }
      const int b = tab_b[lr], t = tab_t[lr];
      float* crow = e.C + (int64_t)b * e.c_bpitch + (int64_t)t * e.ldc;
      const int tr = t * e.r_stride + e.r_off;
      const bool r_row_ok = e.R != nullptr && (unsigned)tr < (unsigned)e.r_t;
      const float* rrow = e.R + (int64_t)b * e.r_bpitch + (int64_t)tr * e.r_ld - e.r_col0;
      if (vec) {
        v += bias;
        if (e.relu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = v[c] < 0.f ? 0.f : v[c];
        }
        if (rcol_ok && r_row_ok) {
          if (e.r_s16) {                               // S16 residual: this lane's 4 columns are half of a group
            const int g0 = (n - e.r_col0) & ~7, hf = ((n - e.r_col0) >> 2) & 1;
            const _Float16* rp = reinterpret_cast<const _Float16*>(rrow + e.r_col0 + g0);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += ((float)rp[hf * 4 + c] + (float)rp[8 + hf * 4 + c]) * rscale;
          } else {
            v += *reinterpret_cast<const f32x4*>(rrow + n);
          }
        }
        *reinterpret_cast<f32x4*>(crow + n) = v;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int nn = n + c;
          if (nn >= p.N) continue;
          float x = v[c] + (e.bias != nullptr ? e.bias[nn] : 0.f);
          if (e.relu) x = x < 0.f ? 0.f : x;
          const int rcc = nn - e.r_col0;
          if (r_row_ok && rcc >= 0 && rcc < e.r_cols) x += rrow[nn];
          crow[nn] = x;
          amax = fmaxf(amax, fabsf(x));
        }
      }
    }
    epi_stage_sync();
  }
  }
  VP3D_TR(5);
  if (!partial && e.amax_out != nullptr) {             // max|stored value| of the whole launch (S16 exponent of the result)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    float* red = reinterpret_cast<float*>(smem);       // (wave 0's staging region)
    __syncthreads();                                   // ... which wave 0 may still be reading
    if (lane == 0) red[w] = amax;
    __syncthreads();
    if (tid == 0) {
      float m = red[0];
      for (int i = 1; i < C::NW; ++i) m = fmaxf(m, red[i]);
      s16_atomic_bound(e.amax_out, m);
    }
  }
  VP3D_TR(6);
  };   // run_segment

  if constexpr (!SK) {
    int tile_m, tile_n;
    if (nt_unit >= nt_tiles * p.splits) return;        // (the grid is 8 x per_xcd: the last share may be short)
    tile_from_linear(nt_unit - split * nt_tiles, p.m_tiles, p.n_tiles, tile_m, tile_n);
    const int kt_begin = split * p.kt_per_split;
    run_segment(tile_m, tile_n, kt_begin, min(nkt_all, kt_begin + p.kt_per_split), 0);
  } else {
    // units in the linear tile order: unit = tile * (K-tiles per tile) + K-tile.  A whole-tile workgroup owns one tile's
    // units, a stream-K workgroup its equal share of the shared tiles' units
    int64_t u, u_end;
    if ((int)blockIdx.x < p.sk_dp_blocks) {
      u = (int64_t)((bid & 7) * (sk_T_dp >> 3) + (bid >> 3)) * nkt_all;
      u_end = u + nkt_all;
    } else {
      const int g = (int)blockIdx.x - p.sk_dp_blocks;               // runs on XCD g % 8 (sk_dp_blocks % 8 == 0): the workgroups
      sk_rank = (g & 7) * (p.sk_blocks >> 3) + (g >> 3);            // of one XCD share contiguous tiles
      u = (int64_t)sk_T_dp * nkt_all + (int64_t)sk_rank * sk_U / p.sk_blocks;
      u_end = (int64_t)sk_T_dp * nkt_all + (int64_t)(sk_rank + 1) * sk_U / p.sk_blocks;
    }
    // Order of a stream-K workgroup's segments: the (partial) TAIL of its first tile is run LAST.  All shares are equal, so
    // every workgroup then walks K-tile position t (heads / whole tiles) or t + K/BK - share (tails) at step t: the
    // workgroups stay in lockstep along K and keep sharing operand panels in L2, as the whole-tile rounds do (in unit
    // order every workgroup starts at a different K offset: measured 1.6x slower than the plain launch).
    int64_t tail_u = -1, tail_end = 0;
    if (u % nkt_all != 0) {
      tail_u = u;
      tail_end = min(u_end, (u / nkt_all + 1) * nkt_all);
      u = tail_end;
    }
    bool first = true;
    for (;;) {
      if (u >= u_end) {
        if (tail_u < 0) break;
        u = tail_u;
        u_end = tail_end;
        tail_u = -1;
      }
      const int L = (int)(u / nkt_all);
      const int kb = (int)(u - (int64_t)L * nkt_all);
      const int ke = (int)min((int64_t)nkt_all, kb + (u_end - u));
      u += ke - kb;
      int tile_m, tile_n;
      tile_from_linear(L, p.m_tiles, p.n_tiles, tile_m, tile_n);
      if (!first) __syncthreads();                   // the previous segment's epilogue is done with the row table / staging
      first = false;
      run_segment(tile_m, tile_n, kb, ke, L - sk_T_dp);
    }
  }
}

// Finish of a split-K launch whose result needs the fused epilogue (forward / dgrad of the small-M layers): sums the
// partial matrices [splits][M][N] and applies bias / ReLU / residual / 64-row-slab BatchNorm statistics / amax.
// blockIdx.x = 64-row slab, blockIdx.y = 64-column strip; thread (rg, cq) owns rows rg + 16 i and one float4 column.
__global__ void __launch_bounds__(256) k_s16_finish(const float* __restrict__ part, int splits, int64_t part_stride, int M,
                                                    int N, int vec, int t_dst, const Epi e) {
  __shared__ float red[16][64];
  __shared__ float mean_s[64];
  const int rg = threadIdx.x >> 4, cq = threadIdx.x & 15;
  const int m_base = blockIdx.x * 64;
  const int n = blockIdx.y * 64 + cq * 4;
  const int cnt = min(64, M - m_base);
  const bool nok = n < N;
  f32x4 raw[4];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) raw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (vec) {
    // slices summed in slice order; the loads of 4 slices x 4 rows are issued together (one slice per iteration left
    // every add waiting for its own load: ~1 us per slice)
    for (int sp0 = 0; sp0 < splits; sp0 += 4) {
      f32x4 v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool ok = nok && rg + 16 * i < cnt && sp0 + u < splits;
          v[u][i] = ok ? *reinterpret_cast<const f32x4*>(part + (int64_t)(m_base + rg + 16 * i) * N + n +
                                                         (int64_t)(sp0 + u) * part_stride)
                       : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) raw[i] += v[u][i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rg + 16 * i;
      if (nok && r < cnt) {
        const float* src = part + (int64_t)(m_base + r) * N + n;
        for (int sp = 0; sp < splits; ++sp)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (n + c < N) raw[i][c] += src[(int64_t)sp * part_stride + c];
      }
    }
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
      const int nn = blockIdx.y * 64 + threadIdx.x;
      // (fin: the strip's last workgroup reads every slab's statistics -- agent-scope stores go straight past this XCD's L2)
      if (nn < N) {
        if (e.fin_tickets != nullptr) __hip_atomic_store(e.stat_sum + (int64_t)blockIdx.x * N + nn, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else e.stat_sum[(int64_t)blockIdx.x * N + nn] = t;
      }
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
      const int nn = blockIdx.y * 64 + threadIdx.x;
      if (nn < N) {
        if (e.fin_tickets != nullptr) __hip_atomic_store(e.stat_m2 + (int64_t)blockIdx.x * N + nn, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else e.stat_m2[(int64_t)blockIdx.x * N + nn] = t;
      }
    }
    // ---- BatchNorm finalize folded into this pass (vp3d_s16_fin): the LAST workgroup of a 64-column strip to get here merges
    // the strip's slabs in fp64 and writes the coefficients + running statistics -- vp3d_bn_finalize's arithmetic in its
    // summation order (slab s belongs to group s % FIN_GROUPS = 64; groups folded by the same tree), so the outputs are
    // bit-identical to the separate launch this removes from the forward's dependent chain.  The hand-over is the RED
    // instance's: statistics published with agent-scope stores (acknowledged = visible), one ticket per strip, the last arriver
    // acquires.
    if (e.fin_tickets != nullptr) {
      __shared__ int fin_last;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        // (a formal release / acquire pair since round 6: the ticket is a RELEASE at agent scope -- cumulative over the
        //  workgroup's statistics stores through the barrier above -- and every thread of the last arriver acquires below)
        const int tk = __hip_atomic_fetch_add(e.fin_tickets + blockIdx.y, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const int last = tk == (int)gridDim.x - 1 ? 1 : 0;
        if (last) __hip_atomic_store(e.fin_tickets + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero for the next launch
        fin_last = last;
      }
      __syncthreads();
      if (fin_last) {                                  // (workgroup-uniform)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // all threads: the plain loads of the other slabs' statistics follow
        constexpr int FG = 64;                         // vp3d_elementwise.hip: FIN_GROUPS
        // 16 columns per round (4 rounds): thread -> (column, 4 of the 64 groups); a group sums its slabs s = g, g + 64, ... in
        // order, then the 64 group sums of a column are folded by the tree
        __shared__ double g1[FG][17], g2[FG][17];      // (17: padded)
        const int nslab = (int)gridDim.x;
        const float momentum = e.fin_momentum_dev != nullptr ? e.fin_momentum_dev[0] : e.fin_momentum;
        for (int c0 = 0; c0 < 64; c0 += 16) {          // 16 columns per round: thread -> (column c0 + (tid & 15), groups (tid >> 4) * 4 .. + 3)
          const int cl = threadIdx.x & 15, gq = threadIdx.x >> 4;      // gq in 0..15: groups gq * 4 .. gq * 4 + 3
          const int nn = blockIdx.y * 64 + c0 + cl;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int g = gq * 4 + u;
            double a1 = 0.0, a2 = 0.0;
            if (nn < N) {
              for (int s = g; s < nslab; s += FG) {
                const int64_t left = (int64_t)M - (int64_t)s * 64;
                const double cn = (double)(left < 64 ? left : 64);
                const double sum = (double)e.stat_sum[(int64_t)s * N + nn];
                a1 += sum;
                a2 += (double)e.stat_m2[(int64_t)s * N + nn] + sum * sum / cn;
              }
            }
            g1[g][cl] = a1;
            g2[g][cl] = a2;
          }
          __syncthreads();
          for (int o = FG / 2; o >= 1; o >>= 1) {      // the fixed-shape tree of k_bn_finalize
            for (int i = threadIdx.x; i < o * 16; i += 256) {
              const int g = i >> 4, c = i & 15;
              g1[g][c] += g1[g + o][c];
              g2[g][c] += g2[g + o][c];
            }
            __syncthreads();
          }
          if (threadIdx.x < 16 && nn < N) {
            const double S1 = g1[0][cl], S2 = g2[0][cl];
            const double mean = S1 / (double)M;
            double var = (S2 - S1 * mean) / (double)M;
            if (var < 0.0) var = 0.0;
            const double invstd = 1.0 / sqrt(var + (double)e.fin_eps);
            const float sc = (float)((double)e.fin_gamma[nn] * invstd);
            e.fin_save_mean[nn] = (float)mean;
            e.fin_save_invstd[nn] = (float)invstd;
            e.fin_scale[nn] = sc;
            e.fin_shift[nn] = e.fin_beta[nn] - (float)mean * sc;
            if (e.fin_running_mean != nullptr) e.fin_running_mean[nn] = (1.f - momentum) * e.fin_running_mean[nn] + momentum * (float)mean;
            if (e.fin_running_var != nullptr) {
              const double unbiased = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
              e.fin_running_var[nn] = (1.f - momentum) * e.fin_running_var[nn] + momentum * (float)unbiased;
            }
          }
          __syncthreads();
        }
        if (blockIdx.y == 0 && threadIdx.x == 0 && e.fin_nbt != nullptr) e.fin_nbt[0] += 1;
      }
    }
  }
  float amax = 0.f;
  if (nok) {
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
      if (vec) {
        f32x4 v = raw[i];
        if (e.bias != nullptr) v += *reinterpret_cast<const f32x4*>(e.bias + n);
        if (e.relu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = v[c] < 0.f ? 0.f : v[c];
        }
        const int rc = n - e.r_col0;
        if (r_row_ok && rc >= 0 && rc < e.r_cols) v += *reinterpret_cast<const f32x4*>(rrow + n);
        *reinterpret_cast<f32x4*>(crow + n) = v;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
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
          amax = fmaxf(amax, fabsf(v));
        }
      }
    }
  }
  if (e.amax_out != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) mean_s[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) s16_atomic_bound(e.amax_out, fmaxf(fmaxf(mean_s[0], mean_s[1]), fmaxf(mean_s[2], mean_s[3])));
  }
}

// fp32 rows -> S16 rows (elementwise; one thread per 8-element group)
__global__ void __launch_bounds__(256) k_split_rows(int64_t groups, int groups_per_row, const float* __restrict__ src,
                                                    int64_t ld_src, float* __restrict__ dst, int64_t ld_dst,
                                                    const float* __restrict__ bound) {
  const float inv = bound != nullptr ? s16_pow2(-s16_exp_of(bound)) : 1.f;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= groups) return;
  const int64_t row = g / groups_per_row;
  const int gc = (int)(g - row * groups_per_row);
  const float* s = src + row * ld_src + gc * 8;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(s);
  const f32x4 v1 = *reinterpret_cast<const f32x4*>(s + 4);
  const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  f16x8 hi, lo;
  s16_split8(v, inv, hi, lo);
  f16x8* d = reinterpret_cast<f16x8*>(dst + row * ld_dst + gc * 8);
  d[0] = hi;
  d[1] = lo;
}

// *bound = max(*bound, max|src|)   (the caller zeroes *bound; non-negative floats order like their bit patterns)
__global__ void __launch_bounds__(256) k_amax(int64_t n, const float* __restrict__ src, float* __restrict__ bound, float floor_) {
  float m = floor_;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) ? n >> 2 : 0;      // 16-B loads when aligned
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {          // four 16-byte loads in flight per thread
    const f32x4 v0 = s4[i], v1 = s4[i + stride], v2 = s4[i + 2 * stride], v3 = s4[i + 3 * stride];
#pragma unroll
    for (int e = 0; e < 4; ++e) m = fmaxf(fmaxf(m, fmaxf(fabsf(v0[e]), fabsf(v1[e]))), fmaxf(fabsf(v2[e]), fabsf(v3[e])));
  }
  for (; i < n4; i += stride) {
    const f32x4 v = s4[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(src[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) s16_atomic_bound(bound, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// ---------------------------------------------------------------------------------------------------------
// k_tn_s16: weight gradient from the S16 ROWS of dy and of the layer input (both k-major, k = row m), transposing on
// the LDS read (ds_read_b64_tr_b16) instead of consuming producer-written transposed copies:
//     part[split][na][tap*C_in + ci] = sum_{m in slice}  A[m][na] * B[m*taps + tap][ci]
// A stage holds 32 rows x 256 channels of each operand: one 1-KiB LDS-DMA piece per row (whole 1-KiB runs of the 4-KB
// HBM rows), rows at a 1040-byte LDS pitch.  Fragment of v_mfma_f32_32x32x16_f16 (lane = column lane % 32, k-half
// lane / 32, 8 k values) = two transpose reads of 4 k each: per 16-lane group the 16 source addresses form a
// [4 rows][16 channels] tile and lane i receives channel i of the four rows (measured semantics:
// profiles/r01_tr_b16_probe.txt).  Rows per read: {0,1,8,9} + 2h (second read: + 4) of the 16-row k-step -- with the
// pitch == 4 dwords (mod 64 banks) the four rows of a read fall on disjoint banks; the k order inside a step is a
// permutation, the same for both operands, so the products pair up.  256x256 tile, 8 waves of 128x64, 2-stage ring.
// Prototype measurements (tools/ubench/wgrad_tr.hip): 0.419 ms for M 27,648 x 1024 x 3072 (NT on copies: 0.438).
// ---------------------------------------------------------------------------------------------------------
typedef short v4s_t __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
constexpr int TN_BM = 256, TN_BK = 32, TN_NT = 512, TN_PITCH = 1040;
constexpr int TN_OP_B = TN_BK * TN_PITCH;                    // operand image of 32 rows x 256 channels
// Narrow B operand (128 channels per row: the expand conv's im2row rows): a 1-KiB LDS-DMA piece holds TWO 512-byte
// rows, so the image is 16 row PAIRS at a pitch of 1056 B (= 264 dwords: pairs {0,4} / {1,5} / {2,6} / {3,7} of a
// transpose read fall on disjoint bank halves; the two rows of a pair share banks: 2-way on 1/5 of the reads)
constexpr int TN_PAIR = 1056, TN_OPB128_B = 16 * TN_PAIR;
template <int CB>
struct TnGeo {
  static constexpr int BN = CB * 128;                        // 256 (CB = 2) or 128 (CB = 1) columns of B per tile
  static constexpr int STAGE_B = TN_OP_B + (CB == 2 ? TN_OP_B : TN_OPB128_B);
  static constexpr int SMEM_B = 2 * STAGE_B;
};

struct TnArgs {
  const float* A;          // dy rows  [Mk][lda]   (S16, 4-byte units)
  const float* B;          // x rows   [Mk*taps][ldb]
  float* part;             // [splits][NA][NB]
  const float* bound_a;
  const float* bound_b;
  int Mk, lda, ldb, NA, NB, taps, c_in;
  uint32_t a_bytes, b_bytes;
  int m_tiles, n_tiles, splits, kt_per_split, per_xcd;
};

// 8 k values of one column: two transpose reads, `second` bytes (4 rows) apart
__device__ __forceinline__ f16x8 tn_frag(const char* p, int second = 4 * TN_PITCH) {
  const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)p);
  const v4s_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(p + second));
  const s16x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(f16x8, v);
}

template <int CB>
__global__ void __launch_bounds__(TN_NT, 2) k_tn_s16(const TnArgs p) {
  constexpr int RB = 4;
  using G = TnGeo<CB>;
  __shared__ __attribute__((aligned(16))) char smem[G::SMEM_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / 4, wn = w % 4;
  // (K slice, tile) units in slice-major order, XCD x (= workgroup id % 8: its own L2) takes the contiguous share
  // [x * per_xcd, (x + 1) * per_xcd): the units that run side by side on an XCD are tiles of the SAME K slice, which stream the
  // same operand rows in step, so a row panel is fetched once per XCD for all of them.  (Tile-major shares -- every slice of a
  // few tiles per XCD -- fetched 680 / 1614 MB for the 231 / 466 MB of the two 27,648-row launches: profiles/r04_step_table.txt.)
  // Measured against the tile-major shares, same process, shuffled order (profiles/r04_tn_slice_major_ab.txt): the two 27,648-row
  // launches + their reductions 449 -> 437 and 166 -> 156 us, the step 4.225 -> 4.198 ms; results are bit-identical (the
  // partial of a (slice, tile) unit does not depend on which workgroup forms it).
  int tile_m, tile_n;
  const int tiles = p.m_tiles * p.n_tiles;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int u = xcd * p.per_xcd + q;
  if (q >= p.per_xcd || u >= tiles * p.splits) return;
  const int split = u / tiles;
  tile_from_linear(u - split * tiles, p.m_tiles, p.n_tiles, tile_m, tile_n);
  const int na0 = tile_m * TN_BM, nb0 = tile_n * G::BN;
  const int tap = nb0 / p.c_in, ci0 = nb0 - tap * p.c_in;        // a column tile lies inside one tap (C_in % BN == 0)
  const int nkt_all = (p.Mk + TN_BK - 1) / TN_BK;
  const int kt_begin = split * p.kt_per_split;
  const int nkt = max(0, min(nkt_all, kt_begin + p.kt_per_split) - kt_begin);

  f32x16 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LDS-DMA: wave w owns rows 4w .. 4w+3 of both operands; rows >= Mk lie beyond num_records and arrive as zeros
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  constexpr int NPB = CB == 2 ? 4 : 2;                       // B pieces per wave: one per row, or one per row pair
  int a_vo[4], b_vo[NPB];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = (int64_t)kt_begin * TN_BK + w * 4 + i;
    const int64_t ao = (r * p.lda + na0) * 4 + lane * 16;
    a_vo[i] = ao < (int64_t)p.a_bytes ? (int)ao : kOob;
  }
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int64_t r = (int64_t)kt_begin * TN_BK + w * 4 + (CB == 2 ? i : 2 * i + (lane >> 5));
    const int64_t bo = ((r * p.taps + tap) * p.ldb + ci0) * 4 + (CB == 2 ? lane : (lane & 31)) * 16;
    b_vo[i] = bo < (int64_t)p.b_bytes ? (int)bo : kOob;
  }
  const int a_step = TN_BK * p.lda * 4, b_step = TN_BK * p.taps * p.ldb * 4;
  auto issue = [&](int stage) {
    char* sA = smem + stage * G::STAGE_B;
#pragma unroll
    for (int i = 0; i < 4; ++i) blds16(rsA, a_vo[i], sA + (w * 4 + i) * TN_PITCH);
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      blds16(rsB, b_vo[i], sA + TN_OP_B + (CB == 2 ? (w * 4 + i) * TN_PITCH : (w * 2 + i) * TN_PAIR));
#pragma unroll
    for (int i = 0; i < 4; ++i)             // (offsets past the end stay past the end: operands are < 2 GiB, kOob = 2^31)
      a_vo[i] = (unsigned)a_vo[i] + (unsigned)a_step < 0x80000000u ? a_vo[i] + a_step : kOob;
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      b_vo[i] = (unsigned)b_vo[i] + (unsigned)b_step < 0x80000000u ? b_vo[i] + b_step : kOob;
  };

  const int g = lane >> 4, sl = lane & 15, h = g >> 1;
  const int rsel = ((sl >> 2) & 1) + 8 * ((sl >> 3) & 1) + 2 * h;          // {0,1,8,9}[sl >> 2] + 2h
  const int col_lane = (g & 1) * 64 + ((sl & 3) >> 1) * 32 + ((sl & 3) & 1) * 8;
  const int a_off = wm * (RB * 32) * 4 + rsel * TN_PITCH + col_lane;
  // row r of the narrow B image sits at (r >> 1) * TN_PAIR + (r & 1) * 512
  const int b_off = TN_OP_B + (CB == 2 ? wn * (CB * 32) * 4 + rsel * TN_PITCH
                                       : wn * 128 + (4 * ((sl >> 3) & 1) + h) * TN_PAIR + ((sl >> 2) & 1) * 512) + col_lane;
  constexpr int B_SECOND = CB == 2 ? 4 * TN_PITCH : 2 * TN_PAIR;          // 4 rows further
  constexpr int B_KSTEP = CB == 2 ? 16 * TN_PITCH : 8 * TN_PAIR;          // 16 rows further

  if (nkt > 0) {
    issue(0);
    int st = 0;
    for (int it = 0; it < nkt; ++it) {
      wait_vmcnt<0>();
      __syncthreads();
      const char* sS = smem + st * G::STAGE_B;
      // All fragment reads of the tile first, THEN the LDS-DMA of the next tile (other stage), then the MFMAs: hipcc
      // cannot prove that a transpose read does not alias an LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of
      // the first read that follows one -- issued ahead of the reads (as k_nt_s16 does) the DMA would be waited for
      // before this tile's compute instead of overlapping it.
      f16x8 ah[2][RB], al[2][RB], bh[2][CB], bl[2][CB];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          ah[ks][i] = tn_frag(sS + a_off + i * 128 + ks * 16 * TN_PITCH);
          al[ks][i] = tn_frag(sS + a_off + i * 128 + ks * 16 * TN_PITCH + 16);
        }
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          bh[ks][j] = tn_frag(sS + b_off + j * 128 + ks * B_KSTEP, B_SECOND);
          bl[ks][j] = tn_frag(sS + b_off + j * 128 + ks * B_KSTEP + 16, B_SECOND);
        }
      }
      if (it + 1 < nkt) issue(st ^ 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
      }
      st ^= 1;
    }
  }

  // raw, scaled partial matrix of this K-slice (vp3d_wgrad_reduce sums the slices): 128-B runs per store
  const float scale = s16_pow2(s16_exp_of(p.bound_a) + s16_exp_of(p.bound_b));
  float* out = p.part + (int64_t)split * p.NA * p.NB;
  const int hh = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        const int na = na0 + (wm * RB + i) * 32 + r, nb = nb0 + (wn * CB + j) * 32 + cl;
        out[(int64_t)na * p.NB + nb] = acc[i][j][reg] * scale;
      }
}

template <class C>
int launch_cfg(hipStream_t s, RowsGemmArgs a, int splits, int m_begin = 0, int m_end = -1) {
  a.m_begin = m_begin;
  a.m_end = m_end < 0 ? a.M : m_end;
  a.m_tiles = (a.m_end - a.m_begin + C::BM - 1) / C::BM;
  a.n_tiles = (a.N + C::BN - 1) / C::BN;
  const int positions = 8 * ((a.m_tiles * a.n_tiles + 7) / 8);       // 8 equal XCD shares
  const int nkt = a.K / C::BKE;
  a.pos_full = 8 * ((a.m_tiles * a.n_tiles * splits + 7) / 8);        // grid: 8 equal XCD shares of the (K range, tile) units
  a.tail_pos = 0;
  a.splits = splits;
  a.kt_per_split = (nkt + splits - 1) / splits;
  if (a.epi.act_scale != nullptr) {
    if constexpr (C::BUF && !C::PIPE && C::BKE == 32 && !C::MIX) {          // (instantiated for the planner's configurations only)
      VP3D_LAUNCH((k_nt_s16<C, true>), dim3(a.pos_full), dim3(C::NT), 0, s, a);
      return check_launch("nt_s16(act)");
    } else {
      set_error("nt_s16: the fused activation epilogue exists for tile configurations 20 / 22 / 30 only");
      return VP3D_E_INVALID;
    }
  }
  if (a.epi.red) {
    if constexpr (C::BUF && !C::PIPE && C::BKE == 32 && C::NT == 2 * C::BN && (!C::MIX || C::RB == 4)) {
      VP3D_REQUIRE(splits == 1 && a.epi.vec && a.N % C::BN == 0 && a.epi.ab_c % C::BN == 0 && a.m_begin == 0 && a.m_end == a.M,
                   "nt_s16: the fused BatchNorm-backward sums need one K slice, 16-byte aligned fp32 output and c_out, c_up "
                   "multiples of the %d-column tile", C::BN);
      VP3D_LAUNCH((k_nt_s16<C, false, false, true>), dim3(positions), dim3(C::NT), 0, s, a);
      return check_launch("nt_s16(red)");
    } else {
      set_error("nt_s16: the fused BatchNorm-backward sums exist for tile configurations 20 / 22 / 28 only (operands below 2 GiB)");
      return VP3D_E_INVALID;
    }
  }
  VP3D_LAUNCH((k_nt_s16<C>), dim3(a.pos_full), dim3(C::NT), 0, s, a);
  return check_launch("nt_s16");
}

#ifdef VP3D_BUILD_EXPERIMENTS
// ---- stream-K geometry (host): which tiles are shared, by how many workgroups, workspace ------------------------------------
struct SkGeom {
  int t_dp, t_sk, blocks, max_seg;
  int64_t ws_floats;
};
constexpr int kSkMinUnits = 8;          // K-tiles per stream-K workgroup at least (below that the fix-up dominates)
template <class C>
bool sk_geometry(int M, int N, int K, SkGeom* g) {
  const int64_t T = (int64_t)((M + C::BM - 1) / C::BM) * ((N + C::BN - 1) / C::BN);
  const int P = 256 * C::OCC, nkt = K / C::BKE;
  if (T <= 0 || T >= (1 << 30) || nkt < kSkMinUnits) return false;
  const int rem = (int)(T % P);
  if (rem == 0) return false;                          // whole rounds: nothing to balance
  int t_sk = rem;
  if (T >= P && rem < P / 2) t_sk += P;                // at least half a tile of work per workgroup
  const int64_t U = (int64_t)t_sk * nkt;
  int G = P;
  while (G > 8 && U < (int64_t)G * kSkMinUnits) G >>= 1;
  if (U < (int64_t)G * kSkMinUnits) return false;
  const int per = (int)(U / G);
  g->t_dp = (int)T - t_sk;
  g->t_sk = t_sk;
  g->blocks = G;
  g->max_seg = (nkt + per - 1) / per + 1;
  g->ws_floats = (int64_t)t_sk * g->max_seg * C::BM * C::BN;
  return true;
}

template <class C>
int launch_cfg_sk(hipStream_t s, RowsGemmArgs a, float* ws, int64_t ws_floats, int32_t* tickets) {
  SkGeom g;
  if (!sk_geometry<C>(a.M, a.N, a.K, &g)) return launch_cfg<C>(s, a, 1);     // whole rounds (or too little K): plain launch
  VP3D_REQUIRE(ws != nullptr && aligned16(ws) && ws_floats >= g.ws_floats && tickets != nullptr,
               "nt_s16: the stream-K configuration needs its workspace (%lld floats) and %d zeroed tickets (vp3d_nt_s16_workspace)",
               (long long)g.ws_floats, g.t_sk);
  a.m_begin = 0;
  a.m_end = a.M;
  a.m_tiles = (a.M + C::BM - 1) / C::BM;
  a.n_tiles = (a.N + C::BN - 1) / C::BN;
  a.pos_full = g.t_dp + g.blocks;
  a.tail_pos = 0;
  a.splits = 1;
  a.kt_per_split = a.K / C::BKE;
  a.sk_dp_blocks = g.t_dp;
  a.sk_tiles = g.t_sk;
  a.sk_blocks = g.blocks;
  a.sk_max_seg = g.max_seg;
  a.sk_ws = ws;
  a.sk_cnt = tickets;
  VP3D_LAUNCH((k_nt_s16<C, false, true>), dim3(g.t_dp + g.blocks), dim3(C::NT), 0, s, a);
  return check_launch("nt_s16(stream-K)");
}

#endif  // VP3D_BUILD_EXPERIMENTS

}  // namespace

// Workspace of a launch in configuration cfg: floats (0 = none) and zeroed int32 tickets (0 = none).
void nt_s16_workspace(int M, int N, int K, int cfg, int splits, int raw, int64_t* ws_floats, int32_t* tickets) {
  *ws_floats = (splits > 1 || raw) ? (int64_t)(splits > 1 ? splits : 1) * M * N : 0;
  *tickets = 0;
#ifdef VP3D_BUILD_EXPERIMENTS
  SkGeom g;
  bool sk = false;
  if (cfg == 120) sk = sk_geometry<Cfg<2, 2, 2, 2, 2, 32, 0, 1>>(M, N, K, &g);
  if (cfg == 122) sk = sk_geometry<Cfg<2, 4, 4, 2, 2, 32, 0, 1>>(M, N, K, &g);
  if (sk) {
    *ws_floats = g.ws_floats;
    *tickets = g.t_sk;
  }
#else
  (void)K; (void)cfg;
#endif
}

// 1 when the library was built with -DVP3D_BUILD_EXPERIMENTS: the measured-and-not-adopted S16 GEMM instances (stream-K
// 120 / 122, register-pipelined 10 / 13 / 21 / 23, the 256x128 / 128x256 pairs 24 / 25, four-wave 256x256 26, the hybrid
// launch pair 30: DESIGN.md 4.6-4.7) exist.  The default build leaves them out: 40 % of the library's compile time, the only
// instances with register spills, and surface the planner never uses.
int nt_s16_has_experiments() {
#ifdef VP3D_BUILD_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

// Tile configuration and split-K factor of an [M,N,K] S16 GEMM: cost model in microseconds, least-squares fit (10 % rms)
// to the tools/s16_tune.py sweep on MI355X (22 shapes of the training step x 2 tilings x up to 10 split factors; its
// picks are within 1.5 % of the measured optimum summed over the step).  A CU works through ceil(workgroups/256)
// workgroups ("rounds"); per 32-element K-tile a 128x128 tile takes 0.66 us of a CU when two workgroups share it
// (0.86 us alone) + 3.4 us per tile; a 256x256 tile 2.22 us (1.68-2.22 us when at most one round is in flight: fewer
// waves compete for the LDS-DMA path) + 8.3 us per tile (row table, first DMA latency, epilogue).  The finishing pass
// of a split forward/dgrad launch streams the partial matrices (mostly Infinity-Cache resident) once each way, a raw
// (wgrad) launch leaves that to vp3d_wgrad_reduce.
constexpr double kLaunchUs = 3.1;
static double cost_128(int64_t wgs, double nk) {
  const int64_t per_cu = (wgs + 255) / 256;
  return (double)per_cu * (nk * (per_cu >= 2 ? 0.655 : 0.862) + 3.44);
}
static double cost_256(int64_t wgs, double nk) {
  const int64_t per_cu = (wgs + 255) / 256;
  const double fill = wgs < 256 ? (double)wgs / 256.0 : 1.0;
  return (double)per_cu * (nk * (per_cu >= 2 ? 2.22 : 1.68 + 0.54 * fill) + 8.3);
}
// rows [0, m_split) = the largest whole number of m-tiles whose 256x256 tiles fit in whole rounds of 256 (0: no such split)
static int hybrid_split_rows(int M, int N) {
  const int n256 = (N + 255) / 256;
  const int64_t tiles = (int64_t)((M + 255) / 256) * n256;
  const int64_t full_rounds = tiles / 256;
  if (full_rounds < 1 || tiles % 256 == 0) return 0;
  return (int)((full_rounds * 256 / n256) * 256);
}

// one K-tile of a 224-row tile against a 256-row one: tools/ubench/kloop.hip, 2.18 vs 2.44 us per K-tile and round
static double cost_224(int64_t wgs, double nk) {
  const int64_t per_cu = (wgs + 255) / 256;
  const double fill = wgs < 256 ? (double)wgs / 256.0 : 1.0;
  return (double)per_cu * (nk * 0.893 * (per_cu >= 2 ? 2.22 : 1.68 + 0.54 * fill) + 7.6);
}

// 160-row tiles (configuration 29: wave rows of 3 + 2 row blocks): 5/7 of a 224-row tile's MFMAs per K-tile, a little more than
// 5/7 of its time (fragment reads per MFMA 0.60 instead of 0.52, the same barrier and DMA issue)
static double cost_160(int64_t wgs, double nk) {
  const int64_t per_cu = (wgs + 255) / 256;
  const double fill = wgs < 256 ? (double)wgs / 256.0 : 1.0;
  return (double)per_cu * (nk * 0.67 * (per_cu >= 2 ? 2.22 : 1.68 + 0.54 * fill) + 6.6);
}

// rows per BatchNorm statistics slab that configuration cfg writes
int nt_s16_stat_slab_rows(int cfg) { return (cfg == 28 || cfg == 29) ? 32 : 64; }

void plan_nt_s16(int M, int N, int K, int allow_split, int raw, int* cfg_out, int* splits_out, int allow_mix) {
  const int nkt = K / 32;
  double best = 1e30;
  int best_cfg = 0, best_s = 1;
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int bt = cfg == 0 ? 128 : 256;
    const int64_t tiles = (int64_t)((M + bt - 1) / bt) * ((N + bt - 1) / bt);
    static const int kSplits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};     // the factors the sweep measured
    for (int si = 0; si < (allow_split ? 10 : 1); ++si) {
      const int s = kSplits[si];
      if (s > 1 && nkt / s < 6) break;
      const double nk = (double)((nkt + s - 1) / s);
      double cost = kLaunchUs + (cfg == 0 ? cost_128(tiles * s, nk) : cost_256(tiles * s, nk));
      if (s > 1) cost += (raw ? 0.0 : 5.5) + (double)s * (double)M * (double)N * (raw ? 4.0 : 8.0) / 7.4e6;
      if (cost < best * 0.98) {
        best = cost;
        best_cfg = cfg == 0 ? 20 : 22;              // buffer-descriptor DMA variants (the launcher falls back for >= 2 GiB operands)
        best_s = s;
      }
    }
  }
  // Hybrid 30: the 256x256 configuration on whole rounds of 256 tiles and the 128x128 one on the rows of the fractional
  // last round (4x finer tiles on 2x the slots instead of 256 - f idle CUs for a whole tile time).
  // (Re-measured with the round-2 kernel, tools/dgrad_epi_bench.py + tools/env_ab.py: the plain 256x256 launch is now
  // faster on every shape of the step -- 518 vs 568 us on M 27,648 x N 3072 x K 1024, -0.7 % step time -- because the
  // two launches serialise; the hybrid stays available as the explicit configuration 30.)
  constexpr bool kPlanHybrid = false;
  if (kPlanHybrid && best_cfg == 22 && best_s == 1) {
    const int m_split = hybrid_split_rows(M, N);
    if (m_split > 0 && m_split < M) {
      const int n256 = (N + 255) / 256, n128 = (N + 127) / 128;
      const double nk = (double)nkt;
      const int64_t ta = (int64_t)(m_split / 256) * n256, tb = (int64_t)((M - m_split + 127) / 128) * n128;
      // (the two launches serialise, which the sum under-counts by ~10 %: measured hybrid / single = 0.96 where the sum
      // says 0.91 [M 27,648 x N 3072 x K 1024], 1.07 where it says 0.955 [M 27,648 x N 1024 x K 3072])
      const double hybrid = 2.0 * kLaunchUs + cost_256(ta, nk) + cost_128(tb, nk);
      if (hybrid < best * 0.93) best_cfg = 30;
    }
  }
  // Configuration 28 (224 x 256 tiles, one K slice): where the 256-row tiling leaves most of its last round idle
  // One-slice launches are eligible from half a round of 256-row tiles up.  Round 4, alternating processes on one box, three
  // pairs (a temporary knob): eligibility from 256 tiles -> 4.478-4.486 ms, from 128 -> 4.426-4.433 ms (-1.1 %), 32 and 1 like 128
  // -- the 144-tile launches of the 9,216-row layers run better as 168 tiles of 224 rows (and better still as 232 tiles of 160
  // rows, below) than as 128 x 128 tiles or K slices of 256 x 256 ones; the cost model decides per launch.
  constexpr int min_tiles = 128;
  // K slices for the two mixed tilings as well: the M = 3,072 layers are 20 row tiles of 160 rows -- 80 tiles x 3 slices = 240
  // workgroups are one 94 %-full round where 192 tiles of 128 x 128 x 4 slices = 1.5 rounds of 512 slots (step 4.275-4.299 ->
  // 4.201-4.232 ms, alternating processes, a temporary knob)
  constexpr bool mix_splits = true;
  constexpr bool allow_160 = true;
  if (allow_mix && !raw && N % 256 == 0) {
    const bool big = (int64_t)((M + 255) / 256) * (N / 256) >= min_tiles;      // eligibility of the one-slice launches
    static const int kSplits[] = {1, 2, 3, 4, 6, 8};
    for (int which = 0; which < 2; ++which) {
      if (which == 1 && !allow_160) break;
      const int rows = which == 0 ? 224 : 160;
      const int64_t tiles = (int64_t)((M + rows - 1) / rows) * (N / 256);
      for (int si = 0; si < ((allow_split && mix_splits) ? 6 : 1); ++si) {
        const int sp = kSplits[si];
        if (sp > 1 && nkt / sp < 6) break;
        if (sp == 1 && !big) continue;
        const double nk = (double)((nkt + sp - 1) / sp);
        double cost = kLaunchUs + (which == 0 ? cost_224(tiles * sp, nk) : cost_160(tiles * sp, nk));
        if (sp > 1) cost += 5.5 + (double)sp * (double)M * (double)N * 8.0 / 7.4e6;
        if (cost < best * 0.97) {
          best = cost;
          best_cfg = which == 0 ? 28 : 29;
          best_s = sp;
        }
      }
    }
  }
  *cfg_out = best_cfg;
  *splits_out = best_s;
}

// cfg < 0: planned.  splits > 1 needs `ws` (splits*M*N floats): forward/dgrad launches then run k_s16_finish for the fused
// epilogue; with raw_partials the partial matrices ARE the result (wgrad: vp3d_wgrad_reduce sums them).
int launch_nt_s16(hipStream_t s, const RowsGemmArgs& a_in, int cfg, int splits, float* ws, int64_t ws_floats,
                  bool raw_partials, int32_t* tickets) {
  RowsGemmArgs a = a_in;
  a.sk_dp_blocks = a.sk_tiles = a.sk_blocks = a.sk_max_seg = 0;
  a.sk_ws = nullptr;
  a.sk_cnt = nullptr;
  VP3D_REQUIRE(a.K % KQ == 0 && a.c_src % KQ == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && aligned16(a.A) && aligned16(a.B) &&
                   aligned16(a.zeros),
               "nt_s16: the split-fp16 GEMM needs channel counts %% 32 == 0 and 16-byte aligned S16 rows");
  if (cfg < 0) {
    int pc, ps;
    plan_nt_s16(a.M, a.N, a.K, ws != nullptr, raw_partials ? 1 : 0, &pc, &ps, 0);
    cfg = pc;
    if (splits <= 0) splits = ps;
  }
  if (splits <= 0) splits = 1;
  VP3D_REQUIRE(splits <= a.K / KQ, "nt_s16: splits=%d exceeds the %d K-tiles", splits, a.K / KQ);
  if (splits > 1) {
    VP3D_REQUIRE(ws != nullptr && aligned16(ws) && ws_floats >= (int64_t)splits * a.M * a.N,
                 "nt_s16: split-K needs a workspace of splits*M*N floats");
    a.part = ws;
    a.part_floats = (int64_t)a.M * a.N;            // stride between the partial matrices
  }
  a.epi.vec = (a.N % 4 == 0) && (a.epi.ldc % 4 == 0) && (a.epi.c_bpitch % 4 == 0) && aligned16(a.epi.C) &&
              (a.epi.bias == nullptr || aligned16(a.epi.bias)) &&
              (a.epi.R == nullptr || (aligned16(a.epi.R) && a.epi.r_ld % 4 == 0 && a.epi.r_bpitch % 4 == 0 &&
                                      a.epi.r_col0 % 4 == 0 && a.epi.r_cols % 4 == 0));
  // byte extents for the buffer-descriptor DMA path (configurations >= 20): both operands must stay below 2 GiB
  const int64_t a_rows = (int64_t)(a.M / a.t_dst) * a.t_src;
  const int64_t a_bytes = ((a_rows - 1) * a.lda + a.c_src) * 4, b_bytes = ((int64_t)(a.N - 1) * a.ldb + a.K) * 4;
  a.a_bytes = (uint32_t)a_bytes;
  a.b_bytes = (uint32_t)b_bytes;
#ifndef VP3D_BUILD_EXPERIMENTS
  VP3D_REQUIRE(cfg == 0 || cfg == 4 || cfg == 20 || cfg == 22 || cfg == 28 || cfg == 29,
               "nt_s16: tile configuration %d is an experiment this library was built without (-DVP3D_BUILD_EXPERIMENTS)", cfg);
  (void)tickets;
  if ((cfg == 20 || cfg == 22) && (a_bytes >= ((int64_t)1 << 31) || b_bytes >= ((int64_t)1 << 31)))
    cfg = cfg == 20 ? 0 : 4;                     // operands of 2 GiB and more: the flat-address form of the same tiling
#else
  if (cfg == 120 || cfg == 122) {                // stream-K instances of 20 / 22 (whole-launch configurations: no split-K)
    const bool flat = a_bytes >= ((int64_t)1 << 31) || b_bytes >= ((int64_t)1 << 31);
    if (flat || raw_partials || a.epi.no_out || a.epi.act_scale != nullptr) {
      cfg -= 100;
      splits = 1;
    } else {
      return cfg == 120 ? launch_cfg_sk<Cfg<2, 2, 2, 2, 2, 32, 0, 1>>(s, a, ws, ws_floats, tickets)
                        : launch_cfg_sk<Cfg<2, 4, 4, 2, 2, 32, 0, 1>>(s, a, ws, ws_floats, tickets);
    }
  }
  if (cfg >= 20 && cfg <= 23 && (a_bytes >= ((int64_t)1 << 31) || b_bytes >= ((int64_t)1 << 31))) {
    static const int flat_of[4] = {0, 10, 4, 13};
    cfg = flat_of[cfg - 20];
  }
  if (cfg == 30) {                               // hybrid (see plan_nt_s16): two launches over disjoint row ranges
    const int m_split = hybrid_split_rows(a.M, a.N);
    const bool flat = a_bytes >= ((int64_t)1 << 31) || b_bytes >= ((int64_t)1 << 31);
    if (m_split <= 0 || m_split >= a.M || splits != 1) {
      cfg = flat ? 4 : 22;
    } else {
      int r1 = flat ? launch_cfg<Cfg<2, 4, 4, 2, 2, 32, 0, 0>>(s, a, 1, 0, m_split)
                    : launch_cfg<Cfg<2, 4, 4, 2, 2, 32, 0, 1>>(s, a, 1, 0, m_split);
      if (r1 != VP3D_OK) return r1;
      return flat ? launch_cfg<Cfg<2, 2, 2, 2, 2, 32, 0, 0>>(s, a, 1, m_split, a.M)
                  : launch_cfg<Cfg<2, 2, 2, 2, 2, 32, 0, 1>>(s, a, 1, m_split, a.M);
    }
  }
#endif  // VP3D_BUILD_EXPERIMENTS
  if (cfg == 28 || cfg == 29) {
    VP3D_REQUIRE(!raw_partials && a.epi.act_scale == nullptr && (!a.epi.red || cfg == 28) && a_bytes < ((int64_t)1 << 31) &&
                     b_bytes < ((int64_t)1 << 31),
                 "nt_s16: tile configurations 28 / 29 (224 / 160 x 256) take no raw output, no fused activation / BatchNorm-backward sums and "
                 "operands below 2 GiB");
  }
  VP3D_REQUIRE(a.epi.stat_sum == nullptr || splits > 1 || a.stat_slab_rows == nt_s16_stat_slab_rows(cfg),
               "nt_s16: the statistics buffers were sized for %d-row slabs, tile configuration %d writes %d-row slabs",
               a.stat_slab_rows, cfg, nt_s16_stat_slab_rows(cfg));
  int rc;
  switch (cfg) {
    // flat-address LDS-DMA (operands of any size)
    case 0: rc = launch_cfg<Cfg<2, 2, 2, 2, 2, 32, 0, 0>>(s, a, splits); break;    // 128x128, 4 waves, 2 x 32-element stages, 2 WG/CU
    case 4: rc = launch_cfg<Cfg<2, 4, 4, 2, 2, 32, 0, 0>>(s, a, splits); break;    // 256x256, 8 waves of 128x64, 2 x 32
#ifdef VP3D_BUILD_EXPERIMENTS
    case 10: rc = launch_cfg<Cfg<2, 2, 2, 2, 2, 32, 1, 0>>(s, a, splits); break;   // 128x128, register-pipelined fragments
    case 13: rc = launch_cfg<Cfg<2, 4, 4, 2, 4, 16, 1, 0>>(s, a, splits); break;   // 256x256, register-pipelined, 4 x 16-element ring
#endif
    // buffer-descriptor LDS-DMA (operands < 2 GiB): what plan_nt_s16 picks
    case 20: rc = launch_cfg<Cfg<2, 2, 2, 2, 2, 32, 0, 1>>(s, a, splits); break;
    case 22: rc = launch_cfg<Cfg<2, 4, 4, 2, 2, 32, 0, 1>>(s, a, splits); break;
    case 28: rc = launch_cfg<Cfg<2, 4, 4, 2, 2, 32, 0, 1, 3>>(s, a, splits); break;  // 224x256: wave rows of 4 + 3 row blocks
    case 29: rc = launch_cfg<Cfg<2, 4, 3, 2, 2, 32, 0, 1, 2>>(s, a, splits); break;  // 160x256: wave rows of 3 + 2 row blocks
#ifdef VP3D_BUILD_EXPERIMENTS
    case 21: rc = launch_cfg<Cfg<2, 2, 2, 2, 2, 32, 1, 1>>(s, a, splits); break;   // measured alternatives
    case 23: rc = launch_cfg<Cfg<2, 4, 4, 2, 4, 16, 1, 1>>(s, a, splits); break;
    // two INDEPENDENT 4-wave workgroups of 128x64 sub-tiles per CU (3 x 16-element stages each: 2 x 74 KiB of LDS): the
    // per-wave operand reuse of the 256x256 tiling, but one workgroup's epilogue / barrier waits run under the other's MFMAs
    case 24: rc = launch_cfg<Cfg<2, 2, 4, 2, 3, 16, 0, 1>>(s, a, splits); break;   // 256x128
    case 25: rc = launch_cfg<Cfg<2, 2, 2, 4, 3, 16, 0, 1>>(s, a, splits); break;   // 128x256
    // 256x256 by FOUR waves of 128x128 (one per SIMD; the 256 accumulator registers of a wave live in AGPRs): a third
    // fewer LDS fragment reads per MFMA than the 8-wave tiling, but nothing runs under a wave's stage-start LDS latency
    // and barrier wait: measured 8 % slower than cfg 22 on the large shapes
    case 26: rc = launch_cfg<Cfg<2, 2, 4, 4, 2, 32, 0, 1>>(s, a, splits); break;
#endif  // VP3D_BUILD_EXPERIMENTS
    default:
      set_error("nt_s16: unknown tile configuration %d", cfg);
      return VP3D_E_INVALID;
  }
  if (rc != VP3D_OK || splits == 1 || raw_partials) return rc;
  VP3D_LAUNCH(k_s16_finish, dim3((a.M + 63) / 64, (a.N + 63) / 64), dim3(256), 0, s, ws, splits, a.part_floats, a.M,
                     a.N, (int)(a.epi.vec && a.N % 4 == 0), a.t_dst, a.epi);
  return check_launch("s16_finish");
}

int launch_split_rows(hipStream_t s, int64_t M, int32_t C, const float* src, int64_t ld_src, float* dst, int64_t ld_dst,
                      const float* bound) {
  const int64_t groups = M * (C / 8);
  VP3D_LAUNCH(k_split_rows, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, groups, C / 8, src, ld_src,
                     dst, ld_dst, bound);
  return check_launch("split_rows");
}

// dW partials from S16 rows (k_tn_s16).  c_out, c_in multiples of 256; both operands below 2 GiB.
int launch_wgrad_rows_s16(hipStream_t s, int64_t Mk, const float* dy, int64_t ld_dy, int32_t c_out, const float* dy_bound,
                          const float* x, int64_t ld_x, int32_t taps, int32_t c_in, const float* x_bound, int32_t splits,
                          float* part) {
  const int64_t a_bytes = Mk * ld_dy * 4, b_bytes = Mk * taps * ld_x * 4;
  const bool narrow = c_in == 128;                       // one 128-column tile per tap (the expand conv's im2row rows)
  VP3D_REQUIRE(Mk > 0 && dy && x && part && dy_bound && x_bound && taps >= 1 && c_out > 0 && c_in > 0 && c_out % 256 == 0 &&
                   (c_in % 256 == 0 || narrow) && ld_dy >= c_out && ld_x >= c_in && ld_dy % 8 == 0 && ld_x % 8 == 0 &&
                   aligned16(dy) && aligned16(x) && aligned16(part),
               "wgrad_rows_s16: needs c_out %% 256 == 0, c_in %% 256 == 0 (or == 128) and 16-byte aligned S16 rows");
  VP3D_REQUIRE(a_bytes < ((int64_t)1 << 31) && b_bytes < ((int64_t)1 << 31), "wgrad_rows_s16: operands must stay below 2 GiB");
  const int nkt = (int)((Mk + TN_BK - 1) / TN_BK);
  VP3D_REQUIRE(splits >= 1 && splits <= nkt, "wgrad_rows_s16: splits=%d for %d K-tiles", splits, nkt);
  TnArgs a;
  a.A = dy; a.B = x; a.part = part; a.bound_a = dy_bound; a.bound_b = x_bound;
  a.Mk = (int)Mk; a.lda = (int)ld_dy; a.ldb = (int)ld_x; a.NA = c_out; a.NB = taps * c_in; a.taps = taps; a.c_in = c_in;
  a.a_bytes = (uint32_t)a_bytes; a.b_bytes = (uint32_t)b_bytes;
  a.m_tiles = c_out / TN_BM; a.n_tiles = a.NB / (narrow ? 128 : 256);
  a.splits = splits;
  a.kt_per_split = (nkt + splits - 1) / splits;
  a.per_xcd = (a.m_tiles * a.n_tiles * splits + 7) / 8;            // 8 equal XCD shares of the (slice, tile) units
  const unsigned grid = 8u * a.per_xcd;
  if (narrow) VP3D_LAUNCH(k_tn_s16<1>, dim3(grid), dim3(TN_NT), 0, s, a);
  else VP3D_LAUNCH(k_tn_s16<2>, dim3(grid), dim3(TN_NT), 0, s, a);
  return check_launch("wgrad_rows_s16");
}

int launch_amax(hipStream_t s, int64_t n, const float* src, float* bound, float floor_) {
  const int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);        // >= 4 float4 per thread
  VP3D_LAUNCH(k_amax, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, n, src, bound, floor_);
  return check_launch("amax");
}

#ifdef VP3D_TRACE
int set_trace(void* p) {
  unsigned long long* q = (unsigned long long*)p;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &q, sizeof(q)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace vp3d

#ifdef VP3D_TRACE
extern "C" int vp3d_debug_trace(void* p) { return vp3d::set_trace(p); }
#endif
