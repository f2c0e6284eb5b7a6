// Micro-benchmark (stand-alone): where does the K loop of the 256x256 split-fp16 GEMM tile (k_nt_s16<Cfg<2,4,4,2,2,32,0,1>>:
// 8 waves of 128x64, two 64-KiB LDS stages, operands through buffer-descriptor LDS-DMA) lose its time?  The same loop with
// parts of the operand path removed -- an upper bound of what moving the weight operand off the LDS path could give:
//   ABL 0  the library's loop              1  no B DMA (stale LDS)       2  no B DMA, no B fragment reads (B in registers)
//   ABL 3  no DMA at all (LDS reads kept)  4  MFMAs + barrier only
//   hipcc --offload-arch=gfx950 -O3 -I../../include -I../../videopose3d_amd/csrc kloop.hip -o kloop && ./kloop
#include "vp3d_s16_mma.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <type_traits>
using namespace vp3d;
using namespace vp3d::mma;
namespace vp3d { void set_error(const char*, ...) {} int check_launch(const char*) { return 0; } }

typedef Cfg<2, 4, 4, 2, 2, 32, 0, 1> C22;

template <int ABL>
__global__ void __launch_bounds__(C22::NT, 2) k_loop(const float* __restrict__ A, const float* __restrict__ B, float* out, int M,
                                                     int N, int K, int n_tiles) {
  using C = C22;
  constexpr int RB = C::RB, CB = C::CB, BM = C::BM, BN = C::BN, PA = C::PA, PB = C::PB, BK = C::BKE, ROWB = C::ROWB, RPP = C::RPP;
  constexpr int CPR = ROWB / 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * C::STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / C::WN, wn = w % C::WN;
  const int h = lane >> 5, cl = lane & 31;
  const int tile_m = blockIdx.x / n_tiles, tile_n = blockIdx.x % n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = K / BK;
  f32x16 acc[RB][CB];
  for (int i = 0; i < RB; ++i)
    for (int j = 0; j < CB; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((int64_t)M * K * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)((int64_t)N * K * 4), 0x00020000);
  int a_cur[PA], b_cur[PB];
  for (int i = 0; i < PA; ++i) {
    const int r = (w * PA + i) * RPP + lane / CPR, chunk = (lane & (CPR - 1)) ^ C::swz(r);
    a_cur[i] = ((m0 + r) * K + chunk * 4) * 4;
  }
  for (int i = 0; i < PB; ++i) {
    const int r = (w * PB + i) * RPP + lane / CPR, chunk = (lane & (CPR - 1)) ^ C::swz(r);
    b_cur[i] = ((n0 + r) * K + chunk * 4) * 4;
  }
  auto issue = [&](int stage) {
    char* sA = smem + stage * C::STAGE_B;
    char* sB = sA + C::A_B;
    if (ABL < 3) {
#pragma unroll
      for (int i = 0; i < PA; ++i) blds16(rsA, a_cur[i], sA + (w * PA + i) * 1024);
    }
    if (ABL == 0) {
#pragma unroll
      for (int i = 0; i < PB; ++i) blds16(rsB, b_cur[i], sB + (w * PB + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) a_cur[i] += BK * 4;
#pragma unroll
    for (int i = 0; i < PB; ++i) b_cur[i] += BK * 4;
  };
  const int sw = C::swz(cl);
  const int off0 = ((2 * (0 + h)) ^ sw) * 16, off1 = ((2 * (2 + h)) ^ sw) * 16;
  const int a_row = (wm * RB * 32 + cl) * ROWB, b_row = (wn * CB * 32 + cl) * ROWB;
  f16x8 bh[2][CB], bl[2][CB];
  for (int s = 0; s < 2; ++s)
    for (int j = 0; j < CB; ++j) {
      bh[s][j] = *reinterpret_cast<const f16x8*>(B + (int64_t)(n0 + wn * 64 + j * 32 + cl) * K + s * 16 + h * 8);
      bl[s][j] = *reinterpret_cast<const f16x8*>(B + (int64_t)(n0 + wn * 64 + j * 32 + cl) * K + s * 16 + h * 8 + 4);
    }
  issue(0);
  int st_c = 0, st_i = 1;
  for (int it = 0; it < nkt; ++it) {
    wait_vmcnt<0>();
    __syncthreads();
    issue(st_i);
    const char* sA = smem + st_c * C::STAGE_B + a_row;
    const char* sB = smem + st_c * C::STAGE_B + C::A_B + b_row;
    if (ABL == 0 || ABL == 1 || ABL == 3) {
      compute_tile<RB, CB, 2, ROWB>(sA, sB, acc, off0, off1);
    } else {
      f16x8 ah[2][RB], al[2][RB];
      if (ABL == 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int off = s == 0 ? off0 : off1;
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            ah[s][i] = *reinterpret_cast<const f16x8*>(sA + i * (32 * ROWB) + off);
            al[s][i] = *reinterpret_cast<const f16x8*>(sA + i * (32 * ROWB) + (off ^ 16));
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            ah[s][i] = bh[s][i & 1];
            al[s][i] = bl[s][i & 1];
          }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][i], bh[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bl[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bh[s][j], acc[i][j], 0, 0, 0);
      }
    }
    st_c ^= 1;
    st_i ^= 1;
  }
  wait_vmcnt<0>();
  float s = 0.f;
  for (int i = 0; i < RB; ++i)
    for (int j = 0; j < CB; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(int64_t)blockIdx.x * C::NT + tid] = s;
}

// ---- round 6: the K loop of a CONSUMER-SIDE fused BatchNorm + ReLU + dropout (north_star: "each dilated conv is a fused
// conv+BN+ReLU+dropout kernel"; review item 8): the A operand is not an S16 activation that a producer pass wrote, but the RAW
// fp32 conv output y of the previous layer, and the consumer forms a = keep * relu(y * scale_k + shift_k) itself, per K-tile,
// in registers -- global_load (32 B of y + 1 byte of stored activation bits per (row, 8-channel group)) -> fma / max / select
// -> hi / lo split -> two ds_write_b128 into the same swizzled LDS image the LDS-DMA path writes.  B stays on LDS-DMA.
// Register double buffer: the raw loads of K-tile it + 1 are issued behind the barrier of iteration `it` and converted
// behind its MFMAs.  What the pass-free forward of the 27,648-row 1x1 conv would run its K loop at.
__global__ void __launch_bounds__(C22::NT, 2) k_loop_fused(const float* __restrict__ Y, const float* __restrict__ B, const uint8_t* __restrict__ bits,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, float* out,
                                                           int M, int N, int K, int n_tiles) {
  using C = C22;
  constexpr int RB = C::RB, CB = C::CB, BM = C::BM, BN = C::BN, PB = C::PB, BK = C::BKE, ROWB = C::ROWB, RPP = C::RPP;
  constexpr int CPR = ROWB / 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * C::STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / C::WN, wn = w % C::WN;
  const int h = lane >> 5, cl = lane & 31;
  const int tile_m = blockIdx.x / n_tiles, tile_n = blockIdx.x % n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = K / BK;
  f32x16 acc[RB][CB];
  for (int i = 0; i < RB; ++i)
    for (int j = 0; j < CB; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)((int64_t)N * K * 4), 0x00020000);
  int b_cur[PB];
  for (int i = 0; i < PB; ++i) {
    const int r = (w * PB + i) * RPP + lane / CPR, chunk = (lane & (CPR - 1)) ^ C::swz(r);
    b_cur[i] = ((n0 + r) * K + chunk * 4) * 4;
  }
  // A items of this thread: (row, group) = (idx >> 2, idx & 3) for idx = tid and tid + 512: 4 threads cover a row's 128 bytes
  const int row0 = tid >> 2, g = tid & 3, row1 = row0 + 128;
  const float* y0 = Y + (int64_t)(m0 + row0) * K + g * 8;
  const float* y1 = Y + (int64_t)(m0 + row1) * K + g * 8;
  const uint8_t* q0 = bits + ((int64_t)(m0 + row0) * K + g * 8) / 8;
  const uint8_t* q1 = bits + ((int64_t)(m0 + row1) * K + g * 8) / 8;
  const int lds0 = row0 * ROWB, lds1 = row1 * ROWB;
  const int ch0 = ((2 * g) ^ C::swz(row0)) * 16, cl0 = ((2 * g + 1) ^ C::swz(row0)) * 16;
  const int ch1 = ((2 * g) ^ C::swz(row1)) * 16, cl1 = ((2 * g + 1) ^ C::swz(row1)) * 16;
  f32x4 raw[4];
  uint32_t rb[2];
  auto load_raw = [&](int kt) {
    raw[0] = *reinterpret_cast<const f32x4*>(y0 + kt * BK);
    raw[1] = *reinterpret_cast<const f32x4*>(y0 + kt * BK + 4);
    raw[2] = *reinterpret_cast<const f32x4*>(y1 + kt * BK);
    raw[3] = *reinterpret_cast<const f32x4*>(y1 + kt * BK + 4);
    rb[0] = q0[kt * (BK / 8)];
    rb[1] = q1[kt * (BK / 8)];
  };
  auto convert_store = [&](int kt, int stage) {
    char* sA = smem + stage * C::STAGE_B;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + kt * BK + g * 8), s1 = *reinterpret_cast<const f32x4*>(scale + kt * BK + g * 8 + 4);
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(shift + kt * BK + g * 8), t1 = *reinterpret_cast<const f32x4*>(shift + kt * BK + g * 8 + 4);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yy = e < 4 ? raw[2 * u][e] : raw[2 * u + 1][e - 4];
        const float z = fmaf(yy, e < 4 ? s0[e] : s1[e - 4], e < 4 ? t0[e] : t1[e - 4]);
        v[e] = (z > 0.f && ((rb[u] >> e) & 1u)) ? z * 1.3333334f : 0.f;
      }
      f16x8 hi, lo;
      s16_split8(v, 0.125f, hi, lo);
      *reinterpret_cast<f16x8*>(sA + (u ? lds1 : lds0) + (u ? ch1 : ch0)) = hi;
      *reinterpret_cast<f16x8*>(sA + (u ? lds1 : lds0) + (u ? cl1 : cl0)) = lo;
    }
  };
  auto issue_b = [&](int stage) {
    char* sB = smem + stage * C::STAGE_B + C::A_B;
#pragma unroll
    for (int i = 0; i < PB; ++i) blds16(rsB, b_cur[i], sB + (w * PB + i) * 1024);
#pragma unroll
    for (int i = 0; i < PB; ++i) b_cur[i] += BK * 4;
  };
  const int sw = C::swz(cl);
  const int off0 = ((2 * (0 + h)) ^ sw) * 16, off1 = ((2 * (2 + h)) ^ sw) * 16;
  const int a_row = (wm * RB * 32 + cl) * ROWB, b_row = (wn * CB * 32 + cl) * ROWB;
  load_raw(0);
  issue_b(0);
  convert_store(0, 0);
  int st_c = 0, st_i = 1;
  for (int it = 0; it < nkt; ++it) {
    wait_vmcnt<0>();
    __syncthreads();                                   // (drains lgkmcnt as well: the A image written behind the previous MFMAs)
    const int nx = it + 1 < nkt ? it + 1 : it;
    load_raw(nx);
    issue_b(st_i);
    const char* sA = smem + st_c * C::STAGE_B + a_row;
    const char* sB = smem + st_c * C::STAGE_B + C::A_B + b_row;
    compute_tile<RB, CB, 2, ROWB>(sA, sB, acc, off0, off1);
    convert_store(nx, st_i);
    st_c ^= 1;
    st_i ^= 1;
  }
  wait_vmcnt<0>();
  float s = 0.f;
  for (int i = 0; i < RB; ++i)
    for (int j = 0; j < CB; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(int64_t)blockIdx.x * C::NT + tid] = s;
}

// ---- round 4: the 224 x 256 tile (verdict item 3): 27,648 rows = 123.4 row tiles -> 124 x 4 = 496 tiles = 1.94 rounds of 256
// CUs (432 tiles of 256 x 256 = 1.69 rounds cost 2).  Same 8 waves in the same 2 x 4 grid: wave row 0 keeps its 4 row blocks
// (rows 0..127), wave row 1 has 3 (rows 128..223) -- wave w runs on SIMD w % 4, so every SIMD hosts one wave of each kind and
// the matrix pipes stay balanced (42 instead of 48 MFMAs per SIMD and k-step, 22 instead of 24 fragment reads).
__global__ void __launch_bounds__(C22::NT, 2) k_loop224(const float* __restrict__ A, const float* __restrict__ B, float* out, int M,
                                                        int N, int K, int n_tiles) {
  using C = C22;
  constexpr int CB = C::CB, BN = C::BN, PB = C::PB, BK = C::BKE, ROWB = C::ROWB, RPP = C::RPP;
  constexpr int BM = 224, A_B = BM * ROWB, STAGE_B = A_B + BN * ROWB;
  constexpr int CPR = ROWB / 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / C::WN, wn = w % C::WN;
  const int h = lane >> 5, cl = lane & 31;
  const int tile_m = blockIdx.x / n_tiles, tile_n = blockIdx.x % n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = K / BK;
  f32x16 acc[4][CB];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < CB; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((int64_t)M * K * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)((int64_t)N * K * 4), 0x00020000);
  // 28 A pieces of 8 rows: waves 0-3 issue 4 each, waves 4-7 issue 3 each
  const int pa_n = w < 4 ? 4 : 3, pa_base = w < 4 ? w * 4 : 16 + (w - 4) * 3;
  int a_cur[4], b_cur[PB];
  for (int i = 0; i < 4; ++i) {
    const int r = (pa_base + i) * RPP + lane / CPR, chunk = (lane & (CPR - 1)) ^ C::swz(r);
    a_cur[i] = ((m0 + r) * K + chunk * 4) * 4;
  }
  for (int i = 0; i < PB; ++i) {
    const int r = (w * PB + i) * RPP + lane / CPR, chunk = (lane & (CPR - 1)) ^ C::swz(r);
    b_cur[i] = ((n0 + r) * K + chunk * 4) * 4;
  }
  auto issue = [&](int stage) {
    char* sA = smem + stage * STAGE_B;
    char* sB = sA + A_B;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < pa_n) blds16(rsA, a_cur[i], sA + (pa_base + i) * 1024);
#pragma unroll
    for (int i = 0; i < PB; ++i) blds16(rsB, b_cur[i], sB + (w * PB + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) a_cur[i] += BK * 4;
#pragma unroll
    for (int i = 0; i < PB; ++i) b_cur[i] += BK * 4;
  };
  const int sw = C::swz(cl);
  const int off0 = ((2 * (0 + h)) ^ sw) * 16, off1 = ((2 * (2 + h)) ^ sw) * 16;
  const int a_row = (wm * 4 * 32 + cl) * ROWB, b_row = (wn * CB * 32 + cl) * ROWB;
  issue(0);
  // one specialised loop per wave row (both run the same number of barriers; a branch INSIDE the loop made hipcc spill 71 VGPRs)
  auto loop = [&](auto rbw) {
    constexpr int RBW = decltype(rbw)::value;
    int st_c = 0, st_i = 1;
    for (int it = 0; it < nkt; ++it) {
      wait_vmcnt<0>();
      __syncthreads();
      issue(st_i);
      const char* sA = smem + st_c * STAGE_B + a_row;
      const char* sB = smem + st_c * STAGE_B + A_B + b_row;
      compute_tile<4, CB, 2, ROWB, RBW>(sA, sB, acc, off0, off1);
      st_c ^= 1;
      st_i ^= 1;
    }
  };
  if (wm == 0) loop(std::integral_constant<int, 4>());
  else loop(std::integral_constant<int, 3>());
  wait_vmcnt<0>();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < CB; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(int64_t)blockIdx.x * C::NT + tid] = s;
}

int main() {
  const int M = 32768, N = 1024, K = 3072;          // 128 x 4 = 512 tiles of 256x256: two full rounds of 256 CUs
  std::vector<_Float16> ha((size_t)M * K * 2), hb((size_t)N * K * 2);
  for (auto& v : ha) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  for (auto& v : hb) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  float *A, *B, *out;
  hipMalloc(&A, (size_t)M * K * 4);
  hipMalloc(&B, (size_t)N * K * 4);
  hipMalloc(&out, (size_t)512 * 512 * 4);
  hipMemcpy(A, ha.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int tiles = (M / 256) * (N / 256);
  const char* names[5] = {"library loop", "no B DMA", "no B DMA, no B LDS reads", "no DMA at all", "MFMA + barrier only"};
  for (int abl = 0; abl < 5; ++abl) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      switch (abl) {
        case 0: hipLaunchKernelGGL(k_loop<0>, dim3(tiles), dim3(C22::NT), 0, 0, A, B, out, M, N, K, N / 256); break;
        case 1: hipLaunchKernelGGL(k_loop<1>, dim3(tiles), dim3(C22::NT), 0, 0, A, B, out, M, N, K, N / 256); break;
        case 2: hipLaunchKernelGGL(k_loop<2>, dim3(tiles), dim3(C22::NT), 0, 0, A, B, out, M, N, K, N / 256); break;
        case 3: hipLaunchKernelGGL(k_loop<3>, dim3(tiles), dim3(C22::NT), 0, 0, A, B, out, M, N, K, N / 256); break;
        default: hipLaunchKernelGGL(k_loop<4>, dim3(tiles), dim3(C22::NT), 0, 0, A, B, out, M, N, K, N / 256); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("ABL %d  %-28s %.3f ms   %.0f TFLOP/s algorithmic (%.0f executed)   %.2f us per K-tile\n", abl, names[abl], best,
           2.0 * M * N * K / best / 1e9, 6.0 * M * N * K / best / 1e9, best * 1e3 / 2 / (K / 32));
  }
  {
    // round 6: A through registers with the fused BatchNorm + ReLU + dropout (stored bits), K = 1024 and K = 3072
    uint8_t* bits;
    float *sc, *sh;
    hipMalloc(&bits, (size_t)M * K / 8);
    hipMemset(bits, 0xB7, (size_t)M * K / 8);
    hipMalloc(&sc, K * 4);
    hipMalloc(&sh, K * 4);
    float* Y;                                          // the raw conv output: real fp32 values (the MFMA power draw depends on the data)
    hipMalloc(&Y, (size_t)M * K * 4);
    {
      std::vector<float> hy((size_t)M * K);
      for (auto& v : hy) v = (rand() % 2001 - 1000) / 1000.0f;
      hipMemcpy(Y, hy.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    }
    std::vector<float> hs(K, 1.0f), ht(K, 0.05f);
    hipMemcpy(sc, hs.data(), K * 4, hipMemcpyHostToDevice);
    hipMemcpy(sh, ht.data(), K * 4, hipMemcpyHostToDevice);
    for (int kk : {3072, 1024}) {
      float best = 1e9f, best0 = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_loop_fused, dim3(tiles), dim3(C22::NT), 0, 0, Y, B, bits, sc, sh, out, M, N, kk, N / 256);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_loop<0>, dim3(tiles), dim3(C22::NT), 0, 0, A, B, out, M, N, kk, N / 256);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best0) best0 = ms;
      }
      printf("K = %4d  fused A producer (fp32 y + bits -> BN + ReLU + dropout -> split -> LDS) %.3f ms  %.0f TFLOP/s   |  library loop %.3f ms  %.0f TFLOP/s   (x %.2f)\n",
             kk, best, 2.0 * M * N * kk / best / 1e9, best0, 2.0 * M * N * kk / best0 / 1e9, best / best0);
    }
  }
  {
    // round 6: the per-tile FIXED cost of the library loop (workgroup start, row setup, first DMA latency, drain; the ubench has
    // no epilogue): T(K, rounds) = launch + rounds * (K / 32 * t + X)
    for (int kk : {64, 1024, 3072})
      for (int rounds : {1, 2}) {
        const int mm = rounds * 64 * 256;               // rounds * 64 m-tiles x 4 n-tiles = rounds * 256 tiles
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(k_loop<0>, dim3(rounds * 256), dim3(C22::NT), 0, 0, A, B, out, mm, N, kk, N / 256);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (rep && ms < best) best = ms;
        }
        printf("fixed-cost probe  K = %4d  rounds = %d: %8.1f us   (%.1f us per round)\n", kk, rounds, best * 1e3, best * 1e3 / rounds);
      }
  }
  {
    // the 224 x 256 tile on the same two full rounds: 128 x 4 = 512 tiles over 28,672 rows (operands re-used: A is 32,768 rows)
    const int M2 = 224 * 128;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_loop224, dim3(512), dim3(C22::NT), 0, 0, A, B, out, M2, N, K, N / 256);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("224x256 tile (4 + 3 row blocks per SIMD pair), library loop: %.3f ms   %.0f TFLOP/s algorithmic   %.2f us per K-tile and round\n",
           best, 2.0 * M2 * N * K / best / 1e9, best * 1e3 / 2 / (K / 32));
  }
  return 0;
}
