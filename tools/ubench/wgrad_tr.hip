// Prototype (stand-alone, not part of libvp3d.so): the split-fp16 WEIGHT-GRADIENT GEMM reading the S16 *rows* of dy and of
// the layer input and transposing on the LDS read (ds_read_b64_tr_b16), instead of consuming the transposed copies the
// streaming producers write today (first row).
//
//     dW[na][tap*C_in + ci] = sum_m  A[m][na] * B[m*taps + tap][ci]        A = dy rows [Mk][NA], B = x rows [Mk*taps][C_in]
//
// Both operands are k-major (k = row m): a stage holds 32 rows x 256 channels of each (one 1-KiB LDS-DMA piece per row,
// fully coalesced 4-KB-pitch HBM rows), rows at a 1040-byte LDS pitch.  Fragment of v_mfma_f32_32x32x16_f16 (lane = column
// n = lane % 32, k-half h = lane / 32, 8 k values): two transpose reads of 4 k each; per 16-lane group the 16 source
// addresses are a [4 rows][16 channels] tile.  Rows per read: {0,1,8,9} + 2h (+4 for the second read) of the 16-row
// k-step -- with the pitch == 4 dwords (mod 64 banks) the four rows of a read fall on disjoint banks; the k order inside
// the step is a permutation, identical for both operands, so the products pair up correctly.
//
//   hipcc --offload-arch=gfx950 -O3 wgrad_tr.hip -o wgrad_tr && ./wgrad_tr
// prints the error of a small ragged case against an fp64 reference and the time of the two big wgrad shapes of the
// cfg3 step (today's NT kernel on transposed copies: 438 us / 149 us).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, BK = 32, NW = 8, NT = NW * 64;
constexpr int PITCH = 1040;                 // bytes per LDS row: 256 channels x 4 B + 16
constexpr int OP_B = BK * PITCH;            // one operand of one stage
constexpr int STAGE_B = 2 * OP_B;
constexpr int SMEM_B = 2 * STAGE_B;         // 133,120 B: one workgroup per CU
constexpr int RB = 4, CB = 2;               // wave sub-tile 128 (na) x 64 (nb)

struct Args {
  const float* A;
  const float* B;
  float* part;            // [splits][NA][NB]
  int Mk, lda, ldb, NA, NB, taps, c_in;
  uint32_t a_bytes, b_bytes;
  int m_tiles, n_tiles, pos, splits, kt_per_split;
};

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, int voff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ f16x8 tr8(const char* p) {       // 8 k values of one column: two transpose reads, 4 rows apart
  auto q = (const __attribute__((address_space(3))) v4s*)p;
  auto q2 = (const __attribute__((address_space(3))) v4s*)(p + 4 * PITCH);
  const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)q);
  const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)q2);
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(f16x8, v);
}

__global__ void __launch_bounds__(NT, 2) k_tn_s16(const Args p) {
  __shared__ __attribute__((aligned(16))) char smem[SMEM_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / 4, wn = w % 4;

  const int split = blockIdx.x / p.pos;
  const int bid = blockIdx.x - split * p.pos;
  // equal contiguous XCD shares of the row-major tile list (see tile_of in vp3d_gemm_s16.hip)
  const int per = p.pos >> 3;
  const int L = (bid & 7) * per + (bid >> 3);
  if ((bid >> 3) >= per || L >= p.m_tiles * p.n_tiles) return;
  const int tile_m = L / p.n_tiles, tile_n = L - tile_m * p.n_tiles;
  const int na0 = tile_m * BM, nb0 = tile_n * BN;
  const int tap = nb0 / p.c_in, ci0 = nb0 - tap * p.c_in;

  const int nkt_all = (p.Mk + BK - 1) / BK;
  const int kt_begin = split * p.kt_per_split;
  const int kt_end = min(nkt_all, kt_begin + p.kt_per_split);
  const int nkt = max(0, kt_end - kt_begin);

  f32x16 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- LDS-DMA staging: wave w owns rows 4w .. 4w+3 of both operands; one 1-KiB piece = one row x 256 channels ----
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  int a_vo[4], b_vo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = kt_begin * BK + w * 4 + i;                   // rows >= Mk lie beyond num_records: zeros
    a_vo[i] = (int)(((int64_t)r * p.lda + na0) * 4) + lane * 16;
    b_vo[i] = (int)((((int64_t)r * p.taps + tap) * p.ldb + ci0) * 4) + lane * 16;
  }
  const int a_step = BK * p.lda * 4, b_step = BK * p.taps * p.ldb * 4;
  auto issue = [&](int stage) {
    char* sA = smem + stage * STAGE_B;
#pragma unroll
    for (int i = 0; i < 4; ++i) blds16(rsA, a_vo[i], sA + (w * 4 + i) * PITCH);
#pragma unroll
    for (int i = 0; i < 4; ++i) blds16(rsB, b_vo[i], sA + OP_B + (w * 4 + i) * PITCH);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a_vo[i] += a_step;
      b_vo[i] += b_step;
    }
  };

  // ---- fragment addressing ----
  const int g = lane >> 4, s = lane & 15, h = g >> 1;
  const int rsel = ((s >> 2) & 1) + 8 * ((s >> 3) & 1) + 2 * h;          // {0,1,8,9}[s >> 2] + 2h
  const int off_lane = rsel * PITCH + (g & 1) * 64 + ((s & 3) >> 1) * 32 + ((s & 3) & 1) * 8;
  const int a_off = wm * (RB * 32) * 4 + off_lane;
  const int b_off = OP_B + wn * (CB * 32) * 4 + off_lane;

  if (nkt > 0) {
    issue(0);
    int st = 0;
    for (int it = 0; it < nkt; ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const char* sS = smem + st * STAGE_B;
      // All fragment reads of the tile first, THEN the LDS-DMA of the next tile (other stage), then the MFMAs: hipcc
      // cannot prove that a transpose read does not alias an LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of
      // the first read that follows one -- issued ahead of the reads (as the NT kernel does) the DMA would be waited for
      // before this tile's compute instead of overlapping it.
      f16x8 ah[2][RB], al[2][RB], bh[2][CB], bl[2][CB];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          ah[ks][i] = tr8(sS + a_off + i * 128 + ks * 16 * PITCH);
          al[ks][i] = tr8(sS + a_off + i * 128 + ks * 16 * PITCH + 16);
        }
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          bh[ks][j] = tr8(sS + b_off + j * 128 + ks * 16 * PITCH);
          bl[ks][j] = tr8(sS + b_off + j * 128 + ks * 16 * PITCH + 16);
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

  // ---- raw partial matrix of this K-slice (prototype: straight from the accumulators, 128-B runs) ----
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
        if (na < p.NA && nb < p.NB) out[(int64_t)na * p.NB + nb] = acc[i][j][reg];
      }
}

// fp32 rows -> S16 rows (exponent 0) and the decoded values (what the GEMM really multiplies)
__global__ void k_pack(int64_t groups, const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dec) {
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (int64_t)gridDim.x * blockDim.x) {
    f16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
      const float x = src[gi * 8 + j];
      const _Float16 hv = (_Float16)x;
      hi[j] = hv;
      lo[j] = (_Float16)(x - (float)hv);
      if (dec != nullptr) dec[gi * 8 + j] = (float)hi[j] + (float)lo[j];
    }
    f16x8* d = reinterpret_cast<f16x8*>(dst + gi * 8);
    d[0] = hi;
    d[1] = lo;
  }
}

__global__ void k_fill(int64_t n, float* dst, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    dst[i] = ((float)(x & 0xffffff) / 8388608.0f - 1.0f) * 1.7f;          // uniform in (-1.7, 1.7)
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float run(int Mk, int NA, int taps, int c_in, int splits, bool check) {
  const int NB = taps * c_in;
  const int64_t nA = (int64_t)Mk * NA, nB = (int64_t)Mk * taps * c_in;
  float *A32, *B32, *A16, *B16, *Ad = nullptr, *Bd = nullptr, *part;
  CK(hipMalloc(&A32, nA * 4)); CK(hipMalloc(&B32, nB * 4)); CK(hipMalloc(&A16, nA * 4)); CK(hipMalloc(&B16, nB * 4));
  if (check) { CK(hipMalloc(&Ad, nA * 4)); CK(hipMalloc(&Bd, nB * 4)); }
  CK(hipMalloc(&part, (int64_t)splits * NA * NB * 4));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, nA, A32, 17u);
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, nB, B32, 4242u);
  hipLaunchKernelGGL(k_pack, dim3(2048), dim3(256), 0, 0, nA / 8, A32, A16, Ad);
  hipLaunchKernelGGL(k_pack, dim3(2048), dim3(256), 0, 0, nB / 8, B32, B16, Bd);
  Args a;
  a.A = A16; a.B = B16; a.part = part;
  a.Mk = Mk; a.lda = NA; a.ldb = c_in; a.NA = NA; a.NB = NB; a.taps = taps; a.c_in = c_in;
  a.a_bytes = (uint32_t)(nA * 4); a.b_bytes = (uint32_t)(nB * 4);
  a.m_tiles = (NA + BM - 1) / BM; a.n_tiles = (NB + BN - 1) / BN;
  a.pos = 8 * ((a.m_tiles * a.n_tiles + 7) / 8);
  a.splits = splits;
  const int nkt = (Mk + BK - 1) / BK;
  a.kt_per_split = (nkt + splits - 1) / splits;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < (check ? 1 : 6); ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_tn_s16, dim3(a.pos * splits), dim3(NT), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 || check) best = ms < best ? ms : best;
  }
  CK(hipGetLastError());
  if (check) {
    std::vector<float> hA(nA), hB(nB), hP((size_t)splits * NA * NB);
    CK(hipMemcpy(hA.data(), Ad, nA * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hB.data(), Bd, nB * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hP.data(), part, hP.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    int bad = 0;
    for (int na = 0; na < NA; na += (NA > 512 ? 13 : 3))
      for (int nb = 0; nb < NB; nb += (NB > 2048 ? 17 : 5)) {
        const int tap = nb / c_in, ci = nb % c_in;
        double ref = 0.0, den = 0.0;
        for (int m = 0; m < Mk; ++m) {
          const double x = hA[(size_t)m * NA + na], y = hB[((size_t)m * taps + tap) * c_in + ci];
          ref += x * y;
          den += fabs(x * y);
        }
        double got = 0.0;
        for (int sp = 0; sp < splits; ++sp) got += hP[((size_t)sp * NA + na) * NB + nb];
        const double err = fabs(got - ref) / (den + 1e-30);
        if (err > worst) worst = err;
        if (err > 1e-5 && bad < 5) { printf("   mismatch na=%d nb=%d got %.6f ref %.6f\n", na, nb, got, ref); ++bad; }
      }
    printf("check Mk=%d NA=%d taps=%d C_in=%d splits=%d: worst |err| / sum|a b| = %.3e  (%s)\n", Mk, NA, taps, c_in, splits,
           worst, worst < 1e-6 ? "OK" : "FAIL");
  } else {
    printf("Mk=%6d NA=%5d NB=%5d splits=%2d: %8.3f ms  %7.1f TFLOP/s algorithmic\n", Mk, NA, NB, splits, best,
           2.0 * Mk * NA * NB / best / 1e9);
  }
  CK(hipFree(A32)); CK(hipFree(B32)); CK(hipFree(A16)); CK(hipFree(B16)); CK(hipFree(part));
  if (check) { CK(hipFree(Ad)); CK(hipFree(Bd)); }
  return best;
}

int main(int argc, char** argv) {
  if (argc > 1) {                          // "./wgrad_tr big": correctness at the step's tile geometry (4 x 12 tiles, 16 K-slices)
    run(2000, 1024, 3, 1024, 16, true);
    return 0;
  }
  run(200, 256, 2, 256, 2, true);          // ragged K (200 rows), two taps, two K-slices
  run(1000, 512, 3, 512, 3, true);
  for (int s : {8, 16, 24}) run(27648, 1024, 3, 1024, s, false);      // first block's strided conv (NT today: 438 us)
  for (int s : {8, 16}) run(27648, 1024, 1, 1024, s, false);          // its 1x1 conv (NT today: 149 us)
  return 0;
}
