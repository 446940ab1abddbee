// gemm_tile.h -- 128x128x32 fp32-MFMA "NT" tile engine shared by the matcher kernels.
//   C[i][j] = sum_k A[i][k] * B[j][k]       (A: [M,K] row-major, B: [N,K] row-major)
// 256 threads = 4 waves arranged 2(M) x 2(N); each wave owns a 64x64 sub-tile as 2x2
// v_mfma_f32_32x32x2_f32 accumulators.  K is consumed strictly ascending through one
// accumulator chain per output, i.e. every C[i][j] is the k-ordered fmaf chain from +0 that
// oracle/einx_oracle.c computes (orc_mnn / orc_linear).  Operand tiles are staged through LDS
// with an odd row pitch (33) so the per-lane row-strided ds_read_b32 fragments are
// conflict-free; the next K-slab's global loads are issued before the current slab's MFMAs.
#pragma once
#include "einx_common.h"

namespace einx_gemm {

constexpr int BM = 128, BN = 128, BK = 32, PITCH = BK + 1;
constexpr int LDS_FLOATS = (BM + BN) * PITCH;
constexpr int THREADS = 256;

struct Frag {
  f32x16 acc[2][2];
};

// A rows [i0, i0+128) valid while < Mvalid; B rows [j0, j0+128) valid while < Nvalid.
// K must be a multiple of 4 (rows are 16-byte aligned); K tail beyond a multiple of 32 is
// zero-filled.  lds: LDS_FLOATS floats.
// Optional second A source: columns k >= Ksplit come from A2[i][k - Ksplit] (Ksplit % 32 == 0),
// which evaluates cat([A, A2], -1) @ B^T without materialising the concatenation.
__device__ __forceinline__ void tile_nt(const float* __restrict__ A, int lda, int i0, int Mvalid, const float* __restrict__ B, int ldb,
                                        int j0, int Nvalid, int K, float* lds, Frag& f, const float* __restrict__ A2 = nullptr,
                                        int lda2 = 0, int Ksplit = 0x7fffffff) {
  float* As = lds;
  float* Bs = lds + BM * PITCH;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) f.acc[mt][nt][r] = 0.0f;

  // staging: 128 rows x 8 float4 per operand = 1024 float4 -> 4 per thread per operand
  f32x4 ra[4], rb[4];
  auto issue = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int fidx = tid + i * THREADS;
      const int r = fidx >> 3, c4 = fidx & 7;
      const int k = k0 + c4 * 4;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (i0 + r < Mvalid && k < K)
        va = (k < Ksplit) ? *reinterpret_cast<const f32x4*>(A + (size_t)(i0 + r) * lda + k)
                          : *reinterpret_cast<const f32x4*>(A2 + (size_t)(i0 + r) * lda2 + (k - Ksplit));
      if (j0 + r < Nvalid && k < K) vb = *reinterpret_cast<const f32x4*>(B + (size_t)(j0 + r) * ldb + k);
      ra[i] = va;
      rb[i] = vb;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int fidx = tid + i * THREADS;
      const int r = fidx >> 3, c4 = fidx & 7;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        As[r * PITCH + c4 * 4 + t] = ra[i][t];
        Bs[r * PITCH + c4 * 4 + t] = rb[i][t];
      }
    }
  };
  const int aoff = (wm * 64 + l31) * PITCH + half;
  const int boff = (wn * 64 + l31) * PITCH + half;
  issue(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    commit();
    __syncthreads();
    if (k0 + BK < K) issue(k0 + BK);
    // software-pipelined fragment reads: step kk+1's operands are requested before step kk's MFMAs
    float av[2][2], bv[2][2];
    auto load_frag = [&](int kk, int buf) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) av[buf][mt] = As[aoff + mt * 32 * PITCH + kk * 2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bv[buf][nt] = Bs[boff + nt * 32 * PITCH + kk * 2];
    };
    load_frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      if (kk + 1 < BK / 2) load_frag(kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);  // hipcc would otherwise sink the prefetch next to its use
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          f.acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][mt], bv[kk & 1][nt], f.acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // LDS reusable by the caller's epilogue
}

// element coordinates inside the 128x128 tile for accumulator (mt, nt, r) of this lane
__device__ __forceinline__ int row_of(int mt, int r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave >> 1) * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int col_of(int nt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave & 1) * 64 + nt * 32 + (lane & 31);
}

}  // namespace einx_gemm
