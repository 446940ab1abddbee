// gemm_tile.h -- 128x128x32 fp32-MFMA "NT" tile engine shared by the matcher kernels.
//   C[i][j] = sum_k A[i][k] * B[j][k]       (A: [M,K] row-major, B: [N,K] row-major)
// WAVES waves arranged (WAVES/2)(M) x 2(N); each wave owns a (128/(WAVES/2)) x 64 sub-tile of
// v_mfma_f32_32x32x2_f32 accumulators (8 waves: 32x64 = 1x2 tiles, 4 waves: 64x64 = 2x2).  K is consumed strictly ascending through one
// accumulator chain per output, i.e. every C[i][j] is the k-ordered fmaf chain from +0 that
// oracle/einx_oracle.c computes (orc_mnn / orc_linear).  Operand tiles are staged through LDS
// with an odd row pitch (33) so the per-lane row-strided ds_read_b32 fragments are
// conflict-free; the next K-slab's global loads are issued before the current slab's MFMAs.
#pragma once
#include "einx_common.h"

namespace einx_gemm {

// (measured alternatives that are no longer in the source: 4 waves of 2x2 tiles, BK = 64, fragments of two K-steps per
// ds_read_b64, prefetch distances 2-3, staging ablations -- profiles/r03_notes.md 7-9, tools/experiments/r5_removed_switches.patch)
constexpr int BM = 128, BN = 128, BK = 32, PITCH = BK + 1;
constexpr int LDS_FLOATS = (BM + BN) * PITCH;
constexpr int WAVES = 8;   // waves are arranged (WAVES/2) along M x 2 along N
constexpr int THREADS = WAVES * 64;
constexpr int MT = BM / ((WAVES / 2) * 32);  // 32x32 MFMA tiles per wave along M (4 waves: 2, 8 waves: 1)
constexpr int NT = 2;                        // ... along N (a wave always spans 64 columns)
constexpr int WROWS = MT * 32;               // rows a wave owns
constexpr int C4 = BK / 4;                   // float4 per operand row per K-slab
constexpr int STAGE = 128 * C4 / THREADS;    // float4 per thread per operand per K-slab

struct Frag {
  f32x16 acc[MT][NT];
};

// One tile's operands.  A rows [i0, i0+128) valid while < Mvalid; B rows [j0, j0+128) valid while
// < Nvalid.  K must be a multiple of 4 (rows are 16-byte aligned); a K tail beyond a multiple of
// BK is zero-filled.  Optional second A source: columns k >= Ksplit come from A2[i][k - Ksplit]
// (Ksplit % BK == 0), which evaluates cat([A, A2], -1) @ B^T without materialising the concatenation.
struct Src {
  const float* A;
  const float* A2;
  const float* B;
  int lda, lda2, ldb, i0, Mvalid, j0, Nvalid;
};

// register image of one K-slab of both operands (global -> registers -> LDS)
struct Stage {
  f32x4 ra[STAGE], rb[STAGE];
};

__device__ __forceinline__ void issue_slab(const Src& s, int k0, int K, int Ksplit, Stage& st) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < STAGE; ++i) {
    const int fidx = tid + i * THREADS;
    const int r = fidx / C4, c4 = fidx % C4;
    const int k = k0 + c4 * 4;
    f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
    if (s.i0 + r < s.Mvalid && k < K)
      va = (k < Ksplit) ? *reinterpret_cast<const f32x4*>(s.A + (size_t)(s.i0 + r) * s.lda + k)
                        : *reinterpret_cast<const f32x4*>(s.A2 + (size_t)(s.i0 + r) * s.lda2 + (k - Ksplit));
    if (s.j0 + r < s.Nvalid && k < K) vb = *reinterpret_cast<const f32x4*>(s.B + (size_t)(s.j0 + r) * s.ldb + k);
    st.ra[i] = va;
    st.rb[i] = vb;
  }
}

// Runs one tile.  Precondition: issue_slab(cur, 0, ...) has been called into `st`.  While the last
// K-slab is on the matrix cores the first slab of `next` (if has_next) is requested into `st`, so a
// persistent kernel's next tile starts without an exposed global-memory round trip and its loads
// also fly under the caller's epilogue.  lds: LDS_FLOATS floats; no trailing barrier.
__device__ __forceinline__ void tile_nt_run(const Src& cur, int K, int Ksplit, float* lds, Frag& f, Stage& st, const Src& next,
                                            bool has_next) {
  float* As = lds;
  float* Bs = lds + BM * PITCH;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) f.acc[mt][nt][r] = 0.0f;
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < STAGE; ++i) {
      const int fidx = tid + i * THREADS;
      const int r = fidx / C4, c4 = fidx % C4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        As[r * PITCH + c4 * 4 + t] = st.ra[i][t];
        Bs[r * PITCH + c4 * 4 + t] = st.rb[i][t];
      }
    }
  };
  const int aoff = (wm * WROWS + l31) * PITCH + half;
  const int boff = (wn * 64 + l31) * PITCH + half;
  // Whole tiles (the common case) reload through a predicate-free path: a uniform base pointer that
  // advances with k0 (scalar registers) plus a per-thread element offset that never changes, so a
  // K-slab costs the loads themselves and no vector address arithmetic, compares or selects.
  const bool whole = cur.i0 + BM <= cur.Mvalid && cur.j0 + BN <= cur.Nvalid && K % BK == 0 && (cur.A2 == nullptr || Ksplit % BK == 0);
  // byte offsets of this thread's float4s from the tile's first element (constant over the K loop); the K position is stepped in
  // the wave-uniform buffer descriptor's base (scalar ALU): a slab's loads are `buffer_load_dwordx4 v, v_off, s[rsrc], 0 offen`
  // with no vector address arithmetic at all (round 5: vector instructions displace fp32 MFMA issue, profiles/r05_notes.md 1)
  unsigned offa[STAGE], offa2[STAGE], offb[STAGE];
#pragma unroll
  for (int i = 0; i < STAGE; ++i) {
    const int fidx = tid + i * THREADS;
    const int r = fidx / C4, c4 = fidx % C4;
    offa[i] = (unsigned)(r * cur.lda + c4 * 4) * 4u;
    offa2[i] = (unsigned)(r * cur.lda2 + c4 * 4) * 4u;
    offb[i] = (unsigned)(r * cur.ldb + c4 * 4) * 4u;
  }
  const float* a_tile = cur.A + (size_t)cur.i0 * cur.lda;
  const float* a2_tile = cur.A2 ? cur.A2 + (size_t)cur.i0 * cur.lda2 : nullptr;
  const float* b_tile = cur.B + (size_t)cur.j0 * cur.ldb;
  auto rsrc = [](const float* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7FFFFFFF, 0x00020000); };  // whole tiles: every offset is in range
  auto issue_whole = [&](int k) {
    const __amdgpu_buffer_rsrc_t rb = rsrc(b_tile + k);
    if (k < Ksplit) {
      const __amdgpu_buffer_rsrc_t ra = rsrc(a_tile + k);
#pragma unroll
      for (int i = 0; i < STAGE; ++i) st.ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, offa[i], 0, 0));
    } else {
      const __amdgpu_buffer_rsrc_t ra = rsrc(a2_tile + (k - Ksplit));
#pragma unroll
      for (int i = 0; i < STAGE; ++i) st.ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, offa2[i], 0, 0));
    }
#pragma unroll
    for (int i = 0; i < STAGE; ++i) st.rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, offb[i], 0, 0));
  };
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();  // the previous slab's (or tile's) fragment reads are done
    commit();
    __syncthreads();
    if (k0 + BK < K) {  // in flight under the MFMAs below
      if (whole) issue_whole(k0 + BK);
      else issue_slab(cur, k0 + BK, K, Ksplit, st);
    } else if (has_next) {
      issue_slab(next, 0, K, Ksplit, st);
    }
    // software-pipelined fragment reads: step kk+1's operands are requested before step kk's MFMAs
    constexpr int PF = 1;  // fragment prefetch distance in K-steps
    float av[PF + 1][MT], bv[PF + 1][NT];
    auto load_frag = [&](int kk, int buf) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[buf][mt] = As[aoff + mt * 32 * PITCH + kk * 2];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[buf][nt] = Bs[boff + nt * 32 * PITCH + kk * 2];
    };
#pragma unroll
    for (int kk = 0; kk < PF; ++kk) load_frag(kk, kk % (PF + 1));
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      if (kk + PF < BK / 2) load_frag(kk + PF, (kk + PF) % (PF + 1));
      __builtin_amdgcn_sched_barrier(0);  // hipcc would otherwise sink the prefetch next to its use
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          f.acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk % (PF + 1)][mt], bv[kk % (PF + 1)][nt], f.acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// single-tile form used by the one-tile-per-workgroup kernels
__device__ __forceinline__ void tile_nt(const float* __restrict__ A, int lda, int i0, int Mvalid, const float* __restrict__ B, int ldb,
                                        int j0, int Nvalid, int K, float* lds, Frag& f, const float* __restrict__ A2 = nullptr,
                                        int lda2 = 0, int Ksplit = 0x7fffffff) {
  Src s;
  s.A = A;
  s.A2 = A2;
  s.B = B;
  s.lda = lda;
  s.lda2 = lda2;
  s.ldb = ldb;
  s.i0 = i0;
  s.Mvalid = Mvalid;
  s.j0 = j0;
  s.Nvalid = Nvalid;
  Stage st;
  issue_slab(s, 0, K, Ksplit, st);
  tile_nt_run(s, K, Ksplit, lds, f, st, s, false);
  __syncthreads();  // LDS reusable by the caller's epilogue
}

// row_of(mt, r) = row_base() + row_step(mt, r): lane-dependent part + compile-time part
__device__ __forceinline__ int row_base() {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave >> 1) * WROWS + 4 * (lane >> 5);
}
__host__ __device__ constexpr int row_step(int mt, int r) { return mt * 32 + (r & 3) + 8 * (r >> 2); }
// element coordinates inside the 128x128 tile for accumulator (mt, nt, r) of this lane
__device__ __forceinline__ int row_of(int mt, int r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave >> 1) * WROWS + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int col_of(int nt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave & 1) * 64 + nt * 32 + (lane & 31);
}

}  // namespace einx_gemm
