// match_tiles.h -- tile kernels shared by the MNN matcher (mnn.hip) and LightGlue's final
// assignment (lightglue.hip): similarity tiles on the fp32 matrix cores with fused arg-max /
// softmax statistics / log-assignment epilogues.  See mnn.hip for the design notes.
#pragma once
#include <type_traits>

#include "gemm_tile.h"

namespace einx_match {
namespace {  // internal linkage: each translation unit gets its own copy of the kernels


using namespace einx_gemm;

__device__ __forceinline__ unsigned long long pack_key(float v, int idx) {
  return ((unsigned long long)einx_ordered_key(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
}

struct MnnArgs {
  const float* d0;
  const float* d1;
  const int32_t* n;
  const int32_t* m;
  int cap0, cap1, D;
  unsigned long long* rowkey;  // [B,cap0]
  unsigned long long* colkey;  // [B,cap1]
  float* rowstat;              // [B,cap0,nc64,2]  (max, sumexp) per 64-column chunk
  float* colstat;              // [B,cap1,nr64,2]  per WROWS-row chunk (nr64 = cdiv(cap0, WROWS))
  float* rowlse;               // [B,cap0,2] (max, log-sum-exp)
  float* collse;               // [B,cap1,2]
  float* la;                   // [B,cap0+1,cap1+1]
  int nc64, nr64;
  // LightGlue assignment (lightglue.py:365-377): scores = log_softmax rows + cols + certainties
  const float* cert0;  // [B,cap0] logsigmoid(z0) or null (plain MNN)
  const float* cert1;  // [B,cap1] logsigmoid(z1)
  const float* dust0;  // [B,cap0] logsigmoid(-z0): last column of the assignment
  const float* dust1;  // [B,cap1] logsigmoid(-z1): last row
  // find_nn thresholds (MNN.py:12-22): second-best similarity per row / column as ordered 32-bit keys (MODE 4)
  unsigned* row2;  // [B,cap0]
  unsigned* col2;  // [B,cap1]
};

// MODE 0: arg-max keys.  MODE 1: per-chunk softmax statistics.  MODE 2: write log_assignment.
// MODE 5: MODE 0 and MODE 1 on one visit of the tile, and the raw similarity into a.la (MNN with log_assignment: ONE
//         similarity pass; mnn_lse_kernel + mnn_la_apply_kernel finish the dual softmax in place).
// MODE 6: MODE 0 and MODE 2 on one visit (LightGlue: arg-max of the assignment scores and the log_assignment write).
// MODE 3: write the raw similarity tile to a.la as [B,cap0,cap1] (MNN.py:88 `similarity`).
// MODE 4: second neighbour (topk(2)[1], MNN.py:13-14): per row the maximum over all columns but the arg-max
//         column found by MODE 0 (an equal value at another index counts, as topk returns it), same per column.
// LG: values are LightGlue's assignment scores (needs rowlse/collse from MODE 1 + mnn_lse_kernel).
template <int MODE, bool LG = false>
__global__ __launch_bounds__(THREADS) void mnn_tile_kernel(const MnnArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  // XCD-contiguous order (einx_common.h): the column tiles of one row block, then the row blocks of one pair, share an L2
  const int gx = (int)gridDim.x, gy = (int)gridDim.y;
  int item = xcd_contiguous((int)(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z)), gx * gy * (int)gridDim.z);
  const int jt = item % gx;
  item /= gx;
  const int it = item % gy, b = item / gy;
  const int n = min(a.n[b], a.cap0), m = min(a.m[b], a.cap1);
  const int i0 = it * BM, j0 = jt * BN;
  if (i0 >= n || j0 >= m) return;
  Frag f;
  tile_nt(a.d0 + (size_t)b * a.cap0 * a.D, a.D, i0, n, a.d1 + (size_t)b * a.cap1 * a.D, a.D, j0, m, a.D, lds, f);
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5;
  const float NEG = -einx_u2f(0x7f800000u);

  if (LG && MODE != 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = min(i0 + row_of(mt, r), n - 1);
        const float rm = a.rowlse[((size_t)b * a.cap0 + i) * 2], rl = a.rowlse[((size_t)b * a.cap0 + i) * 2 + 1];
        const float c0 = a.cert0[(size_t)b * a.cap0 + i];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int j = min(j0 + col_of(nt), m - 1);
          const float cm = a.collse[((size_t)b * a.cap1 + j) * 2], cl = a.collse[((size_t)b * a.cap1 + j) * 2 + 1];
          const float sv = f.acc[mt][nt][r];
          f.acc[mt][nt][r] = (((sv - rm) - rl) + ((sv - cm) - cl)) + (c0 + a.cert1[(size_t)b * a.cap1 + j]);
        }
      }
  }

  if ((MODE == 0 || MODE == 5 || MODE == 6) && MT == 1) {
    // Round 4: the arg-max epilogue as packed 64-bit keys (ordered value << 32 | ~index: the maximum key IS the first maximum).
    //  * columns: the lane scans its 16 rows, the two lane halves meet once;
    //  * rows: instead of 16 separate 32-lane reductions of (value, index) pairs (160 cross-lane moves, 16 atomics of 2
    //    lanes), a butterfly reduce-scatter -- at distance 16, 8, 4, 2 a lane keeps one half of its rows and hands the other
    //    half to its partner (8 + 4 + 2 + 1 keys move), distance 1 completes the row -- after which every lane pair owns ONE
    //    row: 30 cross-lane key moves, no LDS address arithmetic (ds_swizzle);
    //  * the keys of the tile's 128 columns (4 row-waves) and 128 rows (2 column-waves) meet in LDS, so the tile issues
    //    256 global atomics instead of 768.
    unsigned long long* kst = reinterpret_cast<unsigned long long*>(lds);  // [4][128] column keys, then [2][128] row keys
    const int wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, l = lane & 31;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float bv = NEG;
      int bi = 0x7fffffff;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + row_of(0, r);
        const float v = f.acc[0][nt][r];
        if (i < n && (v > bv || bi == 0x7fffffff)) {  // ascending i, strict >: the first maximum
          bv = v;
          bi = i;
        }
      }
      unsigned long long key = bi != 0x7fffffff ? pack_key(bv, bi) : 0ull;
      const unsigned long long ok = ((unsigned long long)(unsigned)__shfl_xor((int)(key >> 32), 32, 64) << 32) | (unsigned)__shfl_xor((int)key, 32, 64);
      key = ok > key ? ok : key;  // the other lane half holds the interleaved rows of the same column
      if (half == 0) kst[wm * 128 + wn * 64 + nt * 32 + l] = key;
    }
    {
      unsigned long long key[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ja = j0 + col_of(0), jb = j0 + col_of(1);
        const float va = f.acc[0][0][r], vb = f.acc[0][1][r];
        const bool tb = jb < m && (!(ja < m) || vb > va);  // strict >: the lower column wins a tie
        key[r] = (ja < m || jb < m) ? pack_key(tb ? vb : va, tb ? jb : ja) : 0ull;
      }
      auto swz = [](unsigned long long v, auto pat) -> unsigned long long {
        constexpr int P = decltype(pat)::value;
        return ((unsigned long long)(unsigned)__builtin_amdgcn_ds_swizzle((int)(v >> 32), P) << 32) | (unsigned)__builtin_amdgcn_ds_swizzle((int)v, P);
      };
      auto step = [&](unsigned long long* k, int cnt, bool up, auto pat) {  // k[0..cnt) <- reduce-scatter of k[0..2cnt) with lane ^ distance
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < cnt) {
            const unsigned long long keep = up ? k[q + cnt] : k[q], send = up ? k[q] : k[q + cnt];
            const unsigned long long got = swz(send, pat);
            k[q] = got > keep ? got : keep;
          }
      };
      step(key, 8, (l & 16) != 0, std::integral_constant<int, (16 << 10) | 0x1F>());
      step(key, 4, (l & 8) != 0, std::integral_constant<int, (8 << 10) | 0x1F>());
      step(key, 2, (l & 4) != 0, std::integral_constant<int, (4 << 10) | 0x1F>());
      step(key, 1, (l & 2) != 0, std::integral_constant<int, (2 << 10) | 0x1F>());
      const unsigned long long o = swz(key[0], std::integral_constant<int, (1 << 10) | 0x1F>());
      const unsigned long long k = o > key[0] ? o : key[0];
      const int r = ((l >> 4) & 1) * 8 + ((l >> 3) & 1) * 4 + ((l >> 2) & 1) * 2 + ((l >> 1) & 1);  // the row this lane pair ended up with
      if ((l & 1) == 0) kst[512 + wn * 128 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = k;
    }
    __syncthreads();
    {
      const int t = threadIdx.x;
      if (t < 128) {
        unsigned long long k = kst[t];
#pragma unroll
        for (int w = 1; w < 4; ++w) k = kst[w * 128 + t] > k ? kst[w * 128 + t] : k;
        if (j0 + t < m && k) atomicMax(&a.colkey[(size_t)b * a.cap1 + j0 + t], k);
      } else if (t < 256) {
        const int r = t - 128;
        const unsigned long long ka = kst[512 + r], kb = kst[512 + 128 + r];
        const unsigned long long k = kb > ka ? kb : ka;
        if (i0 + r < n && k) atomicMax(&a.rowkey[(size_t)b * a.cap0 + i0 + r], k);
      }
    }
  } else if (MODE == 0 || MODE == 5 || MODE == 6) {
    // ---- column arg-max over this wave's 64 rows (ascending i, strict > keeps the first) ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = j0 + col_of(nt);
      float bv = NEG;
      int bi = 0x7fffffff;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + row_of(mt, r);
          const float v = f.acc[mt][nt][r];
          if (i < n && (v > bv || bi == 0x7fffffff)) {
            bv = v;
            bi = i;
          }
        }
      // the other lane-half holds the interleaved rows of the same column
      const float ov = __shfl_xor(bv, 32, 64);
      const int oi = __shfl_xor(bi, 32, 64);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
        bv = ov;
        bi = oi;
      }
      if (half == 0 && j < m && bi != 0x7fffffff) atomicMax(&a.colkey[(size_t)b * a.cap1 + j], pack_key(bv, bi));
    }
    // ---- row arg-max over this wave's 64 columns ---------------------------------------------
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + row_of(mt, r);
        float bv = NEG;
        int bj = 0x7fffffff;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int j = j0 + col_of(nt);
          const float v = f.acc[mt][nt][r];
          if (j < m && (bj == 0x7fffffff || v > bv)) {
            bv = v;
            bj = j;
          }
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
          const float ov = __shfl_xor(bv, off, 64);
          const int oj = __shfl_xor(bj, off, 64);
          if (oj != 0x7fffffff && (bj == 0x7fffffff || ov > bv || (ov == bv && oj < bj))) {
            bv = ov;
            bj = oj;
          }
        }
        if ((lane & 31) == 0 && i < n && bj != 0x7fffffff) atomicMax(&a.rowkey[(size_t)b * a.cap0 + i], pack_key(bv, bj));
      }
  }
  if (MODE == 1 || MODE == 5) {
    const int wave = threadIdx.x >> 6;
    const int cchunk = (j0 >> 6) + (wave & 1), rchunk = i0 / WROWS + (wave >> 1);
    // rows: (max, sum exp(v-max)) over this wave's 64 columns
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + row_of(mt, r);
        float v0 = (j0 + col_of(0) < m) ? f.acc[mt][0][r] : NEG;
        float v1 = (j0 + col_of(1) < m) ? f.acc[mt][1][r] : NEG;
        float mx = fmaxf(v0, v1);
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float s = 0.0f;
        if (v0 != NEG) s += einx_expf(v0 - mx);
        if (v1 != NEG) s += einx_expf(v1 - mx);
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if ((lane & 31) == 0 && i < n && cchunk * 64 < m) {
          float* o = a.rowstat + (((size_t)b * a.cap0 + i) * a.nc64 + cchunk) * 2;
          o[0] = mx;
          o[1] = s;
        }
      }
    // columns: over this wave's 64 rows
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = j0 + col_of(nt);
      float mx = NEG;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (i0 + row_of(mt, r) < n) mx = fmaxf(mx, f.acc[mt][nt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float s = 0.0f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (i0 + row_of(mt, r) < n) s += einx_expf(f.acc[mt][nt][r] - mx);
      s += __shfl_xor(s, 32, 64);
      if (half == 0 && j < m && rchunk * WROWS < n) {
        float* o = a.colstat + (((size_t)b * a.cap1 + j) * a.nr64 + rchunk) * 2;
        o[0] = mx;
        o[1] = s;
      }
    }
    if (MODE == 5) {  // the raw similarity goes to the log_assignment buffer; mnn_la_apply_kernel turns it into the dual softmax in place
      const size_t pitch = (size_t)a.cap1 + 1;
      float* la = a.la + (size_t)b * (a.cap0 + 1) * pitch;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + row_of(mt, r);
          if (i >= n) continue;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int j = j0 + col_of(nt);
            if (j < m) la[(size_t)i * pitch + j] = f.acc[mt][nt][r];
          }
        }
    }
  } else if (MODE == 4) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = j0 + col_of(nt);
      const int besti = j < m ? (int)(0xFFFFFFFFu - (unsigned)(a.colkey[(size_t)b * a.cap1 + j] & 0xFFFFFFFFull)) : -1;
      unsigned key = 0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + row_of(mt, r);
          if (i < n && i != besti) key = max(key, einx_ordered_key(f.acc[mt][nt][r]));
        }
      key = max(key, (unsigned)__shfl_xor((int)key, 32, 64));
      if (half == 0 && j < m && key) atomicMax(&a.col2[(size_t)b * a.cap1 + j], key);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + row_of(mt, r);
        const int bestj = i < n ? (int)(0xFFFFFFFFu - (unsigned)(a.rowkey[(size_t)b * a.cap0 + i] & 0xFFFFFFFFull)) : -1;
        unsigned key = 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int j = j0 + col_of(nt);
          if (j < m && j != bestj) key = max(key, einx_ordered_key(f.acc[mt][nt][r]));
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, off, 64));
        if ((lane & 31) == 0 && i < n && key) atomicMax(&a.row2[(size_t)b * a.cap0 + i], key);
      }
  } else if (MODE == 3) {
    float* sim = a.la + (size_t)b * a.cap0 * a.cap1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + row_of(mt, r);
        if (i >= n) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int j = j0 + col_of(nt);
          if (j < m) sim[(size_t)i * a.cap1 + j] = f.acc[mt][nt][r];
        }
      }
  }
  if (MODE == 2 || MODE == 6) {
    const size_t pitch = (size_t)a.cap1 + 1;
    float* la = a.la + (size_t)b * (a.cap0 + 1) * pitch;
    // the (max, lse) pairs of the lane's columns and rows are requested together from clamped addresses: loads guarded per
    // element are serialised by hipcc (one L2 round trip each, 16 x 6 in a row -- longer than the tile's K loop)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 cst[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      cst[nt] = *reinterpret_cast<const f32x2*>(a.collse + ((size_t)b * a.cap1 + min(j0 + col_of(nt), m - 1)) * 2);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x2 rst[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        rst[r] = *reinterpret_cast<const f32x2*>(a.rowlse + ((size_t)b * a.cap0 + min(i0 + row_of(mt, r), n - 1)) * 2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + row_of(mt, r);
        if (i >= n) continue;
        const float rm = rst[r][0], rl = rst[r][1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int j = j0 + col_of(nt);
          if (j >= m) continue;
          const float cm = cst[nt][0], cl = cst[nt][1];
          const float sv = f.acc[mt][nt][r];
          la[(size_t)i * pitch + j] = LG ? sv : ((sv - rm) - rl) + ((sv - cm) - cl);
        }
      }
    }
  }
}

// (max, sum exp(v - max)) of `chunks` per-chunk pairs: the maximum first, then the sum in ascending chunk order.  The pairs are
// requested eight at a time from clamped addresses (a loop of single dependent loads is one memory round trip per chunk).
__device__ __forceinline__ void lse_merge(const float* st, int chunks, float& mx, float& s) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  mx = -einx_u2f(0x7f800000u);
  for (int c0 = 0; c0 < chunks; c0 += 8) {
    f32x2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x2*>(st + 2 * min(c0 + u, chunks - 1));
#pragma unroll
    for (int u = 0; u < 8; ++u) mx = fmaxf(mx, v[u][0]);  // a clamped duplicate cannot change a maximum
  }
  s = 0.0f;
  for (int c0 = 0; c0 < chunks; c0 += 8) {
    f32x2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x2*>(st + 2 * min(c0 + u, chunks - 1));
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (c0 + u < chunks) s = fmaf(v[u][1], einx_expf(v[u][0] - mx), s);
  }
}

// merge per-chunk (max, sumexp) -> (max, log-sum-exp); also zero the dustbin row/column of la
__global__ void mnn_lse_kernel(const MnnArgs a) {
  const int b = blockIdx.y;
  const int n = min(a.n[b], a.cap0), m = min(a.m[b], a.cap1);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const int chunks = (m + 63) / 64;
    const float* st = a.rowstat + ((size_t)b * a.cap0 + t) * a.nc64 * 2;
    float mx, s;
    lse_merge(st, chunks, mx, s);
    a.rowlse[((size_t)b * a.cap0 + t) * 2] = mx;
    a.rowlse[((size_t)b * a.cap0 + t) * 2 + 1] = einx_logf(s);
  }
  if (t < m) {
    const int chunks = (n + WROWS - 1) / WROWS;
    const float* st = a.colstat + ((size_t)b * a.cap1 + t) * a.nr64 * 2;
    float mx, s;
    lse_merge(st, chunks, mx, s);
    a.collse[((size_t)b * a.cap1 + t) * 2] = mx;
    a.collse[((size_t)b * a.cap1 + t) * 2 + 1] = einx_logf(s);
  }
  if (a.la) {
    const size_t pitch = (size_t)a.cap1 + 1;
    float* la = a.la + (size_t)b * (a.cap0 + 1) * pitch;
    if (t <= m) la[(size_t)n * pitch + t] = (a.dust1 && t < m) ? a.dust1[(size_t)b * a.cap1 + t] : 0.0f;
    if (t <= n) la[(size_t)t * pitch + m] = (a.dust0 && t < n) ? a.dust0[(size_t)b * a.cap0 + t] : 0.0f;
  }
}

// log_assignment[i][j] = log_softmax(sim, -1) + log_softmax(sim, -2) (MNN.py:97-99) in place on the raw similarity that
// mnn_tile_kernel<5> left in the buffer: the same expression on the same values as mnn_tile_kernel<2>, without a third
// similarity pass.  One thread per element, rows contiguous along j.
__global__ __launch_bounds__(256) void mnn_la_apply_kernel(const MnnArgs a) {
  const int b = blockIdx.z;
  const int n = min(a.n[b], a.cap0), m = min(a.m[b], a.cap1);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const size_t pitch = (size_t)a.cap1 + 1;
  const float cm = a.collse[((size_t)b * a.cap1 + j) * 2], cl = a.collse[((size_t)b * a.cap1 + j) * 2 + 1];
  for (int i = blockIdx.y; i < n; i += gridDim.y) {  // one row per workgroup unless cap0 exceeds the grid's y limit
    float* p = a.la + (size_t)b * (a.cap0 + 1) * pitch + (size_t)i * pitch + j;
    const float rm = a.rowlse[((size_t)b * a.cap0 + i) * 2], rl = a.rowlse[((size_t)b * a.cap0 + i) * 2 + 1];
    const float sv = *p;
    *p = ((sv - rm) - rl) + ((sv - cm) - cl);
  }
}

// keys -> matches, mutual check, scores (MNN.py:25-32, :100-101)
__global__ void mnn_finalize_kernel(const unsigned long long* rowkey, const unsigned long long* colkey, const int32_t* nn,
                                    const int32_t* mm, int cap0, int cap1, int64_t* m0, int64_t* m1, float* s0, float* s1) {
  const int b = blockIdx.y;
  const int n = min(nn[b], cap0), m = min(mm[b], cap1);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long* rk = rowkey + (size_t)b * cap0;
  const unsigned long long* ck = colkey + (size_t)b * cap1;
  if (t < cap0) {
    int64_t r = -1;
    if (t < n && m > 0) {
      const int j = (int)(0xFFFFFFFFu - (unsigned)(rk[t] & 0xFFFFFFFFull));
      const int back = (int)(0xFFFFFFFFu - (unsigned)(ck[j] & 0xFFFFFFFFull));
      if (back == t) r = j;
    }
    m0[(size_t)b * cap0 + t] = r;
    s0[(size_t)b * cap0 + t] = r > -1 ? 1.0f : 0.0f;
  }
  if (t < cap1) {
    int64_t r = -1;
    if (t < m && n > 0) {
      const int i = (int)(0xFFFFFFFFu - (unsigned)(ck[t] & 0xFFFFFFFFull));
      const int back = (int)(0xFFFFFFFFu - (unsigned)(rk[i] & 0xFFFFFFFFull));
      if (back == t) r = i;
    }
    m1[(size_t)b * cap1 + t] = r;
    s1[(size_t)b * cap1 + t] = r > -1 ? 1.0f : 0.0f;
  }
}

// find_nn with thresholds (MNN.py:12-22) + mutual check on the masked matches (:25-32)
__device__ __forceinline__ int thresh_match(unsigned long long key, unsigned second, int use_ratio, float ratio_sq, int use_dist, float dist_sq) {
  const int idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  const float d0 = 2.0f * (1.0f - einx_ordered_unkey((unsigned)(key >> 32)));
  bool ok = true;
  if (use_ratio && second) ok = ok && (d0 <= ratio_sq * (2.0f * (1.0f - einx_ordered_unkey(second))));
  if (use_dist) ok = ok && (d0 <= dist_sq);
  return ok ? idx : -1;
}

__global__ void mnn_finalize_thresh_kernel(const unsigned long long* rowkey, const unsigned long long* colkey, const unsigned* row2,
                                           const unsigned* col2, const int32_t* nn, const int32_t* mm, int cap0, int cap1, int use_ratio,
                                           float ratio_sq, int use_dist, float dist_sq, int64_t* m0, int64_t* m1, float* s0, float* s1) {
  const int b = blockIdx.y;
  const int n = min(nn[b], cap0), m = min(mm[b], cap1);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long* rk = rowkey + (size_t)b * cap0;
  const unsigned long long* ck = colkey + (size_t)b * cap1;
  const unsigned* r2 = row2 + (size_t)b * cap0;
  const unsigned* c2 = col2 + (size_t)b * cap1;
  if (t < cap0) {
    int64_t r = -1;
    if (t < n && m > 0) {
      const int j = thresh_match(rk[t], r2[t], use_ratio, ratio_sq, use_dist, dist_sq);
      if (j >= 0 && thresh_match(ck[j], c2[j], use_ratio, ratio_sq, use_dist, dist_sq) == t) r = j;
    }
    m0[(size_t)b * cap0 + t] = r;
    s0[(size_t)b * cap0 + t] = r > -1 ? 1.0f : 0.0f;
  }
  if (t < cap1) {
    int64_t r = -1;
    if (t < m && n > 0) {
      const int i = thresh_match(ck[t], c2[t], use_ratio, ratio_sq, use_dist, dist_sq);
      if (i >= 0 && thresh_match(rk[i], r2[i], use_ratio, ratio_sq, use_dist, dist_sq) == t) r = i;
    }
    m1[(size_t)b * cap1 + t] = r;
    s1[(size_t)b * cap1 + t] = r > -1 ? 1.0f : 0.0f;
  }
}

// ascending-i compaction of matched keypoints; one 1024-thread workgroup per pair
__global__ __launch_bounds__(1024) void gather_matches_kernel(const float* k0, const float* k1, const int64_t* m0, const int32_t* nn,
                                                              int cap0, int cap1, int cols, float* o0, float* o1, int32_t* nmatch) {
  __shared__ int scratch[17];
  const int b = blockIdx.x;
  const int n = min(nn[b], cap0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    int64_t j = -1;
    if (i < n) j = m0[(size_t)b * cap0 + i];
    const int v = j > -1 ? 1 : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int w = 0; w < 16; ++w) {
        const int t = scratch[w];
        scratch[w] = run;
        run += t;
      }
      scratch[16] = run;
    }
    __syncthreads();
    const int pos = base + scratch[wave] + incl - v;
    const int total = scratch[16];
    __syncthreads();
    if (v) {
      for (int c = 0; c < cols; ++c) {
        o0[((size_t)b * cap0 + pos) * cols + c] = k0[((size_t)b * cap0 + i) * 3 + c];
        o1[((size_t)b * cap0 + pos) * cols + c] = k1[((size_t)b * cap1 + (int)j) * 3 + c];
      }
    }
    base += total;
  }
  if (tid == 0) nmatch[b] = base;
}

// mnn_finalize_kernel + gather_matches_kernel as ONE launch (one workgroup per pair; round 6: a launch less on the matcher's
// latency-bound tail): the mutual check of rows / columns t, then -- from the same registers -- the order-preserving compaction of
// the matched keypoints.  Same values as the two launches.
__global__ __launch_bounds__(1024) void mnn_finalize_gather_kernel(const unsigned long long* rowkey, const unsigned long long* colkey, const int32_t* nn,
                                                                   const int32_t* mm, int cap0, int cap1, int64_t* m0, int64_t* m1, float* s0, float* s1,
                                                                   const float* k0, const float* k1, int cols, float* o0, float* o1, int32_t* nmatch) {
  __shared__ int scratch[17];
  const int b = blockIdx.x;
  const int n = min(nn[b], cap0), m = min(mm[b], cap1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long* rk = rowkey + (size_t)b * cap0;
  const unsigned long long* ck = colkey + (size_t)b * cap1;
  const int mx = cap0 > cap1 ? cap0 : cap1;
  int base = 0;
  for (int i0 = 0; i0 < mx; i0 += 1024) {
    const int t = i0 + tid;
    int64_t r0 = -1;
    if (t < cap0) {
      if (t < n && m > 0) {
        const int j = (int)(0xFFFFFFFFu - (unsigned)(rk[t] & 0xFFFFFFFFull));
        const int back = (int)(0xFFFFFFFFu - (unsigned)(ck[j] & 0xFFFFFFFFull));
        if (back == t) r0 = j;
      }
      m0[(size_t)b * cap0 + t] = r0;
      s0[(size_t)b * cap0 + t] = r0 > -1 ? 1.0f : 0.0f;
    }
    if (t < cap1) {
      int64_t r1 = -1;
      if (t < m && n > 0) {
        const int i = (int)(0xFFFFFFFFu - (unsigned)(ck[t] & 0xFFFFFFFFull));
        const int back = (int)(0xFFFFFFFFu - (unsigned)(rk[i] & 0xFFFFFFFFull));
        if (back == t) r1 = i;
      }
      m1[(size_t)b * cap1 + t] = r1;
      s1[(size_t)b * cap1 + t] = r1 > -1 ? 1.0f : 0.0f;
    }
    const int v = (t < n && r0 > -1) ? 1 : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int w = 0; w < 16; ++w) {
        const int u = scratch[w];
        scratch[w] = run;
        run += u;
      }
      scratch[16] = run;
    }
    __syncthreads();
    const int pos = base + scratch[wave] + incl - v;
    const int total = scratch[16];
    __syncthreads();
    if (v) {
      for (int c = 0; c < cols; ++c) {
        o0[((size_t)b * cap0 + pos) * cols + c] = k0[((size_t)b * cap0 + t) * 3 + c];
        o1[((size_t)b * cap0 + pos) * cols + c] = k1[((size_t)b * cap1 + (int)r0) * 3 + c];
      }
    }
    base += total;
  }
  if (tid == 0) nmatch[b] = base;
}

// rows [b, 0:counts[b]] of two padded [B,cap,width] arrays -> consecutive rows of two flat arrays
// (pair b starts at sum(counts[:b])): lets the host cut per-pair views with one split call
__global__ __launch_bounds__(256) void compact_rows_kernel(const float* s0, const float* s1, const int32_t* counts, int B, int cap, int width,
                                                           float* d0, float* d1) {
  __shared__ int part[256];
  const int b = blockIdx.x;
  int acc = 0;
  for (int i = threadIdx.x; i < b; i += 256) acc += min(counts[i], cap);
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  const size_t base = (size_t)part[0] * width;
  const int n = min(counts[b], cap) * width;
  const size_t src = (size_t)b * cap * width;
  for (int i = threadIdx.x; i < n; i += 256) {
    d0[base + i] = s0[src + i];
    d1[base + i] = s1[src + i];
  }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }


}  // namespace
}  // namespace einx_match
