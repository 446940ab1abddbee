// lightglue.hip -- LightGlue inference on gfx950 (fp32 end to end, fp32 matrix cores).
//
// Per layer: SelfBlock (Wqkv GEMM -> rotary -> fused softmax(QK^T/8)V -> out_proj GEMM ->
// concat-free FFN GEMM -> LayerNorm+GELU -> FFN GEMM + residual) on both sides, then CrossBlock
// (to_qk / to_v GEMMs, two fused attentions sharing the 1/8 scale, to_out, same FFN).  After the
// last layer MatchAssignment: final_proj/d^(1/4), similarity tiles with double log-softmax +
// matchability (match_tiles.h), mutual filter.  Everything for a batch of pairs is enqueued on one
// stream with device-side keypoint counts; no host synchronisation.
//
// Attention is flash-style per (pair, head, 128 queries): S^T = K Q^T on the matrix cores with the
// query on the MFMA column so that (a) the online-softmax state (running max / sum) is per lane
// and (b) the probability registers are directly the B operand of the PV product O^T = V^T P^T --
// no LDS round trip for P.  K/V blocks of 64 keys are staged through LDS (K with an odd pitch).
//
// Replaces (reference file:line): core/modules/matchers/lightglue.py:137-148 (normalize_keypoints),
// :161-174 (posenc), :151-158 (rotary), :240-272 (SelfBlock), :275-330 (CrossBlock), :365-418
// (assignment + filter_matches), :522-716 (forward).
#include <stdlib.h>

#include "match_tiles.h"

using namespace einx_gemm;
using namespace einx_match;

namespace {

// The shipped configuration (every EI-Nexus YAML): descriptor_dim 256 = 4 heads x 64.  Its kernels are instantiated with these
// as compile-time constants; any other (heads, head_dim in {32, 64, 128}) configuration of lightglue.py:456-461 runs the same
// kernels with the widths taken from the arguments.
constexpr int D = 256;   // descriptor_dim
constexpr int DH = 64;   // head dim

__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ------------------------------------------------------------------------------------------
// positional encoding: enc[b][i][0..dh-1] = cos, [dh..2dh-1] = sin of Wr . normalised keypoint (Wr: [dh/2, 2]),
// each frequency repeated twice (repeat_interleave(2)).
// ------------------------------------------------------------------------------------------
__global__ void lg_posenc_kernel(const float* kpts, const int32_t* cnt, int cap, float s0, float s1, const float* Wr, float* enc, int dh) {
  const int b = blockIdx.y;
  const int n = min(cnt[b], cap);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = dh >> 1;
  const int i = t / nf, f = t - i * nf;
  if (i >= n) return;
  const float sh0 = s0 / 2.0f, sh1 = s1 / 2.0f;
  const float sc = fmaxf(s0, s1) / 2.0f;
  const float* kp = kpts + ((size_t)b * cap + i) * 3;
  const float k0 = (kp[0] - sh0) / sc, k1 = (kp[1] - sh1) / sc;
  float p = fmaf(k0, Wr[f * 2 + 0], 0.0f);
  p = fmaf(k1, Wr[f * 2 + 1], p);
  float sn, cs;
  einx_sincosf(p, &sn, &cs);
  float* e = enc + ((size_t)b * cap + i) * (2 * dh);
  e[2 * f] = cs;
  e[2 * f + 1] = cs;
  e[dh + 2 * f] = sn;
  e[dh + 2 * f + 1] = sn;
}

// ------------------------------------------------------------------------------------------
// batched Linear: Y[b,i,:] = cat(X[b,i,:], X2[b,i,:]) @ W^T + bias  (+ epilogue)
// ------------------------------------------------------------------------------------------
enum { EPI_BIAS = 0, EPI_DIV = 1, EPI_RESID = 2, EPI_ROPE = 3, EPI_ROPE_ANY = 4 };  // ROPE: d = 256, 64-wide heads; ROPE_ANY: g.d / g.dh

struct GemmArgs {
  const float* X;
  const float* X2;
  const float* W;
  const float* bias;
  float* Y;
  const int32_t* cnt;  // per-batch row counts, or null: every batch has `cap` rows
  int cap, ldx, ldx2, Ksplit, K, N, ldy, B;
  float div;
  // EPI_ROPE / EPI_ROPE_ANY (the Wqkv projection of SelfBlock, lightglue.py:252-272): W rows are laid out (head, dim, 3);
  // the tile list walks them as three d-column blocks q | k | v (row stride 3 in W), applies the cached
  // rotary encoding to q and k in the epilogue and writes the three [B,cap,d] buffers directly
  const float* enc;  // [B,cap,2 dh]: cos(dh) | sin(dh)
  float *Yq, *Yk, *Yv;
  int d, dh;  // read by EPI_ROPE_ANY only (dh even: rotary pairs are adjacent columns)
};

// Persistent workgroups: the launch holds as many workgroups as the chip runs at once (a multiple
// of 8) and each walks a list of 128x128 tiles.  Workgroups w and w+8 share an XCD (round-robin
// dispatch), so the tile list of XCD x is built from whole A-row groups: the tilesN tiles that read
// the same 128 rows of X are computed on one XCD at about the same time and X is fetched into that
// L2 once.  The next tile's first K-slab is requested during the current tile's last K-slab, so its
// latency and the epilogue's stores overlap instead of serialising per tile.
template <int EPI>
__global__ __launch_bounds__(THREADS, 2 * THREADS / 256) void lg_gemm_kernel(const GemmArgs g) {  // registers for two resident workgroups per CU
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  constexpr bool ROPE = EPI == EPI_ROPE || EPI == EPI_ROPE_ANY;
  const int rd = EPI == EPI_ROPE ? D : g.d;              // width of the q | k | v blocks (ROPE only)
  const int rdh = EPI == EPI_ROPE ? DH : g.dh;
  const int tpb = EPI == EPI_ROPE ? D / BN : einx_cdiv(rd, BN);  // tiles per q | k | v block
  const int tilesN = ROPE ? 3 * tpb : einx_cdiv(g.N, BN), tilesM = einx_cdiv(g.cap, BM);
  const int groups = tilesM * g.B;                       // A-row groups
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int local_tiles = einx_cdiv(groups, 8) * tilesN;  // tiles in this XCD's list
  // tile L of this XCD's list -> operands; false when the tile does not exist / has no valid rows
  auto locate = [&](int L, Src& s, int& bb, int& nn) {
    const int grp = (L / tilesN) * 8 + xcd;
    if (grp >= groups) return false;
    const int b = grp / tilesM, ti = grp % tilesM;
    const int n = g.cnt ? min(g.cnt[b], g.cap) : g.cap;
    if (ti * BM >= n) return false;
    s.A = g.X + (size_t)b * g.cap * g.ldx;
    s.A2 = g.X2 ? g.X2 + (size_t)b * g.cap * g.ldx2 : nullptr;
    s.lda = g.ldx;
    s.lda2 = g.ldx2;
    s.i0 = ti * BM;
    s.Mvalid = n;
    if (ROPE) {  // tile tj: block t = tj / tpb of (q, k, v), columns hc0.. of that block = W rows (hc0 + r) * 3 + t
      const int tj = L % tilesN;
      s.B = g.W + (size_t)(tj / tpb) * g.K;
      s.ldb = 3 * g.K;
      s.j0 = (tj % tpb) * BN;
      s.Nvalid = rd;
    } else {
      s.B = g.W;
      s.ldb = g.K;
      s.j0 = (L % tilesN) * BN;
      s.Nvalid = g.N;
    }
    bb = b;
    nn = n;
    return true;
  };
  Src cur, nxt;
  int b = 0, n = 0, nb = 0, nn = 0;
  int L = slot;
  while (L < local_tiles && !locate(L, cur, b, n)) L += slots;
  if (L >= local_tiles) return;
  Stage st;
  issue_slab(cur, 0, g.K, g.Ksplit, st);
  for (;;) {
    int Ln = L + slots;
    while (Ln < local_tiles && !locate(Ln, nxt, nb, nn)) Ln += slots;
    const bool more = Ln < local_tiles;
    Frag f;
    tile_nt_run(cur, g.K, g.Ksplit, lds, f, st, nxt, more);
    const int i0 = cur.i0, j0 = cur.j0;
    if (ROPE) {
      const int t = (int)((cur.B - g.W) / g.K);  // 0: q, 1: k, 2: v
      float* Yt = (t == 0 ? g.Yq : t == 1 ? g.Yk : g.Yv) + (size_t)b * g.cap * rd;
      const float* encb = g.enc + (size_t)b * g.cap * (2 * rdh);
      const bool odd = threadIdx.x & 1;  // column parity == lane parity (col_of(nt) = 64*wn + 32*nt + lane%32)
      float bq[NT];
      int hc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        hc[nt] = j0 + col_of(nt);
        bq[nt] = (EPI == EPI_ROPE || hc[nt] < rd) ? g.bias[hc[nt] * 3 + t] : 0.0f;
      }
      // rows go in batches of 4: the 16 cos/sin loads of a batch are issued together, ahead of its stores
      // (the compiler cannot move a load above an earlier store through plain float pointers)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
          float cs[4][NT], sn[4][NT];
          if (t < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = i0 + row_of(mt, r0 + r);
              const float* e = encb + (size_t)(i < n ? i : n - 1) * (2 * rdh);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const int hd = EPI == EPI_ROPE ? (hc[nt] & (DH - 1)) : hc[nt] % rdh;  // column inside its head (any even head width)
                cs[r][nt] = e[hd];
                sn[r][nt] = e[rdh + hd];
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + row_of(mt, r0 + r);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              float v = f.acc[mt][nt][r0 + r] + bq[nt];
              const float partner = __shfl_xor(v, 1, 64);  // the other element of the rotary pair (adjacent column)
              if (t < 2) v = (v * cs[r][nt]) + ((odd ? partner : -partner) * sn[r][nt]);  // rotate_half: (x0,x1) -> (-x1,x0), lightglue.py:151-158
              if (i < n && (EPI == EPI_ROPE || hc[nt] < rd)) Yt[(size_t)i * rd + hc[nt]] = v;
            }
          }
        }
      if (!more) break;
      cur = nxt;
      b = nb;
      n = nn;
      L = Ln;
      continue;
    }
    float* Y = g.Y + (size_t)b * g.cap * g.ldy;
    // epilogue: the bias of a lane's NT columns is loaded once; whole tiles take a guard-free path so
    // that the residual loads / stores of all 16*MT rows are issued back to back instead of one
    // load -> wait -> store round trip per element
    float bj[NT];
    int jj[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = j0 + col_of(nt);
      jj[nt] = j < g.N ? j : -1;
      bj[nt] = j < g.N ? g.bias[j] : 0.0f;
    }
    auto finish = [&](float acc, float bias, float old) {
      float v = acc + bias;
      if (EPI == EPI_DIV) v = v / g.div;
      if (EPI == EPI_RESID) v = old + v;
      return v;
    };
    if (i0 + BM <= n && j0 + BN <= g.N) {
      // uniform row pointer (scalar registers) + one 32-bit lane offset shared by every access
      float* ytile = Y + (size_t)i0 * g.ldy + j0;
      const unsigned lane_off = (unsigned)(row_base() * g.ldy + col_of(0));
      // rows go in batches of 8: eight residual loads in flight, then eight stores
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
          float old[8][NT];
          if (EPI == EPI_RESID) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const float* yrow = ytile + (size_t)row_step(mt, r0 + r) * g.ldy;
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) old[r][nt] = yrow[lane_off + nt * 32];
            }
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            float* yrow = ytile + (size_t)row_step(mt, r0 + r) * g.ldy;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              yrow[lane_off + nt * 32] = finish(f.acc[mt][nt][r0 + r], bj[nt], EPI == EPI_RESID ? old[r][nt] : 0.0f);
          }
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + row_of(mt, r);
          if (i >= n) continue;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (jj[nt] < 0) continue;
            float* y = Y + (size_t)i * g.ldy + jj[nt];
            *y = finish(f.acc[mt][nt][r], bj[nt], EPI == EPI_RESID ? *y : 0.0f);
          }
        }
    }
    if (!more) break;
    cur = nxt;
    b = nb;
    n = nn;
    L = Ln;
  }
}

// ------------------------------------------------------------------------------------------
// The same linears for SMALL grids (single pairs: fewer 128x128 tiles than CUs).  A launch that cannot fill the chip is
// bound by the latency of ONE tile's K loop (8 waves x 32 MFMAs + staging per 32-deep slab, 10-20 us per tile however few
// tiles there are): 64x64 tiles on 4 waves of one 32x32 accumulator each halve the matrix work per slab and give 4x the
// workgroups.  Same K order per output (one k-ordered chain from +0), so results are bit-identical to lg_gemm_kernel;
// same four epilogues.  K % 32 == 0 and N % 64 == 0 (every LightGlue shape); one tile per workgroup, no persistence.
// ------------------------------------------------------------------------------------------
constexpr int SBM = 64, SBN = 64, SBK = 32, SP = SBK + 1;
template <int EPI>
__global__ __launch_bounds__(256) void lg_gemm_small_kernel(const GemmArgs g) {
  __shared__ float As[SBM * SP];
  __shared__ float Bs[SBN * SP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int b = blockIdx.z;
  const int n = g.cnt ? min(g.cnt[b], g.cap) : g.cap;
  const int i0 = (int)blockIdx.y * SBM;
  if (i0 >= n) return;
  const int tj = (int)blockIdx.x;
  int j0, t = 0;
  const float* Wb;
  int ldb;
  constexpr bool ROPE = EPI == EPI_ROPE || EPI == EPI_ROPE_ANY;
  const int rd = EPI == EPI_ROPE ? D : g.d, rdh = EPI == EPI_ROPE ? DH : g.dh;
  if (ROPE) {  // d / 64 tiles of 64 columns per (q | k | v) block; W rows (hc + r) * 3 + t (see lg_gemm_kernel)
    const int tpb = rd / SBN;
    t = tj / tpb;
    j0 = (tj - t * tpb) * SBN;
    Wb = g.W + (size_t)t * g.K;
    ldb = 3 * g.K;
  } else {
    j0 = tj * SBN;
    Wb = g.W;
    ldb = g.K;
  }
  const float* A1 = g.X + (size_t)b * g.cap * g.ldx;
  const float* A2 = g.X2 ? g.X2 + (size_t)b * g.cap * g.ldx2 : nullptr;
  // staging: 64 rows x 8 float4 per operand and slab = 2 float4 per thread; rows past n read row n - 1 (never stored)
  int ra[2], c4a[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + i * 256;
    ra[i] = f >> 3;
    c4a[i] = f & 7;
  }
  f32x4 va[2], vb[2];
  auto issue = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = min(i0 + ra[i], n - 1);
      const int k = k0 + c4a[i] * 4;
      va[i] = (k < g.Ksplit) ? *reinterpret_cast<const f32x4*>(A1 + (size_t)row * g.ldx + k)
                             : *reinterpret_cast<const f32x4*>(A2 + (size_t)row * g.ldx2 + (k - g.Ksplit));
      vb[i] = *reinterpret_cast<const f32x4*>(Wb + (size_t)(j0 + ra[i]) * ldb + k);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  issue(0);
  const int aoff = (wm * 32 + l31) * SP + half, boff = (wn * 32 + l31) * SP + half;
  for (int k0 = 0; k0 < g.K; k0 += SBK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        As[ra[i] * SP + c4a[i] * 4 + e] = va[i][e];
        Bs[ra[i] * SP + c4a[i] * 4 + e] = vb[i][e];
      }
    __syncthreads();
    if (k0 + SBK < g.K) issue(k0 + SBK);
    float av[2], bv[2];
    av[0] = As[aoff];
    bv[0] = Bs[boff];
#pragma unroll
    for (int kk = 0; kk < SBK / 2; ++kk) {
      if (kk + 1 < SBK / 2) {
        av[(kk + 1) & 1] = As[aoff + (kk + 1) * 2];
        bv[(kk + 1) & 1] = Bs[boff + (kk + 1) * 2];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1], bv[kk & 1], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue: C row (r & 3) + 8 (r >> 2) + 4 half of the wave's 32x32 tile, column l31
  const int col = j0 + wn * 32 + l31;  // < N (N % 64 == 0); for EPI_ROPE the column inside the 256-wide q / k / v block
  int rows[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rows[r] = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
  if (ROPE) {
    float* Yt = (t == 0 ? g.Yq : t == 1 ? g.Yk : g.Yv) + (size_t)b * g.cap * rd;
    const float* encb = g.enc + (size_t)b * g.cap * (2 * rdh);
    const float bq = g.bias[col * 3 + t];
    const bool odd = lane & 1;
    float cs[16], sn[16];
    if (t < 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* e = encb + (size_t)min(rows[r], n - 1) * (2 * rdh);
        const int hd = EPI == EPI_ROPE ? (col & (DH - 1)) : col % rdh;
        cs[r] = e[hd];
        sn[r] = e[rdh + hd];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] + bq;
      const float partner = __shfl_xor(v, 1, 64);
      if (t < 2) v = (v * cs[r]) + ((odd ? partner : -partner) * sn[r]);
      if (rows[r] < n) Yt[(size_t)rows[r] * rd + col] = v;
    }
    return;
  }
  float* Y = g.Y + (size_t)b * g.cap * g.ldy;
  const float bj = g.bias[col];
  float old[16];
  if (EPI == EPI_RESID) {
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = Y[(size_t)min(rows[r], n - 1) * g.ldy + col];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = acc[r] + bj;
    if (EPI == EPI_DIV) v = v / g.div;
    if (EPI == EPI_RESID) v = old[r] + v;
    if (rows[r] < n) Y[(size_t)rows[r] * g.ldy + col] = v;
  }
}


// ------------------------------------------------------------------------------------------
// fused attention: out[b,q,h*HD:(h+1)*HD] = softmax_j(scale * Q_h[q] . K_h[j]) V_h[j]
// HD: head dim (32 / 64 / 128); DD: row width d of Q / K / V / O as a compile-time constant (256: the shipped instantiation) or 0 = a.d;
// LD: row stride of Q / K / V when they are column blocks of a wider buffer (the merged to_qk | to_v projection: 512), 0 = the width
// ------------------------------------------------------------------------------------------
struct AttnArgs {
  const float* Q;
  const float* K;
  const float* V;
  float* O;
  const int32_t* nq;
  const int32_t* nk;
  int capq, capk;
  float scale;
  int kv_shift, Btot;  // keys / values of batch entry b come from entry (b + kv_shift) % Btot (cross attention over the two sides stacked in one buffer)
  int d;               // row width (heads * head dim); read when the kernel's DD is 0
  int dh;              // head width when the kernel's DD is 0: <= HD, a multiple of 4; dims dh .. HD - 1 are zero padding
};

constexpr int AKB = 32;   // keys staged per round (32: 124 VGPRs -> three workgroups per CU; 64: 151 -> two)
template <int HD, int DD, int LD = 0>
__global__ __launch_bounds__(256) void lg_attn_kernel(const AttnArgs a) {
  constexpr int KPITCH = HD + 4;  // K rows padded by 4 floats (64-wide heads: 68): 16-byte aligned for ds_read_b128, and the 32 rows a
                                  // half-wave reads start on 16 distinct 4-bank groups (conflict-free)
  constexpr int DH = HD;          // (shadows the file-level constant)
  __shared__ __attribute__((aligned(16))) float Ks[AKB * KPITCH];
  __shared__ __attribute__((aligned(16))) float Vs[AKB * DH];
  const int D = DD ? DD : a.d;
  const int DL = LD ? LD : D;  // row stride of Q / K / V
  // the query blocks of one (pair, head) read the same K / V: keep them on one XCD (see xcd_contiguous)
  const int gx = (int)gridDim.x, gy = (int)gridDim.y;
  int item = xcd_contiguous((int)(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z)), gx * gy * (int)gridDim.z);
  const int qb = item % gx;
  item /= gx;
  const int h = item % gy, b = item / gy;
  const int bk = a.kv_shift ? (b + a.kv_shift) % a.Btot : b;
  const int nq = min(a.nq[b], a.capq), nk = min(a.nk[bk], a.capk);
  const int q0 = qb * 128;
  if (q0 >= nq || nk <= 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int q = q0 + wave * 32 + l31;
  const bool qv = q < nq;
  // MFMA K-step t pairs head dims (t, t + DH/2): lane half h supplies dim t + (DH/2) h, so a lane's K fragments
  // for four consecutive steps are one 16-byte LDS read
  // Head widths that are not 32 / 64 / 128 (lightglue.py:456-461 takes any descriptor_dim // num_heads) run the next larger
  // instantiation on zero-padded heads (DD == 0 only): dims dhr .. HD - 1 of Q, K and V read as 0 -- exact zeros add nothing
  // to a dot product -- and are not stored.  The head's columns start at h * dhr.
  const int dhr = DD ? DH : a.dh;
  const float* Qrow = a.Q + ((size_t)b * a.capq + (qv ? q : nq - 1)) * DL + h * dhr + (DH / 2) * half;
  float qreg[DH / 2];
#pragma unroll
  for (int t = 0; t < DH / 2; t += 4) {
    const f32x4 v = (DD || (DH / 2) * half + t < dhr) ? *reinterpret_cast<const f32x4*>(Qrow + t) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    qreg[t] = v[0];
    qreg[t + 1] = v[1];
    qreg[t + 2] = v[2];
    qreg[t + 3] = v[3];
  }
  const float* Kb = a.K + (size_t)bk * a.capk * DL + h * dhr;
  const float* Vb = a.V + (size_t)bk * a.capk * DL + h * dhr;
  constexpr int OT = DH / 32;  // 32-wide blocks of the output row
  f32x16 o[OT];
#pragma unroll
  for (int mt = 0; mt < OT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[mt][r] = 0.0f;
  const float NEG = -einx_u2f(0x7f800000u);
  const float sl2 = a.scale * 1.44269504088896341f;  // softmax in base 2: exp(x) = 2^(x log2 e)
  float m_run = NEG, l_run = 0.0f;

  // K/V block staging through registers: block kb0+64 is in flight while block kb0 is consumed
  constexpr int R4 = DH / 4;               // float4 per K / V row
  constexpr int ASTG = AKB * R4 / 256;     // float4 per thread per operand per round
  static_assert(ASTG >= 1 && DH % 32 == 0, "head dim 32, 64 or 128");
  f32x4 rk[ASTG], rv[ASTG];
  auto issue = [&](int kb0) {
#pragma unroll
    for (int i = 0; i < ASTG; ++i) {
      const int fidx = tid + i * 256;
      const int row = fidx / R4, c4 = fidx % R4;
      const int key = min(kb0 + row, nk - 1);  // clamped: rows past nk are masked after the QK product
      if (DD || c4 * 4 < dhr) {
        rk[i] = *reinterpret_cast<const f32x4*>(Kb + (size_t)key * DL + c4 * 4);
        rv[i] = *reinterpret_cast<const f32x4*>(Vb + (size_t)key * DL + c4 * 4);
      } else {
        rk[i] = rv[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < ASTG; ++i) {
      const int fidx = tid + i * 256;
      const int row = fidx / R4, c4 = fidx % R4;
      *reinterpret_cast<f32x4*>(Ks + row * KPITCH + c4 * 4) = rk[i];
      *reinterpret_cast<f32x4*>(Vs + row * DH + c4 * 4) = rv[i];
    }
  };
  issue(0);
  for (int kb0 = 0; kb0 < nk; kb0 += AKB) {
    __syncthreads();
    commit();
    __syncthreads();
    if (kb0 + AKB < nk) issue(kb0 + AKB);
#pragma unroll
    for (int sub = 0; sub < AKB / 32; ++sub) {
      const int kbase = kb0 + sub * 32;
      if (kbase >= nk) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
      const float* krow = Ks + (sub * 32 + l31) * KPITCH + (DH / 2) * half;
      f32x4 kf[2];
      kf[0] = *reinterpret_cast<const f32x4*>(krow);
#pragma unroll
      for (int u = 0; u < DH / 8; ++u) {
        if (u + 1 < DH / 8) kf[(u + 1) & 1] = *reinterpret_cast<const f32x4*>(krow + 4 * (u + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u & 1][t], qreg[4 * u + t], s, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      float mx = NEG;
      if (kbase + 32 <= nk) {  // whole block (uniform): no key mask
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = s[r] * sl2;
          mx = fmaxf(mx, s[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + crow(r, half);
          s[r] = key < nk ? s[r] * sl2 : NEG;
          mx = fmaxf(mx, s[r]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      // v_exp_f32 (2^x, ~1 ulp): LightGlue is compared at 1e-4, so the exact-order einx_expf is not
      // needed here; 2^(-inf) = 0 handles masked keys and the first block
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
        psum += s[r];
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
      // once the running maxima have settled alpha is exactly 1 in every lane: skip the 32 multiplications by one
      // (wave-uniform branch; bit-identical, x * 1.0f == x)
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int mt = 0; mt < OT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
      }
      const float* vbase = Vs + (size_t)(sub * 32 + 4 * half) * DH + l31;
      float vf[2][OT];
#pragma unroll
      for (int mt = 0; mt < OT; ++mt) vf[0][mt] = vbase[crow(0, 0) * DH + 32 * mt];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r + 1 < 16) {
#pragma unroll
          for (int mt = 0; mt < OT; ++mt) vf[(r + 1) & 1][mt] = vbase[crow(r + 1, 0) * DH + 32 * mt];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < OT; ++mt) o[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r & 1][mt], s[r], o[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  if (qv) {
    float* orow = a.O + ((size_t)b * a.capq + q) * D + h * dhr;
#pragma unroll
    for (int mt = 0; mt < OT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (DD || mt * 32 + crow(r, half) < dhr) orow[mt * 32 + crow(r, half)] = o[mt][r] / l;
  }
}

// ------------------------------------------------------------------------------------------
// The same attention for SMALL grids (single pairs: 64 workgroups of lg_attn_kernel on 256 CUs, each wave walking all keys of
// its 32 queries alone: 32 key blocks x ~2.4 us).  Here ONE workgroup of four waves owns 32 queries and the waves share the
// matrix work of every key block, on the 16x16x4 instruction (which accumulates its four K products as a sequential fma chain
// like 32x32x2 its two: tools/mfma16_order.hip), so that every output is the SAME chain of operations as in lg_attn_kernel:
//   A  S^T quadrant (16 keys x 16 queries) per wave: K-step g feeds dims (2g, 32+2g, 2g+1, 33+2g) = steps 2g, 2g+1 of the wide kernel
//   B  every wave redoes the block's online softmax for all 32 queries in the wide kernel's lane layout (query = lane % 32, the
//      16 keys crow(r, lane / 32)): same scale, mask, maxima, exponentials, per-half sums -- redundantly, no exchange of state
//   C  O^T tiles (16 dims x 16 queries), two per wave: K-step g feeds keys (crow(2g,0), crow(2g,1), crow(2g+1,0), crow(2g+1,1))
//      = steps 2g, 2g+1 of the wide kernel; the running rescale by alpha is a separate multiplication there and here.
// Bit-identical to lg_attn_kernel<64, 256> (tests/test_lightglue_gpu.py::test_lightglue_small_grid_kernels_equal_the_large_grid_kernels).
// ------------------------------------------------------------------------------------------
constexpr int A16_KP = 68;  // K rows: [slot k = 0..3][g = 0..15] + 4 floats of padding
constexpr int A16_VP = 68;  // V rows: 64 dims + 4
constexpr int A16_SP = 33;  // S / P rows: 32 queries + 1

// Eight waves, two teams: waves 4-7 compute the S^T quadrants of block i + 1 (a chain of 16 dependent matrix steps) while waves
// 0-3 run the softmax and the O^T tiles of block i -- two waves per SIMD whose stalls cover each other; K, V and S are
// double-buffered and a block costs one workgroup barrier.  (Measured on the way, per launch at one pair: all phases in four
// waves with three barriers per block 44 us; the same with phase A moved one block ahead in the SAME waves 47 us; 16 queries per
// workgroup, two workgroups per CU 44 us; lg_attn_kernel 76 us -- profiles/r05_notes.md 5.)
template <int LD>  // row stride of Q / K / V: 256, or 512 when they are column blocks of the merged to_qk | to_v projection
__global__ __launch_bounds__(512) void lg_attn16_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[2][32 * A16_KP];
  __shared__ __attribute__((aligned(16))) float Vs[2][32 * A16_VP];
  __shared__ float Ss[2][32 * A16_SP];
  __shared__ float Ps[4][32 * A16_SP];
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int bk = a.kv_shift ? (b + a.kv_shift) % a.Btot : b;
  const int nq = min(a.nq[b], a.capq), nk = min(a.nk[bk], a.capk);
  const int q0 = qb * 32;
  if (q0 >= nq || nk <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool team_a = wave >= 4;
  const int w4 = wave & 3;
  const int half = lane >> 5, l31 = lane & 31;  // phase B layout
  const int k4 = lane >> 4, c = lane & 15;      // MFMA 16x16x4 layout: K slot, row / column
  const int kq = w4 >> 1, qq = w4 & 1;          // phase A quadrant: keys 16 kq.., queries 16 qq..
  // Q fragment of phase A: query q0 + 16 qq + c, dims of slot k4 in step order: (k4 & 1) * 32 + 2 g + (k4 >> 1)
  float qf[16];
  {
    const int q = min(q0 + 16 * qq + c, nq - 1);
    const float* Qrow = a.Q + ((size_t)b * a.capq + q) * LD + h * DH + (k4 & 1) * 32 + (k4 >> 1);
#pragma unroll
    for (int g = 0; g < 16; ++g) qf[g] = team_a ? Qrow[2 * g] : 0.0f;
  }
  const float* Kb = a.K + (size_t)bk * a.capk * LD + h * DH;
  const float* Vb = a.V + (size_t)bk * a.capk * LD + h * DH;
  f32x4 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) o[t][i] = 0.0f;
  const float NEG = -einx_u2f(0x7f800000u);
  const float sl2 = a.scale * 1.44269504088896341f;
  float m_run = NEG, l_run = 0.0f;
  // staging: 32 keys x 16 float4 per operand = one float4 per thread and operand; rows past nk repeat row nk - 1 (masked in B)
  const int srow = tid >> 4, sc4 = tid & 15;
  auto load_row = [&](const float* base, int kb0) {
    return *reinterpret_cast<const f32x4*>(base + (size_t)min(kb0 + srow, nk - 1) * LD + sc4 * 4);
  };
  auto commit_k = [&](float* dst, const f32x4& r) {  // dims d0 .. d0+3 (d0 = 4 sc4): slot (d >= 32) + 2 (d & 1), position (d & 31) >> 1
    const int hi = sc4 >> 3, g0 = (sc4 & 7) * 2;
    float* kr = dst + srow * A16_KP;
    kr[hi * 16 + g0] = r[0];
    kr[hi * 16 + g0 + 1] = r[2];
    kr[(hi + 2) * 16 + g0] = r[1];
    kr[(hi + 2) * 16 + g0 + 1] = r[3];
  };
  auto commit_v = [&](float* dst, const f32x4& r) { *reinterpret_cast<f32x4*>(dst + srow * A16_VP + sc4 * 4) = r; };
  // phase A of one block: this wave's S^T quadrant from ks into ss
  auto phase_a = [&](const float* ks, float* ss) {
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* kr = ks + (16 * kq + c) * A16_KP + k4 * 16;
    f32x4 kf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) kf[u] = *reinterpret_cast<const f32x4*>(kr + 4 * u);
#pragma unroll
    for (int g = 0; g < 16; ++g) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[g >> 2][g & 3], qf[g], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) ss[(16 * kq + 4 * k4 + i) * A16_SP + 16 * qq + c] = acc[i];
  };
  const int nblk = (nk + 31) >> 5;
  // K / V rows travel global -> registers -> LDS with TWO blocks of flight time (a block is ~1 us, an L2 round trip under this
  // access pattern about as long: with one block of distance the loop waited ~9 us per launch for its loads)
  f32x4 rk, rv, rk_new, rv_new;
  {  // prologue: K(0), V(0), K(1) staged, K(2) and V(1) requested; S(0) computed
    const f32x4 r0 = load_row(Kb, 0);
    rv = load_row(Vb, 0);
    rk = load_row(Kb, 32);
    commit_k(Ks[0], r0);
    commit_v(Vs[0], rv);
    commit_k(Ks[1], rk);
    rk = load_row(Kb, 64);
    rv = load_row(Vb, 32);
    __syncthreads();
    if (team_a) phase_a(Ks[0], Ss[0]);
    __syncthreads();
  }
  float* Pw = Ps[w4];
  for (int blk = 0; blk < nblk; ++blk) {
    const int cur = blk & 1, nxt = cur ^ 1;
    const int kbase = blk * 32;
    // requested one block ago, committed at the end of this one: K(blk + 2) -> Ks[cur], V(blk + 1) -> Vs[nxt]; requested now for
    // the next iteration's commit: K(blk + 3), V(blk + 2)
    const bool more1 = blk + 1 < nblk, more2 = blk + 2 < nblk;
    rk_new = load_row(Kb, kbase + 96);
    rv_new = load_row(Vb, kbase + 64);
    if (team_a) {
      // ---- A of the NEXT block
      if (more1) phase_a(Ks[nxt], Ss[nxt]);
    } else {
      // ---- B: the block's online softmax, as in lg_attn_kernel (lane = query l31, keys crow(r, half)); every wave of the team
      // redoes it for all 32 queries (no exchange of state)
      const float* ss = Ss[cur];
      float s[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = ss[crow(r, half) * A16_SP + l31];
      // phase C's V operands do not depend on the softmax: requested with the S values (one LDS round trip for both)
      const float* vs = Vs[cur];
      float vfr[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) vfr[g] = vs[crow(2 * g + (k4 >> 1), k4 & 1) * A16_VP + 16 * w4 + c];
      __builtin_amdgcn_sched_barrier(0);
      float mx = NEG;
      if (kbase + 32 <= nk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = s[r] * sl2;
          mx = fmaxf(mx, s[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + crow(r, half);
          s[r] = key < nk ? s[r] * sl2 : NEG;
          mx = fmaxf(mx, s[r]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
        psum += s[r];
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) Pw[crow(r, half) * A16_SP + l31] = s[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave's own LDS writes -> its own reads (in-order LDS)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- C: O^T tiles (dims 16 w4 .., queries 16 t ..)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float al = __shfl(alpha, 16 * t + c, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[t][i] *= al;
      }
      float pr[8][2];  // all P operands in one batch (the chain of matrix steps then runs without LDS waits)
#pragma unroll
      for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t) pr[g][t] = Pw[crow(2 * g + (k4 >> 1), k4 & 1) * A16_SP + 16 * t + c];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(vfr[g], pr[g][t], o[t], 0, 0, 0);
    }
    // Ks[cur] was last read by phase A of THIS block (previous iteration, before its barrier); Vs[nxt] by phase C of the previous block
    if (more2) commit_k(Ks[cur], rk);
    if (more1) commit_v(Vs[nxt], rv);
    rk = rk_new;
    rv = rv_new;
    __syncthreads();
  }
  if (team_a) return;
  const float l = l_run + __shfl_xor(l_run, 32, 64);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float lq = __shfl(l, 16 * t + c, 64);
    const int q = q0 + 16 * t + c;
    if (q < nq) {
      float* orow = a.O + ((size_t)b * a.capq + q) * D + h * DH + 16 * w4 + 4 * k4;
      *reinterpret_cast<f32x4*>(orow) = f32x4{o[t][0] / lq, o[t][1] / lq, o[t][2] / lq, o[t][3] / lq};
    }
  }
}

// LayerNorm(512, eps 1e-5, biased variance) + exact GELU, in place; one wave per token
__global__ __launch_bounds__(256) void lg_ln_gelu_kernel(float* hbuf, const int32_t* cnt, int cap, const float* g, const float* be) {
  const int b = blockIdx.y;
  const int n = min(cnt[b], cap);
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  float* row = hbuf + ((size_t)b * cap + i) * 512;
  float v[8], gv[8], bv[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = row[lane + 64 * t];
  // gamma / beta are requested with the row: read next to the stores below they are one L2 round trip per element
  // (the compiler cannot move a load above a store through plain float pointers)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    gv[t] = g[lane + 64 * t];
    bv[t] = be[lane + 64 * t];
  }
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < 8; ++t) s += v[t];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / 512.0f;
  float q = 0.0f;
#pragma unroll
  for (int t = 0; t < 8; ++t) q = fmaf(v[t] - mean, v[t] - mean, q);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
  const float rstd = 1.0f / sqrtf(q / 512.0f + 1e-5f);
#pragma unroll
  for (int t = 0; t < 8; ++t) row[lane + 64 * t] = einx_geluf(fmaf((v[t] - mean) * rstd, gv[t], bv[t]));
}

// the same for any row width that is a multiple of 64 (configurations other than d = 256): lane partial sums over
// columns lane, lane + 64, ... ascending, then the same butterfly
__global__ __launch_bounds__(256) void lg_ln_gelu_any_kernel(float* hbuf, const int32_t* cnt, int cap, int width, const float* g, const float* be) {
  const int b = blockIdx.y;
  const int n = min(cnt[b], cap);
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  float* row = hbuf + ((size_t)b * cap + i) * width;
  float s = 0.0f;
  for (int c = lane; c < width; c += 64) s += row[c];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)width;
  float q = 0.0f;
  for (int c = lane; c < width; c += 64) q = fmaf(row[c] - mean, row[c] - mean, q);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
  const float rstd = 1.0f / sqrtf(q / (float)width + 1e-5f);
  for (int c = lane; c < width; c += 64) row[c] = einx_geluf(fmaf((row[c] - mean) * rstd, g[c], be[c]));
}

// matchability logit z = x . w + b per token, plus logsigmoid(z) and logsigmoid(-z)
__global__ __launch_bounds__(256) void lg_matchability_kernel(const float* x, const int32_t* cnt, int cap, int d, const float* w, const float* bm,
                                                              float* cert, float* dust) {
  const int b = blockIdx.y;
  const int n = min(cnt[b], cap);
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const float* row = x + ((size_t)b * cap + i) * d;
  float s = 0.0f;
  for (int c = lane; c < d; c += 64) s = fmaf(row[c], w[c], s);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) {
    const float z = s + bm[0];
    cert[(size_t)b * cap + i] = einx_logsigmoidf(z);
    dust[(size_t)b * cap + i] = einx_logsigmoidf(-z);
  }
}

// filter_matches (lightglue.py:402-418) from the packed arg-max keys of the assignment scores
__global__ void lg_finalize_kernel(const unsigned long long* rowkey, const unsigned long long* colkey, const int32_t* nn, const int32_t* mm,
                                   int cap0, int cap1, float th, int64_t* m0, int64_t* m1, float* s0, float* s1) {
  const int b = blockIdx.y;
  const int n = min(nn[b], cap0), m = min(mm[b], cap1);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long* rk = rowkey + (size_t)b * cap0;
  const unsigned long long* ck = colkey + (size_t)b * cap1;
  auto idx_of = [](unsigned long long k) { return (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)); };
  auto val_of = [](unsigned long long k) { return einx_ordered_unkey((unsigned)(k >> 32)); };
  if (t < cap0) {
    int64_t r = -1;
    float sc = 0.0f;
    if (t < n && m > 0) {
      const int j = idx_of(rk[t]);
      const bool mutual = idx_of(ck[j]) == t;
      sc = mutual ? einx_expf(val_of(rk[t])) : 0.0f;
      if (mutual && sc > th) r = j;
    }
    m0[(size_t)b * cap0 + t] = r;
    s0[(size_t)b * cap0 + t] = sc;
  }
  if (t < cap1) {
    int64_t r = -1;
    float sc = 0.0f;
    if (t < m && n > 0) {
      const int i = idx_of(ck[t]);
      const int back = idx_of(rk[i]);
      const bool mutual1 = back == t;
      // mscores1 = mscores0[m1] where mutual1; mscores0[i] is non-zero only if i is mutual too,
      // which is the same condition (back == t  <=>  i's best is t and t's best is i)
      const float ms0 = mutual1 ? einx_expf(val_of(rk[i])) : 0.0f;
      sc = ms0;
      if (mutual1 && ms0 > th) r = i;
    }
    m1[(size_t)b * cap1 + t] = r;
    s1[(size_t)b * cap1 + t] = sc;
  }
}

// dst_bstride: elements between consecutive batches of dst (cap*width for a plain [B,cap,width] copy)
__global__ void lg_copy_rows_kernel(const float* src, float* dst, const int32_t* cnt, int cap, int width, size_t dst_bstride) {
  const int b = blockIdx.y;
  const int n = min(cnt[b], cap);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * width) return;
  dst[(size_t)b * dst_bstride + t] = src[(size_t)b * cap * width + t];
}

// normalize_keypoints (lightglue.py:137-148): (kpt - size/2) / (max(size)/2), first two columns
__global__ void lg_normalize_kpts_kernel(const float* kpts, int cols, size_t total, float s0, float s1, float* out, int out_cols) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const float sc = fmaxf(s0, s1) / 2.0f;
  out[t * out_cols + 0] = (kpts[t * cols + 0] - s0 / 2.0f) / sc;
  out[t * out_cols + 1] = (kpts[t * cols + 1] - s1 / 2.0f) / sc;
  for (int c = 2; c < out_cols; ++c) out[t * out_cols + c] = 0.0f;
}

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Side {
  const float* kpts;
  const float* desc;
  const int32_t* cnt;
  int cap;
  float *x, *enc, *q, *k, *v, *ctx, *msg, *h, *cert, *dust;
};

// d: descriptor_dim, dh: head dim -- buffers [tokens, d] (x, q, k, v, ctx, msg), [tokens, 2 dh] (enc), [tokens, 2 d] (h)
size_t side_bytes(int B, int cap, int d, int dh) {
  const size_t tok = (size_t)B * cap;
  return al(tok * d * 4) + al(tok * 2 * dh * 4) + 3 * al(tok * d * 4) + 2 * al(tok * d * 4) + al(tok * 2 * d * 4) + 2 * al(tok * 4);
}

char* carve_side(Side& s, char* p, int B, int cap, int d, int dh) {
  const size_t tok = (size_t)B * cap;
  auto take = [&](size_t bytes) { float* r = (float*)p; p += al(bytes); return r; };
  s.x = take(tok * d * 4);
  s.enc = take(tok * 2 * dh * 4);
  s.q = take(tok * d * 4);
  s.k = take(tok * d * 4);
  s.v = take(tok * d * 4);
  s.ctx = take(tok * d * 4);
  s.msg = take(tok * d * 4);
  s.h = take(tok * 2 * d * 4);
  s.cert = take(tok * 4);
  s.dust = take(tok * 4);
  return p;
}

// Both sides stacked: every buffer is [2B, cap, width], side 0 = entries 0..B-1, side 1 = entries B..2B-1.  LightGlue applies
// the same weights to both sides, so every per-side launch becomes one launch over 2B entries (cross attention reads
// the partner entry's keys / values); s0 / s1 stay views of the halves.  Never larger than two separate sides.
char* carve_stacked(Side& s0, Side& s1, char* p, int B, int cap, int d, int dh) {
  const size_t tok = (size_t)B * cap;
  auto take = [&](size_t width, float*& a0, float*& a1) {
    a0 = (float*)p;
    a1 = a0 + tok * width;
    p += al(2 * tok * width * 4);
  };
  take(d, s0.x, s1.x);
  take(2 * dh, s0.enc, s1.enc);
  take(d, s0.q, s1.q);
  take(d, s0.k, s1.k);
  take(d, s0.v, s1.v);
  take(d, s0.ctx, s1.ctx);
  take(d, s0.msg, s1.msg);
  take(2 * d, s0.h, s1.h);
  take(1, s0.cert, s1.cert);
  take(1, s0.dust, s1.dust);
  return p;
}

__global__ void lg_stack_counts_kernel(const int32_t* n, const int32_t* m, int B, int32_t* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 2 * B) out[t] = t < B ? n[t] : m[t - B];
}

// persistent launch size: resident workgroups of the tile kernels on this device (a multiple of
// 8 so that every XCD gets the same number of slots), never more than the tiles there are
unsigned gemm_grid(int tiles) {
  static int resident = 0;
  if (resident == 0) {
    int dev = 0, cus = 256, per_cu = 2;
    if (hipGetDevice(&dev) == hipSuccess) {
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lg_gemm_kernel<EPI_BIAS>, THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    }
    resident = (cus * per_cu) & ~7;
    if (resident < 8) resident = 8;
  }
  const int want = (tiles + 7) & ~7;
  return (unsigned)(want < resident ? want : resident);
}

// fewer 128x128 tiles than CUs (and shapes the small kernel takes): 64x64 tiles, one per workgroup
bool small_grid(int N, int cap, int B, int K, int Ksplit) {
  const long max_tiles = 256;
  const long tiles = (long)einx_cdiv(N, BN) * einx_cdiv(cap, BM) * B;
  return tiles < max_tiles && K % SBK == 0 && N % SBN == 0 && (Ksplit >= K || Ksplit % SBK == 0);
}

int gemm(hipStream_t st, int epi, const Side& s, int B, const float* X, int ldx, const float* X2, int ldx2, int Ksplit, int K, const float* W,
         const float* bias, int N, float* Y, int ldy, float div = 1.0f) {
  GemmArgs g;
  g.X = X;
  g.X2 = X2;
  g.W = W;
  g.bias = bias;
  g.Y = Y;
  g.cnt = s.cnt;
  g.cap = s.cap;
  g.ldx = ldx;
  g.ldx2 = ldx2;
  g.Ksplit = Ksplit;
  g.K = K;
  g.N = N;
  g.ldy = ldy;
  g.div = div;
  g.B = B;
  if (small_grid(N, s.cap, B, K, Ksplit)) {
    const dim3 sg((unsigned)(N / SBN), (unsigned)einx_cdiv(s.cap, SBM), (unsigned)B);
    EINX_PROF("lg_gemm_small_kernel", st);
    if (epi == EPI_BIAS) hipLaunchKernelGGL(lg_gemm_small_kernel<EPI_BIAS>, sg, dim3(256), 0, st, g);
    else if (epi == EPI_DIV) hipLaunchKernelGGL(lg_gemm_small_kernel<EPI_DIV>, sg, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(lg_gemm_small_kernel<EPI_RESID>, sg, dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  const dim3 grid(gemm_grid(einx_cdiv(N, BN) * einx_cdiv(s.cap, BM) * B));
  EINX_PROF("lg_gemm_kernel", st);
  if (epi == EPI_BIAS) hipLaunchKernelGGL(lg_gemm_kernel<EPI_BIAS>, grid, dim3(THREADS), 0, st, g);
  else if (epi == EPI_DIV) hipLaunchKernelGGL(lg_gemm_kernel<EPI_DIV>, grid, dim3(THREADS), 0, st, g);
  else hipLaunchKernelGGL(lg_gemm_kernel<EPI_RESID>, grid, dim3(THREADS), 0, st, g);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the model's widths, as the launchers need them
struct Dims {
  int d, heads, dh;
  bool shipped() const { return d == D && dh == DH; }
};

// fused Wqkv projection + rotary split: X [B,cap,d] -> q, k (rotary applied), v, each [B,cap,d]
int gemm_qkv_rope(hipStream_t st, const Side& s, int B, const Dims& dm, const float* X, const float* Wqkv, const float* bqkv, float* q, float* k,
                  float* v) {
  const int D = dm.d;  // (shadows the file-level constant)
  GemmArgs g{};
  g.X = X;
  g.X2 = nullptr;
  g.W = Wqkv;
  g.bias = bqkv;
  g.Y = nullptr;
  g.cnt = s.cnt;
  g.cap = s.cap;
  g.ldx = D;
  g.ldx2 = 0;
  g.Ksplit = 0x7fffffff;
  g.K = D;
  g.N = 3 * D;
  g.ldy = D;
  g.div = 1.0f;
  g.B = B;
  g.enc = s.enc;
  g.Yq = q;
  g.Yk = k;
  g.Yv = v;
  g.d = dm.d;
  g.dh = dm.dh;
  const int tiles = 3 * einx_cdiv(D, BN) * einx_cdiv(s.cap, BM) * B;  // three blocks (q | k | v) of whole tiles each
  if (tiles < 256 && D % SBN == 0 && D % SBK == 0) {  // as small_grid()
    const dim3 sg((unsigned)(3 * D / SBN), (unsigned)einx_cdiv(s.cap, SBM), (unsigned)B);
    EINX_PROF("lg_gemm_small_kernel", st);
    if (dm.shipped()) hipLaunchKernelGGL(lg_gemm_small_kernel<EPI_ROPE>, sg, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(lg_gemm_small_kernel<EPI_ROPE_ANY>, sg, dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  const dim3 grid(gemm_grid(tiles));
  EINX_PROF("lg_gemm_kernel", st);
  if (dm.shipped()) hipLaunchKernelGGL(lg_gemm_kernel<EPI_ROPE>, grid, dim3(THREADS), 0, st, g);
  else hipLaunchKernelGGL(lg_gemm_kernel<EPI_ROPE_ANY>, grid, dim3(THREADS), 0, st, g);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// merged: Q, K, V are column blocks of a [.., 2 d] buffer (row stride 512; shipped widths only)
int attn(hipStream_t st, int B, const Dims& dm, const float* Q, const int32_t* nq, int capq, const float* K, const float* V, const int32_t* nk,
         int capk, float* O, int kv_shift = 0, bool merged = false) {
  AttnArgs a;
  a.kv_shift = kv_shift;
  a.Btot = B;
  a.Q = Q;
  a.K = K;
  a.V = V;
  a.O = O;
  a.nq = nq;
  a.nk = nk;
  a.capq = capq;
  a.capk = capk;
  a.d = dm.d;
  a.dh = dm.dh;
  a.scale = 1.0f / sqrtf((float)dm.dh);  // SDPA scale (self) = (dh^-1/4)^2 (cross, lightglue.py:316); 64-wide heads: exactly 0.125
  const dim3 grid((unsigned)einx_cdiv(capq, 128), (unsigned)dm.heads, (unsigned)B);
  EINX_PROF("lg_attn_kernel", st);
  // the latency form (same bits) while ITS grid fits the chip twice: fewer wide workgroups than CUs alone is not enough -- three
  // pairs of 1024 keypoints (768 latency-form workgroups of eight waves, 1.5 rounds) ran 3.67 instead of 3.26 ms; one pair
  // 1.82 instead of 2.39 ms, two pairs 2.51 instead of 2.68 (tools/lg_bench.py --batch 1 / 2 / 3 --skip-linear)
  if (dm.shipped() && (long)grid.x * grid.y * grid.z < 256 && (long)einx_cdiv(capq, 32) * dm.heads * B <= 512) {
    const dim3 g16((unsigned)einx_cdiv(capq, 32), (unsigned)dm.heads, (unsigned)B);
    if (merged) hipLaunchKernelGGL(lg_attn16_kernel<2 * D>, g16, dim3(512), 0, st, a);
    else hipLaunchKernelGGL(lg_attn16_kernel<D>, g16, dim3(512), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (merged && !dm.shipped()) return -1;
  if (merged) hipLaunchKernelGGL((lg_attn_kernel<64, D, 2 * D>), grid, dim3(256), 0, st, a);
  else if (dm.shipped()) hipLaunchKernelGGL((lg_attn_kernel<64, D>), grid, dim3(256), 0, st, a);
  else if (dm.dh <= 32) hipLaunchKernelGGL((lg_attn_kernel<32, 0>), grid, dim3(256), 0, st, a);  // (narrower heads: zero padded)
  else if (dm.dh <= 64) hipLaunchKernelGGL((lg_attn_kernel<64, 0>), grid, dim3(256), 0, st, a);
  else if (dm.dh <= 128) hipLaunchKernelGGL((lg_attn_kernel<128, 0>), grid, dim3(256), 0, st, a);
  else if (dm.dh <= 256) hipLaunchKernelGGL((lg_attn_kernel<256, 0>), grid, dim3(256), 0, st, a);
  else return -1;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ffn(cat[x,msg]) + residual, in place on s.x
int ffn(hipStream_t st, const Side& s, int B, const Dims& dm, const float* msg, const float* w0, const float* b0, const float* g, const float* be,
        const float* w3, const float* b3) {
  const int D = dm.d;  // (shadows the file-level constant)
  // (one fused launch for ffn.0 + LayerNorm + GELU was measured in round 3 and is slower: tools/experiments/lg_ffn0_ln_gelu_kernel.hip.txt)
  if (gemm(st, EPI_BIAS, s, B, s.x, D, msg, D, D, 2 * D, w0, b0, 2 * D, s.h, 2 * D)) return -1;
  {
    EINX_PROF("lg_ln_gelu_kernel", st);
    const dim3 lg((unsigned)einx_cdiv(s.cap, 4), (unsigned)B);
    if (2 * D == 512) hipLaunchKernelGGL(lg_ln_gelu_kernel, lg, dim3(256), 0, st, s.h, s.cnt, s.cap, g, be);
    else hipLaunchKernelGGL(lg_ln_gelu_any_kernel, lg, dim3(256), 0, st, s.h, s.cnt, s.cap, 2 * D, g, be);
  }
  if (hipGetLastError() != hipSuccess) return -1;
  return gemm(st, EPI_RESID, s, B, s.h, 2 * D, nullptr, 0, 0x7fffffff, 2 * D, w3, b3, D, s.x, D);
}

}  // namespace

EINX_EXPORT int einx_linear(const float* x, int M, int K, const float* w, const float* bias, int N, float* y, int accumulate, void* stream) {
  EINX_CHECK_ARG(x && w && bias && y, "null pointer");
  EINX_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 4 == 0, "bad shape (K must be a multiple of 4)");
  GemmArgs g;
  g.X = x;
  g.X2 = nullptr;
  g.W = w;
  g.bias = bias;
  g.Y = y;
  g.cnt = nullptr;
  g.cap = M;
  g.ldx = K;
  g.ldx2 = 0;
  g.Ksplit = 0x7fffffff;
  g.K = K;
  g.N = N;
  g.ldy = N;
  g.div = 1.0f;
  g.B = 1;
  const dim3 grid(gemm_grid(einx_cdiv(N, BN) * einx_cdiv(M, BM)));
  if (accumulate) hipLaunchKernelGGL(lg_gemm_kernel<EPI_RESID>, grid, dim3(THREADS), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(lg_gemm_kernel<EPI_BIAS>, grid, dim3(THREADS), 0, (hipStream_t)stream, g);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_normalize_keypoints(const float* kpts, int rows, int cols, float h, float w, float* out, int out_cols,
                                         void* stream) {
  EINX_CHECK_ARG(kpts && out, "null pointer");
  EINX_CHECK_ARG(rows > 0 && cols >= 2 && out_cols >= 2, "bad shape");
  hipLaunchKernelGGL(lg_normalize_kpts_kernel, dim3((unsigned)einx_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, kpts, cols,
                     (size_t)rows, h, w, out, out_cols);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

namespace {
// lightglue.py:456-461: head_dim = descriptor_dim // num_heads.  The attention kernel is instantiated for 32-, 64- and 128-wide
// heads; other widths run the next larger one on zero-padded heads (round 6).  Multiples of 4 (16-byte rows of a head; the
// rotary encoding needs an even width anyway) up to 256 (<256, 0>: 256 + 100 registers, one workgroup per SIMD -- rare widths, not tuned).
bool dims_of(int d, int heads, Dims& dm) {
  if (d <= 0 || heads <= 0 || d % heads != 0) return false;
  dm.d = d;
  dm.heads = heads;
  dm.dh = d / heads;
  return dm.dh % 4 == 0 && dm.dh <= 256;
}
}  // namespace

EINX_EXPORT size_t einx_lg_ws_bytes_heads(int B, int cap0, int cap1, int d, int heads, int input_dim) {
  Dims dm;
  if (B <= 0 || cap0 <= 0 || cap1 <= 0 || !dims_of(d, heads, dm)) return 0;
  (void)input_dim;
  return side_bytes(B, cap0, d, dm.dh) + side_bytes(B, cap1, d, dm.dh) + einx_mnn_ws_bytes(B, cap0, cap1) + al((size_t)2 * B * 4) + 1024;
}

EINX_EXPORT size_t einx_lg_ws_bytes(int B, int cap0, int cap1, int d, int input_dim) {  // 64-wide heads (the LightGlue default)
  return d > 0 && d % 64 == 0 ? einx_lg_ws_bytes_heads(B, cap0, cap1, d, d / 64, input_dim) : 0;
}

EINX_EXPORT int einx_lightglue(const einx_lg_weights* w, const float* kpts0, const float* desc0, const int32_t* n, int cap0,
                               const float* kpts1, const float* desc1, const int32_t* m, int cap1, int B, float h0, float w0, float h1,
                               float w1, void* ws, int64_t* matches0, int64_t* matches1, float* scores0, float* scores1, float* la,
                               float* ref0, float* ref1, int ref_layers, void* stream) {
  EINX_CHECK_ARG(w && kpts0 && desc0 && n && kpts1 && desc1 && m && ws && matches0 && matches1 && scores0 && scores1, "null pointer");
  Dims dm;
  EINX_CHECK_ARG(dims_of(w->d, w->heads, dm), "descriptor_dim must be num_heads x head_dim with head_dim a multiple of 4, at most 256");
  const int D = dm.d;  // (shadows the file-level constant: every width below is the model's)
  EINX_CHECK_ARG(w->struct_size == sizeof(einx_lg_weights) && w->layer_size == sizeof(einx_lg_layer),
                 "einx_lg_weights::struct_size / layer_size do not match this library (header / library ABI mismatch)");
  EINX_CHECK_ARG(w->n_layers >= 1 && w->layers, "no layers");
  EINX_CHECK_ARG(B > 0 && cap0 > 0 && cap1 > 0, "bad shape");
  EINX_CHECK_ARG(w->input_dim % 4 == 0 && w->input_dim > 0, "input_dim must be a multiple of 4");
  EINX_CHECK_ARG((w->input_dim == D) == (w->in_w == nullptr), "input_proj must be given exactly when input_dim != d");
  EINX_CHECK_ARG(ref_layers == 0 || ref_layers == 1 || ref_layers == w->n_layers, "ref_layers must be 0/1 (last layer) or n_layers");
  const bool all_layers = ref_layers > 1;
  hipStream_t st = (hipStream_t)stream;
  Side s0{}, s1{};
  s0.kpts = kpts0;
  s0.desc = desc0;
  s0.cnt = n;
  s0.cap = cap0;
  s1.kpts = kpts1;
  s1.desc = desc1;
  s1.cnt = m;
  s1.cap = cap1;
  char* p = (char*)ws;
  // equal capacities (every shipped configuration): the two sides are stacked and every layer runs ONCE over 2B entries --
  // half the launches, and at small batch twice the workgroups per launch (a single pair: 8.0 -> see profiles/r03_notes.md)
  const bool stacked = cap0 == cap1;
  Side sb{};
  if (stacked) {
    p = carve_stacked(s0, s1, p, B, cap0, D, dm.dh);
    int32_t* cnt2 = (int32_t*)p;
    p += al((size_t)2 * B * 4);
    hipLaunchKernelGGL(lg_stack_counts_kernel, dim3((unsigned)einx_cdiv(2 * B, 256)), dim3(256), 0, st, n, m, B, cnt2);
    sb = s0;
    sb.cnt = cnt2;
  } else {
    p = carve_side(s0, p, B, cap0, D, dm.dh);
    p = carve_side(s1, p, B, cap1, D, dm.dh);
  }
  void* mnn_ws = p;
  Side* sides[2] = {&s0, &s1};
  Side* run[2] = {stacked ? &sb : &s0, &s1};  // what the shared-weight stages iterate over
  const int nrun = stacked ? 1 : 2, Br = stacked ? 2 * B : B;
  const float sz[2][2] = {{h0, w0}, {h1, w1}};
#define LG_CHECK(expr)                                                    \
  do {                                                                    \
    if ((expr) != 0 || hipGetLastError() != hipSuccess) {                 \
      einx_set_error("einx_lightglue: kernel launch failed at %s", #expr); \
      return EINX_ERR_LAUNCH;                                             \
    }                                                                     \
  } while (0)
  LG_CHECK(0);
  // ---- input projection (or copy) + positional encodings (per side: own inputs, own image size) --------
  for (int sd = 0; sd < 2; ++sd) {
    Side& s = *sides[sd];
    if (w->in_w) {
      LG_CHECK(gemm(st, EPI_BIAS, s, B, s.desc, w->input_dim, nullptr, 0, 0x7fffffff, w->input_dim, w->in_w, w->in_b, D, s.x, D));
    } else {
      const size_t per = (size_t)s.cap * D;
      hipLaunchKernelGGL(lg_copy_rows_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, st, s.desc, s.x, s.cnt, s.cap, D, per);
      LG_CHECK(0);
    }
    hipLaunchKernelGGL(lg_posenc_kernel, dim3((unsigned)einx_cdiv(s.cap * (dm.dh / 2), 256), (unsigned)B), dim3(256), 0, st, s.kpts, s.cnt, s.cap,
                       sz[sd][0], sz[sd][1], w->Wr, s.enc, dm.dh);
    LG_CHECK(0);
  }
  // ---- transformer layers --------------------------------------------------------------------
  for (int li = 0; li < w->n_layers; ++li) {
    const einx_lg_layer& L = w->layers[li];
    for (int sd = 0; sd < nrun; ++sd) {
      Side& s = *run[sd];
      LG_CHECK(gemm_qkv_rope(st, s, Br, dm, s.x, L.Wqkv, L.bqkv, s.q, s.k, s.v));
      LG_CHECK(attn(st, Br, dm, s.q, s.cnt, s.cap, s.k, s.v, s.cnt, s.cap, s.ctx));
      if (L.Wo) {
        LG_CHECK(gemm(st, EPI_BIAS, s, Br, s.ctx, D, nullptr, 0, 0x7fffffff, D, L.Wo, L.bo, D, s.msg, D));
        LG_CHECK(ffn(st, s, Br, dm, s.msg, L.sf0_w, L.sf0_b, L.sln_g, L.sln_b, L.sf3_w, L.sf3_b));
      } else {  // out_proj folded into the FFN's first Linear at load time: message = context
        LG_CHECK(ffn(st, s, Br, dm, s.ctx, L.sf0_w, L.sf0_b, L.sln_g, L.sln_b, L.sf3_w, L.sf3_b));
      }
    }
    // to_qk and to_v read the same rows: with the merged weight image [Wqk; Wv] (shipped widths) they are ONE launch writing
    // qk | v side by side into the FFN's hidden buffer (free until ffn.0 runs), and the attention reads them at row stride 2 d --
    // the same k-ordered chain and bias per output, one launch less per layer
    const bool merged = L.Wqk_v && L.bqk_v && dm.shipped();
    for (int sd = 0; sd < nrun; ++sd) {
      Side& s = *run[sd];
      if (merged) {
        LG_CHECK(gemm(st, EPI_BIAS, s, Br, s.x, D, nullptr, 0, 0x7fffffff, D, L.Wqk_v, L.bqk_v, 2 * D, s.h, 2 * D));
      } else {
        LG_CHECK(gemm(st, EPI_BIAS, s, Br, s.x, D, nullptr, 0, 0x7fffffff, D, L.Wqk, L.bqk, D, s.q, D));
        LG_CHECK(gemm(st, EPI_BIAS, s, Br, s.x, D, nullptr, 0, 0x7fffffff, D, L.Wv, L.bv, D, s.v, D));
      }
    }
    if (stacked) {  // entry b attends to the keys / values of its partner entry (b + B) mod 2B
      if (merged) LG_CHECK(attn(st, Br, dm, sb.h, sb.cnt, sb.cap, sb.h, sb.h + D, sb.cnt, sb.cap, sb.ctx, B, true));
      else LG_CHECK(attn(st, Br, dm, sb.q, sb.cnt, sb.cap, sb.q, sb.v, sb.cnt, sb.cap, sb.ctx, B));
    } else if (merged) {
      LG_CHECK(attn(st, B, dm, s0.h, s0.cnt, s0.cap, s1.h, s1.h + D, s1.cnt, s1.cap, s0.ctx, 0, true));
      LG_CHECK(attn(st, B, dm, s1.h, s1.cnt, s1.cap, s0.h, s0.h + D, s0.cnt, s0.cap, s1.ctx, 0, true));
    } else {
      LG_CHECK(attn(st, B, dm, s0.q, s0.cnt, s0.cap, s1.q, s1.v, s1.cnt, s1.cap, s0.ctx));
      LG_CHECK(attn(st, B, dm, s1.q, s1.cnt, s1.cap, s0.q, s0.v, s0.cnt, s0.cap, s1.ctx));
    }
    for (int sd = 0; sd < nrun; ++sd) {
      Side& s = *run[sd];
      if (L.Wco) {
        LG_CHECK(gemm(st, EPI_BIAS, s, Br, s.ctx, D, nullptr, 0, 0x7fffffff, D, L.Wco, L.bco, D, s.msg, D));
        LG_CHECK(ffn(st, s, Br, dm, s.msg, L.cf0_w, L.cf0_b, L.cln_g, L.cln_b, L.cf3_w, L.cf3_b));
      } else {
        LG_CHECK(ffn(st, s, Br, dm, s.ctx, L.cf0_w, L.cf0_b, L.cln_g, L.cln_b, L.cf3_w, L.cf3_b));
      }
    }
    for (int sd = 0; sd < 2; ++sd) {
      Side& s = *sides[sd];
      float* ref = sd == 0 ? ref0 : ref1;
      if (ref && all_layers) {  // training-mode output: every layer's descriptors (lightglue.py:626-629)
        const size_t per = (size_t)s.cap * D;
        hipLaunchKernelGGL(lg_copy_rows_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, st, s.x, ref + (size_t)li * per,
                           s.cnt, s.cap, D, per * (size_t)w->n_layers);
        LG_CHECK(0);
      }
    }
  }
  // ---- assignment ------------------------------------------------------------------------------
  for (int sd = 0; sd < nrun; ++sd) {
    Side& s = *run[sd];
    LG_CHECK(gemm(st, EPI_DIV, s, Br, s.x, D, nullptr, 0, 0x7fffffff, D, w->proj_w, w->proj_b, D, s.q, D, sqrtf(sqrtf((float)D))));
    hipLaunchKernelGGL(lg_matchability_kernel, dim3((unsigned)einx_cdiv(s.cap, 4), (unsigned)Br), dim3(256), 0, st, s.x, s.cnt, s.cap, D, w->match_w,
                       w->match_b, s.cert, s.dust);
    LG_CHECK(0);
  }
  for (int sd = 0; sd < 2; ++sd) {
    Side& s = *sides[sd];
    float* ref = sd == 0 ? ref0 : ref1;
    if (ref && !all_layers) {
      const size_t per = (size_t)s.cap * D;
      hipLaunchKernelGGL(lg_copy_rows_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, st, s.x, ref, s.cnt, s.cap, D,
                         per);
      LG_CHECK(0);
    }
  }
  MnnArgs a;
  a.d0 = s0.q;
  a.d1 = s1.q;
  a.n = n;
  a.m = m;
  a.cap0 = cap0;
  a.cap1 = cap1;
  a.D = D;
  a.nc64 = einx_cdiv(cap1, 64);
  a.nr64 = einx_cdiv(cap0, WROWS);
  char* q = (char*)mnn_ws;
  a.rowkey = (unsigned long long*)q;
  q += al((size_t)B * cap0 * 8);
  a.colkey = (unsigned long long*)q;
  q += al((size_t)B * cap1 * 8);
  a.rowstat = (float*)q;
  q += al((size_t)B * cap0 * a.nc64 * 8);
  a.colstat = (float*)q;
  q += al((size_t)B * cap1 * a.nr64 * 8);
  a.rowlse = (float*)q;
  q += al((size_t)B * cap0 * 8);
  a.collse = (float*)q;
  a.la = la;
  a.cert0 = s0.cert;
  a.cert1 = s1.cert;
  a.dust0 = s0.dust;
  a.dust1 = s1.dust;
  const size_t keybytes = al((size_t)B * cap0 * 8) + al((size_t)B * cap1 * 8);
  if (hipMemsetAsync(mnn_ws, 0, keybytes, st) != hipSuccess) {
    einx_set_error("einx_lightglue: memset failed");
    return EINX_ERR_LAUNCH;
  }
  const dim3 grid((unsigned)einx_cdiv(cap1, BN), (unsigned)einx_cdiv(cap0, BM), (unsigned)B);
  const int mx = cap0 > cap1 ? cap0 : cap1;
  EINX_PROF("lg_assignment (3 tile passes + lse + finalize)", st);
  hipLaunchKernelGGL((mnn_tile_kernel<1, false>), grid, dim3(THREADS), 0, st, a);
  LG_CHECK(0);
  hipLaunchKernelGGL(mnn_lse_kernel, dim3((unsigned)einx_cdiv(mx + 1, 256), (unsigned)B), dim3(256), 0, st, a);
  LG_CHECK(0);
  if (la) hipLaunchKernelGGL((mnn_tile_kernel<6, true>), grid, dim3(THREADS), 0, st, a);  // arg-max + log_assignment write in one visit
  else hipLaunchKernelGGL((mnn_tile_kernel<0, true>), grid, dim3(THREADS), 0, st, a);
  LG_CHECK(0);
  hipLaunchKernelGGL(lg_finalize_kernel, dim3((unsigned)einx_cdiv(mx, 256), (unsigned)B), dim3(256), 0, st, a.rowkey, a.colkey, n, m, cap0, cap1,
                     w->filter_threshold, matches0, matches1, scores0, scores1);
  LG_CHECK(0);
#undef LG_CHECK
  return EINX_OK;
}
