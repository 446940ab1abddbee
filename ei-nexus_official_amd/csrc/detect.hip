// detect.hip -- keypoint detector post-processing on gfx950: score map, NMS fix-point,
// top-k quantile threshold (radix select), raster-order compaction.  All integer/compare work:
// results are bit-exact with oracle/einx_oracle.c (and hence with the reference's golden
// vectors); no host synchronisation anywhere in the chain.
//
// Replaces (reference file:line): core/modules/utils/detector_util.py:18-77 (logits_to_prob,
// depth_to_space), :138-164 (remove_border_points), :243-337 (fast_nms), :80-135
// (prob_map_to_points_map: quantile threshold), :451-484 (prob_map_to_positions_with_prob),
// core/modules/event_extractors/EventExtractors.py:544-550,561-562 (mask dilation + apply),
// core/modules/utils/util.py:52-66 (unpad_positions), EventExtractors.py:496-515 (filter).
#include <algorithm>

#include "einx_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// K3: one thread per cell (C==65) or per pixel (C==1).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool mask_on(const uint8_t* mask, int b, int H, int W, int h0, int w0, int Hp, int Wp, int y, int x,
                                        int dilate) {
  // mask zero-padded to [Hp,Wp]; optional 3x3 dilation evaluated inside the padded map.  The nine bytes are requested
  // together (clamped addresses, validity folded into the test): no dependent load-and-branch chain per pixel
  const int r = dilate ? 1 : 0;
  bool any = false;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx, sy = yy - h0, sx = xx - w0;
      const bool ok = dy >= -r && dy <= r && dx >= -r && dx <= r && yy >= 0 && yy < Hp && xx >= 0 && xx < Wp && sy >= 0 && sy < H && sx >= 0 && sx < W;
      const uint8_t m = mask[((size_t)b * H + min(max(sy, 0), H - 1)) * W + min(max(sx, 0), W - 1)];
      any |= ok && m != 0;
    }
  return any;
}

// 8 lanes per cell: lane (cell j, row i) of a wave is lane j + 8*i and owns channels 8i..8i+7, i.e.
// row i of the cell's 8x8 pixel block (pixel_shuffle: score[8h+i][8w+k] = prob[8i+k][h][w]).  The 64
// exponentials of a cell are computed once and in parallel; the softmax denominator is then summed
// by every lane in the oracle's order c = 0..64 (values fetched with lane shuffles), so the result
// is bit-identical to the one-thread-per-cell form.  A lane writes its 8 score pixels as two 16-byte
// stores; 8 neighbouring cells give 256 contiguous bytes per row.
__global__ __launch_bounds__(256) void score65_kernel(const float* logits, int B, int hc, int wc, const uint8_t* mask, int H, int W,
                                                      int h0, int w0, int dilate, int border, float* prob, float* score, int32_t* zero_ptr,
                                                      int zero_n, float* crop) {
  // (einx_extract: the NMS pass flags of the detection that follows are zeroed here instead of by a memset launch of their own)
  if ((int)(blockIdx.x * 256 + threadIdx.x) < zero_n) zero_ptr[blockIdx.x * 256 + threadIdx.x] = 0;
  const int cells = hc * wc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 7, i = lane >> 3;
  const int gid = (blockIdx.x * 4 + wave) * 8 + j;
  const bool valid = gid < B * cells;
  const int g = valid ? gid : B * cells - 1;
  const int b = g / cells, cell = g % cells;
  const int h = cell / wc, w = cell % wc;
  const float* l = logits + (size_t)b * 65 * cells + cell;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = l[(size_t)(8 * i + k) * cells];
  const float v64 = l[(size_t)64 * cells];
  float mx = v64;
#pragma unroll
  for (int k = 0; k < 8; ++k) mx = fmaxf(mx, v[k]);
  mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = einx_expf(v[k] - mx);
  const float e64 = einx_expf(v64 - mx);
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < 64; ++c) s = s + __shfl(e[c & 7], j + 8 * (c >> 3), 64);
  s = s + e64;
  if (!valid) return;
  const int Hp = hc * 8, Wp = wc * 8;
  float* pr = prob + (size_t)b * 65 * cells + cell;
  const int y = h * 8 + i;
  // events mask of the lane's 8 pixels (row y, columns 8w..8w+7), dilated 3x3 inside the padded map: the 3 x 10 bytes around
  // them are requested together (clamped addresses, validity folded into the bit), bit k of `cols` = some row has an event in
  // column 8w-1+k; a per-pixel mask_on() walk is up to 72 dependent byte loads per lane
  unsigned cols = 0x3ffu;
  if (mask) {
    cols = 0;
    const int r = dilate ? 1 : 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy, sy = yy - h0;
      const bool rok = dy >= -r && dy <= r && yy >= 0 && yy < Hp && sy >= 0 && sy < H;
      const uint8_t* mrow = mask + ((size_t)b * H + min(max(sy, 0), H - 1)) * W;
      uint8_t m[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) m[k] = mrow[min(max(w * 8 - 1 + k - w0, 0), W - 1)];
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int xx = w * 8 - 1 + k, sx = xx - w0;
        if (rok && xx >= 0 && xx < Wp && sx >= 0 && sx < W && m[k]) cols |= 1u << k;
      }
    }
  }
  const unsigned window = (mask && !dilate) ? 2u : 7u;  // without dilation only the centre column counts
  float out[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float p = e[k] / s;
    pr[(size_t)(8 * i + k) * cells] = p;
    const int x = w * 8 + k;
    float t = p;
    if (!((cols >> k) & window)) t = 0.0f;
    if (border > 0 && (y < border || y >= Hp - border || x < border || x >= Wp - border)) t = 0.0f;
    out[k] = t;
  }
  if (i == 0) pr[(size_t)64 * cells] = e64 / s;
  f32x4* dst = reinterpret_cast<f32x4*>(score + ((size_t)b * Hp + y) * Wp + w * 8);  // 32-byte aligned: Wp % 8 == 0
  dst[0] = f32x4{out[0], out[1], out[2], out[3]};
  dst[1] = f32x4{out[4], out[5], out[6], out[7]};
  if (crop) {  // the un-padded map of the output dict (the reference's `unpad`), written here instead of by a crop + clone afterwards
    const int uy = y - h0;
    if (uy >= 0 && uy < H) {
      float* cr = crop + ((size_t)b * H + uy) * W - w0 + w * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int ux = w * 8 + k - w0;
        if (ux >= 0 && ux < W) cr[k] = out[k];
      }
    }
  }
}

__global__ void score1_kernel(const float* logits, int B, int Hp, int Wp, const uint8_t* mask, int H, int W, int h0, int w0,
                              int dilate, int border, float* prob, float* score, int32_t* zero_ptr, int zero_n, float* crop) {
  const int n = Hp * Wp;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < zero_n) zero_ptr[gid] = 0;
  if (gid >= B * n) return;
  const int b = gid / n, p = gid % n;
  const int y = p / Wp, x = p % Wp;
  float v = einx_sigmoidf(logits[gid]);
  if (mask && !mask_on(mask, b, H, W, h0, w0, Hp, Wp, y, x, dilate)) v = 0.0f;
  if (border > 0 && (y < border || y >= Hp - border || x < border || x >= Wp - border)) v = 0.0f;
  prob[gid] = v;  // the reference's score aliases probability for cell-1 networks
  score[gid] = v;
  if (crop) {
    const int uy = y - h0, ux = x - w0;
    if (uy >= 0 && uy < H && ux >= 0 && ux < W) crop[((size_t)b * H + uy) * W + ux] = v;
  }
}

// get_dense_positions (detector_util.py:504-519) on an unpadded map: row p of image b = (y+0.5, x+0.5, score)
// in "yx" ordering, (x+0.5, y+0.5, score) in "xy"
__global__ void dense_positions_kernel(const float* score, int B, int H, int W, int xy, float* out) {
  const size_t n = (size_t)B * H * W;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  const int p = (int)(gid % ((size_t)H * W));
  const float fy = (float)(p / W) + 0.5f, fx = (float)(p % W) + 0.5f;
  float* o = out + gid * 3;
  o[0] = xy ? fx : fy;
  o[1] = xy ? fy : fx;
  o[2] = score[gid];
}

__global__ void border_kernel(float* score, int B, int Hp, int Wp, int border) {
  const int n = Hp * Wp;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= B * n) return;
  const int p = gid % n;
  const int y = p / Wp, x = p % Wp;
  if (y < border || y >= Hp - border || x < border || x >= Wp - border) score[gid] = 0.0f;
}

// ------------------------------------------------------------------------------------------
// K4: one suppression pass of fast_nms on a 32x64 tile (halo 2R), written as separable window
// scans so that every lane does the same fixed amount of LDS work (the naive 81-tap test makes a
// whole wave wait for its one surviving maximum on every pass):
//   A  R9[y][x]   = max(v[y][x-R..x+R])                                   (row scan)
//   B  ismax[y][x]= v>0 && max(v[y][x-R..x-1]) <  v && max(v[y][x+1..x+R]) <= v
//                        && max(R9[y-R..y-1][x]) <  v && max(R9[y+1..y+R][x]) <= v
//      (first maximum wins: earlier raster taps must be strictly smaller, later ones <=)
//   C  rowor[y][x]= OR(ismax[y][x-R..x+R]);   D  suppressed = OR(rowor[y-R..y+R][x]) && !ismax
// A pass for image b runs only while the previous pass changed that image (flags[b][it-1]); it
// raises flags[b][it] if it zeroes any non-zero pixel.  When a pass changes nothing src == dst, so
// skipped passes leave both ping-pong buffers holding the fix-point.
// ------------------------------------------------------------------------------------------
constexpr int NMS_TH = 32, NMS_TW = 64, NMS_MAXR = 4, NMS_THREADS = 512;

template <int R>
__global__ __launch_bounds__(NMS_THREADS) void nms_pass_kernel(const float* src, float* dst, int Hp, int Wp, int tilesX, int tilesY,
                                                       int32_t* flags, int it, int nIt) {
  constexpr int VH = NMS_TH + 4 * R, VW = NMS_TW + 4 * R;  // values: tile + halo 2R
  constexpr int MH = NMS_TH + 2 * R, MW = NMS_TW + 2 * R;  // is-max: tile + halo R
  __shared__ float vals[VH * VW];
  __shared__ float r9[VH * MW];
  __shared__ uint8_t ismax[MH * MW];
  __shared__ uint8_t rowor[MH * NMS_TW];
  __shared__ int changed;
  int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);  // neighbouring tiles (shared halo) on one XCD
  const int txi = bid % tilesX;
  bid /= tilesX;
  const int tyi = bid % tilesY;
  const int b = bid / tilesY;
  // passes 0 and 1 always run so that BOTH ping-pong buffers hold image b; from then on a pass
  // is skipped once the previous one changed nothing (fix-point reached).
  if (it >= 2 && flags[b * nIt + it - 1] == 0) return;
  const float* s = src + (size_t)b * Hp * Wp;
  float* d = dst + (size_t)b * Hp * Wp;
  const int y0 = tyi * NMS_TH, x0 = txi * NMS_TW;
  const int tid = threadIdx.x;
  if (tid == 0) changed = 0;
  // all of a thread's tile loads are issued before the first is consumed (clamped addresses, then select)
  constexpr int NLOAD = (VH * VW + NMS_THREADS - 1) / NMS_THREADS;
  float staged[NLOAD];
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int i = tid + k * NMS_THREADS;
    const int ii = i < VH * VW ? i : 0;
    const int y = y0 - 2 * R + ii / VW, x = x0 - 2 * R + ii % VW;
    const bool in = y >= 0 && y < Hp && x >= 0 && x < Wp;
    const float v = s[(size_t)(in ? y : 0) * Wp + (in ? x : 0)];
    staged[k] = in ? v : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int i = tid + k * NMS_THREADS;
    if (i < VH * VW) vals[i] = staged[k];
  }
  __syncthreads();
  // A: row-wise window maximum for every staged row, columns of the is-max grid
  for (int i = tid; i < VH * MW; i += NMS_THREADS) {
    const int vy = i / MW, mx = i % MW;
    const float* row = vals + vy * VW + mx;  // window = row[0 .. 2R], centre row[R]
    float m = row[0];
#pragma unroll
    for (int dx = 1; dx <= 2 * R; ++dx) m = fmaxf(m, row[dx]);
    r9[i] = m;
  }
  __syncthreads();
  // B: is-max on tile + halo R
  for (int i = tid; i < MH * MW; i += NMS_THREADS) {
    const int my = i / MW, mx = i % MW;
    const int vy = my + R;
    const float* row = vals + vy * VW + mx;
    const float c = row[R];
    const int y = y0 - R + my, x = x0 - R + mx;
    float left = row[0], right = row[R + 1];
#pragma unroll
    for (int dx = 1; dx < R; ++dx) {
      left = fmaxf(left, row[dx]);
      right = fmaxf(right, row[R + 1 + dx]);
    }
    float above = r9[(vy - R) * MW + mx], below = r9[(vy + 1) * MW + mx];
#pragma unroll
    for (int dy = 1; dy < R; ++dy) {
      above = fmaxf(above, r9[(vy - R + dy) * MW + mx]);
      below = fmaxf(below, r9[(vy + 1 + dy) * MW + mx]);
    }
    const bool ok = c > 0.0f && y >= 0 && y < Hp && x >= 0 && x < Wp && left < c && above < c && right <= c && below <= c;
    ismax[i] = ok ? 1 : 0;
  }
  __syncthreads();
  // C: row-wise OR of is-max for the tile's columns
  for (int i = tid; i < MH * NMS_TW; i += NMS_THREADS) {
    const int my = i / NMS_TW, tx = i % NMS_TW;
    const uint8_t* row = ismax + my * MW + tx;  // window row[0 .. 2R]
    unsigned o = 0;
#pragma unroll
    for (int dx = 0; dx <= 2 * R; ++dx) o |= row[dx];
    rowor[i] = (uint8_t)o;
  }
  __syncthreads();
  // D: suppress
  int my_changed = 0;
  for (int i = tid; i < NMS_TH * NMS_TW; i += NMS_THREADS) {
    const int ty = i / NMS_TW, tx = i % NMS_TW;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= Hp || x >= Wp) continue;
    const float c = vals[(ty + 2 * R) * VW + tx + 2 * R];
    unsigned o = 0;
#pragma unroll
    for (int dy = 0; dy <= 2 * R; ++dy) o |= rowor[(ty + dy) * NMS_TW + tx];
    // a maximum has no other maximum in its window, so "some maximum nearby and not one myself"
    const bool sup = o != 0 && !ismax[(ty + R) * MW + tx + R];
    float out = c;
    if (c != 0.0f && sup) {
      out = 0.0f;
      my_changed = 1;
    }
    d[(size_t)y * Wp + x] = out;
  }
  if (my_changed) changed = 1;
  __syncthreads();
  if (tid == 0 && changed) atomicOr(&flags[b * nIt + it], 1);
}

// ------------------------------------------------------------------------------------------
// The same pass for R = 4 (every shipped configuration) with register tiling: a work item is eight
// neighbouring columns of one row, read with 16-byte LDS loads, and the boolean planes are bit-packed
// (one byte per eight columns), so a pass issues ~4x fewer LDS instructions -- the LDS pipe, shared by
// the four resident workgroups of a CU, is what bounds the generic kernel above.  Same predicates:
//   ismax = v > 0 && inside && max(left 4) < v && max(right 4) <= v && max(4 rows above of R9) < v
//           && max(4 rows below of R9) <= v;   suppressed = (some ismax in the 9x9 window) && !ismax
// ------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) Nms4Smem {
  static constexpr int R = 4;
  static constexpr int VH = NMS_TH + 4 * R, VW = NMS_TW + 4 * R;  // 48 x 80 values: tile + halo 2R
  static constexpr int MH = NMS_TH + 2 * R, MW = NMS_TW + 2 * R;  // 40 x 72 is-max grid: tile + halo R
  static constexpr int G = MW / 8, TG = NMS_TW / 8;               // 9 column groups of the is-max grid, 8 of the tile
  float vals[VH * VW];
  float r9[VH * MW];
  uint8_t rowok8[VH * G];  // bit k of [vy][g]: centre > 0 and row-wise conditions hold at column g*8+k
  uint8_t ismax8[MH * (G + 1)];
  uint8_t rowor8[MH * TG];
  int changed;
};

// one tile of one pass: s -> d for the 32x64 tile at (y0, x0); raises sm.changed if it zeroes a non-zero pixel.  The caller
// clears sm.changed beforehand (followed by a barrier) and synchronises before reading it.
template <bool COHERENT>
__device__ __forceinline__ void nms4_tile(Nms4Smem& sm, const float* s, float* d, int Hp, int Wp, int y0, int x0) {
  constexpr int R = Nms4Smem::R, VH = Nms4Smem::VH, VW = Nms4Smem::VW, MH = Nms4Smem::MH, MW = Nms4Smem::MW, G = Nms4Smem::G, TG = Nms4Smem::TG;
  const int tid = threadIdx.x;
  constexpr int NLOAD = (VH * VW + NMS_THREADS - 1) / NMS_THREADS;
  float staged[NLOAD];
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int i = tid + k * NMS_THREADS;
    const int ii = i < VH * VW ? i : 0;
    const int y = y0 - 2 * R + ii / VW, x = x0 - 2 * R + ii % VW;
    const bool in = y >= 0 && y < Hp && x >= 0 && x < Wp;
    const float* src_p = s + (size_t)(in ? y : 0) * Wp + (in ? x : 0);
    // COHERENT: the previous pass of the SAME launch wrote these values (finisher): read them from L2, never from a stale L1 line
    const float v = COHERENT ? __hip_atomic_load(const_cast<float*>(src_p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src_p;
    staged[k] = in ? v : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int i = tid + k * NMS_THREADS;
    if (i < VH * VW) sm.vals[i] = staged[k];
  }
  __syncthreads();
  // A: per row and column group: R9 (9-wide row maximum) and the row-wise part of the is-max test
  if (tid < VH * G) {
    const int vy = tid / G, g = tid % G;
    float v[16];
    const float* row = sm.vals + vy * VW + g * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(row + 4 * q);
      v[4 * q] = t[0];
      v[4 * q + 1] = t[1];
      v[4 * q + 2] = t[2];
      v[4 * q + 3] = t[3];
    }
    float m9[8];
    unsigned ok = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float left = v[k], right = v[k + 5];
#pragma unroll
      for (int t = 1; t < 4; ++t) {
        left = fmaxf(left, v[k + t]);
        right = fmaxf(right, v[k + 5 + t]);
      }
      const float c = v[k + 4];
      m9[k] = fmaxf(fmaxf(left, right), c);
      ok |= (c > 0.0f && left < c && right <= c) ? (1u << k) : 0u;
    }
    float* o = sm.r9 + vy * MW + g * 8;
    *reinterpret_cast<f32x4*>(o) = f32x4{m9[0], m9[1], m9[2], m9[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{m9[4], m9[5], m9[6], m9[7]};
    sm.rowok8[tid] = (uint8_t)ok;
  }
  __syncthreads();
  // B: column-wise part on the is-max grid (rows my = 0..39 <-> vy = my + 4)
  if (tid < MH * G) {
    const int my = tid / G, g = tid % G;
    const int vy = my + R;
    float above[8], below[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) above[k] = below[k] = 0.0f;
#pragma unroll
    for (int dy = 1; dy <= R; ++dy) {
      const float* ra = sm.r9 + (vy - dy) * MW + g * 8;
      const float* rb = sm.r9 + (vy + dy) * MW + g * 8;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ra), a1 = *reinterpret_cast<const f32x4*>(ra + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(rb), b1 = *reinterpret_cast<const f32x4*>(rb + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        above[k] = dy == 1 ? a0[k] : fmaxf(above[k], a0[k]);
        above[k + 4] = dy == 1 ? a1[k] : fmaxf(above[k + 4], a1[k]);
        below[k] = dy == 1 ? b0[k] : fmaxf(below[k], b0[k]);
        below[k + 4] = dy == 1 ? b1[k] : fmaxf(below[k + 4], b1[k]);
      }
    }
    const float* crow = sm.vals + vy * VW + g * 8 + R;
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(crow), c1 = *reinterpret_cast<const f32x4*>(crow + 4);
    const unsigned rok = sm.rowok8[vy * G + g];
    const int y = y0 - R + my;
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float c = k < 4 ? c0[k] : c1[k - 4];
      const int x = x0 - R + g * 8 + k;
      const bool okk = ((rok >> k) & 1u) && y >= 0 && y < Hp && x >= 0 && x < Wp && above[k] < c && below[k] <= c;
      bits |= okk ? (1u << k) : 0u;
    }
    sm.ismax8[my * (G + 1) + g] = (uint8_t)bits;
    if (g == 0) sm.ismax8[my * (G + 1) + G] = 0;  // pad byte read by the last tile group
  }
  __syncthreads();
  // C: row-wise OR over the 9-wide window, for the tile's 64 columns (bit k of group t8: is-max columns t8*8+k .. +8)
  if (tid < MH * TG) {
    const int my = tid / TG, t8 = tid % TG;
    const unsigned w = (unsigned)sm.ismax8[my * (G + 1) + t8] | ((unsigned)sm.ismax8[my * (G + 1) + t8 + 1] << 8) |
                       ((unsigned)sm.ismax8[my * (G + 1) + t8 + 2] << 16);
    unsigned r = 0;
#pragma unroll
    for (int dx = 0; dx <= 2 * R; ++dx) r |= w >> dx;
    sm.rowor8[tid] = (uint8_t)(r & 0xFFu);
  }
  __syncthreads();
  // D: column-wise OR and suppression for 8 pixels of one tile row
  int my_changed = 0;
  if (tid < NMS_TH * TG) {
    const int ty = tid / TG, t8 = tid % TG;
    unsigned o8 = 0;
#pragma unroll
    for (int dy = 0; dy <= 2 * R; ++dy) o8 |= sm.rowor8[(ty + dy) * TG + t8];
    // is-max bits of the pixels themselves: grid row ty+R, columns t8*8 + k + R
    const unsigned iw = (unsigned)sm.ismax8[(ty + R) * (G + 1) + t8] | ((unsigned)sm.ismax8[(ty + R) * (G + 1) + t8 + 1] << 8);
    const unsigned is8 = (iw >> R) & 0xFFu;
    const unsigned sup = o8 & ~is8;
    const float* crow = sm.vals + (ty + 2 * R) * VW + t8 * 8 + 2 * R;
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(crow), c1 = *reinterpret_cast<const f32x4*>(crow + 4);
    const int y = y0 + ty;
    if (y < Hp) {
      float out[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float c = k < 4 ? c0[k] : c1[k - 4];
        const bool kill = c != 0.0f && ((sup >> k) & 1u);
        out[k] = kill ? 0.0f : c;
        if (kill && x0 + t8 * 8 + k < Wp) my_changed = 1;
      }
      float* o = d + (size_t)y * Wp + x0 + t8 * 8;
      if (x0 + t8 * 8 + 7 < Wp && (Wp & 3) == 0) {
        *reinterpret_cast<f32x4*>(o) = f32x4{out[0], out[1], out[2], out[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{out[4], out[5], out[6], out[7]};
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (x0 + t8 * 8 + k < Wp) o[k] = out[k];
      }
    }
  }
  if (my_changed) sm.changed = 1;
}

__global__ __launch_bounds__(NMS_THREADS) void nms_pass4_kernel(const float* src, float* dst, int Hp, int Wp, int tilesX, int tilesY,
                                                                int32_t* flags, int it, int nIt) {
  __shared__ Nms4Smem sm;
  int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);  // neighbouring tiles (shared halo) on one XCD
  const int txi = bid % tilesX;
  bid /= tilesX;
  const int tyi = bid % tilesY;
  const int b = bid / tilesY;
  if (it >= 2 && flags[b * nIt + it - 1] == 0) return;
  if (threadIdx.x == 0) sm.changed = 0;  // ordered before the tile's writes by the barriers inside nms4_tile
  nms4_tile<false>(sm, src + (size_t)b * Hp * Wp, dst + (size_t)b * Hp * Wp, Hp, Wp, tyi * NMS_TH, txi * NMS_TW);
  __syncthreads();
  if (threadIdx.x == 0 && sm.changed) atomicOr(&flags[b * nIt + it], 1);
}

// ------------------------------------------------------------------------------------------
// Finisher: images whose fix-point needs more passes than the enqueued wide ones (tie-heavy maps: 14-17 passes on
// quantised scores) are completed HERE, on the device, by one workgroup per such image that keeps ping-ponging between
// the two map buffers until a pass changes nothing -- no host round trip, no sticky pass budget.  Converged images
// (the normal case) cost one flag load.  A tile is revisited only while it or one of its 8 neighbours changed in the
// previous pass: an untouched tile is already identical in both buffers (its last change is at least two passes old),
// so later passes touch only the few contested tiles.  Loads of map values bypass L1 (the same CU wrote them a pass
// earlier in this launch); stores are complete (vmcnt(0) + barrier) before the next pass reads them.
// ------------------------------------------------------------------------------------------
constexpr int NMS_FIN_MAXT = 2048;  // tiles tracked individually; larger maps revisit every tile
// One workgroup finishes one image, so its time is passes x (tiles it revisits): bounded here (a 264x352 map of 8 score
// levels needs 25 passes).  Beyond the bound `not_converged` stays raised and the caller's retry (a larger wide-pass budget,
// then this finisher again) takes over, as for the other radii -- the stream never stalls on one pathological image.
// The launch passes max(256, Hp + Wp) (round 4), so that only serpentine-like maps ever need the host retry.
constexpr int kNmsFinishMaxPasses = 256;
__global__ __launch_bounds__(NMS_THREADS) void nms4_finish_kernel(float* bufA, float* bufB, int Hp, int Wp, int tilesX, int tilesY,
                                                                  int32_t* flags, int nIt, int max_passes) {
  __shared__ Nms4Smem sm;
  __shared__ uint8_t dirty[2][NMS_FIN_MAXT];
  __shared__ int any_changed;
  const int b = blockIdx.x;
  if (flags[b * nIt + nIt - 1] == 0) return;  // fix-point reached within the wide passes
  const int tid = threadIdx.x;
  const int ntiles = tilesX * tilesY;
  const bool track = ntiles <= NMS_FIN_MAXT;
  float* src = bufA + (size_t)b * Hp * Wp;  // the last wide pass wrote bufA
  float* dst = bufB + (size_t)b * Hp * Wp;
  if (track)
    for (int i = tid; i < ntiles; i += NMS_THREADS) {
      dirty[0][i] = 1;
      dirty[1][i] = 0;
    }
  __syncthreads();
  int cur = 0;
  int left = 1;
  for (int pass = 0; pass < max_passes; ++pass) {
    if (tid == 0) any_changed = 0;
    for (int t = 0; t < ntiles; ++t) {
      const int txi = t % tilesX, tyi = t / tilesX;
      bool need = !track;
      if (track) {
        for (int dy = -1; dy <= 1 && !need; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int yy = tyi + dy, xx = txi + dx;
            if (yy >= 0 && yy < tilesY && xx >= 0 && xx < tilesX && dirty[cur][yy * tilesX + xx]) need = true;
          }
      }
      if (!need) continue;  // workgroup-uniform: every thread reads the same LDS bytes
      if (tid == 0) sm.changed = 0;
      __syncthreads();
      nms4_tile<true>(sm, src, dst, Hp, Wp, tyi * NMS_TH, txi * NMS_TW);
      __syncthreads();
      if (tid == 0 && sm.changed) {
        any_changed = 1;
        if (track) dirty[cur ^ 1][t] = 1;
      }
      __syncthreads();
    }
    // this pass's stores have reached L2 before any thread starts the next pass
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    left = any_changed;
    if (track)
      for (int i = tid; i < ntiles; i += NMS_THREADS) dirty[cur][i] = 0;
    cur ^= 1;
    float* tmp = src;
    src = dst;
    dst = tmp;
    __syncthreads();
    if (!left) break;
  }
  // a pass that changes nothing leaves both buffers holding the fix-point, so whichever one the selection kernel reads is final
  if (tid == 0) flags[b * nIt + nIt - 1] = left ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// K5: per image one 1024-thread workgroup: radix select of the order statistics
// sorted[lo], sorted[hi] (ascending), thr = min(b - (b-a)*0.5, det_thr), then raster-order
// stream compaction of (v > thr) with wave ballots.
// ------------------------------------------------------------------------------------------
constexpr int SEL_THREADS = 1024;

struct SelArgs {
  const float* map;  // [B,Hp,Wp] NMS output
  const int32_t* flags;
  float* nms_out;  // [B,H,W] or null
  float* positions;
  int32_t* indices;
  int32_t* counts;
  float* thr_out;
  int32_t* not_converged;
  int Hp, Wp, H, W, h0, w0;
  int top_k, lo, hi, cap, ordering_xy, nIt;
  float det_thr;
};

__device__ __forceinline__ int block_excl_scan(int v, int* total, int* scratch /*>= 17 ints*/) {
  // exclusive prefix sum of v over the 1024-thread block (16 waves); scratch in LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < SEL_THREADS / 64; ++w) {
      const int t = scratch[w];
      scratch[w] = run;
      run += t;
    }
    scratch[16] = run;
  }
  __syncthreads();
  const int res = scratch[wave] + incl - v;
  *total = scratch[16];
  __syncthreads();
  return res;
}

// generic path: works for any input (negative values, dense maps); 6 passes over the map
__device__ void select_compact_generic(const SelArgs& a, unsigned* hist, int* scratch) {
  __shared__ unsigned sh_prefix, sh_rank;
  __shared__ unsigned sh_minkey;
  const int b = blockIdx.x;
  const int N = a.Hp * a.Wp;
  const float* m = a.map + (size_t)b * N;
  const int tid = threadIdx.x;

  float thr = a.det_thr;
  if (a.top_k > 0) {
    float tk = 0.0f;
    if (a.top_k < N) {
      // ---- radix select of the key at ascending rank lo (MSB first, 8 bits per pass) -------
      unsigned prefix = 0, rank = (unsigned)a.lo;
      unsigned less_total = 0;  // number of keys strictly below the selected prefix so far
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += SEL_THREADS) hist[i] = 0;
        __syncthreads();
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < N; i += SEL_THREADS) {
          const unsigned k = einx_ordered_key(m[i]);
          if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          unsigned run = 0, r = rank;
          int dsel = 255;
          for (int dgt = 0; dgt < 256; ++dgt) {
            const unsigned c = hist[dgt];
            if (r < run + c) {
              dsel = dgt;
              break;
            }
            run += c;
          }
          sh_prefix = prefix | ((unsigned)dsel << shift);
          sh_rank = r - run;
          hist[0] = run;  // keys below the chosen digit in this pass
        }
        __syncthreads();
        prefix = sh_prefix;
        rank = sh_rank;
        less_total += hist[0];
        __syncthreads();
      }
      const unsigned key_lo = prefix;
      const float v_lo = einx_ordered_unkey(key_lo);
      float v_hi = v_lo;
      if (a.hi != a.lo) {
        // count of keys <= key_lo decides whether rank hi still sits on the same value
        for (int i = tid; i < 256; i += SEL_THREADS) hist[i] = 0;
        if (tid == 0) sh_minkey = 0xFFFFFFFFu;
        __syncthreads();
        unsigned local_eq = 0, local_min = 0xFFFFFFFFu;
        for (int i = tid; i < N; i += SEL_THREADS) {
          const unsigned k = einx_ordered_key(m[i]);
          if (k == key_lo) ++local_eq;
          if (k > key_lo && k < local_min) local_min = k;
        }
        atomicAdd(&hist[0], local_eq);
        atomicMin(&sh_minkey, local_min);
        __syncthreads();
        const unsigned le = less_total + hist[0];
        if ((unsigned)a.hi >= le) v_hi = einx_ordered_unkey(sh_minkey);
        __syncthreads();
      }
      tk = v_hi - (v_hi - v_lo) * 0.5f;
    }
    thr = fminf(tk, a.det_thr);
  }
  if (tid == 0) {
    a.thr_out[b] = thr;
    a.not_converged[b] = a.nIt > 0 ? a.flags[b * a.nIt + a.nIt - 1] : 0;
  }

  // ---- thresholded map (cropped) + raster-order compaction ----------------------------------
  int base = 0;
  for (int i0 = 0; i0 < N; i0 += SEL_THREADS) {
    const int i = i0 + tid;
    float v = 0.0f;
    bool keep = false;
    int y = 0, x = 0;
    if (i < N) {
      v = m[i];
      y = i / a.Wp;
      x = i % a.Wp;
      const bool above = v > thr;
      const int uy = y - a.h0, ux = x - a.w0;
      const bool inside = uy >= 0 && uy < a.H && ux >= 0 && ux < a.W;
      if (a.nms_out && inside) a.nms_out[((size_t)b * a.H + uy) * a.W + ux] = above ? v : 0.0f;
      keep = above && v > 0.0f && inside;
    }
    int total;
    const int pos = base + block_excl_scan(keep ? 1 : 0, &total, scratch);
    if (keep && pos < a.cap) {
      const float py = ((float)y + 0.5f) - (float)a.h0, px = ((float)x + 0.5f) - (float)a.w0;
      float* o = a.positions + ((size_t)b * a.cap + pos) * 3;
      o[0] = a.ordering_xy ? px : py;
      o[1] = a.ordering_xy ? py : px;
      o[2] = v;
      a.indices[(size_t)b * a.cap + pos] = i;
    }
    base += total;
  }
  if (tid == 0) a.counts[b] = base;
}


// Fast path (the one the extractors hit): after NMS almost every pixel is zero, so one float4
// sweep compacts the non-zero candidates (value + flat index, raster order preserved) into LDS;
// the radix select and the final compaction then run on <= SEL_LCAP candidates instead of on the
// whole map.  Falls back to the generic path when the map is dense or holds negative values.
constexpr int SEL_LCAP = 6144;

__global__ __launch_bounds__(SEL_THREADS) void select_compact_kernel(const SelArgs a) {
  __shared__ unsigned hist[256];
  __shared__ int scratch[17];
  __shared__ float cval[SEL_LCAP];
  __shared__ int cidx[SEL_LCAP];
  __shared__ int sh_bad;
  __shared__ unsigned sh_pre, sh_rk;
  const int b = blockIdx.x;
  const int N = a.Hp * a.Wp;
  const float* m = a.map + (size_t)b * N;
  const int tid = threadIdx.x;
  if (tid == 0) sh_bad = 0;
  __syncthreads();
  // ---- sweep 1: ordered compaction of non-zeros into LDS ---------------------------------------
  // Each of the 16 waves owns one contiguous slice of the map: count its non-zeros, one block-level
  // prefix over the 16 counts, then compact the slice at its offset with wave-local ballots only
  // (no block barrier inside the loops; the second read of the slice comes from L2).
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int NWAVES = SEL_THREADS / 64;
  const int per_wave = ((N + NWAVES - 1) / NWAVES + 255) & ~255;  // multiple of 256 = 64 lanes x 4
  const int w0 = wave * per_wave, w1 = min(N, w0 + per_wave);
  const bool vec = (reinterpret_cast<size_t>(m) & 15) == 0;
  auto load4 = [&](int i, float* v) {
    if (vec && i + 3 < N) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(m + i);
      v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = (i + t < N) ? m[i + t] : 0.0f;
    }
  };
  // whole 16-byte groups only (every shipped geometry): SEL_UB loads per lane are requested together from clamped
  // addresses and zeroed when they lie past the slice -- a loop of one guarded load per iteration is neither unrolled nor
  // pipelined by hipcc and costs a memory round trip per 256 values (23 in a row on a 264x352 map)
  constexpr int SEL_UB = 12;
  const bool fast = vec && (N & 3) == 0 && N >= 4;
  auto load_batch = [&](int i0, f32x4* q) {
#pragma unroll
    for (int u = 0; u < SEL_UB; ++u) {
      const int idx = i0 + u * 256 + lane * 4;
      q[u] = *reinterpret_cast<const f32x4*>(m + min(idx, N - 4));
      if (idx >= w1) q[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  int cnt_w = 0;
  bool neg = false;
  if (fast) {
    for (int i0 = w0; i0 < w1; i0 += 256 * SEL_UB) {
      f32x4 q[SEL_UB];
      load_batch(i0, q);
#pragma unroll
      for (int u = 0; u < SEL_UB; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          cnt_w += (q[u][t] != 0.0f) ? 1 : 0;
          neg = neg || (q[u][t] < 0.0f) || (q[u][t] != q[u][t]);
        }
    }
  } else {
    for (int i0 = w0; i0 < w1; i0 += 256) {
      float v[4];
      load4(i0 + lane * 4, v);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        cnt_w += (v[t] != 0.0f) ? 1 : 0;
        neg = neg || (v[t] < 0.0f) || (v[t] != v[t]);
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) cnt_w += __shfl_xor(cnt_w, off, 64);
  if (neg) sh_bad = 1;
  if (lane == 0) scratch[wave] = cnt_w;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < NWAVES; ++w) {
      const int t = scratch[w];
      scratch[w] = run;
      run += t;
    }
    scratch[16] = run;
  }
  __syncthreads();
  const int nz = scratch[16];
  if (!sh_bad && nz <= SEL_LCAP) {
    int pos_w = scratch[wave];
    // 256 values of the slice in raster order (lane-major, then the lane's four elements).  Rank of a non-zero = non-zeros
    // of lower lanes + of the lane's earlier elements: four ballots and masked population counts (v_mbcnt), no dependent
    // chain of cross-lane shuffles (the 6-step wave scan this replaces was most of the kernel's longest phase)
    auto place = [&](int i, const float* v) {
      int before = 0, total = 0;
      bool nzt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        nzt[t] = v[t] != 0.0f;
        const unsigned long long m = __ballot(nzt[t]);
        before += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        total += __popcll(m);
      }
      int pos = pos_w + before;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (nzt[t]) {
          cval[pos] = v[t];
          cidx[pos] = i + t;
          ++pos;
        }
      pos_w += total;
    };
    if (fast) {
      for (int i0 = w0; i0 < w1; i0 += 256 * SEL_UB) {
        f32x4 q[SEL_UB];
        load_batch(i0, q);
#pragma unroll
        for (int u = 0; u < SEL_UB; ++u) {
          const float v[4] = {q[u][0], q[u][1], q[u][2], q[u][3]};
          if (i0 + u * 256 < w1) place(i0 + u * 256 + lane * 4, v);
        }
      }
    } else {
      for (int i0 = w0; i0 < w1; i0 += 256) {
        float v[4];
        load4(i0 + lane * 4, v);
        place(i0 + lane * 4, v);
      }
    }
  }
  __syncthreads();
  if (sh_bad || nz > SEL_LCAP) {  // uniform across the block: dense map or negative values
    select_compact_generic(a, hist, scratch);
    return;
  }
  // ---- threshold from the order statistics ------------------------------------------------------
  // sorted[lo] by a 4-pass radix select on the candidates (the 256-bin histogram is scanned by 256 threads,
  // not by one); sorted[hi], hi <= lo + 1, is then either the same value (duplicates / zeros reach rank
  // hi) or the smallest candidate above it: one counting pass instead of a second radix select.
  float thr = a.det_thr;
  if (a.top_k > 0) {
    float tk = 0.0f;
    if (a.top_k < N) {
      const int zeros = N - nz;  // all candidates are > 0, so zeros occupy ranks [0, zeros)
      float v0 = 0.0f;
      if (a.lo >= zeros) {
        unsigned prefix = 0, rank = (unsigned)(a.lo - zeros);
        for (int pass = 0; pass < 4; ++pass) {
          const int shift = 24 - 8 * pass;
          for (int i = tid; i < 256; i += SEL_THREADS) hist[i] = 0;
          __syncthreads();
          const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
          for (int i = tid; i < nz; i += SEL_THREADS) {
            const unsigned k = einx_ordered_key(cval[i]);
            if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
          }
          __syncthreads();
          // parallel bin scan: thread d < 256 owns bin d
          const unsigned c = tid < 256 ? hist[tid] : 0u;
          unsigned incl = c;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
          }
          if (lane == 63 && wave < 4) scratch[wave] = (int)incl;
          __syncthreads();
          if (tid < 256) {
            unsigned basew = 0;
            for (int w = 0; w < wave; ++w) basew += (unsigned)scratch[w];
            const unsigned excl = basew + incl - c;
            if ((rank >= excl && rank < excl + c) || (tid == 255 && rank >= excl + c)) {  // the last bin also catches an out-of-range rank
              sh_pre = prefix | ((unsigned)tid << shift);
              sh_rk = rank - excl;
            }
          }
          __syncthreads();
          prefix = sh_pre;
          rank = sh_rk;
          __syncthreads();
        }
        v0 = einx_ordered_unkey(prefix);
      }
      float v1 = v0;
      if (a.hi != a.lo) {
        int le = 0;
        float mn = einx_u2f(0x7f800000u);
        for (int i = tid; i < nz; i += SEL_THREADS) {
          const float v = cval[i];
          le += v <= v0 ? 1 : 0;
          if (v > v0) mn = fminf(mn, v);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          le += __shfl_xor(le, off, 64);
          mn = fminf(mn, __shfl_xor(mn, off, 64));
        }
        __syncthreads();
        if (lane == 0) {
          scratch[wave] = le;
          hist[wave] = einx_f2u(mn);
        }
        __syncthreads();
        int le_all = zeros;
        float mn_all = einx_u2f(0x7f800000u);
        for (int w = 0; w < NWAVES; ++w) {
          le_all += scratch[w];
          mn_all = fminf(mn_all, einx_u2f(hist[w]));
        }
        __syncthreads();
        if (a.hi >= le_all) v1 = mn_all;  // rank hi lies beyond every value <= v0
      }
      tk = v1 - (v1 - v0) * 0.5f;
    }
    thr = fminf(tk, a.det_thr);
  }
  if (tid == 0) {
    a.thr_out[b] = thr;
    a.not_converged[b] = a.nIt > 0 ? a.flags[b * a.nIt + a.nIt - 1] : 0;
  }
  // ---- final raster-order compaction from the candidate list ----------------------------------------
  int base = 0;
  for (int i0 = 0; i0 < nz; i0 += SEL_THREADS) {
    const int i = i0 + tid;
    bool keep = false;
    float v = 0.0f;
    int fi = 0, y = 0, x = 0;
    if (i < nz) {
      v = cval[i];
      fi = cidx[i];
      y = fi / a.Wp;
      x = fi % a.Wp;
      const int uy = y - a.h0, ux = x - a.w0;
      keep = v > thr && uy >= 0 && uy < a.H && ux >= 0 && ux < a.W;
    }
    int total;
    const int pos = base + block_excl_scan(keep ? 1 : 0, &total, scratch);
    if (keep && pos < a.cap) {
      const float py = ((float)y + 0.5f) - (float)a.h0, px = ((float)x + 0.5f) - (float)a.w0;
      float* o = a.positions + ((size_t)b * a.cap + pos) * 3;
      o[0] = a.ordering_xy ? px : py;
      o[1] = a.ordering_xy ? py : px;
      o[2] = v;
      a.indices[(size_t)b * a.cap + pos] = fi;
    }
    base += total;
  }
  if (tid == 0) a.counts[b] = base;
}

// thresholded NMS map, cropped to the unpadded window (the `nms` output): nms_out[b,y,x] = v > thr[b] ? v : 0
__global__ void nms_crop_kernel(const float* map, const float* thr, int B, int Hp, int Wp, int h0, int w0, int H, int W, float* out) {
  const size_t n = (size_t)B * H * W;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  const int b = (int)(gid / ((size_t)H * W));
  const int r = (int)(gid % ((size_t)H * W));
  const float v = map[((size_t)b * Hp + (r / W + h0)) * Wp + r % W + w0];
  out[gid] = v > thr[b] ? v : 0.0f;
}

void topk_ranks(int N, int k, int* lo, int* hi) {
  // fp32 arithmetic of the reference: q = float32(N-k)/float32(N); rank = q*float32(N-1)
  const float q = (float)(N - k) / (float)N;
  const float rank = q * (float)(N - 1);
  *lo = (int)floorf(rank);
  *hi = (int)ceilf(rank);
}

}  // namespace

EINX_EXPORT int einx_score_map(const float* logits, int B, int C, int hc, int wc, const uint8_t* mask, int H, int W, int h0, int w0,
                               int dilate, int border, float* prob, float* score, void* stream) {
  return einx_score_map_zero(logits, B, C, hc, wc, mask, H, W, h0, w0, dilate, border, prob, score, nullptr, 0, nullptr, stream);
}

// the same launch, which also zeroes zero_n int32 words at zero_ptr (einx_extract: the NMS pass flags of the detection that follows;
// returns EINX_ERR_ARG when the launch has fewer threads than words -- the caller then keeps the memset)
int einx_score_map_zero(const float* logits, int B, int C, int hc, int wc, const uint8_t* mask, int H, int W, int h0, int w0, int dilate,
                        int border, float* prob, float* score, int32_t* zero_ptr, int zero_n, float* crop, void* stream) {
  EINX_CHECK_ARG(!crop || (H > 0 && W > 0), "the cropped map needs the un-padded size");
  EINX_CHECK_ARG(logits && prob && score, "null pointer");
  EINX_CHECK_ARG(zero_n == 0 || (zero_ptr && (long)zero_n <= (long)(C == 65 ? einx_cdiv(B * hc * wc, 32) : einx_cdiv(B * hc * wc, 256)) * 256),
                 "more words to zero than threads in the launch");
  EINX_CHECK_ARG(C == 65 || C == 1, "detector head must have 65 or 1 channels");
  EINX_CHECK_ARG(B > 0 && hc > 0 && wc > 0, "bad shape");
  hipStream_t s = (hipStream_t)stream;
  EINX_PROF("score_map (score65/score1)", s);
  if (C == 65) {
    const int n = B * hc * wc;
    hipLaunchKernelGGL(score65_kernel, dim3(einx_cdiv(n, 32)), dim3(256), 0, s, logits, B, hc, wc, mask, H, W, h0, w0, dilate, border,
                       prob, score, zero_ptr, zero_n, crop);
  } else {
    const int n = B * hc * wc;
    hipLaunchKernelGGL(score1_kernel, dim3(einx_cdiv(n, 256)), dim3(256), 0, s, logits, B, hc, wc, mask, H, W, h0, w0, dilate, border,
                       prob, score, zero_ptr, zero_n, crop);
  }
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_remove_border(float* score, int B, int Hp, int Wp, int border, void* stream) {
  EINX_CHECK_ARG(score, "null pointer");
  if (border <= 0) return EINX_OK;
  const int n = B * Hp * Wp;
  hipLaunchKernelGGL(border_kernel, dim3(einx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, score, B, Hp, Wp, border);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_dense_positions(const float* score, int B, int H, int W, int ordering_xy, float* out, void* stream) {
  EINX_CHECK_ARG(score && out, "null pointer");
  EINX_CHECK_ARG(B > 0 && H > 0 && W > 0, "bad shape");
  const size_t n = (size_t)B * H * W;
  hipLaunchKernelGGL(dense_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, score, B, H, W, ordering_xy, out);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT size_t einx_detect_ws_bytes(const einx_detect_params* p) {
  if (!p) return 0;
  const size_t map = (size_t)p->B * p->Hp * p->Wp * sizeof(float);
  const size_t flags = (size_t)p->B * (p->nms_iters > 0 ? p->nms_iters : 1) * sizeof(int32_t);
  return 2 * map + ((flags + 255) & ~(size_t)255) + 256;
}

EINX_EXPORT int einx_detect(const float* score, const einx_detect_params* p, void* ws, float* nms_out, float* positions,
                            int32_t* indices, int32_t* counts, float* thr, int32_t* not_converged, void* stream) {
  return einx_detect_prezeroed(score, p, ws, nms_out, positions, indices, counts, thr, not_converged, 0, nullptr, stream);
}

// where einx_detect keeps its B x nms_iters pass flags inside `ws` (einx_extract has the score kernel zero them)
int32_t* einx_detect_flags(const einx_detect_params* p, void* ws, int* n) {
  const size_t map_bytes = (size_t)p->B * p->Hp * p->Wp * sizeof(float);
  *n = p->radius > 0 ? p->B * p->nms_iters : 0;
  return (int32_t*)((char*)ws + 2 * map_bytes);
}

int einx_detect_prezeroed(const float* score, const einx_detect_params* p, void* ws, float* nms_out, float* positions, int32_t* indices,
                          int32_t* counts, float* thr, int32_t* not_converged, int flags_zeroed, const float** final_map, void* stream) {
  EINX_CHECK_ARG(score && p && ws && positions && indices && counts && thr && not_converged, "null pointer");
  EINX_CHECK_ARG(p->B > 0 && p->Hp > 0 && p->Wp > 0 && p->cap > 0, "bad shape");
  EINX_CHECK_ARG(p->radius >= 0 && p->radius <= NMS_MAXR, "nms radius must be in 0..4");
  EINX_CHECK_ARG(p->radius == 0 || p->nms_iters >= 1, "nms_iters must be >= 1");
  EINX_CHECK_ARG((size_t)p->Hp * p->Wp < (1u << 30), "map too large");
  hipStream_t s = (hipStream_t)stream;
  const int N = p->Hp * p->Wp;
  const size_t map_bytes = (size_t)p->B * N * sizeof(float);
  float* buf0 = (float*)ws;
  float* buf1 = (float*)((char*)ws + map_bytes);
  int32_t* flags = (int32_t*)((char*)ws + 2 * map_bytes);
  const float* cur = score;
  const int nIt = p->radius > 0 ? p->nms_iters : 0;
  if (nIt > 0) {
    if (!flags_zeroed && hipMemsetAsync(flags, 0, (size_t)p->B * nIt * sizeof(int32_t), s) != hipSuccess) {
      einx_set_error("einx_detect: memset failed");
      return EINX_ERR_LAUNCH;
    }
    const int tilesX = einx_cdiv(p->Wp, NMS_TW), tilesY = einx_cdiv(p->Hp, NMS_TH);
    const dim3 grid((unsigned)(tilesX * tilesY * p->B));
    for (int it = 0; it < nIt; ++it) {
      float* dst = (it & 1) ? buf1 : buf0;
      EINX_PROF("nms_pass", s);
      switch (p->radius) {
        case 1: hipLaunchKernelGGL(nms_pass_kernel<1>, grid, dim3(NMS_THREADS), 0, s, cur, dst, p->Hp, p->Wp, tilesX, tilesY, flags, it, nIt); break;
        case 2: hipLaunchKernelGGL(nms_pass_kernel<2>, grid, dim3(NMS_THREADS), 0, s, cur, dst, p->Hp, p->Wp, tilesX, tilesY, flags, it, nIt); break;
        case 3: hipLaunchKernelGGL(nms_pass_kernel<3>, grid, dim3(NMS_THREADS), 0, s, cur, dst, p->Hp, p->Wp, tilesX, tilesY, flags, it, nIt); break;
        default: hipLaunchKernelGGL(nms_pass4_kernel, grid, dim3(NMS_THREADS), 0, s, cur, dst, p->Hp, p->Wp, tilesX, tilesY, flags, it, nIt); break;
      }
      EINX_CHECK_LAUNCH();
      cur = dst;
    }
    if (p->radius == 4) {
      // images that are still changing after the wide passes are finished on the device (see nms4_finish_kernel); for the
      // other radii the caller re-runs with a larger budget when not_converged is raised
      float* last = const_cast<float*>(cur);
      float* other = last == buf0 ? buf1 : buf0;
      EINX_PROF("nms4_finish_kernel", s);
      // bound: realistic slow cases are monotone ramps, whose chain of successive maxima advances >= 5 pixels per pass
      hipLaunchKernelGGL(nms4_finish_kernel, dim3((unsigned)p->B), dim3(NMS_THREADS), 0, s, last, other, p->Hp, p->Wp, tilesX, tilesY, flags, nIt,
                         std::max(kNmsFinishMaxPasses, p->Hp + p->Wp));
      EINX_CHECK_LAUNCH();
    }
  }
  SelArgs a;
  a.map = cur;
  a.flags = flags;
  // the thresholded, cropped map (the `nms` output) is written by nms_crop_kernel below: the selection kernel's fast path never walks
  // the whole map after the threshold is known (its generic path could, and used to write the map a second time)
  a.nms_out = nullptr;
  a.positions = positions;
  a.indices = indices;
  a.counts = counts;
  a.thr_out = thr;
  a.not_converged = not_converged;
  a.Hp = p->Hp;
  a.Wp = p->Wp;
  a.H = p->H;
  a.W = p->W;
  a.h0 = p->h0;
  a.w0 = p->w0;
  a.top_k = p->top_k;
  a.cap = p->cap;
  a.ordering_xy = p->ordering_xy;
  a.nIt = nIt;
  a.det_thr = p->det_thr;
  a.lo = a.hi = 0;
  if (p->top_k > 0 && p->top_k < N) topk_ranks(N, p->top_k, &a.lo, &a.hi);
  {
    EINX_PROF("select_compact_kernel", s);
    hipLaunchKernelGGL(select_compact_kernel, dim3(p->B), dim3(SEL_THREADS), 0, s, a);
  }
  if (final_map) *final_map = a.map;
  if (nms_out && !final_map) {
    EINX_PROF("nms_crop_kernel", s);
    EINX_CHECK_LAUNCH();
    const size_t n = (size_t)p->B * p->H * p->W;
    hipLaunchKernelGGL(nms_crop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.map, thr, p->B, p->Hp, p->Wp, p->h0, p->w0, p->H, p->W,
                       nms_out);
  }
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
