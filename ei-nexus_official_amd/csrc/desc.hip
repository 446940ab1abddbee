// desc.hip -- descriptor post-processing on gfx950: sparse sampling (bilinear / gather) with
// L2 normalisation, dense-map normalisation, dense bilinear upsample + normalisation.
//
// Replaces (reference file:line): core/modules/utils/descriptor_util.py:74-128
// (sparsify_low_resolution_descriptors: grid_sample + F.normalize), :50-71
// (sparsify_full_resolution_descriptors), :21-28 (normalize_descriptors), :131-138
// (upsample_descriptors) and Padder.unpad (core/modules/utils/util.py:34-50) for the dense map.
// Arithmetic order mirrors oracle/einx_oracle.c exactly (bit-equal results).
#include "einx_common.h"

namespace {

__device__ __forceinline__ float wave_butterfly_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}

// one wave per keypoint; lane l owns channels l, l+64, ...
// CL: `raw` is the channels-last copy [B, hc*wc, D] written by normalize_map_tile_kernel: the four
// taps of a keypoint are four contiguous D-float rows (coalesced), instead of 4*D words that each
// sit in a different channel plane (one cache line per lane and tap).
template <bool BILINEAR, bool CL = false>
__global__ __launch_bounds__(256) void desc_sample_kernel(const float* raw, int D, int hc, int wc, int Hp, int Wp,
                                                          const int32_t* indices, const int32_t* counts, int cap, float scale,
                                                          float* out, const EinxWatch watch, int main_blocks, const EinxCrop crop,
                                                          int watch_blocks) {
  const int b = blockIdx.y;
  const int kp = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x >= main_blocks) {
    if ((int)blockIdx.x >= main_blocks + watch_blocks) {  // extra workgroups: the cropped NMS map of image b (einx_extract)
      einx_crop_block(crop, b, (int)blockIdx.x - main_blocks - watch_blocks, (int)threadIdx.x);
      return;
    }
    // spare workgroups of image 0: the extractor's weight watch, four tensors each
    const int t = ((int)blockIdx.x - main_blocks) * 4 + (threadIdx.x >> 6);
    if (b == 0 && t < watch.n) einx_watch_tensor(watch, t, lane);
    return;
  }
  int cnt = counts[b];
  cnt = cnt < cap ? cnt : cap;
  if (kp >= cnt) return;
  const int fi = indices[(size_t)b * cap + kp];
  const size_t plane = (size_t)hc * wc;
  const float* rb = raw + (size_t)b * D * plane;
  float vals[8];  // D <= 512
  float part = 0.0f;
  if (BILINEAR) {
    const int y = fi / Wp, x = fi % Wp;
    const float py = ((float)y + 0.5f) - 0.5f, px = ((float)x + 0.5f) - 0.5f;
    const float gy = 2.0f * (py / (float)(Hp - 1)) - 1.0f;
    const float gx = 2.0f * (px / (float)(Wp - 1)) - 1.0f;
    const float iy = ((gy + 1.0f) * (float)hc - 1.0f) / 2.0f;
    const float ix = ((gx + 1.0f) * (float)wc - 1.0f) / 2.0f;
    const float fx = floorf(ix), fy = floorf(iy);
    const float w = ix - fx, e = 1.0f - w, n = iy - fy, s = 1.0f - n;
    const float nw = s * e, ne = s * w, sw = n * e, se = n * w;
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < wc, vx1 = x1 >= 0 && x1 < wc, vy0 = y0 >= 0 && y0 < hc, vy1 = y1 >= 0 && y1 < hc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + 64 * i;
      float t = 0.0f;
      if (c < D) {
        float a, bb, cc, dd;
        if (CL) {
          a = (vy0 && vx0) ? rb[(size_t)(y0 * wc + x0) * D + c] : 0.0f;
          bb = (vy0 && vx1) ? rb[(size_t)(y0 * wc + x1) * D + c] : 0.0f;
          cc = (vy1 && vx0) ? rb[(size_t)(y1 * wc + x0) * D + c] : 0.0f;
          dd = (vy1 && vx1) ? rb[(size_t)(y1 * wc + x1) * D + c] : 0.0f;
        } else {
          const float* p = rb + (size_t)c * plane;
          a = (vy0 && vx0) ? p[y0 * wc + x0] : 0.0f;
          bb = (vy0 && vx1) ? p[y0 * wc + x1] : 0.0f;
          cc = (vy1 && vx0) ? p[y1 * wc + x0] : 0.0f;
          dd = (vy1 && vx1) ? p[y1 * wc + x1] : 0.0f;
        }
        t = a * nw;
        t = t + bb * ne;
        t = t + cc * sw;
        t = t + dd * se;
        part = fmaf(t, t, part);
      }
      vals[i] = t;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + 64 * i;
      float t = 0.0f;
      if (c < D) {
        t = rb[(size_t)c * plane + fi];
        part = fmaf(t, t, part);
      }
      vals[i] = t;
    }
  }
  const float nrm = sqrtf(wave_butterfly_sum(part));
  const float den = fmaxf(nrm, 1e-12f);
  float* o = out + ((size_t)b * cap + kp) * D;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    if (c < D) o[c] = scale * (vals[i] / den);
  }
}

// one wave per row of a [R,C] matrix: F.normalize(x, dim=1) * scale; lane l accumulates columns
// l, l+64, ... (sequential fmaf) then the xor butterfly, like the sparse descriptor kernels
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* x, int R, int C, float scale, float* out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* r = x + (size_t)row * C;
  float part = 0.0f;
  for (int c = lane; c < C; c += 64) part = fmaf(r[c], r[c], part);
  const float den = fmaxf(sqrtf(wave_butterfly_sum(part)), 1e-12f);
  float* o = out + (size_t)row * C;
  for (int c = lane; c < C; c += 64) o[c] = scale * (r[c] / den);
}

__global__ void random_positions_kernel(const float* u, int R, float s0, float s1, float* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R) return;
  out[t * 3 + 0] = u[t * 2 + 0] * s0;
  out[t * 3 + 1] = u[t * 2 + 1] * s1;
  out[t * 3 + 2] = 0.0f;
}

// normalize_descriptors through LDS: a workgroup stages PT pixels x D channels with fully coalesced loads, NM_BATCH of them
// in flight per thread (clamped addresses, no branch around the load: a predicated loop body is neither unrolled nor
// pipelined by hipcc and costs one HBM round trip per channel), one lane per pixel then walks the channels as the same
// sequential fmaf chain c = 0..D-1, and all threads write the normalised map -- plus, optionally, the channels-last copy of
// the raw map that desc_sample_kernel<.., CL> gathers from (one pixel's D channels per wave pass: contiguous stores, no
// index division).  Element (c, pixel) lives at lds[c*PT + ((pixel + c) & (PT-1))]: conflict-free along pixels and channels.
constexpr int NM_BATCH = 16;
template <int PT>
__global__ __launch_bounds__(256) void normalize_map_tile_kernel(const float* raw, int D, int P, float scale, float* out, float* raw_cl) {
  extern __shared__ float tile[];
  __shared__ float s_den[PT];
  const int b = blockIdx.y, p0 = blockIdx.x * PT;
  const int tid = threadIdx.x;
  const float* rb = raw + (size_t)b * D * P;
  const int px = tid % PT, cgrp = tid / PT;
  constexpr int CG = 256 / PT;  // channels handled per sweep
  const bool pv = p0 + px < P;
  const float* src = rb + (pv ? p0 + px : P - 1);
  {
    int c = cgrp;
    for (; c + (NM_BATCH - 1) * CG < D; c += NM_BATCH * CG) {
      float v[NM_BATCH];
#pragma unroll
      for (int u = 0; u < NM_BATCH; ++u) v[u] = src[(size_t)(c + u * CG) * P];
#pragma unroll
      for (int u = 0; u < NM_BATCH; ++u) tile[(c + u * CG) * PT + ((px + c + u * CG) & (PT - 1))] = pv ? v[u] : 0.0f;
    }
    for (; c < D; c += CG) tile[c * PT + ((px + c) & (PT - 1))] = pv ? src[(size_t)c * P] : 0.0f;
  }
  __syncthreads();
  if (tid < PT) {
    // one lane per pixel walks the channels in order; the LDS reads of 16 channels are in flight at a time so
    // the chain runs at fmaf latency, not at one LDS round trip per channel
    float s = 0.0f;
    int c = 0;
    for (; c + 16 <= D; c += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = tile[(c + u) * PT + ((tid + c + u) & (PT - 1))];
#pragma unroll
      for (int u = 0; u < 16; ++u) s = fmaf(v[u], v[u], s);
    }
    for (; c < D; ++c) {
      const float v = tile[c * PT + ((tid + c) & (PT - 1))];
      s = fmaf(v, v, s);
    }
    s_den[tid] = fmaxf(sqrtf(s), 1e-12f);
  }
  __syncthreads();
  if (pv) {
    const float den = s_den[px];
    float* ob = out + (size_t)b * D * P + p0 + px;
    int c = cgrp;
    for (; c + 3 * CG < D; c += 4 * CG) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = scale * (tile[(c + u * CG) * PT + ((px + c + u * CG) & (PT - 1))] / den);
#pragma unroll
      for (int u = 0; u < 4; ++u) ob[(size_t)(c + u * CG) * P] = v[u];
    }
    for (; c < D; c += CG) ob[(size_t)c * P] = scale * (tile[c * PT + ((px + c) & (PT - 1))] / den);
  }
  if (raw_cl) {
    float* cb = raw_cl + ((size_t)b * P + p0) * D;
    const int wave = tid >> 6, lane = tid & 63;
    const int npx = min(PT, P - p0);
    for (int pixel = wave; pixel < npx; pixel += 4) {
      float* cp = cb + (size_t)pixel * D;
      for (int c = lane; c < D; c += 64) cp[c] = tile[c * PT + ((pixel + c) & (PT - 1))];
    }
  }
}

// thread per pixel, channels walked sequentially (fmaf chain c = 0..D-1)
__global__ void normalize_map_kernel(const float* raw, int B, int D, int P, float scale, float* out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)B * P) return;
  const int b = (int)(gid / P), p = (int)(gid % P);
  const float* r = raw + (size_t)b * D * P + p;
  float s = 0.0f;
  for (int c = 0; c < D; ++c) {
    const float v = r[(size_t)c * P];
    s = fmaf(v, v, s);
  }
  const float den = fmaxf(sqrtf(s), 1e-12f);
  float* o = out + (size_t)b * D * P + p;
  for (int c = 0; c < D; ++c) o[(size_t)c * P] = scale * (r[(size_t)c * P] / den);
}

// upsample_descriptors + normalize, written cropped.  One wave per (64 output columns, band of output
// rows that interpolate between the same two coarse rows y0, y1, image): the horizontal lerps
//   h0 = hx*p[y0][x0] + lx*p[y0][x1],  h1 = hx*p[y1][x0] + lx*p[y1][x1]
// depend on the column only, so they are computed once per channel and shared by every row of the band
// (v = hy*h0 + ly*h1: the same operations, in the same order, as the per-pixel formula of the oracle /
// ATen upsample_bilinear2d), which cuts the coarse-map loads 8x and the flops 2x at the usual 1/8 scale;
// each row of a band is stored as one 256-byte segment per channel.  Pass 1 accumulates the per-pixel
// squared norm as the sequential fmaf chain c = 0..D-1, pass 2 recomputes and writes scale * v / norm.
constexpr int UP_COLS = 384;  // columns per workgroup: whole 346-pixel rows, so a band is one contiguous run per channel
constexpr int UP_ROWS = 8;               // rows per sweep: a band of the usual 1/8 scale (the first band takes two sweeps)
  // 

__global__ __launch_bounds__(UP_COLS) void upsample_band_kernel(const float* raw, int D, int hc, int wc, int Hp, int Wp, int h0, int w0, int H,
                                                           int W, float scale, float* out) {
  extern __shared__ float stage[];  // 2 x UP_ROWS x W floats when the workgroup spans whole rows
  const int lane = threadIdx.x;
  const int x = blockIdx.x * UP_COLS + lane;
  const bool xv = x < W;
  const int j = blockIdx.y, b = blockIdx.z;
  const float sy = (float)hc / (float)Hp, sx = (float)wc / (float)Wp;
  auto coarse_row = [&](int Y, float& ly) {  // y0 and the vertical weight of padded-frame row Y
    float fy = ((float)Y + 0.5f) * sy - 0.5f;
    if (fy < 0.0f) fy = 0.0f;
    const int y0 = (int)fy;
    ly = fy - (float)y0;
    return y0;
  };
  // first cropped row whose y0 is j (rows are monotone in y0); start a little below the estimate
  int Y = j == 0 ? h0 : (int)(((float)j + 0.5f) / sy - 0.5f) - 2;  // band 0 also owns the rows whose source row clamps to 0
  if (Y < h0) Y = h0;
  float ly_tmp;
  while (Y < h0 + H && coarse_row(Y, ly_tmp) < j) ++Y;
  const int y1 = j + (j < hc - 1 ? 1 : 0);
  float fx = ((float)((xv ? x : W - 1) + w0) + 0.5f) * sx - 0.5f;
  if (fx < 0.0f) fx = 0.0f;
  const int x0 = (int)fx, x1 = x0 + (x0 < wc - 1 ? 1 : 0);
  const float lx = fx - (float)x0, hx = 1.0f - lx;
  const size_t plane = (size_t)hc * wc, HW = (size_t)H * W;
  const float* rb = raw + (size_t)b * D * plane;
  const unsigned o00 = (unsigned)(j * wc + x0), o01 = (unsigned)(j * wc + x1), o10 = (unsigned)(y1 * wc + x0), o11 = (unsigned)(y1 * wc + x1);
  while (Y < h0 + H && coarse_row(Y, ly_tmp) == j) {  // sweeps of up to UP_ROWS rows of this band
    int nrow = 0;
    float ly[UP_ROWS], hy[UP_ROWS];
#pragma unroll
    for (int r = 0; r < UP_ROWS; ++r) {
      float l = 0.0f;
      const bool in = Y + r < h0 + H && coarse_row(Y + r, l) == j && nrow == r;
      if (in) nrow = r + 1;
      ly[r] = l;
      hy[r] = 1.0f - l;
    }
    float ssq[UP_ROWS];
#pragma unroll
    for (int r = 0; r < UP_ROWS; ++r) ssq[r] = 0.0f;
    // the four taps of channel c+1 are requested before channel c is consumed (hipcc keeps the loop rolled
    // and would otherwise wait out a full L2 round trip per channel)
    float a00 = rb[o00], a01 = rb[o01], a10 = rb[o10], a11 = rb[o11];
    for (int c = 0; c < D; ++c) {
      const float* pn = rb + (size_t)(c + 1 < D ? c + 1 : c) * plane;
      const float n00 = pn[o00], n01 = pn[o01], n10 = pn[o10], n11 = pn[o11];
      const float t0 = hx * a00 + lx * a01;
      const float t1 = hx * a10 + lx * a11;
#pragma unroll
      for (int r = 0; r < UP_ROWS; ++r) {
        const float v = hy[r] * t0 + ly[r] * t1;
        ssq[r] = fmaf(v, v, ssq[r]);
      }
      a00 = n00;
      a01 = n01;
      a10 = n10;
      a11 = n11;
    }
    float den[UP_ROWS];
#pragma unroll
    for (int r = 0; r < UP_ROWS; ++r) den[r] = fmaxf(sqrtf(ssq[r]), 1e-12f);
    // Stores.  When the workgroup spans whole rows, the nrow rows of a channel are one contiguous run of the
    // output: it is staged in LDS (two buffers, one barrier per channel) and written linearly with 16-byte
    // stores aligned to 16 bytes.  Row pitches like 346 floats put every per-row 256-byte segment across
    // cache-line boundaries, which holds direct row stores at ~2.2 TB/s (tools/store_pattern.hip: 2.1-2.6 TB/s
    // against 3.5-4.0 for aligned runs and 5.5 for a 128-byte-aligned pitch).
    const bool staged = gridDim.x == 1;
    float* ob = out + (size_t)b * D * HW + (size_t)(Y - h0) * W;
    const int len = nrow * W;
    a00 = rb[o00];
    a01 = rb[o01];
    a10 = rb[o10];
    a11 = rb[o11];
    for (int c = 0; c < D; ++c) {
      const float* pn = rb + (size_t)(c + 1 < D ? c + 1 : c) * plane;
      const float n00 = pn[o00], n01 = pn[o01], n10 = pn[o10], n11 = pn[o11];
      const float t0 = hx * a00 + lx * a01;
      const float t1 = hx * a10 + lx * a11;
      a00 = n00;
      a01 = n01;
      a10 = n10;
      a11 = n11;
      float* oc = ob + (size_t)c * HW;
      if (staged) {
        float* buf = stage + (c & 1) * (UP_ROWS * W);
#pragma unroll
        for (int r = 0; r < UP_ROWS; ++r) {
          const float v = hy[r] * t0 + ly[r] * t1;
          if (xv && r < nrow) buf[r * W + x] = scale * (v / den[r]);
        }
        __syncthreads();
        const int head = (int)(((16 - ((size_t)oc & 15)) & 15) >> 2);  // floats up to the first 16-byte boundary
        if (lane < head && lane < len) oc[lane] = buf[lane];
        const int n4 = len > head ? (len - head) >> 2 : 0;
        f32x4* o4 = reinterpret_cast<f32x4*>(oc + head);
        for (int i = lane; i < n4; i += UP_COLS) {
          const float* sb = buf + head + 4 * i;
          o4[i] = f32x4{sb[0], sb[1], sb[2], sb[3]};
        }
        const int tail0 = head + 4 * n4;
        if (tail0 + lane < len) oc[tail0 + lane] = buf[tail0 + lane];
      } else {
#pragma unroll
        for (int r = 0; r < UP_ROWS; ++r) {
          const float v = hy[r] * t0 + ly[r] * t1;
          if (xv && r < nrow) oc[(size_t)r * W + x] = scale * (v / den[r]);
        }
      }
    }
    Y += nrow;
  }
}


// ---- dense descriptor maps, round 3: two kernels instead of the band kernel above for W <= 384, wc <= 63 ---------------
// The output [B,D,H,W] is a pure 2.95 GB stream at the shipped size (B=32); tools/store_pattern2.hip measured what the
// store schedule alone is worth on MI355X: 2.3-2.7 TB/s for per-row dword stores at the 346-float pitch (every 256-byte
// segment straddles cache lines), 5.4 TB/s for "one wave writes the whole ROWS x W run of one (channel, band) with 16-byte
// stores, 16-32 channels per workgroup, channel group fastest in the grid", 6.9 for a plain sequential fill.
//   upsample_den_kernel    per output pixel: den = max(sqrt(sum_c v_c^2), 1e-12) with the sequential fmaf chain c = 0..D-1
//                          (the only part that needs all channels of a pixel), and 1/den; [2,B,H,W] floats of workspace
//   upsample_store_kernel  (channel group, sweep, image) per workgroup, channels in parallel over its four waves: each wave
//                          recomputes v for one channel of the sweep (horizontal lerps shared by the rows, as above), divides
//                          by den, stages the run in its private LDS slab and writes it linearly; no workgroup barrier
//                          after the coarse rows are staged, stores drain under the next channel's arithmetic.
// A sweep = up to UP_ROWS consecutive output rows that interpolate between the same two coarse rows (j, y1): unit
// (j, s) = rows [first row of band j + s UP_ROWS, + UP_ROWS) of band j; units 0..hc-1 are the first sweeps, the few further
// sweeps of taller bands are listed by the host (UpGeom::extra_*), so every workgroup has work.  Both kernels keep the per-element operation order of orc_upsample_normalize (bit-equal).
constexpr int UPS_CC = 32;  // channels per workgroup of the store kernel: 64 (channel, row) pairs = 16 per wave
constexpr int UPS_WAVES = 4;
constexpr int UP_PAIRS = 16;  // (channel, coarse row) pairs a wave of the store kernel stages (once)
constexpr int UPD_PAIRS = 8;  // ... and a wave of the den kernel per round (registers: five of its workgroups per CU)

constexpr int UP_MAX_EXTRA = 8;
struct UpGeom {
  int D, hc, wc, Hp, Wp, h0, w0, H, W;
  int units;  // hc first sweeps + n_extra further sweeps of bands taller than UP_ROWS (band 0 at the shipped size)
  int n_extra;
  short extra_j[UP_MAX_EXTRA], extra_s[UP_MAX_EXTRA];
};

__host__ __device__ inline int up_coarse_row(const UpGeom& g, int Y, float& ly) {  // y0 and the vertical weight of padded row Y
  const float sy = (float)g.hc / (float)g.Hp;
  float fy = ((float)Y + 0.5f) * sy - 0.5f;
  if (fy < 0.0f) fy = 0.0f;
  const int y0 = (int)fy;
  ly = fy - (float)y0;
  return y0;
}
// first padded row (>= h0) whose source row is j; rows are monotone in y0, band 0 also owns the rows clamped to source row 0
__device__ __forceinline__ int up_band_start(const UpGeom& g, int j) {
  const float sy = (float)g.hc / (float)g.Hp;
  int Y = j == 0 ? g.h0 : (int)(((float)j + 0.5f) / sy - 0.5f) - 2;
  if (Y < g.h0) Y = g.h0;
  float t;
  while (Y < g.h0 + g.H && up_coarse_row(g, Y, t) < j) ++Y;
  return Y;
}
// rows of sweep s of band j: first row Y, count nrow (0: nothing to do), vertical weights
__device__ __forceinline__ int up_sweep(const UpGeom& g, int j, int s, int& Y, float* ly, float* hy) {
  Y = up_band_start(g, j) + s * UP_ROWS;
  int nrow = 0;
#pragma unroll
  for (int r = 0; r < UP_ROWS; ++r) {
    float l = 0.0f;
    const bool in = Y + r < g.h0 + g.H && up_coarse_row(g, Y + r, l) == j && nrow == r;
    if (in) nrow = r + 1;
    ly[r] = l;
    hy[r] = 1.0f - l;
  }
  return nrow;
}
__device__ __forceinline__ void up_unit(const UpGeom& g, int unit, int& j, int& s) {
  j = unit;
  s = 0;
  if (unit >= g.hc) {
#pragma unroll
    for (int e = 0; e < UP_MAX_EXTRA; ++e)  // constant indices: the arrays stay in SGPRs
      if (e == unit - g.hc) {
        j = g.extra_j[e];
        s = g.extra_s[e];
      }
  }
}
__device__ __forceinline__ void up_column(const UpGeom& g, int x, int& x0, float& lx, float& hx) {
  const float sx = (float)g.wc / (float)g.Wp;
  float fx = ((float)(x + g.w0) + 0.5f) * sx - 0.5f;
  if (fx < 0.0f) fx = 0.0f;
  x0 = (int)fx;
  lx = fx - (float)x0;
  hx = 1.0f - lx;
}
// Coarse rows j and y1 of channels [c0, c0 + n) -> LDS [n][2][wc + 1] (wc + 1 <= 64); element wc repeats element wc - 1, so
// that the right neighbour x1 = min(x0 + 1, wc - 1) is always the word after x0.  (channel, row) pairs are dealt to the
// waves (at most UP_PAIRS each), lane = x: coalesced row reads, no index divisions, all of a wave's loads in flight at once;
// split in issue / commit so that a round's loads can fly under the previous round's arithmetic.
template <int PAIRS>
struct UpStage {
  float v[PAIRS];
};
template <int PAIRS>
__device__ __forceinline__ void up_stage_issue(UpStage<PAIRS>& st, const float* rb, const UpGeom& g, int j, int y1, int c0, int n, int lane, int wv, int nw) {
  const unsigned plane = (unsigned)(g.hc * g.wc);
  const int np = 2 * n;
  // 32-bit element offsets from one uniform base: the loads address as scalar base + vector offset
  const float* base = rb + (size_t)c0 * plane + (size_t)j * g.wc;
  const unsigned lo = (unsigned)(lane < g.wc ? lane : g.wc - 1);
  const unsigned d1 = (unsigned)((y1 - j) * g.wc);
#pragma unroll
  for (int u = 0; u < PAIRS; ++u) {
    const int q = wv + u * nw;
    const int qc = q < np ? q : np - 1;
    st.v[u] = base[(unsigned)(qc >> 1) * plane + ((qc & 1) ? d1 : 0u) + lo];
  }
}
template <int PAIRS>
__device__ __forceinline__ void up_stage_commit(const UpStage<PAIRS>& st, const UpGeom& g, int n, float* rows, int lane, int wv, int nw) {
  const int pw = g.wc + 1;
  if (lane < pw) {
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) {
      const int q = wv + u * nw;
      if (q < 2 * n) rows[q * pw + lane] = st.v[u];
    }
  }
}

// ws layout: den [B,H,W] then 1/den [B,H,W] (the reciprocal correctly rounded: IEEE division)
// NW waves per workgroup; blockIdx.z walks the column blocks of 64 NW (two half-width workgroups per sweep at W = 346:
// twice as many, half as long workgroups load the CUs more evenly than 34 x B whole-row ones)
template <int NW>
__global__ __launch_bounds__(64 * NW, 8) void upsample_den_kernel(const float* raw, UpGeom g, float* den, float* rden) {
  constexpr int NIT = NW;
  constexpr int CHR = UPD_PAIRS * NIT / 2;  // channels per LDS round: every wave stages UPD_PAIRS (channel, row) pairs
  extern __shared__ float rows[];          // [CHR][2][wc + 1]
  const int b = blockIdx.y;
  int j, s;
  up_unit(g, blockIdx.x, j, s);
  float ly[UP_ROWS], hy[UP_ROWS], ssq[UP_ROWS];
  int Y;
  const int nrow = up_sweep(g, j, s, Y, ly, hy);
  if (nrow == 0) return;  // uniform (a band without rows in the crop window)
  const int tid = threadIdx.x, x = (int)blockIdx.z * 64 * NW + tid, lane = tid & 63, wv = tid >> 6;
  const bool xv = x < g.W;
  const int pw = g.wc + 1;
  int x0;
  float lx, hx;
  up_column(g, xv ? x : g.W - 1, x0, lx, hx);
  const int y1 = j + (j < g.hc - 1 ? 1 : 0);
  const float* rb = raw + (size_t)b * g.D * g.hc * g.wc;
#pragma unroll
  for (int r = 0; r < UP_ROWS; ++r) ssq[r] = 0.0f;
  for (int c0 = 0; c0 < g.D; c0 += CHR) {
    const int n = g.D - c0 < CHR ? g.D - c0 : CHR;
    // the staging registers are not kept across the arithmetic: with <= 64 registers five of these workgroups share a CU and
    // hide each other's load latency (a register prefetch across the loop cost two of them: 217 -> 307 us at B=32)
    UpStage<UPD_PAIRS> st;
    up_stage_issue(st, rb, g, j, y1, c0, n, lane, wv, NIT);
    __syncthreads();  // the previous round's reads are done
    up_stage_commit(st, g, n, rows, lane, wv, NIT);
    __syncthreads();
    const float* rp = rows + x0;
#pragma unroll 1  // (2 / 4 measured the same)
    for (int cl = 0; cl < n; ++cl, rp += 2 * pw) {
      const float t0 = hx * rp[0] + lx * rp[1];
      const float t1 = hx * rp[pw] + lx * rp[pw + 1];
#pragma unroll
      for (int r = 0; r < UP_ROWS; ++r) {
        const float v = hy[r] * t0 + ly[r] * t1;
        ssq[r] = fmaf(v, v, ssq[r]);
      }
    }
  }
  if (xv) {
    const size_t o = ((size_t)b * g.H + (Y - g.h0)) * g.W + x;
#pragma unroll
    for (int r = 0; r < UP_ROWS; ++r)
      if (r < nrow) {
        const float d = fmaxf(sqrtf(ssq[r]), 1e-12f);
        den[o + (size_t)r * g.W] = d;
        rden[o + (size_t)r * g.W] = 1.0f / d;
      }
  }
}

// v / d for the store kernel.  d >= 1e-12 and y = RN(1 / d) come from the den kernel.  Two Newton corrections of
// q = v * y with exact fma remainders give the correctly rounded quotient (after the first step q is faithful, then
// Markstein's theorem applies) as long as nothing under- or overflows and the quotient is a normal number: the kernel
// takes this path for 2^-80 <= |v| <= d < 2^20 only (|q| >= 2^-100).  Checked against IEEE division on 2.5e9 random and boundary-mantissa operands (tools/div_check.c).  Elements
// outside that range (exact zeros among them) make the wave redo the channel with IEEE divisions.
__device__ __forceinline__ float up_div(float v, float d, float y) {
  const float q0 = v * y;
  const float r0 = fmaf(-d, q0, v);
  const float q1 = fmaf(r0, y, q0);
  const float r1 = fmaf(-d, q1, v);
  return fmaf(r1, y, q1);
}

template <int NIT>
__global__ __launch_bounds__(64 * UPS_WAVES) void upsample_store_kernel(const float* raw, const float* den, const float* rden, UpGeom g,
                                                                         float scale, float* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pw = g.wc + 1;
  const int ng = (g.D + UPS_CC - 1) / UPS_CC;
  int id = (int)blockIdx.x;
  const int cg = id % ng;  // channel group fastest: neighbouring workgroups write the same rows of neighbouring channel groups
  id /= ng;
  const int unit = id % g.units, b = id / g.units;
  int j, s;
  up_unit(g, unit, j, s);
  float ly[UP_ROWS], hy[UP_ROWS];
  int Y;
  const int nrow = up_sweep(g, j, s, Y, ly, hy);
  if (nrow == 0) return;  // uniform (a band without rows in the crop window)
  const int c0 = cg * UPS_CC;
  const int nc = g.D - c0 < UPS_CC ? g.D - c0 : UPS_CC;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int run_cap = UP_ROWS * 64 * NIT + 4;  // slab per wave: UP_ROWS rows of W floats (row-major like the output) + slack
  float* rows = smem + UPS_WAVES * run_cap;        // [UPS_CC][2][wc + 1]
  float* slab = smem + wave * run_cap;
  const int y1 = j + (j < g.hc - 1 ? 1 : 0);
  const float* rb = raw + (size_t)b * g.D * g.hc * g.wc;
  UpStage<UP_PAIRS> st;
  up_stage_issue(st, rb, g, j, y1, c0, nc, lane, wave, UPS_WAVES);

  int x0[NIT];
  float lx[NIT], hx[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int x = it * 64 + lane;
    up_column(g, x < g.W ? x : g.W - 1, x0[it], lx[it], hx[it]);
  }
  const bool xvl = (NIT - 1) * 64 + lane < g.W;  // only the last column sweep has lanes outside the row
  const size_t HW = (size_t)g.H * g.W;
  // per-pixel norms of this sweep; rows past nrow / columns past W re-read a valid pixel's norm (clamped address, no
  // branches) and are never copied out
  float dn[NIT][UP_ROWS], yr[NIT][UP_ROWS];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int r = 0; r < UP_ROWS; ++r) {
      const int xc = (it < NIT - 1 || xvl) ? it * 64 + lane : g.W - 1;
      const size_t o = ((size_t)b * g.H + (Y - g.h0 + (r < nrow ? r : nrow - 1))) * g.W + xc;
      dn[it][r] = den[o];
      yr[it][r] = rden[o];
    }
  bool big = false;  // a norm beyond the fast division's range: IEEE divisions for the whole sweep (otherwise |v| <= den < 2^20)
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int r = 0; r < UP_ROWS; ++r) big |= !(dn[it][r] < 0x1p20f);
  big = __any(big);
  up_stage_commit(st, g, nc, rows, lane, wave, UPS_WAVES);
  __syncthreads();  // the only workgroup barrier: from here on every wave works on its own channels and its own LDS slab

  const int len = nrow * g.W;
  for (int cl = wave; cl < nc; cl += UPS_WAVES) {
    float* oc = out + ((size_t)b * g.D + c0 + cl) * HW + (size_t)(Y - g.h0) * g.W;
    const int head = (int)(((16 - ((size_t)oc & 15)) & 15) >> 2);  // floats up to the first 16-byte boundary of the run
    float* buf = slab + ((4 - head) & 3);                           // ... which then sits on a 16-byte boundary of the slab too
    const float* rp = rows + cl * 2 * pw;
    bool odd = big;  // some element outside the fast division's range
    float vmin = 0x1p20f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const float t0 = hx[it] * rp[x0[it]] + lx[it] * rp[x0[it] + 1];
      const float t1 = hx[it] * rp[pw + x0[it]] + lx[it] * rp[pw + x0[it] + 1];
      float q[UP_ROWS];
#pragma unroll
      for (int r = 0; r < UP_ROWS; ++r) {
        const float v = hy[r] * t0 + ly[r] * t1;
        vmin = fminf(vmin, fabsf(v));  // lower end of the fast division's range, tested once per channel; the upper end once per sweep on the norms (|v| <= den < 2^20; a NaN element has a NaN norm)
        q[r] = scale * up_div(v, dn[it][r], yr[it][r]);
      }
      if (it < NIT - 1 || xvl) {
#pragma unroll
        for (int r = 0; r < UP_ROWS; ++r) buf[r * g.W + it * 64 + lane] = q[r];  // rows past nrow land in the slab's slack
      }
    }
    odd |= !(vmin >= 0x1p-80f);
    if (__builtin_expect(__any(odd), 0)) {  // the same values with IEEE divisions
#pragma unroll
      for (int it = 0; it < NIT; ++it) {  // unrolled: a rolled loop would index the per-sweep register arrays dynamically (scratch)
        const float t0 = hx[it] * rp[x0[it]] + lx[it] * rp[x0[it] + 1];
        const float t1 = hx[it] * rp[pw + x0[it]] + lx[it] * rp[pw + x0[it] + 1];
        if (it < NIT - 1 || xvl) {
#pragma unroll
          for (int r = 0; r < UP_ROWS; ++r) buf[r * g.W + it * 64 + lane] = scale * ((hy[r] * t0 + ly[r] * t1) / dn[it][r]);
        }
      }
    }
    // the wave's own LDS writes -> its own LDS reads: the LDS executes one wave's instructions in order; the fences only
    // keep the compiler from moving the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < head && lane < len) oc[lane] = buf[lane];
    const int n4 = len > head ? (len - head) >> 2 : 0;
    f32x4* o4 = reinterpret_cast<f32x4*>(oc + head);
    f32x4 w4[2 * NIT];
#pragma unroll
    for (int k = 0; k < 2 * NIT; ++k) {  // UP_ROWS x 64 NIT floats = 2 NIT float4 per lane at most
      const int i = lane + 64 * k;
      const float* sb = buf + head + 4 * (i < n4 ? i : 0);
      w4[k] = f32x4{sb[0], sb[1], sb[2], sb[3]};  // one ds_read_b128: buf + head is 16-byte aligned
    }
#pragma unroll
    for (int k = 0; k < 2 * NIT; ++k) {
      const int i = lane + 64 * k;
      if (i < n4) o4[i] = w4[k];
    }
    const int tail0 = head + 4 * n4;
    if (tail0 + lane < len) oc[tail0 + lane] = buf[tail0 + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

}  // namespace

EINX_EXPORT int einx_desc_sample(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, int bilinear, int channels_last,
                                 const int32_t* indices, const int32_t* counts, int cap, float scale, float* out, void* stream) {
  EINX_CHECK_ARG(raw && indices && counts && out, "null pointer");
  EINX_CHECK_ARG(B > 0 && D > 0 && D <= 512 && hc > 0 && wc > 0 && cap > 0, "bad shape (D must be <= 512)");
  EINX_CHECK_ARG(bilinear || (Hp == hc && Wp == wc), "gather mode needs a full-resolution map");
  EINX_CHECK_ARG(!channels_last || bilinear, "the channels-last layout is implemented for bilinear sampling");
  EinxWatch none;
  none.table = nullptr;
  none.ref = nullptr;
  none.hash = nullptr;
  none.flag = nullptr;
  none.n = 0;
  none.bit = 0;
  EinxCrop nocrop{};
  nocrop.map = nullptr;
  return einx_desc_sample_watch(raw, B, D, hc, wc, Hp, Wp, bilinear, channels_last, indices, counts, cap, scale, out, none, nocrop, stream);
}

int einx_desc_sample_watch(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, int bilinear, int channels_last, const int32_t* indices,
                           const int32_t* counts, int cap, float scale, float* out, const EinxWatch& watch, const EinxCrop& crop, void* stream) {
  const int main_blocks = einx_cdiv(cap, 4);
  const int watch_blocks = watch.n > 0 ? einx_cdiv(watch.n, 4) : 0;
  const int crop_blocks = crop.map ? einx_cdiv(crop.H * crop.W, 256) : 0;
  dim3 grid((unsigned)(main_blocks + watch_blocks + crop_blocks), (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
  EINX_PROF("desc_sample_kernel", s);
  if (bilinear && channels_last)
    hipLaunchKernelGGL((desc_sample_kernel<true, true>), grid, dim3(256), 0, s, raw, D, hc, wc, Hp, Wp, indices, counts, cap, scale, out, watch, main_blocks, crop, watch_blocks);
  else if (bilinear)
    hipLaunchKernelGGL((desc_sample_kernel<true, false>), grid, dim3(256), 0, s, raw, D, hc, wc, Hp, Wp, indices, counts, cap, scale, out, watch, main_blocks, crop, watch_blocks);
  else
    hipLaunchKernelGGL((desc_sample_kernel<false, false>), grid, dim3(256), 0, s, raw, D, hc, wc, Hp, Wp, indices, counts, cap, scale, out, watch, main_blocks, crop, watch_blocks);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_normalize_map(const float* raw, int B, int D, int P, float scale, float* out, float* raw_cl, void* stream) {
  EINX_CHECK_ARG(raw && out, "null pointer");
  EINX_CHECK_ARG(B > 0 && D > 0 && P > 0, "bad shape");
  hipStream_t s = (hipStream_t)stream;
  EINX_PROF("normalize_map", s);
  if (D <= 512) {  // 32-pixel tiles: 128-byte rows per channel, D * 128 bytes of LDS (4-5 workgroups per CU at D = 256)
    hipLaunchKernelGGL(normalize_map_tile_kernel<32>, dim3((unsigned)einx_cdiv(P, 32), (unsigned)B), dim3(256), (size_t)D * 32 * sizeof(float), s,
                       raw, D, P, scale, out, raw_cl);
  } else {
    EINX_CHECK_ARG(raw_cl == nullptr, "channels-last copy needs D <= 512");
    const size_t n = (size_t)B * P;
    hipLaunchKernelGGL(normalize_map_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw, B, D, P, scale, out);
  }
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_normalize_rows(const float* x, int R, int C, float scale, float* out, void* stream) {
  EINX_CHECK_ARG(x && out, "null pointer");
  EINX_CHECK_ARG(R > 0 && C > 0, "bad shape");
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)einx_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, R, C, scale, out);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_random_positions(const float* u, int R, float size0, float size1, float* out, void* stream) {
  EINX_CHECK_ARG(u && out, "null pointer");
  EINX_CHECK_ARG(R > 0, "bad shape");
  hipLaunchKernelGGL(random_positions_kernel, dim3((unsigned)einx_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, u, R, size0, size1, out);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

// lists the sweeps beyond the first of every band (the same float arithmetic as the kernels: IEEE, no contraction);
// false if there are more than UP_MAX_EXTRA of them (unusual scales: the band kernel handles those)
static bool up_plan_units(UpGeom& g) {
  g.n_extra = 0;
  for (int e = 0; e < UP_MAX_EXTRA; ++e) g.extra_j[e] = g.extra_s[e] = 0;
  int run = 0, prev = -1;
  for (int Y = g.h0; Y < g.h0 + g.H; ++Y) {
    float l;
    const int y0 = up_coarse_row(g, Y, l);
    run = y0 == prev ? run + 1 : 1;
    prev = y0;
    if (run > UP_ROWS && (run - 1) % UP_ROWS == 0) {  // row number UP_ROWS k + 1 of this band opens sweep k
      if (g.n_extra == UP_MAX_EXTRA) return false;
      g.extra_j[g.n_extra] = (short)y0;
      g.extra_s[g.n_extra] = (short)((run - 1) / UP_ROWS);
      ++g.n_extra;
    }
  }
  g.units = g.hc + g.n_extra;
  return true;
}

EINX_EXPORT size_t einx_upsample_ws_bytes(int B, int H, int W) { return (size_t)2 * B * H * W * sizeof(float); }

template <int NIT>
static void launch_upsample(const float* raw, int B, const UpGeom& g, float scale, float* out, float* den, hipStream_t s) {
  float* rden = den + (size_t)B * g.H * g.W;
  const int pw = g.wc + 1;
  const int units = g.units;
  {
    EINX_PROF("upsample_den_kernel", s);
    constexpr int NWD = NIT >= 4 ? (NIT + 1) / 2 : NIT;  // waves per den workgroup
    hipLaunchKernelGGL(upsample_den_kernel<NWD>, dim3((unsigned)units, (unsigned)B, (unsigned)einx_cdiv(g.W, 64 * NWD)), dim3(64 * NWD),
                       (size_t)(UPD_PAIRS * NWD / 2) * 2 * pw * sizeof(float), s, raw, g, den, rden);
  }
  {
    EINX_PROF("upsample_store_kernel", s);
    const int ng = einx_cdiv(g.D, UPS_CC);
    const size_t lds = ((size_t)UPS_WAVES * (UP_ROWS * 64 * NIT + 4) + (size_t)UPS_CC * 2 * pw) * sizeof(float);
    hipLaunchKernelGGL(upsample_store_kernel<NIT>, dim3((unsigned)(ng * units * B)), dim3(64 * UPS_WAVES), lds, s, raw, den, rden, g, scale, out);
  }
}

EINX_EXPORT int einx_upsample_normalize(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, int h0, int w0, int H, int W,
                                        float scale, float* out, void* ws, size_t ws_bytes, void* stream) {
  EINX_CHECK_ARG(raw && out, "null pointer");
  EINX_CHECK_ARG(B > 0 && D > 0 && hc > 0 && wc > 0 && H > 0 && W > 0, "bad shape");
  EINX_CHECK_ARG(h0 >= 0 && w0 >= 0 && h0 + H <= Hp && w0 + W <= Wp, "crop window outside the padded map");
  EINX_CHECK_ARG(B < 65536, "batch too large");
  hipStream_t s = (hipStream_t)stream;
  UpGeom g{D, hc, wc, Hp, Wp, h0, w0, H, W, 0, 0, {0}, {0}};
  // dynamic LDS of the store kernel (four wave slabs + the staged coarse rows) must stay within the 64 KB a launch gets
  // without hipFuncSetAttribute
  const size_t store_lds = ((size_t)UPS_WAVES * (UP_ROWS * 64 * einx_cdiv(W, 64) + 4) + (size_t)UPS_CC * 2 * (wc + 1)) * sizeof(float);
  if (W <= 384 && wc <= 63 && hc < 32768 && store_lds <= 65536 && up_plan_units(g)) {  // the two-kernel path (the shipped geometries); others take the band kernel
    EINX_CHECK_ARG(ws && ws_bytes >= einx_upsample_ws_bytes(B, H, W), "workspace missing or smaller than einx_upsample_ws_bytes");
    EINX_CHECK_ARG((long)einx_cdiv(D, UPS_CC) * g.units * B < (1L << 31), "grid too large");
    float* den = (float*)ws;
    switch (einx_cdiv(W, 64)) {
      case 1: launch_upsample<1>(raw, B, g, scale, out, den, s); break;
      case 2: launch_upsample<2>(raw, B, g, scale, out, den, s); break;
      case 3: launch_upsample<3>(raw, B, g, scale, out, den, s); break;
      case 4: launch_upsample<4>(raw, B, g, scale, out, den, s); break;
      case 5: launch_upsample<5>(raw, B, g, scale, out, den, s); break;
      default: launch_upsample<6>(raw, B, g, scale, out, den, s); break;
    }
    EINX_CHECK_LAUNCH();
    return EINX_OK;
  }
  EINX_PROF("upsample_band_kernel", s);
  const size_t lds = W <= UP_COLS ? (size_t)2 * UP_ROWS * W * sizeof(float) : 0;
  hipLaunchKernelGGL(upsample_band_kernel, dim3((unsigned)einx_cdiv(W, UP_COLS), (unsigned)hc, (unsigned)B), dim3(UP_COLS), lds, s, raw, D,
                     hc, wc, Hp, Wp, h0, w0, H, W, scale, out);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
