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
                                                          float* out) {
  const int b = blockIdx.y;
  const int kp = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
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

// normalize_descriptors through LDS: a workgroup stages PT pixels x D channels with fully coalesced,
// deeply pipelined loads (the thread-per-pixel form below issues D dependent strided loads from far
// too few threads), one lane per pixel then walks the channels as the same sequential fmaf chain
// c = 0..D-1, and all threads write the normalised map -- plus, optionally, the channels-last copy
// of the raw map that desc_sample_kernel<.., CL> gathers from.  Element (c, pixel) lives at
// lds[c*PT + ((pixel + c) & (PT-1))]: conflict-free along pixels and along channels.
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
  for (int c = cgrp; c < D; c += CG) tile[c * PT + ((px + c) & (PT - 1))] = pv ? rb[(size_t)c * P + p0 + px] : 0.0f;
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
    for (int c = cgrp; c < D; c += CG) ob[(size_t)c * P] = scale * (tile[c * PT + ((px + c) & (PT - 1))] / den);
  }
  if (raw_cl) {
    float* cb = raw_cl + ((size_t)b * P + p0) * D;
    for (int e = tid; e < PT * D; e += 256) {
      const int pixel = e / D, c = e % D;
      if (p0 + pixel < P) cb[(size_t)pixel * D + c] = tile[c * PT + ((pixel + c) & (PT - 1))];
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
constexpr int UP_ROWS = 8;    // rows per sweep: a band of the usual 1/8 scale (the first band takes two sweeps)
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

}  // namespace

EINX_EXPORT int einx_desc_sample(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, int bilinear, int channels_last,
                                 const int32_t* indices, const int32_t* counts, int cap, float scale, float* out, void* stream) {
  EINX_CHECK_ARG(raw && indices && counts && out, "null pointer");
  EINX_CHECK_ARG(B > 0 && D > 0 && D <= 512 && hc > 0 && wc > 0 && cap > 0, "bad shape (D must be <= 512)");
  EINX_CHECK_ARG(bilinear || (Hp == hc && Wp == wc), "gather mode needs a full-resolution map");
  EINX_CHECK_ARG(!channels_last || bilinear, "the channels-last layout is implemented for bilinear sampling");
  dim3 grid((unsigned)einx_cdiv(cap, 4), (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
  EINX_PROF("desc_sample_kernel", s);
  if (bilinear && channels_last)
    hipLaunchKernelGGL((desc_sample_kernel<true, true>), grid, dim3(256), 0, s, raw, D, hc, wc, Hp, Wp, indices, counts, cap, scale, out);
  else if (bilinear)
    hipLaunchKernelGGL((desc_sample_kernel<true, false>), grid, dim3(256), 0, s, raw, D, hc, wc, Hp, Wp, indices, counts, cap, scale, out);
  else
    hipLaunchKernelGGL((desc_sample_kernel<false, false>), grid, dim3(256), 0, s, raw, D, hc, wc, Hp, Wp, indices, counts, cap, scale, out);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_normalize_map(const float* raw, int B, int D, int P, float scale, float* out, float* raw_cl, void* stream) {
  EINX_CHECK_ARG(raw && out, "null pointer");
  EINX_CHECK_ARG(B > 0 && D > 0 && P > 0, "bad shape");
  hipStream_t s = (hipStream_t)stream;
  EINX_PROF("normalize_map", s);
  if (D <= 256) {
    hipLaunchKernelGGL(normalize_map_tile_kernel<64>, dim3((unsigned)einx_cdiv(P, 64), (unsigned)B), dim3(256), (size_t)D * 64 * sizeof(float), s,
                       raw, D, P, scale, out, raw_cl);
  } else if (D <= 512) {
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

EINX_EXPORT int einx_upsample_normalize(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, int h0, int w0, int H, int W,
                                        float scale, float* out, void* stream) {
  EINX_CHECK_ARG(raw && out, "null pointer");
  EINX_CHECK_ARG(B > 0 && D > 0 && hc > 0 && wc > 0 && H > 0 && W > 0, "bad shape");
  EINX_CHECK_ARG(h0 >= 0 && w0 >= 0 && h0 + H <= Hp && w0 + W <= Wp, "crop window outside the padded map");
  const size_t lds = W <= UP_COLS ? (size_t)2 * UP_ROWS * W * sizeof(float) : 0;
  hipLaunchKernelGGL(upsample_band_kernel, dim3((unsigned)einx_cdiv(W, UP_COLS), (unsigned)hc, (unsigned)B), dim3(UP_COLS), lds, (hipStream_t)stream, raw, D,
                     hc, wc, Hp, Wp, h0, w0, H, W, scale, out);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
