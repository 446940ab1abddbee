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
    float s = 0.0f;
    for (int c = 0; c < D; ++c) {
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

// thread per output pixel of the CROPPED window; bilinear (align_corners=False) in the padded frame
__global__ void upsample_normalize_kernel(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, int h0, int w0, int H, int W,
                                          float scale, float* out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t HW = (size_t)H * W;
  if (gid >= (size_t)B * HW) return;
  const int b = (int)(gid / HW);
  const int rem = (int)(gid % HW);
  const int y = rem / W + h0, x = rem % W + w0;
  const float sy = (float)hc / (float)Hp, sx = (float)wc / (float)Wp;
  float fy = ((float)y + 0.5f) * sy - 0.5f;
  if (fy < 0.0f) fy = 0.0f;
  const int y0 = (int)fy, y1 = y0 + (y0 < hc - 1 ? 1 : 0);
  const float ly = fy - (float)y0, hy = 1.0f - ly;
  float fx = ((float)x + 0.5f) * sx - 0.5f;
  if (fx < 0.0f) fx = 0.0f;
  const int x0 = (int)fx, x1 = x0 + (x0 < wc - 1 ? 1 : 0);
  const float lx = fx - (float)x0, hx = 1.0f - lx;
  const size_t plane = (size_t)hc * wc;
  const float* rb = raw + (size_t)b * D * plane;
  float s = 0.0f;
  for (int c = 0; c < D; ++c) {
    const float* p = rb + (size_t)c * plane;
    const float v = hy * (hx * p[y0 * wc + x0] + lx * p[y0 * wc + x1]) + ly * (hx * p[y1 * wc + x0] + lx * p[y1 * wc + x1]);
    s = fmaf(v, v, s);
  }
  const float den = fmaxf(sqrtf(s), 1e-12f);
  float* o = out + (size_t)b * D * HW + rem;
  for (int c = 0; c < D; ++c) {
    const float* p = rb + (size_t)c * plane;
    const float v = hy * (hx * p[y0 * wc + x0] + lx * p[y0 * wc + x1]) + ly * (hx * p[y1 * wc + x0] + lx * p[y1 * wc + x1]);
    o[(size_t)c * HW] = scale * (v / den);
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
  const size_t n = (size_t)B * H * W;
  hipLaunchKernelGGL(upsample_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw, B, D, hc, wc,
                     Hp, Wp, h0, w0, H, W, scale, out);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
