// events.hip -- the step BEFORE the extract+match path: raw events
// (x, y, t, p) -> voxel-grid representation and the events mask, on gfx950.
//
// Replaces (reference file:line): datasets/representations.py:8-21 (time_normalization),
// :67-124 (events_to_voxel_grid: trilinear scatter-add + non-zero mean/std normalisation),
// datasets/visualize.py:23-50 (draw_events_accumulation_image) with the `> 0` mask of
// test_events-image_same-time.py:137.
//
// HBM/atomic-bound integer/float scatter: one thread per event, 8 fp32 atomics (the reference's
// put_(accumulate=True)); the count image uses integer atomics and is therefore bit-exact, the
// voxel grid is exact up to fp32 summation order (atomics commute only approximately).
#include "einx_common.h"

namespace {

struct VoxArgs {
  const float* x;
  const float* y;
  const double* t;
  const float* p;
  long long n;
  int bins, H, W;
  float* grid;  // [bins,H,W] of this sample
};

__global__ void voxel_scatter_kernel(const VoxArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  // time_normalization in float64 (numpy), then float32 (torch) exactly as the reference
  const double t0d = a.t[0], tld = a.t[a.n - 1];
  const double den = (tld - t0d) + 1e-8;
  const float tf = (float)((a.t[i] - t0d) / den);
  const float tf0 = (float)(0.0 / den);
  const float tfl = (float)((tld - t0d) / den);
  const float tn = ((float)(a.bins - 1) * (tf - tf0)) / (tfl - tf0);
  const float xf = a.x[i], yf = a.y[i];
  float value = a.p[i];
  if (value < 1.0f) value = -1.0f;
  const int x0 = (int)xf, y0 = (int)yf, t0 = (int)tn;  // .int() truncates toward zero
#pragma unroll
  for (int dx = 0; dx < 2; ++dx)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int xl = x0 + dx, yl = y0 + dy, tl = t0 + dt;
        if (xl < a.W && xl >= 0 && yl < a.H && yl >= 0 && tl >= 0 && tl < a.bins) {
          const float w = value * (1.0f - fabsf((float)xl - xf)) * (1.0f - fabsf((float)yl - yf)) * (1.0f - fabsf((float)tl - tn));
          atomicAdd(&a.grid[((size_t)tl * a.H + yl) * a.W + xl], w);
        }
      }
}

// per-sample statistics over the non-zero voxels: count, sum, sum of squares (fp64)
__global__ __launch_bounds__(256) void voxel_stats_kernel(const float* grid, long long n, double* stats /*[3]*/) {
  __shared__ double sh[3][4];
  double c = 0.0, s = 0.0, q = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = grid[i];
    if (v != 0.0f) {
      c += 1.0;
      s += (double)v;
      q += (double)v * (double)v;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    c += __shfl_xor(c, off, 64);
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sh[0][wave] = c;
    sh[1][wave] = s;
    sh[2][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[0], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
    atomicAdd(&stats[1], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    atomicAdd(&stats[2], sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3]);
  }
}

// (v - mean) / std (unbiased) on the non-zero voxels; std == 0 -> only centre
__global__ void voxel_normalize_kernel(float* grid, long long n, const double* stats) {
  const double cnt = stats[0];
  if (cnt <= 0.0) return;
  const double mean = stats[1] / cnt;
  double var = 0.0;
  if (cnt > 1.0) var = (stats[2] - cnt * mean * mean) / (cnt - 1.0);
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean;
  const float stdf = (float)sqrt(var);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = grid[i];
    if (v != 0.0f) grid[i] = stdf > 0.0f ? (v - meanf) / stdf : (v - meanf);
  }
}

__global__ void events_count_kernel(const float* x, const float* y, long long n, int H, int W, int32_t* cnt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xi = (int)x[i], yi = (int)y[i];
  if (xi >= 0 && xi < W && yi >= 0 && yi < H) atomicAdd(&cnt[yi * W + xi], 1);
}

__global__ __launch_bounds__(256) void minmax_kernel(const int32_t* cnt, int n, int32_t* mm /*[2]: min, max*/) {
  int lo = 0x7fffffff, hi = -0x7fffffff - 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    lo = min(lo, cnt[i]);
    hi = max(hi, cnt[i]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&mm[0], lo);
    atomicMax(&mm[1], hi);
  }
}

// uint8((cnt - min) / (max - min) * 255) > 0, in float64 like numpy
__global__ void events_mask_kernel(const int32_t* cnt, int n, const int32_t* mm, uint8_t* mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double lo = (double)mm[0], hi = (double)mm[1];
  double v = ((double)cnt[i] - lo) / (hi - lo) * 255.0;  // hi == lo gives NaN like numpy -> uint8 0 ... -> mask false
  if (v > 255.0) v = 255.0;
  mask[i] = (v == v && (int)v > 0) ? 1 : 0;
}

}  // namespace

EINX_EXPORT size_t einx_events_ws_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * 32 + (size_t)B * H * W * sizeof(int32_t) + (size_t)B * 8 + 256;
}

EINX_EXPORT int einx_voxel_grid(const float* x, const float* y, const double* t, const float* p, const int64_t* offsets_host, int B,
                                int bins, int H, int W, int normalize, float* grid, void* ws, void* stream) {
  EINX_CHECK_ARG(x && y && t && p && offsets_host && grid && ws, "null pointer");
  EINX_CHECK_ARG(B > 0 && bins > 0 && H > 0 && W > 0, "bad shape");
  hipStream_t s = (hipStream_t)stream;
  const size_t per = (size_t)bins * H * W;
  double* stats = (double*)ws;  // [B][3] (+1 pad)
  if (hipMemsetAsync(grid, 0, per * B * sizeof(float), s) != hipSuccess || hipMemsetAsync(stats, 0, (size_t)B * 32, s) != hipSuccess) {
    einx_set_error("einx_voxel_grid: memset failed");
    return EINX_ERR_LAUNCH;
  }
  for (int b = 0; b < B; ++b) {
    const long long n = offsets_host[b + 1] - offsets_host[b];
    EINX_CHECK_ARG(n >= 0, "offsets must be non-decreasing");
    if (n == 0) continue;
    VoxArgs a;
    a.x = x + offsets_host[b];
    a.y = y + offsets_host[b];
    a.t = t + offsets_host[b];
    a.p = p + offsets_host[b];
    a.n = n;
    a.bins = bins;
    a.H = H;
    a.W = W;
    a.grid = grid + per * b;
    hipLaunchKernelGGL(voxel_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    EINX_CHECK_LAUNCH();
    if (normalize) {
      hipLaunchKernelGGL(voxel_stats_kernel, dim3(256), dim3(256), 0, s, a.grid, (long long)per, stats + 4 * b);
      EINX_CHECK_LAUNCH();
      hipLaunchKernelGGL(voxel_normalize_kernel, dim3(512), dim3(256), 0, s, a.grid, (long long)per, stats + 4 * b);
      EINX_CHECK_LAUNCH();
    }
  }
  return EINX_OK;
}

EINX_EXPORT int einx_events_mask(const float* x, const float* y, const int64_t* offsets_host, int B, int H, int W, void* ws, uint8_t* mask,
                                 void* stream) {
  EINX_CHECK_ARG(x && y && offsets_host && ws && mask, "null pointer");
  EINX_CHECK_ARG(B > 0 && H > 0 && W > 0, "bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int n = H * W;
  int32_t* cnt = (int32_t*)((char*)ws + (size_t)B * 32);
  int32_t* mm = (int32_t*)((char*)ws + (size_t)B * 32 + (size_t)B * n * sizeof(int32_t));
  if (hipMemsetAsync(cnt, 0, (size_t)B * n * sizeof(int32_t), s) != hipSuccess) {
    einx_set_error("einx_events_mask: memset failed");
    return EINX_ERR_LAUNCH;
  }
  for (int b = 0; b < B; ++b) {
    const int32_t init[2] = {0x7fffffff, -0x7fffffff - 1};
    if (hipMemcpyAsync(mm + 2 * b, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) {
      einx_set_error("einx_events_mask: memcpy failed");
      return EINX_ERR_LAUNCH;
    }
    const long long ne = offsets_host[b + 1] - offsets_host[b];
    if (ne > 0) {
      hipLaunchKernelGGL(events_count_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, x + offsets_host[b], y + offsets_host[b], ne, H, W,
                         cnt + (size_t)b * n);
      EINX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(minmax_kernel, dim3(64), dim3(256), 0, s, cnt + (size_t)b * n, n, mm + 2 * b);
    EINX_CHECK_LAUNCH();
    hipLaunchKernelGGL(events_mask_kernel, dim3((unsigned)einx_cdiv(n, 256)), dim3(256), 0, s, cnt + (size_t)b * n, n, mm + 2 * b, mask + (size_t)b * n);
    EINX_CHECK_LAUNCH();
  }
  return EINX_OK;
}
