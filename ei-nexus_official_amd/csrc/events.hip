// events.hip -- the step BEFORE the extract+match path: raw events
// (x, y, t, p) -> voxel-grid representation and the events mask, on gfx950.
//
// Replaces (reference file:line): datasets/representations.py:8-21 (time_normalization),
// :67-124 (events_to_voxel_grid: trilinear scatter-add + non-zero mean/std normalisation),
// datasets/visualize.py:23-50 (draw_events_accumulation_image) with the `> 0` mask of
// test_events-image_same-time.py:137.
//
// Scatter-add (the reference's put_(accumulate=True)): the voxel grid accumulates in LDS tiles (fp32 LDS atomics,
// exact up to summation order), the count image uses integer global atomics and is bit-exact.
#include <mutex>

#include "einx_common.h"

namespace {

// All samples of a batch go through ONE launch per stage: blockIdx.y is the sample, the device copy of the
// offsets array (one small host-to-device copy per call) delimits its events.
struct VoxArgs {
  const float* x;
  const float* y;
  const double* t;
  const float* p;
  const int64_t* offs;  // device [B+1]
  int bins, H, W;
  float* grid;  // [B,bins,H,W]
};

// Scatter through LDS tiles instead of global float atomics: scattered `global_atomic_add_f32` (64 lanes in 64
// different rows) runs at ~0.08 TB/s on this chip, which made the scatter 80 % of the call.  A workgroup owns
// `rows` image rows of one sample for all time bins (bins*rows*W floats in LDS), sweeps that sample's events
// (reading y first and skipping events that cannot touch its rows), accumulates with LDS atomics, then writes
// its slab with plain coalesced stores -- no memset of the grid, and the per-sample statistics of the non-zero
// voxels come from the slab while it is still in LDS.
__global__ __launch_bounds__(1024) void voxel_tile_kernel(const VoxArgs a, int rows, double* stats_all) {
  extern __shared__ float tile[];  // [bins][rows][W]
  __shared__ double sh[3][16];
  __shared__ int wqueue[16 * 128];  // per-wave queue of event indices that touch this slab
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows;
  const int nr = min(rows, a.H - r0);
  const int tid = threadIdx.x;
  const int slab = a.bins * rows * a.W;
  for (int i = tid; i < slab; i += 1024) tile[i] = 0.0f;
  __syncthreads();
  const long long o0 = a.offs[b], n = a.offs[b + 1] - o0;
  if (n > 0) {
    const double* t = a.t + o0;
    const double t0d = t[0], tld = t[n - 1];
    const double den = (tld - t0d) + 1e-8;
    const float tf0 = (float)(0.0 / den);
    const float tfl = (float)((tld - t0d) / den);
    // Every lane tests its events on y alone (8 loads in flight); the few that can touch this slab (rows/H of
    // them) are appended to a per-wave LDS queue and processed 64 at a time with all lanes busy -- without
    // the queue nearly every wave would run the whole body for every event with one or two active lanes.
    // No workgroup barrier inside the sweep: the 16 waves run independently.
    const int lane = tid & 63, wave = tid >> 6;
    int* queue = wqueue + wave * 128;
    int qn = 0;  // wave-uniform
    auto process = [&](long long i) {
      const float yf = a.y[o0 + i];
      const int y0 = (int)yf;  // .int() truncates toward zero
      // time_normalization in float64 (numpy), then float32 (torch) exactly as the reference
      const float tf = (float)((t[i] - t0d) / den);
      const float tn = ((float)(a.bins - 1) * (tf - tf0)) / (tfl - tf0);
      const float xf = a.x[o0 + i];
      float value = a.p[o0 + i];
      if (value < 1.0f) value = -1.0f;
      const int x0 = (int)xf, t0 = (int)tn;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int xl = x0 + dx, yl = y0 + dy, tl = t0 + dt;
            if (xl < a.W && xl >= 0 && yl < r0 + nr && yl >= r0 && yl >= 0 && tl >= 0 && tl < a.bins) {
              const float w = value * (1.0f - fabsf((float)xl - xf)) * (1.0f - fabsf((float)yl - yf)) * (1.0f - fabsf((float)tl - tn));
              atomicAdd(&tile[(tl * rows + (yl - r0)) * a.W + xl], w);
            }
          }
    };
    for (long long ib = tid; ib - lane < n; ib += 8 * 1024) {  // the wave's lanes walk 8 x 64 consecutive events per trip
      float yv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) yv[u] = a.y[o0 + (ib + u * 1024 < n ? ib + u * 1024 : n - 1)];  // clamped address: guarded loads are serialised by hipcc
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long i = ib + u * 1024;
        const int y0 = (int)yv[u];
        const bool hit = i < n && y0 + 1 >= r0 && y0 < r0 + nr;
        const unsigned long long m = __ballot(hit);
        if (m == 0) continue;
        if (hit) queue[qn + __popcll(m & ((1ull << lane) - 1ull))] = (int)i;
        qn += __popcll(m);
        if (qn >= 64) {  // wave-uniform
          process((long long)queue[lane]);
          qn -= 64;
          if (lane < qn) {
            const int v = queue[64 + lane];
            queue[lane] = v;
          }
        }
      }
    }
    if (lane < qn) process((long long)queue[lane]);
  }
  __syncthreads();
  float* grid = a.grid + (size_t)b * a.bins * a.H * a.W;
  double c = 0.0, sm = 0.0, q = 0.0;
  const int rowlen = nr * a.W;
  for (int tb = 0; tb < a.bins; ++tb) {
    const float* src = tile + tb * rows * a.W;
    float* dst = grid + ((size_t)tb * a.H + r0) * a.W;
    for (int i = tid; i < rowlen; i += 1024) {
      const float v = src[i];
      dst[i] = v;
      if (v != 0.0f) {
        c += 1.0;
        sm += (double)v;
        q += (double)v * (double)v;
      }
    }
  }
  if (stats_all) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      c += __shfl_xor(c, off, 64);
      sm += __shfl_xor(sm, off, 64);
      q += __shfl_xor(q, off, 64);
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) {
      sh[0][wave] = c;
      sh[1][wave] = sm;
      sh[2][wave] = q;
    }
    __syncthreads();
    if (tid == 0) {
      double c2 = 0.0, s2 = 0.0, q2 = 0.0;
      for (int w = 0; w < 16; ++w) {
        c2 += sh[0][w];
        s2 += sh[1][w];
        q2 += sh[2][w];
      }
      double* stats = stats_all + 4 * b;
      atomicAdd(&stats[0], c2);
      atomicAdd(&stats[1], s2);
      atomicAdd(&stats[2], q2);
    }
  }
}

// (v - mean) / std (unbiased) on the non-zero voxels; std == 0 -> only centre; grid (blocks, B)
__global__ void voxel_normalize_kernel(float* grid_all, long long n, const double* stats_all) {
  float* grid = grid_all + (size_t)blockIdx.y * n;
  const double* stats = stats_all + 4 * blockIdx.y;
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

__global__ void events_count_kernel(const float* x, const float* y, const int64_t* offs, int H, int W, int32_t* cnt_all) {
  const int b = blockIdx.y;
  const long long o0 = offs[b], n = offs[b + 1] - o0;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xi = (int)x[o0 + i], yi = (int)y[o0 + i];
  if (xi >= 0 && xi < W && yi >= 0 && yi < H) atomicAdd(&cnt_all[(size_t)b * H * W + yi * W + xi], 1);
}

// one 1024-thread workgroup per sample: min and max of the count image
__global__ __launch_bounds__(1024) void minmax_kernel(const int32_t* cnt_all, int n, int32_t* mm_all /*[B][2]: min, max*/) {
  __shared__ int slo[16], shi[16];
  const int32_t* cnt = cnt_all + (size_t)blockIdx.x * n;
  int lo = 0x7fffffff, hi = -0x7fffffff - 1;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {  // eight loads in flight; a clamped index re-reads a valid element (harmless for min / max)
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = cnt[min(i0 + u * 1024, n - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      lo = min(lo, v[u]);
      hi = max(hi, v[u]);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      lo = min(lo, slo[w]);
      hi = max(hi, shi[w]);
    }
    mm_all[2 * blockIdx.x] = lo;
    mm_all[2 * blockIdx.x + 1] = hi;
  }
}

// uint8((cnt - min) / (max - min) * 255) > 0, in float64 like numpy; grid (blocks, B)
__global__ void events_mask_kernel(const int32_t* cnt_all, int n, const int32_t* mm_all, uint8_t* mask_all) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double lo = (double)mm_all[2 * b], hi = (double)mm_all[2 * b + 1];
  double v = ((double)cnt_all[(size_t)b * n + i] - lo) / (hi - lo) * 255.0;  // hi == lo gives NaN like numpy -> uint8 0 ... -> mask false
  if (v > 255.0) v = 255.0;
  mask_all[(size_t)b * n + i] = (v == v && (int)v > 0) ? 1 : 0;
}

}  // namespace

// workspace: [B][4] fp64 statistics | count image int32 [B,H,W] | min/max int32 [B][2] | device offsets int64 [B+1]
EINX_EXPORT size_t einx_events_ws_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * 32 + (size_t)B * H * W * sizeof(int32_t) + (size_t)B * 8 + ((size_t)B + 1) * 8 + 256;
}

namespace {
// The caller's offsets array is pageable memory that it may free as soon as the call returns, so it is first copied
// (synchronously, a few hundred bytes) into a library-owned PINNED staging buffer; the asynchronous host-to-device
// copy then reads that.  One buffer per host thread, reused once the previous copy's event has completed.
struct PinnedOffsets {
  int64_t* p = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
  ~PinnedOffsets() {
    if (p) (void)hipHostFree(p);
    if (done) (void)hipEventDestroy(done);
  }
};
// one staging state per (host thread, device): an event can only be recorded on a stream of the device it was created on
constexpr int kMaxDev = 64;
thread_local PinnedOffsets g_offs[kMaxDev];

// copies the host offsets to the workspace and returns the largest per-sample event count (-1 on bad input)
long long stage_offsets(const int64_t* offsets_host, int B, int H, int W, void* ws, hipStream_t s, int64_t** dev) {
  long long mx = 0;
  for (int b = 0; b < B; ++b) {
    const long long n = offsets_host[b + 1] - offsets_host[b];
    if (n < 0) return -1;
    mx = n > mx ? n : mx;
  }
  char* p = (char*)ws + (size_t)B * 32 + (size_t)B * H * W * sizeof(int32_t) + (size_t)B * 8;
  p = (char*)(((size_t)p + 7) & ~(size_t)7);
  *dev = (int64_t*)p;
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= kMaxDev) return -2;
  PinnedOffsets& st = g_offs[devid];
  const size_t need = (size_t)B + 1;
  if (st.done && hipEventSynchronize(st.done) != hipSuccess) return -2;  // the previous call's copy has left the buffer
  if (need > st.cap) {
    if (st.p) (void)hipHostFree(st.p);
    st.p = nullptr;
    st.cap = 0;
    if (hipHostMalloc((void**)&st.p, need * 2 * sizeof(int64_t), hipHostMallocPortable) != hipSuccess) return -2;
    st.cap = need * 2;
  }
  if (!st.done && hipEventCreateWithFlags(&st.done, hipEventDisableTiming) != hipSuccess) return -2;
  for (size_t i = 0; i < need; ++i) st.p[i] = offsets_host[i];
  if (hipMemcpyAsync(p, st.p, need * 8, hipMemcpyHostToDevice, s) != hipSuccess) return -2;
  if (hipEventRecord(st.done, s) != hipSuccess) return -2;
  return mx;
}

// hipFuncSetAttribute is per device: remember the largest dynamic-LDS size granted on each one
int reserve_voxel_lds(size_t lds) {
  static std::mutex mu;
  static size_t granted[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (lds <= granted[dev]) return 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&voxel_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return -1;
  granted[dev] = lds;
  return 0;
}
}  // namespace

EINX_EXPORT int einx_voxel_grid(const float* x, const float* y, const double* t, const float* p, const int64_t* offsets_host, int B,
                                int bins, int H, int W, int normalize, float* grid, void* ws, void* stream) {
  EINX_CHECK_ARG(x && y && t && p && offsets_host && grid && ws, "null pointer");
  EINX_CHECK_ARG(B > 0 && bins > 0 && H > 0 && W > 0, "bad shape");
  hipStream_t s = (hipStream_t)stream;
  const size_t per = (size_t)bins * H * W;
  double* stats = (double*)ws;  // [B][4]
  int64_t* offs = nullptr;
  const long long mx = stage_offsets(offsets_host, B, H, W, ws, s, &offs);
  EINX_CHECK_ARG(mx != -1, "offsets must be non-decreasing");
  if (mx == -2 || hipMemsetAsync(stats, 0, (size_t)B * 32, s) != hipSuccess) {
    einx_set_error("einx_voxel_grid: memset / copy failed");
    return EINX_ERR_LAUNCH;
  }
  VoxArgs a;
  a.x = x;
  a.y = y;
  a.t = t;
  a.p = p;
  a.offs = offs;
  a.bins = bins;
  a.H = H;
  a.W = W;
  a.grid = grid;
  // rows per workgroup: the slab bins*rows*W floats stays under ~68 KB (+8 KB of per-wave queues) so that two workgroups share a CU
  int rows = (int)(17000 / ((long long)bins * W));
  rows = rows < 1 ? 1 : (rows > H ? H : rows);
  const size_t lds = (size_t)bins * rows * W * sizeof(float);
  EINX_CHECK_ARG(lds <= 150 * 1024, "bins * W too large for the LDS tile scatter");
  if (reserve_voxel_lds(lds) != 0) {
    einx_set_error("einx_voxel_grid: cannot reserve %zu bytes of LDS", lds);
    return EINX_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(voxel_tile_kernel, dim3((unsigned)einx_cdiv(H, rows), (unsigned)B), dim3(1024), lds, s, a, rows, normalize ? stats : nullptr);
  EINX_CHECK_LAUNCH();
  if (normalize) {
    hipLaunchKernelGGL(voxel_normalize_kernel, dim3(128, (unsigned)B), dim3(256), 0, s, grid, (long long)per, stats);
    EINX_CHECK_LAUNCH();
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
  int64_t* offs = nullptr;
  const long long mx = stage_offsets(offsets_host, B, H, W, ws, s, &offs);
  EINX_CHECK_ARG(mx != -1, "offsets must be non-decreasing");
  if (mx == -2 || hipMemsetAsync(cnt, 0, (size_t)B * n * sizeof(int32_t), s) != hipSuccess) {
    einx_set_error("einx_events_mask: memset / copy failed");
    return EINX_ERR_LAUNCH;
  }
  if (mx > 0) {
    hipLaunchKernelGGL(events_count_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)B), dim3(256), 0, s, x, y, offs, H, W, cnt);
    EINX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)B), dim3(1024), 0, s, cnt, n, mm);
  EINX_CHECK_LAUNCH();
  hipLaunchKernelGGL(events_mask_kernel, dim3((unsigned)einx_cdiv(n, 256), (unsigned)B), dim3(256), 0, s, cnt, n, mm, mask);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
