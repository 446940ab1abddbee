// metrics.hip -- the consumers right after the extract+match path: per-pair
// MatchingRatio, MeanMatchingAccuracy@t and ValidDescriptorsDistance (repeatability / descriptor
// distance / angle @t) computed on the device for a whole batch, so the evaluation harness needs no
// per-pair .item() round trips and the multi-GPU job can all-reduce plain sums.
//
// Replaces (reference file:line): core/metrics/matching_metrics.py:30-51 (MatchingRatio),
// :84-156 (MeanMatchingAccuracy), core/metrics/keypoints_metrics.py:160-290
// (ValidDescriptorsDistance), core/metrics/util.py:5-104 (warp_points, keep_true_points).
// Latency-bound integer/compare work on <=1024x1024 point pairs: one wave per keypoint, lanes
// over the other image's keypoints, wave shuffles for the arg-min.
#include "einx_common.h"

namespace {

struct MetArgs {
  const float *k0, *k1, *d0, *d1, *mk0, *mk1, *hom;
  const int32_t *n, *m, *nmatch;
  float* tw0;    // [B,cap0,2] keypoints0 warped by H (x,y)
  float* p1;     // [B,cap1,2] keypoints1 (x,y)
  uint8_t* keep0;
  uint8_t* keep1;
  int32_t* icnt;  // [B][16]: N1, N2, (unused 2..9), mma_good@t.. at 10..
  float* near0;   // [B,cap0,3] per kept keypoint of image 0: nearest distance, descriptor distance, angle
  float* near1;   // [B,cap1,3] same for image 1
  double* out;
  einx_metric_params p;
};

__device__ __forceinline__ void load_h(const float* hom, int b, float* h) {
  if (hom) {
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = hom[b * 9 + i];
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = (i % 4 == 0) ? 1.0f : 0.0f;
  }
}

// torch.mm(H, [x;y;1]) then division by the third row (core/metrics/util.py:17-33)
__device__ __forceinline__ void warp(const float* h, float x, float y, float* ox, float* oy) {
  const float a = fmaf(h[2], 1.0f, fmaf(h[1], y, h[0] * x));
  const float b = fmaf(h[5], 1.0f, fmaf(h[4], y, h[3] * x));
  const float c = fmaf(h[8], 1.0f, fmaf(h[7], y, h[6] * x));
  *ox = a / c;
  *oy = b / c;
}

__device__ __forceinline__ void inverse3(const float* h, float* inv) {
  const float c00 = h[4] * h[8] - h[5] * h[7], c01 = h[5] * h[6] - h[3] * h[8], c02 = h[3] * h[7] - h[4] * h[6];
  const float det = h[0] * c00 + h[1] * c01 + h[2] * c02;
  inv[0] = c00 / det;
  inv[1] = (h[2] * h[7] - h[1] * h[8]) / det;
  inv[2] = (h[1] * h[5] - h[2] * h[4]) / det;
  inv[3] = c01 / det;
  inv[4] = (h[0] * h[8] - h[2] * h[6]) / det;
  inv[5] = (h[2] * h[3] - h[0] * h[5]) / det;
  inv[6] = c02 / det;
  inv[7] = (h[1] * h[6] - h[0] * h[7]) / det;
  inv[8] = (h[0] * h[4] - h[1] * h[3]) / det;
}

constexpr int ICNT = 16;

__global__ void metric_prepare_kernel(const MetArgs a) {
  const int b = blockIdx.y;
  const int n = min(a.n[b], a.p.cap0), m = min(a.m[b], a.p.cap1);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float h[9], hi[9];
  load_h(a.hom, b, h);
  const int xi = a.p.kp_yx ? 1 : 0, yi = a.p.kp_yx ? 0 : 1;
  if (t < n) {
    const float* k = a.k0 + ((size_t)b * a.p.cap0 + t) * 3;
    float wx, wy;
    warp(h, k[xi], k[yi], &wx, &wy);
    a.tw0[((size_t)b * a.p.cap0 + t) * 2] = wx;
    a.tw0[((size_t)b * a.p.cap0 + t) * 2 + 1] = wy;
    const bool keep = wx >= 0.0f && wx < (float)a.p.W1 && wy >= 0.0f && wy < (float)a.p.H1;  // inside img2_shape
    a.keep0[(size_t)b * a.p.cap0 + t] = keep;
    if (keep) atomicAdd(&a.icnt[b * ICNT + 0], 1);
  }
  if (t < m) {
    inverse3(h, hi);
    const float* k = a.k1 + ((size_t)b * a.p.cap1 + t) * 3;
    float wx, wy;
    warp(hi, k[xi], k[yi], &wx, &wy);
    a.p1[((size_t)b * a.p.cap1 + t) * 2] = k[xi];
    a.p1[((size_t)b * a.p.cap1 + t) * 2 + 1] = k[yi];
    const bool keep = wx >= 0.0f && wx < (float)a.p.W0 && wy >= 0.0f && wy < (float)a.p.H0;  // inside img1_shape
    a.keep1[(size_t)b * a.p.cap1 + t] = keep;
    if (keep) atomicAdd(&a.icnt[b * ICNT + 1], 1);
  }
}

// SIDE 0: for each kept warped keypoint of image 0, nearest kept keypoint of image 1 (torch.min(norm,1));
// SIDE 1: for each kept keypoint of image 1, nearest warped keypoint of image 0 (torch.min(norm,0)).
template <int SIDE>
__global__ __launch_bounds__(256) void metric_nearest_kernel(const MetArgs a) {
  const int b = blockIdx.y;
  const int n = min(a.n[b], a.p.cap0), m = min(a.m[b], a.p.cap1);
  const int self_n = SIDE == 0 ? n : m, other_n = SIDE == 0 ? m : n;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= self_n) return;
  const int cs = SIDE == 0 ? a.p.cap0 : a.p.cap1, co = SIDE == 0 ? a.p.cap1 : a.p.cap0;
  const uint8_t* keep_s = (SIDE == 0 ? a.keep0 : a.keep1) + (size_t)b * cs;
  const uint8_t* keep_o = (SIDE == 0 ? a.keep1 : a.keep0) + (size_t)b * co;
  if (!keep_s[i]) {
    if (lane == 0) ((SIDE == 0 ? a.near0 : a.near1) + ((size_t)b * cs + i) * 3)[0] = einx_u2f(0x7f800000u);
    return;
  }
  const float* ps = (SIDE == 0 ? a.tw0 : a.p1) + ((size_t)b * cs + i) * 2;
  const float* po = (SIDE == 0 ? a.p1 : a.tw0) + (size_t)b * co * 2;
  const float sx = ps[0], sy = ps[1];
  float best = einx_u2f(0x7f800000u);
  int bj = 0x7fffffff;
  // four candidates per lane and trip, their flags and coordinates requested together from clamped addresses (one guarded
  // load per iteration is a memory round trip each); evaluated in ascending j as before (strict <: the first minimum wins)
  for (int j0 = lane; j0 < other_n; j0 += 256) {
    uint8_t kk[4];
    float ox[4], oy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jc = min(j0 + 64 * u, other_n - 1);
      kk[u] = keep_o[jc];
      ox[u] = po[2 * jc];
      oy[u] = po[2 * jc + 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + 64 * u;
      if (j >= other_n || !kk[u]) continue;
      // norm of (warped0 - p1): the difference is taken in that direction on both sides
      const float dx = SIDE == 0 ? sx - ox[u] : ox[u] - sx;
      const float dy = SIDE == 0 ? sy - oy[u] : oy[u] - sy;
      const float d = sqrtf(fmaf(dy, dy, dx * dx));
      if (d < best) {
        best = d;
        bj = j;
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oj = __shfl_xor(bj, off, 64);
    if (ob < best || (ob == best && oj < bj)) {
      best = ob;
      bj = oj;
    }
  }
  float* nout = (SIDE == 0 ? a.near0 : a.near1) + ((size_t)b * cs + i) * 3;
  if (bj == 0x7fffffff) {
    if (lane == 0) nout[0] = einx_u2f(0x7f800000u);
    return;
  }
  // descriptor distance and angle of the pair (keypoints_metrics.py:233-247)
  const int D = a.p.D;
  const float* da = (SIDE == 0 ? a.d0 : a.d1) + ((size_t)b * cs + i) * D;
  const float* db = (SIDE == 0 ? a.d1 : a.d0) + ((size_t)b * co + bj) * D;
  // valid_desc1 always indexes image 0 and valid_desc2 image 1, whichever side drives the loop
  const float* v1 = SIDE == 0 ? da : db;
  const float* v2 = SIDE == 0 ? db : da;
  float sq = 0.0f, dot = 0.0f, n1 = 0.0f, n2 = 0.0f;
  for (int c0 = lane; c0 < D; c0 += 256) {  // same per-lane order c = lane, lane + 64, ...; four channel pairs in flight
    float a1[4], a2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cc = min(c0 + 64 * u, D - 1);
      a1[u] = v1[cc];
      a2[u] = v2[cc];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c0 + 64 * u >= D) continue;
      const float x1 = a1[u], x2 = a2[u];
      const float df = x1 - x2;
      sq = fmaf(df, df, sq);
      dot = fmaf(x1, x2, dot);
      n1 = fmaf(x1, x1, n1);
      n2 = fmaf(x2, x2, n2);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sq += __shfl_xor(sq, off, 64);
    dot += __shfl_xor(dot, off, 64);
    n1 += __shfl_xor(n1, off, 64);
    n2 += __shfl_xor(n2, off, 64);
  }
  if (lane == 0) {
    nout[0] = best;
    nout[1] = sqrtf(sq);
    nout[2] = einx_acosf(dot / (sqrtf(n1) * sqrtf(n2))) * 57.29577951308232f;
  }
}

__global__ void metric_mma_kernel(const MetArgs a) {
  const int b = blockIdx.y;
  const int M = min(a.nmatch[b], a.p.cap0);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M) return;
  float h[9];
  load_h(a.hom, b, h);
  const int xi = a.p.kp_yx ? 1 : 0, yi = a.p.kp_yx ? 0 : 1;
  const float* q0 = a.mk0 + ((size_t)b * a.p.cap0 + t) * a.p.cols;
  const float* q1 = a.mk1 + ((size_t)b * a.p.cap0 + t) * a.p.cols;
  float wx, wy;
  warp(h, q0[xi], q0[yi], &wx, &wy);
  const float dx = wx - q1[xi], dy = wy - q1[yi];
  const float d = sqrtf(dx * dx + dy * dy);
  for (int k = 0; k < a.p.n_mma; ++k)
    if (d <= a.p.mma_thr[k]) atomicAdd(&a.icnt[b * ICNT + 10 + k], 1);
}

// one workgroup per pair: fixed-order (deterministic) reduction of the per-keypoint records
__global__ __launch_bounds__(256) void metric_finalize_kernel(const MetArgs a) {
  __shared__ double sh[256];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int n = min(a.n[b], a.p.cap0), m = min(a.m[b], a.p.cap1), M = min(a.nmatch[b], a.p.cap0);
  const int nout = 1 + a.p.n_mma + 3 * a.p.n_vdd;
  double* o = a.out + (size_t)b * nout;
  const int32_t* ic = a.icnt + b * ICNT;
  auto block_sum = [&](double v) {
    sh[tid] = v;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if (tid < off) sh[tid] += sh[tid + off];
      __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
  };
  if (tid == 0) {
    o[0] = (double)M / ((double)(n < m ? n : m) + 1e-8);  // MatchingRatio
    for (int k = 0; k < a.p.n_mma; ++k) o[1 + k] = M > 0 ? (double)((float)ic[10 + k] / (float)M) : 0.0;  // mask.float().mean()
  }
  const int N1 = ic[0], N2 = ic[1];
  for (int t = 0; t < a.p.n_vdd; ++t) {
    double c = 0.0, sd = 0.0, sa = 0.0;
    const float thr = a.p.vdd_thr[t];
    if (a.p.n_vdd > 0) {
      for (int i = tid; i < n; i += 256) {
        const float* r = a.near0 + ((size_t)b * a.p.cap0 + i) * 3;
        if (r[0] <= thr) {
          c += 1.0;
          sd += (double)r[1];
          sa += (double)r[2];
        }
      }
      for (int j = tid; j < m; j += 256) {
        const float* r = a.near1 + ((size_t)b * a.p.cap1 + j) * 3;
        if (r[0] <= thr) {
          c += 1.0;
          sd += (double)r[1];
          sa += (double)r[2];
        }
      }
    }
    c = block_sum(c);
    sd = block_sum(sd);
    sa = block_sum(sa);
    if (tid == 0) {
      double rep = 0.0, vd = 0.0, ang = 0.0;
      if (N1 != 0 && N2 != 0) {
        rep = (double)((float)c / (float)(N1 + N2));
        vd = sd / c;  // 0/0 -> NaN exactly like the reference when nothing is within t
        ang = sa / c;
      }
      if (a.p.rep_nan_if_empty && N1 + N2 == 0) rep = __longlong_as_double(0x7ff8000000000000LL);
      o[1 + a.p.n_mma + 3 * t + 0] = rep;
      o[1 + a.p.n_mma + 3 * t + 1] = vd;
      o[1 + a.p.n_mma + 3 * t + 2] = ang;
    }
  }
}

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

EINX_EXPORT size_t einx_metrics_ws_bytes(const einx_metric_params* p) {
  if (!p || p->B <= 0) return 0;
  const size_t B = p->B;
  return al(B * p->cap0 * 8) + al(B * p->cap1 * 8) + al(B * p->cap0) + al(B * p->cap1) + al(B * ICNT * 4) + al(B * p->cap0 * 12) +
         al(B * p->cap1 * 12) + 256;
}

EINX_EXPORT int einx_pair_metrics(const einx_metric_params* p, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                                  const int32_t* n, const int32_t* m, const float* mk0, const float* mk1, const int32_t* nmatch,
                                  const float* homography, void* ws, double* out, void* stream) {
  EINX_CHECK_ARG(p && kpts0 && kpts1 && desc0 && desc1 && n && m && mk0 && mk1 && nmatch && ws && out, "null pointer");
  EINX_CHECK_ARG(p->B > 0 && p->cap0 > 0 && p->cap1 > 0 && p->D > 0, "bad shape");
  EINX_CHECK_ARG(p->n_mma >= 0 && p->n_mma <= 4 && p->n_vdd >= 0 && p->n_vdd <= 4, "at most 4 thresholds per metric");
  EINX_CHECK_ARG(p->cols == 2 || p->cols == 3, "matched keypoints have 2 or 3 columns");
  hipStream_t s = (hipStream_t)stream;
  MetArgs a;
  a.k0 = kpts0;
  a.k1 = kpts1;
  a.d0 = desc0;
  a.d1 = desc1;
  a.mk0 = mk0;
  a.mk1 = mk1;
  a.hom = homography;
  a.n = n;
  a.m = m;
  a.nmatch = nmatch;
  a.p = *p;
  a.out = out;
  char* q = (char*)ws;
  const size_t B = p->B;
  a.tw0 = (float*)q;
  q += al(B * p->cap0 * 8);
  a.p1 = (float*)q;
  q += al(B * p->cap1 * 8);
  a.keep0 = (uint8_t*)q;
  q += al(B * p->cap0);
  a.keep1 = (uint8_t*)q;
  q += al(B * p->cap1);
  a.icnt = (int32_t*)q;
  q += al(B * ICNT * 4);
  a.near0 = (float*)q;
  q += al(B * p->cap0 * 12);
  a.near1 = (float*)q;
  if (hipMemsetAsync(a.icnt, 0, al(B * ICNT * 4), s) != hipSuccess) {
    einx_set_error("einx_pair_metrics: memset failed");
    return EINX_ERR_LAUNCH;
  }
  const int mx = p->cap0 > p->cap1 ? p->cap0 : p->cap1;
  hipLaunchKernelGGL(metric_prepare_kernel, dim3((unsigned)einx_cdiv(mx, 256), (unsigned)B), dim3(256), 0, s, a);
  EINX_CHECK_LAUNCH();
  if (p->n_vdd > 0) {
    hipLaunchKernelGGL(metric_nearest_kernel<0>, dim3((unsigned)einx_cdiv(p->cap0, 4), (unsigned)B), dim3(256), 0, s, a);
    EINX_CHECK_LAUNCH();
    hipLaunchKernelGGL(metric_nearest_kernel<1>, dim3((unsigned)einx_cdiv(p->cap1, 4), (unsigned)B), dim3(256), 0, s, a);
    EINX_CHECK_LAUNCH();
  }
  if (p->n_mma > 0) {
    hipLaunchKernelGGL(metric_mma_kernel, dim3((unsigned)einx_cdiv(p->cap0, 256), (unsigned)B), dim3(256), 0, s, a);
    EINX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(metric_finalize_kernel, dim3((unsigned)B), dim3(256), 0, s, a);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
