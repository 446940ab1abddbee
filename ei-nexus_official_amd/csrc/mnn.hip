// mnn.hip -- mutual-nearest-neighbour matcher on gfx950.
// sim = desc0 . desc1^T on the fp32 matrix cores (gemm_tile.h, exact k-ordered chains), with the
// row/column arg-max fused into the tile epilogue: each 128x128 tile reduces its rows across
// lanes and its columns across registers, then publishes packed (value, ~index) keys with 64-bit
// atomicMax -- max is order independent, so the result is deterministic, and ties resolve to the
// lowest index exactly like the oracle / torch.topk(1) on CPU.  sim is never written to HBM.
// Optional log_assignment (dual log-softmax) runs two more tile passes (statistics, then write).
//
// Replaces (reference file:line): core/modules/matchers/MNN.py:12-32 (find_nn, mutual_check),
// :88-129 (similarity, log_assignment, matched keypoint gather); lightglue.py:690-698 (gather).
#include "match_tiles.h"

using namespace einx_match;
using namespace einx_gemm;


EINX_EXPORT size_t einx_mnn_ws_bytes(int B, int cap0, int cap1) {
  if (B <= 0 || cap0 <= 0 || cap1 <= 0) return 0;
  const size_t nc64 = (size_t)einx_cdiv(cap1, 64), nr64 = (size_t)einx_cdiv(cap0, WROWS);
  size_t bytes = 0;
  bytes += align256((size_t)B * cap0 * 8) + align256((size_t)B * cap1 * 8);
  bytes += align256((size_t)B * cap0 * nc64 * 8) + align256((size_t)B * cap1 * nr64 * 8);
  bytes += align256((size_t)B * cap0 * 8) + align256((size_t)B * cap1 * 8);
  bytes += align256((size_t)B * cap0 * 4) + align256((size_t)B * cap1 * 4);  // second-neighbour keys (einx_mnn_thresh)
  return bytes;
}

EINX_EXPORT int einx_mnn(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B, int D,
                         void* ws, int64_t* matches0, int64_t* matches1, float* scores0, float* scores1, float* la, void* stream) {
  return einx_mnn_thresh(desc0, n, cap0, desc1, m, cap1, B, D, 0, 0.0f, 0, 0.0f, ws, matches0, matches1, scores0, scores1, la, stream);
}

namespace {
struct GatherOut {  // optional: the matched keypoints of every pair, compacted in row order (einx_gather_matches' outputs)
  const float *k0 = nullptr, *k1 = nullptr;
  int cols = 0;
  float *o0 = nullptr, *o1 = nullptr;
  int32_t* nmatch = nullptr;
};
int mnn_impl(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B, int D, int use_ratio,
             float ratio_sq, int use_dist, float dist_sq, void* ws, int64_t* matches0, int64_t* matches1, float* scores0, float* scores1, float* la,
             const GatherOut& go, void* stream);
}  // namespace

EINX_EXPORT int einx_mnn_thresh(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B, int D,
                                int use_ratio, float ratio_sq, int use_dist, float dist_sq, void* ws, int64_t* matches0, int64_t* matches1,
                                float* scores0, float* scores1, float* la, void* stream) {
  return mnn_impl(desc0, n, cap0, desc1, m, cap1, B, D, use_ratio, ratio_sq, use_dist, dist_sq, ws, matches0, matches1, scores0, scores1, la,
                  GatherOut{}, stream);
}

EINX_EXPORT int einx_mnn_gather(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B, int D,
                                void* ws, int64_t* matches0, int64_t* matches1, float* scores0, float* scores1, float* la, const float* kpts0,
                                const float* kpts1, int cols, float* mk0, float* mk1, int32_t* nmatch, void* stream) {
  EINX_CHECK_ARG(kpts0 && kpts1 && mk0 && mk1 && nmatch && cols >= 1 && cols <= 3, "null pointer / bad cols");
  GatherOut go;
  go.k0 = kpts0;
  go.k1 = kpts1;
  go.cols = cols;
  go.o0 = mk0;
  go.o1 = mk1;
  go.nmatch = nmatch;
  return mnn_impl(desc0, n, cap0, desc1, m, cap1, B, D, 0, 0.0f, 0, 0.0f, ws, matches0, matches1, scores0, scores1, la, go, stream);
}

namespace {
int mnn_impl(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B, int D, int use_ratio,
             float ratio_sq, int use_dist, float dist_sq, void* ws, int64_t* matches0, int64_t* matches1, float* scores0, float* scores1, float* la,
             const GatherOut& go, void* stream) {
  EINX_CHECK_ARG(desc0 && desc1 && n && m && ws && matches0 && matches1 && scores0 && scores1, "null pointer");
  EINX_CHECK_ARG(B > 0 && cap0 > 0 && cap1 > 0 && D > 0 && D % 4 == 0, "bad shape (D must be a multiple of 4)");
  hipStream_t s = (hipStream_t)stream;
  MnnArgs a;
  a.d0 = desc0;
  a.d1 = desc1;
  a.n = n;
  a.m = m;
  a.cap0 = cap0;
  a.cap1 = cap1;
  a.D = D;
  a.nc64 = einx_cdiv(cap1, 64);
  a.nr64 = einx_cdiv(cap0, WROWS);
  char* p = (char*)ws;
  a.rowkey = (unsigned long long*)p;
  p += align256((size_t)B * cap0 * 8);
  a.colkey = (unsigned long long*)p;
  p += align256((size_t)B * cap1 * 8);
  a.rowstat = (float*)p;
  p += align256((size_t)B * cap0 * a.nc64 * 8);
  a.colstat = (float*)p;
  p += align256((size_t)B * cap1 * a.nr64 * 8);
  a.rowlse = (float*)p;
  p += align256((size_t)B * cap0 * 8);
  a.collse = (float*)p;
  p += align256((size_t)B * cap1 * 8);
  a.row2 = (unsigned*)p;
  p += align256((size_t)B * cap0 * 4);
  a.col2 = (unsigned*)p;
  a.la = la;
  a.cert0 = a.cert1 = a.dust0 = a.dust1 = nullptr;
  const size_t keybytes = align256((size_t)B * cap0 * 8) + align256((size_t)B * cap1 * 8);
  if (hipMemsetAsync(ws, 0, keybytes, s) != hipSuccess) {
    einx_set_error("einx_mnn: memset failed");
    return EINX_ERR_LAUNCH;
  }
  const bool thresh = use_ratio || use_dist;
  if (use_ratio && hipMemsetAsync(a.row2, 0, align256((size_t)B * cap0 * 4) + align256((size_t)B * cap1 * 4), s) != hipSuccess) {
    einx_set_error("einx_mnn: memset failed");
    return EINX_ERR_LAUNCH;
  }
  const dim3 grid((unsigned)einx_cdiv(cap1, BN), (unsigned)einx_cdiv(cap0, BM), (unsigned)B);
  if (la) {  // arg-max keys and the log_assignment's softmax statistics from one visit of every tile
    EINX_PROF("mnn_tile_kernel<5>", s);
    hipLaunchKernelGGL(mnn_tile_kernel<5>, grid, dim3(THREADS), 0, s, a);
  } else {
    EINX_PROF("mnn_tile_kernel<0>", s);
    hipLaunchKernelGGL(mnn_tile_kernel<0>, grid, dim3(THREADS), 0, s, a);
  }
  EINX_CHECK_LAUNCH();
  const int mx = cap0 > cap1 ? cap0 : cap1;
  if (thresh) {
    if (use_ratio) {
      hipLaunchKernelGGL(mnn_tile_kernel<4>, grid, dim3(THREADS), 0, s, a);
      EINX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(mnn_finalize_thresh_kernel, dim3((unsigned)einx_cdiv(mx, 256), (unsigned)B), dim3(256), 0, s, a.rowkey, a.colkey, a.row2,
                       a.col2, n, m, cap0, cap1, use_ratio, ratio_sq, use_dist, dist_sq, matches0, matches1, scores0, scores1);
  } else if (go.nmatch) {  // mutual check + matched-keypoint compaction in one launch
    hipLaunchKernelGGL(mnn_finalize_gather_kernel, dim3((unsigned)B), dim3(1024), 0, s, a.rowkey, a.colkey, n, m, cap0, cap1, matches0, matches1,
                       scores0, scores1, go.k0, go.k1, go.cols, go.o0, go.o1, go.nmatch);
  } else {
    hipLaunchKernelGGL(mnn_finalize_kernel, dim3((unsigned)einx_cdiv(mx, 256), (unsigned)B), dim3(256), 0, s, a.rowkey, a.colkey, n, m,
                       cap0, cap1, matches0, matches1, scores0, scores1);
  }
  EINX_CHECK_LAUNCH();
  if (la) {
    hipLaunchKernelGGL(mnn_lse_kernel, dim3((unsigned)einx_cdiv(mx + 1, 256), (unsigned)B), dim3(256), 0, s, a);
    EINX_CHECK_LAUNCH();
    hipLaunchKernelGGL(mnn_la_apply_kernel, dim3((unsigned)einx_cdiv(cap1, 256), (unsigned)(cap0 < 32768 ? cap0 : 32768), (unsigned)B), dim3(256), 0, s, a);
    EINX_CHECK_LAUNCH();
  }
  return EINX_OK;
}
}  // namespace

EINX_EXPORT int einx_similarity(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B,
                                int D, float* sim, void* stream) {
  EINX_CHECK_ARG(desc0 && desc1 && n && m && sim, "null pointer");
  EINX_CHECK_ARG(B > 0 && cap0 > 0 && cap1 > 0 && D > 0 && D % 4 == 0, "bad shape (D must be a multiple of 4)");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sim, 0, (size_t)B * cap0 * cap1 * sizeof(float), s) != hipSuccess) {
    einx_set_error("einx_similarity: memset failed");
    return EINX_ERR_LAUNCH;
  }
  MnnArgs a{};
  a.d0 = desc0;
  a.d1 = desc1;
  a.n = n;
  a.m = m;
  a.cap0 = cap0;
  a.cap1 = cap1;
  a.D = D;
  a.la = sim;
  const dim3 grid((unsigned)einx_cdiv(cap1, BN), (unsigned)einx_cdiv(cap0, BM), (unsigned)B);
  hipLaunchKernelGGL(mnn_tile_kernel<3>, grid, dim3(THREADS), 0, s, a);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_gather_matches(const float* kpts0, const float* kpts1, const int64_t* matches0, const int32_t* n, int cap0, int cap1,
                                    int B, int cols, float* out0, float* out1, int32_t* nmatch, void* stream) {
  EINX_CHECK_ARG(kpts0 && kpts1 && matches0 && n && out0 && out1 && nmatch, "null pointer");
  EINX_CHECK_ARG(B > 0 && cap0 > 0 && cap1 > 0 && (cols == 2 || cols == 3), "bad shape");
  hipLaunchKernelGGL(gather_matches_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, kpts0, kpts1, matches0, n, cap0, cap1,
                     cols, out0, out1, nmatch);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_compact_rows(const float* src0, const float* src1, const int32_t* counts, int B, int cap, int width, float* dst0,
                                  float* dst1, void* stream) {
  EINX_CHECK_ARG(src0 && src1 && counts && dst0 && dst1, "null pointer");
  EINX_CHECK_ARG(B > 0 && cap > 0 && width > 0, "bad shape");
  hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, src0, src1, counts, B, cap, width, dst0, dst1);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
