// Error channel, version and device probing for libeinx_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "einx_common.h"

static thread_local char g_err[512] = "";

void einx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

EINX_EXPORT const char* einx_version(void) { return "einx-hip 0.6 (gfx950, ABI 6)"; }
EINX_EXPORT int einx_abi_version(void) { return EINX_ABI_VERSION; }
EINX_EXPORT const char* einx_last_error(void) { return g_err; }
EINX_EXPORT const char* einx_build_flags(void) {
#ifdef EINX_TIMING_ONLY_BUILD
  return "timing-only";
#else
  return "";
#endif
}

// ---- per-kernel-class timing with HIP events on the launch stream (measurement aid: bench.py's
// roofline_stages).  Off by default: a scope costs one relaxed load when disabled.
namespace {
struct ProfRec {
  const char* name;
  hipEvent_t e0, e1;
};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
volatile int g_prof_on = 0;
int g_prof_gen = 0;  // bumped whenever g_prof_recs is cleared: a scope that straddles einx_profile_enable must not touch a stranger's record
}  // namespace

EinxProfScope::EinxProfScope(const char* name, hipStream_t s) : stream_(s), idx_(-1) {
  if (!g_prof_on) return;
  ProfRec r;
  r.name = name;
  if (hipEventCreate(&r.e0) != hipSuccess) return;
  if (hipEventCreate(&r.e1) != hipSuccess) {
    (void)hipEventDestroy(r.e0);
    return;
  }
  (void)hipEventRecord(r.e0, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  gen_ = g_prof_gen;
  idx_ = (int)g_prof_recs.size();
  g_prof_recs.push_back(r);
}

EinxProfScope::~EinxProfScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (gen_ == g_prof_gen && idx_ < (int)g_prof_recs.size()) (void)hipEventRecord(g_prof_recs[idx_].e1, stream_);
}

bool einx_profile_active() { return g_prof_on != 0; }

EINX_EXPORT int einx_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof_recs.clear();
  ++g_prof_gen;
  g_prof_on = on ? 1 : 0;
  return EINX_OK;
}

EINX_EXPORT int einx_profile_report(char* buf, size_t cap) {
  EINX_CHECK_ARG(buf && cap > 0, "null buffer");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  std::map<std::string, std::pair<int, double>> agg;
  std::vector<std::string> order;
  for (auto& r : g_prof_recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    auto it = agg.find(r.name);
    if (it == agg.end()) {
      agg[r.name] = {1, (double)ms};
      order.push_back(r.name);
    } else {
      it->second.first += 1;
      it->second.second += ms;
    }
  }
  size_t off = 0;
  buf[0] = 0;
  for (auto& n : order) {
    const int w = snprintf(buf + off, cap - off, "%s %d %.6f\n", n.c_str(), agg[n].first, agg[n].second);
    if (w < 0 || (size_t)w >= cap - off) break;
    off += (size_t)w;
  }
  return EINX_OK;
}

EINX_EXPORT int einx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// four elements per thread, requested together (a grid-stride loop of one load per iteration waits for memory once per element)
__global__ __launch_bounds__(256) void einx_div_kernel(float* x, size_t n, float d) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const size_t i = base + (size_t)u * 256;
    v[u] = x[i < n ? i : n - 1];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const size_t i = base + (size_t)u * 256;
    if (i < n) x[i] = v[u] / d;
  }
}

EINX_EXPORT int einx_div_inplace(float* x, size_t n, float divisor, void* stream) {
  EINX_CHECK_ARG(x != nullptr || n == 0, "null tensor");
  if (n == 0) return EINX_OK;
  const size_t blocks = (n + 1023) / 1024;
  EINX_CHECK_ARG(blocks < (1ull << 31), "tensor too large");
  EINX_PROF("einx_div_kernel", (hipStream_t)stream);
  hipLaunchKernelGGL(einx_div_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, divisor);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}


// ------------------------------------------------------------------------------------------
// SuperPointv1's input handling for what the contiguous single-channel fast path (einx_extract's input_div) does not cover:
//   image /= 255.0                      in place on the CALLER's tensor, through its strides (superpoint_extractor.py:372)
//   if C == 3: rgb_to_grayscale(image)  kornia 0.7.1 (requirements.txt:56; absent here): (w_r r + w_g g) + w_b b with the fp32
//                                       weights (0.299, 0.587, 0.114), each product and sum rounded on its own (:375-376)
// One pass: every element is divided where it lies (the caller sees the scaled RGB / strided image afterwards, as with the
// reference) and the network's contiguous [B,1,H,W] input is written beside it.  A thread owns one pixel: four pixels of a row
// per thread when the rows are contiguous and 16-byte aligned would be faster, but this is 0.36 MB per image on a path the
// shipped pipelines (grayscale, contiguous) never take.
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void image_prepare_kernel(float* img, int C, int H, int W, long long sB, long long sC, long long sH, long long sW,
                                                            float d, float* gray, long long npix) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  const long long b = p / ((long long)W * H);
  float* q = img + b * sB + (long long)y * sH + (long long)x * sW;
  if (C == 1) {
    const float v = q[0] / d;
    q[0] = v;
    gray[p] = v;
  } else {
    const float r = q[0] / d, g = q[sC] / d, bl = q[2 * sC] / d;
    q[0] = r;
    q[sC] = g;
    q[2 * sC] = bl;
    gray[p] = (0.299f * r + 0.587f * g) + 0.114f * bl;  // -ffp-contract=off: three products, two sums, each rounded to fp32
  }
}
}  // namespace

EINX_EXPORT int einx_image_prepare(float* image, int B, int C, int H, int W, long long stride_b, long long stride_c, long long stride_h,
                                   long long stride_w, float divisor, float* gray, void* stream) {
  EINX_CHECK_ARG(image && gray, "null pointer");
  EINX_CHECK_ARG(C == 1 || C == 3, "1 (gray) or 3 (RGB) channels");
  EINX_CHECK_ARG(B > 0 && H > 0 && W > 0, "bad shape");
  EINX_CHECK_ARG(divisor != 0.0f, "divisor must not be zero");
  const long long npix = (long long)B * H * W;
  const long long blocks = (npix + 255) / 256;
  EINX_CHECK_ARG(blocks < (1ll << 31), "tensor too large");
  EINX_PROF("image_prepare_kernel", (hipStream_t)stream);
  hipLaunchKernelGGL(image_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, image, C, H, W, stride_b, stride_c, stride_h,
                     stride_w, divisor, gray, npix);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}


// ------------------------------------------------------------------------------------------
// The numeric contract on the device (tests): y[i] = f(x[i]) with the functions of include/einx_math.h as hipcc compiles them
// for gfx950 -- bit-equal to what gcc makes of the same header for the oracle (oracle/einx_oracle.c::orc_math_eval).
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void math_eval_kernel(int fn, const float* x, long long n, float* y) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float sn, cs, r = 0.0f;
  switch (fn) {
    case 0: r = einx_expf(x[i]); break;
    case 1: r = einx_logf(x[i]); break;
    case 2: einx_sincosf(x[i], &sn, &cs); r = sn; break;
    case 3: einx_sincosf(x[i], &sn, &cs); r = cs; break;
    case 4: r = einx_erff(x[i]); break;
    case 5: r = einx_sigmoidf(x[i]); break;
    case 6: r = einx_logsigmoidf(x[i]); break;
    case 7: r = einx_geluf(x[i]); break;
    default: r = einx_acosf(x[i]); break;
  }
  y[i] = r;
}
}  // namespace

EINX_EXPORT int einx_math_eval(int fn, const float* x, long long n, float* y, void* stream) {
  EINX_CHECK_ARG(x && y && n > 0 && fn >= 0 && fn <= 8, "bad arguments");
  const long long blocks = (n + 255) / 256;
  EINX_CHECK_ARG(blocks < (1ll << 31), "too many values");
  hipLaunchKernelGGL(math_eval_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, fn, x, n, y);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}


// ------------------------------------------------------------------------------------------
// Do two streams run side by side?  HIP deals streams onto a few hardware queues / compute pipes; two streams that share one
// serialise, and which ones share depends on what else the process created before (a process group's own streams, a loader's
// copy streams).  The probe holds one 64-lane wave spinning for `spin_us` on each stream between a common start and a common
// end event and returns the elapsed time: about spin_us when the streams overlap, about twice that when they do not.  The
// wave leaves its loop on the constant-rate clock or after a bounded number of polls, whichever comes first.
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void spin_kernel(long long ticks, int max_polls) {
  const long long t0 = wall_clock64();
  for (int i = 0; i < max_polls; ++i) {
    if (wall_clock64() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(32);
  }
}
}  // namespace

EINX_EXPORT int einx_stream_overlap_us(void* stream_a, void* stream_b, int spin_us, float* elapsed_us) {
  EINX_CHECK_ARG(elapsed_us && spin_us > 0 && spin_us <= 5000, "bad arguments (spin_us in 1..5000)");
  hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
  int rate_khz = 0;
  int dev = 0;
  EINX_CHECK_ARG(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess &&
                     rate_khz > 0,
                 "no wall clock rate");
  const long long ticks = (long long)spin_us * rate_khz / 1000;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  int rc = EINX_OK;
  float ms = 0.f;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e2) != hipSuccess ||
      hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) {
    einx_set_error("%s: hipEventCreate failed", __func__);
    rc = EINX_ERR_LAUNCH;
  } else {
    // a waits for nothing; b starts after e0 (recorded on a); a ends after b's wave
    const bool ok = hipEventRecord(e0, a) == hipSuccess && hipStreamWaitEvent(b, e0, 0) == hipSuccess;
    if (ok) {
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks, 1 << 20);
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, ticks, 1 << 20);
    }
    if (!ok || hipGetLastError() != hipSuccess || hipEventRecord(e1, b) != hipSuccess || hipStreamWaitEvent(a, e1, 0) != hipSuccess ||
        hipEventRecord(e2, a) != hipSuccess || hipEventSynchronize(e2) != hipSuccess || hipEventElapsedTime(&ms, e0, e2) != hipSuccess)
    {
      einx_set_error("%s: probe failed: %s", __func__, hipGetErrorString(hipGetLastError()));
      rc = EINX_ERR_LAUNCH;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (e2) (void)hipEventDestroy(e2);
  if (rc == EINX_OK) *elapsed_us = ms * 1000.f;
  return rc;
}


// ------------------------------------------------------------------------------------------
// Content watch of a module's weights (round 4).  The reference's modules are plain nn.Modules: an in-place edit of a
// weight through `p.data` takes effect at the next forward.  Here weights are repacked / folded into kernel-native images,
// and `.data` edits do not move the version counters the host-side cache keys on.  One 64-lane wave per table row hashes every
// word of the row (einx_watch_tensor; the host cuts tensors into rows of EINX_WATCH_CHUNK_WORDS); mode "store" (ref == NULL)
// records the hashes at pack time, mode "compare" raises *stale when a row's hash differs -- read back with the keypoint
// counts the forward fetches anyway.  Exact since round 5 (round 4 sampled 65 words per tensor and missed sparse edits).
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void params_hash_kernel(const EinxWatch w) { einx_watch_tensor(w, (int)blockIdx.x, (int)threadIdx.x); }
}  // namespace

EINX_EXPORT int einx_params_hash(const int64_t* table, int n, uint64_t* hash, const uint64_t* ref, int32_t* stale, void* stream) {
  EINX_CHECK_ARG(table && hash && n > 0, "null pointer / empty table");
  EINX_CHECK_ARG((ref == nullptr) == (stale == nullptr), "ref and stale go together");
  EinxWatch w;
  w.table = table;
  w.ref = (const unsigned long long*)ref;
  w.hash = (unsigned long long*)hash;
  w.flag = stale;
  w.n = n;
  w.bit = 1;
  hipLaunchKernelGGL(params_hash_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, w);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
