// Shared helpers for the gfx950 kernels of libeinx_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/einx.h"
#include "../../include/einx_math.h"

#define EINX_EXPORT extern "C" __attribute__((visibility("default")))

// Experiment switches that drop work to time what is left (WRONG results) or insert delays only exist in builds that say
// so: -DEINX_TIMING_ONLY_BUILD, which einx_build_flags() reports and the Python package refuses to load by default.
#if (defined(EINX_GEMM_ABL) && EINX_GEMM_ABL != 0) || (defined(EINX_UPS_EXP) && EINX_UPS_EXP != 0) || defined(EINX_CONV_ABL_NOLOADS) || \
    (defined(EINX_CONV_STAGGER) && EINX_CONV_STAGGER != 0) || defined(VOX_EXP)
#ifndef EINX_TIMING_ONLY_BUILD
#error "EINX_GEMM_ABL / EINX_UPS_EXP / EINX_CONV_ABL_NOLOADS / EINX_CONV_STAGGER / VOX_EXP are timing-only ablations: add -DEINX_TIMING_ONLY_BUILD"
#endif
#endif

void einx_set_error(const char* fmt, ...);

#define EINX_CHECK_ARG(cond, msg)                 \
  do {                                            \
    if (!(cond)) {                                \
      einx_set_error("%s: %s", __func__, msg);    \
      return EINX_ERR_ARG;                        \
    }                                             \
  } while (0)

#define EINX_CHECK_LAUNCH()                                                   \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) {                                                   \
      einx_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
      return EINX_ERR_LAUNCH;                                                 \
    }                                                                         \
  } while (0)

// RAII timing scope around one kernel launch (or a group of launches) on `stream`; a no-op unless
// einx_profile_enable(1) was called.  Usage: `EinxProfScope prof("lg_gemm", stream);` before the launch.
class EinxProfScope {
 public:
  EinxProfScope(const char* name, hipStream_t s);
  ~EinxProfScope();

 private:
  hipStream_t stream_;
  int idx_;
  int gen_ = 0;
};

#define EINX_PROF(name, stream) EinxProfScope einx_prof_scope_(name, (hipStream_t)(stream))

// Workgroups are dealt round-robin over the 8 XCDs in linear dispatch order (observed placement: speed only, never
// correctness), each XCD with its own L2.  xcd_contiguous() turns the linear workgroup id into a work-item id such that every
// XCD walks ONE contiguous range of work items: neighbouring tiles (shared halo rows, shared K/V blocks) then meet in one
// L2 instead of eight.  A bijection on [0, total) for every total.
#ifndef EINX_NO_XCD_REMAP
__device__ __forceinline__ int xcd_contiguous(int linear, int total) {
  const int per = total >> 3;
  return linear < (per << 3) ? (linear & 7) * per + (linear >> 3) : linear;
}
#else
__device__ __forceinline__ int xcd_contiguous(int linear, int) { return linear; }
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline __host__ __device__ int einx_cdiv(int a, int b) { return (a + b - 1) / b; }
