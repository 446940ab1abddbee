// Error channel, version and device probing for libeinx_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "einx_common.h"

static thread_local char g_err[512] = "";

void einx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

EINX_EXPORT const char* einx_version(void) { return "einx-hip 0.1 (gfx950)"; }
EINX_EXPORT const char* einx_last_error(void) { return g_err; }

EINX_EXPORT int einx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

__global__ void einx_div_kernel(float* x, size_t n, float d) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) x[i] = x[i] / d;
}

EINX_EXPORT int einx_div_inplace(float* x, size_t n, float divisor, void* stream) {
  EINX_CHECK_ARG(x != nullptr || n == 0, "null tensor");
  if (n == 0) return EINX_OK;
  const int threads = 256;
  size_t blocks = (n + threads - 1) / threads;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(einx_div_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, x, n, divisor);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
