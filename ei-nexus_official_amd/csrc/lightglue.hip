// lightglue.hip -- placeholder translation unit; kernels are added below in this round.
#include "einx_common.h"
EINX_EXPORT size_t einx_lg_ws_bytes(int B, int cap0, int cap1, int d, int input_dim) { return 0; }
EINX_EXPORT int einx_lightglue(const einx_lg_weights* w, const float* kpts0, const float* desc0, const int32_t* n, int cap0,
                               const float* kpts1, const float* desc1, const int32_t* m, int cap1, int B, float h0, float w0, float h1,
                               float w1, void* ws, int64_t* matches0, int64_t* matches1, float* scores0, float* scores1, float* la,
                               float* ref0, float* ref1, void* stream) {
  einx_set_error("einx_lightglue: not built yet");
  return EINX_ERR_ARG;
}
