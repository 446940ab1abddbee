// Micro-benchmark (tuning aid, not part of the product): can the fp32 VECTOR pipe (v_fma_f32 chains, the same
// k-ordered fmaf arithmetic the oracle uses) run BESIDE the fp32 MATRIX pipe on one SIMD, and what does the sum
// reach under the power limit?  8 waves per workgroup (two per SIMD): waves with role MFMA run the
// v_mfma_f32_32x32x2_f32 loop of mfma_ceiling.hip, waves with role VALU run 64 independent v_fma_f32 chains.
//   mode 0: all 8 waves MFMA      mode 1: all 8 waves VALU
//   mode 2: waves 0-3 MFMA, 4-7 VALU (one of each per SIMD)    mode 3: waves 0-3 MFMA only (4-7 exit)
//   mode 4: waves 4-7 VALU only (0-3 exit)
//   hipcc -O3 --offload-arch=gfx950 tools/coexec_ceiling.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float rnd(unsigned h) {
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  return (float)(int)h * (1.0f / 2147483648.0f);
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int data) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool do_mfma = MODE == 0 || ((MODE == 2 || MODE == 3) && wave < 4);
  const bool do_valu = MODE == 1 || ((MODE == 2 || MODE == 4) && wave >= 4);
  float t = 0.f;
  if (do_mfma) {
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    float a = data ? rnd(tid * 2654435761u + 1) : 1.0f + lane * 1e-3f;
    float b0 = data ? rnd(tid * 40503u + 7) : 0.5f, b1 = data ? rnd(tid * 9973u + 3) : 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 32; ++s) {  // 64 MFMAs = 64 x 4096 flop... (32x32x2 x 2 flop) per lane-wave
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
      }
    }
    for (int r = 0; r < 16; ++r) t += acc[0][r] + acc[1][r];
  } else if (do_valu) {
    float acc[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) acc[r] = 0.f;
    float x = data ? rnd(tid * 2654435761u + 11) : 1.0f + lane * 1e-3f;
    // weights: wave-uniform values (SGPR operands in the real kernel); here derived from a uniform seed
    float w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = data ? rnd(blockIdx.x * 977u + r * 31u + 5) : 0.001f * (r + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 32; ++s) {  // 32 x 64 v_fma_f32 = the same 64 x 4096... see flop accounting in run()
#pragma unroll
        for (int r = 0; r < 64; ++r) acc[r] = __builtin_fmaf(w[(r + s) & 15], x, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 64; ++r) t += acc[r];
  } else {
    return;
  }
  out[blockIdx.x * blockDim.x + tid] = t;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, int data) {
  const int blocks = 256 * blocks_per_cu, iters = 1500;
  float* out;
  hipMalloc(&out, (size_t)blocks * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, out, 10, data);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, out, iters, data);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per iteration: an MFMA wave does 64 MFMAs x 4096 flop; a VALU wave does 32 x 64 fma x 64 lanes x 2 flop = the same 262144
  const int nm = MODE == 0 ? 8 : (MODE == 2 || MODE == 3) ? 4 : 0;
  const int nv = MODE == 1 ? 8 : (MODE == 2 || MODE == 4) ? 4 : 0;
  const double per = 64.0 * 4096.0;
  const double fm = (double)blocks * nm * iters * per, fv = (double)blocks * nv * iters * per;
  printf("%-52s %d WG/CU data=%d: %7.3f ms  matrix %6.1f + vector %6.1f = %6.1f TFLOP/s\n", name, blocks_per_cu, data, ms, fm / ms / 1e9,
         fv / ms / 1e9, (fm + fv) / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int data = 0; data < 2; ++data) {
    run<0>("8 MFMA waves", 1, data);
    run<1>("8 VALU waves", 1, data);
    run<3>("4 MFMA waves (one per SIMD), others exit", 1, data);
    run<4>("4 VALU waves (one per SIMD), others exit", 1, data);
    run<2>("4 MFMA + 4 VALU waves (one of each per SIMD)", 1, data);
    run<2>("4 MFMA + 4 VALU waves (one of each per SIMD)", 2, data);
  }
  return 0;
}
