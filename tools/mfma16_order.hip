// Does v_mfma_f32_16x16x4_f32 accumulate its four K products as the sequential chain fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))?
// (needed before a 16x16x4 tile can replace 32x32x2 in the k-ordered conv kernels).  Prints which candidate order matches bit for bit.
//   hipcc -O2 --offload-arch=gfx950 tools/mfma16_order.hip -o tools/bin/mfma16_order && tools/bin/mfma16_order
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// lane l: A operand = A[row = l % 16][k = l / 16], B operand = B[k = l / 16][col = l % 16]; C[4 * (l / 16) + i][l % 16] in c[i]
__global__ void k(const float* A, const float* B, const float* C0, float* C, int reps) {
  const int l = threadIdx.x;
  f32x4 c;
  for (int i = 0; i < 4; ++i) c[i] = C0[(4 * (l / 16) + i) * 16 + (l % 16)];
  for (int r = 0; r < reps; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(r * 16 + l % 16) * 4 + l / 16], B[(r * 4 + l / 16) * 16 + l % 16], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) C[(4 * (l / 16) + i) * 16 + (l % 16)] = c[i];
}
int main() {
  const int reps = 5;
  float hA[reps * 64], hB[reps * 64], hC0[256], hC[256];
  srand(7);
  auto rnd = [] { return (float)((double)rand() / RAND_MAX * 2.0 - 1.0) * (1.0f + (rand() % 1000) * 1e-3f); };
  for (auto& v : hA) v = rnd();
  for (auto& v : hB) v = rnd();
  for (auto& v : hC0) v = rnd();
  float *dA, *dB, *dC0, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC0, sizeof hC0); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC0, hC0, sizeof hC0, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC0, dC, reps);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  const int orders[6][4] = {{0, 1, 2, 3}, {3, 2, 1, 0}, {0, 2, 1, 3}, {1, 0, 3, 2}, {2, 3, 0, 1}, {0, 1, 3, 2}};
  for (int o = 0; o < 6; ++o) {
    int bad = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        float acc = hC0[i * 16 + j];
        for (int r = 0; r < reps; ++r)
          for (int t = 0; t < 4; ++t) {
            const int kk = orders[o][t];
            acc = fmaf(hA[(r * 16 + i) * 4 + kk], hB[(r * 4 + kk) * 16 + j], acc);
          }
        bad += memcmp(&acc, &hC[i * 16 + j], 4) != 0;
      }
    printf("order %d%d%d%d: %d of 256 outputs differ\n", orders[o][0], orders[o][1], orders[o][2], orders[o][3], bad);
  }
  // pairwise-tree candidate: (a0b0 + a1b1) + (a2b2 + a3b3) + c with exact products?  just report max abs diff vs sequential for information
  return 0;
}
