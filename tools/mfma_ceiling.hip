// Micro-benchmark (tuning aid, not part of the product): what fraction of the fp32 matrix peak does an
// MFMA loop reach (a) with operands in registers, (b) fed from LDS like conv_block_kernel (3 ds_read_b32
// per 2 MFMAs, prefetch distance PF), for 4 or 8 waves per workgroup.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NWAVES, int PF>
__global__ __launch_bounds__(NWAVES * 64) void k(float* out, int iters, int data) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  // data: 0 = small regular values, 1 = zeros, 2 = random mantissas in [-1,1) (operand toggling drives power -> clocks)
  for (int i = tid; i < 8192; i += NWAVES * 64) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    lds[i] = data == 0 ? (float)(i % 7) * 0.125f : data == 1 ? 0.0f : (float)(int)h * (1.0f / 2147483648.0f);
  }
  __syncthreads();
  f32x16 acc[2];
  for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
  float a = 1.0f + lane * 1e-3f, b0 = 0.5f, b1 = 0.25f;
  if (data == 1) a = b0 = b1 = 0.0f;
  if (data == 2) {
    a = lds[lane];
    b0 = lds[lane + 64];
    b1 = lds[lane + 128];
  }
  const int base = lane + (tid >> 6) * 64;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 36; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
      }
    } else {
      float av[PF + 1], bv0[PF + 1], bv1[PF + 1];
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        av[s] = lds[(base + s * 64) & 8191];
        bv0[s] = lds[(base + s * 96 + 2048) & 8191];
        bv1[s] = lds[(base + s * 96 + 4096) & 8191];
      }
#pragma unroll
      for (int s = 0; s < 36; ++s) {
        if (s + PF < 36) {
          av[(s + PF) % (PF + 1)] = lds[(base + (s + PF) * 64) & 8191];
          bv0[(s + PF) % (PF + 1)] = lds[(base + (s + PF) * 96 + 2048) & 8191];
          bv1[(s + PF) % (PF + 1)] = lds[(base + (s + PF) * 96 + 4096) & 8191];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % (PF + 1)], bv0[s % (PF + 1)], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % (PF + 1)], bv1[s % (PF + 1)], acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float t = 0.f;
  for (int r = 0; r < 16; ++r) t += acc[0][r] + acc[1][r];
  out[blockIdx.x * blockDim.x + tid] = t;
}

template <int MODE, int NWAVES, int PF>
void run(const char* name, int blocks_per_cu, int data = 0) {
  const int blocks = 256 * blocks_per_cu, iters = 2000;
  float* out;
  hipMalloc(&out, (size_t)blocks * NWAVES * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NWAVES, PF>), dim3(blocks), dim3(NWAVES * 64), 0, 0, out, 10, data);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NWAVES, PF>), dim3(blocks), dim3(NWAVES * 64), 0, 0, out, iters, data);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * NWAVES * iters * 72 * 4096.0;
  printf("%-44s %2d blocks/CU: %7.1f TFLOP/s (%.1f %% of 157.3)\n", name, blocks_per_cu, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
  hipFree(out);
}

int main() {
  run<0, 4, 1>("registers, 4 waves/WG", 1);
  run<0, 4, 1>("registers, 4 waves/WG", 2);
  run<0, 8, 1>("registers, 8 waves/WG", 2);
  run<1, 4, 1>("LDS fed, 4 waves/WG, PF=1", 2);
  run<1, 4, 2>("LDS fed, 4 waves/WG, PF=2", 2);
  run<1, 8, 1>("LDS fed, 8 waves/WG, PF=1", 2);
  run<1, 8, 2>("LDS fed, 8 waves/WG, PF=2", 2);
  run<1, 8, 3>("LDS fed, 8 waves/WG, PF=3", 2);
  run<1, 8, 2>("LDS fed, 8 waves/WG, PF=2", 1);
  run<0, 8, 1>("registers, 8 waves/WG, zeros", 2, 1);
  run<0, 8, 1>("registers, 8 waves/WG, random", 2, 2);
  run<1, 8, 2>("LDS fed, 8 waves/WG, PF=2, zeros", 2, 1);
  run<1, 8, 2>("LDS fed, 8 waves/WG, PF=2, random", 2, 2);
  run<1, 8, 2>("LDS fed, 8 waves/WG, PF=2, small values", 2, 0);
  return 0;
}
