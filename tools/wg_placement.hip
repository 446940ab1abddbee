// Micro-benchmark (tuning aid, not part of the product): where does the dispatcher put the workgroups of a 512-thread kernel
// that fits twice on a CU (the conv kernels' shape), and which wave slots do the two co-resident workgroups get?
// Prints, for the first workgroups in linear order: XCC id, SE / SH / CU id, SIMD and wave slot of wave 0, start time.
//   hipcc -O3 --offload-arch=gfx950 tools/wg_placement.hip -o tools/bin/wg_placement && tools/bin/wg_placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct Rec {
  unsigned hw_id, xcc, wave;
  unsigned long long t0, t1;
};

__global__ __launch_bounds__(512, 4) void k(Rec* out, int spin) {
  __shared__ float pad[14000];  // ~55 KB: two workgroups per CU by LDS, like the conv kernels by registers
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, all 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  pad[threadIdx.x] = v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    Rec r;
    r.hw_id = hw;
    r.xcc = xcc;
    r.wave = threadIdx.x >> 6;
    r.t0 = t0;
    r.t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 8 + (threadIdx.x >> 6)] = r;
  }
  if (pad[(threadIdx.x * 7) % 512] == 12345.f) out[0].hw_id = 0;
}

int main() {
  const int blocks = 1536;
  Rec* d;
  hipMalloc(&d, sizeof(Rec) * blocks * 8);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, 20000);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, 20000);
  hipDeviceSynchronize();
  Rec* h = (Rec*)malloc(sizeof(Rec) * blocks * 8);
  hipMemcpy(h, d, sizeof(Rec) * blocks * 8, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (int i = 0; i < blocks * 8; ++i)
    if (h[i].t0 < tmin) tmin = h[i].t0;
  printf("# block wave | xcc se sh cu simd slot | start end (x10 ns, from the first wave)\n");
  for (int b = 0; b < blocks; ++b) {
    if (!(b < 40 || (b >= 256 && b < 272) || (b >= 512 && b < 528) || (b >= 1024 && b < 1032))) continue;
    for (int w = 0; w < 8; w += (b < 4 ? 1 : 4)) {
      const Rec& r = h[b * 8 + w];
      printf("%5d %d | %u %u %u %2u %u %u | %6llu %6llu\n", b, w, r.xcc & 0xf, (r.hw_id >> 13) & 7, (r.hw_id >> 12) & 1, (r.hw_id >> 8) & 15, (r.hw_id >> 4) & 3,
             r.hw_id & 15, r.t0 - tmin, r.t1 - tmin);
    }
  }
  // which linear ids share a CU in the first generation (start within the first microsecond)?
  printf("# co-resident pairs of the first generation (same xcc/se/sh/cu, both started early):\n");
  int shown = 0;
  for (int a = 0; a < blocks && shown < 24; ++a)
    for (int b = a + 1; b < blocks && shown < 24; ++b) {
      const Rec &x = h[a * 8], &y = h[b * 8];
      if (x.t0 - tmin < 200 && y.t0 - tmin < 200 && (x.xcc & 0xf) == (y.xcc & 0xf) && ((x.hw_id >> 8) & 0xff) == ((y.hw_id >> 8) & 0xff)) {
        printf("  blocks %d and %d (slots %u / %u)\n", a, b, x.hw_id & 15, y.hw_id & 15);
        ++shown;
      }
    }
  return 0;
}
