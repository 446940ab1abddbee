// Micro-benchmark (tuning aid, round 3): HBM write rate of candidate store schedules for the dense-descriptor tensor
// [B,D,H,W] = [32,256,260,346] fp32 (2.95 GB), to choose the work decomposition of upsample_store_kernel BEFORE building it.
//   hipcc -O3 --offload-arch=gfx950 tools/store_pattern2.hip -o tools/bin/store_pattern2 && tools/bin/store_pattern2
// A "run" = the ROWS x W floats of one band of one channel = one contiguous piece of the output.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int B = 32, D = 256, H = 260, W = 346, ROWS = 8;
constexpr int NB = (H + ROWS - 1) / ROWS;  // 33 bands

__device__ __forceinline__ void write_run(float* o, int len, int lane, int nl, float v) {
  const int head = (int)((16 - ((size_t)o & 15)) & 15) / 4;
  if (lane < head) o[lane] = v;
  const int n4 = (len - head) / 4;
  float4* o4 = reinterpret_cast<float4*>(o + head);
  for (int i = lane; i < n4; i += nl) o4[i] = make_float4(v, v, v, v);
  const int t0 = head + n4 * 4;
  if (lane < len - t0) o[t0 + lane] = v;
}

// as write_run, but the float4 body starts on a 128-byte boundary (head of up to 31 floats as dword stores): every
// wave-instruction of the body then covers 8 whole cache lines
__device__ __forceinline__ void write_run128(float* o, int len, int lane, float v) {
  int head = (int)((128 - ((size_t)o & 127)) & 127) / 4;
  if (head > len) head = len;
  if (lane < head) o[lane] = v;
  const int n4 = (len - head) / 4;
  float4* o4 = reinterpret_cast<float4*>(o + head);
  for (int i = lane; i < n4; i += 64) o4[i] = make_float4(v, v, v, v);
  const int t0 = head + n4 * 4;
  if (lane < len - t0) o[t0 + lane] = v;
}
template <int NWAVES>
__global__ void k_wave_run128(float* out, int cc) {
  const int ng = D / cc;
  int id = blockIdx.x;
  const int g = id % ng; id /= ng; const int j = id % NB; const int b = id / NB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int y0 = j * ROWS;
  const int rows = y0 + ROWS <= H ? ROWS : H - y0;
  for (int c = g * cc + wave; c < (g + 1) * cc; c += NWAVES)
    write_run128(out + (((size_t)b * D + c) * H + y0) * W, rows * W, lane, (float)c);
}

// wave per run; workgroup of NWAVES waves = (image, band, group of CC channels), waves stride the channels.
// ORDER 0: linear block id -> (band fastest, channel group, image); 1: (channel group fastest, band, image)
template <int NWAVES, int ORDER>
__global__ void k_wave_run(float* out, int cc) {
  const int ng = D / cc;
  int id = blockIdx.x, j, g, b;
  if (ORDER == 0) {
    j = id % NB; id /= NB; g = id % ng; b = id / ng;
  } else {
    g = id % ng; id /= ng; j = id % NB; b = id / NB;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int y0 = j * ROWS;
  const int rows = y0 + ROWS <= H ? ROWS : H - y0;
  for (int c = g * cc + wave; c < (g + 1) * cc; c += NWAVES)
    write_run(out + (((size_t)b * D + c) * H + y0) * W, rows * W, lane, 64, (float)c);
}

// workgroup per channel plane (360 KB contiguous), waves take bands round-robin
template <int NWAVES>
__global__ void k_plane(float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* o = out + (size_t)blockIdx.x * H * W;
  for (int j = wave; j < NB; j += NWAVES) {
    const int y0 = j * ROWS;
    const int rows = y0 + ROWS <= H ? ROWS : H - y0;
    write_run(o + (size_t)y0 * W, rows * W, lane, 64, (float)j);
  }
}

// direct row stores: workgroup of 384 threads spans a row, thread = column, 8 rows x CC channels (one dword per lane per store)
template <int ORDER>
__global__ void k_rows(float* out, int cc) {
  const int ng = D / cc;
  int id = blockIdx.x, j, g, b;
  if (ORDER == 0) {
    j = id % NB; id /= NB; g = id % ng; b = id / ng;
  } else {
    g = id % ng; id /= ng; j = id % NB; b = id / NB;
  }
  const int x = threadIdx.x;
  if (x >= W) return;
  const int y0 = j * ROWS;
  for (int c = g * cc; c < (g + 1) * cc; ++c) {
    float* o = out + (((size_t)b * D + c) * H + y0) * W + x;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
      if (y0 + r < H) o[(size_t)r * W] = (float)(c + r);
  }
}

// direct row stores, wave = (channel, band): 6 column sweeps x 8 rows of dword stores, no LDS staging
template <int NWAVES, int ORDER>
__global__ void k_wave_rows(float* out, int cc) {
  const int ng = D / cc;
  int id = blockIdx.x, j, g, b;
  if (ORDER == 0) {
    j = id % NB; id /= NB; g = id % ng; b = id / ng;
  } else {
    g = id % ng; id /= ng; j = id % NB; b = id / NB;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int y0 = j * ROWS;
  for (int c = g * cc + wave; c < (g + 1) * cc; c += NWAVES) {
    float* o = out + (((size_t)b * D + c) * H + y0) * W;
    for (int x = lane; x < W; x += 64) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
        if (y0 + r < H) o[(size_t)r * W + x] = (float)(c + r);
    }
  }
}

__global__ void k_fill(float* out, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) *reinterpret_cast<float4*>(out + i) = make_float4(1, 2, 3, 4);
}

template <typename F>
void run(const char* name, F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  printf("%-64s %8.0f us  %5.2f TB/s\n", name, ms * 1e3, (double)B * D * H * W * 4 / ms / 1e9);
  fflush(stdout);
}

int main() {
  const size_t n = (size_t)B * D * H * W;
  float* out;
  if (hipMalloc(&out, n * 4) != hipSuccess) return 1;
  run("sequential float4 fill", [&] { hipLaunchKernelGGL(k_fill, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, out, n); });
  for (int cc : {256, 64, 32, 16}) {
    char nm[96];
    const unsigned grid = (unsigned)(NB * (D / cc) * B);
    snprintf(nm, sizeof nm, "wave per run, 4 waves, %3d ch/WG, band fastest", cc);
    run(nm, [&] { hipLaunchKernelGGL((k_wave_run<4, 0>), dim3(grid), dim3(256), 0, 0, out, cc); });
    snprintf(nm, sizeof nm, "wave per run, 4 waves, %3d ch/WG, channel group fastest", cc);
    run(nm, [&] { hipLaunchKernelGGL((k_wave_run<4, 1>), dim3(grid), dim3(256), 0, 0, out, cc); });
    snprintf(nm, sizeof nm, "wave per run, 8 waves, %3d ch/WG, band fastest", cc);
    run(nm, [&] { hipLaunchKernelGGL((k_wave_run<8, 0>), dim3(grid), dim3(512), 0, 0, out, cc); });
    snprintf(nm, sizeof nm, "row stores (384 thr, dword), %3d ch/WG, band fastest", cc);
    run(nm, [&] { hipLaunchKernelGGL((k_rows<0>), dim3(grid), dim3(384), 0, 0, out, cc); });
    snprintf(nm, sizeof nm, "row stores (384 thr, dword), %3d ch/WG, channel group fastest", cc);
    run(nm, [&] { hipLaunchKernelGGL((k_rows<1>), dim3(grid), dim3(384), 0, 0, out, cc); });
    snprintf(nm, sizeof nm, "wave rows (dword, no staging), 4 waves, %3d ch/WG, band fastest", cc);
    run(nm, [&] { hipLaunchKernelGGL((k_wave_rows<4, 0>), dim3(grid), dim3(256), 0, 0, out, cc); });
  }
  for (int cc : {32, 16}) {
    char nm[96];
    const unsigned grid = (unsigned)(NB * (D / cc) * B);
    snprintf(nm, sizeof nm, "wave per run, 128-byte aligned body, 4 waves, %3d ch/WG, cg fastest", cc);
    run(nm, [&] { hipLaunchKernelGGL((k_wave_run128<4>), dim3(grid), dim3(256), 0, 0, out, cc); });
    snprintf(nm, sizeof nm, "wave per run, 16-byte aligned body,  4 waves, %3d ch/WG, cg fastest", cc);
    run(nm, [&] { hipLaunchKernelGGL((k_wave_run<4, 1>), dim3(grid), dim3(256), 0, 0, out, cc); });
  }
  run("workgroup per channel plane, 4 waves", [&] { hipLaunchKernelGGL((k_plane<4>), dim3(B * D), dim3(256), 0, 0, out); });
  run("workgroup per channel plane, 8 waves", [&] { hipLaunchKernelGGL((k_plane<8>), dim3(B * D), dim3(512), 0, 0, out); });
  run("workgroup per channel plane, 16 waves", [&] { hipLaunchKernelGGL((k_plane<16>), dim3(B * D), dim3(1024), 0, 0, out); });
  return 0;
}
