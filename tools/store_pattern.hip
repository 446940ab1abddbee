// Micro-benchmark (tuning aid): HBM write rate of the dense-descriptor store pattern [B,D,H,W] = [32,256,260,346]
// written as 256-byte row segments, for different ways of walking channels / rows.
//   hipcc -O3 --offload-arch=gfx950 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int B = 32, D = 256, H = 260, ROWS = 8;
#ifndef WIDTH
#define WIDTH 346
#endif
constexpr int W = WIDTH;

// mode 0: block = COLS threads, grid (cdiv(W,COLS), cdiv(H,ROWS), B); loops channels [c0, c0+DC) with rows inner
template <int COLS>
__global__ void k_band(float* out, int dc) {
  const int x = blockIdx.x * COLS + threadIdx.x;
  const int y0 = blockIdx.y * ROWS;
  const int b = blockIdx.z / (D / dc), c0 = (blockIdx.z % (D / dc)) * dc;
  if (x >= W) return;
  for (int c = c0; c < c0 + dc; ++c) {
    float* o = out + (((size_t)b * D + c) * H + y0) * W + x;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
      if (y0 + r < H) o[(size_t)r * W] = (float)(c + r);
  }
}
// mode 2: a band of ROWS rows of one channel is one contiguous run of ROWS*W floats; the block writes it with
// 16-byte stores aligned to 16 bytes (scalar head/tail), 1 KB per wave instruction
__global__ void k_run16(float* out, int dc) {
  const int y0 = blockIdx.y * ROWS;
  const int b = blockIdx.z / (D / dc), c0 = (blockIdx.z % (D / dc)) * dc;
  const int rows = y0 + ROWS <= H ? ROWS : H - y0;
  const int len = rows * W;
  for (int c = c0; c < c0 + dc; ++c) {
    float* o = out + (((size_t)b * D + c) * H + y0) * W;
    const int head = (int)((16 - ((size_t)o & 15)) & 15) / 4;  // floats until 16-byte alignment
    if ((int)threadIdx.x < head) o[threadIdx.x] = 1.0f;
    const int n4 = (len - head) / 4;
    float4* o4 = reinterpret_cast<float4*>(o + head);
    for (int i = threadIdx.x; i < n4; i += blockDim.x) o4[i] = make_float4(1, 2, 3, 4);
    const int tail0 = head + n4 * 4;
    if ((int)threadIdx.x < len - tail0) o[tail0 + threadIdx.x] = 2.0f;
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
  printf("%-44s %8.0f us  %5.2f TB/s\n", name, ms * 1e3, (double)B * D * H * W * 4 / ms / 1e9);
}
int main() {
  const size_t n = (size_t)B * D * H * W;
  float* out;
  hipMalloc(&out, n * 4);
  run("sequential float4 fill", [&] { hipLaunchKernelGGL(k_fill, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, out, n); });
  run("band, 64 cols, all 256 channels per wave", [&] { hipLaunchKernelGGL(k_band<64>, dim3((W + 63) / 64, (H + ROWS - 1) / ROWS, B), dim3(64), 0, 0, out, 256); });
  run("band, 384 cols, all 256 channels", [&] { hipLaunchKernelGGL(k_band<384>, dim3(1, (H + ROWS - 1) / ROWS, B), dim3(384), 0, 0, out, 256); });
  run("band, 384 cols, 64 channels per block", [&] { hipLaunchKernelGGL(k_band<384>, dim3(1, (H + ROWS - 1) / ROWS, B * 4), dim3(384), 0, 0, out, 64); });
  run("band, 384 cols, 16 channels per block", [&] { hipLaunchKernelGGL(k_band<384>, dim3(1, (H + ROWS - 1) / ROWS, B * 16), dim3(384), 0, 0, out, 16); });
  run("band, 64 cols, 16 channels per block", [&] { hipLaunchKernelGGL(k_band<64>, dim3((W + 63) / 64, (H + ROWS - 1) / ROWS, B * 16), dim3(64), 0, 0, out, 16); });
  run("contiguous band runs, float4, 256 threads, 256 ch", [&] { hipLaunchKernelGGL(k_run16, dim3(1, (H + ROWS - 1) / ROWS, B), dim3(256), 0, 0, out, 256); });
  run("contiguous band runs, float4, 256 threads, 16 ch", [&] { hipLaunchKernelGGL(k_run16, dim3(1, (H + ROWS - 1) / ROWS, B * 16), dim3(256), 0, 0, out, 16); });
  return 0;
}
