// Shared helpers for the gfx950 kernels of libeinx_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/einx.h"
#include "../../include/einx_math.h"

#define EINX_EXPORT extern "C" __attribute__((visibility("default")))

// Timing-only ablations (switches that drop work to time what is left: WRONG results) live as patch files under
// tools/experiments/, not in these sources (round 5).  A build made from a patched tree must define EINX_TIMING_ONLY_BUILD:
// einx_build_flags() reports it and the Python package refuses to load such a library by default.

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
bool einx_profile_active();  // the per-launch scopes are recording: kernels are meant to be timed alone (no concurrent branches)

// Workgroups are dealt round-robin over the 8 XCDs in linear dispatch order (observed placement: speed only, never
// correctness), each XCD with its own L2.  xcd_contiguous() turns the linear workgroup id into a work-item id such that every
// XCD walks ONE contiguous range of work items: neighbouring tiles (shared halo rows, shared K/V blocks) then meet in one
// L2 instead of eight.  A bijection on [0, total) for every total.
__device__ __forceinline__ int xcd_contiguous(int linear, int total) {
  const int per = total >> 3;
  return linear < (per << 3) ? (linear & 7) * per + (linear >> 3) : linear;
}

// Content watch of a module's weights (common.hip::einx_params_hash): one wave hashes 64 evenly spread 32-bit words + the last
// word of tensor t; with `ref` set a difference raises `bit` in *flag.  Also run by spare workgroups of desc_sample_kernel, so
// that an extractor's watch costs no launch of its own (einx_extract).
struct EinxWatch {
  const int64_t* table;           // device [n][2]: (pointer, number of 32-bit words)
  const unsigned long long* ref;  // device [n] or null (store mode)
  unsigned long long* hash;       // device [n]: written in store mode, scratch otherwise
  int32_t* flag;
  int n, bit;
};
#ifdef __HIPCC__
// one 64-lane wave hashes EVERY word of table row t (round 5; round 4 sampled 65 words per tensor).  Rows are chunks of at most
// a few thousand words (the host cuts the tensors, einx.h: EINX_WATCH_CHUNK_WORDS), so the loads of a row are all in flight
// at once.  term(word, position) = t ^ (t >> 29) with t = (word + 1) * coef(position), coef = (2 position + 1) * odd 64-bit
// constant: every position has its own odd multiplier, so swapping two different words or a +d / -d edit of two words changes
// the wrap-around sum for certain ((w_i - w_j) (c_i - c_j) cannot vanish mod 2^64 for 32-bit words and < 2^12 positions), the
// xor-shift makes the sum non-linear in the words on top, and `word + 1` keeps zero words position-dependent too.  One 32 x 64
// bit multiply per word.  (Round 5 summed ((position << 32) + word) * constant, which only depended on the SUM of a row's
// words: a permutation inside a row or a sum-preserving edit went unseen, ADVICE r5.  A full two-multiply finaliser per word
// doubled the sampling launch the watch rides on: 0.05 -> 0.10 ms per step at B = 32.)
__device__ __forceinline__ unsigned long long einx_watch_term(unsigned long long pos, uint32_t word) {
  const unsigned long long coef = (2ull * pos + 1ull) * 0x9E3779B97F4A7C15ull;
  const unsigned long long t = ((unsigned long long)word + 1ull) * coef;
  return t ^ (t >> 29);
}
__device__ __forceinline__ void einx_watch_tensor(const EinxWatch& w, int t, int lane) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>((uintptr_t)w.table[2 * t]);
  const long long n = w.table[2 * t + 1];
  unsigned long long h = 0;
  long long i0 = 0;
  if ((((uintptr_t)p) & 15) == 0) {  // 16-byte loads, eight per lane in flight
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* p4 = reinterpret_cast<const u32x4*>(p);
    const long long n4 = n >> 2;
    for (long long j0 = 0; j0 < n4; j0 += 64 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long j = j0 + lane + 64 * u;
        v[u] = p4[j < n4 ? j : n4 - 1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long j = j0 + lane + 64 * u;
        if (j < n4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) h += einx_watch_term((unsigned long long)(4 * j + e), v[u][e]);
        }
      }
    }
    i0 = n4 << 2;
  }
  for (long long i = i0 + lane; i < n; i += 64) h += einx_watch_term((unsigned long long)i, p[i]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) h += __shfl_xor(h, off, 64);  // wrap-around sum: order independent
  if (lane == 0) {
    if (w.ref) {
      if (w.ref[t] != h) atomicOr(w.flag, w.bit);
    } else {
      w.hash[t] = h;
    }
  }
}
#endif
// einx_score_map that also zeroes the NMS pass flags of the detection that follows and writes the un-padded score map (crop: [B,1,H,W] or
// null), and einx_detect told so (einx_extract: one launch
// less per network on the latency-bound chain of a single-pair forward)
int einx_score_map_zero(const float* logits, int B, int C, int hc, int wc, const uint8_t* mask, int H, int W, int h0, int w0, int dilate,
                        int border, float* prob, float* score, int32_t* zero_ptr, int zero_n, float* crop, void* stream);
int32_t* einx_detect_flags(const einx_detect_params* p, void* ws, int* n);
// final_map != null: the cropped `nms` output is NOT written here; *final_map receives the buffer that holds the NMS fix-point (the caller
// crops it: einx_extract, on the sampling launch)
int einx_detect_prezeroed(const float* score, const einx_detect_params* p, void* ws, float* nms_out, float* positions, int32_t* indices,
                          int32_t* counts, float* thr, int32_t* not_converged, int flags_zeroed, const float** final_map, void* stream);
// The thresholded NMS map cropped to the un-padded window (the dict's `nms`): out[b,y,x] = v > thr[b] ? v : 0.  Riding on extra workgroups
// of the sampling launch inside einx_extract (both only need the selection's outputs): one launch less on a single pair's chain.
struct EinxCrop {
  const float* map;  // [B,Hp,Wp] the NMS fix-point; null: off
  const float* thr;  // [B]
  float* out;        // [B,H,W]
  int Hp, Wp, h0, w0, H, W;
};
#ifdef __HIPCC__
__device__ __forceinline__ void einx_crop_block(const EinxCrop& c, int b, int blk, int tid) {
  const int r = blk * 256 + tid;
  if (r >= c.H * c.W) return;
  const float v = c.map[((size_t)b * c.Hp + (r / c.W + c.h0)) * c.Wp + r % c.W + c.w0];
  c.out[(size_t)b * c.H * c.W + r] = v > c.thr[b] ? v : 0.0f;
}
#endif
// einx_desc_sample with the extractor's weight watch and the NMS-map crop riding on spare workgroups (einx_extract)
int einx_desc_sample_watch(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, int bilinear, int channels_last, const int32_t* indices,
                           const int32_t* counts, int cap, float scale, float* out, const EinxWatch& watch, const EinxCrop& crop, void* stream);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline __host__ __device__ int einx_cdiv(int a, int b) { return (a + b - 1) / b; }
