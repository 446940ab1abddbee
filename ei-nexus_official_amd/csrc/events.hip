// events.hip -- the step BEFORE the extract+match path: raw events
// (x, y, t, p) -> voxel-grid representation and the events mask, on gfx950.
//
// Replaces (reference file:line): datasets/representations.py:8-21 (time_normalization),
// :67-124 (events_to_voxel_grid: trilinear scatter-add + non-zero mean/std normalisation),
// datasets/visualize.py:23-50 (draw_events_accumulation_image) with the `> 0` mask of
// test_events-image_same-time.py:137.
//
// Scatter-add (round 4: DETERMINISTIC).  The reference's eight put_(accumulate=True) calls add, per voxel, the contributions
// of corner (dx,dy,dt) = (0,0,0) of all events in event order, then corner (0,0,1) ... (1,1,1) -- torch's serial path (and the
// order oracle/einx_oracle.c::orc_voxel_grid restates; with >= 32768 events and several threads torch itself switches to
// unordered atomic adds).  The kernels below reproduce exactly that order without any float atomic: every voxel is owned
// by ONE wave, which walks its events in order and resolves the collisions inside a 64-event batch in lane order.  Two runs
// are bit-identical and `raw` is bit-equal to the oracle at every size.  The count image uses integer atomics (exact).
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>
#include <string.h>

#include "einx_common.h"

namespace {

constexpr int VOX_BANDS = 16;   // column bands = waves of a scatter workgroup (each voxel column has ONE owner wave)
constexpr int VOX_SEGS = 16;    // event segments = waves of a scatter workgroup (each wave sweeps one contiguous share)
constexpr int VOX_TAGS = 128;   // per-wave collision tags
constexpr int VOX_CACHE = 4;    // 64-event chunks of a band list kept in registers over the eight corners

// All samples of a batch go through ONE launch per stage: blockIdx.y is the sample, the device copy of the
// offsets array (one small host-to-device copy per call) delimits its events.
struct VoxArgs {
  const float* x;
  const float* y;
  const double* t;
  const float* p;
  const int64_t* offs;  // device [B+1]
  int bins, H, W;
  int rows, nslab, bw;  // rows per slab, slabs per sample, columns per band
  float* grid;          // [B,bins,H,W]
  float4* rec;          // [N] (xf, yf, t_norm, value) per event
  uint32_t* keys;       // [N] (y0 + 2) << 16 | (x0 + 2), both clamped to [0, 65535]: what the sweep reads instead of x and y
  int32_t* counts;      // [B][nslab][VOX_BANDS][VOX_SEGS] list lengths
  uint32_t* lists;      // [4 N] event indices (in-sample): per slab, per band, per segment, in event order
  double* part;         // [B][nslab][3] statistics of the non-zero voxels of a slab
};

__device__ __forceinline__ uint32_t vox_key(int x0, int y0) {
  const int xc = min(max(x0, -2), 65533) + 2, yc = min(max(y0, -2), 65533) + 2;
  return ((uint32_t)yc << 16) | (uint32_t)xc;
}
__device__ __forceinline__ int vox_key_x(uint32_t k) { return (int)(k & 0xFFFFu) - 2; }
__device__ __forceinline__ int vox_key_y(uint32_t k) { return (int)(k >> 16) - 2; }
// events per segment: a sample's events are split into VOX_SEGS contiguous shares of whole 512-event trips
__device__ __forceinline__ long long vox_seg(long long n) { return ((n + VOX_SEGS * 512 - 1) / (VOX_SEGS * 512)) * 512; }

// which slabs / bands an event can touch (its rows y0, y0+1 and columns x0, x0+1); the SAME tests size the lists
// (voxel_prep_kernel) and fill them (voxel_scatter_kernel)
__device__ __forceinline__ bool vox_hits_slab(int y0, int r0, int nr) { return y0 + 1 >= r0 && y0 < r0 + nr; }
__device__ __forceinline__ bool vox_hits_band(int x0, int band, int bw, int W) {
  return (x0 >= 0 && x0 < W && x0 / bw == band) || (x0 + 1 >= 0 && x0 + 1 < W && (x0 + 1) / bw == band);
}

// Per event: the normalised record (time_normalization in float64 like numpy, then float32 like torch -- one double
// division per event instead of one per slab that sweeps it), the packed cell key, and the lengths of the
// (slab, band, segment) lists it will join.  One workgroup per (segment, sample): the histogram lives in LDS (integer
// atomics: exact) and leaves with plain stores -- no global atomic, no memset of the counts.
constexpr int VOX_PREP_BINS = 4096;  // nslab * VOX_BANDS that fit the LDS histogram; beyond that: global atomics
template <bool LDS_HIST>
__global__ __launch_bounds__(1024) void voxel_prep_kernel(const VoxArgs a) {
  __shared__ int hist[LDS_HIST ? VOX_PREP_BINS : 1];
  const int b = blockIdx.y, seg = blockIdx.x;
  const long long o0 = a.offs[b], n = a.offs[b + 1] - o0;
  const int nb = a.nslab * VOX_BANDS;
  int32_t* counts = a.counts + (size_t)b * nb * VOX_SEGS;
  if (LDS_HIST) {
    for (int j = threadIdx.x; j < nb; j += 1024) hist[j] = 0;
    __syncthreads();
  }
  if (n > 0) {
    const long long sg = vox_seg(n), lo = seg * sg, hi = min(n, lo + sg);
    const double* t = a.t + o0;
    const double t0d = t[0], tld = t[n - 1];
    const double den = (tld - t0d) + 1e-8;
    const float tf0 = (float)(0.0 / den);
    const float tfl = (float)((tld - t0d) / den);
    for (long long i = lo + threadIdx.x; i < hi; i += 1024) {
      const float tf = (float)((t[i] - t0d) / den);
      const float tn = ((float)(a.bins - 1) * (tf - tf0)) / (tfl - tf0);
      const float xf = a.x[o0 + i], yf = a.y[o0 + i];
      float value = a.p[o0 + i];
      if (value < 1.0f) value = -1.0f;
      a.rec[o0 + i] = make_float4(xf, yf, tn, value);
      const uint32_t key = vox_key((int)xf, (int)yf);  // .int() truncates toward zero
      a.keys[o0 + i] = key;
      const int x0 = vox_key_x(key), y0 = vox_key_y(key);
      // rows y0, y0+1 lie in at most two neighbouring slabs, columns x0, x0+1 in at most two neighbouring bands
      const int s_lo = max(0, (y0 < 0 ? -1 : y0 / a.rows)), s_hi = min(a.nslab - 1, (y0 + 1 < 0 ? -1 : (y0 + 1) / a.rows));
      for (int sl = s_lo; sl <= s_hi; ++sl) {
        const int r0 = sl * a.rows, nr = min(a.rows, a.H - r0);
        if (!vox_hits_slab(y0, r0, nr)) continue;
        const int b_lo = max(0, (x0 < 0 ? -1 : x0 / a.bw)), b_hi = min(VOX_BANDS - 1, (x0 + 1 < 0 ? -1 : (x0 + 1) / a.bw));
        for (int bd = b_lo; bd <= b_hi; ++bd)
          if (vox_hits_band(x0, bd, a.bw, a.W)) {
            if (LDS_HIST) atomicAdd(&hist[sl * VOX_BANDS + bd], 1);
            else atomicAdd(&counts[(sl * VOX_BANDS + bd) * VOX_SEGS + seg], 1);
          }
      }
    }
  }
  if (LDS_HIST) {
    __syncthreads();
    for (int j = threadIdx.x; j < nb; j += 1024) counts[j * VOX_SEGS + seg] = hist[j];
  }
}

// Appends up to 64 hits (one per lane, in lane = event order) to the (band, segment) sub-lists they touch.  `curv`: lane v holds
// the cursor of band v.  ONE copy of this code (noinline): inlined at its call sites (and, in its first form, unrolled over the
// 16 bands) the kernel grew to 72 KB of instructions and no longer fitted the instruction cache.
__device__ __noinline__ void vox_distribute(uint32_t ev, bool valid, const uint32_t* keys, uint32_t* lists, const int* sub_base, int seg,
                                            int bw, int W, int& curv, volatile int* vcnt /* 16 ints of this wave */) {
  const int lane = threadIdx.x & 63;
  const int x0 = vox_key_x(keys[valid ? ev : 0]);
  const int b0 = (valid && x0 >= 0 && x0 < W) ? x0 / bw : -1;
  int b1 = (valid && x0 + 1 >= 0 && x0 + 1 < W) ? (x0 + 1) / bw : -1;
  if (b1 == b0) b1 = -1;
  // rank of a lane among the lanes of ITS band from four ballots of the band's bits (instead of one ballot per band):
  // pass 0 the band of x0, pass 1 the band of x0 + 1 where it differs (one column in bw)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int bb = pass == 0 ? b0 : b1;
    const bool act = bb >= 0;
    unsigned long long same = __ballot(act);
    if (same == 0) continue;  // wave-uniform
#pragma unroll
    for (int bit = 0; bit < 4; ++bit) {
      const bool one = (bb >> bit) & 1;
      const unsigned long long m = __ballot(act && one);
      same &= one ? m : ~m;
    }
    const int rank = __popcll(same & ((1ull << lane) - 1ull)), total = __popcll(same);
    const int cur = __shfl(curv, act ? bb : 0, 64);
    if (lane < VOX_BANDS) vcnt[lane] = 0;
    if (act) {
      lists[sub_base[bb * VOX_SEGS + seg] + cur + rank] = ev;
      if (rank == total - 1) vcnt[bb] = total;  // the last lane of a band's group reports its size
    }
    if (lane < VOX_BANDS) curv += vcnt[lane];
  }
}

struct VoxSlab {
  float* tile;
  unsigned* tag;
  int c_lo, c_hi, r0, nr, rows, W, bins;
};

// One corner (dx,dy,dt) of up to 64 events (one per lane, in lane = event order) into the wave's columns of the LDS slab.
__device__ __forceinline__ void vox_corner_add(const VoxSlab sb, const float4 r, bool in, int dx, int dy, int dt) {
  const int lane = threadIdx.x & 63;
  const float xf = r.x, yf = r.y, tn = r.z, value = r.w;
  const int xl = (int)xf + dx, yl = (int)yf + dy, tl = (int)tn + dt;
  const float w = value * (1.0f - fabsf((float)xl - xf)) * (1.0f - fabsf((float)yl - yf)) * (1.0f - fabsf((float)tl - tn));
  // t_norm is NaN when all timestamps are equal (one event, or a burst with one time stamp: 0 / 0 in representations.py:76-80).
  // torch's `.int()` turns NaN into INT_MIN on the CPU, so the reference's range mask drops every such event and the grid stays
  // zero; v_cvt_i32_f32 turns NaN into 0, which would pass the range test and add NaN weights
  bool pend = in && tn == tn && xl >= sb.c_lo && xl < sb.c_hi && yl >= sb.r0 && yl < sb.r0 + sb.nr && yl >= 0 && tl >= 0 && tl < sb.bins && w != 0.0f;
  const int cell = pend ? (tl * sb.rows + (yl - sb.r0)) * sb.W + xl : 0;
  const int h = cell & (VOX_TAGS - 1);
  volatile unsigned* tag = sb.tag;
  volatile float* tile = sb.tile;
  while (__ballot(pend)) {  // lanes on one voxel (or one tag) go in lane order: the lowest pending lane wins the tag
    if (pend) atomicMin(sb.tag + h, (unsigned)lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const bool win = pend && tag[h] == (unsigned)lane;
    if (win) {
      tile[cell] = tile[cell] + w;
      tag[h] = 0xFFFFFFFFu;
      pend = false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// One workgroup per (slab of `rows` image rows, sample); wave w sweeps event segment w and owns column band w.
//   1. sweep (no workgroup barrier): a wave tests the keys of its contiguous share of the sample's events, 512 per trip with
//      8 loads in flight per lane; hits are compacted in event order into a per-wave LDS queue and, 64 at a time, appended to
//      the (band, segment) sub-lists they touch -- ranks from one ballot per band, so every sub-list is in event order, and
//      the sub-lists of a band lie behind each other in segment order: one list per band, in event order.
//   2. scatter: for each of the eight corners in the reference's loop order, a wave walks its band's list 64 events at a
//      time (the first VOX_CACHE chunks stay in registers over the corners).  Lanes whose corner lands on the same voxel are
//      serialised in lane (= event) order with a table of ds_min tags (an integer minimum is order independent); the
//      voxel itself is updated with a plain LDS read-add-write, since no other wave owns its column.  Adding a zero weight
//      cannot change an accumulator that started at +0, so such lanes sit out.
//   3. the slab leaves LDS with plain coalesced stores (no memset of the grid); the statistics of its non-zero voxels are
//      summed in a fixed order and written per slab (no atomics): the normalisation adds them up slab by slab.
__global__ __launch_bounds__(1024, 8) void voxel_scatter_kernel(const VoxArgs a, int want_stats) {
  extern __shared__ float tile[];  // [bins][rows][W]
  __shared__ double sh[3][16];
  __shared__ uint32_t wqueue[16 * 128];
  __shared__ unsigned tags[VOX_BANDS * VOX_TAGS];
  __shared__ int sub_base[VOX_BANDS * VOX_SEGS + 1];  // relative to slab_base
  __shared__ int vcnt_all[16 * VOX_BANDS];
  __shared__ long long red[16];
  __shared__ int wsum[4];
  const int b = blockIdx.y, sl = blockIdx.x;
  const int rows = a.rows, r0 = sl * rows, nr = min(rows, a.H - r0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slab = a.bins * rows * a.W;
  for (int i = tid; i < slab; i += 1024) tile[i] = 0.0f;
  for (int i = tid; i < VOX_BANDS * VOX_TAGS; i += 1024) tags[i] = 0xFFFFFFFFu;
  const long long o0 = a.offs[b], n = a.offs[b + 1] - o0;
  // list bases: sum of the sample's counts before this slab + exclusive prefix over this slab's (band, segment) counts
  const int per_slab = VOX_BANDS * VOX_SEGS;
  const int32_t* counts = a.counts + (size_t)b * a.nslab * per_slab;
  {
    long long acc = 0;
    for (int j = tid; j < sl * per_slab; j += 1024) acc += counts[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) red[wave] = acc;
    int v = 0, incl = 0;
    if (tid < per_slab) {
      v = counts[sl * per_slab + tid];
      incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      if (lane == 63) wsum[wave] = incl;
    }
    __syncthreads();
    if (tid < per_slab) {
      int pre = 0;
      for (int w = 0; w < wave; ++w) pre += wsum[w];
      sub_base[tid] = pre + incl - v;
      if (tid == per_slab - 1) sub_base[per_slab] = pre + incl;
    }
  }
  __syncthreads();
  long long slab_base = 4 * o0;
  for (int w = 0; w < 16; ++w) slab_base += red[w];
  uint32_t* lists = a.lists + slab_base;
  if (n > 0) {
    {  // ---- 1. sweep segment `wave`
      const long long seg = vox_seg(n), lo = wave * seg, hi = min(n, lo + seg);
      const uint32_t* keys = a.keys + o0;
      uint32_t* queue = wqueue + wave * 128;
      int qn = 0;    // wave-uniform
      int curv = 0;  // lane v: cursor of the (band v, this segment) sub-list
      for (long long base = lo; base < hi; base += 512) {
        uint32_t kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const long long i = base + u * 64 + lane;
          kv[u] = keys[i < n ? i : n - 1];  // clamped address: guarded loads are serialised by hipcc
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const long long i = base + u * 64 + lane;
          const bool hit = i < hi && vox_hits_slab(vox_key_y(kv[u]), r0, nr);
          const unsigned long long m = __ballot(hit);
          if (m == 0) continue;
          if (hit) queue[qn + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
          qn += __popcll(m);
          if (qn >= 64) {  // wave-uniform
            vox_distribute(queue[lane], true, keys, lists, sub_base, wave, a.bw, a.W, curv, vcnt_all + wave * VOX_BANDS);
            qn -= 64;
            if (lane < qn) {
              const uint32_t v = queue[64 + lane];
              queue[lane] = v;
            }
          }
        }
      }
      if (qn > 0) vox_distribute(lane < qn ? queue[lane] : 0u, lane < qn, keys, lists, sub_base, wave, a.bw, a.W, curv, vcnt_all + wave * VOX_BANDS);
    }
    // the lists are read by other waves of this workgroup: same CU, same L1 -> a workgroup-scope release is enough
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (an agent-scope __threadfence() here costs ~290 us per launch: L2 write-back)
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    {  // ---- 2. scatter band `wave`
      const uint32_t* mylist = lists + sub_base[wave * VOX_SEGS];
      const int mylen = sub_base[(wave + 1) * VOX_SEGS] - sub_base[wave * VOX_SEGS];
      const float4* rec = a.rec + o0;
      VoxSlab sb;
      sb.tile = tile;
      sb.tag = tags + wave * VOX_TAGS;
      sb.c_lo = wave * a.bw;
      sb.c_hi = min(a.W, sb.c_lo + a.bw);
      sb.r0 = r0;
      sb.nr = nr;
      sb.rows = rows;
      sb.W = a.W;
      sb.bins = a.bins;
      float4 rc[VOX_CACHE];
      {
        uint32_t ix[VOX_CACHE];
#pragma unroll
        for (int k = 0; k < VOX_CACHE; ++k) ix[k] = __builtin_nontemporal_load(mylist + (k * 64 + lane < mylen ? k * 64 + lane : 0));
#pragma unroll
        for (int k = 0; k < VOX_CACHE; ++k) rc[k] = rec[mylen > 0 ? ix[k] : 0];
      }
      const int nchunk = (mylen + 63) >> 6;
      // sensors deliver integer pixel coordinates: a corner with dx = 1 (dy = 1) then weighs every event with an exact zero
      // and is skipped for the whole list (the test is exact: such lanes would sit out one by one anyway)
      bool fx = false, fy = false;
#pragma unroll
      for (int k = 0; k < VOX_CACHE; ++k) {
        const bool in = k * 64 + lane < mylen;
        fx |= in && rc[k].x != (float)(int)rc[k].x;
        fy |= in && rc[k].y != (float)(int)rc[k].y;
      }
#pragma unroll 1
      for (int c = VOX_CACHE * 64; c < mylen; c += 64) {
        const bool in = c + lane < mylen;
        const float4 r = rec[__builtin_nontemporal_load(mylist + (in ? c + lane : 0))];
        fx |= in && r.x != (float)(int)r.x;
        fy |= in && r.y != (float)(int)r.y;
      }
      const bool any_fx = __ballot(fx) != 0, any_fy = __ballot(fy) != 0;
      for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner >> 2, dy = (corner >> 1) & 1, dt = corner & 1;
        if ((dx && !any_fx) || (dy && !any_fy)) continue;
        // the cached chunks take turns in rc[0] (one copy of the corner code; VOX_CACHE turns per corner restore the order),
        // longer lists read their further chunks again per corner
#pragma unroll 1
        for (int k = 0; k < max(nchunk, VOX_CACHE); ++k) {
          float4 r = rc[0];
          if (k < VOX_CACHE) {
#pragma unroll
            for (int q = 0; q + 1 < VOX_CACHE; ++q) rc[q] = rc[q + 1];
            rc[VOX_CACHE - 1] = r;
          } else {
            r = rec[__builtin_nontemporal_load(mylist + (k * 64 + lane < mylen ? k * 64 + lane : 0))];
          }
          if (k < nchunk) vox_corner_add(sb, r, k * 64 + lane < mylen, dx, dy, dt);
        }
      }
    }
  }
  __syncthreads();
  float* grid = a.grid + (size_t)b * a.bins * a.H * a.W;
  double c = 0.0, sm = 0.0, q = 0.0;
  const int rowlen = nr * a.W;
  for (int tb = 0; tb < a.bins; ++tb) {
    const float* src = tile + tb * rows * a.W;
    float* dst = grid + ((size_t)tb * a.H + r0) * a.W;
    for (int i = tid; i < rowlen; i += 1024) {
      const float v = src[i];
      dst[i] = v;
      if (v != 0.0f) {
        c += 1.0;
        sm += (double)v;
        q += (double)v * (double)v;
      }
    }
  }
  if (want_stats) {  // fixed order: thread-strided partials, butterfly over the lanes, the 16 waves in turn
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      c += __shfl_xor(c, off, 64);
      sm += __shfl_xor(sm, off, 64);
      q += __shfl_xor(q, off, 64);
    }
    if (lane == 0) {
      sh[0][wave] = c;
      sh[1][wave] = sm;
      sh[2][wave] = q;
    }
    __syncthreads();
    if (tid == 0) {
      double c2 = 0.0, s2 = 0.0, q2 = 0.0;
      for (int w = 0; w < 16; ++w) {
        c2 += sh[0][w];
        s2 += sh[1][w];
        q2 += sh[2][w];
      }
      double* part = a.part + ((size_t)b * a.nslab + sl) * 3;
      part[0] = c2;
      part[1] = s2;
      part[2] = q2;
    }
  }
}

// (v - mean) / std (unbiased) on the non-zero voxels; std == 0 -> only centre; grid (blocks, B)
__global__ void voxel_normalize_kernel(float* grid_all, long long n, const double* part_all, int nslab) {
  float* grid = grid_all + (size_t)blockIdx.y * n;
  const double* part = part_all + (size_t)blockIdx.y * nslab * 3;
  double cnt = 0.0, sum = 0.0, sq = 0.0;
  for (int sl = 0; sl < nslab; ++sl) {  // slab order: the same bits in every block and every run
    cnt += part[3 * sl];
    sum += part[3 * sl + 1];
    sq += part[3 * sl + 2];
  }
  if (cnt <= 0.0) return;
  const double mean = sum / cnt;
  double var = 0.0;
  if (cnt > 1.0) var = (sq - cnt * mean * mean) / (cnt - 1.0);
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean;
  const float stdf = (float)sqrt(var);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = grid[i];
    if (v != 0.0f) grid[i] = stdf > 0.0f ? (v - meanf) / stdf : (v - meanf);
  }
}

__global__ void events_count_kernel(const float* x, const float* y, const int64_t* offs, int H, int W, int32_t* cnt_all) {
  const int b = blockIdx.y;
  const long long o0 = offs[b], n = offs[b + 1] - o0;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xi = (int)x[o0 + i], yi = (int)y[o0 + i];
  if (xi >= 0 && xi < W && yi >= 0 && yi < H) atomicAdd(&cnt_all[(size_t)b * H * W + yi * W + xi], 1);
}

// one 1024-thread workgroup per sample: min and max of the count image
__global__ __launch_bounds__(1024) void minmax_kernel(const int32_t* cnt_all, int n, int32_t* mm_all /*[B][2]: min, max*/) {
  __shared__ int slo[16], shi[16];
  const int32_t* cnt = cnt_all + (size_t)blockIdx.x * n;
  int lo = 0x7fffffff, hi = -0x7fffffff - 1;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {  // eight loads in flight; a clamped index re-reads a valid element (harmless for min / max)
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = cnt[min(i0 + u * 1024, n - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      lo = min(lo, v[u]);
      hi = max(hi, v[u]);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      lo = min(lo, slo[w]);
      hi = max(hi, shi[w]);
    }
    mm_all[2 * blockIdx.x] = lo;
    mm_all[2 * blockIdx.x + 1] = hi;
  }
}

// uint8((cnt - min) / (max - min) * 255) > 0, in float64 like numpy; grid (blocks, B)
__global__ void events_mask_kernel(const int32_t* cnt_all, int n, const int32_t* mm_all, uint8_t* mask_all) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double lo = (double)mm_all[2 * b], hi = (double)mm_all[2 * b + 1];
  double v = ((double)cnt_all[(size_t)b * n + i] - lo) / (hi - lo) * 255.0;  // hi == lo gives NaN like numpy -> uint8 0 ... -> mask false
  if (v > 255.0) v = 255.0;
  mask_all[(size_t)b * n + i] = (v == v && (int)v > 0) ? 1 : 0;
}

}  // namespace

// workspace: [B][4] fp64 statistics | count image int32 [B,H,W] | min/max int32 [B][2] | device offsets int64 [B+1]
EINX_EXPORT size_t einx_events_ws_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * 32 + (size_t)B * H * W * sizeof(int32_t) + (size_t)B * 8 + ((size_t)B + 1) * 8 + 256;
}

namespace {
// The caller's offsets array is pageable memory that it may free as soon as the call returns, so it is first copied
// (synchronously, a few hundred bytes) into a library-owned PINNED staging buffer; the asynchronous host-to-device
// copy then reads that.  A RING of kOffSlots slots per (host thread, device), each with its own event: a call takes the next
// slot and only waits for the copy that used THAT slot kOffSlots calls ago.  (Rounds 2-5 had one slot: the second call of a
// batch -- einx_events_mask after einx_voxel_grid -- waited for the first call's copy, which an evaluation loop had queued
// behind the previous batch's whole forward, so the host could not run ahead of the device)
constexpr int kOffSlots = 8;
struct PinnedOffsets {
  int64_t* p[kOffSlots] = {nullptr};
  size_t cap[kOffSlots] = {0};
  hipEvent_t done[kOffSlots] = {nullptr};
  unsigned next = 0;
  ~PinnedOffsets() {
    for (int i = 0; i < kOffSlots; ++i) {
      if (p[i]) (void)hipHostFree(p[i]);
      if (done[i]) (void)hipEventDestroy(done[i]);
    }
  }
};
// one staging state per (host thread, device): an event can only be recorded on a stream of the device it was created on
constexpr int kMaxDev = 64;
thread_local PinnedOffsets g_offs[kMaxDev];

// copies the host offsets to the workspace and returns the largest per-sample event count (-1 on bad input)
long long stage_offsets(const int64_t* offsets_host, int B, int64_t* p, hipStream_t s) {
  long long mx = 0;
  for (int b = 0; b < B; ++b) {
    const long long n = offsets_host[b + 1] - offsets_host[b];
    if (n < 0) return -1;
    mx = n > mx ? n : mx;
  }
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= kMaxDev) return -2;
  PinnedOffsets& st = g_offs[devid];
  const int k = (int)(st.next++ % kOffSlots);
  const size_t need = (size_t)B + 1;
  if (st.done[k] && hipEventSynchronize(st.done[k]) != hipSuccess) return -2;  // the copy that used this slot kOffSlots calls ago has left it
  if (need > st.cap[k]) {
    if (st.p[k]) (void)hipHostFree(st.p[k]);
    st.p[k] = nullptr;
    st.cap[k] = 0;
    if (hipHostMalloc((void**)&st.p[k], need * 2 * sizeof(int64_t), hipHostMallocPortable) != hipSuccess) return -2;
    st.cap[k] = need * 2;
  }
  if (!st.done[k] && hipEventCreateWithFlags(&st.done[k], hipEventDisableTiming) != hipSuccess) return -2;
  for (size_t i = 0; i < need; ++i) st.p[k][i] = offsets_host[i];
  if (hipMemcpyAsync(p, st.p[k], need * 8, hipMemcpyHostToDevice, s) != hipSuccess) return -2;
  if (hipEventRecord(st.done[k], s) != hipSuccess) return -2;
  return mx;
}

// hipFuncSetAttribute is per device: remember the largest dynamic-LDS size granted on each one
int reserve_voxel_lds(size_t lds) {
  static std::mutex mu;
  static size_t granted[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (lds <= granted[dev]) return 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&voxel_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return -1;
  granted[dev] = lds;
  return 0;
}
}  // namespace

namespace {
// slab geometry of the scatter: `rows` image rows x all bins stay under ~62 KB of LDS (+17 KB of queues / tags) so that two
// workgroups share a CU
struct VoxGeom {
  int rows, nslab, bw;
};
VoxGeom vox_geom(int bins, int H, int W) {
  VoxGeom g;
  g.rows = (int)(16000 / ((long long)bins * W));
  g.rows = g.rows < 1 ? 1 : (g.rows > H ? H : g.rows);
  g.nslab = einx_cdiv(H, g.rows);
  g.bw = einx_cdiv(W, VOX_BANDS);
  return g;
}
size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

// workspace of einx_voxel_grid: per-slab statistics fp64 [B][nslab][3] | list lengths int32 [B][nslab][16][16] | device offsets
// int64 [B+1] | event records float4 [N] | cell keys uint32 [N] | event lists uint32 [4 N]   (N = total_events = offsets_host[B])
EINX_EXPORT size_t einx_voxel_ws_bytes(int B, int bins, int H, int W, int64_t total_events) {
  if (B <= 0 || bins <= 0 || H <= 0 || W <= 0 || total_events < 0) return 0;
  const VoxGeom g = vox_geom(bins, H, W);
  return al256((size_t)B * g.nslab * 3 * sizeof(double)) + al256((size_t)B * g.nslab * VOX_BANDS * VOX_SEGS * sizeof(int32_t)) +
         al256(((size_t)B + 1) * sizeof(int64_t)) + al256((size_t)total_events * sizeof(float4)) +
         al256((size_t)total_events * sizeof(uint32_t)) + al256((size_t)total_events * 4 * sizeof(uint32_t)) + 256;
}

EINX_EXPORT int einx_voxel_grid(const float* x, const float* y, const double* t, const float* p, const int64_t* offsets_host, int B,
                                int bins, int H, int W, int normalize, float* grid, void* ws, size_t ws_bytes, void* stream) {
  EINX_CHECK_ARG(offsets_host && grid && ws, "null pointer");
  EINX_CHECK_ARG(B > 0 && bins > 0 && H > 0 && W > 0 && H < 60000 && W < 60000, "bad shape");
  EINX_CHECK_ARG(offsets_host[0] == 0, "offsets_host[0] must be 0");
  // a batch without a single event (every sample empty): the event arrays may be NULL, the grids are zero like an empty sample's
  EINX_CHECK_ARG(offsets_host[B] == 0 || (x && y && t && p), "null event arrays");
  hipStream_t s = (hipStream_t)stream;
  const size_t per = (size_t)bins * H * W;
  const VoxGeom g = vox_geom(bins, H, W);
  const int64_t N = offsets_host[B];
  EINX_CHECK_ARG(N >= 0 && N < ((int64_t)1 << 31), "bad event count");
  EINX_CHECK_ARG(ws_bytes >= einx_voxel_ws_bytes(B, bins, H, W, N), "workspace smaller than einx_voxel_ws_bytes");
  char* wp = (char*)(((size_t)ws + 255) & ~(size_t)255);
  VoxArgs a;
  a.part = (double*)wp;
  wp += al256((size_t)B * g.nslab * 3 * sizeof(double));
  a.counts = (int32_t*)wp;
  const size_t counts_bytes = (size_t)B * g.nslab * VOX_BANDS * VOX_SEGS * sizeof(int32_t);
  wp += al256(counts_bytes);
  int64_t* offs = (int64_t*)wp;
  wp += al256(((size_t)B + 1) * sizeof(int64_t));
  a.rec = (float4*)wp;
  wp += al256((size_t)N * sizeof(float4));
  a.keys = (uint32_t*)wp;
  wp += al256((size_t)N * sizeof(uint32_t));
  a.lists = (uint32_t*)wp;
  const long long mx = stage_offsets(offsets_host, B, offs, s);
  EINX_CHECK_ARG(mx != -1, "offsets must be non-decreasing");
  const bool lds_hist = g.nslab * VOX_BANDS <= VOX_PREP_BINS;  // else: counted with global atomics into zeroed counters
  if (mx == -2 || (!lds_hist && hipMemsetAsync(a.counts, 0, counts_bytes, s) != hipSuccess)) {
    einx_set_error("einx_voxel_grid: memset / copy failed");
    return EINX_ERR_LAUNCH;
  }
  a.x = x;
  a.y = y;
  a.t = t;
  a.p = p;
  a.offs = offs;
  a.bins = bins;
  a.H = H;
  a.W = W;
  a.rows = g.rows;
  a.nslab = g.nslab;
  a.bw = g.bw;
  a.grid = grid;
  const size_t lds = (size_t)bins * g.rows * W * sizeof(float);
  EINX_CHECK_ARG(lds <= 140 * 1024, "bins * W too large for the LDS tile scatter");
  if (reserve_voxel_lds(lds) != 0) {
    einx_set_error("einx_voxel_grid: cannot reserve %zu bytes of LDS", lds);
    return EINX_ERR_LAUNCH;
  }
  if (lds_hist) hipLaunchKernelGGL(voxel_prep_kernel<true>, dim3(VOX_SEGS, (unsigned)B), dim3(1024), 0, s, a);
  else hipLaunchKernelGGL(voxel_prep_kernel<false>, dim3(VOX_SEGS, (unsigned)B), dim3(1024), 0, s, a);
  EINX_CHECK_LAUNCH();
  hipLaunchKernelGGL(voxel_scatter_kernel, dim3((unsigned)g.nslab, (unsigned)B), dim3(1024), lds, s, a, normalize);
  EINX_CHECK_LAUNCH();
  if (normalize) {
    hipLaunchKernelGGL(voxel_normalize_kernel, dim3(128, (unsigned)B), dim3(256), 0, s, grid, (long long)per, a.part, g.nslab);
    EINX_CHECK_LAUNCH();
  }
  return EINX_OK;
}

EINX_EXPORT int einx_events_mask(const float* x, const float* y, const int64_t* offsets_host, int B, int H, int W, void* ws, uint8_t* mask,
                                 void* stream) {
  EINX_CHECK_ARG(offsets_host && ws && mask, "null pointer");
  EINX_CHECK_ARG(B > 0 && H > 0 && W > 0, "bad shape");
  EINX_CHECK_ARG(offsets_host[B] == 0 || (x && y), "null event arrays");  // (no event at all: all-false masks)
  hipStream_t s = (hipStream_t)stream;
  const int n = H * W;
  int32_t* cnt = (int32_t*)((char*)ws + (size_t)B * 32);
  int32_t* mm = (int32_t*)((char*)ws + (size_t)B * 32 + (size_t)B * n * sizeof(int32_t));
  int64_t* offs = (int64_t*)(((size_t)((char*)ws + (size_t)B * 32 + (size_t)B * n * sizeof(int32_t) + (size_t)B * 8) + 7) & ~(size_t)7);
  const long long mx = stage_offsets(offsets_host, B, offs, s);
  EINX_CHECK_ARG(mx != -1, "offsets must be non-decreasing");
  if (mx == -2 || hipMemsetAsync(cnt, 0, (size_t)B * n * sizeof(int32_t), s) != hipSuccess) {
    einx_set_error("einx_events_mask: memset / copy failed");
    return EINX_ERR_LAUNCH;
  }
  if (mx > 0) {
    hipLaunchKernelGGL(events_count_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)B), dim3(256), 0, s, x, y, offs, H, W, cnt);
    EINX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)B), dim3(1024), 0, s, cnt, n, mm);
  EINX_CHECK_LAUNCH();
  hipLaunchKernelGGL(events_mask_kernel, dim3((unsigned)einx_cdiv(n, 256), (unsigned)B), dim3(256), 0, s, cnt, n, mm, mask);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}


// ------------------------------------------------------------------------------------------
// Host-side helper, no kernel: concatenate the per-sample event arrays of a batch (the reference's dataset hands one dict of
// numpy arrays per sample, datasets/representations.py:67-80 turns each into tensors) into the four flat arrays the kernels
// read, converting to their types on the way, with `threads` host threads.  A Python host did this with four np.concatenate
// passes on one thread: ~4 ms for 32 x 60k events, half of a B = 32 forward (the harness ran at 0.86 of the forward-only rate).
// The destination is normally page-locked memory that the caller uploads with one asynchronous copy per array.
// ------------------------------------------------------------------------------------------
namespace {
template <typename D, typename S>
void convert_run(D* dst, const void* src, long long i0, long long i1) {
  const S* s_ = (const S*)src;
  for (long long i = i0; i < i1; ++i) dst[i] = (D)s_[i];
}
template <typename D>
bool convert_any(D* dst, const void* src, int type, long long i0, long long i1) {
  switch (type) {
    case EINX_EV_F32:
      if (sizeof(D) == 4) memcpy(dst + i0, (const float*)src + i0, (size_t)(i1 - i0) * 4);
      else convert_run<D, float>(dst, src, i0, i1);
      return true;
    case EINX_EV_F64:
      if (sizeof(D) == 8) memcpy(dst + i0, (const double*)src + i0, (size_t)(i1 - i0) * 8);
      else convert_run<D, double>(dst, src, i0, i1);
      return true;
    case EINX_EV_I64: convert_run<D, int64_t>(dst, src, i0, i1); return true;
    case EINX_EV_I32: convert_run<D, int32_t>(dst, src, i0, i1); return true;
    case EINX_EV_I16: convert_run<D, int16_t>(dst, src, i0, i1); return true;
    case EINX_EV_U16: convert_run<D, uint16_t>(dst, src, i0, i1); return true;
    case EINX_EV_I8: convert_run<D, int8_t>(dst, src, i0, i1); return true;
    case EINX_EV_U8: convert_run<D, uint8_t>(dst, src, i0, i1); return true;
    case EINX_EV_U32: convert_run<D, uint32_t>(dst, src, i0, i1); return true;
    case EINX_EV_U64: convert_run<D, uint64_t>(dst, src, i0, i1); return true;
    default: return false;
  }
}
}  // namespace

EINX_EXPORT int einx_events_pack(const einx_event_arrays* samples, int B, float* x, float* y, double* t, float* p, int64_t* offsets,
                                 int threads) {
  EINX_CHECK_ARG(samples && offsets && B > 0, "null pointer / empty batch");
  long long total = 0;
  offsets[0] = 0;
  for (int b = 0; b < B; ++b) {
    const einx_event_arrays& e = samples[b];
    EINX_CHECK_ARG(e.n >= 0 && (e.n == 0 || (e.x && e.y && e.t && e.p)), "sample with null arrays");
    for (int ty : {e.x_type, e.y_type, e.t_type, e.p_type}) EINX_CHECK_ARG(ty >= 0 && ty <= EINX_EV_U64, "unknown element type");
    total += e.n;
    offsets[b + 1] = total;
  }
  if (total == 0) return EINX_OK;
  EINX_CHECK_ARG(x && y && t && p, "null destination");
  // work items: (sample, field) pieces of at most 64k elements, dealt round-robin to the threads
  struct Item {
    int b, f;
    long long i0, i1;
  };
  std::vector<Item> items;
  constexpr long long kPiece = 65536;
  for (int b = 0; b < B; ++b)
    for (long long i0 = 0; i0 < samples[b].n; i0 += kPiece)
      for (int f = 0; f < 4; ++f) items.push_back({b, f, i0, std::min((long long)samples[b].n, i0 + kPiece)});
  const int nt = std::max(1, std::min(threads, (int)items.size()));
  auto work = [&](int tid) {
    for (size_t k = (size_t)tid; k < items.size(); k += (size_t)nt) {
      const Item& it = items[k];
      const einx_event_arrays& e = samples[it.b];
      const long long o = offsets[it.b];
      if (it.f == 0) convert_any<float>(x + o, e.x, e.x_type, it.i0, it.i1);
      else if (it.f == 1) convert_any<float>(y + o, e.y, e.y_type, it.i0, it.i1);
      else if (it.f == 2) convert_any<double>(t + o, e.t, e.t_type, it.i0, it.i1);
      else convert_any<float>(p + o, e.p, e.p_type, it.i0, it.i1);
    }
  };
  std::vector<std::thread> pool;
  for (int i = 1; i < nt; ++i) pool.emplace_back(work, i);
  work(0);
  for (auto& th : pool) th.join();
  return EINX_OK;
}
