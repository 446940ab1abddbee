// extract.hip -- handle-level entry points: ONE call enqueues a whole extractor (SURVEY 8b's coarse ABI).
//
// einx_extractor_create copies the layer descriptors of one network (weights stay in caller-owned device
// memory, already repacked by einx_conv_repack / folded by einx_bn_fold); einx_extract then enqueues, on the
// given stream and without any host synchronisation,
//   [input /= 255]  ->  backbone convs (replicate pad folded into layer 1)  ->  detector head  ->  descriptor head
//   ->  [coarse-descriptor normalisation + channels-last raw copy]  ->  score map (softmax / sigmoid, pixel
//   shuffle, mask dilation, border)  ->  NMS fix-point + top-k + positions  ->  sparse descriptor sampling,
// i.e. everything VGGExtractor / VGGExtractorNP / SuperPointv1 / SiLKModel .forward does up to the output dict
// (reference core/modules/event_extractors/EventExtractors.py:517-624,:331-434,
// image_extractors/superpoint_extractor.py:345-480, image_extractors/silk_extractor.py:177-257).
// Intermediate activations live in a caller-provided workspace (two ping-pong buffers + the detector's
// workspace); every tensor of the reference's output dict is written to caller-provided outputs.
// The op-level entry points (einx_conv_block, einx_score_map, ...) stay exported for the unit tests; this file
// only sequences them, so a handle-level forward is bit-identical to the op-by-op forward.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "einx_common.h"

// One side (a library-owned stream + fork / join events) per (device, caller stream), shared by EVERY extractor handle of the
// process (round 5; rounds 3-4 kept one per handle): HIP maps streams onto its few hardware queues, so the streams of the fifth or
// sixth handle of a process landed on the queue of a caller's stream and the fork serialised -- bench.py's single-pair leg ran
// 1.06 instead of 0.77 ms per forward once two more streams had been created before its model
// (tools/experiments/r5_eager_after_run2.py).  Round 6: BOUNDED, and no stream is ever destroyed.  The streams come from a POOL of
// EINX_FORK_STREAM_POOL streams per device, created together at the first use on that device and kept for the life of the process;
// a side BORROWS the pool stream that a probe finds running beside its caller (pick_side_stream), and owns only its two events.
// At most EINX_FORK_STREAMS_MAX sides exist; a call on a further stream evicts the least recently used side that no call holds
// (shared_ptr: a side in use outlives its map entry), einx_fork_stream_release drops one explicitly.  (An earlier form of this
// round created up to eight candidate streams per side and destroyed the rejected ones, and destroyed a side's stream at
// eviction: with hipGraphs in the same process, hipGraphLaunch of ROCm 7.2 then crashed in hip::Graph::UpdateStreams after
// some tens of captures -- tools/experiments/r6_modes_crash.py, profiles/r06_notes.md 7.)  hipEventDestroy on events with
// enqueued work is deferred by the runtime until that work has drained.
// `mu` is held while a call enqueues its fork .. join section, so two host threads that enqueue on one stream cannot interleave
// on the events; re-recording an event does not disturb waits that were enqueued on its earlier record.  Two sides may borrow
// the same pool stream (more callers than pool streams): their sections then run one after the other, each between its own events.
struct EinxSide {
  hipStream_t stream = nullptr;  // borrowed from the device's pool: never destroyed
  hipEvent_t fork = nullptr, join = nullptr;
  int dev = 0;
  unsigned long long last_use = 0;
  std::mutex mu;
  EinxSide() = default;
  EinxSide(const EinxSide&) = delete;
  EinxSide& operator=(const EinxSide&) = delete;
  ~EinxSide() {
    int cur = 0;
    const bool sw = (fork || join) && hipGetDevice(&cur) == hipSuccess && cur != dev;
    if (sw) (void)hipSetDevice(dev);
    if (fork) (void)hipEventDestroy(fork);
    if (join) (void)hipEventDestroy(join);
    if (sw) (void)hipSetDevice(cur);
  }
};

struct einx_extractor {
  einx_extractor_desc d;
  std::vector<einx_conv_desc> backbone, det, desc;
  bool has_merged = false;  // det[0] + desc[0] as one layer (see einx_extractor_desc::merged_head0)
  einx_conv_desc merged;
};

namespace {

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Plan {
  int Hp, Wp, h0, w0;       // padded size, top/left pad
  int hc, wc;               // head resolution
  size_t buf_elems[2];      // ping-pong activation buffers (floats per image)
  size_t head_elems;        // scratch for the 3x3 head layers' outputs (floats per image)
};

// Small batches (the reference's own call pattern is one pair per forward): every launch after the backbone leaves most of
// the 256 CUs idle and the call is a chain of ~25 dependent kernels.  The descriptor branch (head convs, coarse
// normalisation) does not depend on the detector branch (head convs, score map, NMS passes, selection) until the sparse
// sampling, so it is enqueued on a second, library-owned stream between a fork and a join event: at B=1 ~90 us of ~540 per
// network leave the critical path.  The rule depends only on the arguments that size the workspace (a second head scratch).
// Round 5: the networks with 1/8-resolution heads fork at EVERY batch size -- at B = 32 the detector branch's latency-bound tail
// (score map, eight NMS passes, selection) then runs beside the descriptor head's convolutions of the same network as well as
// beside the other extractor: SP+MNN 8.52-8.60 -> 8.43-8.47 ms per step, SP+LightGlue B = 64 62.2 -> 61.6 (A/B on one box,
// alternating libraries).  The full-resolution networks (SiLK family) measured the same either way at B = 32 and keep the limit.
constexpr long kForkMaxCells = 8192;  // B x head pixels up to which the two head branches of a full-resolution network run concurrently
bool fork_heads(const einx_extractor* e, const Plan& pl, int B) { return e->d.cell == 8 || (long)B * pl.hc * pl.wc <= kForkMaxCells; }

// the sides of the process: (device, caller stream) -> side, with a use stamp for the LRU bound
typedef std::map<std::pair<int, hipStream_t>, std::shared_ptr<EinxSide>> SideMap;
std::mutex g_sides_mu;
unsigned long long g_side_clock = 0;
SideMap& side_map() {
  static SideMap* m = new SideMap();  // (never destructed: no HIP calls at process exit)
  return *m;
}

// The pool stream for `caller`'s side: one that runs BESIDE it.  HIP deals streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4
// by default) on a handful of compute pipes; which queue a stream got depends on everything the process created before it (torch's
// stream pool, a process group's streams, a loader's copy streams), and a side stream on its caller's queue serialises the fork
// (bench.py under torchrun with 4 queues: einx_stream_overlap_us(caller, fork stream) = 2.1).  So the choice is probed (round 6):
// the pool's streams are tried, least borrowed first, until one overlaps with the caller and, if possible, with the streams the
// host names (einx_fork_stream_prepare_beside) and the side streams of the two most recently used other sides of the device; the
// best one is borrowed.  Skipped (least borrowed stream taken) while the caller is capturing.  A few hundred microseconds per
// stream tried, once per (device, caller stream).  Called with g_sides_mu held and `dev` current.
constexpr int kProbeSpinUs = 100;
struct StreamPool {
  std::vector<hipStream_t> streams;
};
StreamPool* pool_for(int dev) {
  static std::map<int, StreamPool>* pools = new std::map<int, StreamPool>();  // (never destructed: no HIP calls at process exit)
  StreamPool& p = (*pools)[dev];
  if (p.streams.empty()) {
    for (int i = 0; i < EINX_FORK_STREAM_POOL; ++i) {
      hipStream_t st = nullptr;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
      p.streams.push_back(st);
    }
    if (p.streams.empty()) (void)hipGetLastError();
  }
  return p.streams.empty() ? nullptr : &p;
}

hipStream_t pick_side_stream(const SideMap& sides, int dev, hipStream_t caller, void* const* beside, int n_beside) {
  static const bool debug = getenv("EINX_DEBUG_STREAMS") != nullptr;
  static const bool no_probe = getenv("EINX_NO_STREAM_PROBE") != nullptr;  // (diagnostics: no probe, least borrowed stream)
  StreamPool* pool = pool_for(dev);
  if (!pool) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool probe = !no_probe && hipStreamIsCapturing(caller, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
  if (!probe) (void)hipGetLastError();
  // peers: (stream, weight) -- the caller first
  std::vector<std::pair<hipStream_t, int>> peers;
  peers.push_back({caller, 8});
  for (int i = 0; probe && i < n_beside; ++i)
    if ((hipStream_t)beside[i] != caller) peers.push_back({(hipStream_t)beside[i], 2});
  if (probe) {
    std::vector<const EinxSide*> recent;
    for (int k = 0; k < 2; ++k) {
      SideMap::const_iterator best = sides.end();
      for (SideMap::const_iterator it = sides.begin(); it != sides.end(); ++it) {
        if (!it->second || it->first.first != dev || it->first.second == caller || !it->second->stream) continue;
        bool taken = false;
        for (const EinxSide* r : recent) taken = taken || r == it->second.get();
        if (!taken && (best == sides.end() || it->second->last_use > best->second->last_use)) best = it;
      }
      if (best == sides.end()) break;
      recent.push_back(best->second.get());
      // (only the side's OWN stream: its caller's handle may belong to a stream the host has destroyed since)
      bool have = false;
      for (const std::pair<hipStream_t, int>& pr : peers) have = have || pr.first == best->second->stream;
      if (!have) peers.push_back({best->second->stream, 2});
    }
  }
  // pool streams, least borrowed first (ties: pool order)
  const int np = (int)pool->streams.size();
  std::vector<int> borrowed(np, 0), order(np);
  for (SideMap::const_iterator it = sides.begin(); it != sides.end(); ++it)
    for (int i = 0; i < np; ++i)
      if (it->second && it->first.first == dev && it->second->stream == pool->streams[i]) ++borrowed[i];
  for (int i = 0; i < np; ++i) order[i] = i;
  for (int i = 1; i < np; ++i)
    for (int j = i; j > 0 && borrowed[order[j]] < borrowed[order[j - 1]]; --j) {
      const int t = order[j];
      order[j] = order[j - 1];
      order[j - 1] = t;
    }
  int best_i = -1, best_cost = 1 << 30;
  for (int c = 0; c < np; ++c) {
    hipStream_t st = pool->streams[order[c]];
    if (st == caller) continue;
    int cost = 0;
    if (probe) {
      for (const std::pair<hipStream_t, int>& pr : peers) {
        if (pr.first == st) {  // a stream the side has to stay clear of IS this pool stream
          cost += 4 * pr.second;
          continue;
        }
        float us = 0.f;
        if (einx_stream_overlap_us((void*)pr.first, (void*)st, kProbeSpinUs, &us) != EINX_OK) continue;  // (no verdict: no cost)
        const float ratio = us / kProbeSpinUs;
        cost += ratio > 1.6f ? 4 * pr.second : ratio > 1.25f ? pr.second : 0;  // one queue / (probably) one pipe
        if (debug) fprintf(stderr, "[einx streams] caller %p pool stream %d (%p) vs %p: %.2f\n", (void*)caller, order[c], (void*)st, (void*)pr.first, ratio);
      }
    }
    if (debug) fprintf(stderr, "[einx streams] caller %p pool stream %d cost %d (borrowed by %d)\n", (void*)caller, order[c], cost, borrowed[order[c]]);
    if (cost < best_cost) {
      best_cost = cost;
      best_i = order[c];
    }
    if (cost == 0) break;
  }
  return best_i >= 0 ? pool->streams[best_i] : nullptr;
}

// the side of `caller` (keyed on the stream's OWN device, not on the current one); the returned reference keeps it alive
std::shared_ptr<EinxSide> side_for(hipStream_t caller, void* const* beside = nullptr, int n_beside = 0) {
  int dev = 0;
  if (caller) {
    if (hipStreamGetDevice(caller, &dev) != hipSuccess) return nullptr;
  } else if (hipGetDevice(&dev) != hipSuccess) {
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_sides_mu);
  SideMap& sides = side_map();
  std::shared_ptr<EinxSide>& slot = sides[{dev, caller}];
  if (!slot) {
    std::shared_ptr<EinxSide> sd = std::make_shared<EinxSide>();
    sd->dev = dev;
    int cur = 0;
    const bool sw = hipGetDevice(&cur) == hipSuccess && cur != dev;
    if (sw) (void)hipSetDevice(dev);
    sd->stream = pick_side_stream(sides, dev, caller, beside, n_beside);
    const bool ok = sd->stream != nullptr && hipEventCreateWithFlags(&sd->fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&sd->join, hipEventDisableTiming) == hipSuccess;
    if (sw) (void)hipSetDevice(cur);
    if (!ok) {
      sides.erase({dev, caller});
      return nullptr;  // (~EinxSide releases what was created)
    }
    slot = sd;
    // bound: evict least recently used sides nobody holds (use_count 1 = only the map)
    while ((int)sides.size() > EINX_FORK_STREAMS_MAX) {
      SideMap::iterator victim = sides.end();
      for (SideMap::iterator it = sides.begin(); it != sides.end(); ++it)
        if (it->second != sd && it->second.use_count() == 1 && (victim == sides.end() || it->second->last_use < victim->second->last_use)) victim = it;
      if (victim == sides.end()) break;  // every other side is in use right now: over the bound until they return
      sides.erase(victim);
    }
    std::shared_ptr<EinxSide> out = sd;
    out->last_use = ++g_side_clock;
    return out;
  }
  slot->last_use = ++g_side_clock;
  return slot;
}

// Padder.__init__ arithmetic (core/modules/utils/util.py:6-15)
void padder(int h, int w, int p, int* w0, int* w1, int* h0, int* h1) {
  const int hp = (((h / p) + 1) * p - h) % p;
  const int wp = (((w / p) + 1) * p - w) % p;
  *w0 = wp / 2;
  *w1 = wp - wp / 2;
  *h0 = hp / 2;
  *h1 = hp - hp / 2;
}

bool make_plan(const einx_extractor* e, int H, int W, Plan* pl) {
  int w0, w1, h0, h1;
  padder(H, W, e->d.cell, &w0, &w1, &h0, &h1);
  pl->Hp = H + h0 + h1;
  pl->Wp = W + w0 + w1;
  pl->h0 = h0;
  pl->w0 = w0;
  int h = pl->Hp, w = pl->Wp;
  pl->buf_elems[0] = pl->buf_elems[1] = 0;
  const int nb = (int)e->backbone.size();
  for (int i = 0; i < nb; ++i) {
    const einx_conv_desc& c = e->backbone[i];
    if (c.pool) {
      if ((h & 1) || (w & 1)) return false;
      h /= 2;
      w /= 2;
    }
    if (i + 1 < nb) {  // the last backbone layer writes the `backbone_feats` output, not a scratch buffer
      const size_t n = (size_t)c.cout * h * w;
      if (n > pl->buf_elems[i & 1]) pl->buf_elems[i & 1] = n;
    }
  }
  pl->hc = h;
  pl->wc = w;
  size_t he = 0;
  for (size_t i = 0; i + 1 < e->det.size(); ++i) he = std::max(he, (size_t)e->det[i].cout * h * w);
  for (size_t i = 0; i + 1 < e->desc.size(); ++i) he = std::max(he, (size_t)e->desc[i].cout * h * w);
  if (e->has_merged) he = std::max(he, (size_t)e->merged.cout * h * w);
  pl->head_elems = he;
  return h * e->d.cell == pl->Hp && w * e->d.cell == pl->Wp;
}

int topk_cap(int N, int top_k, float det_thr) {
  // rows a batch image can produce: N-1-lo for the quantile threshold of detection_top_k (see einx_detect), else N
  if (top_k <= 0 || top_k >= N || det_thr < 1.0f) return N;
  const float q = (float)(N - top_k) / (float)N;
  const float rank = q * (float)(N - 1);
  const int lo = (int)floorf(rank);
  return N - 1 - lo;
}

void detect_params(const einx_extractor* e, const Plan& pl, int B, int H, int W, int cap, int nms_iters, einx_detect_params* p) {
  p->B = B;
  p->Hp = pl.Hp;
  p->Wp = pl.Wp;
  p->H = H;
  p->W = W;
  p->h0 = pl.h0;
  p->w0 = pl.w0;
  p->radius = e->d.nms_radius;
  p->top_k = e->d.top_k;
  p->det_thr = e->d.det_thr;
  p->ordering_xy = e->d.ordering_xy;
  p->cap = cap;
  p->nms_iters = nms_iters;
}

}  // namespace

EINX_EXPORT einx_extractor* einx_extractor_create(const einx_extractor_desc* d) {
  if (!d) {
    einx_set_error("einx_extractor_create: null descriptor");
    return nullptr;
  }
  if (d->struct_size != sizeof(einx_extractor_desc)) {
    einx_set_error("einx_extractor_create: einx_extractor_desc::struct_size is %zu, this library's is %zu (header / library ABI mismatch: EINX_ABI_VERSION %d)",
                   (size_t)d->struct_size, sizeof(einx_extractor_desc), EINX_ABI_VERSION);
    return nullptr;
  }
  if (!d || !d->backbone || !d->det_head || !d->desc_head || d->n_backbone <= 0 || d->n_det <= 0 || d->n_desc <= 0) {
    einx_set_error("einx_extractor_create: null / empty layer lists");
    return nullptr;
  }
  if (d->cell != 8 && d->cell != 1) {
    einx_set_error("einx_extractor_create: cell must be 8 (SuperPoint-shaped) or 1 (SiLK-shaped)");
    return nullptr;
  }
  einx_extractor* e = new (std::nothrow) einx_extractor();
  if (!e) return nullptr;
  e->d = *d;
  e->backbone.assign(d->backbone, d->backbone + d->n_backbone);
  e->det.assign(d->det_head, d->det_head + d->n_det);
  e->desc.assign(d->desc_head, d->desc_head + d->n_desc);
  e->d.backbone = e->backbone.data();
  e->d.det_head = e->det.data();
  e->d.desc_head = e->desc.data();
  e->d.merged_head0 = nullptr;
  if (d->merged_head0) {
    const einx_conv_desc& m = *d->merged_head0;
    const einx_conv_desc &a = e->det.front(), &b = e->desc.front();
    if (e->det.size() != 2 || e->desc.size() != 2 || m.cin != a.cin || m.cin != b.cin || m.cout != a.cout + b.cout || m.ks != a.ks || m.ks != b.ks ||
        m.relu != a.relu || m.relu != b.relu || m.pool || a.pool || b.pool || !m.w_native || (m.scale == nullptr) != (a.scale == nullptr) ||
        (m.scale == nullptr) != (b.scale == nullptr) || a.cout % 64 != 0) {
      einx_set_error("einx_extractor_create: merged_head0 (cin %d cout %d ks %d relu %d pool %d bn %d) does not describe det_head[0] (%d layers; cin %d cout %d ks %d relu %d bn %d) + "
                     "desc_head[0] (%d layers; cin %d cout %d ks %d relu %d bn %d): two-layer heads of one structure, det_head[0].cout a multiple of 64",
                     m.cin, m.cout, m.ks, m.relu, m.pool, m.scale != nullptr, (int)e->det.size(), a.cin, a.cout, a.ks, a.relu, a.scale != nullptr,
                     (int)e->desc.size(), b.cin, b.cout, b.ks, b.relu, b.scale != nullptr);
      delete e;
      return nullptr;
    }
    e->merged = m;
    e->has_merged = true;
  }
  const int want = d->cell == 8 ? 65 : 1;
  if (e->det.back().cout != want || e->backbone.back().cout != e->det.front().cin || e->backbone.back().cout != e->desc.front().cin) {
    einx_set_error("einx_extractor_create: head shapes do not fit (detector head must end in %d channels)", want);
    delete e;
    return nullptr;
  }
  return e;
}

EINX_EXPORT void einx_extractor_destroy(einx_extractor* e) { delete e; }

EINX_EXPORT int einx_fork_stream_prepare(void* stream) {
  if (!side_for((hipStream_t)stream)) {
    einx_set_error("einx_fork_stream_prepare: could not create the side stream / events");
    return EINX_ERR_LAUNCH;
  }
  return EINX_OK;
}

EINX_EXPORT int einx_fork_stream_prepare_beside(void* stream, void* const* beside, int n_beside) {
  EINX_CHECK_ARG(n_beside >= 0 && n_beside <= 8 && (beside || n_beside == 0), "0..8 streams to stay clear of");
  if (!side_for((hipStream_t)stream, beside, n_beside)) {
    einx_set_error("einx_fork_stream_prepare_beside: could not create the side stream / events");
    return EINX_ERR_LAUNCH;
  }
  return EINX_OK;
}

EINX_EXPORT int einx_fork_stream_release(void* stream) {
  std::vector<std::shared_ptr<EinxSide>> dropped;  // destroyed after the map lock is released
  {
    std::lock_guard<std::mutex> lk(g_sides_mu);
    SideMap& sides = side_map();
    for (SideMap::iterator it = sides.begin(); it != sides.end();) {
      if (it->first.second == (hipStream_t)stream) {
        dropped.push_back(it->second);
        it = sides.erase(it);
      } else {
        ++it;
      }
    }
  }
  return EINX_OK;
}

EINX_EXPORT void* einx_fork_stream_of(void* stream) {
  int dev = 0;
  if (stream ? hipStreamGetDevice((hipStream_t)stream, &dev) != hipSuccess : hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_sides_mu);
  SideMap& sides = side_map();
  SideMap::iterator it = sides.find({dev, (hipStream_t)stream});
  return it == sides.end() ? nullptr : (void*)it->second->stream;
}

EINX_EXPORT int einx_fork_stream_count(void) {
  std::lock_guard<std::mutex> lk(g_sides_mu);
  return (int)side_map().size();
}

EINX_EXPORT int einx_extract_shapes(const einx_extractor* e, int H, int W, einx_extract_shapes_t* s) {
  EINX_CHECK_ARG(e && s, "null pointer");
  Plan pl;
  EINX_CHECK_ARG(H > 0 && W > 0 && make_plan(e, H, W, &pl), "image size does not fit the network's pooling / cell size");
  s->Hp = pl.Hp;
  s->Wp = pl.Wp;
  s->h0 = pl.h0;
  s->w0 = pl.w0;
  s->hc = pl.hc;
  s->wc = pl.wc;
  s->feat_channels = e->backbone.back().cout;
  s->det_channels = e->det.back().cout;
  s->desc_dim = e->desc.back().cout;
  s->cap = topk_cap(pl.Hp * pl.Wp, e->d.top_k, e->d.det_thr);
  return EINX_OK;
}

// nms_iters <= 0 means "the default pass budget" in BOTH the size query and the call (the detector's per-image flag array
// is B x nms_iters words of the workspace)
static inline int nms_budget(int nms_iters) { return nms_iters > 0 ? nms_iters : 8; }

EINX_EXPORT size_t einx_extract_ws_bytes(const einx_extractor* e, int B, int H, int W, int cap, int nms_iters) {
  Plan pl;
  if (!e || B <= 0 || !make_plan(e, H, W, &pl)) return 0;
  einx_detect_params p;
  detect_params(e, pl, B, H, W, cap > 0 ? cap : 1, nms_budget(nms_iters), &p);
  size_t bytes = 0;
  bytes += align256(pl.buf_elems[0] * B * sizeof(float)) + align256(pl.buf_elems[1] * B * sizeof(float));
  bytes += align256(pl.head_elems * B * sizeof(float)) * (fork_heads(e, pl, B) ? 2 : 1);
  bytes += align256(einx_detect_ws_bytes(&p));
  return bytes + 256;
}

EINX_EXPORT int einx_extract(const einx_extractor* e, float* in, const uint8_t* mask, int B, int H, int W, int nms_iters, void* ws,
                             size_t ws_bytes, const einx_extract_out* o, void* stream) {
  return einx_extract_watch(e, in, mask, B, H, W, nms_iters, ws, ws_bytes, o, nullptr, stream);
}

EINX_EXPORT int einx_extract_watch(const einx_extractor* e, float* in, const uint8_t* mask, int B, int H, int W, int nms_iters, void* ws,
                                   size_t ws_bytes, const einx_extract_out* o, const einx_weight_watch* ww, void* stream) {
  EINX_CHECK_ARG(e && in && ws && o, "null pointer");
  EINX_CHECK_ARG(!ww || ww->struct_size == sizeof(einx_weight_watch), "einx_weight_watch::struct_size does not match this library (ABI mismatch)");
  EINX_CHECK_ARG(!ww || ww->n == 0 || (ww->n > 0 && ww->table && ww->ref && ww->scratch && ww->stale), "weight watch with null pointers");
  EINX_CHECK_ARG(o->feats && o->logits && o->raw && o->prob && o->score && o->positions && o->indices && o->counts && o->thr &&
                     o->not_converged && o->sparse_desc,
                 "null output pointer");
  EINX_CHECK_ARG(B > 0 && H > 0 && W > 0 && o->cap > 0, "bad shape");
  Plan pl;
  EINX_CHECK_ARG(make_plan(e, H, W, &pl), "image size does not fit the network's pooling / cell size");
  EINX_CHECK_ARG(e->d.cell == 1 || (o->coarse && o->raw_cl), "cell-8 networks need the coarse / raw_cl outputs");
  EINX_CHECK_ARG(ws_bytes >= einx_extract_ws_bytes(e, B, H, W, o->cap, nms_iters), "workspace smaller than einx_extract_ws_bytes for these arguments");
  char* p = (char*)ws;
  float* buf[2];
  buf[0] = (float*)p;
  p += align256(pl.buf_elems[0] * B * sizeof(float));
  buf[1] = (float*)p;
  p += align256(pl.buf_elems[1] * B * sizeof(float));
  float* head = (float*)p;
  p += align256(pl.head_elems * B * sizeof(float));
  // (while einx_profile_enable(1) records per-launch times the branches stay in line: its numbers are kernels ALONE on the chip)
  bool fork = fork_heads(e, pl, B) && !einx_profile_active();
  {  // a fork nested inside a caller's own fork crashes hipStreamEndCapture (ROCm 7.2): under capture the branches stay in line
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (fork && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) fork = false;
  }
  float* head2 = head;  // the descriptor head's own scratch when the two branches run concurrently
  if (fork) {
    head2 = (float*)p;
    p += align256(pl.head_elems * B * sizeof(float));
  }
  void* det_ws = p;
  int rc;
  if (e->d.input_div != 0.0f && e->d.input_div != 1.0f) {  // SuperPointv1: `image /= 255.0` in place on the caller's tensor
    rc = einx_div_inplace(in, (size_t)B * e->backbone.front().cin * H * W, e->d.input_div, stream);
    if (rc) return rc;
  }
  // ---- backbone
  const float* cur = in;
  int h = pl.Hp, w = pl.Wp;
  const int nb = (int)e->backbone.size();
  int first = 0;
  // the first two layers of a 1-channel network as one launch (large launches only: einx_conv_first_two_fused_ok)
  if (nb >= 2 && 1 == einx_conv_first_two_fused_ok(&e->backbone[0], &e->backbone[1], B, pl.Hp, pl.Wp)) {
    float* out = (2 < nb) ? buf[1] : o->feats;
    if ((rc = einx_conv_first_two_fused(in, B, H, W, pl.h0, pl.w0, pl.Hp, pl.Wp, &e->backbone[0], &e->backbone[1], out, stream))) return rc;
    if (e->backbone[1].pool) {
      h /= 2;
      w /= 2;
    }
    cur = out;
    first = 2;
  }
  for (int i = first; i < nb; ++i) {
    const einx_conv_desc& c = e->backbone[i];
    float* out = (i + 1 < nb) ? buf[i & 1] : o->feats;
    if (i == 0) rc = einx_conv_block(cur, B, H, W, pl.h0, pl.w0, pl.Hp, pl.Wp, &c, out, stream);
    else rc = einx_conv_block(cur, B, h, w, 0, 0, h, w, &c, out, stream);
    if (rc) return rc;
    if (c.pool) {
      h /= 2;
      w /= 2;
    }
    cur = out;
  }
  // ---- heads (the hidden 3x3 layer of each head goes through a scratch buffer)
  auto run_head = [&](const std::vector<einx_conv_desc>& L, float* scratch, float* final_out, void* st) -> int {
    const float* x = o->feats;
    for (size_t i = 0; i < L.size(); ++i) {
      float* out = (i + 1 < L.size()) ? scratch : final_out;
      const int r = einx_conv_block(x, B, h, w, 0, 0, h, w, &L[i], out, st);
      if (r) return r;
      x = out;
    }
    return 0;
  };
  const int D = e->desc.back().cout;
  // single images: the two heads' first layers as one launch into `head` (det channels first); each head's second layer then reads
  // its slice (one image: a channel slice is contiguous)
  const bool merged = e->has_merged && B == 1;
  if (merged && (rc = einx_conv_block(o->feats, B, h, w, 0, 0, h, w, &e->merged, head, stream))) return rc;
  const float* desc_in = head + (size_t)e->det.front().cout * h * w;
  auto det_branch = [&](void* st) -> int {
    if (merged) return einx_conv_block(head, B, h, w, 0, 0, h, w, &e->det[1], o->logits, st);
    return run_head(e->det, head, o->logits, st);
  };
  // descriptor branch: head convs + the dense by-product that depends on `raw` only
  auto desc_branch = [&](void* st) -> int {
    int r = merged ? einx_conv_block(desc_in, B, h, w, 0, 0, h, w, &e->desc[1], o->raw, st) : run_head(e->desc, head2, o->raw, st);
    if (r) return r;
    if (e->d.cell == 8) r = einx_normalize_map(o->raw, B, D, h * w, e->d.desc_scale, o->coarse, o->raw_cl, st);
    return r;
  };
  const std::shared_ptr<EinxSide> sd = fork ? side_for((hipStream_t)stream) : nullptr;  // (held until the call returns)
  std::unique_lock<std::mutex> side_lock;
  if (sd) side_lock = std::unique_lock<std::mutex>(sd->mu);
  if (sd) {  // fork: the descriptor branch runs beside the detector branch
    if (hipEventRecord(sd->fork, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(sd->stream, sd->fork, 0) != hipSuccess) {
      einx_set_error("einx_extract: fork failed");
      return EINX_ERR_LAUNCH;
    }
    rc = desc_branch(sd->stream);
    // the join is enqueued even when a launch failed, so that the caller's stream never runs ahead of the side stream
    const bool ok = hipEventRecord(sd->join, sd->stream) == hipSuccess;
    if (rc) {
      (void)hipStreamWaitEvent((hipStream_t)stream, sd->join, 0);
      return rc;
    }
    if (!ok) {
      einx_set_error("einx_extract: join failed");
      return EINX_ERR_LAUNCH;
    }
    if ((rc = det_branch(stream))) {
      (void)hipStreamWaitEvent((hipStream_t)stream, sd->join, 0);
      return rc;
    }
  } else {
    if ((rc = det_branch(stream))) return rc;
    if ((rc = desc_branch(stream))) return rc;
  }
  einx_detect_params dp;
  detect_params(e, pl, B, H, W, o->cap, nms_budget(nms_iters), &dp);
  // the score kernel also zeroes the detection's NMS pass flags (no memset launch) when it has a thread per flag word
  int nflags = 0;
  int32_t* flags = einx_detect_flags(&dp, det_ws, &nflags);
  const int C_det = e->det.back().cout;
  const long score_threads = (long)(C_det == 65 ? einx_cdiv(B * h * w, 32) : einx_cdiv(B * h * w, 256)) * 256;
  const bool zero_in_score = nflags > 0 && nflags <= score_threads;
  rc = einx_score_map_zero(o->logits, B, C_det, h, w, mask, H, W, pl.h0, pl.w0, e->d.dilate_mask, e->d.border, o->prob, o->score,
                           zero_in_score ? flags : nullptr, zero_in_score ? nflags : 0, o->score_crop, stream);
  if (rc) {
    if (sd) (void)hipStreamWaitEvent((hipStream_t)stream, sd->join, 0);
    return rc;
  }
  // the cropped `nms` output is written by extra workgroups of the sampling launch below (one launch less): the detection only
  // reports which buffer holds the NMS fix-point
  const float* nms_map = nullptr;
  rc = einx_detect_prezeroed(o->score, &dp, det_ws, o->nms, o->positions, o->indices, o->counts, o->thr, o->not_converged, zero_in_score ? 1 : 0,
                             o->nms ? &nms_map : nullptr, stream);
  if (sd && hipStreamWaitEvent((hipStream_t)stream, sd->join, 0) != hipSuccess && !rc) {  // join before the sampler reads `raw`
    einx_set_error("einx_extract: join failed");
    return EINX_ERR_LAUNCH;
  }
  if (rc) return rc;
  const bool bilinear = e->d.cell == 8;
  const bool use_cl = bilinear && D <= 512;
  EinxWatch watch;  // the weight watch rides on spare workgroups of the sampling kernel: no launch of its own
  const bool on = ww && ww->n > 0;
  watch.table = on ? ww->table : nullptr;
  watch.ref = on ? (const unsigned long long*)ww->ref : nullptr;
  watch.hash = on ? (unsigned long long*)ww->scratch : nullptr;
  watch.flag = on ? ww->stale : nullptr;
  watch.n = on ? ww->n : 0;
  watch.bit = 1;
  EinxCrop crop{};
  crop.map = o->nms ? nms_map : nullptr;
  crop.thr = o->thr;
  crop.out = o->nms;
  crop.Hp = pl.Hp;
  crop.Wp = pl.Wp;
  crop.h0 = pl.h0;
  crop.w0 = pl.w0;
  crop.H = H;
  crop.W = W;
  return einx_desc_sample_watch(use_cl ? o->raw_cl : o->raw, B, D, h, w, pl.Hp, pl.Wp, bilinear ? 1 : 0, use_cl ? 1 : 0, o->indices, o->counts,
                                o->cap, e->d.desc_scale, o->sparse_desc, watch, crop, stream);
}
