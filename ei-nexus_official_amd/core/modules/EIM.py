"""EIM: event extractor + image extractor + matcher (reference core/modules/EIM.py:13-100).

Same constructor, checkpoint-prefix loading and `forward` contract.  The difference is the
schedule: both extractors (on two HIP streams) and the matcher are enqueued for the whole batch,
and the host only waits for two small read-backs: the keypoint counts and the match counts that
shape the returned Python lists."""
import ctypes
import os

import torch
from torch import nn

from ..._native import on_input_device
from .Extractors import EventKeypointsExtractor, ImageKeypointsExtractor
from .Matchers import Matcher
from .matchers._batched import full_batch_lists


def _or_all(values):
    """bitwise OR of a list of Python ints (the per-image flag words of one read-back row)"""
    acc = 0
    for v in values:
        acc |= v
    return acc


class EIM(nn.Module):
    def __init__(self, config, device="cuda", logger=None):
        super().__init__()
        self.device = device
        self.config = config
        self.logger = logger
        self.event_extractor = EventKeypointsExtractor(config, logger, device=device)
        self.image_extractor = ImageKeypointsExtractor(config, logger, device=device)
        self.matcher = Matcher(config, logger, device=device)
        if config.pretrain_stage1.model_path is not None:
            m = torch.load(config.pretrain_stage1.model_path, map_location=device)
            self.event_extractor.load_state_dict({k[16:]: v for k, v in m.items() if "event_extractor" in k})
            if logger is not None:
                logger.log_info(f"Loaded pretrain_stage1 model from {config.pretrain_stage1.model_path}")
        if config.pretrain_stage2.model_path is not None:
            m = torch.load(config.pretrain_stage2.model_path, map_location=device)
            self.matcher.load_state_dict({k[8:]: v for k, v in m.items() if "matcher" in k})
            if logger is not None:
                logger.log_info(f"Loaded pretrain_stage2 model from {config.pretrain_stage2.model_path}")

    overlap_extractors = os.environ.get("EINX_OVERLAP", "1") != "0"  # two (independent) extractors on two HIP streams
    # (Measured in round 3 and removed in round 5, tools/experiments/r3_exp15.sh: running the event side's dense kernels beside the
    # image extractor's convolutions and the image side's beside the matcher is slower, 11.69 vs 11.10 ms per step at B=32.)

    _side_streams = {}  # (device type, index, caller stream handle) -> side stream, shared by every model of the process; bounded
    _SIDE_STREAMS_MAX = 16

    def _side_stream(self, device):
        """The stream the event extractor runs on beside the caller's.  HIP maps streams onto a small pool of hardware queues: a
        side stream per model instance means that the third or fourth model of a process gets one that shares a queue with the
        main stream, and its two extractors serialise (measured in bench.py's extra legs: B=1 1.23 ms instead of 0.92).  So all
        instances share one per (device, caller stream), chosen by probing (round 6)."""
        cur = torch.cuda.current_stream(device)
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), cur.cuda_stream)
        st = EIM._side_streams.get(key)
        if st is None:
            from ..._lib import check
            from ... import _native as N
            lib = N.lib()
            if torch.cuda.is_current_stream_capturing():  # no probe (it synchronises) and no stream creation inside a capture
                raise RuntimeError("einx: run one forward on this stream before capturing it (the side streams are chosen then)")
            # a stream that runs BESIDE the caller's: which hardware queue a stream lands on depends on what the process created
            # before it (a process group, a loader), so candidates from torch's pool are probed (einx_stream_overlap_us:
            # elapsed / spin is ~1.1 side by side, ~1.3 on one compute pipe, ~2.1 on one queue) and the first clean one is kept
            best, best_ratio = None, None
            with torch.cuda.device(device):
                for _ in range(8):
                    cand = torch.cuda.Stream(device=device)
                    if cand.cuda_stream == cur.cuda_stream:
                        continue
                    if os.environ.get("EINX_NO_STREAM_PROBE"):  # (diagnostics: first stream of torch's pool, no probe)
                        best = cand
                        break
                    us = ctypes.c_float()
                    if lib.einx_stream_overlap_us(ctypes.c_void_p(cur.cuda_stream), ctypes.c_void_p(cand.cuda_stream), 100, ctypes.byref(us)) != 0:
                        best = best or cand  # (no verdict from the probe: first candidate)
                        break
                    ratio = us.value / 100.0
                    if best is None or ratio < best_ratio - 0.1:
                        best, best_ratio = cand, ratio
                    if ratio < 1.25:
                        break
                if best is None:
                    best = torch.cuda.Stream(device=device)
                while len(EIM._side_streams) >= EIM._SIDE_STREAMS_MAX:  # (oldest first; torch's pool owns the streams)
                    EIM._side_streams.pop(next(iter(EIM._side_streams)))
                st = EIM._side_streams[key] = best
                # the native fork streams of both callers, each beside its caller and clear of the other three streams of a forward
                # (einx.h::einx_fork_stream_prepare_beside)
                arr = (ctypes.c_void_p * 2)(st.cuda_stream, None)
                check(lib.einx_fork_stream_prepare_beside(ctypes.c_void_p(cur.cuda_stream), arr, 1), "einx_fork_stream_prepare_beside")
                arr = (ctypes.c_void_p * 2)(cur.cuda_stream, lib.einx_fork_stream_of(ctypes.c_void_p(cur.cuda_stream)))
                check(lib.einx_fork_stream_prepare_beside(ctypes.c_void_p(st.cuda_stream), arr, 2), "einx_fork_stream_prepare_beside")
        return st

    def _two_streams_pay(self, image):
        """Two streams pay while the latency-bound tails (score map, NMS passes, selection, sampling) are a visible share of the
        forward: always for the 1/8-resolution networks (SP+MNN B=32: 8.4 against 8.8 ms).  The full-resolution networks (SiLK
        family, ~10x the convolution time per image) only up to about one 346x260 image: beyond that one stream is as fast or
        faster (B = 4: 12.5 against 13.3 ms, B = 32: 86.0 against 85.4), and the two-stream step is BIMODAL from process to
        process -- 85 or 95 ms at B = 32, with the streams verifiably on separate hardware queues in both modes: two convolution
        kernels sharing the CUs, profiles/r06_notes.md 9."""
        if all(getattr(e.extractor, "cell_size", 8) == 8 for e in (self.event_extractor, self.image_extractor)):
            return True
        return image.shape[0] * image.shape[-2] * image.shape[-1] <= (1 << 17)

    @on_input_device
    def forward_batched(self, events, image, events_mask=None, image_mask=None, nms_iters=None, prepared=False, before_match=None,
                        image_feats=None):
        """Enqueue the whole pipeline; returns device-side results without synchronising.
        The event and image extractors share nothing, so the event side is enqueued on a second HIP
        stream: its small late layers and its latency-bound NMS/selection kernels overlap the other
        side's convolutions instead of leaving most of the 256 CUs idle.
        image_feats: the image side has been enqueued on this stream already (`enqueue_image`: a host that still has to build
        the events -- packing raw events takes milliseconds -- starts the image network first and packs under it)."""
        if image_feats is not None:
            ev = self.event_extractor.extract_batched(events, events_mask, nms_iters=nms_iters, prepared=prepared)
            im = image_feats
        elif self.overlap_extractors and events.device.type == "cuda" and self._two_streams_pay(image):
            cur = torch.cuda.current_stream(events.device)
            side = self._side_stream(events.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ev = self.event_extractor.extract_batched(events, events_mask, nms_iters=nms_iters, prepared=prepared)
            im = self.image_extractor.extract_batched(image, image_mask, nms_iters=nms_iters, prepared=prepared)
            cur.wait_stream(side)
            # (under graph capture the allocations come from the graph's private pool and are never recycled, and
            # record_stream on them makes hipStreamEndCapture crash)
            for t in (() if torch.cuda.is_current_stream_capturing() else
                      (ev.feats, ev.logits, ev.raw, ev.raw_cl, ev.prob, ev.score, ev.score_crop, ev.sparse_desc, ev.coarse, ev.normalized, ev.det.positions,
                       ev.det.indices, ev.det.counts, ev.det.thr, ev.det.not_converged, ev.det.nms)):
                if t is not None:
                    t.record_stream(cur)  # allocated on the side stream, consumed on the caller's stream
        else:
            ev = self.event_extractor.extract_batched(events, events_mask, nms_iters=nms_iters, prepared=prepared)
            im = self.image_extractor.extract_batched(image, image_mask, nms_iters=nms_iters, prepared=prepared)
        if before_match is not None:
            before_match(ev, im)  # hook of forward(): early read-back of the detection counts
        mr = None
        if self.matcher.matcher is not None and self.matcher.freeze:
            mr = self.matcher.match_batched(ev, im)
        return ev, im, mr

    def _pinned(self, name, shape):
        buf = self._host_bufs.get(name) if hasattr(self, "_host_bufs") else None
        if buf is None or tuple(buf.shape) != tuple(shape):
            if not hasattr(self, "_host_bufs"):
                self._host_bufs = {}
            buf = self._host_bufs[name] = torch.empty(shape, dtype=torch.int32, pin_memory=True)
        return buf

    @on_input_device
    def enqueue_image(self, image, image_mask=None):
        """The image extractor alone, enqueued on the current stream; hand the result to `_enqueue(..., image_feats=)`."""
        return self.image_extractor.extract_batched(image, image_mask)

    @on_input_device
    def _enqueue(self, events, image, events_mask=None, image_mask=None, slot=0, prepared=False, image_feats=None):
        """Device side of one forward, nothing waits: both extractors, the matcher, the two small count read-backs
        (non-blocking copies into pinned buffers of `slot`, each followed by an event) and every output that does
        not depend on the counts."""
        B = events.shape[0]
        p = {"B": B, "slot": slot}

        def det_parts(ev, im):
            # four rows of B words (counts and NMS "needs more passes" flags of both sides), then the extractors' weight-watch
            # words (one each, when a watch rode on the call)
            return [ev.det.counts, im.det.counts, ev.det.not_converged, im.det.not_converged] + \
                   [t for t in (getattr(ev.det, "stale", None), getattr(im.det, "stale", None)) if t is not None]

        def read_detection(ev, im):
            rows = torch.cat(det_parts(ev, im))
            host = self._pinned(f"det{slot}", (int(rows.shape[0]),)).copy_(rows, non_blocking=True)
            p["det_host"], p["stale_host"] = host[:4 * B].view(4, B), host[4 * B:]
            p["det_event"] = torch.cuda.Event()
            p["det_event"].record()

        # graph mode waits for the whole stream anyway: ONE gather + ONE copy node at the end instead of the early detection
        # read-back (which lets `forward` build the feature lists while the matcher still runs) and the match-count read-back
        one_readback = slot == "g"
        ev, im, mr = self.forward_batched(events, image, events_mask, image_mask, before_match=None if one_readback else read_detection,
                                          prepared=prepared, image_feats=image_feats)
        p["ev"], p["im"], p["mr"] = ev, im, mr
        p["args"] = (events, image, events_mask, image_mask)
        p["nm_event"] = None
        if one_readback:
            parts = det_parts(ev, im)
            n_det = 4 * B + len(parts) - 4
            if mr is not None:
                parts += [mr.nmatch] + ([] if getattr(mr, "stale", None) is None else [mr.stale])
            rows = torch.cat(parts)
            host = self._pinned(f"all{slot}", (int(rows.shape[0]),)).copy_(rows, non_blocking=True)
            p["det_host"], p["stale_host"], p["det_event"] = host[:4 * B].view(4, B), host[4 * B:n_det], torch.cuda.Event()
            if mr is not None:
                p["nm_host"] = host[n_det:]
            p["det_event"].record()
        elif mr is not None:
            nm = mr.nmatch if getattr(mr, "stale", None) is None else torch.cat([mr.nmatch, mr.stale])  # + the matcher's weight watch
            p["nm_host"] = self._pinned(f"nmatch{slot}", (int(nm.shape[0]),)).copy_(nm, non_blocking=True)
            p["nm_event"] = torch.cuda.Event()
            p["nm_event"].record()
        ev.prepare()  # count-independent outputs are built while the device still works on the tail
        im.prepare()
        p["pre"] = full_batch_lists(mr) if mr is not None else None
        return p

    @on_input_device
    def _finish(self, p, _rerun=False):
        """Host side of one forward: wait for the two read-backs, cut the per-pair lists."""
        ev, im, mr, pre = p["ev"], p["im"], p["mr"], p["pre"]
        if p["det_event"] is not None:  # None: graph mode, the stream has been synchronised
            p["det_event"].synchronize()
        # (the pinned rows are read with ONE tolist() each and tested in Python: every torch operator on a 4 x B host tensor costs
        # 2-4 us, a dozen of them were a tenth of a single-pair forward_graph call)
        host = p.get("det_rows") or p["det_host"].tolist()
        B = len(host[0])
        nm_host, nm_event = p.get("nm_host"), p["nm_event"]
        flags_ev, flags_im = _or_all(host[2]), _or_all(host[3])
        sh = p.get("stale_host")
        stale = bool(sh is not None and sh.numel() and any(sh.tolist()))  # an extractor's weight watch (its own word, not a bit of the NMS flags)
        if not stale and mr is not None and getattr(mr, "stale", None) is not None:
            if nm_event is not None:
                nm_event.synchronize()
                nm_event = None
            nm_host = nm_host.tolist()
            stale = any(nm_host[B:])
        if stale:
            # a weight was edited through `.data` after the native images were built (the reference's modules would simply use
            # the new values): rebuild the images of all three modules and run this forward again.  SuperPointv1's in-place
            # `image /= 255` has already happened and is not repeated.
            if _rerun:  # the images were rebuilt from the current weights a moment ago: a second alarm is not an edit
                raise RuntimeError("einx: the weight watch still reports stale native weight images after they were rebuilt "
                                   "(weights modified concurrently with the forward?)")
            for mod in (self.event_extractor.extractor, self.image_extractor.extractor, self.matcher.matcher):
                if hasattr(mod, "refresh"):
                    mod.refresh()
            self.reset_graphs()
            return self._finish(self._enqueue(*p["args"], slot=p["slot"], prepared=True), _rerun=True)
        retries = 0
        while flags_ev | flags_im:
            retries += 1
            if retries > 8:  # 8 * 4**8 passes: cannot happen on a finite map (each pass removes at least one pixel or stops)
                raise RuntimeError("einx: the NMS fix-point did not converge within the maximum pass budget")
            # the NMS fix-point of some image needed more passes than were enqueued: redo only the
            # detection tail (and the matcher) with a larger, remembered, pass budget (rare: blocking read-back)
            for flags, bf, wrapper in ((flags_ev, ev, self.event_extractor), (flags_im, im, self.image_extractor)):
                if flags:
                    eng = wrapper.extractor.engine()
                    eng.redetect(bf, eng.grow_nms_iters())
            if mr is not None:
                mr = self.matcher.match_batched(ev, im)
                pre = None
            host = torch.stack([ev.det.counts, im.det.counts, ev.det.not_converged, im.det.not_converged]).tolist()
            flags_ev, flags_im = _or_all(host[2]), _or_all(host[3])
            if mr is not None:
                nm_host = mr.nmatch.tolist()
                nm_event = None
        if retries == 0:
            for wrapper in (self.event_extractor, self.image_extractor):
                eng = getattr(wrapper.extractor, "_engine", None)
                if eng is not None:
                    eng.note_converged()
        n, m = host[0], host[1]
        events_feats = ev.materialize(n)
        image_feats = im.materialize(m)
        if nm_event is not None:
            nm_event.synchronize()
        if nm_host is not None and not isinstance(nm_host, list):
            nm_host = nm_host.tolist()
        self._last_match = mr  # device-side MatchResult of this call (consumed by core.metrics batch_metrics)
        matches = None
        if mr is not None:
            n = [min(v, ev.det.cap) for v in n]
            m = [min(v, im.det.cap) for v in m]
            matches = self.matcher.materialize(mr, n, m, nm_host[:B], prebuilt=pre)
        elif self.matcher.matcher is not None:
            # un-frozen matcher (EIM.py:92-95 -> Matchers.py:204-222): random padding to max_points_num and
            # one stacked call; like the reference it rewrites sparse_positions / sparse_descriptors of
            # the two feature dicts in place
            matches = self.matcher(events_feats, image_feats)
        return events_feats, image_feats, matches

    @on_input_device
    def forward(self, events, image, events_mask=None, image_mask=None):
        """The host reads the counts back in two steps: the per-image keypoint counts are copied (pinned buffer,
        non-blocking) as soon as both extractors are done, so the feature lists are built while the matcher still
        runs; the per-pair match counts follow.  Both copies complete before this function returns: from the
        caller's point of view it is one synchronous forward like the reference's."""
        return self._finish(self._enqueue(events, image, events_mask, image_mask))

    # ------------------------------------------------------------------ latency mode: the forward as ONE hipGraph launch
    def forward_graph(self, events, image, events_mask=None, image_mask=None):
        """Latency mode for single pairs / small batches (the reference's evaluation loop calls the model pair by pair,
        test_events-image_same-time.py:130-194): the device side of `forward` -- both extractors, their side streams, the
        matcher and the two count read-backs, ~60 launches -- is captured ONCE per input geometry into a hipGraph
        (torch.cuda.CUDAGraph) and replayed with one launch per call; the host then only cuts the per-pair lists.
        Same kernels, same outputs as `forward`, with the contract of a captured graph:
          * the returned tensors live in buffers the graph owns: they are valid until the NEXT forward_graph call with the
            same geometry overwrites them (clone what must survive);
          * inputs are copied into the graph's input buffers, so SuperPoint's in-place `image /= 255` (reference quirk)
            happens on that copy, not on the caller's tensor;
          * a forward whose NMS fix-point needs more passes than the captured budget finishes eagerly ON THE GRAPH'S OWN
            BUFFERS (the detection tail and the matcher are redone with a larger budget, as in `forward`; the caller's tensors
            are not touched, so the `/= 255` is never applied to them) and the graph is dropped: the next call captures a new
            one with the grown budget;
          * weights are read at replay time (edits through load_state_dict need a new capture: call `reset_graphs()`);
          * configurations whose keypoint capacity is the whole map (no top-k, or a detection_threshold below 1: the forward then
            sizes its descriptor buffer from the real counts, a host round trip) cannot be captured: NotImplementedError, use
            `forward`;
          * one graph per input geometry, dtype AND configuration: the key holds every attribute that changes what is enqueued
            (`dense_outputs`, `want_log_assignment`, the detector settings, the matcher thresholds), so changing one of them
            captures a new graph instead of replaying the old configuration;
          * dicts returned by EARLIER calls alias the buffers the next replay overwrites -- their tensors, and their unresolved
            lazy entries (`normalized_descriptors`, `dense_*`), which would then be computed from the newer data: read or clone
            what must survive before the next call."""
        key = (events.shape, image.shape, None if events_mask is None else events_mask.shape,
               None if image_mask is None else image_mask.shape, events.device, events.dtype, image.dtype,
               None if events_mask is None else events_mask.dtype, None if image_mask is None else image_mask.dtype, self._graph_config())
        graphs = self.__dict__.setdefault("_graphs", {})
        g = graphs.get(key)
        if g is None:
            g = graphs[key] = self._capture(events, image, events_mask, image_mask)
        cur = torch.cuda.current_stream(events.device)
        for dst, src in zip(g["inputs"], (events, image, events_mask, image_mask)):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        g["graph"].replay()
        cur.synchronize()
        p = g["p"]
        rows = p["det_host"].tolist()
        if _or_all(rows[2]) | _or_all(rows[3]):
            # rare: the captured NMS pass budget was exceeded.  `_finish` redoes the detection tail and the matcher eagerly with
            # a larger, remembered budget on the graph's buffers (the replay has already scaled its own copy of the image);
            # this graph keeps the old budget, so it is dropped and the next call captures afresh
            graphs.pop(key, None)
        for bf, tmpl in ((p["ev"], g["prep_ev"]), (p["im"], g["prep_im"])):
            bf.reuse_prepared(tmpl)
        finish = self._finish if events.device.index not in (None, torch.cuda.current_device()) else EIM._finish.__wrapped__.__get__(self)
        return finish(dict(p, det_event=None, nm_event=None, det_rows=rows))

    def reset_graphs(self):
        self.__dict__.pop("_graphs", None)

    def _graph_config(self):
        """every attribute that changes what `_enqueue` puts on the stream (part of the graph cache key)"""
        def ext(e):
            return tuple(getattr(e, k, None) for k in ("dense_outputs", "detection_top_k", "detection_threshold", "nms_radius", "remove_borders",
                                                       "ordering", "dilate_mask"))
        m = self.matcher.matcher
        mk = None if m is None else (type(m).__name__,) + tuple(repr(getattr(m, k, None)) for k in ("want_log_assignment", "ratio_thresh", "distance_thresh")) \
            + (repr(getattr(getattr(m, "conf", None), "filter_threshold", None)),)
        return (ext(self.event_extractor.extractor), ext(self.image_extractor.extractor), mk, bool(getattr(self, "overlap_extractors", True)))

    @on_input_device
    def _capture(self, events, image, events_mask, image_mask):
        dev = events.device
        static = [None if t is None else t.clone() for t in (events, image, events_mask, image_mask)]
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            for _ in range(3):  # builds the native weight images, the library's side streams / events for THIS stream, the pinned buffers
                static[1].copy_(image)
                warm = self._enqueue(*static, slot="g")
                self._finish(warm)
            static[1].copy_(image)
        stream.synchronize()
        if getattr(warm["ev"], "sized_from_counts", False) or getattr(warm["im"], "sized_from_counts", False):
            # no top-k, or a detection_threshold below 1: the keypoint capacity is the whole map and the descriptor buffer is sized
            # from the real counts -- a host round trip in the middle of the forward, which a captured graph cannot contain
            raise NotImplementedError("einx: forward_graph needs a bounded keypoint capacity (detection_top_k set and detection_threshold >= 1, "
                                      "the shipped configurations); use forward() for this configuration")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            p = self._enqueue(*static, slot="g")
        return {"graph": graph, "inputs": static, "p": p, "prep_ev": p["ev"].prepared_template(), "prep_im": p["im"].prepared_template()}

    def forward_stream(self, batches, depth=2):
        """Throughput mode for evaluation loops (the reference's scripts iterate a DataLoader and call the model once per
        batch, test_events-image_same-time.py:130-194): `batches` yields (events, image[, events_mask[, image_mask]])
        tuples, results come back in order, one `forward` result per batch.  Up to `depth` batches are in flight: the
        next batch's convolutions are enqueued before the host waits for the previous batch's counts, so the
        latency-bound detection / matching tail and the host-side list building hide under them.  Same kernels, same
        outputs as `forward`; every input tensor must stay untouched until its result has been yielded."""
        from collections import deque
        pending = deque()
        k = 0
        for args in batches:
            pending.append(self._enqueue(*args, slot=k % max(depth, 1)))
            k += 1
            if len(pending) >= max(depth, 1):
                yield self._finish(pending.popleft())
        while pending:
            yield self._finish(pending.popleft())

    def count_parameters(self, model):
        return sum(p.numel() for p in model.parameters() if p.requires_grad)
