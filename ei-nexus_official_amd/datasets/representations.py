"""Event representations on the GPU, same names and argument meaning as the
reference's datasets/representations.py:67-124 (`events_to_voxel_grid`) and the events mask built in
datasets/visualize.py:23-50 + test_events-image_same-time.py:137.  Events arrive as the reference's
dict of numpy arrays {"x","y","t","p"}; the result stays on the device, ready for EIM.forward."""
import ctypes

import numpy as np
import torch

from .. import _native as N
from .. import _lib
from .._lib import check


class EventStage:
    """Reusable upload path for the packed events of ONE in-flight batch: page-locked host arrays the samples are concatenated
    into directly (no temporaries) and device arrays they are copied to with non-blocking copies on the current stream, so
    that an evaluation loop can pack and upload batch i + 1 while the device still works on batch i
    (harness.SameTimeEvaluator.run).  The arrays grow to the largest batch seen.  A stage must not be packed into again
    before the work that reads its previous contents has finished."""
    _copy_streams = {}
    _FIELDS = (("x", np.float32, torch.float32), ("y", np.float32, torch.float32), ("t", np.float64, torch.float64), ("p", np.float32, torch.float32))

    def __init__(self, device):
        self.device = torch.device(device)
        self.cap = 0
        self.host, self.dev = {}, {}
        # the transfer overlaps the work already queued on the caller's stream; ONE copy stream per device for the whole process
        # (HIP deals streams onto four compute pipes in creation order: every further stream is one more that can land on the
        # pipe of a stream that matters, einx.h::einx_fork_stream_prepare)
        key = (self.device.type, self.device.index if self.device.index is not None else torch.cuda.current_device())
        if key not in EventStage._copy_streams:
            EventStage._copy_streams[key] = torch.cuda.Stream(self.device)
        self.copy_stream = EventStage._copy_streams[key]

    def _reserve(self, n):
        if n <= self.cap:
            return
        self.cap = max(n, int(self.cap * 1.5))
        for name, _, tdt in self._FIELDS:
            self.host[name] = torch.empty(self.cap, dtype=tdt, pin_memory=True)
            self.dev[name] = torch.empty(self.cap, dtype=tdt, device=self.device)

    # element types einx_events_pack converts from (include/einx.h: EINX_EV_*); anything else goes through float64 first
    _EV_TYPES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int64): 2, np.dtype(np.int32): 3, np.dtype(np.int16): 4,
                 np.dtype(np.uint16): 5, np.dtype(np.int8): 6, np.dtype(np.uint8): 7, np.dtype(np.uint32): 8, np.dtype(np.uint64): 9,
                 np.dtype(np.bool_): 7}
    _threads = None

    @classmethod
    def pack_threads(cls):
        """host threads of the packing helper: what the process may keep busy (affinity, cgroup quota) less the two that drive
        Python and the HIP runtime, at most 8 (38 MB of copies saturate the memory system long before that)"""
        if cls._threads is None:
            from ..placement import pool_threads
            cls._threads = max(1, min(8, int(pool_threads())))
        return cls._threads

    def pack(self, events_list, defer_join=False):
        """One pass: the per-sample arrays are converted and concatenated straight into the page-locked arrays by the library's
        host-side helper on several threads (einx_events_pack; round 5 made four np.concatenate passes on one thread: ~4 ms of a
        9.8 ms batch), then uploaded with one non-blocking copy per array on the copy stream."""
        B = len(events_list)
        arr = (_lib.EventArrays * B)()
        keep = []
        n = 0
        for b, ev in enumerate(events_list):
            fields = []
            for name in ("x", "y", "t", "p"):
                a = np.asarray(ev[name])
                code = self._EV_TYPES.get(a.dtype)
                if code is None or not a.flags["C_CONTIGUOUS"]:
                    a = np.ascontiguousarray(a, None if code is not None else np.float64)
                    code = self._EV_TYPES[a.dtype]
                keep.append(a)
                fields.append((a.ctypes.data, code))
            ln = len(keep[-4])
            if not all(len(k) == ln for k in keep[-4:]):
                raise ValueError(f"sample {b}: x / y / t / p differ in length")
            arr[b] = _lib.EventArrays(fields[0][0], fields[1][0], fields[2][0], fields[3][0], fields[0][1], fields[1][1], fields[2][1], fields[3][1], ln)
            n += ln
        if n == 0:
            return None
        self._reserve(n)
        offs = np.zeros(B + 1, np.int64)
        hp = [ctypes.c_void_p(self.host[name].data_ptr()) for name, _, _ in self._FIELDS]
        check(N.lib().einx_events_pack(arr, B, hp[0], hp[1], hp[2], hp[3], offs.ctypes.data_as(ctypes.c_void_p), self.pack_threads()),
              "einx_events_pack")
        out = []
        with torch.cuda.stream(self.copy_stream):
            for name, _, _ in self._FIELDS:
                out.append(self.dev[name][:n].copy_(self.host[name][:n], non_blocking=True))
        if not defer_join:
            self.join()
        return (*out, offs)

    def join(self):
        """the current stream waits for everything enqueued on the stage's stream so far"""
        done = torch.cuda.Event()
        done.record(self.copy_stream)
        torch.cuda.current_stream(self.device).wait_event(done)  # the kernels that read the arrays are enqueued behind the copies


def _pack(events_list, device, stage=None, defer_join=False):
    if stage is not None:
        packed = stage.pack(events_list, defer_join=defer_join)
        if packed is not None:
            return packed
    xs, ys, ts, ps, offs = [], [], [], [], [0]
    for ev in events_list:
        xs.append(np.asarray(ev["x"], np.float32))
        ys.append(np.asarray(ev["y"], np.float32))
        ts.append(np.asarray(ev["t"], np.float64))
        ps.append(np.asarray(ev["p"], np.float32))
        offs.append(offs[-1] + len(xs[-1]))
    cat = lambda v, dt: torch.from_numpy(np.ascontiguousarray(np.concatenate(v).astype(dt))).to(device)  # noqa: E731
    return cat(xs, np.float32), cat(ys, np.float32), cat(ts, np.float64), cat(ps, np.float32), np.asarray(offs, np.int64)


def events_to_voxel_grid_batch(events_list, input_size, normalize=True, device="cuda", packed=None):
    """list of B event dicts -> voxel grids [B,bins,H,W] (fp32, on `device`).  packed: the result of `_pack` for these events
    (a caller that also needs the events mask packs and uploads the arrays once)."""
    bins, H, W = (int(v) for v in input_size)
    B = len(events_list)
    x, y, t, p, offs = packed if packed is not None else _pack(events_list, device)
    L = N.lib()
    grid = torch.empty((B, bins, H, W), dtype=torch.float32, device=device)
    ws = torch.empty(L.einx_voxel_ws_bytes(B, bins, H, W, int(offs[-1])), dtype=torch.uint8, device=device)
    check(L.einx_voxel_grid(N._ptr(x), N._ptr(y), N._ptr(t), N._ptr(p), offs.ctypes.data_as(ctypes.c_void_p), B, bins, H, W, int(normalize),
                            N._ptr(grid), N._ptr(ws), ws.numel(), N._stream(grid)), "einx_voxel_grid")
    return grid


def events_to_voxel_grid(events, input_size, normalize=True, device="cuda"):
    """Drop-in for datasets/representations.py:67-124 (one sample): returns [bins,H,W].
    Unlike the reference it does not modify the `events` dict in place."""
    return events_to_voxel_grid_batch([events], input_size, normalize, device)[0]


def events_mask_batch(events_list, resolution, device="cuda", packed=None):
    """`draw_events_accumulation_image(events, (W,H)) > 0` for B samples -> bool [B,1,H,W]."""
    W, H = (int(v) for v in resolution)
    B = len(events_list)
    x, y, _, _, offs = packed if packed is not None else _pack(events_list, device)
    L = N.lib()
    mask = torch.empty((B, 1, H, W), dtype=torch.uint8, device=device)
    ws = torch.empty(L.einx_events_ws_bytes(B, H, W), dtype=torch.uint8, device=device)
    check(L.einx_events_mask(N._ptr(x), N._ptr(y), offs.ctypes.data_as(ctypes.c_void_p), B, H, W, N._ptr(ws), N._ptr(mask), N._stream(mask)),
          "einx_events_mask")
    return mask.view(torch.bool)


def events_representation_batch(events_list, input_size, normalize=True, device="cuda", stage=None, on_stage_stream=False):
    """voxel grids [B,bins,H,W] and events masks [B,1,H,W] of B samples from ONE host-side packing and upload of the raw
    event arrays (what test_events-image_same-time.py:130-140 builds per sample with two passes over the events).
    stage: an EventStage -- the upload goes through its page-locked arrays without blocking the host.
    on_stage_stream: the two representation kernels are enqueued on the stage's stream behind the copies as well (an evaluation
    loop: they then run beside the previous batch's forward); the current stream waits for them before it goes on."""
    bins, H, W = (int(v) for v in input_size)
    if stage is not None and on_stage_stream:
        packed = _pack(events_list, device, stage, defer_join=True)
        cur = torch.cuda.current_stream(stage.device)
        with torch.cuda.stream(stage.copy_stream):
            grid = events_to_voxel_grid_batch(events_list, input_size, normalize, device, packed=packed)
            mask = events_mask_batch(events_list, (W, H), device, packed=packed)
        for t in (grid, mask):
            t.record_stream(cur)  # allocated on the stage's stream, consumed on the caller's
        stage.join()
        return grid, mask
    packed = _pack(events_list, device, stage)
    return (events_to_voxel_grid_batch(events_list, input_size, normalize, device, packed=packed),
            events_mask_batch(events_list, (W, H), device, packed=packed))
