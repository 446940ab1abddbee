"""Common machinery of the native extractor front-ends."""
import torch
from torch import nn

from ..._native import on_input_device
from ... import _native as N
from ..._extract import ExtractorEngine
from .net.vgg import block_spec


class NativeExtractor(nn.Module):
    """nn.Module shell: owns the parameters (reference state_dict names), builds the kernel-native
    layer images lazily, and exposes both the reference `forward` (dict, one host sync) and
    `extract_batched` (device-side result, no sync)."""

    kind = None
    cell_size = 8
    uses_batchnorm = True
    dilate_mask = False
    input_div = 0.0  # SuperPointv1: the input tensor is divided by 255 IN PLACE inside the forward (reference quirk)
    padding = 1      # 0: un-padded 3x3 convolutions (cell-1 networks; keypoints mapped back by +9)

    def _init_common(self, nms_radius, detection_top_k, detection_threshold, remove_borders, ordering, descriptor_scale_factor,
                     learnable_descriptor_scale_factor):
        if ordering not in ("xy", "yx"):
            raise AssertionError(ordering)
        self.nms_radius = nms_radius
        self.detection_top_k = detection_top_k
        self.detection_threshold = detection_threshold
        self.remove_borders = remove_borders
        self.ordering = ordering
        self.descriptor_scale_factor = nn.parameter.Parameter(torch.tensor(float(descriptor_scale_factor)),
                                                              requires_grad=learnable_descriptor_scale_factor)
        # "lazy" (default): the reference-complete dict whose dense maps (92 MB / image) are computed when first read
        # (_extract.FeatsDict); True: computed in the forward; False: keys absent
        self.dense_outputs = "lazy"
        self._engine = None
        self._scale_host = None
        self._sig = None
        self._sig_tensors = None
        # runs for THIS module also when a parent's load_state_dict recurses into it (EIM, the
        # Extractor wrappers, ImageImageMatcher), which never calls the child's own load_state_dict
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.refresh())

    # -- cache invalidation: anything that moves or replaces parameters drops the native images
    def _apply(self, fn, *a, **k):
        self._engine = self._scale_host = self._sig_tensors = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = self._scale_host = self._sig_tensors = None
        return super().load_state_dict(*a, **k)

    def _signature(self):
        """(storage, version) of every parameter and buffer: in-place edits made THROUGH the parameter (`with no_grad():
        p.add_(..)` / `p.copy_(..)`, optimiser steps, `load_state_dict`) bump `p._version`, `.to()` changes the storage.
        NOT detected HERE: edits through `p.data` (`p.data.copy_(w)`, `p.data.mul_(..)`, `p.data[i, j] = v`): `.data` is an alias
        with its OWN version counter, so `p._version` stays put.  Those are caught by the device-side content watch instead
        (round 5: a hash over EVERY word of every fp32 parameter / buffer, checked by every forward, which then rebuilds the
        images and runs again).  REPLACING a Parameter object is invisible to both (the watch holds the old storage);
        `refresh()` covers that (tests/test_host_cpu.py::test_data_alias_edits_need_refresh,
        tests/test_boundary_gpu.py::test_data_edits_of_weights_take_effect_at_the_next_forward).  The flat tensor list is cached (walking the module
        tree costs 50-450 us per call); `_apply`, `load_state_dict` and `refresh()` drop it."""
        ts = self._sig_tensors
        if ts is None:
            ts = self._sig_tensors = list(self.parameters()) + list(self.buffers())
        return tuple((t.data_ptr(), t._version) for t in ts)

    def refresh(self):
        """Drop the kernel-native weight images.  Needed after REPLACING Parameter objects; harmless otherwise (in-place edits
        are detected by `_signature`, `.data` edits -- down to a single element -- by the content watch at the next forward)."""
        self._engine = self._scale_host = self._sig_tensors = None

    def _spec(self, block):
        """(conv, BatchNorm or None, relu) of one entry of `_stacks()` (families that list their layers differently override it
        together with `_layer`)"""
        return block_spec(block)[:3]

    def _layer(self, block, pool=False):
        conv, bn, relu, pool = block_spec(block, pool)
        return N.ConvLayer(conv.weight, conv.bias, None if bn is None else N.bn_tuple(bn), relu=relu, pool=pool)

    def _merged_head0(self, det, desc):
        """The two heads' first 3x3 layers read the same backbone features: as ONE layer (output channels of the detector's first,
        then the descriptor's) they are one launch instead of two on a single pair's latency-bound chain (einx.h:
        einx_extractor_desc::merged_head0; used for B == 1).  None when the heads are not two layers of the same structure."""
        if len(det) != 2 or len(desc) != 2:
            return None
        (ca, ba, ra), (cb, bb_, rb) = self._spec(det[0]), self._spec(desc[0])
        same = (ca.kernel_size == cb.kernel_size == (3, 3) and ca.in_channels == cb.in_channels and ra == rb and (ba is None) == (bb_ is None)
                and (ca.bias is None) == (cb.bias is None) and ca.out_channels % 64 == 0 and (ba is None or ba.eps == bb_.eps))
        if not same:
            return None
        cat = lambda a, b: torch.cat([a.detach(), b.detach()], 0)  # noqa: E731
        bn = None if ba is None else (cat(ba.weight, bb_.weight), cat(ba.bias, bb_.bias), cat(ba.running_mean, bb_.running_mean),
                                      cat(ba.running_var, bb_.running_var), ba.eps)
        return N.ConvLayer(cat(ca.weight, cb.weight), None if ca.bias is None else cat(ca.bias, cb.bias), bn, relu=ra, pool=False)

    def _stacks(self):
        """-> (backbone blocks [(block, pool)], detector blocks, descriptor blocks)"""
        raise NotImplementedError

    def engine(self):
        sig = self._signature()
        if sig != self._sig:
            self._engine = self._scale_host = None
            self._sig = sig
        if self._engine is None:
            eng = ExtractorEngine(self.kind, top_k=self.detection_top_k, radius=self.nms_radius, border=self.remove_borders,
                                  det_thr=self.detection_threshold, ordering=self.ordering, cell=self.cell_size)
            eng.valid_crop = 9 if (getattr(self, "padding", 1) == 0 and self.cell_size == 1) else 0
            bb, det, desc = self._stacks()
            eng.backbone = [self._layer(b, p) for b, p in bb]
            eng.det_head = [self._layer(b) for b in det]
            eng.desc_head = [self._layer(b) for b in desc]
            eng.merged_head0 = self._merged_head0(det, desc)
            eng.watch = N.ParamWatch(self._sig_tensors)  # `.data` edits: seen by content (round 4, inside einx_extract), see refresh()
            self._engine = eng
        # the reference's forward reads these attributes at every call (EventExtractors.py:545-556, superpoint_extractor.py:388-406):
        # assigning `extractor.detection_top_k = 500` between two forwards takes effect at the next one (the native handle is
        # keyed on them and rebuilt when one changes)
        eng = self._engine
        eng.top_k, eng.radius, eng.border = self.detection_top_k, self.nms_radius, self.remove_borders
        eng.det_thr, eng.ordering = self.detection_threshold, self.ordering
        return eng

    def _prepare_input(self, x):
        return x

    def _network_input(self, x, prepared):
        """-> (the tensor the network reads, the divisor einx_extract applies to it IN PLACE).  prepared=True: an in-place input
        scaling has already been applied by an earlier call of the same forward and must not run twice."""
        if not (prepared and self.input_div):
            x = self._prepare_input(x)
        return x, (0.0 if prepared else self.input_div)

    def _pooled_padding0_error(self, x, score_mask):
        """VGGExtractor(padding=0) (pooled, un-padded 3x3 layers) constructs in the reference but no forward of it can complete
        (measured with the reference, tests/golden/gen_pad0_pooled.py): its score map is smaller than the padded input (eight valid
        backbone layers and one head layer under three poolings), so with a mask `score[~score_mask] = 0` fails on the shapes
        (EventExtractors.py:553-554), and without one `filter_sparse_feats` indexes the H*W dense descriptors with a validity mask
        over the smaller score map (:608 -> :499-500).  The drop-in raises the same IndexError with the same shapes, before any launch.
        (The EIM / Extractors front-ends never pass `padding` to this class: Extractors.py:46-58.)"""
        from ..._native import padder_pads
        B, _, H, W = x.shape
        w0, w1, h0, h1 = padder_pads(H, W, 8)
        Hp, Wp = H + h0 + h1, W + w0 + w1

        def head_size(n):
            for _ in range(3):
                n = (n - 4) // 2   # two valid 3x3 layers, then MaxPool2d(2, 2)
            return n - 4 - 2       # l4's two layers, the heads' 3x3 layer
        hs, ws = head_size(Hp) * 8, head_size(Wp) * 8
        if hs <= 0 or ws <= 0:
            raise RuntimeError(f"Calculated padded input size per channel: ({Hp} x {Wp}). Kernel size: (3 x 3). Kernel size can't be greater than "
                               "actual input size")
        if score_mask is not None:
            raise IndexError(f"The shape of the mask [{B}, 1, {Hp}, {Wp}] at index 2 does not match the shape of the indexed tensor "
                             f"[{B}, 1, {hs}, {ws}] at index 2")
        raise IndexError(f"The shape of the mask [{hs * ws}] at index 0 does not match the shape of the indexed tensor "
                         f"[{Hp * Wp}, {self.descriptor_dim}] at index 0")  # dense_descriptors are taken before the unpad (:591-592)

    @on_input_device
    def extract_batched(self, x, score_mask=None, nms_iters=None, dense=None, prepared=False, defer_dense=False):
        """prepared=True: an IN-PLACE input scaling (SuperPointv1's `image /= 255`) has already been applied to `x` by an earlier
        call and must not run twice (redo after a `.data` weight edit); extractors that scale a copy do so again.
        defer_dense=True: the dense descriptor map is left to the caller (`bf.run_dense()` on a stream of its choice)."""
        if self.training and self.uses_batchnorm:
            raise RuntimeError("the native path implements eval-mode BatchNorm (running statistics) only; call .eval() first")
        if getattr(self, "padding", 1) == 0 and self.cell_size == 8:
            self._pooled_padding0_error(x, score_mask)
        x, input_div = self._network_input(x, prepared)
        eng = self.engine()
        if self._scale_host is None:  # one device read per engine build, not one host sync per forward
            self._scale_host = float(self.descriptor_scale_factor.detach())
        scale = self._scale_host
        # (the engine's weight watch rides on the call and reports through its own device word, bf.det.stale)
        return eng.run(x, score_mask, scale=scale, dilate_mask=self.dilate_mask, dense=self.dense_outputs if dense is None else dense,
                       nms_iters=nms_iters, input_div=input_div, defer_dense=defer_dense)

    @on_input_device
    def forward(self, x, score_mask=None, **kwargs):
        bf = self.extract_batched(x, score_mask)
        for _ in range(12):
            st = getattr(bf.det, "stale", None)
            flat = torch.cat([bf.det.counts, bf.det.not_converged] + ([] if st is None else [st])).cpu()
            B = bf.det.counts.shape[0]
            host = flat[:2 * B].view(2, B)
            if bool(flat[2 * B:].any()):
                # a weight was edited through `.data` since the native images were built: rebuild them and run again (an input
                # that was scaled in place -- SuperPointv1's `/= 255` -- is not scaled twice)
                self.refresh()
                bf = self.extract_batched(x, score_mask, prepared=True)
                continue
            if not bool(host[1].any()):
                self.engine().note_converged()
                return bf.materialize(host[0].tolist())
            self.engine().redetect(bf, self.engine().grow_nms_iters())  # NMS fix-point needs more passes
        raise RuntimeError("einx: the NMS fix-point did not converge within the maximum pass budget")
