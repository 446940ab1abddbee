"""Round-4 GPU parity tests (-m gpu): slow-converging NMS maps (device finisher bound, host retry), build-flag guard."""
import numpy as np
import pytest
import torch

from helpers import load_pkg

pytestmark = pytest.mark.gpu
pkg = load_pkg()
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _np(t):
    return t.detach().cpu().numpy()


def _ramp(H, W):
    s = (np.arange(W, dtype=np.float32)[None, :] + 1) / np.float32(W + 1)
    return np.broadcast_to(s, (H, W)).copy()[None, None]


def _serpentine(H, W):
    """values increasing along a boustrophedon path through EVERY pixel: each maximum is decided only after the one that
    follows it on the path, 380 passes on 96x96"""
    m = np.zeros((H, W), np.float32)
    v = 1
    for y in range(2, H - 2):
        for x in (range(W) if y % 2 == 0 else range(W - 1, -1, -1)):
            m[y, x] = np.float32(v) / np.float32((H - 4) * W + 1)
            v += 1
    return m[None, None]


def test_nms_slow_converging_maps_finish_on_device_or_through_the_retry(oracle):
    """fast_nms' fix-point (detector_util.py:286-335) on maps that need hundreds of passes.
    * a monotone ramp on a 24x1600 map needs 324 passes: more than the 8 wide + 256 finisher passes of round 3, fewer than the
      finisher's bound of max(256, Hp + Wp) -> converges inside ONE einx_detect call, no host round trip;
    * a serpentine ramp on 96x96 needs 380 > 8 + 256: einx_detect reports not_converged, the callers' retry (budget x4 per
      round, detector_util.fast_nms here, EIM / NativeExtractor.forward alike) reaches the oracle's fix-point."""
    from importlib import import_module
    du = import_module(pkg.__name__ + ".core.modules.utils.detector_util")
    s = _ramp(24, 1600)
    exp_nms, _, _, _, iters = oracle.detect_post(s.copy(), 0, 4, 0, 0.0)
    assert 264 < iters < 1624
    d = pkg.native.detect(_t(s[:, 0]), top_k=0, radius=4, det_thr=float("-inf"), cap=1, nms_iters=8)
    assert int(d.not_converged.sum()) == 0
    assert np.array_equal(_np(d.nms).reshape(exp_nms.shape), exp_nms)
    s = _serpentine(96, 96)
    exp_nms, _, _, _, iters = oracle.detect_post(s.copy(), 0, 4, 0, 0.0)
    assert iters > 8 + 256
    d = pkg.native.detect(_t(s[:, 0]), top_k=0, radius=4, det_thr=float("-inf"), cap=1, nms_iters=8)
    assert int(d.not_converged.sum()) == 1  # honest: the bounded finisher gave up
    got = du.fast_nms(_t(s), nms_dist=4)
    assert np.array_equal(_np(got).reshape(exp_nms.shape), exp_nms)
    # the same map inside a batch next to an ordinary one: only the slow image is redone, both equal the oracle
    both = np.concatenate([s, _ramp(96, 96)], 0)
    exp_b, _, _, _, _ = oracle.detect_post(both.copy(), 0, 4, 0, 0.0)
    assert np.array_equal(_np(du.fast_nms(_t(both), nms_dist=4)).reshape(exp_b.shape), exp_b)


def test_shipped_library_is_not_a_timing_only_build():
    assert pkg.native.lib().einx_build_flags() == b""
