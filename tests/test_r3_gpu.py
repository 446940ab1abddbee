"""Round-3 GPU tests (-m gpu): maximum sizes.

One MI355X holds 288 GB, so a caller may hand the path batches whose activation buffers pass 2^31 ELEMENTS (the point where
a 32-bit element index wraps): B = 384 makes conv1a / conv1b write 384 * 64 * 264 * 352 = 2.28e9 floats into one buffer, and
B = 96 makes the dense descriptor map 96 * 256 * 260 * 346 = 2.21e9 floats.  The batch is built from four distinct pairs
repeated, so every pair of the large batch has a known bit-exact answer: the same pair run in a batch of four (which the
other tests pin to the oracle).
"""
import numpy as np
import pytest
import torch

from helpers import load_pkg, synth

pytestmark = pytest.mark.gpu
pkg = load_pkg()
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    assert torch.cuda.is_available(), "these tests need a HIP device"
    yield
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _model(dense_event=False):
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=11)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model.event_extractor.extractor.dense_outputs = dense_event
    model.image_extractor.extractor.dense_outputs = False
    return model


def _four_pairs(seed):
    ev, mask = synth.synth_events(seed, 4, 5)
    img = synth.synth_image(seed, 4)
    return ev, mask, img


def _tiled(a, B):
    return np.concatenate([a] * (B // a.shape[0]), axis=0)


def test_batch_384_activations_beyond_2_31_elements_equal_the_small_batch():
    B = 384
    assert B * 64 * 264 * 352 > 2**31
    model = _model()
    ev, mask, img = _four_pairs(4242)
    ef4, if4, m4 = model(_t(ev), _t(img), _t(mask))
    ef, imf, m = model(_t(_tiled(ev, B)), _t(_tiled(img, B)), _t(_tiled(mask, B)))
    assert len(ef["sparse_positions"]) == B and len(m["matches0"]) == B
    for b in (0, 1, 2, 3, 189, 190, 191, 192, 193, 362, 363, 380, 381, 382, 383):  # 362 is the first image past 2^31 floats of conv1a
        r = b % 4
        for got, exp in ((ef, ef4), (imf, if4)):
            assert torch.equal(got["sparse_positions"][b], exp["sparse_positions"][r]), f"pair {b}: keypoints"
            assert torch.equal(got["sparse_descriptors"][b], exp["sparse_descriptors"][r]), f"pair {b}: descriptors"
            assert torch.equal(got["score"][b], exp["score"][r]), f"pair {b}: score map"
            assert torch.equal(got["coarse_descriptors"][b], exp["coarse_descriptors"][r]), f"pair {b}: coarse descriptors"
        for k in ("matches0", "matches1", "matching_scores0", "matched_kpts0", "matched_kpts1", "log_assignment"):
            assert torch.equal(m[k][b], m4[k][r]), f"pair {b}: {k}"
    assert int(ef["sparse_positions"][383].shape[0]) > 100


def test_batch_96_dense_descriptor_map_beyond_2_31_elements_equals_the_small_batch():
    B = 96
    assert B * 256 * 260 * 346 > 2**31
    model = _model(dense_event=True)
    ev, mask, img = _four_pairs(4343)
    ef4, _, _ = model(_t(ev), _t(img), _t(mask))
    small = [ef4["normalized_descriptors"][r].clone() for r in range(4)]
    small_dd = [ef4["dense_descriptors"][r].clone() for r in range(4)]
    del ef4
    ef, _, _ = model(_t(_tiled(ev, B)), _t(_tiled(img, B)), _t(_tiled(mask, B)))
    nd = ef["normalized_descriptors"]
    assert tuple(nd.shape) == (B, 256, 260, 346)
    for b in (0, 1, 46, 47, 92, 93, 94, 95):  # 93 is the first image past 2^31 floats
        assert torch.equal(nd[b], small[b % 4]), f"image {b}: dense descriptor map"
        assert torch.equal(ef["dense_descriptors"][b], small_dd[b % 4]), f"image {b}: dense descriptor list entry"
    n = torch.linalg.vector_norm(nd[95], dim=0)
    scale = float(model.event_extractor.extractor.descriptor_scale_factor)
    assert float((n - scale).abs().max()) < 1e-4
