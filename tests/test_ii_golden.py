"""ImageImageMatcher (reference core/modules/ImageImageMatcher.py:75-82) against fixtures generated from the reference
(tests/golden/ii.npz, gen_golden.py::gen_ii): the image extractor on two image batches, a score mask on the FIRST one only
(SuperPointv1 applies it without dilation, superpoint_extractor.py:411-412; SiLKModel.forward(image, *args, **kwargs) ignores it,
silk_extractor.py:177 -- the fixture pins both), then the frozen matcher.
CPU: the oracle against the fixture.  GPU: the kernels against the oracle (bit for bit) and against the fixture."""
import numpy as np
import pytest
import torch

from helpers import Golden, load_pkg, split, state_dict_for, sub_dict, synth
from test_oracle_golden import _check_feats

pkg = load_pkg()
II = Golden("ii")
FTOL = 1e-4


def _inputs(c):
    img0 = synth.synth_image(c["iseed"], c["B"], c["H"], c["W"])
    img1 = synth.synth_image(c["iseed"] + 1000, c["B"], c["H"], c["W"])
    mask0 = synth.uniform01(c["mseed"], (c["B"], 1, c["H"], c["W"])) < np.float32(0.7)
    return img0, img1, mask0


def _oracle_sides(oracle, c, sd):
    it = c["cfg"]["image_extractor"]["type"]
    icfg = c["cfg"]["image_extractor"][it]
    img0, img1, mask0 = _inputs(c)
    kw = dict(top_k=icfg["detection_top_k"], radius=icfg["nms_radius"], border=icfg["remove_borders"], det_thr=icfg["detection_threshold"],
              scale=icfg["descriptor_scale_factor"])
    sub = sub_dict(sd, "image_extractor.extractor.")
    return oracle.extractor_forward(it, sub, img0.copy(), mask0, **kw), oracle.extractor_forward(it, sub, img1.copy(), None, **kw)


def _check_matches_vs_reference(name, got_m0, B):
    ref0 = split(II[f"{name}.m.matches0"], II[f"{name}.m.matches0.lens"])
    ndiff = sum(int((got_m0[b] != ref0[b]).sum()) for b in range(B))
    assert ndiff <= 2, f"{ndiff} match indices differ from the reference"  # arg-max near-ties only (see test_oracle_golden.py)
    return ndiff


@pytest.mark.parametrize("name", list(II.cases))
def test_oracle_image_image_vs_reference(oracle, name):
    c = II.cases[name]
    sd = state_dict_for(c, II)
    f0, f1 = _oracle_sides(oracle, c, sd)
    _check_feats(f"{name}.f0", f0, II)
    _check_feats(f"{name}.f1", f1, II)
    masked_out = f0["score"][~np.broadcast_to(_inputs(c)[2], f0["score"].shape)]
    if c["image_type"] == "silk":  # the reference's SiLKModel.forward swallows the mask (silk_extractor.py:177): nothing is zeroed
        assert int((masked_out != 0).sum()) > 0
    else:
        assert int((masked_out != 0).sum()) == 0
    m0 = [oracle.mnn(f0["sparse_descriptors"][b], f1["sparse_descriptors"][b], want_la=False)["matches0"] for b in range(c["B"])]
    _check_matches_vs_reference(name, m0, c["B"])
    lens = II[f"{name}.m.matched_kpts0.lens"].tolist()
    assert all(abs(int((m0[b] > -1).sum()) - lens[b]) <= 2 for b in range(c["B"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(II.cases))
def test_gpu_image_image_vs_oracle_and_reference(oracle, name):
    assert torch.cuda.is_available(), "needs a HIP device"
    dev = "cuda:0"
    c = II.cases[name]
    cfg = pkg.configs.to_attr(c["cfg"])
    cfg.name = "ImageImageMatcher"
    model = pkg.build_model(cfg, dev, None)
    sd = state_dict_for(c, II)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model.eval()
    model.image_extractor.extractor.dense_outputs = False
    img0, img1, mask0 = _inputs(c)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    f0, f1, m = model(t(img0), t(img1), mask=t(mask0))
    o0, o1 = _oracle_sides(oracle, c, sd)
    as_np = lambda f: {k: (v.cpu().numpy() if torch.is_tensor(v) else [x.cpu().numpy() for x in v]) for k, v in f.items()}  # noqa: E731
    g0, g1 = as_np(f0), as_np(f1)
    for got, exp in ((g0, o0), (g1, o1)):
        for k in ("backbone_feats", "logits", "raw_descriptors", "score", "nms"):
            assert np.array_equal(got[k], exp[k]), f"{k} differs from the oracle"
        for b in range(c["B"]):
            assert np.array_equal(got["sparse_positions"][b], exp["sparse_positions"][b])
            assert np.array_equal(got["sparse_descriptors"][b], exp["sparse_descriptors"][b])
    _check_feats(f"{name}.f0", g0, II)
    _check_feats(f"{name}.f1", g1, II)
    m0 = [m["matches0"][b].cpu().numpy()[0] for b in range(c["B"])]
    for b in range(c["B"]):
        assert np.array_equal(m0[b], oracle.mnn(o0["sparse_descriptors"][b], o1["sparse_descriptors"][b], want_la=False)["matches0"])
    if _check_matches_vs_reference(name, m0, c["B"]) == 0:
        for key in ("matched_kpts0", "matched_kpts1"):
            exp = split(II[f"{name}.m.{key}"], II[f"{name}.m.{key}.lens"])
            for b in range(c["B"]):
                np.testing.assert_allclose(m[key][b].cpu().numpy(), exp[b], atol=FTOL, rtol=0)
    la_shapes = II[f"{name}.m.la_shapes"]
    for b in range(c["B"]):
        assert list(m["log_assignment"][b].shape) == la_shapes[b].tolist()
