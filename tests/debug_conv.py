import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from helpers import load_pkg, synth
from oracle import oracle
import test_gpu_parity as T
pkg = load_pkg()
for shape in T.CONV_SHAPES:
    cin, cout, H, W, ks, relu, bn, pool, fold = shape
    seed = 1000 + cin * 7 + cout
    B = 2
    Hs, Ws = (fold[2], fold[3]) if fold else (H, W)
    x = synth.normalish(seed, (B, cin, Hs, Ws))
    w = synth.synth_param("c.weight", (cout, cin, ks, ks), seed)
    b = synth.uniform(seed + 1, (cout,), -0.5, 0.5)
    bnp = None; scale = shift = None
    if bn:
        g, be = synth.uniform(seed + 2, (cout,), 0.5, 1.5), synth.uniform(seed + 3, (cout,), -0.3, 0.3)
        mu, var = synth.uniform(seed + 4, (cout,), -0.3, 0.3), synth.uniform(seed + 5, (cout,), 0.5, 1.5)
        g[0] = -g[0]
        scale, shift = oracle.bn_fold(g, be, mu, var)
        bnp = (T._t(g), T._t(be), T._t(mu), T._t(var), 1e-5)
    xin = x
    if fold:
        h0, w0 = fold[0], fold[1]
        xin = oracle.pad_replicate(x, (w0, W - Ws - w0, h0, H - Hs - h0))
    exp = oracle.conv_block(xin, w, b, scale, shift, relu=relu, pool=pool)
    layer = pkg.native.ConvLayer(T._t(w), T._t(b), bnp, relu=relu, pool=pool)
    got = T._np(layer(T._t(x), fold=(fold[0], fold[1], H, W) if fold else None))
    d = np.abs(got - exp)
    bad = np.argwhere(got != exp)
    print(shape[:8], "nbad", len(bad), "of", got.size, "maxdiff", d.max(), "first bad", bad[:3].tolist())
    if bn and len(bad):
        sc = layer.scale.cpu().numpy(); sh = layer.shift.cpu().numpy()
        print("   scale equal:", np.array_equal(sc, scale), "shift equal:", np.array_equal(sh, shift), np.abs(sc-scale).max(), np.abs(sh-shift).max())
