"""Locates a film difference found by tools/gpu_fuzz_features.py: worst buffer / pixel, the sample behind it, and the oracle's own
sensitivity of that sample to few-ulp scalings of the geometry (the rounding-noise floor of an ill-conditioned sample)."""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.argv, seed = sys.argv[:1], int(sys.argv[1])
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("ff", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_fuzz_features.py"))
ff = importlib.util.module_from_spec(spec); spec.loader.exec_module(ff)
from gradientdomain_mitsuba_amd import gpt as G
from oracle import gpt_oracle as go
sc, W, H, spp, md, strict, variant, what = ff.make(seed)
print("seed", seed, variant, what, (W, H, spp), "maxDepth", md, "strict", strict)
S, O = G.Scene(sc), go.Scene(sc)
integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
cfg, ocfg = integ.config(spp), go.config(maxDepth=md, spp=spp, strictNormals=strict)
F = G.Film(S); integ.renderBlock(S, F, cfg, (0, 0, W, H)); acc = F.accum()
oacc, _ = O.render(ocfg)
d = np.abs(acc - oacc)[..., :3].max(-1)
b, y, x = np.unravel_index(np.argmax(d / (np.abs(oacc).reshape(5, -1).max(1)[:, None, None] + 1e-300)), d.shape)
print("worst: buffer", G.BUFFER_NAMES[b], "pixel", (x, y), "abs diff", d[b, y, x], "buffer scale", np.abs(oacc[b]).max())
x, y = int(x), int(y)
cands = [(x, y), (x - 1, y), (x + 1, y), (x, y - 1), (x, y + 1)]
for (px, py) in cands:
    if not (0 <= px < W and 0 <= py < H): continue
    for k in range(spp):
        g, o = S.evaluate_point(cfg, px, py, k), O.evaluate_point(ocfg, px, py, k)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            diff = float(np.abs(np.asarray(g[key]) - np.asarray(o[key])).max())
            if diff > 1e-12 * (1 + float(np.abs(np.asarray(o[key])).max())):
                sens = ff.oracle_spread(sc, ocfg, px, py, k, key, o)
                print("  sample", (px, py, k), key, "HIP-oracle diff %.3e" % diff, "value scale %.3e" % float(np.abs(np.asarray(o[key])).max()), "oracle's own spread under ulp scalings %.3e" % sens)
