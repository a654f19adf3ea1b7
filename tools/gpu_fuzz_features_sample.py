"""Prints one sample of a tools/gpu_fuzz_features.py scene from both sides: python tools/gpu_fuzz_features_sample.py SEED PX PY K."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
seed, px, py, k = (int(v) for v in sys.argv[1:5]); sys.argv = sys.argv[:1]
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("ff", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_fuzz_features.py"))
ff = importlib.util.module_from_spec(spec); spec.loader.exec_module(ff)
from gradientdomain_mitsuba_amd import gpt as G
from oracle import gpt_oracle as go
np.set_printoptions(precision=17, linewidth=200)
sc, W, H, spp, md, strict, variant, what = ff.make(seed)
print("seed", seed, variant, what, (W, H, spp), "maxDepth", md, "strict", strict)
S, O = G.Scene(sc), go.Scene(sc)
integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
for depth in ([md] if md < 0 else []) + list(range(2, (md if md > 0 else 12) + 1)):
    cfg, ocfg = integ.__class__(maxDepth=depth, strictNormals=strict).config(spp), go.config(maxDepth=depth, spp=spp, strictNormals=strict)
    g, o = S.evaluate_point(cfg, px, py, k), O.evaluate_point(ocfg, px, py, k)
    worst = max(float(np.abs(np.asarray(g[key]) - np.asarray(o[key])).max()) for key in ("veryDirect", "throughput", "gradients", "neighbours"))
    print("maxDepth", depth, "worst abs diff %.3e" % worst)
    if worst > 1e-10 or depth == md:
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            print(" ", key, "\n   HIP   ", np.asarray(g[key]).ravel(), "\n   oracle", np.asarray(o[key]).ravel())
        if worst > 1e-10: break
