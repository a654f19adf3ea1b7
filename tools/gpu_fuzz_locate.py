"""Locate the pixel / sample where a fuzz-campaign seed differs between the HIP tracer and the oracle: python tools/gpu_fuzz_locate.py SEED"""
import sys, copy
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, scenes
from oracle import gpt_oracle as go

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
W, H = int(rng.integers(17, 44)), int(rng.integers(9, 34))
kind = "random"
kw = dict(seed=seed, environment=(0.5, 0.7, 0.9) if seed % 3 == 0 else None)
if seed % 5 == 1:
    kind = "smooth" if seed % 2 else "bent"; kw = dict(environment=kw["environment"])
if seed % 5 == 2:
    kw["point_light"] = ((float(rng.uniform(100, 450)), float(rng.uniform(200, 500)), float(rng.uniform(100, 450))), (4e4, 3e4, 2e4), bool(seed % 2))
sc = scenes.atrium(W, H, columns=int(rng.integers(4, 12)), segments=int(rng.integers(6, 16))) if seed % 7 == 0 else scenes.cornell_box(W, H, kind, **kw)
md = int(rng.choice([-1, 2, 3, 5, 9])); rr = int(rng.choice([1, 3, 5])); strict = bool(rng.random() < 0.35); thr = float(rng.choice([0.001, 0.02, 0.0]))
spp = int(rng.integers(1, 7))
print("seed %d: %s %dx%d spp %d maxDepth %d rrDepth %d strict %s threshold %g" % (seed, kind, W, H, spp, md, rr, strict, thr))
S = G.Scene(sc); O = go.Scene(sc)
integ = G.GradientPathIntegrator(maxDepth=md, rrDepth=rr, strictNormals=strict, shiftThreshold=thr)
cfg = integ.config(spp); ocfg = go.config(maxDepth=md, rrDepth=rr, strictNormals=strict, spp=spp, shiftThreshold=thr)
for py in range(H):
    for px in range(W):
        gs = [S.evaluate_point(cfg, px, py, s) for s in range(spp)]
        gr = (sum(g["raysTraced"] for g in gs), sum(g["shadowRaysTraced"] for g in gs))
        _, orays = O.render(ocfg, rect=(px, py, px + 1, py + 1))
        if gr != tuple(orays):
            print("pixel (%d, %d): HIP rays %r, oracle %r" % (px, py, gr, tuple(orays)))
            for s in range(spp):
                o = O.evaluate_point(ocfg, px, py, s)
                g = gs[s]
                d = max(float(np.abs(g[k] - o[k]).max()) for k in ("veryDirect", "throughput", "gradients", "neighbours"))
                print("  sample %d: HIP rays %d + %d, oracle %d + %d, depth %d, max abs output difference %.3e" % (s, g["raysTraced"], g["shadowRaysTraced"], o["raysTraced"], o["shadowRaysTraced"], g["depth"], d))
                if d > 0:
                    for k in ("throughput", "gradients", "neighbours"):
                        print("   ", k, "HIP", np.array2string(np.asarray(g[k]).ravel(), precision=10), "\n    ", " " * len(k), "ora", np.array2string(np.asarray(o[k]).ravel(), precision=10))
