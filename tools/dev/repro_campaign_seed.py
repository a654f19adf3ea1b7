"""One probe of tools/gpu_fuzz_campaign.py's seed, in full precision: python tools/dev/repro_campaign_seed.py SEED PX PY S [cpu]
(`cpu`: the oracle only -- no GPU needed)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from gradientdomain_mitsuba_amd import scenes
from oracle import gpt_oracle as go
seed, px, py, s = (int(a) for a in sys.argv[1:5])
cpu = len(sys.argv) > 5
rng = np.random.default_rng(seed)
W, H = int(rng.integers(17, 44)), int(rng.integers(9, 34))
kind = "random"
kw = dict(seed=seed, environment=(0.5, 0.7, 0.9) if seed % 3 == 0 else None)
if seed % 5 == 1:
    kind = "smooth" if seed % 2 else "bent"; kw = dict(environment=kw["environment"])
if seed % 5 == 2:
    kw["point_light"] = ((float(rng.uniform(100, 450)), float(rng.uniform(200, 500)), float(rng.uniform(100, 450))), (4e4, 3e4, 2e4), bool(seed % 2))
if seed % 7 == 0:
    sc = scenes.atrium(W, H, columns=int(rng.integers(4, 12)), segments=int(rng.integers(6, 16)))
else:
    sc = scenes.cornell_box(W, H, kind, **kw)
if seed % 9 == 4:
    sc.thinlens = (float(rng.uniform(2.0, 60.0)), float(rng.uniform(300.0, 1500.0))) if seed % 7 else (float(rng.uniform(0.01, 0.3)), float(rng.uniform(2.0, 30.0)))
if seed % 11 == 3:
    sc.rfilter = scenes.RFILTER_DEFAULTS[1 + seed % 5]
md = int(rng.choice([-1, 2, 3, 5, 9])); rr = int(rng.choice([1, 3, 5])); strict = bool(rng.random() < 0.35); thr = float(rng.choice([0.001, 0.02, 0.0]))
spp = int(rng.integers(1, 7))
print("scene", W, H, kind, kw, "md", md, "rr", rr, "strict", strict, "thr", thr, "spp", spp, "thinlens", getattr(sc, "thinlens", None), "rfilter", getattr(sc, "rfilter", None))
print("materials", getattr(sc, "materials", None))
O = go.Scene(sc); ocfg = go.config(maxDepth=md, rrDepth=rr, strictNormals=strict, spp=spp, shiftThreshold=thr)
o = O.evaluate_point(ocfg, px, py, s)
np.set_printoptions(precision=17)
print("oracle", {k: o[k] for k in ("veryDirect", "throughput", "gradients", "neighbours", "raysTraced", "shadowRaysTraced")})
if not cpu:
    from gradientdomain_mitsuba_amd import gpt as G
    S = G.Scene(sc); integ = G.GradientPathIntegrator(maxDepth=md, rrDepth=rr, strictNormals=strict, shiftThreshold=thr); cfg = integ.config(spp)
    g = S.evaluate_point(cfg, px, py, s)
    print("hip", {k: g[k] for k in ("veryDirect", "throughput", "gradients", "neighbours", "raysTraced", "shadowRaysTraced")})
    for k in ("veryDirect", "throughput", "gradients", "neighbours"):
        print(k, "rel diff", np.abs(g[k] - o[k]) / np.abs(o[k]).clip(1e-300))
