import sys, os, copy
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, scenes
from oracle import gpt_oracle as go
src = open('tools/gpu_fuzz_campaign.py').read()
a = src.index("    rng = np.random.default_rng(seed)"); b = src.index("    for _ in range(40):")
body = "\n".join(l[4:] for l in src[a:b].split("\n"))
seed = int(sys.argv[1]); px, py, s = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ns = dict(np=np, G=G, go=go, scenes=scenes, seed=seed)
exec(body, ns)
S, O, cfg, ocfg, sc = ns["S"], ns["O"], ns["cfg"], ns["ocfg"], ns["sc"]
g, o = S.evaluate_point(cfg, px, py, s), O.evaluate_point(ocfg, px, py, s)
for key in ("veryDirect", "throughput", "gradients", "neighbours"):
    d = np.abs(np.asarray(g[key]) - np.asarray(o[key]))
    print(key, "max abs diff %.3e" % d.max(), "max rel %.3e" % (d / (np.abs(np.asarray(o[key])) + 1e-300)).max())
print(np.asarray(g["gradients"]) - np.asarray(o["gradients"]))
# oracle spread under more perturbations (ulps up to 64)
v0 = np.asarray(sc.verts, np.float64).reshape(-1, 3)
spread = np.zeros_like(np.asarray(o["gradients"]))
for ax in (None, 0, 1, 2):
    for k in (1, 2, 3, 4, 6, 8, 16, 32, 64, -1, -2, -3, -4, -8, -16, -64):
        v = v0.copy()
        if ax is None: v *= 1 + k * 2.0 ** -52
        else: v[:, ax] *= 1 + k * 2.0 ** -52
        sc2 = copy.deepcopy(sc); sc2.verts = v.reshape(np.asarray(sc.verts).shape)
        o2 = go.Scene(sc2).evaluate_point(ocfg, px, py, s)
        spread = np.maximum(spread, np.abs(np.asarray(o2["gradients"]) - np.asarray(o["gradients"])))
print("oracle spread (ulp scalings up to 64):"); print(spread)
