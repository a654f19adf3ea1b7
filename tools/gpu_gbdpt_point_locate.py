import sys, numpy as np
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import scenes
import gradientdomain_mitsuba_amd.gpt as G
import gradientdomain_mitsuba_amd.gbdpt as B
from oracle import gpt_oracle as go
W, H = 40, 30
sc = scenes.cornell_box(W, H, "twosided")
sc.emitters = [("point", (278.0, 400.0, 279.5), (4e4, 3e4, 2e4)), ("point", (120.0, 90.0, 140.0), (1e4, 2e4, 3e4))]
S, O = G.Scene(sc), go.Scene(sc)
integ = B.GBDPTIntegrator(maxDepth=4, lightImage=True)
cfg, ocfg = integ.config(2), go.gbdpt_config(maxDepth=4, lightImage=True, spp=2)
n = 0
for py in range(H):
    for px in range(W):
        for s in range(2):
            g = integ.evaluate_sample(S, cfg, px, py, s)
            o = O.gbdpt_sample(ocfg, px, py, s)
            if (g["raysTraced"], g["shadowRaysTraced"]) != (o["raysTraced"], o["shadowRaysTraced"]):
                n += 1
                print(px, py, s, 'rays', g["raysTraced"], o["raysTraced"], g["shadowRaysTraced"], o["shadowRaysTraced"], 'primal', g["primal"], o["primal"], 'grad diff', np.abs(g["gradients"] - o["gradients"]).max(), 'light', len(g["light"]), len(o["light"]), 'general', g["general"])
print('mismatches', n)
