"""Which samples of the depth-17 glass frame differ between the HIP path and the oracle (tests/test_gbdpt_gpu.py::test_paths_deeper_than_twelve...):
per-sample probe over the whole small frame, ray counts and values."""
import sys
import numpy as np
sys.path.insert(0, ".")
from gradientdomain_mitsuba_amd import scenes
import gradientdomain_mitsuba_amd.gpt as G
import gradientdomain_mitsuba_amd.gbdpt as B
from oracle import gpt_oracle as go
name, md, rr = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("glass", 17, 16)
W, H, spp = 24, 18, 2
sc = scenes.veach_bidir(W, H, specular=True) if name == "veach_specular" else scenes.cornell_box(W, H, name)
S, O = G.Scene(sc), go.Scene(sc)
integ = B.GBDPTIntegrator(maxDepth=md, rrDepth=rr)
cfg, ocfg = integ.config(spp), go.gbdpt_config(maxDepth=md, rrDepth=rr, spp=spp)
F = B.Film(S)
integ.renderBlock(S, F, cfg, (0, 0, W, H))
blk, lgt = F.accum(); st = F.stats()
ob, ol, oc = O.gbdpt_render(ocfg)
print("film rays", (st["raysTraced"], st["shadowRaysTraced"]), (oc["raysTraced"], oc["shadowRaysTraced"]), "block %.3e light %.3e" % (np.abs(blk - ob).max() / np.abs(ob).max(), np.abs(lgt - ol).max() / np.abs(ol).max()))
tot = [0, 0]
for py in range(H):
    for px in range(W):
        for s in range(spp):
            a = integ.evaluate_sample(S, cfg, px, py, s); b = O.gbdpt_sample(ocfg, px, py, s)
            tot[0] += a["raysTraced"]; tot[1] += b["raysTraced"]
            ra, rb = (a["raysTraced"], a["shadowRaysTraced"]), (b["raysTraced"], b["shadowRaysTraced"])
            sc_ = max(np.abs(b["primal"]).max(), np.abs(b["gradients"]).max(), 1e-300)
            dv = max(np.abs(a["primal"] - b["primal"]).max(), np.abs(a["gradients"] - b["gradients"]).max()) / sc_
            if ra != rb or dv > 1e-9 or a["light"].shape != b["light"].shape:
                print(px, py, s, "rays", ra, rb, "rel diff %.3e" % dv, "light", a["light"].shape, b["light"].shape, flush=True)
print("probe totals", tot)
