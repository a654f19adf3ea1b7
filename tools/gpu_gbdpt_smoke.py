"""A quick look at the general form of the G-BDPT sampler on small scenes before a whole GPU suite is spent on it: films against the oracle, with progress
printed as it goes (run it under `timeout`: a kernel that never ends shows as the last line printed).
  gpurun -- 'timeout -s KILL 300 python tools/gpu_gbdpt_smoke.py'"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from gradientdomain_mitsuba_amd import scenes
import gradientdomain_mitsuba_amd.gpt as G
import gradientdomain_mitsuba_amd.gbdpt as B
from oracle import gpt_oracle as go

for name, W, H, spp, md in (("glass", 24, 18, 1, 5), ("glass", 40, 30, 2, 7), ("glossy", 40, 30, 2, -1), ("nearspecular", 40, 30, 2, 6), ("veach_specular", 64, 36, 2, -1)):
    sc = scenes.veach_bidir(W, H, specular=True) if name == "veach_specular" else scenes.cornell_box(W, H, name)
    print(name, W, H, spp, md, "...", flush=True)
    S, O = G.Scene(sc), go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md)
    F = B.Film(S)
    t0 = time.time()
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H)); F.sync()
    print("  rendered in %.2f s" % (time.time() - t0), F.stats(), F.chain_stats(), flush=True)
    blk, lgt = F.accum()
    ob, ol, oc = O.gbdpt_render(go.gbdpt_config(maxDepth=md, spp=spp))
    st = F.stats()
    print("  rays", (st["raysTraced"], st["shadowRaysTraced"]), (oc["raysTraced"], oc["shadowRaysTraced"]),
          "block %.2e light %.2e" % (np.abs(blk - ob).max() / np.abs(ob).max(), np.abs(lgt - ol).max() / max(np.abs(ol).max(), 1e-300)), flush=True)
    F.close(); S.close(); O.close()
print("done", flush=True)
