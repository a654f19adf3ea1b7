"""G-BDPT with specular chains (round 4): single samples of the device's general form (csrc/gbdpt_general.hip.h) against the oracle.
  python tools/gpu_gbdpt_chain_check.py [variant,variant] [samples]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import scenes
import gradientdomain_mitsuba_amd.gpt as G
import gradientdomain_mitsuba_amd.gbdpt as B
from oracle import gpt_oracle as go

variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["glass", "glossy", "nearspecular"]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for name in variants:
    W, H = 40, 30
    sc = scenes.cornell_box(W, H, name)
    S, O = G.Scene(sc), go.Scene(sc)
    for md, li in ((6, True), (-1, False)):
        integ = B.GBDPTIntegrator(maxDepth=md, lightImage=li)
        cfg, ocfg = integ.config(64), go.gbdpt_config(maxDepth=md, lightImage=li, spp=64)
        rng = np.random.default_rng(17)
        bad = gen = 0
        worst = 0.0
        for _ in range(N):
            px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 64))
            g = integ.evaluate_sample(S, cfg, px, py, s)
            o = O.gbdpt_sample(ocfg, px, py, s)
            gen += g["general"]
            scale = max(np.abs(o["primal"]).max(), np.abs(o["gradients"]).max(), 1e-300)
            e = max(np.abs(g["primal"] - o["primal"]).max(), np.abs(g["gradients"] - o["gradients"]).max()) / scale
            rays = (g["raysTraced"], g["shadowRaysTraced"]) == (o["raysTraced"], o["shadowRaysTraced"])
            ok = rays and e <= 1e-9 and g["light"].shape == o["light"].shape and g["overflow"] == 0
            if ok and len(o["light"]):
                ls = np.abs(o["light"][:, 3:]).max() + 1e-300
                ok = np.allclose(g["light"][:, 3:], o["light"][:, 3:], rtol=1e-9, atol=1e-12 * ls) and np.array_equal(g["light"][:, 2], o["light"][:, 2])
            worst = max(worst, e if rays else 1.0)
            if not ok:
                bad += 1
                if bad <= 5:
                    print("  DIFF", name, md, li, (px, py, s), "general", g["general"], "overflow", g["overflow"], "rays", (g["raysTraced"], g["shadowRaysTraced"]), (o["raysTraced"], o["shadowRaysTraced"]), "err %.2e" % e,
                          "light", g["light"].shape, o["light"].shape, flush=True)
        print("%s maxDepth %d lightImage %s: %d samples, %d general, %d differ, worst rel err %.2e" % (name, md, li, N, gen, bad, worst), flush=True)
    S.close(); O.close()
