"""Investigation: the STAGED build of k_render against the general build (GDPT_DEV_GENERAL_KERNEL=1) under strictNormals."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, scenes
variant = sys.argv[1] if len(sys.argv) > 1 else "nearspecular"
W = H = 32
sc = scenes.cornell_box(W, H, variant)
S = G.Scene(sc)
for md in (2, 3, 4, 8):
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=True)
    res = {}
    for mode in ("staged", "general"):
        if mode == "general":
            os.environ["GDPT_DEV_GENERAL_KERNEL"] = "1"
        else:
            os.environ.pop("GDPT_DEV_GENERAL_KERNEL", None)
        F = G.Film(S); F.set_pipeline(2)
        integ.renderBlock(S, F, integ.config(1), (0, 0, W, H))
        res[mode] = F.accum(); F.close()
    d = np.abs(res["staged"] - res["general"])
    print("general checksum", repr(float(res["general"].sum())))
    bad = np.argwhere(d[1][..., :3].max(-1) > 1e-12)
    print("maxDepth", md, "buffers max diff", [float(d[b].max()) for b in range(5)], "pixels differing in -throughput:", len(bad), bad[:6].tolist(), flush=True)
