"""Continuation kernel (k_continue) A/B: film parity against the oracle and against the single-kernel path, then render times."""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes
from oracle import gpt_oracle as go

variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["diffuse", "glossy"]
for variant in variants:
    W, H, spp = 64, 48, 6
    sc = scenes.cornell_box(W, H, variant)
    S = gpt.Scene(sc)
    integ = gpt.GradientPathIntegrator(maxDepth=-1)
    res = {}
    for mode in ("off", "on"):
        if mode == "off":
            os.environ["GDPT_NO_CONTINUATION"] = "1"
        else:
            os.environ.pop("GDPT_NO_CONTINUATION", None)
        F = gpt.Film(S)
        integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H))
        res[mode] = (F.accum(), F.stats())
        F.close()
    oacc, orays = go.Scene(sc).render(go.config(maxDepth=-1, spp=spp))
    for mode in ("off", "on"):
        acc, st = res[mode]
        err = max(float(np.abs(acc[b] - oacc[b]).max() / (np.abs(oacc[b]).max() + 1e-300)) for b in range(5))
        print(variant, mode, "rays", (st["raysTraced"], st["shadowRaysTraced"]) == orays, st["paths"], st["pathLengthSum"], "max rel film diff %.2e" % err, flush=True)
    S.close()

W, H = 1280, 720
for name, desc, spp in (("cornell", scenes.cornell_box(W, H, "diffuse"), 32), ("glossy", scenes.cornell_box(W, H, "glossy"), 16), ("atrium", scenes.atrium(W, H), 8)):
    if len(sys.argv) > 2 and name not in sys.argv[2]:
        continue
    scene = gpt.Scene(desc, device=0)
    integ = gpt.GradientPathIntegrator(maxDepth=-1 if name != "glossy" else 12)
    cfg = integ.config(spp)
    for mode in ("off", "on"):
        if mode == "off":
            os.environ["GDPT_NO_CONTINUATION"] = "1"
        else:
            os.environ.pop("GDPT_NO_CONTINUATION", None)
        film = gpt.Film(scene)
        best = 1e9
        for rep in range(3):
            film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
            best = min(best, film.render_ms())
        st = film.stats()
        print("%s %s: %.1f ms  %.0f Mray/s" % (name, mode, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
        film.close()
    scene.close()
