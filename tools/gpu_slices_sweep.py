"""Render time vs sample slices, full config-2 frame and one 90-row strip of it (the 8-GPU share)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes

W, H = 1280, 720
for name, desc, spp in (("cornell", scenes.cornell_box(W, H, "diffuse"), 64), ("atrium", scenes.atrium(W, H), 16)):
    scene = gpt.Scene(desc, device=0)
    integ = gpt.GradientPathIntegrator(maxDepth=-1)
    cfg = integ.config(spp)
    for (y0, y1) in ((0, H), (360, 450)):
        film = gpt.Film(scene, y0, y1)
        ref = None
        for S in (1, 2, 4, 8, 16, 0):
            if S > spp:
                continue
            film.set_slices(S)
            for rep in range(2):
                film.clear(); integ.renderBlock(scene, film, cfg, (0, y0, W, y1)); film.sync()
            ms = film.render_ms()
            acc = film.accum()
            if ref is None:
                ref = acc
            err = np.abs(acc - ref).max() / np.abs(ref).max()
            st = film.stats()
            print("%s rows %d-%d slices %2d: %.1f ms  %.0f Mray/s  max rel diff vs slices=1 %.1e" % (
                name, y0, y1, S, ms, (st["raysTraced"] + st["shadowRaysTraced"]) / ms / 1e3, err), flush=True)
        film.close()
    scene.close()
