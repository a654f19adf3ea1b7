"""Dev check: the atrium at 3840x2160, frame time per sample per pixel against the number of samples (= chunks of one sample each at the default budget)."""
import os, sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 3840, 2160
scene = gpt.Scene(scenes.atrium(W, H), device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1)
for envs in ({}, {"GDPT_NO_DEFERRED": "1"}):
    for k in ("GDPT_NO_DEFERRED",): os.environ.pop(k, None)
    os.environ.update(envs)
    for spp, sl in ((6, 0), (12, 0), (24, 0), (31, 0), (32, 0), (32, 1), (48, 0), (48, 1)):
        cfg = integ.config(spp)
        film = gpt.Film(scene); film.set_slices(sl); best = 1e9
        for rep in range(2):
            film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync(); best = min(best, film.render_ms())
        print(envs, "spp %d slices %d: %.1f ms = %.2f ms per spp" % (spp, sl, best, best / spp), flush=True)
        film.close()
