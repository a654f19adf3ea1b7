"""Dev check: the atrium at 3840x2160 (config 4's size), a few samples per pixel, with and without the deferred continuation."""
import os, sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 6
scene = gpt.Scene(scenes.atrium(W, H), device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1); cfg = integ.config(spp)
for envs in ({}, {"GDPT_NO_DEFERRED": "1"}, {"GDPT_NO_PIPE": "1"}, {"GDPT_QUEUE_MB": "131072"}):
    for k in ("GDPT_NO_DEFERRED", "GDPT_NO_PIPE", "GDPT_QUEUE_MB"): os.environ.pop(k, None)
    os.environ.update(envs)
    film = gpt.Film(scene); best = 1e9
    for rep in range(2):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync(); best = min(best, film.render_ms())
    st = film.stats()
    print(envs, "%.1f ms  %.0f Mray/s" % (best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
    film.close()
