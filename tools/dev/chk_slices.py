"""Dev check: staged frame time with one sample slice per tile and with the default: python tools/dev/chk_slices.py W H spp [cornell]"""
import os, sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
desc = scenes.cornell_box(W, H, "diffuse") if len(sys.argv) > 4 else scenes.atrium(W, H)
scene = gpt.Scene(desc, device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1)
cfg = integ.config(spp)
for sl in (1, 0, 1, 0):
    film = gpt.Film(scene); film.set_slices(sl); best = 1e9
    for rep in range(2):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync(); best = min(best, film.render_ms())
    print("%dx%d spp %d slices %d: %.1f ms = %.2f ms per spp" % (W, H, spp, sl, best, best / spp), flush=True)
    film.close()
