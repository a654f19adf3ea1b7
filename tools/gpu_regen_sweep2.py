"""Render time against the regeneration threshold of the staged k_render (idle lanes before a wave starts new samples), final pipeline."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 1280, 720
for name, desc, spp in (("cornell", scenes.cornell_box(W, H, "diffuse"), 32), ("atrium", scenes.atrium(W, H), 8)):
    scene = gpt.Scene(desc, device=0)
    integ = gpt.GradientPathIntegrator(maxDepth=-1)
    cfg = integ.config(spp)
    for regen in (1, 8, 16, 32, 48, 56, 64):
        film = gpt.Film(scene); film.set_regeneration(regen)
        best = 1e9
        for rep in range(3):
            film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
            best = min(best, film.render_ms())
        st = film.stats()
        print("%s regen %2d: %.1f ms  %.0f Mray/s" % (name, regen, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
        film.close()
    scene.close()
