"""Render time of the config-2 frame at 32 spp and of the atrium at 8 spp (quick A/B of builds)."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 1280, 720
for name, desc, spp in (("cornell", scenes.cornell_box(W, H, "diffuse"), 32), ("glossy", scenes.cornell_box(W, H, "glossy"), 16), ("atrium", scenes.atrium(W, H), 8)):
    scene = gpt.Scene(desc, device=0)
    film = gpt.Film(scene)
    integ = gpt.GradientPathIntegrator(maxDepth=-1 if name != "glossy" else 12)
    cfg = integ.config(spp)
    best = 1e9
    for rep in range(3):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
        best = min(best, film.render_ms())
    st = film.stats()
    print("%s: %.1f ms  %.0f Mray/s" % (name, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
    film.close(); scene.close()
