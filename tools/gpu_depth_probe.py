"""Mray/s of the render kernel against maxDepth (how much do lanes at different path depths cost each other?)."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes

W, H = 1280, 720
scene = gpt.Scene(scenes.cornell_box(W, H, "diffuse"), device=0)
film = gpt.Film(scene)
for md in (2, 3, 4, 6, -1):
    integ = gpt.GradientPathIntegrator(maxDepth=md)
    cfg = integ.config(32)
    for rep in range(2):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
    ms = film.render_ms(); st = film.stats()
    rays = st["raysTraced"] + st["shadowRaysTraced"]
    print("maxDepth %2d: %.1f ms  %.2f rays/sample  %.0f Mray/s  mean path length %.2f" % (md, ms, rays / (W * H * 32), rays / ms / 1e3, st["pathLengthSum"] / st["paths"]), flush=True)
