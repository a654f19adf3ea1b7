"""Render time against the regeneration threshold and the occupancy build, config-2 Cornell frame and the atrium."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes

W, H = 1280, 720
for name, desc, spp, occs in (("cornell", scenes.cornell_box(W, H, "diffuse"), 32, (2, 3)), ("glossy", scenes.cornell_box(W, H, "glossy"), 16, (2, 3)),
                              ("atrium", scenes.atrium(W, H), 16, (3, 4))):
    scene = gpt.Scene(desc, device=0)
    integ = gpt.GradientPathIntegrator(maxDepth=-1 if name != "glossy" else 12)
    cfg = integ.config(spp)
    film = gpt.Film(scene)
    for occ in occs:
        for regen in (24, 40, 48, 56, 60, 64):
            film.set_occupancy(occ); film.set_regeneration(regen)
            for rep in range(2):
                film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
            ms = film.render_ms(); st = film.stats()
            print("%s occ %d regen %2d: %.1f ms  %.0f Mray/s" % (name, occ, regen, ms, (st["raysTraced"] + st["shadowRaysTraced"]) / ms / 1e3), flush=True)
    film.close(); scene.close()
