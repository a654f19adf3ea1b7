"""Dev check: config 2's frame with the per-sample sums of the first stage in LDS (default) and in registers (gdpt_film_set_occupancy(-2))."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H, spp = 1280, 720, 64
scene = gpt.Scene(scenes.cornell_box(W, H, "diffuse"), device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1); cfg = integ.config(spp)
for occ in (2, -2, 2, -2):
    film = gpt.Film(scene); film.set_occupancy(occ); best = 1e9
    for rep in range(3):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync(); best = min(best, film.render_ms())
    st = film.stats()
    print("occupancy %d: %.1f ms  %.0f Mray/s" % (occ, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
    film.close()
