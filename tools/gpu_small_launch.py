"""Is a small launch inefficient, or is the middle of the Cornell frame just dearer per ray?  Full frame at spp 8 (1/8 of the
work, spread evenly) against the eight 90-row strips at spp 64."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, parallel, scenes

W, H = 1280, 720
scene = gpt.Scene(scenes.cornell_box(W, H, "diffuse"), device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1)

def run(y0, y1, spp, S, seed=5489):
    film = gpt.Film(scene, y0, y1)
    film.set_slices(S)
    cfg = integ.config(spp)
    for rep in range(2):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, y0, W, y1)); film.sync()
    ms = film.render_ms(); st = film.stats(); film.close()
    return ms, st["raysTraced"] + st["shadowRaysTraced"]

for spp in (64, 32, 16, 8, 4):
    ms, rays = run(0, H, spp, 1)
    print("full frame spp %2d: %.1f ms, %.1f Mrays, %.0f Mray/s" % (spp, ms, rays / 1e6, rays / ms / 1e3), flush=True)
tot = 0.0
for (y0, y1) in parallel.row_strips(H, 8):
    ms, rays = run(y0, y1, 64, 0)
    tot += ms
    print("rows %3d-%3d spp 64 auto slices: %.1f ms, %.1f Mrays, %.0f Mray/s" % (y0, y1, ms, rays / 1e6, rays / ms / 1e3), flush=True)
print("sum of strips %.1f ms" % tot)
