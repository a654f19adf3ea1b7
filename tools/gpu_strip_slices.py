"""One 90-row strip of the config-2 frame (the 8-GPU share): render time against the number of sample slices."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes

W, H = 1280, 720
scene = gpt.Scene(scenes.cornell_box(W, H, "diffuse"), device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1)
cfg = integ.config(64)
for (y0, y1) in ((0, 90), (360, 450), (630, 720)):
    film = gpt.Film(scene, y0, y1)
    for S in (1, 2, 4, 7, 8, 12, 16, 32, 64):
        film.set_slices(S)
        best = 1e9
        for rep in range(3):
            film.clear(); integ.renderBlock(scene, film, cfg, (0, y0, W, y1)); film.sync()
            best = min(best, film.render_ms())
        print("rows %3d-%3d slices %2d: %.1f ms" % (y0, y1, S, best), flush=True)
    film.close()
