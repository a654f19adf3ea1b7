"""Dev check: the atrium at 3840x2160 x 24 spp against the queue budget (chunk length), slices 1 and default."""
import os, sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 3840, 2160
scene = gpt.Scene(scenes.atrium(W, H), device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1)
spp = 24
cfg = integ.config(spp)
for mb in (49152, 65536, 98304, 117000, 131072, 180000):
    for sl in (1, 0):
        os.environ["GDPT_QUEUE_MB"] = str(mb)
        film = gpt.Film(scene); film.set_slices(sl); best = 1e9
        for rep in range(2):
            film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync(); best = min(best, film.render_ms())
        print("budget %d MiB slices %d: %.1f ms = %.2f ms per spp" % (mb, sl, best, best / spp), flush=True)
        film.close()
