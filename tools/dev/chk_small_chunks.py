"""Dev check: small launches (config 1's frame; one of eight strips of config 2's) against the chunk length forced through GDPT_QUEUE_MB: more, shorter chunks give the
pipelined chunks something to overlap."""
import os, sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
spp = 64
integ = gpt.GradientPathIntegrator(maxDepth=-1)
cfg = integ.config(spp)
for name, desc, rect, per_spp_mb in (("config 1 frame 512x512", scenes.cornell_box(512, 512, "diffuse"), (0, 0, 512, 512), 2 * 262144 * 1800 / 2**20),
                                     ("config 2 strip 1280x90", scenes.cornell_box(1280, 720, "diffuse"), (0, 270, 1280, 360), 2 * 115200 * 1800 / 2**20),
                                     ("atrium strip 1920x135", scenes.atrium(1920, 1080), (0, 540, 1920, 675), 2 * 261120 * 1800 / 2**20)):
    scene = gpt.Scene(desc, device=0)
    for ch in (64, 32, 16, 8, 4, 2):
        os.environ["GDPT_QUEUE_MB"] = str(int(per_spp_mb * (ch + 0.5)))
        film = gpt.Film(scene); best = 1e9
        for rep in range(3):
            film.clear(); integ.renderBlock(scene, film, cfg, rect); film.sync(); best = min(best, film.render_ms())
        st = film.stats()
        print("%s, chunks of <= %d spp: %.2f ms  %.0f Mray/s" % (name, ch, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
        film.close()
    scene.close()
