"""Render time of the config-2 frame (32 spp) or the atrium (8 spp) against k_continue's refill threshold and the first stage's sample slices:
python tools/gpu_cont_sweep.py {cornell|atrium} REFILL[:SLICES] ...   (GDPT_CONT_WPS=2 with a -DGDPT_DEV_CONT2 build: the 2-wave k_continue of an HBM scene)"""
import sys, os
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 1280, 720
which = sys.argv[1]
desc, spp = (scenes.cornell_box(W, H, "diffuse"), 32) if which == "cornell" else (scenes.atrium(W, H), 8)
scene = gpt.Scene(desc, device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1)
cfg = integ.config(spp)
for a in sys.argv[2:]:
    refill, slices = (int(v) for v in (a.split(":") + ["0"])[:2])
    film = gpt.Film(scene); film.set_pipeline(2, refill); film.set_slices(slices)
    best = 1e9
    for rep in range(3):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
        best = min(best, film.render_ms())
    st = film.stats()
    print("%s cont_wps=%s refill %2d slices %d: %.1f ms  %.0f Mray/s" % (which, os.environ.get("GDPT_CONT_WPS", "-"), refill, slices, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
    film.close()
