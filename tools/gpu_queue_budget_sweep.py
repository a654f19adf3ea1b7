"""Frame time against the sample queue's memory budget (GDPT_QUEUE_MB): the budget sets the chunk length, the chunk length how much of a frame the pipelined chunks overlap.
python tools/gpu_queue_budget_sweep.py [budgets in MiB ...]   (config 2's frame at 64 spp, the atrium 1280x720 at 32 spp)"""
import os
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 1280, 720
budgets = [int(a) for a in sys.argv[1:]] or [12288, 24576, 49152, 73728, 98304]
for name, desc, spp in (("cornell", scenes.cornell_box(W, H, "diffuse"), 64), ("atrium", scenes.atrium(W, H), 32)):
    scene = gpt.Scene(desc, device=0)
    integ = gpt.GradientPathIntegrator(maxDepth=-1)
    cfg = integ.config(spp)
    for mb in budgets:
        os.environ["GDPT_QUEUE_MB"] = str(mb)
        film = gpt.Film(scene)
        best = 1e9
        for rep in range(3):
            film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
            best = min(best, film.render_ms())
        st = film.stats()
        print("%s %d spp, budget %d MiB: %.1f ms  %.0f Mray/s" % (name, spp, mb, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
        film.close()
    scene.close()
