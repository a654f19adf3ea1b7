"""Dev check (scratch build with GDPT_DEV_UNEVEN): equal chunks against full chunks + a shorter last one (a shorter drain of the pipeline)."""
import os, sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 1280, 720
for name, desc, spp in (("cornell", scenes.cornell_box(W, H, "diffuse"), 64), ("atrium", scenes.atrium(W, H), 32), ("cornell", scenes.cornell_box(W, H, "diffuse"), 48)):
    scene = gpt.Scene(desc, device=0)
    integ = gpt.GradientPathIntegrator(maxDepth=-1); cfg = integ.config(spp)
    for ue in (None, "1", "2", None, "1", "2"):
        os.environ.pop("GDPT_DEV_UNEVEN", None)
        if ue: os.environ["GDPT_DEV_UNEVEN"] = ue
        film = gpt.Film(scene); best = 1e9
        for rep in range(4):
            film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync(); best = min(best, film.render_ms())
        print("%s %d spp uneven=%s: %.2f ms" % (name, spp, ue, best), flush=True)
        film.close()
    scene.close()
