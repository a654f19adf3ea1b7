"""Render rate against the BVH leaf size (GDPT_BVH_LEAF, read at scene creation)."""
import os, sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 1280, 720
for name, desc, spp in (("cornell", scenes.cornell_box(W, H, "diffuse"), 32), ("atrium", scenes.atrium(W, H), 8)):
    for leaf in (1, 2, 4, 8):
        os.environ["GDPT_BVH_LEAF"] = str(leaf)
        scene = gpt.Scene(desc, device=0)
        film = gpt.Film(scene)
        integ = gpt.GradientPathIntegrator(maxDepth=-1)
        cfg = integ.config(spp)
        for rep in range(2):
            film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
        ms = film.render_ms(); st = film.stats()
        print("%s leaf<=%d: %.1f ms  %.0f Mray/s" % (name, leaf, ms, (st["raysTraced"] + st["shadowRaysTraced"]) / ms / 1e3), flush=True)
        film.close(); scene.close()
