import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 1280, 720
for name in ("atrium", "cornell"):
    desc = scenes.atrium(W, H) if name == "atrium" else scenes.cornell_box(W, H, "diffuse")
    scene = gpt.Scene(desc, device=0)
    for md in (2, -1):
        integ = gpt.GradientPathIntegrator(maxDepth=md)
        film = gpt.Film(scene)
        integ.renderBlock(scene, film, integ.config(4), (0, 0, W, H)); film.sync()
        st = film.stats(); acc = film.accum()
        print(name, "maxDepth", md, st, "avg len %.3f" % (st["pathLengthSum"] / st["paths"]), "rays/sample %.2f" % ((st["raysTraced"] + st["shadowRaysTraced"]) / st["paths"]),
              "pixels with zero throughput+direct: %.3f" % float(((np.abs(acc[1][..., :3]).sum(-1) + np.abs(acc[4][..., :3]).sum(-1)) == 0).mean()), flush=True)
        film.close()
    scene.close()
