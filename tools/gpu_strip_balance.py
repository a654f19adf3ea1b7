"""Render time of each of N contiguous row strips of the config-2 frame (one GPU, one after the other): the strong-scaling
ceiling of the row-strip partition is mean/max of these."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, parallel, scenes

W, H, SPP = 1280, 720, 64
for name, desc in (("cornell", scenes.cornell_box(W, H, "diffuse")), ("atrium", scenes.atrium(W, H))):
    scene = gpt.Scene(desc, device=0)
    integ = gpt.GradientPathIntegrator(maxDepth=-1)
    cfg = integ.config(SPP if name == "cornell" else 16)
    for world, parts in ((1, 1), (8, 1), (8, 4)):
        strips = parallel.row_strips(H, world * parts)
        per_rank = [0.0] * world
        for i, (y0, y1) in enumerate(strips):
            film = gpt.Film(scene, y0, y1)
            for rep in range(2):
                film.clear(); integ.renderBlock(scene, film, cfg, (0, y0, W, y1)); film.sync()
            per_rank[i % world] += film.render_ms()
            film.close()
        print(name, "ranks", world, "strips/rank", parts, "ms per rank:", " ".join("%.1f" % v for v in per_rank),
              "| balance mean/max = %.3f" % (sum(per_rank) / world / max(per_rank)), flush=True)
    scene.close()
