"""Dev check: what the per-vertex build of the LDS-scene kernels costs -- config 2's Cornell box as it is and with vertex normals equal to the face normals (same picture, <true, true> build)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes
W, H, spp = 1280, 720, 32
flat = scenes.cornell_box(W, H, "diffuse")
smooth = scenes.cornell_box(W, H, "diffuse")
v = np.asarray(smooth.verts, np.float64).reshape(-1, 3, 3)
n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]); n /= np.linalg.norm(n, axis=1, keepdims=True)
smooth.normals = np.concatenate([n, n, n], axis=1)
for e in smooth.emitters:
    if not isinstance(e[0], str):
        smooth.normals[int(e[0]):int(e[0]) + int(e[1])] = 0.0
envd = scenes.cornell_box(W, H, "diffuse", environment=(0.3, 0.4, 0.5))
for name, desc in (("flat", flat), ("vertex normals", smooth), ("flat + environment", envd)):
    scene = gpt.Scene(desc, device=0); integ = gpt.GradientPathIntegrator(maxDepth=-1); cfg = integ.config(spp)
    film = gpt.Film(scene); best = 1e9
    for rep in range(3):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync(); best = min(best, film.render_ms())
    st = film.stats()
    print("%s: %.1f ms  %.0f Mray/s (%d rays)" % (name, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3, st["raysTraced"] + st["shadowRaysTraced"]), flush=True)
    film.close(); scene.close()
