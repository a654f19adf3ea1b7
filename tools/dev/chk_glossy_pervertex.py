"""Dev check: a scene WITH glossy vertices (the in-place path: k_render<STAGED> + k_continue) in HBM with vertex normals = face normals, through whatever build the library picks."""
import os, sys
sys.path.insert(0, '.')
os.environ["GDPT_SCENE_IN_HBM"] = "1"
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes
W, H, spp = 1280, 720, 16
def with_normals(sc):
    v = np.asarray(sc.verts, np.float64).reshape(-1, 3, 3)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]); n /= np.linalg.norm(n, axis=1, keepdims=True)
    sc.normals = np.concatenate([n, n, n], axis=1)
    for e in sc.emitters:
        if not isinstance(e[0], str):
            sc.normals[int(e[0]):int(e[0]) + int(e[1])] = 0.0
    return sc
for name, desc in (("glossy flat", scenes.cornell_box(W, H, "glossy")), ("glossy + vertex normals", with_normals(scenes.cornell_box(W, H, "glossy")))):
    scene = gpt.Scene(desc, device=0); integ = gpt.GradientPathIntegrator(maxDepth=12); cfg = integ.config(spp)
    film = gpt.Film(scene); best = 1e9
    for rep in range(3):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync(); best = min(best, film.render_ms())
    st = film.stats()
    print("%s: %.1f ms  %.0f Mray/s (%d rays)" % (name, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3, st["raysTraced"] + st["shadowRaysTraced"]), flush=True)
    film.close(); scene.close()
