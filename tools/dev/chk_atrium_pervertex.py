"""Dev check: what the per-vertex build of the HBM-scene kernels costs -- the atrium as it is (flat triangles: the <ENV = false, SMOOTH = false> build) and with vertex normals on
every triangle equal to its face normal (the same picture through the <true, true> build a real mesh with normals / textures / an environment runs)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes
W, H, spp = 1280, 720, int(sys.argv[1]) if len(sys.argv) > 1 else 8
flat = scenes.atrium(W, H)
smooth = scenes.atrium(W, H)
v = np.asarray(smooth.verts, np.float64).reshape(-1, 3, 3)
n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]); n /= np.linalg.norm(n, axis=1, keepdims=True)
smooth.normals = np.concatenate([n, n, n], axis=1)
for e in smooth.emitters:                     # (emitter meshes stay flat: all-zero rows)
    if not isinstance(e[0], str):
        smooth.normals[int(e[0]):int(e[0]) + int(e[1])] = 0.0
for name, desc in (("flat", flat), ("vertex normals", smooth)):
    scene = gpt.Scene(desc, device=0); integ = gpt.GradientPathIntegrator(maxDepth=-1); cfg = integ.config(spp)
    film = gpt.Film(scene); best = 1e9
    for rep in range(3):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync(); best = min(best, film.render_ms())
    st = film.stats()
    print("%s: %.1f ms  %.0f Mray/s (%d rays)" % (name, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3, st["raysTraced"] + st["shadowRaysTraced"]), flush=True)
    film.close(); scene.close()
