"""The first-bounce stage at ONE wave per SIMD (512 registers, gdpt_film_set_occupancy(1)) against the default two (256 + scratch): config-2 frame at
32 spp -- render time, and the film bit for bit (VERDICT r4 #4b).   gpurun -- 'timeout 200 python tools/gpu_wps1_check.py'"""
import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes
W, H, spp = 1280, 720, 32
scene = gpt.Scene(scenes.cornell_box(W, H, "diffuse"), device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1)
cfg = integ.config(spp)
out = {}
for occ in (2, 1, 2, 1):
    film = gpt.Film(scene); film.set_occupancy(occ)
    best = 1e9
    for rep in range(3):
        film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
        best = min(best, film.render_ms())
    st = film.stats()
    acc = film.accum()
    print("first-bounce stage at %d wave(s)/SIMD: %.2f ms  %.0f Mray/s" % (occ, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
    if occ in out:
        assert np.array_equal(out[occ][0], acc) and out[occ][1] == st
    out[occ] = (acc, st)
    film.close()
print("films bit-identical:", bool(np.array_equal(out[1][0], out[2][0])), " statistics identical:", out[1][1] == out[2][1])
