"""A/B of the sorted continuation list (gdpt_film::sortQueue, k_queue_keys + radix sort; VERDICT r4 #3d) on the atrium: frame time with GDPT_SORT_QUEUE=0 / 1
in one process, films bit for bit and ray counts identical.  Run on the GPU box: python tools/gpu_sort_queue_ab.py [W H spp]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import scenes, gpt
W, H, spp = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 8)
S = gpt.Scene(scenes.atrium(W, H))
integ = gpt.GradientPathIntegrator(maxDepth=-1)
res = {}
for rep in range(3):
    for mode in ("0", "1"):
        os.environ["GDPT_SORT_QUEUE"] = mode
        F = gpt.Film(S)
        integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H)); F.sync()
        st = F.stats(); ms = F.render_ms(); rays = st['raysTraced'] + st['shadowRaysTraced']
        acc = F.accum()
        print("sort=%s rep %d: %.2f ms, %.1f Mray/s" % (mode, rep, ms, rays / ms / 1e3), flush=True)
        if mode in res:
            assert res[mode][0] == (st['raysTraced'], st['shadowRaysTraced'])
        else:
            res[mode] = ((st['raysTraced'], st['shadowRaysTraced']), [a.copy() for a in acc])
        F.close()
assert res["0"][0] == res["1"][0], "ray counts differ"
for a, b in zip(res["0"][1], res["1"][1]):
    assert np.array_equal(a, b), "films differ"
print("films bit-identical, ray counts identical:", res["0"][0])
