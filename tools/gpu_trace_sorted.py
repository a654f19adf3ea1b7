"""What ray ORDER is worth to the traversal-only kernel: k_intersect on surface-born rays of the atrium (random surface point, random
direction in the hemisphere of its normal: what a path's bounces produce), in random order and sorted by a Morton key of the origin
(+ direction octant).  Run under tools/kt_list.sh for the per-dispatch times."""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes

desc = scenes.atrium(256, 256)
scene = gpt.Scene(desc, device=0)
v = np.asarray(desc.verts, np.float64).reshape(-1, 3, 3)
rng = np.random.default_rng(1)
n = 1 << 22
e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
nrm = np.cross(e1, e2); area = 0.5 * np.linalg.norm(nrm, axis=1); nrm /= (2 * area[:, None] + 1e-300)
tri = rng.choice(len(v), size=n, p=area / area.sum())
a, b = rng.random(n), rng.random(n)
f = a + b > 1; a[f], b[f] = 1 - a[f], 1 - b[f]
o = v[tri, 0] + a[:, None] * e1[tri] + b[:, None] * e2[tri]
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
flip = (d * nrm[tri]).sum(1) < 0
# (the builder orients normals towards the room; either way: leave the surface on the side the direction points to)
o = o + 1e-6 * d
lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)


def morton(bits):
    q = np.minimum(((o - lo) / (hi - lo) * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    key = np.zeros(n, np.uint64)
    for bit in range(bits):
        for ax in range(3):
            key |= ((q[:, ax] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + ax)
    return key


octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | ((d[:, 2] < 0).astype(np.uint64) << np.uint64(2)))
orders = [("random", np.arange(n)),
          ("morton5", np.argsort(morton(5), kind="stable")),
          ("morton5+octant", np.argsort((morton(5) << np.uint64(3)) | octant, kind="stable")),
          ("octant+morton5", np.argsort((octant << np.uint64(15)) | morton(5), kind="stable")),
          ("morton8+octant", np.argsort((morton(8) << np.uint64(3)) | octant, kind="stable")),
          ("octant+morton8", np.argsort((octant << np.uint64(24)) | morton(8), kind="stable"))]
for name, idx in orders:
    oo, dd = np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx])
    for rep in range(2):
        prim, t, p = scene.intersect(oo, dd)
    print(name, "hit fraction %.3f" % (prim >= 0).mean(), scene.trace_stats(oo[:1 << 16], dd[:1 << 16]), flush=True)
scene.close()
