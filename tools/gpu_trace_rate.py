"""Pure traversal on incoherent rays: nodes / triangles per ray (gdpt_scene_trace_stats) of the device tree; run under
rocprofv3 --kernel-trace --stats for the k_intersect time (tools/prof_trace.sh)."""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes

for name, desc in (("cornell", scenes.cornell_box(256, 256, "diffuse")), ("atrium", scenes.atrium(256, 256))):
    scene = gpt.Scene(desc, device=0)
    v = np.asarray(desc.verts, np.float64).reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    rng = np.random.default_rng(1)
    n = 1 << 22
    o = lo + (hi - lo) * rng.random((n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    for rep in range(3):
        prim, t, p = scene.intersect(o, d)
    print(name, "hit fraction %.3f" % (prim >= 0).mean(), scene.trace_stats(o[:1 << 16], d[:1 << 16]), flush=True)
    scene.close()
