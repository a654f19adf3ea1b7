"""Films of a set of small scenes as hashes + counters: run once per library (GDPT_LIB=...) and diff the outputs -- the bit-identity gate of a kernel A/B
(`python tools/gpu_ab_films.py > a.txt; GDPT_LIB=lib/var/libgdpt_B.so python tools/gpu_ab_films.py > b.txt; diff a.txt b.txt`), then the three timing frames
of gpu_quick_perf.py unless NO_PERF=1."""
import hashlib
import os
import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes

W, H = 64, 48
cases = []
for variant in ("diffuse", "rough", "twosided", "glossy", "nearspecular", "glass", "mirrors", "smooth", "bent"):
    for maxDepth, strict in ((-1, False), (2, False), (3, True), (5, True)):
        cases.append((variant + "/d%d%s" % (maxDepth, "s" if strict else ""), scenes.cornell_box(W, H, variant), maxDepth, strict, 6))
cases.append(("diffuse+env/d-1", scenes.cornell_box(W, H, "diffuse", environment=(0.3, 0.4, 0.5)), -1, False, 6))
cases.append(("rough+env/d4s", scenes.cornell_box(W, H, "rough", environment=(0.3, 0.4, 0.5)), 4, True, 6))
cases.append(("diffuse+point/d-1", scenes.cornell_box(W, H, "diffuse", point_light=((278, 400, 279), (3e5, 3e5, 3e5))), -1, False, 6))
cases.append(("atrium/d-1", scenes.atrium(96, 54), -1, False, 4))
cases.append(("atrium/d3s", scenes.atrium(96, 54), 3, True, 4))
for name, desc, maxDepth, strict, spp in cases:
    S = gpt.Scene(desc)
    F = gpt.Film(S)
    integ = gpt.GradientPathIntegrator(maxDepth=maxDepth, strictNormals=strict)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, desc.width, desc.height))
    acc = F.accum(); st = F.stats()
    h = hashlib.sha256(np.ascontiguousarray(acc).tobytes()).hexdigest()[:16]
    print(name, h, st["raysTraced"], st["shadowRaysTraced"], st["paths"], st["pathLengthSum"], flush=True)
    F.close(); S.close()
if not os.environ.get("NO_PERF"):
    exec(open("tools/gpu_quick_perf.py").read())
