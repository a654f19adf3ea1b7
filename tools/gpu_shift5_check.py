"""k_shift5 (one path per lane, GDPT_SHIFT5=1) against k_render<STAGED> (the default): films and ray counts on small scenes of every feature
set, then render times at config 2 / glossy / atrium sizes.  Needs a library built with GDPT_EXTRA_FLAGS=-DGDPT_WITH_SHIFT5 (the kernel is not in the product build: it is slower).
Usage: python tools/gpu_shift5_check.py [check|perf|both]"""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes

what = sys.argv[1] if len(sys.argv) > 1 else "both"


def render(desc, kw, spp, s5, reps=1, hbm=False):
    os.environ["GDPT_SHIFT5"] = "1" if s5 else "0"
    if hbm:
        os.environ["GDPT_SCENE_IN_HBM"] = "1"
    else:
        os.environ.pop("GDPT_SCENE_IN_HBM", None)
    S = gpt.Scene(desc, device=0)
    F = gpt.Film(S)
    integ = gpt.GradientPathIntegrator(**kw)
    cfg = integ.config(spp)
    best = 1e9
    for _ in range(reps):
        F.clear(); integ.renderBlock(S, F, cfg, (0, 0, desc.width, desc.height)); F.sync()
        best = min(best, F.render_ms())
    acc, st = F.accum(), F.stats()
    F.close(); S.close()
    return acc, st, best


if what in ("check", "both"):
    cases = [("diffuse", lambda: scenes.cornell_box(64, 48, "diffuse"), dict(maxDepth=-1)),
             ("diffuse-d2", lambda: scenes.cornell_box(64, 48, "diffuse"), dict(maxDepth=2)),
             ("glossy-strict", lambda: scenes.cornell_box(40, 30, "glossy"), dict(maxDepth=8, strictNormals=True)),
             ("nearspecular-strict", lambda: scenes.cornell_box(40, 30, "nearspecular"), dict(maxDepth=6, strictNormals=True)),
             ("glass", lambda: scenes.cornell_box(40, 30, "glass"), dict(maxDepth=8)),
             ("env-strict", lambda: scenes.cornell_box(48, 30, "glossy", environment=(0.7, 0.9, 1.2)), dict(maxDepth=5, strictNormals=True)),
             ("env-deep", lambda: scenes.cornell_box(48, 30, "diffuse", environment=(0.7, 0.9, 1.2)), dict(maxDepth=4)),
             ("bent-strict", lambda: scenes.cornell_box(40, 30, "bent"), dict(maxDepth=7, strictNormals=True))]
    bad = 0
    for name, mk, kw in cases:
        for hbm in (False, True):
            try:
                desc = mk()
            except Exception as e:
                print(name, "skipped:", e); break
            a0, s0, _ = render(desc, kw, 4, False, hbm=hbm)
            a1, s1, _ = render(desc, kw, 4, True, hbm=hbm)
            same = all(np.array_equal(x, y) for x, y in zip(a0, a1))
            close = all(np.allclose(x, y, rtol=1e-9, atol=1e-12) for x, y in zip(a0, a1))
            rays = (s0["raysTraced"], s0["shadowRaysTraced"]) == (s1["raysTraced"], s1["shadowRaysTraced"])
            print("%-22s %s: bit-identical %s, close %s, rays equal %s (%d/%d vs %d/%d) paths %s" % (name, "hbm" if hbm else "lds", same, close, rays,
                  s0["raysTraced"], s0["shadowRaysTraced"], s1["raysTraced"], s1["shadowRaysTraced"], s0.get("paths") == s1.get("paths")), flush=True)
            bad += not (close and rays)
    print("MISMATCHES:", bad)

if what in ("perf", "both"):
    W, H = 1280, 720
    for name, desc, spp, kw in (("cornell", scenes.cornell_box(W, H, "diffuse"), 32, dict(maxDepth=-1)), ("glossy", scenes.cornell_box(W, H, "glossy"), 16, dict(maxDepth=12)),
                                ("atrium", scenes.atrium(W, H), 8, dict(maxDepth=-1))):
        for s5 in (False, True):
            _, st, ms = render(desc, kw, spp, s5, reps=3)
            print("%s shift5=%d: %.1f ms  %.0f Mray/s" % (name, s5, ms, (st["raysTraced"] + st["shadowRaysTraced"]) / ms / 1e3), flush=True)
