"""Wavefront pipeline (gdpt_film_set_pipeline 3; csrc/gpt_wavefront.hip.h) against the staged pipeline (2): films and statistics must be
bit-identical (the shading passes replay the one bounce() the megakernels run); then render times of the three perf scenes.
  python tools/gpu_wf_check.py [check|perf|both] [iters,iters,...]"""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt, scenes

what = sys.argv[1] if len(sys.argv) > 1 else "both"
iters = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 3, 6, 40]
PIPE = int(os.environ.get("WF_PIPE", "3"))


def render(S, integ, cfg, W, H, stages, it=None, hbm=False):
    if it is not None:
        os.environ["GDPT_WF_ITERS"] = str(it)
    F = gpt.Film(S); F.set_pipeline(stages)
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    r = (F.accum(), F.stats(), F.invalid_puts())
    F.close()
    return r


if what in ("check", "both"):
    env = lambda: scenes.cornell_box(48, 40, "glossy", environment=(0.6, 0.7, 0.9))
    cases = [("diffuse", lambda: scenes.cornell_box(48, 40, "diffuse"), dict(maxDepth=-1)),
             ("glossy", lambda: scenes.cornell_box(48, 40, "glossy"), dict(maxDepth=12)),
             ("glass", lambda: scenes.cornell_box(48, 40, "glass"), dict(maxDepth=14)),
             ("nearspecular-strict", lambda: scenes.cornell_box(48, 40, "nearspecular"), dict(maxDepth=10, strictNormals=True)),
             ("bent-normals-strict", lambda: scenes.cornell_box(48, 40, "bent"), dict(maxDepth=9, strictNormals=True)),
             ("environment", env, dict(maxDepth=8)),
             ("textured", lambda: scenes.textured_cornell_box(48, 36), dict(maxDepth=7)),
             ("atrium", lambda: scenes.atrium(64, 36, columns=8, segments=12), dict(maxDepth=-1))]
    bad = 0
    for hbm in (False, True):
        if hbm:
            os.environ["GDPT_SCENE_IN_HBM"] = "1"
        for name, builder, kw in cases:
            sc = builder()
            W, H, spp = sc.width, sc.height, 5
            S = gpt.Scene(sc)
            integ = gpt.GradientPathIntegrator(**kw)
            cfg = integ.config(spp)
            ref = render(S, integ, cfg, W, H, 2)
            for it in iters:
                got = render(S, integ, cfg, W, H, PIPE, it)
                same = all(np.array_equal(got[0][b], ref[0][b]) for b in range(5))
                d = max(float(np.abs(got[0][b] - ref[0][b]).max() / (np.abs(ref[0][b]).max() + 1e-300)) for b in range(5))
                ok = same and got[1] == ref[1] and got[2] == ref[2]
                bad += not ok
                print("%s %-20s iters %2d: films %s (max rel %.1e) stats %s" % ("HBM" if hbm else "LDS", name, it, "identical" if same else "DIFFER", d, "equal" if got[1] == ref[1] else "DIFFER %r vs %r" % (got[1], ref[1])), flush=True)
            S.close()
    os.environ.pop("GDPT_SCENE_IN_HBM", None)
    print("check:", "OK" if bad == 0 else "%d FAILED" % bad, flush=True)

if what in ("perf", "both"):
    W, H = 1280, 720
    for name, desc, spp in (("cornell", scenes.cornell_box(W, H, "diffuse"), 32), ("glossy", scenes.cornell_box(W, H, "glossy"), 16), ("atrium", scenes.atrium(W, H), 8)):
        if len(sys.argv) > 3 and name not in sys.argv[3]:
            continue
        scene = gpt.Scene(desc, device=0)
        integ = gpt.GradientPathIntegrator(maxDepth=-1 if name != "glossy" else 12)
        cfg = integ.config(spp)
        for stages, it in [(2, None)] + [(PIPE, i) for i in iters]:
            if it is not None:
                os.environ["GDPT_WF_ITERS"] = str(it)
            film = gpt.Film(scene); film.set_pipeline(stages)
            best = 1e9
            for rep in range(3):
                film.clear(); integ.renderBlock(scene, film, cfg, (0, 0, W, H)); film.sync()
                best = min(best, film.render_ms())
            st = film.stats()
            print("%s pipeline %d iters %s: %.1f ms  %.0f Mray/s" % (name, stages, it, best, (st["raysTraced"] + st["shadowRaysTraced"]) / best / 1e3), flush=True)
            film.close()
        scene.close()
