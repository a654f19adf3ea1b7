"""gdpt_render_serial against the oracle's render_serial on the scenes of tools/gpu_fuzz_campaign.py: python tools/gpu_serial_fuzz.py [first [count]].
Every draw of every sample of a film comes from ONE stream: a film that matches says no draw of the device sampler is out of place anywhere in the film."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, scenes
from oracle import gpt_oracle as go

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time(); worst = 0.0; knife = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(17, 44)), int(rng.integers(9, 34))
    kind = "random"
    kw = dict(seed=seed, environment=(0.5, 0.7, 0.9) if seed % 3 == 0 else None)
    if seed % 5 == 1:
        kind = "smooth" if seed % 2 else "bent"; kw = dict(environment=kw["environment"])
    if seed % 5 == 2:
        kw["point_light"] = ((float(rng.uniform(100, 450)), float(rng.uniform(200, 500)), float(rng.uniform(100, 450))), (4e4, 3e4, 2e4), bool(seed % 2))
    sc = scenes.atrium(W, H, columns=int(rng.integers(4, 12)), segments=int(rng.integers(6, 16))) if seed % 7 == 0 else scenes.cornell_box(W, H, kind, **kw)
    if seed % 9 == 4:
        sc.thinlens = (float(rng.uniform(2.0, 60.0)), float(rng.uniform(300.0, 1500.0)))
    if seed % 8 == 5:
        sc.shutter = (0.0, float(rng.uniform(0.01, 1.0)))
    md = int(rng.choice([-1, 2, 3, 5, 9])); rr = int(rng.choice([1, 3, 5])); strict = bool(rng.random() < 0.35); thr = float(rng.choice([0.001, 0.02, 0.0]))
    spp = int(rng.integers(1, 4)); bs = (8, 16, 32)[seed % 3]
    S, O = G.Scene(sc), go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, rrDepth=rr, strictNormals=strict, shiftThreshold=thr)
    F = G.Film(S)
    integ.renderSerial(S, F, integ.config(spp), blockSize=bs, parentSeed=5489 + seed)
    acc, st = F.accum(), F.stats(); F.close()
    oacc, orays = O.render_serial(go.config(maxDepth=md, rrDepth=rr, strictNormals=strict, spp=spp, shiftThreshold=thr), block_size=bs, parent_seed=5489 + seed)
    d = max(np.abs(acc[b] - oacc[b]).max() / (np.abs(oacc[b]).max() + 1e-300) for b in range(5))
    if (st["raysTraced"], st["shadowRaysTraced"]) != orays or d > 1e-9:
        # (a ray count that differs shifts the whole stream behind it: a knife-edge hit shows as a different film, not as a small difference)
        print("DIFF seed %d: rays %r vs %r, worst buffer difference %.3e (%dx%d, %d spp, block %d)" % (seed, (st["raysTraced"], st["shadowRaysTraced"]), orays, d, W, H, spp, bs), flush=True)
        knife += 1
    else:
        worst = max(worst, d)
    S.close(); O.close()
print("seeds %d..%d: %d films equal (worst relative difference %.2e), %d differ; %.0f s" % (first, first + count - 1, count - knife, worst, knife, time.time() - t0))
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import fuzz_summary  # noqa: E402
fuzz_summary.emit("gpu_serial_fuzz", first, count, time.time() - t0, films_equal=count - knife, worst_rel_diff=worst, films_with_knife_edge_difference=knife)
