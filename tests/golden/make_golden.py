#!/usr/bin/env python3
"""Writes the fixtures of this directory.  Run from the repo root:  python tests/golden/make_golden.py

WHAT THESE ARE.  The reference (mmanzi/gradientdomain-mitsuba) has no test vectors for the gpt / poisson_solver path and
cannot be built in this image ("Oracle pinning"), so nothing here is output of the reference.  The fixtures
are:
  * survey_b4_prefixes.json -- the four three-number prefixes SURVEY.md Appendix B.4 reports from the survey stage's
    (shimmed) build of the reference solver on the 64x48 synthetic input of SURVEY.md 8(d): corroboration, not a pin;
  * poisson_restatement.npz / gpt_restatement.json -- outputs of oracle/ itself on seeded inputs, frozen so that a change
    to the restatement (or to the HIP path, which the GPU tests compare with the same files) cannot pass unnoticed.
Inputs are regenerated from their seeds by the tests; only outputs are stored.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gradientdomain_mitsuba_amd import scenes      # noqa: E402
from oracle import gpt_oracle as go               # noqa: E402
from oracle import poisson_oracle as po           # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

POISSON_CASES = [("L2D", 40, 28), ("L1D", 40, 28), ("L2D", 33, 17), ("L2Q", 24, 16)]
GPT_CASES = [("diffuse", -1), ("glossy", 8), ("nearspecular", 8), ("twosided", -1), ("glass", 10), ("glossy+env", 8), ("bent", 8), ("random7", 6)]
ENV = (0.6, 0.8, 1.1)        # radiance of the `constant` environment emitter of the "+env" cases


def build_scene(case, W, H):
    variant, _, env = case.partition("+")
    if variant.startswith("random"):          # fuzzed materials (seed in the name): includes the Phong distribution
        return scenes.cornell_box(W, H, "random", seed=int(variant[6:]), environment=ENV if env else None)
    return scenes.cornell_box(W, H, variant, environment=ENV if env else None)
GPT_SIZE = (48, 36)
GPT_POINTS = 12


def gpt_points(variant):
    rng = np.random.default_rng(sum(map(ord, variant)))
    W, H = GPT_SIZE
    return [(int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 16))) for _ in range(GPT_POINTS)]


def main():
    json.dump({"source": "SURVEY.md Appendix B.4 (survey-stage build of the reference solver with a hand-written windows.h); 64x48 synthetic input, seed 12345",
               "L2D": ["0.501758039", "0.848977268", "0.851127267"], "L1D": ["0.498723149", "0.842597842", "0.85677588"]},
              open(os.path.join(HERE, "survey_b4_prefixes.json"), "w"), indent=1)

    out = {}
    for preset, w, h in POISSON_CASES:
        dx, dy, tp, direct = po.synth_inputs(w, h)
        out["%s_%dx%d" % (preset, w, h)] = po.solve(po.preset(preset), dx, dy, tp, direct, w, h)
    np.savez_compressed(os.path.join(HERE, "poisson_restatement.npz"), **out)

    rec = {}
    W, H = GPT_SIZE
    for variant, md in GPT_CASES:
        O = go.Scene(build_scene(variant, W, H))
        cfg = go.config(maxDepth=md, spp=16)
        pts = []
        for (px, py, s) in gpt_points(variant):
            e = O.evaluate_point(cfg, px, py, s)
            pts.append({"px": px, "py": py, "sample": s, "veryDirect": [float.hex(float(v)) for v in e["veryDirect"]],
                        "throughput": [float.hex(float(v)) for v in e["throughput"]],
                        "gradients": [float.hex(float(v)) for v in e["gradients"].ravel()],
                        "neighbours": [float.hex(float(v)) for v in e["neighbours"].ravel()]})
        rec[variant] = {"maxDepth": md, "points": pts}
        O.close()
    json.dump({"size": list(GPT_SIZE), "spp": 16, "seed": 5489, "cases": rec}, open(os.path.join(HERE, "gpt_restatement.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
