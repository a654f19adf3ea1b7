"""The frozen fixtures of tests/golden/ (see make_golden.py there for what they are and are not) against the restatement
(CPU) and against the HIP path through the C-ABI (GPU).  Inputs are regenerated from their seeds; only outputs are stored."""
import importlib.util
import json
import os

import numpy as np
import pytest

from gradientdomain_mitsuba_amd import scenes
from oracle import gpt_oracle as go
from oracle import poisson_oracle as po

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _build_scene(case, W, H):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    return mg.build_scene(case, W, H)


def _gpt_cases():
    d = json.load(open(os.path.join(G, "gpt_restatement.json")))
    return d, [(v, c["maxDepth"], c["points"]) for v, c in d["cases"].items()]


def _vec(p, key):
    return np.array([float.fromhex(x) for x in p[key]])


def test_survey_prefixes_fixture_matches_the_restatement():
    d = json.load(open(os.path.join(G, "survey_b4_prefixes.json")))
    dx, dy, tp, direct = po.synth_inputs(64, 48)
    for preset in ("L2D", "L1D"):
        rec = po.solve(po.preset(preset), dx, dy, tp, direct, 64, 48)
        assert ["%.9g" % v for v in rec[:3]] == d[preset]


def test_poisson_restatement_is_frozen():
    z = np.load(os.path.join(G, "poisson_restatement.npz"))
    assert len(z.files) == 4
    for key in z.files:
        preset, size = key.split("_")
        w, h = map(int, size.split("x"))
        dx, dy, tp, direct = po.synth_inputs(w, h)
        assert np.array_equal(po.solve(po.preset(preset), dx, dy, tp, direct, w, h), z[key]), key     # same compiler flags -> same bits


def test_gpt_restatement_is_frozen():
    d, cases = _gpt_cases()
    W, H = d["size"]
    for variant, md, pts in cases:
        O = go.Scene(_build_scene(variant, W, H))
        cfg = go.config(maxDepth=md, spp=d["spp"], seed=d["seed"])
        for p in pts:
            e = O.evaluate_point(cfg, p["px"], p["py"], p["sample"])
            for key, got in (("veryDirect", e["veryDirect"]), ("throughput", e["throughput"]), ("gradients", e["gradients"].ravel()), ("neighbours", e["neighbours"].ravel())):
                # libm calls (sin/cos/sqrt/pow) may differ in the last bit between hosts; everything else is plain fp64
                assert np.allclose(got, _vec(p, key), rtol=1e-12, atol=1e-300), (variant, p["px"], p["py"], key)
        O.close()


@pytest.mark.gpu
def test_hip_poisson_against_golden(gpu_required):
    import gradientdomain_mitsuba_amd.poisson as P
    z = np.load(os.path.join(G, "poisson_restatement.npz"))
    tol = {"L2D": 5e-5, "L1D": 5e-4, "L2Q": 2e-4}
    for key in z.files:
        preset, size = key.split("_")
        w, h = map(int, size.split("x"))
        dx, dy, tp, direct = po.synth_inputs(w, h)
        for fusion in (0, 1, 2):
            s = P.Solver(P.Params(preset, 0.2)); s.setFusion(fusion)
            s.importImagesMTS(dx, dy, tp, direct, w, h); s.setupBackend(); s.solveIndirect()
            rec = s.exportImagesMTS(); s.close()
            assert np.abs(rec - z[key]).max() <= tol[preset], (key, fusion)


@pytest.mark.gpu
def test_hip_gpt_against_golden(gpu_required):
    from gradientdomain_mitsuba_amd import gpt
    d, cases = _gpt_cases()
    W, H = d["size"]
    for variant, md, pts in cases:
        S = gpt.Scene(_build_scene(variant, W, H))
        cfg = gpt.GradientPathIntegrator(maxDepth=md).config(d["spp"])
        for p in pts:
            e = S.evaluate_point(cfg, p["px"], p["py"], p["sample"])
            for key, got in (("veryDirect", e["veryDirect"]), ("throughput", e["throughput"]), ("gradients", e["gradients"].ravel()), ("neighbours", e["neighbours"].ravel())):
                assert np.allclose(got, _vec(p, key), rtol=1e-10, atol=1e-14), (variant, p["px"], p["py"], key)
        S.close()
