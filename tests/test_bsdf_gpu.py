"""The device's BSDF models checked on their own (gdpt_bsdf_probe), independently of the render kernels:
  * the reference's chi-square methodology (src/tests/test_chisquare.cpp, include/mitsuba/core/chisquare.h) applied to the HIP sample()/pdf()
    pair -- a statement about the device code alone: whatever sample() draws is distributed as pdf() says, whether or not the restatement
    it was written from is right;
  * sample weight == eval * cos / pdf (the identity every BSDF::sample of the reference promises, bsdf.h:372-396), again device-only;
  * direction by direction parity with the oracle (the same two entry points on the CPU restatement)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import importlib

pytestmark = pytest.mark.gpu

scenes = importlib.import_module("gradientdomain-mitsuba_amd.scenes")


def _gpt():
    return importlib.import_module("gradientdomain-mitsuba_amd.gpt")


MATS = [scenes.diffuse((0.7, 0.6, 0.5)), scenes.roughconductor(0.3, **scenes.CU),
        scenes.roughconductor(0.15, **scenes.AL, distribution=scenes.DISTR_GGX),
        scenes.roughconductor(0.25, **scenes.CU, alphaV=0.1), scenes.roughconductor(0.2, **scenes.AL, sampleVisible=False),
        scenes.roughconductor(0.2, **scenes.AL, distribution=scenes.DISTR_PHONG),
        dict(scenes.roughconductor(0.3, **scenes.CU), twoSided=1)]


def unit(v):
    v = np.asarray(v, float)
    return v / np.linalg.norm(v)


@pytest.mark.parametrize("mat", MATS)
@pytest.mark.parametrize("wi", [(0.35, 0.2, 0.9), (0.8, -0.3, 0.25)])
def test_device_bsdf_samples_follow_the_device_pdf_chi_square(mat, wi):
    from scipy.stats import chi2
    gpt = _gpt()
    rng = np.random.default_rng(23)
    wi = unit(wi)
    NT, NP, N, K = 10, 20, 200000, 32
    (wo, weight, pdf, _typ), _ = gpt.bsdf_probe(mat, wi, samples=rng.random((N, 2)))
    ok = (pdf > 0) & (wo[:, 2] > 0)
    lost = int((~ok).sum())
    it = np.minimum(NT - 1, (wo[ok, 2] * NT).astype(int))
    ip = np.minimum(NP - 1, ((np.arctan2(wo[ok, 1], wo[ok, 0]) % (2 * np.pi)) / (2 * np.pi) * NP).astype(int))
    obs = np.zeros((NT, NP))
    np.add.at(obs, (it, ip), 1)
    # expected counts: the DEVICE pdf integrated over each cell, K x K midpoints
    z = (np.arange(NT)[:, None] + (np.arange(K)[None, :] + 0.5) / K).ravel() / NT
    phi = (np.arange(NP)[:, None] + (np.arange(K)[None, :] + 0.5) / K).ravel() / NP * 2 * np.pi
    Z, PHI = np.meshgrid(z, phi, indexing="ij")
    R = np.sqrt(np.maximum(0.0, 1 - Z * Z))
    dirs = np.stack([R * np.cos(PHI), R * np.sin(PHI), Z], -1).reshape(-1, 3)
    _, (f, p) = gpt.bsdf_probe(mat, wi, dirs=dirs)
    exp = p.reshape(NT, K, NP, K).mean(axis=(1, 3)) * (1.0 / NT) * (2 * np.pi / NP) * N
    assert abs(exp.sum() + lost - N) < 0.02 * N, (exp.sum(), lost)
    o, e = obs.ravel(), exp.ravel()
    big = e >= 5
    stat = ((o[big] - e[big]) ** 2 / e[big]).sum()
    dof = int(big.sum()) - 1
    if (~big).any() and e[~big].sum() > 0:
        stat += (o[~big].sum() - e[~big].sum()) ** 2 / max(e[~big].sum(), 1e-9)
        dof += 1
    # the expected counts come from a 32 x 32 midpoint rule per cell: at 200 000 samples its quadrature error shows in the statistic for the sharpest
    # lobes, hence the 0.1 % level (the reference's test uses 1 % with an adaptive integrator)
    assert dof > 5 and stat < chi2.ppf(0.999, dof), (stat, dof, chi2.ppf(0.999, dof))
    # weight == eval * |cos| / pdf at the sampled direction (device sample() against device eval() / pdf())
    pick = np.flatnonzero(ok)[:4000]
    _, (f2, p2) = gpt.bsdf_probe(mat, wi, dirs=wo[pick])
    assert np.allclose(p2, pdf[pick], rtol=1e-9, atol=1e-300)
    assert np.allclose(weight[pick], f2 / p2[:, None], rtol=1e-9, atol=1e-12)      # eval() already carries the cosine (bsdf.h:398-410)


@pytest.mark.parametrize("mat", MATS + [scenes.conductor(**scenes.AL) if hasattr(scenes, "conductor") else scenes.diffuse((0.2, 0.3, 0.4)),
                                        dict(type=3, eta=(1.5, 1.5, 1.5), reflectance=(1.0, 0.9, 0.8), k=(0.7, 0.8, 0.9))])
def test_device_bsdf_equals_the_oracle_direction_by_direction(mat):
    import gpt_oracle as go
    gpt = _gpt()
    rng = np.random.default_rng(5)
    for wi in ((0.35, 0.2, 0.9), (0.1, -0.6, 0.4), (0.3, 0.3, -0.7)):
        wi = unit(wi)
        smp = rng.random((300, 2))
        (wo, weight, pdf, typ), _ = gpt.bsdf_probe(mat, wi, samples=smp)
        for i in range(len(smp)):
            owo, ow, opdf, otyp = go.bsdf_sample(mat, wi, smp[i, 0], smp[i, 1])
            assert otyp == typ[i] and np.isclose(opdf, pdf[i], rtol=1e-10, atol=1e-300), (i, opdf, pdf[i])
            assert np.allclose(ow, weight[i], rtol=1e-10, atol=1e-300) and np.allclose(owo, wo[i], rtol=0, atol=1e-12)
        dirs = np.array([unit(v) for v in rng.normal(size=(300, 3))])
        for measure in (0, 1):
            _, (f, p) = gpt.bsdf_probe(mat, wi, dirs=dirs, measure=measure)
            for i in range(0, len(dirs), 3):
                of, op = go.bsdf_eval_pdf(mat, wi, dirs[i], measure)
                assert np.allclose(of, f[i], rtol=1e-10, atol=1e-300) and np.isclose(op, p[i], rtol=1e-10, atol=1e-300), (measure, i, of, f[i], op, p[i])
