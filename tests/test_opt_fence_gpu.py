"""The optimisation fence of VERDICT r2 #5: the two tracer units (G-PT, G-BDPT) compiled at -O1 (lib/libgdpt_hip_O1.so, `_build.build_fence`) against the -O3
product, same sources, same flags otherwise.  Both miscompiles met so far (tools/repro/README.md: one 16-byte unit of an offset's
throughput wrong in one k_render instantiation; a 4-wave variant faulting at address 0) were -O3-only and neither announced itself: this
test renders 600 fuzz seeds (tests/fence_worker.py) through both libraries, each in its own process, and asks for identical ray counts and
films equal to accumulation order (the film sums are fp64 atomics: 1e-12 of the buffer's maximum)."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIRST, COUNT = 91000, 600


def run_worker(lib, out):
    env = dict(os.environ)
    env.pop("GDPT_SCENE_IN_HBM", None)
    if lib:
        env["GDPT_LIB"] = lib
    else:
        env.pop("GDPT_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "fence_worker.py"), str(FIRST), str(COUNT), out], env=env, timeout=1500,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, "worker with %s failed (%d):\n%s" % (lib or "the product library", r.returncode, r.stdout[-3000:])
    return np.load(out)


@pytest.mark.gpu
def test_O1_and_O3_builds_of_the_tracer_render_the_same_films(tmp_path):
    b = importlib.import_module("gradientdomain-mitsuba_amd._build")
    assert os.path.exists(b.FENCE_LIB), "%s is not built: run __graft_entry__.build() (it builds the fence library too)" % b.FENCE_LIB
    assert b.fence_stamp() == b.fence_source_hash(), "the -O1 fence library was built from other sources than this tree's: run __graft_entry__.build()"
    o1 = run_worker(b.FENCE_LIB, str(tmp_path / "o1.npz"))
    o3 = run_worker(None, str(tmp_path / "o3.npz"))
    assert sorted(o1.files) == sorted(o3.files)
    films = [k for k in o3.files if k.endswith("/film")]
    assert len(films) >= 2 * COUNT
    assert sum(k.endswith("/gbdpt/film") for k in films) >= COUNT // 3
    strict = sum(int(o3[k[:-5] + "/strict"][0]) for k in films)
    assert strict >= len(films) // 5
    identical, worst = 0, 0.0
    for k in films:
        r = k[:-5] + "/rays"
        assert (o1[r] == o3[r]).all(), (k, o1[r], o3[r])
        a, c = o1[k], o3[k]
        assert np.isfinite(c).all(), k
        for buf in range(a.shape[0]):
            d = float(np.abs(a[buf] - c[buf]).max() / (np.abs(c[buf]).max() + 1e-300))
            worst = max(worst, d)
            assert d <= 1e-12, (k, buf, d)
        identical += int(np.array_equal(a, c))
    print("fence: %d films (%d with strictNormals), %d bit-identical, worst relative difference %.2e" % (len(films), strict, identical, worst))
