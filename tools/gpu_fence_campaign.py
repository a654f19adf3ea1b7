"""The -O1 / -O3 fence beyond the 600 seeds of tests/test_opt_fence_gpu.py: python tools/gpu_fence_campaign.py [first [count]].  Runs
tests/fence_worker.py through lib/libgdpt_hip_O1.so and through the product library in chunks of 500 seeds (a process each) and compares
every film and ray counter.  Prints the first difference and exits non-zero, or a summary."""
import importlib, os, subprocess, sys, tempfile, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import fuzz_summary  # noqa: E402  (tools/fuzz_summary.py: the battery's one-line JSON record)
import numpy as np
b = importlib.import_module("gradientdomain-mitsuba_amd._build")

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
t0 = time.time()
films = identical = 0
worst = 0.0
tmp = tempfile.mkdtemp()
for lo in range(first, first + count, 500):
    n = min(500, first + count - lo)
    out = {}
    for name, lib in (("o1", b.FENCE_LIB), ("o3", None)):
        env = dict(os.environ); env.pop("GDPT_SCENE_IN_HBM", None); env.pop("GDPT_LIB", None)
        if lib:
            env["GDPT_LIB"] = lib
        p = os.path.join(tmp, name + ".npz")
        subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fence_worker.py"), str(lo), str(n), p], env=env, check=True, timeout=3000)
        out[name] = np.load(p)
    for k in out["o3"].files:
        if k.endswith("/rays") and not (out["o1"][k] == out["o3"][k]).all():
            print("DIFFERENT ray counts: %s %r %r" % (k, out["o1"][k], out["o3"][k])); sys.exit(1)
        if k.endswith("/film"):
            a, c = out["o1"][k], out["o3"][k]
            for buf in range(a.shape[0]):
                d = float(np.abs(a[buf] - c[buf]).max() / (np.abs(c[buf]).max() + 1e-300))
                worst = max(worst, d)
                if not d <= 1e-12:
                    print("DIFFERENT film: %s buffer %d rel %g" % (k, buf, d)); sys.exit(1)
            films += 1; identical += int(np.array_equal(a, c))
    print("seeds %d..%d: %d films, %d bit-identical, worst rel %.2e, %.0f s" % (first, lo + n - 1, films, identical, worst, time.time() - t0), flush=True)
print("OK: seeds %d..%d through -O1 and -O3: %d films, ray counts identical, %d films bit-identical, worst relative difference %.2e" % (first, first + count - 1, films, identical, worst))
fuzz_summary.emit("gpu_fence_campaign", first, count, 0.0, films=films, bit_identical_films=identical, worst_rel_diff=worst, ray_counts_identical=True)
