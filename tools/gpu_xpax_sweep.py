"""Occupancy sweep of the fused x_p + stencil kernel of the screened-Poisson CG (kf_xp_Ax) at the HBM-resident size 3840x2160:
tile rows x resident blocks asked of the register allocator -> kernel time by HIP events (gdpt_poisson_profile_kernels), the whole L2D solve, and the
solution against the product variant's (another tiling adds the block partials of p.Ap in another order: rounding only).
  gpurun -- 'timeout 300 python tools/gpu_xpax_sweep.py'"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
import gradientdomain_mitsuba_amd.poisson as P
w, h = 3840, 2160
rng = np.random.default_rng(12345)
yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
gt = np.stack([0.5 + 0.4 * np.sin(0.2 * xx + c) * np.cos(0.15 * yy) for c in range(3)], axis=-1).astype(np.float32)
tp = (gt + 0.2 * (rng.random(gt.shape, dtype=np.float32) - 0.5)).astype(np.float32)
dx = np.zeros_like(gt); dx[:, :-1] = gt[:, 1:] - gt[:, :-1] + 0.01 * (rng.random((h, w - 1, 3), dtype=np.float32) - 0.5)
dy = np.zeros_like(gt); dy[:-1] = gt[1:] - gt[:-1] + 0.01 * (rng.random((h - 1, w, 3), dtype=np.float32) - 0.5)
direct = np.zeros_like(gt)
dx, dy, tp, direct = (a.reshape(-1) for a in (dx, dy, tp, direct))
sv = P.Solver(P.Params("L2D", 0.2))
sv.importImagesMTS(dx, dy, tp, direct, w, h); sv.setupBackend(); sv.solveIndirect()
t = []
for _ in range(3):
    sv.setupBackend(); sv.solveIndirect(); t.append(sv.lastSolveSeconds)
rec = np.empty((h, w, 3), np.float32); sv.exportImagesMTS(rec)
kus = sv.profileKernels(60)
np.save(sys.argv[1], rec)
print(json.dumps(dict(kus=[float(k) for k in kus], solve_ms=1e3 * sorted(t)[1])))
''' % ROOT
ref = None
print("%-10s %12s %12s %12s %14s" % ("rows,blk", "kf_xp_Ax us", "kf_r_rz us", "solve ms", "max|x - x_ref|"))
for var in (os.environ.get("XPAX_VARIANTS", "8,1 8,3 4,1 4,3 4,4 12,1").split()):
    env = dict(os.environ, GDPT_XPAX=var)
    out = "/tmp/xpax_%s.npy" % var.replace(",", "_")
    r = subprocess.run([sys.executable, "-c", CHILD, out], capture_output=True, text=True, env=env, timeout=280)
    if r.returncode != 0:
        print(var, "FAILED", r.stderr[-500:]); continue
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    import numpy as np
    rec = np.load(out)
    if ref is None:
        ref = rec
    print("%-10s %12.1f %12.1f %12.3f %14.3e" % (var, d["kus"][3], d["kus"][1], d["solve_ms"], float(np.abs(rec - ref).max())), flush=True)
