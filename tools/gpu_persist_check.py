import sys, time
sys.path.insert(0, '.')
import numpy as np
import gradientdomain_mitsuba_amd.poisson as P
from oracle import poisson_oracle as po

def run(preset, w, h, fusion, reps=1):
    dx, dy, tp, direct = po.synth_inputs(w, h)
    s = P.Solver(P.Params(preset, 0.2)); s.setFusion(fusion)
    s.importImagesMTS(dx, dy, tp, direct, w, h)
    best = 1e9
    for _ in range(reps):
        s.setupBackend(); s.solveIndirect(); best = min(best, s.lastSolveSeconds)
    rec = s.exportImagesMTS(); it = s.lastIterations; s.close()
    return rec, best, it

for (w, h) in ((64, 48), (256, 96), (512, 512), (1280, 720)):
    for preset in ("L2D", "L1D"):
        a, ta, _ = run(preset, w, h, 1, 3)
        b, tb, it = run(preset, w, h, 2, 3)
        ref = po.solve(po.preset(preset), *po.synth_inputs(w, h), w, h) if w * h <= 512 * 512 or preset == "L2D" else a
        print("%s %4dx%-4d fused %.3f ms (%.1f Gpix-it/s) | persistent %.3f ms (%.1f Gpix-it/s)  max|f2-f1| %.2e  max|f2-oracle| %.2e" % (
            preset, w, h, ta * 1e3, w * h * it / ta / 1e9, tb * 1e3, w * h * it / tb / 1e9, np.abs(a - b).max(), np.abs(b - ref).max()))
