"""Fusion level 3 (single-gather Chronopoulos-Gear recurrence in the persistent CG) against level 2 and the oracle: differences and solve times."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import gradientdomain_mitsuba_amd.poisson as P
from oracle import poisson_oracle as po

for (w, h) in ((64, 48), (192, 100), (260, 37), (512, 512), (1280, 720)):
    dx, dy, tp, direct = po.synth_inputs(w, h)
    for preset in ("L2D", "L1D"):
        out = {}
        for lvl in (2, 3):
            s = P.Solver(P.Params(preset, 0.2)); s.setFusion(lvl)
            best = 1e9
            for rep in range(3):
                s.importImagesMTS(dx, dy, tp, direct, w, h); s.setupBackend(); s.solveIndirect()
                rec = s.exportImagesMTS().copy(); best = min(best, s.lastSolveSeconds)
            out[lvl] = (rec, best, s.profilePersistent(3))
            s.close()
        ref = po.solve(po.preset(preset), dx, dy, tp, direct, w, h) if (w * h <= 300000 or preset == "L2D") else None
        print("%4dx%-4d %s: level 2 %.3f ms (%.1f us/launch), level 3 %.3f ms (%.1f us/launch); |3 - 2| %.2e%s" % (w, h, preset, 1e3 * out[2][1], out[2][2], 1e3 * out[3][1], out[3][2],
              np.abs(out[3][0] - out[2][0]).max(), "" if ref is None else "; vs oracle: level 2 %.2e, level 3 %.2e" % (np.abs(out[2][0] - ref).max(), np.abs(out[3][0] - ref).max())), flush=True)
