"""Poisson solve time and Gpix-iter/s at the BASELINE image sizes (device-resident inputs), L2D and L1D."""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import gradientdomain_mitsuba_amd.poisson as P
from oracle import poisson_oracle as po

dev = torch.device("cuda", 0)
for (w, h) in ((512, 512), (1280, 720), (1920, 1080), (3840, 2160)):
    dx, dy, tp, direct = po.synth_inputs(w, h)
    t = [torch.from_numpy(a.reshape(h, w, 3)).to(dev) for a in (dx, dy, tp, direct)]
    rec = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
    for preset in ("L2D", "L1D"):
        prm = P.Params(preset, 0.2)
        s = P.Solver(prm)
        best = 1e9
        for rep in range(4):
            s.importImagesMTS(t[0], t[1], t[2], t[3], w, h); s.setupBackend(); s.solveIndirect(); s.exportImagesMTS(rec)
            best = min(best, s.lastSolveSeconds)
        iters = prm.irlsIterMax * prm.cgIterMax
        print("%4dx%-4d %s: %8.3f ms  %6.1f Gpix-iter/s  (persistent: %s)" % (w, h, preset, 1e3 * best, w * h * iters / best / 1e9, s.usedPersistent if hasattr(s, "usedPersistent") else "?"), flush=True)
        s.close()
