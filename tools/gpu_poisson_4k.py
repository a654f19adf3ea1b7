import sys, os
sys.path.insert(0, '/root/repo')
import torch
import gradientdomain_mitsuba_amd.poisson as P
from oracle import poisson_oracle as po
dev = torch.device("cuda", 0)
for (w, h) in ((3840, 2160), (2560, 1440)):
    dx, dy, tp, direct = po.synth_inputs(w, h)
    t = [torch.from_numpy(a.reshape(h, w, 3)).to(dev) for a in (dx, dy, tp, direct)]
    rec = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
    s = P.Solver(P.Params("L2D", 0.2))
    best = 1e9
    for rep in range(5):
        s.importImagesMTS(t[0], t[1], t[2], t[3], w, h); s.setupBackend(); s.solveIndirect(); s.exportImagesMTS(rec)
        best = min(best, s.lastSolveSeconds)
    k = s.profileKernels(30)
    print("%dx%d L2D %.3f ms; kernels us: Ax %.1f r_rz %.1f x_p %.1f xp_Ax %.1f -> xp_Ax %.2f TB/s" % (w, h, 1e3 * best, k[0], k[1], k[2], k[3], 72.0 * w * h / k[3] / 1e6))
    s.close()
