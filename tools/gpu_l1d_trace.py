"""One L1D (and L2D) solve at a given size, for a kernel trace: python tools/gpu_l1d_trace.py [W H]  (wrap in tools/kt.sh under timeout)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import gradientdomain_mitsuba_amd.poisson as P
from oracle import poisson_oracle as po
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
dev = torch.device("cuda", 0)
dx, dy, tp, direct = po.synth_inputs(w, h)
t = [torch.from_numpy(a.reshape(h, w, 3)).to(dev) for a in (dx, dy, tp, direct)]
rec = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
for preset in ("L1D",):
    s = P.Solver(P.Params(preset, 0.2))
    for rep in range(3):
        s.importImagesMTS(t[0], t[1], t[2], t[3], w, h); s.setupBackend(); s.solveIndirect(); s.exportImagesMTS(rec)
    print(preset, "%.3f ms" % (1e3 * s.lastSolveSeconds))
    s.close()
