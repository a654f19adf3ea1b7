"""Phase clocks of the persistent CG kernel (needs a build with GDPT_EXTRA_FLAGS=-DGDPT_PT_TIMING)."""
import sys
sys.path.insert(0, '.')
import gradientdomain_mitsuba_amd.poisson as P
from oracle import poisson_oracle as po

for (w, h) in ((64, 48), (512, 512), (1280, 720)):
    dx, dy, tp, direct = po.synth_inputs(w, h)
    s = P.Solver(P.Params("L2D", 0.2)); s.setFusion(2)
    s.importImagesMTS(dx, dy, tp, direct, w, h)
    for _ in range(2):
        s.setupBackend(); s.solveIndirect()
    print(w, h, "%.3f ms" % (s.lastSolveSeconds * 1e3), flush=True)
    s.close()
