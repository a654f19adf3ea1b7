"""kp_cg2 (128-px tiles, 8 px per lane) against the multi-kernel graphs (GDPT_NO_WIDE_PERSISTENT=1) and the oracle at sizes only it covers."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import gradientdomain_mitsuba_amd.poisson as P
from oracle import poisson_oracle as po

dev = torch.device("cuda", 0)
for (w, h) in (((1920, 1080), (1604, 904)) if len(sys.argv) > 1 else ((1920, 1080), (1600, 900), (2048, 1024), (1604, 904), (1412, 1300))):
    dx, dy, tp, direct = po.synth_inputs(w, h)
    t = [torch.from_numpy(a.reshape(h, w, 3)).to(dev) for a in (dx, dy, tp, direct)]
    for preset in ("L2D", "L1D"):
        out = {}
        for mode in ("wide", "graphs"):
            if mode == "graphs":
                os.environ["GDPT_NO_WIDE_PERSISTENT"] = "1"
            else:
                os.environ.pop("GDPT_NO_WIDE_PERSISTENT", None)
            prm = P.Params(preset, 0.2)
            s = P.Solver(prm)
            rec = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
            best = 1e9
            for rep in range(3):
                s.importImagesMTS(t[0], t[1], t[2], t[3], w, h); s.setupBackend(); s.solveIndirect(); s.exportImagesMTS(rec)
                best = min(best, s.lastSolveSeconds)
            out[mode] = (rec.cpu().numpy().copy(), best, getattr(s, "usedPersistent", None))
            s.close()
        a, b = out["wide"][0], out["graphs"][0]
        scale = np.abs(b).max()
        line = "%4dx%-4d %s: wide %.3f ms (persistent %s) | graphs %.3f ms | max |diff| / max = %.2e" % (w, h, preset, 1e3 * out["wide"][1], out["wide"][2], 1e3 * out["graphs"][1], np.abs(a - b).max() / scale)
        if preset == "L2D" and w * h <= 2100000:
            op = po.Params(); po.lib().gdo_params_preset(op, preset.encode()); op.alpha = 0.2
            ref = po.solve(op, dx, dy, tp, direct, w, h)
            line += " | vs oracle %.2e" % (np.abs(a.reshape(-1) - np.asarray(ref).reshape(-1)).max() / scale)
        print(line, flush=True)
