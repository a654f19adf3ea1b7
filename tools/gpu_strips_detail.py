import sys, os; sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/tools') else '.')
import numpy as np, torch
from gradientdomain_mitsuba_amd import gpt as G, scenes
W, H, spp, N = 3840, 2160, 1, 8
S = G.Scene(scenes.atrium(W, H))
integ = G.GradientPathIntegrator(maxDepth=-1)
cfg = integ.config(spp)
F = G.Film(S); integ.renderBlock(S, F, cfg, (0, 0, W, H)); acc = F.accum(); st = F.stats(); F.close()
heights = [404, 56, 35, 793, 36, 127, 582, 127]
b = np.concatenate([[0], np.cumsum(heights)]).tolist()
strips = [(b[i], b[i + 1]) for i in range(N)]
films = [G.Film(S, y0, y1) for (y0, y1) in strips]
for f, (y0, y1) in zip(films, strips): integ.renderBlock(S, f, cfg, (0, y0, W, y1))
n = films[0].halo_bytes() // 8
down = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(N)]; up = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(N)]
for r, f in enumerate(films):
    if r + 1 < N: f.pack_halo(1, down[r])
    if r > 0: f.pack_halo(0, up[r])
for r, f in enumerate(films):
    if r > 0: f.unpack_halo(0, down[r - 1])
    if r + 1 < N: f.unpack_halo(1, up[r + 1])
rays = [0, 0, 0]
for f, (y0, y1) in zip(films, strips):
    a = f.accum(); s2 = f.stats(); rays[0] += s2["raysTraced"]; rays[1] += s2["shadowRaysTraced"]; rays[2] += s2["paths"]
    for k in range(5):
        d = np.abs(a[k] - acc[k][y0:y1])
        bad = np.argwhere(d.max(-1) > 1e-9)
        if len(bad):
            print("strip", (y0, y1), "buffer", k, "pixels", len(bad), "first", [(int(y0 + yy), int(xx)) for yy, xx in bad[:8]])
            yy, xx = bad[0]
            print("   strips", a[k][yy, xx], "one film", acc[k][y0 + yy, xx])
    f.close()
print("rays", rays, "one film", st["raysTraced"], st["shadowRaysTraced"], st["paths"])
