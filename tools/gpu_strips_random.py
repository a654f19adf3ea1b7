"""One film against N strip films with RANDOM boundaries (the partitions bench.py's rebalance step produces are timing-dependent): the strips rendered one after the
other on one device, halos packed / unpacked as parallel.exchange_halos ships them.  Prints every partition whose strips differ from the one-film frame.
  gpurun -- 'python tools/gpu_strips_random.py [trials [W H [spp]]]'"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from gradientdomain_mitsuba_amd import gpt as G, scenes
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
N = 8
S = G.Scene(scenes.atrium(W, H))
integ = G.GradientPathIntegrator(maxDepth=-1)
cfg = integ.config(spp)
F = G.Film(S); integ.renderBlock(S, F, cfg, (0, 0, W, H)); acc = F.accum(); F.close()
rng = np.random.default_rng(12345)
bad = 0
for trial in range(trials):
    while True:
        cuts = np.sort(rng.choice(np.arange(2, H - 1), N - 1, replace=False))
        b = [0] + cuts.tolist() + [H]
        if min(b[i + 1] - b[i] for i in range(N)) >= 2: break
    if trial == 0: b = [0, 312, 563, 854, 1119, 1370, 1649, 1921, 2160] if H == 2160 else b
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
    worst = 0.0; where = None
    for f, (y0, y1) in zip(films, strips):
        a = f.accum()
        for k in range(5):
            d = np.abs(a[k] - acc[k][y0:y1]); sc_ = np.abs(acc[k]).max() + 1e-300
            if d.max() / sc_ > worst: worst = d.max() / sc_; yy, xx, cc = np.unravel_index(d.argmax(), d.shape); where = (k, int(y0 + yy), int(xx), int(cc), float(a[k][yy, xx, cc]), float(acc[k][y0 + yy, xx, cc]))
        f.close()
    flag = worst > 1e-12
    bad += flag
    print("trial", trial, "heights", [y1 - y0 for y0, y1 in strips], "worst relative difference %.3e" % worst, where if flag else "", flush=True)
print("partitions that differ:", bad, "of", trials)
