"""One seed of tools/gpu_gbdpt_fuzz.py (same scene / configuration draw), with the difference of the films localised: per buffer and pixel, and -- by rendering
the device film pixel by pixel -- which pixel's samples the differing splats come from, with the probe's and the oracle's splat lists of those samples.
  gpurun -- 'GBDPT_FUZZ_SPECULAR=1 GBDPT_FUZZ_ENDPOINTS=1 python tools/gpu_gbdpt_fuzz_locate.py 930467'"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, gbdpt as B, scenes
from oracle import gpt_oracle as go

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
W, H = int(rng.integers(12, 36)), int(rng.integers(8, 28))
if os.environ.get("GBDPT_FUZZ_SPECULAR"):
    sc = scenes.veach_bidir(W, H, specular=True) if seed % 5 == 0 else scenes.cornell_box(W, H, "random", seed=seed)
else:
    sc = scenes.veach_bidir(W, H) if seed % 5 == 0 else scenes.cornell_box(W, H, "random_connectable", seed=seed)
md = int(rng.choice([-1, 1, 2, 3, 5, 8, 12, 16, 20])); rr = int(rng.choice([1, 3, 5])); li = bool(rng.random() < 0.7)
if os.environ.get("GBDPT_FUZZ_ENDPOINTS"):
    r2 = np.random.default_rng(1000003 * seed + 17)
    cornell = seed % 5 != 0
    if r2.random() < 0.5:
        sc.thinlens = (float(r2.uniform(2.0, 60.0)), float(r2.uniform(300.0, 1500.0))) if cornell else (float(r2.uniform(0.05, 0.5)), float(r2.uniform(5.0, 15.0)))
        md = min(md, 19)
    if cornell and r2.random() < 0.6:
        pls = [("point", tuple(float(v) for v in r2.uniform(40.0, 510.0, 3)), tuple(float(v) for v in r2.uniform(2e3, 5e4, 3))) for _ in range(int(r2.integers(1, 3)))]
        mode = int(r2.integers(0, 3))
        sc.emitters = sc.emitters + pls if mode == 0 else (pls + sc.emitters if mode == 1 else pls)
    if cornell and r2.random() < 0.35:
        if r2.random() < 0.25: sc.emitters = []
        sc.environment = (tuple(float(v) for v in r2.uniform(0.05, 1.5, 3)), int(r2.integers(0, len(sc.emitters) + 1)))
spp = int(rng.integers(1, 4))
print("seed", seed, "W H", W, H, "maxDepth", md, "rrDepth", rr, "lightImage", li, "spp", spp, "thinlens", getattr(sc, "thinlens", None), "emitters", [e[0] if isinstance(e, tuple) else e for e in sc.emitters], "environment", getattr(sc, "environment", None))
S = G.Scene(sc); O = go.Scene(sc)
integ = B.GBDPTIntegrator(maxDepth=md, rrDepth=rr, lightImage=li)
cfg = integ.config(spp, 5489 + seed); ocfg = go.gbdpt_config(maxDepth=md, rrDepth=rr, lightImage=li, spp=spp, seed=5489 + seed)
F = B.Film(S)
integ.renderBlock(S, F, cfg, (0, 0, W, H)); F.sync()
blk, lgt = F.accum(); print("device", F.stats(), F.chain_stats())
oblk, olgt, oc = O.gbdpt_render(ocfg); print("oracle", oc)
for name, a, b in (("block", blk, oblk), ("light", lgt, olgt)):
    d = np.abs(a - b); sc_ = np.abs(b).max()
    bad = np.argwhere(d.max(-1) > 1e-9 * sc_)
    print(name, "max diff", d.max(), "of", sc_, "entries", len(bad), bad[:12].tolist())
# which pixel's samples: the film of one pixel at a time
n = 0
for py in range(H):
    for px in range(W):
        F.clear(); integ.renderBlock(S, F, cfg, (px, py, px + 1, py + 1)); F.sync()
        b1, l1 = F.accum()
        ob1, ol1, _ = O.gbdpt_render(ocfg, rect=(px, py, px + 1, py + 1))
        if np.abs(l1 - ol1).max() > 1e-9 * max(np.abs(olgt).max(), 1e-300) or np.abs(b1 - ob1).max() > 1e-9 * max(np.abs(oblk).max(), 1e-300):
            n += 1
            if n > 3: continue
            print("pixel", px, py, "alone: light diff", np.abs(l1 - ol1).max(), "block diff", np.abs(b1 - ob1).max())
            for s in range(spp):
                g = integ.evaluate_sample(S, cfg, px, py, s); o = O.gbdpt_sample(ocfg, px, py, s)
                print("  sample", s, "general", g["general"], "probe == oracle:", np.allclose(g["primal"], o["primal"], rtol=1e-9, atol=0), np.asarray(g["light"]).shape, np.asarray(o["light"]).shape,
                      "rays", g["raysTraced"], o["raysTraced"], g["shadowRaysTraced"], o["shadowRaysTraced"])
                print("   oracle splats", np.asarray(o["light"]).reshape(-1, 6)[:, :4].tolist())
            bad = np.argwhere(np.abs(l1 - ol1).max(-1) > 1e-9 * max(np.abs(olgt).max(), 1e-300))
            for (bf, y, x) in bad[:6]: print("   film light[%d][%d][%d]: device" % (bf, y, x), l1[bf, y, x], "oracle", ol1[bf, y, x])
print("pixels whose own film differs:", n)
