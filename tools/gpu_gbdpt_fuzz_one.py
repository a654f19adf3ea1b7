"""One seed of tools/gpu_gbdpt_fuzz.py, film only: where the film differs from the oracle (buffer, pixel), and the samples of that pixel through the probe entry."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
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
md = int(rng.choice([-1, 1, 2, 3, 5, 8, 12])); rr = int(rng.choice([1, 3, 5])); li = bool(rng.random() < 0.7)
spp = int(rng.integers(1, 4))
print("seed", seed, W, H, "maxDepth", md, "rr", rr, "lightImage", li, "spp", spp)
S = G.Scene(sc); O = go.Scene(sc)
integ = B.GBDPTIntegrator(maxDepth=md, rrDepth=rr, lightImage=li)
cfg = integ.config(spp, 5489 + seed); ocfg = go.gbdpt_config(maxDepth=md, rrDepth=rr, lightImage=li, spp=spp, seed=5489 + seed)
for rep in range(3):
    F = B.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H)); F.sync()
    blk, lgt = F.accum(); st = F.stats(); F.close()
    oblk, olgt, orays = O.gbdpt_render(ocfg)
    d = np.abs(blk - oblk)
    b, y, x, ch = np.unravel_index(d.argmax(), d.shape)
    print("rep", rep, "block max diff %.3e at buffer %d pixel (%d, %d) channel %d: %r vs %r; light max diff %.3e; rays %r vs %r" % (d.max(), b, x, y, ch, blk[b, y, x], oblk[b, y, x], np.abs(lgt - olgt).max(), (st["raysTraced"], st["shadowRaysTraced"]), (orays["raysTraced"], orays["shadowRaysTraced"])))
# the samples of that pixel through the probe
tot = np.zeros(15)
for s_ in range(spp):
    g = integ.evaluate_sample(S, cfg, int(x), int(y), s_); o = O.gbdpt_sample(ocfg, int(x), int(y), s_)
    print(" sample", s_, "probe primal", g["primal"], "oracle", o["primal"], "pos", g["position"])
    print("   probe gradients", np.asarray(g["gradients"]).ravel(), "\n   oracle gradients", np.asarray(o["gradients"]).ravel(), "rays", (g["raysTraced"], g["shadowRaysTraced"]), (o["raysTraced"], o["shadowRaysTraced"]))

# every pixel whose block sums differ, with its samples (a dropped put shows as a weight difference of 1)
bad = np.argwhere(np.abs(blk - oblk).max(axis=(0, 3)) > 1e-9 * max(np.abs(oblk).max(), 1e-300))
print("pixels that differ:", bad[:10].tolist(), "invalid puts", st.get("invalidPuts"), orays.get("invalidPuts"))
for (yy, xx) in bad[:4]:
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            px, py = int(xx) + dx, int(yy) + dy
            if not (0 <= px < W and 0 <= py < H): continue
            for s_ in range(spp):
                g = integ.evaluate_sample(S, cfg, px, py, s_); o = O.gbdpt_sample(ocfg, px, py, s_)
                sc_ = max(np.abs(o["primal"]).max(), np.abs(o["gradients"]).max(), 1e-300)
                e = max(np.abs(np.asarray(g["primal"]) - o["primal"]).max(), np.abs(np.asarray(g["gradients"]) - o["gradients"]).max())
                fin = np.isfinite(np.asarray(g["primal"])).all() and np.isfinite(np.asarray(g["gradients"])).all() and np.isfinite(o["primal"]).all() and np.isfinite(o["gradients"]).all()
                if e > 1e-9 * sc_ or not fin or (g["raysTraced"], g["shadowRaysTraced"]) != (o["raysTraced"], o["shadowRaysTraced"]):
                    print("  pixel", (px, py), "sample", s_, "general", g.get("general"), "err %.3e scale %.3e finite %s" % (e, sc_, fin), "rays", (g["raysTraced"], g["shadowRaysTraced"]), (o["raysTraced"], o["shadowRaysTraced"]))
                    print("    probe ", np.asarray(g["primal"]), np.asarray(g["gradients"]).ravel())
                    print("    oracle", o["primal"], np.asarray(o["gradients"]).ravel())
