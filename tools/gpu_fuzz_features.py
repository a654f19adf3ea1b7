"""Fuzz run over the scene features of round 2 against the oracle: python tools/gpu_fuzz_features.py [first [count]].
Per seed a Cornell variant with, drawn from the seed: per-triangle texture coordinates (some degenerate) -> UV tangents; bitmap textures of
random size / filter type (nearest, bilinear, trilinear, ewa) / wrap modes / uv scale and offset / maxAnisotropy on random materials; an
environment map of random size, scale and rotation (or the constant environment, or none); a rectangle light with a random transform
(sampled as the shape samples itself); random integrator settings.  24 evaluatePoint probes and one whole film (five buffers, both ray
counters) through the staged and -- every third seed -- the single-kernel pipeline; every fourth seed through the HBM-scene builds.
Prints the first mismatch and exits non-zero, or a summary."""
import copy, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import fuzz_summary  # noqa: E402  (tools/fuzz_summary.py: the battery's one-line JSON record)
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, scenes
from oracle import gpt_oracle as go

def make(seed):
    rng = np.random.default_rng(seed)
    if seed % 4 == 3: os.environ["GDPT_SCENE_IN_HBM"] = "1"
    else: os.environ.pop("GDPT_SCENE_IN_HBM", None)
    W, H, spp = int(rng.integers(17, 40)), int(rng.integers(11, 30)), int(rng.integers(1, 5))
    variant = str(rng.choice(["diffuse", "glossy", "smooth", "bent", "twosided", "glass", "nearspecular"]))
    sc = scenes.cornell_box(W, H, variant)
    nt = sc.ntri
    what = []
    if rng.random() < 0.7:                                            # texture coordinates (-> UV tangents), some degenerate
        uvs = rng.uniform(-2, 3, (nt, 6)); has = (rng.random(nt) < 0.75).astype(np.uint8)
        deg = np.nonzero(has)[0][::5]
        uvs[deg, 2:4] = uvs[deg, 0:2] + 0.3; uvs[deg, 4:6] = uvs[deg, 0:2] + 0.6
        sc.uvs, sc.tri_has_uv = uvs, has
        what.append("uv")
    if rng.random() < 0.7:                                            # bitmap textures on one to three materials
        texs, mt = [], [-1] * len(sc.materials)
        for m in rng.choice(len(sc.materials), size=int(rng.integers(1, 4)), replace=False):
            w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
            img = scenes.checker_rgb(w, h, int(rng.integers(0, 1000))) * float(rng.choice([0.8, 1.0, 1.4]))
            texs.append(scenes.bitmap_texture(img, wrap=int(rng.integers(0, 5)), wrapV=int(rng.integers(0, 5)), filter=int(rng.integers(0, 4)),
                                              uscale=float(rng.choice([1.0, 3.0, 17.0, 60.0])), vscale=float(rng.choice([1.0, 2.5, 23.0])),
                                              uoffset=float(rng.uniform(-1, 1)), voffset=float(rng.uniform(-1, 1)), maxAnisotropy=float(rng.choice([1.0, 2.0, 8.0, 20.0]))))
            mt[int(m)] = len(texs) - 1
        sc.textures, sc.material_textures = texs, mt
        what.append("tex%d" % len(texs))
    e = rng.random()
    if e < 0.4:                                                       # environment map
        w, h = int(rng.integers(2, 80)), int(rng.integers(2, 40))
        a, b = rng.uniform(0, 2 * np.pi), rng.uniform(-1, 1)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        sc.environment_map = dict(rgb=scenes.sky_map(w, h, seed, sun=float(rng.choice([1.0, 40.0, 3000.0]))), scale=float(rng.uniform(0.3, 2.0)), toWorld=R,
                                  index=int(rng.integers(-1, len(sc.emitters) + 1)))
        what.append("envmap")
    elif e < 0.6:
        sc.environment = ((0.4, 0.5, 0.6), int(rng.integers(0, len(sc.emitters) + 1)))
        what.append("env")
    if rng.random() < 0.5:                                            # a rectangle light somewhere under the ceiling, facing down-ish
        eu = rng.normal(size=3) * 30; eu[1] *= 0.2
        ev = np.cross(eu, np.array([0.0, -1.0, 0.0]) + rng.normal(size=3) * 0.2); ev *= rng.uniform(10, 40) / np.linalg.norm(ev)
        corner = np.array([rng.uniform(100, 400), rng.uniform(300, 480), rng.uniform(100, 400)])
        n = np.cross(eu, ev); n /= np.linalg.norm(n)
        M = np.stack([eu, ev, n, corner + eu + ev], 1)
        P = lambda x, y: M @ np.array([x, y, 0.0, 1.0])
        v = [P(-1, -1), P(1, -1), P(1, 1), P(-1, 1)]
        firstTri = sc.ntri
        sc.verts = np.concatenate([np.asarray(sc.verts, np.float64).reshape(-1, 9), np.array([[v[0], v[1], v[2]], [v[2], v[3], v[0]]]).reshape(2, 9)])
        sc.tri_material = np.concatenate([np.asarray(sc.tri_material, np.int32), np.zeros(2, np.int32)])
        for name, width in (("normals", 9), ("uvs", 6)):
            if getattr(sc, name, None) is not None: setattr(sc, name, np.concatenate([np.asarray(getattr(sc, name)), np.zeros((2, width))]))
        if getattr(sc, "tri_has_uv", None) is not None: sc.tri_has_uv = np.concatenate([sc.tri_has_uv, np.zeros(2, np.uint8)])
        sc.emitters = list(sc.emitters) + [(firstTri, 2, (float(rng.uniform(5, 30)),) * 3, M, tuple(n))]
        if getattr(sc, "environment_map", None) is not None: sc.environment_map["index"] = min(sc.environment_map["index"], len(sc.emitters))
        what.append("rect")
    md = int(rng.choice([-1, 2, 3, 5, 8])); strict = bool(rng.random() < 0.3)
    return sc, W, H, spp, md, strict, variant, what

def oracle_spread(sc, ocfg, px, py, k, key, o):
    """How far the ORACLE's own value of one sample moves under few-ulp perturbations of the geometry: uniform and per-axis scalings, and --
    since a hit on an edge two triangles share is not moved off it by a scaling -- 32 seeded perturbations of every coordinate independently
    (+-4 ulp).  The rounding-noise floor of an ill-conditioned sample."""
    sens = 0.0
    v0 = np.asarray(sc.verts, np.float64).reshape(-1, 3)
    trials = [(ax, kk) for ax in (None, 0, 1, 2) for kk in (1, 2, 3, -1, -2, -3)] + [("each", t) for t in range(32)]
    for ax, kk in trials:
        v = v0.copy()
        if ax is None: v *= 1 + kk * 2.0 ** -52
        elif ax == "each": v *= 1 + np.random.default_rng(kk).integers(-4, 5, v.shape) * 2.0 ** -52
        else: v[:, ax] *= 1 + kk * 2.0 ** -52
        sc2 = copy.deepcopy(sc); sc2.verts = v.reshape(np.asarray(sc.verts).shape)
        O2 = go.Scene(sc2); o2 = O2.evaluate_point(ocfg, px, py, k); O2.close()
        sens = max(sens, float(np.abs(np.asarray(o2[key]) - np.asarray(o[key])).max()))
    return sens


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
worst = 0.0
knife = ill = 0
if __name__ != '__main__': first = count = 0
for seed in range(first, first + count):
    sc, W, H, spp, md, strict, variant, what = make(seed)
    rng = np.random.default_rng(seed + 10 ** 9)
    try:
        S = G.Scene(sc)
    except RuntimeError as ex:
        if "completely black" in str(ex): continue
        raise
    O = go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, strictNormals=strict)
    cfg, ocfg = integ.config(spp), go.config(maxDepth=md, spp=spp, strictNormals=strict)
    for _ in range(24):
        px, py, k = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g, o = S.evaluate_point(cfg, px, py, k), O.evaluate_point(ocfg, px, py, k)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            if not np.allclose(g[key], o[key], rtol=1e-8, atol=1e-12):
                print("MISMATCH seed", seed, variant, what, "sample", (px, py, k), key, g[key], o[key]); sys.exit(1)
    oacc, orays = O.render(ocfg)
    for stages in ((2, 0) if seed % 3 == 0 else (2,)):
        F = G.Film(S); F.set_pipeline(stages)
        integ.renderBlock(S, F, cfg, (0, 0, W, H))
        acc, st = F.accum(), F.stats()
        F.close()
        if (st["raysTraced"], st["shadowRaysTraced"]) != orays:
            # a knife-edge branch (a quantity that is zero in exact arithmetic decides whether a shadow ray is cast)?  Then the
            # buffers still agree; reported and counted, not fatal
            dmax = max(float(np.abs(acc[b] - oacc[b]).max() / (np.abs(oacc[b]).max() + 1e-300)) for b in range(5))
            print("ray counts differ: seed", seed, variant, what, stages, (st["raysTraced"], st["shadowRaysTraced"]), orays, "max film rel diff %.2e" % dmax, flush=True)
            knife += 1
            if dmax > 1e-8: sys.exit(1)
            continue
        for b in range(5):
            d = float(np.abs(acc[b] - oacc[b]).max() / (np.abs(oacc[b]).max() + 1e-300))
            if d <= 1e-8: worst = max(worst, d)
            if d > 1e-8:
                # an ill-conditioned sample?  Find the samples behind the worst pixel and ask the oracle how much ITS value moves
                # when the geometry is scaled by a few ulps; a difference within 20x of that spread is rounding noise of the sample
                dd = np.abs(acc[b] - oacc[b])[..., :3].max(-1)
                y, x = (int(v) for v in np.unravel_index(np.argmax(dd), dd.shape))
                explained = False
                for (px, py) in ((x, y), (x - 1, y), (x + 1, y), (x, y - 1), (x, y + 1)):
                    if not (0 <= px < W and 0 <= py < H): continue
                    for k in range(spp):
                        g, o = S.evaluate_point(cfg, px, py, k), O.evaluate_point(ocfg, px, py, k)
                        for key in ("throughput", "gradients", "neighbours"):
                            diff = float(np.abs(np.asarray(g[key]) - np.asarray(o[key])).max())
                            if diff <= 1e-11 * (1 + float(np.abs(np.asarray(o[key])).max())): continue
                            sens = oracle_spread(sc, ocfg, px, py, k, key, o)
                            if diff > 20 * sens:
                                print("FILM MISMATCH seed", seed, variant, what, stages, G.BUFFER_NAMES[b], d, "sample", (px, py, k), key, diff, "oracle spread", sens); sys.exit(1)
                            explained = True
                if not explained:
                    print("FILM MISMATCH (no differing sample found) seed", seed, variant, what, stages, G.BUFFER_NAMES[b], d); sys.exit(1)
                print("ill-conditioned sample: seed", seed, variant, what, G.BUFFER_NAMES[b], "film rel diff %.2e within the oracle's own spread" % d, flush=True)
                ill += 1
                break
    S.close(); O.close()
    if (seed - first) % 50 == 49: print("seed %d ok, worst film rel diff %.2e, %.0f s" % (seed, worst, time.time() - t0), flush=True)
print("OK: seeds %d..%d, %d films with a ray-count difference and equal buffers, %d with an ill-conditioned sample (within 20x of the oracle's own spread), worst other film rel diff %.2e, %.0f s" % (first, first + count - 1, knife, ill, worst, time.time() - t0))
fuzz_summary.emit("gpu_fuzz_features", first, count, time.time() - t0, knife_edge_ray_count_films=knife, ill_conditioned_films=ill, worst_film_rel_diff=worst)
