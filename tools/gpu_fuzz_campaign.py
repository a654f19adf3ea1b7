"""Long fuzz run of the HIP tracer against the oracle, beyond the seeds in tests/: python tools/gpu_fuzz_campaign.py [first [count]].
Per seed: a fuzzed Cornell scene (random materials / settings; every third with an environment, every fifth with vertex normals or a
point light), 40 evaluatePoint probes and one small whole film (five buffers + both ray counters); every seventh seed the atrium (HBM
BVH) with a fuzzed camera-independent sample set.  Prints the first mismatch and exits non-zero, or a summary."""
import sys, time, copy
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import fuzz_summary  # noqa: E402  (tools/ is on sys.path: the script's own directory)
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, scenes
from oracle import gpt_oracle as go

def perturbed(sc):
    """The scene with its geometry scaled by a few ulps: the whole of it, then one axis at a time (40 variants).  What the oracle itself
    returns for them is the rounding-noise floor of a sample."""
    v0 = np.asarray(sc.verts, np.float64).reshape(-1, 3)
    for ax in (None, 0, 1, 2):
        for k in (1, 2, 3, 4, 6, 8, -1, -2, -3, -4):
            v = v0.copy()
            if ax is None:
                v *= 1 + k * 2.0 ** -52
            else:
                v[:, ax] *= 1 + k * 2.0 ** -52
            sc2 = copy.deepcopy(sc)
            sc2.verts = v.reshape(np.asarray(sc.verts).shape)
            yield sc2
    yield from rotated(sc)          # (round 5: and the scene turned as a whole by tiny angles, below)


def rotated(sc):
    """The scene (geometry, normals, camera, point lights) turned as a whole by k x 2^-30 rad about each axis (18 variants): the same picture with every number rounded
    afresh -- unlike the scalings above it also moves the SAMPLED directions (the frames stop being axis-aligned), so it stands in for the one-ulp differences between
    the device's and glibc's transcendentals.  Scenes whose emitters carry transforms of their own (rectangle lights, an environment map) are left out."""
    if sc.environment_map is not None or any(isinstance(e[0], str) and e[0] != "point" for e in sc.emitters) or any((not isinstance(e[0], str)) and len(e) > 3 for e in sc.emitters):
        return
    if sc.environment is not None:       # (the environment's sphere is drawn about the scene's AXIS-ALIGNED box, which is another box after a rotation: the oracle's values move by
        return                           #  ~4e-10 with the angle itself, not with rounding -- measured; such scenes keep the scalings only)
    v0 = np.asarray(sc.verts, np.float64).reshape(-1, 3)
    n0 = None if sc.normals is None else np.asarray(sc.normals, np.float64).reshape(-1, 3)
    for ax in range(3):
        i, j = [(1, 2), (2, 0), (0, 1)][ax]
        for k in (1, -1, 2, -2, 3, -3):
            th = k * 2.0 ** -30
            R = np.eye(3); R[i, i] = R[j, j] = np.cos(th); R[i, j] = -np.sin(th); R[j, i] = np.sin(th)
            sc2 = copy.deepcopy(sc)
            sc2.verts = (v0 @ R.T).reshape(np.asarray(sc.verts).shape)
            if n0 is not None: sc2.normals = (n0 @ R.T).reshape(np.asarray(sc.normals).shape)
            M = np.array(sc.to_world, np.float64).copy(); M[:3, :3] = R @ M[:3, :3]; M[:3, 3] = R @ M[:3, 3]
            sc2.to_world = M
            sc2.emitters = [(e[0], tuple(R @ np.asarray(e[1], np.float64)), *e[2:]) if isinstance(e[0], str) else e for e in sc.emitters]
            yield sc2


FILM_SCALE = int(os.environ.get("FUZZ_FILM_SCALE", "1")); PROBES = int(os.environ.get("FUZZ_PROBES", "40"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t0 = time.time()
probes = films = illcond = knife = illfilm = 0
worst = worst_ill = 0.0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(17, 44)), int(rng.integers(9, 34))
    W, H = W * FILM_SCALE, H * FILM_SCALE       # (FUZZ_FILM_SCALE=5: films of up to 215 x 165 -- many tiles, waves that refill from long lists, several chunks; same scenes and settings per seed)
    kind = "random"
    kw = dict(seed=seed, environment=(0.5, 0.7, 0.9) if seed % 3 == 0 else None)
    if seed % 5 == 1:
        kind = "smooth" if seed % 2 else "bent"; kw = dict(environment=kw["environment"])
    if seed % 5 == 2:
        kw["point_light"] = ((float(rng.uniform(100, 450)), float(rng.uniform(200, 500)), float(rng.uniform(100, 450))), (4e4, 3e4, 2e4), bool(seed % 2))
    if seed % 7 == 0:
        sc = scenes.atrium(W, H, columns=int(rng.integers(4, 12)), segments=int(rng.integers(6, 16)))
    else:
        sc = scenes.cornell_box(W, H, kind, **kw)
    if seed % 5 == 3 and seed % 7 != 0 and kind == "random":
        # (round 6, r06h: shading normals tilted off the face normals on the RANDOM-material box -- per-vertex data with glossy vertices: the in-place kernels' per-vertex builds, in HBM
        #  the exact <false, true> ones; a generator of its own, so the other draws of the seed stay what they were)
        r2 = np.random.default_rng(seed + 7777777)
        v = np.asarray(sc.verts, np.float64).reshape(-1, 3, 3)
        n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]); n /= np.linalg.norm(n, axis=1, keepdims=True)
        nv = [n + r2.uniform(-0.08, 0.08, 3) for _ in range(3)]
        sc.normals = np.concatenate([a / np.linalg.norm(a, axis=1, keepdims=True) for a in nv], axis=1)
        for e in sc.emitters:
            if not isinstance(e[0], str):
                sc.normals[int(e[0]):int(e[0]) + int(e[1])] = 0.0
    if seed % 9 == 4:                                      # a thin lens instead of the pinhole (two more random numbers per sample, rays from the aperture)
        sc.thinlens = (float(rng.uniform(2.0, 60.0)), float(rng.uniform(300.0, 1500.0))) if seed % 7 else (float(rng.uniform(0.01, 0.3)), float(rng.uniform(2.0, 30.0)))
    if seed % 11 == 3:                                     # a reconstruction filter wider than box (sample log + gather)
        sc.rfilter = scenes.RFILTER_DEFAULTS[1 + seed % 5]
    md = int(rng.choice([-1, 2, 3, 5, 9])); rr = int(rng.choice([1, 3, 5])); strict = bool(rng.random() < 0.35); thr = float(rng.choice([0.001, 0.02, 0.0]))
    spp = int(rng.integers(1, 7))
    S = G.Scene(sc); O = go.Scene(sc)
    integ = G.GradientPathIntegrator(maxDepth=md, rrDepth=rr, strictNormals=strict, shiftThreshold=thr)
    cfg = integ.config(spp); ocfg = go.config(maxDepth=md, rrDepth=rr, strictNormals=strict, spp=spp, shiftThreshold=thr)
    for _ in range(PROBES):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g = S.evaluate_point(cfg, px, py, s); o = O.evaluate_point(ocfg, px, py, s)
        for key in ("veryDirect", "throughput", "gradients", "neighbours"):
            # (a gradient is the DIFFERENCE of two path contributions, w (f(offset) - f(base)): where they nearly cancel its error relative to itself has no bound in any
            #  floating-point implementation, so its bar is 1e-9 of the two contributions it is the difference of -- seed 2515818: a base path through an alpha = 0.0023
            #  conductor lobe, throughput equal to 5e-12, gradient 1/180 of it and "off" by 1.3e-9 of itself)
            atol = 1e-13 + (1e-9 * (np.abs(o["throughput"])[None, :] + np.abs(o["neighbours"])) if key == "gradients" else 0.0)
            if not (np.abs(g[key] - o[key]) <= atol + 1e-9 * np.abs(o[key])).all():
                # ill-conditioned sample (near-specular lobes: D(h) ~ 1/alpha^2)?  Measure the oracle's own sensitivity to few-ulp scalings of
                # the geometry; a difference within 20x of that is rounding noise of the sample, not of the implementation
                sens = np.zeros_like(o[key])
                for sc2 in perturbed(sc):
                    O2 = go.Scene(sc2)
                    sens = np.maximum(sens, np.abs(O2.evaluate_point(ocfg, px, py, s)[key] - o[key]))
                    O2.close()
                if (np.abs(g[key] - o[key]) <= 20 * sens + 1e-9 * np.abs(o[key]) + 1e-13).all():
                    illcond += 1
                    worst_ill = max(worst_ill, float((np.abs(g[key] - o[key]) / np.abs(o[key]).clip(1e-300)).max()))
                    continue
                print("MISMATCH sample: seed %d px %d py %d s %d %s\n%r\n%r\nsensitivity %r" % (seed, px, py, s, key, g[key], o[key], sens)); sys.exit(1)
        probes += 1
    F = G.Film(S)
    F.set_slices(int(rng.integers(0, spp + 1))); F.set_regeneration(int(rng.choice([1, 24, 56, 64])))
    # (round 6: every other film with a sample-queue budget of 2 / 8 / 16 / 24 MiB, i.e. cut into chunks of one to a few samples -- the pipelined chunks' set reuse and stream waits;
    #  the draw comes from a generator of its own so that the seeds' scenes and probes stay what they were)
    os.environ.pop("GDPT_QUEUE_MB", None)
    if seed % 2:
        os.environ["GDPT_QUEUE_MB"] = str((2, 8, 16, 24)[(seed // 2) % 4])
    integ.renderBlock(S, F, cfg, (0, 0, W, H))
    os.environ.pop("GDPT_QUEUE_MB", None)
    acc = F.accum(); st = F.stats()
    oacc, orays = O.render(ocfg)
    if (st["raysTraced"], st["shadowRaysTraced"]) != orays:
        # a branch on a quantity at rounding level (a light sample exactly at grazing incidence, a lobe value at the underflow edge)?  Find the
        # samples whose counts differ and ask the oracle itself: if one-ulp scalings of the geometry make ITS count take the HIP value, the
        # sample sits on a knife edge and the difference is rounding noise
        explained = True
        for py in range(H):
            for px in range(W):
                for s in range(spp):
                    g = S.evaluate_point(cfg, px, py, s); o = O.evaluate_point(ocfg, px, py, s)
                    if (g["raysTraced"], g["shadowRaysTraced"]) == (o["raysTraced"], o["shadowRaysTraced"]):
                        continue
                    seen = set()
                    for sc2 in perturbed(sc):
                        O2 = go.Scene(sc2); r = O2.evaluate_point(ocfg, px, py, s); O2.close()
                        seen.add((r["raysTraced"], r["shadowRaysTraced"]))
                    ok = (g["raysTraced"], g["shadowRaysTraced"]) in seen
                    print("note: seed %d pixel (%d, %d) sample %d: HIP rays %d + %d, oracle %d + %d; oracle under few-ulp scalings: %s -> %s" % (
                        seed, px, py, s, g["raysTraced"], g["shadowRaysTraced"], o["raysTraced"], o["shadowRaysTraced"], sorted(seen), "knife edge" if ok else "UNEXPLAINED"), flush=True)
                    explained &= ok
        if not explained:
            print("MISMATCH ray counts: seed %d %r %r" % (seed, st, orays)); sys.exit(1)
        knife += 1
        F.close(); S.close(); O.close()
        continue
    for b in range(5):
        scale = np.abs(oacc[b]).max() + 1e-300
        d = np.abs(acc[b] - oacc[b]).max() / scale
        worst = max(worst, d)
        if d > 1e-12:
            print("note: seed %d buffer %d rel diff %.2e (%s, %dx%d, spp %d, maxDepth %d)" % (seed, b, d, "atrium" if seed % 7 == 0 else kind, W, H, spp, md), flush=True)
        if d > 1e-9:
            # the same question for a film: how far does the oracle's own film move under few-ulp scalings of the geometry?
            spread = 0.0
            for n, sc2 in enumerate(perturbed(sc)):
                if n % 5 == 0:
                    O2 = go.Scene(sc2); spread = max(spread, float(np.abs(O2.render(ocfg)[0][b] - oacc[b]).max() / scale)); O2.close()
            if d > 20 * spread:
                print("MISMATCH film: seed %d buffer %d rel %g (oracle's own spread %g)" % (seed, b, d, spread)); sys.exit(1)
            illfilm += 1
    films += 1
    F.close(); S.close(); O.close()
    if (seed - first) % 20 == 19:
        print("seed %d: %d probes, %d films ok, worst film rel diff %.2e, %.0f s" % (seed, probes, films, worst, time.time() - t0), flush=True)
print("OK: seeds %d..%d, %d probes (%d outputs beyond 1e-9 on ill-conditioned samples, worst %.1e, each within 20x of the oracle's own sensitivity to few-ulp scalings of the geometry), %d films + %d with a knife-edge ray-count difference (%d film buffers beyond 1e-9, within 20x of the oracle's own spread), worst film rel diff %.2e, %.0f s" % (first, first + count - 1, probes, illcond, worst_ill, films, knife, illfilm, worst, time.time() - t0))
fuzz_summary.emit("gpu_fuzz_campaign", first, count, time.time() - t0, film_scale=FILM_SCALE, probes=probes, films=films, ill_conditioned_outputs=illcond, worst_ill_conditioned_rel=worst_ill, knife_edge_ray_count_films=knife, ill_conditioned_film_buffers=illfilm, worst_film_rel_diff=worst, bars="samples rtol 1e-9 / atol 1e-13 (gradients: + 1e-9 of the two contributions they are the difference of), films 1e-9 of the buffer scale; beyond: within 20x of the oracle's own sensitivity to few-ulp scalings / 2^-30 rad rotations")
