"""Fuzz run of the G-BDPT kernels against the oracle, beyond the seeds in tests/: python tools/gpu_gbdpt_fuzz.py [first [count]].
Per seed: a Cornell box whose four free surfaces draw connectable materials from the seed (diffuse / rough conductors of all three distributions,
anisotropic, one- or two-sided), or (every fifth) the Veach-bidir stand-in; random maxDepth / rrDepth / lightImage; 24 single samples through the
probe entry (primal, four gradients, film position, every light splat, both ray counters) and one small whole film through the wavefront kernels
(camera blocks, light images, ray counters).  Prints the first mismatch and exits non-zero, or a summary.
GBDPT_FUZZ_SPECULAR=1: the free surfaces draw from EVERY material (smooth conductors, dielectrics, rough conductors on both sides of shiftThreshold) and the
Veach-class room comes with its glass egg and mirror: samples with specular chains, i.e. the general form (csrc/gbdpt_general.hip.h).
GBDPT_FUZZ_ENDPOINTS=1: half of the seeds get a thinlens sensor, most Cornell seeds one or two point emitters beside, before or instead of the area light."""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import fuzz_summary  # noqa: E402  (tools/fuzz_summary.py: the battery's one-line JSON record)
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, gbdpt as B, scenes
from oracle import gpt_oracle as go

import copy


def perturbed(sc):
    """The scene with its geometry scaled by a few ulps: the whole of it, then one axis at a time (40 variants).  What the oracle itself
    returns for them is the rounding-noise floor of a sample (as in tools/gpu_fuzz_campaign.py)."""
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
    if sc.environment is not None:       # (to libbidir the environment is a sphere about the scene's AXIS-ALIGNED box: that box is not the same box after a rotation, so the turned scene
        return                           #  is another scene at the 1e-9 level, not the same one rounded afresh)
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


first = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
probes = films = knife = discont = illcond = 0
worst = 0.0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(12, 36)), int(rng.integers(8, 28))
    if os.environ.get("GBDPT_FUZZ_SPECULAR"):       # round 4: everything the path carries -- conductors, dielectrics, rough conductors on both sides of shiftThreshold: the general form
        sc = scenes.veach_bidir(W, H, specular=True) if seed % 5 == 0 else scenes.cornell_box(W, H, "random", seed=seed)
    else:
        sc = scenes.veach_bidir(W, H) if seed % 5 == 0 else scenes.cornell_box(W, H, "random_connectable", seed=seed)
    md = int(rng.choice([-1, 1, 2, 3, 5, 8, 12, 16, 20])); rr = int(rng.choice([1, 3, 5])); li = bool(rng.random() < 0.7)
    if os.environ.get("GBDPT_FUZZ_ENDPOINTS"):      # round 5: the endpoints G-BDPT refused until then -- a thinlens sensor, point emitters beside / before / instead of the area light
        r2 = np.random.default_rng(1000003 * seed + 17)
        cornell = seed % 5 != 0
        if r2.random() < 0.5:
            sc.thinlens = (float(r2.uniform(2.0, 60.0)), float(r2.uniform(300.0, 1500.0))) if cornell else (float(r2.uniform(0.05, 0.5)), float(r2.uniform(5.0, 15.0)))
            md = min(md, 19)
        if cornell and r2.random() < 0.6:
            pls = [("point", tuple(float(v) for v in r2.uniform(40.0, 510.0, 3)), tuple(float(v) for v in r2.uniform(2e3, 5e4, 3))) for _ in range(int(r2.integers(1, 3)))]
            mode = int(r2.integers(0, 3))
            sc.emitters = sc.emitters + pls if mode == 0 else (pls + sc.emitters if mode == 1 else pls)
        if cornell and r2.random() < 0.35:              # a `constant` environment (the box is open at the front), anywhere in the emitter list; now and then the only emitter
            if r2.random() < 0.25: sc.emitters = []
            index = int(r2.integers(0, len(sc.emitters) + 1))
            if r2.random() < 0.5 or sc.thinlens is not None: sc.environment = (tuple(float(v) for v in r2.uniform(0.05, 1.5, 3)), index)   # (a lens + an envmap: refused at scene creation, the G-PT side's filtered lookup)
            else:                                       # an `envmap` one (random small map, now and then with a sun texel; rotated about y)
                a = float(r2.uniform(0.0, 6.28))
                sc.environment_map = dict(rgb=scenes.sky_map(int(r2.choice([8, 16, 32])), int(r2.choice([4, 8, 16])), seed=int(r2.integers(0, 1 << 30)), sun=float(r2.choice([1.5, 40.0]))),
                                          scale=float(r2.uniform(0.3, 2.0)), index=index, toWorld=[[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]])
    spp = int(rng.integers(1, 4))
    S = G.Scene(sc); O = go.Scene(sc)
    integ = B.GBDPTIntegrator(maxDepth=md, rrDepth=rr, lightImage=li)
    cfg = integ.config(spp, 5489 + seed); ocfg = go.gbdpt_config(maxDepth=md, rrDepth=rr, lightImage=li, spp=spp, seed=5489 + seed)
    for _ in range(24):
        px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, spp))
        g = integ.evaluate_sample(S, cfg, px, py, s); o = O.gbdpt_sample(ocfg, px, py, s)
        assert o["unsupported"] == 0, ("oracle scope", seed, px, py, s)
        scale = max(np.abs(o["primal"]).max(), np.abs(o["gradients"]).max(), 1e-300)
        gl, ol = np.asarray(g["light"]).reshape(-1, 6), np.asarray(o["light"]).reshape(-1, 6)
        bad = None
        for key in ("primal", "gradients", "position"):
            if not np.array_equal(np.isfinite(np.asarray(g[key])), np.isfinite(np.asarray(o[key]))): bad = bad or key      # (an invalid value on one side only)
            d = np.nanmax(np.abs(np.nan_to_num(np.asarray(g[key]), nan=0.0, posinf=0.0, neginf=0.0) - np.nan_to_num(np.asarray(o[key]), nan=0.0, posinf=0.0, neginf=0.0)))
            if d > 1e-9 * (scale if key != "position" else 1.0) + 1e-13: bad = bad or key
            elif key != "position" and scale > 1e-6: worst = max(worst, d / scale)     # (relative to the sample's scale, for samples that carry light)
        if gl.shape == ol.shape and len(ol):
            if not np.array_equal(np.isfinite(gl), np.isfinite(ol)): bad = bad or "light splats"
            gl, ol = np.nan_to_num(gl, nan=0.0, posinf=0.0, neginf=0.0), np.nan_to_num(ol, nan=0.0, posinf=0.0, neginf=0.0)
        if gl.shape != ol.shape or (len(ol) and (not np.array_equal(gl[:, 2], ol[:, 2]) or np.abs(gl[:, :2] - ol[:, :2]).max() > 1e-9 or
                                                 np.abs(gl[:, 3:] - ol[:, 3:]).max() > 1e-9 * max(np.abs(ol[:, 3:]).max(), 1e-300) + 1e-13)):
            bad = bad or "light splats"
        if bad:
            # A manifold walk stops on a threshold (relative step below 1e-..., 20 iterations): a sample whose walk sits at that threshold takes one Newton step more
            # or less depending on the last bit of its input, and its value moves by the size of that step.  Ask the oracle how far ITS value moves under few-ulp
            # scalings of the geometry: a difference within 20x of that spread (same splats, same positions) is the sample's conditioning, not the implementation.
            ok = gl.shape == ol.shape and (not len(ol) or (np.array_equal(gl[:, 2], ol[:, 2]) and np.abs(gl[:, :2] - ol[:, :2]).max() <= 1e-9))
            if ok:
                sp_p, sp_g, sp_l = np.zeros(3), np.zeros_like(np.asarray(o["gradients"], float)), np.zeros_like(ol[:, 3:])
                for sc2 in perturbed(sc):
                    O2 = go.Scene(sc2); o2 = O2.gbdpt_sample(ocfg, px, py, s); O2.close()
                    l2 = np.asarray(o2["light"]).reshape(-1, 6)
                    if l2.shape != ol.shape: sp_l = sp_l + np.inf; continue
                    sp_p = np.maximum(sp_p, np.abs(o2["primal"] - o["primal"])); sp_g = np.maximum(sp_g, np.abs(np.asarray(o2["gradients"]) - np.asarray(o["gradients"])))
                    sp_l = np.maximum(sp_l, np.abs(l2[:, 3:] - ol[:, 3:]))
                ok = ((np.abs(np.asarray(g["primal"]) - o["primal"]) <= 20 * sp_p + 1e-9 * scale + 1e-13).all() and
                      (np.abs(np.asarray(g["gradients"]) - np.asarray(o["gradients"])) <= 20 * sp_g + 1e-9 * scale + 1e-13).all() and
                      (not len(ol) or (np.abs(gl[:, 3:] - ol[:, 3:]) <= 20 * sp_l + 1e-9 * max(np.abs(ol[:, 3:]).max(), 1e-300) + 1e-13).all()))
            if not ok:
                print("MISMATCH %s: seed %d px %d py %d s %d\n%r\n%r\n%r\n%r" % (bad, seed, px, py, s, g[bad] if bad in g else gl, o[bad] if bad in o else ol, g["gradients"], o["gradients"])); sys.exit(1)
            illcond += 1
        if (g["raysTraced"], g["shadowRaysTraced"]) != (o["raysTraced"], o["shadowRaysTraced"]):
            knife += 1                     # (outputs equal: a visibility ray inside a wall plane or a lobe at its cut-off, DESIGN.md "G-BDPT" parity)
        probes += 1
    F = B.Film(S)
    integ.renderBlock(S, F, cfg, (0, 0, W, H)); F.sync()
    blk, lgt = F.accum(); st = F.stats()
    F.close()
    oblk, olgt, orays = O.gbdpt_render(ocfg)
    for name, a, b in (("block", blk, oblk), ("light", lgt, olgt)):
        sc_ = max(np.abs(b).max(), 1e-300)
        if np.abs(a - b).max() > 1e-9 * sc_:
            # A connection between two vertices of one wall has a geometry term of 1e-27 or exactly 0 depending on the last bit of its in-plane
            # visibility ray; the reference's estimator then adds the offsets' (finite) terms or nothing at all -- a discontinuity of the
            # estimator itself.  Ask the oracle: if few-ulp scalings of the geometry move ITS film by as much at those pixels, the difference is
            # that knife edge, not the implementation.
            spread = np.zeros_like(b)
            for sc2 in perturbed(sc):
                O2 = go.Scene(sc2)
                pb, pl, _ = O2.gbdpt_render(ocfg)
                spread = np.maximum(spread, np.abs((pb if name == "block" else pl) - b))
                O2.close()
            if (np.abs(a - b) <= 20 * spread + 1e-9 * sc_).all():
                discont += 1
                continue
            print("MISMATCH film %s: seed %d, max |diff| %.3e of %.3e (oracle's own spread there %.3e)" % (name, seed, np.abs(a - b).max(), sc_, spread.ravel()[np.abs(a - b).argmax()])); sys.exit(1)
    if (st["raysTraced"], st["shadowRaysTraced"]) != (orays["raysTraced"], orays["shadowRaysTraced"]):
        knife += 1
    films += 1
    S.close(); O.close()
    if (seed - first) % 20 == 19:
        print("seed %d: %d probes, %d films, %.1f s" % (seed, probes, films, time.time() - t0), flush=True)
print("OK: seeds %d..%d: %d single samples (%d ill-conditioned: within 20x of the oracle's own spread), %d films; worst relative difference %.2e; %d ray-count knife edges (outputs equal), %d films with a pixel on the estimator's own discontinuity (an in-plane connection; within 20x of the oracle's spread under few-ulp scalings)" % (first, first + count - 1, probes, illcond, films, worst, knife, discont))
fuzz_summary.emit("gpu_gbdpt_fuzz", first, count, time.time() - t0, samples=probes, ill_conditioned_samples=illcond, films=films, worst_rel_diff=worst, knife_edge_ray_counts=knife, films_on_estimator_discontinuity=discont)
