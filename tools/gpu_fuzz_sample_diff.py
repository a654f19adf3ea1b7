"""One sample of a campaign seed (tools/gpu_fuzz_campaign.py's scene for SEED) through the device and the oracle, element by element, with the oracle's own spread
under few-ulp scalings of the geometry: python tools/gpu_fuzz_sample_diff.py SEED PX PY S   (GDPT_SCENE_IN_HBM=1 / GDPT_LIB=... select the build)"""
import sys, os, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, scenes
from oracle import gpt_oracle as go
seed, px, py, s = (int(a) for a in sys.argv[1:5])
rng = np.random.default_rng(seed)
W, H = int(rng.integers(17, 44)), int(rng.integers(9, 34))
kind = "random"
kw = dict(seed=seed, environment=(0.5, 0.7, 0.9) if seed % 3 == 0 else None)
if seed % 5 == 1:
    kind = "smooth" if seed % 2 else "bent"; kw = dict(environment=kw["environment"])
if seed % 5 == 2:
    kw["point_light"] = ((float(rng.uniform(100, 450)), float(rng.uniform(200, 500)), float(rng.uniform(100, 450))), (4e4, 3e4, 2e4), bool(seed % 2))
sc = scenes.atrium(W, H, columns=int(rng.integers(4, 12)), segments=int(rng.integers(6, 16))) if seed % 7 == 0 else scenes.cornell_box(W, H, kind, **kw)
if seed % 9 == 4:
    sc.thinlens = (float(rng.uniform(2.0, 60.0)), float(rng.uniform(300.0, 1500.0))) if seed % 7 else (float(rng.uniform(0.01, 0.3)), float(rng.uniform(2.0, 30.0)))
if seed % 11 == 3:
    sc.rfilter = scenes.RFILTER_DEFAULTS[1 + seed % 5]
md = int(rng.choice([-1, 2, 3, 5, 9])); rr = int(rng.choice([1, 3, 5])); strict = bool(rng.random() < 0.35); thr = float(rng.choice([0.001, 0.02, 0.0]))
spp = int(rng.integers(1, 7))
print("seed", seed, kind, W, H, "maxDepth", md, "rrDepth", rr, "strict", strict, "shiftThreshold", thr, "spp", spp, "materials", [(m.get("type"), m.get("alphaU"), m.get("alphaV"), m.get("distribution")) if isinstance(m, dict) else m for m in sc.materials])
S = G.Scene(sc); O = go.Scene(sc)
integ = G.GradientPathIntegrator(maxDepth=md, rrDepth=rr, strictNormals=strict, shiftThreshold=thr)
cfg = integ.config(spp); ocfg = go.config(maxDepth=md, rrDepth=rr, strictNormals=strict, spp=spp, shiftThreshold=thr)
g = S.evaluate_point(cfg, px, py, s); o = O.evaluate_point(ocfg, px, py, s)
v0 = np.asarray(sc.verts, np.float64).reshape(-1, 3)
outs = []
for ax in (None, 0, 1, 2):
    for k in (1, 2, 3, 4, 6, 8, -1, -2, -3, -4):
        v = v0.copy()
        if ax is None: v *= 1 + k * 2.0 ** -52
        else: v[:, ax] *= 1 + k * 2.0 ** -52
        sc2 = copy.deepcopy(sc); sc2.verts = v.reshape(np.asarray(sc.verts).shape)
        O2 = go.Scene(sc2); outs.append(O2.evaluate_point(ocfg, px, py, s)); O2.close()
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
routs = []
for sc2 in rotated(sc):
    O2 = go.Scene(sc2); routs.append(O2.evaluate_point(ocfg, px, py, s)); O2.close()
np.set_printoptions(precision=17, linewidth=200)
print("rays", g["raysTraced"], g["shadowRaysTraced"], o["raysTraced"], o["shadowRaysTraced"], sorted({(r["raysTraced"], r["shadowRaysTraced"]) for r in outs}))
for key in ("veryDirect", "throughput", "gradients", "neighbours"):
    sens = np.max([np.abs(r[key] - o[key]) for r in outs], axis=0)
    print(key, "\n device - oracle (relative):\n", (g[key] - o[key]) / np.abs(o[key]).clip(1e-300), "\n oracle's own spread (relative):\n", sens / np.abs(o[key]).clip(1e-300))
    if routs:
        rs = np.max([np.abs(r[key] - o[key]) for r in routs], axis=0)
        print(" oracle's spread under 18 rigid rotations of the scene by k x 2^-30 rad (relative):\n", rs / np.abs(o[key]).clip(1e-300))
