"""Worker of tests/test_opt_fence_gpu.py: python tests/fence_worker.py FIRST COUNT OUT.npz.  Renders one small film per fuzz seed through the
library the environment selects (GDPT_LIB: the -O1 build; unset: the -O3 product) and writes every film buffer and both ray counters.  The
scenes follow tools/gpu_fuzz_campaign.py's recipe: fuzzed Cornell boxes (random materials; a constant environment, a latitude-longitude map
or none; vertex normals, point lights, thin lenses), every seventh seed the atrium, strictNormals on a third of the seeds; every seed
through the HBM-scene builds (GDPT_SCENE_IN_HBM), every fourth also through the LDS-scene builds; the staged pipeline and the single kernel
built for 4 waves per SIMD (the one that keeps its Lane in scratch).  Every third seed also one G-BDPT film (tools/gpu_gbdpt_fuzz.py's recipe:
connectable Cornell boxes or the Veach-bidir stand-in, random maxDepth / rrDepth / lightImage) through the wavefront kernels: camera blocks,
light image, both ray counters."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from gradientdomain_mitsuba_amd import gpt as G, gbdpt as B, scenes

VARIANTS = (("staged", 2, 2), ("single4", 0, 4))


def scene_for(seed):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(17, 44)), int(rng.integers(9, 34))
    kind = "random"
    kw = dict(seed=seed, environment=(0.5, 0.7, 0.9) if seed % 3 == 0 else None)
    if seed % 5 == 1:
        kind = "smooth" if seed % 2 else "bent"
        kw = dict(environment=kw["environment"])
    if seed % 5 == 2:
        kw["point_light"] = ((float(rng.uniform(100, 450)), float(rng.uniform(200, 500)), float(rng.uniform(100, 450))), (4e4, 3e4, 2e4), bool(seed % 2))
    if seed % 7 == 0:
        sc = scenes.atrium(W, H, columns=int(rng.integers(4, 12)), segments=int(rng.integers(6, 16)))
    else:
        sc = scenes.cornell_box(W, H, kind, **kw)
        if seed % 3 == 1:
            sc.environment_map = dict(rgb=scenes.sky_map(16 + 8 * (seed % 4), 8 + 4 * (seed % 4), seed=seed), scale=float(rng.uniform(0.3, 1.5)), index=-1)
    if seed % 9 == 4 and getattr(sc, "environment_map", None) is None:      # (a thin lens with a bitmap environment is refused by design)
        sc.thinlens = (float(rng.uniform(2.0, 60.0)), float(rng.uniform(300.0, 1500.0))) if seed % 7 else (float(rng.uniform(0.01, 0.3)), float(rng.uniform(2.0, 30.0)))
    md = int(rng.choice([-1, 2, 4, 5, 9]))
    cfg = dict(maxDepth=md, rrDepth=int(rng.choice([1, 3, 5])), strictNormals=bool(rng.random() < 0.35), shiftThreshold=float(rng.choice([0.001, 0.02, 0.0])))
    return sc, cfg, int(rng.integers(1, 5))


def main(first, count, out):
    res = {}
    for seed in range(first, first + count):
        sc, kw, spp = scene_for(seed)
        for hbm in ((1, 0) if seed % 4 == 0 else (1,)):
            if hbm:
                os.environ["GDPT_SCENE_IN_HBM"] = "1"
            else:
                os.environ.pop("GDPT_SCENE_IN_HBM", None)
            S = G.Scene(sc)
            integ = G.GradientPathIntegrator(**kw)
            for name, pipeline, occ in VARIANTS:
                F = G.Film(S)
                F.set_pipeline(pipeline); F.set_occupancy(occ)
                integ.renderBlock(S, F, integ.config(spp), (0, 0, sc.width, sc.height))
                st = F.stats()
                key = "%d/%d/%s" % (seed, hbm, name)
                res[key + "/film"] = np.asarray(F.accum(), np.float64)
                res[key + "/rays"] = np.array([st["raysTraced"], st["shadowRaysTraced"]], np.int64)
                res[key + "/strict"] = np.array([int(kw["strictNormals"])])
                F.close()
            S.close()
        if seed % 3 == 0:
            os.environ.pop("GDPT_SCENE_IN_HBM", None)
            rng = np.random.default_rng(seed + 7)
            W, H = int(rng.integers(12, 36)), int(rng.integers(8, 28))
            sc = scenes.veach_bidir(W, H) if seed % 5 == 0 else scenes.cornell_box(W, H, "random_connectable", seed=seed)
            integ = B.GBDPTIntegrator(maxDepth=int(rng.choice([-1, 1, 2, 3, 5, 8, 12])), rrDepth=int(rng.choice([1, 3, 5])), lightImage=bool(rng.random() < 0.7))
            S = G.Scene(sc); F = B.Film(S)
            integ.renderBlock(S, F, integ.config(int(rng.integers(1, 4)), 5489 + seed), (0, 0, W, H)); F.sync()
            blk, lgt = F.accum(); st = F.stats()
            key = "%d/0/gbdpt" % seed
            res[key + "/film"] = np.concatenate([np.asarray(blk, np.float64).reshape(-1), np.asarray(lgt, np.float64).reshape(-1)])[None]
            res[key + "/rays"] = np.array([st["raysTraced"], st["shadowRaysTraced"]], np.int64)
            res[key + "/strict"] = np.array([0])
            F.close(); S.close()
    np.savez(out, **res)


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
