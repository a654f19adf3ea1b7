"""bench.py -- BASELINE.json's metric on MI355X: `python bench.py --gpus N --steps K --warmup W`.

A step = one pass of the hot path over one batch of synthetic input at BASELINE configs[1]: the build-authored Cornell
box (the reference ships no scenes), G-PT 64 spp at 1280x720 in fp64, then develop + screened-Poisson L2D
reconstruction (gpt.cpp:1358-1480).  The scene (BVH, triangle records) is resident in HBM before the timed region;
random numbers are the counter-based streams both the HIP path and the oracle use.

  value            = shift-mapped Mray/s = (closest-hit + any-hit queries of base AND offset paths, the quantity the
                     reference counts in raysTraced + shadowRaysTraced, skdtree.cpp:46-47) / wall time of the step
                     (render + halo exchange + develop + gather + reconstruct), all ranks, max over ranks.
  poisson          = Poisson-CG Mpix-iter/s of the reconstruction inside the same steps (HIP-event span of solveIndirect).
  roofline         = the TIMED STEP's dominant kernels, the staged render (k_primary + k_first (k_render for scenes with glossy vertices) + k_walk + k_replay + k_continue + k_fold_cont, 98.8 % of a step):
                     not an HBM workload (the scene sits in LDS / L2), so its ceiling is VALU issue: wave-instructions of those
                     kernels per step (committed PMC pass of THIS binary, profiles/*_counters.json, keyed by a hash of csrc/) / their launch
                     duration by HIP events, measured live in this run, against 1024 SIMDs x 2.4 GHz / 4 cycles; `traffic` = their FETCH_SIZE x 2
                     + WRITE_SIZE per step.  Without a matching counter file (sources changed, another configuration) the block falls back
                     to the live byte figure below, so that it is never null.
  tracer_bytes     = SURVEY 8d-B's algorithmic bytes per ray (ray record + hit record + nodes visited x 128 | 64 B (the scene's node layout) + triangles tested x 80 B,
                     counted live on the device tree for 65 536 surface-born rays) x this run's rays/s against the 8 TB/s HBM peak, with
                     the survey's caveat: the traversal is latency- and divergence-bound and is served from LDS / L2, not HBM.
  roofline_hbm_case= the HBM-resident Poisson CG kernel against the 8 TB/s HBM peak (metric A's graded kernel): the fused
                     x_p + stencil kernel `kf_xp_Ax` at 3840x2160 (working set 1.2 GB, beyond the 256 MB Infinity Cache), algorithmic
                     bytes per launch (72 B/px) / launch duration by HIP events on the solver's stream, measured live in this run;
                     `traffic` = FETCH_SIZE x 2 + WRITE_SIZE per launch from the committed PMC pass.  A side measurement at a synthetic
                     size, NOT a kernel of the timed step (whose own solve is the persistent CG below).  frac <= 1 by construction.
  persistent_cg    = the cooperative CG kernel that runs the metric's own 1280x720 solve: NOT an HBM workload (the iterate lives in
                     VGPRs, counter traffic is 25x below the algorithmic bytes), so it is reported as latency-bound with its
                     per-iteration phase budget, never as an HBM fraction.
  tracer_issue     = issue-slot view of the render kernel (98.8 % of a step): VALU wave-instructions per second against
                     1024 SIMDs x 2.4 GHz / 4 cycles (the fp64 VALU issue rate), lane utilisation, scratch traffic (committed SQ / TCC counters).
  cpu_baseline     = the oracle (CPU restatement, kind "port": the reference itself cannot be built here) on a bounded sample.

N > 1 (strong scaling, the image is fixed): one process per GPU; rank r renders a contiguous strip of rows, exchanges
a one-pixel halo with its neighbours by RCCL send/recv, rank 0 gathers the four developed fp32 images and reconstructs.
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: without this RCCL's peer mappings fail (hipIpcGetMemHandle: invalid argument)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, SPP = 1280, 720, 64
MAX_DEPTH = -1           # gpt.cpp:1194 default (unbounded; Russian roulette from depth 5)
PRESET = "L2D"           # configs[1]: "L2 CG reconstruct"
SCENE = "cornell"
# BASELINE.json configs (1-based); the default run is configs[1] = --config 2.  The others are for the record, not bench lines.
CONFIGS = {1: ("cornell", 512, 512, 64, "L2D"), 2: ("cornell", 1280, 720, 64, "L2D"), 3: ("atrium", 1920, 1080, 256, "L1D"), 4: ("atrium", 3840, 2160, 256, "L2D"),
           5: ("veach", 1280, 720, 128, "L2D+L1D")}          # config 5: G-BDPT (bench_gbdpt below)
BYTES_PER_PIX_ITER = {"L2D": 120.0, "L1D": 132.0}     # SURVEY.md 8(d), fp32, reference 3-op formulation
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8 TB/s
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4.0     # wave-instructions/s: 256 CUs x 4 SIMDs, one fp64 VALU wave-instruction per 4 cycles (MI355X_MICROARCH.md: 78.6 TFLOP/s fp64 vector)
HBM_W, HBM_H = 3840, 2160                # the HBM-resident Poisson size (BASELINE configs[3]'s reconstruction)


def _source_hash():
    """Hash of the device sources: PMC counters cannot be read inside a plain bench run, so they come from a committed rocprofv3
    pass (tools/prof_r02.sh -> tools/profile_json.py -> profiles/*_counters.json) and are only quoted while csrc/ is what was profiled."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "gradientdomain-mitsuba_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    try:                                                                       # ... and of the flags they are compiled with (per-unit flags change kernels too)
        import importlib
        h.update(importlib.import_module("gradientdomain_mitsuba_amd._build")._flags_line().encode())
    except Exception:
        pass
    return h.hexdigest()[:16]


def _profiled_counters(suffix="_counters.json"):
    """-> (dict kernel-name-prefix -> counters, file name) of the newest committed counter file whose source hash matches, else ({}, None)."""
    import glob
    sh = _source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("source_hash") == sh:
            return d.get("kernels", {}), os.path.relpath(f, ROOT)
    return {}, None


def _kernel_counters(kernels, prefix, pick=None, **match):
    """Counters of the kernel whose name starts with `prefix`; the same kernel runs at several sizes in one bench run (the 1280x720
    solve and the 3840x2160 one), told apart by grid size: pick = "max_grid" | "min_grid", or exact fields in `match`."""
    c = [v for name, v in kernels.items() if name.startswith(prefix) and all(v.get(k) == w for k, w in match.items())]
    if not c:
        return None
    if pick == "max_grid":
        return max(c, key=lambda v: v.get("grid_x", 0))
    if pick == "min_grid":
        return min(c, key=lambda v: v.get("grid_x", 0))
    return c[0]


def _valu_busy(tot, profiled_s):
    """The fraction of the kernels' cycles in which a SIMD's VALU was executing: SQ_ACTIVE_INST_VALU (quad-cycles a wave spends in VALU instructions,
    MI355X_MICROARCH.md) x 4 / (1024 SIMDs x elapsed cycles), the gfx94x VALUBusy formula rocprofv3 falls back to on gfx950.  Elapsed cycles = GRBM_GUI_ACTIVE of
    the same PMC pass (the chip's real clock under the profiler, not a nominal 2.4 GHz); rocprofv3 reports it summed over the 8 XCDs -- told apart from a
    per-device figure by comparing it with the kernels' duration.  Unlike `frac` (instructions against an all-fp64 issue rate: an upper bound, fp32 and
    integer instructions issue faster) this is a measured busy fraction.  {} when the counter file has no GRBM_GUI_ACTIVE."""
    cyc = tot.get("GRBM_GUI_ACTIVE", 0.0)
    if not cyc or not tot.get("SQ_ACTIVE_INST_VALU") or profiled_s <= 0:
        return {}
    xcds = 8.0 if cyc / (profiled_s * 2.4e9) > 3.0 else 1.0
    busy = tot["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc / xcds)
    return {"valu_busy": round(busy, 4), "effective_clock_ghz_profiled": round(cyc / xcds / profiled_s / 1e9, 3),
            "valu_busy_what": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / %d XCDs) of the committed PMC pass: the measured share of SIMD cycles spent executing VALU instructions" % int(xcds)}


def _traffic_bytes(c):
    """FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both KiB per launch -> bytes."""
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0


def poisson_hbm_roofline(P, dev, preset="L2D"):
    """The HBM-resident case of the Poisson CG: 3840x2160 (x, r, p, Ap, b, e, w2 = 1.2 GB), multi-kernel graphs.  Inputs are made on the
    device (smooth image + noise; the arithmetic does not depend on the values).  -> dict for the bench line."""
    import torch
    w, h = HBM_W, HBM_H
    g = torch.Generator(device=dev); g.manual_seed(12345)
    yy, xx = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    gt = torch.stack([0.5 + 0.4 * torch.sin(0.2 * xx + c) * torch.cos(0.15 * yy) for c in range(3)], dim=-1).contiguous()
    tp = (gt + 0.2 * (torch.rand(gt.shape, device=dev, generator=g) - 0.5)).contiguous()
    dx = torch.zeros_like(gt); dx[:, :-1] = gt[:, 1:] - gt[:, :-1] + 0.01 * (torch.rand((h, w - 1, 3), device=dev, generator=g) - 0.5)
    dy = torch.zeros_like(gt); dy[:-1] = gt[1:] - gt[:-1] + 0.01 * (torch.rand((h - 1, w, 3), device=dev, generator=g) - 0.5)
    direct = torch.zeros_like(gt)
    torch.cuda.synchronize()
    prm = P.Params(preset, 0.2)
    sv = P.Solver(prm)
    sv.importImagesMTS(dx, dy, tp, direct, w, h); sv.setupBackend()
    sv.solveIndirect()                                  # warm-up
    t = []
    for _ in range(3):
        sv.setupBackend(); sv.solveIndirect(); t.append(sv.lastSolveSeconds)
    kus = sv.profileKernels(30)
    pus = sv.profilePersistent(2)
    stream_us = sv.profileStream(10)                    # the yardstick: kf_xp_Ax's access mix and nothing else, here and now
    sv.close()
    solve_s = sorted(t)[1]
    iters = prm.irlsIterMax * prm.cgIterMax
    return dict(w=w, h=h, preset=preset, solve_ms=1e3 * solve_s, mpix_iter_s=w * h * iters / solve_s / 1e6, kus=kus, persistent_us=pus, stream_us=stream_us)



def tracer_bytes_block(scene, desc, rays_per_s, closest_frac):
    """algorithmic bytes per ray of the traversal, counted live on the device BVH (gdpt_scene_trace_stats) for 65 536 rays born on
    the scene's surfaces (area-weighted point, uniform direction: what a path's bounces produce), weighted by this run's closest-hit / any-hit mix."""
    import numpy as np
    v = np.asarray(desc.verts, np.float64).reshape(-1, 3, 3)
    rng = np.random.default_rng(1)
    n = 1 << 16
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
    tri = rng.choice(len(v), size=n, p=area / area.sum())
    u, w = rng.random(n), rng.random(n)
    f = u + w > 1; u[f], w[f] = 1 - u[f], 1 - w[f]
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = v[tri, 0] + u[:, None] * e1[tri] + w[:, None] * e2[tri] + 1e-6 * d
    st = scene.trace_stats(o, d)
    lay = scene.layout()
    NODE_B, TRI_B, RAY_B, HIT_B = float(lay["node_bytes"]), 80.0, 56.0, 28.0       # BvhNode / BvhNodeQ (four children; fp32 boxes when the scene is LDS-resident, 8-bit boxes in HBM), TriIsect (fp64 TriAccel), ray record (o, d, maxt in fp64), hit record (t, u, v, prim)
    b_closest = RAY_B + HIT_B + st["nodes_closest"] * NODE_B + st["tris_closest"] * TRI_B
    b_any = RAY_B + 4.0 + st["nodes_any"] * NODE_B + st["tris_any"] * TRI_B
    bpr = closest_frac * b_closest + (1.0 - closest_frac) * b_any
    ach = bpr * rays_per_s / 1e9
    lds = bool(lay["lds_resident"])
    # (a scene that sits in LDS moves none of these bytes over HBM: the figure is then bytes per ray and a rate, with no ceiling attached)
    return {"bound": "none (scene tables in LDS: not an HBM workload)" if lds else "hbm", "achieved": round(ach, 1), "peak": None if lds else HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None if lds else round(ach / HBM_PEAK_GBS, 4), "traffic": None,
            "bytes_per_ray": round(bpr, 1), "bytes_per_closest_ray": round(b_closest, 1), "bytes_per_shadow_ray": round(b_any, 1), "closest_hit_share_of_rays": round(closest_frac, 4),
            "nodes_visited": {"closest": round(st["nodes_closest"], 2), "any": round(st["nodes_any"], 2)}, "tris_tested": {"closest": round(st["tris_closest"], 2), "any": round(st["tris_any"], 2)},
            "node_bytes": lay["node_bytes"], "scene_lds_resident": lay["lds_resident"],
            "what": "SURVEY 8d-B: (ray record + hit record + nodes_visited x node_bytes + tris_tested x 80 B, counted live by gdpt_scene_trace_stats on 65 536 surface-born rays) x this run's rays/s / 8 TB/s",
            "caveat": "not an HBM-roofline workload: the tables are served from LDS (small scenes) or L2 / Infinity Cache, the traversal is latency- and divergence-bound; path-state traffic is not in the figure"}


def _usable_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container may show 256 CPUs
    and be allowed 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                     # cgroup v2
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())              # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return max(1, min(n, 256))


def cpu_baseline(W, H, spp):
    """Oracle on a bounded sample of the SAME workload, on the host cores of this box.
    Tracer: the reference renders blocks on one worker thread per core (sched.cpp:427-496); here one spawned process per core
    renders a 2-row band of the 1280x720x64spp frame (bands spread evenly over the image), rate = all rays / the slowest
    worker's time; if the pool cannot be used, one 8-row band in this process (cores = 1).
    Solver: full L2D solves on ONE core -- the reference's OpenMP backend is single-threaded off Windows
    (BackendOpenMP.cpp:76-79)."""
    import concurrent.futures as cf
    import multiprocessing as mp
    from oracle import cpu_band, gpt_oracle as go, poisson_oracle as po
    go.build()                                           # compile once, before the workers race for it
    cores = _usable_cores()
    rows = max(2, min(8, int(round(96.0 / cores)) // 2 * 2))          # ~10-30 s of CPU work in all
    cores = max(1, min(cores, H // rows))
    bands = [(W, H, spp, MAX_DEPTH, y, y + rows) for y in [int((k + 0.5) * H / cores) // 2 * 2 for k in range(cores)] if y + rows <= H]
    t0 = time.perf_counter()
    res = None
    if cores > 1:
        try:
            with cf.ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as ex:
                res = list(ex.map(cpu_band.render_band, bands, timeout=180))
        except Exception as e:                           # broken pool, timeout: fall back to one core, say so
            sys.stderr.write("cpu_baseline: multi-process run failed (%s); single-core sample instead\n" % e)
            res = None
    if res is None:
        bands = [(W, H, spp, MAX_DEPTH, H // 2 - 4, H // 2 + 4)]
        res = [cpu_band.render_band(bands[0])]
    wall = time.perf_counter() - t0
    rays, slowest = sum(r for r, _ in res), max(s for _, s in res)
    one = res[len(res) // 2]
    dx, dy, tp, direct = po.synth_inputs(W, H)
    t1 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        po.solve(po.preset(PRESET), dx, dy, tp, direct, W, H)
    dp = time.perf_counter() - t1
    # the same solve on all usable cores (the reference's BackendOpenMP parallelises these loops; off Windows it runs one thread)
    os.environ["OMP_NUM_THREADS"] = str(_usable_cores())
    po.solve_allcores(po.preset(PRESET), dx, dy, tp, direct, W, H)
    t2 = time.perf_counter()
    for _ in range(reps):
        po.solve_allcores(po.preset(PRESET), dx, dy, tp, direct, W, H)
    dpa = time.perf_counter() - t2
    return {"value": round(rays / slowest / 1e6, 3), "unit": "Mray/s", "cores": len(bands), "kind": "port",
            "sample": "%d band(s) of %d rows of the %dx%dx%dspp Cornell render, one process per core (%d rays, slowest worker %.1f s, %.1f s with process start-up) + %d x %s solve on 1 core (%.1f s)" % (
                len(bands), bands[0][5] - bands[0][4], W, H, spp, rays, slowest, wall, reps, PRESET, dp),
            "value_1core": round(one[0] / one[1] / 1e6, 3),
            "poisson_mpix_iter_s": round(W * H * 50 * reps / dp / 1e6, 2), "poisson_cores": 1,
            "poisson_allcores_mpix_iter_s": round(W * H * 50 * reps / dpa / 1e6, 2), "poisson_allcores": _usable_cores()}


def bench_gbdpt(a, rank, local, world, dev):
    """BASELINE configs[4]: Veach-bidir-class scene, G-BDPT 128 spp, 1280x720 (for the record, like configs 1, 3, 4; the metric's own
    configuration is --config 2).  A step = GBDPTIntegrator::render: the bidirectional sampler over every pixel (strips of rows per rank),
    the reduction of the ranks' films onto rank 0, develop, prepareDataForSolver, the L2D and the L1D reconstruction."""
    import torch
    import torch.distributed as dist
    from gradientdomain_mitsuba_amd import gbdpt, gpt, parallel, scenes
    # the scene class the configuration names: a glass egg, a mirror and polished copper in the indirectly lit room (round 4: paths with specular
    # chains run the general form of the sampler, manifold walks and all); --bd-connectable renders rounds 1-3's stand-in (every BSDF connectable)
    desc = scenes.veach_bidir(W, H, specular=not a.bd_connectable)
    scene = gpt.Scene(desc, device=local)
    integ = gbdpt.GBDPTIntegrator(maxDepth=MAX_DEPTH)
    sr = parallel.GBDPTStripRenderer(scene, integ, rank, world, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        sr.render(a.spp)
    barrier()
    t0 = time.perf_counter()
    rays = samples = closest = general = 0
    render_ms = 0.0
    solve = [0.0, 0.0]
    phases = {}
    last_out = None
    for _ in range(a.steps):
        last_out = sr.render(a.spp)
        general += sr.last["chain"]["generalSamples"]
        assert sr.last["chain"]["overflows"] == 0
        rays += sr.last["rays"]; samples += sr.last["samples"]; render_ms += sr.last["render_ms"]; closest += sr.last["closest_rays"]
        solve[0] += sr.last["solve_s"][0]; solve[1] += sr.last["solve_s"][1]
        for k, v in sr.last["phases_ms"].items():
            phases[k] = phases.get(k, 0.0) + v
    barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall, render_ms, float(rays), float(samples)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = t.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = t.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        wall, render_ms, rays, samples = float(mx[0]), float(mx[1]), float(sm[2]), float(sm[3])
    if rank == 0:
        npx = W * H
        # the ceiling figures of this line: SURVEY 8d-B's bytes per ray (live), and -- when a PMC pass of this binary's G-BDPT kernels is
        # committed (profiles/*_counters_gbdpt.json, tools/prof_gbdpt.sh) -- the issue-slot view of the sampler: its wave-instructions PER SAMPLE
        # from that pass x this run's samples/s against the fp64 VALU issue peak, with lane utilisation and waiting share per kernel
        launch_s = render_ms * 1e-3 / a.steps
        tracer_bytes = tracer_bytes_block(scene, desc, rays / launch_s / a.steps, closest / float(max(1, sr.last["rays"] * a.steps)) if world == 1 else 0.5)
        counters, counters_file = _profiled_counters("_counters_gbdpt.json")
        issue = None
        bd = {k: v for k, v in counters.items() if "gdpt_bdk::k_bd" in k and "SQ_INSTS_VALU" in v}         # k_bd_* (every vertex connectable) and k_bdg_* (the general form)
        put = next((v for k, v in bd.items() if "k_bd_put" in k), None)
        if bd and put and put.get("grid_x") and world == 1:
            # k_bd_put runs once per chunk with one thread per sample: its launches x grid = the samples of the profiled run
            prof_samples = sum(v.get("calls", 0) * v.get("grid_x", 0) for k, v in bd.items() if "k_bd_put" in k)
            tot = {f: sum(v.get("calls", 0) * v.get(f, 0.0) for v in bd.values()) for f in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE")}
            busy = _valu_busy(tot, sum(v.get("calls", 0) * v.get("avg_us", 0.0) for v in bd.values()) * 1e-6)
            per_sample = tot["SQ_INSTS_VALU"] / prof_samples
            ach = per_sample * samples / (render_ms * 1e-3)
            issue = {"bound": "valu-issue", "achieved": round(ach / 1e9, 1), "peak": round(VALU_ISSUE_PEAK / 1e9, 1), "unit": "G wave-instr/s", "frac": round(ach / VALU_ISSUE_PEAK, 4),
                     "traffic": round((2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / prof_samples * samples / a.steps),
                     "traffic_what": "FETCH_SIZE x 2 + WRITE_SIZE of the sampler's kernels per sample of the profiled run x this run's samples per step, bytes",
                     "kernel": "k_bd_paths + k_bd_shift + k_bd_connect<*> + k_bdg_shift + k_bdg_offset + k_bdg_connect<*> + k_bdg_light<*> + k_bd_put", "launch_ms_live": round(1e3 * launch_s, 3),
                     "valu_wave_instr_per_sample": round(per_sample, 1), "profiled_samples": prof_samples,
                     "lane_utilisation": round(tot["SQ_THREAD_CYCLES_VALU"] / (tot["SQ_ACTIVE_INST_VALU"] * 64.0), 4) if tot["SQ_ACTIVE_INST_VALU"] else None,
                     "wait_any_frac_of_wave_cycles": round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 4) if tot["SQ_WAVE_CYCLES"] else None,
                     "per_kernel": {k.split("gdpt_bdk::")[1].split("@")[0]: {"calls": v.get("calls"), "avg_us": v.get("avg_us"), "vgpr": v.get("vgpr"),
                                                                           "lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0), 3) if v.get("SQ_ACTIVE_INST_VALU") else None,
                                                                           "wait_any_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3) if v.get("SQ_WAVE_CYCLES") else None}
                                    for k, v in bd.items()},
                     "counters_file": counters_file, **busy,
                     "what": "the sampler's kernels: VALU wave-instructions per sample (committed PMC pass of this binary at the same scene and maxDepth) x this run's samples/s, against 1024 SIMDs x 2.4 GHz / 4 cycles"}
        roofline = issue if issue else dict(tracer_bytes, kernel="k_bd_paths + k_bd_shift + k_bd_connect<*> (traversal part)", counters_file=None,
                                            note="no committed PMC pass of the G-BDPT kernels matches this binary: the live byte figure of SURVEY 8d-B stands in for the issue-slot view")
        out = {"metric": "shift-mapped Mray/s + Poisson-CG Mpix-iter/s, %dx%dx%dspp (G-BDPT)" % (W, H, a.spp),
               "value": round(rays / wall / 1e6, 1), "unit": "Mray/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * wall / a.steps, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "Veach-bidir-class room (build-authored, %d triangles, %s), G-BDPT %d spp, %dx%d, fp64 sampler, L2D + L1D reconstruct (BASELINE configs[4])" % (
                              desc.ntri, "all BSDFs connectable: rounds 1-3's stand-in" if a.bd_connectable else "glass egg + mirror + polished copper: specular chains, manifold walks", a.spp, W, H),
                          "general_form_sample_share": round(general / max(samples, 1.0), 4) if world == 1 else None,
                          "maxDepth": 12, "rrDepth": 5, "lightImage": True, "parallelism": "row strips of camera samples x%d + one film reduction onto rank 0" % world, "strip_rows": [s1 - s0 for (s0, s1) in sr.strips]},
               "rays_per_step": round(rays / a.steps), "rays_per_sample": round(rays / max(samples, 1.0), 2), "msample_s": round(samples / wall / 1e6, 3),
               "render_kernel_ms_per_step": round(render_ms / a.steps, 3), "phases_ms_per_step": {k: round(v / a.steps, 3) for k, v in phases.items()},
               "reduce_bytes_per_rank": sr.last["reduce_bytes"],
               "poisson": {"L2D": {"solve_ms_per_step": round(1e3 * solve[0] / a.steps, 4), "mpix_iter_s": round(npx * 50 * a.steps / solve[0] / 1e6, 1) if solve[0] > 0 else None},
                           "L1D": {"solve_ms_per_step": round(1e3 * solve[1] / a.steps, 4), "mpix_iter_s": round(npx * 1000 * a.steps / solve[1] / 1e6, 1) if solve[1] > 0 else None}, "dtype": "f32"},
               "roofline": roofline, "tracer_bytes": tracer_bytes,
               "roofline_note": "the G-BDPT sampler runs as walk / connect / put launches joined by 11 KB sample records in HBM (DESIGN.md, G-BDPT); the connection kernels (70 % of a frame; one build per item class and phase -- a ray-free filter, the base path, the offsets, each on the survivors of the one before --, MIS weights as recurrences over the record) are latency-bound at 2 waves/SIMD on dependent record and scene-table loads plus fp64 BSDF evaluations -- counter traffic ~0.9 TB/s, a ninth of the HBM peak: no HBM or MFMA fraction applies; the reconstructions are the persistent CG of --config 2"}
        print(json.dumps(out))
        if a.dump:                               # (tests: the last step's developed sampler buffers and both reconstructions)
            import numpy as np
            np.savez(a.dump, strips=np.array(sr.strips), **{k.lstrip("-"): v.cpu().numpy() for k, v in last_out.items()})
    sr.close(); scene.close()
    if world > 1:
        dist.destroy_process_group()


XGMI_LINK_GBS = 64.0     # modelled payload rate of ONE xGMI link in one direction (76.8 GB/s raw per direction, MI355X_MICROARCH.md; ~83 % as payload): what a strip's halo / gather message moves at


def strip_study(a):
    """--strip-study: single-GPU evidence for the N-GPU scaling claim (no 8-GPU node is available to the builder; the driver measures the real curve when one is).
    For N in {2, 4, 8} every rank's strip of the frame is rendered ALONE on this one device -- the partition parallel.row_strips gives, then the partition
    parallel.rebalance_strips makes of the measured times, as `bench.py --gpus N` does after its warm-up -- and timed by the film's HIP events.  The step of an
    N-GPU run is then MODELLED as  max_r(strip render) + halo (pack + unpack measured here, the messages at a stated xGMI rate) + develop of the slowest strip's
    size (measured) + gather of (N - 1) strips onto rank 0 over N - 1 links at once (modelled: the largest strip's four fp32 images at the stated rate) + the
    full-frame reconstruction on rank 0 (measured); predicted_speedup[N] = measured 1-GPU step / that.  What the model leaves out: RCCL's per-message set-up
    (tens of microseconds), the ranks' skew at the barrier, PCIe / host jitter -- all small beside a 100 ms render; what it includes is everything that grows in
    share as a strip shrinks: sample slices, the refill tail of k_continue on a short queue, the fixed cost of develop / solve."""
    import torch
    from gradientdomain_mitsuba_amd import gpt, parallel, scenes
    import gradientdomain_mitsuba_amd.poisson as P
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = {"what": "strip study (bench.py --strip-study): each rank's strip rendered alone on ONE device; N-GPU step modelled from measured parts", "xgmi_link_GBs_modelled": XGMI_LINK_GBS, "configs": {}}
    for config in a.study_configs:
        scene_name, w, h, spp_cfg, preset = CONFIGS[config]
        spp = a.spp if a.spp != SPP else (spp_cfg if config == 2 else min(spp_cfg, 64))
        desc = scenes.cornell_box(w, h, "diffuse") if scene_name == "cornell" else scenes.atrium(w, h, segments=a.atrium_segments)
        scene = gpt.Scene(desc, device=0)
        integ = gpt.GradientPathIntegrator(maxDepth=MAX_DEPTH, reconstructL1=(preset == "L1D"), reconstructL2=(preset != "L1D"))
        cfg = integ.config(spp)

        def render_strip(y0, y1, reps=2):
            """-> (best render ms by HIP events, rays, develop ms, halo pack+unpack ms, halo bytes) of rows [y0, y1) rendered into a strip film of their own"""
            film = gpt.Film(scene, y0, y1)
            best = 1e30
            for _ in range(reps):
                film.clear(); integ.renderBlock(scene, film, cfg, (0, y0, w, y1)); film.sync()
                best = min(best, film.render_ms())
            st = film.stats()
            imgs = torch.empty((4, y1 - y0, w, 3), dtype=torch.float32, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i, b in enumerate((1, 2, 3, 4)):
                film.develop_device(b, imgs[i])
            film.sync(); torch.cuda.synchronize(); dev_ms = 1e3 * (time.perf_counter() - t0)
            n = film.halo_bytes() // 8
            buf = torch.empty(n, dtype=torch.float64, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            film.pack_halo(1 if y1 < h else 0, buf); film.sync()
            film.unpack_halo(1 if y1 < h else 0, buf); film.sync()
            torch.cuda.synchronize(); halo_ms = 1e3 * (time.perf_counter() - t0)
            hb = film.halo_bytes()
            film.close()
            return best, st["raysTraced"] + st["shadowRaysTraced"], dev_ms, halo_ms, hb
        # the 1-GPU step: whole frame + reconstruction (what `bench.py` times at N = 1)
        full_ms, full_rays, full_dev_ms, _, _ = render_strip(0, h)
        sr = parallel.StripRenderer(scene, integ, 0, 1, dev)
        sr.render(spp); sr.render(spp)
        solve_ms = 1e3 * sr.last["solve_s"]
        t0 = time.perf_counter(); sr.render(spp); torch.cuda.synchronize(); step1_ms = 1e3 * (time.perf_counter() - t0)
        sr.close()
        rows = {"workload": "%s %dx%d, %d spp%s, %s" % (scene_name, w, h, spp, "" if spp == spp_cfg else " (configuration: %d; a strip's time is linear in spp above one chunk)" % spp_cfg, preset),
                "one_gpu": {"render_ms": round(full_ms, 2), "develop_ms": round(full_dev_ms, 3), "solve_ms": round(solve_ms, 3), "step_ms_measured": round(step1_ms, 2), "rays": full_rays}, "N": {}}
        for N in (2, 4, 8):
            entry = {}
            strips = parallel.row_strips(h, N)
            for label in ("equal_rows", "rebalanced"):
                res = [render_strip(y0, y1) for (y0, y1) in strips]
                ms = [r[0] for r in res]
                rays = sum(r[1] for r in res)
                slow = max(range(N), key=lambda r: ms[r])
                halo_msg_ms = 1e3 * res[slow][4] / (XGMI_LINK_GBS * 1e9)                          # both neighbours' messages travel on different links at once
                gather_ms = 1e3 * max((y1 - y0) for (y0, y1) in strips[1:]) * w * 3 * 4 * 4 / (XGMI_LINK_GBS * 1e9)   # N - 1 senders, one link each, all into rank 0
                fixed = res[slow][3] + halo_msg_ms + max(r[2] for r in res) + gather_ms + solve_ms
                step = max(ms) + fixed
                entry[label] = {"strip_rows": [y1 - y0 for (y0, y1) in strips], "strip_render_ms": [round(v, 2) for v in ms], "max_over_mean": round(max(ms) / (sum(ms) / N), 4),
                                "sum_of_strips_over_full_frame": round(sum(ms) / full_ms, 4), "rays_sum_equals_frame": rays == full_rays,
                                "fixed_ms": {"halo_pack_unpack": round(res[slow][3], 3), "halo_messages_modelled": round(halo_msg_ms, 4), "develop": round(max(r[2] for r in res), 3),
                                             "gather_modelled": round(gather_ms, 3), "solve_on_rank0": round(solve_ms, 3)},
                                "step_ms_modelled": round(step, 2), "predicted_speedup": round(step1_ms / step, 3), "render_only_speedup": round(full_ms / max(ms), 3)}
                if label == "equal_rows":
                    # rank 0 also reconstructs while the others would already render the next frame: its strip is charged with the solve, as StripRenderer.rebalance does
                    strips = parallel.rebalance_strips(strips, [ms[0] + solve_ms] + ms[1:], min_rows=2)
            rows["N"][str(N)] = entry
        out["configs"][str(config)] = rows
        scene.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json configs index (1-5); 2 is the metric's configuration, 5 = G-BDPT")
    ap.add_argument("--bd-connectable", action="store_true", help="--config 5: the connectable-only stand-in scene of rounds 1-3 instead of the specular one")
    ap.add_argument("--no-rebalance", action="store_true", help="N > 1: keep equal-height strips instead of rebalancing them after the warm-up pass")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the default) | gloo (functional runs of the N>1 path on one GPU)")
    ap.add_argument("--dump", default=None, help="rank 0 writes the last step's reconstruction and the four gathered solver images to this .npz (tests)")
    ap.add_argument("--atrium-segments", type=int, default=48, help="configs 3 / 4: facets per column ring of the atrium (48: 112 908 triangles, the default; 112: 260 364, SURVEY 8d's Sponza size)")
    ap.add_argument("--strip-study", action="store_true", help="one device: every rank's strip of the frame for N = 2, 4, 8 rendered alone; prints the modelled N-GPU step and predicted speedup (profiles/r06*_strip_study.json)")
    ap.add_argument("--study-configs", type=int, nargs="+", default=[2, 4], help="--strip-study: the BASELINE configs to study")
    a = ap.parse_args()
    if a.strip_study:
        return strip_study(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: launch the N ranks here (one process per GPU, rendezvous on 127.0.0.1), exactly as
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` would; rank 0's JSON line is the output.
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s; the launcher's world size is what runs\n" % (a.gpus, os.environ["WORLD_SIZE"]))
    global W, H, PRESET, SCENE
    SCENE, W, H, spp_cfg, PRESET = CONFIGS[a.config]
    if a.spp == SPP:
        a.spp = spp_cfg

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    if a.config == 5:
        return bench_gbdpt(a, rank, local, world, dev)
    from gradientdomain_mitsuba_amd import gpt, parallel, scenes
    import gradientdomain_mitsuba_amd.poisson as P

    # (--atrium-segments 112: the Sponza-class stand-in at SURVEY 8(d)'s ~262 k triangles -- 260 364 -- instead of the default 48 segments' 112 908)
    desc = scenes.cornell_box(W, H, "diffuse") if SCENE == "cornell" else scenes.atrium(W, H, segments=a.atrium_segments)
    scene = gpt.Scene(desc, device=local)
    integ = gpt.GradientPathIntegrator(maxDepth=MAX_DEPTH, reconstructL1=(PRESET == "L1D"), reconstructL2=(PRESET != "L1D"))
    prm = P.Params(PRESET, integ.reconstructAlpha)
    iters = prm.irlsIterMax * prm.cgIterMax
    # the product's multi-GPU render: strips, halo exchange, gather, reconstruction on rank 0 (parallel.StripRenderer)
    sr = parallel.StripRenderer(scene, integ, rank, world, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    last_out = [None]

    def step():
        """-> (rays of this rank, render kernel ms, solve seconds, halo bytes)"""
        last_out[0] = sr.render(a.spp)
        return sr.last["rays"], sr.last["render_ms"], sr.last["solve_s"], sr.last["halo_bytes"]

    for _ in range(a.warmup):
        step()
        if world > 1 and not a.no_rebalance:
            # the reference hands blocks to whichever worker is free; with one strip per GPU the analogue is to move the strip
            # boundaries by the render times of the warm-up pass (same total image; untimed)
            sr.rebalance(min_rows=2)
    barrier()
    t0 = time.perf_counter()
    rays = 0
    render_ms = solve_s = 0.0
    halo = 0
    phase_ms = {}
    for _ in range(a.steps):
        r, ms, ss, hb = step()
        rays += r; render_ms += ms; solve_s += ss; halo = hb
        for k, v in sr.last.get("phases_ms", {}).items():
            phase_ms[k] = phase_ms.get(k, 0.0) + v
    barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall, render_ms, float(rays), solve_s], dtype=torch.float64, device=dev)
    ranks_line = None
    if world > 1:
        # per rank: render-kernel ms and the host phases of a step -- who is the slowest, how uneven the strips are, how much of a step is not render
        mine = torch.tensor([render_ms / a.steps] + [phase_ms.get(k, 0.0) / a.steps for k in ("render", "halo", "develop", "gather", "reconstruct")], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per = [[float(v) for v in r.tolist()] for r in allr]
        rk = [p[0] for p in per]
        slow = max(range(world), key=lambda r: rk[r])
        ranks_line = {"render_kernel_ms": [round(v, 3) for v in rk], "slowest_rank": slow, "imbalance_max_over_mean": round(max(rk) / (sum(rk) / world), 4),
                      "phases_ms_by_rank": {k: [round(p[1 + i], 3) for p in per] for i, k in enumerate(("render", "halo", "develop", "gather", "reconstruct"))},
                      "what": "render_kernel_ms: HIP-event span of a rank's render kernels per step; phases: host wall time per step, each phase ends synchronised (gather on rank 0 includes waiting for the slowest strip)"}
        mx = t.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = t.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        wall, render_ms, solve_s, rays = float(mx[0]), float(mx[1]), float(mx[3]), float(sm[2])
        ranks_line["step_fraction_outside_render"] = round(1.0 - (render_ms / a.steps) / (1e3 * wall / a.steps), 4)
    else:
        rays = float(rays)

    if rank == 0 and a.dump:
        import numpy as np
        np.savez(a.dump, final=last_out[0].cpu().numpy(), images=sr.last["images"].cpu().numpy(), strips=np.array(sr.strips))
    if rank == 0:
        mray = rays / wall / 1e6
        mpix_iter = W * H * iters * a.steps / solve_s / 1e6
        solver, film, strips = sr.solver, sr.film, sr.strips
        kus = solver.profileKernels(50)
        pus = solver.profilePersistent(20)
        bpi = BYTES_PER_PIX_ITER[PRESET]
        solve_achieved = bpi * mpix_iter * 1e6 / 1e9
        counters, counters_file = _profiled_counters()
        st_ = film.stats()
        samples = W * H * a.spp * a.steps
        rays_per_launch = rays / a.steps / world
        launch_s = render_ms * 1e-3 / a.steps
        # --- the render kernel (98.8 % of a step): an issue-slot view, not an HBM one.  Its tables sit in LDS (Cornell) or L2 / Infinity
        # Cache; what limits it is instruction issue under divergence and the latency of its scratch traffic.  Counters per launch come from
        # the committed PMC passes of this binary (null if csrc/ changed since); the launch duration and ray count are this run's.
        tracer_kernels = ("gdpt_tr::k_primary", "gdpt_tr::k_first", "gdpt_tr::k_render", "gdpt_tr::k_walk", "gdpt_tr::k_replay", "gdpt_tr::k_continue", "gdpt_tr::k_fold_cont")
        res = _kernel_counters(counters, "gdpt_tr::k_resolve")          # once per step: the profile's step count
        per_step = {}
        # (the committed counters are those of the DEFAULT workload, config 2 at its own size and spp: another configuration's step has other
        #  instruction counts, and pricing them with this run's duration once printed a "frac" of 1.29 for config 1)
        profiled_workload = a.config == 2 and a.spp == SPP and (W, H) == (1280, 720)
        if res and res.get("calls") and world == 1 and profiled_workload:
            for name, c in counters.items():
                key = next((t for t in tracer_kernels if t in name), None)
                if key and "SQ_INSTS_VALU" in c:
                    n = c.get("calls", 0) / float(res["calls"])
                    acc = per_step.setdefault(key, {"launches_per_step": 0.0, "avg_ms": 0.0})
                    acc["launches_per_step"] += n
                    acc["avg_ms"] += n * c.get("avg_us", 0.0) * 1e-3
                    for f in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"):
                        if f in c:
                            acc[f] = acc.get(f, 0.0) + n * c[f]
        tracer_issue = {"bound": "valu-issue", "unit": "G wave-instr/s", "peak": round(VALU_ISSUE_PEAK / 1e9, 1),
                        "peak_what": "1024 SIMDs x 2.4 GHz / 4 cycles per fp64 VALU wave-instruction",
                        "kernels": "k_primary + k_first (k_render for scenes with glossy vertices) + k_walk + k_replay + k_continue + k_fold_cont (the staged render of one step)", "render_ms_per_step": round(1e3 * launch_s, 3),
                        "counters_file": counters_file if per_step else None}
        if per_step:
            tot = {f: sum(k.get(f, 0.0) for k in per_step.values()) for f in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE")}
            valu = tot["SQ_INSTS_VALU"]
            tracer_issue.update(_valu_busy(tot, sum(k["avg_ms"] for k in per_step.values()) * 1e-3))
            tracer_issue.update({"achieved": round(valu / launch_s / 1e9, 1), "frac": round(valu / launch_s / VALU_ISSUE_PEAK, 4),
                                 "valu_wave_instr_per_ray": round(valu / rays_per_launch, 2),
                                 # (the chunks of a render are pipelined over two streams: under rocprofv3's kernel trace the launches of a step add up to MORE than the step --
                                 #  their durations overlap; the live figure above is HIP events around the step's render on the film's stream, which the second stream is joined to)
                                 "profiled_kernel_ms_sum_per_step": round(sum(k["avg_ms"] for k in per_step.values()), 2),
                                 "profiled_kernel_ms_sum_what": "sum of the committed kernel trace's launch durations per step; more than render_ms_per_step where consecutive chunks' kernels overlap on the film's two streams",
                                 "lane_utilisation": round(tot["SQ_THREAD_CYCLES_VALU"] / (tot["SQ_ACTIVE_INST_VALU"] * 64.0), 4) if tot["SQ_ACTIVE_INST_VALU"] else None,
                                 "wait_any_frac_of_wave_cycles": round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 4) if tot["SQ_WAVE_CYCLES"] else None})
            tb = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
            tracer_issue.update({"fabric_traffic_gb_per_step": round(tb / 1e9, 1), "fabric_traffic_bytes_per_ray": round(tb / rays_per_launch, 1),
                                 "fabric_traffic_what": "FETCH_SIZE x2 + WRITE_SIZE of the step's render kernels: scratch (spilled path state) + the sample queue; the algorithmic HBM bytes are the film records, %.2f GB" % (31 * 8 * 2 * W * H / 1e9),
                                 "per_kernel": {k.split("::")[1]: {"launches_per_step": round(v["launches_per_step"], 2), "ms_per_step": round(v["avg_ms"], 2),
                                                                     "lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0), 3) if v.get("SQ_ACTIVE_INST_VALU") else None,
                                                                     "wait_any_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3) if v.get("SQ_WAVE_CYCLES") else None,
                                                                     "traffic_gb_per_step": round((2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 / 1e9, 1)}
                                                for k, v in per_step.items()}})
        else:
            tracer_issue.update({"achieved": None, "frac": None})
        closest_frac = st_["raysTraced"] / float(max(1, st_["raysTraced"] + st_["shadowRaysTraced"]))
        tracer_bytes = tracer_bytes_block(scene, desc, rays / a.steps / launch_s, closest_frac)
        # --- the persistent CG kernel that runs THIS configuration's solve: latency-bound, never an HBM fraction
        persistent = None
        if pus > 0.0:
            pc = _kernel_counters(counters, "void gdpt::kp_cg")
            persistent = {"bound": "latency", "kernel": "kp_cg", "kernel_avg_us": round(pus, 2), "iterations_per_launch": prm.cgIterMax,
                          "us_per_iteration": round(pus / prm.cgIterMax, 3),
                          "what": "one cooperative launch = cgIterMax CG iterations with the iterate in VGPRs; per iteration two grid-wide all-gathers (p.Ap, r.r) + the ring exchange bound it, not bandwidth",
                          "algorithmic_bytes_if_streamed": bpi * W * H * prm.cgIterMax,
                          "counter_traffic_bytes": _traffic_bytes(pc), "counters_file": counters_file if pc else None,
                          "phase_clocks_us": (pc or {}).get("phase_clocks_us")}
            if persistent["counter_traffic_bytes"]:
                persistent["counter_traffic_gbs"] = round(persistent["counter_traffic_bytes"] / (pus * 1e-6) / 1e9, 1)
        # --- the graded HBM roofline: the same CG on an HBM-resident image (3840x2160), measured live
        hb = poisson_hbm_roofline(P, dev, "L2D") if world == 1 else None
        hbm_case = None
        if hb is not None:
            npx = hb["w"] * hb["h"]
            kb = 72.0 * npx                                    # kf_xp_Ax<unit w>: R r, p, x + W x, p, Ap = 72 B/px (x_p 60 + stencil's p read / Ap write counted once)
            kavg = hb["kus"][3]
            ach = kb / (kavg * 1e-6) / 1e9
            hc = _kernel_counters(counters, "void gdpt::kf_xp_Ax", pick="max_grid")
            iter_bytes = 120.0 * npx
            iter_us = hb["kus"][3] + hb["kus"][1]
            hbm_case = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "scope": "side measurement in this process at a synthetic HBM-resident size (%dx%d), NOT a kernel of the timed step: the step's own kernels are reported under tracer_issue (issue-bound) and persistent_cg (latency-bound) and have no HBM fraction" % (hb["w"], hb["h"]),
                        "synthetic_size": True,
                        "traffic": _traffic_bytes(hc), "traffic_source": counters_file if hc else None,
                        "kernel": "kf_xp_Ax", "kernel_avg_us": round(kavg, 2), "kernel_bytes": kb,
                        "what": "fused x_p + 5-point stencil of the screened-Poisson CG at %dx%d L2D (HBM-resident: 1.2 GB working set), algorithmic 72 B/px per launch / HIP-event launch duration, measured in this run" % (hb["w"], hb["h"]),
                        "iteration": {"kernels_us": {"kf_xp_Ax": round(hb["kus"][3], 2), "kf_r_rz": round(hb["kus"][1], 2)}, "bytes": iter_bytes,
                                      "achieved": round(iter_bytes / (iter_us * 1e-6) / 1e9, 1), "frac": round(iter_bytes / (iter_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                      "what": "one CG iteration = kf_xp_Ax + kf_r_rz against SURVEY 8d's 120 B/pix-iter"},
                        "stream_yardstick": {"us": round(hb["stream_us"], 2), "tb_s": round(kb / (hb["stream_us"] * 1e-6) / 1e12, 3), "frac_of_peak": round(kb / (hb["stream_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                             "frac_of_it": round(hb["stream_us"] / kavg, 3),
                                             "source": "gdpt_poisson_profile_stream: measured in this run, on this solver's own vectors (best of 10 launches by HIP events)",
                                             "what": "a bare streaming kernel with this kernel's access mix (3 coalesced 16-byte reads + 3 non-temporal 16-byte writes per float4 element, no stencil, no reuse): what 72 B/px can be moved in at all on this device; frac_of_it = the yardstick's time / kf_xp_Ax's"},
                        "solve": {"ms": round(hb["solve_ms"], 3), "mpix_iter_s": round(hb["mpix_iter_s"], 1),
                                  "achieved": round(120.0 * hb["mpix_iter_s"] * 1e6 / 1e9, 1), "frac": round(120.0 * hb["mpix_iter_s"] * 1e6 / 1e9 / HBM_PEAK_GBS, 4)}}
        # --- `roofline`: the timed step's dominant kernels.  With the committed counters of this binary: the staged render against the VALU issue
        # peak (what bounds it); otherwise the live byte figure of SURVEY 8d-B, so that every configuration's line carries a recomputable fraction
        if tracer_issue.get("frac") is not None:
            roofline = {"bound": "valu-issue", "achieved": tracer_issue["achieved"], "peak": tracer_issue["peak"], "unit": tracer_issue["unit"], "frac": tracer_issue["frac"],
                        "traffic": round(tracer_issue["fabric_traffic_gb_per_step"] * 1e9), "traffic_what": "FETCH_SIZE x 2 + WRITE_SIZE of the step's render kernels, bytes per step (scratch + sample queue; algorithmic film bytes %.2f GB)" % (31 * 8 * 2 * W * H / 1e9),
                        "kernel": "k_primary + k_first (k_render for scenes with glossy vertices) + k_walk + k_replay + k_continue + k_fold_cont", "share_of_step": round(launch_s / (wall / a.steps), 4),
                        "launch_ms_live": round(1e3 * launch_s, 3), "valu_wave_instr_per_step": round(tracer_issue["achieved"] * 1e9 * launch_s),
                        "lane_utilisation": tracer_issue.get("lane_utilisation"), "wait_any_frac_of_wave_cycles": tracer_issue.get("wait_any_frac_of_wave_cycles"),
                        "valu_busy": tracer_issue.get("valu_busy"), "valu_busy_what": tracer_issue.get("valu_busy_what"), "effective_clock_ghz_profiled": tracer_issue.get("effective_clock_ghz_profiled"),
                        "frac_what": "an UPPER bound on issue use (every VALU instruction priced at the fp64 rate of 4 cycles; fp32 / integer ones issue in 2): valu_busy is the measured busy fraction",
                        "per_kernel": tracer_issue.get("per_kernel"), "counters_file": tracer_issue.get("counters_file"),
                        "profiled_kernel_ms_sum_per_step": tracer_issue.get("profiled_kernel_ms_sum_per_step"), "profiled_kernel_ms_sum_what": tracer_issue.get("profiled_kernel_ms_sum_what"),
                        "what": "the timed step's render kernels: VALU wave-instructions per step (committed PMC pass of this binary) / their launch duration by HIP events in THIS run, against 1024 SIMDs x 2.4 GHz / 4 cycles per fp64 wave-instruction; MFMA is not used (no dense contraction) and the scene is not HBM-resident, so neither the hbm nor the mfma ceiling applies to them"}
        else:
            roofline = dict(tracer_bytes)
            roofline["kernel"] = "k_primary + k_first (k_render for scenes with glossy vertices) + k_walk + k_replay + k_continue + k_fold_cont (traversal part)"
            roofline["counters_file"] = None
            roofline["note"] = "no committed PMC pass matches this binary / configuration: the live byte figure of SURVEY 8d-B stands in for the issue-slot view"
        out = {
            "metric": "shift-mapped Mray/s + Poisson-CG Mpix-iter/s, %dx%dx%dspp" % (W, H, a.spp),
            "value": round(mray, 1), "unit": "Mray/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * wall / a.steps, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s (build-authored, %d triangles), G-PT %d spp, %dx%d, fp64 tracer, %s reconstruct (BASELINE configs[%d])" % ("Cornell box" if SCENE == "cornell" else "atrium (Sponza-class stand-in)", desc.ntri, a.spp, W, H, PRESET, a.config - 1),
                       "maxDepth": MAX_DEPTH, "rrDepth": 5, "parallelism": "row strips x%d + 1-px halo" % world, "strip_rows": [s1 - s0 for (s0, s1) in strips]},
            "rays_per_step": round(rays / a.steps), "rays_per_sample": round(rays / samples, 2), "msample_s": round(samples / wall / 1e6, 2),
            "render_kernel_ms_per_step": round(render_ms / a.steps, 3), "render_kernel_mray_s": round(rays / world / (render_ms * 1e-3) / 1e6 * world, 1),
            "halo_bytes_per_rank": halo,
            "phases_ms_per_step": {k: round(v / a.steps, 4) for k, v in phase_ms.items()},
            "tracer_issue": tracer_issue,
            "poisson": {"value": round(mpix_iter, 1), "unit": "Mpix-iter/s", "preset": PRESET, "solve_ms_per_step": round(1e3 * solve_s / a.steps, 4), "dtype": "f32",
                        "path": "persistent cooperative CG" if pus > 0.0 else "multi-kernel graphs",
                        "kernels_us": {"kf_Ax": round(kus[0], 2), "kf_r_rz": round(kus[1], 2), "kf_x_p": round(kus[2], 2), "kf_xp_Ax": round(kus[3], 2), "kp_cg": round(pus, 2)},
                        "reference_cpu_mpix_iter_s": {"L2D": 65.8, "L1D": 80.8, "what": "BASELINE.md section 2: the reference's own solver, 1 thread, survey-stage shimmed build (supplementary)"}},
            "persistent_cg": persistent,
            "roofline": roofline,
            "tracer_bytes": tracer_bytes,
            "roofline_hbm_case": hbm_case,
        }
        if ranks_line:
            out["ranks"] = ranks_line
        if not a.no_cpu_baseline and a.config == 2 and world == 1:       # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(W, H, a.spp)
        print(json.dumps(out))
    sr.close(); scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
