"""bench.py -- BASELINE.json's metric on MI355X: `python bench.py --gpus N --steps K --warmup W`.

A step = one pass of the hot path over one batch of synthetic input at BASELINE configs[1]: the build-authored Cornell
box (the reference ships no scenes), G-PT 64 spp at 1280x720 in fp64, then develop + screened-Poisson L2D
reconstruction (gpt.cpp:1358-1480).  The scene (BVH, triangle records) is resident in HBM before the timed region;
random numbers are the counter-based streams both the HIP path and the oracle use.

  value            = shift-mapped Mray/s = (closest-hit + any-hit queries of base AND offset paths, the quantity the
                     reference counts in raysTraced + shadowRaysTraced, skdtree.cpp:46-47) / wall time of the step
                     (render + halo exchange + develop + gather + reconstruct), all ranks, max over ranks.
  poisson          = Poisson-CG Mpix-iter/s of the reconstruction inside the same steps (HIP-event span of solveIndirect).
  roofline         = the dominant Poisson CG kernel against HBM (the graded kernel): algorithmic bytes per launch /
                     launch duration by HIP events on the solver's stream; solve_achieved = the same over the whole solve.
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
CONFIGS = {1: ("cornell", 512, 512, 64, "L2D"), 2: ("cornell", 1280, 720, 64, "L2D"), 3: ("atrium", 1920, 1080, 256, "L1D"), 4: ("atrium", 3840, 2160, 256, "L2D")}
BYTES_PER_PIX_ITER = {"L2D": 120.0, "L1D": 132.0}     # SURVEY.md 8(d), fp32, reference 3-op formulation
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8 TB/s
# HBM-side bytes per launch of the dominant CG kernel from the PMC passes of profiles/r01f_hotpath_1280x720x64_pmc.csv
# (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, KiB); PMC counters cannot be read inside a plain bench run.
PROFILED_TRAFFIC = {("kp_cg", 2): (2 * 54328.6 + 105769.4) * 1024.0}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json configs index (1-4); 2 is the metric's configuration")
    ap.add_argument("--no-rebalance", action="store_true", help="N > 1: keep equal-height strips instead of rebalancing them after the warm-up pass")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the default) | gloo (functional runs of the N>1 path on one GPU)")
    a = ap.parse_args()
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

    from gradientdomain_mitsuba_amd import gpt, parallel, scenes
    import gradientdomain_mitsuba_amd.poisson as P

    desc = scenes.cornell_box(W, H, "diffuse") if SCENE == "cornell" else scenes.atrium(W, H)
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

    def step():
        """-> (rays of this rank, render kernel ms, solve seconds, halo bytes)"""
        sr.render(a.spp)
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
    for _ in range(a.steps):
        r, ms, ss, hb = step()
        rays += r; render_ms += ms; solve_s += ss; halo = hb
    barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall, render_ms, float(rays), solve_s], dtype=torch.float64, device=dev)
    if world > 1:
        mx = t.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = t.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        wall, render_ms, solve_s, rays = float(mx[0]), float(mx[1]), float(mx[3]), float(sm[2])
    else:
        rays = float(rays)

    if rank == 0:
        mray = rays / wall / 1e6
        mpix_iter = W * H * iters * a.steps / solve_s / 1e6
        solver, film, strips = sr.solver, sr.film, sr.strips
        kus = solver.profileKernels(50)
        pus = solver.profilePersistent(20)
        bpi = BYTES_PER_PIX_ITER[PRESET]
        solve_achieved = bpi * mpix_iter * 1e6 / 1e9
        if pus > 0.0:
            # the persistent CG kernel: one launch = cgIterMax iterations over the image; algorithmic bytes = SURVEY 8(d)'s
            # per pix-iter figure x the pix-iters of one launch (the iterate itself never leaves the register file)
            kname, kavg, kb = "kp_cg", pus, bpi * W * H * prm.cgIterMax
        else:
            # fused x_p+stencil: R r,p,x + W x,p,Ap = 72 B/px algorithmic (+12 for w in IRLS); with kf_r_rz (48 B/px) it is the iteration
            kname, kavg, kb = "kf_xp_Ax", kus[3], (72.0 if prm.irlsIterMax == 1 else 84.0) * W * H
        achieved = kb / (kavg * 1e-6) / 1e9 if kavg > 0 else 0.0
        # algorithmic bytes per ray = ray record 32 B + hit record 16 B + nodes fetched x 64 B + triangles tested x 80 B
        # (this build's node packet and fp64 TriAccel record), with the traversal counts measured on the device BVH for a
        # secondary-ray-like set (origins uniform in the scene's box, uniform directions), closest-hit and any-hit weighted by
        # the render's own ray mix.  Not an HBM-roofline workload (the Cornell tables are LDS-resident): stated with that caveat.
        import numpy as np
        rng = np.random.default_rng(1)
        vv = np.asarray(desc.verts, np.float64).reshape(-1, 3)
        oo = vv.min(0) + (vv.max(0) - vv.min(0)) * rng.random((1 << 16, 3))
        dd = rng.normal(size=(1 << 16, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        tstat = scene.trace_stats(oo, dd)
        st_ = film.stats()
        fc = st_["raysTraced"] / max(1, st_["raysTraced"] + st_["shadowRaysTraced"])
        bytes_per_ray = 48.0 + 64.0 * (fc * tstat["nodes_closest"] + (1 - fc) * tstat["nodes_any"]) + 80.0 * (fc * tstat["tris_closest"] + (1 - fc) * tstat["tris_any"])
        tracer_gbs = bytes_per_ray * (rays / world / (render_ms * 1e-3) * world) / 1e9
        samples = W * H * a.spp * a.steps
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
            "tracer_roofline": {"bound": "hbm", "achieved": round(tracer_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(tracer_gbs / HBM_PEAK_GBS, 4),
                                "bytes_per_ray": round(bytes_per_ray, 1), "traversal": {k: round(v, 2) for k, v in tstat.items()},
                                "caveat": "algorithmic bytes per ray x render-kernel ray rate (SURVEY 8d-B); the traversal is latency/issue bound and its tables sit in LDS (small scenes) or L2/Infinity Cache -- not an HBM figure"},
            "poisson": {"value": round(mpix_iter, 1), "unit": "Mpix-iter/s", "preset": PRESET, "solve_ms_per_step": round(1e3 * solve_s / a.steps, 4), "dtype": "f32"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": PROFILED_TRAFFIC.get((kname, a.config)),
                         "traffic_source": "profiles/r01f_hotpath_1280x720x64_pmc.csv" if (kname, a.config) in PROFILED_TRAFFIC else None,
                         "what": "dominant Poisson CG kernel: algorithmic bytes per launch (%g B/pix-iter, SURVEY 8d) / HIP-event launch duration" % bpi,
                         "kernel": kname, "kernel_avg_us": round(kavg, 2), "kernel_bytes": kb,
                         "solve_achieved": round(solve_achieved, 1), "solve_frac": round(solve_achieved / HBM_PEAK_GBS, 4),
                         "kernels_us": {"kf_Ax": round(kus[0], 2), "kf_r_rz": round(kus[1], 2), "kf_x_p": round(kus[2], 2), "kf_xp_Ax": round(kus[3], 2), "kp_cg": round(pus, 2)}},
        }
        if not a.no_cpu_baseline and a.config == 2 and world == 1:       # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(W, H, a.spp)
        print(json.dumps(out))
    sr.close(); scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
