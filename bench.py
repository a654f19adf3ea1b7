"""bench.py -- BASELINE.json metric on MI355X.  `python bench.py --gpus N --steps K --warmup W`.

A step = one pass of the hot path over one batch of synthetic input, at BASELINE configs[1]'s size
(1280x720).  Round-1 state: the reconstruction half (screened-Poisson CG, preset L2D = 1 IRLS x 50 CG,
gpt.cpp:1445-1462) is measured; the tracer half (shift-mapped Mray/s) joins `value` when it lands.
Inputs are resident in HBM before the timed region (import + setup are outside it), matching the span
of the reference's m_timerTotal (Solver.cpp:378,500).

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL); the reconstruction does not shard
in this round, so ranks run independent replicas ("replicas only") and value = total
pixel-iterations of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1280, 720
PRESET = "L2D"
# algorithmic bytes per pixel-iteration, fp32, reference 3-op formulation
BYTES_PER_PIX_ITER = {"L2D": 120.0, "L2Q": 120.0, "L1D": 132.0, "L1Q": 132.0, "L1L": 132.0}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline():
    """Oracle (CPU restatement, 1 thread like the reference's BackendOpenMP off-Windows) on a bounded sample."""
    from oracle import poisson_oracle as po
    dx, dy, tp, direct = po.synth_inputs(W, H)
    reps = 6
    t0 = time.perf_counter()
    for _ in range(reps):
        po.solve(po.preset(PRESET), dx, dy, tp, direct, W, H)
    dt = time.perf_counter() - t0
    return {"value": round(W * H * 50 * reps / dt / 1e6, 2), "unit": "Mpix-iter/s", "cores": 1, "kind": "port",
            "sample": "%d x %s solve of the same %dx%d synthetic input (%.1f s)" % (reps, PRESET, W, H, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preset", default=PRESET)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import gradientdomain_mitsuba_amd.poisson as P
    from oracle import poisson_oracle as po   # input generator + cpu_baseline only

    dx, dy, tp, direct = (torch.from_numpy(v).cuda() for v in po.synth_inputs(W, H, seed=12345 + rank))
    prm = P.Params(a.preset, 0.2)
    iters = prm.irlsIterMax * prm.cgIterMax
    s = P.Solver(prm)
    s.importImagesMTS(dx, dy, tp, direct, W, H)

    def step():
        s.setupBackend()          # x0 = T, b: outside the timed span, as in the reference
        s.solveIndirect()
        return s.lastSolveSeconds

    for _ in range(a.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    solve_s = 0.0
    for _ in range(a.steps):
        solve_s += step()
    barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([solve_s, wall], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    solve_s, wall = float(t[0]), float(t[1])

    if rank == 0:
        mpix_iter = W * H * iters * a.steps * world / solve_s / 1e6
        kus = s.profileKernels(50)
        bpi = BYTES_PER_PIX_ITER[a.preset]
        achieved = bpi * (mpix_iter / world) * 1e6 / 1e9            # GB/s per GPU
        # dominant kernel: fused x_p+stencil.  Algorithmic bytes per launch: R r,p,x + W x,p,Ap (+ R w for L1)
        kb = (72.0 if prm.irlsIterMax == 1 else 84.0) * W * H
        out = {
            "metric": "shift-mapped Mray/s + Poisson-CG Mpix-iter/s, 1280x720x64spp (Poisson-CG half; tracer pending)",
            "value": round(mpix_iter, 1), "unit": "Mpix-iter/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * solve_s / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "screened-Poisson %s reconstruct, %dx%d RGB, alpha 0.2 (BASELINE configs[1] size)" % (a.preset, W, H),
                       "cg_iterations_per_step": iters, "parallelism": "replicas x%d" % world},
            "wall_ms_per_step_incl_setup": round(1e3 * wall / a.steps, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "what": "CG iteration, %g B/pix-iter (SURVEY 8d) x pix-iter/s of the timed solves" % bpi,
                         "kernel": "kf_xp_Ax", "kernel_avg_us": round(kus[3], 2),
                         "kernel_achieved": round(kb / (kus[3] * 1e-6) / 1e9, 1) if kus[3] > 0 else None,
                         "kernels_us": {"kf_Ax": round(kus[0], 2), "kf_r_rz": round(kus[1], 2), "kf_x_p": round(kus[2], 2), "kf_xp_Ax": round(kus[3], 2)}},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    s.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
