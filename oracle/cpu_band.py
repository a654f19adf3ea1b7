"""oracle/cpu_band.py -- TEST/BENCH INFRASTRUCTURE (the cpu_baseline leg of bench.py), not product code.

Worker for the multi-core CPU baseline: renders rows [y0, y1) of a Cornell frame with the CPU restatement in its own process
(spawned, so it never inherits a HIP context).  Returns (rays, seconds)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def render_band(args):
    W, H, spp, max_depth, y0, y1 = args
    from gradientdomain_mitsuba_amd import scenes
    from oracle import gpt_oracle as go
    S = go.Scene(scenes.cornell_box(W, H, "diffuse"))
    t0 = time.perf_counter()
    _, rays = S.render(go.config(maxDepth=max_depth, spp=spp), rect=(0, y0, W, y1))
    return int(sum(rays)), time.perf_counter() - t0
