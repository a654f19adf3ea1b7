"""Where the frame of `bench.py --config 4 --gpus 8 --backend gloo --spp 1` (eight ranks on one device) differs from the one-rank frame: image, row, column, values;
with and without the rebalanced partition.  gpurun -- 'python tools/gpu_strips8_locate.py'"""
import sys, os, pathlib, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from tests.test_bench_gpu import run_bench
tmp = pathlib.Path(tempfile.mkdtemp())
one, d1 = run_bench(tmp, 1, config=4, spp=1)
for reb in (False, True):
    eight, d8 = run_bench(tmp, 8, config=4, spp=1, rebalance=reb)
    strips = d8["strips"]
    print("rebalance", reb, "strips", [(int(a), int(b)) for a, b in strips])
    for k in range(d8["images"].shape[0]):
        d = np.abs(d8["images"][k] - d1["images"][k])
        bad = np.argwhere(d.max(-1) > 1e-6 * max(1.0, float(np.abs(d1["images"][k]).max())))
        print(" image", k, "max diff", d.max(), "of", np.abs(d1["images"][k]).max(), "pixels over tolerance", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:20])
        for (y, x) in bad[:5]:
            print("   ", y, x, "eight", d8["images"][k][y, x], "one", d1["images"][k][y, x])
