"""Investigation: which fields of the sample-queue records differ between the lean (STAGED) and the general build of k_render."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, scenes
W = H = 32
NQ = 62
names = ["thr.x", "thr.y", "thr.z", "pdf", "eta", "p.x", "p.y", "p.z", "d.x", "d.y", "d.z", "u", "v", "depth|prim", "rng"] + \
        ["off%d.%s" % (i, k) for i in range(4) for k in ("tx", "ty", "tz", "pdf")] + ["alive"] + ["A%d" % k for k in range(30)]
sc = scenes.cornell_box(W, H, sys.argv[1] if len(sys.argv) > 1 else "nearspecular")
S = G.Scene(sc)
integ = G.GradientPathIntegrator(maxDepth=int(sys.argv[2]) if len(sys.argv) > 2 else 4, strictNormals=True)
q = {}
os.makedirs("gpurun_out", exist_ok=True)
for mode in ("staged", "general"):
    if mode == "general": os.environ["GDPT_DEV_GENERAL_KERNEL"] = "1"
    else: os.environ.pop("GDPT_DEV_GENERAL_KERNEL", None)
    os.environ["GDPT_DEV_DUMP_QUEUE"] = "gpurun_out/q_%s.bin" % mode
    F = G.Film(S); F.set_pipeline(2)
    integ.renderBlock(S, F, integ.config(1), (0, 0, W, H)); F.sync(); F.close()
    q[mode] = np.fromfile("gpurun_out/q_%s.bin" % mode, dtype=np.float64).reshape(NQ, -1)
a, b = q["staged"], q["general"]
ai, bi = a.view(np.int64), b.view(np.int64)
diff = (ai != bi)
done = (ai[13] == -1) & (bi[13] == -1)
diff[:32, done] = False          # a sample that ended in k_render: only its sums are written
print('finished in k_render:', int(done.sum()), 'of', a.shape[1], '; marks differ:', int(((ai[13] == -1) != (bi[13] == -1)).sum()))
slots = np.nonzero(diff.any(0))[0]
print("slots differing:", len(slots), slots[:20].tolist())
for s in slots[:6]:
    tile, t = divmod(int(s), 256); wave, lane = divmod(t, 64)
    px = (tile % 2) * 16 + (wave & 1) * 8 + (lane & 7); py = (tile // 2) * 16 + (wave >> 1) * 8 + (lane >> 3)
    print("slot", s, "pixel", (px, py), "fields:", [(names[k], float(a[k, s]), float(b[k, s])) if k not in (13, 14, 31) else (names[k], hex(int(ai[k, s])), hex(int(bi[k, s]))) for k in np.nonzero(diff[:, s])[0]])
