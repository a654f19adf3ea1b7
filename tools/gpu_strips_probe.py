"""Investigation: gdpt_mitsuba with N strips on one device against the one-strip frame, repeated; prints which buffers / rows differ."""
import os, subprocess, sys, tempfile
sys.path.insert(0, '.')
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from gradientdomain_mitsuba_amd import _build
_build.build()
cli = _build.HOST_BIN
XML = os.path.join(ROOT, "scenes", "cornell_box.xml")

def read_pfm(p):
    with open(p, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split()); f.readline()
        return np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)[::-1]

tmp = tempfile.mkdtemp()
args = ["-D", "width=64", "-D", "height=50", "-D", "spp=5", "-D", "maxDepth=7", "-q"]
subprocess.run([cli, "-o", tmp + "/one", *args, XML], check=True)
ref = {s: read_pfm(tmp + "/one" + s + ".pfm") for s in ("-final", "-throughput", "-dx", "-dy", "-direct")}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for extra in (["--devices", "0,0"], ["-p", "3"]):
        subprocess.run([cli, "-o", tmp + "/s", *args, *extra, XML], check=True)
        out = []
        for s, img in ref.items():
            d = np.abs(read_pfm(tmp + "/s" + s + ".pfm") - img).max(-1)
            rows = np.nonzero(d.max(1) > 1e-5 * float(np.abs(img).max()))[0]
            out.append("%s %.1e rows %s" % (s, float(d.max()), (rows.min(), rows.max(), len(rows)) if len(rows) else "-"))
        print(rep, " ".join(extra), "|", " | ".join(out), flush=True)
