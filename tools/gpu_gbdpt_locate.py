"""Find the samples of a G-BDPT frame on which the HIP path and the oracle disagree in RAY COUNTS (the images agree), by bisection over
rectangles, and print both sides of each such sample."""
import sys
import numpy as np
sys.path.insert(0, ".")
from gradientdomain_mitsuba_amd import scenes
import gradientdomain_mitsuba_amd.gpt as G
import gradientdomain_mitsuba_amd.gbdpt as B
from oracle import gpt_oracle as go

W, H = 1280, 720
sc = scenes.veach_bidir(W, H)
S, O = G.Scene(sc), go.Scene(sc)
integ = B.GBDPTIntegrator(maxDepth=-1)
F = B.Film(S)
cfg, ocfg = integ.config(1), go.gbdpt_config(maxDepth=-1, spp=1)


def counts(rect):
    F.clear()
    integ.renderBlock(S, F, cfg, rect)
    st = F.stats()
    _b, _l, oc = O.gbdpt_render(ocfg, rect)
    return (st["raysTraced"], st["shadowRaysTraced"]), (oc["raysTraced"], oc["shadowRaysTraced"])


found = []


def search(rect):
    g, o = counts(rect)
    if g == o:
        return
    x0, y0, x1, y1 = rect
    if x1 - x0 == 1 and y1 - y0 == 1:
        found.append((x0, y0, g, o))
        return
    if y1 - y0 >= x1 - x0:
        m = (y0 + y1) // 2
        search((x0, y0, x1, m)); search((x0, m, x1, y1))
    else:
        m = (x0 + x1) // 2
        search((x0, y0, m, y1)); search((m, y0, x1, y1))


search((0, 0, W, H))
print("pixels whose ray counts differ:", found)
for (px, py, g, o) in found[:6]:
    a = integ.evaluate_sample(S, cfg, px, py, 0); b = O.gbdpt_sample(ocfg, px, py, 0)
    print(px, py, "rays", (a["raysTraced"], a["shadowRaysTraced"]), (b["raysTraced"], b["shadowRaysTraced"]))
    print(" primal", a["primal"], b["primal"])
    print(" grads", a["gradients"].ravel(), "\n       ", b["gradients"].ravel())
    print(" light", a["light"], "\n", b["light"])
