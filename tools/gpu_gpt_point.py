import sys
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import scenes, gpt
from oracle import gpt_oracle as go
W = H = 32
sc = scenes.cornell_box(W, H, "diffuse")
S = gpt.Scene(sc); O = go.Scene(sc)
for md in (2, 3, 6):
    integ = gpt.GradientPathIntegrator(maxDepth=md)
    nbad = 0
    for (px, py, s) in [(x, y, s) for x in (3, 16, 27) for y in (5, 16, 30) for s in range(3)]:
        g = S.evaluate_point(integ.config(4), px, py, s)
        o = O.evaluate_point(go.config(maxDepth=md, spp=4), px, py, s)
        dT = np.abs(g["throughput"] - o["throughput"]).max(); dG = np.abs(g["gradients"] - o["gradients"]).max(); dN = np.abs(g["neighbours"] - o["neighbours"]).max()
        if max(dT, dG, dN) > 1e-12:
            nbad += 1
            if nbad <= 3:
                print("md", md, (px, py, s), "dT %.3e dG %.3e dN %.3e" % (dT, dG, dN), "gpu rays", g["raysTraced"], g["shadowRaysTraced"], "depth", g["depth"])
                print("   gpu T", g["throughput"], "\n   ora T", o["throughput"])
                print("   gpu N", g["neighbours"].ravel()[:6], "\n   ora N", o["neighbours"].ravel()[:6])
    print("maxDepth", md, "mismatching samples:", nbad, "/ 27")
