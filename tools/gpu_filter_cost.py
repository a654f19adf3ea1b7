"""Render time of the config-2 frame (8 spp) with each reconstruction filter: box = per-pixel sums, the rest = generic atomic puts."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H, spp = 1280, 720, 8
names = ["box", "tent", "gaussian", "mitchell", "catmullrom", "lanczos"]
for k in range(6):
    sc = scenes.cornell_box(W, H, "diffuse"); sc.rfilter = scenes.RFILTER_DEFAULTS[k]
    S = gpt.Scene(sc); F = gpt.Film(S)
    integ = gpt.GradientPathIntegrator(maxDepth=-1)
    cfg = integ.config(spp)
    for rep in range(2):
        F.clear(); integ.renderBlock(S, F, cfg, (0, 0, W, H)); F.sync()
    print("%-10s %.1f ms" % (names[k], F.render_ms()), flush=True)
    F.close(); S.close()
