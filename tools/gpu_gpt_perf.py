import sys, time, itertools
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import scenes, gpt
W, H = 1280, 720
for (variant, spp, md), occ in itertools.product((("diffuse", 32, -1), ("glossy", 16, 12)), (2, -2, 3, 4, -4)):
    sc = scenes.cornell_box(W, H, variant)
    S = gpt.Scene(sc); F = gpt.Film(S); F.set_occupancy(occ)
    integ = gpt.GradientPathIntegrator(maxDepth=md)
    cfg = integ.config(spp)
    integ.renderBlock(S, F, cfg, (0, 0, W, H)); F.sync()
    st = F.stats(); ms = F.render_ms()
    rays = st['raysTraced'] + st['shadowRaysTraced']
    print("occ%+d %s %dx%d spp%d depth%d: %.1f ms, %.1f Mray/s, %.2f Msample/s" % (occ, variant, W, H, spp, md, ms, rays / ms / 1e3, W * H * spp / ms / 1e3))
    F.close(); S.close()
