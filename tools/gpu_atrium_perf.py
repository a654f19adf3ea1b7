import sys, time
sys.path.insert(0, '.')
import numpy as np
from gradientdomain_mitsuba_amd import scenes, gpt
t=time.time(); sc = scenes.atrium(1920, 1080); print("atrium", sc.ntri, "tris, built in %.1fs" % (time.time()-t))
t=time.time(); S = gpt.Scene(sc); print("upload+BVH %.2fs" % (time.time()-t))
for occ in (2, 4):
    F = gpt.Film(S); F.set_occupancy(occ)
    integ = gpt.GradientPathIntegrator(maxDepth=-1)
    integ.renderBlock(S, F, integ.config(8), (0, 0, 1920, 1080)); F.sync()
    st = F.stats(); ms = F.render_ms(); rays = st['raysTraced'] + st['shadowRaysTraced']
    print("occ%d atrium 1920x1080 spp8: %.1f ms, %.1f Mray/s, rays/sample %.1f, avg path len %.2f" % (occ, ms, rays / ms / 1e3, rays / (1920*1080*8), st['pathLengthSum'] / st['paths']))
    F.close()
