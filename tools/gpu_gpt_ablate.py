import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import scenes, gpt
W, H, spp = 1280, 720, 32
sc = scenes.cornell_box(W, H, "diffuse")
S = gpt.Scene(sc)
prev = None
for md in (1, 2, 3, 4, 6, -1):
    F = gpt.Film(S)
    integ = gpt.GradientPathIntegrator(maxDepth=md)
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H)); F.sync()
    st = F.stats(); ms = F.render_ms(); rays = st['raysTraced'] + st['shadowRaysTraced']
    print("maxDepth %2d: %.1f ms, %.0f Mray/s, rays/sample %.2f (closest %.2f shadow %.2f), ns/sample %.2f" % (md, ms, rays / ms / 1e3, rays / (W*H*spp), st['raysTraced']/(W*H*spp), st['shadowRaysTraced']/(W*H*spp), ms * 1e6 / (W*H*spp)))
    F.close()
