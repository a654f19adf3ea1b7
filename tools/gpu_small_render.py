"""Small Cornell renders (debugging aid): python tools/gpu_small_render.py [occupancy [maxDepth [spp [W H]]]]"""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gradientdomain_mitsuba_amd import gpt, scenes
a = [int(v) for v in sys.argv[1:]]
occ = a[0] if len(a) > 0 else 2
md = a[1] if len(a) > 1 else 4
spp = a[2] if len(a) > 2 else 2
W, H = (a[3], a[4]) if len(a) > 4 else (48, 40)
scene = gpt.Scene(scenes.cornell_box(W, H, "diffuse"), device=0)
film = gpt.Film(scene)
film.set_occupancy(occ)
integ = gpt.GradientPathIntegrator(maxDepth=md)
integ.renderBlock(scene, film, integ.config(spp), (0, 0, W, H)); film.sync()
print("ok", occ, md, spp, W, H, film.stats(), flush=True)
