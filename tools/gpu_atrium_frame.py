"""The atrium frame at 1280x720, 8 spp, three launches (profiling target of tools/prof_atrium.sh)."""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gradientdomain_mitsuba_amd import gpt, scenes
W, H = 1280, 720
scene = gpt.Scene(scenes.atrium(W, H), device=0)
film = gpt.Film(scene)
integ = gpt.GradientPathIntegrator(maxDepth=-1)
for rep in range(3):
    film.clear(); integ.renderBlock(scene, film, integ.config(8), (0, 0, W, H)); film.sync()
st = film.stats()
print("atrium %.1f ms, %d rays, avg path length %.2f" % (film.render_ms(), st["raysTraced"] + st["shadowRaysTraced"], st["pathLengthSum"] / st["paths"]), flush=True)
