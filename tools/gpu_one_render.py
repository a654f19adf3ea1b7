"""One scene, a few frames: for kernel traces (rocprofv3 --kernel-trace --stats -- python tools/gpu_one_render.py cornell 32)."""
import sys
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1280, 720)
desc = scenes.atrium(W, H) if name == "atrium" else scenes.cornell_box(W, H, "diffuse" if name == "cornell" else name)
scene = gpt.Scene(desc, device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1 if name != "glossy" else 12)
film = gpt.Film(scene)
import os
if os.environ.get("WF_PIPE"):
    film.set_pipeline(int(os.environ["WF_PIPE"]))          # (3: wavefront continuation, GDPT_WF_ITERS traced bounces)
for rep in range(3):
    film.clear(); integ.renderBlock(scene, film, integ.config(spp), (0, 0, W, H)); film.sync()
    st = film.stats()
    print("%s %d spp: %.1f ms  %.0f Mray/s" % (name, spp, film.render_ms(), (st["raysTraced"] + st["shadowRaysTraced"]) / film.render_ms() / 1e3), flush=True)
