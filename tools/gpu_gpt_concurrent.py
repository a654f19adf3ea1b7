import sys, time, threading
sys.path.insert(0, '.')
from gradientdomain_mitsuba_amd import gpt, scenes
W, H, spp = 1280, 720, 32
desc = scenes.cornell_box(W, H, "diffuse")
scene = gpt.Scene(desc, device=0)
integ = gpt.GradientPathIntegrator(maxDepth=-1)
cfg = integ.config(spp)
films = [gpt.Film(scene), gpt.Film(scene)]
for f in films:
    f.clear(); integ.renderBlock(scene, f, cfg, (0, 0, W, H)); f.sync()
def run(f):
    f.clear(); integ.renderBlock(scene, f, cfg, (0, 0, W, H)); f.sync()
for rep in range(3):
    t0 = time.perf_counter(); run(films[0]); run(films[1]); t1 = time.perf_counter()
    th = [threading.Thread(target=run, args=(f,)) for f in films]
    t2 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t3 = time.perf_counter()
    print("sequential %.1f ms, concurrent %.1f ms (two 32-spp frames)" % (1e3 * (t1 - t0), 1e3 * (t3 - t2)), flush=True)
