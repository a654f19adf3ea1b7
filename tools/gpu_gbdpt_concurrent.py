"""Does the G-BDPT sampler leave the GPU room for more concurrency?  Two films of config 5's scene rendered one after the other and from two host threads
(each film has its own streams; ctypes releases the GIL inside a call).  If the pair finishes in well under twice a single frame, overlapping chunk k + 1's walk
with chunk k's connections (double-buffered records) is worth building.  python tools/gpu_gbdpt_concurrent.py [spp]"""
import sys, time, threading
sys.path.insert(0, ".")
from gradientdomain_mitsuba_amd import scenes
import gradientdomain_mitsuba_amd.gpt as G
import gradientdomain_mitsuba_amd.gbdpt as B
W, H = 1280, 720
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = G.Scene(scenes.veach_bidir(W, H, specular=True))
integ = B.GBDPTIntegrator(maxDepth=-1)
films = [B.Film(S), B.Film(S)]
def one(F, seed):
    integ.renderBlock(S, F, integ.config(spp, seed), (0, 0, W, H)); F.sync()
for F in films: one(F, 1)                      # warm (allocations)
for rep in range(3):
    for F in films: F.clear()
    t0 = time.perf_counter(); one(films[0], 2); one(films[1], 3); seq = time.perf_counter() - t0
    for F in films: F.clear()
    th = [threading.Thread(target=one, args=(F, 2 + i)) for i, F in enumerate(films)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; par = time.perf_counter() - t0
    print("spp %d: two films in sequence %.1f ms, from two threads %.1f ms (x%.2f)" % (spp, 1e3 * seq, 1e3 * par, seq / par), flush=True)
