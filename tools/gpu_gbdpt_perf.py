"""G-BDPT render rate at config 5's geometry (Veach-bidir stand-in, 1280x720) for a few spp."""
import sys, time
sys.path.insert(0, ".")
from gradientdomain_mitsuba_amd import scenes
import gradientdomain_mitsuba_amd.gpt as G
import gradientdomain_mitsuba_amd.gbdpt as B
W, H = 1280, 720
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
name = sys.argv[2] if len(sys.argv) > 2 else "veach"
sc = scenes.veach_bidir(W, H) if name == "veach" else (scenes.veach_bidir(W, H, specular=True) if name == "veach_specular" else scenes.cornell_box(W, H, name))
S = G.Scene(sc)
integ = B.GBDPTIntegrator(maxDepth=-1)
F = B.Film(S)
for rep in range(2):
    F.clear()
    t0 = time.perf_counter()
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H)); F.sync()
    dt = time.perf_counter() - t0
    st = F.stats()
    rays = st["raysTraced"] + st["shadowRaysTraced"]
    print("%s spp %d: %.1f ms (kernel %.1f ms), %.2f Msample/s, %.1f rays/sample, %.1f Mray/s, general-form samples %.1f %%" % (name, spp, 1e3 * dt, F.render_ms(), W * H * spp / dt / 1e6, rays / (W * H * spp), rays / dt / 1e6,
          100.0 * F.chain_stats()["generalSamples"] / (W * H * spp)), F.chain_stats())
