"""Where k_bdg_offset's lanes spend their clocks (development build: bash tools/build_profile_lib.sh, then
   gpurun -- 'GDPT_LIB=$PWD/gradientdomain-mitsuba_amd/lib/libgdpt_hip_prof.so timeout 200 python tools/gpu_gbdpt_profile.py')."""
import ctypes as C, sys
sys.path.insert(0, ".")
from gradientdomain_mitsuba_amd import scenes
from gradientdomain_mitsuba_amd._lib import lib, check
import gradientdomain_mitsuba_amd.gpt as G
import gradientdomain_mitsuba_amd.gbdpt as B
W, H, spp = 1280, 720, 2
S = G.Scene(scenes.veach_bidir(W, H, specular=True))
integ = B.GBDPTIntegrator(maxDepth=-1)
F = B.Film(S)
for rep in range(2):
    F.clear()
    integ.renderBlock(S, F, integ.config(spp), (0, 0, W, H)); F.sync()
out = (C.c_ulonglong * 6)()
check(lib().gdpt_gbdpt_film_profile(F._h, out))
v = [int(x) for x in out]
names = ("generateOffsetPath", "  of which manifoldWalk", "halfJacobian x2", "calcSpecularPDFChange", "radianceProducts", "prepareOffset (all)")
print("render %.1f ms" % F.render_ms())
for n, x in zip(names, v):
    print("%-28s %6.1f %% of the lanes' clocks in prepareOffset" % (n, 100.0 * x / max(v[5], 1)))
