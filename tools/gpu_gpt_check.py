"""Ad-hoc GPU check of the HIP tracer against the oracle (development aid; the real tests live in tests/)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np
from gradientdomain_mitsuba_amd import scenes, gpt
from oracle import gpt_oracle as go

def cmp(variant, W, H, spp, maxDepth):
    sc = scenes.cornell_box(W, H, variant)
    S = gpt.Scene(sc); F = gpt.Film(S)
    integ = gpt.GradientPathIntegrator(maxDepth=maxDepth, reconstructL1=False, reconstructL2=True)
    cfg = integ.config(spp)
    t = time.time(); integ.renderBlock(S, F, cfg, (0, 0, W, H)); F.sync(); dt = time.time() - t
    acc = F.accum(); st = F.stats(); ms = F.render_ms()
    O = go.Scene(sc)
    t = time.time(); oacc, orays = O.render(go.config(maxDepth=maxDepth, spp=spp)); odt = time.time() - t
    print("%s %dx%d spp%d depth%d: gpu %.1f ms (%.1f Mray/s) rays %d+%d | oracle %.2fs rays %d+%d" % (
        variant, W, H, spp, maxDepth, ms, (st['raysTraced'] + st['shadowRaysTraced']) / ms / 1e3, st['raysTraced'], st['shadowRaysTraced'], odt, orays[0], orays[1]))
    for b, name in enumerate(gpt.BUFFER_NAMES):
        d = np.abs(acc[b] - oacc[b]); scale = np.abs(oacc[b]).max() + 1e-30
        bad = (d > 1e-9 * scale).any(axis=-1)
        print("   %-12s max|diff| %.3e (scale %.3e)  pixels differing >1e-9 rel: %d / %d" % (name, d.max(), scale, bad.sum(), W * H))
    return acc, oacc

if __name__ == "__main__":
    cmp("diffuse", 32, 32, 4, 6)
    cmp("diffuse", 64, 48, 8, -1)
    cmp("glossy", 64, 48, 8, 10)
    cmp("nearspecular", 48, 48, 8, 10)
