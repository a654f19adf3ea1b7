"""Trace one sample of a fuzz seed of tools/gpu_gbdpt_fuzz.py on both sides: python tools/gpu_gbdpt_trace_seed.py SEED PX PY SAMPLE (needs the GDPT_BD_TRACE build)."""
import os, sys
os.environ["GPO_TRACE_MAIN"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, gbdpt as B, scenes
from oracle import gpt_oracle as go
seed, px, py, smp = (int(v) for v in sys.argv[1:5])
rng = np.random.default_rng(seed)
W, H = int(rng.integers(12, 36)), int(rng.integers(8, 28))
sc = (scenes.veach_bidir(W, H, specular=True) if seed % 5 == 0 else scenes.cornell_box(W, H, "random", seed=seed)) if os.environ.get("GBDPT_FUZZ_SPECULAR") else (scenes.veach_bidir(W, H) if seed % 5 == 0 else scenes.cornell_box(W, H, "random_connectable", seed=seed))
md = int(rng.choice([-1, 1, 2, 3, 5, 8, 12])); rr = int(rng.choice([1, 3, 5])); li = bool(rng.random() < 0.7)
spp = int(rng.integers(1, 4))
S = G.Scene(sc); O = go.Scene(sc)
integ = B.GBDPTIntegrator(maxDepth=md, rrDepth=rr, lightImage=li)
cfg = integ.config(spp, 5489 + seed); ocfg = go.gbdpt_config(maxDepth=md, rrDepth=rr, lightImage=li, spp=spp, seed=5489 + seed)
print("=== device", flush=True)
integ.evaluate_sample(S, cfg, px, py, smp)
sys.stdout.flush()
print("=== oracle", file=sys.stderr, flush=True)
O.gbdpt_sample(ocfg, px, py, smp)
