"""Trace one G-BDPT sample on both sides (needs the GDPT_BD_TRACE build of gbdpt_capi.hip; the oracle prints with GPO_TRACE_MAIN=1)."""
import os
import sys
os.environ["GPO_TRACE_MAIN"] = "1"
sys.path.insert(0, ".")
from gradientdomain_mitsuba_amd import scenes
import gradientdomain_mitsuba_amd.gpt as G
import gradientdomain_mitsuba_amd.gbdpt as B
from oracle import gpt_oracle as go
W, H = 1280, 720
sc = scenes.veach_bidir(W, H)
S, O = G.Scene(sc), go.Scene(sc)
integ = B.GBDPTIntegrator(maxDepth=-1)
cfg, ocfg = integ.config(1), go.gbdpt_config(maxDepth=-1, spp=1)
for (px, py) in [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]:
    print("=== device", px, py, flush=True)
    integ.evaluate_sample(S, cfg, px, py, 0)
    sys.stdout.flush()
    print("=== oracle", px, py, file=sys.stderr, flush=True)
    O.gbdpt_sample(ocfg, px, py, 0)
