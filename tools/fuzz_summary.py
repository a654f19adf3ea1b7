"""One-line JSON summary of a fuzz battery (the counts of a battery as a checkable record, not prose): every tools/gpu_*fuzz* run ends with
emit(tool, ...), which prints `FUZZ_SUMMARY {...}` and appends the same line to gpurun_out/fuzz/summaries.jsonl on the GPU box (merged back by gpurun; the lines
that are to be judged are copied to profiles/rNN_fuzz_summaries.jsonl).  Fields every line carries: tool, library (path + sha256[:16] of the binary the battery
ran through), environment knobs that select a build (GDPT_SCENE_IN_HBM, GDPT_LIB, GBDPT_FUZZ_*), the seed range, wall seconds, ok."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def library_hash():
    sys.path.insert(0, ROOT)
    from gradientdomain_mitsuba_amd import _build
    path = os.environ.get("GDPT_LIB") or _build.LIB
    h = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] if os.path.exists(path) else None
    return os.path.relpath(path, ROOT), h


def emit(tool, first, count, seconds, ok=True, **fields):
    path, h = library_hash()
    rec = {"tool": tool, "library": path, "library_sha256_16": h, "seeds": [int(first), int(first) + int(count) - 1], "seconds": round(float(seconds), 1), "ok": bool(ok),
           "env": {k: os.environ[k] for k in ("GDPT_SCENE_IN_HBM", "GDPT_LIB", "GBDPT_FUZZ_SPECULAR", "GBDPT_FUZZ_ENDPOINTS", "GDPT_NO_FIRST_STAGE") if k in os.environ},
           "utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
    rec.update(fields)
    line = json.dumps(rec, sort_keys=True)
    print("FUZZ_SUMMARY " + line, flush=True)
    out = os.path.join(ROOT, "gpurun_out", "fuzz")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "summaries.jsonl"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    return rec
