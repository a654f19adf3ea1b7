"""Builds the in-tree gfx950 shared library `lib/libgdpt_hip.so` (hipcc cross-compiles without a GPU)."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "lib", "libgdpt_hip.so")
OBJDIR = os.path.join(PKG, "lib", "obj")
CSRC = os.path.join(PKG, "csrc")
INC = [os.path.join(ROOT, "include", f) for f in ("gdpt_poisson.h", "gdpt_tracer.h")]
# translation units of the library and what each is made of: one object per unit, rebuilt only when its own sources changed
# (the tracer unit is 4 of the 4.5 minutes of hipcc)
# GDPT_WITH_WAVEFRONT=1: a development build that carries the wavefront continuation (gdpt_film_set_pipeline(3), gpt_wave_capi.hip: built, bit-identical to
# the staged pipeline, measured slower -- DESIGN.md); the product library and the GPU suite do not pay for it.
WITH_WAVEFRONT = os.environ.get("GDPT_WITH_WAVEFRONT", "") not in ("", "0")
UNITS = {
    "poisson_capi.hip": ["poisson_kernels.hip.h", "poisson_persistent.hip.h"],
    "gpt_capi.hip": ["gpt_kernels.hip.h", "gpt_render.hip.h", "gpt_scene.hip.h", "gpt_wavefront.hip.h", "gpt_serial.hip.h"],
    "gpt_serial_capi.hip": ["gpt_kernels.hip.h", "gpt_render.hip.h", "gpt_scene.hip.h", "gpt_serial.hip.h"],
    "gbdpt_capi.hip": ["gpt_kernels.hip.h", "gbdpt_kernels.hip.h", "gbdpt_general.hip.h", "gpt_scene.hip.h"],
    "device_capi.hip": [],
}
# per-unit flags: the G-BDPT connection kernel meets its 2-waves-per-SIMD target only when no callee parks spills in AGPRs (one AGPR in a callee
# makes the kernel's unified register count 257)
# -amdgpu-function-calls=0 (every device function inlined into its kernel): a callee is compiled without its kernel's occupancy target and may take AGPRs of its own --
# k_bd_general (the general form of a G-BDPT sample, ~1 000 lines of callees) came out at 256 + 118 registers = ONE wave per SIMD; inlined it is 256 + 0 and two waves:
# 400 -> 342 ms per 568 k general samples (round 4).  The unit takes 5 minutes instead of 40 s (beside gpt_capi.hip's 7).
UNIT_FLAGS = {"gbdpt_capi.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-mllvm", "-amdgpu-function-calls=0"]}
if WITH_WAVEFRONT:
    UNITS["gpt_wave_capi.hip"] = ["gpt_kernels.hip.h", "gpt_render.hip.h", "gpt_scene.hip.h", "gpt_wavefront.hip.h"]
SOURCES = [os.path.join(CSRC, f) for f in UNITS]
# -ffp-contract=off: the per-element arithmetic contract of csrc/poisson_kernels.hip.h (no FMA contraction).
FLAGS = ["--offload-arch=gfx950", os.environ.get("GDPT_OPT", "-O3"), "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-fvisibility=hidden", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include")] + (["-DGDPT_WITH_WAVEFRONT", "-DGDPT_HANDOFF_CONNECTED"] if WITH_WAVEFRONT else [])      # (the wavefront stages carry RAY_CONNECTED offsets only: the hand-over rule of rounds 2-5)


STAMP = LIB + ".flags"          # the flags the library was built with: a development build (GDPT_EXTRA_FLAGS) never passes for the product


def _flags_line():
    # (the checkout's own path is taken out: the same library is up to date wherever the tree is copied to)
    per_unit = sum(([u + ":"] + f for u, f in sorted(UNIT_FLAGS.items())), [])
    return " ".join(FLAGS + os.environ.get("GDPT_EXTRA_FLAGS", "").split() + per_unit).replace(ROOT, "$ROOT")


def _deps(unit):
    return [os.path.join(CSRC, unit)] + [os.path.join(CSRC, f) for f in UNITS[unit]] + INC


def _obj(unit):
    return os.path.join(OBJDIR, unit.replace(".hip", ".o"))


def _flags_changed():
    return not os.path.exists(STAMP) or open(STAMP).read().strip() != _flags_line()


def _unit_stale(unit):
    o = _obj(unit)
    if _flags_changed() or not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in _deps(unit))


def stale():
    """The library is older than a source -- judged by the sources, not by the objects (lib/obj/ does not travel to the GPU box)."""
    if not os.path.exists(LIB) or _flags_changed():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for u in UNITS for d in _deps(u))


def build(force=False, verbose=False):
    if not force and not stale():
        build_host(verbose)
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("GDPT_EXTRA_FLAGS", "").split()
    procs = []
    for unit in UNITS:                                   # the units compile side by side
        if force or _unit_stale(unit):
            cmd = [hipcc] + FLAGS + UNIT_FLAGS.get(unit, []) + extra + ["-c", "-o", _obj(unit), os.path.join(CSRC, unit)]
            if verbose:
                print(" ".join(cmd))
            procs.append((unit, subprocess.Popen(cmd)))
    for unit, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, unit)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(u) for u in UNITS]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_flags_line() + "\n")
    build_host(verbose)
    return LIB


# The optimisation fence (tests/test_opt_fence_gpu.py): the SAME library with the two tracer units compiled at -O1.  Both
# miscompiles met so far were -O3-only (tools/repro/README.md); films and ray counts of the -O3 product are held against this build on the GPU.
FENCE_OPT = "-O1"
FENCE_LIB = os.path.join(PKG, "lib", "libgdpt_hip_O1.so")
FENCE_UNITS = ("gpt_capi.hip", "gbdpt_capi.hip") + (("gpt_wave_capi.hip",) if WITH_WAVEFRONT else ())


def _fence_obj(unit):
    return os.path.join(OBJDIR, unit.replace(".hip", "_O1.o"))


def fence_source_hash():
    """Contents of every source the fence library is made of (+ its optimisation level): what tests/test_opt_fence_gpu.py compares with the
    stamp written at build time -- file times do not survive every way a tree gets copied to a GPU box, contents do."""
    import hashlib
    # (the development flags are part of the configuration: a product built with GDPT_EXTRA_FLAGS is only ever held against a fence built with the same)
    h = hashlib.sha256((FENCE_OPT + " " + os.environ.get("GDPT_EXTRA_FLAGS", "")).encode())
    for d in sorted({d for u in UNITS for d in _deps(u)}):
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()[:16]


def fence_stamp():
    p = FENCE_LIB + ".flags"
    return open(p).read().strip() if os.path.exists(p) else None


def fence_stale():
    if not os.path.exists(FENCE_LIB) or fence_stamp() != fence_source_hash():
        return True
    t = os.path.getmtime(FENCE_LIB)
    return any(os.path.getmtime(d) > t for u in UNITS for d in _deps(u))        # (the other units' product objects are linked in)


def build_fence(force=False, verbose=False):
    """lib/libgdpt_hip_O1.so: gpt_capi.hip (6 min of hipcc) and gbdpt_capi.hip at -O1, started BEFORE the product build so that they run
    side by side with it, linked with the other units' product objects."""
    if not force and not fence_stale():
        return FENCE_LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [FENCE_OPT if f == FLAGS[1] else f for f in FLAGS] + os.environ.get("GDPT_EXTRA_FLAGS", "").split()
    jobs = []
    for u in FENCE_UNITS:
        o = _fence_obj(u)
        if force or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in _deps(u)):
            jobs.append([hipcc] + flags + UNIT_FLAGS.get(u, []) + ["-c", "-o", o, os.path.join(CSRC, u)])
    others = [u for u in UNITS if u not in FENCE_UNITS]
    if not stale():                                      # a tree that arrived with the product library but without lib/obj/
        jobs += [[hipcc] + FLAGS + UNIT_FLAGS.get(u, []) + os.environ.get("GDPT_EXTRA_FLAGS", "").split() + ["-c", "-o", _obj(u), os.path.join(CSRC, u)] for u in others if not os.path.exists(_obj(u))]
    procs = []
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd))
    build(verbose=verbose)
    for cmd, p in zip(jobs, procs):
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", FENCE_LIB] + [_fence_obj(u) for u in FENCE_UNITS] + [_obj(u) for u in others]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(FENCE_LIB + ".flags", "w") as f:
        f.write(fence_source_hash() + "\n")
    return FENCE_LIB


HOST_BIN = os.path.join(PKG, "bin", "gdpt_mitsuba")


def build_host(verbose=False):
    """The C++ host front end (host/gdpt_mitsuba.cpp: scene-XML subset reader + the reference CLI's flags) over the C-ABI."""
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host")] + ([] if verbose else ["-s"]))
    return HOST_BIN
