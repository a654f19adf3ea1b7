"""Builds the in-tree gfx950 shared library `lib/libgdpt_hip.so` (hipcc cross-compiles without a GPU)."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "lib", "libgdpt_hip.so")
SOURCES = [os.path.join(PKG, "csrc", f) for f in ("poisson_capi.hip", "gpt_capi.hip")]
DEPS = SOURCES + [os.path.join(PKG, "csrc", f) for f in ("poisson_kernels.hip.h", "poisson_persistent.hip.h", "gpt_kernels.hip.h", "gpt_render.hip.h")] + \
    [os.path.join(ROOT, "include", f) for f in ("gdpt_poisson.h", "gdpt_tracer.h")]
# -ffp-contract=off: the per-element arithmetic contract of csrc/poisson_kernels.hip.h (no FMA contraction).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include")]


STAMP = LIB + ".flags"          # the flags the library was built with: a development build (GDPT_EXTRA_FLAGS) never passes for the product


def _flags_line():
    # (the checkout's own path is taken out: the same library is up to date wherever the tree is copied to)
    return " ".join(FLAGS + os.environ.get("GDPT_EXTRA_FLAGS", "").split()).replace(ROOT, "$ROOT")


def stale():
    if not os.path.exists(LIB):
        return True
    if not os.path.exists(STAMP) or open(STAMP).read().strip() != _flags_line():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not stale():
        build_host(verbose)
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + os.environ.get("GDPT_EXTRA_FLAGS", "").split() + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_flags_line() + "\n")
    build_host(verbose)
    return LIB


HOST_BIN = os.path.join(PKG, "bin", "gdpt_mitsuba")


def build_host(verbose=False):
    """The C++ host front end (host/gdpt_mitsuba.cpp: scene-XML subset reader + the reference CLI's flags) over the C-ABI."""
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host")] + ([] if verbose else ["-s"]))
    return HOST_BIN
