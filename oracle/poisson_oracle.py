"""oracle/poisson_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes face of oracle/poisson_oracle.c (the sequential-fp32 CPU restatement of the reference's
`poisson::Backend` ops and `poisson::Solver` driver).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  PARITY UNPINNED: see the header of
poisson_oracle.c and DESIGN.md "Oracle pinning".
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgdpt_oracle_poisson.so")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


class Params(C.Structure):
    """Solver::Params solver-configuration fields (Solver.hpp:85-93)."""
    _fields_ = [("alpha", C.c_float), ("irlsIterMax", C.c_int), ("irlsRegInit", C.c_float),
                ("irlsRegIter", C.c_float), ("cgIterMax", C.c_int), ("cgIterCheck", C.c_int),
                ("cgPrecond", C.c_int), ("cgTolerance", C.c_float)]


def build(force=False):
    src = os.path.join(_HERE, "poisson_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.gdo_calc_Px.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, _f32p]
        L.gdo_calc_PTW2x.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, _f32p, _f32p]
        L.gdo_calc_Ax_xAx.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_float, _f32p, _f32p]
        L.gdo_calc_axpy.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_long]
        L.gdo_calc_xdoty.argtypes = [_f32p, _f32p, _f32p, C.c_long]
        L.gdo_calc_r_rz.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_long]
        L.gdo_calc_x_p.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_long]
        L.gdo_calc_w2.argtypes = [_f32p, _f32p, C.c_float, C.c_long]
        L.gdo_calc_MIx.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, _f32p, _f32p]
        L.gdo_params_preset.argtypes = [C.POINTER(Params), C.c_char_p]
        L.gdo_params_preset.restype = C.c_int
        L.gdo_solve.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_int, _f32p, C.c_void_p]
        L.gdo_solve.restype = C.c_long
        L.gdo_evaluate_metrics.argtypes = [_f32p, _f32p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_float, _f32p, C.c_void_p, C.c_void_p]
        L.gdo_gbdpt_prepare_data.argtypes = [C.c_float, _f32p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.gdo_synth_inputs.argtypes = [C.c_int, C.c_int, C.c_uint, _f32p, _f32p, _f32p, C.c_void_p]
        _lib = L
    return _lib


_lib_omp = None


def solve_allcores(params, dx, dy, tp, direct, w, h):
    """The same solve through the OpenMP build of the restatement (libgdpt_oracle_poisson_omp.so).  ONLY for the all-cores CPU
    baseline of bench.py: its dot products are per-thread partial sums, so it is not the checker."""
    global _lib_omp
    if _lib_omp is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "_build", "libgdpt_oracle_poisson_omp.so"))
        L.gdo_solve.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _f32p, C.c_void_p]
        L.gdo_solve.restype = C.c_long
        _lib_omp = L
    keep = [_f(a) if a is not None else None for a in (dx, dy, tp, direct)]
    ptr = [a.ctypes.data_as(C.c_void_p) if a is not None else None for a in keep]
    rec = np.empty(3 * w * h, np.float32)
    _lib_omp.gdo_solve(C.byref(params), ptr[0], ptr[1], ptr[2], ptr[3], w, h, rec, None)
    return rec


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def preset(name, alpha=0.2):
    p = Params()
    if not lib().gdo_params_preset(C.byref(p), name.encode()):
        raise ValueError("unknown preset %r" % name)
    p.alpha = alpha
    return p


def calc_Px(x, w, h, alpha):
    x = _f(x); out = np.empty(9 * w * h, np.float32)
    lib().gdo_calc_Px(out, w, h, alpha, x); return out


def calc_PTW2x(w2, e, w, h, alpha):
    out = np.empty(3 * w * h, np.float32)
    lib().gdo_calc_PTW2x(out, w, h, alpha, _f(w2), _f(e)); return out


def calc_Ax_xAx(w2, x, w, h, alpha):
    Ax = np.empty(3 * w * h, np.float32); s = np.empty(3, np.float32)
    lib().gdo_calc_Ax_xAx(Ax, s, w, h, alpha, _f(w2), _f(x)); return Ax, s


def calc_axpy(a, x, y):
    x = _f(x); out = np.empty_like(x)
    lib().gdo_calc_axpy(out, _f(a), x, _f(y), x.size // 3); return out


def calc_xdoty(x, y):
    x = _f(x); s = np.empty(3, np.float32)
    lib().gdo_calc_xdoty(s, x, _f(y), x.size // 3); return s


def calc_r_rz(r, Ap, rz2, pAp):
    r = _f(r).copy(); rz = np.empty(3, np.float32)
    lib().gdo_calc_r_rz(r, rz, _f(Ap), _f(rz2), _f(pAp), r.size // 3); return r, rz


def calc_x_p(x, p, r, rz, rz2, pAp):
    x = _f(x).copy(); p = _f(p).copy()
    lib().gdo_calc_x_p(x, p, _f(r), _f(rz), _f(rz2), _f(pAp), x.size // 3); return x, p


def calc_w2(e, reg):
    e = _f(e); w2 = np.empty(e.size // 3, np.float32)
    lib().gdo_calc_w2(w2, e, reg, w2.size); return w2


def calc_MIx(w2, x, w, h, alpha):
    out = np.empty(3 * w * h, np.float32)
    lib().gdo_calc_MIx(out, w, h, alpha, _f(w2), _f(x)); return out


def solve(params, dx, dy, tp, direct, w, h, return_x=False):
    """Solver::importImagesMTS/setupBackend/solveIndirect/exportImagesMTS in one call."""
    keep = [_f(a) if a is not None else None for a in (dx, dy, tp, direct)]
    ptr = [a.ctypes.data_as(C.c_void_p) if a is not None else None for a in keep]
    rec = np.empty(3 * w * h, np.float32)
    xo = np.empty(3 * w * h, np.float32) if return_x else None
    iters = lib().gdo_solve(C.byref(params), ptr[0], ptr[1], ptr[2], ptr[3], w, h, rec,
                            xo.ctypes.data_as(C.c_void_p) if return_x else None)
    return (rec, xo, iters) if return_x else rec


def synth_inputs(w, h, seed=12345, with_direct=True):
    """SURVEY.md 8(d) synthetic solver input (dx, dy, throughput, direct)."""
    n3 = 3 * w * h
    dx, dy, tp = (np.empty(n3, np.float32) for _ in range(3))
    direct = np.empty(n3, np.float32) if with_direct else None
    lib().gdo_synth_inputs(w, h, seed, dx, dy, tp, direct.ctypes.data_as(C.c_void_p) if with_direct else None)
    return dx, dy, tp, direct


def evaluate_metrics(x, dx, dy, tp, w, h, alpha):
    """Solver::evaluateMetricsMTS for the iterate x (the indirect solution, before `direct` is added) -> (err[3n], errL1, errL2)."""
    err = np.zeros(3 * w * h, np.float32)
    l1, l2 = C.c_float(0), C.c_float(0)
    tpp = None if tp is None else np.ascontiguousarray(tp, np.float32).ctypes.data_as(C.c_void_p)
    lib().gdo_evaluate_metrics(np.ascontiguousarray(x, np.float32), np.ascontiguousarray(dx, np.float32), np.ascontiguousarray(dy, np.float32), tpp,
                               w, h, C.c_float(alpha), err, C.byref(l1), C.byref(l2))
    return err, float(l1.value), float(l2.value)


def gbdpt_prepare_data(w, data, data2=None, offset=0):
    """GBDPTIntegrator::prepareDataForSolver (gbdpt.cpp:264-280): data, data2 = developed double buffers (3*w*h), -> fp32 solver input."""
    data = np.ascontiguousarray(data, np.float64).ravel()
    out = np.zeros(data.size, np.float32)
    d2 = None
    if data2 is not None:
        d2 = np.ascontiguousarray(data2, np.float64).ravel()
        assert d2.size == data.size
    lib().gdo_gbdpt_prepare_data(C.c_float(w), out, data.ctypes.data_as(C.c_void_p), data.size, None if d2 is None else d2.ctypes.data_as(C.c_void_p), offset)
    return out


def gbdpt_reconstruct(primal, grad_neg_y, grad_neg_x, grad_pos_x, grad_pos_y, w, h, alpha=0.2):
    """The second half of GBDPTIntegrator::render (gbdpt.cpp:178-247): the three prepareDataForSolver calls, then an L2D and an L1D
    solve without a direct image -> (L2 image, L1 image), each 3*w*h fp32."""
    imgf = gbdpt_prepare_data(1.0, primal)
    dyf = gbdpt_prepare_data(1.0, grad_pos_y, grad_neg_y, w)
    dxf = gbdpt_prepare_data(1.0, grad_pos_x, grad_neg_x, 1)
    out = []
    for name in ("L2D", "L1D"):
        out.append(solve(preset(name, alpha), dxf, dyf, imgf, None, w, h))
    return out[0], out[1]


def tonemap_srgb(x, idx, num_pixels, scale, bias):
    """Backend::tonemapSRGB (/root/reference/src/integrators/poisson_solver/Backend.cpp:442-468) in numpy fp32: ABGR_8888 words of
    sRGB(in[i + idx * numPixels] * scale + bias).  Test infrastructure (checker of gdpt_backend_tonemap_srgb)."""
    c = np.asarray(x, np.float32).reshape(-1, 3)[idx * num_pixels:(idx + 1) * num_pixels]
    t = c * np.float32(scale) + np.float32(bias)
    with np.errstate(invalid="ignore"):
        s = np.where(t <= np.float32(0.0031308), np.float32(12.92) * t, np.float32(1.055) * np.power(np.maximum(t, 0), np.float32(1.0 / 2.4), dtype=np.float32) - np.float32(0.055)).astype(np.float32)
    q = np.minimum(np.maximum(s * np.float32(255.0) + np.float32(0.5), np.float32(0.0)), np.float32(255.0)).astype(np.int64)
    return (0xFF000000 | q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16)).astype(np.uint32)


def tonemap_linear(x, idx, num_pixels, num_components, scale_min, scale_max, has_negative):
    """Backend::tonemapLinear (Backend.cpp:472-507) in numpy fp32."""
    total = num_pixels * num_components
    v = np.asarray(x, np.float32).ravel()[idx * total:(idx + 1) * total]
    in_min, in_max = np.float32(v.min()), np.float32(v.max())
    fmin = np.float32(np.finfo(np.float32).tiny)
    raw = (np.float32(0.5) / max(max(-in_min, in_max), fmin)) if has_negative else (np.float32(1.0) / max(in_max, fmin))
    scale = np.float32(min(max(raw, np.float32(scale_min)), np.float32(scale_max)))
    bias = np.float32(0.5 if has_negative else 0.0)
    comp = np.abs(v.reshape(num_pixels, num_components) * scale + bias).astype(np.float32)
    col = np.stack([comp[:, min(k, num_components - 1)] for k in range(3)], axis=1)
    q = np.minimum(np.maximum(col * np.float32(255.0) + np.float32(0.5), np.float32(0.0)), np.float32(255.0)).astype(np.int64)
    return (0xFF000000 | q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16)).astype(np.uint32)
