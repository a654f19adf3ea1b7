"""Host-side mirror of the reference's `poisson::Solver` interface over the C-ABI of include/gdpt_poisson.h.

Same names, argument meaning and call order as /root/reference/src/integrators/poisson_solver/Solver.hpp:
`Params` (+ `setConfigPreset`), `Solver.importImagesMTS / setupBackend / solveIndirect / exportImagesMTS`,
used exactly as gpt.cpp:1445-1462 uses them.  The `Backend` class mirrors the `poisson::Backend` virtuals
(Backend.hpp:66-100) on device vectors.  All arithmetic runs in the gfx950 library; nothing here computes.
"""
import ctypes as C

import numpy as np

from ._lib import GdptError, check, lib

_fp = C.POINTER(C.c_float)


class Params(C.Structure):
    """Solver::Params (Solver.hpp:49-107), solver-configuration subset + device/verbose."""
    _fields_ = [("alpha", C.c_float), ("irlsIterMax", C.c_int), ("irlsRegInit", C.c_float), ("irlsRegIter", C.c_float),
                ("cgIterMax", C.c_int), ("cgIterCheck", C.c_int), ("cgPrecond", C.c_int), ("cgTolerance", C.c_float),
                ("device", C.c_int), ("verbose", C.c_int)]

    def __init__(self, preset=None, alpha=None, **kw):
        super().__init__()
        lib().gdpt_poisson_params_defaults(C.byref(self))      # Params::setDefaults, Solver.cpp:57-88
        if preset is not None and not self.setConfigPreset(preset):
            raise ValueError("unknown preset %r" % (preset,))
        if alpha is not None:
            self.alpha = alpha
        for k, v in kw.items():
            setattr(self, k, v)

    def setConfigPreset(self, preset):
        """Params::setConfigPreset (Solver.cpp:94-178); returns False for an unknown name like the reference."""
        return bool(lib().gdpt_poisson_params_preset(C.byref(self), preset.encode()))


_LOG = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)


def _ptr(a):
    """float* of a numpy fp32 array, a torch tensor, an int address, or None."""
    if a is None:
        return None, None, False
    if isinstance(a, int):
        return C.cast(a, _fp), None, True
    if hasattr(a, "data_ptr"):                                   # torch tensor
        if a.dtype.__str__() != "torch.float32" or not a.is_contiguous():
            raise TypeError("tensor must be contiguous float32")
        return C.cast(a.data_ptr(), _fp), a, a.is_cuda
    arr = np.ascontiguousarray(a, dtype=np.float32)
    return arr.ctypes.data_as(_fp), arr, False


class Solver:
    """poisson::Solver (Solver.hpp:43-157)."""

    def __init__(self, params):
        L = lib()
        self._h = C.c_void_p()
        check(L.gdpt_poisson_create(C.byref(params), C.byref(self._h)))
        self._keep = []
        self._log = None
        self._size = None

    def setLogFunction(self, fn):
        """Params::setLogFunction (Solver.cpp:194-197): fn(str)."""
        self._log = _LOG(lambda msg, _user: fn(msg.decode()))
        check(lib().gdpt_poisson_set_log(self._h, self._log, None))

    def importImagesMTS(self, dx, dy, tp, direct, width, height):
        """Solver::importImagesMTS (Solver.cpp:220-228).  Accepts host numpy arrays or device torch tensors."""
        ptrs, keep, dev = [], [], []
        for a in (dx, dy, tp, direct):
            p, k, d = _ptr(a)
            ptrs.append(p); keep.append(k)
            if a is not None:
                dev.append(d)
        if len(set(dev)) > 1:
            raise TypeError("inputs must be all host or all device")
        self._keep = keep
        self._size = (width, height)
        fn = lib().gdpt_poisson_import_images_device if (dev and dev[0]) else lib().gdpt_poisson_import_images
        check(fn(self._h, ptrs[0], ptrs[1], ptrs[2], ptrs[3], width, height))

    def setupBackend(self):
        check(lib().gdpt_poisson_setup_backend(self._h))

    def solveIndirect(self):
        check(lib().gdpt_poisson_solve_indirect(self._h))

    def solveIndirectAsync(self):
        check(lib().gdpt_poisson_solve_indirect_async(self._h))

    def sync(self):
        check(lib().gdpt_poisson_sync(self._h))

    def exportImagesMTS(self, rec=None):
        """Solver::exportImagesMTS (Solver.cpp:542-582): returns / fills 3*w*h floats."""
        w, h = self._size
        if rec is not None and hasattr(rec, "data_ptr"):                   # a torch tensor: device or host export by where it lives
            import torch
            if rec.dtype != torch.float32 or not rec.is_contiguous() or rec.numel() < 3 * w * h:
                raise ValueError("exportImagesMTS: rec must be a contiguous float32 tensor of at least 3*w*h = %d elements" % (3 * w * h))
            fn = lib().gdpt_poisson_export_images_device if rec.is_cuda else lib().gdpt_poisson_export_images
            check(fn(self._h, C.cast(rec.data_ptr(), _fp)))
            return rec
        if rec is not None and (not isinstance(rec, np.ndarray) or rec.dtype != np.float32 or not rec.flags["C_CONTIGUOUS"]
                                or not rec.flags["WRITEABLE"] or rec.size < 3 * w * h):
            raise ValueError("exportImagesMTS: rec must be a writable C-contiguous float32 array of at least 3*w*h = %d elements" % (3 * w * h))
        out = np.empty(3 * w * h, np.float32) if rec is None else rec
        check(lib().gdpt_poisson_export_images(self._h, out.ctypes.data_as(_fp)))
        return out

    def evaluateMetricsMTS(self):
        """Solver::evaluateMetricsMTS (Solver.cpp:511-541) -> (err float32 [3*w*h], errL1, errL2) for the current iterate."""
        w, h = self._size
        err = np.empty(3 * w * h, np.float32)
        l1, l2 = C.c_float(0.0), C.c_float(0.0)
        check(lib().gdpt_poisson_evaluate_metrics(self._h, err.ctypes.data_as(_fp), C.byref(l1), C.byref(l2)))
        return err, float(l1.value), float(l2.value)

    def profileKernels(self, reps=50):
        """Bench hook: mean standalone microseconds of (stencil, r_rz, x_p, fused x_p+stencil)."""
        us = (C.c_float * 4)()
        check(lib().gdpt_poisson_profile_kernels(self._h, int(reps), us))
        return [float(v) for v in us]

    def profileStream(self, reps=10):
        """Bench hook: microseconds (best of reps) of a bare streaming kernel with kf_xp_Ax's access mix over this solver's own vectors: the
        yardstick of that kernel's HBM fraction, measured in this process on this device.  Clobbers the iterate."""
        us = C.c_float(0.0)
        check(lib().gdpt_poisson_profile_stream(self._h, int(reps), C.byref(us)))
        return float(us.value)

    def profilePersistent(self, reps=20):
        """Bench hook: mean microseconds of one launch of the persistent CG kernel (cgIterMax iterations); 0 if not used."""
        us = C.c_float(0.0)
        check(lib().gdpt_poisson_profile_persistent(self._h, int(reps), C.byref(us)))
        return float(us.value)

    def setFusion(self, level):
        check(lib().gdpt_poisson_set_fusion(self._h, int(level)))

    @property
    def stream(self):
        lib().gdpt_poisson_stream.restype = C.c_void_p
        return lib().gdpt_poisson_stream(self._h)

    @property
    def lastSolveSeconds(self):
        lib().gdpt_poisson_last_solve_seconds.restype = C.c_float
        return float(lib().gdpt_poisson_last_solve_seconds(self._h))

    @property
    def lastIterations(self):
        lib().gdpt_poisson_last_iterations.restype = C.c_long
        return int(lib().gdpt_poisson_last_iterations(self._h))

    def close(self):
        if self._h:
            lib().gdpt_poisson_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reconstruct(dx, dy, tp, direct, width, height, preset="L1D", alpha=0.2):
    """The reconstruction block of GradientPathIntegrator::render (gpt.cpp:1445-1462) in one call."""
    s = Solver(Params(preset, alpha))
    s.importImagesMTS(dx, dy, tp, direct, width, height)
    s.setupBackend()
    s.solveIndirect()
    rec = s.exportImagesMTS()
    s.close()
    return rec


def gbdpt_prepare_data(w, data, data2=None, offset=0):
    """GBDPTIntegrator::prepareDataForSolver (gbdpt.cpp:264-280): the developed double buffer `data` (3*w*h) scaled into the solver's
    fp32 input; with `data2` the entries that have a partner at i + 3*offset are merged with the partner's negated gradient."""
    data = np.ascontiguousarray(data, np.float64).ravel()
    out = np.zeros(data.size, np.float32)
    d2 = None
    if data2 is not None:
        d2 = np.ascontiguousarray(data2, np.float64).ravel()
        if d2.size != data.size:
            raise ValueError("gbdpt_prepare_data: data2 must have the length of data")
    check(lib().gdpt_gbdpt_prepare_data(C.c_float(w), out.ctypes.data_as(_fp), data.ctypes.data_as(C.POINTER(C.c_double)), data.size,
                                        None if d2 is None else d2.ctypes.data_as(C.POINTER(C.c_double)), int(offset)))
    return out


_release_registered = False


def _register_release():
    """The library keeps the three input images and a solver per preset between G-BDPT frames (gdpt_gbdpt_reconstruct_release drops them): released when
    the interpreter exits -- or whenever the host calls gbdpt_reconstruct_release() (a film size that will not come again)."""
    global _release_registered
    if not _release_registered:
        import atexit
        atexit.register(gbdpt_reconstruct_release)
        _release_registered = True


def gbdpt_reconstruct_release():
    check(lib().gdpt_gbdpt_reconstruct_release())


def gbdpt_reconstruct(primal, grad_neg_y, grad_neg_x, grad_pos_x, grad_pos_y, width, height, alpha=0.2, device=-1, l2=True, l1=True):
    """The second half of GBDPTIntegrator::render (gbdpt.cpp:178-247) on the device: the three prepareDataForSolver calls, then the
    L2D and the L1D solve without a direct image.  Buffers in the order of the integrator's MultiFilm ("-primal", "-gradientNegY",
    "-gradientNegX", "-gradientPosX", "-gradientPosY"), developed doubles of 3*width*height each.  -> (L2 image, L1 image) as fp32."""
    n3 = 3 * width * height
    bufs = [np.ascontiguousarray(b, np.float64).ravel() for b in (primal, grad_neg_y, grad_neg_x, grad_pos_x, grad_pos_y)]
    for b in bufs:
        if b.size != n3:
            raise ValueError("gbdpt_reconstruct: every buffer holds 3*width*height values")
    r2 = np.zeros(n3, np.float32) if l2 else None
    r1 = np.zeros(n3, np.float32) if l1 else None
    dp = C.POINTER(C.c_double)
    _register_release()
    check(lib().gdpt_gbdpt_reconstruct(*[b.ctypes.data_as(dp) for b in bufs], width, height, C.c_float(alpha), device,
                                       None if r2 is None else r2.ctypes.data_as(_fp), None if r1 is None else r1.ctypes.data_as(_fp)))
    return r2, r1


def gbdpt_reconstruct_device(bufs, width, height, alpha=0.2, device=-1, l2=True, l1=True):
    """gbdpt_reconstruct with every buffer on the device: `bufs` = five contiguous float64 device tensors of 3*width*height values (the
    developed sampler buffers, MultiFilm order).  -> (L2 tensor or None, L1 tensor or None, (seconds of the L2D solve, of the L1D solve))."""
    import torch
    n3 = 3 * width * height
    for b in bufs:
        if not (b.is_cuda and b.is_contiguous() and b.dtype == torch.float64 and b.numel() == n3):
            raise ValueError("gbdpt_reconstruct_device: five contiguous float64 device tensors of 3*width*height values")
    r2 = torch.empty((height, width, 3), dtype=torch.float32, device=bufs[0].device) if l2 else None
    r1 = torch.empty((height, width, 3), dtype=torch.float32, device=bufs[0].device) if l1 else None
    torch.cuda.current_stream(bufs[0].device).synchronize()           # the library runs on its own streams
    secs = (C.c_float * 2)()
    _register_release()
    check(lib().gdpt_gbdpt_reconstruct_device(*[C.c_void_p(b.data_ptr()) for b in bufs], width, height, C.c_float(alpha), device,
                                              C.c_void_p(r2.data_ptr()) if l2 else None, C.c_void_p(r1.data_ptr()) if l1 else None, secs))
    return r2, r1, (float(secs[0]), float(secs[1]))


class Backend:
    """poisson::Backend virtuals (Backend.hpp:66-100) on device vectors (reference layouts).  Vectors are
    plain device addresses (ints); `upload`/`download` play Backend::write/read."""

    def __init__(self, stream=None):
        self.L = lib()
        self.stream = C.c_void_p(stream)
        self.L.gdpt_backend_alloc.restype = C.c_void_p
        self.L.gdpt_backend_alloc.argtypes = [C.c_size_t]
        self.L.gdpt_backend_free.argtypes = [C.c_void_p]
        self._owned = []

    def allocVector(self, numElems, bytesPerElem):
        p = self.L.gdpt_backend_alloc(numElems * bytesPerElem)
        if not p:
            raise GdptError(self.L.gdpt_last_error().decode())
        self._owned.append(p)
        return p

    def freeVector(self, p):
        """Backend::freeVector (Backend.cpp:78)."""
        if p in self._owned:
            self._owned.remove(p)
            self.L.gdpt_backend_free(C.c_void_p(p))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        p = self.allocVector(arr.size, 4)
        check(self.L.gdpt_backend_write(C.c_void_p(p), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), self.stream))
        return p

    def download(self, p, numFloats):
        out = np.empty(numFloats, np.float32)
        check(self.L.gdpt_backend_read(out.ctypes.data_as(C.c_void_p), C.c_void_p(p), C.c_size_t(out.nbytes), self.stream))
        return out

    def _f(self, p):
        return C.cast(C.c_void_p(p), _fp)

    def set(self, x, y, numFloats):
        check(self.L.gdpt_backend_set(self._f(x), C.c_float(y), C.c_size_t(numFloats), self.stream))

    def copy(self, x, y, nbytes):
        check(self.L.gdpt_backend_copy(C.c_void_p(x), C.c_void_p(y), C.c_size_t(nbytes), self.stream))

    def calc_Px(self, Px, w, h, alpha, x):
        check(self.L.gdpt_backend_calc_Px(self._f(Px), w, h, C.c_float(alpha), self._f(x), self.stream))

    def calc_PTW2x(self, out, w, h, alpha, w2, x):
        check(self.L.gdpt_backend_calc_PTW2x(self._f(out), w, h, C.c_float(alpha), self._f(w2), self._f(x), self.stream))

    def calc_Ax_xAx(self, Ax, xAx, w, h, alpha, w2, x):
        check(self.L.gdpt_backend_calc_Ax_xAx(self._f(Ax), self._f(xAx), w, h, C.c_float(alpha), self._f(w2), self._f(x), self.stream))

    def calc_axpy(self, out, a, x, y, numElems):
        a3 = (C.c_float * 3)(*a)
        check(self.L.gdpt_backend_calc_axpy(self._f(out), a3, self._f(x), self._f(y), numElems, self.stream))

    def calc_xdoty(self, out, x, y, numElems):
        check(self.L.gdpt_backend_calc_xdoty(self._f(out), self._f(x), self._f(y), numElems, self.stream))

    def calc_r_rz(self, r, rz, Ap, rz2, pAp, numElems):
        check(self.L.gdpt_backend_calc_r_rz(self._f(r), self._f(rz), self._f(Ap), self._f(rz2), self._f(pAp), numElems, self.stream))

    def calc_x_p(self, x, p, r, rz, rz2, pAp, numElems):
        check(self.L.gdpt_backend_calc_x_p(self._f(x), self._f(p), self._f(r), self._f(rz), self._f(rz2), self._f(pAp), numElems, self.stream))

    def calc_MIx(self, MIx, w, h, alpha, w2, x):
        check(self.L.gdpt_backend_calc_MIx(self._f(MIx), w, h, C.c_float(alpha), self._f(w2), self._f(x), self.stream))

    def calc_w2(self, w2, e, reg, numElems):
        check(self.L.gdpt_backend_calc_w2(self._f(w2), self._f(e), C.c_float(reg), numElems, self.stream))

    def tonemapSRGB(self, out, x, idx, numPixels, scale, bias):
        """Backend::tonemapSRGB (Backend.cpp:442-468): `out` = numPixels ABGR_8888 words (a device address from allocVector)."""
        check(self.L.gdpt_backend_tonemap_srgb(C.c_void_p(out), self._f(x), int(idx), int(numPixels), C.c_float(scale), C.c_float(bias), self.stream))

    def tonemapLinear(self, out, x, idx, numPixels, numComponents, scaleMin, scaleMax, hasNegative):
        check(self.L.gdpt_backend_tonemap_linear(C.c_void_p(out), self._f(x), int(idx), int(numPixels), int(numComponents), C.c_float(scaleMin), C.c_float(scaleMax),
                                                 int(bool(hasNegative)), self.stream))

    def download_u32(self, p, n):
        out = np.empty(n, np.uint32)
        check(self.L.gdpt_backend_read(out.ctypes.data_as(C.c_void_p), C.c_void_p(p), n * 4, self.stream))
        return out

    def allocTimer(self):
        self.L.gdpt_backend_timer_alloc.restype = C.c_void_p
        return self.L.gdpt_backend_timer_alloc()

    def freeTimer(self, t):
        self.L.gdpt_backend_timer_free(C.c_void_p(t))

    def beginTimer(self, t):
        check(self.L.gdpt_backend_timer_begin(C.c_void_p(t), self.stream))

    def endTimer(self, t):
        """Seconds of device time between beginTimer and endTimer on the backend's stream (Backend.hpp:97-98)."""
        s = C.c_float(0)
        check(self.L.gdpt_backend_timer_end(C.c_void_p(t), self.stream, C.byref(s)))
        return float(s.value)

    def close(self):
        for p in self._owned:
            self.L.gdpt_backend_free(C.c_void_p(p))
        self._owned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
