"""Host-side mirror of the reference's `gbdpt` integrator plugin (gradient-domain bidirectional path tracing, BASELINE config 5) over the
C-ABI of include/gdpt_tracer.h ("G-BDPT") and include/gdpt_poisson.h.

`GBDPTIntegrator` takes the plugin's properties with the reference's names, defaults and error behaviour
(/root/reference/src/integrators/gbdpt/gbdpt.cpp:79-104) and `render()` walks the steps of GBDPTIntegrator::render (:140-262): the MultiFilm
buffers `-L1|-L2, -gradientNegY, -gradientNegX, -gradientPosX, -gradientPosY, -L2|-L1, -primal` (:163), the sampling job
(GBDPTProcess / GBDPTRenderer, gbdpt_proc.cpp), develop, prepareDataForSolver, BOTH reconstructions (L2D and L1D, :215-251).
Scope of the sampler: surface scenes with area emitters, a perspective sensor and the box filter; BSDFs diffuse / roughconductor (the fast form)
and -- round 4 -- conductor, dielectric and rough conductors below shiftThreshold: samples that meet such a SPECULAR vertex run the general form
(offset paths by propagatePerturbation and manifold walks, include/gdpt_tracer.h); all arithmetic runs in lib/libgdpt_hip.so."""
import ctypes as C

import numpy as np

from . import poisson as _poisson
from ._lib import check, lib

SAMPLER_BUFFERS = ("-primal", "-gradientNegY", "-gradientNegX", "-gradientPosX", "-gradientPosY")     # block / light image i of the sampler


class Config(C.Structure):
    _fields_ = [("maxDepth", C.c_int), ("rrDepth", C.c_int), ("lightImage", C.c_int), ("spp", C.c_int),
                ("shiftThreshold", C.c_double), ("seed", C.c_ulonglong)]


class Film:
    """GBDPTWorkResult for the whole crop window (five camera blocks + five light images, gbdpt_wr.{h,cpp}), device-resident."""

    def __init__(self, scene):
        self.scene = scene
        self.width, self.height = scene.width, scene.height
        self._h = C.c_void_p()
        check(lib().gdpt_gbdpt_film_create(scene._h, C.byref(self._h)))

    def clear(self):
        check(lib().gdpt_gbdpt_film_clear(self._h))

    def sync(self):
        check(lib().gdpt_gbdpt_film_sync(self._h))

    def accum(self):
        """-> (block[5,H,W,4] (rgb, weight), light[5,H,W,3]) raw sums"""
        block = np.empty((5, self.height, self.width, 4), np.float64)
        light = np.empty((5, self.height, self.width, 3), np.float64)
        check(lib().gdpt_gbdpt_film_accum(self._h, block.ctypes.data_as(C.c_void_p), light.ctypes.data_as(C.c_void_p)))
        return block, light

    def develop(self, buffer, spp):
        """GBDPTProcess::develop + MultiFilm::developMulti of one buffer -> float64 [H, W, 3] on the host."""
        out = np.empty((self.height, self.width, 3), np.float64)
        check(lib().gdpt_gbdpt_film_develop(self._h, int(buffer), int(spp), out.ctypes.data_as(C.c_void_p)))
        return out

    def develop_device(self, buffer, spp, tensor):
        """develop() into a contiguous float64 device tensor of H*W*3 values (asynchronous on the film's stream; sync() before reading)."""
        import torch
        if not (hasattr(tensor, "data_ptr") and tensor.is_cuda and tensor.is_contiguous() and tensor.dtype == torch.float64 and tensor.numel() >= 3 * self.width * self.height):
            raise ValueError("develop_device: needs a contiguous float64 device tensor of at least H*W*3 elements")
        check(lib().gdpt_gbdpt_film_develop_device(self._h, int(buffer), int(spp), C.c_void_p(tensor.data_ptr())))
        return tensor

    def export_device(self, block, light):
        """The raw sums into two contiguous float64 device tensors ([5,H,W,4] and [5,H,W,3])."""
        check(lib().gdpt_gbdpt_film_export_device(self._h, C.c_void_p(block.data_ptr()), C.c_void_p(light.data_ptr())))

    def import_device(self, block, light):
        check(lib().gdpt_gbdpt_film_import_device(self._h, C.c_void_p(block.data_ptr()), C.c_void_p(light.data_ptr())))

    def stats(self):
        s = (C.c_ulonglong * 4)()
        check(lib().gdpt_gbdpt_film_stats(self._h, s))
        return dict(raysTraced=int(s[0]), shadowRaysTraced=int(s[1]), samples=int(s[2]), invalidPuts=int(s[3]))

    def chain_stats(self):
        """Samples since the last clear that ran in the general form (specular chains), and how often that form's workspace ran out (0)."""
        s = (C.c_ulonglong * 2)()
        check(lib().gdpt_gbdpt_film_chain_stats(self._h, s))
        return dict(generalSamples=int(s[0]), overflows=int(s[1]))

    def render_ms(self):
        lib().gdpt_gbdpt_film_render_ms.restype = C.c_float
        return float(lib().gdpt_gbdpt_film_render_ms(self._h))

    def close(self):
        if self._h:
            lib().gdpt_gbdpt_film_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GBDPTIntegrator:
    """`<integrator type="gbdpt">` (gbdpt.cpp:79-104): same property names, defaults and errors."""

    def __init__(self, maxDepth=-1, rrDepth=5, lightImage=True, shiftThreshold=0.001, reconstructL1=True, reconstructL2=False, reconstructAlpha=0.2):
        if reconstructL1 and reconstructL2:
            raise RuntimeError("Disable 'reconstructL1' or 'reconstructL2': Cannot display two reconstructions at a time!")   # gbdpt.cpp:91-92
        if reconstructAlpha <= 0.0:
            raise RuntimeError("'reconstructAlpha' must be set to a value greater than zero!")                              # :94-95
        if rrDepth <= 0:
            raise RuntimeError("'rrDepth' must be set to a value greater than zero!")                                       # :99-100
        if maxDepth <= 0 and maxDepth != -1:
            raise RuntimeError("'maxDepth' must be set to -1 (infinite) or a value greater than zero!")                      # :102-103
        self.maxDepth, self.rrDepth, self.lightImage, self.shiftThreshold = maxDepth, rrDepth, lightImage, shiftThreshold
        self.reconstructL1, self.reconstructL2, self.reconstructAlpha = reconstructL1, reconstructL2, reconstructAlpha
        self.stats = {}

    def config(self, spp, seed=5489):
        return Config(self.maxDepth, self.rrDepth, int(self.lightImage), spp, self.shiftThreshold, seed)

    def outNames(self):
        """The MultiFilm buffer names in the integrator's order (gbdpt.cpp:163)."""
        return [("-L1" if self.reconstructL1 else "-L2"), "-gradientNegY", "-gradientNegX", "-gradientPosX", "-gradientPosY",
                ("-L2" if self.reconstructL1 else "-L1"), "-primal"]

    def renderBlock(self, scene, film, cfg, rect):
        """GBDPTRenderer::process + GBDPTProcess::processResult for one rectangle (x0, y0, x1, y1)."""
        x0, y0, x1, y1 = rect
        check(lib().gdpt_gbdpt_render_rect(scene._h, C.byref(cfg), x0, y0, x1, y1, film._h))

    def evaluate_sample(self, scene, cfg, px, py, sample, max_light=256):
        """One sample of GBDPTRenderer::process (the probe entry): dict(primal, gradients[4,3], position, light[n,6], ray counters)."""
        out = np.zeros(17, np.float64)
        light = np.zeros((max_light, 6), np.float64)
        n = C.c_int(0)
        cnt = (C.c_ulonglong * 4)()
        check(lib().gdpt_gbdpt_evaluate_sample2(scene._h, C.byref(cfg), px, py, sample, out.ctypes.data_as(C.c_void_p), max_light,
                                                light.ctypes.data_as(C.c_void_p), C.byref(n), cnt))
        # general: the sample met a specular vertex and ran in the general form (manifold walks); overflow: its workspace ran out (must be 0)
        return dict(primal=out[0:3], gradients=out[3:15].reshape(4, 3), position=out[15:17], light=light[:min(n.value, max_light)].copy(),
                    raysTraced=int(cnt[0]), shadowRaysTraced=int(cnt[1]), general=bool(cnt[2]), overflow=int(cnt[3]))

    def render(self, scene, spp, seed=5489, film=None, reconstruct=True):
        """GBDPTIntegrator::render (gbdpt.cpp:140-262).  Returns {suffix: image [H, W, 3]}: the five sampler buffers as developed doubles,
        `-L2` and `-L1` as the solver's float32 reconstructions (both are always computed and written; the `reconstructL*` properties only
        pick the one shown first, :87,235-251)."""
        own = film is None
        film = film or Film(scene)
        cfg = self.config(spp, seed)
        film.clear()
        self.renderBlock(scene, film, cfg, (0, 0, scene.width, scene.height))
        film.sync()
        out = {name: film.develop(i, spp) for i, name in enumerate(SAMPLER_BUFFERS)}
        self.stats = film.stats()
        self.stats["render_ms"] = film.render_ms()
        if reconstruct:
            l2, l1 = _poisson.gbdpt_reconstruct(*[out[n] for n in SAMPLER_BUFFERS], scene.width, scene.height, alpha=self.reconstructAlpha)
            out["-L2"] = l2.reshape(scene.height, scene.width, 3)
            out["-L1"] = l1.reshape(scene.height, scene.width, 3)
        if own:
            film.close()
        return out

    def render_device(self, scene, spp, seed=5489, film=None, device=None):
        """render() with everything resident in HBM: -> dict of device tensors ("-primal" ... float64 [H, W, 3]; "-L2", "-L1" float32) and
        self.last = dict(render_ms, develop_ms, solve_s (L2D, L1D), rays)."""
        import time
        import torch
        own = film is None
        film = film or Film(scene)
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        cfg = self.config(spp, seed)
        film.clear()
        self.renderBlock(scene, film, cfg, (0, 0, scene.width, scene.height))
        film.sync()
        t0 = time.perf_counter()
        bufs = [torch.empty((scene.height, scene.width, 3), dtype=torch.float64, device=dev) for _ in SAMPLER_BUFFERS]
        for i, b in enumerate(bufs):
            film.develop_device(i, spp, b)
        film.sync()
        t1 = time.perf_counter()
        l2, l1, secs = _poisson.gbdpt_reconstruct_device(bufs, scene.width, scene.height, alpha=self.reconstructAlpha)
        out = dict(zip(SAMPLER_BUFFERS, bufs))
        out["-L2"], out["-L1"] = l2, l1
        self.stats = film.stats()
        self.last = dict(render_ms=film.render_ms(), develop_ms=1e3 * (t1 - t0), solve_s=secs, rays=self.stats["raysTraced"] + self.stats["shadowRaysTraced"])
        if own:
            film.close()
        return out
