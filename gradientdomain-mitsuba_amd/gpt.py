"""Host-side mirror of the reference's `gpt` integrator plugin over the C-ABI of include/gdpt_tracer.h.

`GradientPathIntegrator` takes the plugin's properties with the reference's names, defaults and error behaviour
(/root/reference/src/integrators/gpt/gpt.cpp:1190-1213, 1358-1480) and its `render()` walks the same steps:
five MultiFilm buffers `-final -throughput -dx -dy -direct`, block rendering, develop, screened-Poisson
reconstruction (L1D or L2D preset), reconstruction written to `-final`.  All arithmetic runs in lib/libgdpt_hip.so.
"""
import ctypes as C
import os

import numpy as np

from . import poisson as _poisson
from ._lib import GdptError, check, lib

BUFFER_NAMES = ["-final", "-throughput", "-dx", "-dy", "-direct"]      # gpt.cpp:1380
BUFFER_FINAL, BUFFER_THROUGHPUT, BUFFER_DX, BUFFER_DY, BUFFER_VERY_DIRECT = range(5)   # gpt.cpp:76-80


class Material(C.Structure):
    _fields_ = [("type", C.c_int), ("distribution", C.c_int), ("sampleVisible", C.c_int), ("twoSided", C.c_int),
                ("reflectance", C.c_double * 3), ("eta", C.c_double * 3), ("k", C.c_double * 3),
                ("alphaU", C.c_double), ("alphaV", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("rgb", C.c_void_p), ("wrapU", C.c_int), ("wrapV", C.c_int), ("filter", C.c_int),
                ("uscale", C.c_double), ("vscale", C.c_double), ("uoffset", C.c_double), ("voffset", C.c_double), ("scale", C.c_double), ("maxAnisotropy", C.c_double)]


class Emitter(C.Structure):
    _fields_ = [("firstTri", C.c_int), ("numTris", C.c_int), ("radiance", C.c_double * 3), ("position", C.c_double * 3),
                ("rectangle", C.c_int), ("rectToWorld", C.c_double * 12), ("rectNormal", C.c_double * 3)]


class Environment(C.Structure):
    _fields_ = [("radiance", C.c_double * 3), ("index", C.c_int), ("rgb", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("scale", C.c_double), ("toWorld", C.c_double * 9)]


class Camera(C.Structure):
    _fields_ = [("toWorld", C.c_double * 16), ("fovX", C.c_double), ("nearClip", C.c_double), ("farClip", C.c_double),
                ("width", C.c_int), ("height", C.c_int), ("type", C.c_int), ("apertureRadius", C.c_double), ("focusDistance", C.c_double),
                ("shutterOpen", C.c_double), ("shutterClose", C.c_double),
                ("cropOffsetX", C.c_int), ("cropOffsetY", C.c_int), ("fullWidth", C.c_int), ("fullHeight", C.c_int)]


class Config(C.Structure):
    _fields_ = [("maxDepth", C.c_int), ("rrDepth", C.c_int), ("strictNormals", C.c_int), ("spp", C.c_int),
                ("shiftThreshold", C.c_double), ("seed", C.c_ulonglong)]


def _material(m):
    out = Material()
    out.type = m["type"]
    out.distribution = m.get("distribution", 0)
    out.sampleVisible = m.get("sampleVisible", 1)
    out.twoSided = int(m.get("twoSided", 0))
    out.reflectance = (C.c_double * 3)(*m.get("reflectance", (0.5, 0.5, 0.5)))
    out.eta = (C.c_double * 3)(*m.get("eta", (0.0, 0.0, 0.0)))
    out.k = (C.c_double * 3)(*m.get("k", (1.0, 1.0, 1.0)))
    out.alphaU = m.get("alphaU", 0.1)
    out.alphaV = m.get("alphaV", 0.1)
    return out


def bsdf_probe(m, wi, samples=None, dirs=None, measure=0):
    """gdpt_bsdf_probe: the device's BSDF::sample for each (sx, sy) row of `samples` -> (wo, weight, pdf, sampledType) arrays, and
    BSDF::eval / pdf for each row of `dirs` -> (f, pdf) arrays.  A test probe (the models are otherwise reached through the render kernels)."""
    mm = _material(m)
    w = np.ascontiguousarray(wi, np.float64)
    smp = np.ascontiguousarray(samples if samples is not None else np.zeros((0, 2)), np.float64).reshape(-1, 2)
    d = np.ascontiguousarray(dirs if dirs is not None else np.zeros((0, 3)), np.float64).reshape(-1, 3)
    out8 = np.zeros((len(smp), 8)); out4 = np.zeros((len(d), 4))
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().gdpt_bsdf_probe(C.byref(mm), P(w), len(smp), P(smp), P(out8), len(d), P(d), measure, P(out4)))
    return (out8[:, 0:3], out8[:, 3:6], out8[:, 6], out8[:, 7].astype(int)), (out4[:, 0:3], out4[:, 3])


class Scene:
    """Device-resident scene (flat BVH + triangle records in HBM) built from a `scenes.Scene` description."""

    def __init__(self, desc, device=-1):
        self.desc = desc
        self.width, self.height = desc.width, desc.height
        verts = np.ascontiguousarray(desc.verts, dtype=np.float64)
        tm = np.ascontiguousarray(desc.tri_material, dtype=np.int32)
        mats = (Material * len(desc.materials))(*[_material(m) for m in desc.materials])
        ems = (Emitter * max(1, len(desc.emitters)))()
        for i, em in enumerate(desc.emitters):                # (firstTri, numTris, radiance) or ("point", position, intensity)
            if em[0] == "point":
                ems[i].firstTri, ems[i].numTris, ems[i].position, ems[i].radiance = 0, -1, (C.c_double * 3)(*em[1]), (C.c_double * 3)(*em[2])
            else:
                ems[i].firstTri, ems[i].numTris, ems[i].radiance = em[0], em[1], (C.c_double * 3)(*em[2])
                if len(em) >= 5:                                 # the light of a `rectangle` shape: (firstTri, 2, radiance, toWorld 3x4, normal)
                    ems[i].rectangle = 1
                    ems[i].rectToWorld = (C.c_double * 12)(*np.asarray(em[3], np.float64).reshape(12))
                    ems[i].rectNormal = (C.c_double * 3)(*em[4])
        cam = Camera()
        cam.toWorld = (C.c_double * 16)(*np.asarray(desc.to_world, np.float64).ravel())
        cam.fovX, cam.nearClip, cam.farClip, cam.width, cam.height = desc.fov_x, desc.near, desc.far, desc.width, desc.height
        if getattr(desc, "thinlens", None):                 # (apertureRadius, focusDistance) of a `thinlens` sensor
            cam.type, cam.apertureRadius, cam.focusDistance = 1, float(desc.thinlens[0]), float(desc.thinlens[1])
        if getattr(desc, "shutter", None):                  # (shutterOpen, shutterClose) of the sensor: an interval of positive length draws a time sample
            cam.shutterOpen, cam.shutterClose = float(desc.shutter[0]), float(desc.shutter[1])
        if getattr(desc, "crop", None):                     # (cropOffsetX, cropOffsetY, fullWidth, fullHeight): width x height is the crop window of that film
            cam.cropOffsetX, cam.cropOffsetY, cam.fullWidth, cam.fullHeight = (int(v) for v in desc.crop)
        self._h = C.c_void_p()
        envd = getattr(desc, "environment", None)
        env = None
        if envd is not None:                                 # `<emitter type="constant">`: (radiance rgb, position in the emitter list)
            env = Environment((C.c_double * 3)(*envd[0]), int(envd[1]))
        emap = getattr(desc, "environment_map", None)
        self._envmap_keep = None
        if emap is not None:                                 # `<emitter type="envmap">`: dict(rgb [h, w, 3] linear, scale, toWorld 3x3, index)
            if env is not None:
                raise ValueError("a scene has one environment emitter: `environment` or `environment_map`")
            rgb = np.ascontiguousarray(emap["rgb"], dtype=np.float64)
            self._envmap_keep = rgb
            env = Environment((C.c_double * 3)(0.0, 0.0, 0.0), int(emap.get("index", -1)), rgb.ctypes.data_as(C.c_void_p), rgb.shape[1], rgb.shape[0],
                              float(emap.get("scale", 1.0)), (C.c_double * 9)(*np.asarray(emap.get("toWorld", np.eye(3)), np.float64).reshape(9)))
        nrm = getattr(desc, "normals", None)
        nrm = np.ascontiguousarray(nrm, dtype=np.float64) if nrm is not None else None
        # texture coordinates and bitmap textures (scenes.Scene.uvs / tri_has_uv / textures / material_textures)
        uvs = getattr(desc, "uvs", None)
        uvs = np.ascontiguousarray(uvs, dtype=np.float64) if uvs is not None else None
        has = getattr(desc, "tri_has_uv", None)
        has = np.ascontiguousarray(has, dtype=np.uint8) if (has is not None and uvs is not None) else None
        texs = getattr(desc, "textures", None) or []
        keep = []                                            # texel arrays must outlive the call
        tarr = (Texture * max(1, len(texs)))()
        for i, t in enumerate(texs):
            rgb = np.ascontiguousarray(t["rgb"], dtype=np.float64)
            keep.append(rgb)
            tarr[i].height, tarr[i].width = rgb.shape[0], rgb.shape[1]
            tarr[i].rgb = rgb.ctypes.data_as(C.c_void_p)
            tarr[i].wrapU, tarr[i].wrapV, tarr[i].filter = t.get("wrapU", 0), t.get("wrapV", 0), t.get("filter", 1)
            tarr[i].uscale, tarr[i].vscale, tarr[i].uoffset, tarr[i].voffset = t.get("uscale", 1.0), t.get("vscale", 1.0), t.get("uoffset", 0.0), t.get("voffset", 0.0)
            tarr[i].scale = t.get("scale", 1.0)
            tarr[i].maxAnisotropy = t.get("maxAnisotropy", 20.0)
        mtex = getattr(desc, "material_textures", None)
        mtex = np.ascontiguousarray(mtex, dtype=np.int32) if mtex is not None else None
        check(lib().gdpt_scene_create_tex(verts.shape[0], verts.ctypes.data_as(C.c_void_p),
                                          nrm.ctypes.data_as(C.c_void_p) if nrm is not None else None,
                                          uvs.ctypes.data_as(C.c_void_p) if uvs is not None else None,
                                          has.ctypes.data_as(C.c_void_p) if has is not None else None, tm.ctypes.data_as(C.c_void_p),
                                          len(desc.materials), C.byref(mats), mtex.ctypes.data_as(C.c_void_p) if mtex is not None else None,
                                          len(texs), C.byref(tarr) if texs else None, len(desc.emitters), C.byref(ems),
                                          C.byref(env) if env is not None else None, C.byref(cam), device, C.byref(self._h)))

    def intersect(self, origins, dirs):
        od = np.ascontiguousarray(np.concatenate([np.asarray(origins, np.float64), np.asarray(dirs, np.float64)], axis=1))
        n = od.shape[0]
        prim = np.empty(n, np.int32)
        tp = np.empty((n, 4), np.float64)
        check(lib().gdpt_scene_intersect(self._h, n, od.ctypes.data_as(C.c_void_p), prim.ctypes.data_as(C.c_void_p), tp.ctypes.data_as(C.c_void_p)))
        return prim, tp[:, 0], tp[:, 1:4]

    def intersect_record(self, origins, dirs):
        """The filled intersection record of each ray's closest hit as the render kernels form it (fillIntersectionRecord<true>, skdtree.h:343-428):
        (prim, dict of arrays [n, .]: t, p, uv, geoFrame.n, shFrame.n, shFrame.s, dpdu, dpdv, wi)."""
        od = np.ascontiguousarray(np.concatenate([np.asarray(origins, np.float64), np.asarray(dirs, np.float64)], axis=1))
        n = od.shape[0]
        prim = np.empty(n, np.int32)
        rec = np.empty((n, 24), np.float64)
        lib().gdpt_scene_intersect_record.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        check(lib().gdpt_scene_intersect_record(self._h, n, od.ctypes.data_as(C.c_void_p), prim.ctypes.data_as(C.c_void_p), rec.ctypes.data_as(C.c_void_p)))
        return prim, {"t": rec[:, 0], "p": rec[:, 1:4], "uv": rec[:, 4:6], "geoFrame.n": rec[:, 6:9], "shFrame.n": rec[:, 9:12], "shFrame.s": rec[:, 12:15],
                      "dpdu": rec[:, 15:18], "dpdv": rec[:, 18:21], "wi": rec[:, 21:24]}

    def bsphere_radius(self):
        """Scene::getBSphere().radius after Scene::initializeBidirectional (scene.cpp:386-413): kd-tree bounds + sensor + emitters."""
        r = C.c_double(0.0)
        lib().gdpt_scene_bsphere_radius.argtypes = [C.c_void_p, C.c_void_p]
        check(lib().gdpt_scene_bsphere_radius(self._h, C.byref(r)))
        return float(r.value)

    def layout(self):
        """-> dict(nodes, node_bytes (128: fp32 boxes, LDS-resident scene; 64: 8-bit boxes, scene in HBM), lds_resident, stack_entries, table_bytes, leaf_exit)"""
        out = (C.c_longlong * 6)()
        lib().gdpt_scene_layout.argtypes = [C.c_void_p, C.c_void_p]
        check(lib().gdpt_scene_layout(self._h, out))
        return dict(nodes=int(out[0]), node_bytes=int(out[1]), lds_resident=bool(out[2]), stack_entries=int(out[3]), table_bytes=int(out[4]), leaf_exit=int(out[5]))

    def trace_stats(self, origins, dirs):
        """-> dict of mean inner nodes fetched / triangles tested per ray, closest-hit and any-hit (SURVEY 8d-B bytes per ray)."""
        od = np.ascontiguousarray(np.concatenate([np.asarray(origins, np.float64), np.asarray(dirs, np.float64)], axis=1))
        sums = (C.c_ulonglong * 4)()
        check(lib().gdpt_scene_trace_stats(self._h, od.shape[0], od.ctypes.data_as(C.c_void_p), sums))
        n = float(od.shape[0])
        return dict(nodes_closest=sums[0] / n, tris_closest=sums[1] / n, nodes_any=sums[2] / n, tris_any=sums[3] / n)

    def evaluate_point(self, cfg, px, py, sample):
        out = np.zeros(33, np.float64)
        check(lib().gdpt_scene_evaluate_point(self._h, C.byref(cfg), px, py, sample, out.ctypes.data_as(C.c_void_p)))
        return dict(veryDirect=out[0:3], throughput=out[3:6], gradients=out[6:18].reshape(4, 3), neighbours=out[18:30].reshape(4, 3),
                    raysTraced=int(out[30]), shadowRaysTraced=int(out[31]), depth=int(out[32]))

    def close(self):
        if self._h:
            lib().gdpt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Film:
    """The five G-PT buffers over rows [y0, y1) (MultiFilm with `setBuffers`, multifilm.cpp:293-319), device-resident."""

    def __init__(self, scene, y0=0, y1=None):
        self.scene = scene
        self.y0, self.y1 = y0, scene.height if y1 is None else y1
        self.rows, self.width = self.y1 - self.y0, scene.width
        self._h = C.c_void_p()
        check(lib().gdpt_film_create(scene._h, self.y0, self.y1, C.byref(self._h)))
        self.renders_own_border = False                      # True with a filter wider than box: no halo exchange between strips
        rf = getattr(scene.desc, "rfilter", None)
        if rf is not None:                                   # the scene description's <rfilter>
            self.set_rfilter(*rf)

    def clear(self):
        check(lib().gdpt_film_clear(self._h))

    def accum(self):
        out = np.empty((5, self.rows, self.width, 4), np.float64)
        check(lib().gdpt_film_accum(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def accum_rect(self, x0, y0, x1, y1):
        """The five accumulation buffers over the pixels [x0,x1) x [y0,y1) of the film only: [5, y1 - y0, x1 - x0, 4] -- a block with its border."""
        out = np.empty((5, y1 - y0, x1 - x0, 4), np.float64)
        lib().gdpt_film_accum_rect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        check(lib().gdpt_film_accum_rect(self._h, int(x0), int(y0), int(x1), int(y1), out.ctypes.data_as(C.c_void_p)))
        return out

    def develop(self, buffer):
        """developMulti of one buffer -> float32 [rows, W, 3] on the host."""
        out = np.empty((self.rows, self.width, 3), np.float32)
        check(lib().gdpt_film_develop(self._h, buffer, out.ctypes.data_as(C.c_void_p)))
        return out

    @staticmethod
    def _device_tensor(tensor, dtype_name, numel, what):
        """The C-ABI takes raw device pointers: refuse anything that is not a contiguous device tensor of the right type and size."""
        import torch
        if not (hasattr(tensor, "data_ptr") and tensor.is_cuda and tensor.is_contiguous() and tensor.dtype == getattr(torch, dtype_name) and tensor.numel() >= numel):
            raise ValueError("%s: needs a contiguous %s device tensor of at least %d elements" % (what, dtype_name, numel))
        return C.c_void_p(tensor.data_ptr())

    def develop_device(self, buffer, tensor):
        check(lib().gdpt_film_develop_device(self._h, buffer, self._device_tensor(tensor, "float32", 3 * self.rows * self.width, "develop_device")))
        return tensor

    def halo_bytes(self):
        n = C.c_size_t()
        check(lib().gdpt_film_halo_bytes(self._h, C.byref(n)))
        return n.value

    def pack_halo(self, which, tensor):
        check(lib().gdpt_film_pack_halo(self._h, which, self._device_tensor(tensor, "float64", self.halo_bytes() // 8, "pack_halo")))

    def unpack_halo(self, which, tensor):
        check(lib().gdpt_film_unpack_halo(self._h, which, self._device_tensor(tensor, "float64", self.halo_bytes() // 8, "unpack_halo")))

    def stats(self):
        s = (C.c_ulonglong * 4)()
        check(lib().gdpt_film_stats(self._h, s))
        return dict(raysTraced=int(s[0]), shadowRaysTraced=int(s[1]), paths=int(s[2]), pathLengthSum=int(s[3]))

    def invalid_puts(self):
        """Puts dropped by ImageBlock::put's validity check (non-finite, or negative outside dx/dy) since the last clear."""
        n = C.c_ulonglong(0)
        check(lib().gdpt_film_invalid_puts(self._h, C.byref(n)))
        return int(n.value)

    def render_ms(self):
        lib().gdpt_film_render_ms.restype = C.c_float
        return float(lib().gdpt_film_render_ms(self._h))

    def set_pipeline(self, stages, refill_lanes=0):
        """Device staging of a render (0 = one kernel; any other value = the default pipeline: primary pass + general kernel +
        continuation kernel -- the ABI treats 1 like 2, include/gdpt_tracer.h; 3 = the continuation phase starts in wavefront form, an
        opt-in experiment that is bit-identical to 2 and slower).  A tuning knob: results do not depend on it beyond rounding of the per-pixel sums."""
        check(lib().gdpt_film_set_pipeline(self._h, int(stages), int(refill_lanes)))

    def set_occupancy(self, waves_per_simd):
        check(lib().gdpt_film_set_occupancy(self._h, int(waves_per_simd)))

    def set_rfilter(self, kind, p0=0.0, p1=0.0):
        """`<rfilter>` of the film (scenes.RFILTER_*); box is the default and the fast path."""
        lib().gdpt_film_set_rfilter.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        check(lib().gdpt_film_set_rfilter(self._h, int(kind), float(p0), float(p1)))
        self.renders_own_border = int(kind) != 0

    def set_slices(self, slices):
        """Sample slices per launch (0 = chosen per launch); a tuning knob, see include/gdpt_tracer.h."""
        check(lib().gdpt_film_set_slices(self._h, int(slices)))

    def set_regeneration(self, idle_lanes):
        """Idle lanes of a wave before they start new samples together (1..64); a tuning knob."""
        check(lib().gdpt_film_set_regeneration(self._h, int(idle_lanes)))

    def sync(self):
        check(lib().gdpt_film_sync(self._h))

    def cancel(self):
        """Integrator::cancel: stop the frame being rendered into this film (callable from another thread); clear() starts a new one."""
        check(lib().gdpt_film_cancel(self._h))

    def cancelled(self):
        v = C.c_int()
        check(lib().gdpt_film_cancelled(self._h, C.byref(v)))
        return bool(v.value)

    def close(self):
        if self._h:
            lib().gdpt_film_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GradientPathIntegrator:
    """`<integrator type="gpt">` (gpt.cpp:1190-1213): same property names, defaults and errors."""

    def __init__(self, maxDepth=-1, minDepth=-1, rrDepth=5, strictNormals=False, shiftThreshold=0.001,
                 reconstructL1=True, reconstructL2=False, reconstructAlpha=0.2, hideEmitters=False):
        if reconstructL1 and reconstructL2:
            raise RuntimeError("Disable 'reconstructL1' or 'reconstructL2': Cannot display two reconstructions at a time!")  # gpt.cpp:1203
        if reconstructAlpha <= 0.0:
            raise RuntimeError("'reconstructAlpha' must be set to a value greater than zero!")                             # gpt.cpp:1206
        if maxDepth <= 0 and maxDepth != -1:
            raise RuntimeError("'maxDepth' must be set to -1 (infinite) or a value greater than zero!")                     # gpt.cpp:1209
        if hideEmitters:
            raise RuntimeError("Option 'hideEmitters' not implemented for Gradient-Domain Path Tracing!")                   # gpt.cpp:1362
        self.maxDepth, self.rrDepth, self.strictNormals, self.shiftThreshold = maxDepth, rrDepth, strictNormals, shiftThreshold
        self.minDepth = 1                                             # gpt.cpp:1369 overrides whatever was given
        self.reconstructL1, self.reconstructL2, self.reconstructAlpha = reconstructL1, reconstructL2, reconstructAlpha
        self.stats = {}

    def config(self, spp, seed=5489):
        return Config(self.maxDepth, self.rrDepth, int(self.strictNormals), spp, self.shiftThreshold, seed)

    def renderBlock(self, scene, film, cfg, rect):
        """GPTBlockRenderer::process + GPTRenderProcess::processResult for one rectangle (x0, y0, x1, y1)."""
        x0, y0, x1, y1 = rect
        check(lib().gdpt_render_rect(scene._h, C.byref(cfg), x0, y0, x1, y1, film._h))

    def renderSerial(self, scene, film, cfg, blockSize=32, parentSeed=5489):
        """The film as ONE worker of the reference samples it (`mitsuba -p 1`): spiral blocks, Hilbert order inside a block, every random number from one
        SFMT-19937 stream seeded as the worker's cloned sampler is (gdpt_render_serial).  One lane, synchronous: a validation path.  -> random numbers drawn."""
        draws = C.c_ulonglong(0)
        lib().gdpt_render_serial.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_ulonglong, C.c_void_p]
        check(lib().gdpt_render_serial(scene._h, C.byref(cfg), film._h, int(blockSize), int(parentSeed), C.byref(draws)))
        return int(draws.value)

    def render(self, scene, spp, seed=5489, film=None):
        """GradientPathIntegrator::render (gpt.cpp:1358-1480).  Returns {suffix: float32 [H,W,3]}."""
        own = film is None
        film = film or Film(scene)
        cfg = self.config(spp, seed)
        self.renderBlock(scene, film, cfg, (0, film.y0, scene.width, film.y1))
        film.sync()
        out = {name: film.develop(i) for i, name in enumerate(BUFFER_NAMES)}
        self.stats = film.stats()
        self.stats["render_ms"] = film.render_ms()
        if self.reconstructL1 or self.reconstructL2:                  # gpt.cpp:1415-1476
            preset = "L1D" if self.reconstructL1 else "L2D"
            rec = _poisson.reconstruct(out["-dx"].ravel(), out["-dy"].ravel(), out["-throughput"].ravel(), out["-direct"].ravel(),
                                       scene.width, film.rows, preset=preset, alpha=self.reconstructAlpha)
            out["-final"] = rec.reshape(film.rows, scene.width, 3)     # setBitmapMulti(reconstruction, 1, BUFFER_FINAL)
        if own:
            film.close()
        return out


def serial_random(seed, n, cloned=False):
    """n successive Random::nextULong outputs of Mitsuba's SFMT-19937 `Random(seed)` -- cloned: of a Random seeded from that one, as a worker's sampler is
    (host code of the library: no device needed)."""
    out = np.zeros(n, np.uint64)
    lib().gdpt_serial_random.argtypes = [C.c_ulonglong, C.c_int, C.c_int, C.c_void_p]
    check(lib().gdpt_serial_random(int(seed), int(bool(cloned)), int(n), out.ctypes.data_as(C.c_void_p)))
    return out


def serial_pixel_order(width, height, blockSize=32):
    """The pixels of a film in the order one worker of the reference renders them: [(x, y)] (host code of the library)."""
    xy = np.zeros((width * height, 2), np.int32)
    lib().gdpt_serial_pixel_order.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    check(lib().gdpt_serial_pixel_order(int(width), int(height), int(blockSize), xy.ctypes.data_as(C.c_void_p)))
    return xy


def write_pfm(path, rgb):
    """MultiFilm `fileFormat=pfm` output (multifilm.cpp:123-124, 223-235): little-endian float32, bottom row first."""
    a = np.asarray(rgb, np.float32)
    h, w, _ = a.shape
    with open(path, "wb") as f:
        f.write(("PF\n%d %d\n-1.0\n" % (w, h)).encode())
        f.write(a[::-1].astype("<f4").tobytes())


def write_buffers(dest, buffers):
    """MultiFilm::develop naming: `<dest><suffix>.pfm` (multifilm.cpp:453-517)."""
    paths = []
    for suffix, img in buffers.items():
        p = dest + suffix + ".pfm"
        write_pfm(p, img)
        paths.append(p)
    return paths
