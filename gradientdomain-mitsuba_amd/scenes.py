"""Build-authored scenes (the reference ships none: SURVEY.md 8d) as flat arrays for the tracer C-ABI.

A scene is what `gdpt_scene_create` (include/gdpt_tracer.h) takes: a triangle soup (ntri x 9 float64, no vertex
normals/texcoords), one material index per triangle, a material table, area emitters given as contiguous triangle
ranges (one emitter per emissive mesh, as in Mitsuba where an `area` emitter is attached to a shape), and a
perspective sensor.  One-sided BSDFs only (the reference's diffuse/conductor/roughconductor are front-side), so
every face is wound so that cross(p1-p0, p2-p0) points into the space light arrives from.
"""
import math
from dataclasses import dataclass, field

import numpy as np

MAT_DIFFUSE, MAT_CONDUCTOR, MAT_ROUGHCONDUCTOR, MAT_DIELECTRIC = 0, 1, 2, 3
DISTR_BECKMANN, DISTR_GGX, DISTR_PHONG = 0, 1, 2
# reconstruction filters (src/rfilters): (kind, p0, p1) with the reference's default parameters
RFILTER_BOX, RFILTER_TENT, RFILTER_GAUSSIAN, RFILTER_MITCHELL, RFILTER_CATMULLROM, RFILTER_LANCZOS = range(6)
TEXWRAP_REPEAT, TEXWRAP_CLAMP, TEXWRAP_MIRROR, TEXWRAP_ZERO, TEXWRAP_ONE = range(5)      # bitmap.cpp:324-338
TEXFILTER_NEAREST, TEXFILTER_BILINEAR, TEXFILTER_TRILINEAR, TEXFILTER_EWA = 0, 1, 2, 3
RFILTER_DEFAULTS = {0: (0, 0.0, 0.0), 1: (1, 0.0, 0.0), 2: (2, 0.5, 0.0), 3: (3, 1.0 / 3.0, 1.0 / 3.0), 4: (4, 0.0, 0.0), 5: (5, 3.0, 0.0)}


def diffuse(rgb):
    return dict(type=MAT_DIFFUSE, reflectance=tuple(rgb))


def conductor(eta, k, specular=(1.0, 1.0, 1.0)):
    return dict(type=MAT_CONDUCTOR, reflectance=tuple(specular), eta=tuple(eta), k=tuple(k))


def roughconductor(alpha, eta, k, specular=(1.0, 1.0, 1.0), distribution=DISTR_BECKMANN, alphaV=None, sampleVisible=True):
    return dict(type=MAT_ROUGHCONDUCTOR, reflectance=tuple(specular), eta=tuple(eta), k=tuple(k), alphaU=float(alpha),
                alphaV=float(alpha if alphaV is None else alphaV), distribution=distribution, sampleVisible=int(sampleVisible))


def dielectric(int_ior=float(np.float32(1.5046)), ext_ior=float(np.float32(1.000277)), specular_reflectance=(1.0, 1.0, 1.0), specular_transmittance=(1.0, 1.0, 1.0)):
    """`dielectric` (reference src/bsdfs/dielectric.cpp); defaults = bk7 glass in air (src/bsdfs/ior.h, float literals)."""
    eta = int_ior / ext_ior
    return dict(type=MAT_DIELECTRIC, reflectance=tuple(specular_reflectance), eta=(eta, eta, eta), k=tuple(specular_transmittance))


def twosided(inner):
    """`twosided` around a one-sided BRDF (reference src/bsdfs/twosided.cpp): the same model on both faces."""
    m = dict(inner)
    m["twoSided"] = 1
    return m


# measured-looking constants for copper/aluminium at RGB wavelengths (scene authors pass explicit eta/k; data/ior is not needed)
CU = dict(eta=(0.200438, 0.924033, 1.102212), k=(3.912949, 2.452848, 2.142188))
AL = dict(eta=(1.657460, 0.880369, 0.521229), k=(9.223869, 6.269523, 4.837001))


def lookat(origin, target, up):
    """Transform::lookAt (reference src/libcore/transform.cpp:191-214): columns = left, newUp, dir, origin."""
    p, t, u = (np.asarray(v, np.float64) for v in (origin, target, up))
    d = t - p
    d /= np.linalg.norm(d)
    left = np.cross(u, d)
    left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, p
    return m


@dataclass
class Scene:
    verts: np.ndarray            # (ntri, 9) float64
    tri_material: np.ndarray     # (ntri,) int32
    materials: list
    emitters: list               # [(firstTri, numTris, (r,g,b))] area lights, [(firstTri, 2, (r,g,b), toWorld 3x4, normal)] the light of a `rectangle` shape
                                 # (sampled as the shape samples itself) and [("point", (x,y,z), (r,g,b) intensity)], in scene order
    to_world: np.ndarray         # 4x4 camera-to-world
    fov_x: float
    near: float = 1e-2
    far: float = 1e4
    width: int = 512
    height: int = 512
    name: str = "scene"
    environment: tuple = None    # ((r, g, b), position in the scene's emitter list) for `<emitter type="constant">`, or None
    rfilter: tuple = None        # (kind, p0, p1) of the film's reconstruction filter (RFILTER_*); None = box
    normals: np.ndarray = None   # (ntri, 9) per-vertex normals (TriMesh vertex normals), all-zero rows = flat triangle; or None
    uvs: np.ndarray = None       # (ntri, 6) per-vertex texture coordinates u0 v0 u1 v1 u2 v2; or None (its.uv = the hit's barycentrics)
    tri_has_uv: np.ndarray = None   # (ntri,) 1 = the triangle's mesh has texture coordinates (None = all, when uvs is given)
    environment_map: dict = None  # `<emitter type="envmap">`: dict(rgb [h, w, 3] linear latitude-longitude map (top row = up), scale, toWorld 3x3, index); excludes `environment`
    textures: list = None        # bitmap textures: dicts with rgb [h, w, 3] (linear), wrapU/wrapV (TEXWRAP_*), filter (TEXFILTER_*), uscale, vscale, uoffset, voffset, scale
    material_textures: list = None   # per material: texture index on its reflectance / specularReflectance, -1 = constant
    thinlens: tuple = None       # (apertureRadius, focusDistance) of `<sensor type="thinlens">`; None = `perspective`
    shutter: tuple = None        # (shutterOpen, shutterClose) of the sensor (sensor.cpp:26-38); None = (0, 0): no time sample
    crop: tuple = None           # (cropOffsetX, cropOffsetY, fullWidth, fullHeight): width x height above is the crop window of a film of that size (film.cpp:34-48); None = no crop

    @property
    def ntri(self):
        return int(self.verts.shape[0])


class _Builder:
    def __init__(self):
        self.tris, self.mat_ids, self.materials, self.emitters = [], [], [], []
        self.normals = {}            # triangle index -> 9 floats

    def sphere(self, centre, radius, mat, level=1, bend=0.0, seed=0):
        """Icosphere with per-vertex normals (position - centre, optionally bent by up to `bend` radians so that shading
        and geometric normals disagree noticeably)."""
        t = (1 + 5 ** 0.5) / 2
        v = [np.array(q, np.float64) / np.linalg.norm(q) for q in
             [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]]
        f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
             (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
        for _ in range(level):
            nf = []
            for (a, b, c) in f:
                ab, bc, ca = [x / np.linalg.norm(x) for x in (v[a] + v[b], v[b] + v[c], v[c] + v[a])]
                i = len(v); v += [ab, bc, ca]
                nf += [(a, i, i + 2), (b, i + 1, i), (c, i + 2, i + 1), (i, i + 1, i + 2)]
            f = nf
        rng = np.random.default_rng(seed)
        nrm = []
        for q in v:
            n = q.copy()
            if bend > 0:
                d = rng.normal(size=3); d -= n * np.dot(d, n); d /= np.linalg.norm(d)
                a = bend * rng.random()
                n = n * np.cos(a) + d * np.sin(a)
            nrm.append(n / np.linalg.norm(n))
        c = np.asarray(centre, np.float64)
        for (a, b, cc) in f:
            self.normals[len(self.tris)] = np.concatenate([nrm[a], nrm[b], nrm[cc]])
            self.tris.append(np.concatenate([c + radius * v[a], c + radius * v[b], c + radius * v[cc]]))
            self.mat_ids.append(mat)

    def material(self, m):
        self.materials.append(m)
        return len(self.materials) - 1

    def quad(self, a, b, c, d, mat, toward):
        """Two triangles (a,b,c),(a,c,d), re-wound so the face normal has a positive dot with `toward - centroid`."""
        q = [np.asarray(v, np.float64) for v in (a, b, c, d)]
        n = np.cross(q[1] - q[0], q[2] - q[0])
        if np.dot(n, np.asarray(toward, np.float64) - (q[0] + q[1] + q[2] + q[3]) / 4) < 0:
            q = [q[0], q[3], q[2], q[1]]
        self.tris += [np.concatenate([q[0], q[1], q[2]]), np.concatenate([q[0], q[2], q[3]])]
        self.mat_ids += [mat, mat]

    def box(self, top4, height_y0, mat):
        """Cornell-style block: `top4` = top face corners (counter-clockwise seen from above), extruded down to y0."""
        top = [np.asarray(v, np.float64) for v in top4]
        bot = [np.array([v[0], height_y0, v[2]]) for v in top]
        c = sum(top + bot) / 8
        far = lambda pts: c + 1e3 * (sum(pts) / len(pts) - c)   # a point outside the block beyond that face
        self.quad(*top, mat, far(top))
        for i in range(4):
            j = (i + 1) % 4
            self.quad(top[i], top[j], bot[j], bot[i], mat, far([top[i], top[j], bot[j], bot[i]]))

    def emitter(self, first_tri, num_tris, radiance):
        self.emitters.append((first_tri, num_tris, tuple(radiance)))

    def finish(self, **cam):
        sc = Scene(np.asarray(self.tris, np.float64).reshape(-1, 9), np.asarray(self.mat_ids, np.int32),
                   self.materials, self.emitters, **cam)
        if self.normals:
            sc.normals = np.zeros((len(self.tris), 9), np.float64)
            for i, n in self.normals.items():
                sc.normals[i] = n
        return sc


def _random_material(rng):
    """One material drawn from everything the path carries (fuzz tests)."""
    kind = int(rng.integers(0, 6))
    metal = {"eta": tuple(float(v) for v in rng.uniform(0.1, 2.5, 3)), "k": tuple(float(v) for v in rng.uniform(1.5, 7.0, 3))}
    if kind == 0:
        m = diffuse(tuple(float(v) for v in rng.uniform(0.05, 0.9, 3)))
    elif kind == 1:
        m = conductor(**metal)
    elif kind in (2, 3):
        au = float(10.0 ** rng.uniform(-3.3, -0.4))                      # straddles shiftThreshold = 1e-3
        m = roughconductor(au, **metal, distribution=int(rng.choice([DISTR_GGX, DISTR_BECKMANN, DISTR_PHONG])),
                           alphaV=float(au * rng.uniform(0.3, 3.0)) if rng.random() < 0.5 else None, sampleVisible=bool(rng.random() < 0.7))
    elif kind == 4:
        m = dielectric(int_ior=float(rng.uniform(1.2, 2.4)), ext_ior=float(rng.uniform(1.0, 1.1)))
    else:
        m = diffuse(tuple(float(v) for v in rng.uniform(0.05, 0.9, 3)))
    if m["type"] != 3 and rng.random() < 0.3:
        m = twosided(m)
    return m


def _random_connectable_material(rng):
    """One material from what the G-BDPT path carries: every vertex connectable (smooth, roughness well above any shiftThreshold in use)."""
    if rng.random() < 0.4:
        m = diffuse(tuple(float(v) for v in rng.uniform(0.05, 0.9, 3)))
    else:
        metal = {"eta": tuple(float(v) for v in rng.uniform(0.1, 2.5, 3)), "k": tuple(float(v) for v in rng.uniform(1.5, 7.0, 3))}
        au = float(10.0 ** rng.uniform(-1.3, -0.3))
        m = roughconductor(au, **metal, distribution=int(rng.choice([DISTR_GGX, DISTR_BECKMANN, DISTR_PHONG])),
                           alphaV=float(au * rng.uniform(0.5, 2.0)) if rng.random() < 0.5 else None, sampleVisible=bool(rng.random() < 0.7))
    if rng.random() < 0.3:
        m = twosided(m)
    return m


def cornell_box(width=512, height=512, variant="diffuse", seed=0, environment=None, point_light=None):
    """The Cornell box (Cornell Program of Computer Graphics measurement data, 555-unit room), all triangle meshes:
    5 walls, short block, tall block, one area-light quad.  variant: "diffuse" (BASELINE configs 1-2) |
    "glossy" (rough-copper floor, mirror back wall, GGX block: exercises the half-vector shift) | "nearspecular"."""
    b = _Builder()
    white = b.material(diffuse((0.725, 0.71, 0.68)))
    red = b.material(diffuse((0.63, 0.065, 0.05)))
    green = b.material(diffuse((0.14, 0.45, 0.091)))
    lightm = b.material(diffuse((0.78, 0.78, 0.78)))
    floor_m = back_m = white
    if variant == "glossy":           # rough-copper floor, aluminium mirror back wall, GGX block (the box front is open: blocks would mirror the void)
        floor_m = b.material(roughconductor(0.1, **CU))
        back_m = b.material(conductor(**AL))
        tall_m = b.material(roughconductor(0.05, **AL, distribution=DISTR_GGX))
        short_m = white
    elif variant == "rough":          # every BSDF smooth and rougher than shiftThreshold (G-BDPT's connectable-only scope): rough copper floor,
        floor_m = b.material(roughconductor(0.1, **CU))                 # anisotropic rough aluminium back wall, GGX block, Phong-distribution short block
        back_m = b.material(roughconductor(0.2, **AL, alphaV=0.05))
        tall_m = b.material(roughconductor(0.05, **AL, distribution=DISTR_GGX))
        short_m = b.material(twosided(roughconductor(0.3, **CU, distribution=DISTR_PHONG, sampleVisible=False)))
    elif variant == "nearspecular":   # roughness <= shiftThreshold (0.001): a glossy-sampled vertex that is classified GLOSSY
        floor_m = b.material(roughconductor(0.0008, **CU))
        back_m = b.material(roughconductor(0.2, **AL, alphaV=0.05))
        tall_m = short_m = white
    elif variant == "glass":          # the tall block is solid bk7 glass, the short one an aluminium mirror: refraction branch of the half-vector shift
        tall_m = b.material(dielectric())
        short_m = b.material(conductor(**AL))
    elif variant == "slab":           # "glass" with an EXACT rectangular glass block (the measured tall block's opposite faces are parallel to 1e-4 only): a slab
        tall_m = b.material(dielectric())
        short_m = white
    elif variant == "mirrors":        # two mirrors facing each other at an angle: the back wall and the tall block (chains with two specular vertices)
        back_m = b.material(conductor(**AL))
        tall_m = b.material(conductor(**AL))
        short_m = white
    elif variant == "twosided":       # two-sided walls and a free-standing two-sided GGX panel lit and seen from both faces
        white = b.material(twosided(diffuse((0.725, 0.71, 0.68))))
        floor_m = back_m = tall_m = short_m = white
    elif variant in ("smooth", "bent"):  # spheres with interpolated shading normals instead of the blocks ("bent": normals tilted up to 0.6 rad)
        tall_m = b.material(roughconductor(0.12, **AL, distribution=DISTR_GGX))
        short_m = white
    elif variant == "random_connectable":   # fuzz for G-BDPT: the same four surfaces, connectable materials only
        rng = np.random.default_rng(seed)
        floor_m, back_m, tall_m, short_m = (b.material(_random_connectable_material(rng)) for _ in range(4))
    elif variant == "random":         # fuzz: floor, back wall and both blocks draw their materials from `seed`
        rng = np.random.default_rng(seed)
        floor_m, back_m, tall_m, short_m = (b.material(_random_material(rng)) for _ in range(4))
    else:
        tall_m = short_m = white
    room = (278.0, 274.4, 279.6)
    b.quad((552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2), floor_m, room)             # floor
    b.quad((556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0), white, room)   # ceiling
    b.quad((549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2), back_m, room)  # back wall
    b.quad((0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2), green, room)                # right wall
    b.quad((552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0), red, room)      # left wall
    if variant in ("smooth", "bent"):
        b.sphere((186, 90, 169), 90.0, short_m, level=1, bend=0.6 if variant == "bent" else 0.0, seed=1)
        b.sphere((368, 150, 351), 150.0, tall_m, level=1, bend=0.6 if variant == "bent" else 0.0, seed=2)
    else:
        b.box([(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)], 0.0, short_m)         # short block
        if variant == "slab":          # 200 x 60 footprint about (368, 351), turned by atan(3/4): corners are exact in binary floating point to the last bit or two
            cx, cz, ux, uz = 368.0, 351.0, 0.8, 0.6
            corner = lambda su, sv: (cx + su * 100 * ux - sv * 30 * uz, 330, cz + su * 100 * uz + sv * 30 * ux)
            b.box([corner(1, -1), corner(-1, -1), corner(-1, 1), corner(1, 1)], 0.0, tall_m)
        else:
            b.box([(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)], 0.0, tall_m)        # tall block
    if variant == "twosided":
        panel = b.material(twosided(roughconductor(0.15, **AL, distribution=DISTR_GGX)))
        b.quad((60, 20, 150), (200, 20, 60), (200, 300, 60), (60, 300, 150), panel, room)
    first = len(b.tris)
    b.quad((343, 548.3, 227), (343, 548.3, 332), (213, 548.3, 332), (213, 548.3, 227), lightm, room)  # light, just below the ceiling
    b.emitter(first, 2, (17.0, 12.0, 4.0))
    fov = 2 * math.degrees(math.atan(0.0125 / 0.035))      # 0.025 sensor, 0.035 focal length; `fov` is the x-fov (fovAxis = x), so a
                                                            # wide film crops the square original top and bottom and every pixel sees the box
    sc = b.finish(to_world=lookat((278, 273, -800), (278, 273, -799), (0, 1, 0)), fov_x=fov, near=10.0, far=2800.0,
                  width=width, height=height, name="cornell-" + variant)
    if point_light is not None:        # ((x, y, z), (r, g, b) intensity[, first]): a `point` emitter after (or before) the area light
        pl = ("point", tuple(float(c) for c in point_light[0]), tuple(float(c) for c in point_light[1]))
        sc.emitters = [pl] + sc.emitters if len(point_light) > 2 and point_light[2] else sc.emitters + [pl]
    if environment is not None:        # a constant environment emitter seen through the open front and in the wide film's margins
        sc.environment = (tuple(float(v) for v in environment), 0 if variant == "random" and seed % 2 else len(sc.emitters))
    return sc


def atrium(width=1920, height=1080, columns=24, segments=48, seed=7):
    """Sponza-class stand-in (the Sponza asset is not available offline): a two-storey colonnaded atrium, procedurally
    tessellated -- faceted columns, arches, floor tiles, 80 % diffuse / 20 % rough-conductor (Beckmann alpha 0.1), lit by
    an opening-sized area light.  `columns` x `segments` set the triangle count (~columns*segments*2*rings + tiles)."""
    rng = np.random.default_rng(seed)
    b = _Builder()
    stone = [b.material(diffuse(c)) for c in ((0.62, 0.56, 0.47), (0.55, 0.5, 0.43), (0.7, 0.66, 0.6), (0.45, 0.32, 0.25))]
    metal = b.material(roughconductor(0.1, **CU))
    lightm = b.material(diffuse((0.5, 0.5, 0.5)))
    L, Wd, Hh = 40.0, 16.0, 14.0
    room = (0.0, Hh / 2, 0.0)
    tiles = 48
    for i in range(tiles):                                   # floor tiles (each its own quad; some metallic inlays)
        for j in range(tiles // 2):
            x0, x1 = -L / 2 + L * i / tiles, -L / 2 + L * (i + 1) / tiles
            z0, z1 = -Wd / 2 + Wd * j / (tiles // 2), -Wd / 2 + Wd * (j + 1) / (tiles // 2)
            m = metal if rng.random() < 0.2 else stone[(i + j) % 3]
            b.quad((x0, 0, z0), (x1, 0, z0), (x1, 0, z1), (x0, 0, z1), m, room)
    b.quad((-L / 2, 0, -Wd / 2), (-L / 2, Hh, -Wd / 2), (L / 2, Hh, -Wd / 2), (L / 2, 0, -Wd / 2), stone[1], room)
    b.quad((-L / 2, 0, Wd / 2), (-L / 2, Hh, Wd / 2), (L / 2, Hh, Wd / 2), (L / 2, 0, Wd / 2), stone[1], room)
    b.quad((-L / 2, 0, -Wd / 2), (-L / 2, Hh, -Wd / 2), (-L / 2, Hh, Wd / 2), (-L / 2, 0, Wd / 2), stone[3], room)
    b.quad((L / 2, 0, -Wd / 2), (L / 2, Hh, -Wd / 2), (L / 2, Hh, Wd / 2), (L / 2, 0, Wd / 2), stone[3], room)
    b.quad((-L / 2, Hh, -Wd / 2), (L / 2, Hh, -Wd / 2), (L / 2, Hh, Wd / 2), (-L / 2, Hh, Wd / 2), stone[2], room)
    rings = 24
    for ci in range(columns):                                # faceted columns in two rows, two storeys
        side = -1 if ci % 2 == 0 else 1
        cx = -L / 2 + L * (ci // 2 + 0.5) / (columns // 2)
        cz = side * (Wd / 2 - 2.5)
        m = metal if rng.random() < 0.2 else stone[ci % 3]
        for storey in range(2):
            ybase = storey * (Hh / 2)
            for r in range(rings):
                y0, y1 = ybase + (Hh / 2 - 0.6) * r / rings, ybase + (Hh / 2 - 0.6) * (r + 1) / rings
                rad0 = 0.55 + 0.06 * math.sin(3.1 * y0)
                rad1 = 0.55 + 0.06 * math.sin(3.1 * y1)
                for s in range(segments):
                    a0, a1 = 2 * math.pi * s / segments, 2 * math.pi * (s + 1) / segments
                    p = lambda rad, a, y: (cx + rad * math.cos(a), y, cz + rad * math.sin(a))
                    q = [p(rad0, a0, y0), p(rad0, a1, y0), p(rad1, a1, y1), p(rad1, a0, y1)]
                    mid = (cx + 10 * math.cos((a0 + a1) / 2), (y0 + y1) / 2, cz + 10 * math.sin((a0 + a1) / 2))
                    b.quad(*q, m, mid)
    first = len(b.tris)
    b.quad((-6, Hh - 0.05, -3), (6, Hh - 0.05, -3), (6, Hh - 0.05, 3), (-6, Hh - 0.05, 3), lightm, room)
    b.emitter(first, 2, (7.0, 6.5, 5.5))
    return b.finish(to_world=lookat((-17.0, 3.2, 0.6), (0.0, 4.5, 0.0), (0, 1, 0)), fov_x=70.0, near=0.1, far=200.0,
                    width=width, height=height, name="atrium")


def veach_bidir(width=1280, height=720, specular=False):
    """Veach-bidir-class stand-in (BASELINE config 5; the original asset is not available offline): a closed room lit almost entirely
    INDIRECTLY: an up-light inside a shade that only lets light reach the ceiling, and a wall sconce that faces the wall behind it; a table
    with a smooth rough-aluminium sphere (interpolated normals), an egg-sized sphere and a rough-copper block.
    specular=False (rounds 1-3): every BSDF smooth with roughness far above shiftThreshold, the egg diffuse -- no specular chains.
    specular=True (round 4, what the scene's name stands for): the egg is solid GLASS (bk7, interpolated normals), a framed MIRROR hangs on the
    back wall and the block is polished copper (a perfect conductor): caustics under the egg, the room seen in the mirror -- the paths that need
    G-BDPT's manifold walks."""
    b = _Builder()
    wall = b.material(diffuse((0.72, 0.7, 0.66)))
    floor_m = b.material(diffuse((0.45, 0.3, 0.18)))
    wood = b.material(diffuse((0.5, 0.33, 0.2)))
    shade = b.material(twosided(diffuse((0.8, 0.75, 0.6))))
    alu = b.material(roughconductor(0.15, **AL, distribution=DISTR_GGX))
    copper = b.material(roughconductor(0.1, **CU))
    egg = b.material(dielectric() if specular else diffuse((0.75, 0.75, 0.7)))
    if specular:
        copper = b.material(conductor(**CU))
        mirror = b.material(conductor(**AL))
    lightm = b.material(diffuse((0.5, 0.5, 0.5)))
    room = (0.0, 1.5, 0.0)
    X, Y, Z = 4.0, 3.0, 3.0
    b.quad((-X, 0, -Z), (X, 0, -Z), (X, 0, Z), (-X, 0, Z), floor_m, room)
    b.quad((-X, Y, -Z), (X, Y, -Z), (X, Y, Z), (-X, Y, Z), wall, room)
    b.quad((-X, 0, Z), (X, 0, Z), (X, Y, Z), (-X, Y, Z), wall, room)
    b.quad((-X, 0, -Z), (X, 0, -Z), (X, Y, -Z), (-X, Y, -Z), wall, room)
    b.quad((-X, 0, -Z), (-X, 0, Z), (-X, Y, Z), (-X, Y, -Z), wall, room)
    b.quad((X, 0, -Z), (X, 0, Z), (X, Y, Z), (X, Y, -Z), wall, room)
    b.box([(-1.5, 1.0, -0.8), (-1.5, 1.0, 0.8), (1.5, 1.0, 0.8), (1.5, 1.0, -0.8)], 0.9, wood)              # table top
    for (lx, lz) in ((-1.4, -0.7), (-1.4, 0.7), (1.4, -0.7), (1.4, 0.7)):                                     # legs
        b.box([(lx - 0.05, 0.9, lz - 0.05), (lx - 0.05, 0.9, lz + 0.05), (lx + 0.05, 0.9, lz + 0.05), (lx + 0.05, 0.9, lz - 0.05)], 0.0, wood)
    b.sphere((-0.6, 1.3, 0.1), 0.3, alu, level=2)
    b.sphere((0.25, 1.2, -0.3), 0.2, egg, level=2)
    b.box([(0.7, 1.35, 0.0), (0.7, 1.35, 0.4), (1.1, 1.35, 0.4), (1.1, 1.35, 0.0)], 1.0, copper)
    if specular:                                   # a mirror on the wall the camera faces, 2 cm in front of it
        b.quad((-2.2, 0.9, Z - 0.02), (0.6, 0.9, Z - 0.02), (0.6, 2.4, Z - 0.02), (-2.2, 2.4, Z - 0.02), mirror, room)
    # up-light: emitter facing up inside a four-sided shade (two-sided diffuse), open at the top only
    cx, cy, cz, r, hgt = -2.6, 1.9, 0.8, 0.25, 0.5
    for (a0, a1) in (((cx - r, cz - r), (cx + r, cz - r)), ((cx + r, cz - r), (cx + r, cz + r)), ((cx + r, cz + r), (cx - r, cz + r)), ((cx - r, cz + r), (cx - r, cz - r))):
        b.quad((a0[0], cy - 0.05, a0[1]), (a1[0], cy - 0.05, a1[1]), (a1[0], cy + hgt, a1[1]), (a0[0], cy + hgt, a0[1]), shade, (cx, cy, cz))
    b.quad((cx - r, cy - 0.05, cz - r), (cx + r, cy - 0.05, cz - r), (cx + r, cy - 0.05, cz + r), (cx - r, cy - 0.05, cz + r), shade, (cx, cy + 1, cz))
    first = len(b.tris)
    b.quad((cx - 0.15, cy, cz - 0.15), (cx + 0.15, cy, cz - 0.15), (cx + 0.15, cy, cz + 0.15), (cx - 0.15, cy, cz + 0.15), lightm, (cx, cy + 1, cz))
    b.emitter(first, 2, (60.0, 52.0, 40.0))
    # wall sconce: a small panel 10 cm in front of the right wall, emitting TOWARDS the wall; its back is a diffuse plate
    first = len(b.tris)
    b.quad((3.9, 1.6, -1.2), (3.9, 2.0, -1.2), (3.9, 2.0, -0.8), (3.9, 1.6, -0.8), lightm, (5.0, 1.8, -1.0))
    b.emitter(first, 2, (40.0, 40.0, 46.0))
    b.quad((3.88, 1.55, -1.25), (3.88, 2.05, -1.25), (3.88, 2.05, -0.75), (3.88, 1.55, -0.75), shade, (0.0, 1.8, -1.0))
    return b.finish(to_world=lookat((0.4, 1.75, -2.75), (-0.1, 1.1, 0.1), (0, 1, 0)), fov_x=62.0, near=0.05, far=50.0,
                    width=width, height=height, name="veach-bidir-class")


def bitmap_texture(rgb, wrap=TEXWRAP_REPEAT, filter=TEXFILTER_BILINEAR, uscale=1.0, vscale=1.0, uoffset=0.0, voffset=0.0, wrapV=None, conserve=True, maxAnisotropy=20.0):
    """`<texture type="bitmap">` with filterType nearest | bilinear | trilinear | ewa (reference src/textures/bitmap.cpp; ewa is its default).  rgb: [h, w, 3] LINEAR values,
    top row first.  conserve: BSDF::ensureEnergyConservation (src/librender/bsdf.cpp) -- a reflectance texture whose maximum exceeds 1
    is scaled by 0.99 / max."""
    rgb = np.ascontiguousarray(rgb, np.float64)
    mx = float(rgb.max())
    scale = float(np.float32(0.99)) * (1.0 / mx) if (conserve and mx > 1.0) else 1.0      # bsdf.cpp:96: 0.99f * (max / actualMax)
    return dict(rgb=rgb, wrapU=wrap, wrapV=wrap if wrapV is None else wrapV, filter=filter, uscale=uscale, vscale=vscale, uoffset=uoffset, voffset=voffset, scale=scale,
                maxAnisotropy=float(maxAnisotropy))


def checker_rgb(w=16, h=12, seed=3):
    """A small test bitmap: coloured checker with per-texel noise (so that every texel, wrap mode and the bilinear weights matter)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.where(((xx // 2 + yy // 2) % 2)[..., None] == 0, np.array([0.75, 0.2, 0.15]), np.array([0.15, 0.55, 0.8]))
    return np.clip(base + 0.15 * rng.random((h, w, 3)), 0.0, 1.0)


def textured_cornell_box(width=64, height=48, filter=TEXFILTER_BILINEAR, wrap=TEXWRAP_REPEAT, seed=0, uvscale=1.0, maxAnisotropy=20.0, size=(16, 12)):
    """The Cornell box with bitmap textures: the floor (texture coordinates tiled 3 x 2.5 over it, so the wrap mode shows), the back wall
    (a mesh WITHOUT texture coordinates: its.uv = the barycentrics) and a rough-copper short block whose specularReflectance is textured."""
    sc = cornell_box(width, height, "diffuse")
    nt = sc.ntri
    floor_m = len(sc.materials); sc.materials.append(diffuse((0.5, 0.5, 0.5)))
    back_m = len(sc.materials); sc.materials.append(diffuse((0.5, 0.5, 0.5)))
    block_m = len(sc.materials); sc.materials.append(roughconductor(0.15, **CU))
    tm = np.array(sc.tri_material, np.int32).copy()
    tm[0:2] = floor_m; tm[4:6] = back_m; tm[10:20] = block_m              # floor, back wall, the short block (10 triangles after the 5 walls)
    sc.tri_material = tm
    uvs = np.zeros((nt, 6)); has = np.zeros(nt, np.uint8)
    v = np.asarray(sc.verts).reshape(nt, 3, 3)
    for t in (0, 1):                                                      # floor: uv from x and z, running past [0, 1]
        for j in range(3):
            uvs[t, 2 * j] = v[t, j, 0] / 552.8 * 3.0 - 0.7
            uvs[t, 2 * j + 1] = v[t, j, 2] / 559.2 * 2.5 - 0.4
        has[t] = 1
    for t in range(10, 20):                                               # the block: uv from x + y and z
        for j in range(3):
            uvs[t, 2 * j] = (v[t, j, 0] + v[t, j, 1]) / 200.0
            uvs[t, 2 * j + 1] = v[t, j, 2] / 150.0
        has[t] = 1
    sc.uvs, sc.tri_has_uv = uvs, has
    sc.textures = [bitmap_texture(checker_rgb(size[0], size[1], seed), wrap=wrap, filter=filter, uscale=uvscale, vscale=uvscale, maxAnisotropy=maxAnisotropy),
                   bitmap_texture(checker_rgb(7, 5, seed + 1) * 1.3, wrap=TEXWRAP_MIRROR, filter=filter, uscale=2.0 * uvscale, vscale=3.0 * uvscale, uoffset=0.25, voffset=-0.5,
                                  maxAnisotropy=maxAnisotropy),
                   bitmap_texture(checker_rgb(9, 9, seed + 2), wrap=TEXWRAP_CLAMP, wrapV=TEXWRAP_ONE, filter=TEXFILTER_NEAREST)]
    mt = [-1] * len(sc.materials)
    mt[floor_m], mt[back_m], mt[block_m] = 0, 1, 2
    sc.material_textures = mt
    sc.name = "cornell-textured"
    return sc


def sky_map(w=32, h=16, seed=7, sun=40.0):
    """A small latitude-longitude test environment: bluish gradient sky with per-texel noise, a darker ground half and a very bright
    sun patch (so that the importance sampling and the half-precision storage both matter)."""
    rng = np.random.default_rng(seed)
    v = (np.arange(h) + 0.5) / h
    img = np.zeros((h, w, 3))
    img[...] = np.where(v[:, None, None] < 0.5, np.array([0.35, 0.55, 0.9]) * (1.2 - v[:, None, None]), np.array([0.12, 0.1, 0.08]))
    img *= rng.uniform(0.7, 1.3, (h, w, 1))
    img[h // 5: h // 5 + 2, w // 3: w // 3 + 2] = sun * np.array([1.0, 0.9, 0.7])
    return img
