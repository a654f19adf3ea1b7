/*
 * oracle/gpt_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (C++17, double precision like the reference's DOUBLE_PRECISION build,
 * single-threaded, brute-force ray casting) of the reference's G-PT per-sample hot path:
 * GradientPathTracer::evaluatePoint/evaluate, the shift mappings, vertex classification, the 15-put
 * accumulation of GradientPathIntegrator::renderBlock, and the Mitsuba pieces those call for the
 * scene subset the build carries (triangle soups with optional per-vertex normals and texture coordinates (UV tangents), area /
 * rectangle / point emitters, the constant environment and the environment MAP (envmap.cpp), diffuse / conductor / roughconductor
 * (Beckmann, GGX, Phong) / dielectric BSDFs and the twosided adapter, bitmap textures with all four filter types (mipmap_oracle.hpp),
 * perspective sensor with ray differentials, the six reconstruction filters).  GPO_TRACE_MAIN=1 in the environment
 * makes evaluate() print its main path and light-sample decisions (tools/gpu_fuzz_locate.py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED.  The reference tracer cannot be built here (mitsuba.h needs boost; scene loading
 * Xerces-C; films OpenEXR -- none in the image, none may be stood in for) and holds no test or fixture
 * for gpt.  Nothing in this file has been compared with output of the
 * reference.  What checks it (tests/test_gpt_oracle.py): closed-form KATs of the shifts/BSDFs,
 * furnace/energy identities, sample-vs-pdf consistency of the BSDF restatements, convergence of
 * `-throughput` to an independent plain path tracer written against the rendering equation.
 *
 * One deliberate, stated deviation: the reference draws random numbers from one serial SFMT stream
 * in spiral-block x Hilbert-pixel order (independent.cpp:82-103, random.cpp); a GPU cannot consume
 * that, so BOTH this oracle and the HIP path use the same counter-based generator keyed by
 * (seed, pixel, sample) -- see Rng below.  The consumption ORDER within a sample follows the
 * reference.
 *
 * Citations are relative to /root/reference/.  "gpt.cpp" = src/integrators/gpt/gpt.cpp.
 */
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <limits>
#include <vector>

#include "sfmt_random.hpp"
#include "mipmap_oracle.hpp"

#define GPO_API extern "C" __attribute__((visibility("default")))

static const bool g_traceMain = std::getenv("GPO_TRACE_MAIN") != nullptr;      // read once: the hot loops only test a bool

namespace {

typedef double Float;
const Float Epsilon = 1e-7;               // include/mitsuba/core/constants.h:25 (double build)
const Float ShadowEpsilon = 1e-5;         // constants.h:26
const Float DeltaEpsilon = (Float)1e-3f;  // constants.h:31
const Float D_EPSILON = (Float)(1e-14);   // gpt.cpp:63
const Float PI = 3.14159265358979323846;
const Float INV_PI = 0.31830988618379067154;
const Float INF = std::numeric_limits<Float>::infinity();

struct V3 {
    Float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(Float a) : x(a), y(a), z(a) {}
    V3(Float a, Float b, Float c) : x(a), y(b), z(c) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, Float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(Float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, Float s) { Float r = 1.0 / s; return V3(a.x * r, a.y * r, a.z * r); } // TVector3::operator/ multiplies by the reciprocal
inline V3 divc(V3 a, V3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline Float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline Float lengthSquared(V3 a) { return dot(a, a); }
inline Float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a / length(a); }
inline bool isZero(V3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
inline Float maxc(V3 a) { return std::max(a.x, std::max(a.y, a.z)); }
inline Float safe_sqrt(Float v) { return std::sqrt(std::max(0.0, v)); } // math.h:265
inline Float signum(Float v) { return v < 0 ? -1.0 : (v > 0 ? 1.0 : 0.0); }

// ---- counter-based generator shared (by specification, not by code) with the HIP path --------------
// One SplitMix64 stream per (seed, pixel, sample); double in [0,1) from the top 52 bits exactly like
// Random::nextFloat in the double build (random.cpp: ((u64 >> 12) | 0x3ff0000000000000) - 1.0).
inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t s;
    sfmt_oracle::Random *serial = nullptr;   // the reference's own generator: ONE serial SFMT-19937 stream (IndependentSampler::next1D, independent.cpp:94-96)
    Rng(uint64_t seed, uint64_t pixel, uint64_t sample)
    {
        s = mix64(seed + 0x9E3779B97F4A7C15ULL * (pixel + 1));
        s = mix64(s ^ (0xD1B54A32D192ED03ULL * (sample + 1)));
    }
    explicit Rng(sfmt_oracle::Random *stream) : s(0), serial(stream) {}
    Float next1D()
    {
        if (serial) return serial->nextFloat();
        s += 0x9E3779B97F4A7C15ULL;
        const uint64_t bits = (mix64(s) >> 12) | 0x3FF0000000000000ULL;
        Float d;
        std::memcpy(&d, &bits, 8);
        return d - 1.0;
    }
};

// ---- Frame: include/mitsuba/core/frame.h:37-, util.cpp:592-608 ------------------------------------
struct Frame {
    V3 s, t, n;
    V3 toLocal(V3 v) const { return V3(dot(v, s), dot(v, t), dot(v, n)); }
    V3 toWorld(V3 v) const { return s * v.x + t * v.y + n * v.z; }
};
inline Float cosTheta(V3 v) { return v.z; }
inline Float tanTheta(V3 v) { Float t = 1 - v.z * v.z; return t <= 0.0 ? 0.0 : std::sqrt(t) / v.z; } // frame.h:122
inline void coordinateSystem(V3 a, V3 &b, V3 &c)
{ // util.cpp:592-601
    if (std::abs(a.x) > std::abs(a.y)) {
        Float invLen = 1.0 / std::sqrt(a.x * a.x + a.z * a.z);
        c = V3(a.z * invLen, 0.0, -a.x * invLen);
    } else {
        Float invLen = 1.0 / std::sqrt(a.y * a.y + a.z * a.z);
        c = V3(0.0, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}

// ---- materials --------------------------------------------------------------------------------------
enum { MAT_DIFFUSE = 0, MAT_CONDUCTOR = 1, MAT_ROUGHCONDUCTOR = 2, MAT_DIELECTRIC = 3 };
enum { DISTR_BECKMANN = 0, DISTR_GGX = 1, DISTR_PHONG = 2 };     // MicrofacetDistribution::EType, microfacet.h:47-57
const Float RCPOVERFLOW = 0x1p-1024;                              // constants.h:59,97
const Float INV_TWOPI = 0.15915494309189533577;                   // constants.h
// BSDF::EBSDFType bits used here (include/mitsuba/render/bsdf.h): diffuse/glossy reflection are smooth, delta is not
enum { EDiffuseReflection = 0x1, EGlossyReflection = 0x4, EDeltaReflection = 0x10, EDeltaTransmission = 0x20, ESmooth = 0x1 | 0x4, EDelta = 0x10 | 0x20 };
enum { MEASURE_SOLID_ANGLE = 0, MEASURE_DISCRETE = 1 };

} // namespace

// Plain-C scene description, identical in meaning to include/gdpt_tracer.h (defined separately on purpose).
extern "C" {
typedef struct gpo_material {
    int type;            // MAT_*
    int distribution;    // DISTR_* (roughconductor)
    int sampleVisible;   // roughconductor.cpp m_sampleVisible (default true)
    int twoSided;        // wrapped in src/bsdfs/twosided.cpp (the same BRDF on both sides)
    double reflectance[3]; // diffuse: reflectance; conductors: specularReflectance
    double eta[3], k[3];   // dielectric: eta[0] = intIOR/extIOR, reflectance = specularReflectance, k = specularTransmittance
    double alphaU, alphaV;
} gpo_material;

typedef struct gpo_emitter {
    int firstTri, numTris; // the triangles of the emissive mesh (contiguous); numTris == -1: a `point` emitter (src/emitters/point.cpp)
    double radiance[3];    // area: radiance; point: intensity
    double position[3];    // point emitters only
} gpo_emitter;

typedef struct gpo_camera {
    double toWorld[16]; // row-major camera-to-world (Transform::lookAt convention, transform.cpp:191-214)
    double fovX;        // degrees
    double nearClip, farClip;
    int width, height;
    int type;           // 0 perspective, 1 thinlens (src/sensors/thinlens.cpp)
    double apertureRadius, focusDistance;
    double shutterOpen, shutterClose;   // Sensor::Sensor, sensor.cpp:26-38: shutterClose > shutterOpen <=> needsTimeSample() (sensor.h:290)
    int cropOffsetX, cropOffsetY;       // the film's crop window (film.cpp:34-48): width x height is the CROP size, the film itself fullWidth x fullHeight
    int fullWidth, fullHeight;          // (0: no crop).  G-PT only -- the G-BDPT side takes crop == film
} gpo_camera;

typedef struct gpo_config {
    int maxDepth, rrDepth, strictNormals, spp;
    double shiftThreshold;
    unsigned long long seed;
} gpo_config;
}

namespace {

struct TriAccel { // include/mitsuba/render/triaccel.h:37-94
    uint32_t k;
    Float n_u, n_v, n_d, a_u, a_v, b_nu, b_nv, c_nu, c_nv;
};

struct Tri {
    V3 p0, p1, p2;
    TriAccel acc;
    int material, emitter;
    V3 faceNormal; // normalized cross(side1, side2)   (skdtree.h:367-371)
    Frame sh;      // shading frame: n = faceNormal, s,t from dpdu   (skdtree.h:379,396; util.cpp:603-608)
    V3 dpdu;       // its.dpdu: side1, or the UV tangent of a mesh with texture coordinates (skdtree.h:373-380, trimesh.cpp:683-735)
    V3 dpdv;
    V3 geoN;
    bool hasNormals = false;   // per-vertex normals (TriMesh::getVertexNormals): interpolated shading normal, skdtree.h:382-394
    V3 n0, n1, n2;
    bool hasUV = false;        // per-vertex texture coordinates (TriMesh::getVertexTexcoords), skdtree.h:398-405
    Float uv[6] = {0, 0, 0, 0, 0, 0};
};

// `<texture type="bitmap">` (src/textures/bitmap.cpp) as the G-PT path evaluates it.  Texture2D::eval (texture.cpp:112-121) scales and
// offsets its.uv; with filterType nearest / bilinear both BitmapTexture::eval overloads end in MIPMap::evalBox / evalBilinear on
// level 0 (bitmap.cpp:431-452, mipmap.h:566-596,628-633), ray differentials or not; "trilinear" and "ewa" (the default) do the same except
// at the hit of a camera ray, whose UV partials drive the filtered lookup of oracle/mipmap_oracle.hpp.  Texels are Float (the MIP map converts the
// file to Bitmap::EFloat), wrap modes as evalTexel (mipmap.h:503-561).  `scale` is the factor of BSDF::ensureEnergyConservation
// (bsdf.cpp: 0.99 / max when the texture exceeds 1), applied to the interpolated value as ScaleTexture does.
struct Texture {
    int w = 0, h = 0, wrapU = 0, wrapV = 0, filter = 1;      // wrap: 0 repeat, 1 clamp, 2 mirror, 3 zero, 4 one; filter: 0 nearest, 1 bilinear, 2 trilinear, 3 ewa
    Float uscale = 1, vscale = 1, uoffset = 0, voffset = 0, scale = 1;
    std::vector<Float> rgb;                                   // [h][w][3], top row first
    static int modulo(int a, int b) { const int r = a % b; return r < 0 ? r + b : r; }        // math::modulo, math.h
    static int floorToInt(Float v) { return (int)std::floor(v); }
    bool wrap(int &x, int size, int mode, Float &constant) const
    {
        if (x >= 0 && x < size) return true;
        switch (mode) {
            case 0: x = modulo(x, size); return true;
            case 1: x = std::min(std::max(x, 0), size - 1); return true;
            case 2: x = modulo(x, 2 * size); if (x >= size) x = 2 * size - x - 1; return true;
            case 3: constant = 0.0; return false;
            default: constant = 1.0; return false;
        }
    }
    V3 texel(int x, int y) const
    {
        Float c = 0;
        if (!wrap(x, w, wrapU, c)) return V3(c);
        if (!wrap(y, h, wrapV, c)) return V3(c);
        const Float *t = &rgb[((size_t)y * w + x) * 3];
        return V3(t[0], t[1], t[2]);
    }
    // filter 2 = trilinear, 3 = ewa (the reference's default): the MIP pyramid and the filtered lookup of oracle/mipmap_oracle.hpp
    mip_oracle::MipMap mip;
    Float maxAnisotropy = 20;
    // Texture2D::eval(its, filter = true), texture.cpp:112-121: scaled coordinates, and scaled partials if the hit has any
    V3 eval(Float u_, Float v_, bool hasPartials = false, Float dudx = 0, Float dudy = 0, Float dvdx = 0, Float dvdy = 0) const
    {
        const Float ux = u_ * uscale + uoffset, vy = v_ * vscale + voffset;                   // texture.cpp:113
        V3 value;
        if (filter >= 2) {
            Float o[3];
            if (hasPartials) mip.eval(ux, vy, dudx * uscale, dvdx * vscale, dudy * uscale, dvdy * vscale, o);   // BitmapTexture::eval(uv, d0, d1), bitmap.cpp:486-499
            else mip.evalBilinear(0, ux, vy, o);                                                                // BitmapTexture::eval(uv), bitmap.cpp:431-452
            return V3(o[0], o[1], o[2]) * scale;
        }
        if (filter == 0) value = texel(floorToInt(ux * w), floorToInt(vy * h));               // evalBox, mipmap.h:566-569
        else {
            if (!std::isfinite(ux) || !std::isfinite(vy)) return V3(0.0) * scale;             // mipmap.h:576-578
            const Float u = ux * w - 0.5f, v = vy * h - 0.5f;                                 // :586
            const int xPos = floorToInt(u), yPos = floorToInt(v);
            const Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
            value = texel(xPos, yPos) * dx2 * dy2 + texel(xPos, yPos + 1) * dx2 * dy1 + texel(xPos + 1, yPos) * dx1 * dy2 + texel(xPos + 1, yPos + 1) * dx1 * dy1;   // :592-595
        }
        return value * scale;
    }
};

struct Emitter {
    int firstTri, numTris;      // numTris == 0: the environment emitter (`constant`, src/emitters/constant.cpp); -1: `point` (point.cpp)
    V3 radiance;
    V3 position;
    bool onSurface() const { return numTris >= 0; }   // Emitter::isOnSurface: area and constant set EOnSurface, point does not
    std::vector<Float> cdf; // DiscreteDistribution over triangle areas (trimesh.cpp:395-403, pmf.h)
    Float invSurfaceArea;
    bool rectangle = false; // the light of a `rectangle` shape (src/shapes/rectangle.cpp), hit as the two triangles of its createTriMesh() but
    Float rect[12];         // SAMPLED as the shape samples itself: objectToWorld(2u - 1, 2v - 1, 0), its frame's normal, pdf 1 / (|dpdu| |dpdv|)
    V3 rectN;
};

struct Distribution { // include/mitsuba/core/pmf.h
    std::vector<Float> cdf;
    Float sum, normalization;
    Distribution() : cdf(1, 0.0), sum(0), normalization(0) {}
    void append(Float v) { cdf.push_back(cdf.back() + v); }
    Float normalize()
    { // pmf.h:95-108
        sum = cdf.back();
        if (sum > 0) {
            normalization = 1.0 / sum;
            for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= normalization;
            cdf.back() = 1.0;
        } else normalization = 0.0;
        return sum;
    }
    Float operator[](size_t i) const { return cdf[i + 1] - cdf[i]; }
    size_t sample(Float v) const
    { // pmf.h:110-123
        auto entry = std::lower_bound(cdf.begin(), cdf.end(), v);
        size_t index = std::min(cdf.size() - 2, (size_t)std::max((std::ptrdiff_t)0, (std::ptrdiff_t)(entry - cdf.begin()) - 1));
        while ((*this)[index] == 0 && index < cdf.size() - 1) ++index;
        return index;
    }
    size_t sampleReuse(Float &v) const
    { // pmf.h:164-169
        size_t index = sample(v);
        v = (v - cdf[index]) / (cdf[index + 1] - cdf[index]);
        return index;
    }
    size_t sampleReuse(Float &v, Float &pdf) const
    {
        size_t index = sample(v);
        pdf = (*this)[index];
        v = (v - cdf[index]) / (cdf[index + 1] - cdf[index]);
        return index;
    }
};

struct Ray {                    // RayDifferential: a camera ray carries the directions of the rays through the pixels to the right and below
    V3 o, d;                    // (rxOrigin = ryOrigin = o for the perspective camera, perspective.cpp:291-295); every other ray has none
    Float mint, maxt;
    bool hasDifferentials = false;
    V3 rxD, ryD;
    Ray() : mint(Epsilon), maxt(INF) {}
    Ray(V3 o_, V3 d_) : o(o_), d(d_), mint(Epsilon), maxt(INF) {}                       // ray.h: Ray(o, d, time)
    Ray(V3 o_, V3 d_, Float mn, Float mx) : o(o_), d(d_), mint(mn), maxt(mx) {}
};

struct Intersection {
    Float t;
    int prim;
    Float u = 0, v = 0;         // its.uv
    V3 p, wi;
    Frame sh;
    V3 geoN;
    V3 dpdu, dpdv;              // skdtree.h:373-380
    bool hasUVPartials = false; // Intersection::computePartials ran for this hit (only a camera ray can make it, intersection.cpp:11)
    Float dudx = 0, dudy = 0, dvdx = 0, dvdy = 0;
    Intersection() : t(INF), prim(-1) {}
    bool isValid() const { return t != INF; }
};

struct Scene {
    std::vector<Tri> tris;
    std::vector<gpo_material> mats;
    std::vector<int> matTexture;    // per material: index of the bitmap texture on its `reflectance` / `specularReflectance`, -1 = constant
    std::vector<Texture> textures;
    std::vector<Emitter> emitters;
    Distribution emitterPDF; // scene.cpp:357-380, every emitter has sampling weight 1
    gpo_camera cam;
    // camera constants (perspective.cpp:125-163)
    Float aspect, tanHalf;
    V3 aabbMin, aabbMax;
    int rfilterKind = 0;        // the film's reconstruction filter (Film::filterEval)
    double rfilterP0 = 0, rfilterP1 = 0;
    int envIndex = -1;          // position of the environment emitter in `emitters` (scene order), -1: none
    // `<emitter type="envmap">` (src/emitters/envmap.cpp) instead of `constant`: latitude-longitude bitmap in a half-precision MIP map
    // (repeat in u, clamp in v, EWA with maxAnisotropy 10: envmap.cpp:135-138,178-181), importance-sampled through float cdf tables
    struct EnvMap {
        bool present = false;
        mip_oracle::MipMap mip;
        int w = 0, h = 0;
        std::vector<float> cdfRows, cdfCols;
        std::vector<Float> rowWeights;
        Float normalization = 0, scale = 1, pixelSizeX = 0, pixelSizeY = 0;
        Float toWorld[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, toLocal[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // linear part of the emitter's toWorld and its inverse (row-major)
    } envMap;
    V3 bsCenter;                // ConstantBackgroundEmitter::m_sceneBSphere (constant.cpp:67-70)
    Float bsRadius = 0;
    mutable uint64_t raysTraced = 0, shadowRaysTraced = 0; // skdtree.cpp:46-47
    unsigned long long lastInvalidPuts = 0;                // puts dropped by ImageBlock::put's validity check in the last render
    // Oracle-side acceleration for large scenes only (> 64 triangles): a plain midpoint-split bounding-volume tree over double
    // bounds.  It changes which triangles are TESTED, never the test or the answer (closest t wins); small scenes stay brute force.
    struct BNode { V3 lo, hi; int left, right, first, count; };
    std::vector<BNode> bvh;
    std::vector<int> bvhOrder;
};

int triaccel_load(TriAccel &ta, V3 A, V3 B, V3 C)
{ // triaccel.h:61-94
    static const int waldModulo[4] = {1, 2, 0, 1};
    V3 b = C - A, c = B - A, N = cross(c, b);
    uint32_t k = 0;
    for (int j = 0; j < 3; j++)
        if (std::abs(N[j]) > std::abs(N[k])) k = j;
    uint32_t u = waldModulo[k], v = waldModulo[k + 1];
    const Float n_k = N[k], denom = b[u] * c[v] - b[v] * c[u];
    if (denom == 0) { ta.k = 3; return 1; }
    ta.k = k;
    ta.n_u = N[u] / n_k;
    ta.n_v = N[v] / n_k;
    ta.n_d = dot(A, N) / n_k;
    ta.b_nu = b[u] / denom;
    ta.b_nv = -b[v] / denom;
    ta.a_u = A[u];
    ta.a_v = A[v];
    ta.c_nu = c[v] / denom;
    ta.c_nv = -c[u] / denom;
    return 0;
}

inline bool triaccel_intersect(const TriAccel &ta, const Ray &ray, Float mint, Float maxt, Float &u, Float &v, Float &t)
{ // triaccel.h:96-158
    Float o_u, o_v, o_k, d_u, d_v, d_k;
    switch (ta.k) {
    case 0: o_u = ray.o.y; o_v = ray.o.z; o_k = ray.o.x; d_u = ray.d.y; d_v = ray.d.z; d_k = ray.d.x; break;
    case 1: o_u = ray.o.z; o_v = ray.o.x; o_k = ray.o.y; d_u = ray.d.z; d_v = ray.d.x; d_k = ray.d.y; break;
    case 2: o_u = ray.o.x; o_v = ray.o.y; o_k = ray.o.z; d_u = ray.d.x; d_v = ray.d.y; d_k = ray.d.z; break;
    default: return false;
    }
    t = (ta.n_d - o_u * ta.n_u - o_v * ta.n_v - o_k) / (d_u * ta.n_u + d_v * ta.n_v + d_k);
    if (t < mint || t > maxt) return false;
    const Float hu = o_u + t * d_u - ta.a_u;
    const Float hv = o_v + t * d_v - ta.a_v;
    u = hv * ta.b_nu + hu * ta.b_nv;
    v = hu * ta.c_nu + hv * ta.c_nv;
    return u >= 0 && v >= 0 && u + v <= 1.0;
}

// AABB::rayIntersect (include/mitsuba/core/aabb.h) for the scene bounds test of skdtree.cpp:123,211
bool aabb_ray(const Scene &sc, const Ray &ray, Float &nearT, Float &farT)
{
    nearT = -INF; farT = INF;
    for (int i = 0; i < 3; i++) {
        const Float origin = ray.o[i], minVal = sc.aabbMin[i], maxVal = sc.aabbMax[i], dir = ray.d[i];
        if (dir == 0) {
            if (origin < minVal || origin > maxVal) return false;
        } else {
            const Float rcp = 1.0 / dir;
            Float t1 = (minVal - origin) * rcp, t2 = (maxVal - origin) * rcp;
            if (t1 > t2) std::swap(t1, t2);
            nearT = std::max(t1, nearT);
            farT = std::min(t2, farT);
            if (!(nearT <= farT)) return false;
        }
    }
    return true;
}

inline bool boxHit(const Scene::BNode &n, const Ray &ray, Float mint, Float maxt)
{
    Float t0 = mint, t1 = maxt;
    for (int a = 0; a < 3; a++) {
        const Float o = ray.o[a], d = ray.d[a];
        if (d == 0) { if (o < n.lo[a] || o > n.hi[a]) return false; continue; }
        Float ta = (n.lo[a] - o) / d, tb = (n.hi[a] - o) / d;
        if (ta > tb) std::swap(ta, tb);
        // conservative: widen by a relative margin so that rounding in this culling test can never drop a true hit
        ta -= std::abs(ta) * 1e-12 + 1e-300; tb += std::abs(tb) * 1e-12 + 1e-300;
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
        if (t0 > t1) return false;
    }
    return true;
}

// Visits every triangle whose leaf box the ray segment [mint, maxt] touches; `visit` returns the (possibly shortened) maxt,
// or a negative value to stop (any-hit).
template <class F> void bvhVisit(const Scene &sc, const Ray &ray, Float mint, Float maxt, F visit)
{
    if (sc.bvh.empty()) {
        for (size_t i = 0; i < sc.tris.size(); ++i) { maxt = visit((int)i, maxt); if (maxt < 0) return; }
        return;
    }
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const Scene::BNode &n = sc.bvh[stack[--sp]];
        if (!boxHit(n, ray, mint, maxt)) continue;
        if (n.count) {
            for (int i = 0; i < n.count; ++i) { maxt = visit(sc.bvhOrder[n.first + i], maxt); if (maxt < 0) return; }
        } else { stack[sp++] = n.left; stack[sp++] = n.right; }
    }
}

int bvhBuild(Scene &sc, int first, int count)
{
    const int me = (int)sc.bvh.size();
    sc.bvh.push_back(Scene::BNode());
    V3 lo(INF), hi(-INF), clo(INF), chi(-INF);
    for (int i = first; i < first + count; ++i) {
        const Tri &t = sc.tris[sc.bvhOrder[i]];
        const V3 ps[3] = {t.p0, t.p1, t.p2};
        V3 c(0.0);
        for (const V3 &p : ps) {
            lo = V3(std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z));
            hi = V3(std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z));
            c = c + p;
        }
        c = c * (1.0 / 3.0);
        clo = V3(std::min(clo.x, c.x), std::min(clo.y, c.y), std::min(clo.z, c.z));
        chi = V3(std::max(chi.x, c.x), std::max(chi.y, c.y), std::max(chi.z, c.z));
    }
    sc.bvh[me].lo = lo; sc.bvh[me].hi = hi; sc.bvh[me].left = sc.bvh[me].right = -1; sc.bvh[me].first = first; sc.bvh[me].count = 0;
    const V3 ext = chi - clo;
    const int axis = ext.x >= ext.y && ext.x >= ext.z ? 0 : (ext.y >= ext.z ? 1 : 2);
    if (count <= 4 || !(ext[axis] > 0)) { sc.bvh[me].count = count; return me; }
    const Float mid = 0.5 * (clo[axis] + chi[axis]);
    auto centroid = [&](int ti) { const Tri &t = sc.tris[ti]; return (t.p0[axis] + t.p1[axis] + t.p2[axis]) * (1.0 / 3.0); };
    int *b = sc.bvhOrder.data() + first;
    int nl = (int)(std::partition(b, b + count, [&](int ti) { return centroid(ti) < mid; }) - b);
    if (nl == 0 || nl == count) nl = count / 2;
    const int l = bvhBuild(sc, first, nl), r = bvhBuild(sc, first + nl, count - nl);
    sc.bvh[me].left = l; sc.bvh[me].right = r;
    return me;
}

// ShapeKDTree::rayIntersect(ray, its), skdtree.cpp:112-142 + fillIntersectionRecord<true>, skdtree.h:343-428
bool rayIntersect(const Scene &sc, const Ray &ray, Intersection &its)
{
    its.t = INF;
    its.prim = -1;
    Float mint, maxt;
    ++sc.raysTraced;
    if (!aabb_ray(sc, ray, mint, maxt)) return false;
    Float rayMinT = ray.mint;
    if (rayMinT == Epsilon) // adaptive ray epsilon, skdtree.cpp:126-129
        rayMinT *= std::max(std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z)), Epsilon);
    if (rayMinT > mint) mint = rayMinT;
    if (ray.maxt < maxt) maxt = ray.maxt;
    if (!(maxt > mint)) return false;
    Float bu = 0, bv = 0;
    bvhVisit(sc, ray, mint, maxt, [&](int i, Float mx) { // closest hit; the traversal order of the kd-tree is not restated
        Float u, v, t;
        // ties in t between triangles sharing an edge go to the lowest triangle index (brute-force order), whatever the visiting order
        if (triaccel_intersect(sc.tris[i].acc, ray, mint, mx, u, v, t) && (t < its.t || (t == its.t && i < its.prim))) {
            its.t = t; its.prim = i; bu = u; bv = v;
            return t;
        }
        return mx;
    });
    if (its.prim < 0) return false;
    const Tri &tr = sc.tris[its.prim];
    const V3 b(1 - bu - bv, bu, bv);
    its.p = tr.p0 * b.x + tr.p1 * b.y + tr.p2 * b.z;
    its.sh = tr.sh;
    its.geoN = tr.geoN;
    its.dpdu = tr.dpdu; its.dpdv = tr.dpdv;
    its.hasUVPartials = false;
    if (tr.hasNormals) {                                                 // skdtree.h:382-394,426
        its.sh.n = normalize(tr.n0 * b.x + tr.n1 * b.y + tr.n2 * b.z);
        if (dot(tr.faceNormal, its.sh.n) < 0) its.geoN = -tr.faceNormal; // geometric and shading normals face the same way
        const V3 dpdu = tr.dpdu;
        its.sh.s = normalize(dpdu - its.sh.n * dot(its.sh.n, dpdu));     // computeShadingFrame, util.cpp:603-608
        its.sh.t = cross(its.sh.n, its.sh.s);
    }
    if (tr.hasUV) {                                                      // skdtree.h:398-405
        its.u = tr.uv[0] * b.x + tr.uv[2] * b.y + tr.uv[4] * b.z;
        its.v = tr.uv[1] * b.x + tr.uv[3] * b.y + tr.uv[5] * b.z;
    } else { its.u = b.y; its.v = b.z; }
    its.wi = its.sh.toLocal(-ray.d);
    return true;
}

// ShapeKDTree::rayIntersect(ray) (shadow), skdtree.cpp:207-226
bool rayIntersectShadow(const Scene &sc, const Ray &ray)
{
    Float mint, maxt;
    ++sc.shadowRaysTraced;
    if (!aabb_ray(sc, ray, mint, maxt)) return false;
    Float rayMinT = ray.mint;
    if (rayMinT == Epsilon) // no floor here, skdtree.cpp:214-217
        rayMinT *= std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z));
    if (rayMinT > mint) mint = rayMinT;
    if (ray.maxt < maxt) maxt = ray.maxt;
    if (!(maxt > mint)) return false;
    bool hit = false;
    bvhVisit(sc, ray, mint, maxt, [&](int i, Float mx) {
        Float u, v, t;
        if (triaccel_intersect(sc.tris[i].acc, ray, mint, mx, u, v, t)) { hit = true; return (Float)-1.0; }
        return mx;
    });
    return hit;
}

// ---- warps: src/libcore/warp.cpp ------------------------------------------------------------------------
void squareToUniformDiskConcentric(Float sx, Float sy, Float &ox, Float &oy)
{ // warp.cpp:81-102
    Float r1 = 2.0 * sx - 1.0, r2 = 2.0 * sy - 1.0, phi, r;
    if (r1 == 0 && r2 == 0) { r = phi = 0; }
    else if (r1 * r1 > r2 * r2) { r = r1; phi = (PI / 4.0) * (r2 / r1); }
    else { r = r2; phi = (PI / 2.0) - (r1 / r2) * (PI / 4.0); }
    ox = r * std::cos(phi);
    oy = r * std::sin(phi);
}
V3 squareToCosineHemisphere(Float sx, Float sy)
{ // warp.cpp:43-52
    Float px, py;
    squareToUniformDiskConcentric(sx, sy, px, py);
    Float z = safe_sqrt(1.0 - px * px - py * py);
    if (z == 0) z = (Float)1e-10f;
    return V3(px, py, z);
}

// ---- Fresnel: util.cpp:739-761 (Spectrum version, per channel) -----------------------------------------
V3 fresnelConductorExact(Float cosThetaI, V3 eta, V3 k)
{
    Float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    Float out[3];
    for (int c = 0; c < 3; c++) {
        const Float e = eta[c], kk = k[c];
        Float temp1 = e * e - kk * kk - sinThetaI2;
        Float a2pb2 = safe_sqrt(temp1 * temp1 + kk * kk * e * e * 4);
        Float a = safe_sqrt((a2pb2 + temp1) * 0.5);
        Float term1 = a2pb2 + cosThetaI2, term2 = a * (2 * cosThetaI);
        Float Rs2 = (term1 - term2) / (term1 + term2);
        Float term3 = a2pb2 * cosThetaI2 + sinThetaI4, term4 = term2 * sinThetaI2;
        Float Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
        out[c] = 0.5 * (Rp2 + Rs2);
    }
    return V3(out[0], out[1], out[2]);
}

// ---- math: src/libcore/math.cpp:25-72 -------------------------------------------------------------------
Float erfinv_m(Float x)
{
    Float w = -std::log((1.0 - x) * (1.0 + x)), p;
    if (w < 5.0) {
        w = w - 2.5;
        p = 2.81022636e-08; p = 3.43273939e-07 + p * w; p = -3.5233877e-06 + p * w; p = -4.39150654e-06 + p * w;
        p = 0.00021858087 + p * w; p = -0.00125372503 + p * w; p = -0.00417768164 + p * w; p = 0.246640727 + p * w; p = 1.50140941 + p * w;
    } else {
        w = std::sqrt(w) - 3.0;
        p = -0.000200214257; p = 0.000100950558 + p * w; p = 0.00134934322 + p * w; p = -0.00367342844 + p * w;
        p = 0.00573950773 + p * w; p = -0.0076224613 + p * w; p = 0.00943887047 + p * w; p = 1.00167406 + p * w; p = 2.83297682 + p * w;
    }
    return p * x;
}
Float erf_m(Float x)
{
    const Float a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429, p = 0.3275911;
    Float sign = signum(x);
    x = std::abs(x);
    Float t = 1.0 / (1.0 + p * x);
    Float y = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * std::exp(-x * x);
    return sign * y;
}
Float hypot2(Float a, Float b)
{
    Float r;
    if (std::abs(a) > std::abs(b)) { r = b / a; r = std::abs(a) * std::sqrt(1.0 + r * r); }
    else if (b != 0.0) { r = a / b; r = std::abs(b) * std::sqrt(1.0 + r * r); }
    else r = 0.0;
    return r;
}

// ---- MicrofacetDistribution: src/bsdfs/microfacet.h ------------------------------------------------------
struct Microfacet {
    int type;
    Float alphaU, alphaV;
    bool sampleVisible;
    Float exponentU, exponentV;                     // Phong / Ashikhmin-Shirley exponents, computePhongExponent :701-704
    Microfacet(int t, Float au, Float av, bool sv) : type(t), alphaU(std::max(au, (Float)1e-4f)), alphaV(std::max(av, (Float)1e-4f)), sampleVisible(sv)
    { // :135-144
        exponentU = exponentV = 0;
        if (type == DISTR_PHONG) {                  // visible-normal sampling is not supported for Phong
            sampleVisible = false;
            exponentU = std::max(2.0 / (alphaU * alphaU) - 2.0, (Float)0.0);
            exponentV = std::max(2.0 / (alphaV * alphaV) - 2.0, (Float)0.0);
        }
    }
    bool isIsotropic() const { return alphaU == alphaV; }
    Float interpolatePhongExponent(V3 v) const
    { // :554-565
        const Float sinTheta2 = 1.0 - v.z * v.z;
        if (isIsotropic() || sinTheta2 <= RCPOVERFLOW) return exponentU;
        Float invSinTheta2 = 1 / sinTheta2;
        Float cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
        return exponentU * cosPhi2 + exponentV * sinPhi2;
    }
    void sampleFirstQuadrant(Float u1, Float &phi, Float &exponent) const
    { // :707-715
        phi = std::atan(std::sqrt((exponentU + 2.0) / (exponentV + 2.0)) * std::tan(PI * u1 * 0.5));
        Float cosPhi = std::cos(phi), sinPhi = std::sin(phi);
        exponent = exponentU * cosPhi * cosPhi + exponentV * sinPhi * sinPhi;
    }
    Float eval(V3 m) const
    { // :191-234
        if (cosTheta(m) <= 0) return 0.0;
        Float cosTheta2 = m.z * m.z;
        Float beckmannExponent = ((m.x * m.x) / (alphaU * alphaU) + (m.y * m.y) / (alphaV * alphaV)) / cosTheta2;
        Float result;
        if (type == DISTR_BECKMANN) result = std::exp(-beckmannExponent) / (PI * alphaU * alphaV * cosTheta2 * cosTheta2);
        else if (type == DISTR_PHONG) result = std::sqrt((exponentU + 2) * (exponentV + 2)) * INV_TWOPI * std::pow(cosTheta(m), interpolatePhongExponent(m)); // :215-221
        else { Float root = (1.0 + beckmannExponent) * cosTheta2; result = 1.0 / (PI * alphaU * alphaV * root * root); }
        if (result * cosTheta(m) < (Float)1e-20f) result = 0;
        return result;
    }
    Float projectRoughness(V3 v) const
    { // :531-541
        Float invSinTheta2 = 1 / (1.0 - v.z * v.z);
        if (isIsotropic() || invSinTheta2 <= 0) return alphaU;
        Float cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
        return std::sqrt(cosPhi2 * alphaU * alphaU + sinPhi2 * alphaV * alphaV);
    }
    Float smithG1(V3 v, V3 m) const
    { // :477-514
        if (dot(v, m) * cosTheta(v) <= 0) return 0.0;
        Float tanT = std::abs(tanTheta(v));
        if (tanT == 0.0) return 1.0;
        Float alpha = projectRoughness(v);
        if (type == DISTR_BECKMANN || type == DISTR_PHONG) {                    // :489-501
            Float a = 1.0 / (alpha * tanT);
            if (a >= (Float)1.6f) return 1.0;
            Float aSqr = a * a;
            return ((Float)3.535f * a + (Float)2.181f * aSqr) / (1.0 + (Float)2.276f * a + (Float)2.577f * aSqr);
        }
        Float root = alpha * tanT;
        return 2.0 / (1.0 + hypot2(1.0, root));
    }
    Float G(V3 wi, V3 wo, V3 m) const { return smithG1(wi, m) * smithG1(wo, m); }
    Float pdfVisible(V3 wi, V3 m) const
    { // :470-474
        if (cosTheta(wi) == 0) return 0.0;
        return smithG1(wi, m) * std::abs(dot(wi, m)) * eval(m) / std::abs(cosTheta(wi));
    }
    Float pdfAll(V3 m) const { return eval(m) * cosTheta(m); } // :417-420
    Float pdf(V3 wi, V3 m) const { return sampleVisible ? pdfVisible(wi, m) : pdfAll(m); }
    void sampleVisible11(Float thetaI, Float sx, Float sy, Float &slx, Float &sly) const
    { // :573-702
        const Float SQRT_PI_INV = 1 / std::sqrt(PI);
        if (type == DISTR_BECKMANN) {
            if (thetaI < (Float)1e-4f) {
                Float r = std::sqrt(-std::log(1.0 - sx));
                Float ph = 2 * PI * sy;
                slx = r * std::cos(ph); sly = r * std::sin(ph);
                return;
            }
            Float tanThetaI = std::tan(thetaI), cotThetaI = 1 / tanThetaI;
            Float a = -1, c = erf_m(cotThetaI);
            Float sample_x = std::max(sx, (Float)1e-6f);
            Float fit = 1 + thetaI * ((Float)-0.876f + thetaI * ((Float)0.4265f - (Float)0.0594f * thetaI));
            Float b = c - (1 + c) * std::pow(1 - sample_x, fit);
            Float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * std::exp(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c)) b = 0.5 * (a + c);
                Float invErf = erfinv_m(b);
                Float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * std::exp(-invErf * invErf)) - sample_x;
                Float derivative = normalization * (1 - invErf * tanThetaI);
                if (std::abs(value) < (Float)1e-5f) break;
                if (value > 0) c = b; else a = b;
                b -= value / derivative;
            }
            slx = erfinv_m(b);
            sly = erfinv_m(2.0 * std::max(sy, (Float)1e-6f) - 1.0);
            return;
        }
        // GGX
        if (thetaI < (Float)1e-4f) {
            Float r = safe_sqrt(sx / (1 - sx));
            Float ph = 2 * PI * sy;
            slx = r * std::cos(ph); sly = r * std::sin(ph);
            return;
        }
        Float tanThetaI = std::tan(thetaI), a = 1 / tanThetaI;
        Float G1 = 2.0 / (1.0 + safe_sqrt(1.0 + 1.0 / (a * a)));
        Float A = 2.0 * sx / G1 - 1.0;
        if (std::abs(A) == 1) A -= signum(A) * Epsilon;
        Float tmp = 1.0 / (A * A - 1.0);
        Float B = tanThetaI;
        Float D = safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
        Float slope_x_1 = B * tmp - D, slope_x_2 = B * tmp + D;
        slx = (A < 0.0 || slope_x_2 > 1.0 / tanThetaI) ? slope_x_1 : slope_x_2;
        Float S;
        if (sy > (Float)0.5f) { S = 1.0; sy = 2.0 * (sy - 0.5); }
        else { S = -1.0; sy = 2.0 * (0.5 - sy); }
        Float z = (sy * (sy * (sy * (-0.365728915865723) + 0.790235037209296) - 0.424965825137544) + 0.000152998850436920) /
                  (sy * (sy * (sy * (sy * 0.169507819808272 - 0.397203533833404) - 0.232500544458471) + 1) - 0.539825872510702);
        sly = S * z * std::sqrt(1.0 + slx * slx);
    }
    V3 sampleVisibleM(V3 _wi, Float sx, Float sy) const
    { // :421-467
        V3 wi = normalize(V3(alphaU * _wi.x, alphaV * _wi.y, _wi.z));
        Float theta = 0, phi = 0;
        if (wi.z < (Float)0.99999) { theta = std::acos(wi.z); phi = std::atan2(wi.y, wi.x); }
        Float sinPhi = std::sin(phi), cosPhi = std::cos(phi);
        Float slx, sly;
        sampleVisible11(theta, sx, sy, slx, sly);
        Float rx = cosPhi * slx - sinPhi * sly, ry = sinPhi * slx + cosPhi * sly;
        rx *= alphaU; ry *= alphaV;
        Float normalization = 1.0 / std::sqrt(rx * rx + ry * ry + 1.0);
        return V3(-rx * normalization, -ry * normalization, normalization);
    }
    V3 sampleAll(Float sx, Float sy, Float &pdf) const
    { // :300-414
        Float cosThetaM, sinPhiM, cosPhiM, alphaSqr;
        if (type == DISTR_PHONG) {                                              // :349-375
            Float phiM, exponent;
            if (isIsotropic()) { phiM = (2.0 * PI) * sy; exponent = exponentU; }
            else if (sy < (Float)0.25f) sampleFirstQuadrant(4 * sy, phiM, exponent);
            else if (sy < (Float)0.5f) { sampleFirstQuadrant(4 * (0.5 - sy), phiM, exponent); phiM = PI - phiM; }
            else if (sy < (Float)0.75f) { sampleFirstQuadrant(4 * (sy - 0.5), phiM, exponent); phiM += PI; }
            else { sampleFirstQuadrant(4 * (1 - sy), phiM, exponent); phiM = 2 * PI - phiM; }
            sinPhiM = std::sin(phiM); cosPhiM = std::cos(phiM);
            cosThetaM = std::pow(sx, 1.0 / (exponent + 2.0));
            pdf = std::sqrt((exponentU + 2.0) * (exponentV + 2.0)) * INV_TWOPI * std::pow(cosThetaM, exponent + 1.0);
            if (pdf < (Float)1e-20f) pdf = 0;
            Float sinThetaM = std::sqrt(std::max(0.0, 1 - cosThetaM * cosThetaM));
            return V3(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
        }
        if (isIsotropic()) {
            Float ph = (2.0 * PI) * sy;
            sinPhiM = std::sin(ph); cosPhiM = std::cos(ph);
            alphaSqr = alphaU * alphaU;
        } else {
            Float phiM = std::atan(alphaV / alphaU * std::tan(PI + 2 * PI * sy)) + PI * std::floor(2 * sy + 0.5);
            sinPhiM = std::sin(phiM); cosPhiM = std::cos(phiM);
            Float cosSc = cosPhiM / alphaU, sinSc = sinPhiM / alphaV;
            alphaSqr = 1.0 / (cosSc * cosSc + sinSc * sinSc);
        }
        if (type == DISTR_BECKMANN) {
            Float tanThetaMSqr = alphaSqr * -std::log(1.0 - sx);
            cosThetaM = 1.0 / std::sqrt(1.0 + tanThetaMSqr);
            pdf = (1.0 - sx) / (PI * alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM);
        } else {
            Float tanThetaMSqr = alphaSqr * sx / (1.0 - sx);
            cosThetaM = 1.0 / std::sqrt(1.0 + tanThetaMSqr);
            Float temp = 1 + tanThetaMSqr / alphaSqr;
            pdf = INV_PI / (alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
        }
        if (pdf < (Float)1e-20f) pdf = 0;
        Float sinThetaM = std::sqrt(std::max(0.0, 1 - cosThetaM * cosThetaM));
        return V3(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
    }
    V3 sample(V3 wi, Float sx, Float sy, Float &pdf) const
    { // :240-250
        if (sampleVisible) { V3 m = sampleVisibleM(wi, sx, sy); pdf = pdfVisible(wi, m); return m; }
        return sampleAll(sx, sy, pdf);
    }
};

// ---- BSDFs --------------------------------------------------------------------------------------------
struct BSDFSample { V3 wo; Float eta; int sampledType; V3 weight; Float pdf; };

inline V3 rgb(const double *a) { return V3(a[0], a[1], a[2]); }
inline int bsdfType(const gpo_material &m) { return m.type == MAT_DIFFUSE ? EDiffuseReflection : (m.type == MAT_CONDUCTOR ? EDeltaReflection : (m.type == MAT_DIELECTRIC ? (EDeltaReflection | EDeltaTransmission) : EGlossyReflection)); }
inline Float getEta(const gpo_material &m) { return m.type == MAT_DIELECTRIC ? m.eta[0] : 1.0; }   // BSDF::getEta: dielectric.cpp:395, 1 elsewhere
inline Microfacet distr(const gpo_material &m) { return Microfacet(m.distribution, m.alphaU, m.alphaV, m.sampleVisible != 0); }
// BSDF::getRoughness: diffuse.cpp:167-169 (+inf), conductor.cpp:275-277 (0), roughconductor.cpp:437-440
inline Float getRoughness(const gpo_material &m) { return m.type == MAT_DIFFUSE ? INF : ((m.type == MAT_CONDUCTOR || m.type == MAT_DIELECTRIC) ? 0.0 : 0.5 * (m.alphaU + m.alphaV)); }   // dielectric.cpp:399: both components 0

// fresnelDielectricExt, util.cpp:651-681
Float fresnelDielectricExt(Float cosThetaI_, Float &cosThetaT_, Float eta)
{
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0; }
    Float scale = (cosThetaI_ > 0) ? 1 / eta : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0) { cosThetaT_ = 0.0; return 1.0; }
    Float cosThetaI = std::abs(cosThetaI_), cosThetaT = std::sqrt(cosThetaTSqr);
    Float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    Float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5 * (Rs * Rs + Rp * Rp);
}
// SmoothDielectric::refract, dielectric.cpp:222-225
inline V3 dielectricRefract(const gpo_material &m, V3 wi, Float cosThetaT)
{
    const Float eta = m.eta[0], invEta = 1 / eta;
    Float scale = -(cosThetaT < 0 ? invEta : eta);
    return V3(scale * wi.x, scale * wi.y, cosThetaT);
}

// `importance`: bRec.mode == EImportance (libbidir's light subpaths; G-PT only ever transports radiance); `typeMask`: bRec.typeMask restricted
// to one of the delta components (PathVertex::propagatePerturbation, vertex.cpp:696-700) -- both default to what G-PT asks for
V3 bsdfEvalOne(const gpo_material &m, V3 wi, V3 wo, int measure, bool importance = false, int typeMask = EDelta | ESmooth)
{
    if (m.type == MAT_DIELECTRIC) { // dielectric.cpp:227-252
        Float cosThetaT;
        Float F = fresnelDielectricExt(cosTheta(wi), cosThetaT, m.eta[0]);
        if (measure != MEASURE_DISCRETE) return V3(0.0);
        if (cosTheta(wi) * cosTheta(wo) >= 0) {
            if (!(typeMask & EDeltaReflection) || std::abs(dot(V3(-wi.x, -wi.y, wi.z), wo) - 1) > DeltaEpsilon) return V3(0.0);
            return rgb(m.reflectance) * F;
        }
        if (!(typeMask & EDeltaTransmission) || std::abs(dot(dielectricRefract(m, wi, cosThetaT), wo) - 1) > DeltaEpsilon) return V3(0.0);
        Float factor = importance ? 1.0 : (cosThetaT < 0 ? 1 / m.eta[0] : m.eta[0]);
        return rgb(m.k) * factor * factor * (1 - F);
    }
    switch (m.type) {
    case MAT_DIFFUSE: // diffuse.cpp:110-118
        if (measure != MEASURE_SOLID_ANGLE || cosTheta(wi) <= 0 || cosTheta(wo) <= 0) return V3(0.0);
        return rgb(m.reflectance) * (INV_PI * cosTheta(wo));
    case MAT_CONDUCTOR: // conductor.cpp:223-239
        if (!(typeMask & EDeltaReflection) || measure != MEASURE_DISCRETE || cosTheta(wi) <= 0 || cosTheta(wo) <= 0 || std::abs(dot(V3(-wi.x, -wi.y, wi.z), wo) - 1) > DeltaEpsilon) return V3(0.0);
        return rgb(m.reflectance) * fresnelConductorExact(cosTheta(wi), rgb(m.eta), rgb(m.k));
    default: { // roughconductor.cpp:257-293
        if (measure != MEASURE_SOLID_ANGLE || cosTheta(wi) <= 0 || cosTheta(wo) <= 0) return V3(0.0);
        V3 H = normalize(wo + wi);
        Microfacet d = distr(m);
        const Float D = d.eval(H);
        if (D == 0) return V3(0.0);
        const V3 F = fresnelConductorExact(dot(wi, H), rgb(m.eta), rgb(m.k)) * rgb(m.reflectance);
        const Float G = d.G(wi, wo, H);
        Float model = D * G / (4.0 * cosTheta(wi));
        return F * model;
    }
    }
}

Float bsdfPdfOne(const gpo_material &m, V3 wi, V3 wo, int measure, int typeMask = EDelta | ESmooth)
{
    if (m.type == MAT_DIELECTRIC) { // dielectric.cpp:254-275
        Float cosThetaT;
        Float F = fresnelDielectricExt(cosTheta(wi), cosThetaT, m.eta[0]);
        if (measure != MEASURE_DISCRETE) return 0.0;
        const bool sampleReflection = (typeMask & EDeltaReflection) != 0, sampleTransmission = (typeMask & EDeltaTransmission) != 0;
        if (cosTheta(wi) * cosTheta(wo) >= 0) {
            if (!sampleReflection || std::abs(dot(V3(-wi.x, -wi.y, wi.z), wo) - 1) > DeltaEpsilon) return 0.0;
            return sampleTransmission ? F : 1.0;
        }
        if (!sampleTransmission || std::abs(dot(dielectricRefract(m, wi, cosThetaT), wo) - 1) > DeltaEpsilon) return 0.0;
        return sampleReflection ? 1 - F : 1.0;
    }
    switch (m.type) {
    case MAT_DIFFUSE: // diffuse.cpp:120-127
        if (measure != MEASURE_SOLID_ANGLE || cosTheta(wi) <= 0 || cosTheta(wo) <= 0) return 0.0;
        return INV_PI * cosTheta(wo);
    case MAT_CONDUCTOR: // conductor.cpp:241-254
        if (!(typeMask & EDeltaReflection) || measure != MEASURE_DISCRETE || cosTheta(wi) <= 0 || cosTheta(wo) <= 0 || std::abs(dot(V3(-wi.x, -wi.y, wi.z), wo) - 1) > DeltaEpsilon) return 0.0;
        return 1.0;
    default: { // roughconductor.cpp:295-319
        if (measure != MEASURE_SOLID_ANGLE || cosTheta(wi) <= 0 || cosTheta(wo) <= 0) return 0.0;
        V3 H = normalize(wo + wi);
        Microfacet d = distr(m);
        if (d.sampleVisible) return d.eval(H) * d.smithG1(wi, H) / (4.0 * cosTheta(wi));
        return d.pdf(wi, H) / (4 * std::abs(dot(wo, H)));
    }
    }
}

// The pdf-returning BSDF::sample overloads: diffuse.cpp:141-151, conductor.cpp:256-273, roughconductor.cpp:369-418
BSDFSample bsdfSampleOne(const gpo_material &m, V3 wi, Float sx, Float sy, bool importance = false, int typeMask = EDelta | ESmooth)
{
    BSDFSample r;
    r.wo = V3(0.0); r.eta = 1.0; r.sampledType = 0; r.weight = V3(0.0); r.pdf = 0.0; // gpt.cpp:450-454: result.pdf starts at 0
    if (m.type == MAT_DIELECTRIC) { // dielectric.cpp:277-331
        Float cosThetaT;
        Float F = fresnelDielectricExt(cosTheta(wi), cosThetaT, m.eta[0]);
        const bool sampleReflection = (typeMask & EDeltaReflection) != 0, sampleTransmission = (typeMask & EDeltaTransmission) != 0;
        if (!(sampleReflection && sampleTransmission)) {      // one component only (:307-331): it is taken with probability 1 and carries its Fresnel factor
            if (sampleReflection) {
                r.sampledType = EDeltaReflection; r.wo = V3(-wi.x, -wi.y, wi.z); r.eta = 1.0; r.pdf = 1.0;
                r.weight = rgb(m.reflectance) * F;
            } else if (sampleTransmission) {
                r.sampledType = EDeltaTransmission; r.wo = dielectricRefract(m, wi, cosThetaT); r.eta = cosThetaT < 0 ? m.eta[0] : 1 / m.eta[0]; r.pdf = 1.0;
                Float factor = importance ? 1.0 : (cosThetaT < 0 ? 1 / m.eta[0] : m.eta[0]);
                r.weight = rgb(m.k) * (factor * factor * (1 - F));
            }
            return r;
        }
        if (sx <= F) {
            r.sampledType = EDeltaReflection;
            r.wo = V3(-wi.x, -wi.y, wi.z);
            r.eta = 1.0;
            r.pdf = F;
            r.weight = rgb(m.reflectance);
        } else {
            r.sampledType = EDeltaTransmission;
            r.wo = dielectricRefract(m, wi, cosThetaT);
            r.eta = cosThetaT < 0 ? m.eta[0] : 1 / m.eta[0];
            r.pdf = 1 - F;
            Float factor = importance ? 1.0 : (cosThetaT < 0 ? 1 / m.eta[0] : m.eta[0]);
            r.weight = rgb(m.k) * (factor * factor);
        }
        return r;
    }
    switch (m.type) {
    case MAT_DIFFUSE:
        if (cosTheta(wi) <= 0) return r;
        r.wo = squareToCosineHemisphere(sx, sy);
        r.sampledType = EDiffuseReflection;
        r.pdf = INV_PI * cosTheta(r.wo);
        r.weight = rgb(m.reflectance);
        return r;
    case MAT_CONDUCTOR:
        if (cosTheta(wi) <= 0 || !(typeMask & EDeltaReflection)) return r;
        r.sampledType = EDeltaReflection;
        r.wo = V3(-wi.x, -wi.y, wi.z);
        r.pdf = 1;
        r.weight = rgb(m.reflectance) * fresnelConductorExact(cosTheta(wi), rgb(m.eta), rgb(m.k));
        return r;
    default: {
        if (cosTheta(wi) < 0) return r;
        Microfacet d = distr(m);
        Float temporaryPdf = 0;
        V3 mm = d.sample(wi, sx, sy, temporaryPdf);
        if (temporaryPdf == 0) return r;
        r.wo = 2 * dot(wi, mm) * mm - wi;
        r.sampledType = EGlossyReflection;
        if (cosTheta(r.wo) <= 0) return r;
        V3 F = fresnelConductorExact(dot(wi, mm), rgb(m.eta), rgb(m.k)) * rgb(m.reflectance);
        Float weight;
        if (d.sampleVisible) weight = d.smithG1(r.wo, mm);
        else weight = d.eval(mm) * d.G(wi, r.wo, mm) * dot(wi, mm) / (temporaryPdf * cosTheta(wi));
        if (weight > 0) {
            r.pdf = temporaryPdf / (4.0 * dot(r.wo, mm));
            r.weight = F * weight;
        }
        return r;
    }
    }
}

// TwoSided (src/bsdfs/twosided.cpp:100-168) around the one-sided models, nestedBRDF[1] == nestedBRDF[0]
V3 bsdfEval(const gpo_material &m, V3 wi, V3 wo, int measure, bool importance = false, int typeMask = EDelta | ESmooth)
{
    if (!m.twoSided || cosTheta(wi) > 0) return bsdfEvalOne(m, wi, wo, measure, importance, typeMask);
    wi.z *= -1; wo.z *= -1;
    return bsdfEvalOne(m, wi, wo, measure, importance, typeMask);
}
Float bsdfPdf(const gpo_material &m, V3 wi, V3 wo, int measure, int typeMask = EDelta | ESmooth)
{
    if (!m.twoSided || wi.z > 0) return bsdfPdfOne(m, wi, wo, measure, typeMask);
    wi.z *= -1; wo.z *= -1;
    return bsdfPdfOne(m, wi, wo, measure, typeMask);
}
BSDFSample bsdfSample(const gpo_material &m, V3 wi, Float sx, Float sy, bool importance = false, int typeMask = EDelta | ESmooth)
{
    bool flipped = false;
    if (m.twoSided && cosTheta(wi) < 0) { wi.z *= -1; flipped = true; }
    BSDFSample r = bsdfSampleOne(m, wi, sx, sy, importance, typeMask);
    if (flipped && !isZero(r.weight) && r.pdf != 0) r.wo.z *= -1;
    return r;
}
// DirectSamplingRecord(const Intersection&), records.inl:160-164: refN stays 0 when the BSDF has a back side (twosided)
inline V3 refNormal(const gpo_material &m, const Intersection &its) { return (m.twoSided || m.type == MAT_DIELECTRIC) ? V3(0.0) : its.sh.n; }

// ---- emitters -----------------------------------------------------------------------------------------
struct DirectSamplingRecord { V3 ref, refN, p, n, d; Float dist, pdf; int measure; int object; };

// Intersection::Le -> AreaLight::eval, area.cpp:104-109
V3 Le(const Scene &sc, const Intersection &its, V3 d)
{
    const int e = sc.tris[its.prim].emitter;
    if (e < 0) return V3(0.0);
    if (dot(its.sh.n, d) <= 0) return V3(0.0);
    return sc.emitters[e].radiance;
}

// solveQuadratic, util.cpp:447-485
bool solveQuadratic(Float a, Float b, Float c, Float &x0, Float &x1)
{
    if (a == 0) {
        if (b != 0) { x0 = x1 = -c / b; return true; }
        return false;
    }
    Float discrim = b * b - 4.0 * a * c;
    if (discrim < 0) return false;
    Float temp, sqrtDiscrim = std::sqrt(discrim);
    if (b < 0) temp = -0.5 * (b - sqrtDiscrim);
    else temp = -0.5 * (b + sqrtDiscrim);
    x0 = temp / a;
    x1 = c / temp;
    if (x0 > x1) std::swap(x0, x1);
    return true;
}
// BSphere::rayIntersect, bsphere.h:88-95
bool bsphereRayIntersect(const Scene &sc, V3 ro, V3 rd, Float &nearHit, Float &farHit)
{
    V3 o = ro - sc.bsCenter;
    Float A = lengthSquared(rd), B = 2 * dot(o, rd), C = lengthSquared(o) - sc.bsRadius * sc.bsRadius;
    return solveQuadratic(A, B, C, nearHit, farHit);
}
V3 squareToUniformSphere(Float sx, Float sy)
{ // warp.cpp:25-31
    Float z = 1.0 - 2.0 * sy;
    Float r = safe_sqrt(1.0 - z * z);
    Float phi = 2.0 * PI * sx;
    return V3(r * std::cos(phi), r * std::sin(phi), z);
}
// ---- EnvironmentMap, src/emitters/envmap.cpp ----------------------------------------------------------------------------------
inline V3 mul3(const Float *M, V3 v) { return V3(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z); }
inline Float luminance(const Float *c) { return c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f; }     // spectrum.h:725-727
// configure(), envmap.cpp:258-325: marginal and conditional cdfs over luminance x sin(theta), in FLOAT as the reference keeps them
void envMapConfigure(Scene::EnvMap &e)
{
    const mip_oracle::Level &L0 = e.mip.pyramid[0];
    e.w = L0.w; e.h = L0.h;
    e.cdfCols.assign((size_t)(e.w + 1) * e.h, 0.0f);
    e.cdfRows.assign((size_t)e.h + 1, 0.0f);
    e.rowWeights.assign(e.h, 0.0);
    size_t colPos = 0, rowPos = 0;
    Float rowSum = 0.0f;
    e.cdfRows[rowPos++] = 0;
    for (int y = 0; y < e.h; ++y) {
        Float colSum = 0;
        e.cdfCols[colPos++] = 0;
        for (int x = 0; x < e.w; ++x) {
            colSum += luminance(&L0.rgb[((size_t)y * e.w + x) * 3]);
            e.cdfCols[colPos++] = (float)colSum;
        }
        const float normalization = 1.0f / (float)colSum;
        for (int x = 1; x < e.w; ++x) e.cdfCols[colPos - x - 1] *= normalization;
        e.cdfCols[colPos - 1] = 1.0f;
        const Float weight = std::sin((y + 0.5f) * M_PI / e.h);
        e.rowWeights[y] = weight;
        rowSum += colSum * weight;
        e.cdfRows[rowPos++] = (float)rowSum;
    }
    const float normalization = 1.0f / (float)rowSum;
    for (int y = 1; y < e.h; ++y) e.cdfRows[rowPos - y - 1] *= normalization;
    e.cdfRows[rowPos - 1] = 1.0f;
    e.normalization = 1.0f / (rowSum * (2 * M_PI / e.w) * (M_PI / e.h));
    e.pixelSizeX = 2 * M_PI / e.w; e.pixelSizeY = M_PI / e.h;
}
// evalEnvironment, envmap.cpp:378-409: bilinear on level 0, or -- a camera ray -- the EWA lookup with the partials of (u, v) along the differentials
V3 envMapEval(const Scene &sc, const Ray &ray)
{
    const Scene::EnvMap &e = sc.envMap;
    const V3 v = mul3(e.toLocal, ray.d);
    const Float uvx = std::atan2(v.x, -v.z) * INV_TWOPI, uvy = std::acos(std::min(1.0, std::max(-1.0, v.y))) * INV_PI;
    Float o[3];
    if (!ray.hasDifferentials) e.mip.evalBilinear(0, uvx, uvy, o);
    else {
        const V3 dvdx = mul3(e.toLocal, ray.rxD) - v, dvdy = mul3(e.toLocal, ray.ryD) - v;
        const Float t1 = INV_TWOPI / (v.x * v.x + v.z * v.z), t2 = -INV_PI / std::max(safe_sqrt(1.0f - v.y * v.y), Epsilon);
        e.mip.eval(uvx, uvy, t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y, t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y, o);
    }
    return V3(o[0], o[1], o[2]) * e.scale;
}
inline uint32_t envSampleReuse(const float *cdf, uint32_t size, Float &sample)
{ // envmap.cpp:640-645
    const float *entry = std::lower_bound(cdf, cdf + size + 1, (float)sample);
    const uint32_t index = std::min((uint32_t)std::max((ptrdiff_t)0, entry - cdf - 1), size - 1);
    sample = (sample - (Float)cdf[index]) / (Float)(cdf[index + 1] - cdf[index]);
    return index;
}
inline Float intervalToTent(Float sample)
{ // warp.cpp:143-155
    Float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; } else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - std::sqrt(sample));
}
// internalSampleDirection, envmap.cpp:556-594
void envMapSampleDirection(const Scene::EnvMap &e, Float sx, Float sy, V3 &d, V3 &value, Float &pdf)
{
    const uint32_t row = envSampleReuse(e.cdfRows.data(), e.h, sy), col = envSampleReuse(e.cdfCols.data() + (size_t)row * (e.w + 1), e.w, sx);
    const Float posx = (Float)col + intervalToTent(sx), posy = (Float)row + intervalToTent(sy);
    const int xPos = (int)std::floor(posx), yPos = (int)std::floor(posy);
    const Float dx1 = posx - xPos, dx2 = 1.0f - dx1, dy1 = posy - yPos, dy2 = 1.0f - dy1;
    Float a[3], b[3], c[3], dd[3], value1[3], value2[3];
    e.mip.texel(0, xPos, yPos, a); e.mip.texel(0, xPos + 1, yPos, b); e.mip.texel(0, xPos, yPos + 1, c); e.mip.texel(0, xPos + 1, yPos + 1, dd);
    for (int k = 0; k < 3; ++k) { value1[k] = a[k] * dx2 * dy2 + b[k] * dx1 * dy2; value2[k] = c[k] * dx2 * dy1 + dd[k] * dx1 * dy1; }
    value = V3(value1[0] + value2[0], value1[1] + value2[1], value1[2] + value2[2]) * e.scale;
    pdf = (luminance(value1) * e.rowWeights[std::min(std::max(yPos, 0), e.h - 1)] + luminance(value2) * e.rowWeights[std::min(std::max(yPos + 1, 0), e.h - 1)]) * e.normalization;
    const Float phi = e.pixelSizeX * (posx + 0.5f), theta = e.pixelSizeY * (posy + 0.5f);
    const Float sinPhi = std::sin(phi), cosPhi = std::cos(phi), sinTheta = std::sin(theta), cosTheta = std::cos(theta);
    d = V3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    pdf /= std::max(std::abs(sinTheta), Epsilon);
}
// internalPdfDirection, envmap.cpp:597-625
Float envMapPdfDirection(const Scene::EnvMap &e, V3 d)
{
    const Float uvx = std::atan2(d.x, -d.z) * INV_TWOPI, uvy = std::acos(std::min(1.0, std::max(-1.0, d.y))) * INV_PI;
    if (!std::isfinite(uvx) || !std::isfinite(uvy)) return 0.0;
    const Float u = uvx * e.w - 0.5f, v = uvy * e.h - 0.5f;
    const int xPos = (int)std::floor(u), yPos = (int)std::floor(v);
    const Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    Float a[3], b[3], c[3], dd[3], value1[3], value2[3];
    e.mip.texel(0, xPos, yPos, a); e.mip.texel(0, xPos + 1, yPos, b); e.mip.texel(0, xPos, yPos + 1, c); e.mip.texel(0, xPos + 1, yPos + 1, dd);
    for (int k = 0; k < 3; ++k) { value1[k] = a[k] * dx2 * dy2 + b[k] * dx1 * dy2; value2[k] = c[k] * dx2 * dy1 + dd[k] * dx1 * dy1; }
    const Float sinTheta = safe_sqrt(1 - d.y * d.y);
    return (luminance(value1) * e.rowWeights[std::min(std::max(yPos, 0), e.h - 1)] + luminance(value2) * e.rowWeights[std::min(std::max(yPos + 1, 0), e.h - 1)])
        * e.normalization / std::max(std::abs(sinTheta), Epsilon);
}
// Scene::evalEnvironment for whichever environment emitter the scene has
inline V3 evalEnvironment(const Scene &sc, const Ray &ray) { return sc.envMap.present ? envMapEval(sc, ray) : sc.emitters[sc.envIndex].radiance; }

// ConstantBackgroundEmitter::fillDirectSamplingRecord, constant.cpp:245-261
bool envFillDirectSamplingRecord(const Scene &sc, DirectSamplingRecord &dRec, const Ray &ray)
{
    Float nearT, farT;
    if (!bsphereRayIntersect(sc, ray.o, ray.d, nearT, farT) || nearT > 0 || farT < 0) return false;
    dRec.p = ray.o + ray.d * farT;
    dRec.n = normalize(sc.bsCenter - dRec.p);
    dRec.measure = MEASURE_SOLID_ANGLE;
    dRec.object = sc.envIndex;
    dRec.d = ray.d;
    dRec.dist = farT;
    return true;
}
// ConstantBackgroundEmitter::sampleDirect, constant.cpp:179-219
bool bsphereRayIntersect(const Scene &sc, V3 ro, V3 rd, Float &nearHit, Float &farHit);
V3 envSampleDirect(const Scene &sc, const Emitter &em, DirectSamplingRecord &dRec, Float sx, Float sy)
{
    if (sc.envMap.present) {                                      // EnvironmentMap::sampleDirect, envmap.cpp:509-534
        V3 value, dl; Float pdf;
        envMapSampleDirection(sc.envMap, sx, sy, dl, value, pdf);
        const V3 dw = mul3(sc.envMap.toWorld, dl);
        Float nearT, farT;
        dRec.d = dw; dRec.dist = 0.0; dRec.p = dRec.ref; dRec.n = V3(0.0); dRec.measure = MEASURE_SOLID_ANGLE;
        if (isZero(value) || pdf == 0 || !bsphereRayIntersect(sc, dRec.ref, dw, nearT, farT) || nearT >= 0 || farT <= 0) { dRec.pdf = 0.0; return V3(0.0); }
        dRec.pdf = pdf;
        dRec.p = dRec.ref + dw * farT;
        dRec.n = normalize(sc.bsCenter - dRec.p);
        dRec.dist = farT;
        return value / pdf;
    }
    V3 d;
    Float pdf;
    const bool hasN = !(dRec.refN.x == 0 && dRec.refN.y == 0 && dRec.refN.z == 0);
    if (hasN) {
        d = squareToCosineHemisphere(sx, sy);
        pdf = INV_PI * d.z;                                       // squareToCosineHemispherePdf, warp.h
        Frame f; f.n = dRec.refN; coordinateSystem(f.n, f.s, f.t); // Frame(n), frame.h
        d = f.toWorld(d);
    } else {
        d = squareToUniformSphere(sx, sy);
        pdf = 1.0 / (4.0 * PI);
    }
    Float nearT, farT;
    dRec.pdf = 0.0;
    // (the reference leaves d/dist unset on these two early exits; they cannot be taken from inside the sphere)
    dRec.d = d; dRec.dist = 0.0; dRec.p = dRec.ref; dRec.n = V3(0.0); dRec.measure = MEASURE_SOLID_ANGLE;
    if (!bsphereRayIntersect(sc, dRec.ref, d, nearT, farT)) return V3(0.0);
    if (!(nearT < 0 && farT > 0)) return V3(0.0);
    dRec.p = dRec.ref + d * farT;
    dRec.n = normalize(sc.bsCenter - dRec.p);
    dRec.dist = farT;
    dRec.pdf = pdf;
    if (hasN && dot(dRec.d, dRec.refN) <= 0) return V3(0.0);
    return em.radiance / pdf;
}
// ConstantBackgroundEmitter::pdfDirect, constant.cpp:221-236
Float envPdfDirect(const Scene &sc, const DirectSamplingRecord &dRec)
{
    if (sc.envMap.present) {                                      // EnvironmentMap::pdfDirect, envmap.cpp:536-547
        const Float pdfSA = envMapPdfDirection(sc.envMap, mul3(sc.envMap.toLocal, dRec.d));
        return dRec.measure == MEASURE_SOLID_ANGLE ? pdfSA : 0.0;
    }
    const bool hasN = !(dRec.refN.x == 0 && dRec.refN.y == 0 && dRec.refN.z == 0);
    Float pdfSA = hasN ? INV_PI * std::max((Float)0.0, dot(dRec.d, dRec.refN)) : 1.0 / (4.0 * PI);
    if (dRec.measure == MEASURE_SOLID_ANGLE) return pdfSA;
    return 0.0;
}

// Scene::sampleEmitterDirectVisible, scene.cpp:855-879 -> AreaLight::sampleDirect (area.cpp:158-172) ->
// Shape::sampleDirect (shape.cpp:102-116) -> TriMesh::samplePosition (trimesh.cpp:412-423) -> Triangle::sample (triangle.cpp:24-)
V3 sampleEmitterDirectVisible(const Scene &sc, DirectSamplingRecord &dRec, Float sx, Float sy, bool &visible)
{
    Float emPdf;
    size_t index = sc.emitterPDF.sampleReuse(sx, emPdf);
    const Emitter &em = sc.emitters[index];
    V3 value;
    if (em.numTris == 0) {
        value = envSampleDirect(sc, em, dRec, sx, sy);
    } else if (em.numTris < 0) {                                   // PointEmitter::sampleDirect, point.cpp:120-134
        dRec.p = em.position;
        dRec.pdf = 1.0;
        dRec.measure = MEASURE_DISCRETE;
        dRec.d = dRec.p - dRec.ref;
        dRec.dist = length(dRec.d);
        Float invDist = 1.0 / dRec.dist;
        dRec.d = dRec.d * invDist;
        dRec.n = V3(0.0);
        value = em.radiance * (invDist * invDist);
    } else {
    if (em.rectangle) {                                            // Rectangle::samplePosition, rectangle.cpp:200-206
        const Float lx = sx * 2 - 1, ly = sy * 2 - 1;
        const Float *M = em.rect;                                  // Transform::operator()(Point), transform.h:108-124 (affine: w == 1)
        dRec.p = V3(M[0] * lx + M[1] * ly + M[2] * 0.0 + M[3], M[4] * lx + M[5] * ly + M[6] * 0.0 + M[7], M[8] * lx + M[9] * ly + M[10] * 0.0 + M[11]);
        dRec.n = em.rectN;
        dRec.pdf = em.invSurfaceArea;
    } else
    // TriMesh::samplePosition
    {
        const std::vector<Float> &cdf = em.cdf;
        auto entry = std::lower_bound(cdf.begin(), cdf.end(), sy);
        size_t ti = std::min(cdf.size() - 2, (size_t)std::max((std::ptrdiff_t)0, (std::ptrdiff_t)(entry - cdf.begin()) - 1));
        while ((cdf[ti + 1] - cdf[ti]) == 0 && ti < cdf.size() - 1) ++ti;
        sy = (sy - cdf[ti]) / (cdf[ti + 1] - cdf[ti]);
        const Tri &tr = sc.tris[em.firstTri + ti];
        Float a = safe_sqrt(1.0 - sx);             // warp.cpp:76-79 squareToUniformTriangle
        Float bx = 1 - a, by = a * sy;
        V3 sideA = tr.p1 - tr.p0, sideB = tr.p2 - tr.p0;
        dRec.p = tr.p0 + (sideA * bx) + (sideB * by);
        dRec.n = normalize(cross(sideA, sideB));
        dRec.pdf = em.invSurfaceArea;
    }
    // Shape::sampleDirect
    dRec.d = dRec.p - dRec.ref;
    Float distSquared = lengthSquared(dRec.d);
    dRec.dist = std::sqrt(distSquared);
    dRec.d = dRec.d / dRec.dist;
    Float dp = std::abs(dot(dRec.d, dRec.n));
    dRec.pdf *= dp != 0 ? (distSquared / dp) : 0.0;
    dRec.measure = MEASURE_SOLID_ANGLE;
    // AreaLight::sampleDirect
    if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0) value = em.radiance / dRec.pdf;
    else { dRec.pdf = 0.0; value = V3(0.0); }
    }
    dRec.object = (int)index;
    dRec.pdf *= emPdf;
    value = value / emPdf;
    Ray ray(dRec.ref, dRec.d, Epsilon, dRec.dist * (1 - ShadowEpsilon));
    if (rayIntersectShadow(sc, ray)) { visible = false; return V3(0.0); }
    visible = true;
    return value;
}

// Scene::pdfEmitterDirect, scene.cpp:976-979 -> AreaLight::pdfDirect (area.cpp:174-183) -> Shape::pdfDirect (shape.cpp:118-126)
Float pdfEmitterDirect(const Scene &sc, const DirectSamplingRecord &dRec)
{
    const Emitter &em = sc.emitters[dRec.object];
    Float pd = 0.0;
    if (em.numTris == 0) pd = envPdfDirect(sc, dRec);
    else if (em.numTris < 0) pd = dRec.measure == MEASURE_DISCRETE ? 1.0 : 0.0;   // point.cpp:136-138
    else if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0) {
        Float pdfPos = em.invSurfaceArea;
        if (dRec.measure == MEASURE_SOLID_ANGLE) pd = pdfPos * (dRec.dist * dRec.dist) / std::abs(dot(dRec.d, dRec.n));
        else pd = 0.0;
    }
    return pd * (1.0 * sc.emitterPDF.normalization); // Scene::pdfEmitterDiscrete, scene.h:855-857
}

// testEnvironmentVisibility + environmentShift, gpt.cpp:96-114,348-369
struct EnvShiftResult { bool success; Float jacobian; V3 wo; };
EnvShiftResult environmentShift(const Scene &sc, const Ray &mainRay, V3 shiftSourceVertex)
{
    EnvShiftResult r; r.success = false; r.jacobian = 1; r.wo = mainRay.d;
    if (sc.envIndex < 0) return r;
    Ray shadowRay(shiftSourceVertex, mainRay.d);
    DirectSamplingRecord rec;
    rec.dist = 0.0;
    envFillDirectSamplingRecord(sc, rec, shadowRay);
    shadowRay.mint = Epsilon;
    shadowRay.maxt = (1.0 - ShadowEpsilon) * rec.dist;
    r.success = !rayIntersectShadow(sc, shadowRay);
    return r;
}

// ---- sensor: PerspectiveCameraImpl::sampleRayDifferential, perspective.cpp:271-298 -----------------------
// ... and ThinLensCamera::sampleRayDifferential, thinlens.cpp:324-361 (apx, apy = the aperture sample, ignored by the pinhole)
void sampleRay(const Scene &sc, Float px, Float py, Ray &ray, Float apx = 0.5, Float apy = 0.5)
{
    const gpo_camera &c = sc.cam;
    // The crop window (perspective.cpp:126-156): pixelSample is relative to the crop, m_invResolution = 1 / cropSize, and steps 4+5 of m_cameraToSample
    // map the full film's [0, 1]^2 to the crop's -- so the position in the FULL film's unit square is (crop-relative sample + cropOffset) / filmSize;
    // m_aspect (sensor.cpp) and, through relSize * invResolution, the pixel steps m_dx / m_dy are the full film's too.
    const int fullW = c.fullWidth > 0 ? c.fullWidth : c.width, fullH = c.fullWidth > 0 ? c.fullHeight : c.height;
    const Float cropX = c.fullWidth > 0 ? (Float)c.cropOffsetX : 0.0, cropY = c.fullWidth > 0 ? (Float)c.cropOffsetY : 0.0;
    const Float sxn = (px + cropX) * (1.0 / fullW), syn = (py + cropY) * (1.0 / fullH);
    // m_sampleToCamera(Point(sx, sy, 0)) for the composite of perspective.cpp:150-156, written out:
    V3 nearP((1 - 2 * sxn) * c.nearClip * sc.tanHalf, (1 - 2 * syn) / sc.aspect * c.nearClip * sc.tanHalf, c.nearClip);
    // m_dx = sampleToCamera(1/width, 0, 0) - sampleToCamera(0), m_dy likewise (perspective.cpp:160-163, thinlens.cpp:171-174), with the same written-out composite
    const V3 mdx(-2 * (1.0 / fullW) * c.nearClip * sc.tanHalf, 0.0, 0.0), mdy(0.0, -2 * (1.0 / fullH) / sc.aspect * c.nearClip * sc.tanHalf, 0.0);
    const double *M = c.toWorld;
    V3 d, dx, dy, ol(0.0);
    if (c.type == 1) {
        Float tx, ty;
        squareToUniformDiskConcentric(apx, apy, tx, ty);                    // thinlens.cpp:326-327
        ol = V3(tx * c.apertureRadius, ty * c.apertureRadius, 0.0);         // apertureP
        const Float fDist = c.focusDistance / nearP.z;                      // :340-343
        const V3 focusP = nearP * fDist, focusPx = (nearP + mdx) * fDist, focusPy = (nearP + mdy) * fDist;
        d = normalize(focusP - ol);
        dx = normalize(focusPx - ol); dy = normalize(focusPy - ol);         // :356-357
    } else {
        d = normalize(nearP);
        dx = normalize(nearP + mdx); dy = normalize(nearP + mdy);           // perspective.cpp:293-294
    }
    Float invZ = 1.0 / d.z;
    ray.mint = c.nearClip * invZ;
    ray.maxt = c.farClip * invZ;
    ray.o = V3(M[0] * ol.x + M[1] * ol.y + M[2] * ol.z + M[3], M[4] * ol.x + M[5] * ol.y + M[6] * ol.z + M[7], M[8] * ol.x + M[9] * ol.y + M[10] * ol.z + M[11]);   // trafo.transformAffine(apertureP); rxOrigin = ryOrigin = o
    ray.d = V3(M[0] * d.x + M[1] * d.y + M[2] * d.z, M[4] * d.x + M[5] * d.y + M[6] * d.z, M[8] * d.x + M[9] * d.y + M[10] * d.z);
    ray.rxD = V3(M[0] * dx.x + M[1] * dx.y + M[2] * dx.z, M[4] * dx.x + M[5] * dx.y + M[6] * dx.z, M[8] * dx.x + M[9] * dx.y + M[10] * dx.z);
    ray.ryD = V3(M[0] * dy.x + M[1] * dy.y + M[2] * dy.z, M[4] * dy.x + M[5] * dy.y + M[6] * dy.z, M[8] * dy.x + M[9] * dy.y + M[10] * dy.z);
    ray.hasDifferentials = true;
}

// ================================================================================================================
// gpt.cpp
// ================================================================================================================
enum VertexType { VERTEX_TYPE_GLOSSY, VERTEX_TYPE_DIFFUSE };                         // gpt.cpp:122-125
enum RayConnection { RAY_NOT_CONNECTED, RAY_RECENTLY_CONNECTED, RAY_CONNECTED };    // gpt.cpp:127-131

struct RayState { // gpt.cpp:135-173
    Ray ray;
    V3 throughput;
    Float pdf;
    V3 radiance, gradient;
    Intersection its;
    Float eta;
    bool alive;
    RayConnection connection_status;
    RayState() : throughput(0.0), pdf(1.0), radiance(0.0), gradient(0.0), eta(1.0), alive(true), connection_status(RAY_NOT_CONNECTED) {}
    void addRadiance(V3 c, Float w) { radiance = radiance + c * w; }
    void addGradient(V3 c, Float w) { gradient = gradient + c * w; }
};

// getVertexType, gpt.cpp:176-231, for single-component BSDFs
VertexType getVertexType(const gpo_material &m, const gpo_config &cfg, unsigned bsdfTypeMask)
{
    Float lowest = INF;
    bool found_smooth = false, found_dirac = false;
    Float r = getRoughness(m);
    bool skip = false;
    if (r == 0) { found_dirac = true; if (!(bsdfTypeMask & EDelta)) skip = true; }
    else found_smooth = true;
    if (!skip && r < lowest) lowest = r;
    if (!found_smooth && found_dirac && !(bsdfTypeMask & EDelta)) lowest = 0;
    return lowest <= cfg.shiftThreshold ? VERTEX_TYPE_GLOSSY : VERTEX_TYPE_DIFFUSE;
}

struct HalfVectorShiftResult { bool success; Float jacobian; V3 wo; };
V3 reflect(V3 wi, V3 n) { return 2 * dot(wi, n) * n - wi; } // util.cpp:763
V3 refract(V3 wi, V3 n, Float eta)
{ // util.cpp:774-792
    if (eta == 1) return -wi;
    Float cosThetaI = dot(wi, n);
    if (cosThetaI > 0) eta = 1 / eta;
    Float cosThetaTSqr = 1 - (1 - cosThetaI * cosThetaI) * (eta * eta);
    if (cosThetaTSqr <= 0.0) return V3(0.0);
    return n * (cosThetaI * eta - signum(cosThetaI) * std::sqrt(cosThetaTSqr)) - wi * eta;
}

// halfVectorShift, gpt.cpp:242-305
HalfVectorShiftResult halfVectorShift(V3 mainWi, V3 mainWo, V3 shiftedWi, Float mainEta, Float shiftedEta)
{
    HalfVectorShiftResult result;
    result.success = false; result.jacobian = 0; result.wo = V3(0.0);
    if (cosTheta(mainWi) * cosTheta(mainWo) < 0) {
        if (mainEta == 1 || shiftedEta == 1) return result;
        V3 hMain = cosTheta(mainWi) < 0 ? -(mainWi * mainEta + mainWo) : -(mainWi + mainWo * mainEta);
        V3 h = normalize(hMain);
        V3 shiftedWo = refract(shiftedWi, h, shiftedEta);
        if (isZero(shiftedWo)) return result;
        V3 hShifted = cosTheta(shiftedWi) < 0 ? -(shiftedWi * shiftedEta + shiftedWo) : -(shiftedWi + shiftedWo * shiftedEta);
        Float hLengthSquared = lengthSquared(hShifted) / (D_EPSILON + lengthSquared(hMain));
        Float WoDotH = std::abs(dot(mainWo, h)) / (D_EPSILON + std::abs(dot(shiftedWo, h)));
        result.success = true; result.wo = shiftedWo; result.jacobian = hLengthSquared * WoDotH;
    } else {
        V3 h = normalize(mainWi + mainWo);
        V3 shiftedWo = reflect(shiftedWi, h);
        Float WoDotH = dot(shiftedWo, h) / dot(mainWo, h);
        result.success = true; result.wo = shiftedWo; result.jacobian = std::abs(WoDotH);
    }
    return result;
}

// testVisibility, gpt.cpp:84-93
bool testVisibility(const Scene &sc, V3 p1, V3 p2)
{
    Ray shadowRay(p1, p2 - p1, Epsilon, 1.0 - ShadowEpsilon);
    return !rayIntersectShadow(sc, shadowRay);
}

struct ReconnectionShiftResult { bool success; Float jacobian; V3 wo; };
// reconnectShift, gpt.cpp:316-345
ReconnectionShiftResult reconnectShift(const Scene &sc, V3 mainSourceVertex, V3 targetVertex, V3 shiftSourceVertex, V3 targetNormal)
{
    ReconnectionShiftResult result;
    result.success = false; result.jacobian = 0; result.wo = V3(0.0);
    if (!testVisibility(sc, shiftSourceVertex, targetVertex)) return result;
    V3 mainEdge = mainSourceVertex - targetVertex, shiftedEdge = shiftSourceVertex - targetVertex;
    Float mainEdgeLengthSquared = lengthSquared(mainEdge), shiftedEdgeLengthSquared = lengthSquared(shiftedEdge);
    V3 shiftedWo = -shiftedEdge / std::sqrt(shiftedEdgeLengthSquared);
    Float mainOpposingCosine = dot(mainEdge, targetNormal) / std::sqrt(mainEdgeLengthSquared);
    Float shiftedOpposingCosine = dot(shiftedWo, targetNormal);
    result.jacobian = std::abs(shiftedOpposingCosine * mainEdgeLengthSquared) / (D_EPSILON + std::abs(mainOpposingCosine * shiftedEdgeLengthSquared));
    result.success = true;
    result.wo = shiftedWo;
    return result;
}

// its.getBSDF(): the material of the hit, with a textured reflectance resolved at its.uv (diffuse.cpp:107,116,137,149: m_reflectance->eval(its);
// likewise specularReflectance of the conductors and the dielectric)
// Intersection::computePartials, intersection.cpp:5-78: the texture coordinates' partials with respect to a one-pixel step on the screen
inline void computePartials(Intersection &its, const Ray &ray)
{
    if (its.hasUVPartials || !ray.hasDifferentials) return;
    its.hasUVPartials = true;
    if (isZero(its.dpdu) && isZero(its.dpdv)) { its.dudx = its.dvdx = its.dudy = its.dvdy = 0.0; return; }
    const V3 n = its.geoN;                                                               // geoFrame.n
    const Float pp = dot(n, its.p), pox = dot(n, ray.o), poy = dot(n, ray.o), prx = dot(n, ray.rxD), pry = dot(n, ray.ryD);
    if (prx == 0 || pry == 0) { its.dudx = its.dvdx = its.dudy = its.dvdy = 0.0; return; }
    const Float tx = (pp - pox) / prx, ty = (pp - poy) / pry;
    const Float absX = std::abs(n.x), absY = std::abs(n.y), absZ = std::abs(n.z);
    int axes[2];
    if (absX > absY && absX > absZ) { axes[0] = 1; axes[1] = 2; }
    else if (absY > absZ) { axes[0] = 0; axes[1] = 2; }
    else { axes[0] = 0; axes[1] = 1; }
    auto comp = [](const V3 &v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); };
    const Float A[2][2] = {{comp(its.dpdu, axes[0]), comp(its.dpdv, axes[0])}, {comp(its.dpdu, axes[1]), comp(its.dpdv, axes[1])}};
    const V3 px = ray.o + ray.rxD * tx, py = ray.o + ray.ryD * ty;
    const Float Bx[2] = {comp(px, axes[0]) - comp(its.p, axes[0]), comp(px, axes[1]) - comp(its.p, axes[1])};
    const Float By[2] = {comp(py, axes[0]) - comp(its.p, axes[0]), comp(py, axes[1]) - comp(its.p, axes[1])};
    const Float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];                             // solveLinearSystem2x2, util.cpp:527-539
    if (std::abs(det) <= RCPOVERFLOW) {
        its.dudx = 1; its.dvdx = 0;
        its.dudy = 1; its.dvdy = 0;                                                      // (:74-76 writes `dudy = 0; dudy = 1;` and leaves dvdy as it was: taken as 0 here)
        return;
    }
    const Float inverse = 1.0 / det;
    its.dudx = (A[1][1] * Bx[0] - A[0][1] * Bx[1]) * inverse; its.dvdx = (A[0][0] * Bx[1] - A[1][0] * Bx[0]) * inverse;
    its.dudy = (A[1][1] * By[0] - A[0][1] * By[1]) * inverse; its.dvdy = (A[0][0] * By[1] - A[1][0] * By[0]) * inverse;
}

// its.getBSDF(ray): the material of the hit, with a textured reflectance resolved at its.uv (diffuse.cpp:107,116,137,149: m_reflectance->eval(its);
// likewise specularReflectance of the conductors and the dielectric).  A BSDF with a bitmap texture usesRayDifferentials(), so the hit of
// a camera ray gets its UV partials first (shape.h getBSDF(ray)) and Texture2D::eval(its) passes them to the MIP map (texture.cpp:112-121).
inline gpo_material matOf(const Scene &sc, Intersection &its, const Ray &ray)
{
    const int mi = sc.tris[its.prim].material;
    gpo_material m = sc.mats[mi];
    const int ti = mi < (int)sc.matTexture.size() ? sc.matTexture[mi] : -1;
    if (ti >= 0) {
        computePartials(its, ray);
        const V3 r = sc.textures[ti].eval(its.u, its.v, its.hasUVPartials, its.dudx, its.dudy, its.dvdx, its.dvdy);
        m.reflectance[0] = r.x; m.reflectance[1] = r.y; m.reflectance[2] = r.z;
    }
    return m;
}

// GradientPathTracer::evaluate, gpt.cpp:468-1180 (no sub-surface scattering in the carried subset; the environment emitter is `constant`)
void evaluate(const Scene &sc, const gpo_config &cfg, Rng &rng, RayState &main, RayState *shiftedRays, int secondaryCount, V3 &out_veryDirect)
{
    rayIntersect(sc, main.ray, main.its);                                                  // :472
    main.ray.mint = Epsilon;
    for (int i = 0; i < secondaryCount; ++i) { rayIntersect(sc, shiftedRays[i].ray, shiftedRays[i].its); shiftedRays[i].ray.mint = Epsilon; }
    if (!main.its.isValid()) {                                                             // :482-492
        if (sc.envIndex >= 0) out_veryDirect = out_veryDirect + main.throughput * evalEnvironment(sc, main.ray); // scene->evalEnvironment(main.ray): a camera ray, with differentials
        return;
    }
    if (sc.tris[main.its.prim].emitter >= 0) out_veryDirect = out_veryDirect + main.throughput * Le(sc, main.its, -main.ray.d); // :497-499
    for (int i = 0; i < secondaryCount; ++i) if (!shiftedRays[i].its.isValid()) shiftedRays[i].alive = false; // :508-513
    if (cfg.strictNormals) {                                                               // :516-531
        if (dot(main.ray.d, main.its.geoN) * cosTheta(main.its.wi) >= 0) return;
        for (int i = 0; i < secondaryCount; ++i) {
            RayState &s = shiftedRays[i];
            if (s.its.isValid() && dot(s.ray.d, s.its.geoN) * cosTheta(s.its.wi) >= 0) s.alive = false;
            else if (!s.its.isValid()) s.alive = false;
        }
    }
    int depth = 1;                                                                         // :535
    while (depth < cfg.maxDepth || cfg.maxDepth < 0) {                                     // :537
        if (cfg.strictNormals) {                                                           // :541-556
            if (dot(main.ray.d, main.its.geoN) * cosTheta(main.its.wi) >= 0) return;
            for (int i = 0; i < secondaryCount; ++i) {
                RayState &s = shiftedRays[i];
                if (s.alive && dot(s.ray.d, s.its.geoN) * cosTheta(s.its.wi) >= 0) s.alive = false;
            }
        }
        const bool lastSegment = (depth + 1 == cfg.maxDepth);                              // :559
        const gpo_material &mainBSDF = matOf(sc, main.its, main.ray);

        // ---- direct illumination sampling, :565-730 (minDepth is forced to 1, gpt.cpp:1369) ----
        if (bsdfType(mainBSDF) & ESmooth) {
            DirectSamplingRecord dRec;                                                     // records.inl:160-164
            dRec.ref = main.its.p; dRec.refN = refNormal(mainBSDF, main.its);
            const Float lsx = rng.next1D(), lsy = rng.next1D();                            // :572
            bool mainEmitterVisible;
            V3 value = sampleEmitterDirectVisible(sc, dRec, lsx, lsy, mainEmitterVisible);
            V3 mainEmitterRadiance = value * dRec.pdf;                                     // :575
            const V3 mainWoL = main.its.sh.toLocal(dRec.d);
            V3 mainBSDFValue = bsdfEval(mainBSDF, main.its.wi, mainWoL, MEASURE_SOLID_ANGLE); // :588
            const bool emitterOnSurface = sc.emitters[dRec.object].onSurface();
            Float mainBsdfPdf = (emitterOnSurface && dRec.measure == MEASURE_SOLID_ANGLE && mainEmitterVisible) ? bsdfPdf(mainBSDF, main.its.wi, mainWoL, MEASURE_SOLID_ANGLE) : 0; // :592
            Float mainDistanceSquared = lengthSquared(main.its.p - dRec.p);
            Float mainOpposingCosine = dot(dRec.n, (main.its.p - dRec.p)) / std::sqrt(mainDistanceSquared);
            Float mainWeightNumerator = main.pdf * dRec.pdf;                               // :599
            Float mainWeightDenominator = (main.pdf * main.pdf) * ((dRec.pdf * dRec.pdf) + (mainBsdfPdf * mainBsdfPdf));
            if (!cfg.strictNormals || dot(main.its.geoN, dRec.d) * cosTheta(mainWoL) > 0) { // :607
                for (int i = 0; i < secondaryCount; ++i) {
                    RayState &shifted = shiftedRays[i];
                    V3 mainContribution(0.0), shiftedContribution(0.0);
                    Float weight = 0;
                    bool shiftSuccessful = shifted.alive;
                    if (shiftSuccessful) {
                        if (shifted.connection_status == RAY_CONNECTED) {                  // :622-637
                            Float shiftedBsdfPdf = mainBsdfPdf, shiftedDRecPdf = dRec.pdf, jacobian = 1;
                            Float shiftedWeightDenominator = (jacobian * shifted.pdf) * (jacobian * shifted.pdf) * ((shiftedDRecPdf * shiftedDRecPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                            weight = mainWeightNumerator / (D_EPSILON + shiftedWeightDenominator + mainWeightDenominator);
                            mainContribution = main.throughput * (mainBSDFValue * mainEmitterRadiance);
                            shiftedContribution = jacobian * shifted.throughput * (mainBSDFValue * mainEmitterRadiance);
                        } else if (shifted.connection_status == RAY_RECENTLY_CONNECTED) {  // :638-658
                            V3 incomingDirection = normalize(shifted.its.p - main.its.p);
                            V3 wiL = main.its.sh.toLocal(incomingDirection), woL = main.its.sh.toLocal(dRec.d);
                            Float shiftedBsdfPdf = (emitterOnSurface && dRec.measure == MEASURE_SOLID_ANGLE && mainEmitterVisible) ? bsdfPdf(mainBSDF, wiL, woL, MEASURE_SOLID_ANGLE) : 0;
                            Float shiftedDRecPdf = dRec.pdf;
                            V3 shiftedBsdfValue = bsdfEval(mainBSDF, wiL, woL, MEASURE_SOLID_ANGLE);
                            Float jacobian = 1;
                            Float shiftedWeightDenominator = (jacobian * shifted.pdf) * (jacobian * shifted.pdf) * ((shiftedDRecPdf * shiftedDRecPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                            weight = mainWeightNumerator / (D_EPSILON + shiftedWeightDenominator + mainWeightDenominator);
                            mainContribution = main.throughput * (mainBSDFValue * mainEmitterRadiance);
                            shiftedContribution = jacobian * shifted.throughput * (shiftedBsdfValue * mainEmitterRadiance);
                        } else {                                                           // :659-705
                            const gpo_material &shiftedBSDF = matOf(sc, shifted.its, shifted.ray);
                            VertexType mainVertexType = getVertexType(mainBSDF, cfg, ESmooth);
                            VertexType shiftedVertexType = getVertexType(shiftedBSDF, cfg, ESmooth);
                            const bool mainAtPointLight = (dRec.measure == MEASURE_DISCRETE);                        // :667
                            if (g_traceMain) fprintf(stderr, "gpo nee depth %d offset %d: unconnected, main prim %d type %d, shifted prim %d mat type %d vertex type %d\n", depth, i, main.its.prim, (int)mainVertexType, shifted.its.prim, (int)shiftedBSDF.type, (int)shiftedVertexType);
                            if (mainAtPointLight || (mainVertexType == VERTEX_TYPE_DIFFUSE && shiftedVertexType == VERTEX_TYPE_DIFFUSE)) {
                                DirectSamplingRecord shiftedDRec;
                                shiftedDRec.ref = shifted.its.p; shiftedDRec.refN = refNormal(shiftedBSDF, shifted.its);
                                bool shiftedEmitterVisible;
                                V3 sv = sampleEmitterDirectVisible(sc, shiftedDRec, lsx, lsy, shiftedEmitterVisible);
                                V3 shiftedEmitterRadiance = sv * shiftedDRec.pdf;
                                Float shiftedDRecPdf = shiftedDRec.pdf;
                                Float shiftedDistanceSquared = lengthSquared(dRec.p - shifted.its.p);
                                V3 emitterDirection = (dRec.p - shifted.its.p) / std::sqrt(shiftedDistanceSquared);
                                Float shiftedOpposingCosine = -dot(dRec.n, emitterDirection);
                                V3 woL = shifted.its.sh.toLocal(emitterDirection);
                                if (cfg.strictNormals && dot(shifted.its.geoN, emitterDirection) * cosTheta(woL) < 0) {
                                    shiftSuccessful = false;
                                } else {
                                    V3 shiftedBsdfValue = bsdfEval(shiftedBSDF, shifted.its.wi, woL, MEASURE_SOLID_ANGLE);
                                    Float shiftedBsdfPdf = (emitterOnSurface && dRec.measure == MEASURE_SOLID_ANGLE && shiftedEmitterVisible) ? bsdfPdf(shiftedBSDF, shifted.its.wi, woL, MEASURE_SOLID_ANGLE) : 0;   // :693
                                    Float jacobian = std::abs(shiftedOpposingCosine * mainDistanceSquared) / (Epsilon + std::abs(mainOpposingCosine * shiftedDistanceSquared)); // :695 (Epsilon, not D_EPSILON)
                                    Float shiftedWeightDenominator = (jacobian * shifted.pdf) * (jacobian * shifted.pdf) * ((shiftedDRecPdf * shiftedDRecPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                                    weight = mainWeightNumerator / (D_EPSILON + shiftedWeightDenominator + mainWeightDenominator);
                                    mainContribution = main.throughput * (mainBSDFValue * mainEmitterRadiance);
                                    shiftedContribution = jacobian * shifted.throughput * (shiftedBsdfValue * shiftedEmitterRadiance);
                                }
                            }
                        }
                    }
                    if (!shiftSuccessful) {                                                // :708-717
                        weight = mainWeightNumerator / (D_EPSILON + mainWeightDenominator);
                        mainContribution = main.throughput * (mainBSDFValue * mainEmitterRadiance);
                        shiftedContribution = V3(0.0);
                    }
                    main.addRadiance(mainContribution, weight);                            // :723-726
                    shifted.addRadiance(shiftedContribution, weight);
                    shifted.addGradient(shiftedContribution - mainContribution, weight);
                }
            }
        }

        // ---- BSDF sampling and emitter hits, :737-1151 ----
        const Float bsx = rng.next1D(), bsy = rng.next1D();                                // :456
        BSDFSample mainBsdfResult = bsdfSample(mainBSDF, main.its.wi, bsx, bsy);
        if (mainBsdfResult.pdf <= 0.0) break;                                             // :740
        const V3 mainWo = main.its.sh.toWorld(mainBsdfResult.wo);
        Float mainWoDotGeoN = dot(main.its.geoN, mainWo);
        if (cfg.strictNormals && mainWoDotGeoN * cosTheta(mainBsdfResult.wo) <= 0) break;  // :749
        Intersection previousMainIts = main.its;                                           // :754
        const V3 mainBsdfWi = main.its.wi;                                                 // bRec.wi
        bool mainHitEmitter = false;
        V3 mainEmitterRadiance(0.0);
        DirectSamplingRecord mainDRec;
        mainDRec.ref = main.its.p; mainDRec.refN = refNormal(mainBSDF, main.its); mainDRec.object = -1; mainDRec.measure = MEASURE_SOLID_ANGLE;
        VertexType mainVertexType = getVertexType(mainBSDF, cfg, mainBsdfResult.sampledType); // :765
        VertexType mainNextVertexType;
        main.ray = Ray(main.its.p, mainWo);                                                // :768
        const bool mainHitSomething = rayIntersect(sc, main.ray, main.its);
        if (g_traceMain)                                                                  // debugging aid of tools/gpu_fuzz_locate.py
            fprintf(stderr, "gpo main ray depth %d: o %.17g %.17g %.17g d %.17g %.17g %.17g -> prim %d t %.17g\n", depth, main.ray.o.x, main.ray.o.y, main.ray.o.z,
                    main.ray.d.x, main.ray.d.y, main.ray.d.z, mainHitSomething ? main.its.prim : -1, mainHitSomething ? main.its.t : -1.0);
        if (mainHitSomething) {
            if (sc.tris[main.its.prim].emitter >= 0) {                                     // :772-777
                mainEmitterRadiance = Le(sc, main.its, -main.ray.d);
                mainDRec.p = main.its.p; mainDRec.n = main.its.sh.n; mainDRec.d = main.ray.d; mainDRec.dist = main.its.t; // records.inl:170-178
                mainDRec.object = sc.tris[main.its.prim].emitter;
                mainHitEmitter = true;
            }
            mainNextVertexType = getVertexType(matOf(sc, main.its, main.ray), cfg, mainBsdfResult.sampledType); // :785
        } else {                                                                           // :786-804
            if (sc.envIndex < 0) break;
            mainEmitterRadiance = evalEnvironment(sc, main.ray);                           // evalEnvironment
            if (!envFillDirectSamplingRecord(sc, mainDRec, main.ray)) break;
            mainHitEmitter = true;
            mainNextVertexType = VERTEX_TYPE_DIFFUSE;                                      // "environment connection as diffuse"
        }
        Float mainBsdfPdf = mainBsdfResult.pdf, mainPreviousPdf = main.pdf;
        main.throughput = main.throughput * (mainBsdfResult.weight * mainBsdfResult.pdf);  // :810-812
        main.pdf *= mainBsdfResult.pdf;
        main.eta *= mainBsdfResult.eta;
        const Float mainLumPdf = (mainHitEmitter && !(mainBsdfResult.sampledType & EDelta)) ? pdfEmitterDirect(sc, mainDRec) : 0; // :815
        Float mainWeightNumerator = mainPreviousPdf * mainBsdfResult.pdf;                  // :819
        Float mainWeightDenominator = (mainPreviousPdf * mainPreviousPdf) * ((mainLumPdf * mainLumPdf) + (mainBsdfPdf * mainBsdfPdf));

        for (int i = 0; i < secondaryCount; ++i) {                                         // :830
            RayState &shifted = shiftedRays[i];
            V3 mainContribution(0.0), shiftedContribution(0.0);
            Float weight = 0;
            bool postponedShiftEnd = false;
            if (shifted.alive) {
                Float shiftedPreviousPdf = shifted.pdf;
                if (shifted.connection_status == RAY_CONNECTED) {                          // :844-861
                    V3 shiftedBsdfValue = mainBsdfResult.weight * mainBsdfResult.pdf;
                    Float shiftedBsdfPdf = mainBsdfPdf, shiftedLumPdf = mainLumPdf;
                    shifted.throughput = shifted.throughput * shiftedBsdfValue;
                    shifted.pdf *= shiftedBsdfPdf;
                    Float shiftedWeightDenominator = (shiftedPreviousPdf * shiftedPreviousPdf) * ((shiftedLumPdf * shiftedLumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                    weight = mainWeightNumerator / (D_EPSILON + shiftedWeightDenominator + mainWeightDenominator);
                    mainContribution = main.throughput * mainEmitterRadiance;
                    shiftedContribution = shifted.throughput * mainEmitterRadiance;
                } else if (shifted.connection_status == RAY_RECENTLY_CONNECTED) {          // :862-888
                    V3 incomingDirection = normalize(shifted.its.p - main.ray.o);
                    V3 wiL = previousMainIts.sh.toLocal(incomingDirection), woL = previousMainIts.sh.toLocal(main.ray.d);
                    int measure = (mainBsdfResult.sampledType & EDelta) ? MEASURE_DISCRETE : MEASURE_SOLID_ANGLE;
                    V3 shiftedBsdfValue = bsdfEval(mainBSDF, wiL, woL, measure);
                    Float shiftedBsdfPdf = bsdfPdf(mainBSDF, wiL, woL, measure);
                    Float shiftedLumPdf = mainLumPdf;
                    shifted.throughput = shifted.throughput * shiftedBsdfValue;
                    shifted.pdf *= shiftedBsdfPdf;
                    shifted.connection_status = RAY_CONNECTED;
                    Float shiftedWeightDenominator = (shiftedPreviousPdf * shiftedPreviousPdf) * ((shiftedLumPdf * shiftedLumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                    weight = mainWeightNumerator / (D_EPSILON + shiftedWeightDenominator + mainWeightDenominator);
                    mainContribution = main.throughput * mainEmitterRadiance;
                    shiftedContribution = shifted.throughput * mainEmitterRadiance;
                } else {                                                                   // :889-1126
                    const gpo_material &shiftedBSDF = matOf(sc, shifted.its, shifted.ray);
                    VertexType shiftedVertexType = getVertexType(shiftedBSDF, cfg, mainBsdfResult.sampledType);
                    if (mainVertexType == VERTEX_TYPE_DIFFUSE && mainNextVertexType == VERTEX_TYPE_DIFFUSE && shiftedVertexType == VERTEX_TYPE_DIFFUSE) {
                        if (!lastSegment || mainHitEmitter) {                              // :901
                            ReconnectionShiftResult shiftResult;
                            if (main.its.isValid()) shiftResult = reconnectShift(sc, main.ray.o, main.its.p, shifted.its.p, main.its.geoN);
                            else {                                                         // reconnection at infinity, :908-915
                                EnvShiftResult e = environmentShift(sc, main.ray, shifted.its.p);
                                shiftResult.success = e.success; shiftResult.jacobian = e.jacobian; shiftResult.wo = e.wo;
                            }
                            if (!shiftResult.success) { shifted.alive = false; goto shift_failed; }
                            V3 incomingDirection = -shifted.ray.d, outgoingDirection = shiftResult.wo;
                            V3 wiL = shifted.its.sh.toLocal(incomingDirection), woL = shifted.its.sh.toLocal(outgoingDirection);
                            if (cfg.strictNormals && dot(outgoingDirection, shifted.its.geoN) * cosTheta(woL) <= 0) { shifted.alive = false; goto shift_failed; }
                            V3 shiftedBsdfValue = bsdfEval(shiftedBSDF, wiL, woL, MEASURE_SOLID_ANGLE);
                            Float shiftedBsdfPdf = bsdfPdf(shiftedBSDF, wiL, woL, MEASURE_SOLID_ANGLE);
                            shifted.throughput = shifted.throughput * (shiftedBsdfValue * shiftResult.jacobian); // :939-940
                            shifted.pdf *= shiftedBsdfPdf * shiftResult.jacobian;
                            shifted.connection_status = RAY_RECENTLY_CONNECTED;
                            if (mainHitEmitter) {                                          // :944-986
                                V3 shiftedEmitterRadiance;
                                Float shiftedLumPdf;
                                if (main.its.isValid()) {
                                shiftedEmitterRadiance = Le(sc, main.its, -outgoingDirection);
                                DirectSamplingRecord shiftedDRec;
                                shiftedDRec.p = mainDRec.p; shiftedDRec.n = mainDRec.n;
                                shiftedDRec.dist = length(mainDRec.p - shifted.its.p);
                                shiftedDRec.d = (mainDRec.p - shifted.its.p) / shiftedDRec.dist;
                                shiftedDRec.ref = mainDRec.ref; shiftedDRec.refN = shifted.its.sh.n;
                                shiftedDRec.object = mainDRec.object; shiftedDRec.measure = MEASURE_SOLID_ANGLE;
                                shiftedLumPdf = pdfEmitterDirect(sc, shiftedDRec);
                                } else { shiftedEmitterRadiance = mainEmitterRadiance; shiftedLumPdf = mainLumPdf; }   // :972-976
                                Float shiftedWeightDenominator = (shiftedPreviousPdf * shiftedPreviousPdf) * ((shiftedLumPdf * shiftedLumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                                weight = mainWeightNumerator / (D_EPSILON + shiftedWeightDenominator + mainWeightDenominator);
                                mainContribution = main.throughput * mainEmitterRadiance;
                                shiftedContribution = shifted.throughput * shiftedEmitterRadiance;
                            }
                        }
                    } else {                                                               // half-vector duplication, :987-1126
                        V3 tangentSpaceIncomingDirection = shifted.its.sh.toLocal(-shifted.ray.d);
                        V3 tangentSpaceOutgoingDirection;
                        V3 shiftedEmitterRadiance(0.0);
                        {
                            bool bothDelta = (mainBsdfResult.sampledType & EDelta) && (bsdfType(shiftedBSDF) & EDelta);   // :996-1001
                            bool bothSmooth = (mainBsdfResult.sampledType & ESmooth) && (bsdfType(shiftedBSDF) & ESmooth);
                            if (!(bothDelta || bothSmooth)) { shifted.alive = false; goto half_vector_shift_failed; }
                            HalfVectorShiftResult shiftResult = halfVectorShift(mainBsdfWi, mainBsdfResult.wo, shifted.its.sh.toLocal(-shifted.ray.d), getEta(mainBSDF), getEta(shiftedBSDF)); // :1006
                            if (mainBsdfResult.sampledType & EDelta) shiftResult.jacobian = 1;                             // :1008-1011
                            if (shiftResult.success) {
                                shifted.throughput = shifted.throughput * shiftResult.jacobian;
                                shifted.pdf *= shiftResult.jacobian;
                                tangentSpaceOutgoingDirection = shiftResult.wo;
                            } else { shifted.alive = false; goto half_vector_shift_failed; }
                            V3 outgoingDirection = shifted.its.sh.toWorld(tangentSpaceOutgoingDirection);
                            int measure = (mainBsdfResult.sampledType & EDelta) ? MEASURE_DISCRETE : MEASURE_SOLID_ANGLE;
                            shifted.throughput = shifted.throughput * bsdfEval(shiftedBSDF, tangentSpaceIncomingDirection, tangentSpaceOutgoingDirection, measure);
                            shifted.pdf *= bsdfPdf(shiftedBSDF, tangentSpaceIncomingDirection, tangentSpaceOutgoingDirection, measure);
                            if (shifted.pdf == 0) { shifted.alive = false; goto half_vector_shift_failed; }                // :1034
                            if (cfg.strictNormals && dot(outgoingDirection, shifted.its.geoN) * cosTheta(tangentSpaceOutgoingDirection) <= 0) { shifted.alive = false; goto half_vector_shift_failed; }
                            VertexType shiftedVertexType2 = getVertexType(shiftedBSDF, cfg, mainBsdfResult.sampledType);
                            shifted.ray = Ray(shifted.its.p, outgoingDirection);                                           // :1050
                            if (!rayIntersect(sc, shifted.ray, shifted.its)) {                                             // :1052-1074
                                if (sc.envIndex < 0) { shifted.alive = false; goto half_vector_shift_failed; }
                                if (main.its.isValid()) { shifted.alive = false; goto half_vector_shift_failed; }            // no shifts between env and non-env
                                if (mainVertexType == VERTEX_TYPE_DIFFUSE && shiftedVertexType2 == VERTEX_TYPE_DIFFUSE) { shifted.alive = false; goto half_vector_shift_failed; }
                                shiftedEmitterRadiance = evalEnvironment(sc, shifted.ray);
                                postponedShiftEnd = true;
                                goto half_vector_shift_failed;                                                             // (label name only: alive stays true)
                            }
                            if (!main.its.isValid()) { shifted.alive = false; goto half_vector_shift_failed; }             // :1078-1082
                            VertexType shiftedNextVertexType = getVertexType(matOf(sc, shifted.its, shifted.ray), cfg, mainBsdfResult.sampledType);
                            if (mainVertexType == VERTEX_TYPE_DIFFUSE && shiftedVertexType2 == VERTEX_TYPE_DIFFUSE && shiftedNextVertexType == VERTEX_TYPE_DIFFUSE) { // :1089-1093
                                shifted.alive = false; goto half_vector_shift_failed;
                            }
                            if (sc.tris[shifted.its.prim].emitter >= 0) shiftedEmitterRadiance = Le(sc, shifted.its, -shifted.ray.d);
                        }
                    half_vector_shift_failed:
                        if (shifted.alive) {                                               // :1106-1112
                            weight = main.pdf / (shifted.pdf * shifted.pdf + main.pdf * main.pdf);
                            mainContribution = main.throughput * mainEmitterRadiance;
                            shiftedContribution = shifted.throughput * shiftedEmitterRadiance;
                        } else {                                                           // :1113-1124
                            weight = 1.0 / main.pdf;
                            mainContribution = main.throughput * mainEmitterRadiance;
                            shiftedContribution = V3(0.0);
                            shifted.alive = true;
                            postponedShiftEnd = true;
                        }
                    }
                }
            }
        shift_failed:
            if (!shifted.alive) {                                                          // :1130-1136
                weight = mainWeightNumerator / (D_EPSILON + mainWeightDenominator);
                mainContribution = main.throughput * mainEmitterRadiance;
                shiftedContribution = V3(0.0);
            }
            main.addRadiance(mainContribution, weight);                                    // :1140-1146 (depth+1 >= minDepth==1 always)
            shifted.addRadiance(shiftedContribution, weight);
            shifted.addGradient(shiftedContribution - mainContribution, weight);
            if (postponedShiftEnd) shifted.alive = false;
        }
        if (!main.its.isValid()) break;                                                    // :1155-1157
        if (depth++ >= cfg.rrDepth) {                                                      // :1159-1174
            Float q = std::min(maxc(main.throughput / main.pdf) * main.eta * main.eta, (Float)0.95f);
            if (rng.next1D() >= q) break;
            main.pdf *= q;
            for (int i = 0; i < secondaryCount; ++i) shiftedRays[i].pdf *= q;
        }
    }
}

// ---- film: GPTWorkResult::put (gpt_wr.h:56-64) -> ImageBlock::put (imageblock.h:150-199) with the box filter ----
struct Film {
    int W, H;
    std::vector<double> buf[5]; // [H][W][4]: R,G,B,weight (alpha == 1 carried implicitly)
    double filterRadius, filterScale, filterValues[32];
    unsigned long long invalidPuts = 0;   // puts dropped by the validity check of ImageBlock::put
    // the reconstruction filters of src/rfilters/*.cpp; kind: 0 box, 1 tent, 2 gaussian (p0 = stddev), 3 mitchell (p0 = B, p1 = C),
    // 4 catmullrom, 5 lanczos (p0 = lobes)
    static double filterRadiusOf(int kind, double p0)
    {
        switch (kind) {
            case 1: return 1.0;                                       // tent.cpp:34
            case 2: return 4 * p0;                                    // gaussian.cpp:38
            case 3: case 4: return 2.0;                               // mitchell.cpp:35, catmullrom.cpp:32
            case 5: return p0;                                        // lanczos.cpp:35
            default: return 0.5 + (double)1e-5f;                      // box.cpp:38
        }
    }
    static double filterEval(int kind, double p0, double p1, double radius, double x)
    {
        auto cubic = [](double B, double C, double x) {              // mitchell.cpp:55-68, catmullrom.cpp:40-55
            x = std::abs(x);
            double x2 = x * x, x3 = x2 * x;
            if (x < 1) return 1.0 / 6.0 * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
            else if (x < 2) return 1.0 / 6.0 * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
            return 0.0;
        };
        switch (kind) {
            case 1: return std::max(0.0, 1.0 - std::abs(x / radius));                                                 // tent.cpp:42-44
            case 2: { double alpha = -1.0 / (2.0 * p0 * p0); return std::max(0.0, std::exp(alpha * x * x) - std::exp(alpha * radius * radius)); }   // gaussian.cpp:52-57
            case 3: return cubic(p0, p1, x);
            case 4: return cubic(0.0, 0.5, x);
            case 5: {                                                                                                 // lanczos.cpp:43-55
                x = std::abs(x);
                if (x < Epsilon) return 1.0;
                else if (x > radius) return 0.0;
                double x1 = PI * x, x2 = x1 / radius;
                return (std::sin(x1) * std::sin(x2)) / (x1 * x2);
            }
            default: return std::abs(x) <= radius ? 1.0 : 0.0;                                                        // box.cpp:44-46
        }
    }
    Film(int w, int h, int kind = 0, double p0 = 0, double p1 = 0) : W(w), H(h)
    {
        for (auto &b : buf) b.assign((size_t)w * h * 4, 0.0);
        filterRadius = filterRadiusOf(kind, p0);
        double sum = 0;                                           // ReconstructionFilter::configure, rfilter.cpp:37-55
        for (int i = 0; i < 31; ++i) { double v = filterEval(kind, p0, p1, filterRadius, (filterRadius * i) / 31); filterValues[i] = v; sum += v; }
        filterValues[31] = 0.0;
        filterScale = 31 / filterRadius;
        sum *= 2 * filterRadius / 31;
        double normalization = 1.0 / sum;
        for (int i = 0; i < 31; ++i) filterValues[i] *= normalization;
    }
    double evalDiscretized(double x) const { return filterValues[std::min((int)std::abs(x * filterScale), 31)]; } // rfilter.h:76-77
    void put(double px, double py, V3 spec, double weight, int b)
    {
        // Blocks carry a border and are merged by addition with clipping to the film (gpt_proc.cpp:52-56,137-149); net
        // effect on the film: the footprint of imageblock.h:172-176, restricted to [0,W)x[0,H).
        // imageblock.h:154-158: a put with a non-finite channel -- or, where negative values are not allowed (every buffer but
        // dx and dy, gpt_wr.cpp:41-42; blocks are created with warn = true, :38), a negative one -- is dropped whole, value and weight
        // (the reference logs "Invalid sample value" and goes on).  Channels: spec, alpha = 1, weight.
        const bool allowNegative = (b == 2 || b == 3);
        {
            const double chk[5] = {spec.x, spec.y, spec.z, 1.0, weight};
            for (int k = 0; k < 5; ++k)
                if (!std::isfinite(chk[k]) || (!allowNegative && chk[k] < 0)) { invalidPuts++; return; }
        }
        const double posx = px - 0.5, posy = py - 0.5;
        const int x0 = std::max((int)std::ceil(posx - filterRadius), 0), y0 = std::max((int)std::ceil(posy - filterRadius), 0);
        const int x1 = std::min((int)std::floor(posx + filterRadius), W - 1), y1 = std::min((int)std::floor(posy + filterRadius), H - 1);
        const double value[4] = {spec.x, spec.y, spec.z, weight};
        for (int y = y0; y <= y1; ++y) {
            const double wy = evalDiscretized(y - posy);
            for (int x = x0; x <= x1; ++x) {
                const double w = evalDiscretized(x - posx) * wy;
                double *dest = &buf[b][((size_t)y * W + x) * 4];
                for (int k = 0; k < 4; ++k) dest[k] += w * value[k];
            }
        }
    }
};

enum { BUFFER_FINAL = 0, BUFFER_THROUGHPUT = 1, BUFFER_DX = 2, BUFFER_DY = 3, BUFFER_VERY_DIRECT = 4 }; // gpt.cpp:76-80

void renderSample(const Scene &sc, const gpo_config &cfg, Rng &rng, int px, int py, Film &film);

// GradientPathIntegrator::renderBlock, gpt.cpp:1220-1355, for the pixels of [x0,x1) x [y0,y1)
void renderRect(const Scene &sc, const gpo_config &cfg, int x0, int y0, int x1, int y1, Film &film)
{
    for (int py = y0; py < y1; ++py)
        for (int px = x0; px < x1; ++px)
            for (int j = 0; j < cfg.spp; ++j) {
                Rng rng(cfg.seed, (uint64_t)py * sc.cam.width + px, (uint64_t)j);
                renderSample(sc, cfg, rng, px, py, film);
            }
}

// the body of renderBlock's sample loop, gpt.cpp:1254-1352
void renderSample(const Scene &sc, const gpo_config &cfg, Rng &rng, int px, int py, Film &film)
{
    static const double shifts[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};                 // gpt.cpp:410-415
    const double sx = px + rng.next1D(), sy = py + rng.next1D();               // :1261
    double apx = 0.5, apy = 0.5;
    if (sc.cam.type == 1) { apx = rng.next1D(); apy = rng.next1D(); }           // :1262-1264 (needsApertureSample)
    if (sc.cam.shutterClose > sc.cam.shutterOpen) (void)rng.next1D();          // :1265-1267 (needsTimeSample): timeSample -> ray.time = sampleTime(.), sensor.h:202;
                                                                               // every transform here is static (m_worldTransform->eval(time) is one matrix), so the draw is all it does
    RayState mainRay;
    sampleRay(sc, sx, sy, mainRay.ray, apx, apy);                              // evaluatePoint, :397-436: base and offsets share the aperture sample
    mainRay.throughput = V3(1.0);
    RayState shiftedRays[4];
    for (int i = 0; i < 4; ++i) { sampleRay(sc, sx + shifts[i][0], sy + shifts[i][1], shiftedRays[i].ray, apx, apy); shiftedRays[i].throughput = V3(1.0); }
    V3 veryDirect(0.0);
    evaluate(sc, cfg, rng, mainRay, shiftedRays, 4, veryDirect);
    const V3 T = mainRay.radiance;
    enum { RIGHT = 0, BOTTOM = 1, LEFT = 2, TOP = 3 };
    // :1314-1352
    film.put(sx, sy, (8 * veryDirect) + (2 * T), 4.0, BUFFER_FINAL);
    film.put(sx - 1, sy, 2 * shiftedRays[LEFT].radiance, 1.0, BUFFER_FINAL);
    film.put(sx + 1, sy, 2 * shiftedRays[RIGHT].radiance, 1.0, BUFFER_FINAL);
    film.put(sx, sy - 1, 2 * shiftedRays[TOP].radiance, 1.0, BUFFER_FINAL);
    film.put(sx, sy + 1, 2 * shiftedRays[BOTTOM].radiance, 1.0, BUFFER_FINAL);
    film.put(sx, sy, 2 * T, 4.0, BUFFER_THROUGHPUT);
    film.put(sx - 1, sy, 2 * shiftedRays[LEFT].radiance, 1.0, BUFFER_THROUGHPUT);
    film.put(sx + 1, sy, 2 * shiftedRays[RIGHT].radiance, 1.0, BUFFER_THROUGHPUT);
    film.put(sx, sy - 1, 2 * shiftedRays[TOP].radiance, 1.0, BUFFER_THROUGHPUT);
    film.put(sx, sy + 1, 2 * shiftedRays[BOTTOM].radiance, 1.0, BUFFER_THROUGHPUT);
    film.put(sx - 1, sy, -(2 * shiftedRays[LEFT].gradient), 1.0, BUFFER_DX);
    film.put(sx, sy, 2 * shiftedRays[RIGHT].gradient, 1.0, BUFFER_DX);
    film.put(sx, sy - 1, -(2 * shiftedRays[TOP].gradient), 1.0, BUFFER_DY);
    film.put(sx, sy, 2 * shiftedRays[BOTTOM].gradient, 1.0, BUFFER_DY);
    film.put(sx, sy, veryDirect, 1.0, BUFFER_VERY_DIRECT);
}

// What `mitsuba -p 1` accumulates: worker 0's sampler is a clone of the scene's IndependentSampler (renderjob.cpp:59-66), i.e. an
// SFMT-19937 stream seeded by init_by_array from 312 draws of the parent `Random()` (seed 5489; independent.cpp:58,71-80,
// random.cpp:519-524) -- provided nothing drew from the scene's sampler before the render job cloned it, which holds for gpt
// (no preprocess pass samples).  Work units are the spiral blocks of BlockedImageProcess (imageproc.cpp:28-78); inside a block
// GPTBlockRenderer::process walks the pixels in Hilbert order (gpt_proc.cpp:84-87) and renderBlock the samples in index order
// (gpt.cpp:1245-1268; sampler->generate() draws nothing for the independent sampler without requested arrays).  Merging blocks
// into the film is addition (gpt_proc.cpp:137-149), so film sums depend on the order only through fp64 rounding.
void renderSerial(const Scene &sc, const gpo_config &cfg, int blockSize, uint64_t parentSeed, Film &film)
{
    sfmt_oracle::Random parent(parentSeed);
    sfmt_oracle::Random stream(parent);
    Rng rng(&stream);
    sfmt_oracle::HilbertPoints hilbert;                                                    // one curve object per worker, kept between blocks
    for (const sfmt_oracle::Block &b : sfmt_oracle::spiralBlocks(sc.cam.width, sc.cam.height, blockSize)) {
        hilbert.initialize(b.w, b.h);
        for (const auto &pt : hilbert.pts)
            for (int j = 0; j < cfg.spp; ++j) renderSample(sc, cfg, rng, b.x + pt.first, b.y + pt.second, film);
    }
}

// Scene::getAABB() AFTER Scene::initializeBidirectional (scene.cpp:386-413): the kd-tree's (enlarged) bounds, expanded by the sensor's AABB
// (perspective: the camera position, perspective.cpp:444-446 -> AnimatedTransform::getTranslationBounds, track.cpp:79-83; thinlens: the
// spatial bounds of the aperture box (-r,-r,0)..(r,r,0), thinlens.cpp:516-520 -> track.cpp:123-129) and by every emitter's AABB (area: its
// shape's bounds, inside the kd-tree's already, area.cpp:204-206; point: its position, point.cpp:153-155; constant / envmap: the centre of
// m_sceneBSphere, constant.cpp:234-240).  Scene::getBSphere() (scene.h:972-975) is AABB::getBSphere of THIS box (aabb.cpp:44-47) -- the
// yardstick of ManifoldPerturbation::manifoldWalk's reversibility test (mut_manifold.cpp:1219).
static void sensorBounds(const Scene &sc, V3 &mn, V3 &mx)              // m_aabb after `m_aabb.expandBy(m_sensor->getAABB())`, scene.cpp:386-395
{
    mn = sc.aabbMin; mx = sc.aabbMax;
    auto expand = [&](V3 p) {
        mn = V3(std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z));
        mx = V3(std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z));
    };
    const double *M = sc.cam.toWorld;
    if (sc.cam.type == 1) {
        const Float r = sc.cam.apertureRadius;
        for (int j = 0; j < 4; ++j) {                                  // the box is flat in z: four distinct corners
            const Float x = (j & 1) ? r : -r, y = (j & 2) ? r : -r;
            expand(V3(M[0] * x + M[1] * y + M[3], M[4] * x + M[5] * y + M[7], M[8] * x + M[9] * y + M[11]));
        }
    } else expand(V3(M[3], M[7], M[11]));
}
static void sceneBounds(const Scene &sc, V3 &mn, V3 &mx)
{
    sensorBounds(sc, mn, mx);
    const V3 sceneMn = mn, sceneMx = mx;                               // (m_aabb when the environment emitter's createShape reads it, scene.cpp:398-407)
    for (size_t e = 0; e < sc.emitters.size(); ++e) {
        const Emitter &em = sc.emitters[e];
        V3 p;
        if (em.numTris < 0) p = em.position;
        else if (em.numTris == 0) p = (sceneMx + sceneMn) * 0.5;       // AABB(m_sceneBSphere.center): inside the box by construction
        else continue;
        mn = V3(std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z));
        mx = V3(std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z));
    }
}
static Float sceneBSphereRadius(const Scene &sc)
{
    V3 mn, mx;
    sceneBounds(sc, mn, mx);
    const V3 ctr = (mx + mn) * 0.5;                                    // AABB::getCenter, aabb.h:132-134
    return length(ctr - mx);                                          // aabb.cpp:44-47
}

#include "gbdpt_oracle.hpp"

} // namespace

// ================================================================================================================
// C entry points (ctypes)
// ================================================================================================================
struct gpo_scene { Scene sc; };

GPO_API gpo_scene *gpo_scene_create(int ntri, const double *verts, const int *triMaterial, int nmat, const gpo_material *mats,
                                    int nemit, const gpo_emitter *emitters, const gpo_camera *cam)
{
    gpo_scene *h = new gpo_scene;
    Scene &sc = h->sc;
    sc.cam = *cam;
    sc.aspect = cam->fullWidth > 0 ? (double)cam->fullWidth / (double)cam->fullHeight : (double)cam->width / (double)cam->height;   // sensor.cpp: m_aspect = the FILM's size, not the crop's
    sc.tanHalf = std::tan((cam->fovX * 0.5) * (PI / 180.0));
    sc.mats.assign(mats, mats + nmat);
    sc.tris.resize(ntri);
    sc.aabbMin = V3(INF); sc.aabbMax = V3(-INF);
    for (int i = 0; i < ntri; ++i) {
        Tri &t = sc.tris[i];
        t.p0 = V3(verts[9 * i], verts[9 * i + 1], verts[9 * i + 2]);
        t.p1 = V3(verts[9 * i + 3], verts[9 * i + 4], verts[9 * i + 5]);
        t.p2 = V3(verts[9 * i + 6], verts[9 * i + 7], verts[9 * i + 8]);
        triaccel_load(t.acc, t.p0, t.p1, t.p2);
        t.material = triMaterial[i];
        t.emitter = -1;
        V3 side1 = t.p1 - t.p0, side2 = t.p2 - t.p0;
        V3 fn = cross(side1, side2);
        Float len = length(fn);
        if (!isZero(fn)) fn = fn / len;                               // skdtree.h:369-371
        t.faceNormal = fn;
        t.geoN = fn;
        t.sh.n = fn;                                                   // no vertex normals: shFrame.n = faceNormal (skdtree.h:396)
        t.dpdu = side1; t.dpdv = side2;                                // skdtree.h:377-379 (a mesh without UV tangents)
        t.sh.s = normalize(side1 - fn * dot(fn, side1));               // computeShadingFrame, util.cpp:603-608
        t.sh.t = cross(fn, t.sh.s);
        const V3 ps[3] = {t.p0, t.p1, t.p2};
        for (const V3 &p : ps) {
            sc.aabbMin = V3(std::min(sc.aabbMin.x, p.x), std::min(sc.aabbMin.y, p.y), std::min(sc.aabbMin.z, p.z));
            sc.aabbMax = V3(std::max(sc.aabbMax.x, p.x), std::max(sc.aabbMax.y, p.y), std::max(sc.aabbMax.z, p.z));
        }
    }
    {   // GenericKDTree::buildInternal slightly enlarges the scene bounds (gkdtree.h:1213-1219, MTS_KD_AABB_EPSILON 1e-3f,
        // gkdtree.h:50); note the second line sees the already-moved min, as in the reference.
        const Float eps = (Float)1e-3f;
        sc.aabbMin = sc.aabbMin - ((sc.aabbMax - sc.aabbMin) * eps + V3(eps));
        sc.aabbMax = sc.aabbMax + ((sc.aabbMax - sc.aabbMin) * eps + V3(eps));
    }
    for (int e = 0; e < nemit; ++e) {
        Emitter em;
        em.firstTri = emitters[e].firstTri; em.numTris = emitters[e].numTris;
        em.radiance = V3(emitters[e].radiance[0], emitters[e].radiance[1], emitters[e].radiance[2]);
        em.position = V3(emitters[e].position[0], emitters[e].position[1], emitters[e].position[2]);
        if (em.numTris < 0) { em.firstTri = 0; em.invSurfaceArea = 0; sc.emitters.push_back(em); sc.emitterPDF.append(1.0); continue; }
        Distribution d;
        for (int i = 0; i < em.numTris; ++i) {
            const Tri &t = sc.tris[em.firstTri + i];
            d.append(0.5 * length(cross(t.p1 - t.p0, t.p2 - t.p0)));   // Triangle::surfaceArea
            sc.tris[em.firstTri + i].emitter = e;
        }
        Float area = d.normalize();
        em.cdf = d.cdf;
        em.invSurfaceArea = 1.0 / area;
        sc.emitters.push_back(em);
        sc.emitterPDF.append(1.0);                                     // getSamplingWeight() == 1
    }
    if (nemit > 0) sc.emitterPDF.normalize();
    if (ntri > 64) {
        sc.bvhOrder.resize(ntri);
        for (int i = 0; i < ntri; ++i) sc.bvhOrder[i] = i;
        bvhBuild(sc, 0, ntri);
    }
    return h;
}

// Adds `<emitter type="constant">` (src/emitters/constant.cpp) as entry `index` of the scene's emitter list (XML order decides
// which part of the light sample selects it, scene.cpp:855-862).  Bounding sphere as ConstantBackgroundEmitter::createShape
// builds it (constant.cpp:67-70) from Scene::getAABB() at that moment = kd-tree AABB (already enlarged) + the sensor's
// position (scene.cpp:386-395, perspective.cpp:444-446).
GPO_API void gpo_scene_set_environment(gpo_scene *h, const double *radiance, int index)
{
    Scene &sc = h->sc;
    if (sc.envIndex >= 0) return;
    const int n = (int)sc.emitters.size();
    if (index < 0 || index > n) index = n;
    Emitter em;
    em.firstTri = 0; em.numTris = 0; em.radiance = V3(radiance[0], radiance[1], radiance[2]); em.invSurfaceArea = 0;
    sc.emitters.insert(sc.emitters.begin() + index, em);
    for (Tri &t : sc.tris) if (t.emitter >= index) t.emitter++;
    sc.envIndex = index;
    sc.emitterPDF = Distribution();
    for (size_t i = 0; i < sc.emitters.size(); ++i) sc.emitterPDF.append(1.0);
    sc.emitterPDF.normalize();
    V3 mn, mx;
    sensorBounds(sc, mn, mx);                                        // Scene::getAABB() when createShape reads it: kd-tree + sensor (scene.cpp:386-407)
    sc.bsCenter = (mx + mn) * 0.5;                                   // AABB::getCenter, aabb.h:132-134
    sc.bsRadius = std::max(Epsilon, length(sc.bsCenter - mx) * (Float)1.5f);   // aabb.cpp:44-47, constant.cpp:69
}

// Marks area emitter `index` (scene order, before any environment emitter is inserted) as the light of a `rectangle` shape: toWorld12 = rows of its
// 3x4 objectToWorld (flipNormals folded in as the plugin does: toWorld * scale(1, 1, -1)), normal3 = its frame's normal
GPO_API int gpo_scene_set_rectangle_emitter(gpo_scene *h, int index, const double *toWorld12, const double *normal3)
{
    Scene &sc = h->sc;
    if (index < 0 || index >= (int)sc.emitters.size() || sc.emitters[index].numTris != 2) return -1;
    Emitter &em = sc.emitters[index];
    em.rectangle = true;
    for (int k = 0; k < 12; ++k) em.rect[k] = toWorld12[k];
    em.rectN = V3(normal3[0], normal3[1], normal3[2]);
    const V3 dpdu(toWorld12[0] * 2.0, toWorld12[4] * 2.0, toWorld12[8] * 2.0), dpdv(toWorld12[1] * 2.0, toWorld12[5] * 2.0, toWorld12[9] * 2.0);   // rectangle.cpp:103-104
    em.invSurfaceArea = 1.0 / (length(dpdu) * length(dpdv));       // :110,119-121
    return 0;
}

// `<emitter type="envmap">`: rgb = h x w x 3 linear values (top row first: v = 0 is straight up), `scale`, toWorld9 = the linear part of
// the emitter's toWorld (row-major 3x3; identity: +y up, u = 0.5 looks along -z), index = position in the emitter list as for the constant one
GPO_API void gpo_scene_set_envmap(gpo_scene *h, int w, int hgt, const double *rgb, double scale, const double *toWorld9, int index)
{
    Scene &sc = h->sc;
    if (sc.envIndex >= 0) return;
    const double zero[3] = {0, 0, 0};
    gpo_scene_set_environment(h, zero, index);
    Scene::EnvMap &e = sc.envMap;
    e.present = true;
    e.scale = scale;
    for (int k = 0; k < 9; ++k) e.toWorld[k] = toWorld9[k];
    {   // inverse of the 3x3 (Transform::inverse; a rotation in every sensible scene)
        const Float *m = e.toWorld;
        const Float det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        const Float id = 1.0 / det;
        e.toLocal[0] = (m[4] * m[8] - m[5] * m[7]) * id; e.toLocal[1] = (m[2] * m[7] - m[1] * m[8]) * id; e.toLocal[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        e.toLocal[3] = (m[5] * m[6] - m[3] * m[8]) * id; e.toLocal[4] = (m[0] * m[8] - m[2] * m[6]) * id; e.toLocal[5] = (m[2] * m[3] - m[0] * m[5]) * id;
        e.toLocal[6] = (m[3] * m[7] - m[4] * m[6]) * id; e.toLocal[7] = (m[1] * m[6] - m[0] * m[7]) * id; e.toLocal[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    }
    e.mip.build(w, hgt, rgb, mip_oracle::BC_REPEAT, mip_oracle::BC_CLAMP, mip_oracle::FILTER_EWA, 10.0, std::numeric_limits<Float>::infinity(), true);
    envMapConfigure(e);
}
// probes: the environment map's radiance along a direction (no differentials), a light sample (direction, value / pdf, pdf) and its density
GPO_API void gpo_envmap_eval(gpo_scene *h, const double *dir, double *rgb)
{
    Ray r(V3(0.0), V3(dir[0], dir[1], dir[2]));
    const V3 v = envMapEval(h->sc, r);
    rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z;
}
GPO_API void gpo_envmap_sample(gpo_scene *h, double sx, double sy, double *out7)
{
    V3 d, value; Float pdf;
    envMapSampleDirection(h->sc.envMap, sx, sy, d, value, pdf);
    out7[0] = d.x; out7[1] = d.y; out7[2] = d.z; out7[3] = value.x; out7[4] = value.y; out7[5] = value.z; out7[6] = pdf;
}
GPO_API double gpo_envmap_pdf(gpo_scene *h, const double *dirLocal) { return envMapPdfDirection(h->sc.envMap, V3(dirLocal[0], dirLocal[1], dirLocal[2])); }

// Per-vertex normals (9 doubles per triangle; three zero vectors = that triangle has none).  Emitter triangles must be flat:
// AreaLight::eval and TriMesh::samplePosition would otherwise use interpolated normals, which this build does not carry.
GPO_API int gpo_scene_set_normals(gpo_scene *h, const double *n9)
{
    Scene &sc = h->sc;
    for (size_t i = 0; i < sc.tris.size(); ++i) {
        Tri &t = sc.tris[i];
        const double *n = n9 + 9 * i;
        bool any = false;
        for (int k = 0; k < 9; ++k) any = any || n[k] != 0.0;
        if (!any) continue;
        if (t.emitter >= 0) return -1;
        t.hasNormals = true;
        t.n0 = V3(n[0], n[1], n[2]); t.n1 = V3(n[3], n[4], n[5]); t.n2 = V3(n[6], n[7], n[8]);
    }
    return 0;
}

// Per-vertex texture coordinates: 6 doubles per triangle (u0 v0 u1 v1 u2 v2); triangles of meshes without texcoords keep its.uv = (b1, b2).
GPO_API void gpo_scene_set_uvs(gpo_scene *h, const double *uv6, const unsigned char *hasUV)
{
    Scene &sc = h->sc;
    for (size_t i = 0; i < sc.tris.size(); ++i) {
        if (hasUV && !hasUV[i]) continue;
        Tri &t = sc.tris[i];
        t.hasUV = true;
        for (int k = 0; k < 6; ++k) t.uv[k] = uv6[6 * i + k];
        // TriMesh::configure computes UV tangents for every mesh that has texture coordinates (trimesh.cpp:362-386: the second call is
        // unconditional), and fillIntersectionRecord then takes its.dpdu from them (skdtree.h:373-376): the shading frame of such a
        // mesh is oriented along the texture's u axis, not along the first edge.  computeUVTangents, trimesh.cpp:701-735:
        const V3 dP1 = t.p1 - t.p0, dP2 = t.p2 - t.p0;
        const Float dU1x = t.uv[2] - t.uv[0], dU1y = t.uv[3] - t.uv[1], dU2x = t.uv[4] - t.uv[0], dU2y = t.uv[5] - t.uv[1];
        const V3 n = cross(dP1, dP2);
        const Float len = length(n);
        if (len == 0) continue;                                        // (a degenerate triangle keeps its entry as it was; it cannot be hit)
        const Float determinant = dU1x * dU2y - dU1y * dU2x;
        if (determinant == 0) {
            coordinateSystem(n / len, t.dpdu, t.dpdv);                 // degenerate parameterization: arbitrary tangents
        } else {
            const Float invDet = 1.0 / determinant;
            t.dpdu = (dP1 * dU2y - dP2 * dU1y) * invDet;
            t.dpdv = (dP1 * (-dU2x) + dP2 * dU1x) * invDet;
        }
        if (!t.hasNormals) {                                           // the flat frame is a constant of the triangle: redo it with the new dpdu
            t.sh.s = normalize(t.dpdu - t.sh.n * dot(t.sh.n, t.dpdu));
            t.sh.t = cross(t.sh.n, t.sh.s);
        }
    }
}
// Adds a bitmap texture (rgb: h x w x 3 doubles, top row first) and returns its index; params = {wrapU, wrapV, filter}, fparams = {uscale, vscale, uoffset, voffset, scale, maxAnisotropy}
GPO_API int gpo_scene_add_texture(gpo_scene *h, int w, int hgt, const double *rgb, const int *params, const double *fparams)
{
    Texture t;
    t.w = w; t.h = hgt; t.wrapU = params[0]; t.wrapV = params[1]; t.filter = params[2];
    t.uscale = fparams[0]; t.vscale = fparams[1]; t.uoffset = fparams[2]; t.voffset = fparams[3]; t.scale = fparams[4];
    t.maxAnisotropy = fparams[5];
    t.rgb.assign(rgb, rgb + (size_t)w * hgt * 3);
    for (Float &v : t.rgb) if (v < 0) v = 0;                                                  // the MIP map clamps negative texels, mipmap.h:234-242
    if (t.filter >= 2) t.mip.build(w, hgt, t.rgb.data(), t.wrapU, t.wrapV, t.filter, t.maxAnisotropy);
    h->sc.textures.push_back(t);
    return (int)h->sc.textures.size() - 1;
}
GPO_API void gpo_scene_set_material_texture(gpo_scene *h, int material, int texture)
{
    Scene &sc = h->sc;
    if (sc.matTexture.size() < sc.mats.size()) sc.matTexture.resize(sc.mats.size(), -1);
    sc.matTexture[material] = texture;
}
GPO_API void gpo_texture_eval(gpo_scene *h, int texture, double u, double v, double *rgb)
{
    const V3 r = h->sc.textures[texture].eval(u, v);
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}
// the filtered lookup of a hit with UV partials: partials = {dudx, dudy, dvdx, dvdy}
GPO_API void gpo_texture_eval_filtered(gpo_scene *h, int texture, double u, double v, const double *partials, double *rgb)
{
    const V3 r = h->sc.textures[texture].eval(u, v, true, partials[0], partials[1], partials[2], partials[3]);
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}
// MIP pyramid of a trilinear / ewa texture: number of levels; size and texels of one level (rgb may be NULL)
GPO_API int gpo_texture_levels(gpo_scene *h, int texture) { return h->sc.textures[texture].mip.levels(); }
GPO_API void gpo_texture_level(gpo_scene *h, int texture, int level, int *wh, double *rgb)
{
    const mip_oracle::Level &L = h->sc.textures[texture].mip.pyramid[level];
    wh[0] = L.w; wh[1] = L.h;
    if (rgb) std::memcpy(rgb, L.rgb.data(), sizeof(double) * L.rgb.size());
}

// `<rfilter>` of the film: kind as in Film::filterEval, p0/p1 its parameters (defaults are the caller's business)
GPO_API void gpo_scene_set_rfilter(gpo_scene *h, int kind, double p0, double p1) { h->sc.rfilterKind = kind; h->sc.rfilterP0 = p0; h->sc.rfilterP1 = p1; }

GPO_API void gpo_scene_destroy(gpo_scene *h) { delete h; }

// Renders pixels [x0,x1) x [y0,y1).  accum: 5 buffers x H x W x 4 doubles (R,G,B,weight sums; film-sized; contributions
// of the rendered pixels only).  rays[2]: closest-hit and shadow ray counts (skdtree.cpp:46-47 semantics).
GPO_API void gpo_render(gpo_scene *h, const gpo_config *cfg, int x0, int y0, int x1, int y1, double *accum, unsigned long long *rays)
{
    Scene &sc = h->sc;
    sc.raysTraced = sc.shadowRaysTraced = 0;
    Film film(sc.cam.width, sc.cam.height, sc.rfilterKind, sc.rfilterP0, sc.rfilterP1);
    renderRect(sc, *cfg, x0, y0, x1, y1, film);
    const size_t n = (size_t)sc.cam.width * sc.cam.height * 4;
    for (int b = 0; b < 5; ++b) std::memcpy(accum + b * n, film.buf[b].data(), n * sizeof(double));
    if (rays) { rays[0] = sc.raysTraced; rays[1] = sc.shadowRaysTraced; }
    sc.lastInvalidPuts = film.invalidPuts;
}

// puts the last gpo_render dropped as invalid (ImageBlock::put's "Invalid sample value" warnings)
GPO_API unsigned long long gpo_last_invalid_puts(gpo_scene *h) { return h->sc.lastInvalidPuts; }

// The reference's own sample stream: the whole film as a 1-core run of the reference accumulates it (see renderSerial).
// blockSize: Scene::getBlockSize() (32 by default, `-b`).  parentSeed: the seed of the scene sampler's Random (5489, random.h:113).
GPO_API void gpo_render_serial(gpo_scene *h, const gpo_config *cfg, int blockSize, unsigned long long parentSeed, double *accum, unsigned long long *rays)
{
    Scene &sc = h->sc;
    sc.raysTraced = sc.shadowRaysTraced = 0;
    Film film(sc.cam.width, sc.cam.height, sc.rfilterKind, sc.rfilterP0, sc.rfilterP1);
    renderSerial(sc, *cfg, blockSize, parentSeed, film);
    const size_t n = (size_t)sc.cam.width * sc.cam.height * 4;
    for (int b = 0; b < 5; ++b) std::memcpy(accum + b * n, film.buf[b].data(), n * sizeof(double));
    if (rays) { rays[0] = sc.raysTraced; rays[1] = sc.shadowRaysTraced; }
    sc.lastInvalidPuts = film.invalidPuts;
}

// ---- `Random` (SFMT-19937) and the work order, for the pinning tests ----
GPO_API void *gpo_random_create(unsigned long long seed) { return new sfmt_oracle::Random((uint64_t)seed); }
GPO_API void *gpo_random_clone(void *parent) { return new sfmt_oracle::Random(*static_cast<sfmt_oracle::Random *>(parent)); }   // Random(Random *)
GPO_API void gpo_random_destroy(void *r) { delete static_cast<sfmt_oracle::Random *>(r); }
GPO_API void gpo_random_ulongs(void *r, int n, unsigned long long *out) { for (int i = 0; i < n; ++i) out[i] = static_cast<sfmt_oracle::Random *>(r)->nextULong(); }
GPO_API void gpo_random_floats(void *r, int n, double *out) { for (int i = 0; i < n; ++i) out[i] = static_cast<sfmt_oracle::Random *>(r)->nextFloat(); }
GPO_API void gpo_random_floats_single(void *r, int n, float *out) { for (int i = 0; i < n; ++i) out[i] = static_cast<sfmt_oracle::Random *>(r)->nextFloatSingle(); }
GPO_API unsigned gpo_random_uint(void *r, unsigned n) { return static_cast<sfmt_oracle::Random *>(r)->nextUInt(n); }
GPO_API void gpo_random_seed_array(void *r, const unsigned long long *key, unsigned long long length) { static_cast<sfmt_oracle::Random *>(r)->seed((const uint64_t *)key, (uint64_t)length); }
GPO_API void gpo_random_set(void *r, void *other) { static_cast<sfmt_oracle::Random *>(r)->set(*static_cast<sfmt_oracle::Random *>(other)); }
// -> number of blocks; out4 (x, y, w, h per block) may be null to query the count
GPO_API int gpo_spiral_blocks(int width, int height, int blockSize, int *out4)
{
    const std::vector<sfmt_oracle::Block> b = sfmt_oracle::spiralBlocks(width, height, blockSize);
    if (out4) for (size_t i = 0; i < b.size(); ++i) { out4[4 * i] = b[i].x; out4[4 * i + 1] = b[i].y; out4[4 * i + 2] = b[i].w; out4[4 * i + 3] = b[i].h; }
    return (int)b.size();
}
GPO_API int gpo_hilbert_points(int w, int h, unsigned char *outXY)
{
    sfmt_oracle::HilbertPoints hp;
    hp.initialize(w, h);
    if (outXY) for (size_t i = 0; i < hp.pts.size(); ++i) { outXY[2 * i] = hp.pts[i].first; outXY[2 * i + 1] = hp.pts[i].second; }
    return (int)hp.pts.size();
}

// MultiFilm::developMulti -> weight division of fmtconv.cpp:955-1058: invWeight = w != 0 ? 1/w : w ; rgb * invWeight
GPO_API void gpo_develop(const double *accum, int numPixels, double *rgbOut)
{
    for (int i = 0; i < numPixels; ++i) {
        const double w = accum[4 * i + 3], inv = (w != 0) ? 1.0 / w : w;
        for (int c = 0; c < 3; ++c) rgbOut[3 * i + c] = accum[4 * i + c] * inv;
    }
}

// One sample's raw outputs (for KATs): out = veryDirect(3), throughput(3), gradients[4](12), neighbourThroughputs[4](12)
GPO_API void gpo_evaluate_point(gpo_scene *h, const gpo_config *cfg, int px, int py, int sampleIndex, double *out30)
{
    const Scene &sc = h->sc;
    static const double shifts[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};
    Rng rng(cfg->seed, (uint64_t)py * sc.cam.width + px, (uint64_t)sampleIndex);
    const double sx = px + rng.next1D(), sy = py + rng.next1D();
    double apx = 0.5, apy = 0.5;
    if (sc.cam.type == 1) { apx = rng.next1D(); apy = rng.next1D(); }
    if (sc.cam.shutterClose > sc.cam.shutterOpen) (void)rng.next1D();
    RayState mainRay;
    sampleRay(sc, sx, sy, mainRay.ray, apx, apy);
    mainRay.throughput = V3(1.0);
    RayState sh[4];
    for (int i = 0; i < 4; ++i) { sampleRay(sc, sx + shifts[i][0], sy + shifts[i][1], sh[i].ray, apx, apy); sh[i].throughput = V3(1.0); }
    V3 vd(0.0);
    evaluate(sc, *cfg, rng, mainRay, sh, 4, vd);
    double *o = out30;
    *o++ = vd.x; *o++ = vd.y; *o++ = vd.z;
    *o++ = mainRay.radiance.x; *o++ = mainRay.radiance.y; *o++ = mainRay.radiance.z;
    for (int i = 0; i < 4; ++i) { *o++ = sh[i].gradient.x; *o++ = sh[i].gradient.y; *o++ = sh[i].gradient.z; }
    for (int i = 0; i < 4; ++i) { *o++ = sh[i].radiance.x; *o++ = sh[i].radiance.y; *o++ = sh[i].radiance.z; }
}

// The same, plus the sample's own ray counts: rays2 = {closest-hit, shadow} queries issued by this sample alone.
GPO_API void gpo_evaluate_point_counted(gpo_scene *h, const gpo_config *cfg, int px, int py, int sampleIndex, double *out30, uint64_t *rays2)
{
    const Scene &sc = h->sc;
    const uint64_t r0 = sc.raysTraced, s0 = sc.shadowRaysTraced;
    gpo_evaluate_point(h, cfg, px, py, sampleIndex, out30);
    rays2[0] = sc.raysTraced - r0; rays2[1] = sc.shadowRaysTraced - s0;
}

// ---- small probes for known-answer tests ----
GPO_API void gpo_half_vector_shift(const double *mainWi, const double *mainWo, const double *shiftedWi, double mainEta, double shiftedEta, double *out5)
{
    HalfVectorShiftResult r = halfVectorShift(V3(mainWi[0], mainWi[1], mainWi[2]), V3(mainWo[0], mainWo[1], mainWo[2]), V3(shiftedWi[0], shiftedWi[1], shiftedWi[2]), mainEta, shiftedEta);
    out5[0] = r.success; out5[1] = r.jacobian; out5[2] = r.wo.x; out5[3] = r.wo.y; out5[4] = r.wo.z;
}
GPO_API void gpo_bsdf_eval_pdf(const gpo_material *m, const double *wi, const double *wo, int measure, double *out4)
{
    V3 f = bsdfEval(*m, V3(wi[0], wi[1], wi[2]), V3(wo[0], wo[1], wo[2]), measure);
    out4[0] = f.x; out4[1] = f.y; out4[2] = f.z; out4[3] = bsdfPdf(*m, V3(wi[0], wi[1], wi[2]), V3(wo[0], wo[1], wo[2]), measure);
}
GPO_API void gpo_bsdf_sample(const gpo_material *m, const double *wi, double sx, double sy, double *out8)
{
    BSDFSample s = bsdfSample(*m, V3(wi[0], wi[1], wi[2]), sx, sy);
    out8[0] = s.wo.x; out8[1] = s.wo.y; out8[2] = s.wo.z; out8[3] = s.weight.x; out8[4] = s.weight.y; out8[5] = s.weight.z; out8[6] = s.pdf; out8[7] = s.sampledType;
}
GPO_API void gpo_fresnel_conductor(double cosThetaI, const double *eta, const double *k, double *out3)
{
    V3 f = fresnelConductorExact(cosThetaI, V3(eta[0], eta[1], eta[2]), V3(k[0], k[1], k[2]));
    out3[0] = f.x; out3[1] = f.y; out3[2] = f.z;
}
GPO_API int gpo_intersect(gpo_scene *h, const double *o, const double *d, double *out_t_p_wi /*7*/)
{
    Ray r(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]));
    Intersection its;
    if (!rayIntersect(h->sc, r, its)) return -1;
    out_t_p_wi[0] = its.t; out_t_p_wi[1] = its.p.x; out_t_p_wi[2] = its.p.y; out_t_p_wi[3] = its.p.z;
    out_t_p_wi[4] = its.wi.x; out_t_p_wi[5] = its.wi.y; out_t_p_wi[6] = its.wi.z;
    return its.prim;
}
// The filled intersection record of one ray (ShapeKDTree::rayIntersect + fillIntersectionRecord<true>, skdtree.cpp:112-142, skdtree.h:343-428): what
// the reference's own src/tests/test_dgeom.cpp:35-178 asserts on.  out24 = t, p(3), uv(2), geoFrame.n(3), shFrame.n(3), shFrame.s(3), dpdu(3), dpdv(3), wi(3).
GPO_API int gpo_intersect_record(gpo_scene *h, const double *o, const double *d, double *out24)
{
    Ray r(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]));
    Intersection its;
    if (!rayIntersect(h->sc, r, its)) return -1;
    double *q = out24;
    *q++ = its.t;
    const V3 v3[] = {its.p};
    *q++ = v3[0].x; *q++ = v3[0].y; *q++ = v3[0].z;
    *q++ = its.u; *q++ = its.v;
    const V3 rest[] = {its.geoN, its.sh.n, its.sh.s, its.dpdu, its.dpdv, its.wi};
    for (const V3 &v : rest) { *q++ = v.x; *q++ = v.y; *q++ = v.z; }
    return its.prim;
}
GPO_API void gpo_camera_ray_ap(gpo_scene *h, double px, double py, double apx, double apy, double *out14)
{   // with the aperture sample and the two differential directions: o(3), d(3), mint, maxt, rxD(3), ryD(3)
    Ray r;
    sampleRay(h->sc, px, py, r, apx, apy);
    double *o = out14;
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.mint; o[7] = r.maxt;
    o[8] = r.rxD.x; o[9] = r.rxD.y; o[10] = r.rxD.z; o[11] = r.ryD.x; o[12] = r.ryD.y; o[13] = r.ryD.z;
}
GPO_API void gpo_camera_ray(gpo_scene *h, double px, double py, double *out8)
{
    Ray r;
    sampleRay(h->sc, px, py, r);
    out8[0] = r.o.x; out8[1] = r.o.y; out8[2] = r.o.z; out8[3] = r.d.x; out8[4] = r.d.y; out8[5] = r.d.z; out8[6] = r.mint; out8[7] = r.maxt;
}
GPO_API double gpo_rng(unsigned long long seed, unsigned long long pixel, unsigned long long sample, int n)
{
    Rng r(seed, pixel, sample);
    double v = 0;
    for (int i = 0; i <= n; ++i) v = r.next1D();
    return v;
}

// An INDEPENDENT plain path tracer (throughput-only, MIS of light and BSDF sampling, written against the rendering
// equation rather than gpt.cpp) used only to check that E[-throughput] of the G-PT restatement is the radiance integral.
// Returns mean radiance (excluding directly visible emitters) of `n` paths through the centre region of pixel (px,py).
GPO_API void gpo_reference_pt(gpo_scene *h, const gpo_config *cfg, int px, int py, int n, double *rgbOut)
{
    const Scene &sc = h->sc;
    V3 sum(0.0);
    for (int j = 0; j < n; ++j) {
        Rng rng(cfg->seed ^ 0xABCDEF1234567ULL, (uint64_t)py * sc.cam.width + px, (uint64_t)j);
        Ray ray;
        const Float fx = px + rng.next1D(), fy = py + rng.next1D();
        Float ax = 0.5, ay = 0.5;
        if (sc.cam.type == 1) { ax = rng.next1D(); ay = rng.next1D(); }
        sampleRay(sc, fx, fy, ray, ax, ay);
        Intersection its;
        if (!rayIntersect(sc, ray, its)) continue;
        V3 beta(1.0), L(0.0);
        for (int depth = 1; depth < cfg->maxDepth || cfg->maxDepth < 0; ++depth) {
            const Ray plain;                                       // (this checker looks textures up unfiltered)
            const gpo_material &m = matOf(sc, its, plain);
            if (bsdfType(m) & ESmooth) {
                DirectSamplingRecord dRec;
                dRec.ref = its.p; dRec.refN = refNormal(m, its);
                bool vis;
                const Float u1 = rng.next1D(), u2 = rng.next1D();
                V3 val = sampleEmitterDirectVisible(sc, dRec, u1, u2, vis);
                if (vis && dRec.pdf > 0) {
                    V3 woL = its.sh.toLocal(dRec.d);
                    V3 f = bsdfEval(m, its.wi, woL, MEASURE_SOLID_ANGLE);
                    Float pb = (sc.emitters[dRec.object].onSurface() && dRec.measure == MEASURE_SOLID_ANGLE) ? bsdfPdf(m, its.wi, woL, MEASURE_SOLID_ANGLE) : 0.0;   // a point light cannot be hit by BSDF sampling
                    Float wgt = (dRec.pdf * dRec.pdf) / (dRec.pdf * dRec.pdf + pb * pb);
                    L = L + beta * f * val * wgt;
                }
            }
            const Float b1 = rng.next1D(), b2 = rng.next1D();
            BSDFSample s = bsdfSample(m, its.wi, b1, b2);
            if (s.pdf <= 0) break;
            V3 wo = its.sh.toWorld(s.wo);
            DirectSamplingRecord q;
            q.ref = its.p; q.refN = refNormal(m, its);
            Ray next(its.p, wo);
            beta = beta * s.weight;
            if (!rayIntersect(sc, next, its)) {
                if (sc.envIndex >= 0 && envFillDirectSamplingRecord(sc, q, next)) {       // the environment, MIS against its light sampling
                    Float pl = (s.sampledType & EDelta) ? 0.0 : pdfEmitterDirect(sc, q);
                    Float wgt = (s.pdf * s.pdf) / (s.pdf * s.pdf + pl * pl);
                    L = L + beta * evalEnvironment(sc, next) * wgt;
                }
                break;
            }
            if (sc.tris[its.prim].emitter >= 0) {
                V3 le = Le(sc, its, -next.d);
                q.p = its.p; q.n = its.sh.n; q.d = next.d; q.dist = its.t; q.object = sc.tris[its.prim].emitter; q.measure = MEASURE_SOLID_ANGLE;
                Float pl = (s.sampledType & EDelta) ? 0.0 : pdfEmitterDirect(sc, q);
                Float wgt = (s.pdf * s.pdf) / (s.pdf * s.pdf + pl * pl);
                L = L + beta * le * wgt;
            }
            if (depth >= cfg->rrDepth) {
                Float qv = std::min(maxc(beta), 0.95);
                if (rng.next1D() >= qv) break;
                beta = beta / qv;
            }
        }
        sum = sum + L;
    }
    rgbOut[0] = sum.x / n; rgbOut[1] = sum.y / n; rgbOut[2] = sum.z / n;
}

// ---- G-BDPT (oracle/gbdpt_oracle.hpp) ---------------------------------------------------------------------------------------------
extern "C" {
typedef struct gpo_gbdpt_config { int maxDepth, rrDepth, lightImage, spp; double shiftThreshold; unsigned long long seed; } gpo_gbdpt_config;
}
static gb::Config gbConfig(const gpo_gbdpt_config *cfg)
{
    gb::Config c;
    c.maxDepth = cfg->maxDepth; c.rrDepth = cfg->rrDepth; c.lightImage = cfg->lightImage; c.spp = cfg->spp; c.shiftThreshold = cfg->shiftThreshold; c.seed = cfg->seed;
    return c;
}
// one sample of GBDPTRenderer::process: out = primal(3), gradient[4](3 each), sample position(2); light splats as (x, y, buffer, r, g, b);
// Scene::getBSphere().radius after Scene::initializeBidirectional (scene.cpp:386-413, aabb.cpp:44-47)
GPO_API double gpo_scene_bsphere_radius(gpo_scene *h) { return sceneBSphereRadius(h->sc); }

// counters = closest-hit rays, shadow rays, unsupported events (gpo_gbdpt_render: + invalid puts, manifold walks entered / converged, propagated chain vertices)
GPO_API void gpo_gbdpt_sample(gpo_scene *h, const gpo_gbdpt_config *cfg, int px, int py, int sampleIndex, double *out17, int maxLight, double *lightOut, int *nLight,
                              unsigned long long *counters)
{
    gb::Ctx ctx{h->sc, gbConfig(cfg)};
    gb::cameraSetup(ctx);
    const uint64_t r0 = h->sc.raysTraced, s0 = h->sc.shadowRaysTraced;
    Rng rng(cfg->seed, (uint64_t)py * h->sc.cam.width + px, (uint64_t)sampleIndex);
    gb::Pool pool;
    gb::Tracer tr(ctx, rng, pool);
    gb::Tracer::SampleResult r;
    tr.processSample(px, py, r);
    out17[0] = r.primal.x; out17[1] = r.primal.y; out17[2] = r.primal.z;
    for (int k = 0; k < 4; ++k) { out17[3 + 3 * k] = r.gradient[k].x; out17[4 + 3 * k] = r.gradient[k].y; out17[5 + 3 * k] = r.gradient[k].z; }
    out17[15] = r.posX; out17[16] = r.posY;
    *nLight = (int)r.light.size();
    for (int i = 0; i < (int)r.light.size() && i < maxLight; ++i) {
        double *o = lightOut + 6 * i;
        o[0] = r.light[i].x; o[1] = r.light[i].y; o[2] = r.light[i].buffer; o[3] = r.light[i].value.x; o[4] = r.light[i].value.y; o[5] = r.light[i].value.z;
    }
    counters[0] = h->sc.raysTraced - r0; counters[1] = h->sc.shadowRaysTraced - s0; counters[2] = ctx.unsupported;
}
// known-answer probe of the specular manifold on the sensor subpath of one sample (gb::Tracer::manifoldProbe)
GPO_API void gpo_manifold_probe(gpo_scene *h, const gpo_gbdpt_config *cfg, int px, int py, int sampleIndex, const double *delta3, double *out32)
{
    gb::Ctx ctx{h->sc, gbConfig(cfg)};
    gb::cameraSetup(ctx);
    Rng rng(cfg->seed, (uint64_t)py * h->sc.cam.width + px, (uint64_t)sampleIndex);
    gb::Pool pool;
    gb::Tracer tr(ctx, rng, pool);
    tr.manifoldProbe(px, py, delta3, out32);
}
// TriMesh::getNormalDerivative(its, dndu, dndv, shadingFrame = true) at the hit of one ray (trimesh.cpp:745-822; the G-BDPT manifold walk's input,
// manifold.cpp:101-122): out6 = dndu(3), dndv(3).  src/tests/test_dgeom.cpp:112-119,170-176 hold two vectors for it.
GPO_API int gpo_normal_derivative(gpo_scene *h, const double *o, const double *d, double *out6)
{
    Ray r(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]));
    Intersection its;
    if (!rayIntersect(h->sc, r, its)) return -1;
    gpo_gbdpt_config cfg = {12, 5, 1, 1, 0.001, 5489ULL};
    gb::Ctx ctx{h->sc, gbConfig(&cfg)};
    Rng rng(cfg.seed, 0, 0);
    gb::Pool pool;
    gb::Tracer tr(ctx, rng, pool);
    V3 dndu, dndv;
    tr.normalDerivative(its, dndu, dndv);
    out6[0] = dndu.x; out6[1] = dndu.y; out6[2] = dndu.z; out6[3] = dndv.x; out6[4] = dndv.y; out6[5] = dndv.z;
    return its.prim;
}
GPO_API void gpo_manifold_probe2(gpo_scene *h, const gpo_gbdpt_config *cfg, int px, int py, int sampleIndex, const double *delta3, double *out48)
{
    gb::Ctx ctx{h->sc, gbConfig(cfg)};
    gb::cameraSetup(ctx);
    Rng rng(cfg->seed, (uint64_t)py * h->sc.cam.width + px, (uint64_t)sampleIndex);
    gb::Pool pool;
    gb::Tracer tr(ctx, rng, pool);
    tr.manifoldProbe2(px, py, delta3, out48);
}
// GBDPTRenderer::process over the pixels of [x0,x1) x [y0,y1): the five camera blocks [5][H][W][4] and the five light images [5][H][W][3]
GPO_API void gpo_gbdpt_render(gpo_scene *h, const gpo_gbdpt_config *cfg, int x0, int y0, int x1, int y1, double *block, double *light, unsigned long long *counters)
{
    gb::Ctx ctx{h->sc, gbConfig(cfg)};
    gb::cameraSetup(ctx);
    const uint64_t r0 = h->sc.raysTraced, s0 = h->sc.shadowRaysTraced;
    const int W = h->sc.cam.width, H = h->sc.cam.height;
    gb::Film film(W, H);
    for (int py = y0; py < y1; ++py)
        for (int px = x0; px < x1; ++px)
            for (int j = 0; j < cfg->spp; ++j) {
                Rng rng(cfg->seed, (uint64_t)py * W + px, (uint64_t)j);
                gb::Pool pool;
                gb::Tracer tr(ctx, rng, pool);
                gb::Tracer::SampleResult r;
                tr.processSample(px, py, r);
                film.add(r);
            }
    for (int b = 0; b < 5; ++b) {
        std::memcpy(block + (size_t)b * W * H * 4, film.block[b].data(), sizeof(double) * (size_t)W * H * 4);
        std::memcpy(light + (size_t)b * W * H * 3, film.light[b].data(), sizeof(double) * (size_t)W * H * 3);
    }
    counters[0] = h->sc.raysTraced - r0; counters[1] = h->sc.shadowRaysTraced - s0; counters[2] = ctx.unsupported; counters[3] = film.invalidPuts;
    counters[4] = ctx.walks; counters[5] = ctx.walksOk; counters[6] = ctx.propagated;
}
