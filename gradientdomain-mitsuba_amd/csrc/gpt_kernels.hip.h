// gpt_kernels.hip.h -- gfx950 kernels of the gradient-domain path tracer's per-pixel sampling.
//
// What the reference does per sample on a CPU thread (/root/reference/src/integrators/gpt/gpt.cpp:
// renderBlock :1220-1355, evaluatePoint :397-436, evaluate :468-1180, the shifts :242-369, vertex
// classification :176-231) and the Mitsuba pieces those call (cited inline, paths relative to
// /root/reference/), re-designed for CDNA4:
//
//   * one lane = one pixel, walking its spp base paths with the four offset paths carried alongside;
//     a wave = an 8x8 pixel tile (coherent primaries), a block = 4 waves.  Lanes whose base path has
//     ended wait until at least regenMin (56) lanes of the wave are idle, then regenerate together.  The first bounce of a
//     sample (five primaries, four unconnected offsets: ~11 rays) costs as much as all its later bounces together, and in a
//     wave that mixes the two both codes run every iteration with partial exec masks: measured on the config-2 frame,
//     regenerating at 24 idle lanes gives 3.5 Gray/s, at 56 4.0 Gray/s, never (64) 3.6 Gray/s -- the late threshold keeps the
//     first bounces together and still overlaps the few long Russian-roulette survivors with the next generation
//     (persistent wavefront, no cross-lane state shuffling).
//   * BVH2 flattened to HBM: 64-byte inner nodes holding both children's fp32 bounds (rounded outward, tested in fp64) and
//     references (leaves of <= 2 triangles are referenced directly and cost no node fetch), triangles in
//     leaf order as 80-byte projection records (the reference's TriAccel test, triaccel.h:96-158, in fp64)
//     plus 160-byte shading records.  Scenes whose node+triangle arrays fit the LDS budget are staged into
//     LDS once per block; the traversal stack always lives in LDS ([level][lane], conflict-free).
//   * fp64 throughout, like the reference's DOUBLE_PRECISION build (MI355X vector fp64 is half the fp32
//     rate, not 1/16th).  Compiled with -ffp-contract=off.
//   * film: per-pixel sample SUMS in HBM (31 doubles per pixel, component-major so a wave's update is one
//     contiguous run per component) instead of 15 scattered splats per sample; a resolve kernel gathers
//     the 4 neighbours and reproduces the ImageBlock::put arithmetic.  Samples whose box-filter footprint
//     is not the expected single pixel (|u| within 1e-5 of a pixel edge) take an exact generic path with
//     fp64 atomics into a spill film.
//   * random numbers: one SplitMix64 counter stream per (seed, pixel, sample); double in [0,1) from the top
//     52 bits as Random::nextFloat does (src/libcore/random.cpp).  Consumption order per sample = SURVEY A.4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gdpt_tr {

typedef double Float;
#define GD_EPSILON        1e-7                 /* include/mitsuba/core/constants.h:25 (double build) */
#define GD_SHADOW_EPSILON 1e-5                 /* constants.h:26 */
#define GD_DELTA_EPSILON  ((double)1e-3f)      /* constants.h:31 */
#define GD_D_EPSILON      1e-14                /* gpt.cpp:63 */
#define GD_PI             3.14159265358979323846
#define GD_INV_PI         0.31830988618379067154
#define GD_INV_TWOPI      0.15915494309189533577
#define GD_INF            (__builtin_huge_val())

constexpr int TBLK = 256;          // threads per block (16x16 px)
constexpr int STACK_DEPTH = 40;    // most BVH traversal stack entries per lane (LDS; the launches allocate what the scene's tree needs: up to three per 4-wide node)
constexpr int REGEN_MIN = 56;      // default number of idle lanes in a wave before they regenerate together (ConfigD::regenMin)
constexpr int SLICE_FILL = 2;      // sample slices: aim at this many work items per resident block slot ...
constexpr int LOG_CHUNK = 16;      // wider reconstruction filters: samples per pixel logged between two gathers (32 doubles each)
constexpr int SLICE_MIN_SPP = 8;   // ... but never fewer samples than this per slice (the end of a slice runs with idle lanes)
constexpr int NREC = 31;           // per-pixel record components
constexpr int NQ = 74;             // doubles per continuation record (layout: gpt_render.hip.h, q_store / q_load)
constexpr int LDS_SCENE_BYTES = 40 * 1024;   // node + triangle + shading + material + emitter tables of a small scene

struct d3 { Float x, y, z; };
__device__ __forceinline__ d3 mk(Float x, Float y, Float z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 mk(Float a) { return mk(a, a, a); }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 operator-(d3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ d3 operator*(d3 a, Float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ d3 operator*(Float s, d3 a) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ d3 operator*(d3 a, d3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ d3 operator/(d3 a, Float s) { const Float r = 1.0 / s; return mk(a.x * r, a.y * r, a.z * r); } // TVector3::operator/ multiplies by the reciprocal
__device__ __forceinline__ Float dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ Float len2(d3 a) { return dot(a, a); }
__device__ __forceinline__ Float len(d3 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ d3 normalize(d3 a) { return a / len(a); }
__device__ __forceinline__ Float maxc(d3 a) { return fmax(a.x, fmax(a.y, a.z)); }
__device__ __forceinline__ Float safe_sqrt(Float v) { return sqrt(fmax(0.0, v)); }
__device__ __forceinline__ Float signum(Float v) { return v < 0 ? -1.0 : (v > 0 ? 1.0 : 0.0); }
__device__ __forceinline__ Float comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
__device__ __forceinline__ bool is_finite_d(Float v) { return (v - v) == 0; }

// ---- device scene ---------------------------------------------------------------------------------
// 64 B inner node: the bounds of BOTH children (fp32, rounded outward on the host) and their references, so a visit is one
// record and a leaf never costs a node fetch.  Reference: top bit set = leaf, (first triangle << 3) | (count - 1); else the
// index of an inner node.
typedef float f2 __attribute__((ext_vector_type(2)));
// A node has FOUR children (the binary SAH tree with every second level folded into its parent, gpt_capi.hip), in one of two layouts the scene
// picks by where its tables live:
struct BvhNode {                // 128 B -- scenes staged into LDS (SceneD::quantNodes == 0): fp32 boxes, the cheapest test; LDS bytes cost little
    f2 b[4][3];                 // per child and axis: (lo, hi) -- one packed-FMA operand of the slab test
    uint32_t child[4];          // BVH_NONE: no such child (its box is never looked at)
    uint32_t pad[4];
};
struct BvhNodeQ {               // 64 B -- scenes in HBM (quantNodes == 1): boxes as 8-bit offsets on the node's own grid, plane = org + q * scale exactly
    float org[3];               // (scale is a power of two), lower planes rounded down and upper planes up on the host.  Half the tree's footprint in an L2
    float scale[3];             // that the render kernels' scratch traffic washes through (atrium frame 72.4 -> 67.0 ms), for ~30 more VALU operations per
    uint32_t child[4];          // node -- which the LDS-resident scenes, bound by issue, pay for and get nothing back (Cornell 61.2 -> 65.2 ms: hence two layouts)
    uint32_t qlo[3], qhi[3];    // per axis: byte i = child i's lower / upper plane
};
constexpr uint32_t BVH_LEAF = 0x80000000u;
constexpr uint32_t BVH_NONE = 0xffffffffu;      // (also what a finished traversal holds: a leaf's first triangle is < 2^28, so no leaf reference looks like this)
struct TriIsect {           // 80 B: the reference's TriAccel (triaccel.h:37-57) in fp64
    Float n_u, n_v, n_d, a_u, a_v, b_nu, b_nv, c_nu, c_nv;
    int k, pad;
};
struct TriShade {           // 160 B: what fillIntersectionRecord needs (skdtree.h:343-428), constant per flat triangle
    d3 p0, p1, p2;
    d3 n, s, t;             // shading frame == geometric frame normal for meshes without vertex normals
    int material, emitter;  // emitter = -1 if none
    int origIndex, smooth;  // smooth = 1: the triangle has per-vertex normals (TriNormals table), its frame depends on the hit
};
struct TriNormals { d3 n0, n1, n2, dpdu; };         // 96 B: per-vertex normals in leaf order (only scenes that have any) + its.dpdu: the
                                                    // first edge, or the UV tangent of a mesh with texture coordinates (skdtree.h:373-380)
struct MaterialD {           // 112 B (a multiple of 16: tables are staged into LDS with 16-byte copies)
    int type, distribution, sampleVisible, twoSided;
    d3 reflectance, eta, k;
    Float alphaU, alphaV;
    int tex, pad2;           // bitmap texture on `reflectance` / `specularReflectance` (index into SceneD::tex), -1 = the constant above
};
struct TriUV { Float uv[6]; };   // 48 B: per-vertex texture coordinates (u0 v0 u1 v1 u2 v2) in leaf order (only scenes that have any)
// `<texture type="bitmap">` as the G-PT path evaluates it (src/textures/bitmap.cpp:431-452 -> MIPMap::evalBox / evalBilinear on level 0,
// mipmap.h:566-596; with filterType "trilinear" / "ewa" the hit of a camera ray instead gets TMIPMap::eval over the pyramid, :628-712).
// Texels are doubles (the reference's MIP map holds Float), [h][w][3], top row first, level after level, in HBM.
constexpr int TEX_MAX_LEVELS = 16, TEX_LUT_SIZE = 64;     // (MTS_MIPMAP_LUT_SIZE, mipmap.h:37)
struct TexD {
    int w, h, wrapU, wrapV;     // wrap: 0 repeat, 1 clamp, 2 mirror, 3 zero, 4 one (ReconstructionFilter::EBoundaryCondition as bitmap.cpp:324-338 names them)
    int filter, levels;         // 0 nearest (evalBox), 1 bilinear, 2 trilinear, 3 ewa; MIP levels (1 for nearest / bilinear)
    Float uscale, vscale, uoffset, voffset;   // Texture2D, texture.cpp:27-45,113
    Float scale;                // BSDF::ensureEnergyConservation's ScaleTexture factor (1 = none)
    const Float *texels;        // level 0, then the levels of the pyramid one after the other (host-built: gpt_capi.hip build_pyramid)
    int lw[TEX_MAX_LEVELS], lh[TEX_MAX_LEVELS];
    unsigned loff[TEX_MAX_LEVELS];            // first texel of a level in `texels`
    Float ratioX[TEX_MAX_LEVELS], ratioY[TEX_MAX_LEVELS];   // m_sizeRatio
    Float maxAnisotropy;
    Float lut[TEX_LUT_SIZE];    // m_weightLut (Gaussian), mipmap.h:297-301
};
// `<emitter type="envmap">` (src/emitters/envmap.cpp): the latitude-longitude map in a MIP pyramid whose texels went through half precision
// (TMIPMap<Spectrum, SpectrumHalf>; kept here as the doubles those halves are), repeat in u / clamp in v, EWA with maxAnisotropy 10; float
// cdf tables over luminance x sin(theta) for light sampling (envmap.cpp:258-325).  Host-built (gpt_capi.hip).
struct EnvMapD {
    TexD tex;
    const float *cdfRows, *cdfCols;         // (h + 1) and h x (w + 1)
    const Float *rowWeights;                // sin((y + 0.5) pi / h)
    Float normalization, scale, pixelSizeX, pixelSizeY;
    Float toWorld[9], toLocal[9];           // linear part of the emitter's toWorld and its inverse, row-major
};
static_assert(sizeof(BvhNode) == 128 && sizeof(BvhNodeQ) == 64 && sizeof(TriIsect) % 16 == 0 && sizeof(TriShade) % 16 == 0 && sizeof(MaterialD) % 16 == 0 && sizeof(TriNormals) % 16 == 0, "LDS staging copies 16-byte words");
struct EmitterD {           // numTris == 0: the environment emitter (`constant`, src/emitters/constant.cpp); -1: `point` (point.cpp)
    int firstEmTri, numTris, cdfOffset, rectangle;     // rectangle: a `rectangle` shape's light -- sampled as Rectangle::samplePosition does
    d3 radiance;            // point: intensity
    Float invSurfaceArea;
    d3 position;            // point emitters only
    Float pad2;
    Float rect[12];         // rectangle: rows of its 3x4 objectToWorld
    d3 rectN;               // rectangle: its frame's normal
    Float pad3;
};
struct EmTri { d3 p0, p1, p2; };
static_assert(sizeof(EmitterD) % 16 == 0, "LDS staging copies 16-byte words");
struct CameraD {
    Float m[12];            // rows of the 3x4 camera-to-world
    Float nearClip, farClip, tanHalf, aspect, invW, invH;
    int width, height;
    int thinlens, needsTime;// needsTime: shutterClose > shutterOpen -- a sample draws its time sample (Sensor::needsTimeSample, sensor.h:290).  thinlens 1: `thinlens` sensor (thinlens.cpp): rays start on the aperture and pass through the focus point of their pixel
    Float apertureRadius, focusDistance;
    Float cropX, cropY;     // crop window of the film: image pixel (x, y) = pixel (x + cropX, y + cropY) of the full film, whose size invW / invH / aspect are taken from
};
struct SceneD {
    const BvhNode *nodes;
    const TriIsect *isect;
    const TriShade *shade;
    const MaterialD *mats;
    const EmitterD *emitters;
    const EmTri *emTris;
    const Float *emCdf;         // per-emitter triangle-area cdfs, concatenated
    const Float *emitterCdf;    // scene-level emitter cdf (numEmitters + 1)
    Float emitterNormalization;
    int numNodes, numTris, numEmitters, numMats, ldsScene;
    int numEmTris, numEmCdf;    // entries of emTris / emCdf (emitterCdf has numEmitters + 1)
    int ldsBytes;               // bytes of the staged tables (16-byte words per table), 0 if the scene is not LDS-resident
    uint32_t rootRef;
    float boundM;               // largest |coordinate| of the node bounds
    int quantNodes;             // `nodes` holds BvhNodeQ records (scenes that are not LDS-resident)
    const TriNormals *vn;       // per-vertex normals in leaf order, nullptr if the scene has none
    const TriUV *uv;            // per-vertex texture coordinates in leaf order, nullptr if the scene has none
    const unsigned char *hasUV; // per triangle: 1 = its mesh has texture coordinates (else its.uv = the barycentrics, skdtree.h:403-405)
    const TexD *tex;            // bitmap textures
    int numTex;
    int envIndex;               // position of the environment emitter in the emitter list, -1: none
    const EnvMapD *envMap;      // never null (a zeroed record when the scene has no environment map): a load through it may be hoisted above the test of
    int hasEnvMap;              // this flag -- the environment is an `envmap` (else `constant`: EmitterD::radiance)
    d3 bsCenter;                // its bounding sphere (ConstantBackgroundEmitter::m_sceneBSphere)
    Float bsRadius;
    CameraD cam;
};
struct ConfigD {
    int maxDepth, rrDepth, strictNormals, spp;
    int regenMin;               // idle lanes of a wave before they start new samples together (tuning, not a reference parameter)
    int sBase, sCount;          // the samples [sBase, sBase + sCount) of every pixel are rendered by this launch (spp = the whole count)
    int handoffEarly;           // staged pipeline: a sample leaves the first-stage kernel as soon as no offset is RAY_NOT_CONNECTED (1: LDS-resident scenes, round 6) or only
                                // when every offset is RAY_CONNECTED (0: HBM-resident scenes -- their 128-register k_continue pays more for the extra bounce than the first stage saves)
    Float shiftThreshold;
    unsigned long long seed;
};
struct FilmD {
    Float *rec;                 // [NREC][recRows][W] per-pixel sample sums; row index = y - (y0 - 1)
    Float *recExtra;            // [slices-1][NREC][recRows][W]: the sums of sample slices 1.. of a launch, folded into rec after it
    Float *spill;               // [5][spillRows][W][4] exact generic puts (R,G,B,weight); row index = y - (y0 - 2): TWO rows beyond the strip on either side -- a sample
                                // within 1e-5 of a pixel edge lands in both pixels (the box filter's radius is 0.5 + 1e-5, box.cpp:38), so its neighbour puts reach two
                                // rows from its own: the border of the reference's blocks is rfilter border 1 + extraBorder 1 (gpt_wr.cpp:31-44)
    const Float *fValues;       // nullptr: box filter (the fast path below); else the 32-entry table of ReconstructionFilter::configure
    Float fRadius, fScale;      // of that table (rfilter.cpp:37-55)
    Float *log;                 // wider filters: sample log [32][logChunk][logRows][W] (30 sums, sx, sy of every sample of a chunk), gathered by k_gather_log
    int logChunk;
    int logY0, logRows;         // the log covers the film's rows plus the filter's reach above and below (clipped to the image): a strip renders those rows
                                // itself instead of receiving them, so its gathered rows are bit-identical to the same rows of a whole-image film
    // Continuation queue (gpt_render.hip.h, k_continue): a sample whose four offset paths are all connected to the base path (or dead) leaves
    // the general kernel; its state travels through HBM to a lean kernel that runs the rest of the base path with dense waves.
    Float *qRec;                // [NQ][qCapacity] records, slot = (sample - sBase) * qPixels + tile * TBLK + thread; nullptr: no hand-off
    unsigned *qList;            // slots of the handed-off samples in arrival order
    unsigned *qCount;           // [0] entries in qList, [1] next entry to take
    unsigned qCapacity, qPixels;
    // Primary hits of the launch's samples (k_primary): [15][qCapacity] doubles (t, u, v of the base ray and the four offset rays) and
    // [5][qCapacity] leaf-order triangle indices, in the same slots as the continuation records.  nullptr: start_path traces them itself.
    Float *pHit;
    int *pPrim;
    unsigned long long *stats;  // [5]: closest rays, shadow rays, paths, path length sum, puts dropped as invalid
    const int *cancel;          // device flag set by gdpt_film_cancel: waves stop starting samples (Integrator::cancel, the `stop` flag of gpt.cpp:1246,1254)
    int W, H, y0, y1, recRows, spillRows;
    size_t recStride;           // recRows * W
};

// ---- RNG --------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
#ifdef GDPT_SERIAL_STREAM
// gpt_serial_capi.hip's build of the sampler: every draw comes from ONE SFMT-19937 stream, as `mitsuba -p 1` consumes it (IndependentSampler::next1D,
// independent.cpp:94-96 -> Random::nextFloat, random.cpp:616-626).  The 624 state words and the read index (word 624) sit in LDS; the lane that runs the
// serial film owns them.  init() is what a per-sample stream needs and a serial one must not have: the position carries over from sample to sample.
struct Rng {
    uint32_t *w;
    __device__ __forceinline__ void init(uint64_t, uint64_t, uint64_t) {}
    __device__ __forceinline__ uint64_t position() const { return 0; }             // (a sample never leaves its lane in the serial form: nothing to carry in a queue record)
    __device__ __forceinline__ void set_position(uint64_t) {}
    __device__ void regenerate();
    __device__ __forceinline__ Float next1D()
    {
        if (w[624] >= 624u) { regenerate(); w[624] = 0; }
        const uint32_t i = w[624];
        const uint64_t u = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);          // State::gen_rand64, random.cpp:285-293
        w[624] = i + 2;
        w[625]++;                                                                   // draws so far (returned to the host: tests hold it against the oracle's count)
        return __longlong_as_double((long long)((u >> 12) | 0x3FF0000000000000ULL)) - 1.0;
    }
};
#else
struct Rng {
    uint64_t s;
    __device__ __forceinline__ void init(uint64_t seed, uint64_t pixel, uint64_t sample)
    {
        s = mix64(seed + 0x9E3779B97F4A7C15ULL * (pixel + 1));
        s = mix64(s ^ (0xD1B54A32D192ED03ULL * (sample + 1)));
    }
    __device__ __forceinline__ uint64_t position() const { return s; }             // what a queue record carries of the stream
    __device__ __forceinline__ void set_position(uint64_t p) { s = p; }
    __device__ __forceinline__ Float next1D()
    {
        s += 0x9E3779B97F4A7C15ULL;
        return __longlong_as_double((long long)((mix64(s) >> 12) | 0x3FF0000000000000ULL)) - 1.0;
    }
};
#endif

// ---- frames -------------------------------------------------------------------------------------------
struct Frame3 { d3 s, t, n; };
__device__ __forceinline__ d3 toLocal(const Frame3 &f, d3 v) { return mk(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }
__device__ __forceinline__ d3 toWorld(const Frame3 &f, d3 v) { return f.s * v.x + f.t * v.y + f.n * v.z; }
__device__ __forceinline__ Float tanTheta(d3 v) { const Float t = 1 - v.z * v.z; return t <= 0.0 ? 0.0 : sqrt(t) / v.z; } // frame.h:122

// ---- ray / hit ----------------------------------------------------------------------------------------
struct Hit { Float t, u, v; int prim; };   // prim = index in leaf order, -1 = miss

// Scene accessors: LDS-staged copy for small scenes, HBM otherwise.
struct SceneView {
    const BvhNode *nodes;
    const TriIsect *isect;
    const TriShade *shade;
    const MaterialD *mats;
    const EmitterD *emitters;
    const EmTri *emTris;        // light-sampling tables: LDS copies for small scenes (sample_emitter_direct walks them with dependent loads)
    const Float *emCdf;
    const Float *emitterCdf;
    const TriNormals *vn;       // nullptr: no triangle has vertex normals
    const TriUV *uv;            // nullptr: no triangle has texture coordinates
    const unsigned char *hasUV;
    const TexD *tex;
    uint32_t rootRef;
    float boundM;               // largest |coordinate| of the node bounds (error bound of the fp32 slab test)
    int quant;                  // nodes are BvhNodeQ records (uniform; a compile-time 0 in the kernels that stage the scene into LDS)
    int leafExit;               // trace() leaves its inner-node loop once half of the wave's lanes on their way hold a leaf: a compile-time 1 in every kernel that reads
                                // the tables from HBM, 0 in the kernels that stage the scene into LDS
};

// TriAccel::rayIntersect, triaccel.h:96-158
__device__ __forceinline__ bool tri_test(const TriIsect &ta, d3 o, d3 d, Float mint, Float maxt, Float &u, Float &v, Float &t)
{
    Float o_u, o_v, o_k, d_u, d_v, d_k;
    if (ta.k == 0) { o_u = o.y; o_v = o.z; o_k = o.x; d_u = d.y; d_v = d.z; d_k = d.x; }
    else if (ta.k == 1) { o_u = o.z; o_v = o.x; o_k = o.y; d_u = d.z; d_v = d.x; d_k = d.y; }
    else if (ta.k == 2) { o_u = o.x; o_v = o.y; o_k = o.z; d_u = d.x; d_v = d.y; d_k = d.z; }
    else return false;
    t = (ta.n_d - o_u * ta.n_u - o_v * ta.n_v - o_k) / (d_u * ta.n_u + d_v * ta.n_v + d_k);
    if (t < mint || t > maxt) return false;
    const Float hu = o_u + t * d_u - ta.a_u;
    const Float hv = o_v + t * d_v - ta.a_v;
    u = hv * ta.b_nu + hu * ta.b_nv;
    v = hu * ta.c_nu + hv * ta.c_nv;
    return u >= 0 && v >= 0 && u + v <= 1.0;
}

// Ray/box slab test, conservative in fp32.  The bounds are fp32 already (rounded outward on the host); per ray the test keeps
// rdf = fl32(1/d) and, per axis, the two offsets a0/a1 = fl32(o*rdf) -/+ E so that t = b*rdf - a is ONE packed FMA for a (lo, hi)
// pair, already pushed outward by the error bound E = 2^-21 (M + |o|) |rdf| of that axis (M = largest |bound| of the scene): the
// exact slab parameter differs from the computed one by at most 2^-24 (|b rdf| + |o rdf| + |t|) (roundings of rdf, o*rdf and
// the FMA).  The plane the ray meets first gets -E, the other +E (chosen by the sign of rdf), so a box is never culled that the exact
// test would visit; the extra visits cost time, not correctness -- triangles are still tested in fp64.  14 VALU instructions per
// box (3 v_pk_fma_f32, 3 min, 3 max, max3/min3 with the ray interval, 1 compare) against 28 fp64 ones.
struct RayF {
    f2 rx, ry, rz;              // (rdf, rdf)
    f2 ax, ay, az;              // -(a0, a1)
    float mint, maxt;           // ray interval, rounded outward
};
__device__ __forceinline__ float slab_rdf(Float d) { return fabs(d) < 1e-30 ? (d < 0 ? -1e30f : 1e30f) : fmaxf(fminf((float)(1.0 / d), 1e30f), -1e30f); }
__device__ __forceinline__ float up_f(Float v) { return (float)v * (1.0f + 0x1p-22f) + 1e-37f; }
__device__ __forceinline__ float down_f(Float v) { return (float)v * (1.0f - 0x1p-22f) - 1e-37f; }
__device__ __forceinline__ void slab_axis(Float o, Float d, float M, f2 &r, f2 &na)
{
    const float rdf = slab_rdf(d);
    const float orf = (float)(o * (Float)rdf);
    const float E = 0x1p-21f * (M + fabsf((float)o) + 0x1p-100f) * fabsf(rdf) + 1e-37f;
    const float s = rdf < 0 ? -E : E;
    r = (f2){rdf, rdf};
    na = (f2){-(orf + s), -(orf - s)};            // t(lo) = lo*rdf - orf - s,  t(hi) = hi*rdf - orf + s
}
__device__ __forceinline__ RayF ray_f(d3 o, d3 d, Float mint, Float maxt, float M)
{
    RayF R;
    slab_axis(o.x, d.x, M, R.rx, R.ax);
    slab_axis(o.y, d.y, M, R.ry, R.ay);
    slab_axis(o.z, d.z, M, R.rz, R.az);
    R.mint = fmaxf(down_f(mint), 0.0f); R.maxt = up_f(maxt);       // (>= 0, still <= mint: a box's entry parameter is then a non-negative float, whose bits order as the value does)
    return R;
}
__device__ __forceinline__ bool box_test(const f2 (&b)[3], const RayF &R, float &tn)
{
    const f2 tx = __builtin_elementwise_fma(b[0], R.rx, R.ax), ty = __builtin_elementwise_fma(b[1], R.ry, R.ay), tz = __builtin_elementwise_fma(b[2], R.rz, R.az);
    const float n = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), R.mint));
    const float f = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), R.maxt));
    tn = n;
    return n <= f;
}
// A child box of a BvhNodeQ: plane = org + q * scale, so its slab parameter t = plane * rdf - a = q * (scale * rdf) + (org * rdf - a): per node and axis one
// product (exact: scale is a power of two) and one packed FMA give (sr, c), per child and axis ONE packed FMA t = q * sr + c as before.  Roundings: rdf,
// o * rdf, the sum inside a, c and t: each at most 2^-24 of a term bounded by (M + |o|) |rdf| -- five of them against the E = 8 x 2^-24 (M + |o|) |rdf| the
// ray's offsets carry.  M covers |org| and every decoded plane (host: boundM).
__device__ __forceinline__ bool box_test(f2 qx, f2 qy, f2 qz, f2 sx, f2 sy, f2 sz, f2 cx, f2 cy, f2 cz, const RayF &R, float &tn)
{
    const f2 tx = __builtin_elementwise_fma(qx, sx, cx), ty = __builtin_elementwise_fma(qy, sy, cy), tz = __builtin_elementwise_fma(qz, sz, cz);
    const float n = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), R.mint));
    const float f = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), R.maxt));
    tn = n;
    return n <= f;
}

// One inner node of the traversal: the four slab tests, the children the ray enters sorted by entry parameter (a five-comparator network on
// (key, child) pairs; the key of a child the ray misses -- or that is not there -- is all ones, so it sorts behind every hit), the far ones
// pushed far-to-near, the nearest returned; BVH_NONE when nothing is entered and the stack is empty.  A node ends two levels of the binary tree: half
// the dependent round trips per ray.  The ORDER in which nodes are visited only decides how much is culled: closest hits are the minimum over all
// triangles whose boxes the ray enters, any-hit is an OR.
__device__ __forceinline__ void sort2(uint32_t &ka, uint32_t &ca, uint32_t &kb, uint32_t &cb)
{
    const bool s = kb < ka;
    const uint32_t k = s ? kb : ka, c = s ? cb : ca;
    kb = s ? ka : kb; cb = s ? ca : cb;
    ka = k; ca = c;
}
__device__ __forceinline__ uint32_t node_sorted(uint32_t (&k)[4], uint32_t (&c)[4], int *stack, int &sp)
{
    sort2(k[0], c[0], k[1], c[1]); sort2(k[2], c[2], k[3], c[3]);
    sort2(k[0], c[0], k[2], c[2]); sort2(k[1], c[1], k[3], c[3]);
    sort2(k[1], c[1], k[2], c[2]);
    if (k[0] == BVH_NONE) {
        if (sp == 0) return BVH_NONE;
        sp--;
        return (uint32_t)stack[sp * TBLK];
    }
    if (k[3] != BVH_NONE && sp < STACK_DEPTH) { stack[sp * TBLK] = (int)c[3]; sp++; }
    if (k[2] != BVH_NONE && sp < STACK_DEPTH) { stack[sp * TBLK] = (int)c[2]; sp++; }
    if (k[1] != BVH_NONE && sp < STACK_DEPTH) { stack[sp * TBLK] = (int)c[1]; sp++; }
    return c[0];
}
__device__ __forceinline__ uint32_t node_step(const BvhNode &n, const RayF &R, int *stack, int &sp)
{
    uint32_t k[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float tn;
        const bool h = box_test(n.b[i], R, tn) && n.child[i] != BVH_NONE;
        k[i] = h ? __float_as_uint(tn) : BVH_NONE;
        c[i] = n.child[i];
    }
    return node_sorted(k, c, stack, sp);
}
__device__ __forceinline__ uint32_t node_step(const BvhNodeQ &n, const RayF &R, int *stack, int &sp)
{
    uint32_t k[4], c[4];
    const float srx = n.scale[0] * R.rx.x, sry = n.scale[1] * R.ry.x, srz = n.scale[2] * R.rz.x;
    const f2 sx = (f2){srx, srx}, sy = (f2){sry, sry}, sz = (f2){srz, srz};
    const f2 cx = __builtin_elementwise_fma((f2){n.org[0], n.org[0]}, R.rx, R.ax), cy = __builtin_elementwise_fma((f2){n.org[1], n.org[1]}, R.ry, R.ay),
             cz = __builtin_elementwise_fma((f2){n.org[2], n.org[2]}, R.rz, R.az);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float tn;
        const f2 qx = (f2){(float)((n.qlo[0] >> (8 * i)) & 255u), (float)((n.qhi[0] >> (8 * i)) & 255u)};      // (v_cvt_f32_ubyte<i>)
        const f2 qy = (f2){(float)((n.qlo[1] >> (8 * i)) & 255u), (float)((n.qhi[1] >> (8 * i)) & 255u)};
        const f2 qz = (f2){(float)((n.qlo[2] >> (8 * i)) & 255u), (float)((n.qhi[2] >> (8 * i)) & 255u)};
        const bool h = box_test(qx, qy, qz, sx, sy, sz, cx, cy, cz, R, tn) && n.child[i] != BVH_NONE;
        k[i] = h ? __float_as_uint(tn) : BVH_NONE;
        c[i] = n.child[i];
    }
    return node_sorted(k, c, stack, sp);
}
// the step on whichever layout the scene's nodes have (a uniform branch; the LDS-scene kernels hold quant == 0 at compile time)
__device__ __forceinline__ uint32_t node_step(const SceneView &sv, uint32_t ref, const RayF &R, int *stack, int &sp)
{
    if (sv.quant) return node_step(reinterpret_cast<const BvhNodeQ *>(sv.nodes)[ref], R, stack, sp);
    return node_step(sv.nodes[ref], R, stack, sp);
}

// ShapeKDTree::rayIntersect (closest, skdtree.cpp:112-142) / rayIntersect(ray) (shadow, :207-226) on the BVH.
// The adaptive epsilon of :126-129 / :214-217 is applied by the callers (ray_mint_*).  Returns closest hit
// (ANY = false) or whether anything is hit (ANY = true).
struct TravCount { unsigned nodes, tris; };     // traversal statistics of the probe kernel (never live in the render kernel)
template <bool ANY, bool COUNT = false>
__device__ __forceinline__ bool trace(const SceneView &sv, int *stack /* [STACK_DEPTH][TBLK] + tid */, d3 o, d3 d, Float mint, Float maxt, Hit &hit, TravCount *tc = nullptr)
{
    hit.prim = -1;
    hit.t = GD_INF;
    if (!(maxt > mint)) return false;
    RayF R = ray_f(o, d, mint, maxt, sv.boundM);
    constexpr uint32_t DONE = BVH_NONE;
    int sp = 0;
    uint32_t ref = sv.rootRef;
    while (true) {
        // inner nodes: every lane walks down to ITS next leaf ("while-while": the node code and the triangle code each run with fuller exec masks than one
        // loop that alternates per lane) -- or, in the kernels that read the tables from HBM (leafExit), only until half of the lanes still on their way hold a
        // leaf: a lane needs ~2 steps from one leaf to its next, the slowest of 64 several times that, and the lanes already at a leaf would wait for it with
        // their triangle records' latency still ahead of them.  Leaving early only changes WHEN a lane's leaf is tested, not what the lane visits or in which
        // order (a lane that still holds an inner node passes the leaf part).  Atrium frame 70.3 -> 64 ms (threshold sweep: 2/8 66.0, 3/8 64.4, 4/8 and 5/8 63.8),
        // traversal-only kernel 1.25 -> 1.04 ms per 4.2 M rays; the LDS-resident box loses 1-4 % with it (a leaf costs it a few cycles of LDS latency, the
        // extra passes through the loop more): those kernels keep the plain loop.
        if (sv.leafExit) {
            while (true) {
                const bool inner = !(ref & BVH_LEAF);
                const unsigned long long innerMask = __ballot(inner);
                if (innerMask == 0) break;
                if ((int)__popcll(__ballot(!inner && ref != DONE)) >= (int)__popcll(innerMask)) break;
                if (inner) {
                    if (COUNT) tc->nodes++;
                    ref = node_step(sv, ref, R, stack, sp);
                }
            }
        } else {
            while (!(ref & BVH_LEAF)) {
                if (COUNT) tc->nodes++;
                ref = node_step(sv, ref, R, stack, sp);
            }
        }
        if (ref == DONE) break;
        if (!(ref & BVH_LEAF)) continue;
        const uint32_t first = (ref & ~BVH_LEAF) >> 3, cnt = (ref & 7u) + 1u;
        if (COUNT) tc->tris += cnt;
        for (uint32_t i = 0; i < cnt; i++) {
            Float u, v, t;
            // the whole 80-byte record in one go: read field by field as the test proceeds (k, then the plane, then the edge terms) it is
            // three dependent LDS / L2 round trips per triangle, and the traversal is bound by exactly that latency chain, not by bandwidth.
            // (Round 5 measured fetching the leaf's SECOND record before the first test -- a leaf holds one or two triangles -- in the kernels that read
            //  the tables from HBM: bit-identical hits, atrium frame 61.9 -> 65.1 ms: the 20 extra registers cost the 128-register builds more in spills
            //  than the overlapped round trip returns.  Not kept.)
            const TriIsect ta = sv.isect[first + i];
            if (tri_test(ta, o, d, mint, maxt, u, v, t)) {
                if (ANY) return true;
                maxt = t;
                R.maxt = up_f(t);
                hit.t = t; hit.u = u; hit.v = v; hit.prim = (int)(first + i);
            }
        }
        if (sp == 0) break;
        sp--;
        ref = (uint32_t)stack[sp * TBLK];
    }
    return hit.prim >= 0;
}

// The same traversal as a real function.  Used where little caller state is live (the five primary rays of a fresh sample): the
// callee gets its own tight register allocation and the call costs almost nothing; measured 47 -> 23 ms on the primary-only
// 1280x720x32 Cornell pass.  Inside bounce(), where ~150 registers of path state are live, inlining is the faster form.
// Arguments of a real call travel in VGPRs: only the four fields of the scene view the traversal reads are passed, which keeps the
// callee within the 128 registers of the 4-wave builds (with the whole view it needs 132 and costs them a wave per SIMD).
__device__ __noinline__ Hit trace_closest_fn(const BvhNode *nodes, const TriIsect *isect, uint32_t rootRef, float boundM, int quant, int leafExit, int *stack, d3 o, d3 d, Float mint, Float maxt)
{
    SceneView sv;
    sv.nodes = nodes; sv.isect = isect; sv.rootRef = rootRef; sv.boundM = boundM; sv.quant = quant; sv.leafExit = leafExit;
    Hit h;
    trace<false>(sv, stack, o, d, mint, maxt, h);
    return h;
}
__device__ __forceinline__ Hit trace_closest_call(const SceneView &sv, int *stack, d3 o, d3 d, Float mint, Float maxt)
{
    return trace_closest_fn(sv.nodes, sv.isect, sv.rootRef, sv.boundM, sv.quant, sv.leafExit, stack, o, d, mint, maxt);
}

__device__ __forceinline__ Float ray_mint_closest(d3 o, Float mint)
{ // skdtree.cpp:126-129
    if (mint == GD_EPSILON) mint *= fmax(fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z)), GD_EPSILON);
    return mint;
}
__device__ __forceinline__ Float ray_mint_shadow(d3 o, Float mint)
{ // skdtree.cpp:214-217 (no floor)
    if (mint == GD_EPSILON) mint *= fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z));
    return mint;
}

// ---- warps (src/libcore/warp.cpp) ----------------------------------------------------------------------
__device__ __forceinline__ d3 squareToCosineHemisphere(Float sx, Float sy)
{ // warp.cpp:43-52, 81-102
    const Float r1 = 2.0 * sx - 1.0, r2 = 2.0 * sy - 1.0;
    Float phi, r;
    if (r1 == 0 && r2 == 0) { r = phi = 0; }
    else if (r1 * r1 > r2 * r2) { r = r1; phi = (GD_PI / 4.0) * (r2 / r1); }
    else { r = r2; phi = (GD_PI / 2.0) - (r1 / r2) * (GD_PI / 4.0); }
    const Float px = r * cos(phi), py = r * sin(phi);
    Float z = safe_sqrt(1.0 - px * px - py * py);
    if (z == 0) z = (Float)1e-10f;
    return mk(px, py, z);
}

// ---- Fresnel (util.cpp:739-761, per channel) ----------------------------------------------------------------
__device__ __forceinline__ Float fresnel1(Float cosThetaI, Float e, Float kk)
{
    const Float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    const Float temp1 = e * e - kk * kk - sinThetaI2;
    const Float a2pb2 = safe_sqrt(temp1 * temp1 + kk * kk * e * e * 4);
    const Float a = safe_sqrt((a2pb2 + temp1) * 0.5);
    const Float term1 = a2pb2 + cosThetaI2, term2 = a * (2 * cosThetaI);
    const Float Rs2 = (term1 - term2) / (term1 + term2);
    const Float term3 = a2pb2 * cosThetaI2 + sinThetaI4, term4 = term2 * sinThetaI2;
    const Float Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
    return 0.5 * (Rp2 + Rs2);
}
__device__ __forceinline__ d3 fresnelConductorExact(Float c, d3 eta, d3 k) { return mk(fresnel1(c, eta.x, k.x), fresnel1(c, eta.y, k.y), fresnel1(c, eta.z, k.z)); }

// ---- math.cpp:25-72 -----------------------------------------------------------------------------------------
__device__ __forceinline__ Float erfinv_m(Float x)
{
    Float w = -log((1.0 - x) * (1.0 + x)), p;
    if (w < 5.0) {
        w = w - 2.5;
        p = 2.81022636e-08; p = 3.43273939e-07 + p * w; p = -3.5233877e-06 + p * w; p = -4.39150654e-06 + p * w;
        p = 0.00021858087 + p * w; p = -0.00125372503 + p * w; p = -0.00417768164 + p * w; p = 0.246640727 + p * w; p = 1.50140941 + p * w;
    } else {
        w = sqrt(w) - 3.0;
        p = -0.000200214257; p = 0.000100950558 + p * w; p = 0.00134934322 + p * w; p = -0.00367342844 + p * w;
        p = 0.00573950773 + p * w; p = -0.0076224613 + p * w; p = 0.00943887047 + p * w; p = 1.00167406 + p * w; p = 2.83297682 + p * w;
    }
    return p * x;
}
__device__ __forceinline__ Float erf_m(Float x)
{
    const Float a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429, p = 0.3275911;
    const Float sign = signum(x);
    x = fabs(x);
    const Float t = 1.0 / (1.0 + p * x);
    const Float y = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * exp(-x * x);
    return sign * y;
}
__device__ __forceinline__ Float hypot2(Float a, Float b)
{
    Float r;
    if (fabs(a) > fabs(b)) { r = b / a; r = fabs(a) * sqrt(1.0 + r * r); }
    else if (b != 0.0) { r = a / b; r = fabs(b) * sqrt(1.0 + r * r); }
    else r = 0.0;
    return r;
}

// ---- MicrofacetDistribution (src/bsdfs/microfacet.h): Beckmann (0), GGX (1), Phong / Ashikhmin-Shirley (2) -------------------
struct Mf { int type; Float aU, aV; bool sv; };
__device__ __forceinline__ Mf mf_of(const MaterialD &m)
{ // :70-71,135-144: visible-normal sampling is not supported for Phong
    Mf d; d.type = m.distribution; d.aU = fmax(m.alphaU, (Float)1e-4f); d.aV = fmax(m.alphaV, (Float)1e-4f); d.sv = m.sampleVisible != 0 && m.distribution != 2; return d;
}
// The Phong pieces are real calls (pow is large, and the two common distributions should not pay for its registers).
__device__ __forceinline__ Float phong_exponent(Float a) { return fmax(2.0 / (a * a) - 2.0, (Float)0.0); }   // computePhongExponent, :701-704
__device__ __noinline__ Float mf_phong_eval(Float aU, Float aV, d3 m)
{ // :215-221 with interpolatePhongExponent :554-565
    const Float eU = phong_exponent(aU), eV = phong_exponent(aV);
    const Float sinTheta2 = 1.0 - m.z * m.z;
    Float e = eU;
    if (!(aU == aV || sinTheta2 <= 0x1p-1024)) { const Float inv = 1 / sinTheta2; e = eU * (m.x * m.x * inv) + eV * (m.y * m.y * inv); }
    return sqrt((eU + 2) * (eV + 2)) * GD_INV_TWOPI * pow(m.z, e);
}
__device__ __noinline__ d3 mf_phong_sample(Float aU, Float aV, Float sx, Float sy, Float &pdf)
{ // sampleAll, :349-375 with sampleFirstQuadrant :707-715
    const Float eU = phong_exponent(aU), eV = phong_exponent(aV);
    Float phiM, exponent;
    auto quadrant = [&](Float u1) {
        phiM = atan(sqrt((eU + 2.0) / (eV + 2.0)) * tan(GD_PI * u1 * 0.5));
        const Float c = cos(phiM), sn = sin(phiM);
        exponent = eU * c * c + eV * sn * sn;
    };
    if (aU == aV) { phiM = (2.0 * GD_PI) * sy; exponent = eU; }
    else if (sy < (Float)0.25f) quadrant(4 * sy);
    else if (sy < (Float)0.5f) { quadrant(4 * (0.5 - sy)); phiM = GD_PI - phiM; }
    else if (sy < (Float)0.75f) { quadrant(4 * (sy - 0.5)); phiM += GD_PI; }
    else { quadrant(4 * (1 - sy)); phiM = 2 * GD_PI - phiM; }
    const Float sinPhiM = sin(phiM), cosPhiM = cos(phiM);
    const Float cosThetaM = pow(sx, 1.0 / (exponent + 2.0));
    pdf = sqrt((eU + 2.0) * (eV + 2.0)) * GD_INV_TWOPI * pow(cosThetaM, exponent + 1.0);
    if (pdf < (Float)1e-20f) pdf = 0;
    const Float sinThetaM = sqrt(fmax((Float)0.0, 1 - cosThetaM * cosThetaM));
    return mk(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
}
__device__ __forceinline__ Float mf_eval(const Mf &d, d3 m)
{ // :191-234
    if (m.z <= 0) return 0.0;
    const Float cosTheta2 = m.z * m.z;
    const Float be = ((m.x * m.x) / (d.aU * d.aU) + (m.y * m.y) / (d.aV * d.aV)) / cosTheta2;
    Float result;
    if (d.type == 0) result = exp(-be) / (GD_PI * d.aU * d.aV * cosTheta2 * cosTheta2);
    else if (d.type == 2) result = mf_phong_eval(d.aU, d.aV, m);
    else { const Float root = (1.0 + be) * cosTheta2; result = 1.0 / (GD_PI * d.aU * d.aV * root * root); }
    if (result * m.z < (Float)1e-20f) result = 0;
    return result;
}
__device__ __forceinline__ Float mf_projectRoughness(const Mf &d, d3 v)
{ // :531-541
    const Float invSinTheta2 = 1 / (1.0 - v.z * v.z);
    if (d.aU == d.aV || invSinTheta2 <= 0) return d.aU;
    const Float cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
    return sqrt(cosPhi2 * d.aU * d.aU + sinPhi2 * d.aV * d.aV);
}
__device__ __forceinline__ Float mf_G1(const Mf &d, d3 v, d3 m)
{ // :477-514
    if (dot(v, m) * v.z <= 0) return 0.0;
    const Float tanT = fabs(tanTheta(v));
    if (tanT == 0.0) return 1.0;
    const Float alpha = mf_projectRoughness(d, v);
    if (d.type != 1) {                                        // Beckmann and Phong share the rational approximation, :489-501
        const Float a = 1.0 / (alpha * tanT);
        if (a >= (Float)1.6f) return 1.0;
        const Float aSqr = a * a;
        return ((Float)3.535f * a + (Float)2.181f * aSqr) / (1.0 + (Float)2.276f * a + (Float)2.577f * aSqr);
    }
    const Float root = alpha * tanT;
    return 2.0 / (1.0 + hypot2(1.0, root));
}
__device__ __forceinline__ Float mf_pdfVisible(const Mf &d, d3 wi, d3 m)
{ // :470-474
    if (wi.z == 0) return 0.0;
    return mf_G1(d, wi, m) * fabs(dot(wi, m)) * mf_eval(d, m) / fabs(wi.z);
}
__device__ void mf_sampleVisible11(const Mf &d, Float thetaI, Float sx, Float sy, Float &slx, Float &sly)
{ // :573-702
    const Float SQRT_PI_INV = 1 / sqrt(GD_PI);
    if (d.type == 0) {
        if (thetaI < (Float)1e-4f) {
            const Float r = sqrt(-log(1.0 - sx)), ph = 2 * GD_PI * sy;
            slx = r * cos(ph); sly = r * sin(ph);
            return;
        }
        const Float tanThetaI = tan(thetaI), cotThetaI = 1 / tanThetaI;
        Float a = -1, c = erf_m(cotThetaI);
        const Float sample_x = fmax(sx, (Float)1e-6f);
        const Float fit = 1 + thetaI * ((Float)-0.876f + thetaI * ((Float)0.4265f - (Float)0.0594f * thetaI));
        Float b = c - (1 + c) * pow(1 - sample_x, fit);
        const Float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * exp(-cotThetaI * cotThetaI));
        int it = 0;
        while (++it < 10) {
            if (!(b >= a && b <= c)) b = 0.5 * (a + c);
            const Float invErf = erfinv_m(b);
            const Float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * exp(-invErf * invErf)) - sample_x;
            const Float derivative = normalization * (1 - invErf * tanThetaI);
            if (fabs(value) < (Float)1e-5f) break;
            if (value > 0) c = b; else a = b;
            b -= value / derivative;
        }
        slx = erfinv_m(b);
        sly = erfinv_m(2.0 * fmax(sy, (Float)1e-6f) - 1.0);
        return;
    }
    if (thetaI < (Float)1e-4f) {
        const Float r = safe_sqrt(sx / (1 - sx)), ph = 2 * GD_PI * sy;
        slx = r * cos(ph); sly = r * sin(ph);
        return;
    }
    const Float tanThetaI = tan(thetaI), a = 1 / tanThetaI;
    const Float G1 = 2.0 / (1.0 + safe_sqrt(1.0 + 1.0 / (a * a)));
    Float A = 2.0 * sx / G1 - 1.0;
    if (fabs(A) == 1) A -= signum(A) * GD_EPSILON;
    const Float tmp = 1.0 / (A * A - 1.0);
    const Float B = tanThetaI;
    const Float D = safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
    const Float slope_x_1 = B * tmp - D, slope_x_2 = B * tmp + D;
    slx = (A < 0.0 || slope_x_2 > 1.0 / tanThetaI) ? slope_x_1 : slope_x_2;
    Float S;
    if (sy > (Float)0.5f) { S = 1.0; sy = 2.0 * (sy - 0.5); }
    else { S = -1.0; sy = 2.0 * (0.5 - sy); }
    const Float z = (sy * (sy * (sy * (-0.365728915865723) + 0.790235037209296) - 0.424965825137544) + 0.000152998850436920) /
                    (sy * (sy * (sy * (sy * 0.169507819808272 - 0.397203533833404) - 0.232500544458471) + 1) - 0.539825872510702);
    sly = S * z * sqrt(1.0 + slx * slx);
}
__device__ d3 mf_sample(const Mf &d, d3 _wi, Float sx, Float sy, Float &pdf)
{ // :240-250 -> sampleVisible :421-467 / sampleAll :300-414
    if (d.sv) {
        const d3 wi = normalize(mk(d.aU * _wi.x, d.aV * _wi.y, _wi.z));
        Float theta = 0, phi = 0;
        if (wi.z < (Float)0.99999) { theta = acos(wi.z); phi = atan2(wi.y, wi.x); }
        const Float sinPhi = sin(phi), cosPhi = cos(phi);
        Float slx, sly;
        mf_sampleVisible11(d, theta, sx, sy, slx, sly);
        Float rx = cosPhi * slx - sinPhi * sly, ry = sinPhi * slx + cosPhi * sly;
        rx *= d.aU; ry *= d.aV;
        const Float normalization = 1.0 / sqrt(rx * rx + ry * ry + 1.0);
        const d3 m = mk(-rx * normalization, -ry * normalization, normalization);
        pdf = mf_pdfVisible(d, _wi, m);
        return m;
    }
    if (d.type == 2) return mf_phong_sample(d.aU, d.aV, sx, sy, pdf);
    Float cosThetaM, sinPhiM, cosPhiM, alphaSqr;
    if (d.aU == d.aV) {
        const Float ph = (2.0 * GD_PI) * sy;
        sinPhiM = sin(ph); cosPhiM = cos(ph);
        alphaSqr = d.aU * d.aU;
    } else {
        const Float phiM = atan(d.aV / d.aU * tan(GD_PI + 2 * GD_PI * sy)) + GD_PI * floor(2 * sy + 0.5);
        sinPhiM = sin(phiM); cosPhiM = cos(phiM);
        const Float cosSc = cosPhiM / d.aU, sinSc = sinPhiM / d.aV;
        alphaSqr = 1.0 / (cosSc * cosSc + sinSc * sinSc);
    }
    if (d.type == 0) {
        const Float tanThetaMSqr = alphaSqr * -log(1.0 - sx);
        cosThetaM = 1.0 / sqrt(1.0 + tanThetaMSqr);
        pdf = (1.0 - sx) / (GD_PI * d.aU * d.aV * cosThetaM * cosThetaM * cosThetaM);
    } else {
        const Float tanThetaMSqr = alphaSqr * sx / (1.0 - sx);
        cosThetaM = 1.0 / sqrt(1.0 + tanThetaMSqr);
        const Float temp = 1 + tanThetaMSqr / alphaSqr;
        pdf = GD_INV_PI / (d.aU * d.aV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
    }
    if (pdf < (Float)1e-20f) pdf = 0;
    const Float sinThetaM = sqrt(fmax(0.0, 1 - cosThetaM * cosThetaM));
    return mk(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
}

// ---- BSDFs --------------------------------------------------------------------------------------------------
enum { EDiffuseReflection = 0x1, EGlossyReflection = 0x4, EDeltaReflection = 0x10, EDeltaTransmission = 0x20, ESmooth = 0x5, EDelta = 0x30 };
enum { MEASURE_SOLID_ANGLE = 0, MEASURE_DISCRETE = 1 };
__device__ __forceinline__ int bsdfType(const MaterialD &m) { return m.type == 0 ? EDiffuseReflection : (m.type == 1 ? EDeltaReflection : (m.type == 3 ? (EDeltaReflection | EDeltaTransmission) : EGlossyReflection)); }
__device__ __forceinline__ Float bsdf_eta(const MaterialD &m) { return m.type == 3 ? m.eta.x : 1.0; }      // BSDF::getEta (dielectric.cpp:395; 1 elsewhere)

// fresnelDielectricExt, util.cpp:651-681
__device__ __forceinline__ Float fresnelDielectricExt(Float cosThetaI_, Float &cosThetaT_, Float eta)
{
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0; }
    const Float scale = (cosThetaI_ > 0) ? 1 / eta : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0) { cosThetaT_ = 0.0; return 1.0; }
    const Float cosThetaI = fabs(cosThetaI_), cosThetaT = sqrt(cosThetaTSqr);
    const Float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    const Float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5 * (Rs * Rs + Rp * Rp);
}
// SmoothDielectric::refract, dielectric.cpp:222-225
__device__ __forceinline__ d3 dielectric_refract(const MaterialD &m, d3 wi, Float cosThetaT)
{
    const Float eta = m.eta.x, invEta = 1 / eta;
    const Float scale = -(cosThetaT < 0 ? invEta : eta);
    return mk(scale * wi.x, scale * wi.y, cosThetaT);
}

// BSDF::eval and BSDF::pdf together (every call site of the hot path wants both):
// diffuse.cpp:110-127, conductor.cpp:223-254, roughconductor.cpp:257-319
// R: the material's reflectance / specularReflectance at the vertex (reflectance_at: the constant, or its bitmap texture at its.uv)
__device__ void bsdf_eval_pdf(const MaterialD &m, d3 R, d3 wi, d3 wo, int measure, d3 &f, Float &pdf)
{
    f = mk(0.0); pdf = 0.0;
    if (m.twoSided && !(wi.z > 0)) { wi.z = -wi.z; wo.z = -wo.z; }      // TwoSided::eval/pdf, twosided.cpp:100-124
    if (m.type == 3) {                                                   // SmoothDielectric::eval/pdf, dielectric.cpp:227-275 (ERadiance)
        Float cosThetaT;
        const Float F = fresnelDielectricExt(wi.z, cosThetaT, m.eta.x);
        if (measure != MEASURE_DISCRETE) return;
        if (wi.z * wo.z >= 0) {
            if (fabs(dot(mk(-wi.x, -wi.y, wi.z), wo) - 1) > GD_DELTA_EPSILON) return;
            f = R * F; pdf = F;
        } else {
            if (fabs(dot(dielectric_refract(m, wi, cosThetaT), wo) - 1) > GD_DELTA_EPSILON) return;
            const Float factor = cosThetaT < 0 ? 1 / m.eta.x : m.eta.x;
            f = m.k * factor * factor * (1 - F); pdf = 1 - F;
        }
        return;
    }
    if (wi.z <= 0 || wo.z <= 0) return;
    if (m.type == 0) {
        if (measure != MEASURE_SOLID_ANGLE) return;
        f = R * (GD_INV_PI * wo.z);
        pdf = GD_INV_PI * wo.z;
    } else if (m.type == 1) {
        if (measure != MEASURE_DISCRETE || fabs(dot(mk(-wi.x, -wi.y, wi.z), wo) - 1) > GD_DELTA_EPSILON) return;
        f = R * fresnelConductorExact(wi.z, m.eta, m.k);
        pdf = 1.0;
    } else {
        if (measure != MEASURE_SOLID_ANGLE) return;
        const d3 H = normalize(wo + wi);
        const Mf d = mf_of(m);
        const Float D = mf_eval(d, H);
        const Float G1i = mf_G1(d, wi, H);
        if (d.sv) pdf = D * G1i / (4.0 * wi.z);
        else pdf = (D * H.z) / (4 * fabs(dot(wo, H)));
        if (D == 0) return;
        const d3 F = fresnelConductorExact(dot(wi, H), m.eta, m.k) * R;
        const Float G = G1i * mf_G1(d, wo, H);
        const Float model = D * G / (4.0 * wi.z);
        f = F * model;
    }
}

struct BSDFSample { d3 wo, weight; Float pdf, eta; int sampledType; };
// the pdf-returning BSDF::sample overloads: diffuse.cpp:141-151, conductor.cpp:256-273, roughconductor.cpp:369-418
__device__ void bsdf_sample_one(const MaterialD &m, d3 R, d3 wi, Float sx, Float sy, BSDFSample &r)
{
    r.wo = mk(0.0); r.weight = mk(0.0); r.pdf = 0.0; r.eta = 1.0; r.sampledType = 0;   // gpt.cpp:450-454: pdf starts at 0
    if (m.type == 3) {                                                    // SmoothDielectric::sample, dielectric.cpp:277-305
        Float cosThetaT;
        const Float F = fresnelDielectricExt(wi.z, cosThetaT, m.eta.x);
        if (sx <= F) {
            r.sampledType = EDeltaReflection;
            r.wo = mk(-wi.x, -wi.y, wi.z);
            r.pdf = F;
            r.weight = R;
        } else {
            r.sampledType = EDeltaTransmission;
            r.wo = dielectric_refract(m, wi, cosThetaT);
            r.eta = cosThetaT < 0 ? m.eta.x : 1 / m.eta.x;
            r.pdf = 1 - F;
            const Float factor = cosThetaT < 0 ? 1 / m.eta.x : m.eta.x;
            r.weight = m.k * (factor * factor);
        }
        return;
    }
    if (m.type == 0) {
        if (wi.z <= 0) return;
        r.wo = squareToCosineHemisphere(sx, sy);
        r.sampledType = EDiffuseReflection;
        r.pdf = GD_INV_PI * r.wo.z;
        r.weight = R;
    } else if (m.type == 1) {
        if (wi.z <= 0) return;
        r.sampledType = EDeltaReflection;
        r.wo = mk(-wi.x, -wi.y, wi.z);
        r.pdf = 1;
        r.weight = R * fresnelConductorExact(wi.z, m.eta, m.k);
    } else {
        if (wi.z < 0) return;
        const Mf d = mf_of(m);
        Float temporaryPdf = 0;
        const d3 mm = mf_sample(d, wi, sx, sy, temporaryPdf);
        if (temporaryPdf == 0) return;
        r.wo = 2 * dot(wi, mm) * mm - wi;
        r.sampledType = EGlossyReflection;
        if (r.wo.z <= 0) return;
        const d3 F = fresnelConductorExact(dot(wi, mm), m.eta, m.k) * R;
        Float weight;
        if (d.sv) weight = mf_G1(d, r.wo, mm);
        else weight = mf_eval(d, mm) * (mf_G1(d, wi, mm) * mf_G1(d, r.wo, mm)) * dot(wi, mm) / (temporaryPdf * wi.z);
        if (weight > 0) {
            r.pdf = temporaryPdf / (4.0 * dot(r.wo, mm));
            r.weight = F * weight;
        }
    }
}

__device__ __forceinline__ void bsdf_sample(const MaterialD &m, d3 R, d3 wi, Float sx, Float sy, BSDFSample &r)
{ // TwoSided::sample (pdf overload), twosided.cpp:148-168, around the one-sided models
    const bool flipped = m.twoSided && wi.z < 0;
    if (flipped) wi.z = -wi.z;
    bsdf_sample_one(m, R, wi, sx, sy, r);
    if (flipped && !(r.weight.x == 0 && r.weight.y == 0 && r.weight.z == 0) && r.pdf != 0) r.wo.z = -r.wo.z;
}

// getVertexType (gpt.cpp:176-231) for single-component BSDFs; getRoughness: diffuse.cpp:167, conductor.cpp:275, roughconductor.cpp:437
__device__ __forceinline__ bool vertex_is_diffuse(const MaterialD &m, const ConfigD &cfg, int bsdfTypeMask)
{
    const Float r = m.type == 0 ? GD_INF : ((m.type == 1 || m.type == 3) ? 0.0 : 0.5 * (m.alphaU + m.alphaV));
    Float lowest = GD_INF;
    bool found_smooth = false, found_dirac = false, skip = false;
    if (r == 0) { found_dirac = true; if (!(bsdfTypeMask & EDelta)) skip = true; }
    else found_smooth = true;
    if (!skip && r < lowest) lowest = r;
    if (!found_smooth && found_dirac && !(bsdfTypeMask & EDelta)) lowest = 0;
    return !(lowest <= cfg.shiftThreshold);
}

// ---- emitters -----------------------------------------------------------------------------------------------
struct DRec { d3 ref, refN, p, n, d; Float dist, pdf; int object; int offSurfaceDiscrete; };   // last: a `point` emitter was sampled (measure EDiscrete, not EOnSurface)

__device__ __forceinline__ int cdf_sample(const Float *cdf, int n /*entries = n+1*/, Float v)
{ // DiscreteDistribution::sample, pmf.h:110-123: lower_bound, step back one, skip zero-probability entries
    int lo = 0, hi = n + 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] < v) lo = mid + 1; else hi = mid; }
    int index = lo - 1;
    if (index < 0) index = 0;
    if (index > n - 1) index = n - 1;
    while ((cdf[index + 1] - cdf[index]) == 0 && index < n) ++index;
    return index;
}

// Scene::sampleEmitterDirectVisible (scene.cpp:855-879) minus the shadow ray, which the caller casts:
// AreaLight::sampleDirect (area.cpp:158-172) -> Shape::sampleDirect (shape.cpp:102-116) -> TriMesh::samplePosition
// (trimesh.cpp:412-423) -> Triangle::sample (triangle.cpp:24-).  Returns value (already / emPdf); dRec.pdf includes emPdf.
// ---- the `constant` environment emitter (src/emitters/constant.cpp) ------------------------------------------------
__device__ __forceinline__ bool solve_quadratic(Float a, Float b, Float c, Float &x0, Float &x1)
{ // util.cpp:447-485
    if (a == 0) {
        if (b != 0) { x0 = x1 = -c / b; return true; }
        return false;
    }
    const Float discrim = b * b - 4.0 * a * c;
    if (discrim < 0) return false;
    const Float sqrtDiscrim = sqrt(discrim);
    const Float temp = (b < 0) ? -0.5 * (b - sqrtDiscrim) : -0.5 * (b + sqrtDiscrim);
    x0 = temp / a;
    x1 = c / temp;
    if (x0 > x1) { const Float t = x0; x0 = x1; x1 = t; }
    return true;
}
__device__ __forceinline__ bool bsphere_hit(const SceneD &S, d3 ro, d3 rd, Float &nearT, Float &farT)
{ // bsphere.h:88-95
    const d3 o = ro - S.bsCenter;
    return solve_quadratic(len2(rd), 2 * dot(o, rd), len2(o) - S.bsRadius * S.bsRadius, nearT, farT);
}
__device__ __forceinline__ bool is_zero(d3 v) { return v.x == 0 && v.y == 0 && v.z == 0; }
#ifndef GDPT_X_MASK
#define GDPT_X_MASK 0
#endif
#define GDPT_HAS_ENVMAP_N(S, n) (!((GDPT_X_MASK >> (n)) & 1) && (S).hasEnvMap)
// EnvironmentMap (defined after the texture lookups they use)
__device__ d3 envmap_eval_call(const EnvMapD &e, d3 d, bool hasDifferentials, d3 rxD, d3 ryD);
template <bool INL> __device__ __forceinline__ d3 envmap_eval(const EnvMapD &e, d3 d, bool hasDifferentials, d3 rxD, d3 ryD);
__device__ void envmap_sample_direction(const EnvMapD &e, Float sx, Float sy, d3 &d, d3 &value, Float &pdf);
__device__ Float envmap_pdf_direction(const EnvMapD &e, d3 dLocal);
__device__ __forceinline__ d3 mul3(const Float *M, d3 v) { return mk(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z); }
// Scene::evalEnvironment(ray) for a ray WITHOUT differentials (every ray but the camera's): the constant, or the map's level 0
template <bool INL>
__device__ __forceinline__ d3 env_radiance(const SceneD &S, const SceneView &V, d3 d)
{
    if (GDPT_HAS_ENVMAP_N(S, 0)) return envmap_eval<INL>(*S.envMap, d, false, mk(0.0), mk(0.0));
    return V.emitters[S.envIndex].radiance;
}
// ConstantBackgroundEmitter::fillDirectSamplingRecord, constant.cpp:245-261
__device__ __forceinline__ bool env_fill_drec(const SceneD &S, DRec &dRec, d3 o, d3 d)
{
    Float nearT, farT;
    if (!bsphere_hit(S, o, d, nearT, farT) || nearT > 0 || farT < 0) return false;
    dRec.p = o + d * farT;
    dRec.n = normalize(S.bsCenter - dRec.p);
    dRec.object = S.envIndex;
    dRec.d = d;
    dRec.dist = farT;
    return true;
}
// ConstantBackgroundEmitter::sampleDirect, constant.cpp:179-219
__device__ __forceinline__ d3 env_sample_direct(const SceneD &S, d3 radiance, DRec &dRec, Float sx, Float sy)
{
    if (GDPT_HAS_ENVMAP_N(S, 1)) {                                       // EnvironmentMap::sampleDirect, envmap.cpp:509-534
        d3 value, dl;
        Float pdf, nearT, farT;
        envmap_sample_direction(*S.envMap, sx, sy, dl, value, pdf);
        const d3 dw = mul3(S.envMap->toWorld, dl);
        dRec.d = dw; dRec.dist = 0.0; dRec.p = dRec.ref; dRec.n = mk(0.0);
        if (is_zero(value) || pdf == 0 || !bsphere_hit(S, dRec.ref, dw, nearT, farT) || nearT >= 0 || farT <= 0) { dRec.pdf = 0.0; return mk(0.0); }
        dRec.pdf = pdf;
        dRec.p = dRec.ref + dw * farT;
        dRec.n = normalize(S.bsCenter - dRec.p);
        dRec.dist = farT;
        return value / pdf;
    }
    d3 d;
    Float pdf;
    const bool hasN = !is_zero(dRec.refN);
    if (hasN) {
        d = squareToCosineHemisphere(sx, sy);
        pdf = GD_INV_PI * d.z;
        Frame3 f;                                                    // Frame(n): coordinateSystem, util.cpp:592-601
        f.n = dRec.refN;
        if (fabs(f.n.x) > fabs(f.n.y)) { const Float il = 1.0 / sqrt(f.n.x * f.n.x + f.n.z * f.n.z); f.t = mk(f.n.z * il, 0.0, -f.n.x * il); }
        else { const Float il = 1.0 / sqrt(f.n.y * f.n.y + f.n.z * f.n.z); f.t = mk(0.0, f.n.z * il, -f.n.y * il); }
        f.s = cross(f.t, f.n);
        d = toWorld(f, d);
    } else {
        const Float z = 1.0 - 2.0 * sy, r = safe_sqrt(1.0 - z * z), phi = 2.0 * GD_PI * sx;    // warp.cpp:25-31
        d = mk(r * cos(phi), r * sin(phi), z);
        pdf = 1.0 / (4.0 * GD_PI);
    }
    Float nearT, farT;
    dRec.pdf = 0.0;
    dRec.d = d; dRec.dist = 0.0; dRec.p = dRec.ref; dRec.n = mk(0.0);
    if (!bsphere_hit(S, dRec.ref, d, nearT, farT)) return mk(0.0);
    if (!(nearT < 0 && farT > 0)) return mk(0.0);
    dRec.p = dRec.ref + d * farT;
    dRec.n = normalize(S.bsCenter - dRec.p);
    dRec.dist = farT;
    dRec.pdf = pdf;
    if (hasN && dot(dRec.d, dRec.refN) <= 0) return mk(0.0);
    return radiance / pdf;
}

template <bool ENV>
__device__ d3 sample_emitter_direct(const SceneD &S, const SceneView &V, DRec &dRec, Float sx, Float sy)
{
    const int index = cdf_sample(V.emitterCdf, S.numEmitters, sx);
    const Float emPdf = V.emitterCdf[index + 1] - V.emitterCdf[index];
    sx = (sx - V.emitterCdf[index]) / (V.emitterCdf[index + 1] - V.emitterCdf[index]);
    const EmitterD em = V.emitters[index];
    d3 value;
    dRec.offSurfaceDiscrete = 0;
    if (ENV && em.numTris == 0) {
        value = env_sample_direct(S, em.radiance, dRec, sx, sy);
    } else if (ENV && em.numTris < 0) {                       // PointEmitter::sampleDirect, point.cpp:120-134
        dRec.p = em.position;
        dRec.pdf = 1.0;
        dRec.offSurfaceDiscrete = 1;
        dRec.d = dRec.p - dRec.ref;
        dRec.dist = len(dRec.d);
        const Float invDist = 1.0 / dRec.dist;
        dRec.d = dRec.d * invDist;
        dRec.n = mk(0.0);
        value = em.radiance * (invDist * invDist);
    } else if (em.rectangle) {                                // Rectangle::samplePosition, rectangle.cpp:200-206, then Shape::sampleDirect (shape.cpp:102-116)
        const Float lx = sx * 2 - 1, ly = sy * 2 - 1;
        dRec.p = mk(em.rect[0] * lx + em.rect[1] * ly + em.rect[2] * 0.0 + em.rect[3], em.rect[4] * lx + em.rect[5] * ly + em.rect[6] * 0.0 + em.rect[7],
                    em.rect[8] * lx + em.rect[9] * ly + em.rect[10] * 0.0 + em.rect[11]);
        dRec.n = em.rectN;
        dRec.pdf = em.invSurfaceArea;
        dRec.d = dRec.p - dRec.ref;
        const Float distSquared = len2(dRec.d);
        dRec.dist = sqrt(distSquared);
        dRec.d = dRec.d / dRec.dist;
        const Float dp = fabs(dot(dRec.d, dRec.n));
        dRec.pdf *= dp != 0 ? (distSquared / dp) : 0.0;
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0) value = em.radiance / dRec.pdf;
        else { dRec.pdf = 0.0; value = mk(0.0); }
    } else {
        const Float *cdf = V.emCdf + em.cdfOffset;
        const int ti = cdf_sample(cdf, em.numTris, sy);
        sy = (sy - cdf[ti]) / (cdf[ti + 1] - cdf[ti]);
        const EmTri tr = V.emTris[em.firstEmTri + ti];
        const Float a = safe_sqrt(1.0 - sx);                  // warp.cpp:76-79
        const Float bx = 1 - a, by = a * sy;
        const d3 sideA = tr.p1 - tr.p0, sideB = tr.p2 - tr.p0;
        dRec.p = tr.p0 + (sideA * bx) + (sideB * by);
        dRec.n = normalize(cross(sideA, sideB));
        dRec.pdf = em.invSurfaceArea;
        dRec.d = dRec.p - dRec.ref;
        const Float distSquared = len2(dRec.d);
        dRec.dist = sqrt(distSquared);
        dRec.d = dRec.d / dRec.dist;
        const Float dp = fabs(dot(dRec.d, dRec.n));
        dRec.pdf *= dp != 0 ? (distSquared / dp) : 0.0;
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0) value = em.radiance / dRec.pdf;
        else { dRec.pdf = 0.0; value = mk(0.0); }
    }
    dRec.object = index;
    dRec.pdf *= emPdf;
    value = value / emPdf;
    return value;
}
template <bool ENV>
__device__ __forceinline__ Float pdf_emitter_direct(const SceneD &S, const SceneView &V, int object, d3 d, d3 refN, d3 n, Float dist)
{
    Float pd = 0.0;
    if (ENV && S.envIndex >= 0 && object == S.envIndex) pd = GDPT_HAS_ENVMAP_N(S, 2) ? envmap_pdf_direction(*S.envMap, mul3(S.envMap->toLocal, d))                 // envmap.cpp:536-547
                                                                    : (is_zero(refN) ? 1.0 / (4.0 * GD_PI) : GD_INV_PI * fmax((Float)0.0, dot(d, refN)));   // constant.cpp:221-236
    else if (dot(d, refN) >= 0 && dot(d, n) < 0) pd = V.emitters[object].invSurfaceArea * (dist * dist) / fabs(dot(d, n));
    return pd * (1.0 * S.emitterNormalization);
}

// ---- sensor: perspective.cpp:271-298 with the composite of :150-156 written out (the crop window: sample position in the full film's [0, 1]^2 =
// (crop-relative position + crop offset) / full size, which is what steps 4+5 of m_cameraToSample undo) ----------------
// apx, apy: the aperture sample (gpt.cpp:1262-1264), read by the thinlens sensor only (thinlens.cpp:324-361: a point of the aperture disk,
// squareToUniformDiskConcentric * apertureRadius; the ray goes from there through the pixel's point on the focal plane)
__device__ __forceinline__ void camera_ray(const CameraD &c, Float px, Float py, Float apx, Float apy, d3 &o, d3 &d, Float &mint, Float &maxt)
{
    const Float sxn = (px + c.cropX) * c.invW, syn = (py + c.cropY) * c.invH;
    const d3 nearP = mk((1 - 2 * sxn) * c.nearClip * c.tanHalf, (1 - 2 * syn) / c.aspect * c.nearClip * c.tanHalf, c.nearClip);
    d3 dl, ol = mk(0.0);
    if (c.thinlens) {
        const Float r1 = 2.0 * apx - 1.0, r2 = 2.0 * apy - 1.0;                    // warp.cpp:81-102
        Float phi, r;
        if (r1 == 0 && r2 == 0) { r = phi = 0; }
        else if (r1 * r1 > r2 * r2) { r = r1; phi = (GD_PI / 4.0) * (r2 / r1); }
        else { r = r2; phi = (GD_PI / 2.0) - (r1 / r2) * (GD_PI / 4.0); }
        ol = mk(r * cos(phi) * c.apertureRadius, r * sin(phi) * c.apertureRadius, 0.0);
        const Float fDist = c.focusDistance / nearP.z;
        dl = normalize(nearP * fDist - ol);
    } else dl = normalize(nearP);
    const Float invZ = 1.0 / dl.z;
    mint = c.nearClip * invZ;
    maxt = c.farClip * invZ;
    o = mk(c.m[0] * ol.x + c.m[1] * ol.y + c.m[2] * ol.z + c.m[3], c.m[4] * ol.x + c.m[5] * ol.y + c.m[6] * ol.z + c.m[7], c.m[8] * ol.x + c.m[9] * ol.y + c.m[10] * ol.z + c.m[11]);
    d = mk(c.m[0] * dl.x + c.m[1] * dl.y + c.m[2] * dl.z, c.m[4] * dl.x + c.m[5] * dl.y + c.m[6] * dl.z, c.m[8] * dl.x + c.m[9] * dl.y + c.m[10] * dl.z);
}

// ---- shifts -------------------------------------------------------------------------------------------------
// refract(wi, n, eta), util.cpp:774-792: the zero vector on total internal reflection
__device__ __forceinline__ d3 refract_dir(d3 wi, d3 n, Float eta)
{
    if (eta == 1) return -wi;
    const Float cosThetaI = dot(wi, n);
    if (cosThetaI > 0) eta = 1 / eta;
    const Float cosThetaTSqr = 1 - (1 - cosThetaI * cosThetaI) * (eta * eta);
    if (cosThetaTSqr <= 0.0) return mk(0.0);
    return n * (cosThetaI * eta - signum(cosThetaI) * sqrt(cosThetaTSqr)) - wi * eta;
}

// halfVectorShift, gpt.cpp:242-305
__device__ __forceinline__ bool half_vector_shift(d3 mainWi, d3 mainWo, d3 shiftedWi, Float mainEta, Float shiftedEta, Float &jacobian, d3 &wo)
{
    if (mainWi.z * mainWo.z < 0) {                    // refraction, :245-290
        if (mainEta == 1 || shiftedEta == 1) return false;
        const d3 hMain = mainWi.z < 0 ? -(mainWi * mainEta + mainWo) : -(mainWi + mainWo * mainEta);
        const d3 h = normalize(hMain);
        wo = refract_dir(shiftedWi, h, shiftedEta);
        if (wo.x == 0 && wo.y == 0 && wo.z == 0) return false;
        const d3 hShifted = shiftedWi.z < 0 ? -(shiftedWi * shiftedEta + wo) : -(shiftedWi + wo * shiftedEta);
        const Float hLengthSquared = len2(hShifted) / (GD_D_EPSILON + len2(hMain));
        const Float WoDotH = fabs(dot(mainWo, h)) / (GD_D_EPSILON + fabs(dot(wo, h)));
        jacobian = hLengthSquared * WoDotH;
        return true;
    }
    const d3 h = normalize(mainWi + mainWo);          // reflection, :291-302
    wo = 2 * dot(shiftedWi, h) * h - shiftedWi;       // reflect(), util.cpp:763
    jacobian = fabs(dot(wo, h) / dot(mainWo, h));
    return true;
}

// ---- per-lane path state ------------------------------------------------------------------------------------
enum { RAY_NOT_CONNECTED = 0, RAY_RECENTLY_CONNECTED = 1, RAY_CONNECTED = 2 };   // gpt.cpp:127-131

struct Vertex {             // the part of Mitsuba's Intersection the path keeps; wi = toLocal(frame(prim), -rayD) is recomputed
    d3 p;                   // position
    int prim;               // leaf-order triangle, -1 = invalid
    Float u, v;             // barycentrics of the hit (only kept live in builds for scenes with vertex normals)
};
struct Offset {             // RayState of an offset path, gpt.cpp:135-173 (its radiance/gradient sums live in the Acc)
    d3 throughput;
    Float pdf;
    Vertex v;
    d3 rayD;                // direction of the ray that arrived at v
    int alive, status;
};

__device__ __forceinline__ Frame3 frame_of(const TriShade &t) { Frame3 f; f.s = t.s; f.t = t.t; f.n = t.n; return f; }

// Shading frame and geometric normal at a vertex (fillIntersectionRecord, skdtree.h:367-397,426).  Flat triangles: the constants of
// the triangle.  With per-vertex normals (SMOOTH builds): n = normalize(sum b_i n_i), the geometric normal is flipped to the side of
// n, and (s, t) come from computeShadingFrame(n, its.dpdu) (util.cpp:603-608); its.dpdu = p1 - p0, or the UV tangent of a textured mesh.
struct Shading { Frame3 fr; d3 geoN; };
template <bool SMOOTH>
__device__ __forceinline__ Shading shading_at(const SceneView &S, const Vertex &v)
{
    const TriShade &ts = S.shade[v.prim];
    Shading sh;
    sh.fr = frame_of(ts);
    sh.geoN = ts.n;
    if (SMOOTH && ts.smooth) {
        const TriNormals vn = S.vn[v.prim];
        const d3 b = mk(1 - v.u - v.v, v.u, v.v);
        const d3 n = normalize(vn.n0 * b.x + vn.n1 * b.y + vn.n2 * b.z);
        if (dot(ts.n, n) < 0) sh.geoN = -ts.n;
        const d3 dpdu = vn.dpdu;
        sh.fr.n = n;
        sh.fr.s = normalize(dpdu - n * dot(n, dpdu));
        sh.fr.t = cross(n, sh.fr.s);
    }
    return sh;
}

// fillIntersectionRecord<true>, skdtree.h:343-428: barycentric position (frame: shading_at; wi: local_wi)
__device__ __forceinline__ void fill_vertex(const SceneView &S, const Hit &h, d3 rayD, Vertex &v)
{
    v.prim = h.prim;
    if (h.prim < 0) return;
    const TriShade &ts = S.shade[h.prim];
    const d3 b = mk(1 - h.u - h.v, h.u, h.v);
    v.p = ts.p0 * b.x + ts.p1 * b.y + ts.p2 * b.z;
    v.u = h.u; v.v = h.v;
}

// MIPMap::evalTexel's boundary handling (mipmap.h:503-561): false = the lookup is the constant `c` (zero / one modes)
__device__ __forceinline__ bool tex_wrap(int &x, int size, int mode, Float &c)
{
    if (x >= 0 && x < size) return true;
    if (mode == 0) { x %= size; if (x < 0) x += size; return true; }                                   // math::modulo
    if (mode == 1) { x = x < 0 ? 0 : size - 1; return true; }
    if (mode == 2) { x %= 2 * size; if (x < 0) x += 2 * size; if (x >= size) x = 2 * size - x - 1; return true; }
    c = mode == 3 ? 0.0 : 1.0;
    return false;
}
// Real calls, and calls inside them: a kernel is charged the largest register count among its callees, and the 4-wave builds must stay at 128 --
// the lookups are therefore cut into pieces that each fit (tex_ewa ~ the size of one footprint loop), at the price of nested calls on a cold path.
#define GDPT_COLD_CALL __noinline__
__device__ __forceinline__ d3 tex_texel(const TexD &t, int level, int x, int y)
{ // evalTexel, mipmap.h:503-563
    Float c = 0;
    const int w = t.lw[level], h = t.lh[level];
    if (!tex_wrap(x, w, t.wrapU, c)) return mk(c);
    if (!tex_wrap(y, h, t.wrapV, c)) return mk(c);
    const Float *p = t.texels + ((size_t)t.loff[level] + (size_t)y * w + x) * 3;
    return mk(p[0], p[1], p[2]);
}
__device__ __forceinline__ d3 tex_box(const TexD &t, int level, Float u, Float v) { return tex_texel(t, level, (int)floor(u * t.lw[level]), (int)floor(v * t.lh[level])); }   // :566-569
__device__ __forceinline__ d3 tex_bilinear(const TexD &t, int level, Float u_, Float v_)
{ // evalBilinear, mipmap.h:575-596
    if (!is_finite_d(u_) || !is_finite_d(v_)) return mk(0.0);
    if (level >= t.levels) return tex_box(t, t.levels - 1, u_, v_);
    const Float u = u_ * t.lw[level] - 0.5, v = v_ * t.lh[level] - 0.5;
    const int xPos = (int)floor(u), yPos = (int)floor(v);
    const Float dx1 = u - xPos, dx2 = 1.0 - dx1, dy1 = v - yPos, dy2 = 1.0 - dy1;
    return tex_texel(t, level, xPos, yPos) * dx2 * dy2 + tex_texel(t, level, xPos, yPos + 1) * dx2 * dy1 + tex_texel(t, level, xPos + 1, yPos) * dx1 * dy2 + tex_texel(t, level, xPos + 1, yPos + 1) * dx1 * dy1;
}
__device__ __forceinline__ d3 tex_ewa(const TexD &t, int level, Float u_, Float v_, Float A, Float B, Float C)
{ // evalEWA, mipmap.h:744-833
    if (!is_finite_d(A + B + C + u_ + v_)) return mk(0.0);
    if (level >= t.levels) return tex_box(t, t.levels - 1, u_, v_);
    const Float u = u_ * t.lw[level] - 0.5, v = v_ * t.lh[level] - 0.5;
    const Float rx = t.ratioX[level], ry = t.ratioY[level];
    A /= rx * rx; B /= rx * ry; C /= ry * ry;
    const Float invDet = 1.0 / (-B * B + 4.0 * A * C), deltaU = 2.0 * sqrt(C * invDet), deltaV = 2.0 * sqrt(A * invDet);
    const int u0 = (int)ceil(u - deltaU), u1 = (int)floor(u + deltaU), v0 = (int)ceil(v - deltaV), v1 = (int)floor(v + deltaV);
    const Float As = A * TEX_LUT_SIZE, Bs = B * TEX_LUT_SIZE, Cs = C * TEX_LUT_SIZE;
    d3 result = mk(0.0);
    Float denominator = 0.0;
    const Float ddq = 2 * As, uu0 = (Float)u0 - u;
    for (int vt = v0; vt <= v1; ++vt) {
        const Float vv = (Float)vt - v;
        Float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv;
        Float dq = As * (2 * uu0 + 1) + Bs * vv;
        for (int ut = u0; ut <= u1; ++ut) {
            if (q < (Float)TEX_LUT_SIZE) {
                const uint32_t qi = (uint32_t)q;
                if (qi < (uint32_t)TEX_LUT_SIZE) {
                    const Float weight = t.lut[(int)q];
                    result = result + tex_texel(t, level, ut, vt) * weight;
                    denominator += weight;
                }
            }
            q += dq;
            dq += ddq;
        }
    }
    if (denominator == 0) return tex_bilinear(t, level, u_, v_);
    return result / denominator;
}
__device__ __forceinline__ Float tex_hypot2(Float a, Float b)
{ // math::hypot2, math.cpp:89-101
    Float r;
    if (fabs(a) > fabs(b)) { r = b / a; r = fabs(a) * sqrt(1.0 + r * r); }
    else if (b != 0.0) { r = a / b; r = fabs(b) * sqrt(1.0 + r * r); }
    else r = 0.0;
    return r;
}
__device__ __forceinline__ Float tex_log2(Float v) { const Float invLn2 = 1.0 / log(2.0); return log(v) * invLn2; }   // math::log2, math.cpp:108-111
// TMIPMap::eval(uv, d0, d1), mipmap.h:628-712 (trilinear / ewa; d0 = (dudx, dvdx), d1 = (dudy, dvdy), already scaled by uscale / vscale)
__device__ __forceinline__ d3 tex_filtered(const TexD &t, Float u, Float v, Float d0x, Float d0y, Float d1x, Float d1y)
{
    const Float du0 = d0x * t.w, dv0 = d0y * t.h, du1 = d1x * t.w, dv1 = d1y * t.h;
    Float A = dv0 * dv0 + dv1 * dv1, B = -2.0 * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25;
    const Float root = tex_hypot2(A - C, B), Aprime = 0.5 * (A + C - root), Cprime = 0.5 * (A + C + root);
    const Float majorRadius = Aprime != 0 ? sqrt(F / Aprime) : 0;
    Float minorRadius = Cprime != 0 ? sqrt(F / Cprime) : 0;
    if (t.filter == 2 || !(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
        const Float level = tex_log2(fmax(majorRadius, GD_EPSILON));
        const int ilevel = (int)floor(level);
        if (ilevel < 0) return tex_bilinear(t, 0, u, v);
        const Float a = level - ilevel;
        return tex_bilinear(t, ilevel, u, v) * (1.0 - a) + tex_bilinear(t, ilevel + 1, u, v) * a;
    }
    if (minorRadius * t.maxAnisotropy < majorRadius) {
        minorRadius = majorRadius / t.maxAnisotropy;
        const Float theta = 0.5 * atan(B / (A - C));
        const Float sinTheta = sin(theta), cosTheta = cos(theta);
        const Float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta,
                    sin2Theta = 2 * sinTheta * cosTheta;
        A = a2 * cosTheta2 + b2 * sinTheta2;
        B = (a2 - b2) * sin2Theta;
        C = a2 * sinTheta2 + b2 * cosTheta2;
        F = a2 * b2;
    }
    const Float scale = 1.0 / F;
    A *= scale; B *= scale; C *= scale;
    const Float level = fmax((Float)0.0, tex_log2(minorRadius));
    const int ilevel = (int)level;
    const Float a = level - ilevel;
    if (majorRadius < 1 || !(A > 0 && C > 0)) return tex_bilinear(t, ilevel, u, v);
    return tex_ewa(t, ilevel, u, v, A, B, C) * (1.0 - a) + tex_ewa(t, ilevel + 1, u, v, A, B, C) * a;
}
// Texture2D::eval(its) (texture.cpp:112-121) -> BitmapTexture::eval: level 0 by evalBox / evalBilinear (bitmap.cpp:431-452), or -- a hit with
// UV partials under filterType trilinear / ewa -- the filtered lookup (bitmap.cpp:486-499).  partials = (dudx, dudy, dvdx, dvdy).
__device__ __forceinline__ d3 tex_eval_impl(const TexD &t, Float u_, Float v_, bool hasPartials, Float dudx, Float dudy, Float dvdx, Float dvdy)   // a real call: textured vertices only; inlined at its six sites it cost every per-vertex build ~25 % (register pressure)
{
    const Float ux = u_ * t.uscale + t.uoffset, vy = v_ * t.vscale + t.voffset;
    d3 value;
    if (t.filter == 0) value = tex_box(t, 0, ux, vy);
    else if (t.filter == 1 || !hasPartials) value = tex_bilinear(t, 0, ux, vy);
    else value = tex_filtered(t, ux, vy, dudx * t.uscale, dvdx * t.vscale, dudy * t.uscale, dvdy * t.vscale);
    return value * t.scale;
}
__device__ GDPT_COLD_CALL d3 tex_eval_call(const TexD &t, Float u_, Float v_, bool hasPartials, Float dudx, Float dudy, Float dvdx, Float dvdy) { return tex_eval_impl(t, u_, v_, hasPartials, dudx, dudy, dvdx, dvdy); }
// INL: the 4-wave builds inline the lookup (a callee with the EWA loop in it takes ~150 registers of its own accord, a kernel is charged the
// largest count among its callees, and that would cost those builds a wave per SIMD); the 2-wave builds call it (inlined at their unrolled
// sites it cost them 25 %)
template <bool INL>
__device__ __forceinline__ d3 tex_eval(const TexD &t, Float u_, Float v_, bool hasPartials, Float dudx, Float dudy, Float dvdx, Float dvdy)
{
    if constexpr (INL) return tex_eval_impl(t, u_, v_, hasPartials, dudx, dudy, dvdx, dvdy);
    else return tex_eval_call(t, u_, v_, hasPartials, dudx, dudy, dvdx, dvdy);
}
// ---- EnvironmentMap, src/emitters/envmap.cpp ------------------------------------------------------------------------------------
__device__ __forceinline__ Float lum3(d3 c) { return c.x * (Float)0.212671f + c.y * (Float)0.715160f + c.z * (Float)0.072169f; }   // spectrum.h:725-727
// evalEnvironment, envmap.cpp:378-409
__device__ __forceinline__ d3 envmap_eval_impl(const EnvMapD &e, d3 d, bool hasDifferentials, d3 rxD, d3 ryD)
{
    const d3 v = mul3(e.toLocal, d);
    const Float uvx = atan2(v.x, -v.z) * GD_INV_TWOPI, uvy = acos(fmin(1.0, fmax(-1.0, v.y))) * GD_INV_PI;
    d3 value;
    if (!hasDifferentials) value = tex_bilinear(e.tex, 0, uvx, uvy);
    else {
        const d3 dvdx = mul3(e.toLocal, rxD) - v, dvdy = mul3(e.toLocal, ryD) - v;
        const Float t1 = GD_INV_TWOPI / (v.x * v.x + v.z * v.z), t2 = -GD_INV_PI / fmax(safe_sqrt(1.0 - v.y * v.y), GD_EPSILON);
        value = tex_filtered(e.tex, uvx, uvy, t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y, t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y);
    }
    return value * e.scale;
}
__device__ GDPT_COLD_CALL d3 envmap_eval_call(const EnvMapD &e, d3 d, bool hasDifferentials, d3 rxD, d3 ryD) { return envmap_eval_impl(e, d, hasDifferentials, rxD, ryD); }
template <bool INL>
__device__ __forceinline__ d3 envmap_eval(const EnvMapD &e, d3 d, bool hasDifferentials, d3 rxD, d3 ryD)
{
    if constexpr (INL) return envmap_eval_impl(e, d, hasDifferentials, rxD, ryD);
    else return envmap_eval_call(e, d, hasDifferentials, rxD, ryD);
}
__device__ __forceinline__ int envmap_sample_reuse(const float *cdf, int size, Float &sample)
{ // envmap.cpp:640-645: std::lower_bound over size + 1 floats for (float) sample
    const float key = (float)sample;
    int lo = 0, n = size + 1;
    while (n > 0) { const int half = n >> 1; if (cdf[lo + half] < key) { lo += half + 1; n -= half + 1; } else n = half; }
    int index = lo - 1;
    if (index < 0) index = 0;
    if (index > size - 1) index = size - 1;
    sample = (sample - (Float)cdf[index]) / (Float)(cdf[index + 1] - cdf[index]);
    return index;
}
__device__ __forceinline__ Float interval_to_tent(Float sample)
{ // warp.cpp:143-155
    Float sign;
    if (sample < 0.5) { sign = 1; sample *= 2; } else { sign = -1; sample = 2 * (sample - 0.5); }
    return sign * (1 - sqrt(sample));
}
// internalSampleDirection, envmap.cpp:556-594
__device__ GDPT_COLD_CALL void envmap_sample_direction(const EnvMapD &e, Float sx, Float sy, d3 &d, d3 &value, Float &pdf)
{
    const int w = e.tex.w, h = e.tex.h;
    const int row = envmap_sample_reuse(e.cdfRows, h, sy), col = envmap_sample_reuse(e.cdfCols + (size_t)row * (w + 1), w, sx);
    const Float posx = (Float)col + interval_to_tent(sx), posy = (Float)row + interval_to_tent(sy);
    const int xPos = (int)floor(posx), yPos = (int)floor(posy);
    const Float dx1 = posx - xPos, dx2 = 1.0 - dx1, dy1 = posy - yPos, dy2 = 1.0 - dy1;
    const d3 value1 = tex_texel(e.tex, 0, xPos, yPos) * dx2 * dy2 + tex_texel(e.tex, 0, xPos + 1, yPos) * dx1 * dy2;
    const d3 value2 = tex_texel(e.tex, 0, xPos, yPos + 1) * dx2 * dy1 + tex_texel(e.tex, 0, xPos + 1, yPos + 1) * dx1 * dy1;
    value = (value1 + value2) * e.scale;
    pdf = (lum3(value1) * e.rowWeights[min(max(yPos, 0), h - 1)] + lum3(value2) * e.rowWeights[min(max(yPos + 1, 0), h - 1)]) * e.normalization;
    const Float phi = e.pixelSizeX * (posx + 0.5), theta = e.pixelSizeY * (posy + 0.5);
    const Float sinPhi = sin(phi), cosPhi = cos(phi), sinTheta = sin(theta), cosTheta = cos(theta);
    d = mk(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    pdf /= fmax(fabs(sinTheta), GD_EPSILON);
}
// internalPdfDirection, envmap.cpp:597-625
__device__ GDPT_COLD_CALL Float envmap_pdf_direction(const EnvMapD &e, d3 d)
{
    const int w = e.tex.w, h = e.tex.h;
    const Float uvx = atan2(d.x, -d.z) * GD_INV_TWOPI, uvy = acos(fmin(1.0, fmax(-1.0, d.y))) * GD_INV_PI;
    if (!is_finite_d(uvx) || !is_finite_d(uvy)) return 0.0;
    const Float u = uvx * w - 0.5, v = uvy * h - 0.5;
    const int xPos = (int)floor(u), yPos = (int)floor(v);
    const Float dx1 = u - xPos, dx2 = 1.0 - dx1, dy1 = v - yPos, dy2 = 1.0 - dy1;
    const d3 value1 = tex_texel(e.tex, 0, xPos, yPos) * dx2 * dy2 + tex_texel(e.tex, 0, xPos + 1, yPos) * dx1 * dy2;
    const d3 value2 = tex_texel(e.tex, 0, xPos, yPos + 1) * dx2 * dy1 + tex_texel(e.tex, 0, xPos + 1, yPos + 1) * dx1 * dy1;
    const Float sinTheta = safe_sqrt(1 - d.y * d.y);
    return (lum3(value1) * e.rowWeights[min(max(yPos, 0), h - 1)] + lum3(value2) * e.rowWeights[min(max(yPos + 1, 0), h - 1)]) * e.normalization / fmax(fabs(sinTheta), GD_EPSILON);
}
// the two differential directions of the camera ray through film position (sxp, syp): trafo(normalize(nearP + m_dx / m_dy)), perspective.cpp:291-295
__device__ __forceinline__ void camera_differentials(const CameraD &c, Float sxp, Float syp, d3 &rxD, d3 &ryD)
{
    const Float sxn = (sxp + c.cropX) * c.invW, syn = (syp + c.cropY) * c.invH;
    const d3 nearP = mk((1 - 2 * sxn) * c.nearClip * c.tanHalf, (1 - 2 * syn) / c.aspect * c.nearClip * c.tanHalf, c.nearClip);
    const d3 mdx = mk(-2 * c.invW * c.nearClip * c.tanHalf, 0.0, 0.0), mdy = mk(0.0, -2 * c.invH / c.aspect * c.nearClip * c.tanHalf, 0.0);
    const d3 lx = normalize(nearP + mdx), ly = normalize(nearP + mdy);
    rxD = mk(c.m[0] * lx.x + c.m[1] * lx.y + c.m[2] * lx.z, c.m[4] * lx.x + c.m[5] * lx.y + c.m[6] * lx.z, c.m[8] * lx.x + c.m[9] * lx.y + c.m[10] * lx.z);
    ryD = mk(c.m[0] * ly.x + c.m[1] * ly.y + c.m[2] * ly.z, c.m[4] * ly.x + c.m[5] * ly.y + c.m[6] * ly.z, c.m[8] * ly.x + c.m[9] * ly.y + c.m[10] * ly.z);
}
// its.dpdu / its.dpdv of a hit on triangle `prim`: the edges, or the UV tangents of a mesh with texture coordinates (skdtree.h:373-380, trimesh.cpp:701-735)
__device__ __forceinline__ void tri_partials(const SceneView &S, int prim, d3 &dpdu, d3 &dpdv)
{
    const TriShade &ts = S.shade[prim];
    const d3 dP1 = ts.p1 - ts.p0, dP2 = ts.p2 - ts.p0;
    dpdu = dP1; dpdv = dP2;
    if (S.uv && S.hasUV[prim]) {
        const TriUV q = S.uv[prim];
        const Float dU1x = q.uv[2] - q.uv[0], dU1y = q.uv[3] - q.uv[1], dU2x = q.uv[4] - q.uv[0], dU2y = q.uv[5] - q.uv[1];
        const d3 n = cross(dP1, dP2);
        const Float nlen = len(n), determinant = dU1x * dU2y - dU1y * dU2x;
        if (nlen != 0) {
            if (determinant == 0) {
                const d3 a = n * (1.0 / nlen);
                if (fabs(a.x) > fabs(a.y)) { const Float il = 1.0 / sqrt(a.x * a.x + a.z * a.z); dpdv = mk(a.z * il, 0.0, -a.x * il); }
                else { const Float il = 1.0 / sqrt(a.y * a.y + a.z * a.z); dpdv = mk(0.0, a.z * il, -a.y * il); }
                dpdu = cross(dpdv, a);
            } else {
                const Float invDet = 1.0 / determinant;
                dpdu = (dP1 * dU2y - dP2 * dU1y) * invDet;
                dpdv = (dP1 * (-dU2x) + dP2 * dU1x) * invDet;
            }
        }
    }
}
// The hit of a CAMERA ray on a textured material: Intersection::getBSDF(ray) runs computePartials first (shape.h; intersection.cpp:5-78:
// the texture coordinates' change per pixel step, from the two differential rays of perspective.cpp:291-295 and the triangle's dpdu /
// dpdv), and the lookup is the filtered one.  (sxp, syp) = the film position of this path's camera ray.  A real call like tex_eval.
template <bool INL>
__device__ __forceinline__ d3 tex_eval_primary_impl(const SceneView &S, const CameraD &c, const TexD &t, const Vertex &v, d3 geoN, Float tu, Float tv, Float sxp, Float syp)
{
    // differential directions (origins = the camera position)
    d3 rxD, ryD;
    camera_differentials(c, sxp, syp, rxD, ryD);
    const d3 o = mk(c.m[3], c.m[7], c.m[11]);
    d3 dpdu, dpdv;
    tri_partials(S, v.prim, dpdu, dpdv);
    Float dudx = 0, dudy = 0, dvdx = 0, dvdy = 0;
    const Float pp = dot(geoN, v.p), po = dot(geoN, o), prx = dot(geoN, rxD), pry = dot(geoN, ryD);
    if (!(is_zero(dpdu) && is_zero(dpdv)) && !(prx == 0 || pry == 0)) {
        const Float tx = (pp - po) / prx, ty = (pp - po) / pry;
        const Float absX = fabs(geoN.x), absY = fabs(geoN.y), absZ = fabs(geoN.z);
        int a0, a1;
        if (absX > absY && absX > absZ) { a0 = 1; a1 = 2; }
        else if (absY > absZ) { a0 = 0; a1 = 2; }
        else { a0 = 0; a1 = 1; }
        const Float A00 = comp(dpdu, a0), A01 = comp(dpdv, a0), A10 = comp(dpdu, a1), A11 = comp(dpdv, a1);
        const d3 px = o + rxD * tx, py = o + ryD * ty;
        const Float Bx0 = comp(px, a0) - comp(v.p, a0), Bx1 = comp(px, a1) - comp(v.p, a1), By0 = comp(py, a0) - comp(v.p, a0), By1 = comp(py, a1) - comp(v.p, a1);
        const Float det = A00 * A11 - A01 * A10;                                         // solveLinearSystem2x2, util.cpp:527-539
        if (fabs(det) <= 0x1p-1024) { dudx = 1; dvdx = 0; dudy = 1; dvdy = 0; }          // (:66-76; the reference leaves dvdy unset there: taken as 0)
        else {
            const Float inverse = 1.0 / det;
            dudx = (A11 * Bx0 - A01 * Bx1) * inverse; dvdx = (A00 * Bx1 - A10 * Bx0) * inverse;
            dudy = (A11 * By0 - A01 * By1) * inverse; dvdy = (A00 * By1 - A10 * By0) * inverse;
        }
    }
    return tex_eval<INL>(t, tu, tv, true, dudx, dudy, dvdx, dvdy);
}
__device__ GDPT_COLD_CALL d3 tex_eval_primary_call(const SceneView &S, const CameraD &c, const TexD &t, const Vertex &v, d3 geoN, Float tu, Float tv, Float sxp, Float syp) { return tex_eval_primary_impl<false>(S, c, t, v, geoN, tu, tv, sxp, syp); }
// m_reflectance->eval(its) / m_specularReflectance->eval(its): the constant, or the bitmap at its.uv (skdtree.h:398-405: interpolated
// texture coordinates, or the barycentrics (b1, b2) for a mesh without any).  PERVERTEX builds only; flat untextured scenes compile it out.
// primary: v is the hit of the camera ray through film position (sxp, syp) -- the only hits that have UV partials.
template <bool PERVERTEX, bool INL = false>
__device__ __forceinline__ d3 reflectance_at(const SceneView &S, const MaterialD &m, const Vertex &v, bool primary = false, const CameraD *cam = nullptr, Float sxp = 0, Float syp = 0)
{
    if (!PERVERTEX || m.tex < 0) return m.reflectance;
    Float tu = v.u, tv = v.v;
    if (S.uv && S.hasUV[v.prim]) {
        const TriUV t = S.uv[v.prim];
        const Float b0 = 1 - v.u - v.v;
        tu = t.uv[0] * b0 + t.uv[2] * v.u + t.uv[4] * v.v;
        tv = t.uv[1] * b0 + t.uv[3] * v.u + t.uv[5] * v.v;
    }
    const TexD &t = S.tex[m.tex];
    if (primary && t.filter >= 2) {
        if constexpr (INL) return tex_eval_primary_impl<true>(S, *cam, t, v, shading_at<PERVERTEX>(S, v).geoN, tu, tv, sxp, syp);
        else return tex_eval_primary_call(S, *cam, t, v, shading_at<PERVERTEX>(S, v).geoN, tu, tv, sxp, syp);
    }
    return tex_eval<INL>(t, tu, tv, false, 0.0, 0.0, 0.0, 0.0);
}

// its.wi = its.toLocal(-ray.d), skdtree.h:427
template <bool SMOOTH>
__device__ __forceinline__ d3 local_wi(const SceneView &S, const Vertex &v, d3 rayD) { return toLocal(shading_at<SMOOTH>(S, v).fr, -rayD); }

// AreaLight::eval via Intersection::Le, area.cpp:104-109
__device__ __forceinline__ d3 emitted(const SceneView &S, int prim, d3 d)
{
    const TriShade &ts = S.shade[prim];
    if (ts.emitter < 0 || dot(ts.n, d) <= 0) return mk(0.0);
    return S.emitters[ts.emitter].radiance;
}

// ---- per-sample sums (evaluatePoint outputs): T(3) veryDirect(3) neighbour throughput[4](12) gradient[4](12) -----------------
// Either 30 registers per lane, or -- when the block's LDS budget allows (small scenes) -- an LDS slab [k][lane], which takes 60
// VGPRs of long-lived state out of the register allocator's way.
enum { ACC_T = 0, ACC_VD = 3, ACC_NBR = 6, ACC_GRAD = 18, ACC_N = 30 };
template <bool IN_LDS> struct Acc;
template <> struct Acc<false> {
    Float a[ACC_N];
    __device__ __forceinline__ void zero() { for (int k = 0; k < ACC_N; k++) a[k] = 0.0; }
    __device__ __forceinline__ void add3(int k, d3 v) { a[k] += v.x; a[k + 1] += v.y; a[k + 2] += v.z; }
    __device__ __forceinline__ d3 get3(int k) const { return mk(a[k], a[k + 1], a[k + 2]); }
    __device__ __forceinline__ Float get(int k) const { return a[k]; }
    __device__ __forceinline__ void set(int k, Float v) { a[k] = v; }
};
template <> struct Acc<true> {
    Float *p;               // LDS, already offset by the lane
    __device__ __forceinline__ void zero() { for (int k = 0; k < ACC_N; k++) p[k * TBLK] = 0.0; }
    __device__ __forceinline__ void add3(int k, d3 v) { p[k * TBLK] += v.x; p[(k + 1) * TBLK] += v.y; p[(k + 2) * TBLK] += v.z; }
    __device__ __forceinline__ d3 get3(int k) const { return mk(p[k * TBLK], p[(k + 1) * TBLK], p[(k + 2) * TBLK]); }
    __device__ __forceinline__ Float get(int k) const { return p[k * TBLK]; }
    __device__ __forceinline__ void set(int k, Float v) { p[k * TBLK] = v; }
};

// ---- film ---------------------------------------------------------------------------------------------------
// record components: 0 count | 1..3 T | 4..6 veryDirect | 7+3d+c neighbour throughput d | 19+3d+c gradient d, d = R,B,L,T
struct FilterD { Float radius, scale, c; };     // box.cpp:38, rfilter.cpp:37-55: every in-range table entry == c = 1/(2r)
__device__ __forceinline__ FilterD box_filter()
{
    FilterD f;
    f.radius = 0.5 + (Float)1e-5f;
    f.scale = 31 / f.radius;
    Float sum = 0;
    for (int i = 0; i < 31; i++) sum += 1.0;
    sum *= 2 * f.radius / 31;
    f.c = 1.0 * (1.0 / sum);
    return f;
}
__device__ __forceinline__ Float eval_discretized(const FilterD &f, Float x)
{ // rfilter.h:76-77
    int idx = (int)fabs(x * f.scale);
    if (idx > 31) idx = 31;
    return idx < 31 ? f.c : 0.0;
}

// The validity check of ImageBlock::put (imageblock.h:154-158; the blocks of a GPTWorkResult are created with warn = true and only
// dx / dy allow negative values, gpt_wr.cpp:38-42): a put with a non-finite channel, or a negative one where that is not allowed, is
// dropped whole -- value AND weight.  (The alpha and weight channels are positive constants.)
__device__ __forceinline__ bool is_finite(Float v) { return (v - v) == 0; }
__device__ __forceinline__ bool put_valid(d3 spec, int b)
{
    const bool allowNegative = (b == 2 || b == 3);
    if (!(is_finite(spec.x) && is_finite(spec.y) && is_finite(spec.z))) return false;
    return allowNegative || !(spec.x < 0 || spec.y < 0 || spec.z < 0);
}

// ImageBlock::put (imageblock.h:150-199) restricted to the film, with fp64 atomics: the exact generic path.
__device__ void spill_put(const FilmD &F, const FilterD &flt, Float px, Float py, d3 spec, Float weight, int b)
{
    if (!put_valid(spec, b)) { atomicAdd(&F.stats[4], 1ULL); return; }
    const bool box = F.fValues == nullptr;
    const Float radius = box ? flt.radius : F.fRadius, scale = box ? flt.scale : F.fScale;
    auto evalD = [&](Float x) -> Float {                       // evalDiscretized, rfilter.h:76-77
        int idx = (int)fabs(x * scale);
        if (idx > 31) idx = 31;
        return box ? (idx < 31 ? flt.c : 0.0) : F.fValues[idx];
    };
    const Float posx = px - 0.5, posy = py - 0.5;
    int x0 = (int)ceil(posx - radius), y0 = (int)ceil(posy - radius);
    int x1 = (int)floor(posx + radius), y1 = (int)floor(posy + radius);
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > F.W - 1) x1 = F.W - 1;
    if (y1 > F.H - 1) y1 = F.H - 1;
    for (int y = y0; y <= y1; ++y) {
        if (y < F.y0 - 2 || y > F.y1 + 1) continue;           // outside this film's rows + its two-row halo: another strip's sample
        const Float wy = evalD(y - posy);
        for (int x = x0; x <= x1; ++x) {
            const Float w = evalD(x - posx) * wy;
            Float *dest = F.spill + (((size_t)b * F.spillRows + (y - (F.y0 - 2))) * F.W + x) * 4;
            atomicAdd(dest + 0, w * spec.x);
            atomicAdd(dest + 1, w * spec.y);
            atomicAdd(dest + 2, w * spec.z);
            atomicAdd(dest + 3, w * weight);
        }
    }
}

// true iff the put at (px,py) covers exactly pixel (ex,ey) before clipping
__device__ __forceinline__ bool single_pixel(const FilterD &flt, Float px, Float py, int ex, int ey)
{
    const Float posx = px - 0.5, posy = py - 0.5;
    return (int)ceil(posx - flt.radius) == ex && (int)floor(posx + flt.radius) == ex &&
           (int)ceil(posy - flt.radius) == ey && (int)floor(posy + flt.radius) == ey;
}

} // namespace gdpt_tr
