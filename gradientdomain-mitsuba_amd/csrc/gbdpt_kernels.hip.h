// gbdpt_kernels.hip.h -- the G-BDPT sampler (BASELINE config 5) as a gfx950 kernel: one lane = one (pixel, sample).
//
// What the reference computes per sample (src/integrators/gbdpt/gbdpt_proc.cpp:152-252 GBDPTRenderer::process, :259-534 evaluate): an emitter
// and a sensor subpath by an alternating random walk (src/libbidir/path.cpp:548-631), the four offset paths of the sensor subpath (pixel
// shifts (0,-1) (-1,0) (1,0) (0,1); src/libbidir/mut_manifold.cpp:806-936 generateOffsetPathGBDPT), then every connection (s, t) of the two
// subpaths for the base and the four offset paths with the MIS weights of path.cpp:49-378 (power 2 over the base strategies for the primal
// value, balance over base + offset strategies for each gradient), light-tracing connections (t = 1) splatted into light images together with
// their own four offset paths (gbdpt_proc.cpp:356-376,568-590).
//
// The reference walks heap-allocated PathVertex / PathEdge objects shared by pointer between base and offset paths.  Here a path is a pair
// of fixed arrays of compact vertex records per lane (position, triangle + barycentrics, the two transport weights and densities, the edge
// that arrived) and an offset path is three records (sensor sample, the shifted first vertex b', the re-connected vertex c') next to the
// base arrays; `SV` below resolves an index of "the offset sensor subpath" to the right record.  All arithmetic is fp64, in the reference's
// operation order (parity with oracle/gbdpt_oracle.hpp is held at 1e-10 per sample).
//
// SCOPE: every surface vertex connectable (Path::isConnectable_GBDPT, path.cpp:30-47): smooth BSDFs with roughness >= shiftThreshold.  The
// C-ABI refuses other scenes (gbdpt_capi.hip); then no specular chain exists: propagatePerturbation / manifoldWalk are never entered,
// SpecularManifold::det is 1 (manifold.cpp:774) and every generalized geometry term of calcSpecularPDFChange (path.cpp:403-421) is a
// product of plain G terms divided by the same product, i.e. exactly 1.
#pragma once
#include "gpt_kernels.hip.h"

namespace gdpt_bd {
using namespace gdpt_tr;

enum { ERadiance = 0, EImportance = 1 };                                                   // include/mitsuba/render/common.h:33-43
enum { M_INVALID = 0, M_SOLID = 1, M_AREA = 3, M_DISCRETE = 4 };                           // EMeasure, common.h:56-67
enum { T_INVALID = 0, T_SENSOR_SUPER = 1, T_EMITTER_SUPER = 2, T_SENSOR_SAMPLE = 4, T_EMITTER_SAMPLE = 8, T_SURFACE = 16 };   // vertex.h:67-87

constexpr int BD_DEFAULT_DEPTH = 12;               // gbdpt_proc.cpp:103-106: maxDepth -1 renders as 12
#ifndef GDPT_BD_MAX_DEPTH          /* (a development build may size the records for another cap: -DGDPT_BD_MAX_DEPTH=12 is round 4's, tools/build_depth12_lib.sh) */
#define GDPT_BD_MAX_DEPTH 20
#endif
constexpr int BD_MAX_DEPTH = GDPT_BD_MAX_DEPTH;                   // the deepest maxDepth the records are sized for (round 5: 12 until then; the reference takes any positive value,
                                                   // gbdpt.cpp:102-103 -- a record holds whole subpaths, so a cap there has to be; the C-ABI refuses beyond it and says so)
constexpr int NSV = BD_MAX_DEPTH + 2;              // sensor subpath records: supernode, sensor sample, up to maxDepth surface vertices
constexpr int NEV = BD_MAX_DEPTH + 1;              // emitter subpath records: supernode, emitter sample, up to maxDepth - 1 surface vertices
constexpr int NMIS = NSV + NEV;                    // pdfImp / pdfRad entries of a full path
constexpr int BD_MAX_LIGHT = 5 * BD_MAX_DEPTH;     // light-image splats of one sample (5 per emitter vertex)
static_assert(BD_MAX_DEPTH + 4 <= 32, "strategy masks are 32-bit words, items pack s and t in 5 bits each");

struct BdConfig {
    int maxDepth, rrDepth, lightImage, spp;
    Float shiftThreshold;
    unsigned long long seed;
    int sBase, sCount;                             // the samples [sBase, sBase + sCount) of every pixel are rendered by this launch
    int hittableEmitters;                          // !Scene::hasDegenerateEmitters (scene.cpp:388,410-411): some emitter is not a point -- the sensor subpath takes one more step, gbdpt_proc.cpp:120-122
    Float sceneRadius;                             // m_scene->getBSphere().radius (the kd-tree's enlarged bounds): the yardstick of the manifold walk's reversibility test
};
struct BdCam {                                     // perspective sensor quantities beyond CameraD (perspective.cpp:167-173,190-247)
    Float invLin[9];                               // linear part of the inverse camera-to-world (trafo.inverse() applied to a direction)
    d3 pos, dir;                                   // trafo(Point(0)), trafo(Vector(0, 0, 1))
    Float rectX, rectY, normalization;             // m_imageRect half extents, 1 / its area
    Float aperturePdf;                             // `thinlens` sensor (CameraD::thinlens): 1 / (pi r^2), thinlens.cpp:213 (0 for the pinhole)
};

struct BV {                                        // PathVertex
    int type, measure, degenerate, componentType;
    int prim, object;                              // surface: leaf-order triangle; emitter samples (and surfaces on emitters after cast()): emitter index
    Float u, v;                                    // surface: barycentrics; sensor sample: film position (pRec.uv)
    d3 p, n;                                       // position; pRec.n of a sensor / emitter sample (a surface keeps its frames in the scene tables)
    d3 w[2];                                       // weight[ERadiance], weight[EImportance]
    Float pdf[2];
    Float rr;                                      // rrWeight
};
struct BE { d3 d; Float length; Float tr[2]; };    // PathEdge: direction (along the light path), length, pdf[mode] == weight[mode] (1, or the 0/1 of a supernode edge)

__device__ __forceinline__ bool bv_connectable(const BV &v) { return !v.degenerate && v.measure != M_DISCRETE; }   // vertex.h:750
// vertex.h:592-596: a surface vertex, or an endpoint sample whose sensor / emitter is EOnSurface -- area lights and both perspective sensors are, a `point`
// emitter is not (point.cpp:56): its samples carry prim == BV_OFF_SURFACE
constexpr int BV_OFF_SURFACE = -2;
// The constant environment emitter as libbidir sees it: a SHAPE (Scene::initializeBidirectional, scene.cpp:397-408; ConstantBackgroundEmitter::createShape,
// constant.cpp:67-93): a `sphere` of m_sceneBSphere with flipped normals, the emitter as its child and an all-absorbing diffuse BSDF (Shape::configure).
// Scene::rayIntersectAll tests it after the kd-tree, so a ray that leaves the geometry ends in a SURFACE vertex on that sphere -- connectable (a smooth BSDF), black,
// castable to an emitter sample.  Such a vertex carries prim == BV_ENV_PRIM and its (inward) normal in BV::n, which triangle vertices leave unused.
constexpr int BV_ENV_PRIM = -3;
// development aid (-DGDPT_BD_CHECK_PRIM): every table load indexed by a vertex's `prim` records an index outside the triangle tables in g_bdPrimTrap (site, prim,
// count) and goes on with triangle 0, so the kernel completes and the host can say where (gdpt_gbdpt_evaluate_sample2 prints it and fills its workspace with 0x7f
// bytes first, so a read of a vertex nobody wrote shows up whatever the allocation held before)
#ifdef GDPT_BD_CHECK_PRIM
__device__ int g_bdPrimTrap[4];
__device__ __forceinline__ int bd_prim_trap(int prim, int numTris, int site)
{
    if ((unsigned)prim < (unsigned)numTris) return prim;
    if (atomicAdd(&g_bdPrimTrap[2], 1) == 0) { g_bdPrimTrap[0] = site; g_bdPrimTrap[1] = prim; }
    return 0;
}
#define BD_PRIM(c, prim, where) bd_prim_trap((int)(prim), (c).S->numTris, (int)sizeof(where) * 1000 + __LINE__)
#else
// The product: every such index is CLAMPED into the tables.  A vertex that is not a triangle hit (prim -1 of an endpoint sample, BV_OFF_SURFACE, BV_ENV_PRIM, or a
// record nobody wrote yet) never has its table entry USED -- the trap build above shows that, on workspaces filled with 0x7f bytes -- but the load itself may be issued
// ahead of the test that discards it: the product build of round 5's environment cases faulted in k_gbdpt_sample on a TriNormals load (rocgdb), only after other tests
// had left their bytes in the allocation, and not once in the build that clamps.  Two integer instructions per table access beside fp64 shading.
__device__ __forceinline__ int bd_prim_clamp(int prim, int numTris) { return min(max(prim, 0), numTris - 1); }
#define BD_PRIM(c, prim, where) bd_prim_clamp((int)(prim), (c).S->numTris)
#endif
__device__ __forceinline__ bool bv_on_surface(const BV &v) { return v.type == T_SURFACE || ((v.type == T_EMITTER_SAMPLE || v.type == T_SENSOR_SAMPLE) && v.prim != BV_OFF_SURFACE); }
__device__ __forceinline__ bool bv_super(const BV &v) { return (v.type & 3) != 0; }
__device__ __forceinline__ void bv_clear(BV &v)
{
    v.type = T_INVALID; v.measure = M_INVALID; v.degenerate = 0; v.componentType = 0; v.prim = -1; v.object = -1; v.u = v.v = 0.0;
    v.p = mk(0.0); v.n = mk(0.0); v.w[0] = v.w[1] = mk(0.0); v.pdf[0] = v.pdf[1] = 0.0; v.rr = 0.0;
}
__device__ __forceinline__ void be_clear(BE &e) { e.d = mk(0.0); e.length = 0.0; e.tr[0] = e.tr[1] = 0.0; }

struct Surf { Frame3 fr; d3 geoN; MaterialD m; d3 R; };   // what its.getBSDF() / its.shFrame / its.geoFrame give at a surface vertex

struct Ctx {
    const SceneD *S;
    SceneView V;
    BdCam cam;
    BdConfig cfg;
    int *stack;
    Rng rng;
    unsigned nClosest, nShadow;
};

__device__ __forceinline__ Surf surf_of(const Ctx &c, const BV &v)
{
    if (v.prim == BV_ENV_PRIM) {                                                           // (sphere.cpp:248-251 sets shFrame.n only; nothing that survives the black BSDF reads s and t: Frame(n) stands in)
        Surf s;
        s.fr.n = v.n;
        if (fabs(v.n.x) > fabs(v.n.y)) { const Float il = 1.0 / sqrt(v.n.x * v.n.x + v.n.z * v.n.z); s.fr.t = mk(v.n.z * il, 0.0, -v.n.x * il); }
        else { const Float il = 1.0 / sqrt(v.n.y * v.n.y + v.n.z * v.n.z); s.fr.t = mk(0.0, v.n.z * il, -v.n.y * il); }
        s.fr.s = cross(s.fr.t, s.fr.n);
        s.geoN = v.n;
        s.m = c.V.mats[0]; s.m.type = 0; s.m.twoSided = 0; s.m.reflectance = mk(0.0); s.m.tex = -1;   // Shape::configure: "light source & no BSDF -> an all-absorbing BSDF" (diffuse, reflectance 0)
        s.R = mk(0.0);
        return s;
    }
    // (max(prim, 0): the table loads below go through C++ references, which the compiler may hoist above the test of prim -- a load from shade[-3] then reads
    //  whatever lies in front of the table, or faults when nothing does: seen as a memory violation in scenes whose tables happened to start an allocation)
    Vertex vx; vx.p = v.p; vx.prim = max(BD_PRIM(c, v.prim, "surf_of"), 0); vx.u = v.u; vx.v = v.v;
    const Shading sh = shading_at<true>(c.V, vx);
    Surf s;
    s.fr = sh.fr; s.geoN = sh.geoN;
    s.m = c.V.mats[c.V.shade[vx.prim].material];
    s.R = reflectance_at<true, false>(c.V, s.m, vx);                                       // libbidir asks its.getBSDF() without a ray: no UV partials
    return s;
}
__device__ __forceinline__ d3 bv_sh_normal(const Ctx &c, const BV &v)
{
    if (v.type != T_SURFACE || v.prim == BV_ENV_PRIM) return v.n;
    Vertex vx; vx.p = v.p; vx.prim = max(BD_PRIM(c, v.prim, "bv_sh_normal"), 0); vx.u = v.u; vx.v = v.v;
    return shading_at<true>(c.V, vx).fr.n;
}
__device__ __forceinline__ d3 bv_geo_normal(const Ctx &c, const BV &v)
{
    if (v.type != T_SURFACE || v.prim == BV_ENV_PRIM) return v.n;
    Vertex vx; vx.p = v.p; vx.prim = max(BD_PRIM(c, v.prim, "bv_geo_normal"), 0); vx.u = v.u; vx.v = v.v;
    return shading_at<true>(c.V, vx).geoN;
}
__device__ __forceinline__ Float mat_roughness(const MaterialD &m) { return m.type == 0 ? GD_INF : ((m.type == 1 || m.type == 3) ? 0.0 : 0.5 * (m.alphaU + m.alphaV)); }
// Path::isConnectable_GBDPT, path.cpp:30-47
__device__ __forceinline__ bool connectable_gbdpt(const Ctx &c, const BV &v)
{
    if (!bv_connectable(v)) return false;
    if (v.type & (T_SENSOR_SUPER | T_EMITTER_SUPER | T_SENSOR_SAMPLE | T_EMITTER_SAMPLE)) return true;
    if (v.prim == BV_ENV_PRIM) return true;                                                // (diffuse: roughness infinite)
    return !(mat_roughness(c.V.mats[c.V.shade[max(BD_PRIM(c, v.prim, "connectable_gbdpt"), 0)].material]) < c.cfg.shiftThreshold);
}

__device__ __forceinline__ int prim_emitter(const Ctx &c, int prim) { return prim == BV_ENV_PRIM ? c.S->envIndex : c.V.shade[max(BD_PRIM(c, prim, "prim_emitter"), 0)].emitter; }
__device__ __forceinline__ int prim_material(const Ctx &c, int prim) { return prim == BV_ENV_PRIM ? -1 : c.V.shade[max(BD_PRIM(c, prim, "prim_material"), 0)].material; }

// ---- sensor --------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ d3 cam_to_local(const Ctx &c, d3 d) { return mul3(c.cam.invLin, d); }
__device__ __forceinline__ d3 cam_to_world(const Ctx &c, d3 d)
{
    const Float *M = c.S->cam.m;
    return mk(M[0] * d.x + M[1] * d.y + M[2] * d.z, M[4] * d.x + M[5] * d.y + M[6] * d.z, M[8] * d.x + M[9] * d.y + M[10] * d.z);
}
// PerspectiveCameraImpl::importance, perspective.cpp:190-247
__device__ __forceinline__ Float importance(const Ctx &c, d3 d)
{
    const Float cosT = d.z;
    if (cosT <= 0) return 0.0;
    const Float inv = 1.0 / cosT;
    const Float px = d.x * inv, py = d.y * inv;
    if (!(px >= -c.cam.rectX && px <= c.cam.rectX && py >= -c.cam.rectY && py <= c.cam.rectY)) return 0.0;
    return c.cam.normalization * inv * inv * inv;
}
// m_cameraToSample(P).xy for crop == film: the film position, in [0,1]^2, that the camera-space point P projects to
__device__ __forceinline__ void camera_to_sample(const CameraD &cam, d3 P, Float &sx, Float &sy)
{
    sx = 0.5 * (1 - P.x / (P.z * cam.tanHalf)); sy = 0.5 * (1 - P.y * cam.aspect / (P.z * cam.tanHalf));
}
__device__ __forceinline__ d3 cam_to_local_point(const Ctx &c, d3 p) { return mul3(c.cam.invLin, p - c.cam.pos); }   // trafo.inverse().transformAffine(p)
// ThinLensCamera::importance, thinlens.cpp:231-291: p a point of the aperture, d the direction from it (camera space); the pixel is the one whose focus-plane
// point the ray passes through
__device__ __forceinline__ Float importance_lens(const Ctx &c, d3 p, d3 d)
{
    const Float cosT = d.z;
    if (cosT <= 0) return 0.0;
    const Float inv = 1.0 / cosT;
    Float sx, sy;
    camera_to_sample(c.S->cam, p + d * (c.S->cam.focusDistance * inv), sx, sy);
    if (sx < 0 || sx > 1 || sy < 0 || sy > 1) return 0.0;
    return c.cam.normalization * inv * inv * inv;
}
// Sensor::evalDirection == pdfDirection of a sensor sample at world position p towards world direction d (perspective.cpp:373-391, thinlens.cpp:420-437)
__device__ __forceinline__ Float sensor_direction(const Ctx &c, d3 p, d3 d)
{
    return c.S->cam.thinlens ? importance_lens(c, cam_to_local_point(c, p), cam_to_local(c, d)) : importance(c, cam_to_local(c, d));
}
// normalize(m_sampleToCamera(Point(sx, sy, 0))), perspective.cpp:150-156 written out for crop == film
__device__ __forceinline__ d3 sample_to_camera_dir(const Ctx &c, Float sxn, Float syn)
{
    const CameraD &cam = c.S->cam;
    return normalize(mk((1 - 2 * sxn) * cam.nearClip * cam.tanHalf, (1 - 2 * syn) / cam.aspect * cam.nearClip * cam.tanHalf, cam.nearClip));
}
// the direction of sensor->sampleRay(ray, film position, aperture sample (0.5, 0.5)) in camera space: perspective.cpp:249-269; thinlens.cpp:293-322, whose
// aperture point is then the lens centre: normalize(focusP - 0)
__device__ __forceinline__ d3 centre_ray_dir(const Ctx &c, Float sxn, Float syn)
{
    const CameraD &cam = c.S->cam;
    if (!cam.thinlens) return sample_to_camera_dir(c, sxn, syn);
    const d3 nearP = mk((1 - 2 * sxn) * cam.nearClip * cam.tanHalf, (1 - 2 * syn) / cam.aspect * cam.nearClip * cam.tanHalf, cam.nearClip);
    return normalize(nearP * (cam.focusDistance / nearP.z));
}
// getFocusDistance(): the `focusDistance` of a thinlens sensor; a pinhole's default is the far clip, sensor.cpp:162
__device__ __forceinline__ Float focus_distance(const CameraD &cam) { return cam.thinlens ? cam.focusDistance : cam.farClip; }
// PerspectiveCameraImpl::getSamplePosition, perspective.cpp:393-410; ThinLensCamera::getSamplePosition, thinlens.cpp:536-557 (pWorld: the sensor sample's
// point of the aperture; the pinhole does not read it)
__device__ __forceinline__ bool sensor_sample_position(const Ctx &c, d3 pWorld, d3 dWorld, Float &ox, Float &oy)
{
    const CameraD &cam = c.S->cam;
    const d3 l = cam_to_local(c, dWorld);
    if (l.z <= 0) return false;
    if (cam.thinlens) {
        Float sx, sy;
        camera_to_sample(cam, cam_to_local_point(c, pWorld) + l * (cam.focusDistance / l.z), sx, sy);
        if (sx < 0 || sx > 1 || sy < 0 || sy > 1) return false;
        ox = sx * cam.width; oy = sy * cam.height;
        return true;
    }
    const Float sx = 0.5 * (1 - l.x / (l.z * cam.tanHalf)), sy = 0.5 * (1 - l.y * cam.aspect / (l.z * cam.tanHalf));
    if (sx < 0 || sx > 1 || sy < 0 || sy > 1) return false;
    ox = sx * cam.width; oy = sy * cam.height;
    return true;
}

// ---- emitters ------------------------------------------------------------------------------------------------------------------------
// `envmap` as libbidir's endpoint (round 5): positions as the constant environment has them (uniform on the sphere of createShape, envmap.cpp:331-343,412-430), directions
// from the map itself -- sampleDirection / pdfDirection / evalDirection (envmap.cpp:455-498) importance-sample, price and look up a direction of the map whatever the
// position (envmap.cpp:432-454 calls it a compromise).  m_power = surfaceArea * m_scale / m_normalization (envmap.cpp:326-328)
__device__ __forceinline__ Float envmap_power(const Ctx &c) { const Float surfaceArea = 4 * GD_PI * c.S->bsRadius * c.S->bsRadius; return surfaceArea * c.S->envMap->scale / c.S->envMap->normalization; }
__device__ __forceinline__ bool is_envmap(const Ctx &c, int object) { return c.S->hasEnvMap && object == c.S->envIndex; }
// EnvironmentMap::evalDirection, envmap.cpp:482-498: bilinear on level 0 along -d (map space), times m_normalization (not m_scale); the measure is not consulted
__device__ __noinline__ d3 envmap_eval_direction(const Ctx &c, d3 d)
{
    const EnvMapD &e = *c.S->envMap;
    const d3 v = -mul3(e.toLocal, d);
    const Float uvx = atan2(v.x, -v.z) * GD_INV_TWOPI, uvy = acos(fmin(1.0, fmax(-1.0, v.y))) * GD_INV_PI;
    return tex_bilinear(e.tex, 0, uvx, uvy) * e.normalization;
}
// Scene::sampleEmitterPosition (scene.cpp:985-1001) -> AreaLight::samplePosition (area.cpp:93-97) -> TriMesh / Rectangle::samplePosition
__device__ __forceinline__ d3 sample_emitter_position(const Ctx &c, BV &succ, Float sx, Float sy, Float &pdfOut)
{
    const SceneView &V = c.V;
    const int index = cdf_sample(V.emitterCdf, c.S->numEmitters, sx);
    const Float emPdf = V.emitterCdf[index + 1] - V.emitterCdf[index];
    sx = (sx - V.emitterCdf[index]) / (V.emitterCdf[index + 1] - V.emitterCdf[index]);
    const EmitterD em = V.emitters[index];
    if (em.numTris == 0) {                                                                 // ConstantBackgroundEmitter::samplePosition, constant.cpp:110-120
        const Float z = 1.0 - 2.0 * sy, r = safe_sqrt(1.0 - z * z), phi = 2.0 * GD_PI * sx;  // warp::squareToUniformSphere, warp.cpp:25-31
        const d3 d = mk(r * cos(phi), r * sin(phi), z);
        const Float R = c.S->bsRadius;
        succ.p = c.S->bsCenter + d * R; succ.n = -d; succ.object = index;
        Float pdf = 1 / (4 * GD_PI * R * R);
        pdf *= emPdf;
        pdfOut = pdf;
        const Float surfaceArea = 4 * GD_PI * R * R;
        if (c.S->hasEnvMap) return mk(envmap_power(c)) / emPdf;                              // EnvironmentMap::samplePosition, envmap.cpp:412-422: the same sphere, Spectrum(m_power)
        return (em.radiance * surfaceArea * GD_PI) / emPdf;                                 // m_power, constant.cpp:100
    }
    if (em.numTris < 0) {                                                                  // PointEmitter::samplePosition, point.cpp:79-87
        succ.p = em.position; succ.n = mk(0.0); succ.object = index; succ.prim = BV_OFF_SURFACE;
        Float pdf = 1.0;
        pdf *= emPdf;
        pdfOut = pdf;
        return (em.radiance * (4 * GD_PI)) / emPdf;
    }
    if (em.rectangle) {
        const Float lx = sx * 2 - 1, ly = sy * 2 - 1;
        succ.p = mk(em.rect[0] * lx + em.rect[1] * ly + em.rect[2] * 0.0 + em.rect[3], em.rect[4] * lx + em.rect[5] * ly + em.rect[6] * 0.0 + em.rect[7],
                    em.rect[8] * lx + em.rect[9] * ly + em.rect[10] * 0.0 + em.rect[11]);
        succ.n = em.rectN;
        succ.u = sx; succ.v = sy;
    } else {
        const Float *cdf = V.emCdf + em.cdfOffset;
        const int ti = cdf_sample(cdf, em.numTris, sy);
        sy = (sy - cdf[ti]) / (cdf[ti + 1] - cdf[ti]);
        const EmTri tr = V.emTris[em.firstEmTri + ti];
        const Float a = safe_sqrt(1.0 - sx);
        const Float bx = 1 - a, by = a * sy;
        const d3 sideA = tr.p1 - tr.p0, sideB = tr.p2 - tr.p0;
        succ.p = tr.p0 + (sideA * bx) + (sideB * by);
        succ.n = normalize(cross(sideA, sideB));
        succ.u = bx; succ.v = by;
    }
    Float pdf = em.invSurfaceArea;
    succ.object = index;
    pdf *= emPdf;
    pdfOut = pdf;
    const Float area = 1.0 / em.invSurfaceArea;
    const d3 power = em.radiance * GD_PI * area;
    return power / emPdf;
}
__device__ __forceinline__ Float pdf_emitter_position(const Ctx &c, int object, int measure)   // scene.cpp:1003-1006 (pRec.measure = measure, vertex.cpp:925-927)
{
    if (c.V.emitters[object].numTris < 0) return (measure == M_DISCRETE ? 1.0 : 0.0) * (1.0 * c.S->emitterNormalization);   // point.cpp:93-95
    if (c.V.emitters[object].numTris == 0) return (1 / (4 * GD_PI * c.S->bsRadius * c.S->bsRadius)) * (1.0 * c.S->emitterNormalization);   // constant.cpp:126-128
    return c.V.emitters[object].invSurfaceArea * (1.0 * c.S->emitterNormalization);
}
__device__ __forceinline__ Float area_direction(d3 d, d3 n, int measure)                    // AreaLight::evalDirection / pdfDirection, area.cpp:124-142
{
    Float dp = dot(d, n);
    if (measure != M_SOLID || dp < 0) dp = 0.0;
    return GD_INV_PI * dp;
}
// Emitter::evalDirection == pdfDirection of an emitter sample: AreaLight (above) or PointEmitter, point.cpp:107-115
#define GD_INV_FOURPI 0.07957747154594766788
__device__ __forceinline__ Float emitter_direction(const Ctx &c, const BV &v, d3 d, int measure)
{
    if (v.prim == BV_OFF_SURFACE) return measure == M_SOLID ? GD_INV_FOURPI : 0.0;
    if (is_envmap(c, v.object)) return envmap_pdf_direction(*c.S->envMap, -mul3(c.S->envMap->toLocal, d));   // EnvironmentMap::pdfDirection, envmap.cpp:476-480 (no measure test)
    return area_direction(d, v.n, measure);
}
// Emitter::evalDirection as a spectrum: the envmap's is coloured (envmap.cpp:482-498), every other emitter's equals its pdfDirection
__device__ __forceinline__ d3 emitter_eval_direction(const Ctx &c, const BV &v, d3 d, int measure)
{
    if (is_envmap(c, v.object)) return envmap_eval_direction(c, d);
    return mk(emitter_direction(c, v, d, measure));
}
__device__ __forceinline__ int bsdf_measure(int m) { return m == M_DISCRETE ? MEASURE_DISCRETE : MEASURE_SOLID_ANGLE; }

// ---- rays ----------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool closest_hit(Ctx &c, d3 o, d3 d, Float mint, Float maxt, Hit &h)
{
    c.nClosest++;
    return trace<false>(c.V, c.stack, o, d, ray_mint_closest(o, mint), maxt, h);
}
__device__ __forceinline__ bool any_hit(Ctx &c, d3 o, d3 d, Float mint, Float maxt)
{
    c.nShadow++;
    Hit h;
    return trace<true>(c.V, c.stack, o, d, ray_mint_shadow(o, mint), maxt, h);
}
__device__ __forceinline__ void fill_surface(const Ctx &c, const Hit &h, BV &succ)
{
    const TriShade &ts = c.V.shade[h.prim];
    const d3 b = mk(1 - h.u - h.v, h.u, h.v);
    succ.type = T_SURFACE;
    succ.prim = h.prim; succ.u = h.u; succ.v = h.v;
    succ.p = ts.p0 * b.x + ts.p1 * b.y + ts.p2 * b.z;
    const MaterialD &m = c.V.mats[ts.material];
    succ.degenerate = !((bsdfType(m) & ESmooth) || ts.emitter >= 0);                       // edge.cpp:44-45
    succ.object = ts.emitter;
}
// PathEdge::sampleNext (edge.cpp:27-71) and PathEdge::perturbDirection (:73-131) without media: the ray's closest hit becomes the successor
__device__ __forceinline__ bool edge_extend(Ctx &c, BE &e, d3 o, d3 d, BV &succ, int mode, bool perturb, Float dist)
{
    Hit h;
    bool surface = closest_hit(c, o, d, GD_EPSILON, GD_INF, h);
    if (perturb && dist <= 0) return false;
    bool envHit = false;
    if (!surface && c.S->envIndex >= 0) {                                                   // Scene::rayIntersectAll, scene.cpp:736-760: the special shapes after the kd-tree; Sphere::rayIntersect, sphere.cpp:163-187
        const Float mint = ray_mint_closest(o, GD_EPSILON);                                // (the sphere encloses the geometry: it only ever answers when the kd-tree found nothing)
        const d3 oc = o - c.S->bsCenter;
        Float nearT, farT;
        if (solve_quadratic(len2(d), 2 * dot(oc, d), len2(oc) - c.S->bsRadius * c.S->bsRadius, nearT, farT) && nearT <= GD_INF && farT >= mint) {
            if (nearT < mint) { if (!(farT > GD_INF)) { h.t = farT; envHit = true; } }
            else { h.t = nearT; envHit = true; }
        }
        surface = envHit;
    }
    if (!surface) return false;
    if (envHit) h.prim = 0;                                                                // (fill_surface's table loads may be hoisted above the branch below: they must be in bounds either way)
    fill_surface(c, h, succ);
    if (envHit) {                                                                          // Sphere::fillIntersectionRecord, sphere.cpp:209-255 (flipped normals); edge.cpp:44-45
        bv_clear(succ);
        succ.type = T_SURFACE; succ.prim = BV_ENV_PRIM;
        succ.p = o + d * h.t;
        succ.n = -normalize(succ.p - c.S->bsCenter);
        succ.degenerate = 0; succ.object = c.S->envIndex;
    }
    e.length = h.t;
    e.d = mode == ERadiance ? -d : d;
    if (e.length == 0) return false;
    e.tr[0] = e.tr[1] = 1.0;
    return true;
}
// PathEdge::connect, edge.cpp:221-287
// (knownVisible: the caller has traced this very segment in an earlier launch and found it free -- phase 2 of a connection builds its base path again for the
//  state the offsets share with it; the ray is neither traced nor counted a second time)
__device__ __forceinline__ bool edge_connect(Ctx &c, BE &e, const BV &vs, const BV &vt, bool knownVisible = false)
{
    if (vs.type == T_EMITTER_SUPER || vt.type == T_SENSOR_SUPER) {
        const Float rad = vt.type == T_SENSOR_SUPER ? 1.0 : 0.0;
        e.d = mk(0.0); e.length = 0.0; e.tr[ERadiance] = rad; e.tr[EImportance] = 1 - rad;
    } else {
        e.d = vs.p - vt.p;
        e.length = len(e.d);
        e.d = e.d / e.length;
        if (!knownVisible && any_hit(c, vt.p, e.d, bv_on_surface(vt) ? GD_EPSILON : 0.0, e.length * (bv_on_surface(vs) ? (1 - GD_SHADOW_EPSILON) : 1.0))) return false;
        e.tr[0] = e.tr[1] = 1.0;
    }
    e.d = -e.d;
    return true;
}
// PathEdge::pathConnectAndCollapse, edge.cpp:442-574 (no media, no ENull BSDF: a surface in between is an occluder)
__device__ __forceinline__ bool edge_path_connect(Ctx &c, BE &e, const BV &vs, const BV &vt, bool knownVisible = false)
{
    if (vs.type == T_EMITTER_SUPER || vt.type == T_SENSOR_SUPER) {
        const Float rad = vt.type == T_SENSOR_SUPER ? 1.0 : 0.0;
        e.length = 0.0; e.d = mk(0.0); e.tr[ERadiance] = rad; e.tr[EImportance] = 1 - rad;
    } else {
        e.d = vs.p - vt.p;
        e.length = len(e.d);
        if (e.length == 0) return false;
        e.d = e.d / e.length;
        e.tr[0] = e.tr[1] = 1.0;
        Hit h;
        if (!knownVisible && closest_hit(c, vt.p, e.d, bv_on_surface(vt) ? GD_EPSILON : 0.0, e.length * (bv_on_surface(vs) ? (1 - GD_SHADOW_EPSILON) : 1.0), h)) return false;
    }
    e.d = -e.d;
    return true;
}
// PathEdge::evalCached(pred, succ, EGeneralizedGeometricTerm), edge.cpp:169-219: cosines at connectable surface ends, inverse square, transmittance
__device__ __forceinline__ Float edge_geometry_term(const Ctx &c, const BE &e, const BV &pred, const BV &succ)
{
    Float result = 1.0;
    if (e.length == 0) return result;
    if (bv_on_surface(pred) && bv_connectable(pred)) result *= fabs(dot(bv_sh_normal(c, pred), e.d));
    if (bv_on_surface(succ) && bv_connectable(succ)) result *= fabs(dot(bv_sh_normal(c, succ), e.d));
    result /= e.length * e.length;
    result *= e.tr[EImportance] * e.tr[EImportance];                                        // weight[EImportance] * pdf[EImportance]
    return result;
}

// ---- BSDFs as libbidir asks for them: transport mode (dielectric.cpp:248-250,296-298: radiance is scaled when it crosses the interface,
// importance is not) and a component mask (PathVertex::propagatePerturbation takes ONE delta component, vertex.cpp:696-700) -----------------
__device__ void bd_eval_pdf(const MaterialD &m, d3 R, d3 wi, d3 wo, int measure, bool importance, int typeMask, d3 &f, Float &pdf)
{
    if (m.type == 3) {
        f = mk(0.0); pdf = 0.0;
        Float cosThetaT;
        const Float F = fresnelDielectricExt(wi.z, cosThetaT, m.eta.x);
        if (measure != MEASURE_DISCRETE) return;
        const bool sampleReflection = (typeMask & EDeltaReflection) != 0, sampleTransmission = (typeMask & EDeltaTransmission) != 0;
        if (wi.z * wo.z >= 0) {
            if (!sampleReflection || fabs(dot(mk(-wi.x, -wi.y, wi.z), wo) - 1) > GD_DELTA_EPSILON) return;
            f = R * F; pdf = sampleTransmission ? F : 1.0;
        } else {
            if (!sampleTransmission || fabs(dot(dielectric_refract(m, wi, cosThetaT), wo) - 1) > GD_DELTA_EPSILON) return;
            const Float factor = importance ? 1.0 : (cosThetaT < 0 ? 1 / m.eta.x : m.eta.x);
            f = m.k * factor * factor * (1 - F); pdf = sampleReflection ? 1 - F : 1.0;
        }
        return;
    }
    if (m.type == 1 && !(typeMask & EDeltaReflection)) { f = mk(0.0); pdf = 0.0; return; }
    bsdf_eval_pdf(m, R, wi, wo, measure, f, pdf);
}
__device__ void bd_sample(const MaterialD &m, d3 R, d3 wi, Float sx, Float sy, bool importance, int typeMask, BSDFSample &r)
{
    if (m.type == 3) {
        r.wo = mk(0.0); r.weight = mk(0.0); r.pdf = 0.0; r.eta = 1.0; r.sampledType = 0;
        Float cosThetaT;
        const Float F = fresnelDielectricExt(wi.z, cosThetaT, m.eta.x);
        const bool sampleReflection = (typeMask & EDeltaReflection) != 0, sampleTransmission = (typeMask & EDeltaTransmission) != 0;
        const Float factor = importance ? 1.0 : (cosThetaT < 0 ? 1 / m.eta.x : m.eta.x);
        if (sampleReflection && sampleTransmission) {
            if (sx <= F) { r.sampledType = EDeltaReflection; r.wo = mk(-wi.x, -wi.y, wi.z); r.pdf = F; r.weight = R; }
            else { r.sampledType = EDeltaTransmission; r.wo = dielectric_refract(m, wi, cosThetaT); r.eta = cosThetaT < 0 ? m.eta.x : 1 / m.eta.x; r.pdf = 1 - F; r.weight = m.k * (factor * factor); }
        } else if (sampleReflection) { r.sampledType = EDeltaReflection; r.wo = mk(-wi.x, -wi.y, wi.z); r.pdf = 1.0; r.weight = R * F; }
        else if (sampleTransmission) { r.sampledType = EDeltaTransmission; r.wo = dielectric_refract(m, wi, cosThetaT); r.eta = cosThetaT < 0 ? m.eta.x : 1 / m.eta.x; r.pdf = 1.0; r.weight = m.k * (factor * factor * (1 - F)); }
        return;
    }
    if (m.type == 1 && !(typeMask & EDeltaReflection)) { r.wo = mk(0.0); r.weight = mk(0.0); r.pdf = 0.0; r.eta = 1.0; r.sampledType = 0; return; }
    bsdf_sample(m, R, wi, sx, sy, r);
}
constexpr int BD_ALL = EDelta | ESmooth;

// ---- PathVertex ----------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void adjoint(BV &v, int mode, d3 wiL, d3 woL, Float wiDotGeoN, Float woDotGeoN)   // vertex.cpp:213-221,611-619
{
    if (mode == EImportance) v.w[EImportance] = v.w[EImportance] * fabs((wiL.z * woDotGeoN) / (woL.z * wiDotGeoN));
    else v.w[EImportance] = v.w[EImportance] * fabs((woL.z * wiDotGeoN) / (wiL.z * woDotGeoN));
}
__device__ __forceinline__ void to_area(const Ctx &c, BV &v, int mode, const BV &pred, const BE &predEdge, const BE &succEdge, const BV &succ, d3 rayD)
{                                                                                          // vertex.cpp:294-307,663-676
    if (v.measure != M_SOLID) return;
    v.measure = M_AREA;
    v.pdf[mode] /= succEdge.length * succEdge.length;
    if (bv_on_surface(succ)) v.pdf[mode] *= fabs(dot(rayD, bv_geo_normal(c, succ)));
    if (predEdge.length != 0.0) {
        v.pdf[1 - mode] /= predEdge.length * predEdge.length;
        if (bv_on_surface(pred)) v.pdf[1 - mode] *= fabs(dot(predEdge.d, bv_geo_normal(c, pred)));
    }
}
// PathVertex::sampleNext, vertex.cpp:35-310 (the sensor endpoints go through sample_sensor)
__device__ __noinline__ bool sample_next(Ctx &c, BV &v, const BV *pred, const BE *predEdge, BE &succEdge, BV &succ, int mode, bool russianRoulette, d3 &throughput)
{
    d3 ro, rd;
    be_clear(succEdge); bv_clear(succ);
    v.rr = 1.0;
    if (v.type == T_EMITTER_SUPER) {
        const Float sx = c.rng.next1D(), sy = c.rng.next1D();
        Float ppdf;
        const d3 result = sample_emitter_position(c, succ, sx, sy, ppdf);
        if (is_zero(result)) return false;
        v.w[EImportance] = result;
        v.pdf[EImportance] = ppdf;
        v.measure = succ.prim == BV_OFF_SURFACE ? M_DISCRETE : M_AREA;
        succ.type = T_EMITTER_SAMPLE;
        succ.degenerate = 0;
        succEdge.tr[EImportance] = 1.0;
        return true;
    } else if (v.type == T_EMITTER_SAMPLE) {                                               // AreaLight::sampleDirection, area.cpp:114-122
        const Float sx = c.rng.next1D(), sy = c.rng.next1D();
        if (v.prim == BV_OFF_SURFACE) {                                                    // PointEmitter::sampleDirection, point.cpp:97-105: not EOnSurface, no cosine
            const Float z = 1.0 - 2.0 * sy, r = safe_sqrt(1.0 - z * z);                     // warp::squareToUniformSphere, warp.cpp:25-31
            rd = mk(r * cos(2.0 * GD_PI * sx), r * sin(2.0 * GD_PI * sx), z);
            v.w[EImportance] = mk(1.0);
            v.w[ERadiance] = mk(1.0) * GD_INV_FOURPI;
            v.pdf[EImportance] = GD_INV_FOURPI;
            v.pdf[ERadiance] = 1.0;
            v.measure = M_SOLID;
            ro = v.p;
        } else if (is_envmap(c, v.object)) {                                               // EnvironmentMap::sampleDirection, envmap.cpp:455-474
            d3 value, dl; Float dpdf;
            envmap_sample_direction(*c.S->envMap, sx, sy, dl, value, dpdf);
            rd = mul3(c.S->envMap->toWorld, -dl);
            if (is_zero(value) || dpdf == 0) return false;                                  // ("be wary of roundoff errors": Spectrum(0) -> sampleNext fails, vertex.cpp:106-107)
            const d3 result = (value * c.S->envMap->normalization) / (dpdf * c.S->envMap->scale);
            if (is_zero(result)) return false;
            v.w[EImportance] = result;
            v.w[ERadiance] = result * dpdf * (1.0 / fabs(dot(rd, v.n)));                      // (EOnSurface, envmap.cpp:107)
            v.pdf[EImportance] = dpdf;
            v.pdf[ERadiance] = 1.0;
            v.measure = M_SOLID;
            ro = v.p;
        } else {
        const d3 local = squareToCosineHemisphere(sx, sy);
        Frame3 fr; fr.n = v.n;
        if (fabs(fr.n.x) > fabs(fr.n.y)) { const Float il = 1.0 / sqrt(fr.n.x * fr.n.x + fr.n.z * fr.n.z); fr.t = mk(fr.n.z * il, 0.0, -fr.n.x * il); }   // coordinateSystem, util.cpp:592-601
        else { const Float il = 1.0 / sqrt(fr.n.y * fr.n.y + fr.n.z * fr.n.z); fr.t = mk(0.0, fr.n.z * il, -fr.n.y * il); }
        fr.s = cross(fr.t, fr.n);
        rd = toWorld(fr, local);
        const Float dpdf = GD_INV_PI * local.z;
        v.w[EImportance] = mk(1.0);
        v.w[ERadiance] = mk(1.0) * dpdf * (1.0 / fabs(dot(rd, v.n)));
        v.pdf[EImportance] = dpdf;
        v.pdf[ERadiance] = 1.0;
        v.measure = M_SOLID;
        ro = v.p;
        }
    } else if (v.type == T_SURFACE) {
        const Surf sf = surf_of(c, v);
        const d3 wi = normalize(pred->p - v.p);
        const d3 wiL = toLocal(sf.fr, wi);
        const Float sx = c.rng.next1D(), sy = c.rng.next1D();
        BSDFSample bs;
        bd_sample(sf.m, sf.R, wiL, sx, sy, mode == EImportance, BD_ALL, bs);
        v.w[mode] = bs.weight; v.pdf[mode] = bs.pdf;
        if (is_zero(v.w[mode])) return false;
        v.measure = (bs.sampledType & ESmooth) ? M_SOLID : M_DISCRETE;
        v.componentType = bs.sampledType;
        const d3 wo = toWorld(sf.fr, bs.wo);
        const Float wiDotGeoN = dot(sf.geoN, wi), woDotGeoN = dot(sf.geoN, wo);
        if (wiDotGeoN * wiL.z <= 0 || woDotGeoN * bs.wo.z <= 0) return false;
        d3 fRev; Float pRev;
        bd_eval_pdf(sf.m, sf.R, bs.wo, wiL, bsdf_measure(v.measure), (1 - mode) == EImportance, BD_ALL, fRev, pRev);   // bRec.reverse() (flips the mode too); bsdf->pdf
        v.pdf[1 - mode] = pRev;
        if (v.pdf[1 - mode] <= 0x1p-1024) return false;
        if (sf.m.type != 3) {                                                              // (the dielectric is the one ENonSymmetric BSDF of the subset, dielectric.cpp:87-98)
            v.w[1 - mode] = v.w[mode] * (v.pdf[mode] / v.pdf[1 - mode]);
            if (v.measure == M_SOLID) v.w[1 - mode] = v.w[1 - mode] * fabs(wiL.z / bs.wo.z);
        } else v.w[1 - mode] = fRev / v.pdf[1 - mode];
        adjoint(v, mode, wiL, bs.wo, wiDotGeoN, woDotGeoN);
        if (mode == ERadiance && bs.eta != 1) throughput = throughput * (bs.eta * bs.eta);     // "for BDPT & russian roulette, track radiance * eta^2", vertex.cpp:225-227
        ro = v.p; rd = wo;
    } else return false;
    throughput = throughput * v.w[mode];
    if (russianRoulette) {
        const Float q = fmin(maxc(throughput), (Float)0.95f);
        if (c.rng.next1D() > q) { v.measure = M_INVALID; return false; }
        v.rr = 1.0 / q;
        throughput = throughput * v.rr;
    }
    if (!edge_extend(c, succEdge, ro, rd, succ, mode, false, 0.0)) { v.measure = M_INVALID; return false; }
    to_area(c, v, mode, *pred, *predEdge, succEdge, succ, rd);
    return true;
}
// PathVertex::sampleSensor, vertex.cpp:312-384 (perspective: the direction sample maps to pixels; no aperture sample)
__device__ __noinline__ int sample_sensor(Ctx &c, BV &v0, int px, int py, BE &e0, BV &v1, BE &e1, BV &v2)
{
    be_clear(e0); bv_clear(v1);
    const Float sx = c.rng.next1D(), sy = c.rng.next1D();
    v1.p = c.cam.pos; v1.n = c.cam.dir; v1.object = -2;
    v0.w[ERadiance] = mk(1.0); v0.pdf[ERadiance] = 1.0; v0.measure = M_DISCRETE; v0.rr = 1.0;
    const CameraD &cam = c.S->cam;
    if (cam.thinlens) {                                                                    // the aperture sample (:324-325) and ThinLensCamera::samplePosition, thinlens.cpp:363-376
        const Float ax = c.rng.next1D(), ay = c.rng.next1D();
        const Float r1 = 2.0 * ax - 1.0, r2 = 2.0 * ay - 1.0;                              // squareToUniformDiskConcentric, warp.cpp:81-102
        Float phi, r;
        if (r1 == 0 && r2 == 0) { r = phi = 0; }
        else if (r1 * r1 > r2 * r2) { r = r1; phi = (GD_PI / 4.0) * (r2 / r1); }
        else { r = r2; phi = (GD_PI / 2.0) - (r1 / r2) * (GD_PI / 4.0); }
        const d3 ol = mk(r * cos(phi) * cam.apertureRadius, r * sin(phi) * cam.apertureRadius, 0.0);
        v1.p = mk(cam.m[0] * ol.x + cam.m[1] * ol.y + cam.m[2] * ol.z + cam.m[3], cam.m[4] * ol.x + cam.m[5] * ol.y + cam.m[6] * ol.z + cam.m[7],
                  cam.m[8] * ol.x + cam.m[9] * ol.y + cam.m[10] * ol.z + cam.m[11]);
        v0.pdf[ERadiance] = c.cam.aperturePdf; v0.measure = M_AREA;
    }
    v1.type = T_SENSOR_SAMPLE; v1.degenerate = 0;
    e0.tr[ERadiance] = 1.0;
    const Float spx = (px + sx) * cam.invW, spy = (py + sy) * cam.invH;
    v1.u = spx * cam.width; v1.v = spy * cam.height;
    d3 dl = sample_to_camera_dir(c, spx, spy);
    if (cam.thinlens) {                                                                    // ThinLensCamera::sampleDirection, thinlens.cpp:386-418: through the pixel's point of the focus plane
        d3 nearP = mk((1 - 2 * spx) * cam.nearClip * cam.tanHalf, (1 - 2 * spy) / cam.aspect * cam.nearClip * cam.tanHalf, cam.nearClip);
        nearP.x = nearP.x * (cam.focusDistance / nearP.z);
        nearP.y = nearP.y * (cam.focusDistance / nearP.z);
        nearP.z = cam.focusDistance;
        dl = normalize(nearP - cam_to_local_point(c, v1.p));
    }
    const d3 d = cam_to_world(c, dl);
    const Float dpdf = c.cam.normalization / (dl.z * dl.z * dl.z);
    be_clear(e1); bv_clear(v2);
    v1.w[EImportance] = mk(1.0) * dpdf * (1.0 / fabs(dot(d, v1.n)));
    v1.w[ERadiance] = mk(1.0);
    v1.pdf[EImportance] = 1.0; v1.pdf[ERadiance] = dpdf;
    v1.rr = 1.0;
    v1.measure = M_SOLID;
    if (!edge_extend(c, e1, v1.p, d, v2, ERadiance, false, 0.0)) { v1.measure = M_INVALID; return 1; }
    v1.measure = M_AREA;
    v1.pdf[ERadiance] /= e1.length * e1.length;
    if (bv_on_surface(v2)) v1.pdf[ERadiance] *= fabs(dot(d, bv_geo_normal(c, v2)));
    return 2;
}
// PathVertex::perturbDirection, vertex.cpp:488-679 (mode is always ERadiance on the G-BDPT path: the sensor sample is what gets perturbed;
// the surface branch belongs to propagatePerturbation through glossy chains, which connectable-only scenes never enter)
__device__ bool perturb_direction(Ctx &c, BV &v, const BV &pred, const BE &predEdge, BE &succEdge, BV &succ, d3 d, Float dist)
{
    be_clear(succEdge); bv_clear(succ);
    if (v.degenerate) return false;
    if (v.type != T_SENSOR_SAMPLE) return false;
    const Float value = sensor_direction(c, v.p, d), prob = value;
    if (value == 0 || prob <= 0x1p-1024) return false;
    v.w[EImportance] = mk(value) * (1.0 / fabs(dot(d, v.n)));
    v.w[ERadiance] = mk(value) / prob;
    v.pdf[EImportance] = 1.0;
    v.pdf[ERadiance] = prob;
    v.measure = M_SOLID;
    if (!edge_extend(c, succEdge, v.p, d, succ, ERadiance, true, dist)) { v.measure = M_INVALID; return false; }
    to_area(c, v, ERadiance, pred, predEdge, succEdge, succ, d);
    return true;
}
// PathVertex::eval, vertex.cpp:781-913
__device__ d3 bv_eval(const Ctx &c, const BV &v, const BV *pred, const BV *succ, int mode, int measure = M_AREA)
{
    if (v.type == T_EMITTER_SUPER) {
        if (mode != EImportance || pred != nullptr || succ->type != T_EMITTER_SAMPLE) return mk(0.0);
        if (succ->prim == BV_OFF_SURFACE) return measure == M_DISCRETE ? c.V.emitters[succ->object].radiance * (4 * GD_PI) : mk(0.0);   // PointEmitter::evalPosition, point.cpp:89-91
        if (is_envmap(c, succ->object)) return mk(envmap_power(c) * (1 / (4 * GD_PI * c.S->bsRadius * c.S->bsRadius)));   // EnvironmentMap::evalPosition, envmap.cpp:424-426
        return c.V.emitters[succ->object].radiance * GD_PI;
    } else if (v.type == T_SENSOR_SUPER) {
        if (mode != ERadiance || pred != nullptr || succ->type != T_SENSOR_SAMPLE) return mk(0.0);
        if (c.S->cam.thinlens) return mk(measure == M_AREA ? c.cam.aperturePdf : 0.0);     // thinlens.cpp:378-380
        return mk(measure == M_DISCRETE ? 1.0 : 0.0);
    } else if (v.type == T_EMITTER_SAMPLE || v.type == T_SENSOR_SAMPLE) {
        const bool emitter = v.type == T_EMITTER_SAMPLE;
        const int fwd = emitter ? EImportance : ERadiance, superT = emitter ? T_EMITTER_SUPER : T_SENSOR_SUPER;
        d3 target;
        if (mode == fwd && pred->type == superT) target = succ->p;
        else if (mode == 1 - fwd && succ->type == superT) target = pred->p;
        else return mk(0.0);
        const d3 wo = normalize(target - v.p);
        const int dm = measure == M_AREA ? M_SOLID : measure;
        d3 result = emitter ? emitter_eval_direction(c, v, wo, dm) : mk(dm != M_SOLID ? 0.0 : sensor_direction(c, v.p, wo));
        const Float dp = fabs(dot(v.n, wo));
        if (measure != M_DISCRETE && dp != 0) result = result / dp;
        return result;
    } else if (v.type == T_SURFACE) {
        const Surf sf = surf_of(c, v);
        const d3 wi = normalize(pred->p - v.p), wo = normalize(succ->p - v.p);
        const d3 wiL = toLocal(sf.fr, wi), woL = toLocal(sf.fr, wo);
        if (measure == M_AREA) measure = M_SOLID;
        d3 result; Float pdfUnused;
        bsdf_eval_pdf(sf.m, sf.R, wiL, woL, bsdf_measure(measure), result, pdfUnused);
        const Float wiDotGeoN = dot(sf.geoN, wi), woDotGeoN = dot(sf.geoN, wo);
        if (wiDotGeoN * wiL.z <= 0 || woDotGeoN * woL.z <= 0) return mk(0.0);
        if (mode == EImportance) result = result * fabs((wiL.z * woDotGeoN) / (woL.z * wiDotGeoN));
        if (measure != M_DISCRETE && woL.z != 0) result = result / fabs(woL.z);
        return result;
    }
    return mk(0.0);
}
// PathVertex::evalPdf, vertex.cpp:915-1022
__device__ Float bv_eval_pdf(const Ctx &c, const BV &v, const BV *pred, const BV *succ, int mode, int measure = M_AREA)
{
    d3 wo = mk(0.0);
    Float dist = 0.0, result = 0.0;
    if (v.type == T_EMITTER_SUPER) {
        if (mode != EImportance || pred != nullptr || succ->type != T_EMITTER_SAMPLE) return 0.0;
        return pdf_emitter_position(c, succ->object, measure);
    } else if (v.type == T_SENSOR_SUPER) {
        if (mode != ERadiance || pred != nullptr || succ->type != T_SENSOR_SAMPLE) return 0.0;
        if (c.S->cam.thinlens) return measure == M_AREA ? c.cam.aperturePdf : 0.0;         // thinlens.cpp:382-384
        return measure == M_DISCRETE ? 1.0 : 0.0;
    } else if (v.type == T_EMITTER_SAMPLE) {
        if (mode == ERadiance && succ->type == T_EMITTER_SUPER) return 1.0;
        else if (mode != EImportance || pred->type != T_EMITTER_SUPER) return 0.0;
        wo = succ->p - v.p;
        dist = len(wo); wo = wo / dist;
        result = emitter_direction(c, v, wo, measure == M_AREA ? M_SOLID : measure);
    } else if (v.type == T_SENSOR_SAMPLE) {
        if (mode == EImportance && succ->type == T_SENSOR_SUPER) return 1.0;
        else if (mode != ERadiance || pred->type != T_SENSOR_SUPER) return 0.0;
        wo = succ->p - v.p;
        dist = len(wo); wo = wo / dist;
        result = (measure == M_AREA ? M_SOLID : measure) != M_SOLID ? 0.0 : sensor_direction(c, v.p, wo);
    } else if (v.type == T_SURFACE) {
        const Surf sf = surf_of(c, v);
        wo = succ->p - v.p;
        dist = len(wo); wo = wo / dist;
        const d3 wi = normalize(pred->p - v.p);
        const d3 wiL = toLocal(sf.fr, wi), woL = toLocal(sf.fr, wo);
        d3 fUnused;
        bsdf_eval_pdf(sf.m, sf.R, wiL, woL, bsdf_measure(measure == M_AREA ? M_SOLID : measure), fUnused, result);
        const Float wiDotGeoN = dot(sf.geoN, wi), woDotGeoN = dot(sf.geoN, wo);
        if (wiDotGeoN * wiL.z <= 0 || woDotGeoN * woL.z <= 0) return 0.0;
    } else return 0.0;
    if (measure == M_AREA) {
        result /= dist * dist;
        if (bv_on_surface(*succ)) result *= fabs(dot(wo, bv_geo_normal(c, *succ)));
    }
    return result;
}
// PathVertex::cast to EEmitterSample, vertex.cpp:1115-1135 (PositionSamplingRecord(its): n = its.shFrame.n, records.inl:154-155)
__device__ bool bv_cast_emitter(const Ctx &c, BV &v)
{
    if (v.type == T_EMITTER_SAMPLE) return true;
    if (v.type != T_SURFACE) return false;
    const int em = prim_emitter(c, v.prim);
    if (em < 0) return false;
    v.n = bv_sh_normal(c, v);
    v.type = T_EMITTER_SAMPLE;
    v.object = em;
    v.measure = M_AREA;
    v.degenerate = 0;
    return true;
}
// PathVertex::update, vertex.cpp:1165-1211
__device__ bool bv_update(const Ctx &c, BV &v, const BV *pred, const BV *succ, int mode, int measure)
{
    v.pdf[mode] = bv_eval_pdf(c, v, pred, succ, mode, measure);
    v.pdf[1 - mode] = bv_eval_pdf(c, v, succ, pred, 1 - mode, measure);
    v.w[mode] = bv_eval(c, v, pred, succ, mode, measure);
    v.w[1 - mode] = bv_eval(c, v, succ, pred, 1 - mode, measure);
    if (is_zero(v.w[mode]) || v.pdf[mode] <= 0x1p-1024) return false;
    Float weightFwd = v.pdf[mode] <= 0x1p-1024 ? 0.0 : 1 / v.pdf[mode], weightBkw = v.pdf[1 - mode] <= 0x1p-1024 ? 0.0 : 1 / v.pdf[1 - mode];
    v.measure = measure;
    if (!bv_super(v) && measure == M_AREA) {
        const d3 shN = bv_sh_normal(c, v);
        if (!bv_super(*pred)) {
            d3 d = pred->p - v.p;
            const Float invDistSqr = 1.0 / len2(d);
            weightBkw *= invDistSqr;
            d = d * sqrt(invDistSqr);
            if (bv_on_surface(v) && bv_connectable(v)) weightBkw *= fabs(dot(shN, d));
            if (bv_on_surface(*pred)) weightBkw *= fabs(dot(bv_geo_normal(c, *pred), d));
        }
        if (!bv_super(*succ)) {
            d3 d = succ->p - v.p;
            const Float invDistSqr = 1.0 / len2(d);
            weightFwd *= invDistSqr;
            d = d * sqrt(invDistSqr);
            if (bv_on_surface(v) && bv_connectable(v)) weightFwd *= fabs(dot(shN, d));
            if (bv_on_surface(*succ)) weightFwd *= fabs(dot(bv_geo_normal(c, *succ), d));
        }
        if (v.type == T_SURFACE) v.componentType = ESmooth;
    }
    v.w[mode] = v.w[mode] * weightFwd;
    v.w[1 - mode] = v.w[1 - mode] * weightBkw;
    return true;
}
// PathVertex::connect with explicit measures, vertex.cpp:1348-1370 (no sensor shapes: a connection into the sensor supernode cannot occur here)
__device__ bool bv_connect(Ctx &c, const BV *pred, BV &vs, BE &edge, BV &vt, const BV *succ, int vsMeasure, int vtMeasure, bool knownVisible = false)
{
    if (vs.type == T_EMITTER_SUPER) { if (!bv_cast_emitter(c, vt)) return false; }
    else if (vt.type == T_SENSOR_SUPER) return false;
    if (!bv_update(c, vs, pred, &vt, EImportance, vsMeasure)) return false;
    if (!bv_update(c, vt, succ, &vs, ERadiance, vtMeasure)) return false;
    return edge_connect(c, edge, vs, vt, knownVisible);
}

// ---- offset paths: ManifoldPerturbation::generateOffsetPathGBDPT, mut_manifold.cpp:806-936, for a chain a - b - c of adjacent vertices -------
struct Offset {
    BV a, b, c;                 // the perturbed sensor sample, the new first vertex, the clone of c re-connected to it
    BE eab, ebc;                // proposal.edge(a - 1), proposal.edge(q)
    int success, couldConnectAfterB;
    Float jacobian;             // halfJacobian_GBDPT(offset) / halfJacobian_GBDPT(base), path.cpp:380-394 (G(a-1,a)/G(b,a) and det are 1 for this chain)
};
// source chain: srcC - srcB - srcA - S0 (towards the sensor), predC = the vertex before c (nullptr if c is the emitter supernode);
// eSA = the edge between the sensor sample and the sensor supernode; distAB = source.edge(a - 1)->length
__device__ bool generate_offset(Ctx &c, const BV &srcA, const BV &S0, const BE &eSA, const BV &srcB, const BV &srcC, const BV *predC, Float distAB,
                                Float shX, Float shY, bool lightPath, Offset &o)
{
    o.success = 0; o.couldConnectAfterB = 0; o.jacobian = 1.0;
    if (!bv_connectable(srcA)) return false;
    o.a = srcA; o.c = srcC;                                                                // the deep copies of a and c, :863-864
    // perturbDirection, mut_manifold.cpp:938-986
    const CameraD &cam = c.S->cam;
    const Float ppx = srcA.u + shX, ppy = srcA.v + shY;                                     // source.getSamplePosition() + offset
    const d3 rd = cam_to_world(c, centre_ray_dir(c, ppx * cam.invW, ppy * cam.invH));        // sensor->sampleRay, perspective.cpp:249-269
    const Float focusDistance = focus_distance(cam) / fabs(dot(c.cam.dir, rd));
    const d3 d = normalize((c.cam.pos + rd * focusDistance) - srcA.p);
    if (!perturb_direction(c, o.a, S0, eSA, o.eab, o.b, d, distAB)) return false;
    if (!bv_connectable(o.b)) return false;
    o.couldConnectAfterB = bv_connect(c, predC, o.c, o.ebc, o.b, &o.a, bv_connectable(srcC) ? M_AREA : M_DISCRETE, bv_connectable(srcB) ? M_AREA : M_DISCRETE);
    if (lightPath && !o.couldConnectAfterB) return false;
    sensor_sample_position(c, o.a.p, o.b.p - o.a.p, o.a.u, o.a.v);                                 // updateSamplePosition, :912-913
    o.a.rr = srcA.rr; o.b.rr = srcB.rr; o.c.rr = srcC.rr;                                   // :918-923
    if (o.b.type == T_SURFACE && o.b.componentType == 0) o.b.componentType = srcB.componentType;
    Float jy = 1.0; jy /= o.a.pdf[ERadiance];
    Float jx = 1.0; jx /= srcA.pdf[ERadiance];
    // (b of a light path with s = 1 is the emitter sample.  A `point` emitter's sample has the shading normal 0 (point.cpp:82), so Path::G(a - 1, a) == G(b, a) == 0
    //  and the reference's quotient is 0 / 0: it warns "Invalid Path::halfJacobian" and carries the NaN into the gradient, which the film then drops as an invalid
    //  put.  Only reached when the shifted ray happens to end on an AREA light, so that the offset path connects at all.)
    if (srcB.type == T_EMITTER_SAMPLE && srcB.prim == BV_OFF_SURFACE) jx *= __builtin_nan("");
    o.jacobian = jy / jx;
    o.success = 1;
    return true;
}

// ---- the sample --------------------------------------------------------------------------------------------------------------------------
struct Sample {
    BV X[NSV], Y[NEV];          // sensor / emitter subpath vertices
    BE EX[NSV], EY[NEV];        // EX[i]: edge between X[i] and X[i+1]; EY likewise
    int nX, nY;                 // vertex counts
    BV XTc, Y1c;                // connectPath's clones: the last sensor vertex and the emitter vertex it is connected to (createShiftablePath, gbdpt_proc.cpp:600-662)
    BE eConn;                   // the edge between them
    int connS;                  // 1: connected to Y[1]; 0: the last sensor vertex lies on an emitter and is connected to the emitter supernode
    Offset off[4];
    // combineImportanceData / combineRadianceData (gbdpt_proc.cpp:544-565): products of weights / densities along the emitter subpath and
    // along the sensor subpath and each of its four offsets, up to vertex i
    d3 impW[NEV]; Float impP[NEV];
    d3 radW[5][NSV]; Float radP[5][NSV];
    Float posX, posY;           // the sample's film position (sensorSubpath[0].vertex(1)->getSamplePosition())
};

// vertex i of "sensorSubpath[k]" (gbdpt_proc.cpp:224: the reversed proposal): base records except the three the shift replaced, and the
// clone that createShiftablePath put at the connection end
__device__ __forceinline__ const BV &SV(const Sample &sm, int k, int i)
{
    if (k == 0) return sm.X[i];
    const Offset &o = sm.off[k - 1];
    if (i == 1) return o.a;
    if (i == 2) return o.b;
    if (i == 3) return o.c;
    if (i == sm.nX - 1) return sm.XTc;
    return sm.X[i];
}
__device__ __forceinline__ const BE &SE(const Sample &sm, int k, int i)                     // edge between SV(k, i) and SV(k, i + 1)
{
    if (k == 0) return sm.EX[i];
    const Offset &o = sm.off[k - 1];
    if (i == 1) return o.eab;
    if (i == 2) return o.ebc;
    return sm.EX[i];
}

// the densities of Path::miWeight*NoSweep_GBDPT (path.cpp:99-132,264-307) for the path Y[0..s] + connection + SV(k, t..0), pdfImp / pdfRad[0..s+t+1]
struct PathRef {
    const Sample *sm; int k;             // sensor side: sensorSubpath[k]
    const BV *ev; const BE *ee; int ne;  // emitter side: the base emitter subpath, or (light paths) a patched copy given by the overrides below
    const BV *ovS, *ovSm1; const BE *ovEm1;   // overrides for emitter vertices s and s - 1 and edge s - 1 (offset light paths), or nullptr
    const BV *ovT;                       // override for the sensor vertex t (the perturbed sensor sample of an offset light path), or nullptr
};
__device__ __forceinline__ const BV &PE(const PathRef &p, int i, int s) { if (p.ovS && i == s) return *p.ovS; if (p.ovSm1 && i == s - 1) return *p.ovSm1; return p.ev[i]; }
__device__ __forceinline__ const BE &PEe(const PathRef &p, int i, int s) { if (p.ovEm1 && i == s - 1) return *p.ovEm1; return p.ee[i]; }
__device__ __forceinline__ const BV &PS(const PathRef &p, int i, int t) { if (p.ovT && i == t) return *p.ovT; return SV(*p.sm, p.k, i); }

// The densities of the reference are two arrays pdfImp[0..n], pdfRad[0..n] (n = s + t + 1) per path; strategy p has the density
// value[p] = pdfImp[1] .. pdfImp[p] * pdfRad[p + 1] .. pdfRad[n - 1].  The first wavefront form kept those arrays (and the prefix / suffix products)
// per lane: 1.5 KB of dynamically indexed scratch per connection, written and read back -- the PMC pass of k_bd_connect showed 64 GB written and
// 103 GB read PER LAUNCH, 7.6 KB per connection, 82 % of the wave cycles waiting.  Nothing needs the arrays: both weights are ratios of SUMS of
// strategy densities, and a sum over p of (prefix product) x (suffix product) is a Horner recurrence over the entries in reverse order,
//     R(p) = pdfRad[p + 1] R(p + 1),  G(p) = a_p R(p) + pdfImp[p + 1] G(p + 1),  G(n - 1) = a_(n-1), R(n - 1) = 1   =>   sum_p a_p value[p] = G(0)
// (a_p = 1 for an allowed strategy), and the same with squared entries for the power heuristic.  The entries are read straight from the
// sample's record as the recurrence walks the path; the four densities evaluated AT the connection are the only ones computed.
struct ConnPdfs { Float impS, impT, radS, radT; };      // pdfImp[s + 1], pdfImp[s + 2], pdfRad[s - 1], pdfRad[s] (path.cpp:99-132,264-307)
__device__ ConnPdfs conn_pdfs(const Ctx &c, const PathRef &p, const BE &connectionEdge, int s, int t)
{
    const BV *vsPred = s > 0 ? &PE(p, s - 1, s) : nullptr, *vtPred = t > 0 ? &PS(p, t - 1, t) : nullptr;
    const BV &vs = PE(p, s, s), &vt = PS(p, t, t);
    ConnPdfs cp;
    cp.impS = bv_eval_pdf(c, vs, vsPred, &vt, EImportance, M_AREA) * connectionEdge.tr[EImportance];
    cp.impT = t > 0 ? bv_eval_pdf(c, vt, &vs, vtPred, EImportance, M_AREA) * SE(*p.sm, p.k, t - 1).tr[EImportance] : 0.0;
    cp.radS = s > 0 ? bv_eval_pdf(c, vs, &vt, vsPred, ERadiance, M_AREA) * PEe(p, s - 1, s).tr[ERadiance] : 0.0;
    cp.radT = bv_eval_pdf(c, vt, vtPred, &vs, ERadiance, M_AREA) * connectionEdge.tr[ERadiance];
    return cp;
}
// entry i (1 <= i <= n - 1) of the two arrays
__device__ __forceinline__ Float mis_imp(const PathRef &p, const ConnPdfs &cp, int s, int t, int i)
{
    if (i <= s) return PE(p, i - 1, s).pdf[EImportance] * PEe(p, i - 1, s).tr[EImportance];
    if (i == s + 1) return cp.impS;
    if (i == s + 2) return cp.impT;
    const int v = t + s + 2 - i;                                                            // sensor vertex t - 1 .. 1
    return PS(p, v, t).pdf[EImportance] * SE(*p.sm, p.k, v - 1).tr[EImportance];
}
__device__ __forceinline__ Float mis_rad(const PathRef &p, const ConnPdfs &cp, int s, int t, int i)
{
    if (i <= s - 2) return PE(p, i + 1, s).pdf[ERadiance] * PEe(p, i, s).tr[ERadiance];
    if (i == s - 1) return cp.radS;
    if (i == s) return cp.radT;
    const int v = t + s + 1 - i;                                                            // sensor vertex t .. 1
    return PS(p, v - 1, t).pdf[ERadiance] * SE(*p.sm, p.k, v - 1).tr[ERadiance];
}
// NOTE on path.cpp:143-167,309-349 (area densities next to a non-connectable vertex converted to projected solid angle): the loops run over
// i in [1, k-3] / [3, k-1] and fire only where connectableStrict[i] && !connectableStrict[i +- 1]; with every surface vertex connectable
// the only non-connectable vertex is the sensor supernode (index k), outside both ranges: nothing to convert.

// sum over the allowed strategies of value[p] (sum1) and of value[p]^2 (sum2, base path only), and value[s] = the density of the strategy in use
template <bool SQUARES>
__device__ __forceinline__ void mis_sums(const PathRef &p, const ConnPdfs &cp, int s, int t, unsigned allowed, Float &valueS, Float &sum1, Float &sum2)
{
    const int n = s + t + 1;
    Float R = 1.0, G = (allowed >> (n - 1)) & 1u ? 1.0 : 0.0, R2 = 1.0, G2 = G, vS = 1.0;
    for (int q = n - 2; q >= 0; --q) {
        const int i = q + 1;
        const Float im = mis_imp(p, cp, s, t, i), ra = mis_rad(p, cp, s, t, i);
        vS *= i <= s ? im : ra;
        R *= ra;
        const bool a = (allowed >> q) & 1u;
        G = (a ? R : 0.0) + im * G;
        if (SQUARES) { R2 *= ra * ra; G2 = (a ? R2 : 0.0) + (im * im) * G2; }
    }
    valueS = vS; sum1 = G; sum2 = G2;
}

// Path::miWeightBaseNoSweep_GBDPT (path.cpp:49-201; exponent 2, geomTerm 1) and miWeightGradNoSweep_GBDPT (:204-378; exponent 1), restructured:
// the reference rebuilds the base path's densities for each of the five paths of a connection and forms every strategy's density p_i by an
// O(n) product (O(n^2) per weight).  Here each path's sums are one O(n) recurrence, the base path's are formed ONCE per connection and kept for
// the four gradient weights (sum_p (b_p + j o_p) = sum_p b_p + j sum_p o_p) -- same factors, another association (a few ulp).
struct MisBase { Float valueS, sum1; ConnPdfs cp; unsigned allowed; int n; };
__device__ Float mi_weight_base(const Ctx &c, const Sample &sm, const PathRef &base, const BE &baseEdge, int s, int t, MisBase &mb)
{
    const int k = s + t + 1;
    mb.n = k; mb.allowed = 0;
    const bool lightImage = c.cfg.lightImage != 0;
    for (int p = 0; p < k; ++p) {
        const int tPrime = k - p - 1;
        // connectable[] of the BASE path (path.cpp:76-95,241-260): position p is emitter vertex p (p <= s) or sensor vertex k - p
        const BV &vp = p <= s ? sm.Y[p] : sm.X[k - p], &vp1 = p + 1 <= s ? sm.Y[p + 1] : sm.X[k - p - 1];
        if (connectable_gbdpt(c, vp) && connectable_gbdpt(c, vp1) && (lightImage || tPrime > 1)) mb.allowed |= 1u << p;
    }
    mb.cp = conn_pdfs(c, base, baseEdge, s, t);
    Float sum2;
    mis_sums<true>(base, mb.cp, s, t, mb.allowed, mb.valueS, mb.sum1, sum2);
    return (Float)((mb.valueS * mb.valueS) / sum2);                                         // pow(p_i * 1, 2.0); tPrime == t <=> p == s
}
__device__ Float mi_weight_grad(const Ctx &c, const MisBase &mb, const PathRef &offset, const BE &offsetEdge, int s, int t, Float jDet)
{
    const ConnPdfs cp = conn_pdfs(c, offset, offsetEdge, s, t);
    Float oS, o1, o2;
    mis_sums<false>(offset, cp, s, t, mb.allowed, oS, o1, o2);
    return (Float)(mb.valueS / (mb.sum1 + o1 * jDet));                                       // pow(x, 1.0) == x
}
// The gradient weight of a connection whose sensor vertex t lies beyond the shifted part of the offset path (5 <= t < the last sensor vertex):
// the offset path shares vs, vt, their predecessors and the connection edge with the base path, so the four densities evaluated at the
// connection are the base path's (no BSDF call) and only the entries of sensor vertices 0..3 differ, which the walk reads from the offset's record.
__device__ __forceinline__ bool shares_connection(const Sample &sm, int t) { return t >= 5 && t < sm.nX - 1; }
__device__ Float mi_weight_grad_shared(const MisBase &mb, const PathRef &base, int k, int s, int t, Float jDet)
{
    PathRef off = base;
    off.k = k;
    Float oS, o1, o2;
    mis_sums<false>(off, mb.cp, s, t, mb.allowed, oS, o1, o2);
    return (Float)(mb.valueS / (mb.sum1 + o1 * jDet));
}

struct LightSplat { Float x, y; int buffer; d3 value; };
struct SampleOut { d3 primal, gradient[4]; Float posX, posY; int nLight; LightSplat light[BD_MAX_LIGHT]; };
struct PairOut { d3 primal, gradient[4]; int nLight; LightSplat light[5]; };   // one connection (s, t): t >= 2 adds to the sample's sums, t == 1 splats

// First half of GBDPTRenderer::process (the sample loop body, gbdpt_proc.cpp:152-229): the two subpaths, the connected base path, its four
// offset paths, the prefix products.  Everything the connections need is in `sm` afterwards (which may live in HBM: the wavefront form).
// (two functions: as one, the allocator takes 256 VGPRs + 70 AGPRs for it and the walk kernel runs at one wave per SIMD; a register budget
// cannot be attached to a device function, and inlining it into the kernel crashes the backend)
__device__ __noinline__ bool walk_paths(Ctx &c, Sample &sm, int px, int py)
{
    const BdConfig &cfg = c.cfg;
    const int emitterDepth = cfg.maxDepth + (c.S->cam.thinlens ? 1 : 0), sensorDepth = cfg.maxDepth + (cfg.hittableEmitters ? 1 : 0);   // :110-122: one more emitter step unless the sensor is a point (pinhole), one more sensor step if an emitter can be hit
    // ---- Path::alternatingRandomWalkFromPixel, path.cpp:548-631 ----
    bv_clear(sm.X[0]); sm.X[0].type = T_SENSOR_SUPER; sm.X[0].degenerate = 1;               // makeEndpoint, vertex.cpp:27-33
    bv_clear(sm.Y[0]); sm.Y[0].type = T_EMITTER_SUPER; sm.Y[0].degenerate = 0;
    sm.nX = 1; sm.nY = 1;
    int t = sample_sensor(c, sm.X[0], px, py, sm.EX[0], sm.X[1], sm.EX[1], sm.X[2]);
    sm.nX = 1 + t;
    bool walkT = t == 2, walkS = true;
    d3 thrS = mk(1.0), thrT = mk(1.0);
    int s = 0;
    do {
        if (walkT && t < sensorDepth) {
            if (sample_next(c, sm.X[t], &sm.X[t - 1], &sm.EX[t - 1], sm.EX[t], sm.X[t + 1], ERadiance, cfg.rrDepth != -1 && t >= cfg.rrDepth, thrT)) { t++; sm.nX++; }
            else walkT = false;
        } else walkT = false;
        if (walkS && s < emitterDepth) {
            if (sample_next(c, sm.Y[s], s > 0 ? &sm.Y[s - 1] : nullptr, s > 0 ? &sm.EY[s - 1] : nullptr, sm.EY[s], sm.Y[s + 1], EImportance, cfg.rrDepth != -1 && s >= cfg.rrDepth, thrS)) { s++; sm.nY++; }
            else walkS = false;
        } else walkS = false;
    } while (walkS || walkT);
    sm.posX = sm.X[1].u; sm.posY = sm.X[1].v;
    for (int k = 0; k < 4; k++) { sm.off[k].success = 0; sm.off[k].couldConnectAfterB = 0; sm.off[k].jacobian = 1.0; }
    if (sm.nY < 2) { sm.nY = 0; return false; }                                             // (no emitter could be sampled: a scene without power; no connections)
    return true;
}
// walk_shift in two stages, so that the frame kernels can run them on different lanes (round 5): the base stage (one lane per sample: the connected base
// path, the emitter side's prefix products, the base path's own) and ONE offset path (one lane per sample and offset: the four depend on the connected base
// path only, and each writes its own record and products).  Returns whether the sample has offset paths at all.
__device__ __noinline__ bool walk_shift_base(Ctx &c, Sample &sm)
{
    // ---- createShiftablePath(connectPath, emitterSubpath, sensorSubpath, 1, last), gbdpt_proc.cpp:600-662 ----
    const int T = sm.nX - 1;
    sm.connS = 1;
    if (sm.X[T].type == T_SURFACE && prim_emitter(c, sm.X[T].prim) >= 0) sm.connS = 0;
    sm.Y1c = sm.Y[sm.connS];
    sm.XTc = sm.X[T];
    be_clear(sm.eConn);
    bv_cast_emitter(c, sm.XTc);
    bv_connect(c, sm.connS == 1 ? &sm.Y[0] : nullptr, sm.Y1c, sm.eConn, sm.XTc, T >= 1 ? &sm.X[T - 1] : nullptr, bv_connectable(sm.Y1c) ? M_AREA : M_DISCRETE, bv_connectable(sm.XTc) ? M_AREA : M_DISCRETE);
    if (T == 1) sensor_sample_position(c, sm.XTc.p, sm.Y1c.p - sm.XTc.p, sm.XTc.u, sm.XTc.v);          // (the clone's film position: never read again)
    // ---- combineImportanceData / combineRadianceData, gbdpt_proc.cpp:544-565 ----
    const int nE = sm.nY, nS = sm.nX;
    sm.impW[0] = mk(1.0); sm.impP[0] = 1.0;
    for (int i = 1; i < nE; ++i) {
        sm.impW[i] = sm.impW[i - 1] * sm.Y[i - 1].w[EImportance] * sm.Y[i - 1].rr * sm.EY[i - 1].tr[EImportance];
        sm.impP[i] = sm.impP[i - 1] * sm.Y[i - 1].pdf[EImportance] * sm.Y[i - 1].rr * sm.EY[i - 1].tr[EImportance];
    }
    for (int k = 0; k <= 4; k++) {
        sm.radW[k][0] = mk(1.0); sm.radP[k][0] = 1.0;
        for (int i = 1; i < nS; ++i) { sm.radW[k][i] = mk(0.0); sm.radP[k][i] = 0.0; }
    }
    for (int i = 1; i < nS; ++i) {
        const BV &pv = SV(sm, 0, i - 1);
        const BE &pe = SE(sm, 0, i - 1);
        sm.radW[0][i] = sm.radW[0][i - 1] * pv.w[ERadiance] * pv.rr * pe.tr[ERadiance];
        sm.radP[0][i] = sm.radP[0][i - 1] * pv.pdf[ERadiance] * pv.rr * pe.tr[ERadiance];
    }
    // connectPath = [Y0 (connS = 1), Y1c, XTc, X[T-1], ..., X[2], X[1], X[0]]: a = X[1], b = X[2] (or XTc if T == 2), c = X[3] (XTc if T == 3, Y1c if T == 2)
    // muRec.extra[0] = a <= 2 (gbdpt_proc.cpp:200) <=> the connected path has at most four vertices: T + connS < 3
    return T + sm.connS >= 3 && T >= 2;
}
__device__ __noinline__ void walk_shift_offset(Ctx &c, Sample &sm, int k)                   // k = 0..3: sm.off[k], radW / radP[k + 1]
{
    const Float shifts[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};                           // :101,265
    const int T = sm.nX - 1, nS = sm.nX;
    const BV &srcB = T == 2 ? sm.XTc : sm.X[2];
    const BV &srcC = T == 2 ? sm.Y1c : (T == 3 ? sm.XTc : sm.X[3]);
    const BV *predC = T == 2 ? &sm.Y[0] : (T == 3 ? &sm.Y1c : (T == 4 ? &sm.XTc : &sm.X[4]));
    generate_offset(c, sm.X[1], sm.X[0], sm.EX[0], srcB, srcC, predC, sm.EX[1].length, shifts[k][0], shifts[k][1], false, sm.off[k]);
    if (!sm.off[k].success) return;
    // sensorSubpath[k + 1] has vertexCount = nS + connS + 1 >= nS entries for a successful shift
    for (int i = 1; i < nS; ++i) {
        const BV &pv = SV(sm, k + 1, i - 1);
        const BE &pe = SE(sm, k + 1, i - 1);
        sm.radW[k + 1][i] = sm.radW[k + 1][i - 1] * pv.w[ERadiance] * pv.rr * pe.tr[ERadiance];
        sm.radP[k + 1][i] = sm.radP[k + 1][i - 1] * pv.pdf[ERadiance] * pv.rr * pe.tr[ERadiance];
    }
}
__device__ __noinline__ void walk_shift(Ctx &c, Sample &sm)
{
    if (walk_shift_base(c, sm))
        for (int k = 0; k < 4; k++) walk_shift_offset(c, sm, k);
}
__device__ __forceinline__ void walk_sample(Ctx &c, Sample &sm, int px, int py) { if (walk_paths(c, sm, px, py)) walk_shift(c, sm); }
// the range of sensor vertices connected to emitter vertex s (gbdpt_proc.cpp:311-319)
__device__ __forceinline__ void pair_range(const BdConfig &cfg, int nS, int s, int &minT, int &maxT)
{
    minT = max(2 - s, cfg.lightImage ? 1 : 2);
    maxT = min(nS - 1, cfg.maxDepth + 1 - s);
}

// One connection (s, t) of GBDPTRenderer::evaluate (gbdpt_proc.cpp:319-527): the base path and its four offsets.  Reads `sm` only.
// Returns false when the connection contributes nothing.
// CLS: the item class the connection belongs to -- 0: light tracing (t == 1), 1: sensor vertex t inside or at the end of the shifted part, 2: beyond it
// (shares_connection).  A compile-time class lets each build drop the other classes' locals (the cloned end vertices and the transient offset of a
// light path are 1.2 KB of vertex records that live in scratch because they are passed by reference).
// PHASE: 0 = the whole connection (the probe entry); 3 = the part of the base path that needs no visibility ray of the connection (end points
// connectable, facing each other, non-zero throughput; for light tracing: the sensor connection): a filter in front of phase 1; 1 = the base path only: returns whether it carries anything and its primal term -- most
// connections end here (blocked, back-facing, zero throughput), and a wave in which one lane goes on to the four offsets while the others wait ran
// at 17 % lane utilisation; 2 = the offsets of a connection that survived phase 1: the base path is evaluated again for the state the offsets
// share with it (its rays are not counted twice), the gradients are the output.  Phases 1 and 2 run as two launches with the survivors compacted
// in between.
template <int CLS, int PHASE = 0>
__device__ bool connect_pair(Ctx &c, const Sample &sm, int s, int t, PairOut &po)
{
    constexpr bool T1 = CLS == 0;
    const unsigned nClosest0 = c.nClosest, nShadow0 = c.nShadow;
    const Float shifts[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};
    const int vert_b = 2;                                                                  // connectPath.vertexCount() - 1 - extra[1]: b is sensor vertex 2
    const int nE = sm.nY;
    d3 value[5]; Float miW[5], valuePdf[5];
    po.nLight = 0;
    Float samplePosX = sm.posX, samplePosY = sm.posY;
    if (T1) {
        if (!sensor_sample_position(c, sm.X[1].p, sm.Y[s].p - sm.X[1].p, samplePosX, samplePosY) || !connectable_gbdpt(c, sm.Y[s])) return false;
        if (PHASE == 3) return true;             // (light tracing's ray-free filter, round 5: most emitter vertices lie outside the sensor's frustum -- its base-path launch ran at 9 % lane utilisation)
    }
    // light-tracing connections (t == 1): the base path Y[0..s-1], Ysc, S1c, X[0] and its four offsets (gbdpt_proc.cpp:356-376)
    BV Ysc, S1c; BE eL;
    bool pathSuccess0 = true;
    if (T1) {                                                                          // createShiftablePath(connectedBasePath, emitter, sensor, s, 1)
        Ysc = sm.Y[s]; S1c = sm.X[1]; be_clear(eL);
        // (phase 2 runs on the survivors of phase 1: their base path's sensor connection and -- below -- connection edge were traced there and found free)
        pathSuccess0 = bv_connect(c, &sm.Y[s - 1], Ysc, eL, S1c, &sm.X[0], bv_connectable(Ysc) ? M_AREA : M_DISCRETE, bv_connectable(S1c) ? M_AREA : M_DISCRETE, PHASE == 2);
        sensor_sample_position(c, S1c.p, Ysc.p - S1c.p, S1c.u, S1c.v);
    }
    if (PHASE == 1 && !T1) { c.nClosest = nClosest0; c.nShadow = nShadow0; }               // (the rays up to here were counted by phase 3; light tracing has no phase 3:
                                                                                           //  its filter IS a visibility ray -- the sensor connection -- and a launch of its own for it cost more than it saved)
    BE connEdge, connEdgeBase;
    d3 connPartsBase = mk(0.0);
    Float geomBase = 0.0;
    bool successConnectBase = false;
    Offset lo;                                                                             // the offset of a light path (transient)
    MisBase misBase;                                                                       // the base path's strategy densities of this connection (k = 0), reused by k = 1..4
    BV vtBaseCast;                                                                         // s == 0: the base path's sensor vertex t as the emitter sample it was cast to
    Float jacLP[4] = {1.0, 1.0, 1.0, 1.0};
    for (int k = 0; k <= ((PHASE == 1 || PHASE == 3) ? 0 : 4); k++) {
        miW[k] = 1.0 / (s + t + 1);
        bool ok = k == 0 ? true : (sm.off[k - 1].success != 0);
        value[k] = mk(0.0); valuePdf[k] = 0.0;
        d3 impWk = sm.impW[s]; Float impPk = sm.impP[s];
        const d3 radWk = sm.radW[T1 ? 0 : k][t]; const Float radPk = sm.radP[T1 ? 0 : k][t];
        bool lightOffset = false;
        if (T1 && k == 0) ok = pathSuccess0;
        if (T1 && k > 0 && !is_zero(value[0])) {
            if (!pathSuccess0) ok = false;
            else {                                                                         // createShiftedLightPath, :568-590
                ok = generate_offset(c, S1c, sm.X[0], sm.EX[0], Ysc, sm.Y[s - 1], s >= 2 ? &sm.Y[s - 2] : nullptr, eL.length,
                                     shifts[k - 1][0], shifts[k - 1][1], true, lo);
                if (ok) {
                    jacLP[k - 1] = lo.jacobian;
                    impPk = 1.0; impWk = mk(1.0);
                    for (int i = 1; i <= s; ++i) {                                         // the offset emitter subpath: Y[0..s-2], lo.c, lo.b
                        const BV &pv = i - 1 == s - 1 ? lo.c : sm.Y[i - 1];
                        const Float etr = i - 1 == s - 1 ? lo.ebc.tr[EImportance] : sm.EY[i - 1].tr[EImportance];
                        impWk = impWk * pv.w[EImportance] * pv.rr * etr;
                        impPk = impPk * pv.pdf[EImportance] * pv.rr * etr;
                    }
                    lightOffset = true;
                }
            }
        }
        Float geomTerm = 0.0;
        do {
            if (!(ok && pathSuccess0 && (k == 0 || (valuePdf[0] > 0 && !is_zero(value[0]))))) break;
            if (k > 0 && !T1 && !sm.off[k - 1].couldConnectAfterB && t > vert_b) break;
            // the connection end points: emitter side vs (with its predecessor), sensor side vt (with its predecessor)
            const BV *vsPred, *vtPred; const BV *vsP;
            BV vtLocal;                                                                    // s == 0: the sensor vertex is cast to an emitter sample (a copy: the cast of
            const BV *vtP;                                                                 // the reference mutates the shared vertex, which no later evaluation reads)
            if (lightOffset) { vsP = &lo.b; vsPred = &lo.c; }
            else { vsP = &sm.Y[s]; vsPred = s > 0 ? &sm.Y[s - 1] : nullptr; }
            vtP = &SV(sm, T1 ? 0 : k, t); vtPred = &SV(sm, T1 ? 0 : k, t - 1);
            if (vsP->type == T_EMITTER_SUPER) {
                vtLocal = *vtP;
                if (!bv_cast_emitter(c, vtLocal) || vtLocal.degenerate) { valuePdf[k] = radPk; break; }
                vtP = &vtLocal;
                if (k == 0) vtBaseCast = vtLocal;
                const d3 connParts = (k > 0 && t > vert_b + 1) ? connPartsBase : bv_eval(c, *vsP, vsPred, vtP, EImportance) * bv_eval(c, *vtP, vtPred, vsP, ERadiance);
                if (k == 0) connPartsBase = connParts;
                value[k] = radWk * connParts;
                valuePdf[k] = radPk;
            } else {
                if (!connectable_gbdpt(c, *vsP) || !connectable_gbdpt(c, *vtP)) { valuePdf[k] = impPk * radPk; break; }
                const d3 connParts = (k > 0 && t > vert_b + 1) ? connPartsBase : bv_eval(c, *vsP, vsPred, vtP, EImportance) * bv_eval(c, *vtP, vtPred, vsP, ERadiance);
                if (k == 0) connPartsBase = connParts;
                value[k] = impWk * radWk * connParts;
                valuePdf[k] = impPk * radPk;
            }
            if (is_zero(value[k]) || valuePdf[k] == 0) break;
            if (PHASE == 3) return true;                                                    // (k == 0: both end points face each other and carry throughput -- worth a visibility ray)
            const bool successConnect = (k > 0 && t > vert_b) ? successConnectBase : edge_path_connect(c, connEdge, *vsP, *vtP, PHASE == 2 && k == 0);
            if (k == 0) successConnectBase = successConnect;
            if (!successConnect) { value[k] = mk(0.0); break; }
            geomTerm = (k > 0 && t > vert_b) ? geomBase : edge_geometry_term(c, connEdge, *vsP, *vtP);
            value[k] = value[k] * geomTerm;
            valuePdf[k] *= 1.0;                                                            // genGeomTerm (calcSpecularPDFChange): 1 without specular chains
            if (is_zero(value[k]) || valuePdf[k] == 0) break;
            PathRef base; base.sm = &sm; base.k = 0; base.ev = sm.Y; base.ee = sm.EY; base.ne = nE; base.ovS = base.ovSm1 = nullptr; base.ovEm1 = nullptr; base.ovT = nullptr;
            if (s == 0) base.ovT = &vtBaseCast;                                            // (the cast vertex: miWeight sees the emitter sample, gbdpt_proc.cpp:402)
            if (k == 0) {
                connEdgeBase = connEdge;
                geomBase = geomTerm;
                miW[0] = mi_weight_base(c, sm, base, connEdgeBase, s, t, misBase) / valuePdf[0];
            } else {
                if (CLS == 2) miW[k] = mi_weight_grad_shared(misBase, base, k, s, t, sm.off[k - 1].jacobian) / valuePdf[0];
                else {
                    PathRef off = base;
                    off.k = T1 ? 0 : k;
                    if (lightOffset) { off.ovS = &lo.b; off.ovSm1 = &lo.c; off.ovEm1 = &lo.ebc; }
                    if (s == 0) off.ovT = vtP;
                    miW[k] = mi_weight_grad(c, misBase, off, connEdge, s, t, T1 ? jacLP[k - 1] : sm.off[k - 1].jacobian) / valuePdf[0];
                }
            }
        } while (false);
#ifdef GDPT_BD_TRACE
        printf("st %d %d k %d ok %d value %.17g %.17g %.17g pdf %.17g miW %.17g geom %.17g rays %u %u\n", s, t, k, (int)ok, value[k].x, value[k].y, value[k].z, valuePdf[k], miW[k], geomTerm, c.nClosest, c.nShadow);
#endif
        if (is_zero(value[k]) || is_zero(value[0])) { value[k] = mk(0.0); miW[k] = miW[0]; valuePdf[k] = valuePdf[0]; }
        if (PHASE == 2 && k == 0) { c.nClosest = nClosest0; c.nShadow = nShadow0; }        // (counted by phase 1)
    }
    if (PHASE == 3 || is_zero(value[0])) return false;
    const d3 mainRad = value[0] * (valuePdf[0] * miW[0]);
    po.primal = mainRad;
    if (PHASE != 2 && T1) { LightSplat &ls = po.light[po.nLight++]; ls.x = samplePosX; ls.y = samplePosY; ls.buffer = 0; ls.value = mainRad; }
    if (PHASE == 1) return true;
    const d3 fx = value[0] * valuePdf[0];
    for (int n = 0; n < 4; n++) {
        const d3 fy = value[n + 1] * valuePdf[n + 1] * (T1 ? jacLP[n] : sm.off[n].jacobian);
        const d3 gradVal = (fy - fx) * ((Float)2.0 * miW[n + 1]);
        po.gradient[n] = gradVal;
        if (T1) { LightSplat &ls = po.light[po.nLight++]; ls.x = samplePosX; ls.y = samplePosY; ls.buffer = n + 1; ls.value = gradVal; }
    }
    return true;
}

// GBDPTRenderer::process + evaluate for one sample in ONE lane, connections in the reference's order (the probe entry; the frame kernels
// run walk_sample and connect_pair as separate launches)
__device__ void process_sample(Ctx &c, Sample &sm, int px, int py, SampleOut &out)
{
    out.primal = mk(0.0); out.nLight = 0;
    for (int k = 0; k < 4; k++) out.gradient[k] = mk(0.0);
    walk_sample(c, sm, px, py);
    out.posX = sm.posX; out.posY = sm.posY;
    PairOut po;
    for (int s = sm.nY - 1; s >= 0; --s) {
        int minT, maxT;
        pair_range(c.cfg, sm.nX, s, minT, maxT);
        for (int t = maxT; t >= minT; --t) {
            if (!(t == 1 ? connect_pair<0>(c, sm, s, t, po) : (shares_connection(sm, t) ? connect_pair<2>(c, sm, s, t, po) : connect_pair<1>(c, sm, s, t, po)))) continue;
            if (t >= 2) { out.primal = out.primal + po.primal; for (int n = 0; n < 4; n++) out.gradient[n] = out.gradient[n] + po.gradient[n]; }
            for (int i = 0; i < po.nLight && out.nLight < BD_MAX_LIGHT; i++) out.light[out.nLight++] = po.light[i];
        }
    }
}

} // namespace gdpt_bd
