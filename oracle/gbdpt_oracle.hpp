/*
 * oracle/gbdpt_oracle.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Included by gpt_oracle.cpp inside its anonymous namespace (it uses
 * that file's scene, ray casting, BSDF, emitter and sensor restatements).
 *
 * CPU restatement of the reference's G-BDPT per-sample hot path (BASELINE config 5):
 *   GBDPTRenderer::process / evaluate / createShiftablePath / createShiftedLightPath / combine*Data
 *                                                          src/integrators/gbdpt/gbdpt_proc.cpp:86-256,259-534,544-662
 *   GBDPTWorkResult::putSample / putLightSample, GBDPTProcess::develop      gbdpt_wr.h:56-62, gbdpt_proc.cpp:694-706, multifilm.cpp:317-362
 *   Path::alternatingRandomWalkFromPixel, miWeightBaseNoSweep_GBDPT, miWeightGradNoSweep_GBDPT, halfJacobian_GBDPT,
 *   calcSpecularPDFChange, G, isConnectable_GBDPT          src/libbidir/path.cpp:26-454,548-631
 *   PathVertex::sampleNext / sampleSensor / perturbDirection / eval / evalPdf / cast / update / connect /
 *   getSamplePosition / updateSamplePosition               src/libbidir/vertex.cpp:35-384,488-679,781-1022,1115-1211,1298-1370
 *   PathEdge::sampleNext / perturbDirection / connect / pathConnectAndCollapse / evalCached     src/libbidir/edge.cpp:27-131,169-287,442-574
 *   ManifoldPerturbation::computeMuRec / getSpecularChainEndGBDPT / generateOffsetPathGBDPT / perturbDirection
 *                                                          src/libbidir/mut_manifold.cpp:806-986,1230-1296
 *   ManifoldPerturbation::propagatePerturbation / manifoldWalk (stage C, round 4)                 src/libbidir/mut_manifold.cpp:989-1227
 *   PathVertex::propagatePerturbation                                                           src/libbidir/vertex.cpp:681-790
 *   SpecularManifold::init / computeTangents / project / move / update / det / multiG / G       src/libbidir/manifold.cpp:59-951
 *   PerspectiveCamera::importance / samplePosition / sampleDirection / pdfDirection / evalDirection / getSamplePosition
 *                                                          src/sensors/perspective.cpp:190-247,300-410
 *   AreaLight::samplePosition / evalPosition / pdfPosition / sampleDirection / evalDirection / pdfDirection   src/emitters/area.cpp:93-142
 *   Scene::sampleEmitterPosition / pdfEmitterPosition      src/librender/scene.cpp:985-1006
 *
 * SCOPE: surface interactions only (no media), area (triangle-mesh / rectangle) emitters, the perspective sensor, the box filter (the only one
 * G-BDPT supports, gbdpt.cpp:70-71); BSDFs diffuse / roughconductor (plain or two-sided, textured or not) and -- stage C -- conductor, dielectric
 * and rough conductors below shiftThreshold: paths with SPECULAR CHAINS.  Their offset paths follow the reference's three mechanisms: the chain
 * between the sensor a and the first connectable vertex b is re-created vertex by vertex (a specular vertex takes its delta component again, a
 * glossy one keeps its half vector), the chain between b and the next connectable vertex c follows b by a Newton walk on the specular manifold
 * (reversibility-checked; a failed walk falls back to the unshifted chain), and Jacobians / MIS weights carry SpecularManifold's generalized
 * geometry terms and determinants.  Not carried: media, directional endpoints (EPinnedDirection), index-matched ENull transmission; the one
 * place the reference calls Eigen (the inverse and determinant of the mixed glossy / specular system of SpecularManifold::det) is restated as
 * Gauss-Jordan / LU with partial pivoting.  A BSDF is identified by its material index where the reference compares BSDF pointers.
 * `unsupported` counts what falls outside (a chain through a non-surface vertex); tests assert it is zero.
 *
 * PARITY UNPINNED, like the rest of this oracle: nothing here was compared with output of the reference (it cannot be built in this image).
 * Random numbers: the counter-based stream of gpt_oracle.cpp (one per pixel and sample), consumed in the reference's order.
 */
namespace gb {

enum { ERadiance = 0, EImportance = 1 };                                                   // include/mitsuba/render/common.h:33-43
enum { EInvalidMeasure = 0, ESolidAngle = 1, ELength = 2, EArea = 3, EDiscrete = 4 };      // common.h:56-67
enum { EInvalid = 0, ESensorSupernode = 1, EEmitterSupernode = 2, ESensorSample = 4, EEmitterSample = 8, ESurfaceInteraction = 16,
       ESupernode = 3 };                                                                   // include/mitsuba/bidir/vertex.h:67-87
enum { EValueImp = 0x01, EValueRad = 0x02, ECosineImp = 0x04, ECosineRad = 0x08, EInverseSquareFalloff = 0x10, ETransmittance = 0x20,
       EGeometricTerm = 0x04 | 0x08 | 0x10, EGeneralizedGeometricTerm = 0x04 | 0x08 | 0x10 | 0x20 };   // include/mitsuba/bidir/edge.h:171-185

const Float INV_FOURPI = 0.07957747154594766788;
struct Config { int maxDepth, rrDepth, lightImage, spp; Float shiftThreshold; uint64_t seed; };

struct PRec { V3 p, n; Float pdf = 0; int measure = 0; Float uvx = 0, uvy = 0; int object = -1;       // PositionSamplingRecord, common.h:70-150
              bool onSurface = true; };        // object->getType() & EOnSurface: area lights and both perspective sensors; a `point` emitter is not (point.cpp:56)

struct Vertex {                                     // PathVertex, vertex.h
    int type = EInvalid;
    bool degenerate = false;
    int measure = EInvalidMeasure;
    int componentType = 0, sampledComponentIndex = 0;
    V3 weight[2];
    Float pdf[2] = {0, 0};
    Float rrWeight = 0;
    Intersection its;                               // ESurfaceInteraction
    gpo_material mat;                               // its.getBSDF() with its texture resolved at its.uv (no ray differentials in libbidir)
    PRec prec;                                      // sensor / emitter samples (the reference keeps both in one union: cast() overwrites)
    bool isSupernode() const { return (type & ESupernode) != 0; }
    bool isConnectable() const { return !degenerate && measure != EDiscrete; }             // vertex.h:750
    bool isSurface() const { return type == ESurfaceInteraction; }
    bool isOnSurface() const { return type == ESurfaceInteraction || ((type == EEmitterSample || type == ESensorSample) && prec.onSurface); }   // vertex.h:592-596
    V3 position() const { return type == ESurfaceInteraction ? its.p : prec.p; }           // vertex.cpp:1213-1229
    V3 shadingNormal() const { return type == ESurfaceInteraction ? its.sh.n : prec.n; }   // :1231-1243
    V3 geometricNormal() const { return type == ESurfaceInteraction ? its.geoN : prec.n; } // :1245-1257
};

struct Edge {                                       // PathEdge, edge.h
    V3 d;
    Float length = 0;
    V3 weight[2];
    Float pdf[2] = {0, 0};
};

struct Path {                                       // Path, path.h: m_vertices / m_edges of POINTERS (base and offset paths share vertices)
    std::vector<Vertex *> v;
    std::vector<Edge *> e;
    int length() const { return (int)e.size(); }
    int vertexCount() const { return (int)v.size(); }
    Vertex *vertexOrNull(int i) const { return (i < 0 || i >= (int)v.size()) ? nullptr : v[i]; }
    Edge *edgeOrNull(int i) const { return (i < 0 || i >= (int)e.size()) ? nullptr : e[i]; }
    void reverse() { std::reverse(v.begin(), v.end()); std::reverse(e.begin(), e.end()); }   // path.cpp:633-636
    void clear() { v.clear(); e.clear(); }
};

struct Pool {                                       // MemoryPool: per sample, released wholesale
    std::deque<Vertex> vs;
    std::deque<Edge> es;
    Vertex *allocVertex() { vs.emplace_back(); return &vs.back(); }
    Edge *allocEdge() { es.emplace_back(); return &es.back(); }
    Vertex *clone(const Vertex *o) { vs.push_back(*o); return &vs.back(); }
};

struct Ctx {
    const Scene &sc;
    Config cfg;
    Float invLin[9];                                // linear part of the inverse camera transform (trafo.inverse() applied to a direction)
    V3 camPos, camDir;
    Float rectX, rectY, normalization;              // m_imageRect half extents and 1 / its area, perspective.cpp:167-173
    bool thinlens = false;                          // `thinlens` sensor (thinlens.cpp): the position sample is a point of the aperture disk (EArea), not a delta
    Float aperturePdf = 0;                          // 1 / (pi r^2), thinlens.cpp:213
    uint64_t unsupported = 0;
    // specular-chain statistics of the offset paths (the reference's statsUsedManifold / statsMWSuccess / statsUsedPropagation counters,
    // mut_manifold.cpp:33-75): manifold walks entered, walks that converged reversibly, chain vertices re-created by propagatePerturbation
    uint64_t walks = 0, walksOk = 0, propagated = 0;
};

inline void cameraSetup(Ctx &c)
{
    const double *M = c.sc.cam.toWorld;
    const Float a = M[0], b = M[1], cc = M[2], d = M[4], e = M[5], f = M[6], g = M[8], h = M[9], i = M[10];
    const Float A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;                    // adjugate / determinant, a fixed operation order
    const Float det = a * A + b * B + cc * C, r = 1.0 / det;
    c.invLin[0] = A * r; c.invLin[1] = (cc * h - b * i) * r; c.invLin[2] = (b * f - cc * e) * r;
    c.invLin[3] = B * r; c.invLin[4] = (a * i - cc * g) * r; c.invLin[5] = (cc * d - a * f) * r;
    c.invLin[6] = C * r; c.invLin[7] = (b * g - a * h) * r; c.invLin[8] = (a * e - b * d) * r;
    c.camPos = V3(M[3], M[7], M[11]);                                                      // trafo(Point(0))
    c.camDir = V3(M[2], M[6], M[10]);                                                      // trafo(Vector(0, 0, 1)), perspective.cpp:303-305
    c.rectX = c.sc.tanHalf; c.rectY = c.sc.tanHalf / c.sc.aspect;                           // sampleToCamera(0,0,0) / z and (1,1,0) / z written out (crop == film)
    c.normalization = 1.0 / ((2 * c.rectX) * (2 * c.rectY));
    c.thinlens = c.sc.cam.type == 1;
    if (c.thinlens) c.aperturePdf = 1 / (PI * c.sc.cam.apertureRadius * c.sc.cam.apertureRadius);
}
inline V3 camToLocal(const Ctx &c, V3 d) { return mul3(c.invLin, d); }
inline V3 camToLocalPoint(const Ctx &c, V3 p) { return mul3(c.invLin, p - c.camPos); }      // trafo.inverse().transformAffine(p)
inline V3 camToWorld(const Ctx &c, V3 d)
{
    const double *M = c.sc.cam.toWorld;
    return V3(M[0] * d.x + M[1] * d.y + M[2] * d.z, M[4] * d.x + M[5] * d.y + M[6] * d.z, M[8] * d.x + M[9] * d.y + M[10] * d.z);
}

// PerspectiveCameraImpl::importance, perspective.cpp:190-247
inline Float importance(const Ctx &c, V3 d)
{
    const Float cosT = cosTheta(d);
    if (cosT <= 0) return 0.0;
    const Float invCosTheta = 1.0 / cosT;
    const Float px = d.x * invCosTheta, py = d.y * invCosTheta;
    if (!(px >= -c.rectX && px <= c.rectX && py >= -c.rectY && py <= c.rectY)) return 0.0;   // AABB2::contains
    return c.normalization * invCosTheta * invCosTheta * invCosTheta;
}
// m_cameraToSample(P).xy for crop == film (the inverse of sampleToCameraDir's map): the film position, in [0,1]^2, that the camera-space point P projects to
inline void cameraToSample(const Ctx &c, V3 P, Float &sx, Float &sy)
{
    sx = 0.5 * (1 - P.x / (P.z * c.sc.tanHalf)); sy = 0.5 * (1 - P.y * c.sc.aspect / (P.z * c.sc.tanHalf));
}
// ThinLensCamera::importance, thinlens.cpp:231-291: p a point of the aperture, d the direction from it (both in camera space); the pixel is the one whose
// focus-plane point the ray passes through
inline Float importanceLens(const Ctx &c, V3 p, V3 d, Float *ox = nullptr, Float *oy = nullptr)
{
    const Float cosT = cosTheta(d);
    if (cosT <= 0) return 0.0;
    const Float invCosTheta = 1.0 / cosT;
    Float sx, sy;
    cameraToSample(c, p + d * (c.sc.cam.focusDistance * invCosTheta), sx, sy);
    if (sx < 0 || sx > 1 || sy < 0 || sy > 1) return 0.0;
    if (ox) { *ox = sx * c.sc.cam.width; *oy = sy * c.sc.cam.height; }
    return c.normalization * invCosTheta * invCosTheta * invCosTheta;
}
// Sensor::evalDirection == pdfDirection of a sensor sample at world position p towards world direction d (perspective.cpp:373-391, thinlens.cpp:420-437)
inline Float sensorDirection(const Ctx &c, V3 p, V3 d)
{
    return c.thinlens ? importanceLens(c, camToLocalPoint(c, p), camToLocal(c, d)) : importance(c, camToLocal(c, d));
}
// m_sampleToCamera(Point(sx, sy, 0)) normalised: the direction through a film position given in [0,1]^2 (perspective.cpp:150-156 written out)
inline V3 sampleToCameraDir(const Ctx &c, Float sxn, Float syn)
{
    const gpo_camera &cam = c.sc.cam;
    return normalize(V3((1 - 2 * sxn) * cam.nearClip * c.sc.tanHalf, (1 - 2 * syn) / c.sc.aspect * cam.nearClip * c.sc.tanHalf, cam.nearClip));
}
// PerspectiveCameraImpl::getSamplePosition, perspective.cpp:393-410 (m_cameraToSample written out: the inverse of the map above)
// ... and ThinLensCamera::getSamplePosition, thinlens.cpp:536-557 (pWorld: the sensor sample's point of the aperture; the pinhole does not read it)
inline bool sensorSamplePosition(const Ctx &c, V3 pWorld, V3 dWorld, Float &ox, Float &oy)
{
    const V3 local = camToLocal(c, dWorld);
    if (local.z <= 0) return false;
    if (c.thinlens) {
        const V3 localP = camToLocalPoint(c, pWorld);
        Float sx, sy;
        cameraToSample(c, localP + local * (c.sc.cam.focusDistance / local.z), sx, sy);
        if (sx < 0 || sx > 1 || sy < 0 || sy > 1) return false;
        ox = sx * c.sc.cam.width; oy = sy * c.sc.cam.height;
        return true;
    }
    const Float sx = 0.5 * (1 - local.x / (local.z * c.sc.tanHalf)), sy = 0.5 * (1 - local.y * c.sc.aspect / (local.z * c.sc.tanHalf));
    if (sx < 0 || sx > 1 || sy < 0 || sy > 1) return false;
    ox = sx * c.sc.cam.width; oy = sy * c.sc.cam.height;
    return true;
}

// ---- emitters ------------------------------------------------------------------------------------------------------------------------
inline Float envInvSurfaceArea(const Scene &sc) { return 1 / (4 * PI * sc.bsRadius * sc.bsRadius); }     // constant.cpp:97-100
// `envmap` as libbidir's endpoint (round 5): positions as the constant environment has them (uniform on the sphere of createShape, envmap.cpp:331-343,412-430), directions
// from the map itself: sampleDirection / pdfDirection / evalDirection (envmap.cpp:455-498) importance-sample, price and look up a direction of the map whatever the position
// (the comment at envmap.cpp:432-454 calls it a compromise); m_power = surfaceArea * m_scale / m_normalization (envmap.cpp:326-328)
inline Float envMapPower(const Scene &sc) { const Float surfaceArea = 4 * PI * sc.bsRadius * sc.bsRadius; return surfaceArea * sc.envMap.scale / sc.envMap.normalization; }
inline bool isEnvMap(const Scene &sc, int object) { return sc.envMap.present && object == sc.envIndex; }
// EnvironmentMap::evalDirection, envmap.cpp:482-498: bilinear on level 0 along -d (map space), times m_normalization (not m_scale); the measure is not consulted
inline V3 envMapEvalDirection(const Scene &sc, V3 d)
{
    const V3 v = -mul3(sc.envMap.toLocal, d);
    const Float uvx = std::atan2(v.x, -v.z) * INV_TWOPI, uvy = std::acos(std::min(1.0, std::max(-1.0, v.y))) * INV_PI;
    Float o[3];
    sc.envMap.mip.evalBilinear(0, uvx, uvy, o);
    return V3(o[0], o[1], o[2]) * sc.envMap.normalization;
}
// Scene::sampleEmitterPosition (scene.cpp:985-1001) -> AreaLight::samplePosition (area.cpp:93-97) -> TriMesh / Rectangle::samplePosition
inline V3 sampleEmitterPosition(const Ctx &c, PRec &pRec, Float sx, Float sy)
{
    const Scene &sc = c.sc;
    Float emPdf;
    const size_t index = sc.emitterPDF.sampleReuse(sx, emPdf);
    const Emitter &em = sc.emitters[index];
    if (em.numTris == 0) {                                                                 // ConstantBackgroundEmitter::samplePosition, constant.cpp:110-120
        const V3 d = squareToUniformSphere(sx, sy);
        pRec.p = sc.bsCenter + d * sc.bsRadius; pRec.n = -d; pRec.measure = EArea; pRec.pdf = envInvSurfaceArea(sc); pRec.object = (int)index;
        pRec.pdf *= emPdf;
        const Float surfaceArea = 4 * PI * sc.bsRadius * sc.bsRadius;
        if (sc.envMap.present) return V3(envMapPower(sc)) / emPdf;                         // EnvironmentMap::samplePosition, envmap.cpp:412-422: the same sphere, Spectrum(m_power)
        return (em.radiance * surfaceArea * PI) / emPdf;                                   // m_power, constant.cpp:100
    }
    if (em.numTris < 0) {                                                                  // PointEmitter::samplePosition, point.cpp:79-87
        pRec.p = em.position; pRec.n = V3(0.0); pRec.pdf = 1.0; pRec.measure = EDiscrete; pRec.object = (int)index; pRec.onSurface = false;
        pRec.pdf *= emPdf;
        return (em.radiance * (4 * PI)) / emPdf;
    }
    if (em.rectangle) {                                                                    // rectangle.cpp:210-216
        const Float lx = sx * 2 - 1, ly = sy * 2 - 1;
        const Float *M = em.rect;
        pRec.p = V3(M[0] * lx + M[1] * ly + M[2] * 0.0 + M[3], M[4] * lx + M[5] * ly + M[6] * 0.0 + M[7], M[8] * lx + M[9] * ly + M[10] * 0.0 + M[11]);
        pRec.n = em.rectN;
        pRec.uvx = sx; pRec.uvy = sy;
    } else {                                                                               // trimesh.cpp:412-423, triangle.cpp:24-59
        const std::vector<Float> &cdf = em.cdf;
        auto entry = std::lower_bound(cdf.begin(), cdf.end(), sy);
        size_t ti = std::min(cdf.size() - 2, (size_t)std::max((std::ptrdiff_t)0, (std::ptrdiff_t)(entry - cdf.begin()) - 1));
        while ((cdf[ti + 1] - cdf[ti]) == 0 && ti < cdf.size() - 1) ++ti;
        sy = (sy - cdf[ti]) / (cdf[ti + 1] - cdf[ti]);
        const Tri &tr = sc.tris[em.firstTri + ti];
        const Float a = safe_sqrt(1.0 - sx);
        const Float bx = 1 - a, by = a * sy;
        const V3 sideA = tr.p1 - tr.p0, sideB = tr.p2 - tr.p0;
        pRec.p = tr.p0 + (sideA * bx) + (sideB * by);
        pRec.n = normalize(cross(sideA, sideB));
        pRec.uvx = bx; pRec.uvy = by;
    }
    pRec.pdf = em.invSurfaceArea;
    pRec.measure = EArea;
    pRec.object = (int)index;
    pRec.pdf *= emPdf;
    const Float area = 1.0 / em.invSurfaceArea;
    const V3 power = em.radiance * PI * area;                                              // m_power = m_radiance * M_PI * getSurfaceArea(), area.cpp:196
    return power / emPdf;
}
inline Float pdfEmitterPosition(const Ctx &c, const PRec &pRec, int measure)               // scene.cpp:1003-1006 (pRec.measure = measure, vertex.cpp:925-927)
{
    if (c.sc.emitters[pRec.object].numTris < 0) return (measure == EDiscrete ? 1.0 : 0.0) * (1.0 * c.sc.emitterPDF.normalization);   // point.cpp:93-95
    if (c.sc.emitters[pRec.object].numTris == 0) return envInvSurfaceArea(c.sc) * (1.0 * c.sc.emitterPDF.normalization);             // constant.cpp:126-128
    return c.sc.emitters[pRec.object].invSurfaceArea * (1.0 * c.sc.emitterPDF.normalization);
}
// Emitter::evalDirection == pdfDirection of an emitter sample: AreaLight (below), PointEmitter::evalDirection / pdfDirection, point.cpp:107-115
inline Float areaDirection(V3 d, V3 n, int measure);
inline Float emitterDirection(const Ctx &c, const PRec &pRec, V3 d, int measure)
{
    if (c.sc.emitters[pRec.object].numTris < 0) return measure == ESolidAngle ? INV_FOURPI : 0.0;
    if (isEnvMap(c.sc, pRec.object)) return envMapPdfDirection(c.sc.envMap, -mul3(c.sc.envMap.toLocal, d));   // EnvironmentMap::pdfDirection, envmap.cpp:476-480 (no measure test)
    return areaDirection(d, pRec.n, measure);
}
// Emitter::evalDirection as a spectrum: the envmap's is coloured (envmap.cpp:482-498), every other emitter's equals its pdfDirection
inline V3 emitterEvalDirection(const Ctx &c, const PRec &pRec, V3 d, int measure)
{
    if (isEnvMap(c.sc, pRec.object)) return envMapEvalDirection(c.sc, d);
    return V3(emitterDirection(c, pRec, d, measure));
}
inline Float areaDirection(V3 d, V3 n, int measure)                                        // AreaLight::evalDirection / pdfDirection, area.cpp:124-142
{
    Float dp = dot(d, n);
    if (measure != ESolidAngle || dp < 0) dp = 0.0;
    return INV_PI * dp;
}

inline int bsdfMeasure(int m) { return m == EDiscrete ? MEASURE_DISCRETE : MEASURE_SOLID_ANGLE; }
inline bool hasSmooth(const gpo_material &m) { return (bsdfType(m) & ESmooth) != 0; }

// Path::isConnectable_GBDPT, path.cpp:30-47
inline bool isConnectableGBDPT(const Vertex *va, Float threshold)
{
    if (!va->isConnectable()) return false;
    if (va->type & (ESupernode | ESensorSample | EEmitterSample)) return true;
    const Float roughness = getRoughness(va->mat);
    if (roughness < threshold) return false;
    return true;
}

// ---- the environment emitter as libbidir sees it: a SHAPE ---------------------------------------------------------------------------------
// Scene::initializeBidirectional asks every emitter for a shape (scene.cpp:397-408); ConstantBackgroundEmitter::createShape (constant.cpp:67-93) answers with a
// `sphere` of m_sceneBSphere (the bounding sphere of kd-tree + sensor, radius x 1.5) with flipped normals, the emitter as its child and -- Shape::configure gives a
// light source without a BSDF an all-absorbing one (shape.cpp) -- a `diffuse` BSDF of reflectance 0.  Scene::rayIntersectAll (scene.cpp:736-760) tests it after the
// kd-tree: a ray that leaves the geometry ends in a SURFACE vertex on that sphere, connectable (a smooth BSDF), black, and castable to an emitter sample.  So the
// constant environment is an area light whose shape is a sphere.  Its hits carry prim == ENV_PRIM.
constexpr int ENV_PRIM = -2;
inline bool hasEnvShape(const Scene &sc) { return sc.envIndex >= 0; }                    // (`constant` and `envmap` alike: constant.cpp:67-93, envmap.cpp:331-376)
inline int emitterOfPrim(const Scene &sc, int prim) { return prim == ENV_PRIM ? sc.envIndex : sc.tris[prim].emitter; }
inline int materialOfPrim(const Scene &sc, int prim) { return prim == ENV_PRIM ? -1 : sc.tris[prim].material; }
inline gpo_material blackDiffuse() { gpo_material m; std::memset(&m, 0, sizeof m); m.type = MAT_DIFFUSE; return m; }
// Scene::rayIntersectAll(ray, its): the kd-tree, then the special shapes with maxt = the hit found so far (Sphere::rayIntersect / fillIntersectionRecord,
// sphere.cpp:163-187,209-255: the quadratic in double as there; the sphere encloses everything, so it only ever answers when the kd-tree found nothing)
inline bool rayIntersectAll(const Scene &sc, const Ray &ray, Intersection &its)
{
    const bool result = rayIntersect(sc, ray, its);
    if (!hasEnvShape(sc)) return result;
    const Float maxt = result ? its.t : ray.maxt;
    Float mint = ray.mint;
    if (mint == Epsilon) mint *= std::max(std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z)), Epsilon);
    const V3 o = ray.o - sc.bsCenter;
    const Float A = lengthSquared(ray.d), B = 2 * dot(o, ray.d), C = lengthSquared(o) - sc.bsRadius * sc.bsRadius;
    Float nearT, farT;
    if (!solveQuadratic(A, B, C, nearT, farT)) return result;
    if (!(nearT <= maxt && farT >= mint)) return result;
    Float t;
    if (nearT < mint) { if (farT > maxt) return result; t = farT; } else t = nearT;
    its = Intersection();
    its.t = t; its.prim = ENV_PRIM;
    its.p = ray.o + ray.d * t;
    const V3 local = its.p - sc.bsCenter;                                                  // (the shape's transform is translate(centre): the scale went into m_radius, sphere.cpp:108-118)
    const Float theta = std::acos(std::min((Float)1.0, std::max((Float)-1.0, local.z / sc.bsRadius)));   // math::safe_acos
    Float phi = std::atan2(local.y, local.x);
    if (phi < 0) phi += 2 * PI;
    its.u = phi * (0.5 * INV_PI); its.v = theta * INV_PI;
    its.dpdu = V3(-local.y, local.x, 0.0) * (2 * PI);
    its.geoN = normalize(its.p - sc.bsCenter);
    const Float zrad = std::sqrt(local.x * local.x + local.y * local.y);
    if (zrad > 0) {
        const Float invZRad = 1.0 / zrad, cosPhi = local.x * invZRad, sinPhi = local.y * invZRad;
        its.dpdv = V3(local.z * cosPhi, local.z * sinPhi, -std::sin(theta) * sc.bsRadius) * PI;
    } else its.dpdv = V3(local.z * 0.0, local.z * 1.0, -std::sin(theta) * sc.bsRadius) * PI;
    its.geoN = -its.geoN;                                                                  // m_flipNormals
    its.sh.n = its.geoN;                                                                   // (the reference sets shFrame.n only; s and t are read by nothing that survives
    coordinateSystem(its.sh.n, its.sh.s, its.sh.t);                                        //  the black BSDF: a frame built from n stands in)
    its.wi = its.sh.toLocal(-ray.d);
    return true;
}

struct Tracer {
    Ctx &c;
    Rng &rng;
    Pool &pool;
    Tracer(Ctx &c_, Rng &r_, Pool &p_) : c(c_), rng(r_), pool(p_) {}

    // ---- PathEdge ----------------------------------------------------------------------------------------------------------------
    void fillSurface(Vertex *succ) const
    {
        succ->type = ESurfaceInteraction;
        Ray none;
        succ->mat = succ->its.prim == ENV_PRIM ? blackDiffuse() : matOf(c.sc, succ->its, none);
        succ->degenerate = !(hasSmooth(succ->mat) || emitterOfPrim(c.sc, succ->its.prim) >= 0);   // edge.cpp:44-45 (no sensor shapes)
    }
    // PathEdge::sampleNext, edge.cpp:27-71 (no media)
    bool edgeSampleNext(Edge *e, const Ray &ray, Vertex *succ, int mode) const
    {
        if (!rayIntersectAll(c.sc, ray, succ->its)) return false;
        fillSurface(succ);
        e->length = succ->its.t;
        if (e->length == 0) return false;
        e->weight[ERadiance] = e->weight[EImportance] = V3(1.0);
        e->pdf[ERadiance] = e->pdf[EImportance] = 1.0;
        e->d = ray.d;
        if (mode == ERadiance) e->d = -e->d;
        return true;
    }
    // PathEdge::perturbDirection, edge.cpp:73-131 (no media: wantMedium is false, desiredType is not consulted otherwise)
    bool edgePerturbDirection(Edge *e, const Ray &ray, Float dist, Vertex *succ, int mode) const
    {
        const bool surface = rayIntersectAll(c.sc, ray, succ->its);
        if (dist <= 0) return false;
        if (!surface) return false;
        fillSurface(succ);
        e->length = succ->its.t;
        e->d = ray.d;
        if (mode == ERadiance) e->d = -e->d;
        if (e->length == 0) return false;
        e->weight[ERadiance] = e->weight[EImportance] = V3(1.0);
        e->pdf[ERadiance] = e->pdf[EImportance] = 1.0;
        return true;
    }
    // PathEdge::connect, edge.cpp:221-287
    bool edgeConnect(Edge *e, const Vertex *vs, const Vertex *vt) const
    {
        if (vs->type == EEmitterSupernode || vt->type == ESensorSupernode) {
            const Float radianceTransport = vt->type == ESensorSupernode ? 1.0 : 0.0, importanceTransport = 1 - radianceTransport;
            e->d = V3(0.0); e->length = 0.0;
            e->pdf[ERadiance] = radianceTransport; e->pdf[EImportance] = importanceTransport;
            e->weight[ERadiance] = V3(radianceTransport); e->weight[EImportance] = V3(importanceTransport);
        } else {
            const V3 vsp = vs->position(), vtp = vt->position();
            e->d = vsp - vtp;
            e->length = length(e->d);
            e->d = e->d / e->length;
            Ray ray(vtp, e->d, vt->isOnSurface() ? Epsilon : 0.0, e->length * (vs->isOnSurface() ? (1 - ShadowEpsilon) : 1.0));
            // (rayIntersectAll's shadow form also asks the environment's sphere, sphere.cpp:189-207 -- which answers "no" to every segment with both ends inside or
            //  on it: nearT < mint and farT > maxt.  Every vertex lies inside or on that sphere, so the kd-tree alone decides.)
            if (rayIntersectShadow(c.sc, ray)) return false;
            e->weight[ERadiance] = e->weight[EImportance] = V3(1.0);
            e->pdf[ERadiance] = e->pdf[EImportance] = 1.0;
        }
        e->d = -e->d;
        return true;
    }
    // PathEdge::pathConnectAndCollapse, edge.cpp:442-574 (no media, no ENull BSDFs: any surface in between is an occluder)
    bool edgePathConnectAndCollapse(Edge *e, const Vertex *vs, const Vertex *vt, int &interactions) const
    {
        if (vs->type == EEmitterSupernode || vt->type == ESensorSupernode) {
            const Float radianceTransport = vt->type == ESensorSupernode ? 1.0 : 0.0, importanceTransport = 1 - radianceTransport;
            e->length = 0.0; e->d = V3(0.0);
            e->pdf[ERadiance] = radianceTransport; e->pdf[EImportance] = importanceTransport;
            e->weight[ERadiance] = V3(radianceTransport); e->weight[EImportance] = V3(importanceTransport);
            interactions = 0;
        } else {
            const V3 vsp = vs->position(), vtp = vt->position();
            e->d = vsp - vtp;
            e->length = length(e->d);
            interactions = 0;
            if (e->length == 0) return false;
            e->d = e->d / e->length;
            const Float lengthFactor = vs->isOnSurface() ? (1 - ShadowEpsilon) : 1.0;
            Ray ray(vtp, e->d, vt->isOnSurface() ? Epsilon : 0.0, e->length * lengthFactor);
            e->weight[ERadiance] = e->weight[EImportance] = V3(1.0);
            e->pdf[ERadiance] = e->pdf[EImportance] = 1.0;
            Intersection its;
            if (rayIntersect(c.sc, ray, its)) return false;
        }
        e->d = -e->d;
        return true;
    }
    // PathEdge::evalCached, edge.cpp:169-219 (channel-wise; the callers here need a scalar for geometry terms, a spectrum otherwise)
    V3 edgeEvalCached(const Edge *e, const Vertex *pred, const Vertex *succ, unsigned what) const
    {
        V3 result(1.0);
        if (e->length == 0) {
            if (what & EValueImp) result = result * (pred->weight[EImportance] * pred->pdf[EImportance]);
            if (what & EValueRad) result = result * (succ->weight[ERadiance] * succ->pdf[ERadiance]);
        } else {
            if (what & EValueImp) {
                Float tmp = pred->pdf[EImportance];
                if (pred->isConnectable()) {
                    tmp *= e->length * e->length;
                    if (succ->isOnSurface()) tmp /= dot(succ->geometricNormal(), e->d);
                    if (pred->isOnSurface() && !(what & ECosineImp)) tmp /= dot(pred->shadingNormal(), e->d);
                }
                result = result * (pred->weight[EImportance] * std::abs(tmp));
            } else if ((what & ECosineImp) && pred->isOnSurface() && pred->isConnectable()) {
                result = result * std::abs(dot(pred->shadingNormal(), e->d));
            }
            if (what & EValueRad) {
                Float tmp = succ->pdf[ERadiance];
                if (succ->isConnectable()) {
                    tmp *= e->length * e->length;
                    if (pred->isOnSurface()) tmp /= dot(pred->geometricNormal(), e->d);
                    if (succ->isOnSurface() && !(what & ECosineRad)) tmp /= dot(succ->shadingNormal(), e->d);
                }
                result = result * (succ->weight[ERadiance] * std::abs(tmp));
            } else if ((what & ECosineRad) && succ->isOnSurface() && succ->isConnectable()) {
                result = result * std::abs(dot(succ->shadingNormal(), e->d));
            }
            if (what & EInverseSquareFalloff) result = result / (e->length * e->length);
            if (what & ETransmittance) result = result * (e->weight[EImportance] * e->pdf[EImportance]);
        }
        return result;
    }

    // ---- PathVertex -----------------------------------------------------------------------------------------------------------------
    // the ESurfaceInteraction branch shared by sampleNext and perturbDirection: the adjoint-BSDF factor for shading normals, vertex.cpp:213-221,611-619
    static void adjoint(Vertex *v, int mode, V3 wiL, V3 woL, Float wiDotGeoN, Float woDotGeoN)
    {
        if (mode == EImportance) v->weight[EImportance] = v->weight[EImportance] * std::abs((cosTheta(wiL) * woDotGeoN) / (cosTheta(woL) * wiDotGeoN));
        else v->weight[EImportance] = v->weight[EImportance] * std::abs((cosTheta(woL) * wiDotGeoN) / (cosTheta(wiL) * woDotGeoN));
    }
    void toArea(Vertex *v, int mode, const Vertex *pred, const Edge *predEdge, const Edge *succEdge, const Vertex *succ, V3 rayD) const
    {                                                                                      // vertex.cpp:294-307,663-676
        if (v->measure == ESolidAngle) {
            v->measure = EArea;
            v->pdf[mode] /= succEdge->length * succEdge->length;
            if (succ->isOnSurface()) v->pdf[mode] *= std::abs(dot(rayD, succ->geometricNormal()));
            if (predEdge->length != 0.0) {
                v->pdf[1 - mode] /= predEdge->length * predEdge->length;
                if (pred->isOnSurface()) v->pdf[1 - mode] *= std::abs(dot(predEdge->d, pred->geometricNormal()));
            }
        }
    }
    // PathVertex::sampleNext, vertex.cpp:35-310
    bool sampleNext(Vertex *v, const Vertex *pred, const Edge *predEdge, Edge *succEdge, Vertex *succ, int mode, bool russianRoulette, V3 *throughput)
    {
        Ray ray;
        *succEdge = Edge(); *succ = Vertex();
        v->rrWeight = 1.0;
        switch (v->type) {
        case EEmitterSupernode: {
            const Float sx = rng.next1D(), sy = rng.next1D();
            const V3 result = sampleEmitterPosition(c, succ->prec, sx, sy);
            if (isZero(result)) return false;
            v->weight[EImportance] = result;
            v->pdf[EImportance] = succ->prec.pdf;
            v->measure = succ->prec.measure;
            succ->type = EEmitterSample;
            succ->degenerate = false;                                                      // neither area nor point lights have EDeltaDirection
            succEdge->weight[EImportance] = V3(1.0);
            succEdge->pdf[EImportance] = 1.0;
            return true;
        }
        case EEmitterSample: {                                                             // :97-121, AreaLight::sampleDirection area.cpp:114-122
            const Float sx = rng.next1D(), sy = rng.next1D();
            if (c.sc.emitters[v->prec.object].numTris < 0) {                               // PointEmitter::sampleDirection, point.cpp:97-105: not EOnSurface, no cosine
                const Float z = 1.0 - 2.0 * sy, r = safe_sqrt(1.0 - z * z);                 // warp::squareToUniformSphere, warp.cpp:25-31
                const Float sinPhi = std::sin(2.0 * PI * sx), cosPhi = std::cos(2.0 * PI * sx);
                const V3 d(r * cosPhi, r * sinPhi, z);
                const V3 result(1.0);
                v->weight[EImportance] = result;
                v->weight[ERadiance] = result * INV_FOURPI;
                v->pdf[EImportance] = INV_FOURPI;
                v->pdf[ERadiance] = 1.0;
                v->measure = ESolidAngle;
                ray = Ray(v->prec.p, d);
                break;
            }
            if (isEnvMap(c.sc, v->prec.object)) {                                          // EnvironmentMap::sampleDirection, envmap.cpp:455-474
                V3 value, dl; Float dpdf;
                envMapSampleDirection(c.sc.envMap, sx, sy, dl, value, dpdf);
                const V3 d = mul3(c.sc.envMap.toWorld, -dl);
                if (isZero(value) || dpdf == 0) return false;                               // ("be wary of roundoff errors": Spectrum(0) -> sampleNext fails, vertex.cpp:106-107)
                const V3 result = (value * c.sc.envMap.normalization) / (dpdf * c.sc.envMap.scale);
                if (isZero(result)) return false;
                v->weight[EImportance] = result;
                v->weight[ERadiance] = result * dpdf * (1.0 / std::abs(dot(d, v->prec.n)));   // (EOnSurface, envmap.cpp:107)
                v->pdf[EImportance] = dpdf;
                v->pdf[ERadiance] = 1.0;
                v->measure = ESolidAngle;
                ray = Ray(v->prec.p, d);
                break;
            }
            const V3 local = squareToCosineHemisphere(sx, sy);
            Frame fr; fr.n = v->prec.n; coordinateSystem(fr.n, fr.s, fr.t);
            const V3 d = fr.toWorld(local);
            const Float dpdf = INV_PI * cosTheta(local);                                   // squareToCosineHemispherePdf, warp.h
            const V3 result(1.0);
            v->weight[EImportance] = result;
            v->weight[ERadiance] = result * dpdf * (1.0 / std::abs(dot(d, v->prec.n)));
            v->pdf[EImportance] = dpdf;
            v->pdf[ERadiance] = 1.0;
            v->measure = ESolidAngle;
            ray = Ray(v->prec.p, d);
            break;
        }
        case ESurfaceInteraction: {                                                        // :149-231
            const Intersection &its = v->its;
            const V3 wi = normalize(pred->position() - its.p);
            const V3 wiL = its.sh.toLocal(wi);
            const Float sx = rng.next1D(), sy = rng.next1D();
            const BSDFSample bs = bsdfSample(v->mat, wiL, sx, sy, mode == EImportance);
            v->weight[mode] = bs.weight; v->pdf[mode] = bs.pdf;
            if (isZero(v->weight[mode])) return false;
            v->measure = (bs.sampledType & ESmooth) ? ESolidAngle : EDiscrete;             // BSDF::getMeasure, bsdf.h:313-324
            v->componentType = bs.sampledType;
            v->sampledComponentIndex = (bs.sampledType & EDeltaTransmission) ? 1 : 0;      // dielectric.cpp: component 1 is the transmission
            const V3 wo = its.sh.toWorld(bs.wo);
            const Float wiDotGeoN = dot(its.geoN, wi), woDotGeoN = dot(its.geoN, wo);
            if (wiDotGeoN * cosTheta(wiL) <= 0 || woDotGeoN * cosTheta(bs.wo) <= 0) return false;
            v->pdf[1 - mode] = bsdfPdf(v->mat, bs.wo, wiL, bsdfMeasure(v->measure));         // bRec.reverse()
            if (v->pdf[1 - mode] <= RCPOVERFLOW) return false;
            if (v->mat.type != MAT_DIELECTRIC) {                                             // (dielectric.cpp:87-98 is the one ENonSymmetric BSDF of the subset)
                v->weight[1 - mode] = v->weight[mode] * (v->pdf[mode] / v->pdf[1 - mode]);
                if (v->measure == ESolidAngle) v->weight[1 - mode] = v->weight[1 - mode] * std::abs(cosTheta(wiL) / cosTheta(bs.wo));   // (of the REVERSED record: wo = old wi)
            } else v->weight[1 - mode] = bsdfEval(v->mat, bs.wo, wiL, bsdfMeasure(v->measure), (1 - mode) == EImportance) / v->pdf[1 - mode];   // bRec.reverse() flips the mode too
            adjoint(v, mode, wiL, bs.wo, wiDotGeoN, woDotGeoN);
            if (throughput && mode == ERadiance && bs.eta != 1) *throughput = *throughput * (bs.eta * bs.eta);   // "for BDPT & russian roulette, track radiance * eta^2", :225-227
            ray = Ray(its.p, wo);
            break;
        }
        default:
            return false;                                                                  // (the sensor endpoints go through sampleSensor)
        }
        if (throughput) {
            *throughput = *throughput * v->weight[mode];
            if (russianRoulette) {
                const Float q = std::min(maxc(*throughput), (Float)0.95f);
                if (rng.next1D() > q) { v->measure = EInvalidMeasure; return false; }
                v->rrWeight = 1.0 / q;
                *throughput = *throughput * v->rrWeight;
            }
        }
        if (!edgeSampleNext(succEdge, ray, succ, mode)) { v->measure = EInvalidMeasure; return false; }
        toArea(v, mode, pred, predEdge, succEdge, succ, ray.d);
        return true;
    }
    // PathVertex::sampleSensor, vertex.cpp:312-384 (perspective: the direction sample maps to pixels, no aperture sample)
    int sampleSensor(Vertex *v0, int px, int py, Edge *e0, Vertex *v1, Edge *e1, Vertex *v2)
    {
        *e0 = Edge(); *v1 = Vertex();
        const Float sx = rng.next1D(), sy = rng.next1D();
        PRec &pRec = v1->prec;
        pRec = PRec();
        pRec.p = c.camPos; pRec.n = c.camDir; pRec.pdf = 1.0; pRec.measure = EDiscrete; pRec.object = -2;   // samplePosition, perspective.cpp:300-308
        V3 apertureP(0.0);
        if (c.thinlens) {                                                                  // the aperture sample (:324-325) and ThinLensCamera::samplePosition, thinlens.cpp:363-376
            const Float ax = rng.next1D(), ay = rng.next1D();
            Float tx, ty;
            squareToUniformDiskConcentric(ax, ay, tx, ty);
            apertureP = V3(tx * c.sc.cam.apertureRadius, ty * c.sc.cam.apertureRadius, 0.0);
            const double *M = c.sc.cam.toWorld;
            pRec.p = V3(M[0] * apertureP.x + M[1] * apertureP.y + M[2] * apertureP.z + M[3], M[4] * apertureP.x + M[5] * apertureP.y + M[6] * apertureP.z + M[7],
                        M[8] * apertureP.x + M[9] * apertureP.y + M[10] * apertureP.z + M[11]);
            pRec.pdf = c.aperturePdf; pRec.measure = EArea;
        }
        v0->weight[ERadiance] = V3(1.0);
        v0->pdf[ERadiance] = pRec.pdf;
        v0->measure = pRec.measure;
        v0->rrWeight = 1.0;
        v1->type = ESensorSample;
        v1->degenerate = false;
        e0->weight[ERadiance] = V3(1.0);
        e0->pdf[ERadiance] = 1.0;
        // sampleDirection, perspective.cpp:318-345
        const Float spx = (px + sx) * (1.0 / c.sc.cam.width), spy = (py + sy) * (1.0 / c.sc.cam.height);
        pRec.uvx = spx * c.sc.cam.width; pRec.uvy = spy * c.sc.cam.height;
        V3 dl = sampleToCameraDir(c, spx, spy);
        if (c.thinlens) {                                                                  // ThinLensCamera::sampleDirection, thinlens.cpp:386-418: through the pixel's point of the focus plane
            const gpo_camera &cam = c.sc.cam;
            V3 nearP((1 - 2 * spx) * cam.nearClip * c.sc.tanHalf, (1 - 2 * spy) / c.sc.aspect * cam.nearClip * c.sc.tanHalf, cam.nearClip);
            nearP.x = nearP.x * (cam.focusDistance / nearP.z);
            nearP.y = nearP.y * (cam.focusDistance / nearP.z);
            nearP.z = cam.focusDistance;
            dl = normalize(nearP - camToLocalPoint(c, pRec.p));                            // (apertureP = trafo.inverse().transformAffine(pRec.p): the round trip of the reference)
        }
        const V3 d = camToWorld(c, dl);
        const Float dpdf = c.normalization / (dl.z * dl.z * dl.z);
        *e1 = Edge(); *v2 = Vertex();
        v1->weight[EImportance] = V3(1.0) * dpdf * (1.0 / std::abs(dot(d, pRec.n)));
        v1->weight[ERadiance] = V3(1.0);
        v1->pdf[EImportance] = 1.0;
        v1->pdf[ERadiance] = dpdf;
        v1->rrWeight = 1.0;
        v1->measure = ESolidAngle;
        Ray ray(pRec.p, d);
        if (!edgeSampleNext(e1, ray, v2, ERadiance)) { v1->measure = EInvalidMeasure; return 1; }
        if (v1->measure == ESolidAngle) {
            v1->measure = EArea;
            v1->pdf[ERadiance] /= e1->length * e1->length;
            if (v2->isOnSurface()) v1->pdf[ERadiance] *= std::abs(dot(ray.d, v2->geometricNormal()));
        }
        return 2;
    }
    // PathVertex::perturbDirection, vertex.cpp:488-679
    bool perturbDirection(Vertex *v, const Vertex *pred, const Edge *predEdge, Edge *succEdge, Vertex *succ, V3 d, Float dist, int mode)
    {
        Ray ray(v->position(), d);
        *succEdge = Edge(); *succ = Vertex();
        succ->measure = EInvalidMeasure;
        if (v->degenerate) return false;
        switch (v->type) {
        case ESensorSample: {                                                              // :526-547
            const Float value = sensorDirection(c, v->prec.p, d), prob = value;             // evalDirection == pdfDirection, perspective.cpp:373-391
            if (value == 0 || prob <= RCPOVERFLOW) return false;
            v->weight[EImportance] = V3(value) * (1.0 / std::abs(dot(d, v->prec.n)));
            v->weight[ERadiance] = V3(value) / prob;
            v->pdf[EImportance] = 1.0;
            v->pdf[ERadiance] = prob;
            v->measure = ESolidAngle;
            break;
        }
        case ESurfaceInteraction: {                                                        // :549-621
            const Intersection &its = v->its;
            const V3 wi = normalize(pred->position() - its.p), wo = d;
            const V3 wiL = its.sh.toLocal(wi), woL = its.sh.toLocal(wo);
            const V3 value = bsdfEval(v->mat, wiL, woL, MEASURE_SOLID_ANGLE);
            const Float prob = bsdfPdf(v->mat, wiL, woL, MEASURE_SOLID_ANGLE);
            if (isZero(value) || prob <= RCPOVERFLOW) return false;
            v->weight[mode] = value / prob;
            v->pdf[mode] = prob;
            const Float wiDotGeoN = dot(its.geoN, wi), woDotGeoN = dot(its.geoN, wo);
            if (wiDotGeoN * cosTheta(wiL) <= 0 || woDotGeoN * cosTheta(woL) <= 0) return false;
            v->measure = ESolidAngle;
            v->componentType = ESmooth;
            v->pdf[1 - mode] = bsdfPdf(v->mat, woL, wiL, MEASURE_SOLID_ANGLE);
            if (v->pdf[1 - mode] <= RCPOVERFLOW) return false;
            v->weight[1 - mode] = v->weight[mode] * std::abs((v->pdf[mode] * cosTheta(wiL)) / (v->pdf[1 - mode] * cosTheta(woL)));   // (reversed record)
            adjoint(v, mode, wiL, woL, wiDotGeoN, woDotGeoN);
            break;
        }
        default:
            return false;                                                                  // (an emitter sample is never perturbed on the G-BDPT path: mode is always ERadiance)
        }
        if (!edgePerturbDirection(succEdge, ray, dist, succ, mode)) { v->measure = EInvalidMeasure; return false; }
        toArea(v, mode, pred, predEdge, succEdge, succ, ray.d);
        return true;
    }
    // PathVertex::eval, vertex.cpp:781-913
    V3 eval(const Vertex *v, const Vertex *pred, const Vertex *succ, int mode, int measure = EArea) const
    {
        switch (v->type) {
        case EEmitterSupernode:
            if (mode != EImportance || pred != nullptr || succ->type != EEmitterSample) return V3(0.0);
            if (c.sc.emitters[succ->prec.object].numTris < 0)                              // PointEmitter::evalPosition, point.cpp:89-91
                return measure == EDiscrete ? c.sc.emitters[succ->prec.object].radiance * (4 * PI) : V3(0.0);
            if (isEnvMap(c.sc, succ->prec.object)) return V3(envMapPower(c.sc) * envInvSurfaceArea(c.sc));   // EnvironmentMap::evalPosition, envmap.cpp:424-426
            return c.sc.emitters[succ->prec.object].radiance * PI;                         // AreaLight::evalPosition, area.cpp:99-101
        case ESensorSupernode:
            if (mode != ERadiance || pred != nullptr || succ->type != ESensorSample) return V3(0.0);
            if (c.thinlens) return V3(measure == EArea ? c.aperturePdf : 0.0);             // thinlens.cpp:378-380
            return V3(measure == EDiscrete ? 1.0 : 0.0);                                   // perspective.cpp:310-312
        case EEmitterSample: {
            V3 target;
            if (mode == EImportance && pred->type == EEmitterSupernode) target = succ->position();
            else if (mode == ERadiance && succ->type == EEmitterSupernode) target = pred->position();
            else return V3(0.0);
            const V3 wo = normalize(target - v->prec.p);
            V3 result = emitterEvalDirection(c, v->prec, wo, measure == EArea ? ESolidAngle : measure);
            const Float dp = std::abs(dot(v->prec.n, wo));
            if (measure != EDiscrete && dp != 0) result = result / dp;
            return result;
        }
        case ESensorSample: {
            V3 target;
            if (mode == ERadiance && pred->type == ESensorSupernode) target = succ->position();
            else if (mode == EImportance && succ->type == ESensorSupernode) target = pred->position();
            else return V3(0.0);
            const V3 wo = normalize(target - v->prec.p);
            V3 result((measure == EArea ? ESolidAngle : measure) != ESolidAngle ? 0.0 : sensorDirection(c, v->prec.p, wo));
            const Float dp = std::abs(dot(v->prec.n, wo));
            if (measure != EDiscrete && dp != 0) result = result / dp;
            return result;
        }
        case ESurfaceInteraction: {
            const Intersection &its = v->its;
            const V3 wi = normalize(pred->position() - its.p), wo = normalize(succ->position() - its.p);
            const V3 wiL = its.sh.toLocal(wi), woL = its.sh.toLocal(wo);
            if (measure == EArea) measure = ESolidAngle;
            V3 result = bsdfEval(v->mat, wiL, woL, bsdfMeasure(measure), mode == EImportance);
            const Float wiDotGeoN = dot(its.geoN, wi), woDotGeoN = dot(its.geoN, wo);
            if (wiDotGeoN * cosTheta(wiL) <= 0 || woDotGeoN * cosTheta(woL) <= 0) return V3(0.0);
            if (mode == EImportance) result = result * std::abs((cosTheta(wiL) * woDotGeoN) / (cosTheta(woL) * wiDotGeoN));
            if (measure != EDiscrete && cosTheta(woL) != 0) result = result / std::abs(cosTheta(woL));
            return result;
        }
        }
        return V3(0.0);
    }
    // PathVertex::evalPdf, vertex.cpp:915-1022
    Float evalPdf(const Vertex *v, const Vertex *pred, const Vertex *succ, int mode, int measure = EArea) const
    {
        V3 wo(0.0);
        Float dist = 0.0, result = 0.0;
        switch (v->type) {
        case EEmitterSupernode:
            if (mode != EImportance || pred != nullptr || succ->type != EEmitterSample) return 0.0;
            return pdfEmitterPosition(c, succ->prec, measure);
        case ESensorSupernode:
            if (mode != ERadiance || pred != nullptr || succ->type != ESensorSample) return 0.0;
            if (c.thinlens) return measure == EArea ? c.aperturePdf : 0.0;                 // thinlens.cpp:382-384
            return measure == EDiscrete ? 1.0 : 0.0;                                       // perspective.cpp:314-316
        case EEmitterSample:
            if (mode == ERadiance && succ->type == EEmitterSupernode) return 1.0;
            else if (mode != EImportance || pred->type != EEmitterSupernode) return 0.0;
            wo = succ->position() - v->prec.p;
            dist = length(wo); wo = wo / dist;
            result = emitterDirection(c, v->prec, wo, measure == EArea ? ESolidAngle : measure);
            break;
        case ESensorSample:
            if (mode == EImportance && succ->type == ESensorSupernode) return 1.0;
            else if (mode != ERadiance || pred->type != ESensorSupernode) return 0.0;
            wo = succ->position() - v->prec.p;
            dist = length(wo); wo = wo / dist;
            result = (measure == EArea ? ESolidAngle : measure) != ESolidAngle ? 0.0 : sensorDirection(c, v->prec.p, wo);
            break;
        case ESurfaceInteraction: {
            const Intersection &its = v->its;
            wo = succ->position() - its.p;
            dist = length(wo); wo = wo / dist;
            const V3 wi = normalize(pred->position() - its.p);
            const V3 wiL = its.sh.toLocal(wi), woL = its.sh.toLocal(wo);
            result = bsdfPdf(v->mat, wiL, woL, bsdfMeasure(measure == EArea ? ESolidAngle : measure));
            const Float wiDotGeoN = dot(its.geoN, wi), woDotGeoN = dot(its.geoN, wo);
            if (wiDotGeoN * cosTheta(wiL) <= 0 || woDotGeoN * cosTheta(woL) <= 0) return 0.0;
            break;
        }
        default:
            return 0.0;
        }
        if (measure == EArea) {
            result /= dist * dist;
            if (succ->isOnSurface()) result *= std::abs(dot(wo, succ->geometricNormal()));
        }
        return result;
    }
    // PathVertex::cast, vertex.cpp:1115-1163 (no sensor shapes: a cast to ESensorSample of a surface vertex always fails)
    bool cast(Vertex *v, int desired) const
    {
        if (desired == v->type) return true;
        if (desired == EEmitterSample) {
            if (v->type != ESurfaceInteraction) return false;
            const int em = emitterOfPrim(c.sc, v->its.prim);
            if (em < 0) return false;
            v->type = desired;
            PRec pRec;                                                                     // PositionSamplingRecord(its), records.inl:154-155
            pRec.p = v->its.p; pRec.n = v->its.sh.n; pRec.measure = EArea; pRec.uvx = v->its.u; pRec.uvy = v->its.v;
            pRec.object = em; pRec.pdf = 0.0;
            v->prec = pRec;
            v->measure = pRec.measure;
            v->degenerate = false;
            return true;
        }
        return false;
    }
    // PathVertex::update, vertex.cpp:1165-1211
    bool update(Vertex *v, const Vertex *pred, const Vertex *succ, int mode, int measure) const
    {
        v->pdf[mode] = evalPdf(v, pred, succ, mode, measure);
        v->pdf[1 - mode] = evalPdf(v, succ, pred, 1 - mode, measure);
        v->weight[mode] = eval(v, pred, succ, mode, measure);
        v->weight[1 - mode] = eval(v, succ, pred, 1 - mode, measure);
        if (isZero(v->weight[mode]) || v->pdf[mode] <= RCPOVERFLOW) return false;
        Float weightFwd = v->pdf[mode] <= RCPOVERFLOW ? 0.0 : 1 / v->pdf[mode], weightBkw = v->pdf[1 - mode] <= RCPOVERFLOW ? 0.0 : 1 / v->pdf[1 - mode];
        v->measure = measure;
        if (!v->isSupernode() && measure == EArea) {
            if (!pred->isSupernode()) {
                V3 d = pred->position() - v->position();
                const Float invDistSqr = 1.0 / lengthSquared(d);
                weightBkw *= invDistSqr;
                d = d * std::sqrt(invDistSqr);
                if (v->isOnSurface() && v->isConnectable()) weightBkw *= std::abs(dot(v->shadingNormal(), d));
                if (pred->isOnSurface()) weightBkw *= std::abs(dot(pred->geometricNormal(), d));
            }
            if (!succ->isSupernode()) {
                V3 d = succ->position() - v->position();
                const Float invDistSqr = 1.0 / lengthSquared(d);
                weightFwd *= invDistSqr;
                d = d * std::sqrt(invDistSqr);
                if (v->isOnSurface() && v->isConnectable()) weightFwd *= std::abs(dot(v->shadingNormal(), d));
                if (succ->isOnSurface()) weightFwd *= std::abs(dot(succ->geometricNormal(), d));
            }
            if (v->isSurface()) v->componentType = ESmooth;
        }
        v->weight[mode] = v->weight[mode] * weightFwd;
        v->weight[1 - mode] = v->weight[1 - mode] * weightBkw;
        return true;
    }
    // PathVertex::connect with explicit measures, vertex.cpp:1348-1370
    bool connect(const Vertex *pred, Vertex *vs, Edge *edge, Vertex *vt, const Vertex *succ, int vsMeasure, int vtMeasure) const
    {
        if (vs->type == EEmitterSupernode) { if (!cast(vt, EEmitterSample)) return false; }
        else if (vt->type == ESensorSupernode) { if (!cast(vs, ESensorSample)) return false; }
        if (!update(vs, pred, vt, EImportance, vsMeasure)) return false;
        if (!update(vt, succ, vs, ERadiance, vtMeasure)) return false;
        return edgeConnect(edge, vs, vt);
    }
    bool getSamplePosition(const Vertex *v, const Vertex *other, Float &ox, Float &oy) const   // vertex.cpp:1308-1316
    {
        return sensorSamplePosition(c, v->position(), other->position() - v->position(), ox, oy);
    }
    bool updateSamplePosition(Vertex *v, const Vertex *other) const                        // :1298-1306
    {
        return sensorSamplePosition(c, v->position(), other->position() - v->position(), v->prec.uvx, v->prec.uvy);
    }

    // ---- Path ------------------------------------------------------------------------------------------------------------------------
    // Path::alternatingRandomWalkFromPixel, path.cpp:548-631
    void alternatingRandomWalkFromPixel(Path &emitterPath, int nEmitterSteps, Path &sensorPath, int nSensorSteps, int px, int py, int rrStart)
    {
        Vertex *curVertexS = emitterPath.v[0], *curVertexT = sensorPath.v[0], *predVertexS = nullptr, *predVertexT = nullptr;
        Edge *predEdgeS = nullptr, *predEdgeT = nullptr;
        Vertex *v1 = pool.allocVertex(), *v2 = pool.allocVertex();
        Edge *e0 = pool.allocEdge(), *e1 = pool.allocEdge();
        int t = sampleSensor(curVertexT, px, py, e0, v1, e1, v2);
        if (t >= 1) { sensorPath.e.push_back(e0); sensorPath.v.push_back(v1); }
        if (t == 2) { sensorPath.e.push_back(e1); sensorPath.v.push_back(v2); predVertexT = v1; curVertexT = v2; predEdgeT = e1; }
        else curVertexT = nullptr;
        V3 throughputS(1.0), throughputT(1.0);
        int s = 0;
        do {
            if (curVertexT && (t < nSensorSteps || nSensorSteps == -1)) {
                Vertex *succVertexT = pool.allocVertex(); Edge *succEdgeT = pool.allocEdge();
                if (sampleNext(curVertexT, predVertexT, predEdgeT, succEdgeT, succVertexT, ERadiance, rrStart != -1 && t >= rrStart, &throughputT)) {
                    sensorPath.e.push_back(succEdgeT); sensorPath.v.push_back(succVertexT);
                    predVertexT = curVertexT; curVertexT = succVertexT; predEdgeT = succEdgeT;
                    t++;
                } else curVertexT = nullptr;
            } else curVertexT = nullptr;
            if (curVertexS && (s < nEmitterSteps || nEmitterSteps == -1)) {
                Vertex *succVertexS = pool.allocVertex(); Edge *succEdgeS = pool.allocEdge();
                if (sampleNext(curVertexS, predVertexS, predEdgeS, succEdgeS, succVertexS, EImportance, rrStart != -1 && s >= rrStart, &throughputS)) {
                    emitterPath.e.push_back(succEdgeS); emitterPath.v.push_back(succVertexS);
                    predVertexS = curVertexS; curVertexS = succVertexS; predEdgeS = succEdgeS;
                    s++;
                } else curVertexS = nullptr;
            } else curVertexS = nullptr;
        } while (curVertexS || curVertexT);
    }

    // PathVertex::propagatePerturbation, vertex.cpp:681-790: re-creates a SPECULAR vertex of a perturbed chain -- the delta component named by
    // componentType is taken deterministically (the sample (0.5, 0.5) is never consulted by a one-component request) and its ray traced to `dist`
    bool propagatePerturbation(Vertex *v, const Vertex *pred, const Edge *predEdge, Edge *succEdge, Vertex *succ, int componentType, Float dist, int mode)
    {
        const Intersection &its = v->its;
        if (!(bsdfType(v->mat) & EDelta)) return false;
        *succEdge = Edge(); *succ = Vertex();
        const V3 wi = normalize(pred->position() - its.p);
        const V3 wiL = its.sh.toLocal(wi);
        const BSDFSample bs = bsdfSample(v->mat, wiL, 0.5, 0.5, mode == EImportance, componentType);
        if (isZero(bs.weight)) return false;
        const V3 wo = its.sh.toWorld(bs.wo);
        const Float wiDotGeoN = dot(its.geoN, wi), woDotGeoN = dot(its.geoN, wo);
        if (wiDotGeoN * cosTheta(wiL) <= 0 || woDotGeoN * cosTheta(bs.wo) <= 0) return false;
        const Float prob = bsdfPdf(v->mat, wiL, bs.wo, MEASURE_DISCRETE);                   // bRec.typeMask = BSDF::EAll
        if (prob <= RCPOVERFLOW) return false;
        v->weight[mode] = bsdfEval(v->mat, wiL, bs.wo, MEASURE_DISCRETE, mode == EImportance) / prob;
        v->pdf[mode] = prob;
        v->measure = EDiscrete;
        v->componentType = componentType;
        if (isZero(v->weight[mode]) || prob <= RCPOVERFLOW) return false;
        v->pdf[1 - mode] = bsdfPdf(v->mat, bs.wo, wiL, MEASURE_DISCRETE);
        if (v->pdf[1 - mode] <= RCPOVERFLOW) return false;
        if (v->mat.type != MAT_DIELECTRIC) v->weight[1 - mode] = v->weight[mode];
        else v->weight[1 - mode] = bsdfEval(v->mat, bs.wo, wiL, MEASURE_DISCRETE, (1 - mode) == EImportance) / v->pdf[1 - mode];
        adjoint(v, mode, wiL, bs.wo, wiDotGeoN, woDotGeoN);
        (void)predEdge;
        Ray ray(its.p, wo);
        if (!edgePerturbDirection(succEdge, ray, dist, succ, mode)) { v->measure = EInvalidMeasure; return false; }
        return true;
    }

    // ---- SpecularManifold, src/libbidir/manifold.cpp (surface interactions and position-pinned endpoints: no media, no directional endpoints) ----
    struct M2 {                                     // Matrix2x2, core/matrix.h:455-530
        Float m[2][2];
        M2() { m[0][0] = m[0][1] = m[1][0] = m[1][1] = 0; }
        M2(Float a, Float b, Float cc, Float d) { m[0][0] = a; m[0][1] = b; m[1][0] = cc; m[1][1] = d; }
        void setZero() { m[0][0] = m[0][1] = m[1][0] = m[1][1] = 0; }
        void setIdentity() { m[0][0] = m[1][1] = 1; m[0][1] = m[1][0] = 0; }
        Float det() const { return m[0][0] * m[1][1] - m[0][1] * m[1][0]; }
        bool invert(M2 &t) const
        {
            const Float d = m[0][0] * m[1][1] - m[0][1] * m[1][0];
            if (std::abs(d) <= RCPOVERFLOW) return false;
            const Float invDet = 1 / d;
            t.m[0][0] = m[1][1] * invDet; t.m[0][1] = -m[0][1] * invDet; t.m[1][1] = m[0][0] * invDet; t.m[1][0] = -m[1][0] * invDet;
            return true;
        }
        M2 operator*(const M2 &o) const
        {
            M2 r;
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { Float sum = 0; for (int k = 0; k < 2; ++k) sum += m[i][k] * o.m[k][j]; r.m[i][j] = sum; }   // Matrix::operator*, matrix.h
            return r;
        }
        M2 operator-(const M2 &o) const { return M2(m[0][0] - o.m[0][0], m[0][1] - o.m[0][1], m[1][0] - o.m[1][0], m[1][1] - o.m[1][1]); }
        M2 operator-() const { return M2(-m[0][0], -m[0][1], -m[1][0], -m[1][1]); }
    };
    enum { EPinnedPosition = 0, EReflection = 2, ERefraction = 3, EMovable = 5 };             // manifold.h:88-95
    struct SimpleVertex {                           // manifold.h:98-134
        bool degenerate = false;
        int type = EPinnedPosition;
        V3 p, dpdu, dpdv, n, gn, dndu, dndv, m;
        Float eta = 1.0;
        int object = -1;                            // the BSDF: here the material index
        M2 a, b, c, u, Tp;
        SimpleVertex() {}
        SimpleVertex(int t, V3 pos) : type(t), p(pos), dpdu(0.0), dpdv(0.0), n(0.0), gn(0.0), dndu(0.0), dndv(0.0), m(0.0) {}
        V3 map(Float uu, Float vv) const { const Float tx = Tp.m[0][0] * uu + Tp.m[0][1] * vv, ty = Tp.m[1][0] * uu + Tp.m[1][1] * vv; return dpdu * tx + dpdv * ty; }
    };
    std::vector<SimpleVertex> mVerts, mProposal;
    int mIterations = 0;
    // TriMesh::getNormalDerivative, trimesh.cpp:745-822 (shadingFrame = true)
    void normalDerivative(const Intersection &its, V3 &dndu, V3 &dndv) const
    {
        const Tri &tr = c.sc.tris[its.prim];
        dndu = dndv = V3(0.0);
        if (!tr.hasNormals) return;
        const V3 rel = its.p - tr.p0, du = tr.p1 - tr.p0, dv = tr.p2 - tr.p0;
        const Float b1 = dot(du, rel), b2 = dot(dv, rel), a11 = dot(du, du), a12 = dot(du, dv), a22 = dot(dv, dv);
        Float det = a11 * a22 - a12 * a12;
        if (det == 0) return;
        Float invDet = 1.0 / det;
        const Float u = (a22 * b1 - a12 * b2) * invDet, v = (-a12 * b1 + a11 * b2) * invDet, w = 1 - u - v;
        V3 N = tr.n1 * u + tr.n2 * v + tr.n0 * w;
        const Float il = 1.0 / length(N); N = N * il;
        dndu = (tr.n1 - tr.n0) * il; dndu = dndu - N * dot(N, dndu);
        dndv = (tr.n2 - tr.n0) * il; dndv = dndv - N * dot(N, dndv);
        if (tr.hasUV) {
            const Float d1x = tr.uv[2] - tr.uv[0], d1y = tr.uv[3] - tr.uv[1], d2x = tr.uv[4] - tr.uv[0], d2y = tr.uv[5] - tr.uv[1];
            det = d1x * d2y - d1y * d2x;
            if (det == 0) { dndu = dndv = V3(0.0); return; }
            invDet = 1.0 / det;
            const V3 du_ = (dndu * d2y - dndv * d1y) * invDet, dv_ = (dndv * d1x - dndu * d2x) * invDet;
            dndu = du_; dndv = dv_;
        }
    }
    // the surface part of SimpleVertex from an intersection: position, normals, an orthonormal parameterization at p (manifold.cpp:101-122,480-507)
    void manifoldSurface(SimpleVertex &v, const Intersection &its) const
    {
        v.p = its.p; v.gn = its.geoN; v.n = its.sh.n; v.dpdu = its.dpdu; v.dpdv = its.dpdv;
        normalDerivative(its, v.dndu, v.dndv);
        Float invLen = 1 / length(v.dpdu);
        v.dpdu = v.dpdu * invLen; v.dndu = v.dndu * invLen;
        const Float dp = dot(v.dpdu, v.dpdv);
        const V3 dpdv = v.dpdv - v.dpdu * dp, dndv = v.dndv - v.dndu * dp;
        invLen = 1 / length(dpdv);
        v.dpdv = dpdv * invLen; v.dndv = dndv * invLen;
    }
    // SpecularManifold::init, manifold.cpp:59-170
    bool manifoldInit(const Path &path, int start, int end)
    {
        const int step = start < end ? 1 : -1;
        if (path.v[start]->isSupernode()) start += step;
        if (path.v[end]->isSupernode()) end -= step;
        const Vertex *vs = path.v[start], *ve = path.v[end];
        mVerts.clear();
        mVerts.push_back(SimpleVertex(EPinnedPosition, vs->position()));   // (area lights and the perspective sensor: no EDeltaDirection endpoint)
        for (int i = start + step; i != end; i += step) {
            const Vertex *pred = path.v[i - step], *vertex = path.v[i], *succ = path.v[i + step];
            SimpleVertex v(EPinnedPosition, V3(0.0));
            if (!vertex->isSurface()) return false;
            manifoldSurface(v, vertex->its);
            v.object = c.sc.tris[vertex->its.prim].material;
            v.degenerate = !vertex->isConnectable();
            const V3 wPred = pred->position() - v.p, wSucc = succ->position() - v.p;
            if (dot(v.gn, wPred) * dot(v.gn, wSucc) < 0) { v.type = ERefraction; v.eta = getEta(vertex->mat); }
            else { v.type = EReflection; v.eta = 1.0; }
            mVerts.push_back(v);
        }
        mVerts.push_back(SimpleVertex(EMovable, ve->position()));
        return true;
    }
    // SpecularManifold::computeTangents, manifold.cpp:172-400
    bool manifoldTangents()
    {
        const int n = (int)mVerts.size() - 1;
        mVerts[0].Tp.setZero();
        mVerts[mVerts.size() - 1].Tp.setIdentity();
        if (mVerts.size() == 2) return true;
        for (int i = 0; i < n; ++i) {
            SimpleVertex *v = &mVerts[i];
            V3 wo = v[1].p - v[0].p;
            Float ilo = length(wo);
            if (ilo == 0) return false;
            ilo = 1 / ilo; wo = wo * ilo;
            if (v[0].type == EPinnedPosition) { v[0].a.setZero(); v[0].b.setIdentity(); v[0].c.setZero(); continue; }
            V3 wi = v[-1].p - v[0].p;
            Float ili = length(wi);
            if (ili == 0) return false;
            ili = 1 / ili; wi = wi * ili;
            if (v[0].type != EReflection && v[0].type != ERefraction) return false;
            Float eta = v[0].eta;
            const bool normalizeH = !(v[0].type == ERefraction && eta == 1);
            V3 H;
            Float ilh;
            if (normalizeH) {
                if (dot(wi, v[0].gn) < 0) eta = 1 / eta;
                H = wi + wo * eta;
                ilh = 1 / length(H);
                H = H * ilh;
            } else { H = wi + wo; ilh = 1.0; }
            const Float dot_H_n = dot(v[0].n, H), dot_H_dndu = dot(v[0].dndu, H), dot_H_dndv = dot(v[0].dndv, H), dot_u_n = dot(v[0].dpdu, v[0].n), dot_v_n = dot(v[0].dpdv, v[0].n);
            V3 s_ = v[0].dpdu - v[0].n * dot_u_n, t_ = v[0].dpdv - v[0].n * dot_v_n;
            ilo *= eta * ilh; ili *= ilh;
            V3 dH_du = (v[-1].dpdu - wi * dot(wi, v[-1].dpdu)) * ili, dH_dv = (v[-1].dpdv - wi * dot(wi, v[-1].dpdv)) * ili;
            if (normalizeH) { dH_du = dH_du - H * dot(dH_du, H); dH_dv = dH_dv - H * dot(dH_dv, H); }
            v[0].a = M2(dot(dH_du, s_), dot(dH_dv, s_), dot(dH_du, t_), dot(dH_dv, t_));
            dH_du = -v[0].dpdu * (ili + ilo) + wi * (dot(wi, v[0].dpdu) * ili) + wo * (dot(wo, v[0].dpdu) * ilo);
            dH_dv = -v[0].dpdv * (ili + ilo) + wi * (dot(wi, v[0].dpdv) * ili) + wo * (dot(wo, v[0].dpdv) * ilo);
            if (normalizeH) { dH_du = dH_du - H * dot(dH_du, H); dH_dv = dH_dv - H * dot(dH_dv, H); }
            v[0].b = M2(dot(dH_du, s_) - dot(v[0].dpdu, v[0].dndu) * dot_H_n - dot_u_n * dot_H_dndu,
                        dot(dH_dv, s_) - dot(v[0].dpdu, v[0].dndv) * dot_H_n - dot_u_n * dot_H_dndv,
                        dot(dH_du, t_) - dot(v[0].dpdv, v[0].dndu) * dot_H_n - dot_v_n * dot_H_dndu,
                        dot(dH_dv, t_) - dot(v[0].dpdv, v[0].dndv) * dot_H_n - dot_v_n * dot_H_dndv);
            dH_du = (v[1].dpdu - wo * dot(wo, v[1].dpdu)) * ilo;
            dH_dv = (v[1].dpdv - wo * dot(wo, v[1].dpdv)) * ilo;
            if (normalizeH) { dH_du = dH_du - H * dot(dH_du, H); dH_dv = dH_dv - H * dot(dH_dv, H); }
            v[0].c = M2(dot(dH_du, s_), dot(dH_dv, s_), dot(dH_du, t_), dot(dH_dv, t_));
            s_ = normalize(s_);
            t_ = cross(v[0].n, s_);
            v[0].m = V3(dot(s_, H), dot(t_, H), dot(v[0].n, H));
            if (dot(H, v[0].gn) < 0) v[0].m = -v[0].m;
        }
        M2 Li;
        if (!mVerts[0].b.invert(Li)) return false;
        for (int i = 0; i < n - 1; ++i) {
            mVerts[i].u = Li * mVerts[i].c;
            const M2 temp = mVerts[i + 1].b - mVerts[i + 1].a * mVerts[i].u;
            if (!temp.invert(Li)) return false;
        }
        mVerts[n - 1].Tp = -(Li * mVerts[n - 1].c);
        for (int i = n - 2; i >= 0; --i) mVerts[i].Tp = -(mVerts[i].u * mVerts[i + 1].Tp);
        return true;
    }
    static V3 reflectAbout(V3 wi, V3 n) { return n * (2 * dot(wi, n)) - wi; }                // util.cpp:763-765
    static V3 refractAbout(V3 wi, V3 n, Float eta)                                           // util.cpp:774-792
    {
        if (eta == 1) return -wi;
        const Float cosThetaI = dot(wi, n);
        if (cosThetaI > 0) eta = 1 / eta;
        const Float cosThetaTSqr = 1 - (1 - cosThetaI * cosThetaI) * (eta * eta);
        if (cosThetaTSqr <= 0.0) return V3(0.0);
        return n * (cosThetaI * eta - (cosThetaI < 0 ? -1.0 : (cosThetaI > 0 ? 1.0 : 0.0)) * std::sqrt(cosThetaTSqr)) - wi * eta;
    }
    // SpecularManifold::project, manifold.cpp:402-510
    bool manifoldProject(V3 d)
    {
        const SimpleVertex &last = mVerts[mVerts.size() - 1];
        const Float du = dot(d, last.dpdu), dv = dot(d, last.dpdv);
        Ray ray;
        Intersection its;
        mProposal.clear();
        for (size_t i = 0; i < mVerts.size(); ++i) {
            mProposal.push_back(mVerts[i]);
            SimpleVertex &vertex = mProposal[i];
            if (i == 0) {
                const V3 p0 = mVerts[0].p + mVerts[0].map(du, dv), p1 = mVerts[1].p + mVerts[1].map(du, dv);
                ray = Ray(p0, normalize(p1 - p0));
                vertex.p = ray.o;
                continue;
            } else if (vertex.type == EMovable) {
                const Float dp = dot(ray.d, vertex.n);
                if (std::abs(dp) < Epsilon) return false;
                const Float t = dot(vertex.p - ray.o, vertex.n) / dp;
                vertex.p = ray.o + ray.d * t;
                break;
            } else if (vertex.type == EReflection || vertex.type == ERefraction) {
                if (!rayIntersect(c.sc, ray, its)) return false;
                const V3 n = its.sh.n;
                V3 s_ = its.dpdu;
                s_ = normalize(s_ - n * dot(n, s_));
                const V3 t_ = cross(n, s_);
                const V3 m = s_ * vertex.m.x + t_ * vertex.m.y + n * vertex.m.z;
                V3 out;
                if (vertex.type == EReflection) out = reflectAbout(-ray.d, m);
                else {
                    Ray none;
                    out = refractAbout(-ray.d, m, getEta(matOf(c.sc, its, none)));
                    if (isZero(out)) return false;
                }
                ray = Ray(its.p, out);
            } else return false;
            if (vertex.object != c.sc.tris[its.prim].material) return false;
            manifoldSurface(vertex, its);
        }
        return true;
    }
    // SpecularManifold::move, manifold.cpp:512-635
    bool manifoldMove(V3 target, V3 n)
    {
        SimpleVertex &last = mVerts[mVerts.size() - 1];
        if (mVerts.size() == 2 && mVerts[0].type == EPinnedPosition) return true;
        const Float invScale = 1.0 / std::max(std::max(std::abs(target.x), std::abs(target.y)), std::abs(target.z));
        Float stepSize = 1;
        coordinateSystem(n, last.dpdu, last.dpdv);
        last.n = n;
        mIterations = 0;
        while (mIterations < 20) {                                                           // MTS_MANIFOLD_MAX_ITERATIONS, manifold.h:27
            const V3 rel = target - mVerts[mVerts.size() - 1].p;
            Float dist = length(rel), newDist;
            if (dist * invScale < Epsilon) {                                                 // MTS_MANIFOLD_EPSILON
                dist = length(mVerts[mVerts.size() - 1].p - mVerts[mVerts.size() - 2].p);
                if (dist * invScale < Epsilon) return false;
                return true;
            }
            mIterations++;
            if (!manifoldTangents()) return false;
            bool failure = false;
            if (!manifoldProject(rel * stepSize)) failure = true;
            else {
                newDist = length(target - mProposal[mProposal.size() - 1].p);
                if (newDist > dist) failure = true;
            }
            if (!failure) {
                mProposal.swap(mVerts);
                stepSize = std::min((Float)1.0, stepSize * 2.0);
                continue;
            }
            stepSize /= 2.0;
        }
        return false;
    }
    // SpecularManifold::update, manifold.cpp:637-757 (start > end: the walk runs from c towards b, mode = ERadiance)
    bool manifoldUpdate(Path &path, int start, int end)
    {
        const int step = start < end ? 1 : -1, mode = start < end ? EImportance : ERadiance;
        const int last = (int)mVerts.size() - 2;
        for (int j = 0, i = start; j < last; ++j, i += step) {
            const SimpleVertex &v = mVerts[j], &vn = mVerts[j + 1];
            Vertex *pred = path.vertexOrNull(i - step), *vertex = path.v[i], *succ = path.v[i + step];
            const int predEdgeIdx = (mode == EImportance) ? i - step : i - step - 1;
            Edge *predEdge = path.edgeOrNull(predEdgeIdx), *succEdge = path.e[predEdgeIdx + step];
            V3 d = vn.p - v.p;
            const Float len = length(d);
            d = d / len;
            if (!v.degenerate) {
                if (!perturbDirection(vertex, pred, predEdge, succEdge, succ, d, len, mode)) return false;
            } else {
                const int compType = v.type == ERefraction ? EDeltaTransmission : EDeltaReflection;   // (no index-matched ENull components in the subset)
                if (!propagatePerturbation(vertex, pred, predEdge, succEdge, succ, compType, len, mode)) return false;
            }
            const Float relerr = length(vn.p - succ->position()) / std::max(std::max(std::abs(vn.p.x), std::abs(vn.p.y)), std::abs(vn.p.z));
            if (relerr > (Float)1e-3f) return false;
        }
        return true;
    }
    // dense inverse-with-determinant of the small system of SpecularManifold::det's mixed case (the reference calls Eigen's inverse() /
    // determinant() -- partial-pivot LU; restated as Gauss-Jordan with partial pivoting: the same matrix, another rounding)
    static bool denseInverse(std::vector<Float> &A, int n, std::vector<Float> &Ai)
    {
        Ai.assign((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) Ai[(size_t)i * n + i] = 1.0;
        for (int col = 0; col < n; ++col) {
            int piv = col;
            for (int r = col + 1; r < n; ++r) if (std::abs(A[(size_t)r * n + col]) > std::abs(A[(size_t)piv * n + col])) piv = r;
            if (A[(size_t)piv * n + col] == 0) return false;
            if (piv != col) for (int k = 0; k < n; ++k) { std::swap(A[(size_t)piv * n + k], A[(size_t)col * n + k]); std::swap(Ai[(size_t)piv * n + k], Ai[(size_t)col * n + k]); }
            const Float inv = 1.0 / A[(size_t)col * n + col];
            for (int k = 0; k < n; ++k) { A[(size_t)col * n + k] *= inv; Ai[(size_t)col * n + k] *= inv; }
            for (int r = 0; r < n; ++r) {
                if (r == col) continue;
                const Float f = A[(size_t)r * n + col];
                if (f == 0) continue;
                for (int k = 0; k < n; ++k) { A[(size_t)r * n + k] -= f * A[(size_t)col * n + k]; Ai[(size_t)r * n + k] -= f * Ai[(size_t)col * n + k]; }
            }
        }
        return true;
    }
    static Float denseDet(std::vector<Float> A, int n)
    {
        Float det = 1.0;
        for (int col = 0; col < n; ++col) {
            int piv = col;
            for (int r = col + 1; r < n; ++r) if (std::abs(A[(size_t)r * n + col]) > std::abs(A[(size_t)piv * n + col])) piv = r;
            if (A[(size_t)piv * n + col] == 0) return 0.0;
            if (piv != col) { for (int k = 0; k < n; ++k) std::swap(A[(size_t)piv * n + k], A[(size_t)col * n + k]); det = -det; }
            det *= A[(size_t)col * n + col];
            for (int r = col + 1; r < n; ++r) {
                const Float f = A[(size_t)r * n + col] / A[(size_t)col * n + col];
                if (f == 0) continue;
                for (int k = col; k < n; ++k) A[(size_t)r * n + k] -= f * A[(size_t)col * n + k];
            }
        }
        return det;
    }
    // Path::G, path.cpp:424-454: the plain geometry term of an edge, the generalized one of SpecularManifold::G across a specular chain
    Float pathG(const Path &p, int i, int j)
    {
        if (i >= j) return 1.0;
        if (j != i + 1) return manifoldG(p, i, j);
        const Float cosI = std::abs(dot(p.e[i]->d, p.v[i]->shadingNormal())), cosJ = std::abs(dot(p.e[i]->d, p.v[j]->shadingNormal()));
        const Float len = p.e[i]->length;
        return cosI * cosJ / (len * len);
    }
    // SpecularManifold::G (manifold.cpp:900-951) and multiG (:871-898)
    Float manifoldG(const Path &p, int a, int b)
    {
        if (std::abs(a - b) == 1) {
            if (a > b) std::swap(a, b);
            return edgeEvalCached(p.e[a], p.v[a], p.v[b], EGeometricTerm).x;
        }
        const int step = b > a ? 1 : -1;
        if (!manifoldInit(p, a, b)) { c.unsupported++; return 0.0; }
        SimpleVertex &last = mVerts[mVerts.size() - 1];
        const Vertex *vb = p.v[b];
        if (!vb->isOnSurface()) last.n = p.e[a < b ? (b - 1) : b]->d;
        else last.n = vb->shadingNormal();
        coordinateSystem(last.n, last.dpdu, last.dpdv);
        if (!manifoldTangents()) return 0.0;                                                 // "non-manifold configuration"
        const V3 d = mVerts[1].p - mVerts[0].p;
        const Float lengthSqr = lengthSquared(d), invLength = 1 / std::sqrt(lengthSqr);
        Float result = length(cross(mVerts[1].map(1, 0), mVerts[1].map(0, 1))) / lengthSqr;
        if (p.v[a]->isOnSurface()) result *= std::abs(dot(d, p.v[a]->shadingNormal())) * invLength;
        if (p.v[a + step]->isOnSurface()) result *= std::abs(dot(d, p.v[a + step]->shadingNormal())) * invLength;
        return result;
    }
    Float multiG(const Path &p, int a, int b)
    {
        if (a == 0) ++a; else if (a == p.length()) --a;
        if (b == 0) ++b; else if (b == p.length()) --b;
        const int step = b > a ? 1 : -1;
        while (!p.v[b]->isConnectable()) b -= step;
        while (!p.v[a]->isConnectable()) a += step;
        Float result = 1;
        for (int i = a + step, start = a; i != b + step; i += step)
            if (p.v[i]->isConnectable()) { result *= manifoldG(p, start, i); start = i; }
        return result;
    }
    // SpecularManifold::det, manifold.cpp:759-775: a chain with at most one glossy vertex between a and c needs no derivative
    Float manifoldDet(const Path &p, int a, int b, int cI)
    {
        const int k = p.length();
        if (a == 0 || a == k) std::swap(a, cI);
        const int step = b > a ? 1 : -1;
        int nGlossy = 0, nSpecular = 0;
        for (int i = a + step; i != cI; i += step) { if (p.v[i]->isConnectable()) ++nGlossy; else ++nSpecular; }
        if (nGlossy <= 1) return 1.0;
        if (!manifoldInit(p, a, cI)) { c.unsupported++; return 0.0; }
        const int b_idx = std::abs(b - a);
        SimpleVertex &vb = mVerts[b_idx];
        vb.n = p.v[b]->shadingNormal();
        coordinateSystem(vb.n, vb.dpdu, vb.dpdv);
        if (!manifoldTangents()) return 0.0;
        mVerts[b_idx].a.setZero(); mVerts[b_idx].b.setIdentity(); mVerts[b_idx].c.setZero();
        if (nSpecular == 0) {                                                                // glossy vertices only: the block tridiagonal determinant, manifold.cpp:800-822
            M2 Di, D = mVerts[1].b;
            Float det = D.det();
            for (size_t i = 2; i < mVerts.size() - 1; ++i) {
                if (!D.invert(Di)) return 0.0;
                D = mVerts[i].b - mVerts[i].a * Di * mVerts[i - 1].c;
                det *= D.det();
            }
            return std::abs(1 / det);
        }
        const int nv = nGlossy + nSpecular, N = 2 * nv;                                      // glossy and specular vertices, :823-867
        std::vector<Float> A((size_t)N * N, 0.0), Ai;
        for (int j = 0; j < nv; ++j) {
            const int i = j;
            auto put = [&](int cj, const M2 &mm) { A[(size_t)(2 * i) * N + 2 * cj] = mm.m[0][0]; A[(size_t)(2 * i) * N + 2 * cj + 1] = mm.m[0][1]; A[(size_t)(2 * i + 1) * N + 2 * cj] = mm.m[1][0]; A[(size_t)(2 * i + 1) * N + 2 * cj + 1] = mm.m[1][1]; };
            if (j - 1 >= 0) put(j - 1, mVerts[j + 1].a);
            put(j, mVerts[j + 1].b);
            if (j + 1 < nv) put(j + 1, mVerts[j + 1].c);
        }
        if (!denseInverse(A, N, Ai)) return 0.0;
        for (int i = 0; i < nv; ++i) {
            if (!mVerts[i + 1].degenerate) continue;
            for (int k = 0; k < N; ++k) { Ai[(size_t)(2 * i) * N + k] = 0; Ai[(size_t)(2 * i + 1) * N + k] = 0; Ai[(size_t)k * N + 2 * i] = 0; Ai[(size_t)k * N + 2 * i + 1] = 0; }
            Ai[(size_t)(2 * i) * N + 2 * i] = 1; Ai[(size_t)(2 * i + 1) * N + 2 * i + 1] = 1;
        }
        return std::abs(denseDet(Ai, N));
    }
    // Path::halfJacobian_GBDPT, path.cpp:380-394
    Float halfJacobian(const Path &p, int a, int b, int cI)
    {
        Float value = 1.0;
        value /= p.v[a]->pdf[ERadiance];
        value *= pathG(p, a - 1, a) / pathG(p, b, a);
        value *= manifoldDet(p, a, b, cI);
        return value;
    }
    // Path::calcSpecularPDFChange, path.cpp:403-421
    Float calcSpecularPDFChange(const Path &p, int cI, bool lightpath = false)
    {
        Float value(1.0);
        const int k = p.length() - 1;
        cI = std::max(1, cI);
        for (int i = cI + 1; i <= k; i++)
            if (p.v[lightpath ? i - 1 : i]->isConnectable()) value *= pathG(p, i - 1, i);
        if (value <= Float(0.0)) return 1.0;
        return multiG(p, cI, k) / value;
    }

    struct MuRec { int l = 0, m = 0; int extra[5] = {0, 0, 0, 0, 0}; };
    // ManifoldPerturbation::getSpecularChainEndGBDPT, mut_manifold.cpp:1230-1262
    int getSpecularChainEnd(const Path &path, int pos, int step) const
    {
        while (true) {
            if (pos < 0 || pos > path.length()) return -1;
            const Vertex *vertex = path.v[pos];
            if (!vertex->isSurface()) break;
            const Float roughness = getRoughness(vertex->mat);
            if (vertex->isConnectable() && roughness >= c.cfg.shiftThreshold) break;
            pos += step;
        }
        return pos;
    }
    // ManifoldPerturbation::computeMuRec, mut_manifold.cpp:1264-1296
    bool computeMuRec(const Path &source, MuRec &mu) const
    {
        const int k = source.length();
        if (!source.v[k - 1]->isConnectable()) return false;
        const int step = -1, a = k - 1;
        int b, cI;
        if ((b = getSpecularChainEnd(source, a + step, step)) == -1) return false;
        if ((cI = getSpecularChainEnd(source, b + step, step)) == -1) return false;
        mu.l = std::min(a, cI); mu.m = std::max(a, cI);
        mu.extra[0] = a; mu.extra[1] = b; mu.extra[2] = cI; mu.extra[3] = step; mu.extra[4] = ERadiance;
        return true;
    }
    // ManifoldPerturbation::perturbDirection, mut_manifold.cpp:938-986
    bool mutPerturbDirection(const Path &source, Path &proposal, int step, int a, Float offX, Float offY)
    {
        const Vertex *succ_old = source.v[a + step];
        (void)succ_old;
        const Edge *succEdge_old = source.e[a - 1];
        Vertex *pred = proposal.v[a - step], *vertex = proposal.v[a], *succ = proposal.v[a + step];
        Edge *predEdge = proposal.e[a - 1 - step], *succEdge = proposal.e[a - 1];
        const Float ppx = source.v[source.length() - 1]->prec.uvx + offX, ppy = source.v[source.length() - 1]->prec.uvy + offY;   // Path::getSamplePosition, path.h:540-542
        // sensor->sampleRay(ray, proposalSamplePosition, (0.5, 0.5), 0), perspective.cpp:249-269
        V3 dl = sampleToCameraDir(c, ppx * (1.0 / c.sc.cam.width), ppy * (1.0 / c.sc.cam.height));
        if (c.thinlens) {                                                                  // ThinLensCamera::sampleRay with the aperture's centre, thinlens.cpp:293-322: normalize(focusP - 0)
            const gpo_camera &cam = c.sc.cam;
            const V3 nearP((1 - 2 * ppx * (1.0 / cam.width)) * cam.nearClip * c.sc.tanHalf, (1 - 2 * ppy * (1.0 / cam.height)) / c.sc.aspect * cam.nearClip * c.sc.tanHalf, cam.nearClip);
            dl = normalize(nearP * (cam.focusDistance / nearP.z));
        }
        const V3 rd = camToWorld(c, dl);
        const V3 ro = c.camPos;
        // focusDistance = getFocusDistance() / absDot(worldTransform(0, 0, 1), ray.d); the default focus distance is the far clip (sensor.cpp:162)
        const Float focusDistance = (c.thinlens ? c.sc.cam.focusDistance : c.sc.cam.farClip) / std::abs(dot(c.camDir, rd));
        const V3 d = normalize((ro + rd * focusDistance) - source.v[a]->position());
        return perturbDirection(vertex, pred, predEdge, succEdge, succ, d, succEdge_old->length, ERadiance);
    }
    // ManifoldPerturbation::propagatePerturbation, mut_manifold.cpp:989-1149 (surface interactions; mode = ERadiance, step = -1): the vertices
    // between a and b follow the perturbed first segment deterministically -- a glossy one keeps its half vector, a specular one its component
    bool mutPropagatePerturbation(const Path &source, Path &proposal, int step, int a, int b, int mode)
    {
        for (int i = a + step; i != b; i += step) {
            const Vertex *pred_old = source.v[i - step], *vertex_old = source.v[i], *succ_old = source.v[i + step];
            const Edge *succEdge_old = source.e[mode == EImportance ? i : i - 1];
            Vertex *pred = proposal.v[i - step], *vertex = proposal.v[i], *succ = proposal.v[i + step];
            Edge *predEdge = proposal.e[mode == EImportance ? i - step : i - 1 - step], *succEdge = proposal.e[mode == EImportance ? i : i - 1];
            if (!vertex_old->isSurface()) return false;
            c.propagated++;
            const Intersection &its_old = vertex_old->its, &its_new = vertex->its;
            const V3 wi_old = its_old.sh.toLocal(normalize(pred_old->position() - its_old.p)), wo_old = its_old.sh.toLocal(normalize(succ_old->position() - its_old.p));
            const bool reflection = cosTheta(wi_old) * cosTheta(wo_old) > 0;
            const Float eta = getEta(vertex_old->mat);
            const V3 wi_world = normalize(pred->position() - vertex->position());
            V3 wo_world(0.0);
            if (materialOfPrim(c.sc, its_old.prim) != materialOfPrim(c.sc, its_new.prim)) return false;      // its_old.getBSDF() != its_new.getBSDF()
            if (vertex_old->isConnectable()) {
                V3 m(0.0);
                if (reflection) m = normalize(wi_old + wo_old);
                else if (eta != 1) m = normalize(wi_old.z < 0 ? (wi_old * eta + wo_old) : (wi_old + wo_old * eta));
                m = its_new.sh.toWorld(m.z > 0 ? m : -m);
                if (reflection) wo_world = reflectAbout(wi_world, m);
                else if (eta != 1) { wo_world = refractAbout(wi_world, m, eta); if (isZero(wo_world)) return false; }
                else wo_world = -wi_world;
                if (!perturbDirection(vertex, pred, predEdge, succEdge, succ, wo_world, succEdge_old->length, mode)) return false;
            } else {
                const int component = reflection ? EDeltaReflection : EDeltaTransmission;
                if (!propagatePerturbation(vertex, pred, predEdge, succEdge, succ, component, succEdge_old->length, mode)) return false;
            }
        }
        return true;
    }
    // ManifoldPerturbation::manifoldWalk, mut_manifold.cpp:1151-1227: the specular chain between b and c follows b's displacement by a Newton walk
    // on the specular manifold (pinned at c), checked for reversibility
    bool mutManifoldWalk(const Path &source, Path &proposal, int step, int b, int cI)
    {
        (void)step;
        const Vertex *vb_old = source.v[b], *vb_new = proposal.v[b];
        V3 n1 = vb_old->geometricNormal(), n2 = vb_new->geometricNormal();
        V3 rel = vb_new->position() - vb_old->position();
        Float len = length(rel);
        if (len == 0) return false;
        rel = rel / len;
        if (dot(n1, n2) < 0) n1 = -n1;
        V3 n = n1 + n2;
        n = n - rel * dot(rel, n);
        len = length(n);
        if (len == 0) return false;
        n = n / len;
        c.walks++;
        if (!manifoldInit(source, cI, b)) return false;
        const V3 p0 = mVerts[1].p;
        if (!manifoldMove(vb_new->position(), n)) return false;
        if (!manifoldUpdate(proposal, cI, b)) return false;
        if (!manifoldMove(vb_old->position(), n)) return false;
        const V3 p1 = mVerts[1].p;
        const Float radius = sceneBSphereRadius(c.sc);                                       // m_scene->getBSphere().radius: the box of Scene::initializeBidirectional
                                                                                             // (kd-tree bounds + sensor + emitters, scene.cpp:386-413; gbdpt_proc.cpp:80 calls it)
        const Float relerr = length(p0 - p1) / radius;
        if (relerr > 10.0 * Epsilon) return false;
        c.walksOk++;
        return true;
    }
    // ManifoldPerturbation::generateOffsetPathGBDPT, mut_manifold.cpp:806-936
    bool generateOffsetPath(const Path &source, Path &proposal, MuRec &mu, Float offX, Float offY, bool &couldConnectBehindB, bool lightPath)
    {
        const int k = source.length();
        if (!source.v[k - 1]->isConnectable()) return false;
        const int step = -1, a = k - 1;
        int b, cI;
        if ((b = getSpecularChainEnd(source, a + step, step)) == -1) return false;
        if ((cI = getSpecularChainEnd(source, b + step, step)) == -1) return false;
        const int l = std::min(a, cI), m = std::max(a, cI), q = std::min(b, b + step);
        mu = MuRec();
        mu.l = l; mu.m = m;
        mu.extra[0] = a; mu.extra[1] = b; mu.extra[2] = cI; mu.extra[3] = step; mu.extra[4] = ERadiance;
        proposal.clear();
        for (int i = 0; i < l + 1; ++i) { proposal.v.push_back(source.v[i]); if (i + 1 < l + 1) proposal.e.push_back(source.e[i]); }   // append(source, 0, l + 1)
        proposal.e.push_back(pool.allocEdge());
        for (int i = l + 1; i < m; ++i) { proposal.v.push_back(pool.allocVertex()); proposal.e.push_back(pool.allocEdge()); }
        for (int i = m; i < k + 1; ++i) { proposal.v.push_back(source.v[i]); if (i + 1 < k + 1) proposal.e.push_back(source.e[i]); }   // append(source, m, k + 1)
        proposal.v[a] = pool.clone(proposal.v[a]);
        proposal.v[cI] = pool.clone(proposal.v[cI]);
        if (!mutPerturbDirection(source, proposal, step, a, offX, offY)) return false;
        if (!mutPropagatePerturbation(source, proposal, step, a, b, ERadiance)) return false;
        if (!proposal.v[b]->isConnectable()) return false;
        if (std::abs(b - cI) > 1) {                                                          // :882-896
            const bool walkSuccess = mutManifoldWalk(source, proposal, step, b, cI);
            if (!walkSuccess && lightPath) return false;
            if (!walkSuccess) {              // "change state of proposal to as if no MW should have been done in the first place"
                for (int i = b + step; i != cI; i += step) proposal.v[i] = pool.clone(source.v[i]);
                mu.extra[2] = b + step;
            }
        }
        couldConnectBehindB = connect(proposal.vertexOrNull(q - 1), proposal.v[q], proposal.e[q], proposal.v[q + 1], proposal.vertexOrNull(q + 2),
                                      source.v[q]->isConnectable() ? EArea : EDiscrete, source.v[q + 1]->isConnectable() ? EArea : EDiscrete);
        if (lightPath && !couldConnectBehindB) return false;
        if (m >= k - 1) updateSamplePosition(proposal.v[k - 1], proposal.v[k - 2]);
        for (int i = 0; i <= proposal.length(); i++) {
            proposal.v[i]->rrWeight = source.v[i]->rrWeight;
            proposal.v[i]->sampledComponentIndex = source.v[i]->sampledComponentIndex;
            if (proposal.v[i]->type == ESurfaceInteraction && proposal.v[i]->componentType == 0) proposal.v[i]->componentType = source.v[i]->componentType;
        }
        return true;
    }

    // ---- MIS weights, path.cpp:49-378 ------------------------------------------------------------------------------------------------
    struct MisArrays { std::vector<Float> pdfImp, pdfRad; };
    void collectPdfs(const Path &emitterSubpath, const Edge *connectionEdge, const Path &sensorSubpath, int s, int t, std::vector<Float> &pdfImp, std::vector<Float> &pdfRad) const
    {
        const int k = s + t + 1, n = k + 1;
        const Vertex *vsPred = emitterSubpath.vertexOrNull(s - 1), *vtPred = sensorSubpath.vertexOrNull(t - 1), *vs = emitterSubpath.v[s], *vt = sensorSubpath.v[t];
        pdfImp.assign(n, 0.0); pdfRad.assign(n, 0.0);
        int pos = 0;
        pdfImp[pos++] = 1.0;
        for (int i = 0; i < s; ++i) pdfImp[pos++] = emitterSubpath.v[i]->pdf[EImportance] * emitterSubpath.e[i]->pdf[EImportance];
        pdfImp[pos++] = evalPdf(vs, vsPred, vt, EImportance, EArea) * connectionEdge->pdf[EImportance];
        if (t > 0) {
            pdfImp[pos++] = evalPdf(vt, vs, vtPred, EImportance, EArea) * sensorSubpath.e[t - 1]->pdf[EImportance];
            for (int i = t - 1; i > 0; --i) pdfImp[pos++] = sensorSubpath.v[i]->pdf[EImportance] * sensorSubpath.e[i - 1]->pdf[EImportance];
        }
        pos = 0;
        if (s > 0) {
            for (int i = 0; i < s - 1; ++i) pdfRad[pos++] = emitterSubpath.v[i + 1]->pdf[ERadiance] * emitterSubpath.e[i]->pdf[ERadiance];
            pdfRad[pos++] = evalPdf(vs, vt, vsPred, ERadiance, EArea) * emitterSubpath.e[s - 1]->pdf[ERadiance];
        }
        pdfRad[pos++] = evalPdf(vt, vtPred, vs, ERadiance, EArea) * connectionEdge->pdf[ERadiance];
        for (int i = t; i > 0; --i) pdfRad[pos++] = sensorSubpath.v[i - 1]->pdf[ERadiance] * sensorSubpath.e[i - 1]->pdf[ERadiance];
        pdfRad[pos++] = 1.0;
    }
    // the conversions of area densities next to a non-connectable vertex into projected solid angle, path.cpp:143-167,309-349
    static void stripGeometry(const Path &emitterSubpath, const Path &sensorSubpath, int s, int k, const std::vector<char> &connectableStrict,
                              std::vector<Float> &pdfImp, std::vector<Float> &pdfRad)
    {
        for (int i = 1; i <= k - 3; ++i) {
            if (i == s || !(connectableStrict[i] && !connectableStrict[i + 1])) continue;
            const Vertex *cur = i <= s ? emitterSubpath.v[i] : sensorSubpath.v[k - i];
            const Vertex *succ = i + 1 <= s ? emitterSubpath.v[i + 1] : sensorSubpath.v[k - i - 1];
            const Edge *edge = i < s ? emitterSubpath.e[i] : sensorSubpath.e[k - i - 1];
            pdfImp[i + 1] *= edge->length * edge->length / std::abs((succ->isOnSurface() ? dot(edge->d, succ->geometricNormal()) : 1) * (cur->isOnSurface() ? dot(edge->d, cur->geometricNormal()) : 1));
        }
        for (int i = k - 1; i >= 3; --i) {
            if (i - 1 == s || !(connectableStrict[i] && !connectableStrict[i - 1])) continue;
            const Vertex *cur = i <= s ? emitterSubpath.v[i] : sensorSubpath.v[k - i];
            const Vertex *succ = i - 1 <= s ? emitterSubpath.v[i - 1] : sensorSubpath.v[k - i + 1];
            const Edge *edge = i <= s ? emitterSubpath.e[i - 1] : sensorSubpath.e[k - i];
            pdfRad[i - 1] *= edge->length * edge->length / std::abs((succ->isOnSurface() ? dot(edge->d, succ->geometricNormal()) : 1) * (cur->isOnSurface() ? dot(edge->d, cur->geometricNormal()) : 1));
        }
    }
    void classify(const Path &emitterSubpath, const Path &sensorSubpath, int s, int t, std::vector<char> &connectable, std::vector<char> &connectableStrict, std::vector<char> &isNull) const
    {
        connectable.clear(); connectableStrict.clear(); isNull.clear();
        auto add = [&](const Vertex *v) {
            const bool cn = isConnectableGBDPT(v, c.cfg.shiftThreshold);
            connectable.push_back(cn); connectableStrict.push_back(v->isConnectable());
            isNull.push_back(false);                                                       // isNullInteraction(): no ENull components in the subset
        };
        for (int i = 0; i <= s; ++i) add(emitterSubpath.v[i]);
        for (int i = t; i >= 0; --i) add(sensorSubpath.v[i]);
    }
    // Path::miWeightBaseNoSweep_GBDPT, path.cpp:49-201
    Float miWeightBase(const Path &emitterSubpath, const Edge *connectionEdge, const Path &sensorSubpath, int s, int t, bool lightImage, Float exponent, Float geomTermX) const
    {
        const int k = s + t + 1;
        std::vector<char> connectable, connectableStrict, isNull;
        classify(emitterSubpath, sensorSubpath, s, t, connectable, connectableStrict, isNull);
        std::vector<Float> pdfImp, pdfRad;
        collectPdfs(emitterSubpath, connectionEdge, sensorSubpath, s, t, pdfImp, pdfRad);
        stripGeometry(emitterSubpath, sensorSubpath, s, k, connectableStrict, pdfImp, pdfRad);
        double sum_p = 0.0, p_st = 0.0;
        for (int p = 0; p < s + t + 1; ++p) {
            double p_i = 1.0;
            for (int i = 1; i < p + 1; ++i) p_i *= pdfImp[i];
            for (int i = p + 1; i < s + t + 1; ++i) p_i *= pdfRad[i];
            const int tPrime = k - p - 1;
            const bool allowedToConnect = (connectable[p] || isNull[p]) && connectable[p + 1];
            if (allowedToConnect && (lightImage || tPrime > 1)) sum_p += std::pow(p_i * geomTermX, exponent);
            if (tPrime == t) p_st = std::pow(p_i * geomTermX, exponent);
        }
        return (Float)(p_st / sum_p);
    }
    // Path::miWeightGradNoSweep_GBDPT, path.cpp:204-378
    Float miWeightGrad(const Path &emitterSubpath, const Edge *connectionEdge, const Path &sensorSubpath,
                       const Path &offsetEmitterSubpath, const Edge *offsetConnectionEdge, const Path &offsetSensorSubpath,
                       int s, int t, bool lightImage, Float jDet, Float exponent, Float geomTermX, Float geomTermY) const
    {
        const int k = s + t + 1;
        std::vector<char> connectable, connectableStrict, isNull;
        classify(emitterSubpath, sensorSubpath, s, t, connectable, connectableStrict, isNull);
        std::vector<Float> pdfImp, pdfRad, offsetPdfImp, offsetPdfRad;
        collectPdfs(emitterSubpath, connectionEdge, sensorSubpath, s, t, pdfImp, pdfRad);
        collectPdfs(offsetEmitterSubpath, offsetConnectionEdge, offsetSensorSubpath, s, t, offsetPdfImp, offsetPdfRad);
        stripGeometry(emitterSubpath, sensorSubpath, s, k, connectableStrict, pdfImp, pdfRad);
        stripGeometry(offsetEmitterSubpath, offsetSensorSubpath, s, k, connectableStrict, offsetPdfImp, offsetPdfRad);   // (the base path's flags select the rows, :310,331)
        double sum_p_i = 0.0, p_st = 0.0;
        for (int p = 0; p < s + t + 1; ++p) {
            double value = 1.0, oValue = 1.0;
            for (int i = 1; i < p + 1; ++i) { value *= pdfImp[i]; oValue *= offsetPdfImp[i]; }
            for (int i = p + 1; i < s + t + 1; ++i) { value *= pdfRad[i]; oValue *= offsetPdfRad[i]; }
            const int tPrime = k - p - 1;
            const bool allowedToConnect = (connectable[p] || isNull[p]) && connectable[p + 1];
            if (allowedToConnect && (lightImage || tPrime > 1)) sum_p_i += std::pow(value * geomTermX, exponent) + std::pow(oValue * jDet * geomTermY, exponent);
            if (tPrime == t) p_st = std::pow(value * geomTermX, exponent);
        }
        return (Float)(p_st / sum_p_i);
    }

    // ---- GBDPTRenderer ---------------------------------------------------------------------------------------------------------------
    struct ShiftPathData {                                                                 // gbdpt_proc.cpp:29-42
        std::vector<double> jacobianDet, genGeomTerm;
        MuRec muRec;
        bool couldConnectAfterB = false, success = false;
        explicit ShiftPathData(int n) : jacobianDet(n + 3, 1.0), genGeomTerm(n + 3, 1.0) {}
    };
    // GBDPTRenderer::createShiftablePath, gbdpt_proc.cpp:600-662
    bool createShiftablePath(Path &connectedPath, Path &emitterSubpath, Path &sensorSubpath, int s, int t, int &memPointer)
    {
        connectedPath.clear();
        while (!isConnectableGBDPT(sensorSubpath.v[t], c.cfg.shiftThreshold)) { t--; sensorSubpath.v.pop_back(); sensorSubpath.e.pop_back(); }   // removeAndReleaseLastElement
        if (sensorSubpath.v[t]->type == ESurfaceInteraction && emitterOfPrim(c.sc, sensorSubpath.v[t]->its.prim) >= 0) s = 0;
        for (memPointer = 0; memPointer < s; memPointer++) { connectedPath.v.push_back(emitterSubpath.v[memPointer]); connectedPath.e.push_back(emitterSubpath.e[memPointer]); }
        connectedPath.v.push_back(pool.clone(emitterSubpath.v[memPointer]));
        connectedPath.e.push_back(pool.allocEdge());
        connectedPath.v.push_back(pool.clone(sensorSubpath.v[t]));
        cast(connectedPath.v[memPointer + 1], EEmitterSample);
        for (int i = t - 1; i >= 0; i--) { connectedPath.v.push_back(sensorSubpath.v[i]); connectedPath.e.push_back(sensorSubpath.e[i]); }
        const bool pathSuccess = connect(connectedPath.vertexOrNull(memPointer - 1), connectedPath.v[memPointer], connectedPath.e[memPointer], connectedPath.v[memPointer + 1],
                                         connectedPath.vertexOrNull(memPointer + 2),
                                         connectedPath.v[memPointer]->isConnectable() ? EArea : EDiscrete, connectedPath.v[memPointer + 1]->isConnectable() ? EArea : EDiscrete);
        if (t == 1) updateSamplePosition(connectedPath.v[connectedPath.vertexCount() - 2], connectedPath.v[connectedPath.vertexCount() - 3]);
        return pathSuccess;
    }
    // GBDPTRenderer::createShiftedLightPath, gbdpt_proc.cpp:568-590
    void createShiftedLightPath(Path &base, Path &offset, double &jacobian, bool &pathSuccess, V3 &offsetWeight, Float &offsetPdf, MuRec &mu, Float shX, Float shY, int s)
    {
        jacobian = 1.0;
        bool couldConnectWithB = false;
        pathSuccess = generateOffsetPath(base, offset, mu, shX, shY, couldConnectWithB, true);
        if (pathSuccess) {
            jacobian = halfJacobian(offset, mu.extra[0], mu.extra[1], mu.extra[2]) / halfJacobian(base, mu.extra[0], mu.extra[1], mu.extra[2]);
            offsetPdf = 1.0;
            offsetWeight = V3(1.0);
            for (int i = 1; i <= s; ++i) {
                offsetWeight = offsetWeight * offset.v[i - 1]->weight[EImportance] * offset.v[i - 1]->rrWeight * offset.e[i - 1]->weight[EImportance];
                offsetPdf = offsetPdf * offset.v[i - 1]->pdf[EImportance] * offset.v[i - 1]->rrWeight * offset.e[i - 1]->pdf[EImportance];
            }
        }
    }

    struct Splat { Float x, y; int buffer; V3 value; };
    struct SampleResult { V3 primal; V3 gradient[4]; Float posX = 0, posY = 0; std::vector<Splat> light; };

    // GBDPTRenderer::evaluate, gbdpt_proc.cpp:259-534
    void evaluate(SampleResult &wr, Path &emitterSubpath, std::vector<Path> &sensorSubpath, std::vector<ShiftPathData> &pathData, int vert_b)
    {
        static const Float shifts[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};                // :265
        const int neighbourCount = 4;
        const Config &cfg = c.cfg;
        const Float initialX = sensorSubpath[0].v[1]->prec.uvx, initialY = sensorSubpath[0].v[1]->prec.uvy;
        wr.posX = initialX; wr.posY = initialY;
        const int nE = emitterSubpath.vertexCount(), nS = sensorSubpath[0].vertexCount();
        std::vector<V3> importanceWeights(nE);
        std::vector<Float> importancePdf(nE);
        std::vector<std::vector<V3>> radianceWeights(neighbourCount + 1, std::vector<V3>(nS));
        std::vector<std::vector<Float>> radiancePdf(neighbourCount + 1, std::vector<Float>(nS, 0.0));
        // combineImportanceData / combineRadianceData, :544-565
        importanceWeights[0] = V3(1.0); importancePdf[0] = 1.0;
        for (int i = 1; i < nE; ++i) {
            importanceWeights[i] = importanceWeights[i - 1] * emitterSubpath.v[i - 1]->weight[EImportance] * emitterSubpath.v[i - 1]->rrWeight * emitterSubpath.e[i - 1]->weight[EImportance];
            importancePdf[i] = importancePdf[i - 1] * emitterSubpath.v[i - 1]->pdf[EImportance] * emitterSubpath.v[i - 1]->rrWeight * emitterSubpath.e[i - 1]->pdf[EImportance];
        }
        for (int k = 0; k <= neighbourCount; k++) {
            radianceWeights[k][0] = V3(1.0); radiancePdf[k][0] = 1.0;
            for (int i = 1; i < nS; ++i)
                if (pathData[k].success && i < sensorSubpath[k].vertexCount()) {
                    radianceWeights[k][i] = radianceWeights[k][i - 1] * sensorSubpath[k].v[i - 1]->weight[ERadiance] * sensorSubpath[k].v[i - 1]->rrWeight * sensorSubpath[k].e[i - 1]->weight[ERadiance];
                    radiancePdf[k][i] = radiancePdf[k][i - 1] * sensorSubpath[k].v[i - 1]->pdf[ERadiance] * sensorSubpath[k].v[i - 1]->rrWeight * sensorSubpath[k].e[i - 1]->pdf[ERadiance];
                }
        }
        V3 primal(0.0), gradient[4];
        Path offsetEmitterSubpath, connectedBasePath;
        V3 geomTermBase, connectionPartsBase, offsetImportanceWeight;
        Edge connectionEdge, connectionEdgeBase;
        bool successConnectBase = false;
        Float offsetImportancePdf = 0;
        std::vector<V3> value(neighbourCount + 1);
        std::vector<Float> miWeight(neighbourCount + 1), valuePdf(neighbourCount + 1);
        std::vector<double> jacobianLP(neighbourCount), genGeomTermLP(neighbourCount + 1);
        bool pathSuccess[5];

        for (int s = nE - 1; s >= 0; --s) {
            const int minT = std::max(2 - s, cfg.lightImage ? 1 : 2);
            int maxT = nS - 1;
            if (cfg.maxDepth != -1) maxT = std::min(maxT, cfg.maxDepth + 1 - s);
            for (int t = maxT; t >= minT; --t) {
                Float samplePosX = initialX, samplePosY = initialY;
                if (t == 1) {
                    if ((sensorSubpath[0].v[t]->type == ESensorSample && !getSamplePosition(sensorSubpath[0].v[t], emitterSubpath.v[s], samplePosX, samplePosY))
                        || !isConnectableGBDPT(emitterSubpath.v[s], cfg.shiftThreshold))
                        continue;
                }
                int memPointer = 0;
                MuRec muRec;
                for (int k = 0; k <= neighbourCount; k++) {
                    miWeight[k] = 1.0 / (s + t + 1);
                    pathSuccess[k] = pathData[k].success;
                    value[k] = V3(0.0);
                    valuePdf[k] = 0.0;
                    const V3 *importanceWeightTmp = &importanceWeights[s], *radianceWeightTmp = &radianceWeights[t == 1 ? 0 : k][t];
                    const Float *importancePdfTmp = &importancePdf[s], *radiancePdfTmp = &radiancePdf[t == 1 ? 0 : k][t];
                    const Path *sensorSubpathTmp = &sensorSubpath[k], *emitterSubpathTmp = &emitterSubpath;
                    if (t == 1 && k == 0) {
                        pathSuccess[0] = createShiftablePath(connectedBasePath, emitterSubpath, sensorSubpath[0], s, 1, memPointer);
                        computeMuRec(connectedBasePath, muRec);
                        genGeomTermLP[0] = calcSpecularPDFChange(connectedBasePath, muRec.extra[2], true);
                    }
                    if (t == 1 && k > 0 && !isZero(value[0])) {
                        if (!pathSuccess[0]) pathSuccess[k] = false;
                        else {
                            createShiftedLightPath(connectedBasePath, offsetEmitterSubpath, jacobianLP[k - 1], pathSuccess[k], offsetImportanceWeight, offsetImportancePdf, muRec,
                                                   shifts[k - 1][0], shifts[k - 1][1], s);
                            if (pathSuccess[k]) {
                                genGeomTermLP[k] = calcSpecularPDFChange(offsetEmitterSubpath, muRec.extra[2], true);
                                importanceWeightTmp = &offsetImportanceWeight;
                                importancePdfTmp = &offsetImportancePdf;
                                emitterSubpathTmp = &offsetEmitterSubpath;
                            }
                            sensorSubpathTmp = &sensorSubpath[0];
                        }
                    }
                    V3 geomTerm(0.0);
                    do {
                        if (pathSuccess[k] && pathSuccess[0] && (k == 0 || (valuePdf[0] > 0 && !isZero(value[0])))) {
                            if (!pathData[k].couldConnectAfterB && t > vert_b) break;
                            Vertex *vsPred = emitterSubpathTmp->vertexOrNull(s - 1), *vtPred = sensorSubpathTmp->vertexOrNull(t - 1);
                            Vertex *vs = emitterSubpathTmp->v[s], *vt = sensorSubpathTmp->v[t];
                            const int remaining = cfg.maxDepth - s - t + 1;
                            if (vs->type == EEmitterSupernode) {
                                if (!cast(vt, EEmitterSample) || vt->degenerate) { valuePdf[k] = *radiancePdfTmp; break; }
                                const V3 connectionParts = (k > 0 && t > vert_b + 1) ? connectionPartsBase : eval(vs, vsPred, vt, EImportance) * eval(vt, vtPred, vs, ERadiance);
                                if (k == 0) connectionPartsBase = connectionParts;
                                value[k] = *radianceWeightTmp * connectionParts;
                                valuePdf[k] = *radiancePdfTmp;
                            } else if (vt->type == ESensorSupernode) {                      // (t >= 1 here: not reached)
                                valuePdf[k] = *importancePdfTmp;
                                break;
                            } else {
                                if (!isConnectableGBDPT(vs, cfg.shiftThreshold) || !isConnectableGBDPT(vt, cfg.shiftThreshold) || vs->type == 0 || vt->type == 0) {
                                    valuePdf[k] = *importancePdfTmp * *radiancePdfTmp;
                                    break;
                                }
                                const V3 connectionParts = (k > 0 && t > vert_b + 1) ? connectionPartsBase : eval(vs, vsPred, vt, EImportance) * eval(vt, vtPred, vs, ERadiance);
                                if (k == 0) connectionPartsBase = connectionParts;
                                value[k] = *importanceWeightTmp * *radianceWeightTmp * connectionParts;
                                valuePdf[k] = *importancePdfTmp * *radiancePdfTmp;
                                vs->measure = vt->measure = EArea;
                            }
                            if (isZero(value[k]) || valuePdf[k] == 0) break;
                            int interactions = remaining;
                            const bool successConnect = (k > 0 && t > vert_b) ? successConnectBase : edgePathConnectAndCollapse(&connectionEdge, vs, vt, interactions);
                            if (k == 0) successConnectBase = successConnect;
                            if (!successConnect) { value[k] = V3(0.0); break; }
                            geomTerm = (k > 0 && t > vert_b) ? geomTermBase : edgeEvalCached(&connectionEdge, vs, vt, EGeneralizedGeometricTerm);
                            value[k] = value[k] * geomTerm;
                            valuePdf[k] *= (t < 2 ? genGeomTermLP[k] : pathData[k].genGeomTerm[t]);
                            if (isZero(value[k]) || valuePdf[k] == 0) break;
                            if (k == 0) {
                                connectionEdgeBase = connectionEdge;
                                geomTermBase = geomTerm;
                                miWeight[0] = miWeightBase(emitterSubpath, &connectionEdgeBase, sensorSubpath[0], s, t, cfg.lightImage != 0, 2.0,
                                                           (t < 2 ? genGeomTermLP[0] : pathData[0].genGeomTerm[t])) / valuePdf[0];
                            } else {
                                miWeight[k] = miWeightGrad(emitterSubpath, &connectionEdgeBase, sensorSubpath[0], *emitterSubpathTmp, &connectionEdge, *sensorSubpathTmp, s, t,
                                                           cfg.lightImage != 0, (t < 2 ? jacobianLP[k - 1] : pathData[k].jacobianDet[t]), 1.0,
                                                           (t < 2 ? genGeomTermLP[0] : pathData[0].genGeomTerm[t]), (t < 2 ? genGeomTermLP[k] : pathData[k].genGeomTerm[t])) / valuePdf[0];
                            }
                        }
                    } while (false);
                    if (g_traceMain && t < 2 && k > 0) std::fprintf(stderr, "   light path: jacobian %.17g genGeomTerm %.17g (base %.17g)\n", jacobianLP[k - 1], genGeomTermLP[k], genGeomTermLP[0]);
                    if (g_traceMain) std::fprintf(stderr, "st %d %d k %d ok %d value %.17g %.17g %.17g pdf %.17g miW %.17g geom %.17g rays %llu %llu\n", s, t, k, (int)pathSuccess[k], value[k].x, value[k].y, value[k].z, valuePdf[k], miWeight[k], geomTerm.x,
                                                  (unsigned long long)c.sc.raysTraced, (unsigned long long)c.sc.shadowRaysTraced);
                    if (isZero(value[k]) || isZero(value[0])) {
                        value[k] = V3(0.0);
                        miWeight[k] = miWeight[0];
                        valuePdf[k] = valuePdf[0];
                    }
                }
                if (isZero(value[0])) continue;
                const V3 mainRad = value[0] * (valuePdf[0] * miWeight[0]);                   // Spectrum mainRad = valuePdf[0] * miWeight[0] * value[0]: (Float * Float) * Spectrum
                if (t >= 2) primal = primal + mainRad;
                else wr.light.push_back({samplePosX, samplePosY, 0, mainRad});
                const V3 fx = value[0] * valuePdf[0];
                for (int n = 0; n < neighbourCount; n++) {
                    const V3 fy = value[n + 1] * valuePdf[n + 1] * (t < 2 ? jacobianLP[n] : pathData[n + 1].jacobianDet[t]);
                    const V3 gradVal = (fy - fx) * (Float(2.0) * miWeight[n + 1]);           // Float(2.f) * miWeight * (fy - fx)
                    if (t >= 2) gradient[n] = gradient[n] + gradVal;
                    else wr.light.push_back({samplePosX, samplePosY, n + 1, gradVal});
                }
            }
        }
        wr.primal = primal;
        for (int k = 0; k < neighbourCount; ++k) wr.gradient[k] = gradient[k];
    }

    // the body of GBDPTRenderer::process's sample loop, gbdpt_proc.cpp:152-252
    // Known-answer probe of the manifold (tests/test_gbdpt_oracle.py): the sensor subpath of one sample; for its first chain "connectable
    // vertex i, ONE non-connectable vertex, connectable vertex i + 2": out = found, the three positions, the end's shading normal, the
    // generalized geometry term G(i, i + 2), the plain terms of its two edges, then -- pinned at i, end moved by `delta` within the end's tangent
    // plane -- the walk's iteration count, its success, and the positions the chain vertex and the end arrive at.
    void manifoldProbe(int px, int py, const Float delta[3], Float out[32])
    {
        for (int k = 0; k < 32; ++k) out[k] = 0.0;
        Config &cfg = c.cfg;
        if (c.sc.cam.shutterClose > c.sc.cam.shutterOpen) (void)rng.next1D();
        if (cfg.maxDepth == -1) cfg.maxDepth = 12;
        Path emitterSubpath, sensorSubpath;
        emitterSubpath.v.push_back(pool.allocVertex());
        emitterSubpath.v[0]->type = EEmitterSupernode; emitterSubpath.v[0]->degenerate = false;
        sensorSubpath.v.push_back(pool.allocVertex());
        sensorSubpath.v[0]->type = ESensorSupernode; sensorSubpath.v[0]->degenerate = true;
        alternatingRandomWalkFromPixel(emitterSubpath, 0, sensorSubpath, cfg.maxDepth + 1, px, py, -1);
        const Path &p = sensorSubpath;
        for (int i = 1; i + 2 < p.vertexCount(); ++i) {
            if (!p.v[i]->isConnectable() || p.v[i + 1]->isConnectable() || !p.v[i + 2]->isConnectable() || !p.v[i + 1]->isSurface() || !p.v[i + 2]->isSurface()) continue;
            out[0] = 1.0;
            const V3 pa = p.v[i]->position(), pm = p.v[i + 1]->position(), pb = p.v[i + 2]->position(), nb = p.v[i + 2]->shadingNormal(), na = p.v[i]->shadingNormal(), nm = p.v[i + 1]->shadingNormal();
            out[1] = pa.x; out[2] = pa.y; out[3] = pa.z; out[4] = pm.x; out[5] = pm.y; out[6] = pm.z; out[7] = pb.x; out[8] = pb.y; out[9] = pb.z;
            out[10] = nb.x; out[11] = nb.y; out[12] = nb.z;
            out[13] = manifoldG(p, i, i + 2);
            out[14] = pathG(p, i, i + 1); out[15] = pathG(p, i + 1, i + 2);
            out[26] = na.x; out[27] = na.y; out[28] = na.z; out[29] = nm.x; out[30] = nm.y; out[31] = nm.z;
            if (!manifoldInit(p, i, i + 2)) return;
            const V3 target = pb + V3(delta[0], delta[1], delta[2]);
            const bool ok = manifoldMove(target, nb);
            out[16] = mIterations; out[17] = ok ? 1.0 : 0.0;
            out[18] = mVerts[1].p.x; out[19] = mVerts[1].p.y; out[20] = mVerts[1].p.z;
            out[21] = mVerts[2].p.x; out[22] = mVerts[2].p.y; out[23] = mVerts[2].p.z;
            out[24] = (Float)(p.v[i + 1]->mat.type); out[25] = (Float)i;
            return;
        }
    }

    // The same for a chain with TWO non-connectable vertices: "connectable vertex i, specular i + 1, specular i + 2, connectable i + 3" of the sensor
    // subpath (two facing mirrors; both faces of a glass slab).  out: [0] found, [1..12] the four positions, [13..24] their shading normals,
    // [25] SpecularManifold::G(i, i + 3), [26] multiG(i, i + 3) (-1 when i + 3 is the path's last vertex), [27] [28] the material types of the chain vertices, [29] eta of the first;
    // then -- pinned at i, the end moved by `delta` -- [30] iterations, [31] success, [32..40] where the chain vertices and the end arrive.
    void manifoldProbe2(int px, int py, const Float delta[3], Float out[48])
    {
        for (int k = 0; k < 48; ++k) out[k] = 0.0;
        Config &cfg = c.cfg;
        if (c.sc.cam.shutterClose > c.sc.cam.shutterOpen) (void)rng.next1D();
        if (cfg.maxDepth == -1) cfg.maxDepth = 12;
        Path emitterSubpath, sensorSubpath;
        emitterSubpath.v.push_back(pool.allocVertex());
        emitterSubpath.v[0]->type = EEmitterSupernode; emitterSubpath.v[0]->degenerate = false;
        sensorSubpath.v.push_back(pool.allocVertex());
        sensorSubpath.v[0]->type = ESensorSupernode; sensorSubpath.v[0]->degenerate = true;
        alternatingRandomWalkFromPixel(emitterSubpath, 0, sensorSubpath, cfg.maxDepth + 1, px, py, -1);
        const Path &p = sensorSubpath;
        for (int i = 1; i + 3 < p.vertexCount(); ++i) {
            if (!p.v[i]->isConnectable() || p.v[i + 1]->isConnectable() || p.v[i + 2]->isConnectable() || !p.v[i + 3]->isConnectable()) continue;
            if (!p.v[i]->isSurface() || !p.v[i + 1]->isSurface() || !p.v[i + 2]->isSurface() || !p.v[i + 3]->isSurface()) continue;
            out[0] = 1.0;
            for (int q = 0; q < 4; ++q) {
                const V3 pp = p.v[i + q]->position(), nn = p.v[i + q]->shadingNormal();
                out[1 + 3 * q] = pp.x; out[2 + 3 * q] = pp.y; out[3 + 3 * q] = pp.z;
                out[13 + 3 * q] = nn.x; out[14 + 3 * q] = nn.y; out[15 + 3 * q] = nn.z;
            }
            out[25] = manifoldG(p, i, i + 3);
            out[26] = i + 3 < p.length() ? multiG(p, i, i + 3) : -1.0;                         // (multiG steps back from the path's last vertex: it takes it for a supernode, path.cpp:423-454)
            out[27] = (Float)(p.v[i + 1]->mat.type); out[28] = (Float)(p.v[i + 2]->mat.type); out[29] = p.v[i + 1]->mat.eta[0];
            if (!manifoldInit(p, i, i + 3)) return;
            const V3 target = p.v[i + 3]->position() + V3(delta[0], delta[1], delta[2]);
            const bool ok = manifoldMove(target, p.v[i + 3]->shadingNormal());
            out[30] = mIterations; out[31] = ok ? 1.0 : 0.0;
            for (int q = 0; q < 3; ++q) { out[32 + 3 * q] = mVerts[1 + q].p.x; out[33 + 3 * q] = mVerts[1 + q].p.y; out[34 + 3 * q] = mVerts[1 + q].p.z; }
            return;
        }
    }

    void processSample(int px, int py, SampleResult &wr)
    {
        static const Float shifts[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};                // :101
        Config &cfg = c.cfg;
        if (c.sc.cam.shutterClose > c.sc.cam.shutterOpen) (void)rng.next1D();               // :156-157 (needsTimeSample; the subpaths' `time` moves nothing: static transforms)
        if (cfg.maxDepth == -1) cfg.maxDepth = 12;                                         // :103-106
        int emitterDepth = cfg.maxDepth, sensorDepth = cfg.maxDepth;
        // the perspective sensor is degenerate (EDeltaPosition): no extra emitter step; area emitters can be hit: one more sensor step (:116-122)
        if (c.thinlens && emitterDepth != -1) ++emitterDepth;                              // "go one extra step if the sensor can be intersected": not EDeltaPosition, :117-118
        bool degenerateEmitters = true;                                                    // Scene::hasDegenerateEmitters, scene.cpp:388,410-411: every emitter is EDeltaPosition
        for (const Emitter &em : c.sc.emitters) if (em.numTris >= 0) degenerateEmitters = false;
        if (!degenerateEmitters && sensorDepth != -1) ++sensorDepth;
        const int neighborCount = 4;
        std::vector<ShiftPathData> pathData(neighborCount + 1, ShiftPathData(sensorDepth + 3));
        pathData[0].success = true;
        pathData[0].couldConnectAfterB = true;
        Path emitterSubpath;
        std::vector<Path> sensorSubpath(neighborCount + 1);
        emitterSubpath.v.push_back(pool.allocVertex());                                    // Path::initialize -> makeEndpoint, vertex.cpp:27-33
        emitterSubpath.v[0]->type = EEmitterSupernode; emitterSubpath.v[0]->degenerate = false;
        sensorSubpath[0].v.push_back(pool.allocVertex());
        sensorSubpath[0].v[0]->type = ESensorSupernode; sensorSubpath[0].v[0]->degenerate = true;
        alternatingRandomWalkFromPixel(emitterSubpath, emitterDepth, sensorSubpath[0], sensorDepth, px, py, cfg.rrDepth);
        Path connectPath;
        int ptx = 0;
        createShiftablePath(connectPath, emitterSubpath, sensorSubpath[0], 1, sensorSubpath[0].vertexCount() - 1, ptx);
        computeMuRec(connectPath, pathData[0].muRec);
        for (int v = pathData[0].muRec.extra[0] - 1; v >= 0; v--) {
            const int idx = connectPath.vertexCount() - 1 - v;
            if (isConnectableGBDPT(connectPath.v[v], cfg.shiftThreshold) && v >= pathData[0].muRec.extra[2])
                pathData[0].genGeomTerm.at(idx) = calcSpecularPDFChange(connectPath, v);
            else
                pathData[0].genGeomTerm.at(idx) = pathData[0].genGeomTerm.at(idx - 1);
        }
        for (int k = 0; k < neighborCount; k++) {
            pathData[k + 1].success = pathData[0].muRec.extra[0] <= 2 ? false
                : generateOffsetPath(connectPath, sensorSubpath[k + 1], pathData[k + 1].muRec, shifts[k][0], shifts[k][1], pathData[k + 1].couldConnectAfterB, false);
            if (pathData[k + 1].success) {
                for (int v = pathData[k + 1].muRec.extra[0] - 1; v >= 0; v--) {
                    const int idx = connectPath.vertexCount() - 1 - v;
                    if (isConnectableGBDPT(connectPath.v[v], cfg.shiftThreshold) && v >= pathData[k + 1].muRec.extra[2]) {
                        const int a = pathData[k + 1].muRec.extra[0];
                        const int b = v >= pathData[k + 1].muRec.extra[1] ? v : pathData[k + 1].muRec.extra[1];
                        const int cI = v >= pathData[k + 1].muRec.extra[1] ? v - 1 : pathData[k + 1].muRec.extra[2];
                        const double jx = halfJacobian(connectPath, a, b, cI), jy = halfJacobian(sensorSubpath[k + 1], a, b, cI);
                        pathData[k + 1].jacobianDet.at(idx) = jy / jx;
                        pathData[k + 1].genGeomTerm.at(idx) = calcSpecularPDFChange(sensorSubpath[k + 1], v);
                    } else {
                        pathData[k + 1].jacobianDet.at(idx) = pathData[k + 1].jacobianDet.at(idx - 1);
                        pathData[k + 1].genGeomTerm.at(idx) = pathData[k + 1].genGeomTerm.at(idx - 1);
                    }
                }
            }
            sensorSubpath[k + 1].reverse();
        }
        const int v_b = connectPath.vertexCount() - 1 - pathData[0].muRec.extra[1];
        evaluate(wr, emitterSubpath, sensorSubpath, pathData, v_b);
    }
};

// GBDPTWorkResult (camera blocks: spectrum, alpha, weight; light images: spectrum) + GBDPTProcess::develop, for the box filter
struct Film {
    int W, H;
    std::vector<double> block[5];    // [H][W][4]: R, G, B, weight
    std::vector<double> light[5];    // [H][W][3]
    unsigned long long invalidPuts = 0;
    double filterRadius, filterScale, filterValues[32];
    Film(int w, int h) : W(w), H(h)
    {
        for (auto &b : block) b.assign((size_t)w * h * 4, 0.0);
        for (auto &b : light) b.assign((size_t)w * h * 3, 0.0);
        filterRadius = 0.5 + (double)1e-5f;                                                 // box.cpp:38
        double sum = 0;                                                                     // rfilter.cpp:37-55
        for (int i = 0; i < 31; ++i) { filterValues[i] = 1.0; sum += 1.0; }
        filterValues[31] = 0.0;
        filterScale = 31 / filterRadius;
        sum *= 2 * filterRadius / 31;
        for (int i = 0; i < 31; ++i) filterValues[i] *= 1.0 / sum;
    }
    double evalDiscretized(double x) const { return filterValues[std::min((int)std::abs(x * filterScale), 31)]; }
    // ImageBlock::put (imageblock.h:150-210) with negative values allowed (gbdpt_proc.cpp:175-179); channels = nch values + (alpha, weight) for the blocks
    void put(std::vector<double> &buf, int stride, double px, double py, V3 spec, bool withWeight)
    {
        if (!std::isfinite(spec.x) || !std::isfinite(spec.y) || !std::isfinite(spec.z)) { invalidPuts++; return; }
        const double posx = px - 0.5, posy = py - 0.5;
        const int x0 = std::max((int)std::ceil(posx - filterRadius), 0), y0 = std::max((int)std::ceil(posy - filterRadius), 0);
        const int x1 = std::min((int)std::floor(posx + filterRadius), W - 1), y1 = std::min((int)std::floor(posy + filterRadius), H - 1);
        for (int y = y0; y <= y1; ++y) {
            const double wy = evalDiscretized(y - posy);
            for (int x = x0; x <= x1; ++x) {
                const double w = evalDiscretized(x - posx) * wy;
                double *dest = &buf[((size_t)y * W + x) * stride];
                dest[0] += w * spec.x; dest[1] += w * spec.y; dest[2] += w * spec.z;
                if (withWeight) dest[3] += w * 1.0;
            }
        }
    }
    void add(const Tracer::SampleResult &r)
    {
        put(block[0], 4, r.posX, r.posY, r.primal, true);                                   // putSample, gbdpt_proc.cpp:531-533
        for (int k = 0; k < 4; ++k) put(block[k + 1], 4, r.posX, r.posY, r.gradient[k], true);
        for (const auto &s : r.light) put(light[s.buffer], 3, s.x, s.y, s.value, false);    // putLightSample, :514,525
    }
};

} // namespace gb
