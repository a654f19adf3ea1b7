// gbdpt_general.hip.h -- the GENERAL form of the G-BDPT sample (round 4: stage C of SURVEY.md 8f-1): paths with SPECULAR CHAINS.
//
// gbdpt_kernels.hip.h carries the samples whose surface vertices are all connectable (Path::isConnectable_GBDPT): an offset path is three
// records, every generalized geometry term is 1, the MIS sums are recurrences.  A sample that meets a perfectly specular BSDF (conductor,
// dielectric) or a rough conductor below shiftThreshold needs what that form leaves out:
//   * ManifoldPerturbation::propagatePerturbation (mut_manifold.cpp:989-1149): the chain between the sensor and the first connectable vertex b
//     is re-created vertex by vertex (PathVertex::propagatePerturbation / perturbDirection, vertex.cpp:488-790);
//   * ManifoldPerturbation::manifoldWalk (:1151-1227) over SpecularManifold::{init, computeTangents, project, move, update}
//     (manifold.cpp:59-757): the chain between b and the next connectable vertex c follows b by a Newton walk on the specular manifold;
//   * SpecularManifold::{G, multiG, det} (manifold.cpp:759-951) in Path::{G, halfJacobian_GBDPT, calcSpecularPDFChange} (path.cpp:380-454), and the
//     conversion of area densities next to non-connectable vertices in the MIS weights (path.cpp:143-167,309-349).
// Offset paths then replace a variable number of vertices, so this form keeps the reference's own shape: paths are lists of indices into a pool
// of vertex / edge records (the reference's paths share PathVertex objects by pointer, and evaluate() mutates shared vertices -- cast(), the
// measure of connected end points -- which index lists reproduce for free), one lane runs one sample from the connected base path to its last
// connection -- in the probe entry.  The frame kernels (gbdpt_capi.hip, round 5) run the stages of a sample as separate launches over the
// sample's record in HBM: the shift stage (one lane per sample: connected base path, four offset paths, Jacobians; GTr::prepare), the connections
// (one lane per (s, t): GTr::connectPair<false>), the light-tracing connections (one lane per (s, 1), the only ones that allocate and walk
// manifolds: connectPair<true>).  Samples without a specular vertex never enter any of it.  The two subpaths come from the same walk
// (k_bd_paths) as the fast form's.
//
// One restatement in two places: this file follows oracle/gbdpt_oracle.hpp function by function (that file cites the reference line by line and
// is held by closed forms and by the estimator's expectation, tests/test_gbdpt_oracle.py); parity of the two is held per sample at 1e-9
// (tests/test_gbdpt_gpu.py).
#pragma once
#include "gbdpt_kernels.hip.h"

namespace gdpt_bd {

constexpr int GP_LEN = 32;                         // vertices of a path (the connected path: <= BD_MAX_DEPTH + 4)
constexpr int GV_REGION = 2 * BD_MAX_DEPTH + 4;    // records of ONE offset path: <= BD_MAX_DEPTH + 3 new + <= BD_MAX_DEPTH + 1 re-cloned after a failed walk
constexpr int GV_POOL = NSV + NEV + 2 + 4 * GV_REGION, GE_POOL = GV_POOL;   // vertex / edge records of one sample: both subpaths, the clones of createShiftablePath (2), four offset paths
constexpr int GM_MAX = BD_MAX_DEPTH + 4;           // vertices of a specular manifold: a path has at most BD_MAX_DEPTH + 3 vertices, so no chain is ever too long for it
                                                   // (12 until the fuzz of round 4: a 9-vertex chain between two long subpaths left both half-Jacobians 0 and their ratio NaN)

struct GPath {                                     // Path: m_vertices / m_edges as indices into the pool
    short v[GP_LEN], e[GP_LEN];
    int nv, ne;
    __device__ __forceinline__ int length() const { return ne; }
    __device__ __forceinline__ void clear() { nv = ne = 0; }
    __device__ __forceinline__ void pushV(int i) { if (nv < GP_LEN) v[nv] = (short)i; nv++; }
    __device__ __forceinline__ void pushE(int i) { if (ne < GP_LEN) e[ne] = (short)i; ne++; }
    __device__ __forceinline__ void reverse()
    {
        for (int i = 0, j = nv - 1; i < j; i++, j--) { const short t = v[i]; v[i] = v[j]; v[j] = t; }
        for (int i = 0, j = ne - 1; i < j; i++, j--) { const short t = e[i]; e[i] = e[j]; e[j] = t; }
    }
};
struct M2 {                                        // Matrix2x2, core/matrix.h:455-530
    Float m[2][2];
    __device__ __forceinline__ void setZero() { m[0][0] = m[0][1] = m[1][0] = m[1][1] = 0; }
    __device__ __forceinline__ void setIdentity() { m[0][0] = m[1][1] = 1; m[0][1] = m[1][0] = 0; }
    __device__ __forceinline__ Float det() const { return m[0][0] * m[1][1] - m[0][1] * m[1][0]; }
    __device__ __forceinline__ bool invert(M2 &t) const
    {
        const Float d = m[0][0] * m[1][1] - m[0][1] * m[1][0];
        if (fabs(d) <= 0x1p-1024) return false;
        const Float invDet = 1 / d;
        t.m[0][0] = m[1][1] * invDet; t.m[0][1] = -m[0][1] * invDet; t.m[1][1] = m[0][0] * invDet; t.m[1][0] = -m[1][0] * invDet;
        return true;
    }
};
__device__ __forceinline__ M2 m2(Float a, Float b, Float c, Float d) { M2 r; r.m[0][0] = a; r.m[0][1] = b; r.m[1][0] = c; r.m[1][1] = d; return r; }
__device__ __forceinline__ M2 m2mul(const M2 &a, const M2 &b)
{
    M2 r;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { Float sum = 0; for (int k = 0; k < 2; ++k) sum += a.m[i][k] * b.m[k][j]; r.m[i][j] = sum; }
    return r;
}
__device__ __forceinline__ M2 m2sub(const M2 &a, const M2 &b) { return m2(a.m[0][0] - b.m[0][0], a.m[0][1] - b.m[0][1], a.m[1][0] - b.m[1][0], a.m[1][1] - b.m[1][1]); }
__device__ __forceinline__ M2 m2neg(const M2 &a) { return m2(-a.m[0][0], -a.m[0][1], -a.m[1][0], -a.m[1][1]); }
enum { MV_PINNED = 0, MV_REFLECTION = 2, MV_REFRACTION = 3, MV_MOVABLE = 5 };                // manifold.h:88-95
struct MV {                                        // SpecularManifold::SimpleVertex, manifold.h:98-134
    int degenerate, type, object, pad;
    d3 p, dpdu, dpdv, n, gn, dndu, dndv, m;
    Float eta;
    M2 a, b, c, u, Tp;
    __device__ __forceinline__ d3 map(Float uu, Float vv) const { const Float tx = Tp.m[0][0] * uu + Tp.m[0][1] * vv, ty = Tp.m[1][0] * uu + Tp.m[1][1] * vv; return dpdu * tx + dpdv * ty; }
};
__device__ __forceinline__ void mv_init(MV &v, int type, d3 p)
{
    v.degenerate = 0; v.type = type; v.object = -1; v.pad = 0; v.p = p;
    v.dpdu = v.dpdv = v.n = v.gn = v.dndu = v.dndv = v.m = mk(0.0); v.eta = 1.0;
    v.a.setZero(); v.b.setZero(); v.c.setZero(); v.u.setZero(); v.Tp.setZero();
}
struct MuRec { int l, m; int extra[5]; };

// What a sample's stages share, in three pieces (round 5: round 4 kept all of it in ONE 56 KB per-lane workspace and ran a sample from its connected
// base path to its last connection in one lane -- 10 % lane utilisation, 385 GB of fabric traffic per launch):
//   GSamp     the SAMPLE's state, in HBM, written once by the shift stage (k_bdg_shift) and read-only afterwards: the pool of vertex / edge records,
//             the paths as index lists, the per-offset Jacobians and generalized geometry terms, the prefix products -- what every connection reads;
//   GScratch  what a LANE needs while it runs manifold code or builds a light path: the manifold's two vertex lists, the dense system, a small
//             transient pool (indices >= GV_POOL) -- only the shift stage and the light-tracing connections have one (persistent lanes);
constexpr int GL_POOL = GV_REGION + 4;             // transient records of ONE light-tracing connection: two clones + one offset path at a time
struct GSamp {
    BV v[GV_POOL]; BE e[GE_POOL];
    int nv, ne;
    int nvBase, neBase;                            // records in use after the base stage (both subpaths + the clones of createShiftablePath): offset path k takes [nvBase + GV_REGION k, + GV_REGION) (44 records at BD_MAX_DEPTH 20)
    int voidSample;                                // a pool ran out while the sample's paths were built: its connections are skipped (counted; asserted zero by tests and bench)
    GPath emitter, sensor[5], connect;
    MuRec mu[5];
    int success[5], couldConnectAfterB[5];
    Float jacobianDet[5][NSV + 4], genGeomTerm[5][NSV + 4];
    d3 impW[NEV + 1]; Float impP[NEV + 1];
    d3 radW[5][NSV + 1]; Float radP[5][NSV + 1];
    int vert_b;                                    // connectPath.vertexCount() - 1 - muRec.extra[1]: the sensor-side index of b
    unsigned connE, strictE, connS, strictS;       // bit i: vertex i of the emitter subpath / of sensor[0] is connectable in the sense of Path::isConnectable_GBDPT / of PathVertex::isConnectable
    unsigned lid;                                  // the sample's record in the chunk
};
struct GScratch {
    MV mv[GM_MAX], mp[GM_MAX];
    int nm, nmp, mIterations, pad;
    Float A[4 * (GM_MAX - 2) * (GM_MAX - 2)], Ai[4 * (GM_MAX - 2) * (GM_MAX - 2)];   // the dense system of SpecularManifold::det's mixed case: (2 (GM_MAX - 2)) squared, twice
    BV lv[GL_POOL]; BE le[GL_POOL];
    int nlv, nle;
    GPath offsetEmitter, connectedBase;
};
struct GWork { GSamp s; GScratch x; };            // both for one lane: the probe entry (one sample, start to end, in one lane)

// its.dpdu / its.dpdv of a triangle hit (skdtree.h:373-380, trimesh.cpp:683-735): the edges, or the UV tangents of a mesh with texture coordinates
__device__ void tri_partials(const Ctx &c, int prim, d3 &dpdu, d3 &dpdv)
{
    prim = BD_PRIM(c, prim, "tri_partials");
    const TriShade &ts = c.V.shade[prim];
    const d3 dP1 = ts.p1 - ts.p0, dP2 = ts.p2 - ts.p0;
    dpdu = dP1; dpdv = dP2;
    if (c.V.uv && c.V.hasUV[prim]) {
        const TriUV t = c.V.uv[prim];
        const Float du1 = t.uv[2] - t.uv[0], dv1 = t.uv[3] - t.uv[1], du2 = t.uv[4] - t.uv[0], dv2 = t.uv[5] - t.uv[1];
        const Float determinant = du1 * dv2 - dv1 * du2;
        if (determinant == 0) {                    // trimesh.cpp:708-714: a degenerate parameterization falls back to a frame about the face normal
            const d3 n = normalize(cross(dP1, dP2));
            if (fabs(n.x) > fabs(n.y)) { const Float il = 1.0 / sqrt(n.x * n.x + n.z * n.z); dpdv = mk(n.z * il, 0.0, -n.x * il); }
            else { const Float il = 1.0 / sqrt(n.y * n.y + n.z * n.z); dpdv = mk(0.0, n.z * il, -n.y * il); }
            dpdu = cross(dpdv, n);
        } else {
            const Float invDet = 1.0 / determinant;
            dpdu = (dP1 * dv2 - dP2 * dv1) * invDet;
            dpdv = (dP2 * du1 - dP1 * du2) * invDet;
        }
    }
}

// HAS_X: the lane has a GScratch (manifold code, transient pool).  The connection kernel for t >= 2 builds GTrT<false>: every pool index is the sample's.
template <bool HAS_X>
struct GTrT {
    Ctx &c;
    GSamp &W;
    GScratch *X;                                   // nullptr in the connection kernel for t >= 2: nothing there allocates or touches a manifold
    bool localAlloc = false;                       // allocations go to the lane's transient pool (a light-tracing connection), not to the sample's
#ifdef GDPT_BD_PROFILE   /* development: lane clocks per section of prepareOffset (tools/gpu_gbdpt_profile.py) */
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0};
#define GPROF(i, stmt) do { const unsigned long long t0_ = clock64(); stmt; prof[i] += clock64() - t0_; } while (0)
#else
#define GPROF(i, stmt) do { stmt; } while (0)
#endif
    bool lightWalks = false;                       // connectPair<true, *>: the light path's offset paths will walk a manifold (set with its base path)
    int lightK = 0;                                // connectPair<true, 2>: 1..4 = build this one of the light path's four offset paths only (0: all four)
    unsigned overflow = 0;                         // a pool or a list ran out (the sample's result is then void: counted, asserted zero by the tests)
    __device__ GTrT(Ctx &c_, GSamp &w_, GScratch *x_) : c(c_), W(w_), X(x_) {}
    __device__ GTrT(Ctx &c_, GWork &w_) : c(c_), W(w_.s), X(&w_.x) {}

    // ---- pool: indices < GV_POOL are the sample's records, the others the lane's transient ones ----
    // (region: the lane builds ONE offset path of a sample while other lanes build the others -- each in its own slice of the sample's pool)
    int regV = -1, regV1 = 0, regE = 0, regE1 = 0;
    __device__ void setRegion(int k) { regV = W.nvBase + k * GV_REGION; regV1 = regV + GV_REGION; regE = W.neBase + k * GV_REGION; regE1 = regE + GV_REGION; }
    __device__ int allocV()
    {
        if (HAS_X && localAlloc) { if (X->nlv >= GL_POOL) { overflow++; return GV_POOL + GL_POOL - 1; } bv_clear(X->lv[X->nlv]); return GV_POOL + X->nlv++; }
        // (an overflowing region hands out the last slot of the lane's OWN region again -- the sample is void from here on, but another lane may be building its
        //  offset path in the next region at this moment and must not see a vertex clobbered under it)
        if (regV >= 0) { if (regV >= regV1 || regV >= GV_POOL) { overflow++; return regV1 - 1 < GV_POOL ? regV1 - 1 : GV_POOL - 1; } bv_clear(W.v[regV]); return regV++; }
        if (W.nv >= GV_POOL) { overflow++; return GV_POOL - 1; } bv_clear(W.v[W.nv]); return W.nv++;
    }
    __device__ int allocE()
    {
        if (HAS_X && localAlloc) { if (X->nle >= GL_POOL) { overflow++; return GE_POOL + GL_POOL - 1; } be_clear(X->le[X->nle]); return GE_POOL + X->nle++; }
        if (regV >= 0) { if (regE >= regE1 || regE >= GE_POOL) { overflow++; return regE1 - 1 < GE_POOL ? regE1 - 1 : GE_POOL - 1; } be_clear(W.e[regE]); return regE++; }
        if (W.ne >= GE_POOL) { overflow++; return GE_POOL - 1; } be_clear(W.e[W.ne]); return W.ne++;
    }
    __device__ __forceinline__ BV &PV(int i) { if (!HAS_X) return W.v[i]; return i < GV_POOL ? W.v[i] : X->lv[i - GV_POOL]; }
    __device__ __forceinline__ BE &PE_(int i) { if (!HAS_X) return W.e[i]; return i < GE_POOL ? W.e[i] : X->le[i - GE_POOL]; }
    __device__ int cloneV(int i) { const int j = allocV(); PV(j) = PV(i); return j; }
    __device__ BV &V_(const GPath &p, int i) { return PV(p.v[i]); }
    __device__ BE &E_(const GPath &p, int i) { return PE_(p.e[i]); }
    __device__ BV *VN(const GPath &p, int i) { return (i < 0 || i >= p.nv) ? nullptr : &PV(p.v[i]); }
    __device__ BE *EN(const GPath &p, int i) { return (i < 0 || i >= p.ne) ? nullptr : &PE_(p.e[i]); }

    // ---- PathVertex with transport modes (the fast form's helpers are radiance-only where the BSDF is symmetric) ----
    __device__ d3 gEval(const BV &v, const BV *pred, const BV *succ, int mode, int measure = M_AREA)
    {
        if (v.type != T_SURFACE) return bv_eval(c, v, pred, succ, mode, measure);
        const Surf sf = surf_of(c, v);
        const d3 wi = normalize(pred->p - v.p), wo = normalize(succ->p - v.p);
        const d3 wiL = toLocal(sf.fr, wi), woL = toLocal(sf.fr, wo);
        if (measure == M_AREA) measure = M_SOLID;
        d3 result; Float pdfUnused;
        bd_eval_pdf(sf.m, sf.R, wiL, woL, bsdf_measure(measure), mode == EImportance, BD_ALL, result, pdfUnused);
        const Float wiDotGeoN = dot(sf.geoN, wi), woDotGeoN = dot(sf.geoN, wo);
        if (wiDotGeoN * wiL.z <= 0 || woDotGeoN * woL.z <= 0) return mk(0.0);
        if (mode == EImportance) result = result * fabs((wiL.z * woDotGeoN) / (woL.z * wiDotGeoN));
        if (measure != M_DISCRETE && woL.z != 0) result = result / fabs(woL.z);
        return result;
    }
    // PathVertex::update, vertex.cpp:1165-1211
    __device__ bool gUpdate(BV &v, const BV *pred, const BV *succ, int mode, int measure)
    {
        v.pdf[mode] = bv_eval_pdf(c, v, pred, succ, mode, measure);
        v.pdf[1 - mode] = bv_eval_pdf(c, v, succ, pred, 1 - mode, measure);
        v.w[mode] = gEval(v, pred, succ, mode, measure);
        v.w[1 - mode] = gEval(v, succ, pred, 1 - mode, measure);
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
    // PathVertex::connect with explicit measures, vertex.cpp:1348-1370
    __device__ bool gConnect(const BV *pred, BV &vs, BE &edge, BV &vt, const BV *succ, int vsMeasure, int vtMeasure, bool knownVisible = false)
    {
        if (vs.type == T_EMITTER_SUPER) { if (!bv_cast_emitter(c, vt)) return false; }
        else if (vt.type == T_SENSOR_SUPER) return false;                                    // (no sensor shapes)
        if (!gUpdate(vs, pred, &vt, EImportance, vsMeasure)) return false;
        if (!gUpdate(vt, succ, &vs, ERadiance, vtMeasure)) return false;
        return edge_connect(c, edge, vs, vt, knownVisible);
    }
    // PathVertex::perturbDirection, vertex.cpp:488-679: the sensor sample, or a glossy surface vertex of a chain
    __device__ bool gPerturbDirection(BV &v, const BV *pred, const BE *predEdge, BE &succEdge, BV &succ, d3 d, Float dist, int mode)
    {
        be_clear(succEdge); bv_clear(succ);
        if (v.degenerate) return false;
        if (v.type == T_SENSOR_SAMPLE) {
            const Float value = sensor_direction(c, v.p, d), prob = value;
            if (value == 0 || prob <= 0x1p-1024) return false;
            v.w[EImportance] = mk(value) * (1.0 / fabs(dot(d, v.n)));
            v.w[ERadiance] = mk(value) / prob;
            v.pdf[EImportance] = 1.0; v.pdf[ERadiance] = prob;
            v.measure = M_SOLID;
        } else if (v.type == T_SURFACE) {
            const Surf sf = surf_of(c, v);
            const d3 wi = normalize(pred->p - v.p), wo = d;
            const d3 wiL = toLocal(sf.fr, wi), woL = toLocal(sf.fr, wo);
            d3 value; Float prob;
            bd_eval_pdf(sf.m, sf.R, wiL, woL, MEASURE_SOLID_ANGLE, mode == EImportance, BD_ALL, value, prob);
            if (is_zero(value) || prob <= 0x1p-1024) return false;
            v.w[mode] = value / prob;
            v.pdf[mode] = prob;
            const Float wiDotGeoN = dot(sf.geoN, wi), woDotGeoN = dot(sf.geoN, wo);
            if (wiDotGeoN * wiL.z <= 0 || woDotGeoN * woL.z <= 0) return false;
            v.measure = M_SOLID;
            v.componentType = ESmooth;
            d3 fRev; Float pRev;
            bd_eval_pdf(sf.m, sf.R, woL, wiL, MEASURE_SOLID_ANGLE, (1 - mode) == EImportance, BD_ALL, fRev, pRev);
            v.pdf[1 - mode] = pRev;
            if (v.pdf[1 - mode] <= 0x1p-1024) return false;
            v.w[1 - mode] = v.w[mode] * fabs((v.pdf[mode] * wiL.z) / (v.pdf[1 - mode] * woL.z));
            adjoint(v, mode, wiL, woL, wiDotGeoN, woDotGeoN);
        } else return false;
        if (!edge_extend(c, succEdge, v.p, d, succ, mode, true, dist)) { v.measure = M_INVALID; return false; }
        to_area(c, v, mode, *pred, *predEdge, succEdge, succ, d);
        return true;
    }
    // PathVertex::propagatePerturbation, vertex.cpp:681-790
    __device__ bool gPropagatePerturbation(BV &v, const BV *pred, BE &succEdge, BV &succ, int componentType, Float dist, int mode)
    {
        const Surf sf = surf_of(c, v);
        if (!(bsdfType(sf.m) & EDelta)) return false;
        be_clear(succEdge); bv_clear(succ);
        const d3 wi = normalize(pred->p - v.p);
        const d3 wiL = toLocal(sf.fr, wi);
        BSDFSample bs;
        bd_sample(sf.m, sf.R, wiL, 0.5, 0.5, mode == EImportance, componentType, bs);
        if (is_zero(bs.weight)) return false;
        const d3 wo = toWorld(sf.fr, bs.wo);
        const Float wiDotGeoN = dot(sf.geoN, wi), woDotGeoN = dot(sf.geoN, wo);
        if (wiDotGeoN * wiL.z <= 0 || woDotGeoN * bs.wo.z <= 0) return false;
        d3 f; Float prob;
        bd_eval_pdf(sf.m, sf.R, wiL, bs.wo, MEASURE_DISCRETE, mode == EImportance, BD_ALL, f, prob);
        if (prob <= 0x1p-1024) return false;
        v.w[mode] = f / prob;
        v.pdf[mode] = prob;
        v.measure = M_DISCRETE;
        v.componentType = componentType;
        if (is_zero(v.w[mode])) return false;
        d3 fRev; Float pRev;
        bd_eval_pdf(sf.m, sf.R, bs.wo, wiL, MEASURE_DISCRETE, (1 - mode) == EImportance, BD_ALL, fRev, pRev);
        v.pdf[1 - mode] = pRev;
        if (v.pdf[1 - mode] <= 0x1p-1024) return false;
        if (sf.m.type != 3) v.w[1 - mode] = v.w[mode];
        else v.w[1 - mode] = fRev / v.pdf[1 - mode];
        adjoint(v, mode, wiL, bs.wo, wiDotGeoN, woDotGeoN);
        if (!edge_extend(c, succEdge, v.p, wo, succ, mode, true, dist)) { v.measure = M_INVALID; return false; }
        return true;
    }
    // PathEdge::evalCached, edge.cpp:169-219 (scalar: every caller here asks for geometry terms)
    __device__ Float gEdgeEvalCached(const BE &e, const BV &pred, const BV &succ, unsigned what)
    {
        enum { EValueImp = 0x01, EValueRad = 0x02, ECosineImp = 0x04, ECosineRad = 0x08, EInverseSquareFalloff = 0x10, ETransmittance = 0x20 };
        Float result = 1.0;
        if (e.length == 0) return result;                                                    // (no EValue* request here)
        if ((what & ECosineImp) && bv_on_surface(pred) && bv_connectable(pred)) result *= fabs(dot(bv_sh_normal(c, pred), e.d));
        if ((what & ECosineRad) && bv_on_surface(succ) && bv_connectable(succ)) result *= fabs(dot(bv_sh_normal(c, succ), e.d));
        if (what & EInverseSquareFalloff) result /= e.length * e.length;
        if (what & ETransmittance) result *= e.tr[EImportance] * e.tr[EImportance];
        return result;
    }

    // ---- SpecularManifold, manifold.cpp ----
    __device__ void normalDerivative(const BV &v, d3 &dndu, d3 &dndv)                        // TriMesh::getNormalDerivative, trimesh.cpp:745-822
    {
        const int prim = BD_PRIM(c, v.prim, "normalDerivative");
        const TriShade &ts = c.V.shade[prim];
        dndu = dndv = mk(0.0);
        if (!(c.V.vn && ts.smooth)) return;
        const TriNormals vn = c.V.vn[prim];
        const d3 rel = v.p - ts.p0, du = ts.p1 - ts.p0, dv = ts.p2 - ts.p0;
        const Float b1 = dot(du, rel), b2 = dot(dv, rel), a11 = dot(du, du), a12 = dot(du, dv), a22 = dot(dv, dv);
        Float det = a11 * a22 - a12 * a12;
        if (det == 0) return;
        Float invDet = 1.0 / det;
        const Float u = (a22 * b1 - a12 * b2) * invDet, vv = (-a12 * b1 + a11 * b2) * invDet, w = 1 - u - vv;
        d3 N = vn.n1 * u + vn.n2 * vv + vn.n0 * w;
        const Float il = 1.0 / len(N); N = N * il;
        dndu = (vn.n1 - vn.n0) * il; dndu = dndu - N * dot(N, dndu);
        dndv = (vn.n2 - vn.n0) * il; dndv = dndv - N * dot(N, dndv);
        if (c.V.uv && c.V.hasUV[prim]) {
            const TriUV t = c.V.uv[prim];
            const Float d1x = t.uv[2] - t.uv[0], d1y = t.uv[3] - t.uv[1], d2x = t.uv[4] - t.uv[0], d2y = t.uv[5] - t.uv[1];
            det = d1x * d2y - d1y * d2x;
            if (det == 0) { dndu = dndv = mk(0.0); return; }
            invDet = 1.0 / det;
            const d3 du_ = (dndu * d2y - dndv * d1y) * invDet, dv_ = (dndv * d1x - dndu * d2x) * invDet;
            dndu = du_; dndv = dv_;
        }
    }
    __device__ void manifoldSurface(MV &m, const BV &v)                                       // manifold.cpp:101-122,480-507
    {
        Vertex vx; vx.p = v.p; vx.prim = BD_PRIM(c, v.prim, "manifoldSurface"); vx.u = v.u; vx.v = v.v;
        const Shading sh = shading_at<true>(c.V, vx);
        m.p = v.p; m.gn = sh.geoN; m.n = sh.fr.n;
        tri_partials(c, v.prim, m.dpdu, m.dpdv);
        normalDerivative(v, m.dndu, m.dndv);
        Float invLen = 1 / len(m.dpdu);
        m.dpdu = m.dpdu * invLen; m.dndu = m.dndu * invLen;
        const Float dp = dot(m.dpdu, m.dpdv);
        const d3 dpdv = m.dpdv - m.dpdu * dp, dndv = m.dndv - m.dndu * dp;
        invLen = 1 / len(dpdv);
        m.dpdv = dpdv * invLen; m.dndv = dndv * invLen;
    }
    __device__ bool manifoldInit(const GPath &path, int start, int end)                      // manifold.cpp:59-170
    {
        const int step = start < end ? 1 : -1;
        if (bv_super(V_(path, start))) start += step;
        if (bv_super(V_(path, end))) end -= step;
        X->nm = 0;
        if ((end - start) * step + 1 > GM_MAX) return false;
        mv_init(X->mv[X->nm++], MV_PINNED, V_(path, start).p);
        for (int i = start + step; i != end; i += step) {
            const BV &pred = V_(path, i - step), &vertex = V_(path, i), &succ = V_(path, i + step);
            MV &m = X->mv[X->nm++];
            mv_init(m, MV_PINNED, mk(0.0));
            if (vertex.type != T_SURFACE) return false;
            manifoldSurface(m, vertex);
            m.object = c.V.shade[BD_PRIM(c, vertex.prim, "manifoldInit")].material;
            m.degenerate = !bv_connectable(vertex);
            const d3 wPred = pred.p - m.p, wSucc = succ.p - m.p;
            if (dot(m.gn, wPred) * dot(m.gn, wSucc) < 0) { m.type = MV_REFRACTION; m.eta = bsdf_eta(c.V.mats[m.object]); }
            else { m.type = MV_REFLECTION; m.eta = 1.0; }
        }
        mv_init(X->mv[X->nm++], MV_MOVABLE, V_(path, end).p);
        return true;
    }
    __device__ bool manifoldTangents()                                                       // manifold.cpp:172-400
    {
        const int n = X->nm - 1;
        X->mv[0].Tp.setZero();
        X->mv[X->nm - 1].Tp.setIdentity();
        if (X->nm == 2) return true;
        for (int i = 0; i < n; ++i) {
            MV *v = &X->mv[i];
            d3 wo = v[1].p - v[0].p;
            Float ilo = len(wo);
            if (ilo == 0) return false;
            ilo = 1 / ilo; wo = wo * ilo;
            if (v[0].type == MV_PINNED) { v[0].a.setZero(); v[0].b.setIdentity(); v[0].c.setZero(); continue; }
            d3 wi = v[-1].p - v[0].p;
            Float ili = len(wi);
            if (ili == 0) return false;
            ili = 1 / ili; wi = wi * ili;
            if (v[0].type != MV_REFLECTION && v[0].type != MV_REFRACTION) return false;
            Float eta = v[0].eta;
            const bool normalizeH = !(v[0].type == MV_REFRACTION && eta == 1);
            d3 H;
            Float ilh;
            if (normalizeH) {
                if (dot(wi, v[0].gn) < 0) eta = 1 / eta;
                H = wi + wo * eta;
                ilh = 1 / len(H);
                H = H * ilh;
            } else { H = wi + wo; ilh = 1.0; }
            const Float dot_H_n = dot(v[0].n, H), dot_H_dndu = dot(v[0].dndu, H), dot_H_dndv = dot(v[0].dndv, H), dot_u_n = dot(v[0].dpdu, v[0].n), dot_v_n = dot(v[0].dpdv, v[0].n);
            d3 s_ = v[0].dpdu - v[0].n * dot_u_n, t_ = v[0].dpdv - v[0].n * dot_v_n;
            ilo *= eta * ilh; ili *= ilh;
            d3 dH_du = (v[-1].dpdu - wi * dot(wi, v[-1].dpdu)) * ili, dH_dv = (v[-1].dpdv - wi * dot(wi, v[-1].dpdv)) * ili;
            if (normalizeH) { dH_du = dH_du - H * dot(dH_du, H); dH_dv = dH_dv - H * dot(dH_dv, H); }
            v[0].a = m2(dot(dH_du, s_), dot(dH_dv, s_), dot(dH_du, t_), dot(dH_dv, t_));
            dH_du = -v[0].dpdu * (ili + ilo) + wi * (dot(wi, v[0].dpdu) * ili) + wo * (dot(wo, v[0].dpdu) * ilo);
            dH_dv = -v[0].dpdv * (ili + ilo) + wi * (dot(wi, v[0].dpdv) * ili) + wo * (dot(wo, v[0].dpdv) * ilo);
            if (normalizeH) { dH_du = dH_du - H * dot(dH_du, H); dH_dv = dH_dv - H * dot(dH_dv, H); }
            v[0].b = m2(dot(dH_du, s_) - dot(v[0].dpdu, v[0].dndu) * dot_H_n - dot_u_n * dot_H_dndu,
                        dot(dH_dv, s_) - dot(v[0].dpdu, v[0].dndv) * dot_H_n - dot_u_n * dot_H_dndv,
                        dot(dH_du, t_) - dot(v[0].dpdv, v[0].dndu) * dot_H_n - dot_v_n * dot_H_dndu,
                        dot(dH_dv, t_) - dot(v[0].dpdv, v[0].dndv) * dot_H_n - dot_v_n * dot_H_dndv);
            dH_du = (v[1].dpdu - wo * dot(wo, v[1].dpdu)) * ilo;
            dH_dv = (v[1].dpdv - wo * dot(wo, v[1].dpdv)) * ilo;
            if (normalizeH) { dH_du = dH_du - H * dot(dH_du, H); dH_dv = dH_dv - H * dot(dH_dv, H); }
            v[0].c = m2(dot(dH_du, s_), dot(dH_dv, s_), dot(dH_du, t_), dot(dH_dv, t_));
            s_ = normalize(s_);
            t_ = cross(v[0].n, s_);
            v[0].m = mk(dot(s_, H), dot(t_, H), dot(v[0].n, H));
            if (dot(H, v[0].gn) < 0) v[0].m = -v[0].m;
        }
        M2 Li;
        if (!X->mv[0].b.invert(Li)) return false;
        for (int i = 0; i < n - 1; ++i) {
            X->mv[i].u = m2mul(Li, X->mv[i].c);
            const M2 temp = m2sub(X->mv[i + 1].b, m2mul(X->mv[i + 1].a, X->mv[i].u));
            if (!temp.invert(Li)) return false;
        }
        X->mv[n - 1].Tp = m2neg(m2mul(Li, X->mv[n - 1].c));
        for (int i = n - 2; i >= 0; --i) X->mv[i].Tp = m2neg(m2mul(X->mv[i].u, X->mv[i + 1].Tp));
        return true;
    }
    __device__ static d3 reflectAbout(d3 wi, d3 n) { return n * (2 * dot(wi, n)) - wi; }      // util.cpp:763-765
    __device__ static d3 refractAbout(d3 wi, d3 n, Float eta)                                 // util.cpp:774-792
    {
        if (eta == 1) return -wi;
        const Float cosThetaI = dot(wi, n);
        if (cosThetaI > 0) eta = 1 / eta;
        const Float cosThetaTSqr = 1 - (1 - cosThetaI * cosThetaI) * (eta * eta);
        if (cosThetaTSqr <= 0.0) return mk(0.0);
        return n * (cosThetaI * eta - (cosThetaI < 0 ? -1.0 : (cosThetaI > 0 ? 1.0 : 0.0)) * sqrt(cosThetaTSqr)) - wi * eta;
    }
    __device__ bool manifoldProject(d3 d)                                                    // manifold.cpp:402-510
    {
        const MV &last = X->mv[X->nm - 1];
        const Float du = dot(d, last.dpdu), dv = dot(d, last.dpdv);
        d3 ro = mk(0.0), rd = mk(0.0);
        X->nmp = 0;
        for (int i = 0; i < X->nm; ++i) {
            X->mp[X->nmp++] = X->mv[i];
            MV &vertex = X->mp[i];
            if (i == 0) {
                const d3 p0 = X->mv[0].p + X->mv[0].map(du, dv), p1 = X->mv[1].p + X->mv[1].map(du, dv);
                ro = p0; rd = normalize(p1 - p0);
                vertex.p = ro;
                continue;
            } else if (vertex.type == MV_MOVABLE) {
                const Float dp = dot(rd, vertex.n);
                if (fabs(dp) < GD_EPSILON) return false;
                const Float t = dot(vertex.p - ro, vertex.n) / dp;
                vertex.p = ro + rd * t;
                break;
            } else if (vertex.type == MV_REFLECTION || vertex.type == MV_REFRACTION) {
                Hit h;
                if (!closest_hit(c, ro, rd, GD_EPSILON, GD_INF, h)) return false;
                BV hv; bv_clear(hv);
                fill_surface(c, h, hv);
                Vertex vx; vx.p = hv.p; vx.prim = hv.prim; vx.u = hv.u; vx.v = hv.v;
                const d3 n = shading_at<true>(c.V, vx).fr.n;
                d3 s_, dpdvUnused;
                tri_partials(c, hv.prim, s_, dpdvUnused);
                s_ = normalize(s_ - n * dot(n, s_));
                const d3 t_ = cross(n, s_);
                const d3 m = s_ * vertex.m.x + t_ * vertex.m.y + n * vertex.m.z;
                d3 out;
                if (vertex.type == MV_REFLECTION) out = reflectAbout(-rd, m);
                else { out = refractAbout(-rd, m, bsdf_eta(c.V.mats[c.V.shade[hv.prim].material])); if (is_zero(out)) return false; }
                ro = hv.p; rd = out;
                if (vertex.object != c.V.shade[hv.prim].material) return false;
                manifoldSurface(vertex, hv);
            } else return false;
        }
        return true;
    }
    __device__ bool manifoldMove(d3 target, d3 n)                                            // manifold.cpp:512-635
    {
        MV &last = X->mv[X->nm - 1];
        if (X->nm == 2 && X->mv[0].type == MV_PINNED) return true;
        const Float invScale = 1.0 / fmax(fmax(fabs(target.x), fabs(target.y)), fabs(target.z));
        Float stepSize = 1;
        if (fabs(n.x) > fabs(n.y)) { const Float il = 1.0 / sqrt(n.x * n.x + n.z * n.z); last.dpdv = mk(n.z * il, 0.0, -n.x * il); }   // coordinateSystem(n, dpdu, dpdv), util.cpp:592-601
        else { const Float il = 1.0 / sqrt(n.y * n.y + n.z * n.z); last.dpdv = mk(0.0, n.z * il, -n.y * il); }
        last.dpdu = cross(last.dpdv, n);
        last.n = n;
        X->mIterations = 0;
        while (X->mIterations < 20) {
            const d3 rel = target - X->mv[X->nm - 1].p;
            Float dist = len(rel), newDist;
            if (dist * invScale < GD_EPSILON) {
                dist = len(X->mv[X->nm - 1].p - X->mv[X->nm - 2].p);
                if (dist * invScale < GD_EPSILON) return false;
                return true;
            }
            X->mIterations++;
            if (!manifoldTangents()) return false;
            bool failure = false;
            if (!manifoldProject(rel * stepSize)) failure = true;
            else {
                newDist = len(target - X->mp[X->nmp - 1].p);
                if (newDist > dist) failure = true;
            }
            if (!failure) {
                for (int i = 0; i < X->nmp; i++) X->mv[i] = X->mp[i];                             // m_proposal.swap(m_vertices): the old vertices are never read again
                X->nm = X->nmp;
                stepSize = fmin((Float)1.0, stepSize * 2.0);
                continue;
            }
            stepSize /= 2.0;
        }
        return false;
    }
    __device__ bool manifoldUpdate(GPath &path, int start, int end)                          // manifold.cpp:637-757
    {
        const int step = start < end ? 1 : -1, mode = start < end ? EImportance : ERadiance;
        const int last = X->nm - 2;
        for (int j = 0, i = start; j < last; ++j, i += step) {
            const MV &v = X->mv[j], &vn = X->mv[j + 1];
            BV *pred = VN(path, i - step); BV &vertex = V_(path, i), &succ = V_(path, i + step);
            const int predEdgeIdx = (mode == EImportance) ? i - step : i - step - 1;
            BE *predEdge = EN(path, predEdgeIdx); BE &succEdge = E_(path, predEdgeIdx + step);
            d3 d = vn.p - v.p;
            const Float length = len(d);
            d = d / length;
            if (!v.degenerate) { if (!gPerturbDirection(vertex, pred, predEdge, succEdge, succ, d, length, mode)) return false; }
            else if (!gPropagatePerturbation(vertex, pred, succEdge, succ, v.type == MV_REFRACTION ? EDeltaTransmission : EDeltaReflection, length, mode)) return false;
            const Float relerr = len(vn.p - succ.p) / fmax(fmax(fabs(vn.p.x), fabs(vn.p.y)), fabs(vn.p.z));
            if (relerr > (Float)1e-3f) return false;
        }
        return true;
    }
    // the dense inverse and determinant of SpecularManifold::det's mixed case (Gauss-Jordan / LU with partial pivoting, as the oracle)
    __device__ bool denseInverse(int n)
    {
        Float *A = X->A, *Ai = X->Ai;
        for (int i = 0; i < n * n; ++i) Ai[i] = 0.0;
        for (int i = 0; i < n; ++i) Ai[i * n + i] = 1.0;
        for (int col = 0; col < n; ++col) {
            int piv = col;
            for (int r = col + 1; r < n; ++r) if (fabs(A[r * n + col]) > fabs(A[piv * n + col])) piv = r;
            if (A[piv * n + col] == 0) return false;
            if (piv != col) for (int k = 0; k < n; ++k) { Float t = A[piv * n + k]; A[piv * n + k] = A[col * n + k]; A[col * n + k] = t; t = Ai[piv * n + k]; Ai[piv * n + k] = Ai[col * n + k]; Ai[col * n + k] = t; }
            const Float inv = 1.0 / A[col * n + col];
            for (int k = 0; k < n; ++k) { A[col * n + k] *= inv; Ai[col * n + k] *= inv; }
            for (int r = 0; r < n; ++r) {
                if (r == col) continue;
                const Float f = A[r * n + col];
                if (f == 0) continue;
                for (int k = 0; k < n; ++k) { A[r * n + k] -= f * A[col * n + k]; Ai[r * n + k] -= f * Ai[col * n + k]; }
            }
        }
        return true;
    }
    __device__ Float denseDet(int n)                                                         // of X->Ai, destroyed
    {
        Float *A = X->Ai;
        Float det = 1.0;
        for (int col = 0; col < n; ++col) {
            int piv = col;
            for (int r = col + 1; r < n; ++r) if (fabs(A[r * n + col]) > fabs(A[piv * n + col])) piv = r;
            if (A[piv * n + col] == 0) return 0.0;
            if (piv != col) { for (int k = 0; k < n; ++k) { const Float t = A[piv * n + k]; A[piv * n + k] = A[col * n + k]; A[col * n + k] = t; } det = -det; }
            det *= A[col * n + col];
            for (int r = col + 1; r < n; ++r) {
                const Float f = A[r * n + col] / A[col * n + col];
                if (f == 0) continue;
                for (int k = col; k < n; ++k) A[r * n + k] -= f * A[col * n + k];
            }
        }
        return det;
    }

    // ---- Path: geometry terms, Jacobians (path.cpp:380-454; manifold.cpp:759-951) ----
    __device__ Float manifoldG(const GPath &p, int a, int b)
    {
        if (abs(a - b) == 1) {
            if (a > b) { const int t = a; a = b; b = t; }
            return gEdgeEvalCached(E_(p, a), V_(p, a), V_(p, b), 0x04 | 0x08 | 0x10);        // EGeometricTerm
        }
        const int step = b > a ? 1 : -1;
        if (!manifoldInit(p, a, b)) return 0.0;
        MV &last = X->mv[X->nm - 1];
        const BV &vb = V_(p, b);
        last.n = bv_on_surface(vb) ? bv_sh_normal(c, vb) : E_(p, a < b ? (b - 1) : b).d;
        if (fabs(last.n.x) > fabs(last.n.y)) { const Float il = 1.0 / sqrt(last.n.x * last.n.x + last.n.z * last.n.z); last.dpdv = mk(last.n.z * il, 0.0, -last.n.x * il); }
        else { const Float il = 1.0 / sqrt(last.n.y * last.n.y + last.n.z * last.n.z); last.dpdv = mk(0.0, last.n.z * il, -last.n.y * il); }
        last.dpdu = cross(last.dpdv, last.n);
        if (!manifoldTangents()) return 0.0;
        const d3 d = X->mv[1].p - X->mv[0].p;
        const Float lengthSqr = len2(d), invLength = 1 / sqrt(lengthSqr);
        Float result = len(cross(X->mv[1].map(1, 0), X->mv[1].map(0, 1))) / lengthSqr;
        if (bv_on_surface(V_(p, a))) result *= fabs(dot(d, bv_sh_normal(c, V_(p, a)))) * invLength;
        if (bv_on_surface(V_(p, a + step))) result *= fabs(dot(d, bv_sh_normal(c, V_(p, a + step)))) * invLength;
        return result;
    }
    __device__ Float pathG(const GPath &p, int i, int j)
    {
        if (i >= j) return 1.0;
        if (j != i + 1) return manifoldG(p, i, j);
        const BE &e = E_(p, i);
        const Float cosI = fabs(dot(e.d, bv_sh_normal(c, V_(p, i)))), cosJ = fabs(dot(e.d, bv_sh_normal(c, V_(p, j))));
        return cosI * cosJ / (e.length * e.length);
    }
    // The geometry terms of ONE path, each computed once: calcSpecularPDFChange is asked for every connectable vertex v of a path in turn, and its
    // products over [v, k] share every factor with the call before (the plain terms of the edges: two shading-frame lookups each; the chains' generalized
    // terms: a manifold each).  The products themselves are formed as before, factor by factor in the same order -- same values, bit for bit.
    struct GCache { unsigned haveE = 0, haveS = 0; Float e[GP_LEN], s[GP_LEN]; };
    __device__ Float multiG(const GPath &p, int a, int b, GCache *gc = nullptr)
    {
        if (a == 0) ++a; else if (a == p.length()) --a;
        if (b == 0) ++b; else if (b == p.length()) --b;
        const int step = b > a ? 1 : -1;
        while (!bv_connectable(V_(p, b))) b -= step;
        while (!bv_connectable(V_(p, a))) a += step;
        Float result = 1;
        for (int i = a + step, start = a; i != b + step; i += step)
            if (bv_connectable(V_(p, i))) {
                Float g;
                if (gc && step > 0) {                                                        // (a segment is named by where it starts: its end is the next connectable vertex)
                    if (!((gc->haveS >> start) & 1u)) { gc->s[start] = manifoldG(p, start, i); gc->haveS |= 1u << start; }
                    g = gc->s[start];
                } else g = manifoldG(p, start, i);
                result *= g; start = i;
            }
        return result;
    }
    __device__ Float manifoldDet(const GPath &p, int a, int b, int cI)
    {
        const int k = p.length();
        if (a == 0 || a == k) { const int t = a; a = cI; cI = t; }
        const int step = b > a ? 1 : -1;
        int nGlossy = 0, nSpecular = 0;
        for (int i = a + step; i != cI; i += step) { if (bv_connectable(V_(p, i))) ++nGlossy; else ++nSpecular; }
        if (nGlossy <= 1) return 1.0;
        if (!manifoldInit(p, a, cI)) return 0.0;
        const int b_idx = abs(b - a);
        MV &vb = X->mv[b_idx];
        vb.n = bv_sh_normal(c, V_(p, b));
        if (fabs(vb.n.x) > fabs(vb.n.y)) { const Float il = 1.0 / sqrt(vb.n.x * vb.n.x + vb.n.z * vb.n.z); vb.dpdv = mk(vb.n.z * il, 0.0, -vb.n.x * il); }
        else { const Float il = 1.0 / sqrt(vb.n.y * vb.n.y + vb.n.z * vb.n.z); vb.dpdv = mk(0.0, vb.n.z * il, -vb.n.y * il); }
        vb.dpdu = cross(vb.dpdv, vb.n);
        if (!manifoldTangents()) return 0.0;
        X->mv[b_idx].a.setZero(); X->mv[b_idx].b.setIdentity(); X->mv[b_idx].c.setZero();
        if (nSpecular == 0) {
            M2 Di, D = X->mv[1].b;
            Float det = D.det();
            for (int i = 2; i < X->nm - 1; ++i) {
                if (!D.invert(Di)) return 0.0;
                D = m2sub(X->mv[i].b, m2mul(m2mul(X->mv[i].a, Di), X->mv[i - 1].c));
                det *= D.det();
            }
            return fabs(1 / det);
        }
        const int nv = nGlossy + nSpecular, N = 2 * nv;
        for (int i = 0; i < N * N; ++i) X->A[i] = 0.0;
        for (int j = 0; j < nv; ++j) {
            const int i = j;
            for (int q = -1; q <= 1; ++q) {
                const int cj = j + q;
                if (cj < 0 || cj >= nv) continue;
                const M2 &mm = q < 0 ? X->mv[j + 1].a : (q == 0 ? X->mv[j + 1].b : X->mv[j + 1].c);
                X->A[(2 * i) * N + 2 * cj] = mm.m[0][0]; X->A[(2 * i) * N + 2 * cj + 1] = mm.m[0][1]; X->A[(2 * i + 1) * N + 2 * cj] = mm.m[1][0]; X->A[(2 * i + 1) * N + 2 * cj + 1] = mm.m[1][1];
            }
        }
        if (!denseInverse(N)) return 0.0;
        for (int i = 0; i < nv; ++i) {
            if (!X->mv[i + 1].degenerate) continue;
            for (int q = 0; q < N; ++q) { X->Ai[(2 * i) * N + q] = 0; X->Ai[(2 * i + 1) * N + q] = 0; X->Ai[q * N + 2 * i] = 0; X->Ai[q * N + 2 * i + 1] = 0; }
            X->Ai[(2 * i) * N + 2 * i] = 1; X->Ai[(2 * i + 1) * N + 2 * i + 1] = 1;
        }
        return fabs(denseDet(N));
    }
    __device__ Float halfJacobian(const GPath &p, int a, int b, int cI)                      // path.cpp:380-394
    {
        Float value = 1.0;
        value /= V_(p, a).pdf[ERadiance];
        value *= pathG(p, a - 1, a) / pathG(p, b, a);
        value *= manifoldDet(p, a, b, cI);
        return value;
    }
    __device__ Float calcSpecularPDFChange(const GPath &p, int cI, bool lightpath = false, GCache *gc = nullptr)   // path.cpp:403-421
    {
        Float value = 1.0;
        const int k = p.length() - 1;
        cI = max(1, cI);
        for (int i = cI + 1; i <= k; i++)
            if (bv_connectable(V_(p, lightpath ? i - 1 : i))) {
                Float g;
                if (gc) {
                    if (!((gc->haveE >> i) & 1u)) { gc->e[i] = pathG(p, i - 1, i); gc->haveE |= 1u << i; }
                    g = gc->e[i];
                } else g = pathG(p, i - 1, i);
                value *= g;
            }
        if (value <= (Float)0.0) return 1.0;
        return multiG(p, cI, k, gc) / value;
    }

    // ---- ManifoldPerturbation, mut_manifold.cpp ----
    __device__ int getSpecularChainEnd(const GPath &path, int pos, int step)                 // :1230-1262
    {
        while (true) {
            if (pos < 0 || pos > path.length()) return -1;
            const BV &vertex = V_(path, pos);
            if (vertex.type != T_SURFACE) break;
            const Float roughness = vertex.prim == BV_ENV_PRIM ? GD_INF : mat_roughness(c.V.mats[c.V.shade[max(vertex.prim, 0)].material]);   // (the environment's sphere: a black diffuse BSDF)
            if (bv_connectable(vertex) && roughness >= c.cfg.shiftThreshold) break;
            pos += step;
        }
        return pos;
    }
    __device__ bool computeMuRec(const GPath &source, MuRec &mu)                             // :1264-1296
    {
        const int k = source.length();
        mu.l = mu.m = 0; for (int i = 0; i < 5; i++) mu.extra[i] = 0;
        if (!bv_connectable(V_(source, k - 1))) return false;
        const int step = -1, a = k - 1;
        int b, cI;
        if ((b = getSpecularChainEnd(source, a + step, step)) == -1) return false;
        if ((cI = getSpecularChainEnd(source, b + step, step)) == -1) return false;
        mu.l = min(a, cI); mu.m = max(a, cI);
        mu.extra[0] = a; mu.extra[1] = b; mu.extra[2] = cI; mu.extra[3] = step; mu.extra[4] = ERadiance;
        return true;
    }
    __device__ bool mutPerturbDirection(const GPath &source, GPath &proposal, int step, int a, Float offX, Float offY)   // :938-986
    {
        const BE &succEdge_old = E_(source, a - 1);
        BV &pred = V_(proposal, a - step), &vertex = V_(proposal, a), &succ = V_(proposal, a + step);
        BE &predEdge = E_(proposal, a - 1 - step), &succEdge = E_(proposal, a - 1);
        const CameraD &cam = c.S->cam;
        const BV &sensor = V_(source, source.length() - 1);
        const Float ppx = sensor.u + offX, ppy = sensor.v + offY;
        const d3 rd = cam_to_world(c, centre_ray_dir(c, ppx * cam.invW, ppy * cam.invH));
        const Float focusDistance = focus_distance(cam) / fabs(dot(c.cam.dir, rd));
        const d3 d = normalize((c.cam.pos + rd * focusDistance) - V_(source, a).p);
        return gPerturbDirection(vertex, &pred, &predEdge, succEdge, succ, d, succEdge_old.length, ERadiance);
    }
    __device__ bool mutPropagatePerturbation(const GPath &source, GPath &proposal, int step, int a, int b, int mode)   // :989-1149
    {
        for (int i = a + step; i != b; i += step) {
            const BV &pred_old = V_(source, i - step), &vertex_old = V_(source, i), &succ_old = V_(source, i + step);
            const BE &succEdge_old = E_(source, mode == EImportance ? i : i - 1);
            BV &pred = V_(proposal, i - step), &vertex = V_(proposal, i), &succ = V_(proposal, i + step);
            BE &predEdge = E_(proposal, mode == EImportance ? i - step : i - 1 - step), &succEdge = E_(proposal, mode == EImportance ? i : i - 1);
            if (vertex_old.type != T_SURFACE) return false;
            const Surf so = surf_of(c, vertex_old), sn = surf_of(c, vertex);
            const d3 wi_old = toLocal(so.fr, normalize(pred_old.p - vertex_old.p)), wo_old = toLocal(so.fr, normalize(succ_old.p - vertex_old.p));
            const bool reflection = wi_old.z * wo_old.z > 0;
            const Float eta = bsdf_eta(so.m);
            const d3 wi_world = normalize(pred.p - vertex.p);
            d3 wo_world = mk(0.0);
            if (prim_material(c, vertex_old.prim) != prim_material(c, vertex.prim)) return false;
            if (bv_connectable(vertex_old)) {
                d3 m = mk(0.0);
                if (reflection) m = normalize(wi_old + wo_old);
                else if (eta != 1) m = normalize(wi_old.z < 0 ? (wi_old * eta + wo_old) : (wi_old + wo_old * eta));
                m = toWorld(sn.fr, m.z > 0 ? m : -m);
                if (reflection) wo_world = reflectAbout(wi_world, m);
                else if (eta != 1) { wo_world = refractAbout(wi_world, m, eta); if (is_zero(wo_world)) return false; }
                else wo_world = -wi_world;
                if (!gPerturbDirection(vertex, &pred, &predEdge, succEdge, succ, wo_world, succEdge_old.length, mode)) return false;
            } else {
                if (!gPropagatePerturbation(vertex, &pred, succEdge, succ, reflection ? EDeltaReflection : EDeltaTransmission, succEdge_old.length, mode)) return false;
            }
        }
        return true;
    }
    __device__ bool mutManifoldWalk(const GPath &source, GPath &proposal, int b, int cI)     // :1151-1227
    {
        const BV &vb_old = V_(source, b), &vb_new = V_(proposal, b);
        d3 n1 = bv_geo_normal(c, vb_old), n2 = bv_geo_normal(c, vb_new);
        d3 rel = vb_new.p - vb_old.p;
        Float l = len(rel);
        if (l == 0) return false;
        rel = rel / l;
        if (dot(n1, n2) < 0) n1 = -n1;
        d3 n = n1 + n2;
        n = n - rel * dot(rel, n);
        l = len(n);
        if (l == 0) return false;
        n = n / l;
        if (!manifoldInit(source, cI, b)) return false;
        const d3 p0 = X->mv[1].p;
        if (!manifoldMove(vb_new.p, n)) return false;
        if (!manifoldUpdate(proposal, cI, b)) return false;
        if (!manifoldMove(vb_old.p, n)) return false;
        const d3 p1 = X->mv[1].p;
        const Float relerr = len(p0 - p1) / c.cfg.sceneRadius;
        if (relerr > 10.0 * GD_EPSILON) return false;
        return true;
    }
    __device__ bool generateOffsetPath(const GPath &source, GPath &proposal, MuRec &mu, Float offX, Float offY, int &couldConnectBehindB, bool lightPath)   // :806-936
    {
        const int k = source.length();
        couldConnectBehindB = 0;
        if (!bv_connectable(V_(source, k - 1))) return false;
        const int step = -1, a = k - 1;
        int b, cI;
        if ((b = getSpecularChainEnd(source, a + step, step)) == -1) return false;
        if ((cI = getSpecularChainEnd(source, b + step, step)) == -1) return false;
        const int l = min(a, cI), m = max(a, cI), q = min(b, b + step);
        mu.l = l; mu.m = m;
        mu.extra[0] = a; mu.extra[1] = b; mu.extra[2] = cI; mu.extra[3] = step; mu.extra[4] = ERadiance;
        proposal.clear();
        for (int i = 0; i < l + 1; ++i) { proposal.pushV(source.v[i]); if (i + 1 < l + 1) proposal.pushE(source.e[i]); }
        proposal.pushE(allocE());
        for (int i = l + 1; i < m; ++i) { proposal.pushV(allocV()); proposal.pushE(allocE()); }
        for (int i = m; i < k + 1; ++i) { proposal.pushV(source.v[i]); if (i + 1 < k + 1) proposal.pushE(source.e[i]); }
        proposal.v[a] = (short)cloneV(proposal.v[a]);
        proposal.v[cI] = (short)cloneV(proposal.v[cI]);
        if (!mutPerturbDirection(source, proposal, step, a, offX, offY)) return false;
        if (!mutPropagatePerturbation(source, proposal, step, a, b, ERadiance)) return false;
        if (!bv_connectable(V_(proposal, b))) return false;
        if (abs(b - cI) > 1) {
            bool walkSuccess;
            GPROF(1, walkSuccess = mutManifoldWalk(source, proposal, b, cI));
            if (!walkSuccess && lightPath) return false;
            if (!walkSuccess) {
                for (int i = b + step; i != cI; i += step) proposal.v[i] = (short)cloneV(source.v[i]);
                mu.extra[2] = b + step;
            }
        }
        couldConnectBehindB = gConnect(VN(proposal, q - 1), V_(proposal, q), E_(proposal, q), V_(proposal, q + 1), VN(proposal, q + 2),
                                       bv_connectable(V_(source, q)) ? M_AREA : M_DISCRETE, bv_connectable(V_(source, q + 1)) ? M_AREA : M_DISCRETE) ? 1 : 0;
        if (lightPath && !couldConnectBehindB) return false;
        if (m >= k - 1) { BV &s1 = V_(proposal, k - 1); sensor_sample_position(c, s1.p, V_(proposal, k - 2).p - s1.p, s1.u, s1.v); }
        for (int i = 0; i <= proposal.length(); i++) {
            if (proposal.v[i] == source.v[i]) continue;                                  // (a shared record: rr and componentType are its own already -- and other lanes may be reading it)
            BV &pv = V_(proposal, i); const BV &sv = V_(source, i);
            pv.rr = sv.rr;
            if (pv.type == T_SURFACE && pv.componentType == 0) pv.componentType = sv.componentType;
        }
        return true;
    }

    // ---- MIS weights, path.cpp:49-378 ----
    // The reference fills pdfImp / pdfRad[0..k] per path (Path::miWeight*NoSweep_GBDPT: collect, :99-132,264-307; convert the area densities next
    // to a non-connectable vertex to projected solid angle, :143-167,309-349) and forms every strategy's density by an O(k) product: O(k^2) per
    // weight, five paths per connection, the base path's arrays rebuilt for each of the four gradient weights.  Round 4 kept those arrays in the
    // lane's workspace.  Here (round 5), as in the fast form (gbdpt_kernels.hip.h, mis_sums): both weights are ratios of SUMS of strategy densities
    //     value[p] = pdfImp[1] .. pdfImp[p] * pdfRad[p + 1] .. pdfRad[k - 1],
    // and such a sum is a Horner recurrence over the entries in reverse order, R(p) = pdfRad[p + 1] R(p + 1), G(p) = a_p R(p) + pdfImp[p + 1] G(p + 1)
    // (a_p = 1 for an allowed strategy; squared entries for the power heuristic).  An entry is read from the sample's record when the recurrence
    // reaches it, with its conversion factor where the path has a specular vertex next to it; the connectable flags of the base path are two bit
    // masks per subpath made once per sample (GSamp::flags); the base path's sums are formed ONCE per connection and serve the four gradient
    // weights (sum_p (b_p gX + j gY o_p) = gX sum_p b_p + j gY sum_p o_p).  Same factors, another association: a few ulp.
    struct GConn { Float impS, impT, radS, radT; };       // pdfImp[s + 1], pdfImp[s + 2], pdfRad[s - 1], pdfRad[s]: the four densities evaluated AT the connection
    struct GMisBase { Float valueS, sum1, sum2; unsigned allowed, strict; GConn cp; };
    // (ovT: the sensor-side end point as the emitter sample a connection to the emitter supernode casts it to -- a lane-local copy, see connectPair)
    __device__ GConn connPdfs(const GPath &emitterSubpath, const BE &connectionEdge, const GPath &sensorSubpath, int s, int t, const BV *ovT)
    {
        const BV *vsPred = VN(emitterSubpath, s - 1), *vtPred = VN(sensorSubpath, t - 1); const BV &vs = V_(emitterSubpath, s), &vt = ovT ? *ovT : V_(sensorSubpath, t);
        GConn cp;
        cp.impS = bv_eval_pdf(c, vs, vsPred, &vt, EImportance, M_AREA) * connectionEdge.tr[EImportance];
        cp.impT = t > 0 ? bv_eval_pdf(c, vt, &vs, vtPred, EImportance, M_AREA) * E_(sensorSubpath, t - 1).tr[EImportance] : 0.0;
        cp.radS = s > 0 ? bv_eval_pdf(c, vs, &vt, vsPred, ERadiance, M_AREA) * E_(emitterSubpath, s - 1).tr[ERadiance] : 0.0;
        cp.radT = bv_eval_pdf(c, vt, vtPred, &vs, ERadiance, M_AREA) * connectionEdge.tr[ERadiance];
        return cp;
    }
    // path vertex i (0..k) of "emitter subpath [0..s] + sensor subpath [t..0]"
    __device__ __forceinline__ const BV &pathV(const GPath &emitterSubpath, const GPath &sensorSubpath, int s, int k, int i, const BV *ovT)
    {
        if (i <= s) return V_(emitterSubpath, i);
        return (ovT && k - i == k - s - 1 /* = t */) ? *ovT : V_(sensorSubpath, k - i);
    }
    // the factor that turns the area density across path edge (e, e + 1), e != s, into a projected-solid-angle one (path.cpp:158-166,340-348)
    __device__ Float edgeFactor(const GPath &emitterSubpath, const GPath &sensorSubpath, int s, int k, int e, const BV *ovT)
    {
        const BV &va = pathV(emitterSubpath, sensorSubpath, s, k, e, ovT), &vb = pathV(emitterSubpath, sensorSubpath, s, k, e + 1, ovT);
        const BE &edge = e < s ? E_(emitterSubpath, e) : E_(sensorSubpath, k - e - 1);
        return edge.length * edge.length / fabs((bv_on_surface(vb) ? dot(edge.d, bv_geo_normal(c, vb)) : 1) * (bv_on_surface(va) ? dot(edge.d, bv_geo_normal(c, va)) : 1));
    }
    // entry i (1 <= i <= k - 1) of pdfImp / pdfRad after the conversion loops; strict = connectableStrict[] of the BASE path as a bit mask
    __device__ __forceinline__ Float misImp(const GPath &emitterSubpath, const GPath &sensorSubpath, const GConn &cp, unsigned strict, int s, int t, int i, const BV *ovT)
    {
        const int k = s + t + 1;
        Float v;
        if (i <= s) v = V_(emitterSubpath, i - 1).pdf[EImportance] * E_(emitterSubpath, i - 1).tr[EImportance];
        else if (i == s + 1) v = cp.impS;
        else if (i == s + 2) v = cp.impT;
        else { const int q = t + s + 2 - i; v = V_(sensorSubpath, q).pdf[EImportance] * E_(sensorSubpath, q - 1).tr[EImportance]; }       // sensor vertex t - 1 .. 1
        // pdfImp[j + 1] of the loop j = 1 .. k - 3, j != s, where connectableStrict[j] && !connectableStrict[j + 1]
        const int j = i - 1;
        if (j >= 1 && j <= k - 3 && j != s && ((strict >> j) & 1u) && !((strict >> (j + 1)) & 1u)) v *= edgeFactor(emitterSubpath, sensorSubpath, s, k, j, ovT);
        return v;
    }
    __device__ __forceinline__ Float misRad(const GPath &emitterSubpath, const GPath &sensorSubpath, const GConn &cp, unsigned strict, int s, int t, int i, const BV *ovT)
    {
        const int k = s + t + 1;
        Float v;
        if (i <= s - 2) v = V_(emitterSubpath, i + 1).pdf[ERadiance] * E_(emitterSubpath, i).tr[ERadiance];
        else if (i == s - 1) v = cp.radS;
        else if (i == s) v = cp.radT;
        else { const int q = t + s + 1 - i; v = V_(sensorSubpath, q - 1).pdf[ERadiance] * E_(sensorSubpath, q - 1).tr[ERadiance]; }         // sensor vertex t .. 1
        // pdfRad[j - 1] of the loop j = k - 1 .. 3, j - 1 != s, where connectableStrict[j] && !connectableStrict[j - 1]
        const int j = i + 1;
        if (j >= 3 && j <= k - 1 && i != s && ((strict >> j) & 1u) && !((strict >> i) & 1u)) v *= edgeFactor(emitterSubpath, sensorSubpath, s, k, i, ovT);
        return v;
    }
    template <bool SQUARES>
    __device__ void misSums(const GPath &emitterSubpath, const GPath &sensorSubpath, const GConn &cp, unsigned allowed, unsigned strict, int s, int t, const BV *ovT, Float &valueS, Float &sum1, Float &sum2)
    {
        const int n = s + t + 1;
        Float R = 1.0, G = (allowed >> (n - 1)) & 1u ? 1.0 : 0.0, R2 = 1.0, G2 = G, vS = 1.0;
        for (int q = n - 2; q >= 0; --q) {
            const int i = q + 1;
            const Float im = misImp(emitterSubpath, sensorSubpath, cp, strict, s, t, i, ovT), ra = misRad(emitterSubpath, sensorSubpath, cp, strict, s, t, i, ovT);
            vS *= i <= s ? im : ra;
            R *= ra;
            const bool a = (allowed >> q) & 1u;
            G = (a ? R : 0.0) + im * G;
            if (SQUARES) { R2 *= ra * ra; G2 = (a ? R2 : 0.0) + (im * im) * G2; }
        }
        valueS = vS; sum1 = G; sum2 = G2;
    }
    // the two flag masks of the base path "emitter [0..s] + sensor[0] [t..0]" (path.cpp:76-95,241-260) from the sample's per-subpath masks
    __device__ void pathFlags(int s, int t, const BV *ovT, unsigned &conn, unsigned &strict)
    {
        const int k = s + t + 1;
        conn = W.connE & ((2u << s) - 1u); strict = W.strictE & ((2u << s) - 1u);
        for (int p = s + 1; p <= k; ++p) {
            const int q = k - p;
            bool cg, cs;
            if (ovT && q == t) { cg = connectable_gbdpt(c, *ovT); cs = bv_connectable(*ovT); }
            else { cg = (W.connS >> q) & 1u; cs = (W.strictS >> q) & 1u; }
            conn |= (cg ? 1u : 0u) << p; strict |= (cs ? 1u : 0u) << p;
        }
    }
    // Path::miWeightBaseNoSweep_GBDPT (path.cpp:49-201): (p_st geomTermX)^2 / sum over the allowed strategies of (p_i geomTermX)^2 -- geomTermX cancels
    __device__ Float miWeightBase(const GPath &emitterSubpath, const BE &connectionEdge, const GPath &sensorSubpath, int s, int t, bool lightImage, const BV *ovT, GMisBase &mb)
    {
        const int k = s + t + 1;
        unsigned conn;
        pathFlags(s, t, ovT, conn, mb.strict);
        mb.allowed = conn & (conn >> 1);                                                     // connectable[p] && connectable[p + 1]
        if (!lightImage) mb.allowed &= k >= 2 ? ((1u << (k - 2)) - 1u) : 0u;                 // tPrime = k - p - 1 > 1
        mb.allowed &= (1u << k) - 1u;
        mb.cp = connPdfs(emitterSubpath, connectionEdge, sensorSubpath, s, t, ovT);
        misSums<true>(emitterSubpath, sensorSubpath, mb.cp, mb.allowed, mb.strict, s, t, ovT, mb.valueS, mb.sum1, mb.sum2);
        return (Float)((mb.valueS * mb.valueS) / mb.sum2);
    }
    // Path::miWeightGradNoSweep_GBDPT (path.cpp:204-378): p_st geomTermX / sum over the allowed strategies of (value_i geomTermX + oValue_i jDet geomTermY);
    // sharedEnds: the offset path's end points, their predecessors and the connection edge ARE the base path's records (a connection beyond the
    // shifted part): its four connection densities are the base path's, no BSDF is evaluated
    __device__ Float miWeightGrad(const GMisBase &mb, const GPath &offsetEmitterSubpath, const BE &offsetConnectionEdge, const GPath &offsetSensorSubpath,
                                  int s, int t, Float jDet, Float geomTermX, Float geomTermY, const BV *oOvT, bool sharedEnds)
    {
        const GConn cp = sharedEnds ? mb.cp : connPdfs(offsetEmitterSubpath, offsetConnectionEdge, offsetSensorSubpath, s, t, oOvT);
        Float oS, o1, o2;
        misSums<false>(offsetEmitterSubpath, offsetSensorSubpath, cp, mb.allowed, mb.strict, s, t, oOvT, oS, o1, o2);
        return (Float)((mb.valueS * geomTermX) / (mb.sum1 * geomTermX + o1 * jDet * geomTermY));
    }

    // ---- GBDPTRenderer, gbdpt_proc.cpp ----
    __device__ bool createShiftablePath(GPath &connectedPath, GPath &emitterSubpath, GPath &sensorSubpath, int s, int t, int &memPointer, bool knownVisible = false)   // :600-662
    {
        connectedPath.clear();
        while (!connectable_gbdpt(c, V_(sensorSubpath, t))) { t--; sensorSubpath.nv--; sensorSubpath.ne--; }
        if (V_(sensorSubpath, t).type == T_SURFACE && prim_emitter(c, V_(sensorSubpath, t).prim) >= 0) s = 0;
        for (memPointer = 0; memPointer < s; memPointer++) { connectedPath.pushV(emitterSubpath.v[memPointer]); connectedPath.pushE(emitterSubpath.e[memPointer]); }
        connectedPath.pushV(cloneV(emitterSubpath.v[memPointer]));
        connectedPath.pushE(allocE());
        connectedPath.pushV(cloneV(sensorSubpath.v[t]));
        bv_cast_emitter(c, V_(connectedPath, memPointer + 1));
        for (int i = t - 1; i >= 0; i--) { connectedPath.pushV(sensorSubpath.v[i]); connectedPath.pushE(sensorSubpath.e[i]); }
        const bool pathSuccess = gConnect(VN(connectedPath, memPointer - 1), V_(connectedPath, memPointer), E_(connectedPath, memPointer), V_(connectedPath, memPointer + 1),
                                          VN(connectedPath, memPointer + 2),
                                          bv_connectable(V_(connectedPath, memPointer)) ? M_AREA : M_DISCRETE, bv_connectable(V_(connectedPath, memPointer + 1)) ? M_AREA : M_DISCRETE, knownVisible);
        if (t == 1) { BV &s1 = V_(connectedPath, connectedPath.nv - 2); sensor_sample_position(c, s1.p, V_(connectedPath, connectedPath.nv - 3).p - s1.p, s1.u, s1.v); }
        return pathSuccess;
    }

    // GBDPTRenderer::process from the connected base path on (gbdpt_proc.cpp:186-229): createShiftablePath, the four offset paths with their
    // Jacobians and generalized geometry terms, the prefix products of combineImportanceData / combineRadianceData (:544-565).  Everything a
    // connection reads is in W afterwards, and W is not written again.  The two subpaths are W.emitter / W.sensor[0] (loadSubpaths).
    // Two stages, so that the frame kernels can run them on different lanes: prepareBase (one lane per sample) and prepareOffset(k) (one lane per
    // sample AND offset path: the four offset paths of a sample depend on its connected base path only, each writes its own slots and pool slice).
    __device__ void prepareBase()
    {
        GPath &emitterSubpath = W.emitter;
        for (int k = 0; k < 5; k++) {
            W.success[k] = 0; W.couldConnectAfterB[k] = 0;
            for (int i = 0; i < NSV + 4; i++) { W.jacobianDet[k][i] = 1.0; W.genGeomTerm[k][i] = 1.0; }
            W.mu[k].l = W.mu[k].m = 0; for (int i = 0; i < 5; i++) W.mu[k].extra[i] = 0;
        }
        W.success[0] = 1; W.couldConnectAfterB[0] = 1;
        W.voidSample = 0;
        GPath &connectPath = W.connect;
        int ptx = 0;
        createShiftablePath(connectPath, emitterSubpath, W.sensor[0], 1, W.sensor[0].nv - 1, ptx);
        computeMuRec(connectPath, W.mu[0]);
        GCache gc;
        for (int v = W.mu[0].extra[0] - 1; v >= 0; v--) {
            const int idx = connectPath.nv - 1 - v;
            if (connectable_gbdpt(c, V_(connectPath, v)) && v >= W.mu[0].extra[2]) W.genGeomTerm[0][idx] = calcSpecularPDFChange(connectPath, v, false, &gc);
            else W.genGeomTerm[0][idx] = W.genGeomTerm[0][idx - 1];
        }
        W.nvBase = W.nv; W.neBase = W.ne;
        W.vert_b = connectPath.nv - 1 - W.mu[0].extra[1];
        const int nE = emitterSubpath.nv, nS = W.sensor[0].nv;
        W.connE = W.strictE = W.connS = W.strictS = 0u;
        for (int i = 0; i < nE; ++i) { const BV &v = V_(emitterSubpath, i); W.connE |= (connectable_gbdpt(c, v) ? 1u : 0u) << i; W.strictE |= (bv_connectable(v) ? 1u : 0u) << i; }
        for (int i = 0; i < nS; ++i) { const BV &v = V_(W.sensor[0], i); W.connS |= (connectable_gbdpt(c, v) ? 1u : 0u) << i; W.strictS |= (bv_connectable(v) ? 1u : 0u) << i; }
        W.impW[0] = mk(1.0); W.impP[0] = 1.0;
        for (int i = 1; i < nE; ++i) {
            const BV &pv = V_(emitterSubpath, i - 1); const BE &pe = E_(emitterSubpath, i - 1);
            W.impW[i] = W.impW[i - 1] * pv.w[EImportance] * pv.rr * pe.tr[EImportance];
            W.impP[i] = W.impP[i - 1] * pv.pdf[EImportance] * pv.rr * pe.tr[EImportance];
        }
        for (int k = 0; k <= 4; k++) { W.radW[k][0] = mk(1.0); W.radP[k][0] = 1.0; for (int i = 1; i < nS; ++i) { W.radW[k][i] = mk(0.0); W.radP[k][i] = 0.0; } }
        for (int k = 1; k <= 4; k++) W.sensor[k].clear();
        radianceProducts(0);
    }
    __device__ void radianceProducts(int k)
    {
        const int nS = W.sensor[0].nv;
        for (int i = 1; i < nS; ++i) {
            W.radW[k][i] = mk(0.0); W.radP[k][i] = 0.0;
            if (W.success[k] && i < W.sensor[k].nv) {
                const BV &pv = V_(W.sensor[k], i - 1); const BE &pe = E_(W.sensor[k], i - 1);
                W.radW[k][i] = W.radW[k][i - 1] * pv.w[ERadiance] * pv.rr * pe.tr[ERadiance];
                W.radP[k][i] = W.radP[k][i - 1] * pv.pdf[ERadiance] * pv.rr * pe.tr[ERadiance];
            }
        }
    }
    __device__ bool hasOffsets() const { return W.mu[0].extra[0] > 2; }                      // gbdpt_proc.cpp:200
    __device__ bool offsetsWalk() const { return abs(W.mu[0].extra[1] - W.mu[0].extra[2]) > 1; }   // (b and c are not adjacent: generateOffsetPath enters manifoldWalk, mut_manifold.cpp:882)
    __device__ void prepareOffset(int k)                                                     // k = 0..3: the offset path sensor[k + 1]
    {
        const Float shifts[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};
        GPath &connectPath = W.connect;
        GPath &off = W.sensor[k + 1];
        off.clear();
        GPROF(0, W.success[k + 1] = !hasOffsets() ? 0 : (generateOffsetPath(connectPath, off, W.mu[k + 1], shifts[k][0], shifts[k][1], W.couldConnectAfterB[k + 1], false) ? 1 : 0));
        if (W.success[k + 1]) {
            int lastB = -1, lastC = -1;
            double lastJ = 1.0;
            GCache gc;
            for (int v = W.mu[k + 1].extra[0] - 1; v >= 0; v--) {
                const int idx = connectPath.nv - 1 - v;
                if (connectable_gbdpt(c, V_(connectPath, v)) && v >= W.mu[k + 1].extra[2]) {
                    const int a = W.mu[k + 1].extra[0];
                    const int b = v >= W.mu[k + 1].extra[1] ? v : W.mu[k + 1].extra[1];
                    const int cI = v >= W.mu[k + 1].extra[1] ? v - 1 : W.mu[k + 1].extra[2];
                    // (below b the triple (a, b, c) no longer changes with v: the same two half-Jacobians -- 22 % of this stage's clocks when they were
                    //  computed for every v; same inputs, same quotient, bit for bit)
                    if (b != lastB || cI != lastC) {
                        double jx, jy;
                        GPROF(2, jx = halfJacobian(connectPath, a, b, cI); jy = halfJacobian(off, a, b, cI));
                        lastJ = jy / jx; lastB = b; lastC = cI;
                    }
                    W.jacobianDet[k + 1][idx] = lastJ;
                    GPROF(3, W.genGeomTerm[k + 1][idx] = calcSpecularPDFChange(off, v, false, &gc));
                } else {
                    W.jacobianDet[k + 1][idx] = W.jacobianDet[k + 1][idx - 1];
                    W.genGeomTerm[k + 1][idx] = W.genGeomTerm[k + 1][idx - 1];
                }
            }
        }
        off.reverse();
        GPROF(4, radianceProducts(k + 1));
    }
    __device__ void prepare()
    {
        prepareBase();
        for (int k = 0; k < 4; k++) prepareOffset(k);
    }
    // the connections of emitter vertex s (gbdpt_proc.cpp:311-319; the sensor subpath may have lost trailing non-connectable vertices in createShiftablePath)
    __device__ __forceinline__ void pairRange(int s, int &minT, int &maxT) const
    {
        minT = max(2 - s, c.cfg.lightImage ? 1 : 2);
        maxT = min(W.sensor[0].nv - 1, c.cfg.maxDepth + 1 - s);
    }

    // ONE connection (s, t) of GBDPTRenderer::evaluate (gbdpt_proc.cpp:319-527): the base path and its four offsets.  Reads W, writes nothing
    // of it: the reference casts the sensor-side end point to an emitter sample IN PLACE at s = 0 (:402) and sets the measure of connected end points
    // (:439) on vertices the five paths share -- no later connection reads either (vertex t is not on the path of a smaller t, s = 0 is the last s;
    // a measure that is not EDiscrete stays so), so a connection works on a local copy of the cast vertex and leaves the measures alone, and
    // connections may run side by side in any order.  T1: a light-tracing connection (t == 1): its own shiftable path, four offset paths with
    // their own manifold walks, all in the lane's transient pool (X); t >= 2 needs no X.  Returns false when the connection contributes nothing.
    // PHASE (as the fast form's connect_pair; light tracing's phase 3 is the test whether the sensor sees the emitter vertex at all): 0 = the whole connection (the probe entry); 3 = the part of the base path that
    // needs no visibility ray (end points connectable, facing each other, throughput): a filter in front of 1 = the base path only: visibility,
    // geometry term, MIS weight -> whether it carries anything and its primal term (most connections end here, and a wave in which one lane goes
    // on to the four offsets while the others wait runs at a fifth of its lanes); 2 = the four offsets of a survivor of phase 1: the base path is
    // evaluated again for the state the offsets share with it (its ray is not counted twice), the gradients are the output.
    template <bool T1, int PHASE = 0>
    __device__ bool connectPair(int s, int t, PairOut &po)
    {
        const unsigned nClosest0 = c.nClosest, nShadow0 = c.nShadow;
        const Float shifts[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};
        const BdConfig &cfg = c.cfg;
        GPath &emitterSubpath = W.emitter;
        const int vert_b = W.vert_b;
        po.nLight = 0;
        Float samplePosX = V_(W.sensor[0], 1).u, samplePosY = V_(W.sensor[0], 1).v;
        if constexpr (T1) {
            const BV &v1 = V_(W.sensor[0], 1);
            if ((v1.type == T_SENSOR_SAMPLE && !sensor_sample_position(c, v1.p, V_(emitterSubpath, s).p - v1.p, samplePosX, samplePosY)) || !connectable_gbdpt(c, V_(emitterSubpath, s))) return false;
            if (PHASE == 3) return true;                                                    // (light tracing's ray-free filter: the emitter vertex is connectable and the sensor sees it)
            X->nlv = X->nle = 0;
            localAlloc = true;
        }
        Float geomTermBase = 0.0; d3 connectionPartsBase = mk(0.0), offsetImportanceWeight = mk(0.0);
        BE connectionEdge, connectionEdgeBase;
        be_clear(connectionEdge); be_clear(connectionEdgeBase);
        bool successConnectBase = false;
        Float offsetImportancePdf = 0;
        d3 value[5]; Float miWeight[5], valuePdf[5];
        double jacobianLP[4] = {1.0, 1.0, 1.0, 1.0}, genGeomTermLP[5] = {1.0, 1.0, 1.0, 1.0, 1.0};
        bool pathSuccess[5];
        int memPointer = 0;
        MuRec muRec; muRec.l = muRec.m = 0; for (int i = 0; i < 5; i++) muRec.extra[i] = 0;
        GMisBase misBase;                                                                    // the base path's strategy sums of this connection (k = 0), reused by k = 1..4
        BV vtBaseCast, vtCast;                                                               // s == 0: the sensor-side end point as the emitter sample it is cast to (base path / path k)
        int markV = 0, markE = 0;
        for (int k = 0; k <= ((PHASE == 1 || PHASE == 3) ? 0 : 4); k++) {
            if (T1 && PHASE == 2 && lightK && k != 0 && k != lightK) continue;              // (this lane builds ONE of the light path's four offsets; the other lanes the others)
            miWeight[k] = 1.0 / (s + t + 1);
            pathSuccess[k] = W.success[k] != 0;
            value[k] = mk(0.0);
            valuePdf[k] = 0.0;
            d3 importanceWeightTmp = W.impW[s], radianceWeightTmp = W.radW[T1 ? 0 : k][t];
            Float importancePdfTmp = W.impP[s], radiancePdfTmp = W.radP[T1 ? 0 : k][t];
            const GPath *sensorSubpathTmp = &W.sensor[k], *emitterSubpathTmp = &emitterSubpath;
            if constexpr (T1) if (k == 0) {
                // (phase 2 runs on the survivors of phase 1: the base path's sensor connection and -- below -- its connection edge were traced there and found free)
                pathSuccess[0] = createShiftablePath(X->connectedBase, emitterSubpath, W.sensor[0], s, 1, memPointer, PHASE == 2);
                computeMuRec(X->connectedBase, muRec);
                lightWalks = abs(muRec.extra[1] - muRec.extra[2]) > 1;                       // (its offset paths enter a manifold walk: b and c are not adjacent, mut_manifold.cpp:882)
                genGeomTermLP[0] = calcSpecularPDFChange(X->connectedBase, muRec.extra[2], true);
                markV = X->nlv; markE = X->nle;
            }
            if constexpr (T1) if (k > 0 && !is_zero(value[0])) {
                if (!pathSuccess[0]) pathSuccess[k] = false;
                else {
                    // createShiftedLightPath, :568-590 (the records of the previous offset are dead: its values are in value / valuePdf / miWeight)
                    X->nlv = markV; X->nle = markE;
                    jacobianLP[k - 1] = 1.0;
                    int couldConnectWithB = 0;
                    pathSuccess[k] = generateOffsetPath(X->connectedBase, X->offsetEmitter, muRec, shifts[k - 1][0], shifts[k - 1][1], couldConnectWithB, true);
                    if (pathSuccess[k]) {
                        jacobianLP[k - 1] = halfJacobian(X->offsetEmitter, muRec.extra[0], muRec.extra[1], muRec.extra[2]) / halfJacobian(X->connectedBase, muRec.extra[0], muRec.extra[1], muRec.extra[2]);
                        offsetImportancePdf = 1.0;
                        offsetImportanceWeight = mk(1.0);
                        for (int i = 1; i <= s; ++i) {
                            const BV &pv = V_(X->offsetEmitter, i - 1); const BE &pe = E_(X->offsetEmitter, i - 1);
                            offsetImportanceWeight = offsetImportanceWeight * pv.w[EImportance] * pv.rr * pe.tr[EImportance];
                            offsetImportancePdf = offsetImportancePdf * pv.pdf[EImportance] * pv.rr * pe.tr[EImportance];
                        }
                        genGeomTermLP[k] = calcSpecularPDFChange(X->offsetEmitter, muRec.extra[2], true);
                        importanceWeightTmp = offsetImportanceWeight;
                        importancePdfTmp = offsetImportancePdf;
                        emitterSubpathTmp = &X->offsetEmitter;
                    }
                    sensorSubpathTmp = &W.sensor[0];
                }
            }
            Float geomTerm = 0.0;
            do {
                if (!(pathSuccess[k] && pathSuccess[0] && (k == 0 || (valuePdf[0] > 0 && !is_zero(value[0]))))) break;
                if (!W.couldConnectAfterB[k] && t > vert_b) break;
                const BV *vsPred = VN(*emitterSubpathTmp, s - 1), *vtPred = VN(*sensorSubpathTmp, t - 1);
                const BV &vs = PV(emitterSubpathTmp->v[s]);
                const BV *vtP = &PV(sensorSubpathTmp->v[t]);
                if (vs.type == T_EMITTER_SUPER) {
                    vtCast = *vtP;
                    if (!bv_cast_emitter(c, vtCast) || vtCast.degenerate) { valuePdf[k] = radiancePdfTmp; break; }
                    vtP = &vtCast;
                    if (k == 0) vtBaseCast = vtCast;
                    const d3 connectionParts = (k > 0 && t > vert_b + 1) ? connectionPartsBase : gEval(vs, vsPred, vtP, EImportance) * gEval(*vtP, vtPred, &vs, ERadiance);
                    if (k == 0) connectionPartsBase = connectionParts;
                    value[k] = radianceWeightTmp * connectionParts;
                    valuePdf[k] = radiancePdfTmp;
                } else if (vtP->type == T_SENSOR_SUPER) { valuePdf[k] = importancePdfTmp; break; }
                else {
                    if (!connectable_gbdpt(c, vs) || !connectable_gbdpt(c, *vtP) || vs.type == 0 || vtP->type == 0) { valuePdf[k] = importancePdfTmp * radiancePdfTmp; break; }
                    const d3 connectionParts = (k > 0 && t > vert_b + 1) ? connectionPartsBase : gEval(vs, vsPred, vtP, EImportance) * gEval(*vtP, vtPred, &vs, ERadiance);
                    if (k == 0) connectionPartsBase = connectionParts;
                    value[k] = importanceWeightTmp * radianceWeightTmp * connectionParts;
                    valuePdf[k] = importancePdfTmp * radiancePdfTmp;
                }
                if (is_zero(value[k]) || valuePdf[k] == 0) break;
                if (PHASE == 3) return true;                                                // (k == 0: both end points face each other and carry throughput -- worth a visibility ray)
                const bool successConnect = (k > 0 && t > vert_b) ? successConnectBase : edge_path_connect(c, connectionEdge, vs, *vtP, PHASE == 2 && k == 0);
                if (k == 0) successConnectBase = successConnect;
                if (!successConnect) { value[k] = mk(0.0); break; }
                geomTerm = (k > 0 && t > vert_b) ? geomTermBase : gEdgeEvalCached(connectionEdge, vs, *vtP, 0x04 | 0x08 | 0x10 | 0x20);
                value[k] = value[k] * geomTerm;
                valuePdf[k] *= (t < 2 ? genGeomTermLP[k] : W.genGeomTerm[k][t]);
                if (is_zero(value[k]) || valuePdf[k] == 0) break;
                const BV *ovBase = vs.type == T_EMITTER_SUPER ? &vtBaseCast : nullptr;      // (the cast vertex: miWeight sees the emitter sample, gbdpt_proc.cpp:402)
                if (k == 0) {
                    connectionEdgeBase = connectionEdge;
                    geomTermBase = geomTerm;
                    miWeight[0] = miWeightBase(emitterSubpath, connectionEdgeBase, W.sensor[0], s, t, cfg.lightImage != 0, ovBase, misBase) / valuePdf[0];
                } else {
                    // (a path k whose end points, their predecessors and -- t > vert_b -- connection edge are the base path's own records)
                    const bool sharedEnds = !T1 && t > vert_b && t >= 1 && sensorSubpathTmp->v[t] == W.sensor[0].v[t] && sensorSubpathTmp->v[t - 1] == W.sensor[0].v[t - 1];
                    miWeight[k] = miWeightGrad(misBase, *emitterSubpathTmp, connectionEdge, *sensorSubpathTmp, s, t,
                                               (t < 2 ? jacobianLP[k - 1] : W.jacobianDet[k][t]), (t < 2 ? genGeomTermLP[0] : W.genGeomTerm[0][t]), (t < 2 ? genGeomTermLP[k] : W.genGeomTerm[k][t]),
                                               vs.type == T_EMITTER_SUPER ? vtP : nullptr, sharedEnds) / valuePdf[0];
                }
            } while (false);
#ifdef GDPT_BD_TRACE
            printf("G st %d %d k %d ok %d value %.17g %.17g %.17g pdf %.17g miW %.17g geom %.17g rays %u %u\n", s, t, k, (int)pathSuccess[k], value[k].x, value[k].y, value[k].z, valuePdf[k], miWeight[k], geomTerm, c.nClosest, c.nShadow);
#endif
            if (is_zero(value[k]) || is_zero(value[0])) { value[k] = mk(0.0); miWeight[k] = miWeight[0]; valuePdf[k] = valuePdf[0]; }
            if (PHASE == 2 && k == 0) { c.nClosest = nClosest0; c.nShadow = nShadow0; }     // (counted by phase 1)
        }
        if constexpr (T1) localAlloc = false;
        if (PHASE == 3 || is_zero(value[0])) return false;
        const d3 mainRad = value[0] * (valuePdf[0] * miWeight[0]);
        po.primal = mainRad;
        if (T1 && PHASE != 2) { LightSplat &ls = po.light[po.nLight++]; ls.x = samplePosX; ls.y = samplePosY; ls.buffer = 0; ls.value = mainRad; }
        if (PHASE == 1) return true;
        const d3 fx = value[0] * valuePdf[0];
        for (int n = 0; n < 4; n++) {
            if (T1 && PHASE == 2 && lightK && n + 1 != lightK) continue;
            const d3 fy = value[n + 1] * valuePdf[n + 1] * (Float)(t < 2 ? jacobianLP[n] : W.jacobianDet[n + 1][t]);
            const d3 gradVal = (fy - fx) * ((Float)2.0 * miWeight[n + 1]);
            po.gradient[n] = gradVal;
            if (T1) { LightSplat &ls = po.light[po.nLight++]; ls.x = samplePosX; ls.y = samplePosY; ls.buffer = n + 1; ls.value = gradVal; }
        }
        return true;
    }

    // GBDPTRenderer::evaluate (:259-534) in ONE lane, connections in the reference's order: the probe entry (the frame kernels run prepare and
    // connectPair as separate launches: k_bdg_shift, k_bdg_connect, k_bdg_light)
    __device__ void evaluate(SampleOut &wr)
    {
        wr.posX = V_(W.sensor[0], 1).u; wr.posY = V_(W.sensor[0], 1).v;
        wr.nLight = 0;
        wr.primal = mk(0.0);
        for (int k = 0; k < 4; ++k) wr.gradient[k] = mk(0.0);
        PairOut po;
        for (int s = W.emitter.nv - 1; s >= 0; --s) {
            int minT, maxT;
            pairRange(s, minT, maxT);
            for (int t = maxT; t >= minT; --t) {
                if (!(t == 1 ? connectPair<true>(s, t, po) : connectPair<false>(s, t, po))) continue;
                if (t >= 2) { wr.primal = wr.primal + po.primal; for (int n = 0; n < 4; n++) wr.gradient[n] = wr.gradient[n] + po.gradient[n]; }
                for (int i = 0; i < po.nLight && wr.nLight < BD_MAX_LIGHT; i++) wr.light[wr.nLight++] = po.light[i];
            }
        }
    }
    __device__ void processSample(SampleOut &wr) { prepare(); evaluate(wr); }

    // the two subpaths of a walked sample -> pool records and index lists
    __device__ void loadSubpaths(const Sample &sm)
    {
        W.nv = W.ne = 0; overflow = 0;
        W.emitter.clear(); W.sensor[0].clear();
        for (int i = 0; i < sm.nY; i++) { const int j = allocV(); W.v[j] = sm.Y[i]; W.emitter.pushV(j); if (i + 1 < sm.nY) { const int e = allocE(); W.e[e] = sm.EY[i]; W.emitter.pushE(e); } }
        for (int i = 0; i < sm.nX; i++) { const int j = allocV(); W.v[j] = sm.X[i]; W.sensor[0].pushV(j); if (i + 1 < sm.nX) { const int e = allocE(); W.e[e] = sm.EX[i]; W.sensor[0].pushE(e); } }
    }
};
using GTr = GTrT<true>;

// does the sample need the general form?  (a surface vertex of either subpath that is not connectable in the sense of Path::isConnectable_GBDPT)
__device__ bool sample_needs_general(const Ctx &c, const Sample &sm)
{
    for (int i = 2; i < sm.nX; i++) if (sm.X[i].type == T_SURFACE && !connectable_gbdpt(c, sm.X[i])) return true;
    for (int i = 2; i < sm.nY; i++) if (sm.Y[i].type == T_SURFACE && !connectable_gbdpt(c, sm.Y[i])) return true;
    return false;
}

} // namespace gdpt_bd
