// gpt_render.hip.h -- the render / resolve / develop kernels of the gfx950 G-PT sampler.
// See gpt_kernels.hip.h for the design notes and the device-side restatements these kernels call.
#pragma once
#include "gpt_kernels.hip.h"

#include <type_traits>

namespace gdpt_tr {

// The per-offset parts of a bounce are written once, as a generic lambda over (index, offset), and expanded by for_offsets:
//  * UNROLL (the 2-wave builds): four instantiations with a compile-time index -- off[i] and the sums' slots are plain scalars, the
//    Lane is register-allocated.  `#pragma unroll` is NOT what does this: the shift code is past LLVM's -pragma-unroll-threshold
//    (16384 IR instructions x 4), the pragma was silently ignored, and a run-time index into a member array stops SROA for the whole
//    struct -- the 2-wave builds ran with the entire Lane in scratch until this was found (config 2: 7.4 -> 8.5 Gray/s).
//  * rolled (the 4-wave builds, 128 VGPRs): ONE copy of the shift code, off[] indexed at run time, i.e. the Lane and register-held
//    sums live in scratch BY DESIGN -- measured on the atrium against the two alternatives: unrolled 1.81 Gray/s, rolled over a copy
//    picked by a uniform switch (Lane in registers, spilled by the allocator instead) 1.76, this 2.33.
#ifdef GDPT_FORCE_ROLLED      /* investigation build: the 2-wave kernels with the rolled loop, i.e. the configuration of "the lean-build fault"  */
#define GDPT_UNROLL_OFFSETS(WPS) false
#else
#define GDPT_UNROLL_OFFSETS(WPS) ((WPS) <= 2)
#endif
// queue records are written once and read a whole chunk of samples later (14 GB per 32-sample chunk at 1280x720): streaming stores keep them
// from evicting what the render kernels do re-use from L2 (their scratch, the scene tables of the HBM-resident builds)
__device__ __forceinline__ void qst(Float *p, Float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ Float qld(const Float *p) { return __builtin_nontemporal_load(p); }
// pixel shift of offset path i (gpt.cpp:410-415: right, down, left, up)
__device__ __forceinline__ Float offset_shift_x(int i) { return i == 0 ? 1.0 : (i == 2 ? -1.0 : 0.0); }
__device__ __forceinline__ Float offset_shift_y(int i) { return i == 1 ? 1.0 : (i == 3 ? -1.0 : 0.0); }
template <bool UNROLL, class BODY>
__device__ __forceinline__ void for_offsets(Offset (&off)[4], BODY &&body)
{
    if constexpr (UNROLL) {
        body(std::integral_constant<int, 0>{}, off[0]); body(std::integral_constant<int, 1>{}, off[1]);
        body(std::integral_constant<int, 2>{}, off[2]); body(std::integral_constant<int, 3>{}, off[3]);
    } else {
#ifdef GDPT_FORCE_ROLLED
#pragma unroll          /* (as the loop was written when the fault was seen: a full-unroll request that LLVM drops for its size) */
#else
#pragma unroll 1
#endif
        for (int i = 0; i < 4; i++) body(i, off[i]);
    }
}
// the three sums an offset adds to per bounce (gpt.cpp:723-726, 1140-1146)
template <class ACC>
__device__ __forceinline__ void add_offset_sums(ACC &A, int i, d3 t, d3 n, d3 g)
{
    A.add3(ACC_T, t);
    A.add3(ACC_NBR + 3 * i, n);
    A.add3(ACC_GRAD + 3 * i, g);
}

// ---- the offsets of a sample once they are JOINED to the base path (RAY_CONNECTED / RAY_RECENTLY_CONNECTED / dead): what a bounce does for them is a pure function of
// their four (throughput, pdf) pairs and a handful of the base path's values of that bounce (gpt.cpp:622-658 for the emitter sample, :844-888 for the BSDF sample).  The
// two structs are those values; the two functions are that arithmetic for the DEFERRED form (k_walk + k_replay, below): the walker logs the structs per bounce, the replay
// applies them -- the same doubles and the same operations in the same order as the RAY_CONNECTED / RAY_RECENTLY_CONNECTED branches of bounce(), which k_continue and the
// general kernel keep in place (measured: routing them through these functions cost the in-place kernels 1.5-3 %).  That the two copies agree to the last bit is what
// test_every_shipped_instantiation... and test_staged_pipeline... hold: the staged pipeline (deferred) against the single general kernel (in place) and the oracle.
struct NeeShared {              // the emitter-sample half of a bounce, :565-730
    Float dRecPdf, mainBsdfPdf, num, den;       // dRec.pdf, mainBsdfPdf, mainWeightNumerator, mainWeightDenominator
    d3 X, contribAll;                           // mainBSDFValue * mainEmitterRadiance; mainContributionAll
    // read by a RAY_RECENTLY_CONNECTED offset only (its re-evaluation of the base vertex's BSDF, :638-658):
    d3 radiance, woL;                           // mainEmitterRadiance; toLocal(mfr, dRec.d)
    int visSA;                                  // lightOnSurfaceSA && mainEmitterVisible
};
struct BsdfShared {             // the BSDF-sample half of a bounce, :737-1151
    d3 W;                                       // bs.weight * bs.pdf
    Float mainBsdfPdf, lumPdf, num, den;        // bs.pdf, mainLumPdf, mainWeightNumerator, mainWeightDenominator
    d3 radiance, contrib;                       // mainEmitterRadiance of the new vertex; mainContribution
    d3 woL; int measure;                        // RAY_RECENTLY_CONNECTED only: toLocal(mfr, L.rayD), the sampled component's measure (:862-888)
};
struct BaseVertexRef {          // the base vertex a RAY_RECENTLY_CONNECTED offset re-evaluates: previousMainIts
    const MaterialD *bsdf; d3 R; Frame3 fr; d3 p;
};
// What a RAY_RECENTLY_CONNECTED offset's re-evaluation of the base vertex's BSDF yields -- the only part of a joined offset's bounce that touches the scene
// (:638-658 for the emitter sample, :862-888 for the BSDF sample): k_continue forms it where it is used, k_walk forms it for the four offsets of a sample's first
// bounce after the hand-over and logs it, so that k_replay is arithmetic only.
struct RecentTerm { d3 f; Float pdf; };         // emitter-sample half: f * mainEmitterRadiance, shiftedBsdfPdf; BSDF-sample half: f, shiftedBsdfPdf
__device__ __forceinline__ RecentTerm recent_nee(const BaseVertexRef &b, d3 recentVertex, const NeeShared &n)
{
    const d3 incoming = normalize(recentVertex - b.p);
    d3 f;
    Float pdfRaw;
    bsdf_eval_pdf(*b.bsdf, b.R, toLocal(b.fr, incoming), n.woL, MEASURE_SOLID_ANGLE, f, pdfRaw);
    RecentTerm r;
    r.pdf = n.visSA ? pdfRaw : 0;
    r.f = f * n.radiance;
    return r;
}
__device__ __forceinline__ RecentTerm recent_bsdf(const BaseVertexRef &b, d3 recentVertex, const BsdfShared &m)
{
    const d3 incoming = normalize(recentVertex - b.p);
    RecentTerm r;
    bsdf_eval_pdf(*b.bsdf, b.R, toLocal(b.fr, incoming), m.woL, m.measure, r.f, r.pdf);
    return r;
}
template <class ACC>
__device__ __forceinline__ void nee_offset_joined(const NeeShared &n, const RecentTerm &rt, int status, const Offset &s, int i, ACC &A)
{
    d3 shiftedContribution = mk(0.0);
    Float weight = 0;
    if (s.alive) {
        if (status == RAY_CONNECTED) {                                           // :622-637
            const Float den = (s.pdf * s.pdf) * ((n.dRecPdf * n.dRecPdf) + (n.mainBsdfPdf * n.mainBsdfPdf));
            weight = n.num / (GD_D_EPSILON + den + n.den);
            shiftedContribution = 1.0 * s.throughput * n.X;
        } else {                                                                 // RAY_RECENTLY_CONNECTED, :638-658
            const Float shiftedBsdfPdf = rt.pdf;
            const Float den = (s.pdf * s.pdf) * ((n.dRecPdf * n.dRecPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
            weight = n.num / (GD_D_EPSILON + den + n.den);
            shiftedContribution = 1.0 * s.throughput * rt.f;
        }
    } else {                                                                     // :708-717
        weight = n.num / (GD_D_EPSILON + n.den);
        shiftedContribution = mk(0.0);
    }
    add_offset_sums(A, i, n.contribAll * weight, shiftedContribution * weight, (shiftedContribution - n.contribAll) * weight);   // :723-726
}
template <class ACC>
__device__ __forceinline__ void bsdf_offset_joined(const BsdfShared &m, const RecentTerm &rt, Offset &s, int i, ACC &A)
{
    d3 shiftedContribution = mk(0.0);
    Float weight = 0;
    if (s.alive) {
        const Float shiftedPreviousPdf = s.pdf;
        if (s.status == RAY_CONNECTED) {                                         // :844-861
            s.throughput = s.throughput * m.W;
            s.pdf *= m.mainBsdfPdf;
            const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((m.lumPdf * m.lumPdf) + (m.mainBsdfPdf * m.mainBsdfPdf));
            weight = m.num / (GD_D_EPSILON + den + m.den);
            shiftedContribution = s.throughput * m.radiance;
        } else {                                                                 // RAY_RECENTLY_CONNECTED, :862-888
            const Float shiftedBsdfPdf = rt.pdf;
            s.throughput = s.throughput * rt.f;
            s.pdf *= shiftedBsdfPdf;
            s.status = RAY_CONNECTED;
            const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((m.lumPdf * m.lumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
            weight = m.num / (GD_D_EPSILON + den + m.den);
            shiftedContribution = s.throughput * m.radiance;
        }
    } else {                                                                     // :1130-1136 (shift_failed)
        weight = m.num / (GD_D_EPSILON + m.den);
        shiftedContribution = mk(0.0);
    }
    add_offset_sums(A, i, m.contrib * weight, shiftedContribution * weight, (shiftedContribution - m.contrib) * weight);   // :1140-1146
}

struct Lane {
    // base path ("main" RayState, gpt.cpp:135-173)
    d3 throughput;
    Float pdf, eta;
    Vertex v;
    d3 rayO, rayD;
    int depth;
    Offset off[4];
    Float sx, sy;
    Rng rng;
    unsigned nClosest, nShadow;
};

// ---- how a bounce gets its rays traced ------------------------------------------------------------------------------------------------
// bounce() asks for every ray of evaluate()'s main loop through a tracer object; the ray sites of one bounce are numbered:
//   0      the base path's emitter sample (shadow ray)              gpt.cpp:572 (sampleEmitterDirect's visibility test)
//   1 + i  offset i's own emitter sample (shadow ray)               gpt.cpp:676
//   5      the base path's extension (closest hit)                  gpt.cpp:768-777
//   6 + i  offset i's ray: the reconnection's visibility test (shadow ray, :84-93,96-114) or the half-vector shift's extension (closest hit, :1050)
// Sites 0-5 depend on the state at the top of the bounce only ("level 1"); sites 6-9 also on the hit of site 5 ("level 2").
// MODE 0: the ray is traced in place (the megakernels).  The wavefront pipeline (gpt_wavefront.hip.h) runs the SAME source three ways:
// MODE 1 writes the level-1 rays to a queue and stops; MODE 2 replays the bounce with their results and writes the level-2 rays; MODE 3
// replays it with all results and is the only one that keeps what it computed -- one source, so the arithmetic is identical by construction.
enum { SITE_NEE = 0, SITE_NEE_OFF = 1, SITE_EXT = 5, SITE_OFF = 6, N_SITES = 10 };
struct InlineTracer {
    static constexpr int MODE = 0;
    const SceneView &sv;
    int *stack;
    __device__ __forceinline__ bool occluded(int, Lane &L, d3 o, d3 d, Float maxt) const
    {
        Hit h;
        L.nShadow++;
        return trace<true>(sv, stack, o, d, ray_mint_shadow(o, GD_EPSILON), maxt, h);
    }
    __device__ __forceinline__ void closest(int, Lane &L, d3 o, d3 d, Hit &h) const
    {
        L.nClosest++;
        trace<false>(sv, stack, o, d, ray_mint_closest(o, GD_EPSILON), GD_INF, h);
    }
    // the offsets' states: here the Lane's own (a wavefront pass may keep them in the sample's record instead and fetch each one where it is used).
    // MODIFIES: the body changes the offset (the BSDF-sampling part of a bounce; the emitter-sampling part only reads it)
    template <bool UNROLL, bool MODIFIES, class BODY>
    __device__ __forceinline__ void each_offset(Lane &L, BODY &&body) const { for_offsets<UNROLL>(L.off, body); }
    __device__ __forceinline__ void scale_offset_pdfs(Lane &L, Float q) const
    {
#pragma unroll
        for (int i = 0; i < 4; i++) L.off[i].pdf *= q;
    }
};

// ---- the deferred form of the continuation (round 6): k_walk runs the BASE path alone and logs, per bounce, the values its joined offsets would have read
// (NeeShared / BsdfShared / the roulette factor); k_replay applies them to the four offsets and the sample's sums afterwards.  Why: an ablation of k_continue
// (config-2 chunk, 32.3 ms) -- without its emitter-sample half 27.0 ms, without the shadow ray 28.8 ms, without the OFFSETS' part of a bounce 18.5 ms, and the base path
// alone at three waves per SIMD with 23 stores per bounce in place of them 13.8 ms: more than half of the kernel was the four (throughput, pdf) pairs, the re-evaluations
// of a recently connected offset and the 27 sums of 4 x 64 lanes riding along through every traversal (56 more live registers, the 60 KB sums tile that pins the
// kernel at two waves per SIMD).  The offsets never feed back into the base path (gpt.cpp:844-888), so they need not be there.
// Log: component-major doubles, [field][entry]; entry = position in the round's list; WK bounce records per round + one record of the RecentTerms of the sample's
// first bounce after the hand-over (the walker evaluates them: it has the base vertex at hand).
constexpr int WK = 4;            // bounces a sample walks per round
constexpr int WF = 25;           // doubles per bounce record
constexpr int WX = 32;           // doubles of the recent-terms record that follows the WK bounce records
// bounce record: 0 flags (1 emitter-sample half, 2 BSDF-sample half, 4 roulette factor) | 1-4 NeeShared dRecPdf, mainBsdfPdf, num, den | 5-7 X | 8-10 contribAll |
//                11-13 W | 14-17 BsdfShared mainBsdfPdf, lumPdf, num, den | 18-20 radiance | 21-23 contrib | 24 q
// recent-terms record (fields WK x WF + ...): 8 i + 0..3 offset i's RecentTerm of the emitter-sample half, 8 i + 4..7 of the BSDF-sample half -- written in the sample's first
// bounce after an early hand-over, for the offsets that are RAY_RECENTLY_CONNECTED there
constexpr int WLOG = WK * WF + WX;   // doubles per entry
struct WalkTracer : InlineTracer {
    Float *log;                 // the entry's column: wLog + e
    size_t cap;                 // entries per field
    int k;                      // record of the current bounce
    unsigned flags;             // halves of the current bounce logged so far
    unsigned recentMask;        // offsets that are RAY_RECENTLY_CONNECTED in this bounce (0: none, or not the sample's first bounce after the hand-over)
    const d3 *recent;           // their last own vertices
    __device__ __forceinline__ WalkTracer(const SceneView &v, int *stk, Float *l, size_t c, int k_, unsigned mask, const d3 *rv) : InlineTracer{v, stk}, log(l), cap(c), k(k_), flags(0), recentMask(mask), recent(rv) {}
    __device__ __forceinline__ void putf(int field, Float v) const { qst(&log[(size_t)field * cap], v); }
    __device__ __forceinline__ void put(int rec, int field, Float v) const { putf(rec * WF + field, v); }
    __device__ __forceinline__ void put3(int rec, int field, d3 v) const { put(rec, field, v.x); put(rec, field + 1, v.y); put(rec, field + 2, v.z); }
    __device__ __forceinline__ void log_nee(const NeeShared &n, const BaseVertexRef &b)
    {
        put(k, 1, n.dRecPdf); put(k, 2, n.mainBsdfPdf); put(k, 3, n.num); put(k, 4, n.den); put3(k, 5, n.X); put3(k, 8, n.contribAll);
        flags |= 1u;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if ((recentMask >> i) & 1u) {
                const RecentTerm r = recent_nee(b, recent[i], n);
                putf(WK * WF + 8 * i + 0, r.f.x); putf(WK * WF + 8 * i + 1, r.f.y); putf(WK * WF + 8 * i + 2, r.f.z); putf(WK * WF + 8 * i + 3, r.pdf);
            }
    }
    __device__ __forceinline__ void log_bsdf(const BsdfShared &m, const BaseVertexRef &b)
    {
        put3(k, 11, m.W); put(k, 14, m.mainBsdfPdf); put(k, 15, m.lumPdf); put(k, 16, m.num); put(k, 17, m.den); put3(k, 18, m.radiance); put3(k, 21, m.contrib);
        flags |= 2u;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if ((recentMask >> i) & 1u) {
                const RecentTerm r = recent_bsdf(b, recent[i], m);
                putf(WK * WF + 8 * i + 4, r.f.x); putf(WK * WF + 8 * i + 5, r.f.y); putf(WK * WF + 8 * i + 6, r.f.z); putf(WK * WF + 8 * i + 7, r.pdf);
            }
    }
    __device__ __forceinline__ void log_rr(Float q) { put(k, 24, q); flags |= 4u; }
};
struct AccNone { };             // (k_walk: bounce<PH_WALK> adds to no sums)

// Starts base path `sample` of pixel (px,py): evaluatePoint (gpt.cpp:397-436) + the prologue of evaluate (:468-531).
// Returns false if the base path is already over.
// CALLS: the five primary traversals go through the real-call form (the 2-wave builds); the 4-wave builds inline them, because the callee
// needs 132 registers and would cost them a wave per SIMD.
// STAGED: the kernel is the staged pipeline's (primary hits and a sample queue are always there): the traversal loop below is compiled out.
template <bool ENV, bool SMOOTH, bool CALLS, class ACC, bool STAGED = false>
__device__ __forceinline__ bool start_path(const SceneD &S, const SceneView &sv, const ConfigD &cfg, int *stack, Lane &L, ACC &A, int px, int py, int sample,
                                           const FilmD *F = nullptr, unsigned slot = 0)
{
    const Float shx[4] = {1.0, 0.0, -1.0, 0.0}, shy[4] = {0.0, 1.0, 0.0, -1.0};   // gpt.cpp:410-415
    L.rng.init(cfg.seed, (uint64_t)py * S.cam.width + px, (uint64_t)sample);
    L.sx = px + L.rng.next1D();                                                  // gpt.cpp:1261
    L.sy = py + L.rng.next1D();
    Float apx = 0.5, apy = 0.5;
    if (S.cam.thinlens) { apx = L.rng.next1D(); apy = L.rng.next1D(); }          // gpt.cpp:1262-1264 (needsApertureSample)
    if (S.cam.needsTime) (void)L.rng.next1D();                                   // gpt.cpp:1265-1267 (needsTimeSample): ray.time of static transforms
    L.throughput = mk(1.0); L.pdf = 1.0; L.eta = 1.0; L.depth = 1;
    A.zero();
    // five primary rays: traversal in ONE rolled loop (one copy of the traversal code), results parked in a small array
    Hit hits[5];
    const bool traced = STAGED || (F && F->pHit);   // the five primary rays were traced by k_primary (a lean traversal-only kernel at full occupancy)
    if (traced) {
#pragma unroll
        for (int r = 0; r < 5; r++) {
            hits[r].t = qld(&F->pHit[(size_t)(3 * r) * F->qCapacity + slot]);
            hits[r].u = qld(&F->pHit[(size_t)(3 * r + 1) * F->qCapacity + slot]);
            hits[r].v = qld(&F->pHit[(size_t)(3 * r + 2) * F->qCapacity + slot]);
            hits[r].prim = __builtin_nontemporal_load(&F->pPrim[(size_t)r * F->qCapacity + slot]);
        }
        L.nClosest += 5;
    }
#pragma unroll 1
    for (int r = 0; r < (traced ? 0 : 5); r++) {
        d3 o, d;
        Float mint, maxt;
        const Float ox = r == 1 ? 1.0 : (r == 3 ? -1.0 : 0.0), oy = r == 2 ? 1.0 : (r == 4 ? -1.0 : 0.0);
        camera_ray(S.cam, L.sx + ox, L.sy + oy, apx, apy, o, d, mint, maxt);
        L.nClosest++;
        if constexpr (CALLS) hits[r] = trace_closest_call(sv, stack, o, d, ray_mint_closest(o, mint), maxt);
        else trace<false>(sv, stack, o, d, ray_mint_closest(o, mint), maxt, hits[r]);
    }
#pragma unroll
    for (int r = 0; r < 5; r++) {
        d3 o, d;
        Float mint, maxt;
        camera_ray(S.cam, L.sx + (r ? shx[r - 1] : 0.0), L.sy + (r ? shy[r - 1] : 0.0), apx, apy, o, d, mint, maxt);
        const Hit h = hits[r];
        if (r == 0) { fill_vertex(sv, h, d, L.v); L.rayO = o; L.rayD = d; }
        else {
            Offset &s = L.off[r - 1];
            fill_vertex(sv, h, d, s.v);
            s.rayD = d;
            s.throughput = mk(1.0); s.pdf = 1.0;
            s.alive = h.prim >= 0;                                               // :508-513
            s.status = RAY_NOT_CONNECTED;
        }
    }
    if (L.v.prim < 0) {                                                          // :482-492
        if (ENV && S.envIndex >= 0) {                                            // scene->evalEnvironment(main.ray): a camera ray (differentials: the map's EWA lookup, envmap.cpp:390-405)
            d3 Le = sv.emitters[S.envIndex].radiance;                             // (constant.cpp:241-243)
            if (GDPT_HAS_ENVMAP_N(S, 3)) { d3 rxD, ryD; camera_differentials(S.cam, L.sx, L.sy, rxD, ryD); Le = envmap_eval<!CALLS>(*S.envMap, L.rayD, true, rxD, ryD); }
            A.add3(ACC_VD, L.throughput * Le);
        }
        return false;
    }
    A.add3(ACC_VD, L.throughput * emitted(sv, L.v.prim, -L.rayD));                // :497-499
    if (cfg.strictNormals) {                                                     // :516-531
        { const Shading sh = shading_at<SMOOTH>(sv, L.v); if (dot(L.rayD, sh.geoN) * toLocal(sh.fr, -L.rayD).z >= 0) return false; }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            Offset &s = L.off[i];
            if (s.alive) { const Shading sh = shading_at<SMOOTH>(sv, s.v); if (dot(s.rayD, sh.geoN) * toLocal(sh.fr, -s.rayD).z >= 0) s.alive = 0; }
        }
    }
    return true;
}

// One iteration of the main loop of evaluate (gpt.cpp:537-1175).  Returns false when the base path has ended.
// ENV: the scene may have an environment emitter (compiled out otherwise: its branches cost the closed scenes 5-8 %).
// SMOOTH: the scene has triangles with per-vertex normals (shading frame and geometric normal depend on the hit).
// PH: which connection states of an offset path the build carries (the others, and with them their code, are compiled out):
//   PH_ALL     all three, decided per offset at run time (the general kernel: scenes with specular chains, the single-kernel pipeline, the probes)
//   PH_CONN    every offset path is RAY_CONNECTED or dead (the wavefront development build's continuation): every use of an offset's own vertex and direction is gone
//   PH_FIRST   every offset path is RAY_NOT_CONNECTED or dead AND no vertex of the scene can be classified glossy (no delta BSDF, every roughness above
//              cfg.shiftThreshold -- decided by the host per launch, gpt_capi.hip): the first bounce of a sample in such a scene.  getVertexType (gpt.cpp:176-231)
//              answers "diffuse" for every vertex there, so the half-vector shift (:987-1126) can never be taken and only the reconnection shift is compiled in
//   PH_JOINED  every offset path is RAY_RECENTLY_CONNECTED, RAY_CONNECTED or dead (the continuation kernel): the offsets follow the base path arithmetically
//              (:622-658, :844-888); of an offset's own state only the position of its last own vertex is still read (the re-evaluation at previousMainIts)
// The strict-normals test of the offsets (:547-554) is dropped in the builds without RAY_NOT_CONNECTED: it reads the offset's LAST OWN vertex and direction,
// which stop changing when the offset connects, and with those very values it already passed at the top of the bounce in which the offset connected -- it cannot fire again.
// INL: the cold texture / environment-map lookups are inlined (4-wave builds) or real calls (2-wave builds), see tex_eval in gpt_kernels.hip.h
enum { PH_ALL = 0, PH_CONN = 1, PH_FIRST = 2, PH_JOINED = 3, PH_WALK = 4 };
template <int PH>
__device__ __forceinline__ int offset_status(const Offset &s) { return PH == PH_CONN ? (int)RAY_CONNECTED : (PH == PH_FIRST ? (int)RAY_NOT_CONNECTED : s.status); }
//   PH_WALK    the base path alone (the deferred form's k_walk): like PH_JOINED no offset is RAY_NOT_CONNECTED, but the offsets are not here at all -- where a bounce
//              would update them it writes the bounce's NeeShared / BsdfShared through the tracer (TR::log_nee / log_bsdf / log_rr) for k_replay
template <bool ENV, bool SMOOTH, int PH, bool UNROLL, bool INL, class TR, class ACC>
__device__ __forceinline__ bool bounce(const SceneD &S, const SceneView &sv, const ConfigD &cfg, TR &tr, Lane &L, ACC &A)
{
    constexpr bool JOINED = (PH == PH_CONN || PH == PH_JOINED || PH == PH_WALK);  // no offset is RAY_NOT_CONNECTED
    constexpr bool WALK = (PH == PH_WALK);
    constexpr bool RECON = (PH == PH_FIRST);                                      // every vertex is "diffuse": reconnection shifts only
    if (!(L.depth < cfg.maxDepth || cfg.maxDepth < 0)) return false;             // :537
    const TriShade &mts = sv.shade[L.v.prim];
    const Shading msh = shading_at<SMOOTH>(sv, L.v);
    const Frame3 mfr = msh.fr;
    const d3 mGeoN = msh.geoN;
    const d3 mainWi = toLocal(mfr, -L.rayD);                                     // its.wi
    if (cfg.strictNormals) {                                                     // :541-556
        if (dot(L.rayD, mGeoN) * mainWi.z >= 0) return false;
        if (!JOINED) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Offset &s = L.off[i];
                if (s.alive) { const Shading sh = shading_at<SMOOTH>(sv, s.v); if (dot(s.rayD, sh.geoN) * toLocal(sh.fr, -s.rayD).z >= 0) s.alive = 0; }
            }
        }
    }
    const bool lastSegment = (L.depth + 1 == cfg.maxDepth);                      // :559
    const MaterialD &mainBSDF = sv.mats[mts.material];
    const d3 mainR = reflectance_at<SMOOTH, INL>(sv, mainBSDF, L.v, L.depth == 1, &S.cam, L.sx, L.sy);                  // m_reflectance->eval(its): the constant or its bitmap texture at its.uv

    // ================= direct illumination sampling, :565-730 =================
    if constexpr (TR::MODE == 2) { if (bsdfType(mainBSDF) & ESmooth) { L.rng.next1D(); L.rng.next1D(); } }   // (level 2 only needs the stream to advance, :572)
    else if (bsdfType(mainBSDF) & ESmooth) {
        DRec dRec;
        dRec.ref = L.v.p; dRec.refN = (mainBSDF.twoSided || mainBSDF.type == 3) ? mk(0.0) : mfr.n;       // records.inl:160-164 (no refN behind a back-sided BSDF)
        const Float lsx = L.rng.next1D(), lsy = L.rng.next1D();                  // :572
        d3 value = sample_emitter_direct<ENV>(S, sv, dRec, lsx, lsy);
        const bool mainEmitterVisible = !tr.occluded(SITE_NEE, L, dRec.ref, dRec.d, dRec.dist * (1 - GD_SHADOW_EPSILON)); // scene.cpp:869-876
        if (!mainEmitterVisible) value = mk(0.0);
        const d3 mainEmitterRadiance = value * dRec.pdf;                         // :575
        const d3 mainWoL = toLocal(mfr, dRec.d);
        d3 mainBSDFValue;
        Float mainBsdfPdfRaw;
        bsdf_eval_pdf(mainBSDF, mainR, mainWi, mainWoL, MEASURE_SOLID_ANGLE, mainBSDFValue, mainBsdfPdfRaw);   // :588
        const bool lightOnSurfaceSA = !(ENV && dRec.offSurfaceDiscrete);          // emitter->isOnSurface() && dRec.measure == ESolidAngle
        const Float mainBsdfPdf = (lightOnSurfaceSA && mainEmitterVisible) ? mainBsdfPdfRaw : 0;       // :592
        const Float mainDistanceSquared = len2(L.v.p - dRec.p);
        const Float mainOpposingCosine = dot(dRec.n, (L.v.p - dRec.p)) / sqrt(mainDistanceSquared);
        const Float mainWeightNumerator = L.pdf * dRec.pdf;                      // :599-600
        const Float mainWeightDenominator = (L.pdf * L.pdf) * ((dRec.pdf * dRec.pdf) + (mainBsdfPdf * mainBsdfPdf));
        const d3 mainContributionAll = L.throughput * (mainBSDFValue * mainEmitterRadiance);
        if (!cfg.strictNormals || dot(mGeoN, dRec.d) * mainWoL.z > 0) {         // :607
            if constexpr (WALK) {               // (the base path alone: what the joined offsets read of this half goes to the log, k_replay applies it)
                NeeShared n;
                n.dRecPdf = dRec.pdf; n.mainBsdfPdf = mainBsdfPdf; n.num = mainWeightNumerator; n.den = mainWeightDenominator;
                n.X = mainBSDFValue * mainEmitterRadiance; n.contribAll = mainContributionAll;
                n.radiance = mainEmitterRadiance; n.woL = mainWoL; n.visSA = (lightOnSurfaceSA && mainEmitterVisible) ? 1 : 0;
                BaseVertexRef bref; bref.bsdf = &mainBSDF; bref.R = mainR; bref.fr = mfr; bref.p = L.v.p;
                tr.log_nee(n, bref);
            } else
            tr.template each_offset<UNROLL, false>(L, [&](auto ic, Offset &s) __attribute__((always_inline)) {
                const int i = ic;
                d3 shiftedContribution = mk(0.0);
                Float weight = 0;
                bool assigned = false;          // false: weight and both contributions stay 0 (:613-615 with no branch taken)
                bool shiftSuccessful = s.alive != 0;
                const int status = offset_status<PH>(s);
                if (shiftSuccessful) {
                    if (status == RAY_CONNECTED) {                               // :622-637
                        const Float den = (s.pdf * s.pdf) * ((dRec.pdf * dRec.pdf) + (mainBsdfPdf * mainBsdfPdf));
                        weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                        shiftedContribution = 1.0 * s.throughput * (mainBSDFValue * mainEmitterRadiance);
                        assigned = true;
                    } else if (status == RAY_RECENTLY_CONNECTED) {               // :638-658
                        const d3 incoming = normalize(s.v.p - L.v.p);
                        d3 f;
                        Float pdfRaw;
                        bsdf_eval_pdf(mainBSDF, mainR, toLocal(mfr, incoming), toLocal(mfr, dRec.d), MEASURE_SOLID_ANGLE, f, pdfRaw);
                        const Float shiftedBsdfPdf = (lightOnSurfaceSA && mainEmitterVisible) ? pdfRaw : 0;
                        const Float den = (s.pdf * s.pdf) * ((dRec.pdf * dRec.pdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                        weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                        shiftedContribution = 1.0 * s.throughput * (f * mainEmitterRadiance);
                        assigned = true;
                    } else if constexpr (!JOINED) {                              // :659-705
                        const TriShade &sts = sv.shade[s.v.prim];
                        const MaterialD &shiftedBSDF = sv.mats[sts.material];
                        if (RECON || !lightOnSurfaceSA || (vertex_is_diffuse(mainBSDF, cfg, ESmooth) && vertex_is_diffuse(shiftedBSDF, cfg, ESmooth))) {   // mainAtPointLight || both diffuse, :667-672
                            const Shading ssh = shading_at<SMOOTH>(sv, s.v);
                            const Frame3 sfr = ssh.fr;
                            DRec sRec;
                            sRec.ref = s.v.p; sRec.refN = (shiftedBSDF.twoSided || shiftedBSDF.type == 3) ? mk(0.0) : sfr.n;
                            d3 sv_ = sample_emitter_direct<ENV>(S, sv, sRec, lsx, lsy);
                            const bool shiftedEmitterVisible = !tr.occluded(SITE_NEE_OFF + i, L, sRec.ref, sRec.d, sRec.dist * (1 - GD_SHADOW_EPSILON));
                            if constexpr (TR::MODE == 1) return;
                            if (!shiftedEmitterVisible) sv_ = mk(0.0);
                            const d3 shiftedEmitterRadiance = sv_ * sRec.pdf;
                            const Float shiftedDRecPdf = sRec.pdf;
                            const Float shiftedDistanceSquared = len2(dRec.p - s.v.p);
                            const d3 emitterDirection = (dRec.p - s.v.p) / sqrt(shiftedDistanceSquared);
                            const Float shiftedOpposingCosine = -dot(dRec.n, emitterDirection);
                            const d3 woL = toLocal(sfr, emitterDirection);
                            if (cfg.strictNormals && dot(ssh.geoN, emitterDirection) * woL.z < 0) {
                                shiftSuccessful = false;
                            } else {
                                d3 f;
                                Float pdfRaw;
                                bsdf_eval_pdf(shiftedBSDF, reflectance_at<SMOOTH, INL>(sv, shiftedBSDF, s.v, L.depth == 1, &S.cam, L.sx + offset_shift_x(i), L.sy + offset_shift_y(i)), toLocal(sfr, -s.rayD), woL, MEASURE_SOLID_ANGLE, f, pdfRaw);
                                const Float shiftedBsdfPdf = (lightOnSurfaceSA && shiftedEmitterVisible) ? pdfRaw : 0;
                                const Float jacobian = fabs(shiftedOpposingCosine * mainDistanceSquared) / (GD_EPSILON + fabs(mainOpposingCosine * shiftedDistanceSquared)); // :695
                                const Float den = (jacobian * s.pdf) * (jacobian * s.pdf) * ((shiftedDRecPdf * shiftedDRecPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                                weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                                shiftedContribution = jacobian * s.throughput * (f * shiftedEmitterRadiance);
                                assigned = true;
                            }
                        }
                    }
                }
                if (!shiftSuccessful) {                                          // :708-717
                    weight = mainWeightNumerator / (GD_D_EPSILON + mainWeightDenominator);
                    shiftedContribution = mk(0.0);
                    assigned = true;
                }
                const d3 mainContribution = assigned ? mainContributionAll : mk(0.0);
                add_offset_sums(A, i, mainContribution * weight, shiftedContribution * weight, (shiftedContribution - mainContribution) * weight);   // :723-726
            });
        }
    }

    // ================= BSDF sampling and emitter hits, :737-1151 =================
    const Float bsx = L.rng.next1D(), bsy = L.rng.next1D();                      // :456
    BSDFSample bs;
    bsdf_sample(mainBSDF, mainR, mainWi, bsx, bsy, bs);
    if (bs.pdf <= 0.0) return false;                                             // :740
    const d3 mainWo = toWorld(mfr, bs.wo);
    if (cfg.strictNormals && dot(mGeoN, mainWo) * bs.wo.z <= 0) return false;    // :749
    const d3 prevP = L.v.p, prevWi = mainWi;                                      // previousMainIts, :754
    const bool mainVertexDiffuse = vertex_is_diffuse(mainBSDF, cfg, bs.sampledType);     // :765
    L.rayO = prevP;
    L.rayD = mainWo;                                                              // :768
    Float hitT;                                                                   // Intersection::t of the new base vertex
    bool mainHitEnvV = false;                                                     // the base path left the scene into the environment emitter
    DRec envRec;                                                                  // mainDRec of an environment hit (:790-797)
    {
        Hit h;
        tr.closest(SITE_EXT, L, L.rayO, L.rayD, h);
        if constexpr (TR::MODE == 1) return true;                                // (the level-1 rays of this bounce are out)
        if (h.prim < 0) {                                                        // :786-804
            if (!ENV || S.envIndex < 0) return false;
            envRec.ref = prevP; envRec.refN = (mainBSDF.twoSided || mainBSDF.type == 3) ? mk(0.0) : mfr.n;
            if (!env_fill_drec(S, envRec, L.rayO, L.rayD)) return false;
            mainHitEnvV = true;
            L.v.prim = -1;
        } else {
            fill_vertex(sv, h, L.rayD, L.v);
        }
        hitT = h.t;
    }
    const bool mainHitEnv = ENV && mainHitEnvV;
    const TriShade &nts = sv.shade[mainHitEnv ? 0 : L.v.prim];
    const bool mainHitEmitter = mainHitEnv || nts.emitter >= 0;                   // :772-777, :793
    const d3 mainEmitterRadiance = mainHitEnv ? env_radiance<INL>(S, sv, L.rayD) : (mainHitEmitter ? emitted(sv, L.v.prim, -L.rayD) : mk(0.0));
    const bool mainNextVertexDiffuse = mainHitEnv ? true : vertex_is_diffuse(sv.mats[nts.material], cfg, bs.sampledType);  // :785, :799
    const Float mainBsdfPdf = bs.pdf, mainPreviousPdf = L.pdf;
    L.throughput = L.throughput * (bs.weight * bs.pdf);                          // :810-812
    L.pdf *= bs.pdf;
    L.eta *= bs.eta;
    // mainDRec: ref = previous vertex, refN = its shading normal; setQuery (records.inl:170-178): p, n, d, dist
    const Float mainLumPdf = (mainHitEmitter && !(bs.sampledType & EDelta))
        ? (mainHitEnv ? pdf_emitter_direct<ENV>(S, sv, S.envIndex, envRec.d, envRec.refN, envRec.n, envRec.dist)
                      : pdf_emitter_direct<ENV>(S, sv, nts.emitter, L.rayD, (mainBSDF.twoSided || mainBSDF.type == 3) ? mk(0.0) : mfr.n, nts.n, hitT)) : 0;  // :815
    const Float mainWeightNumerator = mainPreviousPdf * bs.pdf;                   // :819-820
    const Float mainWeightDenominator = (mainPreviousPdf * mainPreviousPdf) * ((mainLumPdf * mainLumPdf) + (mainBsdfPdf * mainBsdfPdf));
    const d3 mainContribution = L.throughput * mainEmitterRadiance;
    const int measure = (bs.sampledType & EDelta) ? MEASURE_DISCRETE : MEASURE_SOLID_ANGLE;

    if constexpr (WALK) {
        BsdfShared m;
        m.W = bs.weight * bs.pdf; m.mainBsdfPdf = mainBsdfPdf; m.lumPdf = mainLumPdf; m.num = mainWeightNumerator; m.den = mainWeightDenominator;
        m.radiance = mainEmitterRadiance; m.contrib = mainContribution; m.woL = toLocal(mfr, L.rayD); m.measure = measure;
        BaseVertexRef bref; bref.bsdf = &mainBSDF; bref.R = mainR; bref.fr = mfr; bref.p = L.rayO;
        tr.log_bsdf(m, bref);
    } else
    tr.template each_offset<UNROLL, true>(L, [&](auto ic, Offset &s) __attribute__((always_inline)) {      // :830
        const int i = ic;
        d3 shiftedContribution = mk(0.0);
        Float weight = 0;
        bool assigned = false;
        bool postponedShiftEnd = false;
        if (s.alive) {
            const Float shiftedPreviousPdf = s.pdf;
            const int status = offset_status<PH>(s);
            if (status == RAY_CONNECTED) {                                       // :844-861
                s.throughput = s.throughput * (bs.weight * bs.pdf);
                s.pdf *= mainBsdfPdf;
                const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((mainLumPdf * mainLumPdf) + (mainBsdfPdf * mainBsdfPdf));
                weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                shiftedContribution = s.throughput * mainEmitterRadiance;
                assigned = true;
            } else if (status == RAY_RECENTLY_CONNECTED) {                       // :862-888
                const d3 incoming = normalize(s.v.p - L.rayO);
                d3 f;
                Float shiftedBsdfPdf;
                bsdf_eval_pdf(mainBSDF, mainR, toLocal(mfr, incoming), toLocal(mfr, L.rayD), measure, f, shiftedBsdfPdf);
                s.throughput = s.throughput * f;
                s.pdf *= shiftedBsdfPdf;
                s.status = RAY_CONNECTED;
                const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((mainLumPdf * mainLumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                shiftedContribution = s.throughput * mainEmitterRadiance;
                assigned = true;
            } else if constexpr (!JOINED) {                                      // :889-1126
                const TriShade &sts = sv.shade[s.v.prim];
                const MaterialD &shiftedBSDF = sv.mats[sts.material];
                const Shading ssh = shading_at<SMOOTH>(sv, s.v);
                const Frame3 sfr = ssh.fr;
                const bool shiftedVertexDiffuse = vertex_is_diffuse(shiftedBSDF, cfg, bs.sampledType);
                const d3 shiftedR = reflectance_at<SMOOTH, INL>(sv, shiftedBSDF, s.v, L.depth == 1, &S.cam, L.sx + offset_shift_x(i), L.sy + offset_shift_y(i));   // (depth 1: still the offset's camera-ray hit)
                if (RECON || (mainVertexDiffuse && mainNextVertexDiffuse && shiftedVertexDiffuse)) {
                    // ---- reconnection shift, :897-986 ----
                    if (!lastSegment || mainHitEmitter) {                        // :901
                        // reconnectShift, gpt.cpp:316-345; environmentShift + testEnvironmentVisibility, :96-114,348-369
                        bool visible;
                        if (mainHitEnv) {
                            DRec er;
                            er.dist = 0.0;
                            env_fill_drec(S, er, s.v.p, L.rayD);
                            visible = !tr.occluded(SITE_OFF + i, L, s.v.p, L.rayD, (1.0 - GD_SHADOW_EPSILON) * er.dist);
                        } else visible = !tr.occluded(SITE_OFF + i, L, s.v.p, L.v.p - s.v.p, 1.0 - GD_SHADOW_EPSILON);     // testVisibility, gpt.cpp:84-93: unnormalised direction, maxt = 1 - ShadowEpsilon
                        if constexpr (TR::MODE == 2) return;
                        if (!visible) { s.alive = 0; }
                        else if (mainHitEnv) {
                            // reconnection at infinity: J = 1, wo = the base direction (:364-366); radiance and light pdf of the base (:972-976)
                            const d3 shiftedWo = L.rayD;
                            const d3 woL = toLocal(sfr, shiftedWo);
                            if (cfg.strictNormals && dot(shiftedWo, ssh.geoN) * woL.z <= 0) { s.alive = 0; }
                            else {
                                d3 f;
                                Float shiftedBsdfPdf;
                                bsdf_eval_pdf(shiftedBSDF, shiftedR, toLocal(sfr, -s.rayD), woL, MEASURE_SOLID_ANGLE, f, shiftedBsdfPdf);
                                s.throughput = s.throughput * (f * 1.0);
                                s.pdf *= shiftedBsdfPdf * 1.0;
                                s.status = RAY_RECENTLY_CONNECTED;
                                const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((mainLumPdf * mainLumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                                weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                                shiftedContribution = s.throughput * mainEmitterRadiance;
                                assigned = true;
                            }
                        } else {
                            const d3 mainEdge = L.rayO - L.v.p, shiftedEdge = s.v.p - L.v.p;
                            const Float mainEdgeLengthSquared = len2(mainEdge), shiftedEdgeLengthSquared = len2(shiftedEdge);
                            const d3 shiftedWo = -shiftedEdge / sqrt(shiftedEdgeLengthSquared);
                            const d3 nGeoN = SMOOTH ? shading_at<SMOOTH>(sv, L.v).geoN : nts.n;      // main.rRec.its.geoFrame.n, :911
                            const Float mainOpposingCosine = dot(mainEdge, nGeoN) / sqrt(mainEdgeLengthSquared);
                            const Float shiftedOpposingCosine = dot(shiftedWo, nGeoN);
                            const Float jacobian = fabs(shiftedOpposingCosine * mainEdgeLengthSquared) / (GD_D_EPSILON + fabs(mainOpposingCosine * shiftedEdgeLengthSquared));
                            const d3 woL = toLocal(sfr, shiftedWo);
                            if (cfg.strictNormals && dot(shiftedWo, ssh.geoN) * woL.z <= 0) { s.alive = 0; }
                            else {
                                d3 f;
                                Float shiftedBsdfPdf;
                                bsdf_eval_pdf(shiftedBSDF, shiftedR, toLocal(sfr, -s.rayD), woL, MEASURE_SOLID_ANGLE, f, shiftedBsdfPdf);
                                s.throughput = s.throughput * (f * jacobian);    // :939-940
                                s.pdf *= shiftedBsdfPdf * jacobian;
                                s.status = RAY_RECENTLY_CONNECTED;
                                if (mainHitEmitter) {                            // :944-986
                                    const d3 shiftedEmitterRadiance = emitted(sv, L.v.prim, -shiftedWo);
                                    const Float sdist = len(L.v.p - s.v.p);
                                    const d3 sd = (L.v.p - s.v.p) / sdist;
                                    const Float shiftedLumPdf = pdf_emitter_direct<ENV>(S, sv, nts.emitter, sd, sfr.n, nts.n, sdist);
                                    const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((shiftedLumPdf * shiftedLumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                                    weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                                    shiftedContribution = s.throughput * shiftedEmitterRadiance;
                                    assigned = true;
                                }
                            }
                        }
                    }
                } else if constexpr (!RECON) {
                    // ---- half-vector duplication shift, :987-1126 ----
                    d3 shiftedEmitterRadiance = mk(0.0);
                    bool envEnd = false;
                    const d3 tsIn = toLocal(sfr, -s.rayD);
                    const bool bothDelta = (bs.sampledType & EDelta) && (bsdfType(shiftedBSDF) & EDelta);     // :996-1001
                    const bool bothSmooth = (bs.sampledType & ESmooth) && (bsdfType(shiftedBSDF) & ESmooth);
                    bool ok = bothDelta || bothSmooth;
                    d3 tsOut = mk(0.0);
                    if (ok) {
                        Float jacobian;
                        ok = half_vector_shift(prevWi, bs.wo, tsIn, bsdf_eta(mainBSDF), bsdf_eta(shiftedBSDF), jacobian, tsOut);   // :1006
                        if (bs.sampledType & EDelta) jacobian = 1;               // :1008-1011
                        if (ok) { s.throughput = s.throughput * jacobian; s.pdf *= jacobian; }
                    }
                    if (ok) {
                        const d3 outgoing = toWorld(sfr, tsOut);
                        d3 f;
                        Float p;
                        bsdf_eval_pdf(shiftedBSDF, shiftedR, tsIn, tsOut, measure, f, p);
                        s.throughput = s.throughput * f;
                        s.pdf *= p;
                        if (s.pdf == 0) ok = false;                              // :1034
                        else if (cfg.strictNormals && dot(outgoing, ssh.geoN) * tsOut.z <= 0) ok = false;
                        else {
                            Hit h;
                            tr.closest(SITE_OFF + i, L, s.v.p, outgoing, h);                                            // :1050-1052
                            if constexpr (TR::MODE == 2) return;
                            if (h.prim < 0) {                                    // :1052-1074
                                if (!ENV || S.envIndex < 0 || !mainHitEnv || (mainVertexDiffuse && shiftedVertexDiffuse)) ok = false;
                                else { shiftedEmitterRadiance = env_radiance<INL>(S, sv, outgoing); envEnd = true; }
                            } else if (mainHitEnv) ok = false;                    // :1078-1082: no shifts between env and non-env
                            else {
                                s.rayD = outgoing;
                                fill_vertex(sv, h, outgoing, s.v);
                                const TriShade &snts = sv.shade[s.v.prim];
                                const bool shiftedNextVertexDiffuse = vertex_is_diffuse(sv.mats[snts.material], cfg, bs.sampledType);
                                if (mainVertexDiffuse && shiftedVertexDiffuse && shiftedNextVertexDiffuse) ok = false;   // :1089-1093
                                else if (snts.emitter >= 0) shiftedEmitterRadiance = emitted(sv, s.v.prim, -outgoing);
                            }
                        }
                    }
                    if (ok) {                                                    // :1106-1112
                        weight = L.pdf / (s.pdf * s.pdf + L.pdf * L.pdf);
                        shiftedContribution = s.throughput * shiftedEmitterRadiance;
                        if (envEnd) postponedShiftEnd = true;                    // :1073: the offset path ends with this segment
                    } else {                                                     // :1113-1124
                        weight = 1.0 / L.pdf;
                        shiftedContribution = mk(0.0);
                        postponedShiftEnd = true;                                // alive stays true for this accumulation
                    }
                    assigned = true;
                }
            }
        }
        if (!s.alive) {                                                          // :1130-1136 (shift_failed)
            weight = mainWeightNumerator / (GD_D_EPSILON + mainWeightDenominator);
            shiftedContribution = mk(0.0);
            assigned = true;
        }
        const d3 mc = assigned ? mainContribution : mk(0.0);
        add_offset_sums(A, i, mc * weight, shiftedContribution * weight, (shiftedContribution - mc) * weight);   // :1140-1146
        if (postponedShiftEnd) s.alive = 0;
    });
    if constexpr (TR::MODE == 2) return true;                                    // (the level-2 rays of this bounce are out)

    if (mainHitEnv) return false;                                                // :1155-1157
    if (L.depth++ >= cfg.rrDepth) {                                              // :1159-1174
        const Float q = fmin(maxc(L.throughput / L.pdf) * L.eta * L.eta, (Float)0.95f);
        if (L.rng.next1D() >= q) return false;
        L.pdf *= q;
        if constexpr (WALK) tr.log_rr(q); else tr.scale_offset_pdfs(L, q);
    }
    return true;
}

// Accumulates one finished sample: the 15 puts of gpt.cpp:1314-1352.  Fast path = per-pixel sums (every put covers
// exactly its expected pixel); otherwise the exact generic path.
// A sample's sums can stand for 15 accepted single-pixel puts (the per-pixel record) iff every sum is valid as ImageBlock::put checks it and
// every put covers exactly its expected pixel.
template <class ACC>
__device__ __forceinline__ bool sample_is_fast(const FilmD &F, const FilterD &flt, Float sx, Float sy, const ACC &A, int px, int py)
{
    Float nonFinite = 0.0, lowest = 0.0;
#pragma unroll
    for (int k = 0; k < ACC_N; k++) { const Float v = A.get(k); nonFinite += v - v; if (k < ACC_GRAD) lowest = fmin(lowest, v); }
    return nonFinite == 0.0 && !(lowest < 0.0) && F.fValues == nullptr && single_pixel(flt, sx, sy, px, py) && single_pixel(flt, sx - 1, sy, px - 1, py) &&
           single_pixel(flt, sx + 1, sy, px + 1, py) && single_pixel(flt, sx, sy - 1, px, py - 1) && single_pixel(flt, sx, sy + 1, px, py + 1);
}

template <class ACC>
__device__ __forceinline__ void finish_path(const FilmD &F, const FilterD &flt, Float sx, Float sy, const ACC &A, int px, int py, int logSlot, Float *pixelSums = nullptr)
{
    if (F.log) {
        // a reconstruction filter wider than box: no put here -- the sample's sums and position go to the log, and k_gather_log
        // evaluates every put from the side of the pixel that receives it (no atomics, a fixed order)
        const size_t plane = (size_t)F.logRows * F.W, at = (size_t)logSlot * plane + (size_t)(py - F.logY0) * F.W + px, comp = (size_t)F.logChunk * plane;
#pragma unroll
        for (int k = 0; k < ACC_N; k++) F.log[(size_t)k * comp + at] = A.get(k);
        F.log[(size_t)30 * comp + at] = sx;
        F.log[(size_t)31 * comp + at] = sy;
        return;
    }
    enum { RIGHT = 0, BOTTOM = 1, LEFT = 2, TOP = 3 };
    // (other reconstruction filters than box spread every put over several pixels: they always take the generic path)
    // The per-pixel sums stand for 15 puts that are all accepted.  A sample with a non-finite sum, or a negative throughput / very-direct
    // sum (only dx and dy accept negative values), takes the generic path, which checks every put as ImageBlock::put does.
    const bool fast = sample_is_fast(F, flt, sx, sy, A, px, py);
    if (fast && pixelSums) {
        // (k_fold_cont: the pixel's samples of a chunk are summed in registers, the record is touched once)
        pixelSums[0] += 1.0;
#pragma unroll
        for (int k = 0; k < 3; k++) { pixelSums[1 + k] += A.get(ACC_T + k); pixelSums[4 + k] += A.get(ACC_VD + k); }
#pragma unroll
        for (int k = 0; k < 12; k++) { pixelSums[7 + k] += A.get(ACC_NBR + k); pixelSums[19 + k] += A.get(ACC_GRAD + k); }
    } else if (fast) {
        Float *r = F.rec + (size_t)(py - (F.y0 - 1)) * F.W + px;
        const size_t st = F.recStride;
        r[0] += 1.0;
        const d3 T = A.get3(ACC_T), vd = A.get3(ACC_VD);
        r[1 * st] += T.x; r[2 * st] += T.y; r[3 * st] += T.z;
        r[4 * st] += vd.x; r[5 * st] += vd.y; r[6 * st] += vd.z;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const d3 nb = A.get3(ACC_NBR + 3 * d), g = A.get3(ACC_GRAD + 3 * d);
            r[(7 + 3 * d) * st] += nb.x; r[(8 + 3 * d) * st] += nb.y; r[(9 + 3 * d) * st] += nb.z;
            r[(19 + 3 * d) * st] += g.x; r[(20 + 3 * d) * st] += g.y; r[(21 + 3 * d) * st] += g.z;
        }
    } else {
        const d3 T = A.get3(ACC_T), vd = A.get3(ACC_VD);
        spill_put(F, flt, sx, sy, (8 * vd) + (2 * T), 4.0, 0);
        spill_put(F, flt, sx - 1, sy, 2 * A.get3(ACC_NBR + 3 * LEFT), 1.0, 0);
        spill_put(F, flt, sx + 1, sy, 2 * A.get3(ACC_NBR + 3 * RIGHT), 1.0, 0);
        spill_put(F, flt, sx, sy - 1, 2 * A.get3(ACC_NBR + 3 * TOP), 1.0, 0);
        spill_put(F, flt, sx, sy + 1, 2 * A.get3(ACC_NBR + 3 * BOTTOM), 1.0, 0);
        spill_put(F, flt, sx, sy, 2 * T, 4.0, 1);
        spill_put(F, flt, sx - 1, sy, 2 * A.get3(ACC_NBR + 3 * LEFT), 1.0, 1);
        spill_put(F, flt, sx + 1, sy, 2 * A.get3(ACC_NBR + 3 * RIGHT), 1.0, 1);
        spill_put(F, flt, sx, sy - 1, 2 * A.get3(ACC_NBR + 3 * TOP), 1.0, 1);
        spill_put(F, flt, sx, sy + 1, 2 * A.get3(ACC_NBR + 3 * BOTTOM), 1.0, 1);
        spill_put(F, flt, sx - 1, sy, -(2 * A.get3(ACC_GRAD + 3 * LEFT)), 1.0, 2);
        spill_put(F, flt, sx, sy, 2 * A.get3(ACC_GRAD + 3 * RIGHT), 1.0, 2);
        spill_put(F, flt, sx, sy - 1, -(2 * A.get3(ACC_GRAD + 3 * TOP)), 1.0, 3);
        spill_put(F, flt, sx, sy, 2 * A.get3(ACC_GRAD + 3 * BOTTOM), 1.0, 3);
        spill_put(F, flt, sx, sy, vd, 1.0, 4);
    }
}

// Block prologue shared by the render and the continuation kernel.  Dynamic LDS: [traversal stack: stackDepth x TBLK ints][per-sample
// sums (ACC_LDS only)][staged scene tables (LDS_SCENE only)]; sized by the host from the actual BVH depth and table bytes so that small
// scenes leave room for more resident blocks per CU.
template <bool LDS_SCENE, bool ACC_LDS>
__device__ __forceinline__ void block_setup(const SceneD &S, int stackDepth, unsigned char *s_dyn, SceneView &sv, int *&stack, unsigned char *&s_acc, size_t extraBytes = 0 /* more room at s_acc (k_shift5's mailboxes) */)
{
    int *s_stack = reinterpret_cast<int *>(s_dyn);
#ifdef GDPT_LDS_POISON              /* investigation build: every dynamic-LDS word starts as a signalling pattern (a NaN as double, a huge
                                       index as int), so that any read of a stack slot, sum or table word before its first write shows up
                                       as a wrong film or a fault instead of passing by luck */
    {
        const unsigned words = (unsigned)(((size_t)stackDepth * TBLK * sizeof(int) + (ACC_LDS ? sizeof(Float) * ACC_N * TBLK : 0) + (LDS_SCENE ? S.ldsBytes : 0)) / 4);
        for (unsigned i = threadIdx.x; i < words; i += TBLK) reinterpret_cast<unsigned *>(s_dyn)[i] = 0x7ff7dead;
        __syncthreads();
    }
#endif
#ifdef GDPT_LAYOUT_SCENE_FIRST      /* investigation build: the round-1 layout [stack][scene][sums] that faulted in one build */
    unsigned char *s_scene = s_dyn + (size_t)stackDepth * TBLK * sizeof(int);
    s_acc = s_scene + S.ldsBytes;
#else
    s_acc = s_dyn + (size_t)stackDepth * TBLK * sizeof(int);
    unsigned char *s_scene = s_acc + (ACC_LDS ? sizeof(Float) * ACC_N * TBLK : 0) + extraBytes;
#endif
    if (LDS_SCENE) {
        // stage node packets, triangle records and the shading tables through LDS once per block (coalesced 16-byte copies);
        // the compile-time branch lets the compiler address them with ds_read instead of flat loads
        // (every table is allocated in whole 16-byte words, see upload() on the host)
        const int nb[8] = {S.numNodes * (int)sizeof(BvhNode), S.numTris * (int)sizeof(TriIsect), S.numTris * (int)sizeof(TriShade),
                           S.numMats * (int)sizeof(MaterialD), S.numEmitters * (int)sizeof(EmitterD),
                           S.numEmTris * (int)sizeof(EmTri), S.numEmCdf * (int)sizeof(Float), (S.numEmitters + 1) * (int)sizeof(Float)};
        const void *src[8] = {S.nodes, S.isect, S.shade, S.mats, S.emitters, S.emTris, S.emCdf, S.emitterCdf};
        int off = 0, offs[8];
        for (int a = 0; a < 8; a++) {
            offs[a] = off;
            const uint4 *g = reinterpret_cast<const uint4 *>(src[a]);
            uint4 *l = reinterpret_cast<uint4 *>(s_scene + off);
            for (int i = threadIdx.x; i < (nb[a] + 15) / 16; i += TBLK) l[i] = g[i];
            off += (nb[a] + 15) & ~15;
        }
        __syncthreads();
        sv.nodes = reinterpret_cast<const BvhNode *>(s_scene + offs[0]);
        sv.isect = reinterpret_cast<const TriIsect *>(s_scene + offs[1]);
        sv.shade = reinterpret_cast<const TriShade *>(s_scene + offs[2]);
        sv.mats = reinterpret_cast<const MaterialD *>(s_scene + offs[3]);
        sv.emitters = reinterpret_cast<const EmitterD *>(s_scene + offs[4]);
        sv.emTris = reinterpret_cast<const EmTri *>(s_scene + offs[5]);
        sv.emCdf = reinterpret_cast<const Float *>(s_scene + offs[6]);
        sv.emitterCdf = reinterpret_cast<const Float *>(s_scene + offs[7]);
    } else { sv.nodes = S.nodes; sv.isect = S.isect; sv.shade = S.shade; sv.mats = S.mats; sv.emitters = S.emitters; sv.emTris = S.emTris; sv.emCdf = S.emCdf; sv.emitterCdf = S.emitterCdf; }
    sv.rootRef = S.rootRef; sv.boundM = S.boundM; sv.quant = LDS_SCENE ? 0 : S.quantNodes; sv.leafExit = LDS_SCENE ? 0 : 1;   // (a scene staged into LDS always has fp32 nodes)
    sv.vn = S.vn;                      // per-vertex normals, texture coordinates and textures stay in HBM
    sv.uv = S.uv; sv.hasUV = S.hasUV; sv.tex = S.tex;
    stack = s_stack + threadIdx.x;
}

// ---- continuation records ---------------------------------------------------------------------------------------------------------
// A sample leaves the first-stage kernel as soon as none of its four offset paths is RAY_NOT_CONNECTED any more (round 6; until round 5: as soon as
// each was RAY_CONNECTED, one bounce later) -- for diffuse and rough scenes after the FIRST bounce.  From there the offsets are four (throughput, pdf)
// pairs that follow the base path arithmetically (gpt.cpp:622-658,844-888; a RAY_RECENTLY_CONNECTED offset re-evaluates the base vertex's BSDF with the
// direction from its own last vertex, whose position travels with it), so the rest of the base path runs in k_continue: one small state per lane, two
// of the three connection states, waves refilled from the queue as lanes finish -- instead of the general kernel's deep-bounce phase, where under half
// of a wave's lanes are alive and every bounce drags the code of all three connection states along.  Record = NQ doubles, component-major ([k][slot]):
//   0-2 throughput | 3 pdf | 4 eta | 5-7 v.p | 8-10 rayD | 11-12 v.u, v.v | 13 prim (low 32 bits), depth (high) -- all ones once finished |
//   14 rng state | 15+4i..18+4i offset i: throughput, pdf | 31 alive mask (bits 0-3), RAY_RECENTLY_CONNECTED mask (bits 4-7) |
//   32-61 the sample's 30 sums so far (finished: its final sums) | 62+3i..64+3i offset i: position of its last own vertex (read while RAY_RECENTLY_CONNECTED)
// ConfigD::handoffEarly = 0 (HBM-resident scenes; the wavefront development build, whose stages carry RAY_CONNECTED offsets only): the hand-over rule of rounds 2-5.
constexpr unsigned long long Q_DONE = ~0ULL;
__device__ __forceinline__ void q_store_main(const FilmD &F, unsigned slot, const Lane &L)
{
    Float *q = F.qRec + slot;
    const size_t st = F.qCapacity;
    qst(&q[0 * st], L.throughput.x); qst(&q[1 * st], L.throughput.y); qst(&q[2 * st], L.throughput.z);
    qst(&q[3 * st], L.pdf); qst(&q[4 * st], L.eta);
    qst(&q[5 * st], L.v.p.x); qst(&q[6 * st], L.v.p.y); qst(&q[7 * st], L.v.p.z);
    qst(&q[8 * st], L.rayD.x); qst(&q[9 * st], L.rayD.y); qst(&q[10 * st], L.rayD.z);
    qst(&q[11 * st], L.v.u); qst(&q[12 * st], L.v.v);
    qst(&q[13 * st], __longlong_as_double((long long)(((unsigned long long)(unsigned)L.depth << 32) | (unsigned)L.v.prim)));
    qst(&q[14 * st], __longlong_as_double((long long)L.rng.position()));
}
template <class ACC>
__device__ __forceinline__ void q_store(const FilmD &F, unsigned slot, const Lane &L, const ACC &A)
{
    q_store_main(F, slot, L);
    Float *q = F.qRec + slot;
    const size_t st = F.qCapacity;
    unsigned alive = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const Offset &o = L.off[i];
        qst(&q[(15 + 4 * i) * st], o.throughput.x); qst(&q[(16 + 4 * i) * st], o.throughput.y); qst(&q[(17 + 4 * i) * st], o.throughput.z); qst(&q[(18 + 4 * i) * st], o.pdf);
        alive |= (o.alive ? 1u : 0u) << i;
        const bool rc = o.alive && o.status == RAY_RECENTLY_CONNECTED;          // (only the early hand-over leaves any)
        alive |= (rc ? 16u : 0u) << i;
        if (rc) { qst(&q[(62 + 3 * i) * st], o.v.p.x); qst(&q[(63 + 3 * i) * st], o.v.p.y); qst(&q[(64 + 3 * i) * st], o.v.p.z); }
    }
    qst(&q[31 * st], __longlong_as_double((long long)alive));
#pragma unroll
    for (int k = 0; k < ACC_N; k++) qst(&q[(32 + k) * st], A.get(k));
}
__device__ __forceinline__ void q_load_main(const FilmD &F, unsigned slot, Lane &L)          // the base path's part of a record
{
    const Float *q = F.qRec + slot;
    const size_t st = F.qCapacity;
    L.throughput = mk(qld(&q[0 * st]), qld(&q[1 * st]), qld(&q[2 * st]));
    L.pdf = qld(&q[3 * st]); L.eta = qld(&q[4 * st]);
    L.v.p = mk(qld(&q[5 * st]), qld(&q[6 * st]), qld(&q[7 * st]));
    L.rayD = mk(qld(&q[8 * st]), qld(&q[9 * st]), qld(&q[10 * st]));
    L.v.u = qld(&q[11 * st]); L.v.v = qld(&q[12 * st]);
    const unsigned long long pk = (unsigned long long)__double_as_longlong(qld(&q[13 * st]));
    L.v.prim = (int)(unsigned)(pk & 0xffffffffu); L.depth = (int)(unsigned)(pk >> 32);
    L.rng.set_position((uint64_t)__double_as_longlong(qld(&q[14 * st])));
}
__device__ __forceinline__ void q_load_lane(const FilmD &F, unsigned slot, Lane &L)          // the path's part of a record: base path and offsets
{
    q_load_main(F, slot, L);
    const Float *q = F.qRec + slot;
    const size_t st = F.qCapacity;
    const unsigned alive = (unsigned)__double_as_longlong(qld(&q[31 * st]));
#pragma unroll
    for (int i = 0; i < 4; i++) {
        Offset &o = L.off[i];
        o.throughput = mk(qld(&q[(15 + 4 * i) * st]), qld(&q[(16 + 4 * i) * st]), qld(&q[(17 + 4 * i) * st])); o.pdf = qld(&q[(18 + 4 * i) * st]);
        o.alive = (alive >> i) & 1; o.status = RAY_CONNECTED;
        if ((alive >> (4 + i)) & 1) {
            o.status = RAY_RECENTLY_CONNECTED;
            o.v.p = mk(qld(&q[(62 + 3 * i) * st]), qld(&q[(63 + 3 * i) * st]), qld(&q[(64 + 3 * i) * st]));
        }
    }
}
template <class ACC>
__device__ __forceinline__ void q_load(const FilmD &F, unsigned slot, Lane &L, ACC &A)
{
    q_load_lane(F, slot, L);
    const Float *q = F.qRec + slot;
    const size_t st = F.qCapacity;
#pragma unroll
    for (int k = 0; k < ACC_N; k++) A.set(k, qld(&q[(32 + k) * st]));
}
// a sample that ended in the render kernel: its sums go to its slot like a continued one's (k_fold_cont adds them to the pixel)
template <class ACC>
__device__ __forceinline__ void q_finish(const FilmD &F, unsigned slot, const ACC &A)
{
    Float *q = F.qRec + slot;
    const size_t st = F.qCapacity;
#pragma unroll
    for (int k = 0; k < ACC_N; k++) qst(&q[(32 + k) * st], A.get(k));
    qst(&q[13 * st], __longlong_as_double((long long)Q_DONE));
}
// the hand-over test: no offset path of the sample is still on its own (early: RAY_NOT_CONNECTED is what keeps a sample; else anything but RAY_CONNECTED does)
__device__ __forceinline__ bool all_connected(const Lane &L, bool early)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; i++) ok = ok && (!L.off[i].alive || L.off[i].status == RAY_CONNECTED || (early && L.off[i].status == RAY_RECENTLY_CONNECTED));
    return ok;
}

// STAGED: the build the staged pipeline launches (gdpt_film_set_pipeline(2)): primary hits come from k_primary and every sample ends in
// its queue slot, so the primary traversals, finish_path and the exact generic puts (15 inlined copies with atomics) are not in it at all
// -- code that never runs there, but that the register allocator would otherwise keep values alive for.
template <bool LDS_SCENE, bool ACC_LDS, int WAVES_PER_SIMD, bool ENV, bool SMOOTH, bool STAGED = false>
__global__ __launch_bounds__(TBLK, WAVES_PER_SIMD) void k_render(SceneD S, ConfigD cfg, FilmD F, int rx0, int ry0, int rx1, int ry1, int tilesX, int tiles, int slices, int stackDepth, int sceneBytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    SceneView sv;
    int *stack;
    unsigned char *s_acc;
    block_setup<LDS_SCENE, ACC_LDS>(S, stackDepth, s_dyn, sv, stack, s_acc);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // work item = (16x16 pixel tile, slice of the spp samples).  Slices exist so that a launch smaller than the chip (a strip of
    // a multi-GPU frame) or the tail of a large one still fills it: each slice sums into its own record plane, folded in order.
    const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int s0 = cfg.sBase + (int)((long long)cfg.sCount * slice / slices), s1 = cfg.sBase + (int)((long long)cfg.sCount * (slice + 1) / slices);
    if (slice > 0) F.rec = F.recExtra + (size_t)(slice - 1) * NREC * F.recStride;
    const int px = rx0 + tx * 16 + (wave & 1) * 8 + (lane & 7), py = ry0 + ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool valid = px < rx1 && py < ry1;
    const FilterD flt = box_filter();

    Lane L;
    L.nClosest = L.nShadow = 0;
    Acc<ACC_LDS> A;
    if constexpr (ACC_LDS) A.p = reinterpret_cast<Float *>(s_acc) + threadIdx.x;
    int next = valid ? s0 : s1;         // next sample to start
    bool active = false;
    unsigned slot = 0;                  // queue slot of the lane's current sample
    bool pending = false;               // (no queue) a finished sample whose sums still sit in A: flushed to the pixel record when the lane
                                        // regenerates (together with >= regenMin others) or at the end -- not one lane at a time
                                        // (31 loads + 31 stores per flush; done per finished path they were ~1e9 wave-level memory
                                        // instructions per frame issued for one or two lanes each, with their latency exposed)
    unsigned long long pathLen = 0, paths = 0;
    while (true) {
        const bool idle = !active;
        const unsigned long long idleMask = __ballot(idle);
        const unsigned long long wantMask = __ballot(idle && next < s1);
        if (wantMask == 0 && idleMask == ~0ULL) break;
        // regenerate together: when enough lanes wait, or nothing else is running in this wave
        if (idle && next < s1 && (__popcll(wantMask) >= cfg.regenMin || idleMask == ~0ULL)) {
            if (__hip_atomic_load(F.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { next = s1; continue; }   // cancelled: no new samples; running paths finish
            if constexpr (!STAGED) { if (pending) finish_path(F, flt, L.sx, L.sy, A, px, py, next - 1 - cfg.sBase); }
#ifdef GDPT_PROBE_STAGED_WITH_COLD_CODE
            else { if (cfg.spp < 0) finish_path(F, flt, L.sx, L.sy, A, px, py, next - 1 - cfg.sBase); }
#endif
            slot = (unsigned)(next - cfg.sBase) * F.qPixels + (unsigned)tile * TBLK + threadIdx.x;
            active = start_path<ENV, SMOOTH, (WAVES_PER_SIMD <= 2), Acc<ACC_LDS>, STAGED>(S, sv, cfg, stack, L, A, px, py, next, &F, slot);
            next++;
            if (!active) { paths++; pathLen += L.depth; if (STAGED || F.qRec) q_finish(F, slot, A); else pending = true; }
        }
        if (active) {
            const InlineTracer tr = {sv, stack};
            if (!bounce<ENV, SMOOTH, PH_ALL, GDPT_UNROLL_OFFSETS(WAVES_PER_SIMD), (WAVES_PER_SIMD > 2)>(S, sv, cfg, tr, L, A)) {
                active = false;
                paths++; pathLen += L.depth;
                // with a queue every sample's sums go to its slot (coalesced, write-only) and k_fold_cont adds them to the pixel once per
                // chunk; without one they wait in A for the lane's next regeneration (`pending`)
                if (STAGED || F.qRec) q_finish(F, slot, A); else pending = true;
            } else if ((STAGED || F.qRec) && all_connected(L, cfg.handoffEarly != 0)) {
                // every offset is connected or dead: the rest of this base path belongs to k_continue (the sums so far travel with it)
                q_store(F, slot, L, A);
                const unsigned long long mask = __ballot(true);
                const int leader = __ffsll((unsigned long long)mask) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&F.qCount[0], (unsigned)__popcll(mask));
                base = __shfl(base, leader);
                F.qList[base + __popcll(mask & ((1ULL << lane) - 1ULL))] = slot;
                active = false;
            }
        }
    }
    if constexpr (!STAGED) { if (pending) finish_path(F, flt, L.sx, L.sy, A, px, py, next - 1 - cfg.sBase); }
    // statistics: wave-level integer reduction, one atomic per wave and counter
    const unsigned c0 = __builtin_amdgcn_wave_reduce_add_u32(L.nClosest, 0), c1 = __builtin_amdgcn_wave_reduce_add_u32(L.nShadow, 0);
    const unsigned c2 = __builtin_amdgcn_wave_reduce_add_u32((unsigned)paths, 0), c3 = __builtin_amdgcn_wave_reduce_add_u32((unsigned)pathLen, 0);
    if (lane == 0) {
        atomicAdd(&F.stats[0], (unsigned long long)c0);
        atomicAdd(&F.stats[1], (unsigned long long)c1);
        atomicAdd(&F.stats[2], (unsigned long long)c2);
        atomicAdd(&F.stats[3], (unsigned long long)c3);
    }
}

// The first stage of a sample in a scene WITHOUT glossy vertices (PH_FIRST above; round 6): start_path from k_primary's hits and exactly ONE bounce --
// five emitter samples, the base path's extension, four reconnections -- then the sample goes to k_continue (its offsets RAY_RECENTLY_CONNECTED or
// dead) or is over.  Against k_render<STAGED> on such a scene: no second bounce (its temporaries never coexist with the first bounce's state), only
// the RAY_NOT_CONNECTED code and only the reconnection shift compiled in, and no regeneration logic -- every lane of a wave is at the same point of
// the same sample index, so the loop over a tile's samples is uniform.  Same work items (tile x sample slice), same slots, same record as k_render.
template <bool LDS_SCENE, bool ACC_LDS, int WAVES_PER_SIMD, bool ENV, bool SMOOTH>
__global__ __launch_bounds__(TBLK, WAVES_PER_SIMD) void k_first(SceneD S, ConfigD cfg, FilmD F, int rx0, int ry0, int rx1, int ry1, int tilesX, int tiles, int slices, int stackDepth)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    SceneView sv;
    int *stack;
    unsigned char *s_acc;
    block_setup<LDS_SCENE, ACC_LDS>(S, stackDepth, s_dyn, sv, stack, s_acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int s0 = cfg.sBase + (int)((long long)cfg.sCount * slice / slices), s1 = cfg.sBase + (int)((long long)cfg.sCount * (slice + 1) / slices);
    const int px = rx0 + tx * 16 + (wave & 1) * 8 + (lane & 7), py = ry0 + ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool valid = px < rx1 && py < ry1;
    unsigned nClosest = 0, nShadow = 0, paths = 0, pathLen = 0;
    Acc<ACC_LDS> A;
    if constexpr (ACC_LDS) A.p = reinterpret_cast<Float *>(s_acc) + threadIdx.x;
#pragma unroll 1
    for (int sample = s0; sample < s1; sample++) {
        if (__hip_atomic_load(F.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;                         // cancelled: no new samples
        bool handOver = false;
        const unsigned slot = (unsigned)(sample - cfg.sBase) * F.qPixels + (unsigned)tile * TBLK + threadIdx.x;
        Lane L;
        if (valid) {
            L.nClosest = L.nShadow = 0;
            bool over = !start_path<ENV, SMOOTH, false, Acc<ACC_LDS>, true>(S, sv, cfg, stack, L, A, px, py, sample, &F, slot);
            if (!over) {
                const InlineTracer tr = {sv, stack};
                over = !bounce<ENV, SMOOTH, PH_FIRST, GDPT_UNROLL_OFFSETS(WAVES_PER_SIMD), (WAVES_PER_SIMD > 2)>(S, sv, cfg, tr, L, A);
                // an offset that is alive and still on its own: the base path's next segment would have been its last and met no emitter (:901) -- the path
                // ends at the depth test of the next bounce (:537), before anything else of that bounce is evaluated
                if (!over && !all_connected(L, true)) over = true;
            }
            nClosest += L.nClosest; nShadow += L.nShadow;
            if (over) { paths++; pathLen += (unsigned)L.depth; q_finish(F, slot, A); }
            else { q_store(F, slot, L, A); handOver = true; }
        }
        const unsigned long long mask = __ballot(handOver);
        if (handOver) {
            const int leader = __ffsll((unsigned long long)mask) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&F.qCount[0], (unsigned)__popcll(mask));
            base = __shfl(base, leader);
            F.qList[base + __popcll(mask & ((1ULL << lane) - 1ULL))] = slot;
        }
    }
    const unsigned c0 = __builtin_amdgcn_wave_reduce_add_u32(nClosest, 0), c1 = __builtin_amdgcn_wave_reduce_add_u32(nShadow, 0);
    const unsigned c2 = __builtin_amdgcn_wave_reduce_add_u32(paths, 0), c3 = __builtin_amdgcn_wave_reduce_add_u32(pathLen, 0);
    if (lane == 0) {
        atomicAdd(&F.stats[0], (unsigned long long)c0);
        atomicAdd(&F.stats[1], (unsigned long long)c1);
        atomicAdd(&F.stats[2], (unsigned long long)c2);
        atomicAdd(&F.stats[3], (unsigned long long)c3);
    }
}

// The five primary rays of every sample of the launch (evaluatePoint, gpt.cpp:397-436: the base ray and the four pixel-shifted ones),
// traced ahead of the render kernel by a kernel that is traversal only: ~70 registers, 5+ waves per SIMD, coherent rays -- inside the
// render kernel the same traversals run at 2 waves per SIMD under 256 registers of path state and were 30 % of its time.
// One thread = one (pixel, sample); blockIdx = tile + tiles * (sample - sBase).
template <bool LDS_SCENE>
__global__ __launch_bounds__(TBLK) void k_primary(SceneD S, ConfigD cfg, FilmD F, int rx0, int ry0, int rx1, int ry1, int tilesX, int tiles, int stackDepth)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    SceneView sv;
    int *stack;
    unsigned char *s_acc;
    block_setup<LDS_SCENE, false>(S, stackDepth, s_dyn, sv, stack, s_acc);
    const int tile = blockIdx.x % tiles, sRel = blockIdx.x / tiles, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int px = rx0 + tx * 16 + (wave & 1) * 8 + (lane & 7), py = ry0 + ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (px >= rx1 || py >= ry1) return;
    Rng rng;
    rng.init(cfg.seed, (uint64_t)py * S.cam.width + px, (uint64_t)(cfg.sBase + sRel));
    const Float sx = px + rng.next1D(), sy = py + rng.next1D();                  // gpt.cpp:1261
    Float apx = 0.5, apy = 0.5;
    if (S.cam.thinlens) { apx = rng.next1D(); apy = rng.next1D(); }               // gpt.cpp:1262-1264
    // (the time sample of gpt.cpp:1265-1267 comes after these draws and is not needed here: k_primary only traces the five primary rays)
    const unsigned slot = (unsigned)sRel * F.qPixels + (unsigned)tile * TBLK + threadIdx.x;
#pragma unroll 1
    for (int r = 0; r < 5; r++) {
        d3 o, d;
        Float mint, maxt;
        const Float ox = r == 1 ? 1.0 : (r == 3 ? -1.0 : 0.0), oy = r == 2 ? 1.0 : (r == 4 ? -1.0 : 0.0);     // gpt.cpp:410-415
        camera_ray(S.cam, sx + ox, sy + oy, apx, apy, o, d, mint, maxt);
        Hit h;
        trace<false>(sv, stack, o, d, ray_mint_closest(o, mint), maxt, h);
        qst(&F.pHit[(size_t)(3 * r) * F.qCapacity + slot], h.t);
        qst(&F.pHit[(size_t)(3 * r + 1) * F.qCapacity + slot], h.u);
        qst(&F.pHit[(size_t)(3 * r + 2) * F.qCapacity + slot], h.v);
        __builtin_nontemporal_store(h.prim, &F.pPrim[(size_t)r * F.qCapacity + slot]);
    }
}

// The continuation kernel: persistent waves that run handed-off base paths (all offsets connected or dead) to their end.  A lane that
// finishes writes the sample's final sums back into its record and marks it; idle lanes take the next entries of the queue together
// (one atomic per wave and refill), so waves stay dense whatever the path lengths are.
// PH: PH_JOINED (the records may hold RAY_RECENTLY_CONNECTED offsets: the early hand-over) or PH_CONN (RAY_CONNECTED only)
template <bool LDS_SCENE, bool ACC_LDS, int WAVES_PER_SIMD, bool ENV, bool SMOOTH, int PH = PH_JOINED>
__global__ __launch_bounds__(TBLK, WAVES_PER_SIMD) void k_continue(SceneD S, ConfigD cfg, FilmD F, int stackDepth, int refillMin)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    SceneView sv;
    int *stack;
    unsigned char *s_acc;
    block_setup<LDS_SCENE, ACC_LDS>(S, stackDepth, s_dyn, sv, stack, s_acc);
    const int lane = threadIdx.x & 63;
    const unsigned total = __hip_atomic_load(&F.qCount[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Lane L;
    L.nClosest = L.nShadow = 0;
    L.depth = 0; L.v.prim = 0;
    Acc<ACC_LDS> A;
    if constexpr (ACC_LDS) A.p = reinterpret_cast<Float *>(s_acc) + threadIdx.x;
    bool active = false, exhausted = false;
    unsigned slot = 0;
    unsigned long long pathLen = 0, paths = 0;
    while (true) {
        const unsigned long long idleMask = __ballot(!active);
        if (!exhausted && (__popcll(idleMask) >= refillMin || idleMask == ~0ULL)) {
            const int leader = __ffsll((unsigned long long)idleMask) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&F.qCount[1], (unsigned)__popcll(idleMask));
            base = __shfl(base, leader);
            if (base + (unsigned)__popcll(idleMask) >= total) exhausted = true;        // (uniform: the queue has been handed out)
            if (!active) {
                const unsigned e = base + (unsigned)__popcll(idleMask & ((1ULL << lane) - 1ULL));
                if (e < total) { slot = F.qList[e]; q_load(F, slot, L, A); active = true; }
            }
        }
        if (__ballot(active) == 0) { if (exhausted) break; continue; }
        const InlineTracer tr = {sv, stack};
        if (active && !bounce<ENV, SMOOTH, PH, true, (WAVES_PER_SIMD > 2)>(S, sv, cfg, tr, L, A)) {
            active = false;
            paths++; pathLen += L.depth;
            q_finish(F, slot, A);
        }
    }
    const unsigned c0 = __builtin_amdgcn_wave_reduce_add_u32(L.nClosest, 0), c1 = __builtin_amdgcn_wave_reduce_add_u32(L.nShadow, 0);
    const unsigned c2 = __builtin_amdgcn_wave_reduce_add_u32((unsigned)paths, 0), c3 = __builtin_amdgcn_wave_reduce_add_u32((unsigned)pathLen, 0);
    if (lane == 0) {
        atomicAdd(&F.stats[0], (unsigned long long)c0);
        atomicAdd(&F.stats[1], (unsigned long long)c1);
        atomicAdd(&F.stats[2], (unsigned long long)c2);
        atomicAdd(&F.stats[3], (unsigned long long)c3);
    }
}

// The base paths of one round of the deferred continuation: persistent waves like k_continue's, one lane = one base path for up to WK bounces.  cnt: [0] entries of
// listIn (written by the producer: the first-stage kernel or the previous round), [1] this kernel's cursor, [2] entries appended to listOut (the samples that are still
// alive after WK bounces: their base state goes back into their queue record).  wInfo[e] = records written | 256 if the path ended.
template <bool LDS_SCENE, int WAVES_PER_SIMD, bool ENV, bool SMOOTH>
__global__ __launch_bounds__(TBLK, WAVES_PER_SIMD) void k_walk(SceneD S, ConfigD cfg, FilmD F, const unsigned *__restrict__ listIn, unsigned *__restrict__ cnt, unsigned *__restrict__ listOut,
                                                               Float *__restrict__ wLog, unsigned *__restrict__ wInfo, int firstRound, int stackDepth, int refillMin)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    SceneView sv;
    int *stack;
    unsigned char *s_acc;
    block_setup<LDS_SCENE, false>(S, stackDepth, s_dyn, sv, stack, s_acc);
    const int lane = threadIdx.x & 63;
    const unsigned total = __hip_atomic_load(&cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Lane L;
    L.nClosest = L.nShadow = 0;
    L.depth = 0; L.v.prim = 0;
    AccNone A;
    bool active = false, exhausted = false;
    unsigned slot = 0, e = 0, recentMask = 0;
    d3 recent[4];                   // the last own vertices of the sample's RAY_RECENTLY_CONNECTED offsets (read in its first bounce here only)
#pragma unroll
    for (int i = 0; i < 4; i++) recent[i] = mk(0.0);
    int nrec = 0;
    unsigned long long pathLen = 0, paths = 0;
    while (true) {
        const unsigned long long idleMask = __ballot(!active);
        if (!exhausted && (__popcll(idleMask) >= refillMin || idleMask == ~0ULL)) {
            const int leader = __ffsll((unsigned long long)idleMask) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&cnt[1], (unsigned)__popcll(idleMask));
            base = __shfl(base, leader);
            if (base + (unsigned)__popcll(idleMask) >= total) exhausted = true;
            if (!active) {
                e = base + (unsigned)__popcll(idleMask & ((1ULL << lane) - 1ULL));
                if (e < total) {
                    slot = listIn[e];
                    q_load_main(F, slot, L);
                    recentMask = firstRound ? (((unsigned)__double_as_longlong(qld(&F.qRec[(size_t)31 * F.qCapacity + slot])) >> 4) & 15u) : 0u;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if ((recentMask >> i) & 1u) {
                            const Float *q = F.qRec + slot;
                            recent[i] = mk(qld(&q[(size_t)(62 + 3 * i) * F.qCapacity]), qld(&q[(size_t)(63 + 3 * i) * F.qCapacity]), qld(&q[(size_t)(64 + 3 * i) * F.qCapacity]));
                        }
                    nrec = 0;
                    active = true;
                }
            }
        }
        if (__ballot(active) == 0) { if (exhausted) break; continue; }
        bool goesOn = false;            // alive after WK bounces: to the next round
        if (active) {
            WalkTracer tr(sv, stack, wLog + e, (size_t)F.qCapacity, nrec, nrec == 0 ? recentMask : 0u, recent);
            const bool go = bounce<ENV, SMOOTH, PH_WALK, true, (WAVES_PER_SIMD > 2)>(S, sv, cfg, tr, L, A);
            if (tr.flags) { tr.put(nrec, 0, __longlong_as_double((long long)tr.flags)); nrec++; }
            if (!go || nrec == WK) {
                __builtin_nontemporal_store((unsigned)nrec | (go ? 0u : 256u), &wInfo[e]);
                if (go) { q_store_main(F, slot, L); goesOn = true; }
                else { paths++; pathLen += L.depth; }
                active = false;
            }
        }
        const unsigned long long onMask = __ballot(goesOn);
        if (onMask) {
            const int leader = __ffsll((unsigned long long)onMask) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&cnt[2], (unsigned)__popcll(onMask));
            base = __shfl(base, leader);
            if (goesOn) listOut[base + __popcll(onMask & ((1ULL << lane) - 1ULL))] = slot;
        }
    }
    const unsigned c0 = __builtin_amdgcn_wave_reduce_add_u32(L.nClosest, 0), c1 = __builtin_amdgcn_wave_reduce_add_u32(L.nShadow, 0);
    const unsigned c2 = __builtin_amdgcn_wave_reduce_add_u32((unsigned)paths, 0), c3 = __builtin_amdgcn_wave_reduce_add_u32((unsigned)pathLen, 0);
    if (lane == 0) {
        atomicAdd(&F.stats[0], (unsigned long long)c0);
        atomicAdd(&F.stats[1], (unsigned long long)c1);
        atomicAdd(&F.stats[2], (unsigned long long)c2);
        atomicAdd(&F.stats[3], (unsigned long long)c3);
    }
}

// The offsets and sums of one round: one lane per entry of the round's list (consecutive lanes = consecutive entries: every log read is coalesced) applies the entry's
// bounce records in order -- the emitter-sample half for offsets 0..3, the BSDF-sample half for offsets 0..3, the roulette factor: the order bounce() itself has --
// to the sample's four offsets and its sums, then writes the final sums (the path ended: k_fold_cont takes them) or the offsets and sums back (next round).  Arithmetic
// only: the one thing a joined offset needs from the scene, the RecentTerms of the first bounce, is in the log.
#ifndef GDPT_RENDER_DEVICE_FUNCTIONS_ONLY      /* (a plain kernel: gpt_capi.hip's alone, like the ones further down) */
#ifndef GDPT_REPLAY_WPS
#define GDPT_REPLAY_WPS 2
#endif
__global__ __launch_bounds__(TBLK, GDPT_REPLAY_WPS) void k_replay(FilmD F, const unsigned *__restrict__ listIn, const unsigned *__restrict__ cnt, const Float *__restrict__ wLog,
                                                              const unsigned *__restrict__ wInfo)
{
    const unsigned total = __hip_atomic_load(&cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const size_t cap = F.qCapacity;
    for (unsigned e = blockIdx.x * TBLK + threadIdx.x; e < total; e += gridDim.x * TBLK) {
        const unsigned slot = listIn[e], info = __builtin_nontemporal_load(&wInfo[e]);
        const int nrec = (int)(info & 255u);
        const bool over = (info >> 8) != 0;
        Float *q = F.qRec + slot;
        const Float *lg = wLog + e;
        auto ldf = [&](int field) -> Float { return qld(&lg[(size_t)field * cap]); };
        auto ld = [&](int rec, int field) -> Float { return ldf(rec * WF + field); };
        auto ld3 = [&](int rec, int field) -> d3 { return mk(ld(rec, field), ld(rec, field + 1), ld(rec, field + 2)); };
        Offset off[4];
        Acc<false> A;                                                    // (the veryDirect sums stay where the first stage left them: slots ACC_VD are neither read nor written)
        const unsigned alive = (unsigned)__double_as_longlong(qld(&q[31 * cap]));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            Offset &o = off[i];
            o.throughput = mk(qld(&q[(15 + 4 * i) * cap]), qld(&q[(16 + 4 * i) * cap]), qld(&q[(17 + 4 * i) * cap])); o.pdf = qld(&q[(18 + 4 * i) * cap]);
            o.alive = (alive >> i) & 1; o.status = ((alive >> (4 + i)) & 1) ? RAY_RECENTLY_CONNECTED : RAY_CONNECTED;
        }
#pragma unroll
        for (int k = 0; k < ACC_N; k++) A.a[k] = (k >= ACC_VD && k < ACC_VD + 3) ? 0.0 : qld(&q[(32 + k) * cap]);
        for (int k = 0; k < nrec; k++) {
            // (the whole record in one go: its flags and both halves' fields are independent loads -- a half that was not logged holds stale doubles that are not used)
            const unsigned flags = (unsigned)__double_as_longlong(ld(k, 0));
            NeeShared n;
            n.dRecPdf = ld(k, 1); n.mainBsdfPdf = ld(k, 2); n.num = ld(k, 3); n.den = ld(k, 4); n.X = ld3(k, 5); n.contribAll = ld3(k, 8);
            BsdfShared m;
            m.W = ld3(k, 11); m.mainBsdfPdf = ld(k, 14); m.lumPdf = ld(k, 15); m.num = ld(k, 16); m.den = ld(k, 17); m.radiance = ld3(k, 18); m.contrib = ld3(k, 21);
            const Float qf = ld(k, 24);
            if (flags & 1u) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    RecentTerm rt; rt.f = mk(0.0); rt.pdf = 0.0;           // (read where it is used, in the sample's first bounce after the hand-over only)
                    if (off[i].status == RAY_RECENTLY_CONNECTED && off[i].alive) { rt.f = mk(ldf(WK * WF + 8 * i + 0), ldf(WK * WF + 8 * i + 1), ldf(WK * WF + 8 * i + 2)); rt.pdf = ldf(WK * WF + 8 * i + 3); }
                    nee_offset_joined(n, rt, off[i].status, off[i], i, A);
                }
            }
            if (flags & 2u) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    RecentTerm rt; rt.f = mk(0.0); rt.pdf = 0.0;
                    if (off[i].status == RAY_RECENTLY_CONNECTED && off[i].alive) { rt.f = mk(ldf(WK * WF + 8 * i + 4), ldf(WK * WF + 8 * i + 5), ldf(WK * WF + 8 * i + 6)); rt.pdf = ldf(WK * WF + 8 * i + 7); }
                    bsdf_offset_joined(m, rt, off[i], i, A);
                }
            }
            if (flags & 4u) {
#pragma unroll
                for (int i = 0; i < 4; i++) off[i].pdf *= qf;
            }
        }
#pragma unroll
        for (int k = 0; k < ACC_N; k++) if (!(k >= ACC_VD && k < ACC_VD + 3)) qst(&q[(32 + k) * cap], A.a[k]);
        if (over) qst(&q[13 * cap], __longlong_as_double((long long)Q_DONE));
        else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const Offset &o = off[i];
                qst(&q[(15 + 4 * i) * cap], o.throughput.x); qst(&q[(16 + 4 * i) * cap], o.throughput.y); qst(&q[(17 + 4 * i) * cap], o.throughput.z); qst(&q[(18 + 4 * i) * cap], o.pdf);
            }
            qst(&q[31 * cap], __longlong_as_double((long long)(alive & 15u)));
        }
    }
}

#endif // GDPT_RENDER_DEVICE_FUNCTIONS_ONLY (k_replay)

#ifndef GDPT_RENDER_DEVICE_FUNCTIONS_ONLY      /* (gpt_wave_capi.hip takes the device functions and kernel templates above; the plain kernels below belong to gpt_capi.hip) */
// finish_path for the samples of a chunk, all of which left their final sums in their queue slots (from the render kernel or from
// k_continue): one thread per pixel of the launch adds them in sample order -- a fixed association, so a render is reproducible bit for
// bit -- in registers, and touches the pixel's record once.  Reads are coalesced ([component][slot], consecutive lanes = consecutive slots).
__global__ __launch_bounds__(TBLK) void k_fold_cont(SceneD S, ConfigD cfg, FilmD F, int rx0, int ry0, int rx1, int ry1, int tilesX)
{
    const int tile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int px = rx0 + tx * 16 + (wave & 1) * 8 + (lane & 7), py = ry0 + ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (px >= rx1 || py >= ry1) return;
    const FilterD flt = box_filter();
    const size_t st = F.qCapacity;
    Float P[NREC];
#pragma unroll
    for (int k = 0; k < NREC; k++) P[k] = 0.0;
    for (int s = 0; s < cfg.sCount; s++) {
        const unsigned slot = (unsigned)s * F.qPixels + (unsigned)tile * TBLK + threadIdx.x;
        const Float *q = F.qRec + slot;
        if ((unsigned long long)__double_as_longlong(qld(&q[13 * st])) != Q_DONE) continue;         // (a cancelled frame: the sample was never started)
        Acc<false> A;
#pragma unroll
        for (int k = 0; k < ACC_N; k++) A.a[k] = qld(&q[(32 + k) * st]);
        Rng rng;
        rng.init(cfg.seed, (uint64_t)py * S.cam.width + px, (uint64_t)(cfg.sBase + s));
        const Float sx = px + rng.next1D(), sy = py + rng.next1D();              // the sample's position, as start_path drew it (gpt.cpp:1261)
        finish_path(F, flt, sx, sy, A, px, py, s, P);
    }
    if (P[0] != 0.0) {
        Float *r = F.rec + (size_t)(py - (F.y0 - 1)) * F.W + px;
#pragma unroll
        for (int k = 0; k < NREC; k++) r[(size_t)k * F.recStride] += P[k];
    }
}

// Wider reconstruction filters: the 15 puts of gpt.cpp:1314-1352 for the `count` logged samples of every pixel, evaluated from
// the side of the receiving pixel.  One thread = one output pixel: it visits the pixels within reach (filter radius + the one-pixel
// shift of the neighbour puts), reads each sample's position, forms the weights of the five put positions from the discretised
// filter (ImageBlock::put, imageblock.h:150-199: w = wx * wy; beyond the radius the table gives 0) and adds into its own 5 x 4
// sums in (row, column, sample) order -- no atomics, reproducible.  Adjacent threads read adjacent log entries.
// [sx0, sx1) x [sy0, sy1): the pixels whose samples this chunk logged -- the film's rows and their reach (a whole strip), or a block of the film (round 5:
// gdpt_render_rect on a sub-rectangle, GPTBlockRenderer's unit: the block's samples land within the filter's reach AROUND the block, gpt_wr.cpp:31-44,
// and blocks add up); the receiving pixels [ox0, ..) x [oy0, ..) are the film's pixels within reach of those.
__global__ __launch_bounds__(TBLK) void k_gather_log(FilmD F, int count, int sx0, int sy0, int sx1, int sy1, int ox0, int oy0, int ox1, int oy1)
{
    enum { RIGHT = 0, BOTTOM = 1, LEFT = 2, TOP = 3 };
    const int x = ox0 + blockIdx.x * 16 + (threadIdx.x & 15), y = oy0 + blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= ox1 || y >= oy1) return;
    if (__hip_atomic_load(F.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;      // a cancelled chunk is incomplete: dropped
    const int R = (int)ceil(F.fRadius) + 1;                 // (the host sizes logY0 / logRows with the same reach)
    const size_t plane = (size_t)F.logRows * F.W, comp = (size_t)F.logChunk * plane;
    auto evalD = [&](Float d) -> Float { int idx = (int)fabs(d * F.fScale); if (idx > 31) idx = 31; return F.fValues[idx]; };
    Float o[5][4];
    unsigned invalid = 0;
    for (int b = 0; b < 5; b++) for (int k = 0; k < 4; k++) o[b][k] = 0.0;
    for (int yy = max(sy0, y - R); yy <= min(sy1 - 1, y + R); yy++)
        for (int xx = max(sx0, x - R); xx <= min(sx1 - 1, x + R); xx++)
            for (int c = 0; c < count; c++) {
                const size_t at = (size_t)c * plane + (size_t)(yy - F.logY0) * F.W + xx;
                const Float sx = F.log[(size_t)30 * comp + at], sy = F.log[(size_t)31 * comp + at];
                // distances of this pixel from the put positions (pos = sample - 0.5; neighbour puts one pixel away)
                const Float dx0 = x - (sx - 0.5), dy0 = y - (sy - 0.5);
                const Float wx0 = evalD(dx0), wxm = evalD(dx0 + 1.0), wxp = evalD(dx0 - 1.0);     // puts at sx, sx - 1, sx + 1
                const Float wy0 = evalD(dy0), wym = evalD(dy0 + 1.0), wyp = evalD(dy0 - 1.0);
                const Float w0 = wx0 * wy0, wL = wxm * wy0, wR = wxp * wy0, wT = wx0 * wym, wB = wx0 * wyp;
                const bool own = xx == x && yy == y;
                if (!own && w0 == 0 && wL == 0 && wR == 0 && wT == 0 && wB == 0) continue;
                auto get3 = [&](int k0) -> d3 { return mk(F.log[(size_t)k0 * comp + at], F.log[(size_t)(k0 + 1) * comp + at], F.log[(size_t)(k0 + 2) * comp + at]); };
                const d3 T = get3(ACC_T), vd = get3(ACC_VD);
                const d3 nL = 2 * get3(ACC_NBR + 3 * LEFT), nR = 2 * get3(ACC_NBR + 3 * RIGHT), nT = 2 * get3(ACC_NBR + 3 * TOP), nB = 2 * get3(ACC_NBR + 3 * BOTTOM);
                auto put = [&](int b, Float w, d3 spec, Float weight) {
                    if (!put_valid(spec, b)) { if (own) invalid++; return; }      // ImageBlock::put drops an invalid put whole; counted once, by the sample's own pixel
                    o[b][0] += w * spec.x; o[b][1] += w * spec.y; o[b][2] += w * spec.z; o[b][3] += w * weight;
                };
                put(0, w0, (8 * vd) + (2 * T), 4.0); put(0, wL, nL, 1.0); put(0, wR, nR, 1.0); put(0, wT, nT, 1.0); put(0, wB, nB, 1.0);
                put(1, w0, 2 * T, 4.0);              put(1, wL, nL, 1.0); put(1, wR, nR, 1.0); put(1, wT, nT, 1.0); put(1, wB, nB, 1.0);
                put(2, wL, -(2 * get3(ACC_GRAD + 3 * LEFT)), 1.0); put(2, w0, 2 * get3(ACC_GRAD + 3 * RIGHT), 1.0);
                put(3, wT, -(2 * get3(ACC_GRAD + 3 * TOP)), 1.0);  put(3, w0, 2 * get3(ACC_GRAD + 3 * BOTTOM), 1.0);
                put(4, w0, vd, 1.0);
            }
    for (int b = 0; b < 5; b++)
        for (int k = 0; k < 4; k++) F.spill[(((size_t)b * F.spillRows + (y - (F.y0 - 2))) * F.W + x) * 4 + k] += o[b][k];
    if (invalid) atomicAdd(&F.stats[4], (unsigned long long)invalid);
}

// rec += the slice planes, in slice order (a fixed association, so a render is reproducible bit for bit), and clear them
__global__ __launch_bounds__(TBLK) void k_fold_slices(FilmD F, int slices)
{
    const size_t n = (size_t)NREC * F.recStride;
    for (size_t i = (size_t)blockIdx.x * TBLK + threadIdx.x; i < n; i += (size_t)gridDim.x * TBLK) {
        Float v = F.rec[i];
        for (int s = 0; s < slices - 1; s++) { v += F.recExtra[(size_t)s * n + i]; F.recExtra[(size_t)s * n + i] = 0.0; }
        F.rec[i] = v;
    }
}

// ---- resolve: per-pixel sums -> the five accumulation buffers, as 15 puts per sample would have produced -------------
// out[5][rows][W][4] (R,G,B,weight), rows = y1 - y0.
__global__ __launch_bounds__(TBLK) void k_resolve(FilmD F, Float *__restrict__ out)
{
    const FilterD flt = box_filter();
    const Float w = flt.c * flt.c;                        // weightX * weightY of imageblock.h:186-191
    const int rows = F.y1 - F.y0;
    const int n = rows * F.W;
    for (int i = blockIdx.x * TBLK + threadIdx.x; i < n; i += gridDim.x * TBLK) {
        const int x = i % F.W, y = F.y0 + i / F.W;
        const size_t st = F.recStride;
        auto rec = [&](int xx, int yy, int k) -> Float {
            if (xx < 0 || xx >= F.W || yy < 0 || yy >= F.H) return 0.0;
            return F.rec[(size_t)k * st + (size_t)(yy - (F.y0 - 1)) * F.W + xx];
        };
        const Float cnt = rec(x, y, 0);
        const Float cL = rec(x - 1, y, 0), cR = rec(x + 1, y, 0), cT = rec(x, y - 1, 0), cB = rec(x, y + 1, 0);
        Float o[5][4];
        for (int c = 0; c < 3; c++) {
            const Float T = rec(x, y, 1 + c), vd = rec(x, y, 4 + c);
            // neighbour throughput landing here: sample at x-1 shifts RIGHT(0); at x+1 LEFT(2); at y-1 BOTTOM(1); at y+1 TOP(3)
            const Float nb = 2 * rec(x - 1, y, 7 + 0 + c) + 2 * rec(x + 1, y, 7 + 6 + c) + 2 * rec(x, y - 1, 7 + 3 + c) + 2 * rec(x, y + 1, 7 + 9 + c);
            o[0][c] = w * ((8 * vd + 2 * T) + nb);
            o[1][c] = w * (2 * T + nb);
            o[2][c] = w * (2 * rec(x, y, 19 + 0 + c) - 2 * rec(x + 1, y, 19 + 6 + c));      // +2 g_RIGHT here, -2 g_LEFT of the sample at x+1
            o[3][c] = w * (2 * rec(x, y, 19 + 3 + c) - 2 * rec(x, y + 1, 19 + 9 + c));      // +2 g_BOTTOM here, -2 g_TOP of the sample at y+1
            o[4][c] = w * vd;
        }
        o[0][3] = o[1][3] = w * (4 * cnt + cL + cR + cT + cB);
        o[2][3] = w * (cnt + cR);
        o[3][3] = w * (cnt + cB);
        o[4][3] = w * cnt;
        for (int b = 0; b < 5; b++)
            for (int k = 0; k < 4; k++) {
                const size_t si = (((size_t)b * F.spillRows + (y - (F.y0 - 2))) * F.W + x) * 4 + k;
                out[(((size_t)b * rows + (y - F.y0)) * F.W + x) * 4 + k] = o[b][k] + F.spill[si];
            }
    }
}

// MultiFilm::developMulti: rgb * (w != 0 ? 1/w : w) (fmtconv.cpp:955-1058), cast to fp32 (gpt.cpp:1439-1442)
__global__ __launch_bounds__(TBLK) void k_develop(const Float *__restrict__ accumBuf /* [n][4] */, float *__restrict__ rgb, int n)
{
    for (int i = blockIdx.x * TBLK + threadIdx.x; i < n; i += gridDim.x * TBLK) {
        const Float wgt = accumBuf[4 * (size_t)i + 3], inv = (wgt != 0) ? 1.0 / wgt : wgt;
        for (int c = 0; c < 3; c++) rgb[3 * (size_t)i + c] = (float)(accumBuf[4 * (size_t)i + c] * inv);
    }
}

// halo exchange payload: [NREC][W] records of an owned boundary row, then [2][5][W][4]: the spill of the two halo rows beyond it (near, far)
__global__ __launch_bounds__(TBLK) void k_pack_halo(FilmD F, int which, Float *__restrict__ buf)
{
    const int ownRow = which == 0 ? F.y0 : F.y1 - 1, step = which == 0 ? -1 : 1;       // the halo rows: ownRow + step (near), ownRow + 2 * step (far)
    const int n1 = NREC * F.W, n2 = 5 * F.W * 4;
    for (int i = blockIdx.x * TBLK + threadIdx.x; i < n1 + 2 * n2; i += gridDim.x * TBLK) {
        if (i < n1) { const int k = i / F.W, x = i % F.W; buf[i] = F.rec[(size_t)k * F.recStride + (size_t)(ownRow - (F.y0 - 1)) * F.W + x]; }
        else {
            const int far = (i - n1) / n2, j = (i - n1) % n2, b = j / (F.W * 4), r = j % (F.W * 4);
            buf[i] = F.spill[((size_t)b * F.spillRows + (ownRow + (1 + far) * step - (F.y0 - 2))) * F.W * 4 + r];
        }
    }
}
// receive from the neighbour on side `which`: its boundary-row records become my halo row; its spill of my boundary row and of the row inside it is added
__global__ __launch_bounds__(TBLK) void k_unpack_halo(FilmD F, int which, const Float *__restrict__ buf)
{
    const int ownRow = which == 0 ? F.y0 : F.y1 - 1, haloRow = which == 0 ? F.y0 - 1 : F.y1, step = which == 0 ? 1 : -1;   // the neighbour's near row is my boundary row, its far row the one inside it
    const int n1 = NREC * F.W, n2 = 5 * F.W * 4;
    for (int i = blockIdx.x * TBLK + threadIdx.x; i < n1 + 2 * n2; i += gridDim.x * TBLK) {
        if (i < n1) { const int k = i / F.W, x = i % F.W; F.rec[(size_t)k * F.recStride + (size_t)(haloRow - (F.y0 - 1)) * F.W + x] = buf[i]; }
        else {
            const int far = (i - n1) / n2, j = (i - n1) % n2, b = j / (F.W * 4), r = j % (F.W * 4);
            const int row = ownRow + far * step;
            if (row >= F.y0 && row < F.y1) F.spill[((size_t)b * F.spillRows + (row - (F.y0 - 2))) * F.W * 4 + r] += buf[i];     // (a one-row strip has no row inside its boundary row)
        }
    }
}

// probe: closest hit of arbitrary rays (tests)
__global__ __launch_bounds__(TBLK) void k_intersect(SceneD S, int n, const Float *__restrict__ od, int *__restrict__ prim, Float *__restrict__ tp)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    SceneView sv;
    sv.nodes = S.nodes; sv.isect = S.isect; sv.shade = S.shade; sv.mats = S.mats; sv.emitters = S.emitters; sv.emTris = S.emTris; sv.emCdf = S.emCdf; sv.emitterCdf = S.emitterCdf; sv.rootRef = S.rootRef; sv.boundM = S.boundM; sv.quant = S.quantNodes; sv.leafExit = 1; sv.vn = S.vn; sv.uv = S.uv; sv.hasUV = S.hasUV; sv.tex = S.tex;
    const int i = blockIdx.x * TBLK + threadIdx.x;
    if (i >= n) return;
    const d3 o = mk(od[6 * i], od[6 * i + 1], od[6 * i + 2]), d = mk(od[6 * i + 3], od[6 * i + 4], od[6 * i + 5]);
    Hit h;
    trace<false>(sv, s_stack + threadIdx.x, o, d, ray_mint_closest(o, GD_EPSILON), GD_INF, h);
    Vertex v;
    fill_vertex(sv, h, d, v);
    prim[i] = h.prim < 0 ? -1 : sv.shade[h.prim].origIndex;
    tp[4 * i] = h.t;
    tp[4 * i + 1] = h.prim < 0 ? 0.0 : v.p.x; tp[4 * i + 2] = h.prim < 0 ? 0.0 : v.p.y; tp[4 * i + 3] = h.prim < 0 ? 0.0 : v.p.z;
}

// probe: the filled intersection record of arbitrary rays (fillIntersectionRecord<true>, skdtree.h:343-428) as the render kernels form it -- position by
// fill_vertex, frames by shading_at, texture coordinates as reflectance_at interpolates them, dpdu / dpdv by tri_partials.  What the reference's own
// src/tests/test_dgeom.cpp:35-178 asserts on.  rec24 per ray: t, p(3), uv(2), geoFrame.n(3), shFrame.n(3), shFrame.s(3), dpdu(3), dpdv(3), wi(3).
__global__ __launch_bounds__(TBLK) void k_intersect_record(SceneD S, int n, const Float *__restrict__ od, int *__restrict__ prim, Float *__restrict__ rec24)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    SceneView sv;
    sv.nodes = S.nodes; sv.isect = S.isect; sv.shade = S.shade; sv.mats = S.mats; sv.emitters = S.emitters; sv.emTris = S.emTris; sv.emCdf = S.emCdf; sv.emitterCdf = S.emitterCdf; sv.rootRef = S.rootRef; sv.boundM = S.boundM; sv.quant = S.quantNodes; sv.leafExit = 1; sv.vn = S.vn; sv.uv = S.uv; sv.hasUV = S.hasUV; sv.tex = S.tex;
    const int i = blockIdx.x * TBLK + threadIdx.x;
    if (i >= n) return;
    const d3 o = mk(od[6 * i], od[6 * i + 1], od[6 * i + 2]), d = mk(od[6 * i + 3], od[6 * i + 4], od[6 * i + 5]);
    Hit h;
    trace<false>(sv, s_stack + threadIdx.x, o, d, ray_mint_closest(o, GD_EPSILON), GD_INF, h);
    Float *r = rec24 + (size_t)24 * i;
    prim[i] = h.prim < 0 ? -1 : sv.shade[h.prim].origIndex;
    for (int k = 0; k < 24; k++) r[k] = 0.0;
    r[0] = h.t;
    if (h.prim < 0) return;
    Vertex v;
    fill_vertex(sv, h, d, v);
    const Shading sh = shading_at<true>(sv, v);
    Float tu = v.u, tv = v.v;                                                    // its.uv, skdtree.h:398-405 (as reflectance_at)
    if (sv.uv && sv.hasUV[v.prim]) {
        const TriUV t = sv.uv[v.prim];
        const Float b0 = 1 - v.u - v.v;
        tu = t.uv[0] * b0 + t.uv[2] * v.u + t.uv[4] * v.v;
        tv = t.uv[1] * b0 + t.uv[3] * v.u + t.uv[5] * v.v;
    }
    d3 dpdu, dpdv;
    tri_partials(sv, v.prim, dpdu, dpdv);
    const d3 wi = toLocal(sh.fr, -d);
    r[1] = v.p.x; r[2] = v.p.y; r[3] = v.p.z; r[4] = tu; r[5] = tv;
    const d3 out[6] = {sh.geoN, sh.fr.n, sh.fr.s, dpdu, dpdv, wi};
    for (int k = 0; k < 6; k++) { r[6 + 3 * k] = out[k].x; r[7 + 3 * k] = out[k].y; r[8 + 3 * k] = out[k].z; }
}

// probe: traversal statistics -- inner nodes fetched and triangles tested, closest-hit and any-hit, summed over n rays (od: origin,
// direction; maxt = infinity).  sums[0..3] = nodes (closest), tris (closest), nodes (any), tris (any).
__global__ __launch_bounds__(TBLK) void k_trace_stats(SceneD S, int n, const Float *__restrict__ od, unsigned long long *__restrict__ sums)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    SceneView sv;
    sv.nodes = S.nodes; sv.isect = S.isect; sv.shade = S.shade; sv.mats = S.mats; sv.emitters = S.emitters; sv.emTris = S.emTris; sv.emCdf = S.emCdf; sv.emitterCdf = S.emitterCdf; sv.rootRef = S.rootRef; sv.boundM = S.boundM; sv.quant = S.quantNodes; sv.leafExit = 1; sv.vn = S.vn; sv.uv = S.uv; sv.hasUV = S.hasUV; sv.tex = S.tex;
    const int i = blockIdx.x * TBLK + threadIdx.x;
    TravCount c0 = {0, 0}, c1 = {0, 0};
    if (i < n) {
        const d3 o = mk(od[6 * i], od[6 * i + 1], od[6 * i + 2]), d = mk(od[6 * i + 3], od[6 * i + 4], od[6 * i + 5]);
        Hit h;
        trace<false, true>(sv, s_stack + threadIdx.x, o, d, ray_mint_closest(o, GD_EPSILON), GD_INF, h, &c0);
        trace<true, true>(sv, s_stack + threadIdx.x, o, d, ray_mint_shadow(o, GD_EPSILON), GD_INF, h, &c1);
    }
    const unsigned v[4] = {c0.nodes, c0.tris, c1.nodes, c1.tris};
    for (int k = 0; k < 4; k++) {
        const unsigned w = __builtin_amdgcn_wave_reduce_add_u32(v[k], 0);
        if ((threadIdx.x & 63) == 0) atomicAdd(&sums[k], (unsigned long long)w);
    }
}

// probe: the raw outputs of evaluatePoint (gpt.cpp:397-436) for one (pixel, sample): veryDirect(3), throughput(3),
// gradients[4](12), neighbourThroughputs[4](12), then closest/shadow ray counts and the final depth as doubles.
__global__ __launch_bounds__(TBLK) void k_eval_point(SceneD S, ConfigD cfg, int px, int py, int sample, Float *__restrict__ out33)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    SceneView sv;
    sv.nodes = S.nodes; sv.isect = S.isect; sv.shade = S.shade; sv.mats = S.mats; sv.emitters = S.emitters; sv.emTris = S.emTris; sv.emCdf = S.emCdf; sv.emitterCdf = S.emitterCdf; sv.rootRef = S.rootRef; sv.boundM = S.boundM; sv.quant = S.quantNodes; sv.leafExit = 1; sv.vn = S.vn; sv.uv = S.uv; sv.hasUV = S.hasUV; sv.tex = S.tex;
    Lane L;
    L.nClosest = L.nShadow = 0;
    Acc<false> A;
    bool active = start_path<true, true, true>(S, sv, cfg, s_stack, L, A, px, py, sample);
    const InlineTracer tr = {sv, s_stack};
    while (active) active = bounce<true, true, PH_ALL, false, false>(S, sv, cfg, tr, L, A);
    Float *o = out33;
    for (int k = 0; k < 3; k++) *o++ = A.a[ACC_VD + k];
    for (int k = 0; k < 3; k++) *o++ = A.a[ACC_T + k];
    for (int k = 0; k < 12; k++) *o++ = A.a[ACC_GRAD + k];
    for (int k = 0; k < 12; k++) *o++ = A.a[ACC_NBR + k];
    *o++ = (Float)L.nClosest; *o++ = (Float)L.nShadow; *o++ = (Float)L.depth;
}

// probe: the device BSDF models on their own (tests: chi-square of sample() against pdf(), oracle parity per direction)
__global__ __launch_bounds__(TBLK) void k_bsdf_probe(MaterialD m, d3 wi, int nSamples, int nDirs, int measure, const Float *__restrict__ in, Float *__restrict__ out)
{
    const int i = blockIdx.x * TBLK + threadIdx.x;
    if (i < nSamples) {
        BSDFSample r;
        bsdf_sample(m, m.reflectance, wi, in[2 * i], in[2 * i + 1], r);
        Float *o = out + (size_t)8 * i;
        o[0] = r.wo.x; o[1] = r.wo.y; o[2] = r.wo.z; o[3] = r.weight.x; o[4] = r.weight.y; o[5] = r.weight.z; o[6] = r.pdf; o[7] = (Float)r.sampledType;
    } else if (i < nSamples + nDirs) {
        const int j = i - nSamples;
        const Float *w = in + (size_t)2 * nSamples + (size_t)3 * j;
        d3 f; Float pdf;
        bsdf_eval_pdf(m, m.reflectance, wi, mk(w[0], w[1], w[2]), measure, f, pdf);
        Float *o = out + (size_t)8 * nSamples + (size_t)4 * j;
        o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = pdf;
    }
}

#endif // GDPT_RENDER_DEVICE_FUNCTIONS_ONLY

} // namespace gdpt_tr
