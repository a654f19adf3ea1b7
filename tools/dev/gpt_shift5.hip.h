// gpt_shift5.hip.h -- the shift stage of the staged G-PT pipeline with ONE PATH PER LANE.
//
// k_render<STAGED> walks a sample in one lane: the base path and its four offset paths side by side, the per-offset parts of a bounce
// expanded four times.  That is 62 doubles of path state plus the temporaries of five paths per lane: the 2-wave builds spill ~1.2 KB
// per lane, the 4-wave builds keep the whole state in scratch by design (DESIGN.md "Where the time goes").  k_shift5 gives every path its
// own lane: a wave holds twelve samples, lanes [12 r, 12 r + 12) carry role r of them (r = 0: the base path, r = 1..4: offset path r - 1;
// lanes 60..63 idle), so a lane keeps ONE RayState (gpt.cpp:135-173) and one bounce's temporaries of ONE path.  What an offset path needs
// of its base path each bounce (the emitter sample, the BSDF sample, the new base vertex, the MIS terms) goes through a per-wave mailbox
// in LDS; the shadow rays of a bounce's emitter sampling -- the base path's and the four offsets' -- are traced in one pass.
//
// The arithmetic is bounce()'s (gpt_render.hip.h), statement by statement, with the same random numbers in the same order; the throughput
// sum of a sample is accumulated by its base lane from the four offsets' weights in offset order, so a sample's sums are bit-identical to
// k_render<STAGED>'s and k_continue / k_fold_cont take over unchanged.
#pragma once
#include "../../gradientdomain-mitsuba_amd/csrc/gpt_render.hip.h"

namespace gdpt_tr {

constexpr int S5_K = 12;            // samples per wave
// mailbox fields (doubles; [field][sample of the wave])
enum {
    // written at the top of a bounce, valid to its end
    MB_FR = 0,                      // base shading frame s, t, n (9)
    MB_R = 9,                       // reflectance of the base BSDF at the base vertex (3)
    MB_P = 12,                      // base vertex position (3): its.p of this bounce / previousMainIts.p after the BSDF sample
    MB_MAT = 15,                    // base material index
    // emitter sampling
    MB_LS = 16,                     // the emitter sample (2)
    MB_DP = 18, MB_DN = 21,         // dRec.p, dRec.n (3 + 3)
    MB_DD = 24,                     // dRec.d (3)
    MB_DPDF = 27,                   // dRec.pdf
    MB_BE = 28,                     // mainBSDFValue * mainEmitterRadiance (3)
    MB_ERAD = 31,                   // mainEmitterRadiance (3)
    MB_BPDF = 34,                   // mainBsdfPdf (0 unless the light is on a surface and visible)
    MB_NUM = 35, MB_DEN = 36,       // mainWeightNumerator, mainWeightDenominator
    MB_CALL = 37,                   // mainContributionAll / mainContribution (3)
    MB_DIST2 = 40, MB_OPCOS = 41,   // mainDistanceSquared, mainOpposingCosine
    // BSDF sampling
    MB_WO = 42,                     // bs.wo (local, 3)
    MB_WI = 45,                     // wi of the base path at the base vertex (local, 3)
    MB_BWP = 48,                    // bs.weight * bs.pdf (3)
    MB_BSPDF = 51,                  // bs.pdf
    MB_LUM = 52,                    // mainLumPdf
    MB_NP = 53,                     // new base vertex (3)
    MB_RD = 56,                     // base ray direction of this segment (world, 3)
    MB_NGN = 59,                    // geometric normal at the new base vertex (3)
    MB_NPRIM = 62,                  // new base triangle
    MB_LPDF = 63,                   // base pdf after the BSDF sample (L.pdf)
    MB_Q = 64,                      // Russian-roulette survival probability (1 if none was played)
    MB_W = 65,                      // the four offsets' weights of the current accumulation (4); NaN-free: 0 when nothing was assigned
    MB_ASG = 69,                    // ... and whether a contribution was assigned (4)
    MB_GN = 73,                     // geometric normal at the base vertex (3)
    MB_VAL = 76,                    // the emitter sample's value before the shadow ray (3)
    MB_DDIST = 79,                  // dRec.dist
    MB_N = 80
};

struct P5 {                         // one path: RayState of gpt.cpp:135-173 (base: alive = 1, status unused)
    d3 throughput;
    Float pdf;
    Vertex v;
    d3 rayD;
    int alive, status;
};

__device__ __forceinline__ void wave_sync5()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave = twelve samples x five paths.  Work: the launch's sample slots (k_primary's numbering: slot = sample * qPixels + tile * 256 +
// pixel of the tile) in equal contiguous ranges per wave; a wave's twelve sample seats are refilled together when `regen` of them are idle.
template <bool LDS_SCENE, int WAVES_PER_SIMD, bool ENV, bool SMOOTH>
__global__ __launch_bounds__(TBLK, WAVES_PER_SIMD) void k_shift5(SceneD S, ConfigD cfg, FilmD F, int rx0, int ry0, int rx1, int ry1, int tilesX, int tiles, int stackDepth, int regen)
{
    constexpr bool INL = WAVES_PER_SIMD > 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    SceneView sv;
    int *stack;
    unsigned char *s_box;
    block_setup<LDS_SCENE, false>(S, stackDepth, s_dyn, sv, stack, s_box, sizeof(Float) * MB_N * S5_K * (TBLK / 64));

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = lane / S5_K, k = lane - role * S5_K;          // role 5 (lanes 60..63): no path
    const bool seat = role < 5;
    const bool isBase = role == 0;
    const int oi = role - 1;                                       // offset index of this lane
    Float *mb = reinterpret_cast<Float *>(s_box) + (size_t)wave * MB_N * S5_K + k;
    auto put = [&](int f, Float v) { mb[f * S5_K] = v; };
    auto get = [&](int f) -> Float { return mb[f * S5_K]; };
    auto put3 = [&](int f, d3 v) { mb[f * S5_K] = v.x; mb[(f + 1) * S5_K] = v.y; mb[(f + 2) * S5_K] = v.z; };
    auto get3 = [&](int f) -> d3 { return mk(mb[f * S5_K], mb[(f + 1) * S5_K], mb[(f + 2) * S5_K]); };
    // a flag of the base lane of this lane's sample, seen by all five of its lanes (converged code only)
    auto of_base = [&](bool c) -> bool { return (__ballot(isBase && c) >> k) & 1ULL; };

    // the wave's range of sample slots
    const unsigned long long totalSlots = (unsigned long long)cfg.sCount * F.qPixels;
    const unsigned nWaves = gridDim.x * (TBLK / 64), waveId = blockIdx.x * (TBLK / 64) + wave;
    unsigned next = (unsigned)(totalSlots * waveId / nWaves);
    const unsigned end = (unsigned)(totalSlots * (waveId + 1) / nWaves);

    P5 me;
    me.alive = 0; me.status = RAY_NOT_CONNECTED; me.v.prim = -1;
    Float eta = 1.0, sx = 0, sy = 0;
    Rng rng; rng.s = 0;
    int depth = 0, px = 0, py = 0;
    d3 sumA = mk(0.0), sumB = mk(0.0);          // base: throughput sum (ACC_T), very direct (ACC_VD); offset i: ACC_NBR + 3 i, ACC_GRAD + 3 i
    bool live = false;                           // this lane's sample is running
    unsigned slot = 0;
    unsigned nClosest = 0, nShadow = 0, paths = 0, pathLen = 0;

    // ends a sample in this kernel: its final sums go to its slot (k_fold_cont adds them to the pixel)
    auto finish = [&]() {
        Float *q = F.qRec + slot;
        const size_t st = F.qCapacity;
        if (isBase) {
            q[(32 + ACC_T) * st] = sumA.x; q[(33 + ACC_T) * st] = sumA.y; q[(34 + ACC_T) * st] = sumA.z;
            q[(32 + ACC_VD) * st] = sumB.x; q[(33 + ACC_VD) * st] = sumB.y; q[(34 + ACC_VD) * st] = sumB.z;
            q[13 * st] = __longlong_as_double((long long)Q_DONE);
            paths++; pathLen += (unsigned)depth;
        } else {
            q[(32 + ACC_NBR + 3 * oi) * st] = sumA.x; q[(33 + ACC_NBR + 3 * oi) * st] = sumA.y; q[(34 + ACC_NBR + 3 * oi) * st] = sumA.z;
            q[(32 + ACC_GRAD + 3 * oi) * st] = sumB.x; q[(33 + ACC_GRAD + 3 * oi) * st] = sumB.y; q[(34 + ACC_GRAD + 3 * oi) * st] = sumB.z;
        }
        live = false;
    };

    while (true) {
        // ---------------- regeneration: idle seats take the next slots of the wave's range together ----------------
        {
            const unsigned idleSeats = (unsigned)(__ballot(isBase && !live) & 0xFFFULL);
            const unsigned nIdle = (unsigned)__popc(idleSeats);
            if (nIdle == S5_K && next >= end) break;
            if (next < end && (nIdle >= (unsigned)regen || nIdle == S5_K)) {
                if (__hip_atomic_load(F.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { next = end; continue; }   // cancelled: no new samples
                const bool mine = seat && !live;
                const unsigned cand = next + (unsigned)__popc(idleSeats & ((1u << k) - 1u));
                next = min(end, next + nIdle);
                bool start = mine && cand < end;
                if (start) {
                    slot = cand;
                    const unsigned t = slot & (TBLK - 1), tile = (slot / TBLK) % (unsigned)tiles, sRel = slot / F.qPixels;
                    const int tx = (int)tile % tilesX, ty = (int)tile / tilesX, w4 = (int)(t >> 6), l6 = (int)(t & 63);
                    px = rx0 + tx * 16 + (w4 & 1) * 8 + (l6 & 7); py = ry0 + ty * 16 + (w4 >> 1) * 8 + (l6 >> 3);
                    start = px < rx1 && py < ry1;
                    if (start) {
                        // ---- evaluatePoint (gpt.cpp:397-436) + the prologue of evaluate (:468-531), this lane's path ----
                        const int sample = cfg.sBase + (int)sRel;
                        rng.init(cfg.seed, (uint64_t)py * S.cam.width + px, (uint64_t)sample);
                        sx = px + rng.next1D();                                              // gpt.cpp:1261
                        sy = py + rng.next1D();
                        Float apx = 0.5, apy = 0.5;
                        if (S.cam.thinlens) { apx = rng.next1D(); apy = rng.next1D(); }     // gpt.cpp:1262-1264
                        if (S.cam.needsTime) (void)rng.next1D();                             // gpt.cpp:1265-1267
                        d3 o, d;
                        Float mint, maxt;
                        camera_ray(S.cam, sx + (isBase ? 0.0 : offset_shift_x(oi)), sy + (isBase ? 0.0 : offset_shift_y(oi)), apx, apy, o, d, mint, maxt);
                        Hit h;
                        h.t = F.pHit[(size_t)(3 * role) * F.qCapacity + slot];
                        h.u = F.pHit[(size_t)(3 * role + 1) * F.qCapacity + slot];
                        h.v = F.pHit[(size_t)(3 * role + 2) * F.qCapacity + slot];
                        h.prim = F.pPrim[(size_t)role * F.qCapacity + slot];
                        nClosest++;
                        fill_vertex(sv, h, d, me.v);
                        me.rayD = d;
                        me.throughput = mk(1.0); me.pdf = 1.0; eta = 1.0; depth = 1;
                        me.alive = isBase ? 1 : (h.prim >= 0);                               // :508-513
                        me.status = RAY_NOT_CONNECTED;
                        sumA = mk(0.0); sumB = mk(0.0);
                        live = true;
                    }
                }
                // the base path's start: misses, emission, strict normals (start_path)
                bool over = false;
                if (start && isBase) {
                    if (me.v.prim < 0) {                                                     // :482-492
                        if (ENV && S.envIndex >= 0) {
                            d3 Le = sv.emitters[S.envIndex].radiance;
                            if (GDPT_HAS_ENVMAP_N(S, 3)) { d3 rxD, ryD; camera_differentials(S.cam, sx, sy, rxD, ryD); Le = envmap_eval<INL>(*S.envMap, me.rayD, true, rxD, ryD); }
                            sumB = sumB + me.throughput * Le;
                        }
                        over = true;
                    } else {
                        sumB = sumB + me.throughput * emitted(sv, me.v.prim, -me.rayD);       // :497-499
                        if (cfg.strictNormals) { const Shading sh = shading_at<SMOOTH>(sv, me.v); if (dot(me.rayD, sh.geoN) * toLocal(sh.fr, -me.rayD).z >= 0) over = true; }
                    }
                }
                const bool overAll = of_base(start && over);
                if (start && !isBase && !overAll && cfg.strictNormals && me.alive) {         // :516-531
                    const Shading sh = shading_at<SMOOTH>(sv, me.v);
                    if (dot(me.rayD, sh.geoN) * toLocal(sh.fr, -me.rayD).z >= 0) me.alive = 0;
                }
                if (start && overAll) finish();
            }
        }
        const bool run = seat && live;

        // =====================================================================================================================
        // one iteration of the main loop of evaluate (gpt.cpp:537-1175) for every running sample of the wave
        // =====================================================================================================================
        // Values that cross a phase boundary live in the mailbox, not in registers: a lane's registers would otherwise hold the base role's
        // and the offset role's temporaries of every phase at once (the first version of this kernel spilled 1.5 KB per lane that way).
        bool endNow = false;                 // base: the base path ends here
        bool smooth = false, gate = false, lightSA = true, mainDiffuseSm = false;
        // ---- N0 (base): top of the bounce, emitter sample ----
        if (run && isBase) {
            if (!(depth < cfg.maxDepth || cfg.maxDepth < 0)) endNow = true;                 // :537
            else {
                const TriShade &mts = sv.shade[me.v.prim];
                const Shading msh = shading_at<SMOOTH>(sv, me.v);
                const d3 mainWi = toLocal(msh.fr, -me.rayD);
                if (cfg.strictNormals && dot(me.rayD, msh.geoN) * mainWi.z >= 0) endNow = true; // :541-546
                else {
                    const MaterialD &mainBSDF = sv.mats[mts.material];
                    const d3 mainR = reflectance_at<SMOOTH, INL>(sv, mainBSDF, me.v, depth == 1, &S.cam, sx, sy);
                    put3(MB_FR, msh.fr.s); put3(MB_FR + 3, msh.fr.t); put3(MB_FR + 6, msh.fr.n); put3(MB_GN, msh.geoN);
                    put3(MB_R, mainR); put3(MB_P, me.v.p); put(MB_MAT, (Float)mts.material); put3(MB_WI, mainWi);
                    smooth = (bsdfType(mainBSDF) & ESmooth) != 0;
                    if (smooth) {
                        DRec dRec;
                        dRec.ref = me.v.p; dRec.refN = (mainBSDF.twoSided || mainBSDF.type == 3) ? mk(0.0) : msh.fr.n;
                        const Float lsx = rng.next1D(), lsy = rng.next1D();                  // :572
                        const d3 value = sample_emitter_direct<ENV>(S, sv, dRec, lsx, lsy);
                        lightSA = !(ENV && dRec.offSurfaceDiscrete);
                        const d3 mainWoL = toLocal(msh.fr, dRec.d);
                        gate = !cfg.strictNormals || dot(msh.geoN, dRec.d) * mainWoL.z > 0; // :607
                        mainDiffuseSm = vertex_is_diffuse(mainBSDF, cfg, ESmooth);
                        put(MB_LS, lsx); put(MB_LS + 1, lsy);
                        put3(MB_DP, dRec.p); put3(MB_DN, dRec.n); put3(MB_DD, dRec.d); put(MB_DPDF, dRec.pdf); put(MB_DDIST, dRec.dist);
                        put3(MB_VAL, value);
                    }
                }
            }
        }
        const bool ended0 = of_base(run && endNow);
        const bool smoothS = of_base(run && smooth), gateS = of_base(run && gate), lightSAS = of_base(run && lightSA), mainDiffS = of_base(run && mainDiffuseSm);
        const bool depth1 = of_base(run && depth == 1);
        wave_sync5();
        // ---- N0' (offsets): strict normals of the offsets' own vertices (:547-554), then the own emitter sample where the shift needs one ----
        bool ownNee = false;
        d3 sValue = mk(0.0), sD = mk(0.0);
        Float sPdf = 0, sDist = 0;
        if (run && !isBase && !ended0) {
            if (cfg.strictNormals && me.alive) { const Shading sh = shading_at<SMOOTH>(sv, me.v); if (dot(me.rayD, sh.geoN) * toLocal(sh.fr, -me.rayD).z >= 0) me.alive = 0; }
            if (smoothS && gateS && me.alive && me.status == RAY_NOT_CONNECTED) {            // :659-672
                const MaterialD &shiftedBSDF = sv.mats[sv.shade[me.v.prim].material];
                if (!lightSAS || (mainDiffS && vertex_is_diffuse(shiftedBSDF, cfg, ESmooth))) {
                    DRec sRec;
                    sRec.ref = me.v.p; sRec.refN = (shiftedBSDF.twoSided || shiftedBSDF.type == 3) ? mk(0.0) : shading_at<SMOOTH>(sv, me.v).fr.n;
                    sValue = sample_emitter_direct<ENV>(S, sv, sRec, get(MB_LS), get(MB_LS + 1));
                    sD = sRec.d; sDist = sRec.dist; sPdf = sRec.pdf;
                    ownNee = true;
                }
            }
        }
        // ---- N1: the shadow rays of the emitter samples, base path and offsets in one pass (scene.cpp:869-876) ----
        bool occluded = false;
        {
            const bool doBase = run && isBase && smooth;
            if (doBase || ownNee) {
                const d3 o = me.v.p, d = doBase ? get3(MB_DD) : sD;
                const Float maxt = (doBase ? get(MB_DDIST) : sDist) * (1 - GD_SHADOW_EPSILON);
                Hit h;
                nShadow++;
                occluded = trace<true>(sv, stack, o, d, ray_mint_shadow(o, GD_EPSILON), maxt, h);
            }
        }
        // ---- N2 (base): the base path's direct illumination terms, :575-606 ----
        if (run && isBase && smooth) {
            const MaterialD &mainBSDF = sv.mats[(int)get(MB_MAT)];
            Frame3 mfr; mfr.s = get3(MB_FR); mfr.t = get3(MB_FR + 3); mfr.n = get3(MB_FR + 6);
            const bool mainEmitterVisible = !occluded;
            const d3 value = mainEmitterVisible ? get3(MB_VAL) : mk(0.0);
            const Float dRecPdf = get(MB_DPDF);
            const d3 dP = get3(MB_DP);
            const d3 mainEmitterRadiance = value * dRecPdf;                                  // :575
            const d3 mainWoL = toLocal(mfr, get3(MB_DD));
            d3 mainBSDFValue;
            Float mainBsdfPdfRaw;
            bsdf_eval_pdf(mainBSDF, get3(MB_R), get3(MB_WI), mainWoL, MEASURE_SOLID_ANGLE, mainBSDFValue, mainBsdfPdfRaw);   // :588
            const Float mainBsdfPdf = (lightSA && mainEmitterVisible) ? mainBsdfPdfRaw : 0;   // :592
            const Float mainDistanceSquared = len2(me.v.p - dP);
            const Float mainOpposingCosine = dot(get3(MB_DN), (me.v.p - dP)) / sqrt(mainDistanceSquared);
            const Float mainWeightNumerator = me.pdf * dRecPdf;                              // :599-600
            const Float mainWeightDenominator = (me.pdf * me.pdf) * ((dRecPdf * dRecPdf) + (mainBsdfPdf * mainBsdfPdf));
            const d3 mainContributionAll = me.throughput * (mainBSDFValue * mainEmitterRadiance);
            if (gate) {
                put3(MB_BE, mainBSDFValue * mainEmitterRadiance); put3(MB_ERAD, mainEmitterRadiance); put(MB_BPDF, mainBsdfPdf);
                put(MB_NUM, mainWeightNumerator); put(MB_DEN, mainWeightDenominator); put3(MB_CALL, mainContributionAll);
                put(MB_DIST2, mainDistanceSquared); put(MB_OPCOS, mainOpposingCosine);
                put(MB_LUM, mainEmitterVisible ? 1.0 : 0.0);
            }
        }
        wave_sync5();
        // ---- N3 (offsets): weights and contributions of the emitter sample, :610-726 ----
        if (run && !isBase && !ended0 && smoothS && gateS) {
            const Float dRecPdf = get(MB_DPDF), mainBsdfPdf = get(MB_BPDF);
            const Float mainWeightNumerator = get(MB_NUM), mainWeightDenominator = get(MB_DEN);
            d3 shiftedContribution = mk(0.0);
            Float weight = 0;
            bool assigned = false;
            bool shiftSuccessful = me.alive != 0;
            if (shiftSuccessful) {
                if (me.status == RAY_CONNECTED) {                                            // :622-637
                    const Float den = (me.pdf * me.pdf) * ((dRecPdf * dRecPdf) + (mainBsdfPdf * mainBsdfPdf));
                    weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                    shiftedContribution = 1.0 * me.throughput * get3(MB_BE);
                    assigned = true;
                } else if (me.status == RAY_RECENTLY_CONNECTED) {                            // :638-658
                    const MaterialD &mainBSDF = sv.mats[(int)get(MB_MAT)];
                    Frame3 bfr; bfr.s = get3(MB_FR); bfr.t = get3(MB_FR + 3); bfr.n = get3(MB_FR + 6);
                    const d3 incoming = normalize(me.v.p - get3(MB_P));
                    d3 f;
                    Float pdfRaw;
                    bsdf_eval_pdf(mainBSDF, get3(MB_R), toLocal(bfr, incoming), toLocal(bfr, get3(MB_DD)), MEASURE_SOLID_ANGLE, f, pdfRaw);
                    const bool mainEmitterVisible = get(MB_LUM) != 0.0;
                    const Float shiftedBsdfPdf = (lightSAS && mainEmitterVisible) ? pdfRaw : 0;
                    const Float den = (me.pdf * me.pdf) * ((dRecPdf * dRecPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                    weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                    shiftedContribution = 1.0 * me.throughput * (f * get3(MB_ERAD));
                    assigned = true;
                } else if (ownNee) {                                                         // :659-705
                    const MaterialD &shiftedBSDF = sv.mats[sv.shade[me.v.prim].material];
                    const Shading ssh = shading_at<SMOOTH>(sv, me.v);
                    const bool shiftedEmitterVisible = !occluded;
                    if (!shiftedEmitterVisible) sValue = mk(0.0);
                    const d3 shiftedEmitterRadiance = sValue * sPdf;
                    const Float shiftedDRecPdf = sPdf;
                    const d3 dP = get3(MB_DP);
                    const Float shiftedDistanceSquared = len2(dP - me.v.p);
                    const d3 emitterDirection = (dP - me.v.p) / sqrt(shiftedDistanceSquared);
                    const Float shiftedOpposingCosine = -dot(get3(MB_DN), emitterDirection);
                    const d3 woL = toLocal(ssh.fr, emitterDirection);
                    if (cfg.strictNormals && dot(ssh.geoN, emitterDirection) * woL.z < 0) {
                        shiftSuccessful = false;
                    } else {
                        d3 f;
                        Float pdfRaw;
                        bsdf_eval_pdf(shiftedBSDF, reflectance_at<SMOOTH, INL>(sv, shiftedBSDF, me.v, depth1, &S.cam, sx + offset_shift_x(oi), sy + offset_shift_y(oi)), toLocal(ssh.fr, -me.rayD), woL, MEASURE_SOLID_ANGLE, f, pdfRaw);
                        const Float shiftedBsdfPdf = (lightSAS && shiftedEmitterVisible) ? pdfRaw : 0;
                        const Float jacobian = fabs(shiftedOpposingCosine * get(MB_DIST2)) / (GD_EPSILON + fabs(get(MB_OPCOS) * shiftedDistanceSquared)); // :695
                        const Float den = (jacobian * me.pdf) * (jacobian * me.pdf) * ((shiftedDRecPdf * shiftedDRecPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                        weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                        shiftedContribution = jacobian * me.throughput * (f * shiftedEmitterRadiance);
                        assigned = true;
                    }
                }
            }
            if (!shiftSuccessful) {                                                          // :708-717
                weight = mainWeightNumerator / (GD_D_EPSILON + mainWeightDenominator);
                shiftedContribution = mk(0.0);
                assigned = true;
            }
            const d3 mainContribution = assigned ? get3(MB_CALL) : mk(0.0);
            sumA = sumA + shiftedContribution * weight;                                     // :723-726
            sumB = sumB + (shiftedContribution - mainContribution) * weight;
            put(MB_W + oi, weight); put(MB_ASG + oi, assigned ? 1.0 : 0.0);
        }
        wave_sync5();
        // ---- N4 (base): the throughput sum, offset by offset as bounce() adds it ----
        if (run && isBase && smooth && gate) {
            const d3 mainContributionAll = get3(MB_CALL);
#pragma unroll
            for (int i = 0; i < 4; i++) { const d3 mc = get(MB_ASG + i) != 0.0 ? mainContributionAll : mk(0.0); sumA = sumA + mc * get(MB_W + i); }
        }

        // ================= BSDF sampling and emitter hits, :737-1151 =================
        BSDFSample bs;
        bool lastSegment = false, mainVertexDiffuse = false, mainNextVertexDiffuse = false, mainHitEnv = false, mainHitEmitter = false;
        int ntsEmitter = -1;
        bs.sampledType = 0;
        // ---- B0 (base): sample the BSDF ----
        bool extend = false;
        if (run && isBase && !endNow) {
            const MaterialD &mainBSDF = sv.mats[(int)get(MB_MAT)];
            lastSegment = (depth + 1 == cfg.maxDepth);                                       // :559
            const Float bsx = rng.next1D(), bsy = rng.next1D();                              // :456
            bsdf_sample(mainBSDF, get3(MB_R), get3(MB_WI), bsx, bsy, bs);
            if (bs.pdf <= 0.0) endNow = true;                                                // :740
            else {
                Frame3 mfr; mfr.s = get3(MB_FR); mfr.t = get3(MB_FR + 3); mfr.n = get3(MB_FR + 6);
                const d3 mainWo = toWorld(mfr, bs.wo);
                if (cfg.strictNormals && dot(get3(MB_GN), mainWo) * bs.wo.z <= 0) endNow = true;   // :749
                else {
                    mainVertexDiffuse = vertex_is_diffuse(mainBSDF, cfg, bs.sampledType);    // :765
                    me.rayD = mainWo;                                                        // :768 (the ray starts at previousMainIts.p = MB_P)
                    extend = true;
                }
            }
        }
        // ---- B1 (base): the next segment of the base path ----
        Hit nh;
        nh.prim = -1; nh.t = 0;
        if (extend) {
            const d3 o = me.v.p;
            nClosest++;
            trace<false>(sv, stack, o, me.rayD, ray_mint_closest(o, GD_EPSILON), GD_INF, nh);
        }
        // ---- B2 (base): the new base vertex, :772-826 ----
        if (extend) {
            const MaterialD &mainBSDF = sv.mats[(int)get(MB_MAT)];
            const d3 prevP = me.v.p, mfrN = get3(MB_FR + 6);
            bool mainHitEnvV = false;
            DRec envRec;
            if (nh.prim < 0) {                                                               // :786-804
                if (!ENV || S.envIndex < 0) endNow = true;
                else {
                    envRec.ref = prevP; envRec.refN = (mainBSDF.twoSided || mainBSDF.type == 3) ? mk(0.0) : mfrN;
                    if (!env_fill_drec(S, envRec, prevP, me.rayD)) endNow = true;
                    else { mainHitEnvV = true; me.v.prim = -1; }
                }
            } else fill_vertex(sv, nh, me.rayD, me.v);
            if (!endNow) {
                mainHitEnv = ENV && mainHitEnvV;
                const TriShade &nts = sv.shade[mainHitEnv ? 0 : me.v.prim];
                mainHitEmitter = mainHitEnv || nts.emitter >= 0;                             // :772-777, :793
                ntsEmitter = nts.emitter;
                const d3 mainEmitterRadiance = mainHitEnv ? env_radiance<INL>(S, sv, me.rayD) : (mainHitEmitter ? emitted(sv, me.v.prim, -me.rayD) : mk(0.0));
                mainNextVertexDiffuse = mainHitEnv ? true : vertex_is_diffuse(sv.mats[nts.material], cfg, bs.sampledType);  // :785, :799
                const Float mainBsdfPdf = bs.pdf, mainPreviousPdf = me.pdf;
                me.throughput = me.throughput * (bs.weight * bs.pdf);                        // :810-812
                me.pdf *= bs.pdf;
                eta *= bs.eta;
                const Float mainLumPdf = (mainHitEmitter && !(bs.sampledType & EDelta))
                    ? (mainHitEnv ? pdf_emitter_direct<ENV>(S, sv, S.envIndex, envRec.d, envRec.refN, envRec.n, envRec.dist)
                                  : pdf_emitter_direct<ENV>(S, sv, nts.emitter, me.rayD, (mainBSDF.twoSided || mainBSDF.type == 3) ? mk(0.0) : mfrN, nts.n, nh.t)) : 0;  // :815
                const Float mainWeightNumerator = mainPreviousPdf * bs.pdf;                   // :819-820
                const Float mainWeightDenominator = (mainPreviousPdf * mainPreviousPdf) * ((mainLumPdf * mainLumPdf) + (mainBsdfPdf * mainBsdfPdf));
                const d3 mainContribution = me.throughput * mainEmitterRadiance;
                put3(MB_WO, bs.wo); put3(MB_BWP, bs.weight * bs.pdf); put(MB_BSPDF, bs.pdf); put(MB_LUM, mainLumPdf);
                put3(MB_NP, me.v.p); put3(MB_RD, me.rayD); put(MB_NPRIM, (Float)me.v.prim); put(MB_LPDF, me.pdf);
                put3(MB_NGN, mainHitEnv ? mk(0.0) : (SMOOTH ? shading_at<SMOOTH>(sv, me.v).geoN : nts.n));   // main.rRec.its.geoFrame.n, :911
                put3(MB_ERAD, mainEmitterRadiance); put(MB_NUM, mainWeightNumerator); put(MB_DEN, mainWeightDenominator); put3(MB_CALL, mainContribution);
            }
        }
        const bool ended1 = of_base(run && endNow);              // (includes ended0)
        const bool lastSegS = of_base(run && lastSegment), mvdS = of_base(run && mainVertexDiffuse), mnvdS = of_base(run && mainNextVertexDiffuse);
        const bool hitEnvS = of_base(run && mainHitEnv), hitEmS = of_base(run && mainHitEmitter);
        const int sampledTypeS = (int)__shfl(bs.sampledType, k), ntsEmitterS = (int)__shfl(ntsEmitter, k);
        wave_sync5();
        // ---- B3 (offsets): :830-1146 ----
        if (run && !isBase && !ended1) {
            const int measureS = (sampledTypeS & EDelta) ? MEASURE_DISCRETE : MEASURE_SOLID_ANGLE;
            const Float mainWeightNumerator = get(MB_NUM), mainWeightDenominator = get(MB_DEN), mainLumPdf = get(MB_LUM), mainBsdfPdf = get(MB_BSPDF);
            const d3 mainEmitterRadiance = get3(MB_ERAD);
            d3 shiftedContribution = mk(0.0);
            Float weight = 0;
            bool assigned = false;
            bool postponedShiftEnd = false;
            if (me.alive) {
                const Float shiftedPreviousPdf = me.pdf;
                if (me.status == RAY_CONNECTED) {                                            // :844-861
                    me.throughput = me.throughput * get3(MB_BWP);
                    me.pdf *= mainBsdfPdf;
                    const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((mainLumPdf * mainLumPdf) + (mainBsdfPdf * mainBsdfPdf));
                    weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                    shiftedContribution = me.throughput * mainEmitterRadiance;
                    assigned = true;
                } else if (me.status == RAY_RECENTLY_CONNECTED) {                            // :862-888
                    const MaterialD &mainBSDF = sv.mats[(int)get(MB_MAT)];
                    Frame3 bfr; bfr.s = get3(MB_FR); bfr.t = get3(MB_FR + 3); bfr.n = get3(MB_FR + 6);
                    const d3 incoming = normalize(me.v.p - get3(MB_P));
                    d3 f;
                    Float shiftedBsdfPdf;
                    bsdf_eval_pdf(mainBSDF, get3(MB_R), toLocal(bfr, incoming), toLocal(bfr, get3(MB_RD)), measureS, f, shiftedBsdfPdf);
                    me.throughput = me.throughput * f;
                    me.pdf *= shiftedBsdfPdf;
                    me.status = RAY_CONNECTED;
                    const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((mainLumPdf * mainLumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                    weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                    shiftedContribution = me.throughput * mainEmitterRadiance;
                    assigned = true;
                } else {                                                                     // :889-1126
                    const TriShade &sts = sv.shade[me.v.prim];
                    const MaterialD &shiftedBSDF = sv.mats[sts.material];
                    const Shading sh2 = shading_at<SMOOTH>(sv, me.v);
                    const Frame3 sfr = sh2.fr;
                    const bool shiftedVertexDiffuse = vertex_is_diffuse(shiftedBSDF, cfg, sampledTypeS);
                    const d3 shiftedR = reflectance_at<SMOOTH, INL>(sv, shiftedBSDF, me.v, depth1, &S.cam, sx + offset_shift_x(oi), sy + offset_shift_y(oi));
                    const d3 baseP = get3(MB_NP), baseD = get3(MB_RD), baseO = get3(MB_P);
                    if (mvdS && mnvdS && shiftedVertexDiffuse) {
                        // ---- reconnection shift, :897-986 ----
                        if (!lastSegS || hitEmS) {                                           // :901
                            bool visible;
                            {
                                d3 vo = me.v.p, vd;
                                Float vmax;
                                if (hitEnvS) {                                               // environmentShift + testEnvironmentVisibility, :96-114,348-369
                                    DRec er;
                                    er.dist = 0.0;
                                    env_fill_drec(S, er, me.v.p, baseD);
                                    vd = baseD; vmax = (1.0 - GD_SHADOW_EPSILON) * er.dist;
                                } else { vd = baseP - me.v.p; vmax = 1.0 - GD_SHADOW_EPSILON; }   // testVisibility, :84-93
                                Hit h;
                                nShadow++;
                                visible = !trace<true>(sv, stack, vo, vd, ray_mint_shadow(vo, GD_EPSILON), vmax, h);
                            }
                            if (!visible) { me.alive = 0; }
                            else if (hitEnvS) {
                                const d3 shiftedWo = baseD;
                                const d3 woL = toLocal(sfr, shiftedWo);
                                if (cfg.strictNormals && dot(shiftedWo, sh2.geoN) * woL.z <= 0) { me.alive = 0; }
                                else {
                                    d3 f;
                                    Float shiftedBsdfPdf;
                                    bsdf_eval_pdf(shiftedBSDF, shiftedR, toLocal(sfr, -me.rayD), woL, MEASURE_SOLID_ANGLE, f, shiftedBsdfPdf);
                                    me.throughput = me.throughput * (f * 1.0);
                                    me.pdf *= shiftedBsdfPdf * 1.0;
                                    me.status = RAY_RECENTLY_CONNECTED;
                                    const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((mainLumPdf * mainLumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                                    weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                                    shiftedContribution = me.throughput * mainEmitterRadiance;
                                    assigned = true;
                                }
                            } else {
                                const d3 mainEdge = baseO - baseP, shiftedEdge = me.v.p - baseP;
                                const Float mainEdgeLengthSquared = len2(mainEdge), shiftedEdgeLengthSquared = len2(shiftedEdge);
                                const d3 shiftedWo = -shiftedEdge / sqrt(shiftedEdgeLengthSquared);
                                const d3 nGeoN = get3(MB_NGN);
                                const Float mainOpposingCosine = dot(mainEdge, nGeoN) / sqrt(mainEdgeLengthSquared);
                                const Float shiftedOpposingCosine = dot(shiftedWo, nGeoN);
                                const Float jacobian = fabs(shiftedOpposingCosine * mainEdgeLengthSquared) / (GD_D_EPSILON + fabs(mainOpposingCosine * shiftedEdgeLengthSquared));
                                const d3 woL = toLocal(sfr, shiftedWo);
                                if (cfg.strictNormals && dot(shiftedWo, sh2.geoN) * woL.z <= 0) { me.alive = 0; }
                                else {
                                    d3 f;
                                    Float shiftedBsdfPdf;
                                    bsdf_eval_pdf(shiftedBSDF, shiftedR, toLocal(sfr, -me.rayD), woL, MEASURE_SOLID_ANGLE, f, shiftedBsdfPdf);
                                    me.throughput = me.throughput * (f * jacobian);          // :939-940
                                    me.pdf *= shiftedBsdfPdf * jacobian;
                                    me.status = RAY_RECENTLY_CONNECTED;
                                    if (hitEmS) {                                            // :944-986
                                        const int nprim = (int)get(MB_NPRIM);
                                        const d3 shiftedEmitterRadiance = emitted(sv, nprim, -shiftedWo);
                                        const Float sdist = len(baseP - me.v.p);
                                        const d3 sd = (baseP - me.v.p) / sdist;
                                        const Float shiftedLumPdf = pdf_emitter_direct<ENV>(S, sv, ntsEmitterS, sd, sfr.n, sv.shade[nprim].n, sdist);
                                        const Float den = (shiftedPreviousPdf * shiftedPreviousPdf) * ((shiftedLumPdf * shiftedLumPdf) + (shiftedBsdfPdf * shiftedBsdfPdf));
                                        weight = mainWeightNumerator / (GD_D_EPSILON + den + mainWeightDenominator);
                                        shiftedContribution = me.throughput * shiftedEmitterRadiance;
                                        assigned = true;
                                    }
                                }
                            }
                        }
                    } else {
                        // ---- half-vector duplication shift, :987-1126 ----
                        const MaterialD &mainBSDF = sv.mats[(int)get(MB_MAT)];
                        const Float basePdf = get(MB_LPDF);
                        d3 shiftedEmitterRadiance = mk(0.0);
                        bool envEnd = false;
                        const d3 tsIn = toLocal(sfr, -me.rayD);
                        const bool bothDelta = (sampledTypeS & EDelta) && (bsdfType(shiftedBSDF) & EDelta);     // :996-1001
                        const bool bothSmooth = (sampledTypeS & ESmooth) && (bsdfType(shiftedBSDF) & ESmooth);
                        bool ok = bothDelta || bothSmooth;
                        d3 tsOut = mk(0.0);
                        if (ok) {
                            Float jacobian;
                            ok = half_vector_shift(get3(MB_WI), get3(MB_WO), tsIn, bsdf_eta(mainBSDF), bsdf_eta(shiftedBSDF), jacobian, tsOut);   // :1006
                            if (sampledTypeS & EDelta) jacobian = 1;                         // :1008-1011
                            if (ok) { me.throughput = me.throughput * jacobian; me.pdf *= jacobian; }
                        }
                        if (ok) {
                            const d3 outgoing = toWorld(sfr, tsOut);
                            d3 f;
                            Float p;
                            bsdf_eval_pdf(shiftedBSDF, shiftedR, tsIn, tsOut, measureS, f, p);
                            me.throughput = me.throughput * f;
                            me.pdf *= p;
                            if (me.pdf == 0) ok = false;                                     // :1034
                            else if (cfg.strictNormals && dot(outgoing, sh2.geoN) * tsOut.z <= 0) ok = false;
                            else {
                                Hit h;
                                nClosest++;
                                trace<false>(sv, stack, me.v.p, outgoing, ray_mint_closest(me.v.p, GD_EPSILON), GD_INF, h);   // :1050-1052
                                if (h.prim < 0) {                                            // :1052-1074
                                    if (!ENV || S.envIndex < 0 || !hitEnvS || (mvdS && shiftedVertexDiffuse)) ok = false;
                                    else { shiftedEmitterRadiance = env_radiance<INL>(S, sv, outgoing); envEnd = true; }
                                } else if (hitEnvS) ok = false;                               // :1078-1082
                                else {
                                    me.rayD = outgoing;
                                    fill_vertex(sv, h, outgoing, me.v);
                                    const TriShade &snts = sv.shade[me.v.prim];
                                    const bool shiftedNextVertexDiffuse = vertex_is_diffuse(sv.mats[snts.material], cfg, sampledTypeS);
                                    if (mvdS && shiftedVertexDiffuse && shiftedNextVertexDiffuse) ok = false;   // :1089-1093
                                    else if (snts.emitter >= 0) shiftedEmitterRadiance = emitted(sv, me.v.prim, -outgoing);
                                }
                            }
                        }
                        if (ok) {                                                            // :1106-1112
                            weight = basePdf / (me.pdf * me.pdf + basePdf * basePdf);
                            shiftedContribution = me.throughput * shiftedEmitterRadiance;
                            if (envEnd) postponedShiftEnd = true;                            // :1073
                        } else {                                                             // :1113-1124
                            weight = 1.0 / basePdf;
                            shiftedContribution = mk(0.0);
                            postponedShiftEnd = true;
                        }
                        assigned = true;
                    }
                }
            }
            if (!me.alive) {                                                                 // :1130-1136 (shift_failed)
                weight = mainWeightNumerator / (GD_D_EPSILON + mainWeightDenominator);
                shiftedContribution = mk(0.0);
                assigned = true;
            }
            const d3 mc = assigned ? get3(MB_CALL) : mk(0.0);
            sumA = sumA + shiftedContribution * weight;                                     // :1140-1146
            sumB = sumB + (shiftedContribution - mc) * weight;
            put(MB_W + oi, weight); put(MB_ASG + oi, assigned ? 1.0 : 0.0);
            if (postponedShiftEnd) me.alive = 0;
        }
        wave_sync5();
        // ---- B6 (base): throughput sum, environment end, Russian roulette (:1155-1174) ----
        bool rrPlayed = false;
        if (run && isBase && !endNow) {
            const d3 mainContribution = get3(MB_CALL);
#pragma unroll
            for (int i = 0; i < 4; i++) { const d3 mc = get(MB_ASG + i) != 0.0 ? mainContribution : mk(0.0); sumA = sumA + mc * get(MB_W + i); }
            if (mainHitEnv) endNow = true;                                                   // :1155-1157
            else if (depth++ >= cfg.rrDepth) {
                const Float q = fmin(maxc(me.throughput / me.pdf) * eta * eta, (Float)0.95f);
                if (rng.next1D() >= q) endNow = true;
                else { me.pdf *= q; put(MB_Q, q); rrPlayed = true; }
            }
        }
        const bool ended2 = of_base(run && endNow), rrS = of_base(run && rrPlayed);
        wave_sync5();
        if (run && !isBase && !ended2 && rrS) me.pdf *= get(MB_Q);

        // ---------------- end of the bounce: finished, handed over, or on to the next one ----------------
        if (run && ended2) finish();
        {
            // every offset connected or dead: the rest of this base path belongs to k_continue (the sums so far travel with it)
            const unsigned long long open = __ballot(run && !isBase && !ended2 && me.alive && me.status != RAY_CONNECTED);
            const bool anyOpen = ((open >> (S5_K + k)) | (open >> (2 * S5_K + k)) | (open >> (3 * S5_K + k)) | (open >> (4 * S5_K + k))) & 1ULL;
            const bool hand = run && !ended2 && !anyOpen;
            const unsigned long long aliveBits = __ballot(run && !isBase && me.alive);
            const unsigned handSeats = (unsigned)(__ballot(hand && isBase) & 0xFFFULL);
            if (handSeats) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&F.qCount[0], (unsigned)__popc(handSeats));
                base = __shfl(base, 0);
                if (hand) {
                    Float *q = F.qRec + slot;
                    const size_t st = F.qCapacity;
                    if (isBase) {
                        q[0 * st] = me.throughput.x; q[1 * st] = me.throughput.y; q[2 * st] = me.throughput.z;
                        q[3 * st] = me.pdf; q[4 * st] = eta;
                        q[5 * st] = me.v.p.x; q[6 * st] = me.v.p.y; q[7 * st] = me.v.p.z;
                        q[8 * st] = me.rayD.x; q[9 * st] = me.rayD.y; q[10 * st] = me.rayD.z;
                        q[11 * st] = me.v.u; q[12 * st] = me.v.v;
                        q[13 * st] = __longlong_as_double((long long)(((unsigned long long)(unsigned)depth << 32) | (unsigned)me.v.prim));
                        q[14 * st] = __longlong_as_double((long long)rng.s);
                        unsigned alive = 0;
#pragma unroll
                        for (int i = 0; i < 4; i++) alive |= (unsigned)((aliveBits >> ((i + 1) * S5_K + k)) & 1ULL) << i;
                        q[31 * st] = __longlong_as_double((long long)alive);
                        q[(32 + ACC_T) * st] = sumA.x; q[(33 + ACC_T) * st] = sumA.y; q[(34 + ACC_T) * st] = sumA.z;
                        q[(32 + ACC_VD) * st] = sumB.x; q[(33 + ACC_VD) * st] = sumB.y; q[(34 + ACC_VD) * st] = sumB.z;
                        F.qList[base + (unsigned)__popc(handSeats & ((1u << k) - 1u))] = slot;
                    } else {
                        q[(15 + 4 * oi) * st] = me.throughput.x; q[(16 + 4 * oi) * st] = me.throughput.y; q[(17 + 4 * oi) * st] = me.throughput.z; q[(18 + 4 * oi) * st] = me.pdf;
                        q[(32 + ACC_NBR + 3 * oi) * st] = sumA.x; q[(33 + ACC_NBR + 3 * oi) * st] = sumA.y; q[(34 + ACC_NBR + 3 * oi) * st] = sumA.z;
                        q[(32 + ACC_GRAD + 3 * oi) * st] = sumB.x; q[(33 + ACC_GRAD + 3 * oi) * st] = sumB.y; q[(34 + ACC_GRAD + 3 * oi) * st] = sumB.z;
                    }
                    live = false;
                }
            }
        }
    }
    // statistics: wave-level integer reduction, one atomic per wave and counter
    const unsigned c0 = __builtin_amdgcn_wave_reduce_add_u32(nClosest, 0), c1 = __builtin_amdgcn_wave_reduce_add_u32(nShadow, 0);
    const unsigned c2 = __builtin_amdgcn_wave_reduce_add_u32(paths, 0), c3 = __builtin_amdgcn_wave_reduce_add_u32(pathLen, 0);
    if (lane == 0) {
        atomicAdd(&F.stats[0], (unsigned long long)c0);
        atomicAdd(&F.stats[1], (unsigned long long)c1);
        atomicAdd(&F.stats[2], (unsigned long long)c2);
        atomicAdd(&F.stats[3], (unsigned long long)c3);
    }
}

} // namespace gdpt_tr
