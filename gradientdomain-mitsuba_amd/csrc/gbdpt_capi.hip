// gbdpt_capi.hip -- C-ABI (include/gdpt_tracer.h, "G-BDPT") over the gfx950 G-BDPT sampler of gbdpt_kernels.hip.h: what GBDPTRenderer::process
// computes for a rectangle of pixels (src/integrators/gbdpt/gbdpt_proc.cpp:86-256), the five camera blocks and five light images of
// GBDPTWorkResult (gbdpt_wr.{h,cpp}) merged as GBDPTProcess::processResult / develop do (gbdpt_proc.cpp:694-763, multifilm.cpp:317-362).
// No CPU fallback: every value comes from the kernels below.
#include "../../include/gdpt_tracer.h"
#include "gbdpt_general.hip.h"
#include "gpt_scene.hip.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace gdpt_tr;
using namespace gdpt_bd;

extern "C" int gdpt_internal_fail(int code, const char *msg);

namespace gdpt_bdk {        // (named: rocprofv3 prints kernels of an anonymous namespace without their names)

int bfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return gdpt_internal_fail(code, buf);
}
#define BHIPCHK(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return bfail(GDPT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

__device__ __forceinline__ SceneView hbm_scene_view(const SceneD &S)
{
    SceneView sv;
    sv.nodes = S.nodes; sv.isect = S.isect; sv.shade = S.shade; sv.mats = S.mats; sv.emitters = S.emitters; sv.emTris = S.emTris; sv.emCdf = S.emCdf;
    sv.emitterCdf = S.emitterCdf; sv.rootRef = S.rootRef; sv.boundM = S.boundM; sv.quant = S.quantNodes; sv.leafExit = 1; sv.vn = S.vn; sv.uv = S.uv; sv.hasUV = S.hasUV; sv.tex = S.tex;
    return sv;
}

// ImageBlock::put (imageblock.h:150-210) with the box filter and negative values allowed (gbdpt_proc.cpp:175-179): a put with a non-finite
// channel is dropped whole; the footprint [ceil(p - r), floor(p + r)] is one pixel except within 1e-5 of a pixel edge (box.cpp:38).
// fp64 atomics: a light sample lands on any pixel, and the samples of one pixel are spread over lanes.
__device__ void film_put(Float *buf, int stride, int W, int H, Float px, Float py, d3 spec, bool withWeight, unsigned long long *invalid)
{
    if (!is_finite(spec.x) || !is_finite(spec.y) || !is_finite(spec.z)) { atomicAdd(invalid, 1ULL); return; }
    const FilterD flt = box_filter();
    const Float posx = px - 0.5, posy = py - 0.5;
    const int x0 = max((int)ceil(posx - flt.radius), 0), y0 = max((int)ceil(posy - flt.radius), 0);
    const int x1 = min((int)floor(posx + flt.radius), W - 1), y1 = min((int)floor(posy + flt.radius), H - 1);
    for (int y = y0; y <= y1; ++y) {
        const Float wy = eval_discretized(flt, y - posy);
        for (int x = x0; x <= x1; ++x) {
            const Float w = eval_discretized(flt, x - posx) * wy;
            Float *dest = buf + ((size_t)y * W + x) * stride;
            atomicAdd(dest + 0, w * spec.x); atomicAdd(dest + 1, w * spec.y); atomicAdd(dest + 2, w * spec.z);
            if (withWeight) atomicAdd(dest + 3, w * 1.0);
        }
    }
}

// ---- the frame in wavefront form ------------------------------------------------------------------------------------------------------
// The reference evaluates a sample's connections in a serial double loop (gbdpt_proc.cpp:311-528).  One lane doing that holds both subpaths,
// the four offset paths, the prefix products and the MIS arrays (17 KB) for the whole loop: 1 wave per SIMD, every access a scratch round
// trip (2.0 Msample/s).  Here a chunk of samples goes through three launches joined by HBM records:
//   k_bd_walk     one lane per sample: subpaths, connected base path, offset paths, prefix products -> a `Sample` record (11 KB) in HBM, the
//                 sample's connections (s, t) appended to an item list, its film position;
//   k_bd_connect  one lane per CONNECTION (on average ~25 per sample): reads the few vertices it needs from its sample's record (lanes of a
//                 wave mostly share one record: the loads coalesce), evaluates the base path and the four offsets, adds primal / gradient
//                 terms to the sample's 15 sums (fp64 atomics) or splats light-tracing terms into the light images;
//   k_bd_put      one lane per sample: the five camera-block puts of its sums (GBDPTWorkResult::putSample, gbdpt_proc.cpp:531-533).
// Same arithmetic per connection as the one-lane form (process_sample, kept as the probe); the sums of a sample are added in arrival order
// instead of (s, t) order: rounding of the last bits only.
constexpr int BD_ITEMS_PER_SAMPLE = 2 * BD_MAX_DEPTH + (BD_MAX_DEPTH - 1) * BD_MAX_DEPTH / 2 + 2;   // >= the sum over s of the t-range at maxDepth d (pair_range): 2 d + (d - 1) d / 2 (90 at 12, 230 at 20)
constexpr unsigned BD_CHUNK = 1u << 21;            // most samples per chunk (sizeof(Sample) + nine item lists of BD_ITEMS_PER_SAMPLE x 4 B + the sums per sample: the launcher shrinks the
                                                   // chunk to what the device has free -- at BD_MAX_DEPTH 20 a sample costs 1.6x the record and 2.4x the lists it did at 12)

// The two subpaths of every sample (Path::alternatingRandomWalkFromPixel, path.cpp:548-631) with PERSISTENT lanes: the subpaths of a sample have between
// 3 and 25 vertices (Russian roulette), and with one sample per lane a wave took as long as its longest pair (40 % lane utilisation).  Here a
// lane walks through its share of the chunk's samples in ONE flat loop -- every iteration either starts the next sample (sample_sensor) or advances
// the current one by one step of each subpath (sample_next), exactly the body of walk_paths' loop -- so the lanes of a wave are at different
// samples but in the same code.  Which lane walks which sample does not matter: a sample's random numbers are its own, its record is recs[lid].
__global__ __launch_bounds__(TBLK, 2) void k_bd_paths(SceneD S, BdCam cam, BdConfig cfg, int x0, int y0, int x1, int y1, long long first, unsigned count, Sample *__restrict__ recs,
                                                   unsigned long long *__restrict__ stats)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    Ctx c;
    c.S = &S; c.V = hbm_scene_view(S); c.cam = cam; c.cfg = cfg; c.stack = s_stack + threadIdx.x; c.nClosest = c.nShadow = 0;
    const unsigned total = gridDim.x * TBLK;
    unsigned next = blockIdx.x * TBLK + threadIdx.x, lid = 0, done = 0;
    const int emitterDepth = cfg.maxDepth + (S.cam.thinlens ? 1 : 0), sensorDepth = cfg.maxDepth + (cfg.hittableEmitters ? 1 : 0);   // gbdpt_proc.cpp:110-122: one more emitter step unless the sensor is a point (pinhole); one more sensor step if an emitter can be hit
    bool have = false, walkT = false, walkS = false;
    int s = 0, t = 0;
    d3 thrS = mk(1.0), thrT = mk(1.0);
    while (true) {
        if (!have) {
            if (next >= count) break;
            lid = next; next += total;
            const int w = x1 - x0;
            const long long gid = first + lid;
            const int sIdx = (int)(gid % cfg.spp);
            const long long pix = gid / cfg.spp;
            const int px = x0 + (int)(pix % w), py = y0 + (int)(pix / w);
            c.rng.init(cfg.seed, (uint64_t)py * S.cam.width + px, (uint64_t)sIdx);
            if (S.cam.needsTime) (void)c.rng.next1D();                  // gbdpt_proc.cpp:156-157: the time sample comes first
            Sample &sm = recs[lid];
            bv_clear(sm.X[0]); sm.X[0].type = T_SENSOR_SUPER; sm.X[0].degenerate = 1;       // makeEndpoint, vertex.cpp:27-33
            bv_clear(sm.Y[0]); sm.Y[0].type = T_EMITTER_SUPER; sm.Y[0].degenerate = 0;
            sm.nX = 1; sm.nY = 1;
            t = sample_sensor(c, sm.X[0], px, py, sm.EX[0], sm.X[1], sm.EX[1], sm.X[2]);
            sm.nX = 1 + t;
            walkT = t == 2; walkS = true;
            thrS = mk(1.0); thrT = mk(1.0);
            s = 0;
            have = true;
        } else {
            Sample &sm = recs[lid];
            if (walkT && t < sensorDepth) {
                if (sample_next(c, sm.X[t], &sm.X[t - 1], &sm.EX[t - 1], sm.EX[t], sm.X[t + 1], ERadiance, cfg.rrDepth != -1 && t >= cfg.rrDepth, thrT)) { t++; sm.nX++; }
                else walkT = false;
            } else walkT = false;
            if (walkS && s < emitterDepth) {
                if (sample_next(c, sm.Y[s], s > 0 ? &sm.Y[s - 1] : nullptr, s > 0 ? &sm.EY[s - 1] : nullptr, sm.EY[s], sm.Y[s + 1], EImportance, cfg.rrDepth != -1 && s >= cfg.rrDepth, thrS)) { s++; sm.nY++; }
                else walkS = false;
            } else walkS = false;
            if (!(walkS || walkT)) {
                sm.posX = sm.X[1].u; sm.posY = sm.X[1].v;
                for (int k = 0; k < 4; k++) { sm.off[k].success = 0; sm.off[k].couldConnectAfterB = 0; sm.off[k].jacobian = 1.0; }
                if (sm.nY < 2) sm.nY = 0;                                                    // (no emitter could be sampled: a scene without power; no connections)
                have = false; done++;
            }
        }
    }
    const unsigned a = __builtin_amdgcn_wave_reduce_add_u32(c.nClosest, 0), b = __builtin_amdgcn_wave_reduce_add_u32(c.nShadow, 0);
    const unsigned n = __builtin_amdgcn_wave_reduce_add_u32(done, 0);
    if ((threadIdx.x & 63) == 0) { atomicAdd(stats + 0, (unsigned long long)a); atomicAdd(stats + 1, (unsigned long long)b); atomicAdd(stats + 2, (unsigned long long)n); }
}

// One lane per sample: the connected base path and its four offset paths, the prefix products (walk_shift), the sample's connections appended to
// the three item lists.
__global__ __launch_bounds__(TBLK, 2) void k_bd_shift(SceneD S, BdCam cam, BdConfig cfg, unsigned count, Sample *__restrict__ recs,
                                                   unsigned *__restrict__ items, size_t itemStride, unsigned *__restrict__ itemCount, Float *__restrict__ acc, unsigned long long *__restrict__ stats,
                                                   unsigned *__restrict__ genList, unsigned *__restrict__ genCount, unsigned *__restrict__ offList)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    const unsigned lid = blockIdx.x * TBLK + threadIdx.x;
    Ctx c;
    c.S = &S; c.V = hbm_scene_view(S); c.cam = cam; c.cfg = cfg; c.stack = s_stack + threadIdx.x; c.nClosest = c.nShadow = 0;
    if (lid < count) {
        Sample &sm = recs[lid];
        // a sample that met a specular vertex (conductor, dielectric, a rough conductor below shiftThreshold) goes to the general form
        // (gbdpt_general.hip.h, k_bd_general): no offsets, no connections here
        const bool general = sm.nY >= 2 && sample_needs_general(c, sm);
        if (general) genList[atomicAdd(genCount, 1u)] = lid;
        // (round 5: the four offset paths of the sample go to k_bd_offs, one lane each: genCount[1] counts them)
        if (sm.nY >= 2 && !general && walk_shift_base(c, sm)) { const unsigned at = atomicAdd(genCount + 1, 4u); for (unsigned k = 0; k < 4; k++) offList[at + k] = (lid << 2) | k; }
        for (int k = 0; k < 15; k++) acc[(size_t)lid * 15 + k] = 0.0;
        unsigned n = 0;
        if (!general)
        for (int s = sm.nY - 1; s >= 0; --s) { int minT, maxT; pair_range(cfg, sm.nX, s, minT, maxT); if (maxT >= minT) n += (unsigned)(maxT - minT + 1); }
        if (n) {
            // three item lists, by the shape of the work: light-tracing connections (t == 1: a base connection + four offset paths, each with its
            // own rays), connections inside or at the end of the shifted part (every path of the five evaluates its own BSDFs and densities),
            // connections beyond it (the offsets share everything with the base path but seven densities) -- a wave runs ONE of the three codes
            unsigned cnt[3] = {0, 0, 0};
            for (int s = sm.nY - 1; s >= 0; --s) {
                int minT, maxT;
                pair_range(cfg, sm.nX, s, minT, maxT);
                for (int t = maxT; t >= minT; --t) cnt[t == 1 ? 0 : (shares_connection(sm, t) ? 2 : 1)]++;
            }
            unsigned at[3];
            for (int q = 0; q < 3; q++) at[q] = cnt[q] ? atomicAdd(itemCount + q, cnt[q]) : 0u;
            for (int s = sm.nY - 1; s >= 0; --s) {
                int minT, maxT;
                pair_range(cfg, sm.nX, s, minT, maxT);
                for (int t = maxT; t >= minT; --t) {
                    const int q = t == 1 ? 0 : (shares_connection(sm, t) ? 2 : 1);
                    items[(size_t)q * itemStride + at[q]++] = (lid << 10) | ((unsigned)s << 5) | (unsigned)t;
                }
            }
        }
    }
    const unsigned a = __builtin_amdgcn_wave_reduce_add_u32(c.nClosest, 0), b = __builtin_amdgcn_wave_reduce_add_u32(c.nShadow, 0);
    if ((threadIdx.x & 63) == 0) { atomicAdd(stats + 0, (unsigned long long)a); atomicAdd(stats + 1, (unsigned long long)b); }
}

// One lane per (sample, offset path) of the fast form: generateOffsetPathGBDPT for a chain a - b - c of adjacent vertices (perturbed sensor direction, the new
// first vertex, its re-connection) and the offset path's prefix products -- four lanes per sample instead of one lane running the four in turn.
__global__ __launch_bounds__(TBLK, 2) void k_bd_offs(SceneD S, BdCam cam, BdConfig cfg, Sample *__restrict__ recs, const unsigned *__restrict__ offList, const unsigned *__restrict__ nOff,
                                                  unsigned long long *__restrict__ stats)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    Ctx c;
    c.S = &S; c.V = hbm_scene_view(S); c.cam = cam; c.cfg = cfg; c.stack = s_stack + threadIdx.x; c.nClosest = c.nShadow = 0;
    const unsigned n = __hip_atomic_load(nOff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned i = blockIdx.x * TBLK + threadIdx.x; i < n; i += gridDim.x * TBLK) {
        const unsigned it = offList[i];
        walk_shift_offset(c, recs[it >> 2], (int)(it & 3u));
    }
    const unsigned a = __builtin_amdgcn_wave_reduce_add_u32(c.nClosest, 0), b = __builtin_amdgcn_wave_reduce_add_u32(c.nShadow, 0);
    if ((threadIdx.x & 63) == 0) { atomicAdd(stats + 0, (unsigned long long)a); atomicAdd(stats + 1, (unsigned long long)b); }
}

// Three launches per class, each over the survivors of the one before (lists compacted with one atomic per wave):
//   PHASE 3  every connection of the class's item list: the part of the base path that needs no visibility ray of the connection;
//   PHASE 1  the base path: visibility, geometry term, MIS weight; adds the primal term;
//   PHASE 2  the four offsets of the connections whose base path carries anything: the gradient terms.
template <int CLS, int PHASE>
__global__ __launch_bounds__(TBLK, 2) void k_bd_connect(SceneD S, BdCam cam, BdConfig cfg, const Sample *__restrict__ recs, const unsigned *__restrict__ in, unsigned nIn,
                                                     const unsigned *__restrict__ nInDev, unsigned *__restrict__ out, unsigned *__restrict__ nOut, Float *__restrict__ acc,
                                                     Float *__restrict__ light, unsigned long long *__restrict__ stats)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    const unsigned i = blockIdx.x * TBLK + threadIdx.x;
    Ctx c;
    c.S = &S; c.V = hbm_scene_view(S); c.cam = cam; c.cfg = cfg; c.stack = s_stack + threadIdx.x; c.nClosest = c.nShadow = 0;
    const unsigned n = nInDev ? __hip_atomic_load(nInDev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : nIn;
    if (blockIdx.x * TBLK >= n) return;                                                     // (the grid is sized for the item list: its tail has nothing to do)
    bool keep = false;
    unsigned it = 0;
    if (i < n) {
        it = in[i];
        const unsigned lid = it >> 10;
        const int s = (int)((it >> 5) & 31u), t = (int)(it & 31u);
        PairOut po;
        if (connect_pair<CLS, PHASE>(c, recs[lid], s, t, po)) {
            keep = true;
            if (PHASE != 3) {
                const int W = S.cam.width, H = S.cam.height;
                const size_t plane3 = (size_t)W * H * 3;
                if (CLS != 0) {
                    Float *a = acc + (size_t)lid * 15;
                    if (PHASE == 1) { atomicAdd(a + 0, po.primal.x); atomicAdd(a + 1, po.primal.y); atomicAdd(a + 2, po.primal.z); }
                    else for (int g = 0; g < 4; g++) { atomicAdd(a + 3 + 3 * g, po.gradient[g].x); atomicAdd(a + 4 + 3 * g, po.gradient[g].y); atomicAdd(a + 5 + 3 * g, po.gradient[g].z); }
                } else
                    for (int k = 0; k < po.nLight; k++) film_put(light + po.light[k].buffer * plane3, 3, W, H, po.light[k].x, po.light[k].y, po.light[k].value, false, stats + 3);   // putLightSample, :514,525
            }
        }
    }
    // The survivors' list and the ray counters take ONE atomic per block, not per wave: a phase-3 launch of config 5's frame is 490 k waves that do little else, and
    // three atomics per wave on three fixed addresses held it at ~12 ns per wave whatever the work (5.8 ms of a 164 ms frame for the camera connections alone).
    __shared__ unsigned s_keep[TBLK / 64], s_base, s_rays[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long mask = PHASE != 2 ? __ballot(keep) : 0ULL;
    if (threadIdx.x < 2) s_rays[threadIdx.x] = 0;
    if (PHASE != 2 && lane == 0) s_keep[wv] = (unsigned)__popcll(mask);
    __syncthreads();
    if (PHASE != 3) {                                                                       // (phase 3 traces nothing)
        const unsigned a = __builtin_amdgcn_wave_reduce_add_u32(c.nClosest, 0), b = __builtin_amdgcn_wave_reduce_add_u32(c.nShadow, 0);
        if (lane == 0) { if (a) atomicAdd(&s_rays[0], a); if (b) atomicAdd(&s_rays[1], b); }
    }
    if (PHASE != 2 && threadIdx.x == 0) {
        unsigned total = 0;
        for (int w = 0; w < TBLK / 64; w++) total += s_keep[w];
        s_base = total ? atomicAdd(nOut, total) : 0u;
    }
    __syncthreads();
    if (PHASE != 2 && keep) {
        unsigned base = s_base;
        for (int w = 0; w < wv; w++) base += s_keep[w];
        out[base + __popcll(mask & ((1ULL << lane) - 1ULL))] = it;
    }
    if (PHASE != 3 && threadIdx.x == 0) { if (s_rays[0]) atomicAdd(stats + 0, (unsigned long long)s_rays[0]); if (s_rays[1]) atomicAdd(stats + 1, (unsigned long long)s_rays[1]); }
}

// The general form of a sample (specular chains; gbdpt_general.hip.h) in staged launches per pass over <= gsCap listed samples (round 5; round 4 ran a
// sample from its connected base path to its last connection in ONE lane with a 56 KB workspace: 80 % of a config-5 frame at 10 % lane utilisation):
//   k_bdg_shift    one lane per sample (persistent lanes, a manifold scratch each): the connected base path, its generalized geometry terms, the prefix
//                  products, the flag masks -> the sample's GSamp record; appends the sample's connections and its four offset paths to item lists;
//   k_bdg_offset   one lane per (sample, offset path), twice: the offset paths that enter a manifold walk and those that do not -- perturbation,
//                  propagation, walk, re-connection, Jacobians, generalized geometry terms, prefix products -> the record, read-only from here on;
//   k_bdg_connect  one lane per connection (s, t >= 2), ~25 per sample: reads the sample's record (lanes of a wave mostly share one), forms its MIS sums
//                  as recurrences over it, allocates nothing; adds to the sample's 15 sums;
//   k_bdg_light    one lane per light-tracing connection (s, 1): the only ones that build paths of their own (clones + four offset paths with manifold
//                  walks) -- in the lane's transient pool (persistent lanes); splats into the light images.
constexpr int GD_ITEMS = BD_ITEMS_PER_SAMPLE, GD_LIGHT = NEV + 3;       // connection items (t >= 2) / light items (<= NEV) per general sample
// gCount (16 counters of a pass): [0] connection items, [1] light items, [2] k_bdg_shift's cursor, [3] k_bdg_light<1>'s, [4] survivors of connection phase 3,
// [5] of phase 1, [6] [14] light offset-path items without / with a manifold walk, [7] [15] k_bdg_light<2>'s cursors into them, [8] [9] offset-path items without / with a
// manifold walk, [10] [11] their cursors, [12] light items that passed the ray-free test, [13] k_bdg_light<1>'s cursor.
__global__ __launch_bounds__(TBLK, 2) void k_bdg_shift(SceneD S, BdCam cam, BdConfig cfg, const Sample *__restrict__ recs, const unsigned *__restrict__ genList, unsigned first, unsigned count,
                                                    GSamp *__restrict__ gsamp, GScratch *__restrict__ scratch, unsigned *__restrict__ gItems, unsigned *__restrict__ gLight, unsigned *__restrict__ gOff,
                                                    size_t offStride, unsigned *__restrict__ gCount, unsigned long long *__restrict__ stats)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    const unsigned lane = blockIdx.x * TBLK + threadIdx.x;
    Ctx c;
    c.S = &S; c.V = hbm_scene_view(S); c.cam = cam; c.cfg = cfg; c.stack = s_stack + threadIdx.x; c.nClosest = c.nShadow = 0;
    unsigned overflow = 0, done = 0;
    for (unsigned i = atomicAdd(gCount + 2, 1u); i < count; i = atomicAdd(gCount + 2, 1u)) {
        const unsigned lid = genList[first + i];
        GSamp &W = gsamp[i];
        GTr g(c, W, &scratch[lane]);
        g.loadSubpaths(recs[lid]);
        g.prepareBase();
        W.lid = lid;
        done++;
        if (g.overflow) { overflow += g.overflow; W.voidSample = 1; continue; }              // (a pool ran out: the sample is void -- no connections, counted; asserted zero by tests and bench)
        if (g.hasOffsets()) {                                                              // its four offset paths: one lane each (k_bdg_offset), lists by "enters a manifold walk"
            const int q = g.offsetsWalk() ? 1 : 0;
            const unsigned at = atomicAdd(gCount + 8 + q, 4u);
            for (unsigned k = 0; k < 4; k++) gOff[(size_t)q * offStride + at + k] = (i << 2) | k;
        }
        unsigned nC = 0, nL = 0;
        for (int s = W.emitter.nv - 1; s >= 0; --s) {
            int minT, maxT;
            g.pairRange(s, minT, maxT);
            if (maxT < minT) continue;
            if (minT == 1) { nL++; minT = 2; }
            if (maxT >= minT) nC += (unsigned)(maxT - minT + 1);
        }
        unsigned atC = nC ? atomicAdd(gCount + 0, nC) : 0u, atL = nL ? atomicAdd(gCount + 1, nL) : 0u;
        for (int s = W.emitter.nv - 1; s >= 0; --s) {
            int minT, maxT;
            g.pairRange(s, minT, maxT);
            for (int t = maxT; t >= minT; --t) {
                const unsigned it = (i << 10) | ((unsigned)s << 5) | (unsigned)t;
                if (t == 1) gLight[atL++] = it; else gItems[atC++] = it;
            }
        }
    }
    const unsigned r0 = __builtin_amdgcn_wave_reduce_add_u32(c.nClosest, 0), r1 = __builtin_amdgcn_wave_reduce_add_u32(c.nShadow, 0);
    const unsigned r2 = __builtin_amdgcn_wave_reduce_add_u32(done, 0), r3 = __builtin_amdgcn_wave_reduce_add_u32(overflow, 0);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(stats + 0, (unsigned long long)r0); atomicAdd(stats + 1, (unsigned long long)r1);
        if (r2) atomicAdd(stats + 4, (unsigned long long)r2);
        if (r3) atomicAdd(stats + 5, (unsigned long long)r3);
    }
}

// One lane per (sample, offset path): ManifoldPerturbation::generateOffsetPathGBDPT, the path's half-Jacobians and generalized geometry terms, its prefix
// products (GTr::prepareOffset) -- into the sample's record, each lane its own slots and its own slice of the pool.  Two launches: the offset paths whose
// chain b..c holds specular vertices (a manifold walk, <= 2 x 20 Newton steps with re-traced chains) and the others, so that a wave runs one of the two.
__global__ __launch_bounds__(TBLK, 2) void k_bdg_offset(SceneD S, BdCam cam, BdConfig cfg, GSamp *__restrict__ gsamp, GScratch *__restrict__ scratch, const unsigned *__restrict__ in,
                                                     const unsigned *__restrict__ nIn, unsigned *__restrict__ cursor, unsigned long long *__restrict__ stats)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    const unsigned lane = blockIdx.x * TBLK + threadIdx.x;
    Ctx c;
    c.S = &S; c.V = hbm_scene_view(S); c.cam = cam; c.cfg = cfg; c.stack = s_stack + threadIdx.x; c.nClosest = c.nShadow = 0;
    const unsigned n = __hip_atomic_load(nIn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned overflow = 0;
    // (a wave takes 64 consecutive items = the four offset paths of 16 samples: the lanes of a sample share its base path)
    for (unsigned base = 0; ; ) {
        if ((threadIdx.x & 63) == 0) base = atomicAdd(cursor, 64u);
        base = __shfl(base, 0);
        if (base >= n) break;
        const unsigned i = base + (threadIdx.x & 63);
        if (i < n) {
            const unsigned it = in[i];
            GSamp &W = gsamp[it >> 2];
            GTr g(c, W, &scratch[lane]);
            g.setRegion((int)(it & 3u));
#ifdef GDPT_BD_PROFILE
            const unsigned long long tAll = clock64();
#endif
            g.prepareOffset((int)(it & 3u));
#ifdef GDPT_BD_PROFILE
            g.prof[5] = clock64() - tAll;
            for (int q = 0; q < 6; q++) atomicAdd(stats + 8 + q, g.prof[q]);
#endif
            if (g.overflow) { overflow += g.overflow; W.voidSample = 1; }
        }
    }
    const unsigned r0 = __builtin_amdgcn_wave_reduce_add_u32(c.nClosest, 0), r1 = __builtin_amdgcn_wave_reduce_add_u32(c.nShadow, 0), r3 = __builtin_amdgcn_wave_reduce_add_u32(overflow, 0);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(stats + 0, (unsigned long long)r0); atomicAdd(stats + 1, (unsigned long long)r1);
        if (r3) atomicAdd(stats + 5, (unsigned long long)r3);
    }
}

// Three launches, each over the survivors of the one before (lists compacted with one atomic per wave), as k_bd_connect's: PHASE 3 the ray-free part of
// the base path, PHASE 1 the base path (visibility, MIS weight; adds the primal term), PHASE 2 the four offsets (the gradient terms).
template <int PHASE>
__global__ __launch_bounds__(TBLK, 2) void k_bdg_connect(SceneD S, BdCam cam, BdConfig cfg, GSamp *__restrict__ gsamp, const unsigned *__restrict__ in, const unsigned *__restrict__ nIn,
                                                      unsigned *__restrict__ out, unsigned *__restrict__ nOut, Float *__restrict__ acc, unsigned long long *__restrict__ stats)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    Ctx c;
    c.S = &S; c.V = hbm_scene_view(S); c.cam = cam; c.cfg = cfg; c.stack = s_stack + threadIdx.x; c.nClosest = c.nShadow = 0;
    const unsigned n = __hip_atomic_load(nIn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned i = blockIdx.x * TBLK + threadIdx.x; i < n; i += gridDim.x * TBLK) {
        const unsigned it = in[i];
        GSamp &W = gsamp[it >> 10];
        GTrT<false> g(c, W, nullptr);
        PairOut po;
        const bool keep = !W.voidSample && g.connectPair<false, PHASE>((int)((it >> 5) & 31u), (int)(it & 31u), po);
        if (keep && PHASE != 3) {
            Float *a = acc + (size_t)W.lid * 15;
            if (PHASE == 1) { atomicAdd(a + 0, po.primal.x); atomicAdd(a + 1, po.primal.y); atomicAdd(a + 2, po.primal.z); }
            else for (int k = 0; k < 4; k++) { atomicAdd(a + 3 + 3 * k, po.gradient[k].x); atomicAdd(a + 4 + 3 * k, po.gradient[k].y); atomicAdd(a + 5 + 3 * k, po.gradient[k].z); }
        }
        if (PHASE != 2) {
            const unsigned long long mask = __ballot(keep);
            if (mask) {
                const int lane = threadIdx.x & 63, leader = __ffsll((unsigned long long)mask) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(nOut, (unsigned)__popcll(mask));
                base = __shfl(base, leader);
                if (keep) out[base + __popcll(mask & ((1ULL << lane) - 1ULL))] = it;
            }
        }
    }
    const unsigned r0 = __builtin_amdgcn_wave_reduce_add_u32(c.nClosest, 0), r1 = __builtin_amdgcn_wave_reduce_add_u32(c.nShadow, 0);
    if ((threadIdx.x & 63) == 0) { atomicAdd(stats + 0, (unsigned long long)r0); atomicAdd(stats + 1, (unsigned long long)r1); }
}

// PHASE 3: every light-tracing connection's ray-free test (the emitter vertex is connectable and inside the sensor's frustum: most are not) -> survivors.
// PHASE 1: the base path of a survivor (its own shiftable path: clones, the sensor connection's visibility ray, MIS weight); splats the primal term, lists
// the connections that carry anything, four items each -- one per offset path -- in one of two lists: offset paths that will walk a manifold, and the others.
// PHASE 2 (once per list, so that a wave runs one of the two): ONE offset path of a connection per lane (perturbed sensor direction, propagation, manifold walk,
// re-connection), the base path built again for what the offset shares with it (its rays not counted twice); splats the offset's gradient term.
template <int PHASE>
__global__ __launch_bounds__(TBLK, 2) void k_bdg_light(SceneD S, BdCam cam, BdConfig cfg, GSamp *__restrict__ gsamp, GScratch *__restrict__ scratch, const unsigned *__restrict__ in, const unsigned *__restrict__ nIn,
                                                    unsigned *__restrict__ cursor, unsigned *__restrict__ out, unsigned *__restrict__ nOut, unsigned *__restrict__ outWalk, unsigned *__restrict__ nOutWalk,
                                                    Float *__restrict__ light, unsigned long long *__restrict__ stats)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    const unsigned lane = blockIdx.x * TBLK + threadIdx.x;
    Ctx c;
    c.S = &S; c.V = hbm_scene_view(S); c.cam = cam; c.cfg = cfg; c.stack = s_stack + threadIdx.x; c.nClosest = c.nShadow = 0;
    const unsigned n = __hip_atomic_load(nIn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int Wd = S.cam.width, H = S.cam.height;
    const size_t plane3 = (size_t)Wd * H * 3;
    unsigned overflow = 0;
    // (PHASE 2 and 3: a wave takes 64 consecutive items -- in phase 2 the four offset paths of 16 connections, so that the lanes of a connection stay side by side;
    //  PHASE 1: a lane takes the next connection when it is done with its own: their work differs by whether the visibility ray is reached at all)
    unsigned wbase = 0;
    for (unsigned i = 0; ; ) {
        if (PHASE != 1) {
            if ((threadIdx.x & 63) == 0) wbase = atomicAdd(cursor, 64u);
            wbase = __shfl(wbase, 0);
            if (wbase >= n) break;
            i = wbase + (threadIdx.x & 63);
        } else { i = atomicAdd(cursor, 1u); if (i >= n) break; }
        bool keep = false;
        unsigned it = 0;
        if (i < n) {
            // an item: PHASE 3 / 1 (sample << 10) | (s << 5) | 1; PHASE 2 (sample << 7) | (s << 2) | (offset - 1)
            it = in[i];
            const unsigned smp = PHASE != 2 ? it >> 10 : it >> 7;
            const int es = PHASE != 2 ? (int)((it >> 5) & 31u) : (int)((it >> 2) & 31u);
            if (!gsamp[smp].voidSample) {
                GTr g(c, gsamp[smp], &scratch[lane]);
                PairOut po;
                if (PHASE == 2) g.lightK = (int)(it & 3u) + 1;
                const bool ok = g.connectPair<true, PHASE>(es, 1, po);
                if (g.overflow) overflow += g.overflow;
                else if (ok) {
                    keep = true;
                    if (PHASE == 1) {
                        unsigned *o = g.lightWalks ? outWalk : out;
                        const unsigned at = atomicAdd(g.lightWalks ? nOutWalk : nOut, 4u);
                        for (unsigned k = 0; k < 4; k++) o[at + k] = (smp << 7) | ((unsigned)es << 2) | k;
                    }
                    if (PHASE != 3)
                        for (int k = 0; k < po.nLight; k++) film_put(light + po.light[k].buffer * plane3, 3, Wd, H, po.light[k].x, po.light[k].y, po.light[k].value, false, stats + 3);   // putLightSample, :514,525
                }
            }
        }
        if (PHASE == 3) {
            const unsigned long long mask = __ballot(keep);
            if (mask) {
                const int ln = threadIdx.x & 63, leader = __ffsll((unsigned long long)mask) - 1;
                unsigned base = 0;
                if (ln == leader) base = atomicAdd(nOut, (unsigned)__popcll(mask));
                base = __shfl(base, leader);
                if (keep) out[base + __popcll(mask & ((1ULL << ln) - 1ULL))] = it;
            }
        }
    }
    const unsigned r0 = __builtin_amdgcn_wave_reduce_add_u32(c.nClosest, 0), r1 = __builtin_amdgcn_wave_reduce_add_u32(c.nShadow, 0), r3 = __builtin_amdgcn_wave_reduce_add_u32(overflow, 0);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(stats + 0, (unsigned long long)r0); atomicAdd(stats + 1, (unsigned long long)r1);
        if (r3) atomicAdd(stats + 5, (unsigned long long)r3);
    }
}

__global__ __launch_bounds__(TBLK) void k_bd_put(const Sample *__restrict__ recs, const Float *__restrict__ acc, unsigned count, int W, int H, Float *__restrict__ block, unsigned long long *__restrict__ stats)
{
    const unsigned lid = blockIdx.x * TBLK + threadIdx.x;
    if (lid >= count) return;
    const Float px = recs[lid].posX, py = recs[lid].posY;
    const Float *a = acc + (size_t)lid * 15;
    const size_t plane4 = (size_t)W * H * 4;
    for (int k = 0; k < 5; k++) film_put(block + k * plane4, 4, W, H, px, py, mk(a[3 * k], a[3 * k + 1], a[3 * k + 2]), true, stats + 3);   // putSample, gbdpt_proc.cpp:531-533
}

// probe: one sample -> primal(3), gradients(12), position(2), light splats (x, y, buffer, r, g, b), counters
__global__ __launch_bounds__(TBLK) void k_gbdpt_sample(SceneD S, BdCam cam, BdConfig cfg, int px, int py, int sample, Float *__restrict__ out17, int maxLight, Float *__restrict__ lightOut,
                                                       int *__restrict__ nLight, unsigned long long *__restrict__ counters, GWork *__restrict__ work)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Ctx c;
    c.S = &S; c.V = hbm_scene_view(S); c.cam = cam; c.cfg = cfg; c.stack = s_stack; c.nClosest = c.nShadow = 0;
    c.rng.init(cfg.seed, (uint64_t)py * S.cam.width + px, (uint64_t)sample);
    if (S.cam.needsTime) (void)c.rng.next1D();                          // gbdpt_proc.cpp:156-157
    Sample sm;
    SampleOut out;
    if (work) {                                     // (the host passes a workspace: the sample may need the general form)
        if (walk_paths(c, sm, px, py) && sample_needs_general(c, sm)) {
            GTr g(c, *work);
            g.loadSubpaths(sm);
            g.processSample(out);
            counters[2] = 1; counters[3] = g.overflow;
        } else {
            c.nClosest = c.nShadow = 0;
            c.rng.init(cfg.seed, (uint64_t)py * S.cam.width + px, (uint64_t)sample);
            if (S.cam.needsTime) (void)c.rng.next1D();
            process_sample(c, sm, px, py, out);
            counters[2] = 0; counters[3] = 0;
        }
    } else process_sample(c, sm, px, py, out);
    out17[0] = out.primal.x; out17[1] = out.primal.y; out17[2] = out.primal.z;
    for (int k = 0; k < 4; k++) { out17[3 + 3 * k] = out.gradient[k].x; out17[4 + 3 * k] = out.gradient[k].y; out17[5 + 3 * k] = out.gradient[k].z; }
    out17[15] = out.posX; out17[16] = out.posY;
    *nLight = out.nLight;
    for (int i = 0; i < out.nLight && i < maxLight; i++) {
        Float *o = lightOut + 6 * i;
        o[0] = out.light[i].x; o[1] = out.light[i].y; o[2] = out.light[i].buffer; o[3] = out.light[i].value.x; o[4] = out.light[i].value.y; o[5] = out.light[i].value.z;
    }
    counters[0] = c.nClosest; counters[1] = c.nShadow;
}

// GBDPTProcess::develop (gbdpt_proc.cpp:694-706) + MultiFilm::developMulti: the camera block is the film storage (setBitmapMulti), the light
// image is added scaled by weight / sampleCount (addBitmapMulti, multifilm.cpp:351-361: a pixel without camera samples gets weight 1), then
// rgb * (1 / weight) (fmtconv.cpp: invWeight = w != 0 ? 1 / w : w).  out: [H][W][3] doubles.
__global__ void k_gbdpt_develop(const Float *__restrict__ block, const Float *__restrict__ light, int npix, Float multiplier, Float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    Float wgt = block[4 * i + 3];
    if (wgt == 0) wgt = 1;
    const Float stored = wgt;
    wgt *= multiplier;
    const Float inv = stored != 0 ? 1.0 / stored : stored;
    for (int k = 0; k < 3; k++) out[3 * i + k] = (block[4 * i + k] + light[3 * i + k] * wgt) * inv;
}

} // namespace gdpt_bdk
using namespace gdpt_bdk;

struct gdpt_gbdpt_film {
    gdpt_scene *scene = nullptr;
    Float *block = nullptr, *light = nullptr;      // [5][H][W][4], [5][H][W][3]
    unsigned long long *stats = nullptr;           // closest rays, shadow rays, samples, invalid puts
    hipStream_t stream = nullptr, gstream = nullptr;   // gstream: the general form's launches of a chunk run beside the fast form's (they share nothing but atomics)
    hipEvent_t e0 = nullptr, e1 = nullptr, eG = nullptr;
    float renderMs = 0.0f;
    bool timed = false;
    int W = 0, H = 0;
    // workspace of the wavefront launches (allocated at the first render, sized for the largest chunk so far)
    Sample *recs = nullptr;
    unsigned *items = nullptr, *itemCount = nullptr;
    Float *acc = nullptr;
    unsigned capacity = 0;
    // the general form (specular chains): the samples that need it; one record per sample of a pass (k_bdg_shift writes it, the connection kernels
    // read it), one scratch per persistent lane of k_bdg_shift / k_bdg_light, the two item lists of a pass and their counters / cursors
    unsigned *genList = nullptr, *genCount = nullptr, *offList = nullptr;     // genCount: [0] entries of the general list, [1] of the fast form's offset-path list
    GSamp *gsamp = nullptr;
    GScratch *gscratch = nullptr;
    unsigned *gItems = nullptr, *gLight = nullptr, *gOff = nullptr, *gCount = nullptr;
    unsigned gsCap = 0, gLanes = 0, gsWant = 0;     // records of a pass, persistent lanes with a scratch, and the call size they were allocated for
    double sceneRadius = 0.0;
};

namespace {

// the G-BDPT path of this library carries connectable vertices only (gbdpt_kernels.hip.h): refuse what would need the specular-chain
// machinery (propagatePerturbation / manifoldWalk), and the emitters the bidirectional layer is not written for here
int check_scope(const gdpt_scene *s, const gdpt_gbdpt_config *cfg)
{
    if (cfg->maxDepth == 0 || cfg->maxDepth < -1) return bfail(GDPT_ERR_INVALID, "'maxDepth' must be set to -1 (infinite) or a value greater than zero!");   // gbdpt.cpp:102-103
    if (cfg->rrDepth <= 0) return bfail(GDPT_ERR_INVALID, "'rrDepth' must be set to a value greater than zero!");                                           // gbdpt.cpp:99-100
    if (cfg->maxDepth > BD_MAX_DEPTH) return bfail(GDPT_ERR_UNSUPPORTED, "G-BDPT: maxDepth up to %d (a sample's record holds both subpaths; -1 renders as 12, gbdpt_proc.cpp:103-106)", BD_MAX_DEPTH);
    if (cfg->spp <= 0) return bfail(GDPT_ERR_INVALID, "G-BDPT: spp must be positive");
    if (s->cropped) return bfail(GDPT_ERR_UNSUPPORTED, "G-BDPT: a film with a crop window is not carried (the sensor's importance and the light image take crop == film)");
    if (s->d.cam.thinlens && (cfg->maxDepth < 0 ? BD_DEFAULT_DEPTH : cfg->maxDepth) > BD_MAX_DEPTH - 1)   // (the extra emitter step of a non-degenerate sensor, gbdpt_proc.cpp:117-118, needs one more record)
        return bfail(GDPT_ERR_UNSUPPORTED, "G-BDPT: maxDepth up to %d with the thinlens sensor", BD_MAX_DEPTH - 1);
    // (round 4: Dirac BSDFs and rough conductors below shiftThreshold are carried -- samples that meet one run the general form, gbdpt_general.hip.h)
    return GDPT_OK;
}

BdCam make_cam(const gdpt_scene *s)
{
    const CameraD &cd = s->d.cam;
    BdCam cam;
    const double a = cd.m[0], b = cd.m[1], cc = cd.m[2], d = cd.m[4], e = cd.m[5], f = cd.m[6], g = cd.m[8], h = cd.m[9], i = cd.m[10];
    const double A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;                    // adjugate / determinant in a fixed operation order
    const double det = a * A + b * B + cc * C, r = 1.0 / det;
    cam.invLin[0] = A * r; cam.invLin[1] = (cc * h - b * i) * r; cam.invLin[2] = (b * f - cc * e) * r;
    cam.invLin[3] = B * r; cam.invLin[4] = (a * i - cc * g) * r; cam.invLin[5] = (cc * d - a * f) * r;
    cam.invLin[6] = C * r; cam.invLin[7] = (b * g - a * h) * r; cam.invLin[8] = (a * e - b * d) * r;
    cam.pos.x = cd.m[3]; cam.pos.y = cd.m[7]; cam.pos.z = cd.m[11];
    cam.dir.x = cd.m[2]; cam.dir.y = cd.m[6]; cam.dir.z = cd.m[10];
    cam.rectX = cd.tanHalf; cam.rectY = cd.tanHalf / cd.aspect;
    cam.normalization = 1.0 / ((2 * cam.rectX) * (2 * cam.rectY));
    cam.aperturePdf = cd.thinlens ? 1 / (GD_PI * cd.apertureRadius * cd.apertureRadius) : 0.0;
    return cam;
}

BdConfig make_cfg(const gdpt_gbdpt_config *cfg, double sceneRadius, bool hittableEmitters)
{
    BdConfig c;
    c.sceneRadius = sceneRadius;
    c.hittableEmitters = hittableEmitters ? 1 : 0;
    c.maxDepth = cfg->maxDepth == -1 ? BD_DEFAULT_DEPTH : cfg->maxDepth;                    // gbdpt_proc.cpp:103-106
    c.rrDepth = cfg->rrDepth; c.lightImage = cfg->lightImage ? 1 : 0; c.spp = cfg->spp;
    c.shiftThreshold = cfg->shiftThreshold; c.seed = cfg->seed;
    c.sBase = 0; c.sCount = cfg->spp;
    return c;
}

} // namespace

extern "C" {

int gdpt_gbdpt_film_create(gdpt_scene *s, gdpt_gbdpt_film **out)
{
    if (!s || !out) return bfail(GDPT_ERR_INVALID, "gbdpt_film_create: null argument");
    BHIPCHK(hipSetDevice(s->device));
    gdpt_gbdpt_film *f = new gdpt_gbdpt_film;
    f->scene = s; f->W = s->d.cam.width; f->H = s->d.cam.height;
    const size_t npix = (size_t)f->W * f->H;
    BHIPCHK(hipMalloc((void **)&f->block, sizeof(Float) * 5 * npix * 4));
    BHIPCHK(hipMalloc((void **)&f->light, sizeof(Float) * 5 * npix * 3));
    BHIPCHK(hipMalloc((void **)&f->stats, sizeof(unsigned long long) * 16));           // [0..3] the public counters; [4] samples run in the general form, [5] workspace overflows
    BHIPCHK(hipMalloc((void **)&f->genCount, sizeof(unsigned) * 2));               // entries of the general list
    BHIPCHK(hipMalloc((void **)&f->gCount, sizeof(unsigned) * 16));                // the counters and cursors of a pass (k_bdg_shift)
    f->sceneRadius = s->bsphereRadius;              // m_scene->getBSphere().radius (gpt_capi.hip: kd-tree bounds + sensor + emitters, scene.cpp:386-413)
    BHIPCHK(hipStreamCreate(&f->stream));
    // The general form's stream is the frame's critical path (persistent lanes with long tails: ~93 ms of kernels per 2 spp frame of config 5's scene against ~58 ms of
    // the fast form's connection launches beside it): it gets the device's highest stream priority, so its workgroups are placed first and the fast form's dense grids
    // fill what is left.  GDPT_BD_GSTREAM_PRIORITY=0: both streams at the default priority (the A/B switch of the measurement in DESIGN.md).
    {
        int least = 0, greatest = 0;
        const char *e = getenv("GDPT_BD_GSTREAM_PRIORITY");
        if ((e && atoi(e) == 0) || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || greatest == least) { (void)hipGetLastError(); BHIPCHK(hipStreamCreate(&f->gstream)); }
        else BHIPCHK(hipStreamCreateWithPriority(&f->gstream, hipStreamDefault, greatest));
    }
    BHIPCHK(hipEventCreate(&f->e0)); BHIPCHK(hipEventCreate(&f->e1)); BHIPCHK(hipEventCreateWithFlags(&f->eG, hipEventDisableTiming));
    *out = f;
    return gdpt_gbdpt_film_clear(f);
}

void gdpt_gbdpt_film_destroy(gdpt_gbdpt_film *f)
{
    if (!f) return;
    hipSetDevice(f->scene->device);
    if (f->stream) hipStreamSynchronize(f->stream);
    if (f->gstream) { hipStreamSynchronize(f->gstream); hipStreamDestroy(f->gstream); }
    if (f->eG) hipEventDestroy(f->eG);
    hipFree(f->block); hipFree(f->light); hipFree(f->stats);
    hipFree(f->recs); hipFree(f->items); hipFree(f->itemCount); hipFree(f->acc);
    hipFree(f->genList); hipFree(f->offList); hipFree(f->genCount); hipFree(f->gsamp); hipFree(f->gscratch); hipFree(f->gItems); hipFree(f->gLight); hipFree(f->gOff); hipFree(f->gCount);
    if (f->e0) hipEventDestroy(f->e0);
    if (f->e1) hipEventDestroy(f->e1);
    if (f->stream) hipStreamDestroy(f->stream);
    delete f;
}

int gdpt_gbdpt_film_clear(gdpt_gbdpt_film *f)
{
    if (!f) return bfail(GDPT_ERR_INVALID, "gbdpt_film_clear: null film");
    BHIPCHK(hipSetDevice(f->scene->device));
    const size_t npix = (size_t)f->W * f->H;
    BHIPCHK(hipMemsetAsync(f->block, 0, sizeof(Float) * 5 * npix * 4, f->stream));
    BHIPCHK(hipMemsetAsync(f->light, 0, sizeof(Float) * 5 * npix * 3, f->stream));
    BHIPCHK(hipMemsetAsync(f->stats, 0, sizeof(unsigned long long) * 16, f->stream));
    f->renderMs = 0.0f; f->timed = false;
    return GDPT_OK;
}

int gdpt_gbdpt_render_rect(gdpt_scene *s, const gdpt_gbdpt_config *cfg, int x0, int y0, int x1, int y1, gdpt_gbdpt_film *f)
{
    if (!s || !cfg || !f || f->scene != s) return bfail(GDPT_ERR_INVALID, "gbdpt_render_rect: null argument, or a film of another scene");
    if (x0 < 0 || y0 < 0 || x1 > f->W || y1 > f->H || x0 >= x1 || y0 >= y1) return bfail(GDPT_ERR_INVALID, "gbdpt_render_rect: rectangle outside the film");
    if (int rc = check_scope(s, cfg)) return rc;
    BHIPCHK(hipSetDevice(s->device));
    const BdCam cam = make_cam(s);
    BdConfig c = make_cfg(cfg, f->sceneRadius, s->hittableEmitters);
    if (f->timed) { float ms = 0; BHIPCHK(hipEventSynchronize(f->e1)); BHIPCHK(hipEventElapsedTime(&ms, f->e0, f->e1)); f->renderMs += ms; f->timed = false; }
    BHIPCHK(hipEventRecord(f->e0, f->stream));
    const long long pixels = (long long)(x1 - x0) * (y1 - y0), total = pixels * c.spp;
    // The chunk: at most BD_CHUNK samples, at most what fits 40 % of the memory the device has free right now (+ what this film already holds) --
    // several films on one GPU (strips wrapped onto a device, a G-PT film resident beside this one) or a partitioned / smaller part each get a
    // share instead of failing -- halved again while the allocation itself fails.  GDPT_BD_CHUNK forces a size (tests of the chunk loop).
    const size_t perSample = sizeof(Sample) + sizeof(unsigned) * 9 * BD_ITEMS_PER_SAMPLE + sizeof(Float) * 15 + 5 * sizeof(unsigned);   // record + three item lists with two survivor lists each + sums + general-list entry + four offset-path items
    // the general form's memory (only scenes that can produce a specular vertex need it): a scratch per persistent lane (two 256-thread blocks per CU:
    // 131 072 x 60 KB = 7.9 GB on 256 CUs) and the records + item lists of a pass (8 samples per lane: 1 048 576 x 55 KB = 58 GB of the 288: a chunk of
    // 2 M samples of the specular Veach scene lists 650 k -- one pass; with half of that it was a full pass and a quarter-full one, and every pass pays the
    // tails of its persistent kernels) -- at most a third of what the device has free, halved until it fits
    bool specularScene = false;
    for (const MaterialD &m : s->hostMats) if (m.type == 1 || m.type == 3 || (m.type == 2 && 0.5 * (m.alphaU + m.alphaV) < cfg->shiftThreshold)) specularScene = true;
    // ... and never more than THIS call can use: a chunk lists at most its own samples (a 24 x 18 film must not take 65 GB -- sixteen fuzz processes on one GPU
    // found that out), a lane of the persistent kernels has at most four items per listed sample to fetch.  A later, larger call re-allocates.
    const unsigned wantCap = (unsigned)std::min<long long>((long long)8 * s->numCUs * 2 * TBLK, std::max<long long>(1, std::min<long long>(total, BD_CHUNK)));
    const unsigned wantLanes = (unsigned)std::min<long long>((long long)s->numCUs * 2 * TBLK, ((long long)4 * wantCap + TBLK - 1) / TBLK * TBLK);
    if (specularScene && (!f->gsamp || (f->gsWant < wantCap && !getenv("GDPT_BD_GENERAL_PASS")))) {
        BHIPCHK(hipStreamSynchronize(f->stream)); BHIPCHK(hipStreamSynchronize(f->gstream));
        hipFree(f->gscratch); hipFree(f->gsamp); hipFree(f->gItems); hipFree(f->gLight); hipFree(f->gOff);
        f->gscratch = nullptr; f->gsamp = nullptr; f->gItems = nullptr; f->gLight = nullptr; f->gOff = nullptr;
        size_t freeB = 0, totalB = 0;
        size_t budget = (size_t)96 << 30;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) budget = std::min(budget, freeB / 3);
        unsigned lanes = wantLanes, cap = wantCap;
        if (const char *e = getenv("GDPT_BD_GENERAL_PASS")) cap = (unsigned)std::max<long long>(1, atoll(e));      // (tests of the pass loop)
        const size_t perSampleG = sizeof(GSamp) + sizeof(unsigned) * (3 * GD_ITEMS + 10 * GD_LIGHT + 8);  // record + the connection list with its two survivor lists + the light lists + the two offset-path lists
        for (;;) {
            while ((size_t)lanes * sizeof(GScratch) + (size_t)cap * perSampleG > budget && (lanes > TBLK || cap > TBLK)) { if (cap > lanes) cap /= 2; else lanes = std::max<unsigned>(TBLK, lanes / 2 / TBLK * TBLK); }
            if (hipMalloc((void **)&f->gscratch, sizeof(GScratch) * (size_t)lanes) == hipSuccess && hipMalloc((void **)&f->gsamp, sizeof(GSamp) * (size_t)cap) == hipSuccess &&
                hipMalloc((void **)&f->gItems, sizeof(unsigned) * 3 * GD_ITEMS * (size_t)cap) == hipSuccess && hipMalloc((void **)&f->gLight, sizeof(unsigned) * 10 * GD_LIGHT * (size_t)cap) == hipSuccess &&
                hipMalloc((void **)&f->gOff, sizeof(unsigned) * 8 * (size_t)cap) == hipSuccess) break;
            (void)hipGetLastError();
            hipFree(f->gscratch); hipFree(f->gsamp); hipFree(f->gItems); hipFree(f->gLight); hipFree(f->gOff);
            f->gscratch = nullptr; f->gsamp = nullptr; f->gItems = nullptr; f->gLight = nullptr; f->gOff = nullptr;
            if (budget <= ((size_t)64 << 20)) return bfail(GDPT_ERR_HIP, "Out of memory! (G-BDPT general-form records: %.1f MB)", ((double)lanes * sizeof(GScratch) + (double)cap * perSampleG) / 1e6);
            budget /= 2;
        }
        f->gLanes = lanes; f->gsCap = cap; f->gsWant = wantCap;
    }
    unsigned chunk = (unsigned)std::min<long long>(total, BD_CHUNK);
    if (const char *e = getenv("GDPT_BD_CHUNK")) chunk = (unsigned)std::max<long long>(1, std::min<long long>(chunk, atoll(e)));
    else {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            const size_t budget = std::max<size_t>((size_t)64 << 20, (size_t)(0.4 * (double)(freeB + (size_t)f->capacity * perSample)));
            chunk = (unsigned)std::max<size_t>(1, std::min<size_t>(chunk, budget / perSample));
        }
    }
    if (chunk > f->capacity || (getenv("GDPT_BD_CHUNK") && chunk != f->capacity)) {
        BHIPCHK(hipStreamSynchronize(f->stream));
        for (;;) {
            hipFree(f->recs); hipFree(f->items); hipFree(f->itemCount); hipFree(f->acc); hipFree(f->genList); hipFree(f->offList);
            f->recs = nullptr; f->items = nullptr; f->itemCount = nullptr; f->acc = nullptr; f->genList = nullptr; f->offList = nullptr; f->capacity = 0;
            if (hipMalloc((void **)&f->genList, sizeof(unsigned) * (size_t)chunk) == hipSuccess && hipMalloc((void **)&f->offList, sizeof(unsigned) * 4 * (size_t)chunk) == hipSuccess &&
                hipMalloc((void **)&f->recs, sizeof(Sample) * (size_t)chunk) == hipSuccess && hipMalloc((void **)&f->items, sizeof(unsigned) * 9 * (size_t)chunk * BD_ITEMS_PER_SAMPLE) == hipSuccess &&
                hipMalloc((void **)&f->itemCount, sizeof(unsigned) * 9) == hipSuccess && hipMalloc((void **)&f->acc, sizeof(Float) * 15 * (size_t)chunk) == hipSuccess) break;
            (void)hipGetLastError();
            if (chunk <= 1024) {
                hipFree(f->recs); hipFree(f->items); hipFree(f->itemCount); hipFree(f->acc); hipFree(f->genList); hipFree(f->offList);
                f->recs = nullptr; f->items = nullptr; f->itemCount = nullptr; f->acc = nullptr; f->genList = nullptr; f->offList = nullptr;
                return bfail(GDPT_ERR_HIP, "Out of memory! (G-BDPT workspace for %u samples: %.1f MB)", chunk, (double)perSample * chunk / 1e6);
            }
            chunk /= 2;
        }
        f->capacity = chunk;
    } else chunk = std::min<unsigned>(chunk, f->capacity);
    for (long long first = 0; first < total; first += chunk) {
        const unsigned count = (unsigned)std::min<long long>(chunk, total - first);
        const size_t itemStride = (size_t)f->capacity * BD_ITEMS_PER_SAMPLE;
        BHIPCHK(hipMemsetAsync(f->itemCount, 0, sizeof(unsigned) * 9, f->stream));
        BHIPCHK(hipMemsetAsync(f->genCount, 0, sizeof(unsigned) * 2, f->stream));
        const unsigned pgrid = std::min<unsigned>((count + TBLK - 1) / TBLK, (unsigned)s->numCUs * 2u);      // persistent: the grid that is resident at 2 waves per SIMD
        hipLaunchKernelGGL(k_bd_paths, dim3(pgrid), dim3(TBLK), 0, f->stream, s->d, cam, c, x0, y0, x1, y1, first, count, f->recs, f->stats);
        hipLaunchKernelGGL(k_bd_shift, dim3((count + TBLK - 1) / TBLK), dim3(TBLK), 0, f->stream, s->d, cam, c, count, f->recs, f->items, itemStride, f->itemCount, f->acc, f->stats, f->genList, f->genCount, f->offList);
        hipLaunchKernelGGL(k_bd_offs, dim3(std::min<unsigned>((4 * count + TBLK - 1) / TBLK, (unsigned)s->numCUs * 32u)), dim3(TBLK), 0, f->stream, s->d, cam, c, f->recs, (const unsigned *)f->offList, (const unsigned *)(f->genCount + 1), f->stats);
        BHIPCHK(hipGetLastError());
        unsigned nItems[3] = {0, 0, 0}, nGen = 0;
        BHIPCHK(hipMemcpyAsync(nItems, f->itemCount, sizeof(unsigned) * 3, hipMemcpyDeviceToHost, f->stream));
        BHIPCHK(hipMemcpyAsync(&nGen, f->genCount, sizeof(unsigned), hipMemcpyDeviceToHost, f->stream));
        BHIPCHK(hipStreamSynchronize(f->stream));                                          // the sizes of the connection launches come from the walk
        if (nGen && !f->gsamp) return bfail(GDPT_ERR_HIP, "G-BDPT: a sample needs the general form in a scene without a specular material");
        // The general form of this chunk's listed samples on a stream of its own, BESIDE the fast form's connection launches below: the two share the
        // chunk's read-only records and otherwise only atomics (sums of disjoint samples, the light images, the counters).  Its kernels are persistent
        // lanes with long tails (a manifold walk is 10-100x an offset path without one); the fast form's dense launches fill the CUs those tails leave
        // idle.  GDPT_BD_NO_OVERLAP=1: one stream (the A/B switch of the measurement in DESIGN.md).
        hipStream_t gs = getenv("GDPT_BD_NO_OVERLAP") ? f->stream : f->gstream;
        const unsigned gPasses = nGen ? (nGen + f->gsCap - 1) / f->gsCap : 0u, gPer = gPasses ? (nGen + gPasses - 1) / gPasses : 0u;   // passes of equal size
        for (unsigned gFirst = 0; gFirst < nGen; gFirst += gPer) {                        // the general form, a pass of <= gsCap samples at a time
            const unsigned gN = std::min(gPer, nGen - gFirst);
            BHIPCHK(hipMemsetAsync(f->gCount, 0, sizeof(unsigned) * 16, gs));
            const unsigned lgrid = std::min((gN + TBLK - 1) / TBLK, f->gLanes / TBLK);
            const size_t offStride = (size_t)4 * f->gsCap;
            hipLaunchKernelGGL(k_bdg_shift, dim3(lgrid), dim3(TBLK), 0, gs, s->d, cam, c, f->recs, f->genList, gFirst, gN, f->gsamp, f->gscratch, f->gItems, f->gLight, f->gOff, offStride, f->gCount, f->stats);
            const dim3 ogrid((unsigned)std::min(((size_t)gN * 4 + TBLK - 1) / TBLK, (size_t)f->gLanes / TBLK));
            for (int q = 0; q < 2; q++)
                hipLaunchKernelGGL(k_bdg_offset, ogrid, dim3(TBLK), 0, gs, s->d, cam, c, f->gsamp, f->gscratch, (const unsigned *)(f->gOff + q * offStride), (const unsigned *)(f->gCount + 8 + q), f->gCount + 10 + q, f->stats);
            const unsigned cgridG = (unsigned)std::min<size_t>(((size_t)gN * 24 + TBLK - 1) / TBLK, (size_t)s->numCUs * 16);   // (grid-stride over the list, whose length only the device knows)
            unsigned *listA = f->gItems + (size_t)GD_ITEMS * f->gsCap, *listB = listA + (size_t)GD_ITEMS * f->gsCap;
            hipLaunchKernelGGL(k_bdg_connect<3>, dim3(cgridG), dim3(TBLK), 0, gs, s->d, cam, c, f->gsamp, (const unsigned *)f->gItems, (const unsigned *)(f->gCount + 0), listA, f->gCount + 4, f->acc, f->stats);
            hipLaunchKernelGGL(k_bdg_connect<1>, dim3(cgridG), dim3(TBLK), 0, gs, s->d, cam, c, f->gsamp, (const unsigned *)listA, (const unsigned *)(f->gCount + 4), listB, f->gCount + 5, f->acc, f->stats);
            hipLaunchKernelGGL(k_bdg_connect<2>, dim3(cgridG), dim3(TBLK), 0, gs, s->d, cam, c, f->gsamp, (const unsigned *)listB, (const unsigned *)(f->gCount + 5), (unsigned *)nullptr, (unsigned *)nullptr, f->acc, f->stats);
            const dim3 lgridL((unsigned)std::min(((size_t)gN * 4 + TBLK - 1) / TBLK, (size_t)f->gLanes / TBLK));
            // light lists: [0] every light-tracing connection, [1] the survivors of the ray-free test, [2] / [3] offset-path items without / with a manifold walk (x 4)
            unsigned *lightF = f->gLight + (size_t)GD_LIGHT * f->gsCap, *lightB = lightF + (size_t)GD_LIGHT * f->gsCap, *lightW = lightB + (size_t)4 * GD_LIGHT * f->gsCap;
            hipLaunchKernelGGL(k_bdg_light<3>, lgridL, dim3(TBLK), 0, gs, s->d, cam, c, f->gsamp, f->gscratch, (const unsigned *)f->gLight, (const unsigned *)(f->gCount + 1), f->gCount + 3, lightF, f->gCount + 12,
                               (unsigned *)nullptr, (unsigned *)nullptr, f->light, f->stats);
            hipLaunchKernelGGL(k_bdg_light<1>, lgridL, dim3(TBLK), 0, gs, s->d, cam, c, f->gsamp, f->gscratch, (const unsigned *)lightF, (const unsigned *)(f->gCount + 12), f->gCount + 13, lightB, f->gCount + 6,
                               lightW, f->gCount + 14, f->light, f->stats);
            hipLaunchKernelGGL(k_bdg_light<2>, lgridL, dim3(TBLK), 0, gs, s->d, cam, c, f->gsamp, f->gscratch, (const unsigned *)lightB, (const unsigned *)(f->gCount + 6), f->gCount + 7, (unsigned *)nullptr, (unsigned *)nullptr,
                               (unsigned *)nullptr, (unsigned *)nullptr, f->light, f->stats);
            hipLaunchKernelGGL(k_bdg_light<2>, lgridL, dim3(TBLK), 0, gs, s->d, cam, c, f->gsamp, f->gscratch, (const unsigned *)lightW, (const unsigned *)(f->gCount + 14), f->gCount + 15, (unsigned *)nullptr, (unsigned *)nullptr,
                               (unsigned *)nullptr, (unsigned *)nullptr, f->light, f->stats);
            BHIPCHK(hipGetLastError());
        }
        if (nGen && gs != f->stream) { BHIPCHK(hipEventRecord(f->eG, gs)); }
        for (int q = 0; q < 3; q++) {
            if (!nItems[q]) continue;
            const dim3 cgrid((nItems[q] + TBLK - 1) / TBLK);
            const unsigned *list = f->items + q * itemStride;
            unsigned *listA = f->items + (3 + q) * itemStride, *nA = f->itemCount + 3 + q, *listB = f->items + (6 + q) * itemStride, *nB = f->itemCount + 6 + q;
#define BD_CONNECT(CLSV) do { \
                hipLaunchKernelGGL((k_bd_connect<CLSV, 3>), cgrid, dim3(TBLK), 0, f->stream, s->d, cam, c, f->recs, list, nItems[q], (const unsigned *)nullptr, listA, nA, f->acc, f->light, f->stats); \
                hipLaunchKernelGGL((k_bd_connect<CLSV, 1>), cgrid, dim3(TBLK), 0, f->stream, s->d, cam, c, f->recs, (const unsigned *)listA, nItems[q], (const unsigned *)nA, listB, nB, f->acc, f->light, f->stats); \
                hipLaunchKernelGGL((k_bd_connect<CLSV, 2>), cgrid, dim3(TBLK), 0, f->stream, s->d, cam, c, f->recs, (const unsigned *)listB, nItems[q], (const unsigned *)nB, (unsigned *)nullptr, (unsigned *)nullptr, f->acc, f->light, f->stats); } while (0)
            if (q == 0) { BD_CONNECT(0);      // light tracing: its phase 3 is the test whether the sensor sees the emitter vertex at all
            } else if (q == 1) BD_CONNECT(1); else BD_CONNECT(2);
#undef BD_CONNECT
            BHIPCHK(hipGetLastError());
        }
        if (nGen && gs != f->stream) BHIPCHK(hipStreamWaitEvent(f->stream, f->eG, 0));        // the sums of the general samples are complete
        hipLaunchKernelGGL(k_bd_put, dim3((count + TBLK - 1) / TBLK), dim3(TBLK), 0, f->stream, f->recs, f->acc, count, f->W, f->H, f->block, f->stats);
        BHIPCHK(hipGetLastError());
    }
    BHIPCHK(hipEventRecord(f->e1, f->stream));
    f->timed = true;
    return GDPT_OK;
}

int gdpt_gbdpt_film_sync(gdpt_gbdpt_film *f)
{
    if (!f) return bfail(GDPT_ERR_INVALID, "gbdpt_film_sync: null film");
    BHIPCHK(hipSetDevice(f->scene->device));
    BHIPCHK(hipStreamSynchronize(f->stream));
    if (f->timed) { float ms = 0; BHIPCHK(hipEventElapsedTime(&ms, f->e0, f->e1)); f->renderMs += ms; f->timed = false; }
    return GDPT_OK;
}

float gdpt_gbdpt_film_render_ms(gdpt_gbdpt_film *f)
{
    if (!f || gdpt_gbdpt_film_sync(f) != GDPT_OK) return -1.0f;
    return f->renderMs;
}

int gdpt_gbdpt_film_accum(gdpt_gbdpt_film *f, double *block, double *light)
{
    if (!f || !block || !light) return bfail(GDPT_ERR_INVALID, "gbdpt_film_accum: null argument");
    if (int rc = gdpt_gbdpt_film_sync(f)) return rc;
    const size_t npix = (size_t)f->W * f->H;
    BHIPCHK(hipMemcpy(block, f->block, sizeof(Float) * 5 * npix * 4, hipMemcpyDeviceToHost));
    BHIPCHK(hipMemcpy(light, f->light, sizeof(Float) * 5 * npix * 3, hipMemcpyDeviceToHost));
    return GDPT_OK;
}

int gdpt_gbdpt_film_develop_device(gdpt_gbdpt_film *f, int buffer, int spp, double *rgbDevice)
{
    if (!f || !rgbDevice || buffer < 0 || buffer > 4 || spp <= 0) return bfail(GDPT_ERR_INVALID, "gbdpt_film_develop: bad argument");
    BHIPCHK(hipSetDevice(f->scene->device));
    const int npix = f->W * f->H;
    hipLaunchKernelGGL(k_gbdpt_develop, dim3((npix + 255) / 256), dim3(256), 0, f->stream, f->block + (size_t)buffer * npix * 4, f->light + (size_t)buffer * npix * 3, npix,
                       (Float)(1.0 / spp), rgbDevice);
    BHIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_gbdpt_film_develop(gdpt_gbdpt_film *f, int buffer, int spp, double *rgbHost)
{
    if (!f || !rgbHost) return bfail(GDPT_ERR_INVALID, "gbdpt_film_develop: null argument");
    BHIPCHK(hipSetDevice(f->scene->device));
    const size_t n = (size_t)f->W * f->H * 3;
    double *d = nullptr;
    BHIPCHK(hipMalloc((void **)&d, sizeof(double) * n));
    int rc = gdpt_gbdpt_film_develop_device(f, buffer, spp, d);
    if (rc == GDPT_OK) { hipError_t e = hipStreamSynchronize(f->stream); if (e == hipSuccess) e = hipMemcpy(rgbHost, d, sizeof(double) * n, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = bfail(GDPT_ERR_HIP, "gbdpt_film_develop: %s", hipGetErrorString(e)); }
    hipFree(d);
    return rc;
}

int gdpt_gbdpt_film_stats(gdpt_gbdpt_film *f, unsigned long long stats[4])
{
    if (!f || !stats) return bfail(GDPT_ERR_INVALID, "gbdpt_film_stats: null argument");
    if (int rc = gdpt_gbdpt_film_sync(f)) return rc;
    BHIPCHK(hipMemcpy(stats, f->stats, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost));
    return GDPT_OK;
}

// the raw sums as device-to-device copies, out of and into the film: what a multi-GPU host reduces between the films of its ranks (the light
// image of a rank holds splats for EVERY pixel, gbdpt_wr.cpp:45-52 -- the reference merges whole work results by addition, :57-63)
int gdpt_gbdpt_film_export_device(gdpt_gbdpt_film *f, double *blockDevice, double *lightDevice)
{
    if (!f || !blockDevice || !lightDevice) return bfail(GDPT_ERR_INVALID, "gbdpt_film_export_device: null argument");
    BHIPCHK(hipSetDevice(f->scene->device));
    const size_t npix = (size_t)f->W * f->H;
    BHIPCHK(hipMemcpyAsync(blockDevice, f->block, sizeof(Float) * 5 * npix * 4, hipMemcpyDeviceToDevice, f->stream));
    BHIPCHK(hipMemcpyAsync(lightDevice, f->light, sizeof(Float) * 5 * npix * 3, hipMemcpyDeviceToDevice, f->stream));
    BHIPCHK(hipStreamSynchronize(f->stream));
    return GDPT_OK;
}

int gdpt_gbdpt_film_import_device(gdpt_gbdpt_film *f, const double *blockDevice, const double *lightDevice)
{
    if (!f || !blockDevice || !lightDevice) return bfail(GDPT_ERR_INVALID, "gbdpt_film_import_device: null argument");
    BHIPCHK(hipSetDevice(f->scene->device));
    const size_t npix = (size_t)f->W * f->H;
    BHIPCHK(hipMemcpyAsync(f->block, blockDevice, sizeof(Float) * 5 * npix * 4, hipMemcpyDeviceToDevice, f->stream));
    BHIPCHK(hipMemcpyAsync(f->light, lightDevice, sizeof(Float) * 5 * npix * 3, hipMemcpyDeviceToDevice, f->stream));
    BHIPCHK(hipStreamSynchronize(f->stream));
    return GDPT_OK;
}

void *gdpt_gbdpt_film_stream(gdpt_gbdpt_film *f) { return f ? (void *)f->stream : nullptr; }

int gdpt_gbdpt_evaluate_sample2(gdpt_scene *s, const gdpt_gbdpt_config *cfg, int px, int py, int sample, double out17[17], int maxLight, double *light6, int *nLight,
                                unsigned long long counters[4])
{
    if (!s || !cfg || !out17 || !nLight || !counters || maxLight < 0 || (maxLight > 0 && !light6)) return bfail(GDPT_ERR_INVALID, "gbdpt_evaluate_sample: bad argument");
    if (px < 0 || py < 0 || px >= s->d.cam.width || py >= s->d.cam.height) return bfail(GDPT_ERR_INVALID, "gbdpt_evaluate_sample: pixel outside the film");
    if (int rc = check_scope(s, cfg)) return rc;
    BHIPCHK(hipSetDevice(s->device));
    const BdCam cam = make_cam(s);
    const BdConfig c = make_cfg(cfg, s->bsphereRadius, s->hittableEmitters);
    struct Bufs {                                  // (freed on every return path)
        double *d = nullptr, *dl = nullptr; int *dn = nullptr; unsigned long long *dc = nullptr; GWork *work = nullptr;
        ~Bufs() { hipFree(d); hipFree(dl); hipFree(dn); hipFree(dc); hipFree(work); }
    } b;
    const int ml = std::max(maxLight, 1);
    BHIPCHK(hipMalloc((void **)&b.d, sizeof(double) * 17));
    BHIPCHK(hipMalloc((void **)&b.dl, sizeof(double) * 6 * ml));
    BHIPCHK(hipMalloc((void **)&b.dn, sizeof(int)));
    BHIPCHK(hipMalloc((void **)&b.dc, sizeof(unsigned long long) * 4));
    BHIPCHK(hipMalloc((void **)&b.work, sizeof(GWork)));
    BHIPCHK(hipMemset(b.dc, 0, sizeof(unsigned long long) * 4));
#ifdef GDPT_BD_CHECK_PRIM
    { int z[4] = {0, 0, 0, 0}; BHIPCHK(hipMemset(b.work, 0x7f, sizeof(GWork))); BHIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_bdPrimTrap), z, sizeof z)); }
#endif
    hipLaunchKernelGGL(k_gbdpt_sample, dim3(1), dim3(TBLK), 0, 0, s->d, cam, c, px, py, sample, b.d, maxLight, b.dl, b.dn, b.dc, b.work);
    BHIPCHK(hipGetLastError());
    BHIPCHK(hipMemcpy(out17, b.d, sizeof(double) * 17, hipMemcpyDeviceToHost));
    BHIPCHK(hipMemcpy(nLight, b.dn, sizeof(int), hipMemcpyDeviceToHost));
    if (maxLight > 0) BHIPCHK(hipMemcpy(light6, b.dl, sizeof(double) * 6 * std::min(maxLight, std::max(*nLight, 0)), hipMemcpyDeviceToHost));
    BHIPCHK(hipMemcpy(counters, b.dc, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost));
#ifdef GDPT_BD_CHECK_PRIM
    { int z[4]; BHIPCHK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_bdPrimTrap), sizeof z)); if (z[2]) fprintf(stderr, "G-BDPT prim trap: pixel (%d, %d) sample %d: %d loads outside the tables, first at site %d (len*1000 + line) with prim %d\n", px, py, sample, z[2], z[0], z[1]); }
#endif
    return GDPT_OK;
}

int gdpt_gbdpt_evaluate_sample(gdpt_scene *s, const gdpt_gbdpt_config *cfg, int px, int py, int sample, double out17[17], int maxLight, double *light6, int *nLight,
                               unsigned long long counters[2])
{
    unsigned long long c4[4] = {0, 0, 0, 0};
    if (!counters) return bfail(GDPT_ERR_INVALID, "gbdpt_evaluate_sample: bad argument");
    const int rc = gdpt_gbdpt_evaluate_sample2(s, cfg, px, py, sample, out17, maxLight, light6, nLight, c4);
    counters[0] = c4[0]; counters[1] = c4[1];
    return rc;
}

#ifdef GDPT_BD_PROFILE
// development (-DGDPT_BD_PROFILE, tools/gpu_gbdpt_profile.py): lane clocks of k_bdg_offset by section -- generateOffsetPath, of which the manifold walk,
// half-Jacobians, calcSpecularPDFChange, prefix products, all of prepareOffset
__attribute__((visibility("default"))) int gdpt_gbdpt_film_profile(gdpt_gbdpt_film *f, unsigned long long out[6])
{
    if (int rc = gdpt_gbdpt_film_sync(f)) return rc;
    BHIPCHK(hipMemcpy(out, f->stats + 8, sizeof(unsigned long long) * 6, hipMemcpyDeviceToHost));
    return GDPT_OK;
}
#endif

int gdpt_gbdpt_film_chain_stats(gdpt_gbdpt_film *f, unsigned long long stats[2])
{
    if (!f || !stats) return bfail(GDPT_ERR_INVALID, "gbdpt_film_chain_stats: null argument");
    if (int rc = gdpt_gbdpt_film_sync(f)) return rc;
    BHIPCHK(hipMemcpy(stats, f->stats + 4, sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost));
    return GDPT_OK;
}

} // extern "C"
