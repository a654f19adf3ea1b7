// gpt_wave_capi.hip -- kernels and launcher of the wavefront pipeline (gpt_wavefront.hip.h has the design notes).
#define GDPT_RENDER_DEVICE_FUNCTIONS_ONLY
#include "gpt_render.hip.h"
#include "gpt_wavefront.hip.h"
#include "gpt_scene.hip.h"

#include <algorithm>
#include <cstdlib>

namespace gdpt_tr {

constexpr int WF_MAX_ITERS = 64;        // wavefront iterations per chunk the counters are laid out for
constexpr int WF_SITES = N_SITES;       // ray sites of a bounce (gpt_render.hip.h: SITE_*)

struct WfD {
    // ray queues, one per kind (0: shadow ray = any hit, 1: closest hit), rewritten by every iteration: [7][rayCap] doubles (origin,
    // direction, maxt; mint follows from the origin, skdtree.cpp:126-129,214-217) and the slot + site each ray's result goes to
    Float *ray[2];
    unsigned *rayId[2];         // (site << 28) | slot
    unsigned rayCap[2];
    // results, [site][cap] / [3 (site - 5) + {t, u, v}][cap]: shadow sites 1 = occluded, 0 = free; closest sites the leaf-order triangle or -1
    int *resI;
    Float *resH;
    unsigned cap;               // stride of the results = slots of the launch's chunk (FilmD::qCapacity)
    // the samples still on their way: iteration `it` reads list[it & 1] and appends the survivors to the other one
    unsigned *list[2];
    // counters of a chunk, zeroed once per chunk: [0 .. 2 W + 1]: entries of the list iteration `it` reads at [2 it] (and, for the kernel that
    // takes the last list over -- k_continue -- its cursor at [2 it + 1]); behind them per iteration the ray counts of the two queues and the
    // tracing waves' cursors into them
    unsigned *counters;
    unsigned *err;              // [WF_COUNTERS] of the same allocation, zeroed when the queues are (re)allocated, never per chunk: a ray that found its queue full (wf_failed)
};
__host__ __device__ __forceinline__ constexpr unsigned wf_list_count(int it) { return 2u * (unsigned)it; }
__host__ __device__ __forceinline__ constexpr unsigned wf_ray_count(int it, int kind) { return 2u * (WF_MAX_ITERS + 1) + 4u * (unsigned)it + (unsigned)kind; }
__host__ __device__ __forceinline__ constexpr unsigned wf_ray_cursor(int it, int kind) { return wf_ray_count(it, kind) + 2u; }      // next ray the tracing waves take
constexpr size_t WF_COUNTERS = 2 * (WF_MAX_ITERS + 1) + 4 * WF_MAX_ITERS;

struct WfQueues { WfD d = {}; size_t slots = 0; };

WfQueues *wf_create() { return new WfQueues(); }
void wf_release(WfQueues *q)
{
    WfD &w = q->d;
    for (int k = 0; k < 2; k++) { if (w.ray[k]) hipFree(w.ray[k]); if (w.rayId[k]) hipFree(w.rayId[k]); if (w.list[k]) hipFree(w.list[k]); }
    if (w.resI) hipFree(w.resI);
    if (w.resH) hipFree(w.resH);
    if (w.counters) hipFree(w.counters);
    w = WfD{};
    q->slots = 0;
}
void wf_destroy(WfQueues *q) { if (q) { wf_release(q); delete q; } }
// (continuation phase: one shadow ray and one extension ray per slot and iteration)
size_t wf_bytes_per_slot() { return 2 * (7 * sizeof(Float) + sizeof(unsigned)) + WF_SITES * sizeof(int) + 15 * sizeof(Float) + 2 * sizeof(unsigned); }
size_t wf_slots(const WfQueues *q) { return q->slots; }
int wf_max_iters() { return WF_MAX_ITERS - 1; }
bool wf_failed(const WfQueues *q)                      // (the caller has synchronised the stream)
{
    unsigned e = 0;
    return q && q->d.err && (hipMemcpy(&e, q->d.err, sizeof e, hipMemcpyDeviceToHost) != hipSuccess || e != 0);
}
bool wf_reserve(WfQueues *q, size_t cap)
{
    wf_release(q);
    WfD &w = q->d;
    bool ok = true;
    for (int k = 0; k < 2 && ok; k++) {
        w.rayCap[k] = (unsigned)cap;
        ok = hipMalloc((void **)&w.ray[k], cap * 7 * sizeof(Float)) == hipSuccess && hipMalloc((void **)&w.rayId[k], cap * sizeof(unsigned)) == hipSuccess &&
             hipMalloc((void **)&w.list[k], cap * sizeof(unsigned)) == hipSuccess;
    }
    ok = ok && hipMalloc((void **)&w.resI, cap * WF_SITES * sizeof(int)) == hipSuccess && hipMalloc((void **)&w.resH, cap * 15 * sizeof(Float)) == hipSuccess &&
         hipMalloc((void **)&w.counters, (WF_COUNTERS + 1) * sizeof(unsigned)) == hipSuccess;
    if (ok) { w.err = w.counters + WF_COUNTERS; ok = hipMemset(w.err, 0, sizeof(unsigned)) == hipSuccess; }
    if (!ok) { (void)hipGetLastError(); wf_release(q); return false; }
    q->slots = cap;
    return true;
}
int wf_begin_chunk(WfQueues *q, hipStream_t stream, int iters, FilmD &fdRender, FilmD &fdContinue)
{
    if (iters < 1 || iters > wf_max_iters() || fdRender.qCapacity > q->slots) return -1;
    if (hipMemsetAsync(q->d.counters, 0, WF_COUNTERS * sizeof(unsigned), stream) != hipSuccess) return -1;
    fdRender.qList = q->d.list[0]; fdRender.qCount = q->d.counters + wf_list_count(0);
    fdContinue.qList = q->d.list[(iters + 1) & 1]; fdContinue.qCount = q->d.counters + wf_list_count(iters + 1);
    return 0;
}

// ---- the tracers of the replayed bounce (MODE 1 / 2: write the level's rays; MODE 2 / 3: read what was traced) --------------------------
struct NullAcc {            // the sums of a pass that only looks for its rays: every contribution is dead code
    __device__ __forceinline__ void zero() {}
    __device__ __forceinline__ void add3(int, d3) {}
    __device__ __forceinline__ d3 get3(int) const { return mk(0.0); }
    __device__ __forceinline__ Float get(int) const { return 0.0; }
    __device__ __forceinline__ void set(int, Float) {}
};

// The sums of a replay, left where they are -- in the sample's record: a contribution is a read-modify-write of its row.  Holding the 30 sums
// in registers for the length of a bounce (60 VGPRs, on top of the path's 62) is what made the first version of the shading kernel spill 650
// registers per lane into 1.6 KB of scratch, and with 400 MB of scratch resident that traffic went to HBM: 12 ms per pass where the records
// themselves are 2 ms.  Only the throughput sum, which every offset of both halves of a bounce adds to (eight times), stays in registers.
struct RecordAcc {
    Float *q;                   // row 32 of the record, at the slot
    size_t st;
    d3 T;
    __device__ __forceinline__ void open(const FilmD &F, unsigned slot)
    {
        q = F.qRec + (size_t)32 * F.qCapacity + slot; st = F.qCapacity;
        T = mk(q[(ACC_T + 0) * st], q[(ACC_T + 1) * st], q[(ACC_T + 2) * st]);
    }
    __device__ __forceinline__ void close() { q[(ACC_T + 0) * st] = T.x; q[(ACC_T + 1) * st] = T.y; q[(ACC_T + 2) * st] = T.z; }
    __device__ __forceinline__ void add3(int k, d3 v)
    {
        if (k == ACC_T) { T.x += v.x; T.y += v.y; T.z += v.z; return; }
        q[k * st] += v.x; q[(k + 1) * st] += v.y; q[(k + 2) * st] += v.z;
    }
};

// REC: the offsets' states are not in the Lane but in the sample's continuation record (every offset connected or dead: throughput and
// pdf, rows 15 + 4 i, and the alive bits): fetched where a bounce uses them, written back where it changes them
template <int MODE_, bool REC = false>
struct WfTracer {
    static constexpr int MODE = MODE_;
    const WfD &Q;
    unsigned slot;
    unsigned *rayCount;         // this iteration's two ray counters
    int n;                      // rays written by this pass
    Float *rec = nullptr;       // REC: the record at the slot, its stride, its alive bits
    size_t recSt = 0;
    unsigned aliveBits = 0;
    template <bool UNROLL, bool MODIFIES, class BODY>
    __device__ __forceinline__ void each_offset(Lane &L, BODY &&body)
    {
        if constexpr (!REC) for_offsets<UNROLL>(L.off, body);
        else if constexpr (MODE == 3) {
            auto one = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                Offset s;
                s.alive = (aliveBits >> i) & 1; s.status = RAY_CONNECTED;
                if (s.alive) { s.throughput = mk(rec[(15 + 4 * i) * recSt], rec[(16 + 4 * i) * recSt], rec[(17 + 4 * i) * recSt]); s.pdf = rec[(18 + 4 * i) * recSt]; }
                else { s.throughput = mk(0.0); s.pdf = 0.0; }
                body(ic, s);
                if (MODIFIES && s.alive) { rec[(15 + 4 * i) * recSt] = s.throughput.x; rec[(16 + 4 * i) * recSt] = s.throughput.y; rec[(17 + 4 * i) * recSt] = s.throughput.z; rec[(18 + 4 * i) * recSt] = s.pdf; }
            };
            one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
        }
        // (MODE 1 of a connected sample: no offset has a ray of its own, and nothing else of the pass is kept)
    }
    __device__ __forceinline__ void scale_offset_pdfs(Lane &L, Float q)
    {
        if constexpr (!REC) {
#pragma unroll
            for (int i = 0; i < 4; i++) L.off[i].pdf *= q;
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; i++) if ((aliveBits >> i) & 1) rec[(18 + 4 * i) * recSt] *= q;
        }
    }
    __device__ __forceinline__ static constexpr int level(int site) { return site < SITE_OFF ? 1 : 2; }
    // one atomic per wave and site: the lanes that reach the site together take consecutive queue entries
    __device__ __forceinline__ void push(int kind, int site, d3 o, d3 d, Float maxt)
    {
        const unsigned long long mask = __ballot(true);
        const int lane = threadIdx.x & 63, leader = __ffsll((unsigned long long)mask) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&rayCount[kind], (unsigned)__popcll(mask));
        base = __shfl(base, leader);
        const unsigned at = base + (unsigned)__popcll(mask & ((1ULL << lane) - 1ULL));
        if (at >= Q.rayCap[kind]) atomicOr(Q.err, 1u);   // (the queues are sized for every site of every slot: should that ever not hold, the replay would read a
                                                        //  stale result -- the render is failed instead, wf_failed / gdpt_film_sync)
        if (at < Q.rayCap[kind]) {
            Float *r = Q.ray[kind] + at;
            const size_t st = Q.rayCap[kind];
            qst(&r[0], o.x); qst(&r[st], o.y); qst(&r[2 * st], o.z);
            qst(&r[3 * st], d.x); qst(&r[4 * st], d.y); qst(&r[5 * st], d.z);
            qst(&r[6 * st], maxt);
            __builtin_nontemporal_store(((unsigned)site << 28) | slot, &Q.rayId[kind][at]);
        }
        n++;
    }
    __device__ __forceinline__ bool occluded(int site, Lane &L, d3 o, d3 d, Float maxt)
    {
        if (MODE == 3 || level(site) < MODE) {
            if (MODE == 3) L.nShadow++;
            return __builtin_nontemporal_load(&Q.resI[(size_t)site * Q.cap + slot]) != 0;
        }
        if (level(site) == MODE) push(0, site, o, d, maxt);
        return false;
    }
    __device__ __forceinline__ void closest(int site, Lane &L, d3 o, d3 d, Hit &h)
    {
        if (MODE == 3 || level(site) < MODE) {
            if (MODE == 3) L.nClosest++;
            const size_t cs = (size_t)3 * (site - SITE_EXT) * Q.cap + slot;
            h.t = qld(&Q.resH[cs]); h.u = qld(&Q.resH[cs + Q.cap]); h.v = qld(&Q.resH[cs + 2 * (size_t)Q.cap]);
            h.prim = __builtin_nontemporal_load(&Q.resI[(size_t)site * Q.cap + slot]);
            return;
        }
        if (level(site) == MODE) push(1, site, o, d, GD_INF);
        h.prim = -1; h.t = GD_INF; h.u = h.v = 0.0;
    }
};

// ---- traversal only ------------------------------------------------------------------------------------------------------------------
// Persistent waves over a ray queue: one lane = one ray at a time; a lane whose ray is finished writes its result and, as soon as
// `refillMin` lanes of its wave are idle, they take the next rays of the queue together (one atomic per refill).  The traversal itself is
// trace() of gpt_kernels.hip.h (while-while, near child first, fp32 slab test, TriAccel in fp64), restated so that a lane can leave and
// enter it between two steps: the same nodes are visited and the same triangles tested in the same order per ray, so hits are identical.
// Why refill: with one batch of 64 rays per wave pass the kernel is ISSUE-bound at 23 % lane utilisation on incoherent rays (PMC, atrium:
// 5 waves per SIMD, each with an instruction in flight 19 % of its cycles) -- every instruction is issued for the slowest ray of a batch.
constexpr uint32_t WF_IDLE = 0xffffffffu, WF_FIN = 0xfffffffeu;       // (both carry the leaf bit: the inner-node loop passes them by)
template <bool LDS_SCENE, bool ANY>
__global__ __launch_bounds__(TBLK) void k_wf_trace(SceneD S, WfD Q, int it, int stackDepth, int refillMin, int leafMin)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    SceneView sv;
    int *stack;
    unsigned char *s_acc;
    block_setup<LDS_SCENE, false>(S, stackDepth, s_dyn, sv, stack, s_acc);
    constexpr int kind = ANY ? 0 : 1;
    const unsigned total = min(__hip_atomic_load(&Q.counters[wf_ray_count(it, kind)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), Q.rayCap[kind]);
    unsigned *cursor = &Q.counters[wf_ray_cursor(it, kind)];
    const size_t st = Q.rayCap[kind];
    const int lane = threadIdx.x & 63;
    d3 o = mk(0.0), d = mk(0.0);
    Float mint = 0.0, maxt = 0.0;
    RayF R = ray_f(o, d, 0.0, 0.0, sv.boundM);
    Hit hit;
    hit.prim = -1; hit.t = GD_INF; hit.u = hit.v = 0.0;
    unsigned id = 0;
    int sp = 0;
    uint32_t ref = WF_IDLE;
    bool exhausted = false;
    while (true) {
        const unsigned long long idle = __ballot(ref == WF_IDLE);
        if (!exhausted && ((int)__popcll(idle) >= refillMin || idle == ~0ULL)) {
            const int leader = __ffsll((unsigned long long)idle) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(cursor, (unsigned)__popcll(idle));
            base = __shfl(base, leader);
            if (base + (unsigned)__popcll(idle) >= total) exhausted = true;             // (uniform: the queue has been handed out)
            if (ref == WF_IDLE) {
                const unsigned i = base + (unsigned)__popcll(idle & ((1ULL << lane) - 1ULL));
                if (i < total) {
                    const Float *r = Q.ray[kind] + i;
                    o = mk(qld(&r[0]), qld(&r[st]), qld(&r[2 * st])); d = mk(qld(&r[3 * st]), qld(&r[4 * st]), qld(&r[5 * st]));
                    maxt = qld(&r[6 * st]);
                    id = __builtin_nontemporal_load(&Q.rayId[kind][i]);
                    mint = ANY ? ray_mint_shadow(o, GD_EPSILON) : ray_mint_closest(o, GD_EPSILON);
                    hit.prim = -1; hit.t = GD_INF; hit.u = hit.v = 0.0;
                    sp = 0;
                    if (!(maxt > mint)) ref = WF_FIN;                                    // trace(): an empty interval hits nothing
                    else { R = ray_f(o, d, mint, maxt, sv.boundM); ref = sv.rootRef; }
                }
            }
        }
        if (__ballot(ref != WF_IDLE) == 0) { if (exhausted) break; continue; }
        // inner nodes: the lanes walk down until `leafMin` of them hold a leaf (or none has an inner node left).  trace() lets every lane
        // walk to ITS next leaf before the wave turns to the triangles; with incoherent rays that is ~7 steps on average and ~35 for the slowest
        // of 64 lanes, i.e. ~20 % of the lanes busy (PMC: 23 %).  Leaving the loop early only changes when a lane's leaf is tested, not what
        // the lane visits or in which order.
        while (true) {
            const bool inner = !(ref & BVH_LEAF);
            if (__ballot(inner) == 0) break;
            if ((int)__popcll(__ballot((ref & BVH_LEAF) != 0 && ref < WF_FIN)) >= leafMin) break;
            if (inner) {
                ref = node_step(sv, ref, R, stack, sp);
                if (ref == BVH_NONE) ref = WF_FIN;
            }
        }
        // leaves: the lanes that hold one test its triangles together
        if ((ref & BVH_LEAF) != 0 && ref < WF_FIN) {
            const uint32_t first = (ref & ~BVH_LEAF) >> 3, cnt = (ref & 7u) + 1u;
            bool found = false;
            for (uint32_t i = 0; i < cnt; i++) {
                Float u, v, t;
                const TriIsect ta = sv.isect[first + i];
                if (tri_test(ta, o, d, mint, maxt, u, v, t)) {
                    hit.t = t; hit.u = u; hit.v = v; hit.prim = (int)(first + i);
                    if (ANY) { found = true; break; }
                    maxt = t;
                    R.maxt = up_f(t);
                }
            }
            if (found || sp == 0) ref = WF_FIN;
            else { sp--; ref = (uint32_t)stack[sp * TBLK]; }
        }
        // finished rays: the result goes to the slot and site the ray came from
        if (ref == WF_FIN) {
            const unsigned slot = id & 0x0fffffffu, site = id >> 28;
            if (ANY) __builtin_nontemporal_store(hit.prim >= 0 ? 1 : 0, &Q.resI[(size_t)site * Q.cap + slot]);
            else {
                const size_t cs = (size_t)3 * (site - SITE_EXT) * Q.cap + slot;
                qst(&Q.resH[cs], hit.t); qst(&Q.resH[cs + Q.cap], hit.u); qst(&Q.resH[cs + 2 * (size_t)Q.cap], hit.v);
                __builtin_nontemporal_store(hit.prim, &Q.resI[(size_t)site * Q.cap + slot]);
            }
            ref = WF_IDLE;
        }
    }
}

// ---- continuation phase: every offset path is connected or dead (the state is the continuation record of gpt_render.hip.h) ------------
// flags: 1 = the rays of the samples' current bounce have been traced: replay it with their results first; 2 = then run the next bounce up
// to its rays.  Without 1 (the first iteration) the records are left as they are; without 2 (the last) the survivors' records are stored
// for the kernel that takes the list over.
template <bool LDS_SCENE, int WAVES_PER_SIMD, bool ENV, bool SMOOTH>
__global__ __launch_bounds__(TBLK, WAVES_PER_SIMD) void k_wf_cont(SceneD S, ConfigD cfg, FilmD F, WfD Q, int it, int flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    SceneView sv;
    int *stack;
    unsigned char *s_acc;
    block_setup<LDS_SCENE, false>(S, 0, s_dyn, sv, stack, s_acc);
    const int lane = threadIdx.x & 63;
    const unsigned total = __hip_atomic_load(&Q.counters[wf_list_count(it)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned *listIn = Q.list[it & 1];
    unsigned *listOut = Q.list[(it + 1) & 1];
    unsigned *outCount = &Q.counters[wf_list_count(it + 1)];
    unsigned *rayCount = &Q.counters[wf_ray_count(it, 0)];
    unsigned nClosest = 0, nShadow = 0, paths = 0, pathLen = 0;
    for (unsigned e = blockIdx.x * TBLK + threadIdx.x; e < total; e += gridDim.x * TBLK) {
        const unsigned slot = listIn[e];
        Lane L;
        L.nClosest = L.nShadow = 0;
        q_load_main(F, slot, L);
        Float *rec = F.qRec + slot;
        if (flags & 1) {
            const unsigned aliveBits = (unsigned)__double_as_longlong(rec[(size_t)31 * F.qCapacity]);
            RecordAcc A;
            A.open(F, slot);
            WfTracer<3, true> tr = {Q, slot, rayCount, 0, rec, F.qCapacity, aliveBits};
            const bool alive = bounce<ENV, SMOOTH, true, true, (WAVES_PER_SIMD > 2)>(S, sv, cfg, tr, L, A);
            A.close();
            nClosest += L.nClosest; nShadow += L.nShadow;
            if (!alive) {                       // (the record's sums are final: it only gets its mark)
                paths++; pathLen += (unsigned)L.depth;
                qst(&rec[(size_t)13 * F.qCapacity], __longlong_as_double((long long)Q_DONE));
                continue;
            }
            q_store_main(F, slot, L);           // (before the next bounce is looked at: the record is dead from here on)
        }
        if (flags & 2) {
            // the next bounce as far as its rays: nothing of this pass is kept but the rays (the replay after the trace starts from the record again)
            const unsigned depth = (unsigned)L.depth;
            NullAcc N;
            WfTracer<1, true> tr = {Q, slot, rayCount, 0};
            bounce<ENV, SMOOTH, true, true, (WAVES_PER_SIMD > 2)>(S, sv, cfg, tr, L, N);
            // a bounce without any ray ends the path before anything is added to the sums (maxDepth / strictNormals at its top, or a
            // delta BSDF whose sample fails): it needs no replay
            if (tr.n == 0) {
                paths++; pathLen += depth;
                qst(&rec[(size_t)13 * F.qCapacity], __longlong_as_double((long long)Q_DONE));
                continue;
            }
        }
        const unsigned long long mask = __ballot(true);
        const int leader = __ffsll((unsigned long long)mask) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(outCount, (unsigned)__popcll(mask));
        base = __shfl(base, leader);
        listOut[base + (unsigned)__popcll(mask & ((1ULL << lane) - 1ULL))] = slot;
    }
    const unsigned c0 = __builtin_amdgcn_wave_reduce_add_u32(nClosest, 0), c1 = __builtin_amdgcn_wave_reduce_add_u32(nShadow, 0);
    const unsigned c2 = __builtin_amdgcn_wave_reduce_add_u32(paths, 0), c3 = __builtin_amdgcn_wave_reduce_add_u32(pathLen, 0);
    if (lane == 0 && (c0 | c1 | c2 | c3)) {
        atomicAdd(&F.stats[0], (unsigned long long)c0);
        atomicAdd(&F.stats[1], (unsigned long long)c1);
        atomicAdd(&F.stats[2], (unsigned long long)c2);
        atomicAdd(&F.stats[3], (unsigned long long)c3);
    }
}

int wf_continue(const gdpt_scene *s, hipStream_t stream, const ConfigD &cfg, const FilmD &fd, WfQueues *queues, int iters, int stackDepth, int sceneBytes)
{
    WfD q = queues->d;
    q.cap = fd.qCapacity;               // (the stride of this launch's slots; the allocation may be larger)
    const dim3 block(TBLK);
    const size_t tlds = (size_t)stackDepth * TBLK * sizeof(int) + sceneBytes;
    // resident blocks per CU: the trace kernel by its LDS (stack + staged scene), at most 8 (one wave per SIMD each); the shading kernel by its registers
    const int traceBlocks = (int)std::max<size_t>(1, std::min<size_t>(8, ((size_t)160 * 1024) / std::max<size_t>(tlds, 1)));
    static const int traceOver = getenv("GDPT_WF_TRACE_OVER") ? std::max(1, atoi(getenv("GDPT_WF_TRACE_OVER"))) : 1;
    static const int refill = getenv("GDPT_WF_REFILL") ? std::max(1, std::min(64, atoi(getenv("GDPT_WF_REFILL")))) : 48;   // idle lanes of a tracing wave before they take new rays
    static const int leafMin = getenv("GDPT_WF_LEAFMIN") ? std::max(1, std::min(64, atoi(getenv("GDPT_WF_LEAFMIN")))) : 24;   // lanes holding a leaf before the wave tests triangles
    static const int shadeWaves = getenv("GDPT_WF_SHADE_WAVES") ? atoi(getenv("GDPT_WF_SHADE_WAVES")) : 2;     // build of the shading kernel (experiments)
    static const int shadeOver = getenv("GDPT_WF_SHADE_OVER") ? std::max(1, atoi(getenv("GDPT_WF_SHADE_OVER"))) : 2;
    const dim3 tgrid((unsigned)(s->numCUs * traceBlocks * traceOver)), sgrid((unsigned)(s->numCUs * (shadeWaves <= 2 ? 2 : 4) * shadeOver));
    const bool lds = s->d.ldsScene != 0;
#define WF_CONT(LDSV, ENVV, SMV) do { \
        if (shadeWaves <= 2) hipLaunchKernelGGL((k_wf_cont<LDSV, 2, ENVV, SMV>), sgrid, block, (size_t)sceneBytes, stream, s->d, cfg, fd, q, it, flags); \
        else hipLaunchKernelGGL((k_wf_cont<LDSV, 4, ENVV, SMV>), sgrid, block, (size_t)sceneBytes, stream, s->d, cfg, fd, q, it, flags); } while (0)
#define WF_CONT_F(LDSV) do { \
        if (s->perVertex) WF_CONT(LDSV, true, true); \
        else if (s->specialEmitters) WF_CONT(LDSV, true, false); \
        else WF_CONT(LDSV, false, false); } while (0)
    for (int it = 0; it <= iters; it++) {
        const int flags = (it > 0 ? 1 : 0) | (it < iters ? 2 : 0);
        if (lds) WF_CONT_F(true); else WF_CONT_F(false);
        if (it == iters) break;
        if (lds) {
            hipLaunchKernelGGL((k_wf_trace<true, true>), tgrid, block, tlds, stream, s->d, q, it, stackDepth, refill, leafMin);
            hipLaunchKernelGGL((k_wf_trace<true, false>), tgrid, block, tlds, stream, s->d, q, it, stackDepth, refill, leafMin);
        } else {
            hipLaunchKernelGGL((k_wf_trace<false, true>), tgrid, block, tlds, stream, s->d, q, it, stackDepth, refill, leafMin);
            hipLaunchKernelGGL((k_wf_trace<false, false>), tgrid, block, tlds, stream, s->d, q, it, stackDepth, refill, leafMin);
        }
    }
#undef WF_CONT_F
#undef WF_CONT
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace gdpt_tr
