// poisson_persistent.hip.h -- the whole CG loop of one IRLS iteration as ONE persistent cooperative kernel.
//
// Why: at the image sizes G-PT renders (1280x720: x, r, p, Ap, w = 55 MB) the CG iterate fits the chip's register file
// (256 CUs x 512 KB), so the only thing that has to leave a CU per iteration is a one-pixel ring of r and two 3-float dot
// products.  Each workgroup owns a 64 x TH tile for the whole solve, keeps x, r, p (and w) of its 4 px per lane in VGPRs,
// stages p (+ring) through LDS for the 5-point stencil, and meets the other workgroups at TWO all-gathers per iteration
// (p.Ap, then r.r) -- the two global reductions the CG recurrence of the reference has (Solver.cpp:466-469); nothing
// else is synchronised and no array is streamed through HBM.  The arithmetic per element is that of kf_Ax / kf_r_rz / kf_x_p
// (reference association order, -ffp-contract=off); only the summation tree of the dot products differs (per-tile partials,
// summed by every workgroup in the same fixed order so that all of them derive bit-identical step sizes).
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, form "8-byte agent-scope atomics on both sides"): a tile
// partial travels as three 8-byte words {float bits, iteration tag}, written with relaxed AGENT-scope atomic stores
// (write-through) into the tile's slot of a G x 3 table; every workgroup polls the whole table (one word per lane, relaxed
// agent-scope atomic loads, s_sleep between polls) until each word carries the tag of the current iteration.  The gather is
// data and barrier in one: no arrival counter, no second round trip.  The boundary ring of r travels the same way -- every
// float of it is an 8-byte {float bits, iteration tag} word the neighbouring tile's ring lane polls -- so nothing has to be
// drained or fenced: each word validates itself.  Slot reuse is safe because a workgroup only writes gather A of iteration i+1
// after it has seen every gather-B word of iteration i, which each workgroup publishes after it consumed gather A of
// iteration i; the ring of iteration i+1 is written after gather A of i+1 completed, which every neighbour publishes after it
// consumed the ring of iteration i.  Correctness does not depend on where a workgroup runs.  Tags carry the launch number in their upper half, so the tables need no clearing between launches; every spin is bounded, and a timeout raises a
// sticky flag that makes every workgroup leave and that the host reports (and recovers from, poisson_capi.hip) instead of
// hanging.  Residency comes from the grid size (tiles <= CUs, one workgroup per CU), checked by hipLaunchCooperativeKernel.
#pragma once
#include "poisson_kernels.hip.h"

namespace gdpt {

constexpr int PT_W = 64;                 // tile width in pixels: 16 lanes x 4 px
constexpr int PT_MAXH = 64;              // tile rows <= 64 (16 lanes per row -> <= 1024 threads)
// LDS image of p: three colour planes of (TH+2) x 64 floats (row 0 / TH+1: the ring rows) and the two ring columns.  A lane's
// 4 px of one colour are one aligned 16-byte access, and 8 consecutive lanes cover the 32 banks: conflict-free.  (The AoS
// layout of the first version put 64 lanes on 8 banks; the stencil phase took 5.5 us of a 15 us iteration.)
constexpr int PT_PLANE = (PT_MAXH + 2) * PT_W;
constexpr int PT_COLL = 3 * PT_PLANE, PT_COLR = PT_COLL + 3 * PT_MAXH, PT_LDS = PT_COLR + 3 * PT_MAXH;
__device__ __forceinline__ int pt_plane(int c, int row, int x) { return c * PT_PLANE + row * PT_W + x; }

// value of `v` in the lane one to the left / right inside the 16-lane row (0 at the row's end)
__device__ __forceinline__ float pt_from_left(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true)); }
__device__ __forceinline__ float pt_from_right(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true)); }
constexpr int PT_HALO = (2 * PT_W + 2 * PT_MAXH) * 3;   // floats a tile publishes per iteration: top, bottom, left, right
constexpr unsigned PT_SPIN_LIMIT = 4000000u;

struct PersistArgs {
    float *x, *r, *p;                    // images (AoS RGB); p is written back at the end
    const float *w2;                     // 3n weights (ignored when UNITW)
    unsigned long long *gat;             // [2][PT_MAXG*3] tagged partials: gather A (p.Ap) and gather B (r.r)
    unsigned long long *halo;            // [tiles][PT_HALO] boundary r of every tile, each float tagged with its iteration
    unsigned *bar;                       // [1] sticky error flag
    float *s_rz;                         // in: r.r of the prologue; out: final r.r
    int W, H, tilesX, tilesY, TH, iters;
    unsigned tagBase;                    // launch number << 16: tags of earlier launches never match
    float alpha;
    int debugFail;                       // test hook (GDPT_DEBUG_PERSISTENT_FAIL=1): raise the time-out flag at once, to exercise the recovery
};

#define PT_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int PT_MAXG = 256;             // workgroups (tiles) at most
constexpr int PT_BAR_WORDS = 64;         // [1] sticky error flag; [16..] phase clocks of the GDPT_PT_TIMING build

// Workgroup barrier for LDS traffic only.  __syncthreads() carries a workgroup-scope fence, i.e. s_waitcnt vmcnt(0): every
// barrier would wait for the write-through stores of the boundary record and the partials to reach memory (~2-3 us with 240
// workgroups storing) although no lane of this workgroup ever reads them.  LDS operations are all this kernel orders here.
__device__ __forceinline__ void pt_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void pt_store(unsigned long long *p, float v, unsigned tag)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | __float_as_uint(v), PT_RLX_AGENT);
}

// Poll a tagged word until it carries `tag` (w: a value already loaded from it); bounded, raises the sticky flag on timeout.
__device__ __forceinline__ float pt_wait(unsigned long long *p, unsigned long long w, unsigned tag, int *s_fail, unsigned *err)
{
    unsigned spins = 0;
    while ((unsigned)(w >> 32) != tag) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if (((spins & 1023u) == 0 && __hip_atomic_load(err, PT_RLX_AGENT) != 0) || spins > PT_SPIN_LIMIT) {
            __hip_atomic_store(err, 1u, PT_RLX_AGENT);
            *s_fail = 1;
            break;
        }
        w = __hip_atomic_load(p, PT_RLX_AGENT);
    }
    return __uint_as_float((unsigned)w);
}

// Wave total in every lane without touching LDS: quad, half-row and row steps as DPP adds (every lane of a 16-lane row ends with
// the row's sum -- the operands only swap sides, so all lanes hold the same bits), then the four row sums through readlane,
// added in a fixed order.  (__shfl_xor compiles to ds_bpermute_b32 here: 6 dependent LDS round trips per reduction, four
// reductions per CG iteration on the critical path.)
__device__ __forceinline__ float pt_wave_sum(float v)
{
#define PT_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
    PT_DPP_ADD(0xB1);        // quad_perm:[1,0,3,2]
    PT_DPP_ADD(0x4E);        // quad_perm:[2,3,0,1]
    PT_DPP_ADD(0x141);       // row_half_mirror
    PT_DPP_ADD(0x140);       // row_mirror
#undef PT_DPP_ADD
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return ((r0 + r1) + r2) + r3;
}

// All-gather + fixed-order sum of one 3-float partial per workgroup.  `mine` (the same in every thread) is published under
// `tag`; on return v holds the totals over the G workgroups, identical in every thread of every workgroup.  gsm: >= 3*G floats.
__device__ __forceinline__ bool pt_allgather_sum(unsigned long long *gat, unsigned tag, int G, int tile, const float (&mine)[3],
                                                 float (&v)[3], float *gsm, int *s_fail, unsigned *err)
{
    const int t = threadIdx.x;
    if (t < 3) pt_store(&gat[tile * 3 + t], mine[t], tag);
    for (int idx = t; idx < 3 * G; idx += blockDim.x)
        gsm[idx] = pt_wait(&gat[idx], __hip_atomic_load(&gat[idx], PT_RLX_AGENT), tag, s_fail, err);
    pt_sync();
    // every wave sums the table itself, in one fixed order: lane l takes tiles l, l+64, ...; then the xor butterfly
    const int ln = t & 63;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int i = ln; i < G; i += 64) { a0 += gsm[3 * i]; a1 += gsm[3 * i + 1]; a2 += gsm[3 * i + 2]; }
    v[0] = pt_wave_sum(a0); v[1] = pt_wave_sum(a1); v[2] = pt_wave_sum(a2);
    return *s_fail == 0;
}

// The same for two partials per workgroup at once (the single-gather CG, fusion level 3): tables gatG and gatD are published together and
// polled together.  gsm: >= 6*G floats.
__device__ __forceinline__ bool pt_allgather_sum6(unsigned long long *gatG, unsigned long long *gatD, unsigned tag, int G, int tile, const float (&g)[3], const float (&d)[3],
                                                  float (&vg)[3], float (&vd)[3], float *gsm, int *s_fail, unsigned *err)
{
    const int t = threadIdx.x;
    if (t < 3) pt_store(&gatG[tile * 3 + t], g[t], tag);
    else if (t < 6) pt_store(&gatD[tile * 3 + (t - 3)], d[t - 3], tag);
    for (int idx = t; idx < 6 * G; idx += blockDim.x) {
        unsigned long long *w = idx < 3 * G ? &gatG[idx] : &gatD[idx - 3 * G];
        gsm[idx] = pt_wait(w, __hip_atomic_load(w, PT_RLX_AGENT), tag, s_fail, err);
    }
    pt_sync();
    const int ln = t & 63;
    float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = ln; i < G; i += 64)
#pragma unroll
        for (int c = 0; c < 3; c++) { a[c] += gsm[3 * i + c]; a[3 + c] += gsm[3 * G + 3 * i + c]; }
#pragma unroll
    for (int c = 0; c < 3; c++) { vg[c] = pt_wave_sum(a[c]); vd[c] = pt_wave_sum(a[3 + c]); }
    return *s_fail == 0;
}

// Sum a[0..2] over a block of up to 16 waves; every thread receives the totals.  sm: >= 64 floats.
__device__ __forceinline__ void pt_block_sum3(float (&a)[3], float *sm, int nwaves)
{
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 3; c++) a[c] = pt_wave_sum(a[c]);
    pt_sync();
    if (ln == 0) { sm[wv * 4 + 0] = a[0]; sm[wv * 4 + 1] = a[1]; sm[wv * 4 + 2] = a[2]; }
    pt_sync();
    float t[3] = {0.0f, 0.0f, 0.0f};
    for (int w = 0; w < nwaves; w++) { t[0] += sm[w * 4]; t[1] += sm[w * 4 + 1]; t[2] += sm[w * 4 + 2]; }
    a[0] = t[0]; a[1] = t[1]; a[2] = t[2];
}

// SINGLE (fusion level 3): the Chronopoulos-Gear form of the same iteration -- w = A r, gamma = r.r and delta = w.r in ONE gather, then
//     beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old), z = w + beta z (= A p), p = r + beta p, x += alpha p, r -= alpha z
// -- algebraically the recurrence of Solver.cpp:466-469 (alpha = r.r / p.Ap, beta = r'.r' / r.r) with p.Ap obtained from delta instead of a second
// reduction; in floating point the iterates differ by rounding (tolerance in tests/test_poisson_gpu.py).  The stencil runs on r, the LDS
// image holds r, and the ring is the neighbours' r as published.  The gather tables alternate by iteration parity: with one gather per iteration
// a fast workgroup would otherwise overwrite a slot a slow one is still polling.
template <bool UNITW, bool SINGLE = false>
__global__ __launch_bounds__(1024) void kp_cg(PersistArgs A)
{
    __shared__ __attribute__((aligned(16))) float lp[PT_LDS];                  // p (SINGLE: r) of the tile with its ring
    __shared__ float sm[64];
    __shared__ float gsm[(SINGLE ? 6 : 3) * PT_MAXG];
    __shared__ float hs[PT_HALO];
    __shared__ int s_fail;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, nwaves = (blockDim.x + 63) >> 6;
    const int tile = blockIdx.x, tX = tile % A.tilesX, tY = tile / A.tilesX;
    const int X0 = tX * PT_W, Y0 = tY * A.TH;
    const int TWv = min(PT_W, A.W - X0), THv = min(A.TH, A.H - Y0);        // valid extent of this tile
    const int x = X0 + 4 * tx, y = Y0 + ty;
    const bool valid = ty < THv && 4 * tx < TWv;
    const int W = A.W, H = A.H, n = W * H;
    const int i = y * W + x;
    const float alphaSqr = A.alpha * A.alpha;
    const int G = A.tilesX * A.tilesY;
    unsigned long long *myHalo = A.halo + (size_t)tile * PT_HALO;

    // ---- load the tile: x, r (p := r, Solver.cpp:405) and the weights, once ----
    float xv[12], rv[12], pv[12];
    float w0[4], w1[4], wv_[4], wu[4], w1l = 0.0f;
#pragma unroll
    for (int k = 0; k < 12; k++) { xv[k] = rv[k] = pv[k] = 0.0f; }
#pragma unroll
    for (int k = 0; k < 4; k++) { w0[k] = w1[k] = wv_[k] = wu[k] = 1.0f; }
    if (valid) {
        const float4 *x4 = reinterpret_cast<const float4 *>(A.x + 3 * (size_t)i), *r4 = reinterpret_cast<const float4 *>(A.r + 3 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float4 a = x4[k], b = r4[k];
            xv[4 * k] = a.x; xv[4 * k + 1] = a.y; xv[4 * k + 2] = a.z; xv[4 * k + 3] = a.w;
            rv[4 * k] = b.x; rv[4 * k + 1] = b.y; rv[4 * k + 2] = b.z; rv[4 * k + 3] = b.w;
        }
#pragma unroll
        for (int k = 0; k < 12; k++) pv[k] = rv[k];
        if (!UNITW) {
            const float4 a = *reinterpret_cast<const float4 *>(A.w2 + i), b = *reinterpret_cast<const float4 *>(A.w2 + n + i), c = *reinterpret_cast<const float4 *>(A.w2 + 2 * n + i);
            w0[0] = a.x; w0[1] = a.y; w0[2] = a.z; w0[3] = a.w;
            w1[0] = b.x; w1[1] = b.y; w1[2] = b.z; w1[3] = b.w;
            wv_[0] = c.x; wv_[1] = c.y; wv_[2] = c.z; wv_[3] = c.w;
            w1l = (x != 0) ? A.w2[n + i - 1] : 0.0f;
            if (y != 0) { const float4 d = *reinterpret_cast<const float4 *>(A.w2 + 2 * n + i - W); wu[0] = d.x; wu[1] = d.y; wu[2] = d.z; wu[3] = d.w; }
            else { wu[0] = wu[1] = wu[2] = wu[3] = 0.0f; }
        } else w1l = 1.0f;
    }
    // ring of p (= r of the neighbouring tiles' boundary, straight from the prologue kernel's output)
    const int ringN = (2 * TWv + 2 * THv) * 3;
    for (int e = t; e < ringN; e += blockDim.x) {
        const int c = e % 3, q = e / 3;
        int gx, gy, slot;
        if (q < TWv) { gx = X0 + q; gy = Y0 - 1; slot = pt_plane(c, 0, q); }
        else if (q < 2 * TWv) { gx = X0 + (q - TWv); gy = Y0 + THv; slot = pt_plane(c, THv + 1, q - TWv); }
        else if (q < 2 * TWv + THv) { gx = X0 - 1; gy = Y0 + (q - 2 * TWv); slot = PT_COLL + c * PT_MAXH + (q - 2 * TWv); }
        else { gx = X0 + TWv; gy = Y0 + (q - 2 * TWv - THv); slot = PT_COLR + c * PT_MAXH + (q - 2 * TWv - THv); }
        lp[slot] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? A.r[3 * ((size_t)gy * W + gx) + c] : 0.0f;
    }
    // where ring float e comes from: neighbour tile, offset in ITS boundary record, my LDS slot, colour
    auto ring_source = [&](int e, int &nt, int &off, int &slot, int &c) -> bool {
        c = e % 3;
        const int q = e / 3;
        bool have;
        if (q < TWv) { have = tY > 0; nt = tile - A.tilesX; off = PT_W * 3 + q * 3 + c; slot = pt_plane(c, 0, q); }                                 // its bottom row
        else if (q < 2 * TWv) { have = Y0 + THv < H; nt = tile + A.tilesX; off = (q - TWv) * 3 + c; slot = pt_plane(c, THv + 1, q - TWv); }        // its top row
        else if (q < 2 * TWv + THv) { have = tX > 0; nt = tile - 1; off = 2 * PT_W * 3 + PT_MAXH * 3 + (q - 2 * TWv) * 3 + c; slot = PT_COLL + c * PT_MAXH + (q - 2 * TWv); }   // its right column
        else { have = X0 + TWv < W; nt = tile + 1; off = 2 * PT_W * 3 + (q - 2 * TWv - THv) * 3 + c; slot = PT_COLR + c * PT_MAXH + (q - 2 * TWv - THv); }      // its left column
        return have;
    };
    int h_slot = 0, h_c = 0;
    unsigned long long *h_ptr = A.halo;
    bool h_have = false;
    if (t < ringN) {
        int nt, off;
        h_have = ring_source(t, nt, off, h_slot, h_c);
        if (h_have) h_ptr = &A.halo[(size_t)nt * PT_HALO + off];
    }
    float rz[3];
    rz[0] = A.s_rz[0]; rz[1] = A.s_rz[1]; rz[2] = A.s_rz[2];
    bool ok = true;
    if (t == 0) s_fail = 0;
    if (A.debugFail && tile == 0 && t == 0) __hip_atomic_store(A.bar + 1, 1u, PT_RLX_AGENT);
    unsigned long long *gatA = A.gat, *gatB = A.gat + 3 * PT_MAXG;
    unsigned *err = A.bar + 1;

#ifdef GDPT_PT_TIMING
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tc = wall_clock64(), tn;
#define PT_TICK(k) { tn = wall_clock64(); tph[k] += tn - tc; tc = tn; }
#else
#define PT_TICK(k)
#endif
    if constexpr (SINGLE) {
        float zv[12];
#pragma unroll
        for (int k = 0; k < 12; k++) zv[k] = 0.0f;
        float gOld[3] = {1.0f, 1.0f, 1.0f}, aOld[3] = {1.0f, 1.0f, 1.0f};
        for (int it = 0; it < A.iters && ok; it++) {
            // ---- w = A r (tile of r staged in LDS), partials r.r and w.r ----
            pt_sync();
            if (valid) {
#pragma unroll
                for (int c = 0; c < 3; c++)
                    *reinterpret_cast<float4 *>(&lp[pt_plane(c, ty + 1, 4 * tx)]) = make_float4(rv[c], rv[3 + c], rv[6 + c], rv[9 + c]);
            }
            float lft[3], rgt[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { lft[c] = pt_from_left(rv[9 + c]); rgt[c] = pt_from_right(rv[c]); }
            pt_sync();
            float wv[12], accG[3] = {0.0f, 0.0f, 0.0f}, accD[3] = {0.0f, 0.0f, 0.0f};
            if (valid) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float4 u4 = *reinterpret_cast<const float4 *>(&lp[pt_plane(c, ty, 4 * tx)]);
                    const float4 d4 = *reinterpret_cast<const float4 *>(&lp[pt_plane(c, ty + 2, 4 * tx)]);
                    const float up[4] = {u4.x, u4.y, u4.z, u4.w}, dn[4] = {d4.x, d4.y, d4.z, d4.w};
                    if (tx == 0) lft[c] = lp[PT_COLL + c * PT_MAXH + ty];
                    if (4 * tx + 4 == TWv) rgt[c] = lp[PT_COLR + c * PT_MAXH + ty];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int xx = x + k;
                        const float xi = rv[3 * k + c];
                        const float wl = (k == 0) ? w1l : w1[k - 1];
                        const float xl = (k == 0) ? lft[c] : rv[3 * (k - 1) + c], xr = (k == 3) ? rgt[c] : rv[3 * (k + 1) + c];
                        float a = w0[k] * xi * alphaSqr;                                   // Backend.cpp:228-233
                        if (xx != 0)     a = a + wl * (xi - xl);
                        if (xx != W - 1) a = a + w1[k] * (xi - xr);
                        if (y != 0)      a = a + wu[k] * (xi - up[k]);
                        if (y != H - 1)  a = a + wv_[k] * (xi - dn[k]);
                        wv[3 * k + c] = a;
                        accG[c] += xi * xi;
                        accD[c] += xi * a;
                    }
                }
            }
            pt_block_sum3(accG, sm, nwaves);
            pt_block_sum3(accD, sm, nwaves);
            float gam[3], del[3], al[3], be[3];
            unsigned long long *tg = A.gat + (size_t)(it & 1) * 6 * PT_MAXG;
            ok = pt_allgather_sum6(tg, tg + 3 * PT_MAXG, A.tagBase + (unsigned)it + 1u, G, tile, accG, accD, gam, del, gsm, &s_fail, err);
            if (!ok) break;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                be[c] = it == 0 ? 0.0f : gam[c] / fmaxf(gOld[c], FLT_MIN);
                const float pAp = it == 0 ? del[c] : del[c] - be[c] * gam[c] / aOld[c];    // p.Ap = w.r - beta gamma / alpha_old
                al[c] = gam[c] / fmaxf(pAp, FLT_MIN);
                gOld[c] = gam[c]; aOld[c] = fmaxf(al[c], FLT_MIN);
            }
            if (valid) {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const int q = 3 * k + c;
                        zv[q] = wv[q] + zv[q] * be[c];
                        pv[q] = it == 0 ? rv[q] : rv[q] + pv[q] * be[c];
                        xv[q] = xv[q] + pv[q] * al[c];
                        rv[q] = rv[q] - zv[q] * al[c];
                    }
                if (ty == 0) for (int k = 0; k < 12; k++) hs[(4 * tx) * 3 + k] = rv[k];
                if (ty == THv - 1) for (int k = 0; k < 12; k++) hs[PT_W * 3 + (4 * tx) * 3 + k] = rv[k];
                if (tx == 0) for (int c = 0; c < 3; c++) hs[2 * PT_W * 3 + ty * 3 + c] = rv[c];
                if (4 * tx + 4 == TWv) for (int c = 0; c < 3; c++) hs[2 * PT_W * 3 + PT_MAXH * 3 + ty * 3 + c] = rv[9 + c];
            }
            pt_sync();
            for (int e = t; e < PT_HALO; e += blockDim.x) pt_store(&myHalo[e], hs[e], A.tagBase + (unsigned)it + 1u);
            // the ring of the next stencil: the neighbours' new r
            for (int e = t; e < ringN; e += blockDim.x) {
                int nt, off, slot, c;
                if (ring_source(e, nt, off, slot, c)) {
                    unsigned long long *hp = &A.halo[(size_t)nt * PT_HALO + off];
                    lp[slot] = pt_wait(hp, __hip_atomic_load(hp, PT_RLX_AGENT), A.tagBase + (unsigned)it + 1u, &s_fail, err);
                }
            }
            rz[0] = gam[0]; rz[1] = gam[1]; rz[2] = gam[2];
        }
    } else
    for (int it = 0; it < A.iters && ok; it++) {
        // ---- Ap = A p (tile staged in LDS), partial p.Ap ----
        pt_sync();
        PT_TICK(5)
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; c++)
                *reinterpret_cast<float4 *>(&lp[pt_plane(c, ty + 1, 4 * tx)]) = make_float4(pv[c], pv[3 + c], pv[6 + c], pv[9 + c]);
        }
        float lft[3], rgt[3];
#pragma unroll
        for (int c = 0; c < 3; c++) { lft[c] = pt_from_left(pv[9 + c]); rgt[c] = pt_from_right(pv[c]); }   // all lanes take part
        pt_sync();
        float Ap[12], acc[3] = {0.0f, 0.0f, 0.0f};
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float4 u4 = *reinterpret_cast<const float4 *>(&lp[pt_plane(c, ty, 4 * tx)]);
                const float4 d4 = *reinterpret_cast<const float4 *>(&lp[pt_plane(c, ty + 2, 4 * tx)]);
                const float up[4] = {u4.x, u4.y, u4.z, u4.w}, dn[4] = {d4.x, d4.y, d4.z, d4.w};
                if (tx == 0) lft[c] = lp[PT_COLL + c * PT_MAXH + ty];
                if (4 * tx + 4 == TWv) rgt[c] = lp[PT_COLR + c * PT_MAXH + ty];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int xx = x + k;
                    const float xi = pv[3 * k + c];
                    const float wl = (k == 0) ? w1l : w1[k - 1];
                    const float xl = (k == 0) ? lft[c] : pv[3 * (k - 1) + c], xr = (k == 3) ? rgt[c] : pv[3 * (k + 1) + c];
                    float a = w0[k] * xi * alphaSqr;                                   // Backend.cpp:228-233
                    if (xx != 0)     a = a + wl * (xi - xl);
                    if (xx != W - 1) a = a + w1[k] * (xi - xr);
                    if (y != 0)      a = a + wu[k] * (xi - up[k]);
                    if (y != H - 1)  a = a + wv_[k] * (xi - dn[k]);
                    Ap[3 * k + c] = a;
                    acc[c] += xi * a;
                }
            }
        }
        pt_block_sum3(acc, sm, nwaves);
        PT_TICK(0)
        float pAp[3], a[3];
        ok = pt_allgather_sum(gatA, A.tagBase + (unsigned)it + 1u, G, tile, acc, pAp, gsm, &s_fail, err);
        if (!ok) break;
        PT_TICK(1)

        // ---- a = rz / pAp ; x += p a ; r -= Ap a ; partial r.r ; publish the boundary of r ----
#pragma unroll
        for (int c = 0; c < 3; c++) a[c] = rz[c] / fmaxf(pAp[c], FLT_MIN);              // Backend.cpp:301
        float acc2[3] = {0.0f, 0.0f, 0.0f};
        if (valid) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    xv[3 * k + c] = xv[3 * k + c] + pv[3 * k + c] * a[c];               // Backend.cpp:343
                    const float ri = rv[3 * k + c] - Ap[3 * k + c] * a[c];              // Backend.cpp:305-307
                    rv[3 * k + c] = ri;
                    acc2[c] += ri * ri;
                }
            // boundary of r -> LDS record (top row, bottom row, left column, right column)
            if (ty == 0) for (int k = 0; k < 12; k++) hs[(4 * tx) * 3 + k] = rv[k];
            if (ty == THv - 1) for (int k = 0; k < 12; k++) hs[PT_W * 3 + (4 * tx) * 3 + k] = rv[k];
            if (tx == 0) for (int c = 0; c < 3; c++) hs[2 * PT_W * 3 + ty * 3 + c] = rv[c];
            if (4 * tx + 4 == TWv) for (int c = 0; c < 3; c++) hs[2 * PT_W * 3 + PT_MAXH * 3 + ty * 3 + c] = rv[9 + c];
        }
        pt_block_sum3(acc2, sm, nwaves);
        // publish the record: consecutive lanes write consecutive tagged words (coalesced write-through, not one fabric write per lane)
        for (int e = t; e < PT_HALO; e += blockDim.x) pt_store(&myHalo[e], hs[e], A.tagBase + (unsigned)it + 1u);
        PT_TICK(2)
        const unsigned long long hw0 = h_have ? __hip_atomic_load(h_ptr, PT_RLX_AGENT) : 0ull;   // in flight during the gather
        float rzn[3], b[3];
        ok = pt_allgather_sum(gatB, A.tagBase + (unsigned)it + 1u, G, tile, acc2, rzn, gsm, &s_fail, err);
        if (!ok) break;
        PT_TICK(3)

        // ---- b = rz_new / rz ; p = r + p b, for the tile and (redundantly) its ring ----
#pragma unroll
        for (int c = 0; c < 3; c++) { b[c] = rzn[c] / fmaxf(rz[c], FLT_MIN); rz[c] = rzn[c]; }   // Backend.cpp:336
        if (valid) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) pv[3 * k + c] = rv[3 * k + c] + pv[3 * k + c] * b[c];  // Backend.cpp:344
        }
        if (h_have) {
            const float rn = pt_wait(h_ptr, hw0, A.tagBase + (unsigned)it + 1u, &s_fail, err);
            lp[h_slot] = rn + lp[h_slot] * b[h_c];
        }
        for (int e = t + blockDim.x; e < ringN; e += blockDim.x) {         // tiles so small that a lane serves several ring floats
            int nt, off, slot, c;
            if (ring_source(e, nt, off, slot, c)) {
                unsigned long long *hp = &A.halo[(size_t)nt * PT_HALO + off];
                const float rn = pt_wait(hp, __hip_atomic_load(hp, PT_RLX_AGENT), A.tagBase + (unsigned)it + 1u, &s_fail, err);
                lp[slot] = rn + lp[slot] * b[c];
            }
        }
        PT_TICK(4)
    }
#ifdef GDPT_PT_TIMING
    if (t == 0 && (tile == 0 || tile == G / 2)) for (int k = 0; k < 6; k++) A.bar[16 + (tile ? 8 : 0) + k] = (unsigned)tph[k];
#endif

    // ---- write the iterate back ----
    if (valid) {
        float4 *x4 = reinterpret_cast<float4 *>(A.x + 3 * (size_t)i), *r4 = reinterpret_cast<float4 *>(A.r + 3 * (size_t)i), *p4 = reinterpret_cast<float4 *>(A.p + 3 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            x4[k] = make_float4(xv[4 * k], xv[4 * k + 1], xv[4 * k + 2], xv[4 * k + 3]);
            r4[k] = make_float4(rv[4 * k], rv[4 * k + 1], rv[4 * k + 2], rv[4 * k + 3]);
            p4[k] = make_float4(pv[4 * k], pv[4 * k + 1], pv[4 * k + 2], pv[4 * k + 3]);
        }
    }
    if (tile == 0 && t == 0) { A.s_rz[0] = rz[0]; A.s_rz[1] = rz[1]; A.s_rz[2] = rz[2]; }
}

// ---- the same for images of 1-2 Mpixel: 128-px wide tiles, 8 px per lane, p in LDS only -------------------------------------------------
// 1920x1080 needs 8 100 px per CU.  A lane of kp_cg holds x, r, p, Ap and the IRLS weights of 4 px in ~125 registers; with 8 px that
// would be 250.  kp_cg2 keeps what has to survive the two gathers in registers (x, r, Ap of two 4-px groups: 72 floats) and everything
// else where it is cheap to fetch each iteration: p lives in the LDS image the stencil reads anyway (three planes of (TH+2) x 128 floats,
// 101 KB; p = r + b p is a read-modify-write of a lane's own pixels there), the IRLS weights are re-read from L2 / MALL (25 MB per
// iteration for the whole image, 98 KB per CU).  Tiles: 128 x TH, TH <= 64, one workgroup of 1024 threads per CU; lane (tx, ty) owns the
// pixels [4 tx, 4 tx + 4) and [64 + 4 tx, 64 + 4 tx + 4) of row ty.  Gathers, tagged halo records, time-outs and the arithmetic per element
// are kp_cg's.  1920x1080 = 15 x 17 tiles of 128 x 64 = 255 workgroups.
constexpr int P2_W = 128;
constexpr int P2_PLANE = (PT_MAXH + 2) * P2_W;
constexpr int P2_COLL = 3 * P2_PLANE, P2_COLR = P2_COLL + 3 * PT_MAXH, P2_LDS = P2_COLR + 3 * PT_MAXH;
constexpr int P2_HALO = (2 * P2_W + 2 * PT_MAXH) * 3;
constexpr int P2_WV = (PT_MAXH + 1) * P2_W;       // LDS image of the vertical IRLS weights: row 0 = the image row above the tile, row r + 1 = tile row r
constexpr size_t P2_SHARED_BYTES = sizeof(float) * (P2_LDS + P2_HALO + 3 * PT_MAXG + 64 + P2_WV) + 16;
__device__ __forceinline__ int p2_plane(int c, int row, int x) { return c * P2_PLANE + row * P2_W + x; }

template <bool UNITW>
__global__ __launch_bounds__(1024) void kp_cg2(PersistArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float p2_dyn[];
    float *lp = p2_dyn;                           // p of the tile with its ring
    float *hs = lp + P2_LDS;
    float *gsm = hs + P2_HALO;
    float *sm = gsm + 3 * PT_MAXG;
    float *wvs = sm + 64;                         // (constant over the launch: read back by ds_read_b128 instead of two float4 loads from L2 per group and iteration)
    int *s_failp = reinterpret_cast<int *>(wvs + P2_WV);
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, nwaves = (blockDim.x + 63) >> 6;
    const int tile = blockIdx.x, tX = tile % A.tilesX, tY = tile / A.tilesX;
    const int X0 = tX * P2_W, Y0 = tY * A.TH;
    const int TWv = min(P2_W, A.W - X0), THv = min(A.TH, A.H - Y0);
    const int y = Y0 + ty;
    const int W = A.W, H = A.H, n = W * H;
    const float alphaSqr = A.alpha * A.alpha;
    const int G = A.tilesX * A.tilesY;
    unsigned long long *myHalo = A.halo + (size_t)tile * P2_HALO;
    const int lx[2] = {4 * tx, 64 + 4 * tx};                                     // the lane's two pixel groups, tile-local x
    const bool valid[2] = {ty < THv && lx[0] < TWv, ty < THv && lx[1] < TWv};
    const int gi[2] = {y * W + X0 + lx[0], y * W + X0 + lx[1]};

    float xv[2][12], rv[2][12], Ap[2][12];
#pragma unroll
    for (int g = 0; g < 2; g++) {
#pragma unroll
        for (int k = 0; k < 12; k++) { xv[g][k] = rv[g][k] = Ap[g][k] = 0.0f; }
        if (valid[g]) {
            const float4 *x4 = reinterpret_cast<const float4 *>(A.x + 3 * (size_t)gi[g]), *r4 = reinterpret_cast<const float4 *>(A.r + 3 * (size_t)gi[g]);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float4 a = x4[k], b = r4[k];
                xv[g][4 * k] = a.x; xv[g][4 * k + 1] = a.y; xv[g][4 * k + 2] = a.z; xv[g][4 * k + 3] = a.w;
                rv[g][4 * k] = b.x; rv[g][4 * k + 1] = b.y; rv[g][4 * k + 2] = b.z; rv[g][4 * k + 3] = b.w;
            }
#pragma unroll
            for (int c = 0; c < 3; c++)                                          // p := r (Solver.cpp:405)
                *reinterpret_cast<float4 *>(&lp[p2_plane(c, ty + 1, lx[g])]) = make_float4(rv[g][c], rv[g][3 + c], rv[g][6 + c], rv[g][9 + c]);
        }
    }
    if (!UNITW) {
#pragma unroll
        for (int g = 0; g < 2; g++)
            if (valid[g]) {
                *reinterpret_cast<float4 *>(&wvs[(ty + 1) * P2_W + lx[g]]) = *reinterpret_cast<const float4 *>(A.w2 + 2 * n + gi[g]);
                if (ty == 0) *reinterpret_cast<float4 *>(&wvs[lx[g]]) = y != 0 ? *reinterpret_cast<const float4 *>(A.w2 + 2 * n + gi[g] - W) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
    }
    const int ringN = (2 * TWv + 2 * THv) * 3;
    for (int e = t; e < ringN; e += blockDim.x) {
        const int c = e % 3, q = e / 3;
        int gx, gy, slot;
        if (q < TWv) { gx = X0 + q; gy = Y0 - 1; slot = p2_plane(c, 0, q); }
        else if (q < 2 * TWv) { gx = X0 + (q - TWv); gy = Y0 + THv; slot = p2_plane(c, THv + 1, q - TWv); }
        else if (q < 2 * TWv + THv) { gx = X0 - 1; gy = Y0 + (q - 2 * TWv); slot = P2_COLL + c * PT_MAXH + (q - 2 * TWv); }
        else { gx = X0 + TWv; gy = Y0 + (q - 2 * TWv - THv); slot = P2_COLR + c * PT_MAXH + (q - 2 * TWv - THv); }
        lp[slot] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? A.r[3 * ((size_t)gy * W + gx) + c] : 0.0f;
    }
    auto ring_source = [&](int e, int &nt, int &off, int &slot, int &c) -> bool {
        c = e % 3;
        const int q = e / 3;
        bool have;
        if (q < TWv) { have = tY > 0; nt = tile - A.tilesX; off = P2_W * 3 + q * 3 + c; slot = p2_plane(c, 0, q); }                                 // its bottom row
        else if (q < 2 * TWv) { have = Y0 + THv < H; nt = tile + A.tilesX; off = (q - TWv) * 3 + c; slot = p2_plane(c, THv + 1, q - TWv); }        // its top row
        else if (q < 2 * TWv + THv) { have = tX > 0; nt = tile - 1; off = 2 * P2_W * 3 + PT_MAXH * 3 + (q - 2 * TWv) * 3 + c; slot = P2_COLL + c * PT_MAXH + (q - 2 * TWv); }   // its right column
        else { have = X0 + TWv < W; nt = tile + 1; off = 2 * P2_W * 3 + (q - 2 * TWv - THv) * 3 + c; slot = P2_COLR + c * PT_MAXH + (q - 2 * TWv - THv); }      // its left column
        return have;
    };
    float rz[3];
    rz[0] = A.s_rz[0]; rz[1] = A.s_rz[1]; rz[2] = A.s_rz[2];
    bool ok = true;
    if (t == 0) *s_failp = 0;
    if (A.debugFail && tile == 0 && t == 0) __hip_atomic_store(A.bar + 1, 1u, PT_RLX_AGENT);
    unsigned long long *gatA = A.gat, *gatB = A.gat + 3 * PT_MAXG;
    unsigned *err = A.bar + 1;

    for (int it = 0; it < A.iters && ok; it++) {
        // ---- Ap = A p from the LDS image, partial p.Ap ----
        pt_sync();
        float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int x = X0 + lx[g];
            // the 17 weights the group's stencil needs: w0, w1 and the left neighbour's w1 from L2 / MALL, the two vertical ones from their LDS image; fetched here, group by group: holding group 0's across the gathers (a prefetch
            // during the wait for the ring) or both groups' at once spills, and cost 10 us per iteration when tried
            const int i = gi[g];
            float w0[4] = {1.0f, 1.0f, 1.0f, 1.0f}, w1[4] = {1.0f, 1.0f, 1.0f, 1.0f}, wv_[4] = {1.0f, 1.0f, 1.0f, 1.0f}, wu[4] = {1.0f, 1.0f, 1.0f, 1.0f}, w1l = 1.0f;
            if (!UNITW && valid[g]) {
                const float4 a = *reinterpret_cast<const float4 *>(A.w2 + i), b = *reinterpret_cast<const float4 *>(A.w2 + n + i);
                const float4 c = *reinterpret_cast<const float4 *>(&wvs[(ty + 1) * P2_W + lx[g]]), d = *reinterpret_cast<const float4 *>(&wvs[ty * P2_W + lx[g]]);
                w0[0] = a.x; w0[1] = a.y; w0[2] = a.z; w0[3] = a.w;
                w1[0] = b.x; w1[1] = b.y; w1[2] = b.z; w1[3] = b.w;
                wv_[0] = c.x; wv_[1] = c.y; wv_[2] = c.z; wv_[3] = c.w;
                w1l = (x != 0) ? A.w2[n + i - 1] : 0.0f;
                wu[0] = d.x; wu[1] = d.y; wu[2] = d.z; wu[3] = d.w;                      // (zeros in the image's first row)
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float4 m4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), u4 = m4, d4 = m4;
                if (valid[g]) {
                    m4 = *reinterpret_cast<const float4 *>(&lp[p2_plane(c, ty + 1, lx[g])]);
                    u4 = *reinterpret_cast<const float4 *>(&lp[p2_plane(c, ty, lx[g])]);
                    d4 = *reinterpret_cast<const float4 *>(&lp[p2_plane(c, ty + 2, lx[g])]);
                }
                float lft = pt_from_left(m4.w), rgt = pt_from_right(m4.x);                    // all lanes take part
                if (valid[g]) {
                    if (tx == 0) lft = (g == 0) ? lp[P2_COLL + c * PT_MAXH + ty] : lp[p2_plane(c, ty + 1, 63)];
                    if (lx[g] + 4 == TWv) rgt = lp[P2_COLR + c * PT_MAXH + ty];
                    else if (tx == 15) rgt = lp[p2_plane(c, ty + 1, 64)];                      // (g == 0 here: the first pixel of the right half)
                    const float pm[4] = {m4.x, m4.y, m4.z, m4.w}, up[4] = {u4.x, u4.y, u4.z, u4.w}, dn[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int xx = x + k;
                        const float xi = pm[k];
                        const float wl = (k == 0) ? w1l : w1[k - 1];
                        const float xl = (k == 0) ? lft : pm[k - 1], xr = (k == 3) ? rgt : pm[k + 1];
                        float a = w0[k] * xi * alphaSqr;                                   // Backend.cpp:228-233
                        if (xx != 0)     a = a + wl * (xi - xl);
                        if (xx != W - 1) a = a + w1[k] * (xi - xr);
                        if (y != 0)      a = a + wu[k] * (xi - up[k]);
                        if (y != H - 1)  a = a + wv_[k] * (xi - dn[k]);
                        Ap[g][3 * k + c] = a;
                        acc[c] += xi * a;
                    }
                }
            }
        }
        pt_block_sum3(acc, sm, nwaves);
        float pAp[3], a[3];
        ok = pt_allgather_sum(gatA, A.tagBase + (unsigned)it + 1u, G, tile, acc, pAp, gsm, s_failp, err);
        if (!ok) break;

        // ---- a = rz / pAp ; x += p a ; r -= Ap a ; partial r.r ; publish the boundary of r ----
#pragma unroll
        for (int c = 0; c < 3; c++) a[c] = rz[c] / fmaxf(pAp[c], FLT_MIN);              // Backend.cpp:301
        float acc2[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (valid[g]) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float4 m4 = *reinterpret_cast<const float4 *>(&lp[p2_plane(c, ty + 1, lx[g])]);
                    const float pm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        xv[g][3 * k + c] = xv[g][3 * k + c] + pm[k] * a[c];             // Backend.cpp:343
                        const float ri = rv[g][3 * k + c] - Ap[g][3 * k + c] * a[c];    // Backend.cpp:305-307
                        rv[g][3 * k + c] = ri;
                        acc2[c] += ri * ri;
                    }
                }
                if (ty == 0) for (int k = 0; k < 12; k++) hs[lx[g] * 3 + k] = rv[g][k];
                if (ty == THv - 1) for (int k = 0; k < 12; k++) hs[P2_W * 3 + lx[g] * 3 + k] = rv[g][k];
                if (lx[g] == 0) for (int c = 0; c < 3; c++) hs[2 * P2_W * 3 + ty * 3 + c] = rv[g][c];
                if (lx[g] + 4 == TWv) for (int c = 0; c < 3; c++) hs[2 * P2_W * 3 + PT_MAXH * 3 + ty * 3 + c] = rv[g][9 + c];
            }
        }
        pt_block_sum3(acc2, sm, nwaves);
        for (int e = t; e < P2_HALO; e += blockDim.x) pt_store(&myHalo[e], hs[e], A.tagBase + (unsigned)it + 1u);
        float rzn[3], b[3];
        ok = pt_allgather_sum(gatB, A.tagBase + (unsigned)it + 1u, G, tile, acc2, rzn, gsm, s_failp, err);
        if (!ok) break;

        // ---- b = rz_new / rz ; p = r + p b in the LDS image, for the tile and (redundantly) its ring ----
#pragma unroll
        for (int c = 0; c < 3; c++) { b[c] = rzn[c] / fmaxf(rz[c], FLT_MIN); rz[c] = rzn[c]; }   // Backend.cpp:336
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (valid[g]) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float4 *q = reinterpret_cast<float4 *>(&lp[p2_plane(c, ty + 1, lx[g])]);
                    const float4 m4 = *q;
                    *q = make_float4(rv[g][c] + m4.x * b[c], rv[g][3 + c] + m4.y * b[c], rv[g][6 + c] + m4.z * b[c], rv[g][9 + c] + m4.w * b[c]);   // Backend.cpp:344
                }
            }
        }
        for (int e = t; e < ringN; e += blockDim.x) {
            int nt, off, slot, c;
            if (ring_source(e, nt, off, slot, c)) {
                unsigned long long *hp = &A.halo[(size_t)nt * P2_HALO + off];
                const float rn = pt_wait(hp, __hip_atomic_load(hp, PT_RLX_AGENT), A.tagBase + (unsigned)it + 1u, s_failp, err);
                lp[slot] = rn + lp[slot] * b[c];
            }
        }
    }

    // ---- write the iterate back ----
    pt_sync();
#pragma unroll
    for (int g = 0; g < 2; g++) {
        if (valid[g]) {
            float4 *x4 = reinterpret_cast<float4 *>(A.x + 3 * (size_t)gi[g]), *r4 = reinterpret_cast<float4 *>(A.r + 3 * (size_t)gi[g]), *p4 = reinterpret_cast<float4 *>(A.p + 3 * (size_t)gi[g]);
            float pv[12];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float4 m4 = *reinterpret_cast<const float4 *>(&lp[p2_plane(c, ty + 1, lx[g])]);
                pv[c] = m4.x; pv[3 + c] = m4.y; pv[6 + c] = m4.z; pv[9 + c] = m4.w;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                x4[k] = make_float4(xv[g][4 * k], xv[g][4 * k + 1], xv[g][4 * k + 2], xv[g][4 * k + 3]);
                r4[k] = make_float4(rv[g][4 * k], rv[g][4 * k + 1], rv[g][4 * k + 2], rv[g][4 * k + 3]);
                p4[k] = make_float4(pv[4 * k], pv[4 * k + 1], pv[4 * k + 2], pv[4 * k + 3]);
            }
        }
    }
    if (tile == 0 && t == 0) { A.s_rz[0] = rz[0]; A.s_rz[1] = rz[1]; A.s_rz[2] = rz[2]; }
}

} // namespace gdpt
