// poisson_persistent.hip.h -- the whole CG loop of one IRLS iteration as ONE persistent cooperative kernel.
//
// Why: at the image sizes G-PT renders (1280x720: x, r, p, Ap, w = 55 MB) the CG iterate fits the chip's register file
// (256 CUs x 512 KB), so the only thing that has to leave a CU per iteration is a one-pixel ring of r and two 3-float dot
// products.  Each workgroup owns a 64 x TH tile for the whole solve, keeps x, r, p (and w) of its 4 px per lane in VGPRs,
// stages p (+ring) through LDS for the 5-point stencil, and meets the other workgroups at TWO grid barriers per iteration
// (after p.Ap, after r.r) -- the two global reductions the CG recurrence of the reference has (Solver.cpp:466-469); nothing
// else is synchronised and no array is streamed through HBM.  The arithmetic per element is that of kf_Ax / kf_r_rz / kf_x_p
// (reference association order, -ffp-contract=off); only the summation tree of the dot products differs (per-tile partials).
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, form "agent-scope atomics on both sides"): everything one
// workgroup hands to another in-launch -- its tile partial and the boundary ring of r -- is written with relaxed AGENT-scope
// atomic stores (write-through, sc1) and read with relaxed agent-scope atomic loads (L1 bypassed), so no release/acquire cache
// maintenance is needed; every storing wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup arrives.  The grid
// barrier is hierarchical: 8 arrival counters (workgroup b reports to counter b % 8, the observed XCD of block b, so that the
// 32 arrivals per counter stay on one L2), the last arriver of each group reports to a top counter, waits for all groups and
// then releases its group through a generation word the others poll with s_sleep.  Grouping by b % 8 is only a speed choice:
// correctness does not depend on where a workgroup runs.  All words are zeroed by a memset node before every launch, every
// spin is bounded, and a timeout raises a sticky flag the host turns into GDPT_ERR_HIP instead of hanging.  Residency comes
// from the grid size (tiles <= CUs, one workgroup per CU) and is checked by hipLaunchCooperativeKernel.
#pragma once
#include "poisson_kernels.hip.h"

namespace gdpt {

constexpr int PT_W = 64;                 // tile width in pixels: 16 lanes x 4 px
constexpr int PT_MAXH = 64;              // tile rows <= 64 (16 lanes per row -> <= 1024 threads)
constexpr int PT_RS = (PT_W + 2) * 3 + 2; // LDS row stride in floats (ring pixel each side, +2 pad)
constexpr int PT_HALO = (2 * PT_W + 2 * PT_MAXH) * 3;   // floats a tile publishes per iteration: top, bottom, left, right
constexpr unsigned PT_SPIN_LIMIT = 4000000u;

struct PersistArgs {
    float *x, *r, *p;                    // images (AoS RGB); p is written back at the end
    const float *w2;                     // 3n weights (ignored when UNITW)
    float4 *part_a, *part_b;             // per-tile partials of p.Ap and r.r
    float *halo;                         // [tiles][PT_HALO] boundary r of every tile
    unsigned *bar;                       // PT_BAR_WORDS words: [1] error flag, arrival counters, top counter, generation words
    float *s_rz;                         // in: r.r of the prologue; out: final r.r
    int W, H, tilesX, tilesY, TH, iters;
    float alpha;
};

#define PT_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int PT_BAR_WORDS = 32 * 20;    // [1] error flag; counters and generation words 128 B apart

__device__ __forceinline__ void pt_store(float *p, float v) { __hip_atomic_store(p, v, PT_RLX_AGENT); }
__device__ __forceinline__ float pt_load(float *p) { return __hip_atomic_load(p, PT_RLX_AGENT); }

__device__ __forceinline__ bool pt_spin(unsigned *word, unsigned target, unsigned *err)
{
    unsigned spins = 0;
    while (__hip_atomic_load(word, PT_RLX_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0 && __hip_atomic_load(err, PT_RLX_AGENT) != 0) return false;
        if (spins > PT_SPIN_LIMIT) { __hip_atomic_store(err, 1u, PT_RLX_AGENT); return false; }
    }
    return true;
}

// Grid barrier number `epoch` (1, 2, ...) over G workgroups.
__device__ __forceinline__ bool pt_barrier(unsigned *bar, unsigned epoch, int G)
{
    __shared__ int s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // EVERY wave drains its write-through stores before arriving
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = blockIdx.x & 7, groups = G < 8 ? G : 8;
        const unsigned members = (unsigned)((G - g + 7) / 8);
        unsigned *cnt = bar + 32 * (1 + g), *top = bar + 32 * 9, *gen = bar + 32 * (10 + g), *err = bar + 1;
        bool ok;
        if (__hip_atomic_fetch_add(cnt, 1u, PT_RLX_AGENT) + 1u == members * epoch) {      // last arriver of its group
            __hip_atomic_fetch_add(top, 1u, PT_RLX_AGENT);
            ok = pt_spin(top, (unsigned)groups * epoch, err);
            __hip_atomic_store(gen, epoch, PT_RLX_AGENT);
        } else ok = pt_spin(gen, epoch, err);
        s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return s_ok != 0;
}

// Sum a[0..2] over a block of up to 16 waves; every thread receives the totals.  sm: >= 64 floats.
__device__ __forceinline__ void pt_block_sum3(float (&a)[3], float *sm, int nwaves)
{
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 3; c++) a[c] = wave_sum(a[c]);
    __syncthreads();
    if (ln == 0) { sm[wv * 4 + 0] = a[0]; sm[wv * 4 + 1] = a[1]; sm[wv * 4 + 2] = a[2]; }
    __syncthreads();
    float t[3] = {0.0f, 0.0f, 0.0f};
    for (int w = 0; w < nwaves; w++) { t[0] += sm[w * 4]; t[1] += sm[w * 4 + 1]; t[2] += sm[w * 4 + 2]; }
    a[0] = t[0]; a[1] = t[1]; a[2] = t[2];
}

// Fixed-order total of the G tile partials (agent-scope atomic loads: L1 is bypassed, the scalar cache never involved).
__device__ __forceinline__ void pt_reduce_parts(float4 *part, int G, float (&v)[3], float *sm, int nwaves)
{
    v[0] = v[1] = v[2] = 0.0f;
    for (int i = threadIdx.x; i < G; i += blockDim.x) {
        float *q = reinterpret_cast<float *>(&part[i]);
        v[0] += pt_load(q); v[1] += pt_load(q + 1); v[2] += pt_load(q + 2);
    }
    pt_block_sum3(v, sm, nwaves);
}

template <bool UNITW>
__global__ __launch_bounds__(1024) void kp_cg(PersistArgs A)
{
    __shared__ __attribute__((aligned(16))) float lp[(PT_MAXH + 2) * PT_RS];   // p of the tile with its ring
    __shared__ float sm[64];
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
    float *myHalo = A.halo + (size_t)tile * PT_HALO;

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
        if (q < TWv) { gx = X0 + q; gy = Y0 - 1; slot = 0 * PT_RS + (q + 1) * 3 + c; }
        else if (q < 2 * TWv) { gx = X0 + (q - TWv); gy = Y0 + THv; slot = (THv + 1) * PT_RS + (q - TWv + 1) * 3 + c; }
        else if (q < 2 * TWv + THv) { gx = X0 - 1; gy = Y0 + (q - 2 * TWv); slot = (q - 2 * TWv + 1) * PT_RS + 0 * 3 + c; }
        else { gx = X0 + TWv; gy = Y0 + (q - 2 * TWv - THv); slot = (q - 2 * TWv - THv + 1) * PT_RS + (TWv + 1) * 3 + c; }
        lp[slot] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? A.r[3 * ((size_t)gy * W + gx) + c] : 0.0f;
    }
    float rz[3];
    rz[0] = A.s_rz[0]; rz[1] = A.s_rz[1]; rz[2] = A.s_rz[2];
    unsigned epoch = 0;
    bool ok = true;

    for (int it = 0; it < A.iters && ok; it++) {
        // ---- Ap = A p (tile staged in LDS), partial p.Ap ----
        __syncthreads();
        if (valid) {
            float *row = lp + (ty + 1) * PT_RS + (4 * tx + 1) * 3;
#pragma unroll
            for (int k = 0; k < 12; k++) row[k] = pv[k];
        }
        __syncthreads();
        float Ap[12], acc[3] = {0.0f, 0.0f, 0.0f};
        if (valid) {
            const float *c0 = lp + (ty + 1) * PT_RS + (4 * tx + 1) * 3;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int xx = x + k;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float xi = pv[3 * k + c];
                    const float wl = (k == 0) ? w1l : w1[k - 1];
                    float a = w0[k] * xi * alphaSqr;                                   // Backend.cpp:228-233
                    if (xx != 0)     a = a + wl * (xi - c0[3 * (k - 1) + c]);
                    if (xx != W - 1) a = a + w1[k] * (xi - c0[3 * (k + 1) + c]);
                    if (y != 0)      a = a + wu[k] * (xi - c0[3 * k + c - PT_RS]);
                    if (y != H - 1)  a = a + wv_[k] * (xi - c0[3 * k + c + PT_RS]);
                    Ap[3 * k + c] = a;
                    acc[c] += xi * a;
                }
            }
        }
        pt_block_sum3(acc, sm, nwaves);
        if (t == 0) { float *q = reinterpret_cast<float *>(&A.part_a[tile]); pt_store(q, acc[0]); pt_store(q + 1, acc[1]); pt_store(q + 2, acc[2]); }
        ok = pt_barrier(A.bar, ++epoch, G);
        if (!ok) break;

        // ---- a = rz / pAp ; x += p a ; r -= Ap a ; partial r.r ; publish the boundary of r ----
        float pAp[3], a[3];
        pt_reduce_parts(A.part_a, G, pAp, sm, nwaves);
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
            if (ty == 0) for (int k = 0; k < 12; k++) pt_store(&myHalo[(4 * tx) * 3 + k], rv[k]);
            if (ty == THv - 1) for (int k = 0; k < 12; k++) pt_store(&myHalo[PT_W * 3 + (4 * tx) * 3 + k], rv[k]);
            if (tx == 0) for (int c = 0; c < 3; c++) pt_store(&myHalo[2 * PT_W * 3 + ty * 3 + c], rv[c]);
            if (4 * tx + 4 == TWv) for (int c = 0; c < 3; c++) pt_store(&myHalo[2 * PT_W * 3 + PT_MAXH * 3 + ty * 3 + c], rv[9 + c]);
        }
        pt_block_sum3(acc2, sm, nwaves);
        if (t == 0) { float *q = reinterpret_cast<float *>(&A.part_b[tile]); pt_store(q, acc2[0]); pt_store(q + 1, acc2[1]); pt_store(q + 2, acc2[2]); }
        ok = pt_barrier(A.bar, ++epoch, G);
        if (!ok) break;

        // ---- b = rz_new / rz ; p = r + p b, for the tile and (redundantly) its ring ----
        float rzn[3], b[3];
        pt_reduce_parts(A.part_b, G, rzn, sm, nwaves);
#pragma unroll
        for (int c = 0; c < 3; c++) { b[c] = rzn[c] / fmaxf(rz[c], FLT_MIN); rz[c] = rzn[c]; }   // Backend.cpp:336
        if (valid) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) pv[3 * k + c] = rv[3 * k + c] + pv[3 * k + c] * b[c];  // Backend.cpp:344
        }
        for (int e = t; e < ringN; e += blockDim.x) {
            const int c = e % 3, q = e / 3;
            int nt, off, slot;          // neighbour tile, offset of the wanted float in ITS halo record, my LDS slot
            bool have;
            if (q < TWv) { have = tY > 0; nt = tile - A.tilesX; off = PT_W * 3 + q * 3 + c; slot = 0 * PT_RS + (q + 1) * 3 + c; }                                 // its bottom row
            else if (q < 2 * TWv) { have = Y0 + THv < H; nt = tile + A.tilesX; off = (q - TWv) * 3 + c; slot = (THv + 1) * PT_RS + (q - TWv + 1) * 3 + c; }        // its top row
            else if (q < 2 * TWv + THv) { have = tX > 0; nt = tile - 1; off = 2 * PT_W * 3 + PT_MAXH * 3 + (q - 2 * TWv) * 3 + c; slot = (q - 2 * TWv + 1) * PT_RS + c; }   // its right column
            else { have = X0 + TWv < W; nt = tile + 1; off = 2 * PT_W * 3 + (q - 2 * TWv - THv) * 3 + c; slot = (q - 2 * TWv - THv + 1) * PT_RS + (TWv + 1) * 3 + c; }      // its left column
            if (have) {
                const float rn = pt_load(&A.halo[(size_t)nt * PT_HALO + off]);
                lp[slot] = rn + lp[slot] * b[c];
            }
        }
    }

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

} // namespace gdpt
