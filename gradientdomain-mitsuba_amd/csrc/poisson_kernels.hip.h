// poisson_kernels.hip.h -- gfx950 kernels of the screened-Poisson IRLS/CG reconstruction.
//
// One hand-written CDNA4 kernel per `poisson::Backend` op of the reference
// (/root/reference/src/integrators/poisson_solver/Backend.cpp; the reference's own GPU code,
// BackendCUDA.cu, is compiled out and is not what this follows) plus the fusions the MI355X design
// adds.  Data layout is the reference's: images are row-major AoS RGB fp32 ("Vec3f"), b/e stack
// [alpha*T ; dx ; dy] as 3n elements, w2 stacks 3n scalars.  Nothing is re-laid-out, so the
// backend-op C-ABI is a drop-in for Backend::Vector contents.
//
// Arithmetic contract: compiled with -ffp-contract=off, every per-element expression is written in
// the reference's association order, so every op WITHOUT a reduction is bit-identical to the CPU
// restatement in oracle/.  Dot products use a fixed tree (lane butterfly -> 4 waves -> <=1024 block
// partials in index order), i.e. deterministic run to run but not the sequential fp32 order of
// Backend.cpp:224-236; tests state the tolerance.
//
// Two families:
//   kg_*  "generic": one lane per float, any W,H.
//   kf_*  "fast":    W % 4 == 0; 16-byte lane accesses; the 5-point stencil stages a 256 px x 4 row
//                    tile (+1 px ring) of p through LDS with fully coalesced dwordx4 loads, then each
//                    lane owns 4 px x RGB (12 outputs) and reads its neighbourhood with ds_read_b128.
//   CDNA4 notes: wave = 64 lanes, block = 4 waves (one per SIMD).  ROCm 7.2's
//   __builtin_amdgcn_wave_reduce_* exists for integer types only, so fp32 sums use the xor butterfly.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

namespace gdpt {

constexpr int BLK  = 256;    // threads per block: 4 waves
constexpr int MAXP = 1024;   // cap on reduction-producing grids == cap on block partials

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum a[0..2] over the block; every thread receives the totals.  sm: >= 16 floats of LDS.
__device__ __forceinline__ void block_sum3(float (&a)[3], float *sm)
{
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 3; c++) a[c] = wave_sum(a[c]);
    __syncthreads();
    if (ln == 0) { sm[wv * 4 + 0] = a[0]; sm[wv * 4 + 1] = a[1]; sm[wv * 4 + 2] = a[2]; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; c++) a[c] = ((sm[c] + sm[4 + c]) + sm[8 + c]) + sm[12 + c];
}

// Reduce G block partials (float4: rgb + pad) in a fixed order; every thread receives the totals.
__device__ __forceinline__ void reduce_parts(const float4 *__restrict__ part, int G, float (&v)[3], float *sm)
{
    v[0] = v[1] = v[2] = 0.0f;
    for (int i = threadIdx.x; i < G; i += BLK) {
        const float4 q = part[i];
        v[0] += q.x; v[1] += q.y; v[2] += q.z;
    }
    block_sum3(v, sm);
}

__device__ __forceinline__ float sel3(int c, float a0, float a1, float a2) { return c == 0 ? a0 : (c == 1 ? a1 : a2); }

// ------------------------------------------------------------------------------------------------
// Scalar plumbing shared by the CG kernels.  A kernel either reads a 3-float device scalar or first
// reduces the previous kernel's block partials (and block 0 publishes the total for later kernels).
// ------------------------------------------------------------------------------------------------
struct Scal3 { float v[3]; };

__device__ __forceinline__ void load3(const float *__restrict__ p, float (&v)[3]) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }
__device__ __forceinline__ void store3(float *p, const float (&v)[3])
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; }
}

// ================================================================================================
// generic family (one lane per float)
// ================================================================================================

// Backend::calc_Px, Backend.cpp:150-174.
__global__ __launch_bounds__(BLK) void kg_Px(float *__restrict__ Px, const float *__restrict__ x, int W, int H, float alpha)
{
    const int n3 = 3 * W * H;
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK) {
        const int i = f / 3, xx = i % W, yy = i / W;
        const float xi = x[f];
        Px[f] = xi * alpha;
        Px[n3 + f] = (xx != W - 1) ? x[f + 3] - xi : 0.0f;
        Px[2 * n3 + f] = (yy != H - 1) ? x[f + 3 * W] - xi : 0.0f;
    }
}

// e = b - P x : Solver.cpp:386-387 (calc_Px then calc_axpy with a = -1; (-1*v)+b == b-v exactly).
__global__ __launch_bounds__(BLK) void kg_residual(float *__restrict__ e, const float *__restrict__ b,
                                                   const float *__restrict__ x, int W, int H, float alpha)
{
    const int n3 = 3 * W * H;
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK) {
        const int i = f / 3, xx = i % W, yy = i / W;
        const float xi = x[f];
        const float g0 = xi * alpha;
        const float g1 = (xx != W - 1) ? x[f + 3] - xi : 0.0f;
        const float g2 = (yy != H - 1) ? x[f + 3 * W] - xi : 0.0f;
        e[f] = b[f] - g0;
        e[n3 + f] = b[n3 + f] - g1;
        e[2 * n3 + f] = b[2 * n3 + f] - g2;
    }
}

// Backend::calc_PTW2x, Backend.cpp:178-205.  Optionally also p = r and block partials of r.r
// (Solver.cpp:403-405 fused: calc_PTW2x, calc_xdoty(rz,r,r), copy(p,r)).
template <bool FUSE_RZ_P>
__global__ __launch_bounds__(BLK) void kg_PTW2x(float *__restrict__ out, float *__restrict__ pcopy, float4 *__restrict__ part,
                                                const float *__restrict__ w2, const float *__restrict__ e, int W, int H, float alpha)
{
    __shared__ float sm[16];
    const int n = W * H, n3 = 3 * n;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK) {
        const int i = f / 3, c = f - 3 * i, xx = i % W, yy = i / W;
        float v = w2[i] * e[f] * alpha;
        if (xx != 0)     v = v + w2[n + i - 1] * e[n3 + f - 3];
        if (xx != W - 1) v = v - w2[n + i] * e[n3 + f];
        if (yy != 0)     v = v + w2[2 * n + i - W] * e[2 * n3 + f - 3 * W];
        if (yy != H - 1) v = v - w2[2 * n + i] * e[2 * n3 + f];
        out[f] = v;
        if (FUSE_RZ_P) {
            pcopy[f] = v;
            const float s = v * v;
            acc[0] += (c == 0) ? s : 0.0f; acc[1] += (c == 1) ? s : 0.0f; acc[2] += (c == 2) ? s : 0.0f;
        }
    }
    if (FUSE_RZ_P) {
        block_sum3(acc, sm);
        if (threadIdx.x == 0) part[blockIdx.x] = make_float4(acc[0], acc[1], acc[2], 0.0f);
    }
}

// Backend::calc_Ax_xAx, Backend.cpp:209-242: Ap and block partials of p.Ap.
__global__ __launch_bounds__(BLK) void kg_Ax(float *__restrict__ Ax, float4 *__restrict__ part, const float *__restrict__ w2,
                                             const float *__restrict__ x, int W, int H, float alpha)
{
    __shared__ float sm[16];
    const int n = W * H, n3 = 3 * n;
    const float alphaSqr = alpha * alpha;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK) {
        const int i = f / 3, c = f - 3 * i, xx = i % W, yy = i / W;
        const float xi = x[f];
        float a = w2[i] * xi * alphaSqr;
        if (xx != 0)     a = a + w2[n + i - 1] * (xi - x[f - 3]);
        if (xx != W - 1) a = a + w2[n + i] * (xi - x[f + 3]);
        if (yy != 0)     a = a + w2[2 * n + i - W] * (xi - x[f - 3 * W]);
        if (yy != H - 1) a = a + w2[2 * n + i] * (xi - x[f + 3 * W]);
        Ax[f] = a;
        const float s = xi * a;
        acc[0] += (c == 0) ? s : 0.0f; acc[1] += (c == 1) ? s : 0.0f; acc[2] += (c == 2) ? s : 0.0f;
    }
    block_sum3(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = make_float4(acc[0], acc[1], acc[2], 0.0f);
}

// Backend::calc_axpy, Backend.cpp:246-262.
__global__ __launch_bounds__(BLK) void kg_axpy(float *out, float a0, float a1, float a2, const float *x, const float *y, int n3)
{
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK)
        out[f] = sel3(f % 3, a0, a1, a2) * x[f] + y[f];
}

// Backend::calc_xdoty, Backend.cpp:266-283: block partials.
__global__ __launch_bounds__(BLK) void kg_xdoty(float4 *__restrict__ part, const float *__restrict__ x, const float *__restrict__ y, int n3)
{
    __shared__ float sm[16];
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK) {
        const int c = f % 3;
        const float s = x[f] * y[f];
        acc[0] += (c == 0) ? s : 0.0f; acc[1] += (c == 1) ? s : 0.0f; acc[2] += (c == 2) ? s : 0.0f;
    }
    block_sum3(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = make_float4(acc[0], acc[1], acc[2], 0.0f);
}

// Backend::calc_MIx, Backend.cpp:387-438: z = K' inv(D) K x with K = I - L inv(D), the incomplete-Poisson preconditioner of
// Ament et al. (cited there).  The reference makes two passes with two n-element temporaries (t = x - Ux/D, DIt = t/D; then
// MIx = t - L*DIt); DIt of the left and upper neighbour is all the second pass needs, so one pass recomputes those two
// values (same expression, same bits) instead of storing them.
__device__ __forceinline__ void mi_t(const float *__restrict__ w2, const float *__restrict__ x, int W, int H, int n, float alphaSqr,
                                     int xx, int yy, int i, int c, float &t, float &DIt)
{
    float Di = w2[i] * alphaSqr;
    float Uxi = 0.0f;
    if (xx != 0)     Di = Di + w2[n + i - 1];
    if (xx != W - 1) { Di = Di + w2[n + i]; Uxi = Uxi - w2[n + i] * x[3 * (i + 1) + c]; }
    if (yy != 0)     Di = Di + w2[2 * n + i - W];
    if (yy != H - 1) { Di = Di + w2[2 * n + i]; Uxi = Uxi - w2[2 * n + i] * x[3 * (i + W) + c]; }
    t = x[3 * i + c] - Uxi / Di;
    DIt = t / Di;
}

__global__ __launch_bounds__(BLK) void kg_MIx(float *__restrict__ MIx, const float *__restrict__ w2, const float *__restrict__ x, int W, int H, float alpha)
{
    const int n = W * H, n3 = 3 * n;
    const float alphaSqr = alpha * alpha;
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK) {
        const int i = f / 3, c = f - 3 * i, yy = i / W, xx = i - yy * W;
        float t, d, tn, dn;
        mi_t(w2, x, W, H, n, alphaSqr, xx, yy, i, c, t, d);
        float L = 0.0f;
        if (xx != 0) { mi_t(w2, x, W, H, n, alphaSqr, xx - 1, yy, i - 1, c, tn, dn); L = L - w2[n + i - 1] * dn; }
        if (yy != 0) { mi_t(w2, x, W, H, n, alphaSqr, xx, yy - 1, i - W, c, tn, dn); L = L - w2[2 * n + i - W] * dn; }
        MIx[f] = t - L;
    }
}

// Fixed-order total of block partials into a 3-float device scalar (and an optional second copy).
__global__ __launch_bounds__(BLK) void k_finalize(float *__restrict__ out, float *__restrict__ out2, const float4 *__restrict__ part, int G)
{
    __shared__ float sm[16];
    float v[3];
    reduce_parts(part, G, v, sm);
    if (threadIdx.x == 0) {
        out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
        if (out2) { out2[0] = v[0]; out2[1] = v[1]; out2[2] = v[2]; }
    }
}

// Backend::calc_r_rz, Backend.cpp:287-315.  pAp comes either from s_pAp or from reducing part_pAp
// (then published to s_pAp_out).  rz2 ("old" r.z) is read from s_rz2 and re-published to s_rz_old.
__global__ __launch_bounds__(BLK) void kg_r_rz(float *__restrict__ r, float4 *__restrict__ part_rz, const float *__restrict__ Ap,
                                               const float *__restrict__ s_rz2, const float *__restrict__ s_pAp,
                                               const float4 *__restrict__ part_pAp, int G_in, float *s_pAp_out, float *s_rz_old_out, int n3)
{
    __shared__ float sm[16];
    float pAp[3], rz2[3], a[3];
    if (part_pAp) { reduce_parts(part_pAp, G_in, pAp, sm); if (s_pAp_out) store3(s_pAp_out, pAp); }
    else load3(s_pAp, pAp);
    load3(s_rz2, rz2);
    if (s_rz_old_out) store3(s_rz_old_out, rz2);
#pragma unroll
    for (int c = 0; c < 3; c++) a[c] = rz2[c] / fmaxf(pAp[c], FLT_MIN);
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK) {
        const int c = f % 3;
        const float ri = r[f] - Ap[f] * sel3(c, a[0], a[1], a[2]);
        r[f] = ri;
        const float s = ri * ri;
        acc[0] += (c == 0) ? s : 0.0f; acc[1] += (c == 1) ? s : 0.0f; acc[2] += (c == 2) ? s : 0.0f;
    }
    block_sum3(acc, sm);
    if (threadIdx.x == 0) part_rz[blockIdx.x] = make_float4(acc[0], acc[1], acc[2], 0.0f);
}

// Backend::calc_x_p, Backend.cpp:319-350.  rz ("new") comes either from s_rz or from reducing part_rz
// (then published to s_rz_out).
__global__ __launch_bounds__(BLK) void kg_x_p(float *__restrict__ x, float *__restrict__ p, const float *__restrict__ r,
                                              const float *__restrict__ s_rz, const float *__restrict__ s_rz2, const float *__restrict__ s_pAp,
                                              const float4 *__restrict__ part_rz, int G_in, float *s_rz_out, int n3)
{
    __shared__ float sm[16];
    float rz[3], rz2[3], pAp[3], a[3], b[3];
    if (part_rz) { reduce_parts(part_rz, G_in, rz, sm); if (s_rz_out) store3(s_rz_out, rz); }
    else load3(s_rz, rz);
    load3(s_rz2, rz2);
    load3(s_pAp, pAp);
#pragma unroll
    for (int c = 0; c < 3; c++) { a[c] = rz2[c] / fmaxf(pAp[c], FLT_MIN); b[c] = rz[c] / fmaxf(rz2[c], FLT_MIN); }
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK) {
        const int c = f % 3;
        const float pi = p[f];
        x[f] = x[f] + pi * sel3(c, a[0], a[1], a[2]);
        p[f] = r[f] + pi * sel3(c, b[0], b[1], b[2]);
    }
}

// Backend::calc_w2 first loop, Backend.cpp:362-368: raw weights and block partial sums (in .x).
// Inside the solver reg = regtab[*counter] is read from device memory (Solver.cpp:395's value, computed on
// the host at setup), so ONE captured graph serves every IRLS iteration; the op-level ABI passes reg_imm.
__global__ __launch_bounds__(BLK) void kg_w2_raw(float *__restrict__ w2, float4 *__restrict__ part, const float *__restrict__ e,
                                                 const float *__restrict__ regtab, const int *__restrict__ counter, float reg_imm, int numElems)
{
    __shared__ float sm[16];
    const float reg = regtab ? regtab[counter[0]] : reg_imm;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int i = blockIdx.x * BLK + threadIdx.x; i < numElems; i += gridDim.x * BLK) {
        const float ex = e[3 * i], ey = e[3 * i + 1], ez = e[3 * i + 2];
        const float len = sqrtf(ex * ex + ey * ey + ez * ez);
        const float wi = 1.0f / (len + reg);
        w2[i] = wi;
        acc[0] += wi;
    }
    block_sum3(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = make_float4(acc[0], 0.0f, 0.0f, 0.0f);
}

// Backend::calc_w2 second loop, Backend.cpp:370-372: coef = numElems / sum, w2 *= coef.  Also advances the
// solver's IRLS counter (no block of THIS kernel reads it).
__global__ __launch_bounds__(BLK) void kg_w2_scale(float *__restrict__ w2, const float4 *__restrict__ part, int G_in, int *counter, int numElems)
{
    __shared__ float sm[16];
    float s[3];
    reduce_parts(part, G_in, s, sm);
    const float coef = (float)numElems / s[0];
    for (int i = blockIdx.x * BLK + threadIdx.x; i < numElems; i += gridDim.x * BLK) w2[i] = w2[i] * coef;
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) counter[0] = counter[0] + 1;
}

// Backend::set, Backend.cpp:104-115.
__global__ __launch_bounds__(BLK) void kg_set(float *x, float y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) x[i] = y;
}

// Solver::setupBackend, Solver.cpp:323-337: b = [alpha*T ; dx ; dy], x0 = T (or 0).
__global__ __launch_bounds__(BLK) void kg_setup(float *__restrict__ b, float *__restrict__ x, const float *__restrict__ dx,
                                                const float *__restrict__ dy, const float *__restrict__ tp, float alpha, int n3)
{
    for (int f = blockIdx.x * BLK + threadIdx.x; f < n3; f += gridDim.x * BLK) {
        const float t = tp ? tp[f] : 0.0f;
        b[f] = tp ? t * alpha : 0.0f;
        b[n3 + f] = dx[f];
        b[2 * n3 + f] = dy[f];
        x[f] = t;
    }
}

// ================================================================================================
// fast family (W % 4 == 0, 16-byte aligned bases)
// ================================================================================================
//
// Flat elementwise kernels: the image is a flat array of n3/4 float4.  A block-iteration covers
// 768 consecutive float4 (= 1024 px): lane t touches float4 T+t, T+256+t, T+512+t, so every wave
// instruction moves one contiguous KiB.  Because T % 3 == 0 and 256 % 3 == 1, float4 number j of lane
// t starts at colour (t + j) % 3: in the lane's ROTATED colour frame (rot = t % 3) the colour of
// component k of load j is the compile-time constant (j + k) % 3, so no per-element selects.

// streaming stores (and loads): the CG vectors of an HBM-resident image are written once and read by the NEXT kernel, 100 MB later -- kept out of
// L2 / MALL they do not evict the rows the neighbouring tile is about to re-read (kf_xp_Ax at 3840x2160: 140 -> 117 us)
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float4 *dst, float4 v)
{
    v4f_nt w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<v4f_nt *>(dst));
}
__device__ __forceinline__ float4 nt_load4(const float4 *src)
{
    const v4f_nt w = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt *>(src));
    return make_float4(w.x, w.y, w.z, w.w);
}
constexpr int FLAT_TILE = 3 * BLK; // float4 per block-iteration

__device__ __forceinline__ void rotate3(const float (&a)[3], int rot, float (&aR)[3])
{
    aR[0] = sel3(rot, a[0], a[1], a[2]);
    aR[1] = sel3(rot, a[1], a[2], a[0]);
    aR[2] = sel3(rot, a[2], a[0], a[1]);
}
__device__ __forceinline__ void unrotate3(const float (&aR)[3], int rot, float (&a)[3])
{
    a[0] = sel3(rot, aR[0], aR[2], aR[1]);
    a[1] = sel3(rot, aR[1], aR[0], aR[2]);
    a[2] = sel3(rot, aR[2], aR[1], aR[0]);
}

__global__ __launch_bounds__(BLK) void kf_r_rz(float4 *__restrict__ r, float4 *__restrict__ part_rz, const float4 *__restrict__ Ap,
                                               const float *__restrict__ s_rz2, const float *__restrict__ s_pAp,
                                               const float4 *__restrict__ part_pAp, int G_in, float *s_pAp_out, float *s_rz_old_out, int total4)
{
    __shared__ float sm[16];
    const int t = threadIdx.x, rot = t % 3;
    // issue the first tile's loads before the scalar reduction so their latency overlaps it
    int T = blockIdx.x * FLAT_TILE;
    float4 rv[3], av[3];
    bool ok[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int q = T + t + BLK * j;
        ok[j] = q < total4;
        if (ok[j]) { rv[j] = r[q]; av[j] = Ap[q]; }
    }
    float pAp[3], rz2[3], a[3], aR[3];
    if (part_pAp) { reduce_parts(part_pAp, G_in, pAp, sm); if (s_pAp_out) store3(s_pAp_out, pAp); }
    else load3(s_pAp, pAp);
    load3(s_rz2, rz2);
    if (s_rz_old_out) store3(s_rz_old_out, rz2);
#pragma unroll
    for (int c = 0; c < 3; c++) a[c] = rz2[c] / fmaxf(pAp[c], FLT_MIN);
    rotate3(a, rot, aR);
    float accR[3] = {0.0f, 0.0f, 0.0f};
    while (T < total4) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (!ok[j]) continue;
            float4 o;
            o.x = rv[j].x - av[j].x * aR[(j + 0) % 3];
            o.y = rv[j].y - av[j].y * aR[(j + 1) % 3];
            o.z = rv[j].z - av[j].z * aR[(j + 2) % 3];
            o.w = rv[j].w - av[j].w * aR[(j + 3) % 3];
            r[T + t + BLK * j] = o;
            accR[(j + 0) % 3] += o.x * o.x;
            accR[(j + 1) % 3] += o.y * o.y;
            accR[(j + 2) % 3] += o.z * o.z;
            accR[(j + 3) % 3] += o.w * o.w;
        }
        T += gridDim.x * FLAT_TILE;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int q = T + t + BLK * j;
            ok[j] = q < total4;
            if (ok[j]) { rv[j] = r[q]; av[j] = Ap[q]; }
        }
    }
    float acc[3];
    unrotate3(accR, rot, acc);
    block_sum3(acc, sm);
    if (t == 0) part_rz[blockIdx.x] = make_float4(acc[0], acc[1], acc[2], 0.0f);
}

__global__ __launch_bounds__(BLK) void kf_x_p(float4 *__restrict__ x, float4 *__restrict__ p, const float4 *__restrict__ r,
                                              const float *__restrict__ s_rz, const float *__restrict__ s_rz2, const float *__restrict__ s_pAp,
                                              const float4 *__restrict__ part_rz, int G_in, float *s_rz_out, int total4)
{
    __shared__ float sm[16];
    const int t = threadIdx.x, rot = t % 3;
    float rz[3], rz2[3], pAp[3], a[3], b[3], aR[3], bR[3];
    if (part_rz) { reduce_parts(part_rz, G_in, rz, sm); if (s_rz_out) store3(s_rz_out, rz); }
    else load3(s_rz, rz);
    load3(s_rz2, rz2);
    load3(s_pAp, pAp);
#pragma unroll
    for (int c = 0; c < 3; c++) { a[c] = rz2[c] / fmaxf(pAp[c], FLT_MIN); b[c] = rz[c] / fmaxf(rz2[c], FLT_MIN); }
    rotate3(a, rot, aR);
    rotate3(b, rot, bR);
    for (int T = blockIdx.x * FLAT_TILE; T < total4; T += gridDim.x * FLAT_TILE) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int q = T + t + BLK * j;
            if (q >= total4) continue;
            const float4 pv = p[q], xv = x[q], rv = r[q];
            float4 xo, po;
            xo.x = xv.x + pv.x * aR[(j + 0) % 3]; po.x = rv.x + pv.x * bR[(j + 0) % 3];
            xo.y = xv.y + pv.y * aR[(j + 1) % 3]; po.y = rv.y + pv.y * bR[(j + 1) % 3];
            xo.z = xv.z + pv.z * aR[(j + 2) % 3]; po.z = rv.z + pv.z * bR[(j + 2) % 3];
            xo.w = xv.w + pv.w * aR[(j + 3) % 3]; po.w = rv.w + pv.w * bR[(j + 3) % 3];
            x[q] = xo;
            p[q] = po;
        }
    }
}

// ---- LDS-tiled 5-point stencil -----------------------------------------------------------------
// Tile = TW px x TH rows, staged with a 1-px ring.  LDS row = [4-float left pad | 3*TW floats | 4-float
// right pad]; interior starts 16-byte aligned, the ring pixel sits in the last/first 3 floats of the pads.
constexpr int TW = 256;               // px per tile row = 64 lanes x 4 px
constexpr int TH = 4;                 // tile rows = waves per block
constexpr int ROW4 = 3 * TW / 4;      // float4 per tile row (192)
constexpr int RS = 3 * TW + 8;        // LDS row stride in floats (776)
constexpr int SROWS = TH + 2;

struct Tile { int x0, y0; };

// Stencil of one lane's 4 px x RGB from the staged tile; returns p.Ap contributions in acc.
// Association order == Backend.cpp:228-233.
template <bool UNITW, bool NT = false>
__device__ __forceinline__ void stencil_lane(const float *__restrict__ tile, int row, int lane, int x0, int y0, int W, int H,
                                             const float *__restrict__ w2, float alphaSqr, float4 *__restrict__ Ax4, float (&acc)[3])
{
    const int x = x0 + 4 * lane, y = y0 + row;
    if (x >= W || y >= H) return;
    const int n = W * H, i = y * W + x;
    const float *c0 = tile + (row + 1) * RS + 4 + 12 * lane;
    float ce[12], up[12], dn[12], lf[3], rt[3];
    {
        const float4 *c4 = reinterpret_cast<const float4 *>(c0);
        const float4 *u4 = reinterpret_cast<const float4 *>(c0 - RS);
        const float4 *d4 = reinterpret_cast<const float4 *>(c0 + RS);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float4 a = c4[k], b = u4[k], c = d4[k];
            ce[4 * k] = a.x; ce[4 * k + 1] = a.y; ce[4 * k + 2] = a.z; ce[4 * k + 3] = a.w;
            up[4 * k] = b.x; up[4 * k + 1] = b.y; up[4 * k + 2] = b.z; up[4 * k + 3] = b.w;
            dn[4 * k] = c.x; dn[4 * k + 1] = c.y; dn[4 * k + 2] = c.z; dn[4 * k + 3] = c.w;
        }
        const float4 l = c4[-1], r = c4[3];
        lf[0] = l.y; lf[1] = l.z; lf[2] = l.w;
        rt[0] = r.x; rt[1] = r.y; rt[2] = r.z;
    }
    float w0[4], w1[4], w1l, wv[4], wu[4];
    if (UNITW) {
#pragma unroll
        for (int k = 0; k < 4; k++) { w0[k] = w1[k] = wv[k] = wu[k] = 1.0f; }
        w1l = 1.0f;
    } else {
        const float4 a = *reinterpret_cast<const float4 *>(w2 + i);
        const float4 b = *reinterpret_cast<const float4 *>(w2 + n + i);
        const float4 c = *reinterpret_cast<const float4 *>(w2 + 2 * n + i);
        w0[0] = a.x; w0[1] = a.y; w0[2] = a.z; w0[3] = a.w;
        w1[0] = b.x; w1[1] = b.y; w1[2] = b.z; w1[3] = b.w;
        wv[0] = c.x; wv[1] = c.y; wv[2] = c.z; wv[3] = c.w;
        w1l = (x != 0) ? w2[n + i - 1] : 0.0f;
        if (y != 0) {
            const float4 d = *reinterpret_cast<const float4 *>(w2 + 2 * n + i - W);
            wu[0] = d.x; wu[1] = d.y; wu[2] = d.z; wu[3] = d.w;
        } else { wu[0] = wu[1] = wu[2] = wu[3] = 0.0f; }
    }
    float out[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xx = x + k;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float xi = ce[3 * k + c];
            const float left = (k == 0) ? lf[c] : ce[3 * (k - 1) + c];
            const float right = (k == 3) ? rt[c] : ce[3 * (k + 1) + c];
            const float wl = (k == 0) ? w1l : w1[k - 1];
            float a = w0[k] * xi * alphaSqr;
            if (xx != 0)     a = a + wl * (xi - left);
            if (xx != W - 1) a = a + w1[k] * (xi - right);
            if (y != 0)      a = a + wu[k] * (xi - up[3 * k + c]);
            if (y != H - 1)  a = a + wv[k] * (xi - dn[3 * k + c]);
            out[3 * k + c] = a;
            acc[c] += xi * a;
        }
    }
    float4 *o = Ax4 + (3 * (size_t)i) / 4;
    if (NT) {       // (kf_xp_Ax: Ap is read next by kf_r_rz, a whole image later)
        nt_store4(&o[0], make_float4(out[0], out[1], out[2], out[3]));
        nt_store4(&o[1], make_float4(out[4], out[5], out[6], out[7]));
        nt_store4(&o[2], make_float4(out[8], out[9], out[10], out[11]));
    } else {
        o[0] = make_float4(out[0], out[1], out[2], out[3]);
        o[1] = make_float4(out[4], out[5], out[6], out[7]);
        o[2] = make_float4(out[8], out[9], out[10], out[11]);
    }
}

// Backend::calc_Ax_xAx (Backend.cpp:209-242), LDS-tiled.
template <bool UNITW>
__global__ __launch_bounds__(BLK) void kf_Ax(float4 *__restrict__ Ax4, float4 *__restrict__ part, const float *__restrict__ w2,
                                             const float *__restrict__ p, int W, int H, float alpha, int tilesX, int tiles)
{
    __shared__ __attribute__((aligned(16))) float tile[SROWS * RS];
    __shared__ float sm[16];
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    const float alphaSqr = alpha * alpha;
    const int row4 = 3 * W / 4; // float4 per image row
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
        const int x0 = (tl % tilesX) * TW, y0 = (tl / tilesX) * TH;
        const int q0 = 3 * x0 / 4;
        __syncthreads();
        for (int it = t; it < SROWS * ROW4; it += BLK) {
            const int rr = it / ROW4, q = it - rr * ROW4, y = y0 - 1 + rr;
            if (y >= 0 && y < H && q0 + q < row4)
                *reinterpret_cast<float4 *>(tile + rr * RS + 4 + 4 * q) = p4[(size_t)y * row4 + q0 + q];
        }
        if (t < SROWS * 6) {
            const int rr = t / 6, s = (t % 6) / 3, c = t % 3, y = y0 - 1 + rr;
            const int xs = s ? x0 + TW : x0 - 1;
            if (y >= 0 && y < H && xs >= 0 && xs < W)
                tile[rr * RS + (s ? 4 + 3 * TW : 1) + c] = p[3 * ((size_t)y * W + xs) + c];
        }
        __syncthreads();
        stencil_lane<UNITW>(tile, wv, ln, x0, y0, W, H, w2, alphaSqr, Ax4, acc);
    }
    block_sum3(acc, sm);
    if (t == 0) part[blockIdx.x] = make_float4(acc[0], acc[1], acc[2], 0.0f);
}

// calc_x_p of iteration k fused into calc_Ax_xAx of iteration k+1 (MI355X design; no reference
// counterpart).  Stages p_new = r + p_old*b for the tile AND its ring straight into LDS, so the
// stencil never re-reads p from HBM; interior rows also store p_new (to the OTHER p buffer: a
// neighbouring block may still be reading p_old for its ring) and update x += p_old*a.
// TH_ = tile rows (ring overhead (TH_+2)/TH_ on the r,p reads); the first tile's loads are issued BEFORE
// the block reduces the previous kernel's partials, so that fixed latency hides under the HBM/MALL fetch.
// MINB: resident blocks per CU the register allocator is asked to leave room for (1 = its own choice).  Round 5's occupancy sweep at 3840x2160 (DESIGN.md,
// "kf_xp_Ax: occupancy sweep"; GDPT_XPAX=rows,blocks selects a variant, tools/gpu_xpax_sweep.py times them).
template <bool UNITW, int TH_, int MINB = 1>
__global__ __launch_bounds__(BLK, MINB) void kf_xp_Ax(float4 *__restrict__ Ax4, float4 *__restrict__ part_pAp, const float *__restrict__ w2,
                                                float4 *__restrict__ x4, const float *__restrict__ p_old, float4 *__restrict__ p_new4,
                                                const float *__restrict__ r, const float *__restrict__ s_rz2, const float *__restrict__ s_pAp,
                                                const float4 *__restrict__ part_rz, int G_in, float *s_rz_out,
                                                int W, int H, float alpha, int tilesX, int tiles)
{
    constexpr int SR = TH_ + 2, ITEMS = SR * ROW4, M = (ITEMS + BLK - 1) / BLK;
    __shared__ __attribute__((aligned(16))) float tile[SR * RS];
    __shared__ float sm[16];
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, rot = t % 3;
    const float alphaSqr = alpha * alpha;
    const int row4 = 3 * W / 4;
    const float4 *po4 = reinterpret_cast<const float4 *>(p_old);
    const float4 *r4 = reinterpret_cast<const float4 *>(r);
    float a[3], b[3], aR[3], bR[3];
    float acc[3] = {0.0f, 0.0f, 0.0f};
    bool first = true;
    for (int tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
        const int x0 = (tl % tilesX) * TW, y0 = (tl / tilesX) * TH_;
        const int q0 = 3 * x0 / 4;
        // ---- issue every load of this tile ----
        float4 pv[M], rv[M], xv[M];
        int gi[M];          // float4 index in the image, -1 = nothing to do
        bool own[M];
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int it = t + BLK * m;
            const int rr = it / ROW4, q = it - rr * ROW4, y = y0 - 1 + rr;
            gi[m] = (it < ITEMS && y >= 0 && y < H && q0 + q < row4) ? y * row4 + q0 + q : -1;
            own[m] = rr >= 1 && rr <= TH_;
            if (gi[m] >= 0) {
                pv[m] = po4[gi[m]];
                rv[m] = r4[gi[m]];
                if (own[m]) xv[m] = x4[gi[m]];
            }
        }
        float hr = 0.0f, hp = 0.0f;
        int hslot = -1, hc = 0;
        if (t < SR * 6) {
            const int rr = t / 6, sd = (t % 6) / 3, c = t % 3, y = y0 - 1 + rr;
            const int xs = sd ? x0 + TW : x0 - 1;
            if (y >= 0 && y < H && xs >= 0 && xs < W) {
                const size_t g = 3 * ((size_t)y * W + xs) + c;
                hr = r[g]; hp = p_old[g];
                hslot = rr * RS + (sd ? 4 + 3 * TW : 1) + c;
                hc = c;
            }
        }
        if (first) {        // scalars of this iteration: rz (new) from the previous kernel's block partials
            float rz[3], rz2[3], pAp[3];
            reduce_parts(part_rz, G_in, rz, sm);
            store3(s_rz_out, rz);
            load3(s_rz2, rz2);
            load3(s_pAp, pAp);
#pragma unroll
            for (int c = 0; c < 3; c++) { a[c] = rz2[c] / fmaxf(pAp[c], FLT_MIN); b[c] = rz[c] / fmaxf(rz2[c], FLT_MIN); }
            rotate3(a, rot, aR);
            rotate3(b, rot, bR);
            first = false;
        }
        __syncthreads();    // the previous tile's stencil reads are done
        // it = t + 256*m; ROW4 % 3 == 0 and 256 % 3 == 1  =>  colour of component k is (rot + m + k) % 3
#pragma unroll
        for (int m = 0; m < M; m++) {
            if (gi[m] < 0) continue;
            const int it = t + BLK * m;
            const int rr = it / ROW4, q = it - rr * ROW4;
            float4 pn;
            pn.x = rv[m].x + pv[m].x * bR[(m + 0) % 3];
            pn.y = rv[m].y + pv[m].y * bR[(m + 1) % 3];
            pn.z = rv[m].z + pv[m].z * bR[(m + 2) % 3];
            pn.w = rv[m].w + pv[m].w * bR[(m + 3) % 3];
            *reinterpret_cast<float4 *>(tile + rr * RS + 4 + 4 * q) = pn;
            if (own[m]) {
                nt_store4(&p_new4[gi[m]], pn);
                float4 xo;
                xo.x = xv[m].x + pv[m].x * aR[(m + 0) % 3];
                xo.y = xv[m].y + pv[m].y * aR[(m + 1) % 3];
                xo.z = xv[m].z + pv[m].z * aR[(m + 2) % 3];
                xo.w = xv[m].w + pv[m].w * aR[(m + 3) % 3];
                nt_store4(&x4[gi[m]], xo);
            }
        }
        if (hslot >= 0) tile[hslot] = hr + hp * sel3(hc, b[0], b[1], b[2]);
        __syncthreads();
#pragma unroll
        for (int row = 0; row < TH_; row += BLK / 64)
            stencil_lane<UNITW, true>(tile, row + wv, ln, x0, y0, W, H, w2, alphaSqr, Ax4, acc);
    }
    block_sum3(acc, sm);
    if (t == 0) part_pAp[blockIdx.x] = make_float4(acc[0], acc[1], acc[2], 0.0f);
}

// The yardstick of kf_xp_Ax's HBM fraction (gdpt_poisson_profile_stream): its access mix and nothing else -- per float4 element three coalesced loads from three
// arrays, three non-temporal stores to three others.
__global__ __launch_bounds__(BLK) void kg_stream33(const float4 *__restrict__ a, const float4 *__restrict__ b, const float4 *__restrict__ c,
                                                   float4 *__restrict__ o0, float4 *__restrict__ o1, float4 *__restrict__ o2, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) {
        const float4 va = a[i], vb = b[i], vc = c[i];
        float4 v;
        v.x = va.x + vb.x + vc.x; v.y = va.y + vb.y + vc.y; v.z = va.z + vb.z + vc.z; v.w = va.w + vb.w + vc.w;
        nt_store4(&o0[i], v);
        v.x += 1.0f; nt_store4(&o1[i], v);
        v.x += 1.0f; nt_store4(&o2[i], v);
    }
}

// ---- Backend::tonemapSRGB / tonemapLinear (Backend.cpp:442-507; the CUDA backend's kernels: BackendCUDA.cu:564-660): the display path of
// poisson::Solver (an ABGR_8888 image of the iterate after every solve, Solver.cpp:506) ------------------------------------------------
__device__ __forceinline__ unsigned pack_abgr(float x, float y, float z)
{
    return 0xFF000000u | ((unsigned)(int)fminf(fmaxf(x * 255.0f + 0.5f, 0.0f), 255.0f) << 0) | ((unsigned)(int)fminf(fmaxf(y * 255.0f + 0.5f, 0.0f), 255.0f) << 8) |
           ((unsigned)(int)fminf(fmaxf(z * 255.0f + 0.5f, 0.0f), 255.0f) << 16);
}
__global__ void kg_tonemap_srgb(unsigned *__restrict__ out, const float *__restrict__ in, int numPixels, float scale, float bias)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numPixels) return;
    float c[3];
    for (int k = 0; k < 3; k++) {
        float t = in[3 * (size_t)i + k];
        t = t * scale + bias;
        c[k] = (t <= 0.0031308f) ? 12.92f * t : 1.055f * powf(t, 1.0f / 2.4f) - 0.055f;        // linear to sRGB
    }
    out[i] = pack_abgr(c[0], c[1], c[2]);
}
// pass A: min / max over all components (order-preserving integer keys, integer atomics); pass B: scale, bias, |.|, pack
__global__ void kg_tonemap_minmax(unsigned *__restrict__ minmax, const float *__restrict__ in, int total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
    if (i < total) lo = hi = in[i];
    unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
    a ^= (unsigned)(((int)a >> 31) | 0x80000000u); b ^= (unsigned)(((int)b >> 31) | 0x80000000u);
    a = __builtin_amdgcn_wave_reduce_min_u32(a, 0); b = __builtin_amdgcn_wave_reduce_max_u32(b, 0);
    if ((threadIdx.x & 63) == 0) { atomicMin(&minmax[0], a); atomicMax(&minmax[1], b); }
}
__global__ void kg_tonemap_linear(unsigned *__restrict__ out, const float *__restrict__ in, const unsigned *__restrict__ minmax, int numPixels, int numComponents, float scaleMin, float scaleMax, int hasNegative)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numPixels) return;
    unsigned a = minmax[0], b = minmax[1];
    a ^= (a & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu; b ^= (b & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu;
    const float inMin = __uint_as_float(a), inMax = __uint_as_float(b);
    const float FMIN = 1.175494351e-38f;
    const float scale = fminf(fmaxf(hasNegative ? 0.5f / fmaxf(fmaxf(-inMin, inMax), FMIN) : 1.0f / fmaxf(inMax, FMIN), scaleMin), scaleMax);
    const float bias = hasNegative ? 0.5f : 0.0f;
    float c[3] = {0.0f, 0.0f, 0.0f};
    for (int k = 0; k < 3; k++) c[k] = (k < numComponents) ? fabsf(in[(size_t)i * numComponents + k] * scale + bias) : c[k - 1];
    out[i] = pack_abgr(c[0], c[1], c[2]);
}

} // namespace gdpt
