// poisson_capi.hip -- C-ABI (include/gdpt_poisson.h) over the gfx950 Poisson kernels.
//
// Solver level mirrors poisson::Solver (reference Solver.cpp:196-582); backend-op level mirrors the
// poisson::Backend virtuals (Backend.hpp:66-100).  Host code here is launch logic only: every byte of
// image arithmetic happens in poisson_kernels.hip.h on the device.  There is no CPU fallback.
#include "../../include/gdpt_poisson.h"
#include "poisson_kernels.hip.h"
#include "poisson_persistent.hip.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <list>
#include <vector>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

using namespace gdpt;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(GDPT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline int imin(int a, int b) { return a < b ? a : b; }
inline bool aligned16(const void *p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

int grid_generic(long n) { return imin(cdiv(n, BLK), 2 * MAXP); }
int grid_reduce(long n) { return imin(cdiv(n, BLK), MAXP); }
int grid_flat(long total4) { return imin(cdiv(total4, FLAT_TILE), MAXP); }

int ensure_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(GDPT_ERR_NO_DEVICE, "no HIP device visible: the gfx950 Poisson backend has no CPU fallback");
    if (device >= 0) {
        if (device >= count) return fail(GDPT_ERR_INVALID, "device %d out of range (%d visible)", device, count);
        HIPCHK(hipSetDevice(device));
    }
    return GDPT_OK;
}

// ---- op launchers shared by both ABI levels -----------------------------------------------------

constexpr int TH_F = 8;   // tile rows of kf_xp_Ax (ring overhead 10/8 on the r,p reads; 31 KB LDS)

struct Lattice {
    int W, H;
    float alpha;
    bool fast() const { return W % 4 == 0; }
    long n() const { return (long)W * H; }
    int tilesX() const { return cdiv(W, TW); }
    int tiles() const { return tilesX() * cdiv(H, TH); }
    int tilesF(int th = TH_F) const { return tilesX() * cdiv(H, th); }     // tiles of the fused x_p+stencil kernel (th rows each)
};

// Ap = A p and block partials of p.Ap; returns the number of partials written.
int launch_Ax(hipStream_t st, const Lattice &L, bool unitw, float *Ap, float4 *part, const float *w2, const float *p)
{
    if (L.fast() && aligned16(Ap) && aligned16(p) && aligned16(w2)) {
        const int G = imin(L.tiles(), MAXP);
        if (unitw) hipLaunchKernelGGL(kf_Ax<true>, dim3(G), dim3(BLK), 0, st, (float4 *)Ap, part, w2, p, L.W, L.H, L.alpha, L.tilesX(), L.tiles());
        else       hipLaunchKernelGGL(kf_Ax<false>, dim3(G), dim3(BLK), 0, st, (float4 *)Ap, part, w2, p, L.W, L.H, L.alpha, L.tilesX(), L.tiles());
        return G;
    }
    const int G = grid_reduce(3 * L.n());
    hipLaunchKernelGGL(kg_Ax, dim3(G), dim3(BLK), 0, st, Ap, part, w2, p, L.W, L.H, L.alpha);
    return G;
}

int launch_r_rz(hipStream_t st, long n3, float *r, float4 *part_rz, const float *Ap, const float *s_rz2, const float *s_pAp,
                const float4 *part_pAp, int G_in, float *s_pAp_out, float *s_rz_old_out)
{
    if (n3 % 4 == 0 && aligned16(r) && aligned16(Ap)) {
        const int G = grid_flat(n3 / 4);
        hipLaunchKernelGGL(kf_r_rz, dim3(G), dim3(BLK), 0, st, (float4 *)r, part_rz, (const float4 *)Ap, s_rz2, s_pAp, part_pAp, G_in, s_pAp_out, s_rz_old_out, (int)(n3 / 4));
        return G;
    }
    const int G = grid_reduce(n3);
    hipLaunchKernelGGL(kg_r_rz, dim3(G), dim3(BLK), 0, st, r, part_rz, Ap, s_rz2, s_pAp, part_pAp, G_in, s_pAp_out, s_rz_old_out, (int)n3);
    return G;
}

void launch_x_p(hipStream_t st, long n3, float *x, float *p, const float *r, const float *s_rz, const float *s_rz2, const float *s_pAp,
                const float4 *part_rz, int G_in, float *s_rz_out)
{
    if (n3 % 4 == 0 && aligned16(x) && aligned16(p) && aligned16(r)) {
        hipLaunchKernelGGL(kf_x_p, dim3(grid_flat(n3 / 4)), dim3(BLK), 0, st, (float4 *)x, (float4 *)p, (const float4 *)r, s_rz, s_rz2, s_pAp, part_rz, G_in, s_rz_out, (int)(n3 / 4));
        return;
    }
    hipLaunchKernelGGL(kg_x_p, dim3(grid_generic(n3)), dim3(BLK), 0, st, x, p, r, s_rz, s_rz2, s_pAp, part_rz, G_in, s_rz_out, (int)n3);
}

} // namespace

// =================================================================================================
// solver level
// =================================================================================================

struct gdpt_poisson_solver {
    gdpt_poisson_params P;
    gdpt_log_fn log_fn = nullptr;
    void *log_user = nullptr;

    int W = -1, H = -1;
    const float *in[4] = {nullptr, nullptr, nullptr, nullptr}; // dx, dy, tp, direct (borrowed)
    bool in_on_device = false;

    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ready = false;

    float *d_in[4] = {nullptr, nullptr, nullptr, nullptr}; // owned staging when inputs are host pointers
    float *b = nullptr, *e = nullptr, *w2 = nullptr, *x = nullptr, *r = nullptr, *p[2] = {nullptr, nullptr}, *Ap = nullptr, *rec = nullptr, *z = nullptr;
    float *x0 = nullptr;        // persistent CG: the iterate a solve started from, should it have to be redone on the multi-kernel path
    bool x0_valid = false;      // x0 holds the CURRENT x (taken by setup_backend; a second solve without setup retakes it)
    bool unchecked = false;     // a persistent solve was enqueued and its time-out flag has not been looked at yet
    float4 *part_pAp = nullptr, *part_rz = nullptr, *part_w = nullptr;
    float *scal = nullptr; // [0..3] pAp  [4..7] rz_old  [8..11] rz_next
    float *regtab = nullptr;
    int *counter = nullptr;
    float alpha_eff = 0.0f;
    const float *dev_direct = nullptr; // device pointer of `direct` (borrowed or staged), or null

    int fusion = 2;             // 0: reference op sequence; 1: x_p fused into the stencil; 2: persistent cooperative CG when the image fits, else 1; 3: 2 with the single-gather recurrence
    unsigned long long *halo = nullptr;   // persistent CG: per-tile boundary records (tagged floats)
    unsigned *bar = nullptr;    // persistent CG: [1] sticky error flag
    unsigned long long *gat = nullptr;  // persistent CG: tagged partial tables (gather A, gather B)
    unsigned ptLaunch = 0;      // persistent CG: launch number (upper half of every tag)
    bool ptWide = false;        // persistent CG: the 128-px wide tiles of kp_cg2
    int ptTilesX = 0, ptTilesY = 0, ptTH = 0;
    bool usedPersistent = false;
    hipGraphExec_t g0 = nullptr, gK = nullptr;
    int graph_fusion = -1;
    float graph_alpha = -1.0f;

    float last_seconds = 0.0f;
    long last_iters = 0;

    float *s_pAp() const { return scal; }
    float *s_rz_old() const { return scal + 4; }
    float *s_rz_next() const { return scal + 8; }
    Lattice lat() const { return Lattice{W, H, alpha_eff}; }

    void log(const char *fmt, ...)
    {
        if (!log_fn) return;
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        log_fn(buf, log_user);
    }

    void release_graphs()
    {
        if (g0) hipGraphExecDestroy(g0);
        if (gK) hipGraphExecDestroy(gK);
        g0 = gK = nullptr;
        graph_fusion = -1;
    }

    void release_buffers()
    {
        release_graphs();
        float **bufs[] = {&d_in[0], &d_in[1], &d_in[2], &d_in[3], &b, &e, &w2, &x, &r, &p[0], &p[1], &Ap, &rec, &x0, &scal, &regtab};
        for (float **q : bufs) { if (*q) hipFree(*q); *q = nullptr; }
        if (z) hipFree(z);
        z = nullptr;
        float4 **pb[] = {&part_pAp, &part_rz, &part_w};
        for (float4 **q : pb) { if (*q) hipFree(*q); *q = nullptr; }
        if (counter) hipFree(counter);
        counter = nullptr;
        if (halo) hipFree(halo);
        if (bar) hipFree(bar);
        if (gat) hipFree(gat);
        halo = nullptr; bar = nullptr; gat = nullptr; ptLaunch = 0;
        ready = false;
    }
};

namespace {

// The per-IRLS-iteration op sequence of Solver::solveIndirect (Solver.cpp:382-405) followed by `cg` CG
// iterations (Solver.cpp:464-470), enqueued on the solver's stream.  first == (irlsIter == 0).
void enqueue_irls_prologue(gdpt_poisson_solver *s, bool first)
{
    const Lattice L = s->lat();
    const long n = L.n(), n3 = 3 * n;
    hipStream_t st = s->stream;
    hipLaunchKernelGGL(kg_residual, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->e, s->b, s->x, L.W, L.H, L.alpha); // e = b - P x
    if (first) {
        hipLaunchKernelGGL(kg_set, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->w2, 1.0f, (size_t)n3);          // w2 = 1
    } else {
        const int G = grid_reduce(n3);
        hipLaunchKernelGGL(kg_w2_raw, dim3(G), dim3(BLK), 0, st, s->w2, s->part_w, s->e, (const float *)s->regtab, (const int *)s->counter, 0.0f, (int)n3);
        hipLaunchKernelGGL(kg_w2_scale, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->w2, s->part_w, G, s->counter, (int)n3);
    }
    const int G = grid_reduce(n3);
    hipLaunchKernelGGL(kg_PTW2x<true>, dim3(G), dim3(BLK), 0, st, s->r, s->p[0], s->part_rz, s->w2, s->e, L.W, L.H, L.alpha); // r, p = r, r.r
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLK), 0, st, s->s_rz_next(), (float *)nullptr, s->part_rz, G);
}

// `cg` iterations in the reference's 3-op form; p stays in p[0].
void enqueue_cg_unfused(gdpt_poisson_solver *s, bool unitw, int cg)
{
    const Lattice L = s->lat();
    const long n3 = 3 * L.n();
    hipStream_t st = s->stream;
    for (int k = 0; k < cg; k++) {
        const int Ga = launch_Ax(st, L, unitw, s->Ap, s->part_pAp, s->w2, s->p[0]);
        const int Gr = launch_r_rz(st, n3, s->r, s->part_rz, s->Ap, s->s_rz_next(), nullptr, s->part_pAp, Ga, s->s_pAp(), s->s_rz_old());
        launch_x_p(st, n3, s->x, s->p[0], s->r, nullptr, s->s_rz_old(), s->s_pAp(), s->part_rz, Gr, s->s_rz_next());
    }
}

// `cg` iterations of the preconditioned form, Solver.cpp:474-489 -- including the reference's choice of handing r (not z) to
// calc_x_p.  first == (cgIter == 0): z = inv(M) r ; rz = r.z ; p = z.
void enqueue_cg_precond(gdpt_poisson_solver *s, bool unitw, int cg, bool first)
{
    const Lattice L = s->lat();
    const long n3 = 3 * L.n();
    hipStream_t st = s->stream;
    const int Gx = grid_reduce(n3);
    if (first) {
        hipLaunchKernelGGL(kg_MIx, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->z, s->w2, s->r, L.W, L.H, L.alpha);
        hipLaunchKernelGGL(kg_xdoty, dim3(Gx), dim3(BLK), 0, st, s->part_rz, s->r, s->z, (int)n3);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLK), 0, st, s->s_rz_next(), (float *)nullptr, s->part_rz, Gx);
        hipMemcpyAsync(s->p[0], s->z, sizeof(float) * n3, hipMemcpyDeviceToDevice, st);
    }
    for (int k = 0; k < cg; k++) {
        const int Ga = launch_Ax(st, L, unitw, s->Ap, s->part_pAp, s->w2, s->p[0]);
        launch_r_rz(st, n3, s->r, s->part_rz, s->Ap, s->s_rz_next(), nullptr, s->part_pAp, Ga, s->s_pAp(), s->s_rz_old());   // its r.r is overwritten below
        hipLaunchKernelGGL(kg_MIx, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->z, s->w2, s->r, L.W, L.H, L.alpha);
        hipLaunchKernelGGL(kg_xdoty, dim3(Gx), dim3(BLK), 0, st, s->part_rz, s->r, s->z, (int)n3);
        launch_x_p(st, n3, s->x, s->p[0], s->r, nullptr, s->s_rz_old(), s->s_pAp(), s->part_rz, Gx, s->s_rz_next());
    }
}

// The fused x_p + stencil kernel in the variant GDPT_XPAX=rows,blocks names (tile rows 4 | 8 | 12, resident blocks per CU asked of the register allocator:
// 1 = its own choice); default = the product's 8 rows.  Returns the grid (= the number of p.Ap partials written).
struct XpAxVariant { int th, minb; };
static XpAxVariant xpax_variant()
{
    static XpAxVariant v = [] {
        XpAxVariant r = {TH_F, 1};
        if (const char *e = getenv("GDPT_XPAX")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2) { r.th = a; r.minb = b; } }
        return r;
    }();
    return v;
}
static int launch_xp_Ax(hipStream_t st, const Lattice &L, bool unitw, float *Ap, float4 *part_pAp, const float *w2, float *x, const float *po, float *pn, const float *r,
                        const float *s_rz2, const float *s_pAp, const float4 *part_rz, int G_in, float *s_rz_out)
{
    const XpAxVariant v = xpax_variant();
    const int tiles = L.tilesF(v.th), Gt = imin(tiles, MAXP);
#define XPAX(U, TH_V, MB) hipLaunchKernelGGL((kf_xp_Ax<U, TH_V, MB>), dim3(Gt), dim3(BLK), 0, st, (float4 *)Ap, part_pAp, w2, (float4 *)x, po, (float4 *)pn, r, s_rz2, s_pAp, part_rz, G_in, s_rz_out, L.W, L.H, L.alpha, L.tilesX(), tiles)
#define XPAX_U(TH_V, MB) do { if (unitw) XPAX(true, TH_V, MB); else XPAX(false, TH_V, MB); } while (0)
    if (v.th == 4 && v.minb == 1) XPAX_U(4, 1);
    else if (v.th == 4 && v.minb == 3) XPAX_U(4, 3);
    else if (v.th == 4 && v.minb == 4) XPAX_U(4, 4);
    else if (v.th == 8 && v.minb == 3) XPAX_U(8, 3);
    else if (v.th == 12 && v.minb == 1) XPAX_U(12, 1);
    else XPAX_U(TH_F, 1);
#undef XPAX_U
#undef XPAX
    return Gt;
}

// `cg` iterations with x_p(k) fused into the stencil of iteration k+1: 2*cg + 1 kernels.
void enqueue_cg_fused(gdpt_poisson_solver *s, bool unitw, int cg)
{
    const Lattice L = s->lat();
    const long n3 = 3 * L.n();
    hipStream_t st = s->stream;
    int Ga = launch_Ax(st, L, unitw, s->Ap, s->part_pAp, s->w2, s->p[0]);
    int Gr = launch_r_rz(st, n3, s->r, s->part_rz, s->Ap, s->s_rz_next(), nullptr, s->part_pAp, Ga, s->s_pAp(), s->s_rz_old());
    for (int k = 1; k < cg; k++) {
        float *po = s->p[(k - 1) & 1], *pn = s->p[k & 1];
        const int Gt = launch_xp_Ax(st, L, unitw, s->Ap, s->part_pAp, s->w2, s->x, po, pn, s->r, s->s_rz_old(), s->s_pAp(), s->part_rz, Gr, s->s_rz_next());
        Gr = launch_r_rz(st, n3, s->r, s->part_rz, s->Ap, s->s_rz_next(), nullptr, s->part_pAp, Gt, s->s_pAp(), s->s_rz_old());
    }
    launch_x_p(st, n3, s->x, s->p[(cg - 1) & 1], s->r, nullptr, s->s_rz_old(), s->s_pAp(), s->part_rz, Gr, s->s_rz_next());
}

bool can_fuse(const gdpt_poisson_solver *s) { return s->fusion >= 1 && s->W % 4 == 0; }

// Persistent CG geometry: 64-px wide tiles, one workgroup per CU, every tile resident.  Returns false when the
// image does not fit that scheme (then the multi-kernel graph path runs).
bool persistent_geometry(gdpt_poisson_solver *s)
{
    if (s->fusion < 2 || s->W % 4 != 0 || s->P.cgTolerance != 0.0f || s->P.verbose || s->P.cgPrecond || s->P.cgIterMax >= 0xffff) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return false;
    // 64-px wide tiles with 4 px per lane (kp_cg), else 128-px wide tiles with 8 px per lane and p in LDS only (kp_cg2: up to ~2 Mpixel)
    for (int wide = 0; wide < 2; wide++) {
        if (wide && getenv("GDPT_NO_WIDE_PERSISTENT")) break;
        const int tw = wide ? P2_W : PT_W;
        const int tilesX = cdiv(s->W, tw);
        const int maxTY = cus / tilesX;
        if (maxTY < 1) continue;
        int TH = cdiv(s->H, maxTY);
        if (TH < 8) TH = imin(8, s->H);
        if (TH > PT_MAXH) continue;
        s->ptTilesX = tilesX; s->ptTH = TH; s->ptTilesY = cdiv(s->H, TH); s->ptWide = wide != 0;
        if (s->ptTilesX * s->ptTilesY <= imin(cus, PT_MAXG)) return true;
    }
    return false;
}

constexpr int PT_LAUNCH_REFUSED = -1000;     // internal: hipLaunchCooperativeKernel failed (never crosses the ABI)
int enqueue_cg_persistent(gdpt_poisson_solver *s, bool unitw, int cg)
{
    PersistArgs A;
    A.x = s->x; A.r = s->r; A.p = s->p[0]; A.w2 = s->w2;
    A.gat = s->gat; A.halo = s->halo; A.bar = s->bar; A.s_rz = s->s_rz_next();
    A.W = s->W; A.H = s->H; A.tilesX = s->ptTilesX; A.tilesY = s->ptTilesY; A.TH = s->ptTH; A.iters = cg; A.alpha = s->alpha_eff;
    { const char *e = getenv("GDPT_DEBUG_PERSISTENT_FAIL"); A.debugFail = (e && e[0] == '1') ? 1 : ((e && e[0] == '2') ? 2 : 0); }   // test hooks: 1 = a gather times out, 2 = the launch is refused
    if (s->ptLaunch == 0 || s->ptLaunch == 0xffffu) {      // fresh tables, or the 16-bit launch number is about to wrap: forget all tags
        HIPCHK(hipMemsetAsync(s->gat, 0, sizeof(unsigned long long) * 12 * PT_MAXG, s->stream));
        HIPCHK(hipMemsetAsync(s->halo, 0, sizeof(unsigned long long) * (size_t)P2_HALO * PT_MAXG, s->stream));
        s->ptLaunch = 0;
    }
    A.tagBase = (++s->ptLaunch) << 16;
    void *args[] = {&A};
    const dim3 grid(s->ptTilesX * s->ptTilesY), block(((16 * s->ptTH + 63) / 64) * 64);
    const bool single = s->fusion >= 3 && !s->ptWide;          // the single-gather recurrence exists for the 64-px kernel
    const void *fn = s->ptWide ? (unitw ? (const void *)kp_cg2<true> : (const void *)kp_cg2<false>)
                   : single ? (unitw ? (const void *)kp_cg<true, true> : (const void *)kp_cg<false, true>) : (unitw ? (const void *)kp_cg<true> : (const void *)kp_cg<false>);
    const size_t shared = s->ptWide ? P2_SHARED_BYTES : 0;
    if (s->ptWide) {
        // 141 KB of dynamic LDS (P2_SHARED_BYTES = 144 144 B): above the 64 KB a kernel gets without asking.  The attribute is PER DEVICE, so the
        // "already raised" note is kept per (device, variant): a process that creates solvers on two GPUs raises it on both
        static std::mutex raisedMutex;
        static std::map<std::pair<int, int>, bool> raised;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(raisedMutex);
        bool &done = raised[std::make_pair(dev, unitw ? 1 : 0)];
        if (!done) { if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shared) != hipSuccess) { (void)hipGetLastError(); return PT_LAUNCH_REFUSED; } done = true; }
    }
    if (A.debugFail == 2 || hipLaunchCooperativeKernel(fn, grid, block, args, shared, s->stream) != hipSuccess) {
        (void)hipGetLastError();            // cooperative launch refused (unsupported, partitioned device, grid not co-resident): the caller falls back
        return PT_LAUNCH_REFUSED;
    }
    return GDPT_OK;
}

int capture(gdpt_poisson_solver *s, bool first, hipGraphExec_t *out)
{
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    enqueue_irls_prologue(s, first);
    if (can_fuse(s)) enqueue_cg_fused(s, first, s->P.cgIterMax);
    else enqueue_cg_unfused(s, first, s->P.cgIterMax);
    HIPCHK(hipStreamEndCapture(s->stream, &graph));
    hipError_t e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(GDPT_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return GDPT_OK;
}

} // namespace

extern "C" int gdpt_internal_fail(int code, const char *msg) { g_err = msg; return code; }

extern "C" {

const char *gdpt_last_error(void) { return g_err.c_str(); }

void gdpt_poisson_params_defaults(gdpt_poisson_params *p)
{
    p->alpha = 0.2f;
    p->device = -1;
    p->verbose = 0;
    gdpt_poisson_params_preset(p, "L1D");
}

int gdpt_poisson_params_preset(gdpt_poisson_params *p, const char *preset)
{
    p->irlsIterMax = 1; p->irlsRegInit = 0.0f; p->irlsRegIter = 0.0f;
    p->cgIterMax = 1; p->cgIterCheck = 100; p->cgPrecond = 0; p->cgTolerance = 0.0f;
    if (!preset) return 0;
    if (!strcmp(preset, "L1D")) { p->irlsIterMax = 20; p->irlsRegInit = 0.05f; p->irlsRegIter = 0.5f; p->cgIterMax = 50; return 1; }
    if (!strcmp(preset, "L1Q")) { p->irlsIterMax = 64; p->irlsRegInit = 1.0f; p->irlsRegIter = 0.7f; p->cgIterMax = 1000; return 1; }
    if (!strcmp(preset, "L1L")) { p->irlsIterMax = 7; p->irlsRegInit = 1.0e-4f; p->irlsRegIter = 1.0e-1f; p->cgIterMax = 20000; p->cgTolerance = 1.0e-20f; return 1; }
    if (!strcmp(preset, "L2D")) { p->cgIterMax = 50; return 1; }
    if (!strcmp(preset, "L2Q")) { p->cgIterMax = 500; return 1; }
    return 0;
}

int gdpt_poisson_create(const gdpt_poisson_params *p, gdpt_poisson_solver **out)
{
    if (!p || !out) return fail(GDPT_ERR_INVALID, "null argument");
    int rc = ensure_device(p->device);
    if (rc) return rc;
    gdpt_poisson_solver *s = new gdpt_poisson_solver;
    s->P = *p;
    // Params::sanitize, Solver.cpp:182-192
    s->P.alpha = fmaxf(s->P.alpha, 0.0f);
    if (s->P.irlsIterMax < 1) s->P.irlsIterMax = 1;
    s->P.irlsRegInit = fmaxf(s->P.irlsRegInit, 0.0f);
    s->P.irlsRegIter = fmaxf(s->P.irlsRegIter, 0.0f);
    if (s->P.cgIterMax < 1) s->P.cgIterMax = 1;
    if (s->P.cgIterCheck < 1) s->P.cgIterCheck = 1;
    s->P.cgTolerance = fmaxf(s->P.cgTolerance, 0.0f);
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) {
        delete s;
        return fail(GDPT_ERR_HIP, "stream/event creation failed");
    }
    *out = s;
    return GDPT_OK;
}

void gdpt_poisson_destroy(gdpt_poisson_solver *s)
{
    if (!s) return;
    if (s->stream) hipStreamSynchronize(s->stream);
    s->release_buffers();
    if (s->ev0) hipEventDestroy(s->ev0);
    if (s->ev1) hipEventDestroy(s->ev1);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

int gdpt_poisson_set_log(gdpt_poisson_solver *s, gdpt_log_fn fn, void *user)
{
    if (!s) return fail(GDPT_ERR_INVALID, "null solver");
    s->log_fn = fn;
    s->log_user = user;
    return GDPT_OK;
}

static int import_common(gdpt_poisson_solver *s, const float *dx, const float *dy, const float *tp, const float *direct, int w, int h, bool dev)
{
    if (!s) return fail(GDPT_ERR_INVALID, "null solver");
    if (!dx || !dy) return fail(GDPT_ERR_INVALID, "dx and dy are required (Solver.cpp:259 asserts them)");
    if (w <= 0 || h <= 0) return fail(GDPT_ERR_INVALID, "image size must be positive (Solver.cpp:260)");
    if ((long)w * h * 9 >= (1L << 31)) return fail(GDPT_ERR_INVALID, "image too large for 32-bit element indices");
    if (s->ready && (w != s->W || h != s->H)) s->release_buffers();
    s->W = w; s->H = h;
    s->in[0] = dx; s->in[1] = dy; s->in[2] = tp; s->in[3] = direct;
    s->in_on_device = dev;
    return GDPT_OK;
}

int gdpt_poisson_import_images(gdpt_poisson_solver *s, const float *dx, const float *dy, const float *tp, const float *direct, int w, int h)
{
    return import_common(s, dx, dy, tp, direct, w, h, false);
}

int gdpt_poisson_import_images_device(gdpt_poisson_solver *s, const float *dx, const float *dy, const float *tp, const float *direct, int w, int h)
{
    return import_common(s, dx, dy, tp, direct, w, h, true);
}

int gdpt_poisson_setup_backend(gdpt_poisson_solver *s)
{
    if (!s || !s->in[0] || !s->in[1] || s->W <= 0) return fail(GDPT_ERR_INVALID, "setup_backend before import_images");
    const long n = (long)s->W * s->H, n3 = 3 * n;
    const size_t B3 = sizeof(float) * n3;
    if (!s->ready) {
        s->log("Using HIP (gfx950) backend\n");
        HIPCHK(hipMalloc(&s->b, 3 * B3));
        HIPCHK(hipMalloc(&s->e, 3 * B3));
        HIPCHK(hipMalloc(&s->w2, B3));
        HIPCHK(hipMalloc(&s->x, B3));
        HIPCHK(hipMalloc(&s->r, B3));
        HIPCHK(hipMalloc(&s->p[0], B3));
        HIPCHK(hipMalloc(&s->p[1], B3));
        HIPCHK(hipMalloc(&s->Ap, B3));
        HIPCHK(hipMalloc(&s->rec, B3));
        if (s->P.cgPrecond) HIPCHK(hipMalloc(&s->z, B3));
        HIPCHK(hipMalloc(&s->part_pAp, sizeof(float4) * MAXP));
        HIPCHK(hipMalloc(&s->part_rz, sizeof(float4) * MAXP));
        HIPCHK(hipMalloc(&s->part_w, sizeof(float4) * MAXP));
        HIPCHK(hipMalloc(&s->scal, sizeof(float) * 16));
        HIPCHK(hipMalloc(&s->regtab, sizeof(float) * (s->P.irlsIterMax + 1)));
        HIPCHK(hipMalloc(&s->counter, sizeof(int) * 4));
        HIPCHK(hipMalloc(&s->halo, sizeof(unsigned long long) * (size_t)P2_HALO * PT_MAXG));      // (P2_HALO > PT_HALO: either kernel's records fit)
        HIPCHK(hipMalloc(&s->bar, sizeof(unsigned) * PT_BAR_WORDS));
        HIPCHK(hipMalloc(&s->gat, sizeof(unsigned long long) * 12 * PT_MAXG));      // two gather tables; four (two per iteration parity) at fusion level 3
        // reg_k = regInit * regIter^(k-1), Solver.cpp:395 (host powf like the reference)
        std::vector<float> reg(s->P.irlsIterMax + 1, 0.0f);
        for (int k = 1; k < s->P.irlsIterMax; k++) reg[k] = s->P.irlsRegInit * powf(s->P.irlsRegIter, (float)(k - 1));
        HIPCHK(hipMemcpy(s->regtab, reg.data(), sizeof(float) * reg.size(), hipMemcpyHostToDevice));
        s->ready = true;
    }
    const float *src[4];
    for (int k = 0; k < 4; k++) {
        src[k] = s->in[k];
        if (s->in[k] && !s->in_on_device) {
            if (!s->d_in[k]) HIPCHK(hipMalloc(&s->d_in[k], B3));
            HIPCHK(hipMemcpyAsync(s->d_in[k], s->in[k], B3, hipMemcpyHostToDevice, s->stream));
            src[k] = s->d_in[k];
        }
    }
    s->alpha_eff = src[2] ? s->P.alpha : 0.0f; // Solver.cpp:319
    hipLaunchKernelGGL(kg_setup, dim3(grid_generic(n3)), dim3(BLK), 0, s->stream, s->b, s->x, src[0], src[1], src[2], s->alpha_eff, (int)n3);
    s->dev_direct = src[3];
    HIPCHK(hipGetLastError());
    s->x0_valid = false;
    if (persistent_geometry(s)) {           // x0, should the cooperative launch have to be redone (gdpt_poisson_sync); its own buffer: export writes s->rec
        if (!s->x0) HIPCHK(hipMalloc(&s->x0, B3));
        HIPCHK(hipMemcpyAsync(s->x0, s->x, B3, hipMemcpyDeviceToDevice, s->stream));
        s->x0_valid = true;
    }
    // (re)capture the per-IRLS-iteration graphs when geometry/alpha/fusion changed
    if (s->graph_fusion != s->fusion || s->graph_alpha != s->alpha_eff) {
        s->release_graphs();
        if (s->P.cgTolerance == 0.0f && !s->P.verbose && !s->P.cgPrecond) {
            int rc = capture(s, true, &s->g0);
            if (rc) return rc;
            if (s->P.irlsIterMax > 1) { rc = capture(s, false, &s->gK); if (rc) return rc; }
        }
        s->graph_fusion = s->fusion;
        s->graph_alpha = s->alpha_eff;
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return GDPT_OK;
}

int gdpt_poisson_solve_indirect_async(gdpt_poisson_solver *s)
{
    if (!s || !s->ready) return fail(GDPT_ERR_INVALID, "solve_indirect before setup_backend");
    hipStream_t st = s->stream;
    HIPCHK(hipEventRecord(s->ev0, st));
    const int one = 1;
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)s->counter, one, 1, st));
    s->last_iters = 0;
    s->usedPersistent = false;
    bool persistent = persistent_geometry(s) && s->ptTilesX * s->ptTilesY <= MAXP && s->x0;
    if (persistent) {
        // fusion level 2: the CG loop of every IRLS iteration is one cooperative launch (poisson_persistent.hip.h)
        const size_t B3 = sizeof(float) * 3 * (size_t)s->W * s->H;
        if (!s->x0_valid) HIPCHK(hipMemcpyAsync(s->x0, s->x, B3, hipMemcpyDeviceToDevice, st));      // a further solve without setup_backend continues from the current x
        s->x0_valid = false;
        s->usedPersistent = true;
        s->unchecked = true;
        HIPCHK(hipMemsetAsync(s->bar, 0, sizeof(unsigned) * PT_BAR_WORDS, st));
        for (int irls = 0; irls < s->P.irlsIterMax; irls++) {
            enqueue_irls_prologue(s, irls == 0);
            int rc = enqueue_cg_persistent(s, irls == 0, s->P.cgIterMax);
            if (rc == PT_LAUNCH_REFUSED && s->g0) {
                // include/gdpt_poisson.h: level 2 falls back to level 1 when the persistent kernel cannot run.  Start over from x0 on the graphs.
                s->log("persistent CG: cooperative launch refused; falling back to the multi-kernel path\n");
                s->fusion = 1;
                s->usedPersistent = false; s->unchecked = false;
                HIPCHK(hipMemcpyAsync(s->x, s->x0, B3, hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemsetD32Async((hipDeviceptr_t)s->counter, one, 1, st));
                s->last_iters = 0;
                persistent = false;
                break;
            }
            if (rc == PT_LAUNCH_REFUSED) return fail(GDPT_ERR_HIP, "cooperative launch refused and no graph path was captured");
            if (rc) return rc;
            s->last_iters += s->P.cgIterMax;
        }
    }
    if (persistent) {
    } else if (s->g0) {
        // cgTolerance == 0: the convergence test of Solver.cpp:438 can only fire on r.z == 0 exactly, and CG
        // steps taken from that state leave x bit-identical (a = 0/FLT_MIN = 0), so no host round trip is needed.
        for (int irls = 0; irls < s->P.irlsIterMax; irls++) {
            HIPCHK(hipGraphLaunch(irls == 0 ? s->g0 : s->gK, st));
            s->last_iters += s->P.cgIterMax;
        }
    } else {
        // cgTolerance > 0, verbose or cgPrecond: reference control flow with the host read of the error every cgIterCheck iterations.
        const long n3 = 3L * s->W * s->H;
        for (int irls = 0; irls < s->P.irlsIterMax; irls++) {
            enqueue_irls_prologue(s, irls == 0);
            for (int cg = 0;;) {
                float rz[3];
                const float *src = s->s_rz_next();
                if (s->P.cgPrecond && cg != 0) {                                   // Solver.cpp:423-429: r.r, not r.z
                    const int G = grid_reduce(n3);
                    hipLaunchKernelGGL(kg_xdoty, dim3(G), dim3(BLK), 0, st, s->part_w, s->r, s->r, (int)n3);
                    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLK), 0, st, s->scal + 12, (float *)nullptr, s->part_w, G);
                    src = s->scal + 12;
                }
                HIPCHK(hipMemcpyAsync(rz, src, sizeof rz, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                const float errL2W = rz[0] + rz[1] + rz[2];
                if (s->P.verbose)
                    s->log("IRLS = %-3d/ %d, CG = %-4d/ %d, errL2W = %9.2e\n", irls, s->P.irlsIterMax, cg, s->P.cgIterMax, errL2W);
                if (cg == s->P.cgIterMax || errL2W <= s->P.cgTolerance) break;
                const int chunk = imin(s->P.cgIterCheck - cg % s->P.cgIterCheck, s->P.cgIterMax - cg);
                if (s->P.cgPrecond) enqueue_cg_precond(s, irls == 0, chunk, cg == 0);
                else enqueue_cg_unfused(s, irls == 0, chunk);
                cg += chunk;
                s->last_iters += chunk;
            }
        }
    }
    HIPCHK(hipEventRecord(s->ev1, st));
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_poisson_sync(gdpt_poisson_solver *s)
{
    if (!s) return fail(GDPT_ERR_INVALID, "null solver");
    HIPCHK(hipStreamSynchronize(s->stream));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->last_seconds = ms * 1.0e-3f;
    if (s->usedPersistent && s->unchecked) {
        s->unchecked = false;
        unsigned flag[2] = {0, 0};
        HIPCHK(hipMemcpy(flag, s->bar, sizeof flag, hipMemcpyDeviceToHost));
#ifdef GDPT_PT_TIMING
        unsigned tk[16];
        HIPCHK(hipMemcpy(tk, s->bar + 16, sizeof tk, hipMemcpyDeviceToHost));
        for (int b = 0; b < 2; b++)
            fprintf(stderr, "persistent CG phase clocks (10 ns, last launch, tile %s): stencil+sum %u | gather A %u | update+sum %u | gather B %u | ring %u | ring wait %u\n",
                    b ? "G/2" : "0", tk[8 * b], tk[8 * b + 1], tk[8 * b + 2], tk[8 * b + 3], tk[8 * b + 4], tk[8 * b + 5]);
#endif
        if (flag[1]) {
            // A gather timed out (the workgroups were not all resident, e.g. the device is shared).  Not an error of the
            // solve: redo it from x0 with the multi-kernel graphs and stay on them.
            s->log("persistent CG: a grid-wide gather timed out; falling back to the multi-kernel path\n");
            s->fusion = 1;
            if (!s->g0) return fail(GDPT_ERR_HIP, "persistent CG timed out and no graph path was captured");
            HIPCHK(hipMemcpyAsync(s->x, s->x0, sizeof(float) * 3 * (size_t)s->W * s->H, hipMemcpyDeviceToDevice, s->stream));
            int rc = gdpt_poisson_solve_indirect_async(s);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(s->stream));
            if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->last_seconds = ms * 1.0e-3f;
        }
    }
    return GDPT_OK;
}

int gdpt_poisson_solve_indirect(gdpt_poisson_solver *s)
{
    int rc = gdpt_poisson_solve_indirect_async(s);
    if (rc) return rc;
    rc = gdpt_poisson_sync(s);
    if (rc) return rc;
    s->log("Execution time = %.2f s\n", s->last_seconds); // Solver.cpp:500
    return GDPT_OK;
}

static int export_common(gdpt_poisson_solver *s, float *dst, hipMemcpyKind kind)
{
    if (!s || !s->ready || !dst) return fail(GDPT_ERR_INVALID, "export_images before setup_backend / null destination");
    if (s->unchecked) { const int rc = gdpt_poisson_sync(s); if (rc) return rc; }     // solve_indirect_async + export: a timed-out persistent solve is redone before anything is exported
    const long n3 = 3L * s->W * s->H;
    const float *final_ = s->x;
    if (s->dev_direct) { // Solver.cpp:563-566: r = direct ; r = 1*r + x
        hipLaunchKernelGGL(kg_axpy, dim3(grid_generic(n3)), dim3(BLK), 0, s->stream, s->rec, 1.0f, 1.0f, 1.0f, s->dev_direct, s->x, (int)n3);
        final_ = s->rec;
    }
    HIPCHK(hipMemcpyAsync(dst, final_, sizeof(float) * n3, kind, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return GDPT_OK;
}

// Solver::evaluateMetricsMTS (Solver.cpp:511-541): e = b - P x on the device, then -- as the reference does through Backend::map -- the
// 3n Vec3f rows on the host: errL1 / errL2 are sequential fp32 sums of their lengths / squared lengths over 3n, err the first n rows.
int gdpt_poisson_evaluate_metrics(gdpt_poisson_solver *s, float *err, float *errL1, float *errL2)
{
    if (!s || !s->ready || !err || !errL1 || !errL2) return fail(GDPT_ERR_INVALID, "evaluate_metrics before setup_backend / null argument");
    if (s->unchecked) { const int rc = gdpt_poisson_sync(s); if (rc) return rc; }
    const long n = (long)s->W * s->H, n3 = 3 * n;
    const Lattice L = s->lat();
    hipLaunchKernelGGL(kg_residual, dim3(grid_generic(n3)), dim3(BLK), 0, s->stream, s->e, s->b, s->x, L.W, L.H, L.alpha);
    HIPCHK(hipGetLastError());
    std::vector<float> e((size_t)3 * n3);
    HIPCHK(hipMemcpyAsync(e.data(), s->e, sizeof(float) * 3 * n3, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    float l1 = 0.0f, l2 = 0.0f;
    for (long i = 0; i < 3 * n; i++) {
        const float sq = e[3 * i] * e[3 * i] + e[3 * i + 1] * e[3 * i + 1] + e[3 * i + 2] * e[3 * i + 2];
        l1 += sqrtf(sq);
        l2 += sq;
    }
    *errL1 = l1 / (float)(n * 3);
    *errL2 = l2 / (float)(n * 3);
    memcpy(err, e.data(), sizeof(float) * n3);
    return GDPT_OK;
}

// ---- G-BDPT's reconstruction stage (gbdpt.cpp:178-247,264-280) ----------------------------------------------------------------------
// prepareDataForSolver, element by element with the promotions of the C++ expressions: `out[i] = w*float(data[i])` is fp32;
// `out[i] *= 0.5` and `out[i] -= 0.5*w*float(data2[io])` go through double (0.5 is a double literal) and round to fp32 on the store.
__global__ __launch_bounds__(BLK) void k_gbdpt_prepare(float w, float *__restrict__ out, const double *__restrict__ data, long len,
                                                       const double *__restrict__ data2, long off3)
{
    for (long i = (long)blockIdx.x * BLK + threadIdx.x; i < len; i += (long)gridDim.x * BLK) {
        float o = w * (float)data[i];
        if (data2) {
            const long io = i + off3;
            if (io >= 0 && io < len) {
                o = (float)((double)o * 0.5);
                o = (float)((double)o - (0.5 * (double)w) * (double)(float)data2[io]);
            }
        }
        out[i] = o;
    }
}

int gdpt_gbdpt_prepare_data_device(float w, float *out, const double *data, int len, const double *data2, int offset, void *stream)
{
    if (!out || !data || len < 0) return fail(GDPT_ERR_INVALID, "gbdpt_prepare_data: null pointer or negative length");
    if (len == 0) return GDPT_OK;
    hipLaunchKernelGGL(k_gbdpt_prepare, dim3(grid_generic(len)), dim3(BLK), 0, (hipStream_t)stream, w, out, data, (long)len, data2, 3L * (long)offset);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_gbdpt_prepare_data(float w, float *out, const double *data, int len, const double *data2, int offset)
{
    if (!out || !data || len < 0) return fail(GDPT_ERR_INVALID, "gbdpt_prepare_data: null pointer or negative length");
    if (len == 0) return GDPT_OK;
    double *dd = nullptr, *dd2 = nullptr;
    float *dout = nullptr;
    int rc = GDPT_OK;
    if (hipMalloc(&dd, sizeof(double) * len) != hipSuccess || hipMalloc(&dout, sizeof(float) * len) != hipSuccess ||
        (data2 && hipMalloc(&dd2, sizeof(double) * len) != hipSuccess)) rc = fail(GDPT_ERR_HIP, "Out of memory!");
    if (!rc && (hipMemcpy(dd, data, sizeof(double) * len, hipMemcpyHostToDevice) != hipSuccess ||
                (data2 && hipMemcpy(dd2, data2, sizeof(double) * len, hipMemcpyHostToDevice) != hipSuccess))) rc = fail(GDPT_ERR_HIP, "upload failed");
    if (!rc) rc = gdpt_gbdpt_prepare_data_device(w, dout, dd, len, dd2, offset, nullptr);
    if (!rc && hipMemcpy(out, dout, sizeof(float) * len, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(GDPT_ERR_HIP, "download failed");
    hipFree(dd); hipFree(dd2); hipFree(dout);
    return rc;
}

// The second half of GBDPTIntegrator::render on DEVICE buffers (five developed double images in, fp32 reconstructions out, host or device).
// The integrator has no handle of its own at this level, so the library keeps what a frame needs from one call to the next: per (device, size,
// alpha) the three fp32 input images and one solver per preset.  Creating and destroying them per frame (round 3) put three hipMallocs and the
// COLD first cooperative launch of a fresh solver inside every frame's reported solve: 6.8 ms for config 5's L2D where config 2's identical
// 1280x720 solve takes 0.65 ms.  gdpt_gbdpt_reconstruct_release() drops the cache.
namespace {
struct GbdptRecon {
    int device = -1, width = 0, height = 0;
    float alpha = 0.0f;
    float *in[3] = {nullptr, nullptr, nullptr};                  // imgf, dyf, dxf
    gdpt_poisson_solver *sv[2] = {nullptr, nullptr};             // L2D, L1D (created on first use)
    unsigned long long stamp = 0;
    int users = 0;                                               // frames in flight on this entry (under g_reconMutex): an entry in use is never evicted
    std::mutex work;                                             // one frame at a time PER ENTRY: hosts that drive several GPUs from several threads solve side by side
    void freeBuffers()                                           // the solvers and images only: the (device, size) key stays, so that frames queued on `work` rebuild into an
    {                                                            // entry that lookups still match and that eviction still destroys on its own device
        for (auto &p : sv) { if (p) gdpt_poisson_destroy(p); p = nullptr; }
        for (auto &f : in) { if (f) hipFree(f); f = nullptr; }
    }
    void drop() { freeBuffers(); width = height = 0; device = -1; }   // (an entry that is being erased)
};
std::mutex g_reconMutex;                                         // guards the list, not the solves
std::list<GbdptRecon> g_recon;
unsigned long long g_reconClock = 0;
constexpr size_t GBDPT_RECON_CACHE = 8;                          // (device, size) pairs kept: strips of a multi-GPU host, a few film sizes
}

static int gbdpt_reconstruct_core(double *const dev[5], int width, int height, float alpha, int device, float *recL2, float *recL1, bool outOnDevice, float *seconds2)
{
    int cur = device;
    if (cur < 0 && hipGetDevice(&cur) != hipSuccess) return fail(GDPT_ERR_HIP, "gbdpt_reconstruct: no current device");
    const int len = 3 * width * height;
    GbdptRecon *R = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_reconMutex);
        for (auto &e : g_recon) if (e.device == cur && e.width == width && e.height == height && e.alpha == alpha) R = &e;
        if (!R) {
            if (g_recon.size() >= GBDPT_RECON_CACHE) {           // evict the idle entry used longest ago (the callers have made `cur` the current device)
                auto old = g_recon.end();
                for (auto it = g_recon.begin(); it != g_recon.end(); ++it) if (it->users == 0 && (old == g_recon.end() || it->stamp < old->stamp)) old = it;
                if (old != g_recon.end()) { (void)hipSetDevice(old->device); old->drop(); (void)hipSetDevice(cur); g_recon.erase(old); }
            }
            g_recon.emplace_back();
            R = &g_recon.back();
            R->device = cur; R->width = width; R->height = height; R->alpha = alpha;
        }
        R->stamp = ++g_reconClock;
        R->users++;
    }
    int rc = GDPT_OK;
    {
        std::lock_guard<std::mutex> frame(R->work);
        for (int k = 0; k < 3 && !rc; k++) if (!R->in[k] && hipMalloc(&R->in[k], sizeof(float) * len) != hipSuccess) rc = fail(GDPT_ERR_HIP, "Out of memory!");
        float **in = R->in;
        if (!rc) rc = gdpt_gbdpt_prepare_data_device(1.0f, in[0], dev[0], len, nullptr, 0, nullptr);            // gbdpt.cpp:206
        if (!rc) rc = gdpt_gbdpt_prepare_data_device(1.0f, in[1], dev[4], len, dev[1], width, nullptr);          // :207  dy: grad[3] (+y) with grad[0] (-y)
        if (!rc) rc = gdpt_gbdpt_prepare_data_device(1.0f, in[2], dev[3], len, dev[2], 1, nullptr);              // :208  dx: grad[2] (+x) with grad[1] (-x)
        if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(GDPT_ERR_HIP, "gbdpt_reconstruct: prepare failed");
        const char *presets[2] = {"L2D", "L1D"};                     // :213-218, both with m_reconstructAlpha
        float *outs[2] = {recL2, recL1};
        for (int k = 0; k < 2 && !rc; k++) {
            if (seconds2) seconds2[k] = 0.0f;
            if (!outs[k]) continue;
            if (!R->sv[k]) {
                gdpt_poisson_params p;
                gdpt_poisson_params_defaults(&p);
                gdpt_poisson_params_preset(&p, presets[k]);
                p.alpha = alpha;
                p.device = device;
                rc = gdpt_poisson_create(&p, &R->sv[k]);
            }
            gdpt_poisson_solver *sv = R->sv[k];
            if (!rc) rc = gdpt_poisson_import_images_device(sv, in[2], in[1], in[0], nullptr, width, height);   // importImagesMTS(dx, dy, img, NULL), :229,243
            if (!rc) rc = gdpt_poisson_setup_backend(sv);
            if (!rc) rc = gdpt_poisson_solve_indirect(sv);
            if (!rc) rc = outOnDevice ? gdpt_poisson_export_images_device(sv, outs[k]) : gdpt_poisson_export_images(sv, outs[k]);
            if (!rc && outOnDevice) rc = gdpt_poisson_sync(sv);
            if (!rc && seconds2) seconds2[k] = gdpt_poisson_last_solve_seconds(sv);
        }
        if (rc) R->freeBuffers();                                // a failed frame leaves nothing half-built behind (the emptied entry keeps its key; it goes when nobody uses it)
    }
    std::lock_guard<std::mutex> lock(g_reconMutex);
    R->users--;
    if (rc && R->users == 0)
        for (auto it = g_recon.begin(); it != g_recon.end(); ++it) if (&*it == R) { g_recon.erase(it); break; }
    return rc;
}

int gdpt_gbdpt_reconstruct_release(void)
{
    std::lock_guard<std::mutex> lock(g_reconMutex);
    int back = 0;
    const bool have = hipGetDevice(&back) == hipSuccess;
    for (auto it = g_recon.begin(); it != g_recon.end();)      // (an entry with a frame in flight on another thread stays)
        if (it->users == 0) { if (it->device >= 0) (void)hipSetDevice(it->device); it->drop(); it = g_recon.erase(it); } else ++it;
    if (have) (void)hipSetDevice(back);
    return GDPT_OK;
}

int gdpt_gbdpt_reconstruct_release_size(int device, int width, int height)
{
    std::lock_guard<std::mutex> lock(g_reconMutex);
    int back = 0;
    const bool have = hipGetDevice(&back) == hipSuccess;
    if (device < 0) device = back;
    for (auto it = g_recon.begin(); it != g_recon.end();)
        if (it->users == 0 && it->device == device && it->width == width && it->height == height) { (void)hipSetDevice(it->device); it->drop(); it = g_recon.erase(it); } else ++it;
    if (have) (void)hipSetDevice(back);
    return GDPT_OK;
}

int gdpt_gbdpt_reconstruct(const double *primal, const double *gradNegY, const double *gradNegX, const double *gradPosX, const double *gradPosY,
                           int width, int height, float alpha, int device, float *recL2, float *recL1)
{
    if (!primal || !gradNegY || !gradNegX || !gradPosX || !gradPosY || width <= 0 || height <= 0)
        return fail(GDPT_ERR_INVALID, "gbdpt_reconstruct: null buffer or empty image");
    if (device >= 0) HIPCHK(hipSetDevice(device));
    const int len = 3 * width * height;
    const double *host[5] = {primal, gradNegY, gradNegX, gradPosX, gradPosY};
    double *dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = GDPT_OK;
    for (int k = 0; k < 5 && !rc; k++)
        if (hipMalloc(&dev[k], sizeof(double) * len) != hipSuccess || hipMemcpy(dev[k], host[k], sizeof(double) * len, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(GDPT_ERR_HIP, "gbdpt_reconstruct: upload failed");
    if (!rc) rc = gbdpt_reconstruct_core(dev, width, height, alpha, device, recL2, recL1, false, nullptr);
    for (double *d : dev) hipFree(d);
    return rc;
}

int gdpt_gbdpt_reconstruct_device(const double *primal, const double *gradNegY, const double *gradNegX, const double *gradPosX, const double *gradPosY,
                                  int width, int height, float alpha, int device, float *recL2, float *recL1, float solveSeconds[2])
{
    if (!primal || !gradNegY || !gradNegX || !gradPosX || !gradPosY || width <= 0 || height <= 0)
        return fail(GDPT_ERR_INVALID, "gbdpt_reconstruct_device: null buffer or empty image");
    if (device >= 0) HIPCHK(hipSetDevice(device));
    double *dev[5] = {const_cast<double *>(primal), const_cast<double *>(gradNegY), const_cast<double *>(gradNegX), const_cast<double *>(gradPosX), const_cast<double *>(gradPosY)};
    return gbdpt_reconstruct_core(dev, width, height, alpha, device, recL2, recL1, true, solveSeconds);
}

int gdpt_poisson_export_images(gdpt_poisson_solver *s, float *rec) { return export_common(s, rec, hipMemcpyDeviceToHost); }
int gdpt_poisson_export_images_device(gdpt_poisson_solver *s, float *rec) { return export_common(s, rec, hipMemcpyDeviceToDevice); }

int gdpt_poisson_solution_device(gdpt_poisson_solver *s, float **x)
{
    if (!s || !s->ready || !x) return fail(GDPT_ERR_INVALID, "no solution yet");
    *x = s->x;
    return GDPT_OK;
}

float gdpt_poisson_last_solve_seconds(const gdpt_poisson_solver *s) { return s ? s->last_seconds : 0.0f; }
long gdpt_poisson_last_iterations(const gdpt_poisson_solver *s) { return s ? s->last_iters : 0; }
void *gdpt_poisson_stream(gdpt_poisson_solver *s) { return s ? (void *)s->stream : nullptr; }

int gdpt_poisson_set_fusion(gdpt_poisson_solver *s, int level)
{
    if (!s) return fail(GDPT_ERR_INVALID, "null solver");
    s->fusion = level < 0 ? 0 : (level > 3 ? 3 : level);
    return GDPT_OK;
}


// Bench hook: average standalone duration (microseconds, HIP events on the handle's stream) of each CG kernel
// at the handle's geometry: us[0] stencil (calc_Ax_xAx), us[1] calc_r_rz, us[2] calc_x_p, us[3] fused x_p+stencil
// (0 when the geometry cannot fuse).  Clobbers the iterate: call setup_backend again before the next solve.
int gdpt_poisson_profile_kernels(gdpt_poisson_solver *s, int reps, float us[4])
{
    if (!s || !s->ready || reps < 1 || !us) return fail(GDPT_ERR_INVALID, "profile_kernels needs a set-up solver");
    const Lattice L = s->lat();
    const long n3 = 3 * L.n();
    const bool unitw = s->P.irlsIterMax == 1;
    hipStream_t st = s->stream;
    hipLaunchKernelGGL(kg_set, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->w2, 1.0f, (size_t)n3);
    hipLaunchKernelGGL(kg_set, dim3(16), dim3(BLK), 0, st, s->scal, 1.0f, (size_t)16);
    hipLaunchKernelGGL(kg_set, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->p[0], 0.5f, (size_t)n3);
    hipLaunchKernelGGL(kg_set, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->r, 0.25f, (size_t)n3);
    const int Gt = imin(L.tilesF(), MAXP);
    int Ga = 0, Gr = 0;
    for (int which = 0; which < 4; which++) {
        us[which] = 0.0f;
        if (which == 3 && !can_fuse(s)) continue;
        for (int pass = 0; pass < 2; pass++) { // pass 0 = warm-up
            const int R = pass ? reps : 3;
            HIPCHK(hipEventRecord(s->ev0, st));
            for (int k = 0; k < R; k++) {
                if (which == 0) Ga = launch_Ax(st, L, unitw, s->Ap, s->part_pAp, s->w2, s->p[0]);
                if (which == 1) Gr = launch_r_rz(st, n3, s->r, s->part_rz, s->Ap, s->s_rz_next(), nullptr, s->part_pAp, Ga, s->s_pAp(), s->s_rz_old());
                if (which == 2) launch_x_p(st, n3, s->x, s->p[0], s->r, nullptr, s->s_rz_old(), s->s_pAp(), s->part_rz, Gr, s->s_rz_next());
                if (which == 3) launch_xp_Ax(st, L, unitw, s->Ap, s->part_pAp, s->w2, s->x, s->p[k & 1], s->p[(k + 1) & 1], s->r, s->s_rz_old(), s->s_pAp(), s->part_rz, Gr, s->s_rz_next());
            }
            HIPCHK(hipEventRecord(s->ev1, st));
            HIPCHK(hipStreamSynchronize(st));
            float ms = 0.0f;
            HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
            if (pass) us[which] = ms * 1000.0f / (float)R;
        }
    }
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

// Bench hook: the yardstick beside kf_xp_Ax's HBM fraction, measured in the SAME process on the same device -- a bare streaming kernel with that kernel's access
// mix (three coalesced 16-byte reads from three arrays, three non-temporal 16-byte writes to three others, no stencil, no reuse) over the handle's own CG
// vectors (x, r, p[0] in; Ap, p[1], rec out: the solver's buffers, clobbered -- call setup_backend again before the next solve).  us = best of `reps` launches.
int gdpt_poisson_profile_stream(gdpt_poisson_solver *s, int reps, float *us)
{
    if (!s || !s->ready || reps < 1 || !us) return fail(GDPT_ERR_INVALID, "profile_stream needs a set-up solver");
    const size_t n4 = (size_t)3 * s->W * s->H / 4;
    hipStream_t st = s->stream;
    float best = 1e30f;
    for (int k = 0; k < reps + 2; k++) {          // (two warm-up launches)
        HIPCHK(hipEventRecord(s->ev0, st));
        hipLaunchKernelGGL(kg_stream33, dim3(1024), dim3(BLK), 0, st, (const float4 *)s->x, (const float4 *)s->r, (const float4 *)s->p[0], (float4 *)s->Ap, (float4 *)s->p[1], (float4 *)s->rec, n4);
        HIPCHK(hipEventRecord(s->ev1, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        if (k >= 2 && ms < best) best = ms;
    }
    *us = best * 1000.0f;
    return GDPT_OK;
}

int gdpt_poisson_profile_persistent(gdpt_poisson_solver *s, int reps, float *us)
{
    if (!s || !s->ready || reps < 1 || !us) return fail(GDPT_ERR_INVALID, "profile_persistent needs a set-up solver");
    *us = 0.0f;
    if (!persistent_geometry(s)) return GDPT_OK;
    const long n3 = 3L * s->W * s->H;
    const bool unitw = s->P.irlsIterMax == 1;
    hipStream_t st = s->stream;
    hipLaunchKernelGGL(kg_set, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->w2, 1.0f, (size_t)n3);
    hipLaunchKernelGGL(kg_set, dim3(16), dim3(BLK), 0, st, s->scal, 1.0f, (size_t)16);
    hipLaunchKernelGGL(kg_set, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->r, 0.25f, (size_t)n3);
    hipLaunchKernelGGL(kg_set, dim3(grid_generic(n3)), dim3(BLK), 0, st, s->x, 0.5f, (size_t)n3);
    HIPCHK(hipMemsetAsync(s->bar, 0, sizeof(unsigned) * PT_BAR_WORDS, st));
    for (int pass = 0; pass < 2; pass++) { // pass 0 = warm-up
        const int R = pass ? reps : 2;
        HIPCHK(hipEventRecord(s->ev0, st));
        for (int k = 0; k < R; k++) {
            int rc = enqueue_cg_persistent(s, unitw, s->P.cgIterMax);
            if (rc) return rc;
        }
        HIPCHK(hipEventRecord(s->ev1, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        if (pass) *us = ms * 1000.0f / (float)R;
    }
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

// =================================================================================================
// backend-op level
// =================================================================================================

int gdpt_backend_calc_MIx(float *MIx, int w, int h, float alpha, const float *w2, const float *x, void *stream)
{
    if (!MIx || !w2 || !x || w <= 0 || h <= 0) return fail(GDPT_ERR_INVALID, "calc_MIx: bad argument");
    const long n3 = 3L * w * h;
    hipLaunchKernelGGL(kg_MIx, dim3(grid_generic(n3)), dim3(BLK), 0, (hipStream_t)stream, MIx, w2, x, w, h, alpha);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

// Backend::tonemapSRGB (Backend.cpp:442-468): out[i] = ABGR_8888 of sRGB(in[i + idx * numPixels] * scale + bias); in: Vec3f per pixel
int gdpt_backend_tonemap_srgb(unsigned *out, const float *in, int idx, int numPixels, float scale, float bias, void *stream)
{
    if (!out || !in || numPixels <= 0 || idx < 0) return fail(GDPT_ERR_INVALID, "tonemap_srgb: bad argument");
    hipLaunchKernelGGL(kg_tonemap_srgb, dim3((numPixels + BLK - 1) / BLK), dim3(BLK), 0, (hipStream_t)stream, out, in + (size_t)idx * numPixels * 3, numPixels, scale, bias);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}
// Backend::tonemapLinear (Backend.cpp:472-507): scale by the range of the `idx`-th block of numPixels * numComponents floats, |.|, pack.
// `out` holds at least two words: the first two serve as the min / max cells of pass A before pass B overwrites them (as BackendCUDA.cu:647-653).
int gdpt_backend_tonemap_linear(unsigned *out, const float *in, int idx, int numPixels, int numComponents, float scaleMin, float scaleMax, int hasNegative, void *stream)
{
    if (!out || !in || numPixels < 2 || numComponents < 1 || idx < 0) return fail(GDPT_ERR_INVALID, "tonemap_linear: bad argument");
    const int total = numPixels * numComponents;
    const float *src = in + (size_t)idx * total;
    hipStream_t st = (hipStream_t)stream;
    unsigned *cells = nullptr;                                                            // (separate cells: pass B reads them while it writes out[0..1])
    HIPCHK(hipMallocAsync((void **)&cells, 2 * sizeof(unsigned), st));
    const unsigned init[2] = {~0u, 0u};
    HIPCHK(hipMemcpyAsync(cells, init, sizeof init, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(kg_tonemap_minmax, dim3((total + BLK - 1) / BLK), dim3(BLK), 0, st, cells, src, total);
    hipLaunchKernelGGL(kg_tonemap_linear, dim3((numPixels + BLK - 1) / BLK), dim3(BLK), 0, st, out, src, cells, numPixels, numComponents, scaleMin, scaleMax, hasNegative);
    HIPCHK(hipGetLastError());
    HIPCHK(hipFreeAsync(cells, st));
    return GDPT_OK;
}
// Backend::allocTimer / freeTimer / beginTimer / endTimer (Backend.hpp:95-98; BackendCUDA.cu: cudaEvent pairs): device time of the work between
// begin and end on `stream`, in seconds -- a host clock around asynchronous launches would time the enqueue
struct gdpt_backend_timer { hipEvent_t e0, e1; };
gdpt_backend_timer *gdpt_backend_timer_alloc(void)
{
    if (ensure_device(-1)) return nullptr;
    gdpt_backend_timer *t = new gdpt_backend_timer;
    if (hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess) { delete t; fail(GDPT_ERR_HIP, "timer_alloc: hipEventCreate failed"); return nullptr; }
    return t;
}
void gdpt_backend_timer_free(gdpt_backend_timer *t) { if (t) { hipEventDestroy(t->e0); hipEventDestroy(t->e1); delete t; } }
int gdpt_backend_timer_begin(gdpt_backend_timer *t, void *stream)
{
    if (!t) return fail(GDPT_ERR_INVALID, "timer_begin: null timer");
    HIPCHK(hipEventRecord(t->e0, (hipStream_t)stream));
    return GDPT_OK;
}
int gdpt_backend_timer_end(gdpt_backend_timer *t, void *stream, float *seconds)
{
    if (!t || !seconds) return fail(GDPT_ERR_INVALID, "timer_end: null argument");
    HIPCHK(hipEventRecord(t->e1, (hipStream_t)stream));
    HIPCHK(hipEventSynchronize(t->e1));
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, t->e0, t->e1));
    *seconds = ms * 1e-3f;
    return GDPT_OK;
}

void *gdpt_backend_alloc(size_t bytes)
{
    void *p = nullptr;
    if (ensure_device(-1)) return nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { fail(GDPT_ERR_HIP, "Out of memory!"); return nullptr; }
    return p;
}

void gdpt_backend_free(void *ptr) { if (ptr) hipFree(ptr); }

int gdpt_backend_set(float *x, float y, size_t numFloats, void *stream)
{
    hipLaunchKernelGGL(kg_set, dim3(grid_generic((long)numFloats)), dim3(BLK), 0, (hipStream_t)stream, x, y, numFloats);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_backend_copy(void *x, const void *y, size_t bytes, void *stream)
{
    HIPCHK(hipMemcpyAsync(x, y, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GDPT_OK;
}

int gdpt_backend_read(void *host, const void *x, size_t bytes, void *stream)
{
    HIPCHK(hipMemcpyAsync(host, x, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return GDPT_OK;
}

int gdpt_backend_write(void *x, const void *host, size_t bytes, void *stream)
{
    HIPCHK(hipMemcpyAsync(x, host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return GDPT_OK;
}

int gdpt_backend_sync(void *stream)
{
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return GDPT_OK;
}

static int lattice_ok(int w, int h)
{
    if (w <= 0 || h <= 0 || (long)w * h * 9 >= (1L << 31)) return fail(GDPT_ERR_INVALID, "bad lattice %dx%d", w, h);
    return GDPT_OK;
}

int gdpt_backend_calc_Px(float *Px, int w, int h, float alpha, const float *x, void *stream)
{
    if (lattice_ok(w, h)) return GDPT_ERR_INVALID;
    hipLaunchKernelGGL(kg_Px, dim3(grid_generic(3L * w * h)), dim3(BLK), 0, (hipStream_t)stream, Px, x, w, h, alpha);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_backend_calc_PTW2x(float *out, int w, int h, float alpha, const float *w2, const float *x, void *stream)
{
    if (lattice_ok(w, h)) return GDPT_ERR_INVALID;
    hipLaunchKernelGGL(kg_PTW2x<false>, dim3(grid_generic(3L * w * h)), dim3(BLK), 0, (hipStream_t)stream, out, (float *)nullptr, (float4 *)nullptr, w2, x, w, h, alpha);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

// scratch for block partials of the op-level reductions: stream-ordered allocation
// Block partials of the op-level reductions: one buffer per (device, stream), allocated once and kept.  (Round 1 took it from the
// stream-ordered allocator on every call -- hipMallocAsync / hipFreeAsync -- and one ad-hoc run saw calc_w2 return inf on the second
// call of a fresh process, i.e. a zero sum of partials, which never reproduced; a pool block handed out again while a free was still
// pending is the one thing in that path that was not plain stream order.  A persistent buffer per stream has no such state: consumers
// follow producers on the same stream, and two streams never share a buffer.)
struct PartScratch {
    float4 *p = nullptr;
    explicit PartScratch(hipStream_t s)
    {
        static std::mutex m;
        static std::map<std::pair<int, hipStream_t>, float4 *> pool;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return;
        std::lock_guard<std::mutex> lock(m);
        float4 *&slot = pool[std::make_pair(dev, s)];
        if (!slot && hipMalloc((void **)&slot, sizeof(float4) * MAXP) != hipSuccess) { slot = nullptr; (void)hipGetLastError(); }
        p = slot;
    }
};

int gdpt_backend_calc_Ax_xAx(float *Ax, float *xAx, int w, int h, float alpha, const float *w2, const float *x, void *stream)
{
    if (lattice_ok(w, h)) return GDPT_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    PartScratch ps(st);
    if (!ps.p) return fail(GDPT_ERR_HIP, "Out of memory!");
    const int G = launch_Ax(st, Lattice{w, h, alpha}, false, Ax, ps.p, w2, x);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLK), 0, st, xAx, (float *)nullptr, ps.p, G);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_backend_calc_axpy(float *out, const float a[3], const float *x, const float *y, int numElems, void *stream)
{
    hipLaunchKernelGGL(kg_axpy, dim3(grid_generic(3L * numElems)), dim3(BLK), 0, (hipStream_t)stream, out, a[0], a[1], a[2], x, y, 3 * numElems);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_backend_calc_xdoty(float *xdoty, const float *x, const float *y, int numElems, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    PartScratch ps(st);
    if (!ps.p) return fail(GDPT_ERR_HIP, "Out of memory!");
    const int G = grid_reduce(3L * numElems);
    hipLaunchKernelGGL(kg_xdoty, dim3(G), dim3(BLK), 0, st, ps.p, x, y, 3 * numElems);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLK), 0, st, xdoty, (float *)nullptr, ps.p, G);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_backend_calc_r_rz(float *r, float *rz, const float *Ap, const float *rz2, const float *pAp, int numElems, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    PartScratch ps(st);
    if (!ps.p) return fail(GDPT_ERR_HIP, "Out of memory!");
    const int G = launch_r_rz(st, 3L * numElems, r, ps.p, Ap, rz2, pAp, nullptr, 0, nullptr, nullptr);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLK), 0, st, rz, (float *)nullptr, ps.p, G);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_backend_calc_x_p(float *x, float *p, const float *r, const float *rz, const float *rz2, const float *pAp, int numElems, void *stream)
{
    launch_x_p((hipStream_t)stream, 3L * numElems, x, p, r, rz, rz2, pAp, nullptr, 0, nullptr);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

int gdpt_backend_calc_w2(float *w2, const float *e, float reg, int numElems, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    PartScratch ps(st);
    if (!ps.p) return fail(GDPT_ERR_HIP, "Out of memory!");
    const int G = grid_reduce(numElems);
    hipLaunchKernelGGL(kg_w2_raw, dim3(G), dim3(BLK), 0, st, w2, ps.p, e, (const float *)nullptr, (const int *)nullptr, reg, numElems);
    hipLaunchKernelGGL(kg_w2_scale, dim3(grid_generic(numElems)), dim3(BLK), 0, st, w2, ps.p, G, (int *)nullptr, numElems);
    HIPCHK(hipGetLastError());
    return GDPT_OK;
}

} // extern "C"
