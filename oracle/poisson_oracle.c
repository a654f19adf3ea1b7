/*
 * oracle/poisson_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, strictly sequential fp32) of the reference's
 * screened-Poisson reconstruction: the `poisson::Backend` ops and the
 * `poisson::Solver` IRLS + CG driver.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load this library; the product path
 * (gradientdomain-mitsuba_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED.  The reference holds no test, golden vector or fixture for this
 * path, and its poisson_solver sources are unbuildable in this
 * image: they are Win32/MSVC translation units (#include <windows.h> at
 * Backend.cpp:31, `__int64` at Backend.hpp:57, `_vscprintf`/`vsprintf_s` at
 * Defs.cpp:53-56 and Solver.cpp:606-609) and the image has no such header or CRT.
 * Writing a stand-in header to force a build is not permitted, so no oracle/_ref
 * exists and nothing here has been compared with output of the reference itself.
 * What checks this file instead (tests/test_poisson_oracle.py): hand-derived
 * known-answer cases on 1x1/2x2/2x3 lattices, operator adjointness
 * <P x, e> == <x, P^T e>, A == P^T diag(w) P against a dense construction, and the
 * normal-equation residual of a converged solve.  Those pin the mathematics the
 * cited lines express, not the reference's bits.
 *
 * All citations are relative to /root/reference/src/integrators/poisson_solver/.
 * Images are row-major AoS RGB fp32 ("Vec3f", Defs.hpp:71-93): element i of an
 * n-element vector occupies floats [3i, 3i+2].  Stacked vectors (b, e) hold 3n
 * elements: [alpha*T ; dx ; dy].  w2 holds 3n scalar weights in the same
 * stacking.  Arithmetic is written operation-by-operation in the reference's
 * association order and the file must be compiled with -ffp-contract=off.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GDO_API __attribute__((visibility("default")))

/* The checker is this file compiled WITHOUT OpenMP: the pragmas below vanish and every sum runs in index order.  The second
 * build (-fopenmp -DGDO_OMP, libgdpt_oracle_poisson_omp.so) exists only for the all-cores CPU baseline of bench.py (SURVEY 8d:
 * the reference's BackendOpenMP parallelises the same loops, BackendOpenMP.cpp:192-330); its sums are per-thread partials. */
#ifdef GDO_OMP
#define GDO_PRAGMA(x) _Pragma(#x)
#define GDO_PARALLEL_FOR GDO_PRAGMA(omp parallel for schedule(static))
#define GDO_PARALLEL_FOR_SUM3 GDO_PRAGMA(omp parallel for schedule(static) reduction(+ : acc[:3]))
#define GDO_PARALLEL_FOR_SUM1 GDO_PRAGMA(omp parallel for schedule(static) reduction(+ : w2sum))
#else
#define GDO_PARALLEL_FOR
#define GDO_PARALLEL_FOR_SUM3
#define GDO_PARALLEL_FOR_SUM1
#endif

static inline float fmax_ref(float a, float b) { return (a > b) ? a : b; } /* Defs.hpp:57 */

/* Backend::calc_Px, Backend.cpp:150-174 (== BackendOpenMP.cpp:192-214).
 * Px[0n+i] = x_i*alpha ; Px[1n+i] = x_{i+1}-x_i (0 in last column) ;
 * Px[2n+i] = x_{i+W}-x_i (0 in last row). */
GDO_API void gdo_calc_Px(float *Px, int w, int h, float alpha, const float *x)
{
    const long n = (long)w * h;
    GDO_PARALLEL_FOR
    for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++)
            for (int c = 0; c < 3; c++) {
                const long i = (long)yy * w + xx;
                const float xi = x[3 * i + c];
                Px[3 * (n * 0 + i) + c] = xi * alpha;
                Px[3 * (n * 1 + i) + c] = (xx != w - 1) ? x[3 * (i + 1) + c] - xi : 0.0f;
                Px[3 * (n * 2 + i) + c] = (yy != h - 1) ? x[3 * (i + w) + c] - xi : 0.0f;
            }
}

/* Backend::calc_PTW2x, Backend.cpp:178-205 (== BackendOpenMP.cpp:218-242).
 * out_i = (w0_i*e0_i)*alpha + w1_{i-1}e1_{i-1} - w1_i e1_i + w2_{i-W}e2_{i-W} - w2_i e2_i. */
GDO_API void gdo_calc_PTW2x(float *out, int w, int h, float alpha, const float *w2, const float *x)
{
    const long n = (long)w * h;
    GDO_PARALLEL_FOR
    for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++)
            for (int c = 0; c < 3; c++) {
                const long i = (long)yy * w + xx;
                float v = w2[n * 0 + i] * x[3 * (n * 0 + i) + c] * alpha;
                if (xx != 0)     v = v + w2[n * 1 + i - 1] * x[3 * (n * 1 + i - 1) + c];
                if (xx != w - 1) v = v - w2[n * 1 + i]     * x[3 * (n * 1 + i) + c];
                if (yy != 0)     v = v + w2[n * 2 + i - w] * x[3 * (n * 2 + i - w) + c];
                if (yy != h - 1) v = v - w2[n * 2 + i]     * x[3 * (n * 2 + i) + c];
                out[3 * i + c] = v;
            }
}

/* Backend::calc_Ax_xAx, Backend.cpp:209-242 (== BackendOpenMP.cpp:246-287, which
 * accumulates the three dot products in separate float scalars in index order:
 * identical rounding to the naive Vec3f accumulation). */
GDO_API void gdo_calc_Ax_xAx(float *Ax, float *xAx, int w, int h, float alpha, const float *w2, const float *x)
{
    const long n = (long)w * h;
    const float alphaSqr = alpha * alpha;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    GDO_PARALLEL_FOR_SUM3
    for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++)
            for (int c = 0; c < 3; c++) {
                const long i = (long)yy * w + xx;
                const float xi = x[3 * i + c];
                float a = w2[n * 0 + i] * xi * alphaSqr;
                if (xx != 0)     a = a + w2[n * 1 + i - 1] * (xi - x[3 * (i - 1) + c]);
                if (xx != w - 1) a = a + w2[n * 1 + i]     * (xi - x[3 * (i + 1) + c]);
                if (yy != 0)     a = a + w2[n * 2 + i - w] * (xi - x[3 * (i - w) + c]);
                if (yy != h - 1) a = a + w2[n * 2 + i]     * (xi - x[3 * (i + w) + c]);
                Ax[3 * i + c] = a;
                acc[c] = acc[c] + xi * a;
            }
    xAx[0] = acc[0]; xAx[1] = acc[1]; xAx[2] = acc[2];
}

/* Backend::calc_axpy, Backend.cpp:246-262: out = a*x + y, a is RGB. In-place safe. */
GDO_API void gdo_calc_axpy(float *out, const float *a, const float *x, const float *y, long numElems)
{
    GDO_PARALLEL_FOR
    for (long i = 0; i < numElems; i++)
        for (int c = 0; c < 3; c++)
            out[3 * i + c] = a[c] * x[3 * i + c] + y[3 * i + c];
}

/* Backend::calc_xdoty, Backend.cpp:266-283: per-channel dot, sequential fp32. */
GDO_API void gdo_calc_xdoty(float *out, const float *x, const float *y, long numElems)
{
    float acc[3] = {0.0f, 0.0f, 0.0f};
    GDO_PARALLEL_FOR_SUM3
    for (long i = 0; i < numElems; i++)
        for (int c = 0; c < 3; c++)
            acc[c] = acc[c] + x[3 * i + c] * y[3 * i + c];
    out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
}

/* Backend::calc_r_rz, Backend.cpp:287-315: a = rz2/max(pAp,FLT_MIN); r -= Ap*a; rz = sum r*r. */
GDO_API void gdo_calc_r_rz(float *r, float *rz, const float *Ap, const float *rz2, const float *pAp, long numElems)
{
    float a[3], acc[3] = {0.0f, 0.0f, 0.0f};
    for (int c = 0; c < 3; c++) a[c] = rz2[c] / fmax_ref(pAp[c], FLT_MIN);
    GDO_PARALLEL_FOR_SUM3
    for (long i = 0; i < numElems; i++)
        for (int c = 0; c < 3; c++) {
            const float ri = r[3 * i + c] - Ap[3 * i + c] * a[c];
            r[3 * i + c] = ri;
            acc[c] = acc[c] + ri * ri;
        }
    rz[0] = acc[0]; rz[1] = acc[1]; rz[2] = acc[2];
}

/* Backend::calc_x_p, Backend.cpp:319-350: x += p*a ; p = r + p*b. */
GDO_API void gdo_calc_x_p(float *x, float *p, const float *r, const float *rz, const float *rz2, const float *pAp, long numElems)
{
    float a[3], b[3];
    for (int c = 0; c < 3; c++) {
        a[c] = rz2[c] / fmax_ref(pAp[c], FLT_MIN);
        b[c] = rz[c] / fmax_ref(rz2[c], FLT_MIN);
    }
    GDO_PARALLEL_FOR
    for (long i = 0; i < numElems; i++)
        for (int c = 0; c < 3; c++) {
            const float pi = p[3 * i + c];
            x[3 * i + c] = x[3 * i + c] + pi * a[c];
            p[3 * i + c] = r[3 * i + c] + pi * b[c];
        }
}

/* Backend::calc_w2, Backend.cpp:354-376: w = 1/(|e_i|_2 + reg) over ALL numElems rows
 * (including the structurally-zero last-column/last-row gradient rows), then
 * scaled by numElems / sum(w). */
GDO_API void gdo_calc_w2(float *w2, const float *e, float reg, long numElems)
{
    float w2sum = 0.0f;
    GDO_PARALLEL_FOR_SUM1
    for (long i = 0; i < numElems; i++) {
        const float ex = e[3 * i], ey = e[3 * i + 1], ez = e[3 * i + 2];
        const float len = sqrtf(ex * ex + ey * ey + ez * ez); /* Defs.hpp:105-106 */
        const float wi = 1.0f / (len + reg);
        w2[i] = wi;
        w2sum = w2sum + wi;
    }
    const float coef = (float)numElems / w2sum;
    GDO_PARALLEL_FOR
    for (long i = 0; i < numElems; i++)
        w2[i] = w2[i] * coef;
}

/* Backend::calc_MIx, Backend.cpp:387-438 (incomplete-Poisson preconditioner; only
 * reachable with cgPrecond=true, which no preset sets). */
GDO_API void gdo_calc_MIx(float *MIx, int w, int h, float alpha, const float *w2, const float *x)
{
    const long n = (long)w * h;
    const float alphaSqr = alpha * alpha;
    float *t = (float *)malloc(sizeof(float) * 3 * n);
    float *DIt = (float *)malloc(sizeof(float) * 3 * n);
    long i = 0;
    for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++, i++)
            for (int c = 0; c < 3; c++) {
                float Di = w2[n * 0 + i] * alphaSqr;
                float Uxi = 0.0f;
                if (xx != 0)     Di = Di + w2[n * 1 + i - 1];
                if (xx != w - 1) { Di = Di + w2[n * 1 + i]; Uxi = Uxi - w2[n * 1 + i] * x[3 * (i + 1) + c]; }
                if (yy != 0)     Di = Di + w2[n * 2 + i - w];
                if (yy != h - 1) { Di = Di + w2[n * 2 + i]; Uxi = Uxi - w2[n * 2 + i] * x[3 * (i + w) + c]; }
                t[3 * i + c] = x[3 * i + c] - Uxi / Di;
                DIt[3 * i + c] = t[3 * i + c] / Di;
            }
    i = 0;
    for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++, i++)
            for (int c = 0; c < 3; c++) {
                float L = 0.0f;
                if (xx != 0) L = L - w2[n * 1 + i - 1] * DIt[3 * (i - 1) + c];
                if (yy != 0) L = L - w2[n * 2 + i - w] * DIt[3 * (i - w) + c];
                MIx[3 * i + c] = t[3 * i + c] - L;
            }
    free(t);
    free(DIt);
}

/* Solver::Params solver-configuration fields, Solver.hpp:85-93, and the presets of
 * Solver::Params::setConfigPreset, Solver.cpp:94-178. */
typedef struct gdo_params {
    float alpha;
    int   irlsIterMax;
    float irlsRegInit;
    float irlsRegIter;
    int   cgIterMax;
    int   cgIterCheck;
    int   cgPrecond;
    float cgTolerance;
} gdo_params;

GDO_API int gdo_params_preset(gdo_params *p, const char *preset)
{
    p->irlsIterMax = 1; p->irlsRegInit = 0.0f; p->irlsRegIter = 0.0f;
    p->cgIterMax = 1; p->cgIterCheck = 100; p->cgPrecond = 0; p->cgTolerance = 0.0f;
    if (!strcmp(preset, "L1D")) { p->irlsIterMax = 20; p->irlsRegInit = 0.05f;  p->irlsRegIter = 0.5f;  p->cgIterMax = 50;   return 1; }
    if (!strcmp(preset, "L1Q")) { p->irlsIterMax = 64; p->irlsRegInit = 1.0f;   p->irlsRegIter = 0.7f;  p->cgIterMax = 1000; return 1; }
    if (!strcmp(preset, "L1L")) { p->irlsIterMax = 7;  p->irlsRegInit = 1.0e-4f; p->irlsRegIter = 1.0e-1f; p->cgIterMax = 20000; p->cgTolerance = 1.0e-20f; return 1; }
    if (!strcmp(preset, "L2D")) { p->cgIterMax = 50;  return 1; }
    if (!strcmp(preset, "L2Q")) { p->cgIterMax = 500; return 1; }
    return 0;
}

/* Solver::Params::sanitize, Solver.cpp:182-192. */
static void gdo_sanitize(gdo_params *p)
{
    p->alpha = fmax_ref(p->alpha, 0.0f);
    if (p->irlsIterMax < 1) p->irlsIterMax = 1;
    p->irlsRegInit = fmax_ref(p->irlsRegInit, 0.0f);
    p->irlsRegIter = fmax_ref(p->irlsRegIter, 0.0f);
    if (p->cgIterMax < 1) p->cgIterMax = 1;
    if (p->cgIterCheck < 1) p->cgIterCheck = 1;
    p->cgTolerance = fmax_ref(p->cgTolerance, 0.0f);
}

/* Solver::setupBackend (Solver.cpp:257-338) + Solver::solveIndirect (Solver.cpp:374-509)
 * + Solver::exportImagesMTS (Solver.cpp:542-582).  `tp` and `direct` may be NULL exactly
 * as in importImagesMTS (Solver.cpp:220-228).  Returns the number of CG iterations
 * executed in total; writes 3*w*h floats to `rec`.  If `x_out` is non-NULL also
 * returns the indirect solution x. */
GDO_API long gdo_solve(const gdo_params *params_in, const float *dx, const float *dy, const float *tp,
                       const float *direct, int w, int h, float *rec, float *x_out)
{
    gdo_params P = *params_in;
    gdo_sanitize(&P);
    const long n = (long)w * h;
    const float alpha = tp ? P.alpha : 0.0f;               /* Solver.cpp:319 */
    const float minus1[3] = {-1.0f, -1.0f, -1.0f};
    const float plus1[3] = {1.0f, 1.0f, 1.0f};

    float *b  = (float *)malloc(sizeof(float) * 9 * n);
    float *e  = (float *)malloc(sizeof(float) * 9 * n);
    float *w2 = (float *)malloc(sizeof(float) * 3 * n);
    float *x  = (float *)malloc(sizeof(float) * 3 * n);
    float *r  = (float *)malloc(sizeof(float) * 3 * n);
    float *z  = (float *)malloc(sizeof(float) * 3 * n);
    float *p  = (float *)malloc(sizeof(float) * 3 * n);
    float *Ap = (float *)malloc(sizeof(float) * 3 * n);
    float rr[3], rzA[3], rzB[3], pAp[3];
    float *rz = rzA, *rz2 = rzB;
    long iters = 0;

    for (long i = 0; i < n; i++)                            /* Solver.cpp:323-329 */
        for (int c = 0; c < 3; c++) {
            b[3 * (n * 0 + i) + c] = tp ? tp[3 * i + c] * alpha : 0.0f;
            b[3 * (n * 1 + i) + c] = dx[3 * i + c];
            b[3 * (n * 2 + i) + c] = dy[3 * i + c];
        }
    if (tp) memcpy(x, tp, sizeof(float) * 3 * n);           /* Solver.cpp:334-337 */
    else    memset(x, 0, sizeof(float) * 3 * n);

    for (int irlsIter = 0; irlsIter < P.irlsIterMax; irlsIter++) {
        gdo_calc_Px(e, w, h, alpha, x);                     /* Solver.cpp:386 */
        gdo_calc_axpy(e, minus1, e, b, 3 * n);              /* Solver.cpp:387 */
        if (irlsIter == 0) {
            for (long i = 0; i < 3 * n; i++) w2[i] = 1.0f;  /* Solver.cpp:391-392 */
        } else {
            const float reg = P.irlsRegInit * powf(P.irlsRegIter, (float)(irlsIter - 1)); /* :395 */
            gdo_calc_w2(w2, e, reg, 3 * n);
        }
        rz = rzA; rz2 = rzB;                                /* Solver.cpp:401-402 */
        gdo_calc_PTW2x(r, w, h, alpha, w2, e);              /* :403 */
        gdo_calc_xdoty(rz, r, r, n);                        /* :404 */
        memcpy(p, r, sizeof(float) * 3 * n);                /* :405 */

        for (int cgIter = 0;; cgIter++) {
            if (cgIter % P.cgIterCheck == 0 || cgIter == P.cgIterMax) { /* :411 */
                float errL2W;
                if (!P.cgPrecond || cgIter == 0) {
                    errL2W = rz[0] + rz[1] + rz[2];
                } else {
                    gdo_calc_xdoty(rr, r, r, n);
                    errL2W = rr[0] + rr[1] + rr[2];
                }
                if (cgIter == P.cgIterMax || errL2W <= P.cgTolerance) /* :438 */
                    break;
            }
            if (!P.cgPrecond) {                             /* :464-470 */
                float *t = rz; rz = rz2; rz2 = t;
                gdo_calc_Ax_xAx(Ap, pAp, w, h, alpha, w2, p);
                gdo_calc_r_rz(r, rz, Ap, rz2, pAp, n);
                gdo_calc_x_p(x, p, r, rz, rz2, pAp, n);
            } else {                                        /* :474-489 */
                if (cgIter == 0) {
                    gdo_calc_MIx(z, w, h, alpha, w2, r);
                    gdo_calc_xdoty(rz, r, z, n);
                    memcpy(p, z, sizeof(float) * 3 * n);
                }
                float *t = rz; rz = rz2; rz2 = t;
                gdo_calc_Ax_xAx(Ap, pAp, w, h, alpha, w2, p);
                gdo_calc_r_rz(r, rz, Ap, rz2, pAp, n);
                gdo_calc_MIx(z, w, h, alpha, w2, r);
                gdo_calc_xdoty(rz, r, z, n);
                /* NB reference passes m_r (not z) to calc_x_p here, Solver.cpp:488. */
                gdo_calc_x_p(x, p, r, rz, rz2, pAp, n);
            }
            iters++;
        }
    }

    if (x_out) memcpy(x_out, x, sizeof(float) * 3 * n);
    if (!direct) memcpy(rec, x, sizeof(float) * 3 * n);     /* Solver.cpp:561-562 */
    else gdo_calc_axpy(rec, plus1, direct, x, n);           /* :565-566: r=direct; r = 1*r + x */

    free(b); free(e); free(w2); free(x); free(r); free(z); free(p); free(Ap);
    return iters;
}

/* Synthetic solver input of SURVEY.md section 8(d) (the generator the survey's oracle probe
 * used; not reference code).  All fp32; LCG s = s*1664525+1013904223 (u32), u=(s>>8)/2^24. */
/* Solver::evaluateMetricsMTS, Solver.cpp:511-541 (public, no caller in the reference): e = b - P x for the given iterate x; errL1 / errL2 =
 * mean length / squared length over the 3n stacked Vec3f rows, sequential fp32; err = the first n rows of e (the alpha*T block). */
GDO_API void gdo_evaluate_metrics(const float *x, const float *dx, const float *dy, const float *tp, int w, int h, float alpha,
                                  float *err, float *errL1, float *errL2)
{
    const long n = (long)w * h;
    float *b = (float *)malloc(sizeof(float) * 9 * n), *e = (float *)malloc(sizeof(float) * 9 * n);
    const float minus1[3] = {-1.0f, -1.0f, -1.0f};
    if (!tp) alpha = 0.0f;                                          /* Solver.cpp:319 */
    for (long i = 0; i < 3 * n; i++) { b[i] = tp ? tp[i] * alpha : 0.0f; b[3 * n + i] = dx[i]; b[6 * n + i] = dy[i]; }    /* Solver.cpp:323-332 */
    gdo_calc_Px(e, w, h, alpha, x);
    gdo_calc_axpy(e, minus1, e, b, 3 * n);
    float l1 = 0.0f, l2 = 0.0f;
    for (long i = 0; i < 3 * n; i++) {
        const float ex = e[3 * i], ey = e[3 * i + 1], ez = e[3 * i + 2];
        const float sq = ex * ex + ey * ey + ez * ez;               /* lenSqr, Defs.hpp */
        l1 += sqrtf(sq);
        l2 += sq;
    }
    *errL1 = l1 / (float)(n * 3);
    *errL2 = l2 / (float)(n * 3);
    memcpy(err, e, sizeof(float) * 3 * n);
    free(b); free(e);
}

/* GBDPTIntegrator::prepareDataForSolver, src/integrators/gbdpt/gbdpt.cpp:264-280 (G-BDPT's reconstruction stage, BASELINE config 5):
 * the developed Float (double) buffer scaled into the solver's fp32 input; with data2, every entry that has a partner at i + 3*offset
 * becomes the mean of its own gradient and the negated gradient the partner recorded towards it ("merge inverse directions into one
 * buffer").  Mixed float/double arithmetic exactly as the C++ expressions promote it. */
GDO_API void gdo_gbdpt_prepare_data(float w, float *out, const double *data, int len, const double *data2, int offset)
{
    for (int i = 0; i < len; i++)
        out[i] = w * (float)data[i];
    if (data2 != NULL) {
        int io;
        for (int i = 0; i < len; i++) {
            io = i + 3 * offset;
            if (io >= 0 && io < len) {
                out[i] *= 0.5;
                out[i] -= 0.5 * w * (float)data2[io];
            }
        }
    }
}

GDO_API void gdo_synth_inputs(int w, int h, unsigned seed, float *dx, float *dy, float *tp, float *direct)
{
    const long n = (long)w * h;
    float *gt = (float *)malloc(sizeof(float) * 3 * n);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++)
                gt[3 * ((long)y * w + x) + c] = 0.5f + 0.4f * sinf(0.2f * (float)x + (float)c) * cosf(0.15f * (float)y);
    unsigned s = seed;
#define GDO_NEXT_U() (s = s * 1664525u + 1013904223u, (float)(s >> 8) * (1.0f / 16777216.0f))
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) {
                const long i = 3 * ((long)y * w + x) + c;
                float u = GDO_NEXT_U();
                tp[i] = gt[i] + 0.2f * (u - 0.5f);
                if (x < w - 1) { u = GDO_NEXT_U(); dx[i] = gt[i + 3] - gt[i] + 0.01f * (u - 0.5f); } else dx[i] = 0.0f;
                if (y < h - 1) { u = GDO_NEXT_U(); dy[i] = gt[i + 3 * w] - gt[i] + 0.01f * (u - 0.5f); } else dy[i] = 0.0f;
            }
#undef GDO_NEXT_U
    if (direct) memset(direct, 0, sizeof(float) * 3 * n);
    free(gt);
}
