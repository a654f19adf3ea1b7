/*
 * include/gdpt_poisson.h -- C-ABI of the MI355X screened-Poisson reconstruction.
 *
 * Drop-in boundary for the reference's `poisson::Solver` / `poisson::Backend` pair
 * (/root/reference/src/integrators/poisson_solver/).  Two levels are exported:
 *
 *   (1) solver level  -- what gpt.cpp:1445-1462 calls: Params + preset, importImagesMTS,
 *       setupBackend, solveIndirect, exportImagesMTS.  One handle == one poisson::Solver.
 *   (2) backend-op level -- one entry per `poisson::Backend` virtual (Backend.hpp:66-100), on
 *       device vectors laid out exactly like the reference's (`Vec3f` AoS, stacked [aT;dx;dy]),
 *       so a `class BackendHIP : public poisson::Backend` is a list of one-line forwards and
 *       is selectable where BackendCUDA was (Solver.cpp:264-274).  See INTEGRATION.md.
 *
 * Plain C: pointers, sizes, integer status codes (0 = ok, <0 = error; text via
 * gdpt_last_error()).  No C++ exceptions cross this boundary, no torch types.  All functions
 * are thread-compatible (one handle per thread); kernels run on the handle's own HIP stream
 * unless a stream is passed.  The library needs a gfx950 device: it never falls back to CPU.
 */
#ifndef GDPT_POISSON_H
#define GDPT_POISSON_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDPT_API __attribute__((visibility("default")))

#define GDPT_OK                 0
#define GDPT_ERR_INVALID       -1   /* bad argument / call order (the reference asserts here)      */
#define GDPT_ERR_HIP           -2   /* a HIP runtime call failed (reference: BackendCUDA -> fail()) */
#define GDPT_ERR_NO_DEVICE     -3   /* no gfx950 device visible; there is no CPU fallback          */
#define GDPT_ERR_UNSUPPORTED   -4   /* a reference feature this build does not carry                */

GDPT_API const char *gdpt_last_error(void);

/* ---- (1) solver level --------------------------------------------------------------------- */

/* Solver::Params solver-configuration fields (Solver.hpp:78-93).  `device` mirrors
 * `cudaDevice` (Solver.hpp:79): -1 = current device.  `verbose` as Solver.hpp:80. */
typedef struct gdpt_poisson_params {
    float alpha;
    int   irlsIterMax;
    float irlsRegInit;
    float irlsRegIter;
    int   cgIterMax;
    int   cgIterCheck;
    int   cgPrecond;      /* Solver.cpp:474-489 (calc_MIx); no preset enables it; host-checked loop */
    float cgTolerance;
    int   device;
    int   verbose;
} gdpt_poisson_params;

typedef struct gdpt_poisson_solver gdpt_poisson_solver;

/* Solver::Params::LogFunction (Solver.hpp:95-96), wired to SLog at gpt.cpp:1454. */
typedef void (*gdpt_log_fn)(const char *message, void *user);

/* Params::setDefaults (Solver.cpp:57-88): alpha 0.2, preset "L1D", device -1. */
GDPT_API void gdpt_poisson_params_defaults(gdpt_poisson_params *p);
/* Params::setConfigPreset (Solver.cpp:94-178): "L1D","L1Q","L1L","L2D","L2Q". Returns 1 if known
 * else 0, exactly as the reference's bool. alpha/device/verbose are left untouched. */
GDPT_API int  gdpt_poisson_params_preset(gdpt_poisson_params *p, const char *preset);

/* Solver::Solver(const Params&) (Solver.cpp:196-218); sanitizes like Params::sanitize. */
GDPT_API int  gdpt_poisson_create(const gdpt_poisson_params *p, gdpt_poisson_solver **out);
/* Solver::~Solver (Solver.cpp:232-253). */
GDPT_API void gdpt_poisson_destroy(gdpt_poisson_solver *s);
GDPT_API int  gdpt_poisson_set_log(gdpt_poisson_solver *s, gdpt_log_fn fn, void *user);

/* Solver::importImagesMTS (Solver.cpp:220-228): BORROWS the four HOST pointers (3*w*h floats,
 * row-major RGB); tp and direct may be NULL with the reference's meaning. */
GDPT_API int  gdpt_poisson_import_images(gdpt_poisson_solver *s, const float *dx, const float *dy,
                                const float *tp, const float *direct, int width, int height);
/* Same contract with DEVICE pointers (the tracer hands its film buffers over without a host trip). */
GDPT_API int  gdpt_poisson_import_images_device(gdpt_poisson_solver *s, const float *dx, const float *dy,
                                       const float *tp, const float *direct, int width, int height);
/* Solver::setupBackend (Solver.cpp:257-338): allocates b,e,w2,x,r,p,Ap; b=[alpha*T;dx;dy]; x0=T. */
GDPT_API int  gdpt_poisson_setup_backend(gdpt_poisson_solver *s);
/* Solver::solveIndirect (Solver.cpp:374-509). Returns when the solve has finished on the device. */
GDPT_API int  gdpt_poisson_solve_indirect(gdpt_poisson_solver *s);
/* Asynchronous form: enqueue on the handle's stream and return; pair with gdpt_poisson_sync. */
GDPT_API int  gdpt_poisson_solve_indirect_async(gdpt_poisson_solver *s);
GDPT_API int  gdpt_poisson_sync(gdpt_poisson_solver *s);
/* Solver::exportImagesMTS (Solver.cpp:542-582): rec = direct + x (or x), 3*w*h floats to HOST. */
/* Solver::evaluateMetricsMTS (Solver.cpp:511-541; public, no caller in the reference): residual e = b - P x of the current iterate;
 * errL1 / errL2 = mean length / squared length of its 3n stacked RGB rows (sequential fp32 on the host, as the reference sums the
 * mapped vector), err (3*w*h floats, host) = the rows of the alpha*T block. */
GDPT_API int  gdpt_poisson_evaluate_metrics(gdpt_poisson_solver *s, float *err, float *errL1, float *errL2);
GDPT_API int  gdpt_poisson_export_images(gdpt_poisson_solver *s, float *rec);
GDPT_API int  gdpt_poisson_export_images_device(gdpt_poisson_solver *s, float *rec_device);
/* Device pointer of the current solution x (3*w*h floats), valid until destroy. */
GDPT_API int  gdpt_poisson_solution_device(gdpt_poisson_solver *s, float **x_device);
/* The "Execution time" of Solver.cpp:500 for the last solve (HIP-event time, seconds). */
GDPT_API float gdpt_poisson_last_solve_seconds(const gdpt_poisson_solver *s);
/* Total CG iterations the last solve executed. */
GDPT_API long gdpt_poisson_last_iterations(const gdpt_poisson_solver *s);
/* The HIP stream (hipStream_t as void*) the handle launches on. */
GDPT_API void *gdpt_poisson_stream(gdpt_poisson_solver *s);
/* 0: reference op sequence, 3 kernels per CG iteration; 1: x_p fused into the next iteration's stencil, 2 kernels per CG
 * iteration; 2 (default): the CG loop of an IRLS iteration as ONE persistent cooperative kernel that keeps the iterate in
 * registers -- used when every 64-px-wide tile gets its own CU (up to ~1 Mpixel on 256 CUs), cgTolerance == 0 and not
 * verbose; otherwise, and if its workgroups turn out not to be co-resident, level 1 runs (images of 1-2 Mpixel: the 128-px
 * tiles of kp_cg2).  Same arithmetic per element at levels 0-2; only the dot products' summation tree differs.
 * 3 (opt-in): level 2 with the Chronopoulos-Gear form of the iteration -- r.r and (A r).r in ONE grid-wide reduction, p.Ap
 * from delta - beta gamma / alpha_old -- algebraically the recurrence of Solver.cpp:466-469, different in rounding (the bars of
 * tests/test_poisson_gpu.py); 64-px kernel only, wider images run level 2. */
GDPT_API int  gdpt_poisson_set_fusion(gdpt_poisson_solver *s, int level);

/* Bench hook (no reference counterpart): mean standalone duration in microseconds, by HIP events on the
 * handle's stream, of the CG kernels at the handle's geometry: us[0] stencil, us[1] r_rz, us[2] x_p,
 * us[3] fused x_p+stencil.  Clobbers the iterate; call setup_backend again before the next solve. */
GDPT_API int  gdpt_poisson_profile_kernels(gdpt_poisson_solver *s, int reps, float us[4]);
/* Same for the persistent CG kernel of fusion level 2: mean duration in microseconds of ONE launch (= cgIterMax CG
 * iterations), 0 when the handle's geometry does not use it.  Clobbers the iterate like profile_kernels. */
GDPT_API int  gdpt_poisson_profile_persistent(gdpt_poisson_solver *s, int reps, float *us);
/* Bench hook: a bare streaming kernel with kf_xp_Ax's access mix (3 coalesced 16-byte reads + 3 non-temporal 16-byte writes per float4 element, no stencil,
 * no reuse) over the handle's own CG vectors: what 72 B/px can be moved in at all on this device, measured in the same process as the kernel it is the
 * yardstick of (best of `reps` launches, microseconds).  Clobbers the iterate: call setup_backend again before the next solve. */
GDPT_API int  gdpt_poisson_profile_stream(gdpt_poisson_solver *s, int reps, float *us);

/* --- G-BDPT's reconstruction stage (BASELINE config 5; the sampler of that integrator is not part of this library) ------------------
 * GBDPTIntegrator::prepareDataForSolver, src/integrators/gbdpt/gbdpt.cpp:264-280: out[i] = w * float(data[i]); with data2, every
 * entry with a partner at i + 3*offset becomes 0.5 * out[i] - 0.5 * w * float(data2[i + 3*offset]) (the gradient towards +x / +y merged
 * with the one the neighbour recorded towards -x / -y).  data, data2: developed Float (double) buffers of len = 3*w*h; HOST pointers. */
GDPT_API int  gdpt_gbdpt_prepare_data(float w, float *out, const double *data, int len, const double *data2, int offset);
/* the same on DEVICE pointers of the current device, enqueued on `stream` (NULL: the default stream), asynchronous */
GDPT_API int  gdpt_gbdpt_prepare_data_device(float w, float *out, const double *data, int len, const double *data2, int offset, void *stream);
/* The second half of GBDPTIntegrator::render, gbdpt.cpp:178-247: the three prepareDataForSolver calls (primal; +y with -y at offset
 * `width`; +x with -x at offset 1), then Solver(L2D) and Solver(L1D) on them with no direct image, both with `alpha`.  Inputs: the five
 * developed buffers (HOST, 3*w*h doubles each); outputs recL2 / recL1 (HOST, 3*w*h floats each; either may be NULL to skip that solve).
 * Everything between the upload and the download stays on `device`. */
GDPT_API int  gdpt_gbdpt_reconstruct(const double *primal, const double *gradNegY, const double *gradNegX, const double *gradPosX, const double *gradPosY,
                                     int width, int height, float alpha, int device, float *recL2, float *recL1);

/* the same with every buffer on `device` (five developed double images in, fp32 reconstructions out; either output may be NULL);
 * solveSeconds (may be NULL): the HIP-event spans of the L2D and the L1D solveIndirect.  The call returns with the results complete. */
GDPT_API int  gdpt_gbdpt_reconstruct_device(const double *primal, const double *gradNegY, const double *gradNegX, const double *gradPosX, const double *gradPosY,
                                            int width, int height, float alpha, int device, float *recL2, float *recL1, float solveSeconds[2]);

/* Both calls keep, per (device, size, alpha), the three fp32 solver inputs and one solver per preset from one frame to the next (an
 * integrator renders many frames of one size); this frees them (all devices). */
GDPT_API int  gdpt_gbdpt_reconstruct_release(void);
/* ... and this frees the idle entries of ONE frame size on ONE device (device < 0: the current one) -- what an integrator hands back when it is done, without
 * taking the cached solvers of other integrators of the process with it. */
GDPT_API int  gdpt_gbdpt_reconstruct_release_size(int device, int width, int height);

/* ---- (2) backend-op level ----------------------------------------------------------------- */
/* Device-pointer forms of the `poisson::Backend` virtuals.  `stream` is a hipStream_t (NULL =
 * default stream).  Vectors use the reference layout; sizes in ELEMENTS as in Backend::Vector. */

GDPT_API void *gdpt_backend_alloc(size_t bytes);                                   /* allocVector  Backend.cpp:60   */
GDPT_API void  gdpt_backend_free(void *ptr);                                       /* freeVector   Backend.cpp:78   */
GDPT_API int   gdpt_backend_set(float *x, float y, size_t numFloats, void *stream);            /* set   :104 */
GDPT_API int   gdpt_backend_copy(void *x, const void *y, size_t bytes, void *stream);          /* copy  :119 */
GDPT_API int   gdpt_backend_read(void *host, const void *x, size_t bytes, void *stream);       /* read  :135 */
GDPT_API int   gdpt_backend_write(void *x, const void *host, size_t bytes, void *stream);      /* write :145 */
GDPT_API int   gdpt_backend_calc_Px(float *Px, int w, int h, float alpha, const float *x, void *stream);                         /* :155 */
GDPT_API int   gdpt_backend_calc_PTW2x(float *out, int w, int h, float alpha, const float *w2, const float *x, void *stream);    /* :178 */
GDPT_API int   gdpt_backend_calc_Ax_xAx(float *Ax, float *xAx, int w, int h, float alpha, const float *w2, const float *x, void *stream); /* :209 */
GDPT_API int   gdpt_backend_calc_axpy(float *out, const float a[3], const float *x, const float *y, int numElems, void *stream); /* :246 */
GDPT_API int   gdpt_backend_calc_xdoty(float *xdoty, const float *x, const float *y, int numElems, void *stream);                /* :266 */
GDPT_API int   gdpt_backend_calc_r_rz(float *r, float *rz, const float *Ap, const float *rz2, const float *pAp, int numElems, void *stream); /* :287 */
GDPT_API int   gdpt_backend_calc_x_p(float *x, float *p, const float *r, const float *rz, const float *rz2, const float *pAp, int numElems, void *stream); /* :319 */
GDPT_API int   gdpt_backend_calc_w2(float *w2, const float *e, float reg, int numElems, void *stream);                           /* :354 */
GDPT_API int   gdpt_backend_calc_MIx(float *MIx, int w, int h, float alpha, const float *w2, const float *x, void *stream);      /* :387 */
/* Backend::tonemapSRGB / tonemapLinear (Backend.cpp:442-507, BackendCUDA.cu:564-660): ABGR_8888 display images of a device vector */
GDPT_API int   gdpt_backend_tonemap_srgb(unsigned *out, const float *in, int idx, int numPixels, float scale, float bias, void *stream);
GDPT_API int   gdpt_backend_tonemap_linear(unsigned *out, const float *in, int idx, int numPixels, int numComponents, float scaleMin, float scaleMax, int hasNegative, void *stream);
/* Backend::allocTimer / freeTimer / beginTimer / endTimer (Backend.hpp:95-98): seconds of DEVICE time between begin and end on `stream` */
typedef struct gdpt_backend_timer gdpt_backend_timer;
GDPT_API gdpt_backend_timer *gdpt_backend_timer_alloc(void);
GDPT_API void  gdpt_backend_timer_free(gdpt_backend_timer *t);
GDPT_API int   gdpt_backend_timer_begin(gdpt_backend_timer *t, void *stream);
GDPT_API int   gdpt_backend_timer_end(gdpt_backend_timer *t, void *stream, float *seconds);
GDPT_API int   gdpt_backend_sync(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GDPT_POISSON_H */
