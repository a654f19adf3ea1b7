// BackendHIP.cpp -- see BackendHIP.hpp.  In Solver::setupBackend (Solver.cpp:264-274), beside the CUDA branch:
//     if (!m_backend && (m_params.backend == "HIP" || m_params.backend == "Auto")) { log("Using HIP backend\n"); m_backend = new BackendHIP(m_params.cudaDevice); }
// This keeps the reference's own solveIndirect loop (three ops per CG iteration, the map() of rz at each check); the fused solver-level
// route is gdpt_poisson_* (INTEGRATION.md section 1).
#include "BackendHIP.hpp"
#include "gdpt_poisson.h"
#include <stdlib.h>
using namespace poisson;

static void ok(int rc) { if (rc) fail("HIP backend: %s", gdpt_last_error()); }      // Defs.cpp:36-45, as BackendCUDA::checkError
static float* F(Backend::Vector* v) { return (float*)v->ptr; }

BackendHIP::BackendHIP(int) {}
Backend::Vector* BackendHIP::allocVector(int n, size_t bpe)
{
    Vector* x = new Vector; x->numElems = n; x->bytesPerElem = bpe; x->bytesTotal = n * bpe;
    x->ptr = gdpt_backend_alloc(x->bytesTotal);
    if (!x->ptr) fail("Out of memory!");
    return x;
}
void  BackendHIP::freeVector(Vector* x) { if (x) gdpt_backend_free(x->ptr); delete x; }
void* BackendHIP::map(Vector* x) { void* h = malloc(x->bytesTotal); ok(gdpt_backend_read(h, x->ptr, x->bytesTotal, 0)); return h; }   // BackendCUDA.cu:154-175
void  BackendHIP::unmap(Vector* x, void* h, bool modified) { if (modified) ok(gdpt_backend_write(x->ptr, h, x->bytesTotal, 0)); free(h); }
void  BackendHIP::set(Vector* x, float y) { ok(gdpt_backend_set(F(x), y, x->bytesTotal / sizeof(float), 0)); }
void  BackendHIP::copy(Vector* x, Vector* y) { ok(gdpt_backend_copy(x->ptr, y->ptr, x->bytesTotal, 0)); }
void  BackendHIP::read(void* p, Vector* x) { ok(gdpt_backend_read(p, x->ptr, x->bytesTotal, 0)); }
void  BackendHIP::write(Vector* x, const void* p) { ok(gdpt_backend_write(x->ptr, p, x->bytesTotal, 0)); }
void  BackendHIP::calc_Px(Vector* Px, PoissonMatrix P, Vector* x) { ok(gdpt_backend_calc_Px(F(Px), P.size.x, P.size.y, P.alpha, F(x), 0)); }
void  BackendHIP::calc_PTW2x(Vector* o, PoissonMatrix P, Vector* w2, Vector* x) { ok(gdpt_backend_calc_PTW2x(F(o), P.size.x, P.size.y, P.alpha, F(w2), F(x), 0)); }
void  BackendHIP::calc_Ax_xAx(Vector* Ax, Vector* xAx, PoissonMatrix P, Vector* w2, Vector* x) { ok(gdpt_backend_calc_Ax_xAx(F(Ax), F(xAx), P.size.x, P.size.y, P.alpha, F(w2), F(x), 0)); }
void  BackendHIP::calc_axpy(Vector* o, Vec3f a, Vector* x, Vector* y) { float a3[3] = {a.x, a.y, a.z}; ok(gdpt_backend_calc_axpy(F(o), a3, F(x), F(y), x->numElems, 0)); }
void  BackendHIP::calc_xdoty(Vector* o, Vector* x, Vector* y) { ok(gdpt_backend_calc_xdoty(F(o), F(x), F(y), x->numElems, 0)); }
void  BackendHIP::calc_r_rz(Vector* r, Vector* rz, Vector* Ap, Vector* rz2, Vector* pAp) { ok(gdpt_backend_calc_r_rz(F(r), F(rz), F(Ap), F(rz2), F(pAp), r->numElems, 0)); }
void  BackendHIP::calc_x_p(Vector* x, Vector* p, Vector* r, Vector* rz, Vector* rz2, Vector* pAp) { ok(gdpt_backend_calc_x_p(F(x), F(p), F(r), F(rz), F(rz2), F(pAp), x->numElems, 0)); }
void  BackendHIP::calc_w2(Vector* w2, Vector* e, float reg) { ok(gdpt_backend_calc_w2(F(w2), F(e), reg, w2->numElems, 0)); }
void  BackendHIP::calc_MIx(Vector* o, PoissonMatrix P, Vector* w2, Vector* x) { ok(gdpt_backend_calc_MIx(F(o), P.size.x, P.size.y, P.alpha, F(w2), F(x), 0)); }

// Backend.cpp:442-507 (the display path of the solver's debug output, Solver.cpp:516-560), on the device: `in` holds a stack of images of out->numElems pixels each, `idx`
// picks one; `out` gets one packed ABGR word per pixel.
void  BackendHIP::tonemapSRGB(Vector* out, Vector* in, int idx, float scale, float bias) { ok(gdpt_backend_tonemap_srgb((unsigned*)out->ptr, F(in), idx, out->numElems, scale, bias, 0)); }
void  BackendHIP::tonemapLinear(Vector* out, Vector* in, int idx, float scaleMin, float scaleMax, bool hasNegative)
{
    ok(gdpt_backend_tonemap_linear((unsigned*)out->ptr, F(in), idx, out->numElems, (int)(in->bytesPerElem / sizeof(float)), scaleMin, scaleMax, hasNegative ? 1 : 0, 0));
}

// BackendCUDA.cu:672-722: a pair of events; endTimer returns the seconds of DEVICE time between them (the base class measures host ticks)
struct TimerHIP : public Backend::Timer { gdpt_backend_timer* t; };
Backend::Timer* BackendHIP::allocTimer(void) { TimerHIP* t = new TimerHIP; t->beginTicks = 0; t->t = gdpt_backend_timer_alloc(); if (!t->t) fail("HIP backend: %s", gdpt_last_error()); return t; }
void  BackendHIP::freeTimer(Timer* timer) { TimerHIP* t = (TimerHIP*)timer; if (t) gdpt_backend_timer_free(t->t); delete t; }
void  BackendHIP::beginTimer(Timer* timer) { ok(gdpt_backend_timer_begin(((TimerHIP*)timer)->t, 0)); }
float BackendHIP::endTimer(Timer* timer) { float s = 0.0f; ok(gdpt_backend_timer_end(((TimerHIP*)timer)->t, 0, &s)); return s; }
