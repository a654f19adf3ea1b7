// tests/mitsuba_mock/Backend.hpp -- COMPILE-ONLY HARNESS (see mock_mitsuba.h): the declarations of poisson::Backend that BackendHIP.{hpp,cpp} override,
// with the signatures of /root/reference/src/integrators/poisson_solver/Backend.hpp:41-100 and Defs.hpp (Vec2i, Vec3f, fail).  Not the reference's file.
#pragma once
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
namespace poisson
{
struct Vec2i { int x, y; };
struct Vec3f { float x, y, z; };
inline void fail(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); exit(1); }     // Defs.cpp:36-45
class Backend
{
public:
    struct Vector { int numElems; size_t bytesPerElem; size_t bytesTotal; void *ptr; };
    struct PoissonMatrix { Vec2i size; float alpha; };
    struct Timer { long long beginTicks; };
    Backend(void) {}
    virtual ~Backend(void) {}
    virtual Vector* allocVector(int numElems, size_t bytesPerElem) = 0;
    virtual void    freeVector(Vector* x) = 0;
    virtual void*   map(Vector* x) = 0;
    virtual void    unmap(Vector* x, void* ptr, bool modified) = 0;
    virtual void    set(Vector* x, float y) = 0;
    virtual void    copy(Vector* x, Vector* y) = 0;
    virtual void    read(void* ptr, Vector* x) = 0;
    virtual void    write(Vector* x, const void* ptr) = 0;
    virtual void    calc_Px(Vector* Px, PoissonMatrix P, Vector* x) = 0;
    virtual void    calc_PTW2x(Vector* PTW2x, PoissonMatrix P, Vector* w2, Vector* x) = 0;
    virtual void    calc_Ax_xAx(Vector* Ax, Vector* xAx, PoissonMatrix P, Vector* w2, Vector* x) = 0;
    virtual void    calc_axpy(Vector* axpy, Vec3f a, Vector* x, Vector* y) = 0;
    virtual void    calc_xdoty(Vector* xdoty, Vector* x, Vector* y) = 0;
    virtual void    calc_r_rz(Vector* r, Vector* rz, Vector* Ap, Vector* rz2, Vector* pAp) = 0;
    virtual void    calc_x_p(Vector* x, Vector* p, Vector* r, Vector* rz, Vector* rz2, Vector* pAp) = 0;
    virtual void    calc_w2(Vector* w2, Vector* e, float reg) = 0;
    virtual void    calc_MIx(Vector* MIx, PoissonMatrix P, Vector* w2, Vector* x) = 0;
    virtual void    tonemapSRGB(Vector* out, Vector* in, int idx, float scale, float bias) = 0;
    virtual void    tonemapLinear(Vector* out, Vector* in, int idx, float scaleMin, float scaleMax, bool hasNegative) = 0;
    virtual Timer*  allocTimer(void) = 0;
    virtual void    freeTimer(Timer* timer) = 0;
    virtual void    beginTimer(Timer* timer) = 0;
    virtual float   endTimer(Timer* timer) = 0;
};
}
