// BackendHIP.hpp -- poisson::Backend over the gfx950 C-ABI (include/gdpt_poisson.h, backend-op level).
// Place next to /src/integrators/poisson_solver/BackendCUDA.hpp; one override per virtual of Backend.hpp:66-100.
#pragma once
#include "Backend.hpp"

namespace poisson
{
class BackendHIP : public Backend
{
public:
    explicit BackendHIP(int device = -1);
    virtual ~BackendHIP(void) {}
    virtual Vector* allocVector(int numElems, size_t bytesPerElem);
    virtual void    freeVector(Vector* x);
    virtual void*   map(Vector* x);
    virtual void    unmap(Vector* x, void* ptr, bool modified);
    virtual void    set(Vector* x, float y);
    virtual void    copy(Vector* x, Vector* y);
    virtual void    read(void* ptr, Vector* x);
    virtual void    write(Vector* x, const void* ptr);
    virtual void    calc_Px(Vector* Px, PoissonMatrix P, Vector* x);
    virtual void    calc_PTW2x(Vector* PTW2x, PoissonMatrix P, Vector* w2, Vector* x);
    virtual void    calc_Ax_xAx(Vector* Ax, Vector* xAx, PoissonMatrix P, Vector* w2, Vector* x);
    virtual void    calc_axpy(Vector* axpy, Vec3f a, Vector* x, Vector* y);
    virtual void    calc_xdoty(Vector* xdoty, Vector* x, Vector* y);
    virtual void    calc_r_rz(Vector* r, Vector* rz, Vector* Ap, Vector* rz2, Vector* pAp);
    virtual void    calc_x_p(Vector* x, Vector* p, Vector* r, Vector* rz, Vector* rz2, Vector* pAp);
    virtual void    calc_w2(Vector* w2, Vector* e, float reg);
    virtual void    calc_MIx(Vector* MIx, PoissonMatrix P, Vector* w2, Vector* x);
    virtual void    tonemapSRGB(Vector* out, Vector* in, int idx, float scale, float bias);
    virtual void    tonemapLinear(Vector* out, Vector* in, int idx, float scaleMin, float scaleMax, bool hasNegative);
    virtual Timer*  allocTimer(void);
    virtual void    freeTimer(Timer* timer);
    virtual void    beginTimer(Timer* timer);
    virtual float   endTimer(Timer* timer);
};
}
