/* oracle/mipmap_oracle.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the MIP map behind `<texture type="bitmap">` as the G-PT path uses it with filterType "trilinear" / "ewa"
 * (the reference's default): the pyramid (TMIPMap's constructor, include/mitsuba/render/mipmap.h:163-304, built by Bitmap::resample =
 * two passes of Resampler<Float>, src/libcore/bitmap.cpp:2230-2330, include/mitsuba/core/rfilter.h:104-468, with the 2-lobed Lanczos
 * filter of src/textures/bitmap.cpp:282-287, src/rfilters/lanczos.cpp:42-54, results clamped to [0, maxValue = 1]) and the filtered
 * lookup (TMIPMap::eval / evalEWA / evalBilinear / evalBox / evalTexel, mipmap.h:503-596,628-712,744-833).  Float = double
 * (DOUBLE_PRECISION build); texels are RGB triples of Float.  PARITY UNPINNED like the rest of the floating-point path.
 */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace mip_oracle {

typedef double Float;
enum { BC_REPEAT = 0, BC_CLAMP = 1, BC_MIRROR = 2, BC_ZERO = 3, BC_ONE = 4 };          // ReconstructionFilter::EBoundaryCondition as the texture's wrap modes
enum { FILTER_NEAREST = 0, FILTER_BILINEAR = 1, FILTER_TRILINEAR = 2, FILTER_EWA = 3 };  // EMIPFilterType
enum { LUT_SIZE = 64 };                                                                 // MTS_MIPMAP_LUT_SIZE, mipmap.h:37

inline int modulo(int a, int b) { const int r = a % b; return r < 0 ? r + b : r; }      // math::modulo
inline int floorToInt(Float v) { return (int)std::floor(v); }
inline int ceilToInt(Float v) { return (int)std::ceil(v); }
inline Float hypot2(Float a, Float b)
{ // math.cpp:89-101
    Float r;
    if (std::abs(a) > std::abs(b)) { r = b / a; r = std::abs(a) * std::sqrt(1.0 + r * r); }
    else if (b != 0.0) { r = a / b; r = std::abs(b) * std::sqrt(1.0 + r * r); }
    else r = 0.0;
    return r;
}
inline Float log2f_(Float v) { const Float invLn2 = 1.0 / std::log(2.0); return std::log(v) * invLn2; }   // math.cpp:108-111 (fastlog == log)

// LanczosSincFilter::eval with lobes = 2, lanczos.cpp:42-54 (Epsilon = 1e-7 in the double build, constants.h:25)
inline Float lanczos2(Float x)
{
    const Float radius = 2.0, Epsilon = 1e-7;
    x = std::abs(x);
    if (x < Epsilon) return 1.0;
    else if (x > radius) return 0.0;
    const Float x1 = M_PI * x, x2 = x1 / radius;
    return (std::sin(x1) * std::sin(x2)) / (x1 * x2);
}

// Resampler<Float> in resampling mode (sourceRes != targetRes), rfilter.h:122-183, and resampleAndClamp, :232-275.  The reference runs
// a boundary-aware loop at both ends and a plain loop in the middle; inside the image `lookup` IS the plain access, so one loop says both.
struct Resampler {
    int bc, sourceRes, targetRes, taps;
    std::vector<int> start;
    std::vector<Float> weights;
    Resampler(int bc_, int sourceRes_, int targetRes_) : bc(bc_), sourceRes(sourceRes_), targetRes(targetRes_)
    {
        Float filterRadius = 2.0, scale = 1.0, invScale = 1.0;
        if (targetRes < sourceRes) { scale = (Float)sourceRes / (Float)targetRes; invScale = 1 / scale; filterRadius *= scale; }
        taps = ceilToInt(filterRadius * 2);
        start.resize(targetRes);
        weights.resize((size_t)taps * targetRes);
        for (int i = 0; i < targetRes; i++) {
            const Float center = (i + (Float)0.5f) / targetRes * sourceRes;
            start[i] = floorToInt(center - filterRadius + (Float)0.5f);
            Float sum = 0;
            for (int j = 0; j < taps; j++) {
                const Float pos = start[i] + j + (Float)0.5f - center;
                const Float weight = lanczos2(pos * invScale);
                weights[(size_t)i * taps + j] = weight;
                sum += weight;
            }
            const Float normalization = 1.0 / sum;
            for (int j = 0; j < taps; j++) weights[(size_t)i * taps + j] = weights[(size_t)i * taps + j] * normalization;
        }
    }
    Float lookup(const Float *source, int pos, size_t stride, int ch) const
    { // rfilter.h:437-458
        if (pos < 0 || pos >= sourceRes) {
            switch (bc) {
                case BC_CLAMP: pos = std::min(std::max(pos, 0), sourceRes - 1); break;
                case BC_REPEAT: pos = modulo(pos, sourceRes); break;
                case BC_MIRROR: pos = modulo(pos, 2 * sourceRes); if (pos >= sourceRes) pos = 2 * sourceRes - pos - 1; break;
                case BC_ZERO: return 0.0;
                default: return 1.0;
            }
        }
        return source[stride * pos + ch];
    }
    // source / target: first sample of the line; strides in SAMPLES (a sample = `channels` values)
    void resampleAndClamp(const Float *source, size_t sourceStride, Float *target, size_t targetStride, int channels, Float lo, Float hi) const
    {
        for (int i = 0; i < targetRes; ++i)
            for (int ch = 0; ch < channels; ++ch) {
                Float result = 0;
                for (int j = 0; j < taps; ++j) result += lookup(source, start[i] + j, sourceStride * channels, ch) * weights[(size_t)i * taps + j];
                target[(size_t)i * targetStride * channels + ch] = std::min(hi, std::max(lo, result));
            }
    }
};

// half(float) -> float of the `half` class the environment map's pyramid is stored in (TMIPMap<Spectrum, SpectrumHalf>, envmap.cpp:101-103):
// round to nearest even on the 10-bit mantissa, gradual underflow, overflow to infinity.  The double texel first becomes a float
// (half's constructor takes one), then a half.
inline Float roundToHalf(Float value)
{
    const float f = (float)value;
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = x & 0x80000000u;
    x &= 0x7fffffffu;
    float r;
    if (x >= 0x7f800000u) { r = f; }                                                     // inf / nan stay
    else if (x >= 0x477ff000u) { const uint32_t inf = sign | 0x7f800000u; std::memcpy(&r, &inf, 4); }   // >= 65520: rounds to infinity
    else if (x < 0x38800000u) {                                                          // below 2^-14: a half denormal = a multiple of 2^-24
        const float q = std::nearbyint(std::fabs(f) * 16777216.0f) * (1.0f / 16777216.0f);   // (round-half-even is the default rounding mode)
        r = sign ? -q : q;
    } else {
        x += 0x00000fffu + ((x >> 13) & 1u);
        x &= 0xffffe000u;
        x |= sign;
        std::memcpy(&r, &x, 4);
    }
    return (Float)r;
}

struct Level { int w = 0, h = 0; std::vector<Float> rgb; Float ratioX = 1, ratioY = 1; };

// Bitmap::resample(rfilter, bcu, bcv, size, 0, maxValue), bitmap.cpp:2230-2330: X pass into a temporary (when the width changes), then Y pass
inline Level resampleImage(const Level &src, int tw, int th, int bcu, int bcv, Float maxValue)
{
    Level out; out.w = tw; out.h = th; out.rgb.assign((size_t)tw * th * 3, 0.0);
    if (src.w == tw && src.h == th) { out.rgb = src.rgb; return out; }
    const Level *source = &src;
    Level temp;
    if (src.w != tw) {
        Resampler r(bcu, src.w, tw);
        Level *dst = &out;
        if (src.h != th) { temp.w = tw; temp.h = src.h; temp.rgb.assign((size_t)tw * src.h * 3, 0.0); dst = &temp; }
        for (int y = 0; y < src.h; ++y) r.resampleAndClamp(&src.rgb[(size_t)y * src.w * 3], 1, &dst->rgb[(size_t)y * tw * 3], 1, 3, 0.0, maxValue);
        source = dst;
    }
    if (source->h != th) {
        Resampler r(bcv, source->h, th);
        for (int x = 0; x < source->w; ++x) r.resampleAndClamp(&source->rgb[(size_t)x * 3], source->w, &out.rgb[(size_t)x * 3], tw, 3, 0.0, maxValue);
    }
    return out;
}

struct MipMap {
    int bcu = 0, bcv = 0, filter = FILTER_EWA;
    Float maxAnisotropy = 20;
    std::vector<Level> pyramid;
    Float weightLut[LUT_SIZE];

    // TMIPMap(bitmap, ...), mipmap.h:163-304 (negative texels are clamped first, :234-242)
    // maxValue: the upper clamp of the resampled levels (1 for bitmap textures, infinity for the environment map, envmap.cpp:178-181);
    // halfStorage: the pyramid keeps its texels as `half` (TMIPMap<Spectrum, SpectrumHalf>) -- each level is resampled from the previous
    // level's FLOAT bitmap and quantized when stored (mipmap.h:226-232,262-264)
    void build(int w, int h, const Float *rgb, int bcu_, int bcv_, int filter_, Float maxAniso, Float maxValue = 1.0, bool halfStorage = false)
    {
        bcu = bcu_; bcv = bcv_; filter = filter_;
        maxAnisotropy = filter == FILTER_EWA ? maxAniso : 1.0;                                // bitmap.cpp:232-235
        pyramid.clear();
        Level cur; cur.w = w; cur.h = h; cur.rgb.assign(rgb, rgb + (size_t)w * h * 3);
        for (Float &v : cur.rgb) if (v < 0) v = 0;
        auto store = [&](const Level &l) { pyramid.push_back(l); if (halfStorage) for (Float &v : pyramid.back().rgb) v = roundToHalf(v); };
        store(cur);
        if (filter != FILTER_NEAREST && filter != FILTER_BILINEAR) {
            int sx = w, sy = h;
            while (sx > 1 || sy > 1) {
                sx = std::max(1, (sx + 1) / 2); sy = std::max(1, (sy + 1) / 2);
                Level next = resampleImage(cur, sx, sy, bcu, bcv, maxValue);
                next.ratioX = (Float)sx / (Float)w; next.ratioY = (Float)sy / (Float)h;
                store(next);
                cur = next;
            }
        }
        // :297-301.  `math::fastexp(-2.0f)` takes the FLOAT overload: (float) exp((double) value) on Linux/x86_64 (math.h:175-187) -- the
        // constant subtracted from the double-precision Gaussian is rounded to single precision
        for (int i = 0; i < LUT_SIZE; ++i) { const Float r2 = (Float)i / (Float)(LUT_SIZE - 1); weightLut[i] = std::exp(-2.0f * r2) - (Float)(float)std::exp(-2.0); }
    }
    int levels() const { return (int)pyramid.size(); }

    void texel(int level, int x, int y, Float out[3]) const
    { // evalTexel, mipmap.h:503-563
        const Level &L = pyramid[level];
        if (x < 0 || x >= L.w) {
            switch (bcu) {
                case BC_REPEAT: x = modulo(x, L.w); break;
                case BC_CLAMP: x = std::min(std::max(x, 0), L.w - 1); break;
                case BC_MIRROR: x = modulo(x, 2 * L.w); if (x >= L.w) x = 2 * L.w - x - 1; break;
                case BC_ZERO: out[0] = out[1] = out[2] = 0.0; return;
                default: out[0] = out[1] = out[2] = 1.0; return;
            }
        }
        if (y < 0 || y >= L.h) {
            switch (bcv) {
                case BC_REPEAT: y = modulo(y, L.h); break;
                case BC_CLAMP: y = std::min(std::max(y, 0), L.h - 1); break;
                case BC_MIRROR: y = modulo(y, 2 * L.h); if (y >= L.h) y = 2 * L.h - y - 1; break;
                case BC_ZERO: out[0] = out[1] = out[2] = 0.0; return;
                default: out[0] = out[1] = out[2] = 1.0; return;
            }
        }
        const Float *t = &L.rgb[((size_t)y * L.w + x) * 3];
        out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
    }
    void evalBox(int level, Float u, Float v, Float out[3]) const
    { // :566-569
        const Level &L = pyramid[level];
        texel(level, floorToInt(u * L.w), floorToInt(v * L.h), out);
    }
    void evalBilinear(int level, Float u_, Float v_, Float out[3]) const
    { // :575-596
        if (!std::isfinite(u_) || !std::isfinite(v_)) { out[0] = out[1] = out[2] = 0.0; return; }
        if (level >= levels()) { evalBox(levels() - 1, u_, v_, out); return; }
        const Level &L = pyramid[level];
        const Float u = u_ * L.w - 0.5f, v = v_ * L.h - 0.5f;
        const int xPos = floorToInt(u), yPos = floorToInt(v);
        const Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
        Float a[3], b[3], c[3], d[3];
        texel(level, xPos, yPos, a); texel(level, xPos, yPos + 1, b); texel(level, xPos + 1, yPos, c); texel(level, xPos + 1, yPos + 1, d);
        for (int k = 0; k < 3; ++k) out[k] = a[k] * dx2 * dy2 + b[k] * dx2 * dy1 + c[k] * dx1 * dy2 + d[k] * dx1 * dy1;
    }
    void evalEWA(int level, Float u_, Float v_, Float A, Float B, Float C, Float out[3]) const
    { // :744-833
        if (!std::isfinite(A + B + C + u_ + v_)) { out[0] = out[1] = out[2] = 0.0; return; }
        if (level >= levels()) { evalBox(levels() - 1, u_, v_, out); return; }
        const Level &L = pyramid[level];
        const Float u = u_ * L.w - 0.5f, v = v_ * L.h - 0.5f;
        A /= L.ratioX * L.ratioX; B /= L.ratioX * L.ratioY; C /= L.ratioY * L.ratioY;
        const Float invDet = 1.0f / (-B * B + 4.0f * A * C), deltaU = 2.0f * std::sqrt(C * invDet), deltaV = 2.0f * std::sqrt(A * invDet);
        const int u0 = ceilToInt(u - deltaU), u1 = floorToInt(u + deltaU), v0 = ceilToInt(v - deltaV), v1 = floorToInt(v + deltaV);
        const Float As = A * LUT_SIZE, Bs = B * LUT_SIZE, Cs = C * LUT_SIZE;
        Float result[3] = {0.0, 0.0, 0.0}, denominator = 0.0f;
        const Float ddq = 2 * As, uu0 = (Float)u0 - u;
        for (int vt = v0; vt <= v1; ++vt) {
            const Float vv = (Float)vt - v;
            Float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv;
            Float dq = As * (2 * uu0 + 1) + Bs * vv;
            for (int ut = u0; ut <= u1; ++ut) {
                if (q < (Float)LUT_SIZE) {
                    const uint32_t qi = (uint32_t)q;
                    if (qi < LUT_SIZE) {
                        const Float weight = weightLut[(int)q];
                        Float t[3];
                        texel(level, ut, vt, t);
                        for (int k = 0; k < 3; ++k) result[k] += t[k] * weight;
                        denominator += weight;
                    }
                }
                q += dq;
                dq += ddq;
            }
        }
        if (denominator == 0) { evalBilinear(level, u_, v_, out); return; }
        for (int k = 0; k < 3; ++k) out[k] = result[k] / denominator;
    }
    // TMIPMap::eval(uv, d0, d1), mipmap.h:628-712
    void eval(Float u, Float v, Float d0x, Float d0y, Float d1x, Float d1y, Float out[3]) const
    {
        if (filter == FILTER_NEAREST) { evalBox(0, u, v, out); return; }
        if (filter == FILTER_BILINEAR) { evalBilinear(0, u, v, out); return; }
        const Level &L0 = pyramid[0];
        const Float du0 = d0x * L0.w, dv0 = d0y * L0.h, du1 = d1x * L0.w, dv1 = d1y * L0.h;
        Float A = dv0 * dv0 + dv1 * dv1, B = -2.0f * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25f;
        const Float root = hypot2(A - C, B), Aprime = 0.5f * (A + C - root), Cprime = 0.5f * (A + C + root);
        Float majorRadius = Aprime != 0 ? std::sqrt(F / Aprime) : 0, minorRadius = Cprime != 0 ? std::sqrt(F / Cprime) : 0;
        if (filter == FILTER_TRILINEAR || !(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
            const Float Epsilon = 1e-7;
            const Float level = log2f_(std::max(majorRadius, Epsilon));
            const int ilevel = floorToInt(level);
            if (ilevel < 0) { evalBilinear(0, u, v, out); return; }
            const Float a = level - ilevel;
            Float p[3], q[3];
            evalBilinear(ilevel, u, v, p); evalBilinear(ilevel + 1, u, v, q);
            for (int k = 0; k < 3; ++k) out[k] = p[k] * (1.0f - a) + q[k] * a;
            return;
        }
        if (minorRadius * maxAnisotropy < majorRadius) {
            minorRadius = majorRadius / maxAnisotropy;
            const Float theta = 0.5f * std::atan(B / (A - C));
            const Float sinTheta = std::sin(theta), cosTheta = std::cos(theta);
            const Float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta,
                        sin2Theta = 2 * sinTheta * cosTheta;
            A = a2 * cosTheta2 + b2 * sinTheta2;
            B = (a2 - b2) * sin2Theta;
            C = a2 * sinTheta2 + b2 * cosTheta2;
            F = a2 * b2;
        }
        const Float scale = 1.0f / F;
        A *= scale; B *= scale; C *= scale;
        const Float level = std::max((Float)0.0f, log2f_(minorRadius));
        const int ilevel = (int)level;
        const Float a = level - ilevel;
        if (majorRadius < 1 || !(A > 0 && C > 0)) { evalBilinear(ilevel, u, v, out); return; }
        Float p[3], q[3];
        evalEWA(ilevel, u, v, A, B, C, p); evalEWA(ilevel + 1, u, v, A, B, C, q);
        for (int k = 0; k < 3; ++k) out[k] = p[k] * (1.0f - a) + q[k] * a;
    }
};

} // namespace mip_oracle
