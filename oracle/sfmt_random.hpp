/* oracle/sfmt_random.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of Mitsuba's `Random` (SFMT-19937, /root/reference/src/libcore/random.cpp) -- the only integer arithmetic on the
 * G-PT path -- and of the order in which `mitsuba -p 1` consumes ONE such stream: spiral blocks
 * (src/librender/imageproc.cpp:28-78) x Hilbert-ordered pixels inside a block (include/mitsuba/core/sfcurve.h:34-107,
 * gpt_proc.cpp:86), samples in index order.
 *
 * PINNED: this generator reproduces the `reference[]` table that the reference's own test holds for it
 * (src/tests/test_random.cpp:436-501, `Random(4321)`, 192 x nextULong) bit for bit -- tests/golden/sfmt_reference.json (data only,
 * extracted by tests/golden/make_sfmt_golden.py), checked by tests/test_sfmt_oracle.py.  It is the one piece of the path the
 * reference lets anyone pin; the tracer arithmetic around it stays "parity unpinned" (DESIGN.md).
 *
 * The recursion is the published SFMT one (Saito & Matsumoto, MCQMC 2006) with the 19937 parameter set of random.cpp:70-95,
 * written here on a flat array of 624 32-bit words (the reference's non-SSE path works on 128-bit unions; the word-level
 * formulation below is the same map).
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace sfmt_oracle {

class Random {
public:
    enum { MEXP = 19937, N128 = MEXP / 128 + 1, N32 = N128 * 4, N64 = N128 * 2, POS1 = 122, SL1 = 18, SR1 = 11 };   // random.cpp:70-87 (SL2 = SR2 = 1 byte)

    explicit Random(uint64_t seedValue = 5489ULL) { seed(seedValue); }                    // random.h:113: the default seed
    explicit Random(Random &parent) { seed(parent); }                                     // random.cpp:479-483

    // Random::seed(uint64_t) -> State::init_gen_rand, random.cpp:400-409
    void seed(uint64_t s)
    {
        uint64_t prev = s;
        put64(0, prev);
        for (int i = 1; i < N64; ++i) {
            prev = 6364136223846793005ULL * (prev ^ (prev >> 62)) + (uint64_t)i;
            put64(i, prev);
        }
        idx = N32;
        certify();
    }

    // Random::seed(Random *), random.cpp:519-524: N64 draws of the parent become the key of init_by_array
    void seed(Random &parent)
    {
        uint64_t buf[N64];
        for (int i = 0; i < N64; ++i) buf[i] = parent.nextULong();
        seed(buf, N64);
    }

    // Random::seed(uint64_t *, uint64_t) -> State::init_by_array over the key seen as 32-bit words, random.cpp:531-540, 411-467
    void seed(const uint64_t *key64, uint64_t length)
    {
        std::vector<uint32_t> key((size_t)length * 2);
        for (uint64_t i = 0; i < length; ++i) { key[2 * i] = (uint32_t)key64[i]; key[2 * i + 1] = (uint32_t)(key64[i] >> 32); }   // little-endian reinterpret_cast
        initByArray(key.data(), (int)key.size());
    }

    // State::gen_rand64, random.cpp:285-293
    uint64_t nextULong()
    {
        if (idx >= N32) { regenerate(); idx = 0; }
        const uint64_t r = (uint64_t)w[idx] | ((uint64_t)w[idx + 1] << 32);
        idx += 2;
        return r;
    }

    // Random::nextUInt, random.cpp:586-596: rejection under the next power-of-two mask
    uint32_t nextUInt(uint32_t n)
    {
        uint32_t mask = n;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t r;
        while ((r = (uint32_t)(nextULong() & mask)) >= n) {}
        return r;
    }

    // Random::nextFloat, DOUBLE_PRECISION build (random.cpp:616-626): 52 mantissa bits of a double in [1,2), minus 1
    double nextFloat()
    {
        const uint64_t u = (nextULong() >> 12) | 0x3ff0000000000000ULL;
        double d;
        std::memcpy(&d, &u, sizeof d);
        return d - 1.0;
    }

    // Random::nextFloat of the single-precision build (random.cpp:630-640), for the record of that variant
    float nextFloatSingle()
    {
        const uint32_t u = (uint32_t)((nextULong() & 0xFFFFFFFFULL) >> 9) | 0x3f800000U;
        float f;
        std::memcpy(&f, &u, sizeof f);
        return f - 1.0f;
    }

    // Random::set, random.cpp:526-529
    void set(const Random &o) { std::memcpy(w, o.w, sizeof w); idx = o.idx; }

private:
    uint32_t w[N32];
    int idx;

    void put64(int i, uint64_t v) { w[2 * i] = (uint32_t)v; w[2 * i + 1] = (uint32_t)(v >> 32); }

    // period_certification, random.cpp:318-345, parity vector random.cpp:92-95
    void certify()
    {
        static const uint32_t parity[4] = {0x00000001U, 0x00000000U, 0x00000000U, 0x13c9e684U};
        uint32_t inner = 0;
        for (int i = 0; i < 4; ++i) inner ^= w[i] & parity[i];
        for (int s = 16; s > 0; s >>= 1) inner ^= inner >> s;
        if (inner & 1) return;
        for (int i = 0; i < 4; ++i)
            for (uint32_t bit = 1; bit != 0; bit <<= 1)
                if (bit & parity[i]) { w[i] ^= bit; return; }
    }

    // one step of the recursion (do_recursion, random.cpp:190-206) on 128-bit lanes held as 4 little-endian words:
    // r = a ^ (a << 8 bits, as a 128-bit integer) ^ ((b >> 11 per word) & mask) ^ (c >> 8 bits, 128-bit) ^ (d << 18 per word)
    static void step(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d)
    {
        static const uint32_t msk[4] = {0xdfffffefU, 0xddfecb7fU, 0xbffaffffU, 0xbffffff6U};
        uint32_t x[4], y[4];
        for (int k = 0; k < 4; ++k) {
            x[k] = (a[k] << 8) | (k > 0 ? a[k - 1] >> 24 : 0u);
            y[k] = (c[k] >> 8) | (k < 3 ? c[k + 1] << 24 : 0u);
        }
        uint32_t out[4];
        for (int k = 0; k < 4; ++k) out[k] = a[k] ^ x[k] ^ ((b[k] >> SR1) & msk[k]) ^ y[k] ^ (d[k] << SL1);
        std::memcpy(r, out, sizeof out);
    }

    // gen_rand_all, random.cpp:369-383
    void regenerate()
    {
        const uint32_t *r1 = w + 4 * (N128 - 2), *r2 = w + 4 * (N128 - 1);
        for (int i = 0; i < N128; ++i) {
            const int j = i + POS1 < N128 ? i + POS1 : i + POS1 - N128;
            step(w + 4 * i, w + 4 * i, w + 4 * j, r1, r2);
            r1 = r2;
            r2 = w + 4 * i;
        }
    }

    // State::init_by_array, random.cpp:411-467
    void initByArray(const uint32_t *key, int keyLength)
    {
        auto f1 = [](uint32_t x) { return (x ^ (x >> 27)) * 1664525U; };
        auto f2 = [](uint32_t x) { return (x ^ (x >> 27)) * 1566083941U; };
        const int size = N32, lag = 11, mid = (size - lag) / 2;            // size >= 623
        std::memset(w, 0x8b, sizeof w);
        int count = keyLength + 1 > N32 ? keyLength + 1 : N32;
        uint32_t r = f1(w[0] ^ w[mid] ^ w[N32 - 1]);
        w[mid] += r;
        r += (uint32_t)keyLength;
        w[mid + lag] += r;
        w[0] = r;
        count--;
        int i = 1, j = 0;
        for (; j < count && j < keyLength; ++j) {
            r = f1(w[i] ^ w[(i + mid) % N32] ^ w[(i + N32 - 1) % N32]);
            w[(i + mid) % N32] += r;
            r += key[j] + (uint32_t)i;
            w[(i + mid + lag) % N32] += r;
            w[i] = r;
            i = (i + 1) % N32;
        }
        for (; j < count; ++j) {
            r = f1(w[i] ^ w[(i + mid) % N32] ^ w[(i + N32 - 1) % N32]);
            w[(i + mid) % N32] += r;
            r += (uint32_t)i;
            w[(i + mid + lag) % N32] += r;
            w[i] = r;
            i = (i + 1) % N32;
        }
        for (j = 0; j < N32; ++j) {
            r = f2(w[i] + w[(i + mid) % N32] + w[(i + N32 - 1) % N32]);
            w[(i + mid) % N32] ^= r;
            r -= (uint32_t)i;
            w[(i + mid + lag) % N32] ^= r;
            w[i] = r;
            i = (i + 1) % N32;
        }
        idx = N32;
        certify();
    }
};

struct Block { int x, y, w, h; };

// BlockedImageProcess::init + generateWork, imageproc.cpp:28-78: blocks spiral outward from the centre block
inline std::vector<Block> spiralBlocks(int width, int height, int blockSize)
{
    enum { ERight = 0, EDown, ELeft, EUp };
    const int nbx = (int)std::ceil((double)width / (double)blockSize), nby = (int)std::ceil((double)height / (double)blockSize);
    const int total = nbx * nby;
    std::vector<Block> out;
    int cx = nbx / 2, cy = nby / 2, direction = ERight, stepsLeft = 1, numSteps = 1;
    while ((int)out.size() < total) {
        const int px = cx * blockSize, py = cy * blockSize;
        out.push_back(Block{px, py, std::min(width - px, blockSize), std::min(height - py, blockSize)});
        if ((int)out.size() == total) break;
        do {
            switch (direction) {
                case ERight: ++cx; break;
                case EDown: ++cy; break;
                case ELeft: --cx; break;
                default: --cy; break;
            }
            if (--stepsLeft == 0) {
                direction = (direction + 1) % 4;
                if (direction == ELeft || direction == ERight) ++numSteps;
                stepsLeft = numSteps;
            }
        } while (cx < 0 || cy < 0 || cx >= nbx || cy >= nby);
    }
    return out;
}

// HilbertCurve2D<uint8_t>::initialize + generate, sfcurve.h:52-103 (positions are uint8_t and wrap as the reference's do)
struct HilbertPoints {
    std::vector<std::pair<uint8_t, uint8_t>> pts;
    uint8_t sx = 0, sy = 0, px = 0, py = 0;
    enum { N = 0, E, S, W };
    void move(int dir)
    {
        switch (dir) { case N: py--; break; case E: px++; break; case S: py++; break; default: px--; break; }
    }
    void gen(int order, int front, int right, int back, int left)
    {
        if (order == 0) { if (px < sx && py < sy) pts.emplace_back(px, py); return; }
        gen(order - 1, left, back, right, front); move(right);
        gen(order - 1, front, right, back, left); move(back);
        gen(order - 1, front, right, back, left); move(left);
        gen(order - 1, right, front, left, back);
    }
    void initialize(int w, int h)
    {
        if ((uint8_t)w == sx && (uint8_t)h == sy) return;                     // sfcurve.h:53-54: kept from the previous block of the same size
        pts.clear();
        sx = (uint8_t)w; sy = (uint8_t)h; px = py = 0;
        const double invLog2 = 1.0 / std::log(2.0);                          // math::fastlog == ::log in the double build (math.h:197-199)
        gen((int)std::ceil(invLog2 * std::log((double)std::max(sx, sy))), N, E, S, W);
    }
};

} // namespace sfmt_oracle
