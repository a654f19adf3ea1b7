// gpt_serial_capi.hip -- the serial form of the G-PT sampler (gpt_serial.hip.h): SFMT-19937 as Mitsuba's `Random` seeds and steps it, the order in
// which a one-worker render visits the film, and the one-lane kernel that renders in that order from that stream.
//
// What `mitsuba -p 1` does (the chain this file follows): worker 0's sampler is a clone of the scene's IndependentSampler (renderjob.cpp:59-66,
// independent.cpp:58,71-80): its Random is seeded by init_by_array from 312 64-bit draws of the parent Random (random.cpp:519-524), whose own state
// comes from the default seed 5489 (random.h:113, random.cpp:400-409).  Work units are BlockedImageProcess's blocks, spiralling out of the centre
// (imageproc.cpp:28-78); inside a block GPTBlockRenderer::process visits the pixels along a Hilbert curve (gpt_proc.cpp:84-87, sfcurve.h:34-107) and
// renderBlock the samples of a pixel in index order, drawing film position, aperture, time, then the path (gpt.cpp:1245-1268).  Blocks merge into the
// film by addition (gpt_proc.cpp:137-149).
#define GDPT_SERIAL_STREAM
#define GDPT_RENDER_DEVICE_FUNCTIONS_ONLY
#include "gpt_render.hip.h"
#include "gpt_serial.hip.h"
#include "gpt_scene.hip.h"
#include "../../include/gdpt_tracer.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace gdpt_tr {

// ---- SFMT-19937 (Saito & Matsumoto 2006; parameter set of random.cpp:70-95): 156 lanes of 128 bits, here four little-endian 32-bit words each --------
constexpr int SF_LANES = 156, SF_WORDS = 4 * SF_LANES, SF_POS = 122, SF_SL = 18, SF_SR = 11;
// lane <- a ^ (a << 8 as a 128-bit number) ^ ((b >> 11 word by word) & mask) ^ (c >> 8 as a 128-bit number) ^ (d << 18 word by word)   (do_recursion, random.cpp:190-206)
__host__ __device__ inline void sf_mix(uint32_t *out, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d)
{
    const uint32_t mask[4] = {0xdfffffefu, 0xddfecb7fu, 0xbffaffffu, 0xbffffff6u};
    uint32_t r[4];
    for (int k = 0; k < 4; k++) {
        const uint32_t up = (a[k] << 8) | (k ? a[k - 1] >> 24 : 0u);
        const uint32_t down = (c[k] >> 8) | (k < 3 ? c[k + 1] << 24 : 0u);
        r[k] = a[k] ^ up ^ ((b[k] >> SF_SR) & mask[k]) ^ down ^ (d[k] << SF_SL);
    }
    for (int k = 0; k < 4; k++) out[k] = r[k];
}
// the whole state one generation on (gen_rand_all, random.cpp:369-383): lane i from itself, lane i + 122 and the two lanes written last
__host__ __device__ inline void sf_generation(uint32_t *w)
{
    int c = SF_LANES - 2, d = SF_LANES - 1;
    for (int i = 0; i < SF_LANES; i++) {
        const int b = i + SF_POS < SF_LANES ? i + SF_POS : i + SF_POS - SF_LANES;
        sf_mix(w + 4 * i, w + 4 * i, w + 4 * b, w + 4 * c, w + 4 * d);
        c = d; d = i;
    }
}
__device__ void Rng::regenerate() { sf_generation(w); }

struct HostRandom {                       // Mitsuba's Random on the host: only what seeding the worker's stream needs
    uint32_t w[SF_WORDS];
    int at = SF_WORDS;
    // period certification (random.cpp:318-345): the parity vector's inner product with the first lane must be odd
    void certify()
    {
        const uint32_t parity[4] = {1u, 0u, 0u, 0x13c9e684u};
        uint32_t p = 0;
        for (int k = 0; k < 4; k++) p ^= w[k] & parity[k];
        p ^= p >> 16; p ^= p >> 8; p ^= p >> 4; p ^= p >> 2; p ^= p >> 1;
        if (p & 1u) return;
        for (int k = 0; k < 4; k++)
            for (int bit = 0; bit < 32; bit++)
                if (parity[k] >> bit & 1u) { w[k] ^= 1u << bit; return; }
    }
    // Random::seed(uint64_t) -> State::init_gen_rand (random.cpp:400-409): a 64-bit recurrence over the state seen as 312 64-bit words
    explicit HostRandom(uint64_t seed)
    {
        uint64_t v = seed;
        for (int i = 0; i < SF_WORDS / 2; i++) {
            if (i) v = 6364136223846793005ULL * (v ^ (v >> 62)) + (uint64_t)i;
            w[2 * i] = (uint32_t)v; w[2 * i + 1] = (uint32_t)(v >> 32);
        }
        certify();
    }
    uint64_t next64()
    {
        if (at >= SF_WORDS) { sf_generation(w); at = 0; }
        const uint64_t r = (uint64_t)w[at] | ((uint64_t)w[at + 1] << 32);
        at += 2;
        return r;
    }
    // Random::seed(Random *) (random.cpp:519-524) -> State::init_by_array (random.cpp:411-467) with the parent's next 312 draws, seen as 624 words, as the key
    explicit HostRandom(HostRandom &parent)
    {
        std::vector<uint32_t> key;
        for (int i = 0; i < SF_WORDS / 2; i++) { const uint64_t v = parent.next64(); key.push_back((uint32_t)v); key.push_back((uint32_t)(v >> 32)); }
        const int n = SF_WORDS, lag = 11, mid = (n - lag) / 2, len = (int)key.size();
        auto scramble1 = [](uint32_t x) { return (x ^ (x >> 27)) * 1664525u; };
        auto scramble2 = [](uint32_t x) { return (x ^ (x >> 27)) * 1566083941u; };
        std::memset(w, 0x8b, sizeof w);
        uint32_t r = scramble1(w[0] ^ w[mid] ^ w[n - 1]);
        w[mid] += r;
        r += (uint32_t)len;
        w[mid + lag] += r;
        w[0] = r;
        const int rounds = std::max(len + 1, n) - 1;
        int i = 1;
        for (int j = 0; j < rounds; j++) {                                  // the key goes in first, the rest of the rounds add the position only
            r = scramble1(w[i] ^ w[(i + mid) % n] ^ w[(i + n - 1) % n]);
            w[(i + mid) % n] += r;
            r += (j < len ? key[j] : 0u) + (uint32_t)i;
            w[(i + mid + lag) % n] += r;
            w[i] = r;
            i = (i + 1) % n;
        }
        for (int j = 0; j < n; j++) {
            r = scramble2(w[i] + w[(i + mid) % n] + w[(i + n - 1) % n]);
            w[(i + mid) % n] ^= r;
            r -= (uint32_t)i;
            w[(i + mid + lag) % n] ^= r;
            w[i] = r;
            i = (i + 1) % n;
        }
        at = SF_WORDS;
        certify();
    }
};

// ---- the order of the pixels ------------------------------------------------------------------------------------------------------------------------
// HilbertCurve2D<uint8_t> (sfcurve.h:52-103): an L-system walk over a 2^order square that keeps the points inside the block; coordinates are bytes
// and wrap as the reference's do.  The curve object lives as long as the worker: a block of the size of the previous one reuses its points (:53-54).
struct BlockCurve {
    std::vector<std::pair<int, int>> points;
    uint8_t w = 0, h = 0, x = 0, y = 0;
    void step(int heading) { if (heading == 0) y--; else if (heading == 1) x++; else if (heading == 2) y++; else x--; }     // north, east, south, west
    void walk(int order, int front, int right, int back, int left)
    {
        if (order == 0) { if (x < w && y < h) points.emplace_back((int)x, (int)y); return; }
        walk(order - 1, left, back, right, front); step(right);
        walk(order - 1, front, right, back, left); step(back);
        walk(order - 1, front, right, back, left); step(left);
        walk(order - 1, right, front, left, back);
    }
    void set_size(int bw, int bh)
    {
        if ((uint8_t)bw == w && (uint8_t)bh == h) return;
        points.clear();
        w = (uint8_t)bw; h = (uint8_t)bh; x = y = 0;
        walk((int)std::ceil(std::log((double)std::max(w, h)) * (1.0 / std::log(2.0))), 0, 1, 2, 3);   // (math::fastlog is ::log in the double build, math.h:197-199)
    }
};
// every pixel of a width x height film in the order a single worker renders them: blocks in BlockedImageProcess's spiral (imageproc.cpp:28-78: from the
// centre block to the right, then down, left, up, one more step every second turn; positions outside the grid are passed over), the block's pixels along its curve
std::vector<uint2> serial_pixel_order(int width, int height, int blockSize)
{
    const int nx = (int)std::ceil((double)width / blockSize), ny = (int)std::ceil((double)height / blockSize);
    std::vector<uint2> order;
    order.reserve((size_t)width * height);
    BlockCurve curve;
    int bx = nx / 2, by = ny / 2, heading = 0, run = 1, left = 1;
    for (int emitted = 0; emitted < nx * ny;) {
        if (bx >= 0 && by >= 0 && bx < nx && by < ny) {
            const int x0 = bx * blockSize, y0 = by * blockSize;
            curve.set_size(std::min(blockSize, width - x0), std::min(blockSize, height - y0));
            for (const auto &p : curve.points) order.push_back(make_uint2((unsigned)(x0 + p.first), (unsigned)(y0 + p.second)));
            emitted++;
        }
        if (heading == 0) bx++; else if (heading == 1) by++; else if (heading == 2) bx--; else by--;
        if (--left == 0) {
            heading = (heading + 1) % 4;
            if (heading == 0 || heading == 2) run++;
            left = run;
        }
    }
    return order;
}

// One lane, the whole film: per pixel of `pixels` and sample index, the sampler's general form (every feature tested at run time, as k_eval_point's)
// on the HBM tables, the sums into the pixel records.
__global__ __launch_bounds__(TBLK) void k_render_serial(SceneD S, ConfigD cfg, FilmD F, const uint2 *__restrict__ pixels, int numPixels, const uint32_t *__restrict__ state, unsigned long long *__restrict__ draws)
{
    __shared__ int s_stack[STACK_DEPTH * TBLK];
    __shared__ uint32_t s_rng[SF_WORDS + 2];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < SF_WORDS; i++) s_rng[i] = state[i];
    s_rng[SF_WORDS] = SF_WORDS; s_rng[SF_WORDS + 1] = 0;
    SceneView sv;
    sv.nodes = S.nodes; sv.isect = S.isect; sv.shade = S.shade; sv.mats = S.mats; sv.emitters = S.emitters; sv.emTris = S.emTris; sv.emCdf = S.emCdf; sv.emitterCdf = S.emitterCdf; sv.rootRef = S.rootRef; sv.boundM = S.boundM; sv.quant = S.quantNodes; sv.leafExit = 1; sv.vn = S.vn; sv.uv = S.uv; sv.hasUV = S.hasUV; sv.tex = S.tex;
    const FilterD flt = box_filter();
    Lane L;
    L.rng.w = s_rng;
    L.nClosest = L.nShadow = 0;
    Acc<false> A;
    unsigned long long paths = 0, pathLen = 0, total = 0;
    for (int p = 0; p < numPixels; p++) {
        const int px = (int)pixels[p].x, py = (int)pixels[p].y;
        for (int j = 0; j < cfg.spp; j++) {
            if (__hip_atomic_load(F.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { p = numPixels; break; }
            bool active = start_path<true, true, true>(S, sv, cfg, s_stack, L, A, px, py, j);
            const InlineTracer tr = {sv, s_stack};
            while (active) active = bounce<true, true, false, false, false>(S, sv, cfg, tr, L, A);
            finish_path(F, flt, L.sx, L.sy, A, px, py, j);
            paths++; pathLen += L.depth;
            total += s_rng[SF_WORDS + 1]; s_rng[SF_WORDS + 1] = 0;
        }
    }
    atomicAdd(&F.stats[0], (unsigned long long)L.nClosest);
    atomicAdd(&F.stats[1], (unsigned long long)L.nShadow);
    atomicAdd(&F.stats[2], paths);
    atomicAdd(&F.stats[3], pathLen);
    if (draws) *draws = total;
}

int serial_render(const gdpt_scene *s, hipStream_t stream, const ConfigD &cfg, const FilmD &fd, int blockSize, unsigned long long parentSeed, unsigned long long *draws)
{
    HostRandom parent(parentSeed);
    const HostRandom worker(parent);
    const std::vector<uint2> order = serial_pixel_order(fd.W, fd.y1 - fd.y0, blockSize);
    uint2 *dOrder = nullptr; uint32_t *dState = nullptr; unsigned long long *dDraws = nullptr;
    hipError_t e = hipMalloc((void **)&dOrder, order.size() * sizeof(uint2));
    if (e == hipSuccess) e = hipMalloc((void **)&dState, sizeof worker.w);
    if (e == hipSuccess) e = hipMalloc((void **)&dDraws, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpyAsync(dOrder, order.data(), order.size() * sizeof(uint2), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dState, worker.w, sizeof worker.w, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_render_serial, dim3(1), dim3(TBLK), 0, stream, s->d, cfg, fd, dOrder, (int)order.size(), dState, dDraws);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);                 // (the host vectors above and the temporaries below end with this call)
    unsigned long long n = 0;
    if (e == hipSuccess) e = hipMemcpy(&n, dDraws, sizeof n, hipMemcpyDeviceToHost);
    if (draws) *draws = n;
    if (dOrder) (void)hipFree(dOrder);
    if (dState) (void)hipFree(dState);
    if (dDraws) (void)hipFree(dDraws);
    return e == hipSuccess ? 0 : (int)e;
}

} // namespace gdpt_tr

extern "C" int gdpt_internal_fail(int code, const char *msg);

extern "C" {

int gdpt_serial_random(unsigned long long seed, int cloned, int n, unsigned long long *out)
{
    if (n < 0 || (n > 0 && !out)) return gdpt_internal_fail(GDPT_ERR_INVALID, "serial_random: bad argument");
    gdpt_tr::HostRandom parent(seed);
    if (!cloned) { for (int i = 0; i < n; i++) out[i] = parent.next64(); return GDPT_OK; }
    gdpt_tr::HostRandom worker(parent);
    for (int i = 0; i < n; i++) out[i] = worker.next64();
    return GDPT_OK;
}

int gdpt_serial_pixel_order(int width, int height, int blockSize, int *xy)
{
    if (width <= 0 || height <= 0 || blockSize <= 0 || blockSize > 255 || !xy) return gdpt_internal_fail(GDPT_ERR_INVALID, "serial_pixel_order: bad argument (block sizes up to 255: the curve's coordinates are bytes)");
    const std::vector<uint2> order = gdpt_tr::serial_pixel_order(width, height, blockSize);
    for (size_t i = 0; i < order.size(); i++) { xy[2 * i] = (int)order[i].x; xy[2 * i + 1] = (int)order[i].y; }
    return GDPT_OK;
}

} // extern "C"
