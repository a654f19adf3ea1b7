// gdpt_host.hpp -- C++ host side above the C-ABI (include/gdpt_poisson.h, include/gdpt_tracer.h), in the reference's own
// language and with the reference's class / method / property names, so that code written against
//   poisson::Solver               (/root/reference/src/integrators/poisson_solver/Solver.hpp:43-157)
//   GradientPathIntegrator        (/root/reference/src/integrators/gpt/gpt.h:71-113, gpt.cpp:1190-1480)
//   MultiFilm                     (/root/reference/src/films/multifilm.cpp: setBuffers / developMulti / develop)
// reads the same here.  Everything numerical happens behind the C-ABI on the GPU; this header only marshals.
// Errors: the reference's Log(EError, ...) throws std::runtime_error (src/libcore/logger.cpp:147) -- so does this.
#pragma once
#include "../../include/gdpt_tracer.h"
#include "exr_writer.hpp"
#include <atomic>
#include <cmath>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace gdpt {

inline std::string format(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return buf;
}

[[noreturn]] inline void logError(const std::string &msg) { throw std::runtime_error(msg); }   // Log(EError, ...)

inline void check(int rc)
{
    if (rc != GDPT_OK) logError(format("gdpt error %d: %s", rc, gdpt_last_error()));
}

/// Properties (include/mitsuba/core/properties.h): typed name -> value map with defaulted getters.
class Properties {
public:
    Properties() {}
    explicit Properties(const std::string &pluginName) : m_plugin(pluginName) {}
    const std::string &getPluginName() const { return m_plugin; }
    void setPluginName(const std::string &n) { m_plugin = n; }
    const std::string &getID() const { return m_id; }
    void setID(const std::string &id) { m_id = id; }
    bool hasProperty(const std::string &n) const { return m_values.count(n) != 0; }
    void setString(const std::string &n, const std::string &v) { m_values[n] = v; }
    void setInteger(const std::string &n, int v) { m_values[n] = std::to_string(v); }
    void setFloat(const std::string &n, double v) { m_values[n] = format("%.17g", v); }
    void setBoolean(const std::string &n, bool v) { m_values[n] = v ? "true" : "false"; }
    std::string getString(const std::string &n, const std::string &def) const { auto it = m_values.find(n); return it == m_values.end() ? def : it->second; }
    std::string getString(const std::string &n) const
    {
        auto it = m_values.find(n);
        if (it == m_values.end()) logError(format("Property \"%s\" has not been specified!", n.c_str()));
        return it->second;
    }
    int getInteger(const std::string &n, int def) const { auto it = m_values.find(n); return it == m_values.end() ? def : std::stoi(it->second); }
    double getFloat(const std::string &n, double def) const { auto it = m_values.find(n); return it == m_values.end() ? def : std::stod(it->second); }
    bool getBoolean(const std::string &n, bool def) const
    {
        auto it = m_values.find(n);
        if (it == m_values.end()) return def;
        if (it->second == "true") return true;
        if (it->second == "false") return false;
        logError(format("Could not parse boolean value \"%s\" -- must be \"true\" or \"false\"", it->second.c_str()));
    }
    const std::map<std::string, std::string> &values() const { return m_values; }

private:
    std::string m_plugin, m_id;
    std::map<std::string, std::string> m_values;
};

namespace poisson {

/// poisson::Solver (Solver.hpp:43-157) over gdpt_poisson_*.
class Solver {
public:
    struct Params {                     // Solver.hpp:49-107 (the fields the solver reads)
        float alpha;
        std::string backend;            // "Auto" | "HIP": the only backend of this build
        int cudaDevice;                 // kept under the reference's name; a HIP device ordinal here
        bool verbose;
        int irlsIterMax;
        float irlsRegInit, irlsRegIter;
        int cgIterMax, cgIterCheck;
        bool cgPrecond;
        float cgTolerance;
        typedef std::function<void(const std::string &)> LogFunction;
        LogFunction logFunc;

        Params() { setDefaults(); }
        void setDefaults()              // Solver.cpp:57-88
        {
            alpha = 0.2f; verbose = false; backend = "Auto"; cudaDevice = -1;
            logFunc = LogFunction([](const std::string &m) { fputs(m.c_str(), stdout); });
            setConfigPreset("L1D");
        }
        bool setConfigPreset(const char *preset)    // Solver.cpp:94-178, forwarded so the numbers live in ONE place
        {
            gdpt_poisson_params p;
            const int ok = gdpt_poisson_params_preset(&p, preset);
            irlsIterMax = p.irlsIterMax; irlsRegInit = p.irlsRegInit; irlsRegIter = p.irlsRegIter;
            cgIterMax = p.cgIterMax; cgIterCheck = p.cgIterCheck; cgPrecond = p.cgPrecond != 0; cgTolerance = p.cgTolerance;
            return ok != 0;
        }
        void setLogFunction(LogFunction f) { logFunc = f; }
    };

    explicit Solver(const Params &params) : m_params(params), m_handle(nullptr)
    {
        if (params.backend != "Auto" && params.backend != "HIP")
            logError(format("Invalid backend specified '%s'!", params.backend.c_str()));                // Solver.cpp:292-294
        gdpt_poisson_params p;
        p.alpha = params.alpha; p.irlsIterMax = params.irlsIterMax; p.irlsRegInit = params.irlsRegInit; p.irlsRegIter = params.irlsRegIter;
        p.cgIterMax = params.cgIterMax; p.cgIterCheck = params.cgIterCheck; p.cgPrecond = params.cgPrecond; p.cgTolerance = params.cgTolerance;
        p.device = params.cudaDevice; p.verbose = params.verbose;
        check(gdpt_poisson_create(&p, &m_handle));
        check(gdpt_poisson_set_log(m_handle, &Solver::logThunk, this));
    }
    ~Solver() { gdpt_poisson_destroy(m_handle); }
    Solver(const Solver &) = delete;
    Solver &operator=(const Solver &) = delete;

    void importImagesMTS(float *dx, float *dy, float *tp, float *direct, int width, int height) { check(gdpt_poisson_import_images(m_handle, dx, dy, tp, direct, width, height)); }
    void importImagesDevice(const float *dx, const float *dy, const float *tp, const float *direct, int width, int height) { check(gdpt_poisson_import_images_device(m_handle, dx, dy, tp, direct, width, height)); }
    void setupBackend() { check(gdpt_poisson_setup_backend(m_handle)); }
    void solveIndirect() { check(gdpt_poisson_solve_indirect(m_handle)); }
    void evaluateMetricsMTS(float *err, float &errL1, float &errL2) { check(gdpt_poisson_evaluate_metrics(m_handle, err, &errL1, &errL2)); }      // Solver.cpp:511-541
    void exportImagesMTS(float *rec) { check(gdpt_poisson_export_images(m_handle, rec)); }
    float lastSolveSeconds() const { return gdpt_poisson_last_solve_seconds(m_handle); }

private:
    static void logThunk(const char *msg, void *self) { static_cast<Solver *>(self)->m_params.logFunc(msg); }
    Params m_params;
    gdpt_poisson_solver *m_handle;
};

} // namespace poisson

/// The reconstruction stage of the G-BDPT integrator (BASELINE config 5) -- the part of GBDPTIntegrator::render that is this library's
/// kind of work (src/integrators/gbdpt/gbdpt.cpp:178-247,264-280); its bidirectional sampler is not carried.
struct GBDPTReconstruction {
    /// GBDPTIntegrator::prepareDataForSolver (gbdpt.cpp:264-280), same arguments
    static void prepareDataForSolver(float w, float *out, const double *data, int len, const double *data2, int offset) { check(gdpt_gbdpt_prepare_data(w, out, data, len, data2, offset)); }
    /// the three prepareDataForSolver calls + Solver(L2D) + Solver(L1D) without a direct image (gbdpt.cpp:206-247); recL2 / recL1 may be null
    static void reconstruct(const double *primal, const double *gradNegY, const double *gradNegX, const double *gradPosX, const double *gradPosY,
                            int width, int height, float reconstructAlpha, float *recL2, float *recL1, int device = -1)
    { check(gdpt_gbdpt_reconstruct(primal, gradNegY, gradNegX, gradPosX, gradPosY, width, height, reconstructAlpha, device, recL2, recL1)); }
};

/// What the scene-XML subset reader produces and gdpt_scene_create consumes.
struct SceneData {
    std::vector<double> verts;              // 9 per triangle
    std::vector<double> normals;            // 9 per triangle (zeros = flat) for the first normals.size()/9 triangles; empty = none
    std::vector<int> triMaterial;
    std::vector<double> uvs;                // 6 per triangle (u0 v0 u1 v1 u2 v2) for the first uvs.size()/6 triangles; empty = no mesh has texture coordinates
    std::vector<unsigned char> triHasUV;    // per triangle of that prefix: its mesh has texture coordinates
    struct Texture { int width = 0, height = 0; std::vector<double> rgb; int wrapU = 0, wrapV = 0, filter = 1; double uscale = 1, vscale = 1, uoffset = 0, voffset = 0, scale = 1, maxAnisotropy = 20; };
    std::vector<Texture> textures;          // `<texture type="bitmap">`
    std::vector<int> materialTexture;       // per material: its reflectance / specularReflectance texture, -1 = constant (shorter than materials = -1)
    std::vector<gdpt_material> materials;
    std::vector<gdpt_emitter> emitters;
    bool hasEnvironment = false;            // <emitter type="constant"> or <emitter type="envmap">
    std::vector<double> envmapRgb;          // envmap: height x width x 3 linear values (environment.width / height), else empty
    gdpt_environment environment = {{0, 0, 0}, -1};
    gdpt_camera camera;
    Properties integrator, film, sampler, rfilter;
    int numTriangles() const { return (int)triMaterial.size(); }
};

/// The three StatsCounters on the path, printed as Statistics::getStats does (src/libcore/statistics.cpp:152-272):
/// "Normal rays traced" / "Shadow rays traced" (skdtree.cpp:46-47) and "Average path length" (gpt.cpp:72,1178-1179).
struct Statistics {
    unsigned long long raysTraced = 0, shadowRaysTraced = 0, paths = 0, pathLengthSum = 0;
    static std::string number(const char *name, double value)
    {   // ENumberValue, statistics.cpp:201-212
        static const char *suffix[] = {"", " K", " M", " G", " T"};
        int s = 0;
        float v = (float)value;
        while (v > 1000.0f && s <= 4) { v /= 1000.0f; s++; }
        return (v - std::floor(v) < 0.001f) ? format("    -  %s : %.0f%s", name, v, suffix[s]) : format("    -  %s : %.3f%s", name, v, suffix[s]);
    }
    std::string getStats() const
    {
        static const char *suffix[] = {"", " K", " M", " G", " T"};
        std::string o = "------------------------------------------------------------\n";
        o += " * Loaded plugins :\n    -  libgdpt_hip.so [G-PT hot path and screened Poisson reconstruction on MI355X]\n";
        int entries = 0;
        if (raysTraced || shadowRaysTraced) {
            o += "\n  * General :\n";
            if (raysTraced) { o += number("Normal rays traced", (double)raysTraced) + "\n"; entries++; }
            if (shadowRaysTraced) { o += number("Shadow rays traced", (double)shadowRaysTraced) + "\n"; entries++; }
        }
        if (pathLengthSum) {                                   // EAverage, statistics.cpp:243-258
            float v2 = (float)pathLengthSum, v3 = (float)paths;
            int s2 = 0, s3 = 0;
            while (v2 > 1000.0f && s2 < 4) { v2 /= 1000.0f; s2++; }
            while (v3 > 1000.0f && s3 < 4) { v3 /= 1000.0f; s3++; }
            o += "\n  * Gradient Path Tracer :\n";
            o += format("    -  Average path length : %.2f (%.2f%s / %.2f%s)\n", (double)pathLengthSum / (double)paths, v2, suffix[s2], v3, suffix[s3]);
            entries++;
        }
        if (!entries) o += " * Statistics:\n     none.\n";
        o += "------------------------------------------------------------";
        return o;
    }
};

/// MultiFilm (src/films/multifilm.cpp): N named buffers over one image, written as <dest><suffix>.pfm.
class MultiFilm {
public:
    explicit MultiFilm(const Properties &props)
    {
        m_width = props.getInteger("cropWidth", props.getInteger("width", 768));                    // film.cpp defaults; the buffers hold the crop window (film.cpp:40-43)
        m_height = props.getInteger("cropHeight", props.getInteger("height", 576));
        m_fileFormat = props.getString("fileFormat", "openexr");                                    // multifilm.cpp:104
        m_componentFormat = props.getString("componentFormat", "float16");                          // multifilm.cpp:116-117
        m_attachLog = props.getBoolean("attachLog", true);                                           // multifilm.cpp:108
        m_exrCompression = props.getString("exrCompression", "zip");                                // (not a reference property: OpenEXR's default ZIP, or "none" / "zips")
        if (m_fileFormat != "pfm" && m_fileFormat != "openexr" && m_fileFormat != "rgbe")
            logError("The \"fileFormat\" parameter must either be equal to \"openexr\", \"pfm\", or \"rgbe\"!");   // multifilm.cpp:126-127
        if (m_exrCompression != "zip" && m_exrCompression != "zips" && m_exrCompression != "none")
            logError(format("MultiFilm: exrCompression \"%s\" is not carried (zip, zips, none)", m_exrCompression.c_str()));
        if (m_componentFormat != "float16" && m_componentFormat != "float32")
            logError(format("MultiFilm: componentFormat \"%s\" is not carried (float16, float32)", m_componentFormat.c_str()));
        if (m_fileFormat == "pfm") m_componentFormat = "float32";                                   // multifilm.cpp:223-235: pfm forces float32
    }
    bool setBuffers(const std::vector<std::string> &names)                                          // multifilm.cpp:293-319
    {
        m_names = names;
        m_images.assign(names.size(), std::vector<float>((size_t)3 * m_width * m_height, 0.0f));
        return true;
    }
    int getWidth() const { return m_width; }
    int getHeight() const { return m_height; }
    std::vector<float> &buffer(size_t i) { return m_images[i]; }
    void setDestinationFile(const std::string &dest) { m_dest = dest; }
    /// MultiFilm::develop (multifilm.cpp:423-518): one file per buffer, then <dest>-log.txt and <dest>-stats.txt (:493-516).
    std::vector<std::string> develop(const std::string &log, const Statistics &stats = Statistics()) const
    {
        std::vector<std::string> written;
        for (size_t i = 0; i < m_names.size(); ++i) {
            if (m_fileFormat == "openexr") {
                const std::string path = m_dest + m_names[i] + ".exr";
                const int comp = m_exrCompression == "zip" ? ExrWriter::ZIP_COMPRESSION : (m_exrCompression == "zips" ? ExrWriter::ZIPS_COMPRESSION : ExrWriter::NO_COMPRESSION);
                if (!ExrWriter::write(path, m_images[i].data(), m_width, m_height, m_componentFormat == "float16", m_attachLog ? log : std::string(), comp))   // :481-484
                    logError(format("Cannot write \"%s\"", path.c_str()));
                written.push_back(path);
                continue;
            }
            if (m_fileFormat == "rgbe") {                                                          // multifilm.cpp:462-463
                const std::string path = m_dest + m_names[i] + ".rgbe";
                if (!ExrWriter::writeRGBE(path, m_images[i].data(), m_width, m_height)) logError(format("Cannot write \"%s\"", path.c_str()));
                written.push_back(path);
                continue;
            }
            const std::string path = m_dest + m_names[i] + ".pfm";
            std::ofstream f(path, std::ios::binary);
            if (!f) logError(format("Cannot write \"%s\"", path.c_str()));
            f << "PF\n" << m_width << " " << m_height << "\n-1.0\n";
            for (int y = m_height - 1; y >= 0; --y)                                                 // PFM stores the bottom row first
                f.write(reinterpret_cast<const char *>(&m_images[i][(size_t)3 * m_width * y]), sizeof(float) * 3 * m_width);
            written.push_back(path);
        }
        std::ofstream lf(m_dest + "-log.txt");
        lf << log;
        std::ofstream sf(m_dest + "-stats.txt");
        sf << stats.getStats();
        return written;
    }

private:
    int m_width, m_height;
    std::string m_fileFormat, m_componentFormat, m_dest, m_exrCompression;
    bool m_attachLog = true;
    std::vector<std::string> m_names;
    std::vector<std::vector<float>> m_images;
};

/// GradientPathIntegrator (gpt.h:71-113; gpt.cpp:1190-1213 constructor, :1358-1480 render).
class GradientPathIntegrator {
public:
    explicit GradientPathIntegrator(const Properties &props)
    {
        m_maxDepth = props.getInteger("maxDepth", -1);
        m_rrDepth = props.getInteger("rrDepth", 5);
        m_strictNormals = props.getBoolean("strictNormals", false);
        m_hideEmitters = props.getBoolean("hideEmitters", false);
        m_shiftThreshold = props.getFloat("shiftThreshold", 0.001);
        m_reconstructL1 = props.getBoolean("reconstructL1", true);
        m_reconstructL2 = props.getBoolean("reconstructL2", false);
        m_reconstructAlpha = props.getFloat("reconstructAlpha", 0.2);
        if (m_reconstructL1 && m_reconstructL2)
            logError("Disable 'reconstructL1' or 'reconstructL2': Cannot display two reconstructions at a time!");
        if (m_reconstructAlpha <= 0.0)
            logError("'reconstructAlpha' must be set to a value greater than zero!");
        if (m_maxDepth <= 0 && m_maxDepth != -1)
            logError("'maxDepth' must be set to -1 (infinite) or a value greater than zero!");
    }

    /// The GPUs a frame is sharded over (row strips + one-pixel halo, the multi-GPU form of the block scheduler: imageproc.cpp:28-78,
    /// gpt_proc.cpp:52-56,137-149).  Device ordinals; an ordinal may repeat (several strips on one GPU: functional runs).  Empty or
    /// one entry: the single-device path.  gdpt_mitsuba's -p sets it.
    void setDevices(const std::vector<int> &devices) { m_devices = devices; }

    /// render (gpt.cpp:1358-1480): five buffers, blocks, develop, reconstruct, -final := reconstruction.
    bool render(const SceneData &sd, MultiFilm &film, int sampleCount, unsigned long long seed, std::string &log)
    {
        if (m_devices.size() > 1) return renderStrips(sd, film, sampleCount, seed, log);
        if (m_hideEmitters) logError("Option 'hideEmitters' not implemented for Gradient-Domain Path Tracing!");
        const std::vector<std::string> outNames = {"-final", "-throughput", "-dx", "-dy", "-direct"};
        if (!film.setBuffers(outNames)) logError("Cannot render image! G-PT has been called without MultiFilm.");
        const int W = film.getWidth(), H = film.getHeight();
        gdpt_scene *scene = nullptr;
        gdpt_film *gf = nullptr;
        createScene(sd, -1, &scene);
        check(gdpt_film_create(scene, 0, H, &gf));
        {
            int kind = GDPT_RFILTER_BOX; double p0 = 0, p1 = 0;
            rfilterOf(sd, kind, p0, p1);
            check(gdpt_film_set_rfilter(gf, kind, p0, p1));
        }
        gdpt_config cfg;
        cfg.maxDepth = m_maxDepth; cfg.rrDepth = m_rrDepth; cfg.strictNormals = m_strictNormals; cfg.spp = sampleCount;
        cfg.shiftThreshold = m_shiftThreshold; cfg.seed = seed;
        log += format("Starting render job (GPT::render) (%ix%i, %i %s, 1 MI355X) ..\n", W, H, sampleCount, sampleCount == 1 ? "sample" : "samples");
        m_film.store(gf);
        check(gdpt_render_rect(scene, &cfg, 0, 0, W, H, gf));
        check(gdpt_film_sync(gf));
        m_film.store(nullptr);
        int cancelled = 0;
        check(gdpt_film_cancelled(gf, &cancelled));
        if (cancelled) {                                                                         // sched->cancel: no develop, no reconstruction, render() == false
            log += "Render job cancelled.\n";
            gdpt_film_destroy(gf);
            gdpt_scene_destroy(scene);
            return false;
        }
        for (int b = 0; b < 5; ++b) check(gdpt_film_develop(gf, b, film.buffer(b).data()));
        unsigned long long st[4];
        check(gdpt_film_stats(gf, st));
        m_stats.raysTraced = st[0]; m_stats.shadowRaysTraced = st[1]; m_stats.paths = st[2]; m_stats.pathLengthSum = st[3];
        const float ms = gdpt_film_render_ms(gf);
        log += format("Render time: %.3f s, %llu rays + %llu shadow rays (%.1f Mray/s), average path length %.3f\n", ms * 1e-3, st[0], st[1],
                      (st[0] + st[1]) / (ms * 1e3), st[2] ? (double)st[3] / st[2] : 0.0);
        if (m_reconstructL1 || m_reconstructL2) {                                               // gpt.cpp:1415-1476
            poisson::Solver::Params params;
            params.setConfigPreset(m_reconstructL1 ? "L1D" : "L2D");
            params.alpha = (float)m_reconstructAlpha;
            params.setLogFunction([&log](const std::string &m) { log += m; });
            poisson::Solver solver(params);
            std::vector<float> rec((size_t)3 * W * H);
            solver.importImagesMTS(film.buffer(2).data(), film.buffer(3).data(), film.buffer(1).data(), film.buffer(4).data(), W, H);
            solver.setupBackend();
            solver.solveIndirect();
            solver.exportImagesMTS(rec.data());
            film.buffer(0) = rec;                                                                // setBitmapMulti(reconstruction, 1, BUFFER_FINAL)
        }
        gdpt_film_destroy(gf);
        gdpt_scene_destroy(scene);
        return true;
    }

    /// One frame over several GPUs of the node, from ONE process: a thread per strip uploads the scene to its device and renders its rows
    /// (GPTBlockRenderer::process over the strip); neighbouring strips then settle their one-pixel borders -- each ships its boundary
    /// row's per-pixel sums and the exact puts it made into the neighbour's row (gdpt_film_pack_halo / unpack_halo), device to device over
    /// xGMI (gdpt_device_copy = hipMemcpyPeer) -- every strip develops its rows of the five buffers on its own device, the solver inputs are
    /// gathered on the first device by peer copies, and that device reconstructs.  With a reconstruction filter wider than box a strip
    /// renders the filter's reach itself and nothing is exchanged (gdpt_film_set_rfilter).  Same strips, payloads and order as
    /// parallel.StripRenderer (the one-process-per-GPU form over RCCL); samples depend on (seed, pixel, sample index) only, so the image
    /// does not depend on the number of devices beyond the rounding of the border sums.
    bool renderStrips(const SceneData &sd, MultiFilm &film, int sampleCount, unsigned long long seed, std::string &log)
    {
        if (m_hideEmitters) logError("Option 'hideEmitters' not implemented for Gradient-Domain Path Tracing!");
        const std::vector<std::string> outNames = {"-final", "-throughput", "-dx", "-dy", "-direct"};
        if (!film.setBuffers(outNames)) logError("Cannot render image! G-PT has been called without MultiFilm.");
        const int W = film.getWidth(), H = film.getHeight(), N = (int)m_devices.size();
        if (2 * N > H) logError("strips need at least two rows each (the halo carries the exact puts of two rows beyond a boundary): fewer devices, or a taller image");
        struct Strip { int device = 0, y0 = 0, y1 = 0; gdpt_scene *scene = nullptr; gdpt_film *film = nullptr; void *out[2] = {nullptr, nullptr}, *in[2] = {nullptr, nullptr}; float *imgs = nullptr; std::string error; };
        std::vector<Strip> strips(N);
        for (int r = 0, y = 0; r < N; ++r) {                       // contiguous rows, earlier strips take the remainder (parallel.row_strips)
            const int n = H / N + (r < H % N ? 1 : 0);
            strips[r].device = m_devices[r]; strips[r].y0 = y; strips[r].y1 = y + n; y += n;
        }
        gdpt_config cfg;
        cfg.maxDepth = m_maxDepth; cfg.rrDepth = m_rrDepth; cfg.strictNormals = m_strictNormals; cfg.spp = sampleCount;
        cfg.shiftThreshold = m_shiftThreshold; cfg.seed = seed;
        int kind = GDPT_RFILTER_BOX; double p0 = 0, p1 = 0;
        rfilterOf(sd, kind, p0, p1);
        log += format("Starting render job (GPT::render) (%ix%i, %i %s, %d strips on %d MI355X) ..\n", W, H, sampleCount, sampleCount == 1 ? "sample" : "samples", N, N);
        auto cleanup = [&]() {
            for (Strip &s : strips) {
                for (int k = 0; k < 2; ++k) { gdpt_device_free(s.device, s.out[k]); gdpt_device_free(s.device, s.in[k]); }
                gdpt_device_free(s.device, s.imgs);
                gdpt_film_destroy(s.film);
                gdpt_scene_destroy(s.scene);
            }
        };
        // (1) one thread per strip: scene upload + render
        std::vector<std::thread> workers;
        for (int r = 0; r < N; ++r)
            workers.emplace_back([&, r]() {
                Strip &s = strips[r];
                try {
                    createScene(sd, s.device, &s.scene);
                    check(gdpt_film_create(s.scene, s.y0, s.y1, &s.film));
                    check(gdpt_film_set_rfilter(s.film, kind, p0, p1));
                    check(gdpt_render_rect(s.scene, &cfg, 0, s.y0, W, s.y1, s.film));
                    check(gdpt_film_sync(s.film));
                } catch (const std::exception &e) { s.error = e.what(); }
            });
        for (std::thread &t : workers) t.join();
        for (const Strip &s : strips) if (!s.error.empty()) { const std::string e = s.error; cleanup(); logError(e); }
        try {
            // (2) borders: box filter only (wider filters: the strips rendered the reach themselves)
            size_t haloBytes = 0;
            if (kind == GDPT_RFILTER_BOX) {
                check(gdpt_film_halo_bytes(strips[0].film, &haloBytes));
                for (int r = 0; r < N; ++r)
                    for (int which = 0; which < 2; ++which) {
                        const int peer = which == 0 ? r - 1 : r + 1;
                        if (peer < 0 || peer >= N) continue;
                        check(gdpt_device_alloc(strips[r].device, haloBytes, &strips[r].out[which]));
                        check(gdpt_device_alloc(strips[r].device, haloBytes, &strips[r].in[which]));
                        check(gdpt_film_pack_halo(strips[r].film, which, strips[r].out[which]));         // every strip packs before anyone unpacks
                    }
                for (int r = 0; r + 1 < N; ++r) {
                    check(gdpt_device_copy(strips[r + 1].device, strips[r + 1].in[0], strips[r].device, strips[r].out[1], haloBytes));      // down
                    check(gdpt_device_copy(strips[r].device, strips[r].in[1], strips[r + 1].device, strips[r + 1].out[0], haloBytes));      // up
                }
                for (int r = 0; r < N; ++r)
                    for (int which = 0; which < 2; ++which)
                        if (strips[r].in[which]) check(gdpt_film_unpack_halo(strips[r].film, which, strips[r].in[which]));
            }
            // (3) develop on every device, gather the five images on the first one
            const int d0 = strips[0].device;
            const size_t imgFloats = (size_t)3 * W * H;
            void *full = nullptr;
            check(gdpt_device_alloc(d0, sizeof(float) * 5 * imgFloats, &full));
            float *fullImgs = static_cast<float *>(full);
            try {
                for (Strip &s : strips) {
                    const size_t rows = (size_t)(s.y1 - s.y0), stripFloats = 3 * rows * W;
                    void *p = nullptr;
                    check(gdpt_device_alloc(s.device, sizeof(float) * 5 * stripFloats, &p));
                    s.imgs = static_cast<float *>(p);
                    for (int b = 0; b < 5; ++b) {
                        check(gdpt_film_develop_device(s.film, b, s.imgs + b * stripFloats));
                        check(gdpt_device_copy(d0, fullImgs + b * imgFloats + (size_t)3 * W * s.y0, s.device, s.imgs + b * stripFloats, sizeof(float) * stripFloats));
                    }
                }
                for (int b = 0; b < 5; ++b) check(gdpt_device_download(d0, film.buffer(b).data(), fullImgs + b * imgFloats, sizeof(float) * imgFloats));
                // statistics and timing: sums over the strips, the slowest strip's render time
                unsigned long long tot[4] = {0, 0, 0, 0};
                float slowest = 0.0f;
                for (Strip &s : strips) {
                    unsigned long long st[4];
                    check(gdpt_film_stats(s.film, st));
                    for (int k = 0; k < 4; ++k) tot[k] += st[k];
                    const float ms = gdpt_film_render_ms(s.film);
                    log += format("  strip rows [%d, %d) on device %d: %.3f s\n", s.y0, s.y1, s.device, ms * 1e-3);
                    slowest = std::max(slowest, ms);
                }
                m_stats.raysTraced = tot[0]; m_stats.shadowRaysTraced = tot[1]; m_stats.paths = tot[2]; m_stats.pathLengthSum = tot[3];
                log += format("Render time: %.3f s (slowest strip), %llu rays + %llu shadow rays (%.1f Mray/s), average path length %.3f, halo %zu bytes per border\n", slowest * 1e-3, tot[0], tot[1],
                              (tot[0] + tot[1]) / (slowest * 1e3), tot[2] ? (double)tot[3] / tot[2] : 0.0, haloBytes);
                // (4) reconstruction on the first device, inputs already resident there (gpt.cpp:1415-1476)
                if (m_reconstructL1 || m_reconstructL2) {
                    poisson::Solver::Params params;
                    params.setConfigPreset(m_reconstructL1 ? "L1D" : "L2D");
                    params.alpha = (float)m_reconstructAlpha;
                    params.cudaDevice = d0;
                    params.setLogFunction([&log](const std::string &m) { log += m; });
                    poisson::Solver solver(params);
                    std::vector<float> rec(imgFloats);
                    solver.importImagesDevice(fullImgs + 2 * imgFloats, fullImgs + 3 * imgFloats, fullImgs + 1 * imgFloats, fullImgs + 4 * imgFloats, W, H);
                    solver.setupBackend();
                    solver.solveIndirect();
                    solver.exportImagesMTS(rec.data());
                    film.buffer(0) = rec;
                }
            } catch (...) { gdpt_device_free(d0, full); throw; }
            gdpt_device_free(d0, full);
        } catch (...) { cleanup(); throw; }
        cleanup();
        return true;
    }

    /// Integrator::cancel (integrator.h:88): callable from another thread while render() runs.
    void cancel()
    {
        gdpt_film *f = m_film.load();
        if (f) gdpt_film_cancel(f);
    }

    const Statistics &getStatistics() const { return m_stats; }

    /// gdpt_scene_create_tex from a SceneData (vertex normals, texture coordinates and bitmap textures padded to the scene's size)
    static void createScene(const SceneData &sd, int device, gdpt_scene **scene)
    {
        const size_t nt = (size_t)sd.numTriangles();
        std::vector<double> normals = sd.normals, uvs = sd.uvs;
        if (!normals.empty()) normals.resize(9 * nt, 0.0);
        std::vector<unsigned char> has = sd.triHasUV;
        if (!uvs.empty()) { uvs.resize(6 * nt, 0.0); has.resize(nt, 0); }
        std::vector<gdpt_texture> tex(sd.textures.size());
        for (size_t i = 0; i < tex.size(); ++i) {
            const SceneData::Texture &t = sd.textures[i];
            tex[i].width = t.width; tex[i].height = t.height; tex[i].rgb = t.rgb.data(); tex[i].wrapU = t.wrapU; tex[i].wrapV = t.wrapV; tex[i].filter = t.filter;
            tex[i].uscale = t.uscale; tex[i].vscale = t.vscale; tex[i].uoffset = t.uoffset; tex[i].voffset = t.voffset; tex[i].scale = t.scale; tex[i].maxAnisotropy = t.maxAnisotropy;
        }
        gdpt_environment envCopy = sd.environment;
        envCopy.rgb = sd.envmapRgb.empty() ? nullptr : sd.envmapRgb.data();
        std::vector<int> mtex = sd.materialTexture;
        mtex.resize(sd.materials.size(), -1);
        check(gdpt_scene_create_tex(sd.numTriangles(), sd.verts.data(), normals.empty() ? nullptr : normals.data(), uvs.empty() ? nullptr : uvs.data(),
                                    uvs.empty() ? nullptr : has.data(), sd.triMaterial.data(), (int)sd.materials.size(), sd.materials.data(),
                                    tex.empty() ? nullptr : mtex.data(), (int)tex.size(), tex.empty() ? nullptr : tex.data(),
                                    (int)sd.emitters.size(), sd.emitters.data(), sd.hasEnvironment ? &envCopy : nullptr, &sd.camera, device, scene));
    }

private:
    /// <rfilter>: src/rfilters/*.cpp with their default parameters
    static void rfilterOf(const SceneData &sd, int &kind, double &p0, double &p1)
    {
        const std::string ft = sd.rfilter.getPluginName();
        kind = GDPT_RFILTER_BOX; p0 = 0; p1 = 0;
        if (ft == "tent") kind = GDPT_RFILTER_TENT;
        else if (ft == "gaussian") { kind = GDPT_RFILTER_GAUSSIAN; p0 = sd.rfilter.getFloat("stddev", 0.5); }
        else if (ft == "mitchell") { kind = GDPT_RFILTER_MITCHELL; p0 = sd.rfilter.getFloat("B", 1.0 / 3.0); p1 = sd.rfilter.getFloat("C", 1.0 / 3.0); }
        else if (ft == "catmullrom") kind = GDPT_RFILTER_CATMULLROM;
        else if (ft == "lanczos") { kind = GDPT_RFILTER_LANCZOS; p0 = (double)sd.rfilter.getInteger("lobes", 3); }
    }

    std::vector<int> m_devices;
    Statistics m_stats;
    std::atomic<gdpt_film *> m_film{nullptr};
    int m_maxDepth, m_rrDepth;
    bool m_strictNormals, m_hideEmitters, m_reconstructL1, m_reconstructL2;
    double m_shiftThreshold, m_reconstructAlpha;
};

/// GBDPTIntegrator (/root/reference/src/integrators/gbdpt/gbdpt.cpp:77-300): constructor :79-104, render :140-262.  The sampler is
/// GBDPTProcess / GBDPTRenderer behind gdpt_gbdpt_render_rect; the film's seven buffers get the names of gbdpt.cpp:163; BOTH reconstructions
/// are computed and written (the reconstructL1 / reconstructL2 properties only choose which one comes first, :87,235-251).
class GBDPTIntegrator {
public:
    explicit GBDPTIntegrator(const Properties &props)
    {
        m_maxDepth = props.getInteger("maxDepth", -1);
        m_rrDepth = props.getInteger("rrDepth", 5);
        m_lightImage = props.getBoolean("lightImage", true);
        m_shiftThreshold = props.getFloat("shiftThreshold", 0.001);
        m_reconstructL1 = props.getBoolean("reconstructL1", true);
        m_reconstructL2 = props.getBoolean("reconstructL2", false);
        m_reconstructAlpha = props.getFloat("reconstructAlpha", 0.2);
        if (m_reconstructL1 && m_reconstructL2)
            logError("Disable 'reconstructL1' or 'reconstructL2': Cannot display two reconstructions at a time!");
        if (m_reconstructAlpha <= 0.0)
            logError("'reconstructAlpha' must be set to a value greater than zero!");
        if (m_rrDepth <= 0)
            logError("'rrDepth' must be set to a value greater than zero!");
        if (m_maxDepth <= 0 && m_maxDepth != -1)
            logError("'maxDepth' must be set to -1 (infinite) or a value greater than zero!");
    }

    ~GBDPTIntegrator() { if (m_lastW > 0) gdpt_gbdpt_reconstruct_release_size(-1, m_lastW, m_lastH); }     // (the library keeps a frame size's solvers between render() calls: this integrator's own size goes back, nobody else's)
    const Statistics &getStatistics() const { return m_stats; }

    /// render (gbdpt.cpp:140-262)
    bool render(const SceneData &sd, MultiFilm &film, int sampleCount, unsigned long long seed, std::string &log)
    {
        const std::vector<std::string> outNames = {(m_reconstructL1 ? "-L1" : "-L2"), "-gradientNegY", "-gradientNegX", "-gradientPosX", "-gradientPosY",
                                                   (m_reconstructL1 ? "-L2" : "-L1"), "-primal"};                          // :163
        if (!film.setBuffers(outNames)) logError("Cannot render image! G-BDPT has been called without MultiFilm.");
        if (sd.rfilter.getPluginName() != "box" && !sd.rfilter.getPluginName().empty())
            logError("G-BDPT supports no pixel filter beside the box filter (gbdpt.cpp:70-71)");
        const int W = film.getWidth(), H = film.getHeight();
        if (m_lastW > 0 && (m_lastW != W || m_lastH != H)) gdpt_gbdpt_reconstruct_release_size(-1, m_lastW, m_lastH);
        m_lastW = W; m_lastH = H;
        gdpt_scene *scene = nullptr;
        gdpt_gbdpt_film *gf = nullptr;
        GradientPathIntegrator::createScene(sd, -1, &scene);
        struct Guard { gdpt_scene *&s; gdpt_gbdpt_film *&f; ~Guard() { gdpt_gbdpt_film_destroy(f); gdpt_scene_destroy(s); } } guard{scene, gf};
        check(gdpt_gbdpt_film_create(scene, &gf));
        gdpt_gbdpt_config cfg;
        cfg.maxDepth = m_maxDepth; cfg.rrDepth = m_rrDepth; cfg.lightImage = m_lightImage; cfg.spp = sampleCount; cfg.shiftThreshold = m_shiftThreshold; cfg.seed = seed;
        log += format("Starting render job (%ix%i, %i %s, 1 MI355X) ..\n", W, H, sampleCount, sampleCount == 1 ? "sample" : "samples");
        check(gdpt_gbdpt_render_rect(scene, &cfg, 0, 0, W, H, gf));
        check(gdpt_gbdpt_film_sync(gf));
        const size_t n3 = (size_t)3 * W * H;
        std::vector<std::vector<double>> dev(5, std::vector<double>(n3));                     // film->developMulti of the primal and the four gradients, :199-202
        for (int b = 0; b < 5; ++b) check(gdpt_gbdpt_film_develop(gf, b, sampleCount, dev[b].data()));
        unsigned long long st[4];
        check(gdpt_gbdpt_film_stats(gf, st));
        m_stats.raysTraced = st[0]; m_stats.shadowRaysTraced = st[1]; m_stats.paths = st[2]; m_stats.pathLengthSum = 0;
        const float ms = gdpt_gbdpt_film_render_ms(gf);
        log += format("Render time: %.3f s, %llu rays + %llu shadow rays (%.1f Mray/s)\n", ms * 1e-3, st[0], st[1], (st[0] + st[1]) / (ms * 1e3));
        std::vector<float> l2(n3), l1(n3);                                                     // prepareDataForSolver + both solves, :209-251
        check(gdpt_gbdpt_reconstruct(dev[0].data(), dev[1].data(), dev[2].data(), dev[3].data(), dev[4].data(), W, H, (float)m_reconstructAlpha, -1, l2.data(), l1.data()));
        film.buffer(0) = m_reconstructL1 ? l1 : l2;
        film.buffer(5) = m_reconstructL1 ? l2 : l1;
        for (int b = 1; b <= 4; ++b) for (size_t i = 0; i < n3; ++i) film.buffer(b)[i] = (float)dev[b][i];
        for (size_t i = 0; i < n3; ++i) film.buffer(6)[i] = (float)dev[0][i];                  // setBitmapMulti(imgBaseBuff, 1, nNeighbours + 2), :255
        return true;
    }

private:
    Statistics m_stats;
    int m_maxDepth, m_rrDepth;
    int m_lastW = 0, m_lastH = 0;
    bool m_lightImage, m_reconstructL1, m_reconstructL2;
    double m_shiftThreshold, m_reconstructAlpha;
};

} // namespace gdpt
