// gdpt_mitsuba -- command line front end with the reference CLI's flags for this path
// (/root/reference/src/mitsuba/mitsuba.cpp:154-250): gdpt_mitsuba [-o dest] [-D key=val]... [-p n] [-b n] [-x] [-q] scene.xml
//   -o  output destination stem: writes <dest>-final|-throughput|-dx|-dy|-direct.{exr|pfm} and <dest>-log.txt, <dest>-stats.txt (multifilm.cpp:453-517)
//   -D  parameter substitution for $key in the scene file
//   -p  the reference's "number of local worker cores" (mitsuba.cpp:187-194) = here the number of GPUs the frame is sharded over: row strips
//       with a one-pixel halo exchanged device to device (host/gdpt_host.hpp renderStrips); more strips than visible GPUs wrap around (a
//       functional run with several strips per device).  --devices a,b,c names the ordinals explicitly.
//   -b  accepted for compatibility (the block size has no meaning for the GPU path) and ignored
//   -x  skip rendering if <dest>-final.pfm exists;  -q  quiet;  --parse-only  load the scene, print a summary, do not touch the GPU
#include "scene_xml.hpp"

#include <sys/stat.h>

int main(int argc, char **argv)
{
    std::string dest, scenePath;
    bool skipExisting = false, quiet = false, parseOnly = false;
    unsigned long long seed = 5489;     // include/mitsuba/core/random.h:113
    int numDevices = 1;
    std::vector<int> deviceList;
    gdpt::SceneLoader loader;
    try {
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            auto need = [&](const char *what) -> std::string { if (i + 1 >= argc) gdpt::logError(std::string("missing value after ") + what); return argv[++i]; };
            if (a == "-o") dest = need("-o");
            else if (a == "-D") { const std::string kv = need("-D"); const size_t eq = kv.find('='); if (eq == std::string::npos) gdpt::logError("-D expects key=value"); loader.params[kv.substr(0, eq)] = kv.substr(eq + 1); }
            else if (a.rfind("-D", 0) == 0 && a.size() > 2) { const std::string kv = a.substr(2); const size_t eq = kv.find('='); if (eq == std::string::npos) gdpt::logError("-D expects key=value"); loader.params[kv.substr(0, eq)] = kv.substr(eq + 1); }
            else if (a == "-p") { numDevices = std::stoi(need("-p")); if (numDevices < 1) gdpt::logError("-p expects a positive number"); }
            else if (a == "--devices") { const std::string l = need("--devices"); size_t at = 0; while (at <= l.size()) { const size_t c = l.find(',', at); deviceList.push_back(std::stoi(l.substr(at, c == std::string::npos ? std::string::npos : c - at))); if (c == std::string::npos) break; at = c + 1; } }
            else if (a == "-b") need(a.c_str());
            else if (a == "-x") skipExisting = true;
            else if (a == "-q") quiet = true;
            else if (a == "--seed") seed = std::stoull(need("--seed"));
            else if (a == "--parse-only") parseOnly = true;
            else if (a == "--pfm2exr") {      // utility (and CPU-testable face of the EXR writer): --pfm2exr in.pfm out.exr [float16|float32]
                const std::string in = need("--pfm2exr"), out = need("--pfm2exr"), fmt = (i + 1 < argc) ? argv[++i] : "float16", cmp = (i + 1 < argc) ? argv[++i] : "zip";
                std::ifstream f(in, std::ios::binary);
                std::string magic; int w = 0, h = 0; float scale = 0;
                f >> magic >> w >> h >> scale;
                f.get();
                if (!f || magic != "PF" || scale >= 0 || w <= 0 || h <= 0) gdpt::logError("--pfm2exr: expected a little-endian colour PFM");
                std::vector<float> img((size_t)3 * w * h);
                for (int y = h - 1; y >= 0; --y) f.read(reinterpret_cast<char *>(&img[(size_t)3 * w * y]), sizeof(float) * 3 * w);
                if (fmt == "rgbe") { if (!gdpt::ExrWriter::writeRGBE(out, img.data(), w, h)) gdpt::logError("cannot write " + out); return 0; }
                const int comp = cmp == "zip" ? gdpt::ExrWriter::ZIP_COMPRESSION : (cmp == "zips" ? gdpt::ExrWriter::ZIPS_COMPRESSION : gdpt::ExrWriter::NO_COMPRESSION);
                if (!gdpt::ExrWriter::write(out, img.data(), w, h, fmt == "float16", "", comp)) gdpt::logError("cannot write " + out);
                return 0;
            }
            else if (a == "--tex2pfm") {      // utility (and CPU-testable face of the texture / environment-map READER): --tex2pfm in.{exr,png,ppm,pfm} out.pfm
                const std::string in = need("--tex2pfm"), out = need("--tex2pfm");
                gdpt::SceneData::Texture t;
                loader.readBitmap(in, t, 1.0);
                std::ofstream o(out, std::ios::binary);
                o << "PF\n" << t.width << " " << t.height << "\n-1.0\n";
                std::vector<float> row((size_t)3 * t.width);
                for (int y = t.height - 1; y >= 0; --y) { for (int x = 0; x < 3 * t.width; ++x) row[x] = (float)t.rgb[(size_t)3 * t.width * y + x]; o.write(reinterpret_cast<const char *>(row.data()), sizeof(float) * row.size()); }
                return 0;
            }
            else if (a == "-h" || a == "--help") { printf("usage: gdpt_mitsuba [-o dest] [-D key=val] [-p gpus] [--devices a,b,..] [-b n] [-x] [-q] [--seed n] [--parse-only] scene.xml\n"); return 0; }
            else if (a[0] == '-') gdpt::logError("unknown option " + a);
            else scenePath = a;
        }
        if (scenePath.empty()) gdpt::logError("no scene file given");
        if (dest.empty()) { dest = scenePath; const size_t dot = dest.rfind(".xml"); if (dot != std::string::npos) dest.erase(dot); }   // mitsuba.cpp: default destination = scene name
        gdpt::SceneData sd = loader.load(scenePath);
        const int spp = sd.sampler.getInteger("sampleCount", 4);                                                                       // independent.cpp default
        if (parseOnly) {
            int smooth = 0;
            double firstN[3] = {0, 0, 0};
            for (size_t t = 0; t * 9 < sd.normals.size(); ++t) {
                bool any = false;
                for (int k = 0; k < 9; ++k) any = any || sd.normals[9 * t + k] != 0.0;
                if (any && !smooth) for (int k = 0; k < 3; ++k) firstN[k] = sd.normals[9 * t + k];
                smooth += any;
            }
            printf("{\"triangles\": %d, \"materials\": %zu, \"emitters\": %zu, \"width\": %d, \"height\": %d, \"fovX\": %.9g, \"sampleCount\": %d, \"maxDepth\": %d, \"firstVertex\": [%.9g, %.9g, %.9g], \"cameraOrigin\": [%.9g, %.9g, %.9g], \"environment\": [%.9g, %.9g, %.9g, %d], \"smoothTriangles\": %d, \"firstNormal\": [%.9g, %.9g, %.9g]}\n",
                   sd.numTriangles(), sd.materials.size(), sd.emitters.size(), sd.camera.width, sd.camera.height, sd.camera.fovX, spp,
                   sd.integrator.getInteger("maxDepth", -1), sd.verts[0], sd.verts[1], sd.verts[2], sd.camera.toWorld[3], sd.camera.toWorld[7], sd.camera.toWorld[11],
                   sd.environment.radiance[0], sd.environment.radiance[1], sd.environment.radiance[2], sd.hasEnvironment ? sd.environment.index : -1,
                   smooth, firstN[0], firstN[1], firstN[2]);
            return 0;
        }
        struct stat stt;
        if (skipExisting && (stat((dest + "-final.pfm").c_str(), &stt) == 0 || stat((dest + "-final.exr").c_str(), &stt) == 0)) { if (!quiet) printf("Skipping %s (output exists)\n", scenePath.c_str()); return 0; }
        if (sd.integrator.getPluginName() == "gbdpt") {                                  // <integrator type="gbdpt">: -L1 -gradient{NegY,NegX,PosX,PosY} -L2 -primal
            if (skipExisting && (stat((dest + "-L1.pfm").c_str(), &stt) == 0 || stat((dest + "-L1.exr").c_str(), &stt) == 0)) { if (!quiet) printf("Skipping %s (output exists)\n", scenePath.c_str()); return 0; }
            gdpt::GBDPTIntegrator bd(sd.integrator);
            gdpt::MultiFilm film(sd.film);
            film.setDestinationFile(dest);
            std::string log;
            bd.render(sd, film, spp, seed, log);
            for (const std::string &p : film.develop(log, bd.getStatistics())) if (!quiet) printf("Writing image to \"%s\" ..\n", p.c_str());
            if (!quiet) fputs(log.c_str(), stdout);
            return 0;
        }
        gdpt::GradientPathIntegrator integrator(sd.integrator);
        if (deviceList.empty() && numDevices > 1) {
            int visible = 0;
            gdpt::check(gdpt_device_count(&visible));
            if (visible <= 0) gdpt::logError("no HIP device visible");
            for (int r = 0; r < numDevices; ++r) deviceList.push_back(r % visible);
        }
        integrator.setDevices(deviceList);
        gdpt::MultiFilm film(sd.film);
        film.setDestinationFile(dest);
        std::string log;
        integrator.render(sd, film, spp, seed, log);
        for (const std::string &p : film.develop(log, integrator.getStatistics())) if (!quiet) printf("Writing image to \"%s\" ..\n", p.c_str());
        if (!quiet) fputs(log.c_str(), stdout);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "Error: %s\n", e.what());       // Log(EError) -> exception -> non-zero exit, as mitsuba.cpp's top-level handler
        return 1;
    }
}
