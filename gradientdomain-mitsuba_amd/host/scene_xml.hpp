// scene_xml.hpp -- reader for the subset of Mitsuba 0.5 scene XML that the carried hot path can render
// (tag table: /root/reference/src/librender/scenehandler.cpp:70-106; grammar: data/schema/scene.xsd).
//
// Carried:  <scene>, <default>, $parameter substitution (+ -D overrides), <integrator type="gpt">, <sensor type="perspective">
// (fov, fovAxis x|y, nearClip, farClip, toWorld), <sampler type="independent">, <film type="multifilm"> (width, height,
// fileFormat="openexr"|"pfm") with <rfilter type="box|tent|gaussian|mitchell|catmullrom|lanczos"> (gaussian when absent, film.cpp:93), <bsdf type="diffuse|conductor|roughconductor|dielectric|twosided"> (top-level with id, or nested in a
// shape), <shape type="obj|serialized|rectangle|cube"> (filename, toWorld, flipNormals, <ref>, nested <bsdf>, nested <emitter type="area">),
// top-level <emitter type="constant"> (radiance), <emitter type="envmap"> (filename, scale, gamma, toWorld) and <emitter type="point"> (position | toWorld, intensity),
// <transform> built from translate / rotate / scale / lookat / matrix, <integer|float|boolean|string|rgb|spectrum>.
// Anything else raises std::runtime_error naming the tag or plugin, like the reference's "unsupported" errors.
#pragma once
#include "gdpt_host.hpp"

#include <array>
#include <iterator>
#include <zlib.h>
#include <cctype>
#include <map>
#include <cmath>
#include <memory>
#include <sstream>

namespace gdpt {
namespace xml {

struct Node {
    std::string tag;
    std::map<std::string, std::string> attr;
    std::vector<std::unique_ptr<Node>> children;
    const std::string &get(const std::string &k) const
    {
        auto it = attr.find(k);
        if (it == attr.end()) logError(format("<%s>: missing attribute \"%s\"", tag.c_str(), k.c_str()));
        return it->second;
    }
    std::string get(const std::string &k, const std::string &def) const { auto it = attr.find(k); return it == attr.end() ? def : it->second; }
};

/// A small non-validating XML parser: elements, attributes, comments, <?...?> declarations.  No entities beyond the basic five.
class Parser {
public:
    explicit Parser(const std::string &text) : s(text), i(0) {}
    std::unique_ptr<Node> parse()
    {
        skipMisc();
        std::unique_ptr<Node> root = element();
        skipMisc();
        return root;
    }

private:
    const std::string &s;
    size_t i;
    [[noreturn]] void fail(const char *what) const
    {
        size_t line = 1;
        for (size_t k = 0; k < i && k < s.size(); ++k) line += s[k] == '\n';
        logError(format("XML parse error near line %zu: %s", line, what));
    }
    void skipWs() { while (i < s.size() && std::isspace((unsigned char)s[i])) ++i; }
    bool starts(const char *p) const { return s.compare(i, std::strlen(p), p) == 0; }
    void skipMisc()
    {
        for (;;) {
            skipWs();
            if (starts("<!--")) { size_t e = s.find("-->", i); if (e == std::string::npos) fail("unterminated comment"); i = e + 3; }
            else if (starts("<?")) { size_t e = s.find("?>", i); if (e == std::string::npos) fail("unterminated declaration"); i = e + 2; }
            else if (starts("<!")) { size_t e = s.find('>', i); if (e == std::string::npos) fail("unterminated <!...>"); i = e + 1; }
            else return;
        }
    }
    std::string name()
    {
        size_t b = i;
        while (i < s.size() && (std::isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '-' || s[i] == ':' || s[i] == '.')) ++i;
        if (b == i) fail("expected a name");
        return s.substr(b, i - b);
    }
    static std::string unescape(const std::string &v)
    {
        std::string o;
        for (size_t k = 0; k < v.size(); ++k) {
            if (v[k] == '&') {
                if (!v.compare(k, 4, "&lt;")) { o += '<'; k += 3; continue; }
                if (!v.compare(k, 4, "&gt;")) { o += '>'; k += 3; continue; }
                if (!v.compare(k, 5, "&amp;")) { o += '&'; k += 4; continue; }
                if (!v.compare(k, 6, "&quot;")) { o += '"'; k += 5; continue; }
                if (!v.compare(k, 6, "&apos;")) { o += '\''; k += 5; continue; }
            }
            o += v[k];
        }
        return o;
    }
    std::unique_ptr<Node> element()
    {
        if (i >= s.size() || s[i] != '<') fail("expected '<'");
        ++i;
        std::unique_ptr<Node> n(new Node);
        n->tag = name();
        for (;;) {
            skipWs();
            if (i >= s.size()) fail("unterminated tag");
            if (s[i] == '/') { if (!starts("/>")) fail("expected '/>'"); i += 2; return n; }
            if (s[i] == '>') { ++i; break; }
            std::string k = name();
            skipWs();
            if (i >= s.size() || s[i] != '=') fail("expected '='");
            ++i;
            skipWs();
            if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) fail("expected a quoted attribute value");
            const char q = s[i++];
            size_t e = s.find(q, i);
            if (e == std::string::npos) fail("unterminated attribute value");
            n->attr[k] = unescape(s.substr(i, e - i));
            i = e + 1;
        }
        for (;;) {
            skipMisc();
            if (i >= s.size()) fail("unterminated element");
            if (starts("</")) {
                i += 2;
                if (name() != n->tag) fail("mismatched closing tag");
                skipWs();
                if (i >= s.size() || s[i] != '>') fail("expected '>'");
                ++i;
                return n;
            }
            if (s[i] == '<') n->children.push_back(element());
            else { while (i < s.size() && s[i] != '<') ++i; }      // character data is not used by the scene grammar
        }
    }
};

} // namespace xml

struct Mat4 {
    double m[16];
    static Mat4 identity() { Mat4 r; for (int k = 0; k < 16; ++k) r.m[k] = (k % 5 == 0); return r; }
    Mat4 operator*(const Mat4 &o) const
    {
        Mat4 r;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) { double v = 0; for (int k = 0; k < 4; ++k) v += m[4 * a + k] * o.m[4 * k + b]; r.m[4 * a + b] = v; }
        return r;
    }
    void point(const double p[3], double out[3]) const
    {
        double w = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
        for (int a = 0; a < 3; ++a) out[a] = (m[4 * a] * p[0] + m[4 * a + 1] * p[1] + m[4 * a + 2] * p[2] + m[4 * a + 3]) / w;
    }
    double det3() const
    {
        return m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    }
    void normal(const double n[3], double out[3]) const       // Transform::operator()(Normal): inverse transpose of the linear part
    {
        const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
        const double inv = 1.0 / det3();
        const double C[9] = {(e * i - f * h) * inv, (f * g - d * i) * inv, (d * h - e * g) * inv,
                             (c * h - b * i) * inv, (a * i - c * g) * inv, (b * g - a * h) * inv,
                             (b * f - c * e) * inv, (c * d - a * f) * inv, (a * e - b * d) * inv};
        for (int r = 0; r < 3; ++r) out[r] = C[3 * r] * n[0] + C[3 * r + 1] * n[1] + C[3 * r + 2] * n[2];
    }
};

class SceneLoader {
public:
    std::map<std::string, std::string> params;      // -D key=value and <default>

    SceneData load(const std::string &path)
    {
        std::ifstream f(path);
        if (!f) logError(format("Cannot open scene file \"%s\"", path.c_str()));
        std::stringstream ss;
        ss << f.rdbuf();
        const size_t slash = path.find_last_of('/');
        m_dir = slash == std::string::npos ? "." : path.substr(0, slash);
        return loadString(ss.str());
    }

    SceneData loadString(const std::string &text)
    {
        xml::Parser parser(text);
        std::unique_ptr<xml::Node> root = parser.parse();
        if (root->tag != "scene") logError("root element must be <scene>");
        SceneData sd;
        std::memset(&sd.camera, 0, sizeof sd.camera);
        bool haveSensor = false, haveIntegrator = false;
        for (auto &c : root->children) {
            const xml::Node &n = *c;
            if (n.tag == "default") { if (!params.count(n.get("name"))) params[n.get("name")] = subst(n.get("value")); }
            else if (n.tag == "integrator") {
                if (subst(n.get("type")) != "gpt" && subst(n.get("type")) != "gbdpt") logError(format("integrator \"%s\" is not carried: this build is the gpt / gbdpt hot path only", n.get("type").c_str()));
                sd.integrator = props(n);
                haveIntegrator = true;
            } else if (n.tag == "sensor") { sensor(n, sd); haveSensor = true; }
            else if (n.tag == "bsdf") { const std::string id = n.get("id", ""); const int idx = bsdf(n, sd); if (!id.empty()) m_bsdfIds[id] = idx; }
            else if (n.tag == "shape") shape(n, sd);
            else if (n.tag == "emitter") {                                       // src/emitters/constant.cpp: the only top-level emitter carried
                if (subst(n.get("type")) == "point") {                           // src/emitters/point.cpp
                    gdpt_emitter e;
                    std::memset(&e, 0, sizeof e);
                    e.numTris = -1;
                    double intensity[3] = {1.0, 1.0, 1.0};
                    bool havePos = false, haveXf = false;
                    for (auto &ec : n.children) {
                        if ((ec->tag == "rgb" || ec->tag == "spectrum") && ec->get("name") == "intensity") rgb3(*ec, intensity);
                        else if (ec->tag == "point" && ec->get("name") == "position") { e.position[0] = std::atof(subst(ec->get("x", "0")).c_str()); e.position[1] = std::atof(subst(ec->get("y", "0")).c_str()); e.position[2] = std::atof(subst(ec->get("z", "0")).c_str()); havePos = true; }
                        else if (ec->tag == "transform" && ec->get("name") == "toWorld") { const Mat4 T = transform(*ec); const double o[3] = {0, 0, 0}; T.point(o, e.position); haveXf = true; }
                        else if (ec->tag == "float" && ec->get("name") == "samplingWeight") { if (std::atof(subst(ec->get("value")).c_str()) != 1.0) logError("emitter \"point\": samplingWeight other than 1 is not carried"); }
                        else logError(format("emitter \"point\": <%s name=\"%s\"> is not carried", ec->tag.c_str(), ec->get("name", "").c_str()));
                    }
                    if (havePos && haveXf) logError("Only one of the parameters 'position' and 'toWorld' can be used!'");          // point.cpp:61-63
                    for (int k = 0; k < 3; ++k) e.radiance[k] = intensity[k];
                    sd.emitters.push_back(e);
                    continue;
                }
                if (subst(n.get("type")) == "envmap") {                          // src/emitters/envmap.cpp: filename, scale, gamma, toWorld
                    if (sd.hasEnvironment) logError("Only one environment emitter can be used at a time!");
                    std::string filename;
                    double scale = 1.0, gamma = 0.0;
                    Mat4 T = Mat4::identity();
                    for (auto &ec : n.children) {
                        const std::string nm = ec->get("name", ""), v = subst(ec->get("value", ""));
                        if (ec->tag == "string" && nm == "filename") filename = v;
                        else if (ec->tag == "float" && nm == "scale") scale = std::stod(v);
                        else if (ec->tag == "float" && nm == "gamma") gamma = std::stod(v);
                        else if (ec->tag == "transform" && nm == "toWorld") T = transform(*ec);
                        else if (ec->tag == "boolean" && nm == "cache") {}
                        else if (ec->tag == "float" && nm == "samplingWeight") { if (std::stod(v) != 1.0) logError("emitter \"envmap\": samplingWeight other than 1 is not carried"); }
                        else logError(format("emitter \"envmap\": <%s name=\"%s\"> is not carried", ec->tag.c_str(), nm.c_str()));
                    }
                    if (filename.empty()) logError("emitter \"envmap\": missing filename");
                    SceneData::Texture img;
                    loadImage(filename[0] == '/' ? filename : m_dir + "/" + filename, gamma, img);
                    sd.envmapRgb = img.rgb;
                    sd.hasEnvironment = true;
                    sd.environment.width = img.width; sd.environment.height = img.height; sd.environment.scale = scale;
                    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) sd.environment.toWorld[3 * r + c] = T.m[4 * r + c];
                    sd.environment.index = (int)sd.emitters.size();
                    continue;
                }
                if (subst(n.get("type")) != "constant") logError(format("top-level emitter \"%s\" is not carried: `constant`, `envmap`, `point` (and `area` on shapes)", n.get("type").c_str()));
                if (sd.hasEnvironment) logError("Only one environment emitter can be used at a time!");          // scene.cpp: addChild
                double radiance[3] = {1.0, 1.0, 1.0};                            // constant.cpp:44: default radiance = D65 white
                for (auto &ec : n.children) {
                    if ((ec->tag == "rgb" || ec->tag == "spectrum") && ec->get("name") == "radiance") rgb3(*ec, radiance);
                    else if (ec->tag == "float" && ec->get("name") == "samplingWeight") { if (std::atof(subst(ec->get("value")).c_str()) != 1.0) logError("emitter \"constant\": samplingWeight other than 1 is not carried"); }
                    else logError(format("emitter \"constant\": <%s name=\"%s\"> is not carried", ec->tag.c_str(), ec->get("name", "").c_str()));
                }
                sd.hasEnvironment = true;
                for (int k = 0; k < 3; ++k) sd.environment.radiance[k] = radiance[k];
                sd.environment.index = (int)sd.emitters.size();                  // its place in the emitter list = XML order
            }
            else logError(format("<%s> is not carried by this build", n.tag.c_str()));
        }
        if (!haveIntegrator) sd.integrator = Properties("gpt");
        if (!haveSensor) logError("scene has no <sensor>");
        if (sd.emitters.empty() && !sd.hasEnvironment) logError("scene has no emitter");
        return sd;
    }

private:
    std::string m_dir;
    std::map<std::string, int> m_bsdfIds;

    std::string subst(const std::string &v) const
    { // $name substitution (scenehandler.cpp: parameter map from -D and <default>)
        std::string o;
        for (size_t k = 0; k < v.size(); ++k) {
            if (v[k] == '$') {
                size_t e = k + 1;
                while (e < v.size() && (std::isalnum((unsigned char)v[e]) || v[e] == '_')) ++e;
                const std::string key = v.substr(k + 1, e - k - 1);
                auto it = params.find(key);
                if (it == params.end()) logError(format("The scene references an undefined parameter \"$%s\" (use -D %s=...)", key.c_str(), key.c_str()));
                o += it->second;
                k = e - 1;
            } else o += v[k];
        }
        return o;
    }

    static std::vector<double> numbers(const std::string &v)
    {
        std::string t = v;
        for (char &c : t) if (c == ',') c = ' ';
        std::stringstream ss(t);
        std::vector<double> out;
        double d;
        while (ss >> d) out.push_back(d);
        return out;
    }

    void rgb3(const xml::Node &n, double out[3]) const
    {
        std::vector<double> v = numbers(subst(n.get("value")));
        if (v.size() == 1) out[0] = out[1] = out[2] = v[0];
        else if (v.size() == 3) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; }
        else logError(format("<%s name=\"%s\">: expected 1 or 3 values (wavelength:value spectra need data files this build does not carry)", n.tag.c_str(), n.get("name").c_str()));
    }

    Properties props(const xml::Node &n) const
    {
        Properties p(subst(n.get("type", "")));
        p.setID(n.get("id", ""));
        for (auto &c : n.children) {
            const std::string &t = c->tag;
            if (t == "integer" || t == "float" || t == "boolean" || t == "string") p.setString(c->get("name"), subst(c->get("value")));
        }
        return p;
    }

    Mat4 transform(const xml::Node &n) const
    { // every child left-multiplies the accumulated transform (scenehandler.cpp: m_transform = op * m_transform)
        Mat4 T = Mat4::identity();
        for (auto &c : n.children) {
            Mat4 M = Mat4::identity();
            auto num = [&](const char *k, double def) { auto it = c->attr.find(k); return it == c->attr.end() ? def : std::stod(subst(it->second)); };
            if (c->tag == "translate") { M.m[3] = num("x", 0); M.m[7] = num("y", 0); M.m[11] = num("z", 0); }
            else if (c->tag == "scale") {
                if (c->attr.count("value")) { const double v = num("value", 1); M.m[0] = M.m[5] = M.m[10] = v; }
                else { M.m[0] = num("x", 1); M.m[5] = num("y", 1); M.m[10] = num("z", 1); }
            } else if (c->tag == "rotate") {
                double ax = num("x", 0), ay = num("y", 0), az = num("z", 0);
                const double len = std::sqrt(ax * ax + ay * ay + az * az);
                if (len == 0) logError("<rotate>: zero axis");
                ax /= len; ay /= len; az /= len;
                const double a = num("angle", 0) * M_PI / 180.0, s = std::sin(a), co = std::cos(a);     // Transform::rotate, transform.cpp
                M.m[0] = ax * ax + (1 - ax * ax) * co; M.m[1] = ax * ay * (1 - co) - az * s; M.m[2] = ax * az * (1 - co) + ay * s;
                M.m[4] = ax * ay * (1 - co) + az * s; M.m[5] = ay * ay + (1 - ay * ay) * co; M.m[6] = ay * az * (1 - co) - ax * s;
                M.m[8] = ax * az * (1 - co) - ay * s; M.m[9] = ay * az * (1 - co) + ax * s; M.m[10] = az * az + (1 - az * az) * co;
            } else if (c->tag == "lookat" || c->tag == "lookAt") {
                std::vector<double> o = numbers(subst(c->get("origin"))), t = numbers(subst(c->get("target"))), u = numbers(subst(c->get("up", "0, 1, 0")));
                if (o.size() != 3 || t.size() != 3 || u.size() != 3) logError("<lookat>: origin/target/up need 3 values each");
                double d[3] = {t[0] - o[0], t[1] - o[1], t[2] - o[2]};
                double dl = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                if (dl == 0) logError("lookAt(): 'origin' and 'target' coincide!");
                for (double &x : d) x /= dl;
                double l[3] = {u[1] * d[2] - u[2] * d[1], u[2] * d[0] - u[0] * d[2], u[0] * d[1] - u[1] * d[0]};   // cross(up, dir)
                double ll = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
                if (ll == 0) logError("lookAt(): the forward and upward direction must be linearly independent!");
                for (double &x : l) x /= ll;
                double nu[3] = {d[1] * l[2] - d[2] * l[1], d[2] * l[0] - d[0] * l[2], d[0] * l[1] - d[1] * l[0]};   // cross(dir, left)
                for (int a = 0; a < 3; ++a) { M.m[4 * a] = l[a]; M.m[4 * a + 1] = nu[a]; M.m[4 * a + 2] = d[a]; M.m[4 * a + 3] = o[a]; }
            } else if (c->tag == "matrix") {
                std::vector<double> v = numbers(subst(c->get("value")));
                if (v.size() != 16) logError("<matrix>: expected 16 values");
                for (int k = 0; k < 16; ++k) M.m[k] = v[k];
            } else logError(format("<%s> inside <transform> is not carried", c->tag.c_str()));
            T = M * T;
        }
        return T;
    }

    void sensor(const xml::Node &n, SceneData &sd)
    {
        const std::string stype = subst(n.get("type"));
        if (stype != "perspective" && stype != "thinlens") logError(format("sensor \"%s\" is not carried: `perspective`, `thinlens`", n.get("type").c_str()));
        Properties p = props(n);
        Mat4 toWorld = Mat4::identity();
        bool haveFilm = false;
        for (auto &c : n.children) {
            if (c->tag == "transform" && c->get("name") == "toWorld") toWorld = transform(*c);
            else if (c->tag == "sampler") {
                if (subst(c->get("type")) != "independent") logError(format("sampler \"%s\" is not carried: `independent` only (G-PT never calls advance(), SURVEY.md 2a)", c->get("type").c_str()));
                sd.sampler = props(*c);
            } else if (c->tag == "film") {
                if (subst(c->get("type")) != "multifilm") logError("Cannot render image! G-PT has been called without MultiFilm.");   // gpt.cpp:1381-1384
                sd.film = props(*c);
                sd.rfilter = Properties("gaussian");                             // Film's default reconstruction filter, film.cpp:93
                for (auto &fc : c->children)
                    if (fc->tag == "rfilter") {
                        const std::string ft = subst(fc->get("type"));
                        if (ft != "box" && ft != "tent" && ft != "gaussian" && ft != "mitchell" && ft != "catmullrom" && ft != "lanczos")
                            logError(format("rfilter \"%s\" is not carried: box, tent, gaussian, mitchell, catmullrom, lanczos", ft.c_str()));
                        sd.rfilter = props(*fc);
                    }
                haveFilm = true;
            }
        }
        if (!haveFilm) logError("the sensor has no <film type=\"multifilm\">");
        const int W = sd.film.getInteger("width", 768), H = sd.film.getInteger("height", 576);
        double fov = p.getFloat("fov", 50.0);
        const std::string axis = p.getString("fovAxis", "x");
        if (axis == "y") fov = 2 * std::atan(std::tan(fov * M_PI / 360.0) * (double)W / H) * 180.0 / M_PI;
        else if (axis != "x") logError(format("fovAxis \"%s\" is not carried (x or y)", axis.c_str()));
        for (int k = 0; k < 16; ++k) sd.camera.toWorld[k] = toWorld.m[k];
        sd.camera.fovX = fov;
        sd.camera.nearClip = p.getFloat("nearClip", 1e-2);
        sd.camera.farClip = p.getFloat("farClip", 1e4);
        // the film's crop window (film.cpp:34-48): the image that is rendered and written is the crop; the sensor's rays come from the full film's raster
        const int cropX = sd.film.getInteger("cropOffsetX", 0), cropY = sd.film.getInteger("cropOffsetY", 0);
        const int cropW = sd.film.getInteger("cropWidth", W), cropH = sd.film.getInteger("cropHeight", H);
        if (cropX < 0 || cropY < 0 || cropW <= 0 || cropH <= 0 || cropX + cropW > W || cropY + cropH > H) logError("Invalid crop window specification!");   // film.cpp:44-48
        sd.camera.width = cropW;
        sd.camera.height = cropH;
        if (cropX != 0 || cropY != 0 || cropW != W || cropH != H) { sd.camera.cropOffsetX = cropX; sd.camera.cropOffsetY = cropY; sd.camera.fullWidth = W; sd.camera.fullHeight = H; }
        sd.camera.shutterOpen = p.getFloat("shutterOpen", 0.0);            // Sensor::Sensor, sensor.cpp:28-30
        sd.camera.shutterClose = p.getFloat("shutterClose", 0.0);
        if (sd.camera.shutterClose < sd.camera.shutterOpen) logError("Shutter opening time must be less than or equal to the shutter closing time!");   // sensor.cpp:33-35
        if (stype == "thinlens") {                                       // thinlens.cpp:236-244, sensor.cpp (ProjectiveCamera: focusDistance, default farClip)
            sd.camera.type = GDPT_SENSOR_THINLENS;
            sd.camera.apertureRadius = p.getFloat("apertureRadius", 0.0);
            if (sd.camera.apertureRadius == 0.0) sd.camera.apertureRadius = 1e-7;        // "Can't have a zero aperture radius -- setting to Epsilon", thinlens.cpp:134-138
            sd.camera.focusDistance = p.getFloat("focusDistance", sd.camera.farClip);
        }
    }

    int bsdf(const xml::Node &n, SceneData &sd)
    {
        if (subst(n.get("type")) == "twosided") {            // src/bsdfs/twosided.cpp with ONE nested BRDF (used for both faces)
            const xml::Node *inner = nullptr;
            for (auto &c : n.children)
                if (c->tag == "bsdf") { if (inner) logError("twosided: a second nested BRDF (different front/back models) is not carried"); inner = c.get(); }
            if (!inner) logError("A nested one-sided material is required!");                        // twosided.cpp:85
            if (subst(inner->get("type")) == "twosided") logError("twosided inside twosided is not carried");
            if (subst(inner->get("type")) == "dielectric") logError("Only materials without a transmission component can be nested!");   // twosided.cpp:96-98
            const int idx = bsdf(*inner, sd);
            sd.materials[idx].twoSided = 1;
            return idx;
        }
        const std::string type = subst(n.get("type"));
        if (type == "dielectric") return dielectric(n, sd);
        gdpt_material m;
        std::memset(&m, 0, sizeof m);
        m.sampleVisible = 1;
        bool haveEta = false, haveK = false;
        int texIndex = -1;
        for (int c = 0; c < 3; ++c) { m.reflectance[c] = type == "diffuse" ? 0.5 : 1.0; m.eta[c] = 0; m.k[c] = 1; }
        m.alphaU = m.alphaV = 0.1;
        for (auto &c : n.children) {
            const std::string nm = c->get("name", "");
            if (c->tag == "rgb" || c->tag == "spectrum" || c->tag == "srgb") {
                double v[3];
                rgb3(*c, v);
                double *dst = nullptr;
                if (nm == "reflectance" || nm == "diffuseReflectance" || nm == "specularReflectance") dst = m.reflectance;
                else if (nm == "eta") { dst = m.eta; haveEta = true; }
                else if (nm == "k") { dst = m.k; haveK = true; }
                else logError(format("bsdf \"%s\": parameter \"%s\" is not carried", type.c_str(), nm.c_str()));
                for (int k = 0; k < 3; ++k) dst[k] = v[k];
            } else if (c->tag == "float") {
                const double v = std::stod(subst(c->get("value")));
                if (nm == "alpha") m.alphaU = m.alphaV = v;
                else if (nm == "alphaU") m.alphaU = v;
                else if (nm == "alphaV") m.alphaV = v;
                else if (nm == "eta") { m.eta[0] = m.eta[1] = m.eta[2] = v; haveEta = true; }
                else if (nm == "k") { m.k[0] = m.k[1] = m.k[2] = v; haveK = true; }
                else logError(format("bsdf \"%s\": parameter \"%s\" is not carried", type.c_str(), nm.c_str()));
            } else if (c->tag == "string") {
                const std::string v = subst(c->get("value"));
                if (nm == "distribution") {
                    if (v == "beckmann") m.distribution = GDPT_DISTR_BECKMANN;
                    else if (v == "ggx") m.distribution = GDPT_DISTR_GGX;
                    else if (v == "phong" || v == "as") m.distribution = GDPT_DISTR_PHONG;          // microfacet.h:108-112
                    else logError("Specified an invalid distribution \"" + v + "\", must be \"beckmann\", \"ggx\", or \"phong\"/\"as\"!");
                } else if (nm == "material") logError("conductor `material` presets need Mitsuba's data/ior tables, which this build does not carry: give explicit eta and k");
                else logError(format("bsdf \"%s\": parameter \"%s\" is not carried", type.c_str(), nm.c_str()));
            } else if (c->tag == "boolean") {
                if (nm == "sampleVisible") m.sampleVisible = subst(c->get("value")) == "true";
                else logError(format("bsdf \"%s\": parameter \"%s\" is not carried", type.c_str(), nm.c_str()));
            } else if (c->tag == "texture") {
                if (nm != "reflectance" && nm != "diffuseReflectance" && nm != "specularReflectance")
                    logError(format("bsdf \"%s\": a texture on \"%s\" is not carried (reflectance / specularReflectance only)", type.c_str(), nm.c_str()));
                texIndex = texture(*c, sd);
            }
            else if (c->tag == "bsdf") logError(format("nested BSDFs (\"%s\") are not carried", type.c_str()));
        }
        if (type == "diffuse") m.type = GDPT_MAT_DIFFUSE;
        else if (type == "conductor") m.type = GDPT_MAT_CONDUCTOR;
        else if (type == "roughconductor") m.type = GDPT_MAT_ROUGHCONDUCTOR;
        else logError(format("bsdf \"%s\" is not carried: diffuse, conductor, roughconductor, dielectric, twosided", type.c_str()));
        if (m.type != GDPT_MAT_DIFFUSE && !(haveEta && haveK)) logError(format("bsdf \"%s\": explicit eta and k are required (the default `material=Cu` needs data/ior)", type.c_str()));
        sd.materials.push_back(m);
        if (texIndex >= 0) {
            sd.materialTexture.resize(sd.materials.size(), -1);
            sd.materialTexture.back() = texIndex;
            // BSDF::ensureEnergyConservation (bsdf.cpp:88-113; diffuse.cpp:95, conductor.cpp / roughconductor.cpp likewise): a reflectance
            // texture whose maximum exceeds 1 is wrapped in a ScaleTexture of 0.99f / max
            SceneData::Texture &t = sd.textures[texIndex];
            double mx = 0.0;
            for (double v : t.rgb) mx = std::max(mx, v);
            t.scale = mx > 1.0 ? (double)0.99f * (1.0 / mx) : 1.0;
        }
        return (int)sd.materials.size() - 1;
    }

    // ---- `<texture type="bitmap">` (src/textures/bitmap.cpp) ------------------------------------------------------------------------
    // Carried: filterType ewa (the default) | trilinear | bilinear | nearest with maxAnisotropy (the library builds the MIP pyramid and
    // filters the lookups at camera-ray hits), wrapMode / wrapModeU / wrapModeV, gamma, uscale / vscale / uoffset / voffset (Texture2D,
    // texture.cpp:27-45).  Files: PFM and uncompressed OpenEXR (linear floats), binary PPM and 8-bit PNG (sRGB unless `gamma` says otherwise),
    // converted to Float as Bitmap::convert does (fmtconv.cpp:1137-1160: value/255 with the float reciprocal, then the sRGB curve).
    int texture(const xml::Node &n, SceneData &sd)
    {
        const std::string type = subst(n.get("type"));
        if (type != "bitmap") logError(format("texture \"%s\" is not carried: bitmap", type.c_str()));
        SceneData::Texture t;
        std::string filename, filterType = "ewa", wrapMode = "repeat", wrapU, wrapV;
        double gamma = 0;
        for (auto &c : n.children) {
            const std::string nm = c->get("name", ""), v = subst(c->get("value", ""));
            if (c->tag == "string" && nm == "filename") filename = v;
            else if (c->tag == "string" && nm == "filterType") { filterType = v; for (char &ch : filterType) ch = (char)std::tolower((unsigned char)ch); }
            else if (c->tag == "string" && nm == "wrapMode") wrapMode = v;
            else if (c->tag == "string" && nm == "wrapModeU") wrapU = v;
            else if (c->tag == "string" && nm == "wrapModeV") wrapV = v;
            else if (c->tag == "float" && nm == "gamma") gamma = std::stod(v);
            else if (c->tag == "float" && nm == "uscale") t.uscale = std::stod(v);
            else if (c->tag == "float" && nm == "vscale") t.vscale = std::stod(v);
            else if (c->tag == "float" && nm == "uoffset") t.uoffset = std::stod(v);
            else if (c->tag == "float" && nm == "voffset") t.voffset = std::stod(v);
            else if (c->tag == "float" && nm == "maxAnisotropy") t.maxAnisotropy = std::stod(v);     // bitmap.cpp:232
            else if (c->tag == "boolean" && nm == "cache") {}                              // the MIP-map cache file: nothing to cache here
            else logError(format("texture \"bitmap\": <%s name=\"%s\"> is not carried", c->tag.c_str(), nm.c_str()));
        }
        if (filterType == "nearest") t.filter = GDPT_TEXFILTER_NEAREST;
        else if (filterType == "bilinear") t.filter = GDPT_TEXFILTER_BILINEAR;
        else if (filterType == "trilinear") t.filter = GDPT_TEXFILTER_TRILINEAR;
        else if (filterType == "ewa") t.filter = GDPT_TEXFILTER_EWA;
        else logError(format("Invalid filter type '%s' -- must be 'ewa', 'trilinear', or 'nearest'!", filterType.c_str()));      // bitmap.cpp:229-230
        auto wrapOf = [&](const std::string &w) {
            if (w == "repeat") return GDPT_TEXWRAP_REPEAT;
            if (w == "clamp") return GDPT_TEXWRAP_CLAMP;
            if (w == "mirror") return GDPT_TEXWRAP_MIRROR;
            if (w == "zero" || w == "black") return GDPT_TEXWRAP_ZERO;
            if (w == "one" || w == "white") return GDPT_TEXWRAP_ONE;
            logError(format("Invalid wrap mode '%s' -- must be 'repeat', 'clamp', 'black', or 'white'!", w.c_str()));           // bitmap.cpp:336-337
        };
        t.wrapU = wrapOf(wrapU.empty() ? wrapMode : wrapU);
        t.wrapV = wrapOf(wrapV.empty() ? wrapMode : wrapV);
        if (filename.empty()) logError("texture \"bitmap\": missing filename");
        loadImage(filename[0] == '/' ? filename : m_dir + "/" + filename, gamma, t);
        sd.textures.push_back(std::move(t));
        return (int)sd.textures.size() - 1;
    }

    // undoGamma of fmtconv.cpp:1093-1102 (gamma == -1: the sRGB curve)
    static double undoGamma(double value, double gamma)
    {
        if (gamma == -1) return value <= 0.04045 ? value * (1.0 / 12.92) : std::pow((value + 0.055) * (1.0 / 1.055), 2.4);
        return std::pow(value, gamma);
    }

    // -> t.width, t.height, t.rgb (linear, top row first).  fileGamma: what the format implies (-1 sRGB for the 8-bit formats, 1 for floats);
    // `gamma` != 0 overrides it (bitmap.cpp:251-252).
public:
    void readBitmap(const std::string &path, SceneData::Texture &t, double gamma) { loadImage(path, gamma, t); }   // the texture / environment-map reader on its own (gdpt_mitsuba --tex2pfm)
private:
    void loadImage(const std::string &path, double gamma, SceneData::Texture &t)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) logError(format("Cannot open texture \"%s\"", path.c_str()));
        std::vector<unsigned char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        auto from8 = [&](const std::vector<unsigned char> &px, int w, int h, int channels) {          // fmtconv.cpp:1137-1160 (uint8 source, Float destination)
            const double g = gamma != 0 ? gamma : -1.0;
            double table[256];
            for (int i = 0; i < 256; ++i) { double v = (double)i * (double)(1.0f / 255); if (g != 1) v = undoGamma(v, g); table[i] = v; }
            t.width = w; t.height = h; t.rgb.resize((size_t)3 * w * h);
            for (size_t i = 0; i < (size_t)w * h; ++i)
                for (int c = 0; c < 3; ++c) t.rgb[3 * i + c] = table[px[i * channels + (channels >= 3 ? c : 0)]];
        };
        if (data.size() >= 2 && data[0] == 'P' && (data[1] == 'F' || data[1] == '6')) {
            size_t at = 2;
            auto token = [&]() { while (at < data.size() && std::isspace(data[at])) ++at; if (at < data.size() && data[at] == '#') { while (at < data.size() && data[at] != '\n') ++at; while (at < data.size() && std::isspace(data[at])) ++at; }
                                 std::string tok; while (at < data.size() && !std::isspace(data[at])) tok += (char)data[at++]; return tok; };
            const int w = std::atoi(token().c_str()), h = std::atoi(token().c_str());
            const double third = std::atof(token().c_str());
            ++at;                                                                                     // the single whitespace after the header
            if (w <= 0 || h <= 0) logError(path + ": bad image header");
            if (data[1] == 'F') {                                                                     // PFM: bottom row first, little endian when the scale is negative
                if (!(third < 0)) logError(path + ": big-endian PFM is not carried");
                if (at + sizeof(float) * 3 * (size_t)w * h > data.size()) logError(path + ": truncated");
                t.width = w; t.height = h; t.rgb.resize((size_t)3 * w * h);
                for (int y = 0; y < h; ++y)
                    for (int x = 0; x < 3 * w; ++x) { float v; std::memcpy(&v, &data[at + sizeof(float) * ((size_t)(h - 1 - y) * 3 * w + x)], sizeof v); t.rgb[(size_t)y * 3 * w + x] = gamma != 0 && gamma != 1 ? undoGamma((double)v, gamma) : (double)v; }
            } else {
                if ((int)third != 255) logError(path + ": only 8-bit PPM is carried");
                if (at + (size_t)3 * w * h > data.size()) logError(path + ": truncated");
                from8(std::vector<unsigned char>(data.begin() + at, data.begin() + at + (size_t)3 * w * h), w, h, 3);
            }
            return;
        }
        if (data.size() >= 8 && data[0] == 0x89 && data[1] == 'P' && data[2] == 'N' && data[3] == 'G') {
            // PNG: 8-bit greyscale / RGB / RGBA (+ grey-alpha), no interlace; zlib inflate + the five scanline filters
            auto be32 = [&](size_t o) { if (o + 4 > data.size()) logError(path + ": truncated"); return ((unsigned)data[o] << 24) | ((unsigned)data[o + 1] << 16) | ((unsigned)data[o + 2] << 8) | data[o + 3]; };
            size_t at = 8;
            unsigned w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
            std::vector<unsigned char> z;
            while (at + 12 <= data.size()) {
                const unsigned len = be32(at);
                const std::string tag(data.begin() + at + 4, data.begin() + at + 8);
                if (at + 12 + len > data.size()) logError(path + ": truncated");
                if (tag == "IHDR") { w = be32(at + 8); h = be32(at + 12); depth = data[at + 16]; ctype = data[at + 17]; interlace = data[at + 20]; }
                else if (tag == "IDAT") z.insert(z.end(), data.begin() + at + 8, data.begin() + at + 8 + len);
                else if (tag == "IEND") break;
                at += 12 + len;
            }
            const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
            if (depth != 8 || !channels || interlace || !w || !h) logError(path + ": only 8-bit non-interlaced grey / RGB / RGBA PNG files are carried");
            const size_t stride = (size_t)w * channels;
            std::vector<unsigned char> raw((stride + 1) * h);
            uLongf outLen = (uLongf)raw.size();
            if (uncompress(raw.data(), &outLen, z.data(), (uLong)z.size()) != Z_OK || outLen != raw.size()) logError(path + ": PNG data is corrupt");
            std::vector<unsigned char> px(stride * h);
            for (unsigned y = 0; y < h; ++y) {
                const unsigned char *in = &raw[(stride + 1) * y + 1], *up = y ? &px[stride * (y - 1)] : nullptr;
                unsigned char *out = &px[stride * y];
                const int ft = raw[(stride + 1) * y];
                for (size_t i = 0; i < stride; ++i) {
                    const int a = i >= (size_t)channels ? out[i - channels] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)channels) ? up[i - channels] : 0;
                    int pred = 0;
                    if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) / 2;
                    else if (ft == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
                    else if (ft != 0) logError(path + ": PNG data is corrupt");
                    out[i] = (unsigned char)(in[i] + pred);
                }
            }
            if (channels == 2 || channels == 4) {                                                     // drop alpha (the texture reads the colour channels: bitmap.cpp:268-275)
                const int keep = channels - 1;
                std::vector<unsigned char> q((size_t)w * h * keep);
                for (size_t i = 0; i < (size_t)w * h; ++i) for (int c = 0; c < keep; ++c) q[i * keep + c] = px[i * channels + c];
                from8(q, (int)w, (int)h, keep);
            } else from8(px, (int)w, (int)h, channels);
            return;
        }
        if (data.size() >= 4 && data[0] == 0x76 && data[1] == 0x2f && data[2] == 0x31 && data[3] == 0x01) {
            // OpenEXR, the subset this build writes itself (uncompressed scanlines, half or float channels B G R in file order)
            size_t at = 8;
            int w = 0, h = 0, minY = 0, compression = -1;
            std::vector<std::pair<std::string, int>> channels;                                        // name, pixel type (1 half, 2 float)
            auto cstr = [&]() { std::string r; while (at < data.size() && data[at]) r += (char)data[at++]; ++at; return r; };
            auto le32 = [&](size_t o) { if (o + 4 > data.size()) logError(path + ": truncated"); unsigned v; std::memcpy(&v, &data[o], 4); return v; };
            while (at < data.size() && data[at]) {
                const std::string name = cstr(), atype = cstr();
                const unsigned size = le32(at); at += 4;
                if (name == "channels") { size_t p = at; while (data[p]) { std::string cn; while (data[p]) cn += (char)data[p++]; ++p; const int pt = (int)le32(p); p += 16; channels.emplace_back(cn, pt); } }
                else if (name == "compression") compression = data[at];
                else if (name == "dataWindow") { w = (int)le32(at + 8) - (int)le32(at) + 1; h = (int)le32(at + 12) - (int)le32(at + 4) + 1; minY = (int)le32(at + 4); }
                at += size;
            }
            ++at;
            if ((compression != 0 && compression != 2 && compression != 3) || w <= 0 || h <= 0 || channels.empty())
                logError(path + ": only uncompressed, ZIPS- and ZIP-compressed scanline OpenEXR files are carried (PIZ / PXR24 / B44 / DWA are not)");
            size_t rowBytes = 0;
            for (auto &c : channels) rowBytes += (size_t)w * (c.second == 1 ? 2 : 4);
            t.width = w; t.height = h; t.rgb.assign((size_t)3 * w * h, 0.0);
            auto half2d = [](unsigned short hv) { const int s = hv >> 15, e = (hv >> 10) & 31, m = hv & 1023; double v = e == 0 ? std::ldexp((double)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : std::ldexp((double)(m + 1024), e - 25)); return s ? -v : v; };
            const int lines = compression == 3 ? 16 : 1, chunks = (h + lines - 1) / lines;
            std::vector<unsigned char> raw, tmp;
            for (int cidx = 0; cidx < chunks; ++cidx) {
                const size_t off = (size_t)(le32(at + 8 * (size_t)cidx)) | ((size_t)le32(at + 8 * (size_t)cidx + 4) << 32);
                const int y0 = (int)le32(off) - minY, y1 = std::min(h, y0 + lines);      // (a chunk stores its y in dataWindow coordinates: cropped renders start above 0)
                const size_t sz = le32(off + 4), want = (size_t)(y1 - y0) * rowBytes;
                if (y0 < 0 || y0 >= h || off + 8 + sz > data.size()) logError(path + ": truncated");
                raw.assign(want, 0);
                if (compression == 0 || sz == want) { if (sz != want) logError(path + ": truncated"); std::memcpy(raw.data(), &data[off + 8], want); }
                else {                                                                                // ImfZip.cpp: zlib, undo the predictor, interleave the two halves
                    tmp.assign(want, 0);
                    uLongf outLen = (uLongf)want;
                    if (uncompress(tmp.data(), &outLen, &data[off + 8], (uLong)sz) != Z_OK || outLen != want) logError(path + ": OpenEXR data is corrupt");
                    for (size_t i = 1; i < want; ++i) tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);
                    const unsigned char *t1 = tmp.data(), *t2 = tmp.data() + (want + 1) / 2;
                    for (size_t i = 0; i < want; ++i) raw[i] = (i & 1) ? *t2++ : *t1++;
                }
                size_t p = 0;
                for (int y = y0; y < y1; ++y)
                    for (auto &c : channels) {
                        const int dst = c.first == "R" ? 0 : c.first == "G" ? 1 : c.first == "B" ? 2 : (c.first == "Y" ? 3 : -1);
                        for (int x = 0; x < w; ++x) {
                            double v;
                            if (c.second == 1) { unsigned short hv; std::memcpy(&hv, &raw[p], 2); p += 2; v = half2d(hv); }
                            else { float fv; std::memcpy(&fv, &raw[p], 4); p += 4; v = (double)fv; }
                            if (gamma != 0 && gamma != 1) v = undoGamma(v, gamma);
                            if (dst == 3) for (int k = 0; k < 3; ++k) t.rgb[((size_t)y * w + x) * 3 + k] = v;
                            else if (dst >= 0) t.rgb[((size_t)y * w + x) * 3 + dst] = v;
                        }
                    }
            }
            return;
        }
        if (data.size() >= 3 && data[0] == 0xff && data[1] == 0xd8 && data[2] == 0xff)
            logError(format("texture \"%s\": JPEG files are not carried (convert to PNG / EXR)", path.c_str()));
        logError(format("texture \"%s\": the file format is not carried (PFM, PPM, 8-bit PNG, uncompressed / ZIP OpenEXR)", path.c_str()));
    }

    /// lookupIOR (src/bsdfs/ior.h:40-80): a number, or one of the named media (float literals there, hence the casts)
    static double lookupIOR(const std::string &v)
    {
        static const struct { const char *name; float value; } table[] = {
            {"vacuum", 1.0f}, {"helium", 1.000036f}, {"hydrogen", 1.000132f}, {"air", 1.000277f}, {"carbon dioxide", 1.00045f}, {"water", 1.3330f},
            {"acetone", 1.36f}, {"ethanol", 1.361f}, {"carbon tetrachloride", 1.461f}, {"glycerol", 1.4729f}, {"benzene", 1.501f},
            {"silicone oil", 1.52045f}, {"bromine", 1.661f}, {"water ice", 1.31f}, {"fused quartz", 1.458f}, {"pyrex", 1.470f},
            {"acrylic glass", 1.49f}, {"polypropylene", 1.49f}, {"bk7", 1.5046f}, {"sodium chloride", 1.544f}, {"amber", 1.55f}, {"pet", 1.5750f},
            {"diamond", 2.419f}};
        std::string lower = v;
        for (char &c : lower) c = (char)std::tolower((unsigned char)c);
        for (const auto &e : table) if (lower == e.name) return (double)e.value;
        logError(format("Unable to find an IOR value for \"%s\"!", v.c_str()));
    }

    int dielectric(const xml::Node &n, SceneData &sd)
    { // src/bsdfs/dielectric.cpp:141-160: intIOR (default bk7), extIOR (default air), specularReflectance, specularTransmittance
        gdpt_material m;
        std::memset(&m, 0, sizeof m);
        m.type = GDPT_MAT_DIELECTRIC;
        m.sampleVisible = 1;
        m.alphaU = m.alphaV = 0.1;
        double intIOR = lookupIOR("bk7"), extIOR = lookupIOR("air");
        for (int c = 0; c < 3; ++c) { m.reflectance[c] = 1.0; m.k[c] = 1.0; }
        for (auto &c : n.children) {
            const std::string nm = c->get("name", "");
            if ((c->tag == "float" || c->tag == "string") && (nm == "intIOR" || nm == "extIOR")) {
                const std::string v = subst(c->get("value"));
                const double ior = c->tag == "float" ? std::stod(v) : lookupIOR(v);
                (nm == "intIOR" ? intIOR : extIOR) = ior;
            } else if ((c->tag == "rgb" || c->tag == "spectrum") && nm == "specularReflectance") rgb3(*c, m.reflectance);
            else if ((c->tag == "rgb" || c->tag == "spectrum") && nm == "specularTransmittance") rgb3(*c, m.k);
            else logError(format("bsdf \"dielectric\": <%s name=\"%s\"> is not carried", c->tag.c_str(), nm.c_str()));
        }
        if (intIOR < 0 || extIOR < 0) logError("The interior and exterior indices of refraction must be positive!");     // dielectric.cpp:152-154
        m.eta[0] = m.eta[1] = m.eta[2] = intIOR / extIOR;
        sd.materials.push_back(m);
        return (int)sd.materials.size() - 1;
    }

    void addTri(SceneData &sd, const Mat4 &T, bool flip, const double *a, const double *b, const double *c, int mat)
    {
        double A[3], B[3], C[3];
        T.point(a, A);
        T.point(flip ? c : b, B);
        T.point(flip ? b : c, C);
        sd.verts.insert(sd.verts.end(), A, A + 3);
        sd.verts.insert(sd.verts.end(), B, B + 3);
        sd.verts.insert(sd.verts.end(), C, C + 3);
        sd.triMaterial.push_back(mat);
    }

    void shape(const xml::Node &n, SceneData &sd)
    {
        const std::string type = subst(n.get("type"));
        Mat4 T = Mat4::identity();
        int mat = -1;
        bool flipNormals = false, faceNormals = false, emits = false, flipTexCoords = true;
        int shapeIndex = 0;
        double radiance[3] = {1, 1, 1};
        std::string filename;
        for (auto &c : n.children) {
            if (c->tag == "transform" && c->get("name") == "toWorld") T = transform(*c);
            else if (c->tag == "ref") {
                auto it = m_bsdfIds.find(c->get("id"));
                if (it == m_bsdfIds.end()) logError(format("Referenced object \"%s\" not found (BSDFs must be declared before the shapes that use them)", c->get("id").c_str()));
                mat = it->second;
            } else if (c->tag == "bsdf") mat = bsdf(*c, sd);
            else if (c->tag == "emitter") {
                if (subst(c->get("type")) != "area") logError(format("emitter \"%s\" is not carried: `area` only", c->get("type").c_str()));
                emits = true;
                for (auto &ec : c->children)
                    if ((ec->tag == "rgb" || ec->tag == "spectrum") && ec->get("name") == "radiance") rgb3(*ec, radiance);
            } else if (c->tag == "string" && c->get("name") == "filename") filename = subst(c->get("value"));
            else if (c->tag == "boolean" && c->get("name") == "flipNormals") flipNormals = subst(c->get("value")) == "true";
            else if (c->tag == "boolean" && c->get("name") == "faceNormals") faceNormals = subst(c->get("value")) == "true";
            else if (c->tag == "boolean" && c->get("name") == "flipTexCoords") flipTexCoords = subst(c->get("value")) == "true";
            else if (c->tag == "integer" && c->get("name") == "shapeIndex") shapeIndex = std::atoi(subst(c->get("value")).c_str());
            else logError(format("shape \"%s\": <%s name=\"%s\"> is not carried", type.c_str(), c->tag.c_str(), c->get("name", "").c_str()));
        }
        if (mat < 0) { gdpt_material m; std::memset(&m, 0, sizeof m); m.type = GDPT_MAT_DIFFUSE; m.sampleVisible = 1; m.reflectance[0] = m.reflectance[1] = m.reflectance[2] = 0.5; m.alphaU = m.alphaV = 0.1; sd.materials.push_back(m); mat = (int)sd.materials.size() - 1; }   // shape.cpp: default diffuse
        const bool flip = flipNormals != (T.det3() < 0);        // rectangle / cube: the analytic shapes keep their outward normal under a mirroring transform
        const int first = sd.numTriangles();
        if (type == "rectangle") {                               // src/shapes/rectangle.cpp: [-1,1]^2 in z = 0, normal +z
            const double v[4][3] = {{-1, -1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 1, 0}};
            addTri(sd, T, flip, v[0], v[1], v[2], mat);
            addTri(sd, T, flip, v[2], v[3], v[0], mat);
            {   // Rectangle::fillIntersectionRecord: its.uv = (0.5 (x + 1), 0.5 (y + 1)) of the local hit point (rectangle.cpp) -- affine in the
                // position, so per-vertex coordinates (0,0) (1,0) (1,1) (0,1) interpolate to the same value
                const double q[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
                const int corners[2][3] = {{0, 1, 2}, {2, 3, 0}};
                sd.uvs.resize(6 * (size_t)sd.numTriangles(), 0.0);
                sd.triHasUV.resize((size_t)sd.numTriangles(), 0);
                for (int t = 0; t < 2; ++t) {
                    const size_t ti = (size_t)sd.numTriangles() - 2 + t;
                    int order[3] = {corners[t][0], corners[t][1], corners[t][2]};
                    if (flip) std::swap(order[1], order[2]);                      // addTri swaps the last two vertices of a flipped triangle
                    for (int j = 0; j < 3; ++j) { sd.uvs[6 * ti + 2 * j] = q[order[j]][0]; sd.uvs[6 * ti + 2 * j + 1] = q[order[j]][1]; }
                    sd.triHasUV[ti] = 1;
                }
            }
        } else if (type == "cube") {
            // src/shapes/cube.cpp:23-29,74-110: a TriMesh of 24 vertices (four per face, each face mapped onto [0,1]^2) and 12 triangles, in
            // the plugin's own vertex and triangle order -- the order fixes the barycentrics, the UV tangents (and with them the shading
            // frames, since every mesh with texture coordinates gets them) and the triangle cdf of a cube used as an area light.  Its
            // per-vertex normals are the face normals, so the interpolated normal is the face normal up to rounding: emitted as flat triangles.
            static const double P[24][3] = {{1, -1, -1}, {1, -1, 1}, {-1, -1, 1}, {-1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, 1, 1}, {1, 1, 1},
                                            {1, -1, -1}, {1, 1, -1}, {1, 1, 1}, {1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}, {-1, -1, 1},
                                            {-1, -1, 1}, {-1, 1, 1}, {-1, 1, -1}, {-1, -1, -1}, {1, 1, -1}, {1, -1, -1}, {-1, -1, -1}, {-1, 1, -1}};
            static const double Q[4][2] = {{0, 1}, {1, 1}, {1, 0}, {0, 0}};                 // texcoords of vertex 4 f + k
            for (int f = 0; f < 6; ++f) {
                const int tri[2][3] = {{4 * f, 4 * f + 1, 4 * f + 2}, {4 * f + 3, 4 * f, 4 * f + 2}};
                for (int t = 0; t < 2; ++t) {
                    addTri(sd, T, flip, P[tri[t][0]], P[tri[t][1]], P[tri[t][2]], mat);
                    int order[3] = {tri[t][0], tri[t][1], tri[t][2]};
                    if (flip) std::swap(order[1], order[2]);                              // addTri swaps the last two vertices of a flipped triangle
                    sd.uvs.resize(6 * (size_t)sd.numTriangles(), 0.0);
                    sd.triHasUV.resize((size_t)sd.numTriangles(), 0);
                    const size_t ti = (size_t)sd.numTriangles() - 1;
                    for (int j = 0; j < 3; ++j) { sd.uvs[6 * ti + 2 * j] = Q[order[j] % 4][0]; sd.uvs[6 * ti + 2 * j + 1] = Q[order[j] % 4][1]; }
                    sd.triHasUV[ti] = 1;
                }
            }
        } else if (type == "obj") {
            if (filename.empty()) logError("shape \"obj\": missing filename");
            loadObj(filename[0] == '/' ? filename : m_dir + "/" + filename, sd, T, flipNormals, faceNormals, mat, flipTexCoords);   // obj.cpp applies no handedness correction
        } else if (type == "serialized") {
            if (filename.empty()) logError("shape \"serialized\": missing filename");
            loadSerialized(filename[0] == '/' ? filename : m_dir + "/" + filename, shapeIndex, sd, T, flipNormals, faceNormals, mat);
        } else logError(format("shape \"%s\" is not carried: obj, serialized, rectangle, cube", type.c_str()));
        if (emits) {
            gdpt_emitter e;
            std::memset(&e, 0, sizeof e);
            e.firstTri = first; e.numTris = sd.numTriangles() - first;
            for (int k = 0; k < 3; ++k) e.radiance[k] = radiance[k];
            if (type == "rectangle") {                           // light samples as Rectangle::samplePosition draws them (rectangle.cpp:80-85,100-107,200-206)
                Mat4 M = T;
                if (flipNormals) for (int r = 0; r < 4; ++r) M.m[4 * r + 2] = -M.m[4 * r + 2];      // m_objectToWorld * Transform::scale(Vector(1, 1, -1))
                e.rectangle = 1;
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) e.rectToWorld[4 * r + c] = M.m[4 * r + c];
                const double up[3] = {0, 0, 1};
                double n[3];
                M.normal(up, n);
                const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                for (int k = 0; k < 3; ++k) e.rectNormal[k] = n[k] / len;
            }
            sd.emitters.push_back(e);
        }
    }

    // TriMesh::computeNormals (trimesh.cpp:608-681) on a mesh of world-space vertices (position 3 + normal 3) and index triples, then
    // the triangles go to the scene: faceNormals drops the normals (flipNormals swaps the first two vertices of every triangle); given
    // normals are kept (flipNormals negates them); a mesh without normals gets angle-weighted vertex normals (flipNormals negates
    // them too).  Vertex normals that equal the face normal of every triangle using them are dropped again: the flat code path gives
    // the same frame without the per-hit interpolation.
    void finishMesh(SceneData &sd, const std::vector<std::array<double, 6>> &vb, std::vector<std::array<int, 3>> idx, bool hasNormals, bool flipNormals, bool faceNormals, int mat,
                    const std::vector<std::array<double, 2>> *uv = nullptr)
    {
        std::vector<std::array<double, 3>> vn(vb.size(), std::array<double, 3>{{0.0, 0.0, 0.0}});
        bool useNormals = false;
        if (faceNormals) {
            if (flipNormals) for (auto &id : idx) std::swap(id[0], id[1]);
        } else if (hasNormals) {
            useNormals = true;
            for (size_t i = 0; i < vb.size(); ++i) for (int k = 0; k < 3; ++k) vn[i][k] = flipNormals ? -vb[i][3 + k] : vb[i][3 + k];
        } else {
            useNormals = true;                                                 // "Computing Vertex Normals from Polygonal Facets", trimesh.cpp:636-672
            for (auto &id : idx) {
                double n[3] = {0, 0, 0};
                for (int i = 0; i < 3; ++i) {
                    const double *v0 = vb[id[i]].data(), *v1 = vb[id[(i + 1) % 3]].data(), *v2 = vb[id[(i + 2) % 3]].data();
                    double a[3], b[3];
                    for (int k = 0; k < 3; ++k) { a[k] = v1[k] - v0[k]; b[k] = v2[k] - v0[k]; }
                    if (i == 0) {
                        n[0] = a[1] * b[2] - a[2] * b[1]; n[1] = a[2] * b[0] - a[0] * b[2]; n[2] = a[0] * b[1] - a[1] * b[0];
                        const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                        if (l == 0) break;
                        for (double &c : n) c /= l;
                    }
                    const double la = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), lb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
                    double u[3], w[3], dt = 0, sp = 0, sm = 0;
                    for (int k = 0; k < 3; ++k) { u[k] = a[k] / la; w[k] = b[k] / lb; dt += u[k] * w[k]; }
                    for (int k = 0; k < 3; ++k) { sp += (w[k] + u[k]) * (w[k] + u[k]); sm += (w[k] - u[k]) * (w[k] - u[k]); }
                    const double angle = dt < 0 ? M_PI - 2 * std::asin(0.5 * std::sqrt(sp)) : 2 * std::asin(0.5 * std::sqrt(sm));   // unitAngle, util.h:305-310
                    for (int k = 0; k < 3; ++k) vn[id[i]][k] += n[k] * angle;
                }
            }
            for (auto &n : vn) {
                double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                if (flipNormals) l *= -1;
                if (l != 0) for (double &c : n) c /= l;
                else { n[0] = 1; n[1] = 0; n[2] = 0; }
            }
        }
        // vertex normals equal to every adjacent face normal are the flat case
        bool allFlat = true;
        for (size_t t = 0; t < idx.size() && useNormals && allFlat; ++t) {
            const double *A = vb[idx[t][0]].data(), *B = vb[idx[t][1]].data(), *C = vb[idx[t][2]].data();
            double a[3], b[3];
            for (int k = 0; k < 3; ++k) { a[k] = B[k] - A[k]; b[k] = C[k] - A[k]; }
            const double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) if (std::abs(vn[idx[t][j]][k] - (l != 0 ? n[k] / l : 0.0)) > 1e-12) allFlat = false;
        }
        if (useNormals && allFlat) useNormals = false;
        for (size_t t = 0; t < idx.size(); ++t) {
            for (int j = 0; j < 3; ++j) sd.verts.insert(sd.verts.end(), vb[idx[t][j]].data(), vb[idx[t][j]].data() + 3);
            sd.triMaterial.push_back(mat);
            if (uv) {                                                             // TriMesh::getVertexTexcoords of this mesh (skdtree.h:398-402)
                sd.uvs.resize(6 * (size_t)sd.numTriangles(), 0.0);
                sd.triHasUV.resize((size_t)sd.numTriangles(), 0);
                for (int j = 0; j < 3; ++j) for (int k = 0; k < 2; ++k) sd.uvs[6 * (size_t)(sd.numTriangles() - 1) + 2 * j + k] = (*uv)[idx[t][j]][k];
                sd.triHasUV[(size_t)sd.numTriangles() - 1] = 1;
            }
            if (useNormals) {
                sd.normals.resize(9 * (size_t)sd.numTriangles(), 0.0);
                for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) sd.normals[9 * (size_t)(sd.numTriangles() - 1) + 3 * j + k] = vn[idx[t][j]][k];
            }
        }
    }

    // `<shape type="serialized">` (src/shapes/serialized.cpp:146-210) reading Mitsuba's compressed mesh format
    // (TriMesh::loadCompressed, trimesh.cpp:175-252: header 0x041C, version 3|4, then a zlib stream holding flags, [name,]
    // vertex and triangle counts, positions, [normals,] [texcoords,] [colors,] uint32 indices; multi-mesh files end with an
    // offset dictionary, trimesh.cpp:272-294).  Positions and normals go through toWorld; a mirroring transform swaps the first
    // two vertices of every triangle (serialized.cpp:197-202); the file's face-normal flag is overridden by the property.
    void loadSerialized(const std::string &path, int shapeIndex, SceneData &sd, const Mat4 &T, bool flipNormals, bool faceNormals, int mat)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) logError(format("Cannot open serialized mesh \"%s\"", path.c_str()));
        std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        auto u16 = [&](size_t o) { if (o + 2 > file.size()) logError(path + ": truncated"); return (unsigned)(file[o] | (file[o + 1] << 8)); };
        auto rd = [&](size_t o, size_t n) { unsigned long long v = 0; if (o + n > file.size()) logError(path + ": truncated"); for (size_t k = 0; k < n; ++k) v |= (unsigned long long)file[o + k] << (8 * k); return v; };
        if (u16(0) == 0x1C04) logError("Encountered a geometry file generated by an old version of Mitsuba. Please re-import the scene to update this file to the current format.");
        if (u16(0) != 0x041C) logError("Encountered an invalid file format!");
        const unsigned version = u16(2);
        if (version != 3 && version != 4) logError("Encountered an incompatible file version!");
        if (shapeIndex < 0) logError("Shape index must be nonnegative!");
        size_t offset = 0;
        if (shapeIndex != 0) {
            const size_t size = file.size();
            if (size < 8) logError(path + ": truncated");
            const unsigned count = (unsigned)rd(size - 4, 4);
            // (trimesh.cpp:279 rejects idx > count only; idx == count would read the count word as an offset, so it is rejected here too, and a
            // dictionary larger than the file is a corrupt file)
            if ((unsigned)shapeIndex >= count || (unsigned long long)count * (version == 4 ? 8 : 4) + 4 > size) logError(format("Unable to unserialize mesh, shape index is out of range! (requested %i out of 0..%i)", shapeIndex, (int)count - 1));
            offset = version == 4 ? (size_t)rd(size - 8 * (size_t)(count - shapeIndex) - 4, 8) : (size_t)rd(size - 4 * (size_t)(count - shapeIndex + 1), 4);
        }
        if (offset > file.size() || file.size() - offset < 4) logError(path + ": sub-mesh offset points outside the file");
        offset += 4;                                            // the (sub)stream's own header
        std::vector<unsigned char> data;
        {
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit(&zs) != Z_OK) logError("zlib: inflateInit failed");
            zs.next_in = file.data() + offset;
            zs.avail_in = (uInt)std::min<size_t>(file.size() - offset, 0xffffffffu);
            unsigned char buf[1 << 16];
            int rc;
            do {
                zs.next_out = buf; zs.avail_out = sizeof buf;
                rc = inflate(&zs, Z_NO_FLUSH);
                if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); logError(path + ": zlib stream is corrupt"); }
                data.insert(data.end(), buf, buf + (sizeof buf - zs.avail_out));
            } while (rc != Z_STREAM_END);
            inflateEnd(&zs);
        }
        size_t p = 0;
        auto need = [&](size_t n) { if (p + n > data.size()) logError(path + ": mesh data is truncated"); };
        auto r32 = [&]() { need(4); unsigned v; std::memcpy(&v, &data[p], 4); p += 4; return v; };
        auto r64 = [&]() { need(8); unsigned long long v; std::memcpy(&v, &data[p], 8); p += 8; return v; };
        const unsigned flags = r32();
        if (version == 4) { need(1); while (data[p] != 0) { ++p; need(1); } ++p; }        // name
        const size_t nv = (size_t)r64(), nt = (size_t)r64();
        const bool dbl = (flags & 0x2000) != 0;
        auto rfl = [&]() -> double { if (dbl) { need(8); double v; std::memcpy(&v, &data[p], 8); p += 8; return v; } need(4); float v; std::memcpy(&v, &data[p], 4); p += 4; return (double)v; };
        std::vector<std::array<double, 6>> vb(nv, std::array<double, 6>{{0, 0, 0, 0, 0, 0}});
        for (size_t i = 0; i < nv; ++i) { double q[3] = {rfl(), rfl(), rfl()}; T.point(q, vb[i].data()); }
        const bool hasNormals = (flags & 0x0001) != 0;
        if (hasNormals)
            for (size_t i = 0; i < nv; ++i) {
                double q[3] = {rfl(), rfl(), rfl()}, n[3];
                T.normal(q, n);
                const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                for (int k = 0; k < 3; ++k) vb[i][3 + k] = n[k] / l;                 // serialized.cpp:191-194
            }
        std::vector<std::array<double, 2>> uv2;
        if (flags & 0x0002) { uv2.resize(nv); for (size_t i = 0; i < nv; ++i) { uv2[i][0] = rfl(); uv2[i][1] = rfl(); } }   // texture coordinates (trimesh.cpp:222-226)
        if (flags & 0x0008) for (size_t i = 0; i < 3 * nv; ++i) rfl();                // vertex colours
        std::vector<std::array<int, 3>> idx(nt);
        const bool mirror = T.det3() < 0;
        for (size_t t = 0; t < nt; ++t) {
            for (int j = 0; j < 3; ++j) { const unsigned v = r32(); if (v >= nv) logError(path + ": triangle references a vertex out of range"); idx[t][j] = (int)v; }
            if (mirror) std::swap(idx[t][0], idx[t][1]);
        }
        finishMesh(sd, vb, idx, hasNormals, flipNormals, faceNormals, mat, uv2.empty() ? nullptr : &uv2);
    }

    // Wavefront OBJ subset (src/shapes/obj.cpp): v, vn, vt, f with v / v/vt / v//vn / v/vt/vn and negative indices, polygons fanned,
    // one mesh per o / g group.  As createMesh does (obj.cpp:608-704): positions and normals go through toWorld, vertices with equal
    // (position, normal, uv) VALUES are merged per mesh; then TriMesh::computeNormals (trimesh.cpp:608-681): faceNormals drops the
    // normals (flipNormals swaps the first two vertices of every triangle); given normals are kept (flipNormals negates them);
    // a mesh without normals gets angle-weighted vertex normals over the merged vertices (flipNormals negates them too).
    // Vertex normals that equal the face normal of every triangle using them are dropped again: the flat code path gives the
    // same frame without the per-hit interpolation.
    void loadObj(const std::string &path, SceneData &sd, const Mat4 &T, bool flipNormals, bool faceNormals, int mat, bool flipTexCoords = true)
    {
        std::ifstream f(path);
        if (!f) logError(format("Cannot open OBJ file \"%s\"", path.c_str()));
        struct V { double d[8]; bool operator<(const V &o) const { return std::lexicographical_compare(d, d + 8, o.d, o.d + 8); } };   // p(3) n(3) uv(2), obj.cpp:577-606
        std::vector<double> pos, nrm, tex;
        struct Corner { int v, t, n; };
        std::vector<std::array<Corner, 3>> tris;
        auto flush = [&]() {
            if (tris.empty()) return;
            std::map<V, int> vmap;
            std::vector<V> vb;
            std::vector<std::array<int, 3>> idx;
            bool hasNormals = false, hasTexcoords = false;
            for (auto &tr : tris) {
                std::array<int, 3> id;
                for (int j = 0; j < 3; ++j) {
                    V v;
                    for (double &c : v.d) c = 0.0;
                    T.point(&pos[3 * tr[j].v], v.d);
                    if (tr[j].n >= 0) {
                        T.normal(&nrm[3 * tr[j].n], v.d + 3);
                        const double l = std::sqrt(v.d[3] * v.d[3] + v.d[4] * v.d[4] + v.d[5] * v.d[5]);
                        if (l != 0) for (int k = 3; k < 6; ++k) v.d[k] /= l;
                        hasNormals = true;
                    }
                    if (tr[j].t >= 0) { v.d[6] = tex[2 * tr[j].t]; v.d[7] = tex[2 * tr[j].t + 1]; hasTexcoords = true; }    // obj.cpp:664-669
                    for (double &c : v.d) if (c == 0) c = 0.0;                      // -0.0 and 0.0 are one key
                    auto it = vmap.find(v);
                    if (it == vmap.end()) { it = vmap.emplace(v, (int)vb.size()).first; vb.push_back(v); }
                    id[j] = it->second;
                }
                idx.push_back(id);
            }
            std::vector<std::array<double, 6>> verts6(vb.size());
            std::vector<std::array<double, 2>> uv2(vb.size());
            for (size_t i = 0; i < vb.size(); ++i) { for (int k = 0; k < 6; ++k) verts6[i][k] = vb[i].d[k]; uv2[i] = {{vb[i].d[6], vb[i].d[7]}}; }
            finishMesh(sd, verts6, idx, hasNormals, flipNormals, faceNormals, mat, hasTexcoords ? &uv2 : nullptr);
            tris.clear();
        };
        std::string line;
        while (std::getline(f, line)) {
            std::stringstream ss(line);
            std::string tag;
            if (!(ss >> tag)) continue;
            if (tag == "v") { double x, y, z; ss >> x >> y >> z; pos.push_back(x); pos.push_back(y); pos.push_back(z); }
            else if (tag == "vn") { double x, y, z; ss >> x >> y >> z; nrm.push_back(x); nrm.push_back(y); nrm.push_back(z); }
            else if (tag == "vt") { double u = 0, v = 0; ss >> u >> v; if (flipTexCoords) v = 1 - v; tex.push_back(u); tex.push_back(v); }   // obj.cpp:304-308
            else if (tag == "o" || tag == "g") flush();
            else if (tag == "f") {
                std::vector<Corner> cs;
                std::string tok;
                while (ss >> tok) {
                    Corner c = {0, -1, -1};
                    const size_t s1 = tok.find('/'), s2 = s1 == std::string::npos ? std::string::npos : tok.find('/', s1 + 1);
                    auto resolve = [&](const std::string &str, size_t count, const char *what) {
                        int i = std::stoi(str);
                        i = i < 0 ? (int)count + i : i - 1;
                        if (i < 0 || i >= (int)count) logError(format("%s: face references %s %s out of range", path.c_str(), what, str.c_str()));
                        return i;
                    };
                    c.v = resolve(tok.substr(0, s1), pos.size() / 3, "vertex");
                    if (s1 != std::string::npos) {
                        const std::string ts = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                        if (!ts.empty()) c.t = resolve(ts, tex.size() / 2, "texture coordinate");
                        if (s2 != std::string::npos && s2 + 1 < tok.size()) c.n = resolve(tok.substr(s2 + 1), nrm.size() / 3, "normal");
                    }
                    cs.push_back(c);
                }
                for (size_t k = 1; k + 1 < cs.size(); ++k) tris.push_back({{cs[0], cs[k], cs[k + 1]}});
            }
        }
        flush();
    }
};

} // namespace gdpt
