// tests/mitsuba_mock/mock_mitsuba.h -- COMPILE-ONLY HARNESS, written for this repository (no Mitsuba source is copied): the minimal declarations of
// the Mitsuba 0.5 symbols that host/mitsuba_plugin/{gpt_hip.cpp,gbdpt_hip.cpp} use, with the signatures the reference's headers give them, so that
// the drop-in sources at least parse, type-check and link against lib/libgdpt_hip.so in CI (tests/test_plugin_sources.py).  Nothing here
// computes anything; it is not a Mitsuba build and pins no behaviour.  Signatures follow (paths relative to /root/reference/include/mitsuba):
//   core/object.h, core/cobject.h (ConfigurableObject, MTS_EXPORT_PLUGIN), core/properties.h, core/logger.h (Log / SLog), core/spectrum.h,
//   core/transform.h, core/bitmap.h, core/rfilter.h, core/sched.h, render/integrator.h (:61-118), render/scene.h, render/sensor.h,
//   render/film.h (:62-79 the Multi* virtuals), render/trimesh.h, render/emitter.h, render/bsdf.h, render/texture.h, render/sampler.h;
//   for the block-process shape of gpt_hip.cpp: core/sched.h (:43-177 WorkUnit / WorkResult / WorkProcessor, :216-330 ParallelProcess, :362-463),
//   core/lock.h, core/statistics.h (:293-299), render/rectwu.h, render/imageblock.h (:59-106), render/imageproc.h, render/renderproc.h (:38-93),
//   render/renderqueue.h (:102).
#pragma once
#include <algorithm>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#define MTS_NAMESPACE_BEGIN namespace mitsuba {
#define MTS_NAMESPACE_END }
#define MTS_DECLARE_CLASS() virtual const Class *getClass() const;
#define MTS_IMPLEMENT_CLASS_S(name, abstract, super) const Class *name::getClass() const { static Class c(#name); return &c; }
#define MTS_IMPLEMENT_CLASS(name, abstract, super) MTS_IMPLEMENT_CLASS_S(name, abstract, super)
#define MTS_EXPORT_PLUGIN(name, descr) extern "C" { void *CreateInstance(const Properties &props) { return new name(props); } const char *GetDescription() { return descr; } }

MTS_NAMESPACE_BEGIN
typedef double Float;                                   // the reference's DOUBLE_PRECISION build
#define Epsilon 1e-7                                   // core/constants.h:25
enum ELogLevel { EDebug, EInfo, EWarn, EError };
inline void mockLog(ELogLevel lvl, const char *fmt, ...)
{
    char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (lvl == EError) throw std::runtime_error(buf);     // logger.cpp:147
    fprintf(stderr, "%s\n", buf);
}
#define Log(level, ...) ::mitsuba::mockLog(level, __VA_ARGS__)
#define SLog(level, ...) ::mitsuba::mockLog(level, __VA_ARGS__)

class Class { public: explicit Class(const char *n) : m_name(n) {} const std::string &getName() const { return m_name; } private: std::string m_name; };
class Object { public: virtual ~Object() {} virtual const Class *getClass() const = 0; void incRef() const {} void decRef() const {} };
template <class T> class ref {
public:
    ref(T *p = nullptr) : m_p(p) {}
    T *operator->() const { return m_p; }
    T *get() const { return m_p; }
    operator T *() const { return m_p; }
private:
    T *m_p;
};
template <class T> class ref_vector : public std::vector<ref<T> > {};

struct Vector { Float x, y, z; Vector(Float a = 0, Float b = 0, Float c = 0) : x(a), y(b), z(c) {} };
struct Point { Float x, y, z; Point(Float a = 0, Float b = 0, Float c = 0) : x(a), y(b), z(c) {} };
inline Point operator+(Point p, Vector v) { return Point(p.x + v.x, p.y + v.y, p.z + v.z); }
struct Normal { Float x, y, z; Normal(Float a = 0, Float b = 0, Float c = 0) : x(a), y(b), z(c) {} };
inline Normal normalize(const Normal &n) { return n; }
struct Point2 { Float x, y; Point2(Float a = 0, Float b = 0) : x(a), y(b) {} };
struct Vector2i { int x, y; Vector2i(int a = 0, int b = 0) : x(a), y(b) {} };
struct Point2i { int x, y; Point2i(int a = 0, int b = 0) : x(a), y(b) {} };
struct Frame { Vector s, t; Normal n; Frame() {} explicit Frame(const Normal &nn) : n(nn) {} };
struct Matrix4x4 { Float m[4][4]; Float operator()(int r, int c) const { return m[r][c]; } };
class Transform {
public:
    const Matrix4x4 &getMatrix() const { return m_m; }
    Point operator()(const Point &p) const { return p; }
    Vector operator()(const Vector &v) const { return v; }
    Normal operator()(const Normal &n) const { return n; }
    Transform operator*(const Transform &) const { return *this; }
    static Transform scale(const Vector &) { return Transform(); }
private:
    Matrix4x4 m_m;
};
class AnimatedTransform { public: const Transform &eval(Float) const { return m_t; } private: Transform m_t; };
class InterpolatedSpectrum { public: explicit InterpolatedSpectrum(const std::string &) {} };          // core/spectrum.h
class Spectrum {
public:
    Spectrum(Float v = 0) { c[0] = c[1] = c[2] = v; }
    void toLinearRGB(Float &r, Float &g, Float &b) const { r = c[0]; g = c[1]; b = c[2]; }
    void fromContinuousSpectrum(const InterpolatedSpectrum &) {}
    Spectrum operator/(Float f) const { Spectrum s; for (int i = 0; i < 3; ++i) s.c[i] = c[i] / f; return s; }
    static Spectrum getD65() { return Spectrum(1); }
    Float c[3];
};
class FileResolver { public: std::string resolve(const std::string &p) const { return p; } void incRef() const {} };   // core/fresolver.h (returns fs::path there)
class Thread { public: static Thread *getThread() { static Thread t; return &t; } FileResolver *getFileResolver() { return &m_fr; } private: FileResolver m_fr; };
struct Triangle { uint32_t idx[3]; };
struct RayDifferential { RayDifferential(const Point &, const Vector &, Float) {} };

class Properties {
public:
    enum EPropertyType { EBoolean, EInteger, EFloat, EPoint, ETransform, EAnimatedTransform, ESpectrum, EString, EData };   // properties.h:52-62
    explicit Properties(const std::string &plugin = "") : m_plugin(plugin) {}
    EPropertyType getType(const std::string &) const { return EFloat; }
    Transform getTransform(const std::string &, const Transform &def) const { return def; }
    const std::string &getPluginName() const { return m_plugin; }
    bool hasProperty(const std::string &) const { return false; }
    int getInteger(const std::string &, int def) const { return def; }
    Float getFloat(const std::string &, Float def) const { return def; }
    Float getFloat(const std::string &) const { return 0; }
    bool getBoolean(const std::string &, bool def) const { return def; }
    std::string getString(const std::string &, const std::string &def) const { return def; }
    Spectrum getSpectrum(const std::string &, const Spectrum &def) const { return def; }
    Spectrum getSpectrum(const std::string &) const { return Spectrum(); }
private:
    std::string m_plugin;
};
class InstanceManager; class RenderJob; class ParallelProcess;
class Stream { public: int readInt() { return 0; } void writeInt(int) {} bool readBool() { return false; } void writeBool(bool) {} };   // core/stream.h
class SerializableObject : public Object {                                             // core/serialization.h
public:
    SerializableObject() {}
    SerializableObject(Stream *, InstanceManager *) {}
    virtual void serialize(Stream *, InstanceManager *) const {}
};
class ConfigurableObject : public SerializableObject {
public:
    ConfigurableObject() {}
    explicit ConfigurableObject(const Properties &p) : m_properties(p) {}
    const Properties &getProperties() const { return m_properties; }                    // cobject.h
protected:
    Properties m_properties;
};

class Bitmap : public Object {
public:
    enum EPixelFormat { ELuminance, ERGB, ESpectrum, ESpectrumAlphaWeight };
    enum EComponentFormat { EUInt8, EFloat16, EFloat32, EFloat64, EFloat = EFloat64 };
    Bitmap(EPixelFormat, EComponentFormat, const Vector2i &size) : m_size(size), m_data((size_t)size.x * size.y * 5 * 8) {}
    int getChannelCount() const { return 5; }
    void clear() { std::fill(m_data.begin(), m_data.end(), 0); }
    const Vector2i &getSize() const { return m_size; }
    int getWidth() const { return m_size.x; }
    int getHeight() const { return m_size.y; }
    float *getFloat32Data() { return reinterpret_cast<float *>(m_data.data()); }
    double *getFloat64Data() { return reinterpret_cast<double *>(m_data.data()); }
    Float *getFloatData() { return reinterpret_cast<Float *>(m_data.data()); }
    ref<Bitmap> convert(EPixelFormat pf, EComponentFormat cf, Float gamma = 1.0) const { (void)gamma; return new Bitmap(pf, cf, m_size); }   // bitmap.h convert(pixelFormat, componentFormat, gamma, ...)
    const Class *getClass() const { static Class c("Bitmap"); return &c; }
private:
    Vector2i m_size; std::vector<char> m_data;
};

class ReconstructionFilter : public ConfigurableObject { public: Float getRadius() const { return 0.5; } const Class *getClass() const { static Class c("BoxFilter"); return &c; } };
class ImageBlock : public Object {                                                       // render/imageblock.h:59-106 (this fork's extraBorder argument)
public:
    ImageBlock(Bitmap::EPixelFormat fmt, const Vector2i &size, const ReconstructionFilter *filter = nullptr, int channels = -1, bool warn = true, int extraBorder = 0)
        : m_size(size), m_border(extraBorder), m_bitmap(new Bitmap(fmt, Bitmap::EFloat, Vector2i(size.x + 2 * extraBorder, size.y + 2 * extraBorder))) { (void)filter; (void)channels; (void)warn; }
    void setAllowNegativeValues(bool) {}
    void setOffset(const Point2i &o) { m_offset = o; }
    const Point2i &getOffset() const { return m_offset; }
    void setSize(const Vector2i &s) { m_size = s; }
    const Vector2i &getSize() const { return m_size; }
    int getBorderSize() const { return m_border; }
    Bitmap *getBitmap() { return m_bitmap; }
    const Bitmap *getBitmap() const { return m_bitmap.get(); }
    void clear() { m_bitmap->clear(); }
    void load(Stream *) {}
    void save(Stream *) const {}
    std::string toString() const { return "ImageBlock[]"; }
    const Class *getClass() const { static Class c("ImageBlock"); return &c; }
private:
    Point2i m_offset; Vector2i m_size; int m_border; ref<Bitmap> m_bitmap;
};
class Film : public ConfigurableObject {
public:
    const Vector2i &getSize() const { return m_size; }                                                           // film.h:35-42
    const Vector2i &getCropSize() const { return m_size; }
    const Point2i &getCropOffset() const { return m_cropOffset; }
    const ReconstructionFilter *getReconstructionFilter() const { return &m_rf; }
    virtual void clear() {}
    virtual bool setBuffers(std::vector<std::string> &) { return true; }                                          // film.h:62-79: the five multi-buffer virtuals
    virtual void setBitmapMulti(const Bitmap *, Float, int) {}
    virtual void addBitmapMulti(const Bitmap *, Float, int) {}
    virtual bool developMulti(const Point2i &, const Vector2i &, const Point2i &, Bitmap *, int) const { return true; }
    virtual void putMulti(const ImageBlock *, int) {}
    const Class *getClass() const { static Class c("MultiFilm"); return &c; }
private:
    Vector2i m_size; Point2i m_cropOffset; ReconstructionFilter m_rf;
};
class Sampler : public ConfigurableObject { public: size_t getSampleCount() const { return 4; } const Class *getClass() const { static Class c("IndependentSampler"); return &c; } };
class Sensor : public ConfigurableObject {
public:
    Film *getFilm() { return &m_film; }                     // sensor.h:255-258: both overloads
    const Film *getFilm() const { return &m_film; }
    const AnimatedTransform *getWorldTransform() const { return &m_t; }
    Float getShutterOpen() const { return 0; }              // sensor.h:275
    Float getShutterOpenTime() const { return 0; }          // sensor.h:281
    const Class *getClass() const { static Class c("PerspectiveCamera"); return &c; }
private:
    Film m_film; AnimatedTransform m_t;
};
class PerspectiveCamera : public Sensor { public: Float getXFov() const { return 40; } Float getNearClip() const { return 0.01; } Float getFarClip() const { return 1e4; } Float getFocusDistance() const { return 1e4; } };

struct Intersection { Frame shFrame, geoFrame; Point2 uv; };
struct DirectSamplingRecord { DirectSamplingRecord(const Point &, Float) {} };
class Texture : public ConfigurableObject {
public:
    virtual ref<Bitmap> getBitmap(const Vector2i &resolutionHint = Vector2i(-1, -1)) const { (void)resolutionHint; return new Bitmap(Bitmap::ERGB, Bitmap::EFloat64, Vector2i(1, 1)); }   // texture.h
    virtual bool isConstant() const { return true; }
    virtual const Texture *getNestedTexture() const { return nullptr; }                // ADDED accessor, not in Mitsuba 0.5
    const Class *getClass() const { static Class c("ConstantSpectrumTexture"); return &c; }
};
class BSDF : public ConfigurableObject {
public:
    virtual Spectrum getDiffuseReflectance(const Intersection &) const { return Spectrum(0.5); }       // bsdf.h
    virtual Spectrum getSpecularReflectance(const Intersection &) const { return Spectrum(1.0); }
    virtual Float getEta() const { return 1.0; }                                         // bsdf.h:451
    virtual const BSDF *getNestedBRDF() const { return nullptr; }                       // ADDED accessor, not in Mitsuba 0.5
    virtual const Texture *getReflectanceTexture() const { return nullptr; }            // ADDED accessor, not in Mitsuba 0.5
    const Class *getClass() const { static Class c("SmoothDiffuse"); return &c; }
};
class Emitter : public ConfigurableObject {
public:
    virtual Spectrum eval(const Intersection &, const Vector &) const { return Spectrum(1); }
    virtual Spectrum evalEnvironment(const RayDifferential &) const { return Spectrum(1); }
    virtual Spectrum sampleDirect(DirectSamplingRecord &, const Point2 &) const { return Spectrum(1); }
    virtual ref<Bitmap> getBitmap(const Vector2i &sizeHint = Vector2i(-1, -1)) const { (void)sizeHint; return new Bitmap(Bitmap::ERGB, Bitmap::EFloat64, Vector2i(2, 1)); }   // emitter.h: the environment map's bitmap
    const AnimatedTransform *getWorldTransform() const { return &m_t; }
    const Class *getClass() const { static Class c("AreaLight"); return &c; }
private:
    AnimatedTransform m_t;
};
class TriMesh;
class Shape : public ConfigurableObject {
public:
    const BSDF *getBSDF() const { return &m_bsdf; }
    bool isEmitter() const { return false; }
    const Emitter *getEmitter() const { return &m_em; }
    virtual ref<TriMesh> createTriMesh();                                               // shape.h:230
    const Class *getClass() const { static Class c("Shape"); return &c; }
protected:
    BSDF m_bsdf; Emitter m_em;
};
class TriMesh : public Shape {
public:
    const Point *getVertexPositions() const { return nullptr; }
    const Normal *getVertexNormals() const { return nullptr; }
    const Point2 *getVertexTexcoords() const { return nullptr; }
    const Triangle *getTriangles() const { return nullptr; }
    size_t getTriangleCount() const { return 0; }
};
inline ref<TriMesh> Shape::createTriMesh() { return nullptr; }
class Scene : public ConfigurableObject {
public:
    Sensor *getSensor() { return &m_sensor; }
    uint32_t getBlockSize() const { return 32; }                                         // scene.h:1128
    void bindUsedResources(ParallelProcess *) const {}                                  // scene.h:1134
    const std::vector<TriMesh *> &getMeshes() const { return m_meshes; }
    const ref_vector<Shape> &getShapes() const { return m_shapes; }
    const ref_vector<Emitter> &getEmitters() const { return m_emitters; }
    const Class *getClass() const { static Class c("Scene"); return &c; }
private:
    PerspectiveCamera m_sensor; std::vector<TriMesh *> m_meshes; ref_vector<Shape> m_shapes; ref_vector<Emitter> m_emitters;
};
// ---- core/sched.h, core/lock.h, core/statistics.h, render/{rectwu,imageproc,renderproc,renderqueue}.h: the block scheduler's surface ----
class Mutex : public Object { public: const Class *getClass() const { static Class c("Mutex"); return &c; } };
class UniqueLock { public: explicit UniqueLock(Mutex *) {} void unlock() {} };
class ProgressReporter { public: ProgressReporter(const std::string &, long long, const void *) {} void update(long long) {} };
class RenderQueue : public Object { public: void signalWorkEnd(const RenderJob *, const ImageBlock *, bool) {} const Class *getClass() const { static Class c("RenderQueue"); return &c; } };
class WorkUnit : public Object { public: virtual void set(const WorkUnit *) = 0; virtual void load(Stream *) = 0; virtual void save(Stream *) const = 0; virtual std::string toString() const = 0; };
class WorkResult : public Object { public: virtual void load(Stream *) = 0; virtual void save(Stream *) const = 0; virtual std::string toString() const = 0; };
class RectangularWorkUnit : public WorkUnit {
public:
    void set(const WorkUnit *) {} void load(Stream *) {} void save(Stream *) const {} std::string toString() const { return "RectangularWorkUnit[]"; }
    const Point2i &getOffset() const { return m_offset; }
    const Vector2i &getSize() const { return m_size; }
    const Class *getClass() const { static Class c("RectangularWorkUnit"); return &c; }
private:
    Point2i m_offset; Vector2i m_size;
};
class WorkProcessor : public SerializableObject {
public:
    virtual ref<WorkUnit> createWorkUnit() const = 0;
    virtual ref<WorkResult> createWorkResult() const = 0;
    virtual ref<WorkProcessor> clone() const = 0;
    virtual void prepare() = 0;
    virtual void process(const WorkUnit *workUnit, WorkResult *workResult, const bool &stop) = 0;
protected:
    WorkProcessor() {}
    WorkProcessor(Stream *s, InstanceManager *m) : SerializableObject(s, m) {}
    SerializableObject *getResource(const std::string &) { return nullptr; }
};
class ParallelProcess : public Object {
public:
    enum EStatus { EUnknown, EPause, ESuccess, EFailure };
    EStatus getReturnStatus() const { return ESuccess; }
    virtual ref<WorkProcessor> createWorkProcessor() const = 0;
    virtual void processResult(const WorkResult *result, bool cancelled) = 0;
    virtual void bindResource(const std::string &, int) {}
};
class BlockedImageProcess : public ParallelProcess { protected: int m_blockSize; };
class BlockedRenderProcess : public BlockedImageProcess {
public:
    BlockedRenderProcess(const RenderJob *parent, RenderQueue *queue, int blockSize)
        : m_queue(queue), m_film(nullptr), m_parent(parent), m_resultCount(0), m_resultMutex(new Mutex()), m_progress(new ProgressReporter("Rendering", 1, parent)),
          m_borderSize(0), m_pixelFormat(Bitmap::ESpectrumAlphaWeight), m_channelCount(-1), m_warnInvalid(false) { m_blockSize = blockSize; }
    void bindResource(const std::string &, int) {}
protected:
    ref<RenderQueue> m_queue; ref<Film> m_film; const RenderJob *m_parent; int m_resultCount; ref<Mutex> m_resultMutex; ProgressReporter *m_progress;
    int m_borderSize; Bitmap::EPixelFormat m_pixelFormat; int m_channelCount; bool m_warnInvalid;
};
class Scheduler : public Object {
public:
    static Scheduler *getInstance() { static Scheduler s; return &s; }
    ConfigurableObject *getResource(int, int = -1) { static Sampler smp; return &smp; }
    int registerResource(SerializableObject *) { return 0; }
    bool unregisterResource(int) { return true; }
    bool schedule(ParallelProcess *) { return true; }
    bool wait(const ParallelProcess *) { return true; }
    bool cancel(ParallelProcess *) { return true; }
    size_t getCoreCount() const { return 1; }
    const Class *getClass() const { static Class c("Scheduler"); return &c; }
};
class Integrator : public ConfigurableObject {                                            // render/integrator.h:61-118
public:
    explicit Integrator(const Properties &p) : ConfigurableObject(p) {}
    Integrator(Stream *, InstanceManager *) {}
    virtual bool preprocess(const Scene *, RenderQueue *, const RenderJob *, int, int, int) { return true; }
    virtual bool render(Scene *, RenderQueue *, const RenderJob *, int, int, int) = 0;
    virtual void cancel() = 0;
    virtual void postprocess(const Scene *, RenderQueue *, const RenderJob *, int, int, int) {}
    virtual void serialize(Stream *, InstanceManager *) const {}
    virtual std::string toString() const { return "Integrator[]"; }
    void bindUsedResources(ParallelProcess *) const {}                                 // cobject.h
};
MTS_NAMESPACE_END
