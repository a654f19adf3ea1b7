// gdpt_mitsuba_scene.h -- a Mitsuba 0.5 `Scene` flattened into the scene subset of include/gdpt_tracer.h; shared by gpt_hip.cpp and gbdpt_hip.cpp.
//
// Everything is read through Mitsuba's PUBLIC interface as the reference's headers give it:
//   ConfigurableObject::getProperties()  (core/cobject.h:77)  -- every plain parameter of a BSDF / texture / emitter / shape;
//   Scene::getShapes / getMeshes / getEmitters / getSensor (render/scene.h:1086-1093), Shape::createTriMesh / getBSDF / getEmitter
//   (render/shape.h:230,445-465), TriMesh accessors (render/trimesh.h), BSDF::getEta / getDiffuseReflectance / getSpecularReflectance
//   (render/bsdf.h:337-340,451), Texture::getBitmap / getMaximum (render/texture.h:62,105), Emitter::getBitmap (render/emitter.h:553).
// Nested objects are children, not properties, and Mitsuba gives no way to ask a BSDF or a texture for them.  THREE one-line
// virtuals therefore have to be added (INTEGRATION.md section 3c lists them with the lines to paste); the code below calls them
// and nothing else that Mitsuba does not already have:
//   BSDF::getNestedBRDF()            (twosided.cpp: return m_nestedBRDF[0])
//   BSDF::getReflectanceTexture()    (diffuse.cpp: m_reflectance; conductor / roughconductor / dielectric: m_specularReflectance)
//   Texture::getNestedTexture()      (scale.cpp: m_nested -- the wrapper BSDF::ensureEnergyConservation puts around a texture above 1)
// Anything outside the subset is refused with Log(EError) -- never rendered approximately.
#pragma once
#include <mitsuba/render/scene.h>
#include <mitsuba/core/bitmap.h>
#include <algorithm>
#include <cctype>
#include <list>
#include <string>
#include <vector>
#include "gdpt_tracer.h"

MTS_NAMESPACE_BEGIN
namespace gdpt_plugin {

struct FlatScene {
	std::vector<double> verts, normals, uvs;
	std::vector<unsigned char> hasUV;
	std::vector<int> triMat, matTex;
	std::vector<gdpt_material> mats;
	std::vector<gdpt_texture> texs;
	std::list<std::vector<double> > pixels;         /* level-0 texels / the environment map; gdpt_texture::rgb and gdpt_environment::rgb point in here */
	std::vector<gdpt_emitter> ems;
	gdpt_environment env;
	gdpt_camera cam;
	bool anyNormals, anyUV, haveEnv;
	FlatScene() : anyNormals(false), anyUV(false), haveEnv(false) { memset(&env, 0, sizeof env); memset(&cam, 0, sizeof cam); }
};

inline std::string lower(std::string s) { std::transform(s.begin(), s.end(), s.begin(), ::tolower); return s; }
inline void rgb3(const Spectrum &s, double out[3]) { Float r, g, b; s.toLinearRGB(r, g, b); out[0] = r; out[1] = g; out[2] = b; }

inline int wrapMode(const std::string &w) {        /* bitmap.cpp:324-338 */
	if (w == "repeat") return GDPT_TEXWRAP_REPEAT;
	if (w == "clamp") return GDPT_TEXWRAP_CLAMP;
	if (w == "mirror") return GDPT_TEXWRAP_MIRROR;
	if (w == "zero" || w == "black") return GDPT_TEXWRAP_ZERO;
	if (w == "one" || w == "white") return GDPT_TEXWRAP_ONE;
	SLog(EError, "Invalid wrap mode \"%s\"", w.c_str());
	return 0;
}

/* level 0 of a bitmap as height x width x 3 linear doubles, top row first */
inline const double *keepPixels(FlatScene &fs, const Bitmap *src) {
	ref<Bitmap> rgb = src->convert(Bitmap::ERGB, Bitmap::EFloat64);
	const size_t n = (size_t) 3 * rgb->getWidth() * rgb->getHeight();
	fs.pixels.push_back(std::vector<double>(rgb->getFloat64Data(), rgb->getFloat64Data() + n));
	return fs.pixels.back().data();
}

/* `<texture type="bitmap">` (optionally inside the `scale` wrapper of BSDF::ensureEnergyConservation, bsdf.cpp:88-112) -> gdpt_texture; -1: constant */
inline int addTexture(FlatScene &fs, const Texture *tex) {
	if (!tex || tex->isConstant()) return -1;
	double scale = 1.0;
	if (tex->getClass()->getName() == "ScalingTexture") {
		scale = tex->getProperties().getFloat("scale");
		tex = tex->getNestedTexture();                                                    /* ADDED accessor */
	}
	if (!tex || tex->getClass()->getName() != "BitmapTexture")
		SLog(EError, "gdpt: texture \"%s\" is not carried (bitmap)", tex ? tex->getClass()->getName().c_str() : "?");
	const Properties &p = tex->getProperties();
	if (p.getString("channel", "") != "") SLog(EError, "gdpt: bitmap textures with a `channel` are not carried");
	gdpt_texture t; memset(&t, 0, sizeof t);
	ref<Bitmap> bmp = tex->getBitmap();                                                   /* bitmap.cpp:482: level 0 of the MIP map */
	t.width = bmp->getWidth(); t.height = bmp->getHeight();
	t.rgb = keepPixels(fs, bmp);
	const std::string wrap = p.getString("wrapMode", "repeat"), filter = lower(p.getString("filterType", "ewa"));   /* bitmap.cpp:213-216 */
	t.wrapU = wrapMode(p.getString("wrapModeU", wrap));
	t.wrapV = wrapMode(p.getString("wrapModeV", wrap));
	t.filter = filter == "nearest" ? GDPT_TEXFILTER_NEAREST : filter == "bilinear" ? GDPT_TEXFILTER_BILINEAR
	         : filter == "trilinear" ? GDPT_TEXFILTER_TRILINEAR : GDPT_TEXFILTER_EWA;
	const Float uvscale = p.getFloat("uvscale", 1.0f);                                     /* texture.cpp:82-91 */
	t.uscale = p.getFloat("uscale", uvscale); t.vscale = p.getFloat("vscale", uvscale);
	t.uoffset = p.getFloat("uoffset", 0.0f); t.voffset = p.getFloat("voffset", 0.0f);
	t.scale = scale;
	t.maxAnisotropy = p.getFloat("maxAnisotropy", 20);                                    /* bitmap.cpp:232 */
	fs.texs.push_back(t);
	return (int) fs.texs.size() - 1;
}

/* conductor.cpp:157-176 / roughconductor.cpp:172-191: eta and k from the properties (or the named material's measured data), over extEta */
inline void conductorIOR(const Properties &p, gdpt_material &m) {
	const std::string material = p.getString("material", "Cu");
	Spectrum intEta, intK;
	if (lower(material) == "none") { intEta = Spectrum(0.0f); intK = Spectrum(1.0f); }
	else {
		ref<FileResolver> fResolver = Thread::getThread()->getFileResolver();
		intEta.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + material + ".eta.spd")));
		intK.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + material + ".k.spd")));
	}
	Float extEta = 1.000277f;                                                              /* "air", src/bsdfs/ior.h:43 */
	if (p.hasProperty("extEta")) {
		if (p.getType("extEta") != Properties::EFloat)
			SLog(EError, "gdpt: give `extEta` as a number (the named-medium table lives in src/bsdfs/ior.h, private to that directory)");
		extEta = p.getFloat("extEta");
	}
	rgb3(p.getSpectrum("eta", intEta) / extEta, m.eta);
	rgb3(p.getSpectrum("k", intK) / extEta, m.k);
}

inline void microfacet(const Properties &p, gdpt_material &m) {      /* microfacet.h:100-139 */
	const std::string d = lower(p.getString("distribution", "beckmann"));
	if (d == "beckmann") m.distribution = GDPT_DISTR_BECKMANN;
	else if (d == "ggx") m.distribution = GDPT_DISTR_GGX;
	else if (d == "phong" || d == "as") m.distribution = GDPT_DISTR_PHONG;
	else SLog(EError, "Specified an invalid distribution \"%s\"", d.c_str());
	m.alphaU = m.alphaV = p.hasProperty("alpha") ? p.getFloat("alpha") : Float(0.1);
	if (p.hasProperty("alphaU")) m.alphaU = p.getFloat("alphaU");
	if (p.hasProperty("alphaV")) m.alphaV = p.getFloat("alphaV");
	m.sampleVisible = p.getBoolean("sampleVisible", true) && m.distribution != GDPT_DISTR_PHONG ? 1 : 0;
}

inline int addMaterial(FlatScene &fs, const BSDF *bsdf) {
	if (!bsdf) SLog(EError, "gdpt: a shape without a BSDF");
	gdpt_material m; memset(&m, 0, sizeof m);
	m.sampleVisible = 1; m.alphaU = m.alphaV = 0.1;
	std::string cls = bsdf->getClass()->getName();
	if (cls == "TwoSidedBRDF") {
		m.twoSided = 1;
		bsdf = bsdf->getNestedBRDF();                                                      /* ADDED accessor */
		if (!bsdf) SLog(EError, "gdpt: `twosided` does not expose its nested BRDF (INTEGRATION.md 3c)");
		cls = bsdf->getClass()->getName();
	}
	const Properties &p = bsdf->getProperties();
	Intersection its;
	if (cls == "SmoothDiffuse") {
		m.type = GDPT_MAT_DIFFUSE;
		rgb3(bsdf->getDiffuseReflectance(its), m.reflectance);                             /* the constant's value; a texture overrides it below */
	} else if (cls == "SmoothConductor" || cls == "RoughConductor") {
		m.type = cls == "SmoothConductor" ? GDPT_MAT_CONDUCTOR : GDPT_MAT_ROUGHCONDUCTOR;
		rgb3(bsdf->getSpecularReflectance(its), m.reflectance);
		conductorIOR(p, m);
		if (m.type == GDPT_MAT_ROUGHCONDUCTOR) microfacet(p, m);
	} else if (cls == "SmoothDielectric") {
		m.type = GDPT_MAT_DIELECTRIC;
		m.eta[0] = m.eta[1] = m.eta[2] = bsdf->getEta();                                   /* intIOR / extIOR, dielectric.cpp:149-156 */
		rgb3(bsdf->getSpecularReflectance(its), m.reflectance);
		rgb3(p.getSpectrum("specularTransmittance", Spectrum(1.0f)), m.k);
	} else
		SLog(EError, "gdpt: BSDF \"%s\" is not carried (diffuse, conductor, roughconductor, dielectric, twosided)", cls.c_str());
	fs.mats.push_back(m);
	fs.matTex.push_back(addTexture(fs, bsdf->getReflectanceTexture()));                    /* ADDED accessor; NULL for a constant */
	return (int) fs.mats.size() - 1;
}

inline void addMesh(FlatScene &fs, const TriMesh *mesh, const Shape *owner) {
	const int first = (int) fs.triMat.size();
	const int mat = addMaterial(fs, owner->getBSDF());
	const Point *P = mesh->getVertexPositions();
	const Normal *N = mesh->getVertexNormals();
	const Point2 *T = mesh->getVertexTexcoords();
	const Triangle *tri = mesh->getTriangles();
	for (size_t t = 0; t < mesh->getTriangleCount(); ++t) {
		for (int k = 0; k < 3; ++k) {
			const uint32_t i = tri[t].idx[k];
			fs.verts.push_back(P[i].x); fs.verts.push_back(P[i].y); fs.verts.push_back(P[i].z);
			fs.normals.push_back(N ? N[i].x : 0); fs.normals.push_back(N ? N[i].y : 0); fs.normals.push_back(N ? N[i].z : 0);
			fs.uvs.push_back(T ? T[i].x : 0); fs.uvs.push_back(T ? T[i].y : 0);
		}
		fs.hasUV.push_back(T ? 1 : 0);
		fs.triMat.push_back(mat);
	}
	fs.anyNormals |= N != NULL; fs.anyUV |= T != NULL;
	if (!owner->isEmitter()) return;
	const Emitter *em = owner->getEmitter();
	if (em->getClass()->getName() != "AreaLight") SLog(EError, "gdpt: shape emitter \"%s\" is not carried (area)", em->getClass()->getName().c_str());
	gdpt_emitter e; memset(&e, 0, sizeof e);
	e.firstTri = first; e.numTris = (int) fs.triMat.size() - first;
	rgb3(em->getProperties().getSpectrum("radiance", Spectrum::getD65()), e.radiance);   /* area.cpp:77 */
	if (owner->getClass()->getName() == "Rectangle") {
		/* light samples are drawn as the shape draws them (rectangle.cpp:81-84,200-206), not per triangle */
		Transform o2w = owner->getProperties().getTransform("toWorld", Transform());
		if (owner->getProperties().getBoolean("flipNormals", false)) o2w = o2w * Transform::scale(Vector(1, 1, -1));
		const Matrix4x4 &M = o2w.getMatrix();
		for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) e.rectToWorld[4 * r + c] = M(r, c);
		const Normal n = normalize(o2w(Normal(0, 0, 1)));
		e.rectNormal[0] = n.x; e.rectNormal[1] = n.y; e.rectNormal[2] = n.z;
		e.rectangle = 1;
	}
	fs.ems.push_back(e);
}

/* Scene -> FlatScene.  Emitters end up in the scene's emitter order, which Scene::sampleEmitterDirect's discrete pick follows (scene.cpp:855-862):
   area lights are appended as their shapes are met (the order in which Scene::addChild registered them), point emitters are inserted at their index. */
inline void flatten(const Scene *scene, const Sensor *sensor, const Vector2i &size, FlatScene &fs) {
	const ref_vector<Shape> &shapes = scene->getShapes();
	for (size_t s = 0; s < shapes.size(); ++s) {
		Shape *shape = const_cast<Shape *>(shapes[s].get());
		const std::string cls = shape->getClass()->getName();
		if (cls == "TriMesh" || cls == "PLYLoader" || cls == "SerializedMesh")
			addMesh(fs, static_cast<const TriMesh *>(shape), shape);          /* an `obj` arrives as its elements: Scene::addChild expands compound shapes */
		else if (cls == "Rectangle") {
			ref<TriMesh> tm = shape->createTriMesh();                                      /* rectangle.cpp:232-260: two triangles, no vertex normals on an emitter */
			addMesh(fs, tm.get(), shape);
		} else
			SLog(EError, "gdpt: shape \"%s\" is not carried (triangle meshes, rectangle)", cls.c_str());
	}
	const ref_vector<Emitter> &emitters = scene->getEmitters();
	for (size_t i = 0; i < emitters.size(); ++i) {
		const Emitter *em = emitters[i].get();
		const std::string cls = em->getClass()->getName();
		const Properties &p = em->getProperties();
		if (cls == "ConstantBackgroundEmitter") {
			rgb3(p.getSpectrum("radiance", Spectrum::getD65()), fs.env.radiance);        /* constant.cpp:49 */
			fs.env.index = (int) i; fs.haveEnv = true;
		} else if (cls == "EnvironmentMap") {
			ref<Bitmap> bmp = em->getBitmap();                                             /* envmap.cpp:635: level 0 of its MIP map */
			fs.env.width = bmp->getWidth(); fs.env.height = bmp->getHeight();
			fs.env.rgb = keepPixels(fs, bmp);
			fs.env.scale = p.getFloat("scale", 1.0f);                                      /* envmap.cpp:188 */
			const Matrix4x4 &M = em->getWorldTransform()->eval(0).getMatrix();
			for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) fs.env.toWorld[3 * r + c] = M(r, c);
			fs.env.index = (int) i; fs.haveEnv = true;
		} else if (cls == "PointEmitter") {
			gdpt_emitter e; memset(&e, 0, sizeof e);
			e.numTris = -1;
			const Point pos = em->getWorldTransform()->eval(0)(Point(0.0f));                /* point.cpp:60-66: `position` becomes the translation of toWorld */
			e.position[0] = pos.x; e.position[1] = pos.y; e.position[2] = pos.z;
			rgb3(p.getSpectrum("intensity", Spectrum::getD65()), e.radiance);            /* point.cpp:68 */
			fs.ems.insert(fs.ems.begin() + std::min(i, fs.ems.size()), e);
		} else if (cls != "AreaLight")
			SLog(EError, "gdpt: emitter \"%s\" is not carried (area, point, constant, envmap)", cls.c_str());
	}
	const std::string scls = sensor->getClass()->getName();
	if (scls != "PerspectiveCameraImpl" && scls != "PerspectiveCamera" && scls != "ThinLens")
		SLog(EError, "gdpt: sensor \"%s\" is not carried (perspective, thinlens)", scls.c_str());
	const PerspectiveCamera *pc = static_cast<const PerspectiveCamera *>(sensor);
	const Matrix4x4 &M = pc->getWorldTransform()->eval(0).getMatrix();
	for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) fs.cam.toWorld[4 * r + c] = M(r, c);
	fs.cam.fovX = pc->getXFov(); fs.cam.nearClip = pc->getNearClip(); fs.cam.farClip = pc->getFarClip();
	fs.cam.width = size.x; fs.cam.height = size.y;                                        /* `size` = the film's crop size: what is rendered */
	{	/* the crop window (perspective.cpp:126-163): the C-ABI takes the rays from the full film's raster */
		const Film *film = sensor->getFilm();
		const Vector2i full = film->getSize();
		const Point2i off = film->getCropOffset();
		if (full.x != size.x || full.y != size.y || off.x != 0 || off.y != 0) {
			fs.cam.cropOffsetX = off.x; fs.cam.cropOffsetY = off.y; fs.cam.fullWidth = full.x; fs.cam.fullHeight = full.y;
		}
	}
	fs.cam.shutterOpen = sensor->getShutterOpen();                                         /* sensor.h:275-281; an interval of positive length <=> needsTimeSample() */
	fs.cam.shutterClose = sensor->getShutterOpen() + sensor->getShutterOpenTime();
	if (scls == "ThinLens") {                                                              /* thinlens.cpp:236-244: both are plain properties */
		fs.cam.type = GDPT_SENSOR_THINLENS;
		fs.cam.apertureRadius = sensor->getProperties().getFloat("apertureRadius");
		if (fs.cam.apertureRadius == 0) fs.cam.apertureRadius = Epsilon;                    /* :134-138 */
		fs.cam.focusDistance = pc->getFocusDistance();                                      /* ProjectiveCamera::getFocusDistance, sensor.h */
	}
}

inline void check(int rc) { if (rc != GDPT_OK) SLog(EError, "gdpt: %s", gdpt_last_error()); }

/* upload to `device` (-1: the current one): BVH build on the host, triangles in leaf order in HBM */
inline gdpt_scene *upload(const FlatScene &fs, int device) {
	gdpt_scene *gs = NULL;
	check(gdpt_scene_create_tex((int) fs.triMat.size(), fs.verts.data(), fs.anyNormals ? fs.normals.data() : NULL,
		fs.anyUV ? fs.uvs.data() : NULL, fs.anyUV ? fs.hasUV.data() : NULL, fs.triMat.data(), (int) fs.mats.size(), fs.mats.data(),
		fs.texs.empty() ? NULL : fs.matTex.data(), (int) fs.texs.size(), fs.texs.empty() ? NULL : fs.texs.data(),
		(int) fs.ems.size(), fs.ems.empty() ? NULL : fs.ems.data(), fs.haveEnv ? &fs.env : NULL, &fs.cam, device, &gs));
	return gs;
}

/* the film's `<rfilter>` -> gdpt_film_set_rfilter arguments; the filters' parameters are plain properties (src/rfilters/<type>.cpp) */
inline void rfilterOf(const ReconstructionFilter *rf, int &kind, double &p0, double &p1) {
	const std::string cls = rf->getClass()->getName();
	const Properties &p = rf->getProperties();
	kind = GDPT_RFILTER_BOX; p0 = p1 = 0;
	if (cls == "BoxFilter") return;
	else if (cls == "TentFilter") kind = GDPT_RFILTER_TENT;
	else if (cls == "GaussianFilter") { kind = GDPT_RFILTER_GAUSSIAN; p0 = p.getFloat("stddev", 0.5f); }                 /* gaussian.cpp:35 */
	else if (cls == "MitchellNetravaliFilter") { kind = GDPT_RFILTER_MITCHELL; p0 = p.getFloat("B", 1.0f / 3.0f); p1 = p.getFloat("C", 1.0f / 3.0f); }   /* mitchell.cpp:37-39 */
	else if (cls == "CatmullRomFilter") kind = GDPT_RFILTER_CATMULLROM;
	else if (cls == "LanczosSincFilter") { kind = GDPT_RFILTER_LANCZOS; p0 = p.getInteger("lobes", 3); }                  /* lanczos.cpp:35 */
	else SLog(EError, "gdpt: reconstruction filter \"%s\" is not carried", cls.c_str());
}

} // namespace gdpt_plugin
MTS_NAMESPACE_END
