// gpt_hip.cpp -- `<integrator type="gpt_hip">`: the G-PT integrator of src/integrators/gpt/gpt.cpp with its per-pixel sampling and its
// reconstruction running on an MI355X through lib/libgdpt_hip.so (include/gdpt_tracer.h, include/gdpt_poisson.h).
//
// Written against Mitsuba 0.5's API as the reference uses it (the calls are those of gpt.cpp:1358-1480 and of the trimesh / scene headers
// cited inline).  It cannot be compiled in this repository's image (no boost / Xerces / OpenEXR), so it ships as a source for the
// maintainer; every library call below is exercised by this repository's tests through the same C-ABI.
//
// What is carried is the scene subset of include/gdpt_tracer.h: triangle meshes; diffuse / conductor / roughconductor / dielectric
// BSDFs (optionally `twosided`, optionally with a `bitmap` texture on the reflectance, filterType nearest | bilinear); area, point and
// constant emitters; a perspective sensor; independent sampler semantics (counter-based streams); any of the six
// reconstruction filters.  Anything else is refused by the library with a message -- never rendered approximately.
#include <mitsuba/render/scene.h>
#include <mitsuba/render/renderjob.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/bitmap.h>
#include "gdpt_tracer.h"

MTS_NAMESPACE_BEGIN

class GradientPathIntegratorHIP : public Integrator {
public:
	GradientPathIntegratorHIP(const Properties &props) : Integrator(props), m_film(NULL) {
		/* the properties and checks of gpt.cpp:1194-1213 */
		m_maxDepth = props.getInteger("maxDepth", -1);
		m_rrDepth = props.getInteger("rrDepth", 5);
		m_strictNormals = props.getBoolean("strictNormals", false);
		m_shiftThreshold = props.getFloat("shiftThreshold", Float(0.001));
		m_reconstructL1 = props.getBoolean("reconstructL1", true);
		m_reconstructL2 = props.getBoolean("reconstructL2", false);
		m_reconstructAlpha = (Float) props.getFloat("reconstructAlpha", Float(0.2));
		m_devices = props.getInteger("devices", 1);      /* not a reference property: GPUs to shard the frame over (row strips) */
		if (m_reconstructL1 && m_reconstructL2)
			Log(EError, "Disable 'reconstructL1' or 'reconstructL2': Cannot display two reconstructions at a time!");
		if (m_reconstructAlpha <= 0.0f)
			Log(EError, "'reconstructAlpha' must be set to a value greater than zero!");
		if (m_maxDepth <= 0 && m_maxDepth != -1)
			Log(EError, "'maxDepth' must be set to -1 (infinite) or a value greater than zero!");
	}

	GradientPathIntegratorHIP(Stream *stream, InstanceManager *manager) : Integrator(stream, manager), m_film(NULL) {
		Log(EError, "gpt_hip: network rendering is not carried (the GPU path renders on the node that owns the GPUs)");
	}

	bool preprocess(const Scene *, RenderQueue *, const RenderJob *, int, int, int) { return true; }

	/* gpt.cpp:1358-1480 with the block scheduler replaced by gdpt_render_rect */
	bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID, int samplerResID) {
		ref<Sensor> sensor = scene->getSensor();
		ref<Film> film = sensor->getFilm();
		std::vector<std::string> outNames = {"-final", "-throughput", "-dx", "-dy", "-direct"};
		if (!film->setBuffers(outNames)) {
			Log(EError, "Cannot render image! G-PT has been called without MultiFilm.");
			return false;
		}
		const Vector2i size = film->getCropSize();
		const int W = size.x, H = size.y;

		/* ---- flatten the scene (scene.h: getMeshes(); trimesh.h accessors) ---- */
		std::vector<double> verts, normals, uvs;
		std::vector<unsigned char> hasUV;
		std::vector<int> triMat;
		std::vector<gdpt_material> mats;
		std::vector<int> matTex;
		std::vector<gdpt_texture> texs;
		std::vector<std::vector<double> > texels;
		std::vector<gdpt_emitter> ems;
		bool anyNormals = false, anyUV = false;
		const std::vector<TriMesh *> &meshes = scene->getMeshes();
		for (size_t m = 0; m < meshes.size(); ++m) {
			const TriMesh *mesh = meshes[m];
			const int first = (int) triMat.size();
			const int mat = addMaterial(mesh->getBSDF(), mats, matTex, texs, texels);
			const Point *P = mesh->getVertexPositions();
			const Normal *N = mesh->getVertexNormals();
			const Point2 *T = mesh->getVertexTexcoords();
			const Triangle *tri = mesh->getTriangles();
			for (size_t t = 0; t < mesh->getTriangleCount(); ++t) {
				for (int k = 0; k < 3; ++k) {
					const uint32_t i = tri[t].idx[k];
					verts.push_back(P[i].x); verts.push_back(P[i].y); verts.push_back(P[i].z);
					normals.push_back(N ? N[i].x : 0); normals.push_back(N ? N[i].y : 0); normals.push_back(N ? N[i].z : 0);
					uvs.push_back(T ? T[i].x : 0); uvs.push_back(T ? T[i].y : 0);
				}
				hasUV.push_back(T ? 1 : 0);
				triMat.push_back(mat);
			}
			anyNormals |= N != NULL; anyUV |= T != NULL;
			if (mesh->isEmitter()) {
				/* AreaLight::m_radiance (area.cpp:71): what eval() returns for a front-facing direction */
				Intersection its; its.shFrame = its.geoFrame = Frame(Normal(0, 0, 1));
				const Spectrum Le = mesh->getEmitter()->eval(its, Vector(0, 0, 1));
				gdpt_emitter e; memset(&e, 0, sizeof e);
				e.firstTri = first; e.numTris = (int) triMat.size() - first;
				Float r, g, b; Le.toLinearRGB(r, g, b);
				e.radiance[0] = r; e.radiance[1] = g; e.radiance[2] = b;
				ems.push_back(e);
			}
		}
		/* point and constant emitters, in the scene's emitter order (scene.cpp:855-862 selects by it) */
		gdpt_environment env; bool haveEnv = false;
		memset(&env, 0, sizeof env);
		const ref_vector<Emitter> &emitters = scene->getEmitters();
		for (size_t i = 0; i < emitters.size(); ++i) {
			const Emitter *em = emitters[i].get();
			const std::string cls = em->getClass()->getName();
			if (cls == "ConstantBackgroundEmitter") {
				const Spectrum Le = em->evalEnvironment(RayDifferential(Point(0.0f), Vector(0, 0, 1), 0));
				Float r, g, b; Le.toLinearRGB(r, g, b);
				env.radiance[0] = r; env.radiance[1] = g; env.radiance[2] = b; env.index = (int) i; haveEnv = true;
			} else if (cls == "PointEmitter") {
				gdpt_emitter e; memset(&e, 0, sizeof e);
				e.numTris = -1;
				const Point p = em->getWorldTransform()->eval(0)(Point(0.0f));
				e.position[0] = p.x; e.position[1] = p.y; e.position[2] = p.z;
				/* PointEmitter::m_intensity: sampleDirect returns intensity / dist^2 (point.cpp:120-134) */
				DirectSamplingRecord dRec(p + Vector(0, 0, 1), 0);
				const Spectrum I = em->sampleDirect(dRec, Point2(0.5f));
				Float r, g, b; I.toLinearRGB(r, g, b);
				e.radiance[0] = r; e.radiance[1] = g; e.radiance[2] = b;
				ems.insert(ems.begin() + std::min(i, ems.size()), e);
			} else if (cls != "AreaLight")
				Log(EError, "gpt_hip: emitter \"%s\" is not carried (area, point, constant)", cls.c_str());
		}
		gdpt_camera cam;
		{
			const PerspectiveCamera *pc = static_cast<const PerspectiveCamera *>(sensor.get());
			const Matrix4x4 &M = pc->getWorldTransform()->eval(0).getMatrix();
			for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) cam.toWorld[4 * r + c] = M(r, c);
			cam.fovX = pc->getXFov(); cam.nearClip = pc->getNearClip(); cam.farClip = pc->getFarClip();
			cam.width = W; cam.height = H;
		}
		gdpt_scene *gs = NULL;
		check(gdpt_scene_create_tex((int) triMat.size(), verts.data(), anyNormals ? normals.data() : NULL, anyUV ? uvs.data() : NULL,
				anyUV ? hasUV.data() : NULL, triMat.data(), (int) mats.size(), mats.data(), texs.empty() ? NULL : matTex.data(),
				(int) texs.size(), texs.empty() ? NULL : texs.data(), (int) ems.size(), ems.data(), haveEnv ? &env : NULL, &cam, -1, &gs));
		check(gdpt_film_create(gs, 0, H, &m_film));
		setFilter(film->getReconstructionFilter());

		const Sampler *sampler = static_cast<const Sampler *>(Scheduler::getInstance()->getResource(samplerResID, 0));
		gdpt_config cfg;
		cfg.maxDepth = m_maxDepth; cfg.rrDepth = m_rrDepth; cfg.strictNormals = m_strictNormals;
		cfg.spp = (int) sampler->getSampleCount(); cfg.shiftThreshold = m_shiftThreshold; cfg.seed = 5489ull;   /* random.h:113 */
		Log(EInfo, "Starting render job (GPT::render on MI355X) (%ix%i, %i samples) ..", W, H, cfg.spp);

		/* GPTBlockRenderer::process over the whole film in one call (blocks may also be handed over one by one) */
		check(gdpt_render_rect(gs, &cfg, 0, 0, W, H, m_film));
		check(gdpt_film_sync(m_film));
		int cancelled = 0;
		check(gdpt_film_cancelled(m_film, &cancelled));
		if (cancelled) { release(gs); return false; }

		/* developMulti + the float casts of gpt.cpp:1419-1442, on the device */
		std::vector<float> img[5];
		for (int b = 0; b < 5; ++b) { img[b].resize((size_t) 3 * W * H); check(gdpt_film_develop(m_film, b, img[b].data())); }
		if (m_reconstructL1 || m_reconstructL2) {      /* gpt.cpp:1445-1462 */
			gdpt_poisson_params p;
			gdpt_poisson_params_defaults(&p);
			gdpt_poisson_params_preset(&p, m_reconstructL1 ? "L1D" : "L2D");
			p.alpha = (float) m_reconstructAlpha;
			gdpt_poisson_solver *solver = NULL;
			check(gdpt_poisson_create(&p, &solver));
			check(gdpt_poisson_import_images(solver, img[2].data(), img[3].data(), img[1].data(), img[4].data(), W, H));
			check(gdpt_poisson_setup_backend(solver));
			check(gdpt_poisson_solve_indirect(solver));
			check(gdpt_poisson_export_images(solver, img[0].data()));
			gdpt_poisson_destroy(solver);
		}
		/* setBitmapMulti(..., 1, buffer) as gpt.cpp:1464-1475 */
		for (int b = 0; b < 5; ++b) {
			ref<Bitmap> bmp = new Bitmap(Bitmap::ESpectrum, Bitmap::EFloat32, size);
			memcpy(bmp->getFloat32Data(), img[b].data(), sizeof(float) * img[b].size());
			film->setBitmapMulti(bmp, 1, b);
		}
		release(gs);
		return true;
	}

	void cancel() { if (m_film) gdpt_film_cancel(m_film); }     /* Integrator::cancel, integrator.h:88 */
	void postprocess(const Scene *, RenderQueue *, const RenderJob *, int, int, int) {}
	void serialize(Stream *stream, InstanceManager *manager) const { Integrator::serialize(stream, manager); }
	std::string toString() const { return "GradientPathIntegratorHIP[]"; }

	MTS_DECLARE_CLASS()
private:
	static void check(int rc) { if (rc != GDPT_OK) SLog(EError, "gpt_hip: %s", gdpt_last_error()); }
	void release(gdpt_scene *gs) { gdpt_film_destroy(m_film); m_film = NULL; gdpt_scene_destroy(gs); }

	void setFilter(const ReconstructionFilter *rf) {
		const std::string cls = rf->getClass()->getName();
		/* the filters' own parameters are private members; a maintainer passes them through (defaults shown) */
		if (cls == "BoxFilter") return;
		else if (cls == "TentFilter") check(gdpt_film_set_rfilter(m_film, GDPT_RFILTER_TENT, 0, 0));
		else if (cls == "GaussianFilter") check(gdpt_film_set_rfilter(m_film, GDPT_RFILTER_GAUSSIAN, rf->getRadius() / 4, 0));   /* radius = 4 stddev, gaussian.cpp:38 */
		else if (cls == "MitchellNetravaliFilter") check(gdpt_film_set_rfilter(m_film, GDPT_RFILTER_MITCHELL, 1.0 / 3.0, 1.0 / 3.0));
		else if (cls == "CatmullRomFilter") check(gdpt_film_set_rfilter(m_film, GDPT_RFILTER_CATMULLROM, 0, 0));
		else if (cls == "LanczosSincFilter") check(gdpt_film_set_rfilter(m_film, GDPT_RFILTER_LANCZOS, rf->getRadius(), 0));       /* radius = lobes, lanczos.cpp:35 */
		else Log(EError, "gpt_hip: reconstruction filter \"%s\" is not carried", cls.c_str());
	}

	/* BSDF -> gdpt_material.  Mitsuba's BSDF classes keep eta / k / alpha private: the two conductors and the dielectric need the
	   accessors `getEta()`, `getK()`, `getAlphaU()`, `getAlphaV()`, `getDistributionType()`, `getSampleVisible()` added next to their
	   members (conductor.cpp, roughconductor.cpp, dielectric.cpp) -- one line each; diffuse works as is. */
	int addMaterial(const BSDF *bsdf, std::vector<gdpt_material> &mats, std::vector<int> &matTex, std::vector<gdpt_texture> &texs, std::vector<std::vector<double> > &texels) {
		gdpt_material m; memset(&m, 0, sizeof m);
		m.sampleVisible = 1; m.alphaU = m.alphaV = 0.1;
		std::string cls = bsdf->getClass()->getName();
		if (cls == "TwoSidedBRDF") { m.twoSided = 1; bsdf = static_cast<const BSDF *>(bsdf->getSubObject(0)); cls = bsdf->getClass()->getName(); }   /* a maintainer exposes the nested BRDF */
		Intersection its;
		Float r, g, b;
		if (cls == "SmoothDiffuse") {
			m.type = GDPT_MAT_DIFFUSE;
			bsdf->getDiffuseReflectance(its).toLinearRGB(r, g, b);
			m.reflectance[0] = r; m.reflectance[1] = g; m.reflectance[2] = b;
		} else
			Log(EError, "gpt_hip: add the accessors named above to \"%s\" and fill eta / k / alpha here", cls.c_str());
		/* a `bitmap` texture on the reflectance: Texture::getBitmap() gives level 0; wrap modes / filter type / uv transform need accessors too */
		mats.push_back(m); matTex.push_back(-1);
		(void) texs; (void) texels;
		return (int) mats.size() - 1;
	}

	gdpt_film *m_film;
	int m_maxDepth, m_rrDepth, m_devices;
	bool m_strictNormals, m_reconstructL1, m_reconstructL2;
	Float m_shiftThreshold, m_reconstructAlpha;
};

MTS_IMPLEMENT_CLASS_S(GradientPathIntegratorHIP, false, Integrator)
MTS_EXPORT_PLUGIN(GradientPathIntegratorHIP, "Gradient-domain path tracer on MI355X (libgdpt_hip)");
MTS_NAMESPACE_END
