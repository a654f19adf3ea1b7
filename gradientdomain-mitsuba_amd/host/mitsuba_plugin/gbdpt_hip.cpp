// gbdpt_hip.cpp -- `<integrator type="gbdpt_hip">`: the G-BDPT integrator of src/integrators/gbdpt/gbdpt.cpp with GBDPTRenderer::process
// (gbdpt_proc.cpp:86-256) and both reconstructions running on an MI355X through lib/libgdpt_hip.so (the G-BDPT section of include/gdpt_tracer.h,
// gdpt_gbdpt_reconstruct_device of include/gdpt_poisson.h).
//
// Scope as the library states it: scenes whose BSDFs are all connectable in the sense of Path::isConnectable_GBDPT (diffuse and rough
// conductors with roughness >= shiftThreshold), area emitters, perspective sensor, box filter; anything else comes back from
// gdpt_gbdpt_render_rect as GDPT_ERR_UNSUPPORTED with a message and is logged as an error -- never rendered approximately.
// tests/test_plugin_sources.py compiles this file against compile-only mock headers (tests/mitsuba_mock) and links it with the library.
#include <mitsuba/render/scene.h>
#include <mitsuba/render/renderjob.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/bitmap.h>
#include "gdpt_tracer.h"
#include "gdpt_poisson.h"
#include "gdpt_mitsuba_scene.h"

MTS_NAMESPACE_BEGIN

class GBDPTIntegratorHIP : public Integrator {
public:
	GBDPTIntegratorHIP(const Properties &props) : Integrator(props) {
		/* the properties and checks of gbdpt.cpp:79-104 */
		m_maxDepth = props.getInteger("maxDepth", -1);
		m_rrDepth = props.getInteger("rrDepth", 5);
		m_lightImage = props.getBoolean("lightImage", true);
		m_shiftThreshold = props.getFloat("shiftThreshold", Float(0.001));
		m_reconstructL1 = props.getBoolean("reconstructL1", true);
		m_reconstructL2 = props.getBoolean("reconstructL2", false);
		m_reconstructAlpha = (Float) props.getFloat("reconstructAlpha", Float(0.2));
		if (m_reconstructL1 && m_reconstructL2)
			Log(EError, "Disable 'reconstructL1' or 'reconstructL2': Cannot display two reconstructions at a time!");   /* gbdpt.cpp:91-92 */
		if (m_reconstructAlpha <= 0.0f)
			Log(EError, "'reconstructAlpha' must be set to a value greater than zero!");
		if (m_rrDepth <= 0)
			Log(EError, "'rrDepth' must be set to a value greater than zero!");
		if (m_maxDepth <= 0 && m_maxDepth != -1)
			Log(EError, "'maxDepth' must be set to -1 (infinite) or a value greater than zero!");
	}

	GBDPTIntegratorHIP(Stream *stream, InstanceManager *manager) : Integrator(stream, manager) {
		Log(EError, "gbdpt_hip: network rendering is not carried (the GPU path renders on the node that owns the GPU)");
	}

	bool preprocess(const Scene *, RenderQueue *, const RenderJob *, int, int, int) { return true; }

	/* gbdpt.cpp:128-254 with the block scheduler replaced by gdpt_gbdpt_render_rect */
	bool render(Scene *scene, RenderQueue *, const RenderJob *, int, int, int samplerResID) {
		ref<Sensor> sensor = scene->getSensor();
		ref<Film> film = sensor->getFilm();
		/* gbdpt.cpp:163: the displayed reconstruction first */
		const char *names[7] = { m_reconstructL1 ? "-L1" : "-L2", "-gradientNegY", "-gradientNegX", "-gradientPosX", "-gradientPosY", m_reconstructL1 ? "-L2" : "-L1", "-primal" };
		std::vector<std::string> outNames(names, names + 7);
		if (!film->setBuffers(outNames)) {
			Log(EError, "Cannot render image! G-BDPT has been called without MultiFilm.");
			return false;
		}
		const Vector2i size = film->getCropSize();
		const int W = size.x, H = size.y;
		int kind; double p0, p1;
		gdpt_plugin::rfilterOf(film->getReconstructionFilter(), kind, p0, p1);
		if (kind != GDPT_RFILTER_BOX) Log(EError, "gbdpt_hip: the G-BDPT path carries the box filter only");

		gdpt_plugin::FlatScene fs;
		gdpt_plugin::flatten(scene, sensor.get(), size, fs);
		gdpt_scene *gs = gdpt_plugin::upload(fs, -1);
		gdpt_gbdpt_film *gf = NULL;
		check(gdpt_gbdpt_film_create(gs, &gf));

		const Sampler *sampler = static_cast<const Sampler *>(Scheduler::getInstance()->getResource(samplerResID, 0));
		gdpt_gbdpt_config cfg;
		cfg.maxDepth = m_maxDepth; cfg.rrDepth = m_rrDepth; cfg.lightImage = m_lightImage;
		cfg.spp = (int) sampler->getSampleCount(); cfg.shiftThreshold = m_shiftThreshold; cfg.seed = 5489ull;
		Log(EInfo, "Starting render job (G-BDPT on MI355X) (%ix%i, %i samples) ..", W, H, cfg.spp);
		check(gdpt_gbdpt_render_rect(gs, &cfg, 0, 0, W, H, gf));
		check(gdpt_gbdpt_film_sync(gf));

		/* GBDPTProcess::develop + developMulti (gbdpt_proc.cpp:694-706, gbdpt.cpp:199-207), then gbdpt.cpp:178-247 */
		const size_t n = (size_t) 3 * W * H;
		std::vector<double> dev[5];
		for (int b = 0; b < 5; ++b) { dev[b].resize(n); check(gdpt_gbdpt_film_develop(gf, b, cfg.spp, dev[b].data())); }
		std::vector<float> recL2(n), recL1(n);
		check(gdpt_gbdpt_reconstruct(dev[0].data(), dev[1].data(), dev[2].data(), dev[3].data(), dev[4].data(), W, H, (float) m_reconstructAlpha, -1, recL2.data(), recL1.data()));
		gdpt_gbdpt_film_destroy(gf);
		gdpt_scene_destroy(gs);
		gdpt_gbdpt_reconstruct_release_size(-1, W, H);       /* the library keeps the solvers of a frame size between calls: a render job is one frame, hand THIS size back */

		/* setBitmapMulti as gbdpt.cpp:222-247: slots 0 and 5 the reconstructions, 1..4 the gradients, 6 the primal */
		for (int b = 0; b < 7; ++b) {
			ref<Bitmap> bmp = new Bitmap(Bitmap::ESpectrum, Bitmap::EFloat32, size);
			float *dst = bmp->getFloat32Data();
			if (b == 0 || b == 5) {
				const std::vector<float> &src = ((b == 0) == m_reconstructL1) ? recL1 : recL2;
				memcpy(dst, src.data(), sizeof(float) * n);
			} else {
				const std::vector<double> &src = dev[b == 6 ? 0 : b];
				for (size_t i = 0; i < n; ++i) dst[i] = (float) src[i];
			}
			film->setBitmapMulti(bmp, 1, b);
		}
		return true;
	}

	void cancel() {}      /* a frame is one asynchronous enqueue; the G-BDPT film has no stop flag (the G-PT one has: gdpt_film_cancel) */
	void postprocess(const Scene *, RenderQueue *, const RenderJob *, int, int, int) {}
	void serialize(Stream *stream, InstanceManager *manager) const { Integrator::serialize(stream, manager); }
	std::string toString() const { return "GBDPTIntegratorHIP[]"; }

	MTS_DECLARE_CLASS()
private:
	static void check(int rc) { if (rc != GDPT_OK) SLog(EError, "gbdpt_hip: %s", gdpt_last_error()); }

	int m_maxDepth, m_rrDepth;
	bool m_lightImage, m_reconstructL1, m_reconstructL2;
	Float m_shiftThreshold, m_reconstructAlpha;
};

MTS_IMPLEMENT_CLASS_S(GBDPTIntegratorHIP, false, Integrator)
MTS_EXPORT_PLUGIN(GBDPTIntegratorHIP, "Gradient-domain bidirectional path tracer on MI355X (libgdpt_hip)");
MTS_NAMESPACE_END
