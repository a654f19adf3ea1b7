// gpt_hip.cpp -- `<integrator type="gpt_hip">`: the G-PT integrator of src/integrators/gpt/gpt.cpp with its per-pixel sampling and its
// reconstruction running on an MI355X through lib/libgdpt_hip.so (include/gdpt_tracer.h, include/gdpt_poisson.h).
//
// Written against Mitsuba 0.5's API as the reference uses it (the calls are those of gpt.cpp:1358-1480 and of the trimesh / scene headers
// cited inline).  It cannot be compiled in this repository's image (no boost / Xerces / OpenEXR), so it ships as a source for the
// maintainer; every library call below is exercised by this repository's tests through the same C-ABI.
//
// What is carried is the scene subset of include/gdpt_tracer.h, flattened by gdpt_mitsuba_scene.h: triangle meshes and `rectangle` shapes; diffuse /
// conductor / roughconductor / dielectric BSDFs (optionally `twosided`, optionally with a `bitmap` texture on the reflectance); area, point,
// constant and envmap emitters; a perspective sensor; independent sampler semantics (counter-based streams); any of the six
// reconstruction filters.  Anything else is refused with a message -- never rendered approximately.  `devices` > 1 shards the frame over
// the GPUs of the node in row strips (one thread per GPU, borders exchanged over xGMI), as host/gdpt_host.hpp does.
//
// Two shapes of render() (property `blocked`, default true on one device):
//  * the reference's own: a BlockedRenderProcess (GPTRenderProcessHIP / GPTBlockRendererHIP / GPTWorkResultHIP below = gpt_proc.{h,cpp} and
//    gpt_wr.{h,cpp}) -- Mitsuba's scheduler hands out RectangularWorkUnits in its spiral, a work unit is rendered by gdpt_render_rect into a
//    device film of the block's rows plus one row above and below, comes back as five ImageBlocks with a one-pixel border and is merged by
//    MultiFilm::putMulti exactly as gpt_proc.cpp:137-149 does -- so `-b blocksize`, the progress bar, the GUI's block preview and per-block
//    cancellation work as with `gpt`.  A GPU wants LARGE blocks (a 32 x 32 block is 1 024 lanes on a 131 072-lane device): use `-b 256` or more.
//  * the whole frame in one gdpt_render_rect call (`blocked` = false, and always with `devices` > 1): no per-block traffic, no block preview.
// To make this plugin THE `gpt` of a build, compile it as src/integrators/gpt/ (SConscript: plugins += env.SharedLibrary('gpt', ['gpt_hip.cpp'])):
// the XML's <integrator type="gpt"> then loads it; the properties are the reference's.
// tests/test_plugin_sources.py compiles this file against compile-only mock headers (tests/mitsuba_mock) and links it with the library.
#include <mitsuba/render/scene.h>
#include <mitsuba/render/renderjob.h>
#include <mitsuba/render/renderproc.h>
#include <mitsuba/render/rectwu.h>
#include <mitsuba/render/imageblock.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/bitmap.h>
#include <mitsuba/core/statistics.h>
#include <mitsuba/core/lock.h>
#include <thread>
#include "gdpt_tracer.h"
#include "gdpt_poisson.h"
#include "gdpt_mitsuba_scene.h"

MTS_NAMESPACE_BEGIN

/* ==================================================================== */
/*   GPTWorkResult (gpt_wr.h:36-97, gpt_wr.cpp:31-86): five ImageBlocks   */
/* ==================================================================== */
class GPTWorkResultHIP : public WorkResult {
public:
	enum { BUFFER_COUNT = 5 };     /* 0: preview / final, 1: throughput, 2: dx, 3: dy, 4: very direct */
	GPTWorkResultHIP(const ReconstructionFilter *rfilter, Vector2i blockSize, int extraBorder) {
		for (int i = 0; i < BUFFER_COUNT; ++i) {
			m_block[i] = new ImageBlock(Bitmap::ESpectrumAlphaWeight, blockSize, rfilter, -1, true, extraBorder);
			m_block[i]->setOffset(Point2i(0, 0));
			m_block[i]->setSize(blockSize);
		}
		m_block[2]->setAllowNegativeValues(true);
		m_block[3]->setAllowNegativeValues(true);
	}
	void clear() { for (int i = 0; i < BUFFER_COUNT; ++i) m_block[i]->clear(); }
	void load(Stream *stream) { for (int i = 0; i < BUFFER_COUNT; ++i) m_block[i]->load(stream); }
	void save(Stream *stream) const { for (int i = 0; i < BUFFER_COUNT; ++i) m_block[i]->save(stream); }
	ImageBlock *getImageBlock(int buffer) { return m_block[buffer].get(); }
	const ImageBlock *getImageBlock(int buffer) const { return m_block[buffer].get(); }
	void setSize(const Vector2i &size) { for (int i = 0; i < BUFFER_COUNT; ++i) m_block[i]->setSize(size); }
	void setOffset(const Point2i &offset) { for (int i = 0; i < BUFFER_COUNT; ++i) m_block[i]->setOffset(offset); }
	std::string toString() const { return m_block[0]->toString(); }
	MTS_DECLARE_CLASS()
private:
	ref<ImageBlock> m_block[BUFFER_COUNT];
};

class GradientPathIntegratorHIP;

/* ==================================================================== */
/*   GPTBlockRenderer (gpt_proc.cpp:47-128): renders work units           */
/* ==================================================================== */
class GPTBlockRendererHIP : public WorkProcessor {
public:
	GPTBlockRendererHIP(int blockSize) : m_integrator(NULL), m_sensor(NULL), m_film(NULL), m_filmY0(-1), m_filmY1(-1), m_blockSize(blockSize) { }
	GPTBlockRendererHIP(Stream *stream, InstanceManager *manager) : WorkProcessor(stream, manager), m_integrator(NULL), m_sensor(NULL), m_film(NULL), m_filmY0(-1), m_filmY1(-1) {
		m_blockSize = stream->readInt();
	}
	ref<WorkUnit> createWorkUnit() const { return new RectangularWorkUnit(); }
	ref<WorkResult> createWorkResult() const;
	void prepare();
	void process(const WorkUnit *workUnit, WorkResult *workResult, const bool &stop);
	void serialize(Stream *stream, InstanceManager *) const { stream->writeInt(m_blockSize); }
	ref<WorkProcessor> clone() const { return new GPTBlockRendererHIP(m_blockSize); }
	MTS_DECLARE_CLASS()
protected:
	virtual ~GPTBlockRendererHIP();
private:
	GradientPathIntegratorHIP *m_integrator;
	Sensor *m_sensor;
	gdpt_film *m_film;              /* the device film of the current band of rows: reused while consecutive work units share their rows */
	int m_filmY0, m_filmY1;
	int m_blockSize;
	std::vector<double> m_accum;
};

/* ==================================================================== */
/*   GPTRenderProcess (gpt_proc.h:41-64, gpt_proc.cpp:131-149)            */
/* ==================================================================== */
class GPTRenderProcessHIP : public BlockedRenderProcess {
public:
	GPTRenderProcessHIP(const RenderJob *parent, RenderQueue *queue, int blockSize) : BlockedRenderProcess(parent, queue, blockSize) { }
	ref<WorkProcessor> createWorkProcessor() const { return new GPTBlockRendererHIP(m_blockSize); }
	void processResult(const WorkResult *result, bool cancelled) {
		const GPTWorkResultHIP *block = static_cast<const GPTWorkResultHIP *>(result);
		UniqueLock lock(m_resultMutex);
		for (int i = 0; i < 5; ++i)
			m_film->putMulti(block->getImageBlock(i), i);
		m_progress->update(++m_resultCount);
		lock.unlock();
		m_queue->signalWorkEnd(m_parent, block->getImageBlock(0), cancelled);
	}
	MTS_DECLARE_CLASS()
protected:
	virtual ~GPTRenderProcessHIP() { }
};

class GradientPathIntegratorHIP : public Integrator {
	friend class GPTBlockRendererHIP;
public:
	GradientPathIntegratorHIP(const Properties &props) : Integrator(props), m_film(NULL), m_blockScene(NULL), m_process(NULL) {
		/* the properties and checks of gpt.cpp:1194-1213 */
		m_maxDepth = props.getInteger("maxDepth", -1);
		m_rrDepth = props.getInteger("rrDepth", 5);
		m_strictNormals = props.getBoolean("strictNormals", false);
		m_shiftThreshold = props.getFloat("shiftThreshold", Float(0.001));
		m_reconstructL1 = props.getBoolean("reconstructL1", true);
		m_reconstructL2 = props.getBoolean("reconstructL2", false);
		m_reconstructAlpha = (Float) props.getFloat("reconstructAlpha", Float(0.2));
		m_devices = props.getInteger("devices", 1);      /* not a reference property: GPUs to shard the frame over (row strips) */
		/* not a reference property.  false (default): the whole frame in one gdpt_render_rect -- what fills a 131 072-lane GPU.  true: the reference's
		   own shape, a BlockedRenderProcess whose workers hand blocks to the GPU (mitsuba -b, progress, per-unit cancel, border merge by putMulti); a
		   32x32 unit is 1 024 lanes, so units of at least 256x256 pixels are asked for whatever Scene::getBlockSize() says (see render()) */
		m_blocked = props.getBoolean("blocked", false);
		if (m_reconstructL1 && m_reconstructL2)
			Log(EError, "Disable 'reconstructL1' or 'reconstructL2': Cannot display two reconstructions at a time!");
		if (m_reconstructAlpha <= 0.0f)
			Log(EError, "'reconstructAlpha' must be set to a value greater than zero!");
		if (m_maxDepth <= 0 && m_maxDepth != -1)
			Log(EError, "'maxDepth' must be set to -1 (infinite) or a value greater than zero!");
	}

	GradientPathIntegratorHIP(Stream *stream, InstanceManager *manager) : Integrator(stream, manager), m_film(NULL), m_blockScene(NULL), m_process(NULL) {
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
		/* (a crop window: flatten() hands the film's size and crop offset to the C-ABI's camera, include/gdpt_tracer.h gdpt_camera; W x H is the crop) */

		gdpt_plugin::FlatScene fs;
		gdpt_plugin::flatten(scene, sensor.get(), size, fs);
		int kind; double p0, p1;
		gdpt_plugin::rfilterOf(film->getReconstructionFilter(), kind, p0, p1);

		const Sampler *sampler = static_cast<const Sampler *>(Scheduler::getInstance()->getResource(samplerResID, 0));
		gdpt_config cfg;
		cfg.maxDepth = m_maxDepth; cfg.rrDepth = m_rrDepth; cfg.strictNormals = m_strictNormals;
		cfg.spp = (int) sampler->getSampleCount(); cfg.shiftThreshold = m_shiftThreshold; cfg.seed = 5489ull;   /* random.h:113 */
		Log(EInfo, "Starting render job (GPT::render on %i MI355X) (%ix%i, %i samples) ..", m_devices, W, H, cfg.spp);

		std::vector<float> img[5];
		for (int b = 0; b < 5; ++b) img[b].resize((size_t) 3 * W * H);
		if (m_devices > 1) {
			if (!renderStrips(fs, cfg, kind, p0, p1, W, H, img)) return false;
		} else if (m_blocked) {
			/* gpt.cpp:1396-1414: "This is a sampling-based integrator - parallelize": the scheduler's workers each drive the GPU through a
			   GPTBlockRendererHIP; the blocks' borders are merged by addition in MultiFilm::putMulti (gpt_proc.cpp:137-149) */
			ref<Scheduler> sched = Scheduler::getInstance();
			m_blockScene = gdpt_plugin::upload(fs, -1);
			m_blockCfg = cfg;
			m_blockFilter[0] = kind; m_blockFilterP[0] = p0; m_blockFilterP[1] = p1;
			ref<BlockedRenderProcess> proc = new GPTRenderProcessHIP(job, queue, std::max((int) scene->getBlockSize(), 256));
			int integratorResID = sched->registerResource(this);
			proc->bindResource("integrator", integratorResID);
			proc->bindResource("scene", sceneResID);
			proc->bindResource("sensor", sensorResID);
			proc->bindResource("sampler", samplerResID);
			scene->bindUsedResources(proc);
			bindUsedResources(proc);
			sched->schedule(proc);
			m_process = proc;
			sched->wait(proc);
			sched->unregisterResource(integratorResID);
			m_process = NULL;
			const bool ok = proc->getReturnStatus() == ParallelProcess::ESuccess;
			gdpt_scene_destroy(m_blockScene);
			m_blockScene = NULL;
			if (!ok) return false;
			/* gpt.cpp:1419-1442: develop the five buffers of the MultiFilm and cast them to float */
			for (int b = 0; b < 5; ++b) {
				ref<Bitmap> bmp = new Bitmap(Bitmap::ESpectrum, Bitmap::EFloat, size);
				film->developMulti(Point2i(0, 0), size, Point2i(0, 0), bmp, b);
				const Float *src = bmp->getFloatData();
				for (size_t i = 0; i < img[b].size(); ++i) img[b][i] = (float) src[i];
			}
		} else {
			/* GPTBlockRenderer::process over the whole film in one call (blocks may also be handed over one by one) */
			gdpt_scene *gs = gdpt_plugin::upload(fs, -1);
			check(gdpt_film_create(gs, 0, H, &m_film));
			check(gdpt_film_set_rfilter(m_film, kind, p0, p1));
			check(gdpt_render_rect(gs, &cfg, 0, 0, W, H, m_film));
			check(gdpt_film_sync(m_film));
			int cancelled = 0;
			check(gdpt_film_cancelled(m_film, &cancelled));
			if (!cancelled)
				for (int b = 0; b < 5; ++b) check(gdpt_film_develop(m_film, b, img[b].data()));
			release(gs);
			if (cancelled) return false;
		}

		/* img[] = developMulti + the float casts of gpt.cpp:1419-1442, done on the device */
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
		return true;
	}

	/* Integrator::cancel, integrator.h:88; gpt.cpp:1482-1485 for the process */
	void cancel() {
		if (m_film) gdpt_film_cancel(m_film);
		if (m_process) Scheduler::getInstance()->cancel(m_process);
	}
	void postprocess(const Scene *, RenderQueue *, const RenderJob *, int, int, int) {}
	void serialize(Stream *stream, InstanceManager *manager) const { Integrator::serialize(stream, manager); }
	std::string toString() const { return "GradientPathIntegratorHIP[]"; }

	MTS_DECLARE_CLASS()
	static void check(int rc) { if (rc != GDPT_OK) SLog(EError, "gpt_hip: %s", gdpt_last_error()); }
private:
	void release(gdpt_scene *gs) { gdpt_film_destroy(m_film); m_film = NULL; gdpt_scene_destroy(gs); }

	/* One thread and one strip film per GPU; border sums exchanged device to device (gdpt_film_pack_halo / unpack_halo over gdpt_device_copy =
	   a peer DMA over xGMI), every strip develops its own rows.  The reference's counterpart is the block border merged by addition
	   (gpt_proc.cpp:52-56,137-149).  Same strips and payloads as host/gdpt_host.hpp renderStrips and parallel.StripRenderer. */
	bool renderStrips(const gdpt_plugin::FlatScene &fs, const gdpt_config &cfg, int kind, double p0, double p1, int W, int H, std::vector<float> img[5]) {
		int have = 0;
		check(gdpt_device_count(&have));
		const int N = m_devices;
		if (N > have || N > H) Log(EError, "gpt_hip: 'devices' = %i, but the node has %i GPUs and the image %i rows", N, have, H);
		struct Strip { int y0, y1; gdpt_scene *scene; gdpt_film *film; void *out[2], *in[2]; std::string error; };
		std::vector<Strip> strips(N);
		for (int r = 0, y = 0; r < N; ++r) {
			const int n = H / N + (r < H % N ? 1 : 0);
			Strip &s = strips[r];
			s.y0 = y; s.y1 = y + n; y += n; s.scene = NULL; s.film = NULL; s.out[0] = s.out[1] = s.in[0] = s.in[1] = NULL;
		}
		std::vector<std::thread> workers;
		for (int r = 0; r < N; ++r)
			workers.emplace_back([&, r]() {
				Strip &s = strips[r];
				try {
					s.scene = gdpt_plugin::upload(fs, r);
					check(gdpt_film_create(s.scene, s.y0, s.y1, &s.film));
					check(gdpt_film_set_rfilter(s.film, kind, p0, p1));
					check(gdpt_render_rect(s.scene, &cfg, 0, s.y0, W, s.y1, s.film));
					check(gdpt_film_sync(s.film));
				} catch (const std::exception &e) { s.error = e.what(); }
			});
		for (size_t i = 0; i < workers.size(); ++i) workers[i].join();
		std::string error;
		for (int r = 0; r < N; ++r) if (!strips[r].error.empty()) error = strips[r].error;
		if (error.empty() && kind == GDPT_RFILTER_BOX) {     /* wider filters: every strip rendered the filter's reach itself */
			size_t bytes = 0;
			check(gdpt_film_halo_bytes(strips[0].film, &bytes));
			for (int r = 0; r < N; ++r)
				for (int which = 0; which < 2; ++which) {
					if ((which == 0 && r == 0) || (which == 1 && r == N - 1)) continue;
					check(gdpt_device_alloc(r, bytes, &strips[r].out[which]));
					check(gdpt_device_alloc(r, bytes, &strips[r].in[which]));
					check(gdpt_film_pack_halo(strips[r].film, which, strips[r].out[which]));       /* every strip packs before anyone unpacks */
				}
			for (int r = 0; r + 1 < N; ++r) {
				check(gdpt_device_copy(r + 1, strips[r + 1].in[0], r, strips[r].out[1], bytes));
				check(gdpt_device_copy(r, strips[r].in[1], r + 1, strips[r + 1].out[0], bytes));
			}
			for (int r = 0; r < N; ++r)
				for (int which = 0; which < 2; ++which)
					if (strips[r].in[which]) check(gdpt_film_unpack_halo(strips[r].film, which, strips[r].in[which]));
		}
		if (error.empty())
			for (int r = 0; r < N; ++r)
				for (int b = 0; b < 5; ++b)
					check(gdpt_film_develop(strips[r].film, b, img[b].data() + (size_t) 3 * W * strips[r].y0));
		for (int r = 0; r < N; ++r) {
			for (int k = 0; k < 2; ++k) { gdpt_device_free(r, strips[r].out[k]); gdpt_device_free(r, strips[r].in[k]); }
			gdpt_film_destroy(strips[r].film);
			gdpt_scene_destroy(strips[r].scene);
		}
		if (!error.empty()) Log(EError, "%s", error.c_str());
		return true;
	}

	gdpt_film *m_film;
	gdpt_scene *m_blockScene;       /* blocked shape: the device scene the workers' GPTBlockRendererHIP render from */
	gdpt_config m_blockCfg;
	int m_blockFilter[1];           /* the film's reconstruction filter, as gdpt_film_set_rfilter takes it */
	double m_blockFilterP[2];
	ParallelProcess *m_process;
	int m_maxDepth, m_rrDepth, m_devices;
	bool m_strictNormals, m_reconstructL1, m_reconstructL2, m_blocked;
	Float m_shiftThreshold, m_reconstructAlpha;
};

/* ---- GPTBlockRendererHIP: needs the integrator's members ---- */
ref<WorkResult> GPTBlockRendererHIP::createWorkResult() const {
	return new GPTWorkResultHIP(m_sensor ? m_sensor->getFilm()->getReconstructionFilter() : NULL, Vector2i(m_blockSize, m_blockSize), 1);    /* gpt_proc.cpp:52-56: extraBorder = 1 */
}

void GPTBlockRendererHIP::prepare() {      /* gpt_proc.cpp:58-72: the resources the process bound */
	m_sensor = static_cast<Sensor *>(getResource("sensor"));
	m_integrator = static_cast<GradientPathIntegratorHIP *>(getResource("integrator"));
}

GPTBlockRendererHIP::~GPTBlockRendererHIP() { if (m_film) gdpt_film_destroy(m_film); }

/* gpt_proc.cpp:74-91 with renderBlock replaced by gdpt_render_rect: the unit's pixels are sampled on the device; the 15 puts of every sample
   (gpt.cpp:1314-1352) land in the unit's rectangle grown by the blocks' border -- the filter's reach + the extra pixel of the neighbour puts
   (gpt_proc.cpp:52-56, imageblock.cpp:23-38) --, which is what comes back in the five ImageBlocks.  With a filter wider than box the C-ABI renders
   a sub-rectangle as exactly such a block (include/gdpt_tracer.h, gdpt_film_set_rfilter).  Unit offsets are relative to the crop window
   (BlockedRenderProcess::bindResource, renderproc.cpp:158-181); the C-ABI's camera has crop == film, which flatten() checks. */
void GPTBlockRendererHIP::process(const WorkUnit *workUnit, WorkResult *workResult, const bool &stop) {
	const RectangularWorkUnit *rect = static_cast<const RectangularWorkUnit *>(workUnit);
	GPTWorkResultHIP *block = static_cast<GPTWorkResultHIP *>(workResult);
	block->setOffset(rect->getOffset());
	block->setSize(rect->getSize());
	block->clear();
	if (stop) return;
	const Vector2i crop = m_sensor->getFilm()->getCropSize();
	const int W = crop.x, H = crop.y;
	/* (a film with high-quality edges hands out units that reach beyond the crop window, renderproc.cpp:166-171: their outside part has no pixels here) */
	const int ux0 = rect->getOffset().x, uy0 = rect->getOffset().y;
	const int x0 = std::max(0, ux0), y0 = std::max(0, uy0), x1 = std::min(W, ux0 + rect->getSize().x), y1 = std::min(H, uy0 + rect->getSize().y);
	if (x0 >= x1 || y0 >= y1) return;
	const int border = block->getImageBlock(0)->getBorderSize();
	const int fy0 = std::max(0, y0 - border), fy1 = std::min(H, y1 + border);          /* the film owns the border rows too: their sums come back resolved */
	const int fx0 = std::max(0, x0 - border), fx1 = std::min(W, x1 + border);
	if (!m_film || fy0 != m_filmY0 || fy1 != m_filmY1) {
		if (m_film) gdpt_film_destroy(m_film);
		m_film = NULL;
		GradientPathIntegratorHIP::check(gdpt_film_create(m_integrator->m_blockScene, fy0, fy1, &m_film));
		GradientPathIntegratorHIP::check(gdpt_film_set_rfilter(m_film, m_integrator->m_blockFilter[0], m_integrator->m_blockFilterP[0], m_integrator->m_blockFilterP[1]));
		m_filmY0 = fy0; m_filmY1 = fy1;
	} else GradientPathIntegratorHIP::check(gdpt_film_clear(m_film));
	GradientPathIntegratorHIP::check(gdpt_render_rect(m_integrator->m_blockScene, &m_integrator->m_blockCfg, x0, y0, x1, y1, m_film));
	GradientPathIntegratorHIP::check(gdpt_film_sync(m_film));
	if (stop) return;                        /* (the scheduler drops the result of a cancelled unit) */
	/* only the block and its border travel back: accum[5][rows][cols][4] = (R, G, B, weight) */
	const int rows = fy1 - fy0, cols = fx1 - fx0;
	m_accum.resize((size_t) 5 * rows * cols * 4);
	GradientPathIntegratorHIP::check(gdpt_film_accum_rect(m_film, fx0, fy0, fx1, fy1, m_accum.data()));
	/* -> the block's bitmap: SPECTRUM_SAMPLES + 2 channels per pixel (spectrum, alpha, weight; GPTWorkResult::put, gpt_wr.h:57-65, writes alpha = 1
	   per put, which MultiFilm's develop never reads: it is set to the weight here) */
	for (int b = 0; b < 5; ++b) {
		ImageBlock *ib = block->getImageBlock(b);
		Bitmap *bmp = ib->getBitmap();
		const int bw = bmp->getWidth(), ch = bmp->getChannelCount();
		Float *dst = bmp->getFloatData();
		for (int yy = fy0; yy < fy1; ++yy)
			for (int xx = fx0; xx < fx1; ++xx) {
				const double *a = &m_accum[((((size_t) b * rows) + (yy - fy0)) * cols + (xx - fx0)) * 4];
				Float *d = dst + ((size_t) (yy - uy0 + border) * bw + (xx - ux0 + border)) * ch;
				for (int c = 0; c < ch - 2; ++c) d[c] = (Float) a[c < 3 ? c : 2];
				d[ch - 2] = (Float) a[3];
				d[ch - 1] = (Float) a[3];
			}
	}
}

MTS_IMPLEMENT_CLASS(GPTWorkResultHIP, false, WorkResult)
MTS_IMPLEMENT_CLASS_S(GPTBlockRendererHIP, false, WorkProcessor)
MTS_IMPLEMENT_CLASS(GPTRenderProcessHIP, false, BlockedRenderProcess)
MTS_IMPLEMENT_CLASS_S(GradientPathIntegratorHIP, false, Integrator)
MTS_EXPORT_PLUGIN(GradientPathIntegratorHIP, "Gradient-domain path tracer on MI355X (libgdpt_hip)");
MTS_NAMESPACE_END
