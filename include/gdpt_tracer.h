/*
 * include/gdpt_tracer.h -- C-ABI of the MI355X gradient-domain path tracer (G-PT sampling stage).
 *
 * Drop-in boundary for the per-block work of the reference's `gpt` integrator plugin
 * (/root/reference/src/integrators/gpt/): what GPTBlockRenderer::process -> GradientPathIntegrator::
 * renderBlock -> GradientPathTracer::evaluatePoint/evaluate compute for a rectangle of pixels
 * (gpt_proc.cpp:75-93, gpt.cpp:1220-1355, 397-436, 468-1180), and what GPTRenderProcess::processResult +
 * MultiFilm::putMulti/developMulti do with the result (gpt_proc.cpp:137-149, multifilm.cpp:366-416).
 * A maintainer's `GradientPathIntegrator::render` keeps its own scheduling and film; it hands this
 * library the scene once and then asks for tiles.  See INTEGRATION.md for the binding.
 *
 * Scene subset carried: triangle soups without vertex normals/texcoords, `area` emitters on
 * meshes, `diffuse` / `conductor` / `roughconductor` BSDFs, `perspective` (or `thinlens`) sensor, `box` rfilter, independent
 * sampling.  Arithmetic is fp64 like the reference's DOUBLE_PRECISION build.  Random numbers: one counter-based
 * stream per (seed, pixel, sample) -- a GPU cannot consume the reference's serial SFMT stream at speed; gdpt_render_serial
 * below is the same sampler fed by that stream in a one-worker render's order, on one lane: the validation path.
 *
 * Plain C; status codes and gdpt_last_error() as in gdpt_poisson.h.  No CPU fallback.
 */
#ifndef GDPT_TRACER_H
#define GDPT_TRACER_H

#include "gdpt_poisson.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GDPT_MAT_DIFFUSE        0   /* src/bsdfs/diffuse.cpp        */
#define GDPT_MAT_CONDUCTOR      1   /* src/bsdfs/conductor.cpp      */
#define GDPT_MAT_ROUGHCONDUCTOR 2   /* src/bsdfs/roughconductor.cpp */
#define GDPT_MAT_DIELECTRIC     3   /* src/bsdfs/dielectric.cpp: eta[0] = intIOR/extIOR, reflectance = specularReflectance, k = specularTransmittance */
#define GDPT_DISTR_BECKMANN     0   /* src/bsdfs/microfacet.h EBeckmann */
#define GDPT_DISTR_GGX          1   /* EGGX */
#define GDPT_DISTR_PHONG        2   /* EPhong: Phong / Ashikhmin-Shirley ("phong" or "as"); never sampled by visible normals */

typedef struct gdpt_material {
    int    type;            /* GDPT_MAT_*                                                        */
    int    distribution;    /* GDPT_DISTR_* (roughconductor `distribution`)                      */
    int    sampleVisible;   /* roughconductor `sampleVisible` (default 1)                        */
    int    twoSided;        /* 1 = wrapped in `twosided` (src/bsdfs/twosided.cpp), same BRDF both sides */
    double reflectance[3];  /* diffuse `reflectance`; conductors `specularReflectance`           */
    double eta[3], k[3];    /* conductors `eta`, `k` (RGB)                                       */
    double alphaU, alphaV;  /* roughconductor `alpha` / `alphaU`,`alphaV`                        */
} gdpt_material;

typedef struct gdpt_emitter {   /* an `area` emitter attached to one mesh (src/emitters/area.cpp), or a `point` emitter (point.cpp) */
    int    firstTri, numTris;   /* area: that mesh's triangles, contiguous in the soup; point: numTris = -1                  */
    double radiance[3];         /* area: radiance; point: intensity                                                           */
    double position[3];         /* point emitters only (the `position` / translation of `toWorld`)                          */
    int    rectangle;           /* 1: the mesh is a `rectangle` shape (src/shapes/rectangle.cpp) given as the two triangles of its createTriMesh():  */
    double rectToWorld[12];     /*    light samples are drawn as the shape draws them, objectToWorld(2u - 1, 2v - 1, 0) (rectangle.cpp:200-206), with  */
    double rectNormal[3];       /*    its frame's normal (normalize(objectToWorld(Normal(0,0,1)))) and pdf 1 / (|dpdu| |dpdv|); rows of the 3x4, incl. flipNormals */
} gdpt_emitter;

typedef struct gdpt_environment {   /* `<emitter type="constant">` (src/emitters/constant.cpp): uniform radiance from all directions -- or, with */
    double radiance[3];             /* `rgb` set, `<emitter type="envmap">` (src/emitters/envmap.cpp): a latitude-longitude bitmap              */
    int    index;               /* its position in the scene's emitter list (XML order; Scene::sampleEmitterDirect picks by it); <0: last */
    const double *rgb;          /* envmap: height x width x 3 LINEAR values, top row first (v = 0: straight up); NULL = the constant emitter.  The library  */
    int    width, height;       /*   keeps it as the plugin does: MIP pyramid with half-precision texels, repeat in u / clamp in v, EWA lookups   */
    double scale;               /*   (maxAnisotropy 10) for camera rays, level-0 bilinear otherwise, float cdfs for light sampling (envmap.cpp:135-138,258-325) */
    double toWorld[9];          /* envmap: linear part of the emitter's `toWorld`, row-major (identity: +y is up, u = 0.5 looks along -z)   */
} gdpt_environment;

typedef struct gdpt_camera {    /* `perspective` sensor (src/sensors/perspective.cpp) or `thinlens` (thinlens.cpp) */
    double toWorld[16];         /* row-major camera-to-world, Transform::lookAt convention         */
    double fovX;                /* degrees (`fov`, fovAxis = x)                                    */
    double nearClip, farClip;
    int    width, height;       /* size of the rendered image in pixels: the film's CROP window (Film::getCropSize; the whole film when there is no crop) */
    int    type;                /* GDPT_SENSOR_PERSPECTIVE (0) | GDPT_SENSOR_THINLENS: `thinlens` (src/sensors/thinlens.cpp), the aperture sample of gpt.cpp:1262-1264 */
    double apertureRadius;      /* thinlens `apertureRadius`                                        */
    double focusDistance;       /* thinlens `focusDistance`                                         */
    double shutterOpen;         /* `shutterOpen` / `shutterClose` (Sensor::Sensor, sensor.cpp:26-38): with shutterClose > shutterOpen every sample draws  */
    double shutterClose;        /*   its time sample (gpt.cpp:1265-1267, gbdpt_proc.cpp:156-157); both 0 = no shutter.  Transforms are static: the time    */
                                /*   of a ray moves nothing, the draw keeps the sample's random stream where the reference's is                           */
    int    cropOffsetX, cropOffsetY;   /* the film's crop window (film.cpp `cropOffsetX/Y`, `cropWidth/Height`): pixel (x, y) of the rendered image is pixel          */
    int    fullWidth, fullHeight;      /*   (x + cropOffsetX, y + cropOffsetY) of a fullWidth x fullHeight film -- the sensor's rays, aspect and pixel differentials  */
                                       /*   come from the FULL film's raster (perspective.cpp:126-163, steps 4+5 of m_cameraToSample).  fullWidth == 0: no crop      */
                                       /*   (the film is width x height).  G-PT only: the G-BDPT entry points return GDPT_ERR_UNSUPPORTED for a cropped camera       */
} gdpt_camera;
#define GDPT_SENSOR_PERSPECTIVE 0
#define GDPT_SENSOR_THINLENS    1

/* GradientPathTracerConfig (gpt.h:35-68) + sampler settings.  minDepth is forced to 1 (gpt.cpp:1369). */
typedef struct gdpt_config {
    int    maxDepth;            /* -1 = unbounded (gpt.cpp:1194)                                   */
    int    rrDepth;             /* 5                                                               */
    int    strictNormals;       /* 0                                                               */
    int    spp;                 /* sampler sampleCount                                             */
    double shiftThreshold;      /* 0.001                                                           */
    unsigned long long seed;    /* 5489 echoes random.h:113                                        */
} gdpt_config;

typedef struct gdpt_scene gdpt_scene;
typedef struct gdpt_film  gdpt_film;

/* Upload a scene: builds the flat BVH on the host, lays triangles out in leaf order in HBM. device: -1 = current. */
GDPT_API int  gdpt_scene_create(int numTris, const double *verts9, const int *triMaterial,
                                int numMaterials, const gdpt_material *materials,
                                int numEmitters, const gdpt_emitter *emitters,
                                const gdpt_camera *camera, int device, gdpt_scene **out);
/* The same with an environment emitter (NULL = none; then at least one area emitter is required).  Carries the environment
 * branches of the integrator: environment hits of base and offset paths, environmentShift / testEnvironmentVisibility
 * (gpt.cpp:96-114,348-369,786-804,1052-1074), light sampling of the environment with the bounding sphere of constant.cpp:67-70. */
GDPT_API int  gdpt_scene_create_env(int numTris, const double *verts9, const int *triMaterial,
                                    int numMaterials, const gdpt_material *materials,
                                    int numEmitters, const gdpt_emitter *emitters, const gdpt_environment *env,
                                    const gdpt_camera *camera, int device, gdpt_scene **out);
/* The same with per-vertex normals: normals9 = 9 doubles per triangle (n0, n1, n2), three zero vectors = that triangle is flat;
 * NULL = no vertex normals at all.  Shading frame and geometric normal then follow fillIntersectionRecord (skdtree.h:382-397):
 * interpolated shading normal, geometric normal flipped to its side, (s, t) from computeShadingFrame with dpdu = p1 - p0.
 * Emitter triangles must be flat (GDPT_ERR_UNSUPPORTED otherwise). */
GDPT_API int  gdpt_scene_create_ex(int numTris, const double *verts9, const double *normals9, const int *triMaterial,
                                   int numMaterials, const gdpt_material *materials,
                                   int numEmitters, const gdpt_emitter *emitters, const gdpt_environment *env,
                                   const gdpt_camera *camera, int device, gdpt_scene **out);
/* `<texture type="bitmap">` (src/textures/bitmap.cpp) on a material's `reflectance` (diffuse) / `specularReflectance` (conductor,
 * roughconductor, dielectric), as the G-PT path evaluates it: Texture2D::eval scales and offsets its.uv (texture.cpp:112-121); `nearest` and
 * `bilinear` look level 0 up by MIPMap::evalBox / evalBilinear (bitmap.cpp:431-452, mipmap.h:566-596,628-633); `trilinear` and `ewa` (the
 * reference's default) do the same EXCEPT at the hit of a camera ray, whose UV partials (Intersection::computePartials from the ray's
 * differentials) select MIP levels / an elliptical footprint (mipmap.h:628-712,744-833).  The library builds the pyramid from level 0
 * as TMIPMap does (Lanczos resampling, values clamped to [0, 1]); negative texels are clamped to 0 like there. */
#define GDPT_TEXWRAP_REPEAT 0      /* ReconstructionFilter::ERepeat ... as bitmap.cpp:324-338 parses wrapMode */
#define GDPT_TEXWRAP_CLAMP  1
#define GDPT_TEXWRAP_MIRROR 2
#define GDPT_TEXWRAP_ZERO   3      /* "zero" / "black" */
#define GDPT_TEXWRAP_ONE    4      /* "one" / "white"  */
#define GDPT_TEXFILTER_NEAREST   0
#define GDPT_TEXFILTER_BILINEAR  1
#define GDPT_TEXFILTER_TRILINEAR 2
#define GDPT_TEXFILTER_EWA       3   /* bitmap.cpp:213 default */
typedef struct gdpt_texture {
    int width, height;
    const double *rgb;              /* height x width x 3 LINEAR values, top row first (what the MIP map's level 0 holds: the file converted to Float) */
    int wrapU, wrapV, filter;
    double uscale, vscale, uoffset, voffset;   /* Texture2D: `uscale`, `vscale`, `uoffset`, `voffset` */
    double scale;                   /* BSDF::ensureEnergyConservation (bsdf.cpp): 0.99 / max when the texture exceeds 1, else 1; applied to the looked-up value */
    double maxAnisotropy;           /* `maxAnisotropy` (bitmap.cpp:232, default 20): bound on the EWA ellipse's aspect ratio; used by GDPT_TEXFILTER_EWA only */
} gdpt_texture;
/* The same as gdpt_scene_create_ex with texture coordinates and bitmap textures.  uvs6 = 6 doubles per triangle (u0 v0 u1 v1 u2 v2), NULL = no
 * mesh has texture coordinates; triHasUV (NULL = all) marks the triangles whose mesh has them -- the others get its.uv = the hit's
 * barycentrics (b1, b2), as fillIntersectionRecord does (skdtree.h:398-405).  materialTexture[i] = texture of material i, -1 = constant. */
GDPT_API int  gdpt_scene_create_tex(int numTris, const double *verts9, const double *normals9, const double *uvs6, const unsigned char *triHasUV,
                                    const int *triMaterial, int numMaterials, const gdpt_material *materials, const int *materialTexture,
                                    int numTextures, const gdpt_texture *textures,
                                    int numEmitters, const gdpt_emitter *emitters, const gdpt_environment *env,
                                    const gdpt_camera *camera, int device, gdpt_scene **out);
GDPT_API void gdpt_scene_destroy(gdpt_scene *s);

/* A film = the five G-PT buffers `-final -throughput -dx -dy -direct` (gpt.cpp:1380) over rows [y0, y1) of the
 * image plus a halo above and below (GPTWorkResult's border, gpt_proc.cpp:52-56, gpt_wr.cpp:31-44: the filter's 1 + extraBorder 1), as
 * per-pixel sample sums on the device: one row of neighbour records, and two rows for the exact puts of the rare samples that sit
 * within 1e-5 of a pixel edge and so land in two pixels (the box filter's radius is 0.5 + 1e-5).  Single GPU: y0 = 0, y1 = height. */
GDPT_API int  gdpt_film_create(gdpt_scene *s, int y0, int y1, gdpt_film **out);
GDPT_API void gdpt_film_destroy(gdpt_film *f);
GDPT_API int  gdpt_film_clear(gdpt_film *f);

/* Render pixels [x0,x1) x [y0,y1) (must lie inside the film's rows) with cfg->spp samples each, ADDING into the
 * film (GPTBlockRenderer::process + processResult).  Asynchronous on the film's stream; gdpt_film_sync waits. */
GDPT_API int  gdpt_render_rect(gdpt_scene *s, const gdpt_config *cfg, int x0, int y0, int x1, int y1, gdpt_film *f);
GDPT_API int  gdpt_film_sync(gdpt_film *f);
/* The film as `mitsuba -p 1` samples it: ONE lane renders every pixel of the (whole-image, box-filter) film with cfg->spp samples each, in the order a
 * single worker does -- BlockedImageProcess's spiral of blockSize x blockSize blocks (imageproc.cpp:28-78; Scene::getBlockSize(), 32), Hilbert order inside
 * a block (gpt_proc.cpp:84-87, sfcurve.h:34-107), samples in index order (gpt.cpp:1245-1268) -- drawing every random number from ONE SFMT-19937 stream:
 * the clone of the scene's IndependentSampler, seeded by init_by_array from 312 draws of a parent Random(parentSeed) (renderjob.cpp:59-66,
 * independent.cpp:71-80, random.cpp:400-467,519-524; parentSeed 5489 = random.h:113).  cfg->seed is not used.  ADDS into the film like gdpt_render_rect;
 * synchronous; minutes per megasample -- a validation path (tests hold it against the oracle's render_serial), never the product's renderer.
 * draws (may be NULL): random numbers consumed. */
GDPT_API int  gdpt_render_serial(gdpt_scene *s, const gdpt_config *cfg, gdpt_film *f, int blockSize, unsigned long long parentSeed, unsigned long long *draws);
/* Host-only probes of the two integer pieces of that path (no device needed; the CPU suite pins them): n successive 64-bit outputs (Random::nextULong,
 * random.cpp:285-293) of Random(seed) -- or, with cloned != 0, of a Random seeded from a parent Random(seed) as a worker's sampler is --; and the pixel order
 * of a width x height film, xy = 2 ints per pixel. */
GDPT_API int  gdpt_serial_random(unsigned long long seed, int cloned, int n, unsigned long long *out);
GDPT_API int  gdpt_serial_pixel_order(int width, int height, int blockSize, int *xy);
/* Integrator::cancel (include/mitsuba/render/integrator.h:88; the `stop` flag polled per pixel and sample, gpt.cpp:1246,1254): may be
 * called from another thread while a gdpt_render_rect is running.  Waves stop starting samples, running base paths finish, the film
 * keeps what was accumulated (with a filter wider than box: the chunks gathered so far) and later gdpt_render_rect calls into this
 * frame render nothing.  gdpt_film_clear starts a new, uncancelled frame.  gdpt_film_cancelled reports the flag. */
GDPT_API int  gdpt_film_cancel(gdpt_film *f);
GDPT_API int  gdpt_film_cancelled(gdpt_film *f, int *out);

/* Halo exchange for row-strip sharding (DESIGN.md "multi-GPU").  A strip's pixels need the per-pixel sample sums of the
 * rows just outside it (the neighbour samples that splat into it) -- the reference's block border merged by addition
 * (gpt_proc.cpp:52-56,137-149).  pack(which) writes this strip's boundary payload for the neighbour ABOVE (which = 0) or
 * BELOW (which = 1) into a device buffer of gdpt_film_halo_bytes(); unpack(which, buf) consumes the payload received
 * from the neighbour on that side.  Payload: the boundary row's records, then the exact puts this strip's samples made into the
 * TWO rows beyond it (round 5; one until then, which lost about one put in 50 000 next to a boundary).  Strips need >= 2 rows.
 * Transport (RCCL send/recv) is the caller's. */
GDPT_API int  gdpt_film_halo_bytes(gdpt_film *f, size_t *bytes);
GDPT_API int  gdpt_film_pack_halo(gdpt_film *f, int which, void *devBuf);
GDPT_API int  gdpt_film_unpack_halo(gdpt_film *f, int which, const void *devBuf);

/* Resolve the per-pixel sums into the five accumulation buffers exactly as 15 ImageBlock::put calls per sample with
 * the box filter would have (gpt.cpp:1314-1352, imageblock.h:150-199): accum[5][rows][width][4] doubles (R,G,B,weight)
 * on the HOST, rows = y1 - y0. */
GDPT_API int  gdpt_film_accum(gdpt_film *f, double *accum);
/* The same for the pixels [x0,x1) x [y0,y1) of the film only: accum[5][y1 - y0][x1 - x0][4] -- what a block-wise host copies back per work unit
 * (GPTBlockRenderer::process's five ImageBlocks with their border, src/integrators/gpt/gpt_proc.cpp:74-91, gpt_wr.cpp:31-44). */
GDPT_API int  gdpt_film_accum_rect(gdpt_film *f, int x0, int y0, int x1, int y1, double *accum);
/* MultiFilm::developMulti (multifilm.cpp:366-416; weight division fmtconv.cpp:955-1058) of one buffer to fp32 RGB
 * (the std::transform casts of gpt.cpp:1439-1442) into a DEVICE buffer of 3*rows*width floats -- the solver's input. */
GDPT_API int  gdpt_film_develop_device(gdpt_film *f, int buffer, float *rgbDevice);
GDPT_API int  gdpt_film_develop(gdpt_film *f, int buffer, float *rgbHost);

/* Counters since the last clear: [0] closest-hit queries, [1] any-hit queries (raysTraced / shadowRaysTraced,
 * skdtree.cpp:46-47,123,151,211), [2] base paths, [3] sum of base-path lengths (avgPathLength, gpt.cpp:72,1178). */
GDPT_API int  gdpt_film_stats(gdpt_film *f, unsigned long long stats[4]);
/* Puts dropped since the last clear by the validity check of ImageBlock::put (imageblock.h:154-158: a non-finite channel, or a
 * negative one in a buffer other than dx / dy, gpt_wr.cpp:38-42) -- the reference's "Invalid sample value" warnings.  A dropped
 * put leaves neither value nor weight; one such sample no longer poisons its pixel (and, through the CG's global dot products,
 * the whole reconstruction). */
GDPT_API int  gdpt_film_invalid_puts(gdpt_film *f, unsigned long long *count);
/* HIP-event time of the render kernels enqueued since the last clear (milliseconds). */
GDPT_API float gdpt_film_render_ms(gdpt_film *f);
GDPT_API void *gdpt_film_stream(gdpt_film *f);
/* Tuning knob (no reference counterpart): which build of the render kernel to launch -- the one compiled for 1, 2 (default),
 * 3 or 4 resident waves per SIMD (register budget 512 / n per lane; builds exist for 2 and 4: 1 runs the 2-wave build, 3 the
 * 4-wave build); a negative value selects the same build with the
 * per-sample sums kept in registers instead of LDS.  Results are identical; only speed differs. */
GDPT_API int  gdpt_film_set_occupancy(gdpt_film *f, int wavesPerSimd);
/* Tuning knob (no reference counterpart): how gdpt_render_rect is staged on the device.  stages = 2 (default): the five primary rays
 * of every sample are traced by a traversal-only kernel (k_primary), the general kernel (k_render) walks a sample until each of its
 * four offset paths is connected to the base path or dead (diffuse / rough scenes: two bounces), the rest of the base path runs in
 * the continuation kernel (k_continue), and every sample's sums are added to its pixel once per chunk of samples (k_fold_cont).
 * stages = 0: everything in k_render (the round-1 form; gdpt_film_set_occupancy chooses its build); 1 is accepted and means 2.  Samples,
 * random numbers, ray counts and the order in which a pixel's samples are summed do not depend on the setting; results agree to
 * rounding of the per-pixel sums.
 * stages = 3 (round 4, opt-in, measured SLOWER than 2 -- DESIGN.md): the first GDPT_WF_ITERS (default 6) bounces of the continuation phase run
 * in wavefront form (csrc/gpt_wavefront.hip.h: rays through HBM queues to traversal-only kernels, shading passes that replay the one bounce()
 * around them), k_continue takes what is left; films, ray counts and statistics are bit-identical to stages = 2.
 * refillLanes: idle lanes of a wave of k_continue before they take new records together (0 = keep the current value, default 48).
 * Environment: GDPT_NO_CONTINUATION and GDPT_QUEUE_MB (memory budget of the sample queue, default
 * 24576) override at render time (GDPT_NO_CONTINUATION set = stages 0). */
GDPT_API int  gdpt_film_set_pipeline(gdpt_film *f, int stages, int refillLanes);

/* ---- multi-device helpers (csrc/device_capi.hip) ---------------------------------------------------------------------------------
 * The reference shards a frame over CPU worker threads by image blocks and merges block borders by addition (imageproc.cpp:28-78,
 * gpt_proc.cpp:52-56,137-149).  A C++ host that shards it over the GPUs of one node (host/gdpt_multi.hpp: one thread and one strip
 * film per device) needs, besides the entry points above (gdpt_scene_create_ex takes the device, a film lives on its scene's device),
 * device memory it can name and copies between devices.  A copy between two GPUs is a peer-to-peer DMA over xGMI. */
GDPT_API int  gdpt_device_count(int *count);
GDPT_API int  gdpt_device_alloc(int device, size_t bytes, void **ptr);
GDPT_API int  gdpt_device_free(int device, void *ptr);
GDPT_API int  gdpt_device_copy(int dstDevice, void *dst, int srcDevice, const void *src, size_t bytes);   /* synchronous */
GDPT_API int  gdpt_device_download(int device, void *host, const void *dev, size_t bytes);
/* The device a scene (and every film created on it) lives on. */
GDPT_API int  gdpt_scene_device(const gdpt_scene *s);
/* Scene::getBSphere().radius (include/mitsuba/render/scene.h:972-975) of the box Scene::initializeBidirectional builds (src/librender/scene.cpp:386-413):
 * the kd-tree's bounds (enlarged by MTS_KD_AABB_EPSILON) + the sensor's AABB + every emitter's AABB -- what ManifoldPerturbation::manifoldWalk
 * measures its reversibility error against (src/libbidir/mut_manifold.cpp:1219). */
GDPT_API int  gdpt_scene_bsphere_radius(const gdpt_scene *s, double *radius);
/* The film's reconstruction filter (`<rfilter type=...>`, src/rfilters/<type>.cpp, discretised as rfilter.cpp:37-55): GDPT_RFILTER_BOX
 * (default: every put covers one pixel, the per-pixel-sums fast path), TENT, GAUSSIAN (p0 = stddev, 0.5), MITCHELL (p0 = B, p1 = C,
 * 1/3 each), CATMULLROM, LANCZOS (p0 = lobes, 3).  The wider filters log every sample and gather the puts per receiving pixel
 * (no atomics; 1.3-2.3x the box render time).  A rectangle that is the WHOLE film renders a strip: a film over a strip of rows then
 * renders the rows within the filter's reach (ceil(radius) + 1 above and below, clipped to the image) as well, so that its own rows come
 * out bit-identical to the same rows of a whole-image film and strips need no exchange (gdpt_film_stats counts those rays too).  A
 * SUB-rectangle renders a block as GPTBlockRenderer::process does (src/integrators/gpt/gpt_proc.cpp:74-91): the samples of its pixels
 * only, put within the filter's reach AROUND the rectangle (GPTWorkResult's bordered ImageBlocks, gpt_wr.cpp:31-44, clipped to the film's
 * rows); the blocks of a film add up to the whole-film render (to the rounding of the sums' order).
 * Call before rendering. */
#define GDPT_RFILTER_BOX        0
#define GDPT_RFILTER_TENT       1
#define GDPT_RFILTER_GAUSSIAN   2
#define GDPT_RFILTER_MITCHELL   3
#define GDPT_RFILTER_CATMULLROM 4
#define GDPT_RFILTER_LANCZOS    5
GDPT_API int  gdpt_film_set_rfilter(gdpt_film *f, int kind, double p0, double p1);
/* Tuning knob (no reference counterpart): into how many slices the spp samples of a launch are split (one work item = one
 * 16x16 tile x one slice).  0 (default) = chosen per launch from the rectangle, spp and the device so that small launches
 * (strips of a multi-GPU frame) still fill the chip.  Samples and their random numbers do not depend on it; the per-pixel sums
 * are folded in slice order, so results differ from slices = 1 only in the association of an fp64 sum. */
GDPT_API int  gdpt_film_set_slices(gdpt_film *f, int slices);
/* Tuning knob (no reference counterpart): how many lanes of a wave must be idle before they start new samples together
 * (1..64, default 56; 64 = only when the whole wave is idle).  Results do not depend on it. */
GDPT_API int  gdpt_film_set_regeneration(gdpt_film *f, int idleLanes);

/* Probe for the roofline note of SURVEY 8(d)-B: traversal statistics of numRays rays (origin, direction; unbounded), traced once as
 * closest-hit and once as any-hit queries: sums[0..3] = inner nodes fetched / triangles tested (closest), the same (any-hit). */
GDPT_API int  gdpt_scene_trace_stats(gdpt_scene *s, int numRays, const double *originsDirs6, unsigned long long sums[4]);
/* What the scene looks like on the device: out[0] = inner nodes of the BVH (four children each), out[1] = bytes per node (128: fp32 boxes, scenes staged
 * into LDS; 64: 8-bit boxes, scenes in HBM), out[2] = 1 if the tables are staged into LDS by the render kernels, out[3] = traversal stack entries the tree can
 * need, out[4] = bytes of the staged tables, out[5] = 1 if the render kernels read the tables from HBM (their traversal leaves its inner-node loop early, gpt_kernels.hip.h trace()). */
GDPT_API int  gdpt_scene_layout(gdpt_scene *s, long long out[6]);
/* Probe for tests: closest hit of one ray on the device -> prim (original triangle index, -1 = miss), t, p[3]. */
GDPT_API int  gdpt_scene_intersect(gdpt_scene *s, int numRays, const double *originsDirs6, int *prim, double *tp4);
/* Probe for tests: the FILLED intersection record of each ray's closest hit as the render kernels form it (ShapeKDTree::rayIntersect(ray, its),
 * src/librender/skdtree.cpp:112-142 + fillIntersectionRecord<true>, include/mitsuba/render/skdtree.h:343-428) -> prim (original triangle index, -1 = miss)
 * and rec24 = t, its.p(3), its.uv(2), geoFrame.n(3), shFrame.n(3), shFrame.s(3), dpdu(3), dpdv(3), wi(3).  This is what the reference's own
 * src/tests/test_dgeom.cpp:35-178 asserts on; tests/golden/dgeom_reference.json holds its vectors. */
GDPT_API int  gdpt_scene_intersect_record(gdpt_scene *s, int numRays, const double *originsDirs6, int *prim, double *rec24);

/* Probe for tests: one sample's raw evaluatePoint outputs (gpt.cpp:397-436): veryDirect(3), throughput(3), gradients[4](12),
 * neighbourThroughputs[4](12), then closest-hit count, any-hit count, final depth. */
GDPT_API int  gdpt_scene_evaluate_point(gdpt_scene *s, const gdpt_config *cfg, int px, int py, int sample, double out33[33]);

/* Probe for tests: the device's BSDF models on their own, for one material with its constant reflectance and one incident direction wi (local
 * frame).  For each of nSamples pairs (sx, sy): the pdf-returning BSDF::sample (diffuse.cpp:141-151, conductor.cpp:256-273, roughconductor.cpp:
 * 369-418, dielectric.cpp:277-305, twosided.cpp:148-168) -> sampled8 = wo(3), weight(3), pdf, sampledType.  For each of nDirs directions wo:
 * BSDF::eval and BSDF::pdf in `measure` (0 solid angle, 1 discrete) -> evalPdf4 = f(3), pdf.  This is what the chi-square test of the reference's
 * src/tests/test_chisquare.cpp checks a BSDF with, run on the HIP side. */
GDPT_API int  gdpt_bsdf_probe(const gdpt_material *m, const double wi[3], int nSamples, const double *samples2, double *sampled8,
                              int nDirs, const double *wo3, int measure, double *evalPdf4);

/* ---- G-BDPT (BASELINE config 5): the per-block work of the reference's `gbdpt` integrator plugin ------------------------------------------
 * What GBDPTRenderer::process -> evaluate compute for a rectangle of pixels (src/integrators/gbdpt/gbdpt_proc.cpp:86-256,259-534 over
 * src/libbidir: Path::alternatingRandomWalkFromPixel, ManifoldPerturbation::generateOffsetPathGBDPT, Path::miWeight{Base,Grad}NoSweep_GBDPT),
 * accumulated as GBDPTWorkResult / GBDPTProcess::processResult do (gbdpt_wr.h:56-62, gbdpt_proc.cpp:708-763): five camera blocks (rgb, weight)
 * and five full-resolution light images, buffer order of the integrator's MultiFilm (gbdpt.cpp:163): 0 primal, 1 gradient towards (0,-1),
 * 2 (-1,0), 3 (+1,0), 4 (0,+1).  Scope: surface scenes; area, point, constant-environment and envmap emitters (round 5); perspective or thinlens sensor
 * WITHOUT a crop window (a cropped gdpt_camera returns GDPT_ERR_UNSUPPORTED); box filter; maxDepth <= 20; BSDFs diffuse and rough conductors
 * (one- or two-sided, textured or not) and -- round 4 -- conductor, dielectric and rough conductors below shiftThreshold.  A sample whose
 * surface vertices are all connectable in the sense of Path::isConnectable_GBDPT runs the fast wavefront form; a sample that meets a
 * SPECULAR vertex runs the general form (csrc/gbdpt_general.hip.h): offset paths by ManifoldPerturbation::propagatePerturbation and
 * manifoldWalk (mut_manifold.cpp:989-1227, SpecularManifold manifold.cpp:59-757), Jacobians and MIS weights with SpecularManifold::{G, multiG,
 * det} -- as staged launches over per-sample records in HBM (round 5: DESIGN.md "the general form as staged launches").
 * Same counter-based random streams as the G-PT path, consumed in the reference's order. */
typedef struct gdpt_gbdpt_config {
    int    maxDepth;            /* -1 renders as 12 (gbdpt_proc.cpp:103-106); at most 20 (a sample record holds whole subpaths; 19 with the thinlens sensor) */
    int    rrDepth;             /* 5 (gbdpt.cpp:82)                                                                         */
    int    lightImage;          /* 1 (gbdpt.cpp:83): connect emitter subpaths to the sensor (t = 1 strategies)             */
    int    spp;
    double shiftThreshold;      /* 0.001 (gbdpt.cpp:85)                                                                     */
    unsigned long long seed;
} gdpt_gbdpt_config;
typedef struct gdpt_gbdpt_film gdpt_gbdpt_film;

GDPT_API int   gdpt_gbdpt_film_create(gdpt_scene *s, gdpt_gbdpt_film **out);       /* GBDPTWorkResult for the whole crop window + GBDPTProcess::m_result */
GDPT_API void  gdpt_gbdpt_film_destroy(gdpt_gbdpt_film *f);
GDPT_API int   gdpt_gbdpt_film_clear(gdpt_gbdpt_film *f);
/* GBDPTRenderer::process over the pixels [x0,x1) x [y0,y1), all cfg->spp samples each; asynchronous on the film's stream */
GDPT_API int   gdpt_gbdpt_render_rect(gdpt_scene *s, const gdpt_gbdpt_config *cfg, int x0, int y0, int x1, int y1, gdpt_gbdpt_film *f);
GDPT_API int   gdpt_gbdpt_film_sync(gdpt_gbdpt_film *f);
GDPT_API float gdpt_gbdpt_film_render_ms(gdpt_gbdpt_film *f);                      /* HIP-event time of the render kernels since the last clear */
GDPT_API void *gdpt_gbdpt_film_stream(gdpt_gbdpt_film *f);
/* raw sums: block[5][H][W][4] (r, g, b, weight), light[5][H][W][3]; HOST pointers */
GDPT_API int   gdpt_gbdpt_film_accum(gdpt_gbdpt_film *f, double *block, double *light);
/* GBDPTProcess::develop + MultiFilm::developMulti (gbdpt_proc.cpp:694-706, multifilm.cpp:317-362): (block + light * weight / spp) / weight
 * as H x W x 3 doubles -- what GBDPTIntegrator::render reads back before prepareDataForSolver (gbdpt.cpp:199-207) */
GDPT_API int   gdpt_gbdpt_film_develop_device(gdpt_gbdpt_film *f, int buffer, int spp, double *rgbDevice);
GDPT_API int   gdpt_gbdpt_film_develop(gdpt_gbdpt_film *f, int buffer, int spp, double *rgbHost);
/* the raw sums copied device-to-device out of / into the film (layouts of gdpt_gbdpt_film_accum; the call returns with the copy complete): a
 * multi-GPU host adds the films of its ranks -- every rank's light images hold splats for the WHOLE image (GBDPTWorkResult::put, gbdpt_wr.cpp:57-63) */
GDPT_API int   gdpt_gbdpt_film_export_device(gdpt_gbdpt_film *f, double *blockDevice, double *lightDevice);
GDPT_API int   gdpt_gbdpt_film_import_device(gdpt_gbdpt_film *f, const double *blockDevice, const double *lightDevice);
GDPT_API int   gdpt_gbdpt_film_stats(gdpt_gbdpt_film *f, unsigned long long stats[4]);   /* closest-hit rays, shadow rays, samples, puts dropped as invalid */
/* probe: ONE sample of GBDPTRenderer::process: out17 = primal(3), gradients(4 x 3), film position(2); its light-image splats as rows
 * (x, y, buffer, r, g, b), at most maxLight of them (*nLight = how many there were); counters = closest-hit / shadow rays */
GDPT_API int   gdpt_gbdpt_evaluate_sample(gdpt_scene *s, const gdpt_gbdpt_config *cfg, int px, int py, int sample, double out17[17],
                                          int maxLight, double *light6, int *nLight, unsigned long long counters[2]);
/* the same with counters[4] = closest-hit rays, shadow rays, 1 if the sample ran in the general form (it met a specular vertex: conductor,
 * dielectric, a rough conductor below shiftThreshold -- offset paths by propagatePerturbation + manifoldWalk, mut_manifold.cpp:989-1227), and the
 * number of times that form's per-sample workspace ran out (0: the result is complete) */
GDPT_API int   gdpt_gbdpt_evaluate_sample2(gdpt_scene *s, const gdpt_gbdpt_config *cfg, int px, int py, int sample, double out17[17],
                                           int maxLight, double *light6, int *nLight, unsigned long long counters[4]);
/* since the last clear: stats[0] samples that ran in the general form (specular chains), stats[1] workspace overflows among them (0 = none) */
GDPT_API int   gdpt_gbdpt_film_chain_stats(gdpt_gbdpt_film *f, unsigned long long stats[2]);

#ifdef __cplusplus
}
#endif
#endif /* GDPT_TRACER_H */
