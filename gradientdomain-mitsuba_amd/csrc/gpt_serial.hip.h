// gpt_serial.hip.h -- the serial form of the G-PT sampler: the whole film by ONE lane that consumes ONE SFMT-19937 stream in the order a one-worker run
// of the reference does (`mitsuba -p 1`).  A validation path, not a fast one: it is what lets the HIP sampler be held against the reference's OWN random
// stream and block order instead of against per-sample streams only.  gpt_capi.hip owns the film; the kernel, the generator
// and the pixel order live in gpt_serial_capi.hip -- a translation unit of its own because it builds the sampler's device functions with another `Rng`
// (gpt_kernels.hip.h, GDPT_SERIAL_STREAM).
#pragma once
#include "gpt_kernels.hip.h"
#include <hip/hip_runtime.h>

struct gdpt_scene;

namespace gdpt_tr {

// Renders every pixel of the film `fd` (rows y0 .. y1 must be the whole image) with cfg.spp samples each, serially; the sums go to the film's pixel records as
// the single-kernel pipeline's do.  blockSize: Scene::getBlockSize() (`-b`, 32); parentSeed: the seed of the scene sampler's Random (5489, random.h:113).
// draws (optional): random numbers consumed.  Asynchronous on `stream` except for the upload of the pixel order and the generator's state.
int serial_render(const gdpt_scene *s, hipStream_t stream, const ConfigD &cfg, const FilmD &fd, int blockSize, unsigned long long parentSeed, unsigned long long *draws);

} // namespace gdpt_tr
