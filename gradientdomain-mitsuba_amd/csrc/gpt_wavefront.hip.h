// gpt_wavefront.hip.h -- the wavefront form of the G-PT sampler's bounces: rays (not paths) travel through HBM queues to traversal-only
// kernels that run at full occupancy, and the shading stages replay bounce() (gpt_render.hip.h) around them.
//
// Why: the megakernels (k_render / k_continue) carry a traversal inlined under 128-256 registers of fp64 path state; on an HBM-resident
// scene (the 113 k-triangle atrium) they sit at ~2 Gray/s with 22 % lane utilisation and 76-80 % of their wave cycles waiting, while a
// traversal-only kernel walks the same tree at 4-10 Gray/s (k_primary / k_intersect).  Here a bounce is split at its ray sites
// (the tracer policy of bounce(), gpt_render.hip.h):
//     k_wf_cont   per sample of the list: [replay the bounce whose rays were just traced, with their results: the only pass that keeps
//                 what it computes] -> [run the next bounce as far as its rays, write them to the ray queues] -> sample back on the list
//     k_wf_trace  per ray of a queue: traversal only (closest hit / any hit), result to the slot the ray came from
// Both replays are instantiations of the ONE bounce() the megakernels run, so films and ray counts are bit-identical by construction
// (tests/test_gpt_gpu.py holds the pipelines against each other and the oracle).  The price of the replay is arithmetic (the emitter and
// BSDF samples of a bounce are computed twice); what it buys is that nothing but the path record itself travels between the passes.
//
// This file: the queue descriptor and the launcher's interface, shared by gpt_capi.hip (which owns the film and its queues) and
// gpt_wave_capi.hip (which holds the kernels: its own translation unit, so that it compiles beside the megakernels).
#pragma once
#include "gpt_kernels.hip.h"
#include <hip/hip_runtime.h>

struct gdpt_scene;

namespace gdpt_tr {

// The film's side of the wavefront pipeline is this interface only (the queues' layout is gpt_wave_capi.hip's business, so that the megakernels'
// translation unit does not recompile when it changes).
struct WfQueues;
WfQueues *wf_create();
void wf_destroy(WfQueues *q);                       // frees the device memory too
size_t wf_bytes_per_slot();                         // device memory the queues need per sample slot of a chunk
bool wf_reserve(WfQueues *q, size_t slots);         // (re)allocates for `slots` sample slots; false: out of memory (nothing is held then)
void wf_release(WfQueues *q);                       // frees the device memory, keeps the handle
size_t wf_slots(const WfQueues *q);                 // slots the queues are allocated for
int wf_max_iters();                                 // most traced bounces wf_continue takes
bool wf_failed(const WfQueues *q);                  // a ray found its queue full since the queues were allocated (call after synchronising): the films rendered since are void
// Start of a chunk: zeroes the chunk's counters (asynchronous on `stream`) and points the descriptors at the sample lists: fdRender (the
// kernel that hands samples over: k_render<STAGED>) appends to the first list, fdContinue (the kernel that takes over what is left after
// `iters` traced bounces: k_continue) reads the last one.  Both descriptors must already hold the chunk's queue geometry (qCapacity).
int wf_begin_chunk(WfQueues *q, hipStream_t stream, int iters, FilmD &fdRender, FilmD &fdContinue);
// Runs `iters` traced bounces of the continuation phase on the samples on the first list (all offsets connected or dead: the state of a
// sample is its continuation record, FilmD::qRec): iters + 1 shading passes around `iters` pairs of trace launches.  Afterwards the samples
// still alive are on the last list with their records up to date.  Asynchronous on `stream`.
int wf_continue(const gdpt_scene *s, hipStream_t stream, const ConfigD &cfg, const FilmD &fd, WfQueues *q, int iters, int stackDepth, int sceneBytes);

} // namespace gdpt_tr
