"""Row-strip sharding of the G-PT hot path over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference shards by 32x32 image blocks over CPU worker threads and merges block borders by addition
(/root/reference/src/librender/imageproc.cpp:28-78, gpt_proc.cpp:52-56,137-149).  Here each rank owns a contiguous strip of
rows; the only coupling is the one-pixel halo: a strip needs the per-pixel sample sums of the row just above and below it
(and passes on the exact-put spill it produced for its neighbours' rows).  That is two point-to-point messages per rank
(<= 2 xGMI links, no collective on the data path).  Reconstruction runs on rank 0 after a gather of the four developed
fp32 images (44 MB at 1280x720; the CG at this size is latency-bound and does not profit from splitting -- DESIGN.md).
`StripRenderer` below is the entry point: one frame = render strip, settle borders, develop, gather, reconstruct on rank 0.
"""
import os
import time

import torch
import torch.distributed as dist


def row_strips(height, world):
    """Contiguous row ranges [y0, y1) per rank; earlier ranks take the remainder."""
    base, rem = divmod(height, world)
    out, y = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((y, y + n))
        y += n
    return out


def rebalance_strips(strips, times, min_rows=1):
    """New contiguous strips [y0, y1) with (approximately) equal render time, from the times the current strips took.
    The reference balances by handing 32x32 blocks to whichever worker is free (sched.cpp:427-496); with one strip per GPU
    the analogue is to move the strip boundaries: cost per row is taken as constant inside each old strip, the new
    boundaries cut the cumulative cost at k/world.  Deterministic (same inputs on every rank -> same partition)."""
    world = len(strips)
    height = strips[-1][1]
    if world == 1 or min(times) <= 0.0:
        return list(strips)
    cum = [0.0]                                   # cumulative cost at row boundaries 0..height
    for (y0, y1), t in zip(strips, times):
        per_row = float(t) / (y1 - y0)
        for _ in range(y0, y1):
            cum.append(cum[-1] + per_row)
    total = cum[-1]
    cuts, y = [0], 0
    for k in range(1, world):
        target = total * k / world
        while y < height and cum[y + 1] <= target:
            y += 1
        # the boundary nearer to the target
        if y < height and (target - cum[y]) > (cum[y + 1] - target):
            y += 1
        y = max(y, cuts[-1] + min_rows)
        y = min(y, height - (world - k) * min_rows)
        cuts.append(y)
    cuts.append(height)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _wire(t):
    """gloo (CPU tests, single-GPU functional runs) cannot move device tensors point-to-point: stage through the host."""
    return t.cpu() if (t.is_cuda and dist.get_backend() == "gloo") else t


def _settle(t):
    """The HIP library launches on its own streams; torch (RCCL receives, copies, cat) on torch's current stream.  Before a
    buffer torch produced is handed to the library, that stream has to be finished (req.wait() only orders the STREAM
    after an RCCL operation, not the host)."""
    if t is not None and t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()


def exchange_halos(film, rank, world, device, group=None):
    """film: object with halo_bytes(), pack_halo(which, tensor), unpack_halo(which, tensor) (gpt.Film or a test double).
    which = 0 talks to rank-1 (the strip above), which = 1 to rank+1 (below).  Returns bytes sent."""
    if world == 1:
        return 0
    n = film.halo_bytes() // 8
    ops, recv, sent = [], {}, 0
    for which, peer in ((0, rank - 1), (1, rank + 1)):
        if peer < 0 or peer >= world:
            continue
        out = torch.empty(n, dtype=torch.float64, device=device)
        film.pack_halo(which, out)
        recv[which] = _wire(torch.empty(n, dtype=torch.float64, device=device))
        ops.append(dist.P2POp(dist.isend, _wire(out), peer, group=group))
        ops.append(dist.P2POp(dist.irecv, recv[which], peer, group=group))
        sent += out.numel() * 8
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for which, buf in recv.items():
        buf = buf.to(device)
        _settle(buf)
        film.unpack_halo(which, buf)
    return sent


def gather_post(shape_lead, dtype, device, strips, width, rank, world, group=None):
    """Rank 0 posts the receives of a gather (one message per rank) and returns (parts, requests) for gather_finish; other ranks get None.
    On a communicator of its own (StripRenderer.gather_group) this may happen BEFORE rank 0 renders: between one pair of ranks RCCL matches
    point-to-point messages in posting order, so on the halo's communicator an early gather receive from rank 1 would take rank 1's halo."""
    if world == 1 or rank != 0:
        return None
    parts = [_wire(torch.empty(tuple(shape_lead) + (y1 - y0, width, 3), dtype=dtype, device=device)) for (y0, y1) in strips]
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, parts[r], r, group=group) for r in range(1, world)])
    return parts, reqs


def gather_finish(strip, posted, rank, world, group=None):
    """Completes a gather: rank 0 waits for the posted receives and returns the assembled rows, the others send their strip (None)."""
    if world == 1:
        return strip
    stacked = strip.dim() == 4
    if rank == 0:
        parts, reqs = posted
        for q in reqs:
            q.wait()
        parts[0] = strip
        full = torch.cat([p.to(strip.device) for p in parts], dim=1 if stacked else 0)
        _settle(full)
        return full
    for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, _wire(strip.contiguous()), 0, group=group)]):
        q.wait()
    _settle(strip)          # the library reuses the strip buffer on its own stream next step
    return None


def gather_rows(strip, strips, width, rank, world, group=None):
    """Gather per-rank strips to rank 0 (None elsewhere).  strip: [rows_r, width, 3] -> [H, width, 3], or a stack of images
    [k, rows_r, width, 3] -> [k, H, width, 3] (one message per rank for all k images instead of k)."""
    if world == 1:
        return strip
    lead = (strip.shape[0],) if strip.dim() == 4 else ()
    return gather_finish(strip, gather_post(lead, strip.dtype, strip.device, strips, width, rank, world, group), rank, world, group)


class StripRenderer:
    """GradientPathIntegrator::render (/root/reference/src/integrators/gpt/gpt.cpp:1358-1480) over the ranks of a process group:
    every rank renders its strip (GPTBlockRenderer::process), strips settle their borders, the four developed solver images
    travel to rank 0 as one message per rank, and rank 0 reconstructs (`poisson::Solver`).

    Borders: with the box filter a sample touches only its own pixel and the four neighbours (the rare sample within 1e-5 of a pixel edge: two pixels and their
    neighbours -- the payload carries the exact puts of two rows for those), so the halo sums are
    exchanged and added (gpt_proc.cpp:52-56,137-149).  With a wider reconstruction filter (Mitsuba's film default is gaussian) a
    strip's film renders the rows within the filter's reach itself (`gdpt_film_set_rfilter`), so nothing is exchanged and the
    result is bit-identical to a one-GPU render; the price is 2 x reach redundant rows per strip.

    scene, integ: gpt.Scene / gpt.GradientPathIntegrator (or test doubles with the same methods); film_factory(scene, y0, y1)
    and solver_factory(preset, alpha) default to gpt.Film and poisson.Solver."""

    def __init__(self, scene, integ, rank, world, device, group=None, strips=None, film_factory=None, solver_factory=None):
        self.scene, self.integ, self.rank, self.world, self.device, self.group = scene, integ, rank, world, device, group
        self.width, self.height = scene.width, scene.height
        if film_factory is None:
            from . import gpt
            film_factory = gpt.Film
        if solver_factory is None:
            from . import poisson

            def solver_factory(preset, alpha):
                return poisson.Solver(poisson.Params(preset, alpha))
        self._film_factory = film_factory
        self.preset = "L1D" if getattr(integ, "reconstructL1", False) else ("L2D" if getattr(integ, "reconstructL2", False) else None)
        self.solver = solver_factory(self.preset, integ.reconstructAlpha) if (rank == 0 and self.preset) else None
        self.film = None
        # The gather's receives: by default rank 0 posts them AFTER the halo exchange, on the halo's own communicator (between one pair of ranks RCCL matches
        # point-to-point messages in posting order, so nothing can be mistaken; one communicator, nothing pending while a strip renders).
        # GDPT_EARLY_GATHER=1 (opt-in, round 5's form): a second communicator on which rank 0 posts them BEFORE it renders, so that they complete while it is
        # still busy with its own strip.  Not the default since round 6: two communicators in concurrent use from one process are not deadlock-safe by
        # RCCL's own documentation (rank 0's pending receive kernel can occupy the device while the halo exchange needs its own to run), the pending receive
        # lives for a whole render -- longer than the process group's watchdog timeout for a long frame unless that is raised -- and it has only ever run
        # over gloo on one device (tests/test_parallel_cpu.py, tests/test_bench_gpu.py): unverified on real multi-GPU RCCL.  Measured worth on one
        # device over gloo: ~1 ms of a 4K frame's gather set-up.
        want_early = os.environ.get("GDPT_EARLY_GATHER", "0") == "1"
        self.gather_group = dist.new_group(ranks=list(range(world))) if (want_early and world > 1 and dist.is_initialized() and group is None) else group
        self.early_gather = world > 1 and want_early and self.gather_group is not group   # (a caller's own group: no second communicator)
        self.set_strips(strips or row_strips(self.height, world))
        self.last = {}
        self._open_links()

    def _open_links(self):
        """RCCL sets a peer-to-peer channel up on its first use (tens of milliseconds each): touch every link the frame loop
        uses -- strip neighbours and every rank -> rank 0 -- once, here, so that the first frame does not pay for it."""
        if self.world == 1:
            return
        one = torch.zeros(1, dtype=torch.float32, device=self.device)
        ops, keep = [], []
        for peer in (self.rank - 1, self.rank + 1):
            if 0 <= peer < self.world:
                keep.append(_wire(torch.empty_like(one)))
                ops.append(dist.P2POp(dist.isend, _wire(one), peer, group=self.group))
                ops.append(dist.P2POp(dist.irecv, keep[-1], peer, group=self.group))
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        gather_rows(one.view(1, 1, 1).expand(1, 1, 3).contiguous(), [(r, r + 1) for r in range(self.world)], 1, self.rank, self.world, self.gather_group)

    def set_strips(self, strips):
        """(Re)partition the image; every rank must pass the same list."""
        assert len(strips) == self.world and strips[0][0] == 0 and strips[-1][1] == self.height
        # (the halo carries the exact puts of TWO rows beyond a boundary -- a sample within 1e-5 of a pixel edge reaches that far, include/gdpt_tracer.h -- and the
        #  neighbour adds the far one to the row INSIDE its boundary row: a one-row strip has none, the put would be lost)
        if self.world > 1 and min(b - a for a, b in strips) < 2:
            raise ValueError("strips of the halo exchange need at least two rows each (%d rows over %d ranks: %r)" % (self.height, self.world, list(strips)))
        self.strips = list(strips)
        self.y0, self.y1 = self.strips[self.rank]
        if self.film is not None:
            self.film.close()
        self.film = self._film_factory(self.scene, self.y0, self.y1)
        # four solver images; without a reconstruction a fifth: buffer 0, the (8 veryDirect + 2 throughput + neighbours) / weight
        # preview that the reference leaves in -final (gpt.cpp:1314-1322)
        self.strip_imgs = torch.empty((4 if self.preset else 5, self.y1 - self.y0, self.width, 3), dtype=torch.float32, device=self.device)
        self.rec = torch.empty((self.height, self.width, 3), dtype=torch.float32, device=self.device) if self.rank == 0 else None

    def rebalance(self, min_rows=2):
        """Move the strip boundaries by the render-kernel times of the last render (collective).  Returns True if they moved."""
        if self.world == 1:
            return False
        # rank 0 also reconstructs while the others already render the next frame: its strip is charged with the solve
        t = torch.tensor([self.last.get("render_ms", 0.0) + 1e3 * self.last.get("solve_s", 0.0)], dtype=torch.float64, device=self.device)
        allt = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(allt, t, group=self.group)
        new = rebalance_strips(self.strips, [float(v.item()) for v in allt], min_rows=min_rows)
        if new == self.strips:
            return False
        self.set_strips(new)
        return True

    def render(self, spp, seed=5489):
        """One frame.  Rank 0 returns the reconstruction [H, W, 3] (device tensor; if the integrator does not reconstruct, the -final
        preview of gpt.cpp:1314-1322 = what gpt.GradientPathIntegrator.render()['-final'] holds in that mode) -- other ranks None.  self.last: rays, render_ms, solve_s, halo_bytes of this rank; on rank 0 also the four
        gathered solver images under "images" ([4, H, W, 3]: throughput, dx, dy, direct; [5, H, W, 3] with the -final preview as a fifth
        when the integrator does not reconstruct)."""
        film, integ = self.film, self.integ
        cfg = integ.config(spp, seed)
        tick = [time.perf_counter()]
        phases = {}

        def lap(name):                                   # host wall time of a phase (each ends synchronised)
            now = time.perf_counter()
            phases[name] = 1e3 * (now - tick[0])
            tick[0] = now
        film.clear()
        post = lambda: gather_post((self.strip_imgs.shape[0],), self.strip_imgs.dtype, self.device, self.strips, self.width, self.rank, self.world, self.gather_group)
        posted = post() if self.early_gather else None
        integ.renderBlock(self.scene, film, cfg, (0, self.y0, self.width, self.y1))       # GPTBlockRenderer::process over the strip
        film.sync()
        lap("render")
        halo = 0
        if not getattr(film, "renders_own_border", False):
            halo = exchange_halos(film, self.rank, self.world, self.device, self.group)
        lap("halo")
        for i, b in enumerate((1, 2, 3, 4) if self.preset else (1, 2, 3, 4, 0)):          # BUFFER_THROUGHPUT, DX, DY, VERY_DIRECT (, BUFFER_FINAL)
            film.develop_device(b, self.strip_imgs[i])                                     # developMulti + float cast, gpt.cpp:1419-1442
        lap("develop")
        full = gather_finish(self.strip_imgs, posted if self.early_gather else post(), self.rank, self.world, self.gather_group)
        lap("gather")
        solve_s = 0.0
        out = None
        if self.rank == 0:
            if self.solver is not None:
                self.solver.importImagesMTS(full[1], full[2], full[0], full[3], self.width, self.height)   # dx, dy, throughput, direct
                self.solver.setupBackend()
                self.solver.solveIndirect()
                self.solver.exportImagesMTS(self.rec)
                solve_s = self.solver.lastSolveSeconds
                out = self.rec
            else:
                out = full[4]                                                              # no reconstruction: -final keeps the preview
        lap("reconstruct")
        st = film.stats()
        self.last = dict(phases_ms=phases, rays=st["raysTraced"] + st["shadowRaysTraced"], render_ms=film.render_ms(), solve_s=solve_s, halo_bytes=halo,
                         images=full if self.rank == 0 else None)
        return out

    def close(self):
        if self.solver is not None:
            self.solver.close()
            self.solver = None
        if self.film is not None:
            self.film.close()
            self.film = None


class GBDPTStripRenderer:
    """GBDPTIntegrator::render (/root/reference/src/integrators/gbdpt/gbdpt.cpp:140-262) over the ranks of a process group.

    The reference hands 32x32 blocks to workers; every worker's result carries camera blocks for its pixels AND five full-resolution light
    images (light-tracing connections land on any pixel, gbdpt_wr.cpp:45-52), and results are merged by addition (GBDPTWorkResult::put,
    :57-63).  Here: rank r renders the samples of a contiguous strip of rows into a film of its own (whole-image buffers), the films are SUMMED
    onto rank 0 -- ONE reduction over RCCL (`dist.reduce` of a single buffer, 258 MB of fp64 sums at 1280x720: the path's one real exchange step; a one-pixel halo
    would not do, a light sample of any rank can hit any pixel) -- and rank 0 develops and runs both reconstructions.

    scene, integ: gpt.Scene / gbdpt.GBDPTIntegrator (or test doubles); film_factory(scene) defaults to gbdpt.Film."""

    def __init__(self, scene, integ, rank, world, device, group=None, strips=None, film_factory=None, reconstruct=None):
        self.scene, self.integ, self.rank, self.world, self.device, self.group = scene, integ, rank, world, device, group
        self.width, self.height = scene.width, scene.height
        if film_factory is None:
            from . import gbdpt
            film_factory = gbdpt.Film
        if reconstruct is None:
            from . import poisson

            def reconstruct(bufs, w, h, alpha):
                return poisson.gbdpt_reconstruct_device(bufs, w, h, alpha=alpha)
        self._reconstruct = reconstruct
        self.film = film_factory(scene)
        self.strips = list(strips or row_strips(self.height, world))
        self.y0, self.y1 = self.strips[rank]
        # the camera blocks and the light images of a film in ONE buffer: one reduction per frame (a collective's set-up costs more than 100 MB more payload)
        n4, n3 = 5 * self.height * self.width * 4, 5 * self.height * self.width * 3
        self.sums = torch.empty(n4 + n3, dtype=torch.float64, device=device)
        self.block = self.sums[:n4].view(5, self.height, self.width, 4)
        self.light = self.sums[n4:].view(5, self.height, self.width, 3)
        self.last = {}

    def render(self, spp, seed=5489):
        """One frame.  Rank 0 returns dict("-primal" ... "-gradientPosY": float64 [H, W, 3], "-L2", "-L1": float32), other ranks None."""
        film, integ = self.film, self.integ
        cfg = integ.config(spp, seed)
        t0 = time.perf_counter()
        film.clear()
        integ.renderBlock(self.scene, film, cfg, (0, self.y0, self.width, self.y1))
        film.sync()
        t1 = time.perf_counter()
        reduce_bytes = 0
        if self.world > 1:
            film.export_device(self.block, self.light)
            ws = _wire(self.sums)
            dist.reduce(ws, 0, op=dist.ReduceOp.SUM, group=self.group)
            reduce_bytes = ws.numel() * 8
            if self.rank == 0:
                if ws is not self.sums:
                    self.sums.copy_(ws)
                _settle(self.sums)
                film.import_device(self.block, self.light)
        t2 = time.perf_counter()
        out = None
        solve_s = (0.0, 0.0)
        if self.rank == 0:
            from .gbdpt import SAMPLER_BUFFERS
            bufs = [torch.empty((self.height, self.width, 3), dtype=torch.float64, device=self.device) for _ in SAMPLER_BUFFERS]
            for i, b in enumerate(bufs):
                film.develop_device(i, spp, b)
            film.sync()
            l2, l1, solve_s = self._reconstruct(bufs, self.width, self.height, integ.reconstructAlpha)
            out = dict(zip(SAMPLER_BUFFERS, bufs))
            out["-L2"], out["-L1"] = l2, l1
        t3 = time.perf_counter()
        st = film.stats()
        self.last = dict(phases_ms=dict(render=1e3 * (t1 - t0), reduce=1e3 * (t2 - t1), develop_reconstruct=1e3 * (t3 - t2)),
                         rays=st["raysTraced"] + st["shadowRaysTraced"], closest_rays=st["raysTraced"], samples=st["samples"], render_ms=film.render_ms(), solve_s=solve_s, reduce_bytes=reduce_bytes,
                         chain=film.chain_stats() if hasattr(film, "chain_stats") else dict(generalSamples=0, overflows=0))
        return out

    def close(self):
        if self.film is not None:
            self.film.close()
            self.film = None
