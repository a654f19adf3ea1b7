"""CPU (gloo, world_size 2 and 3) tests of the row-strip halo protocol of gradientdomain-mitsuba_amd/parallel.py.
The film here is a numpy double of gpt.Film's pack/unpack contract (csrc/gpt_render.hip.h k_pack_halo/k_unpack_halo);
the same protocol against real device films is tested on the GPU in tests/test_gpt_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gradientdomain_mitsuba_amd import parallel

NREC = 31


class FakeFilm:
    """rec[NREC][rows+2][W] (row 0 / -1 = halo), spill[5][rows+4][W][4] (rows 0-1 / -2 -1 = the two halo rows of exact puts beyond the strip: a sample within 1e-5
    of a pixel edge reaches two rows, round 5); same payload layout as the device film: the boundary row's records, then the spill of the near and the far halo row."""

    def __init__(self, W, y0, y1, seed):
        rng = np.random.default_rng(seed)
        self.W, self.y0, self.y1 = W, y0, y1
        rows = y1 - y0
        self.rec = rng.standard_normal((NREC, rows + 2, W))
        self.rec[:, 0] = 0; self.rec[:, -1] = 0
        self.spill = rng.standard_normal((5, rows + 4, W, 4))

    def halo_bytes(self):
        return 8 * (NREC * self.W + 2 * 5 * self.W * 4)

    def pack_halo(self, which, t):
        own, near, far = (1, 1, 0) if which == 0 else (-2, -2, -1)
        t.copy_(torch.from_numpy(np.concatenate([self.rec[:, own].ravel(), self.spill[:, near].ravel(), self.spill[:, far].ravel()])))

    def unpack_halo(self, which, t):
        halo, own, inside = (0, 2, 3) if which == 0 else (-1, -3, -4)
        a = t.numpy()
        n1, n2 = NREC * self.W, 5 * self.W * 4
        self.rec[:, halo] = a[:n1].reshape(NREC, self.W)
        self.spill[:, own] += a[n1:n1 + n2].reshape(5, self.W, 4)                # the neighbour's near halo row is my boundary row,
        if self.y1 - self.y0 >= 2:
            self.spill[:, inside] += a[n1 + n2:].reshape(5, self.W, 4)           # its far halo row the one inside it


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    strips = parallel.row_strips(H, world)
    y0, y1 = strips[rank]
    film = FakeFilm(W, y0, y1, seed=100 + rank)
    before_spill = film.spill.copy()
    sent = parallel.exchange_halos(film, rank, world, torch.device("cpu"))
    strip = torch.from_numpy(np.full((y1 - y0, W, 3), float(rank), np.float32))
    img = parallel.gather_rows(strip, strips, W, rank, world)
    stack = torch.stack([strip + 10.0 * k for k in range(4)])                  # the four solver images of a strip in one message
    img4 = parallel.gather_rows(stack, strips, W, rank, world)
    if rank == 0:
        assert img4.shape == (4, H, W, 3) and all(torch.equal(img4[k], img + 10.0 * k) for k in range(4))
    else:
        assert img4 is None
    dsp = film.spill - before_spill
    q.put((rank, film.rec[:, 0].copy(), film.rec[:, -1].copy(), film.rec[:, 1].copy(), film.rec[:, -2].copy(),
           dsp[:, 2:4].copy(), dsp[:, -4:-2][:, ::-1].copy(), before_spill[:, 0:2][:, ::-1].copy(), before_spill[:, -2:].copy(),     # (boundary row first, then the next one)
           sent, None if img is None else img.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_and_gather_over_gloo(world):
    W, H = 24, 17
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    strips = parallel.row_strips(H, world)
    assert strips[0][0] == 0 and strips[-1][1] == H and all(a[1] == b[0] for a, b in zip(strips, strips[1:]))
    for r in range(world):
        _, halo_top, halo_bot, own_top, own_bot, dsp_top, dsp_bot, sp_halo_top, sp_halo_bot, sent, img = res[r]
        if r > 0:      # my top halo row == upper neighbour's last owned row; the spill of its two bottom halo rows (near, far) was added to my first and second row
            assert np.array_equal(halo_top, res[r - 1][4]) and np.allclose(dsp_top, res[r - 1][8])
        else:
            assert not halo_top.any() and not dsp_top.any()
        if r < world - 1:
            assert np.array_equal(halo_bot, res[r + 1][3]) and np.allclose(dsp_bot, res[r + 1][7])
        else:
            assert not halo_bot.any() and not dsp_bot.any()
        assert sent == 8 * (NREC * W + 40 * W) * ((r > 0) + (r < world - 1))
        if r == 0:
            assert img.shape == (H, W, 3)
            for rr, (y0, y1) in enumerate(strips):
                assert (img[y0:y1] == rr).all()
        else:
            assert img is None


def test_row_strips_cover_the_image():
    for H in (1, 7, 720, 1080):
        for world in (1, 2, 3, 8):
            if world > H:
                continue
            s = parallel.row_strips(H, world)
            assert s[0][0] == 0 and s[-1][1] == H and max(b - a for a, b in s) - min(b - a for a, b in s) <= 1


def test_rebalance_strips_equalises_cost():
    """Boundaries move towards equal render time; partitions stay contiguous, complete and deterministic."""
    import numpy as np
    from gradientdomain_mitsuba_amd import parallel as PP
    strips = PP.row_strips(720, 8)
    times = [36.5, 37.0, 38.0, 39.0, 39.8, 40.5, 41.0, 41.3]
    new = PP.rebalance_strips(strips, times, min_rows=2)
    assert new[0][0] == 0 and new[-1][1] == 720 and all(a[1] == b[0] for a, b in zip(new, new[1:]))
    cost = np.concatenate([np.full(b - a, t / (b - a)) for (a, b), t in zip(strips, times)])
    pred = [cost[a:b].sum() for a, b in new]
    assert max(pred) - min(pred) < 0.6 and max(pred) < max(times) - 1.5
    assert PP.rebalance_strips(strips, times, min_rows=2) == new
    assert PP.rebalance_strips([(0, 4), (4, 8)], [1.0, 3.0]) == [(0, 5), (5, 8)]
    assert PP.rebalance_strips([(0, 1), (1, 2)], [1.0, 9.0]) == [(0, 1), (1, 2)]          # nothing to move
    assert PP.rebalance_strips([(0, 10)], [5.0]) == [(0, 10)]
    assert PP.rebalance_strips(PP.row_strips(16, 4), [0.0, 1.0, 1.0, 1.0]) == PP.row_strips(16, 4)   # no timing yet: unchanged


# ---- parallel.StripRenderer over gloo with doubles of the film / integrator / solver -------------------------------------
class _ToyScene:
    def __init__(self, W, H):
        self.width, self.height = W, H


class _ToyFilm:
    """Integer-valued double of gpt.Film: per-pixel sums that the developed image reads from the row above and below (the one-pixel
    halo of rec) and splats that land on neighbouring rows (spill), so a strip is only right after the halo exchange."""
    own_border = False

    def __init__(self, scene, y0, y1):
        self.W, self.H, self.y0, self.y1 = scene.width, scene.height, y0, y1
        self.renders_own_border = self.own_border
        self.clear()
        self.exchanged = 0

    def clear(self):
        rows = self.y1 - self.y0
        self.rec = np.zeros((NREC, rows + 2, self.W)); self.spill = np.zeros((5, rows + 4, self.W, 4))     # (one halo row of records, two of exact puts: FakeFilm)

    def render(self, spp):
        lo, hi = (max(0, self.y0 - 2), min(self.H, self.y1 + 2)) if self.own_border else (self.y0, self.y1)     # (own border: every row whose puts reach the strip)
        for y in range(lo, hi):
            x = np.arange(self.W)
            if self.y0 <= y < self.y1 or self.own_border:
                if self.y0 - 1 <= y <= self.y1:
                    for k in range(NREC):
                        self.rec[k, y - (self.y0 - 1)] = spp * ((k + 1) * 1000 + 7 * y + x)
            for t in (y - 2, y - 1, y, y + 1, y + 2):                                  # (a put reaches two rows from its sample's: the box filter's 0.5 + 1e-5 radius)
                if 0 <= t < self.H and self.y0 - 2 <= t <= self.y1 + 1 and (not self.own_border or self.y0 <= t < self.y1):
                    for b in range(5):
                        self.spill[b, t - (self.y0 - 2), :, :] += (b + 1) * (3 * y + t) + x[:, None]

    def sync(self):
        pass

    def halo_bytes(self):
        self.exchanged += 1
        return 8 * (NREC * self.W + 2 * 5 * self.W * 4)

    pack_halo = FakeFilm.pack_halo
    unpack_halo = FakeFilm.unpack_halo

    def develop_device(self, b, t):
        r = self.rec[b]
        img = r[1:-1] + 2 * r[:-2] + 3 * r[2:]
        t.copy_(torch.from_numpy((img[:, :, None] + self.spill[b, 2:-2, :, :3]).astype(np.float32)))

    def stats(self):
        return dict(raysTraced=10 * (self.y1 - self.y0), shadowRaysTraced=self.y1 - self.y0)

    def render_ms(self):
        return float((self.y1 - self.y0) * (1 + self.y0))        # later rows cost more: rebalance must move the boundary up

    def close(self):
        pass


class _ToyFilmOwnBorder(_ToyFilm):
    own_border = True


class _ToyIntegrator:
    reconstructL1, reconstructL2, reconstructAlpha = False, True, 0.2

    def config(self, spp, seed=5489):
        return spp

    def renderBlock(self, scene, film, cfg, rect):
        assert rect == (0, film.y0, scene.width, film.y1)
        film.render(cfg)


class _ToySolver:
    lastSolveSeconds = 1e-4

    def importImagesMTS(self, dx, dy, tp, direct, w, h):
        self.v = tp + 2 * dx - dy + 4 * direct

    def setupBackend(self):
        pass

    def solveIndirect(self):
        pass

    def exportImagesMTS(self, rec):
        rec.copy_(self.v)

    def close(self):
        pass


def _toy_render(rank, world, W, H, film_cls, rebalance):
    sr = parallel.StripRenderer(_ToyScene(W, H), _ToyIntegrator(), rank, world, torch.device("cpu"), film_factory=film_cls,
                                solver_factory=lambda preset, alpha: _ToySolver())
    assert sr.preset == "L2D"
    out = sr.render(3)
    moved = False
    if rebalance:
        moved = sr.rebalance(min_rows=2)
        out = sr.render(3)
    res = (None if out is None else out.numpy().copy(), sr.last["rays"], sr.last["halo_bytes"], sr.film.exchanged, list(sr.strips), moved)
    sr.close()
    return res


def _strip_worker(rank, world, port, W, H, own_border, rebalance, q, early="0"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["GDPT_EARLY_GATHER"] = early          # "1": the opt-in second communicator with the gather's receives posted before the render (parallel.StripRenderer)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank,) + _toy_render(rank, world, W, H, _ToyFilmOwnBorder if own_border else _ToyFilm, rebalance))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("early", ["0", "1"], ids=["gather-after-halo", "early-gather"])
@pytest.mark.parametrize("world,own_border,rebalance", [(2, False, False), (3, False, True), (2, True, False), (3, True, True)])
def test_strip_renderer_equals_one_rank(world, own_border, rebalance, early):
    """Strips + halo exchange (box filter) or strips that render their own border (wider filters) + gather + solve on rank 0
    give exactly the one-rank image; rebalancing moves the boundaries and changes nothing in the image."""
    W, H = 12, 19
    whole = _toy_render(0, 1, W, H, _ToyFilmOwnBorder if own_border else _ToyFilm, False)[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strip_worker, args=(r, world, port, W, H, own_border, rebalance, q, early)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], whole)
    assert all(res[r][0] is None for r in range(1, world))
    assert sum(res[r][1] for r in range(world)) == 11 * H
    for r in range(world):
        img, rays, halo, exchanged, strips, moved = res[r]
        assert strips == res[0][4] and moved == rebalance
        if own_border:
            assert halo == 0 and exchanged == 0                     # nothing exchanged: the strip rendered its border itself
        else:
            assert halo == 8 * (NREC * W + 40 * W) * ((r > 0) + (r < world - 1))
    if rebalance:
        assert res[0][4] != parallel.row_strips(H, world) and res[0][4][0][1] > parallel.row_strips(H, world)[0][1]


# ---- parallel.GBDPTStripRenderer over gloo with doubles of the G-BDPT film / integrator / reconstruction ------------------------------
class _ToyBdFilm:
    """Double of gbdpt.Film: camera blocks get a sample's value at its own pixel, light images get splats that land on pixels of OTHER
    strips too (a light-tracing connection reaches any pixel) -- so only the sum over all ranks' films is the frame."""

    def __init__(self, scene):
        self.W, self.H = scene.width, scene.height
        self.clear()

    def clear(self):
        self.block = np.zeros((5, self.H, self.W, 4)); self.light = np.zeros((5, self.H, self.W, 3)); self.n = 0

    def render(self, spp, rect):
        x0, y0, x1, y1 = rect
        for y in range(y0, y1):
            for x in range(x0, x1):
                for b in range(5):
                    self.block[b, y, x] += spp * np.array([x + 1, y + 2, b + 3, 1.0])
                    ty, tx = (7 * y + 3 * x + b) % self.H, (5 * x + y) % self.W            # anywhere on the film
                    self.light[b, ty, tx] += spp * np.array([1.0, x, y])
                self.n += spp

    def sync(self):
        pass

    def export_device(self, block, light):
        block.copy_(torch.from_numpy(self.block)); light.copy_(torch.from_numpy(self.light))

    def import_device(self, block, light):
        self.block, self.light = block.numpy().copy(), light.numpy().copy()

    def develop_device(self, b, spp, t):
        w = self.block[b, ..., 3].copy(); w[w == 0] = 1.0
        t.copy_(torch.from_numpy((self.block[b, ..., :3] + self.light[b] * (w / spp)[..., None]) / w[..., None]))

    def stats(self):
        return dict(raysTraced=3 * self.n, shadowRaysTraced=self.n, samples=self.n, invalidPuts=0)

    def render_ms(self):
        return 1.0

    def close(self):
        pass


class _ToyBdIntegrator:
    reconstructAlpha = 0.2

    def config(self, spp, seed=5489):
        return spp

    def renderBlock(self, scene, film, cfg, rect):
        film.render(cfg, rect)


def _toy_bd_render(rank, world, W, H):
    sr = parallel.GBDPTStripRenderer(_ToyScene(W, H), _ToyBdIntegrator(), rank, world, torch.device("cpu"), film_factory=_ToyBdFilm,
                                     reconstruct=lambda bufs, w, h, alpha: ((bufs[0] + 2 * bufs[3] - bufs[2]).float(), (bufs[0] - bufs[4] + bufs[1]).float(), (1e-4, 2e-4)))
    out = sr.render(3)
    res = (None if out is None else {k: v.numpy().copy() for k, v in out.items()}, sr.last["rays"], sr.last["samples"], sr.last["reduce_bytes"], list(sr.strips))
    sr.close()
    return res


def _bd_worker(rank, world, port, W, H, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank,) + _toy_bd_render(rank, world, W, H))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gbdpt_strip_renderer_equals_one_rank(world):
    """G-BDPT over ranks: strips of camera samples + ONE reduction of the whole films onto rank 0 (light images reach every pixel) give the
    one-rank frame exactly (integer-valued sums), developed buffers and both reconstructions."""
    W, H = 9, 14
    whole = _toy_bd_render(0, 1, W, H)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bd_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res[r][0] is None for r in range(1, world))
    for k, v in whole[0].items():
        assert np.array_equal(res[0][0][k], v), k
    assert sum(res[r][2] for r in range(world)) == whole[2] == 3 * W * H and sum(res[r][1] for r in range(world)) == whole[1]
    assert all(res[r][3] == 8 * 5 * W * H * 7 for r in range(world)) and whole[3] == 0
    assert res[0][4] == parallel.row_strips(H, world)


def test_strips_of_one_row_are_refused_for_the_halo_exchange():
    """The halo carries the exact puts of TWO rows beyond a boundary and the neighbour adds the far one to the row inside its boundary row (include/gdpt_tracer.h):
    a one-row strip has no such row -- the hosts refuse the partition instead of losing the put (parallel.StripRenderer.set_strips; gdpt_host.hpp renderStrips)."""
    import types
    from gradientdomain_mitsuba_amd import parallel as PP
    stub = types.SimpleNamespace(world=2, height=3, rank=0, film=None)
    with pytest.raises(ValueError, match="at least two rows"):
        PP.StripRenderer.set_strips(stub, [(0, 1), (1, 3)])
    assert PP.rebalance_strips([(0, 4), (4, 8)], [1.0, 30.0], min_rows=2) == [(0, 6), (6, 8)]      # (the rebalance step keeps two rows when asked to)
