"""Row-strip sharding of the G-PT hot path over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference shards by 32x32 image blocks over CPU worker threads and merges block borders by addition
(/root/reference/src/librender/imageproc.cpp:28-78, gpt_proc.cpp:52-56,137-149).  Here each rank owns a contiguous strip of
rows; the only coupling is the one-pixel halo: a strip needs the per-pixel sample sums of the row just above and below it
(and passes on the exact-put spill it produced for its neighbours' rows).  That is two point-to-point messages per rank
(<= 2 xGMI links, no collective on the data path).  Reconstruction runs on rank 0 after a gather of the four developed
fp32 images (44 MB at 1280x720; the CG at this size is latency-bound and does not profit from splitting -- DESIGN.md).
"""
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


def gather_rows(strip, strips, width, rank, world, group=None):
    """Gather per-rank strips to rank 0 (None elsewhere).  strip: [rows_r, width, 3] -> [H, width, 3], or a stack of images
    [k, rows_r, width, 3] -> [k, H, width, 3] (one message per rank for all k images instead of k)."""
    if world == 1:
        return strip
    stacked = strip.dim() == 4
    lead = (strip.shape[0],) if stacked else ()
    if rank == 0:
        parts = [_wire(torch.empty(lead + (y1 - y0, width, 3), dtype=strip.dtype, device=strip.device)) for (y0, y1) in strips]
        for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, parts[r], r, group=group) for r in range(1, world)]):
            q.wait()
        parts[0] = strip
        full = torch.cat([p.to(strip.device) for p in parts], dim=1 if stacked else 0)
        _settle(full)
        return full
    for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, _wire(strip.contiguous()), 0, group=group)]):
        q.wait()
    _settle(strip)          # the library reuses the strip buffer on its own stream next step
    return None
