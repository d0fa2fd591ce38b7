"""Clip sharding across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" for the CPU tests).

The path has no cross-clip arithmetic: every normalisation, cutoff search and
softmax is per clip, so multi-GPU execution is a scatter of low-rate clips (+ their prior noise),
an independent `generate` per rank, and a gather of the 48 kHz waveforms.  No all-reduce exists
anywhere, so nothing here depends on ring bandwidth; at B = 256 over 8 GPUs the payload per peer is
32 clips x 1.92 MB = 61 MB (~0.4 ms on one 153 GB/s xGMI link).  The sharded result is bit-identical
to the single-GPU result for the same clips, which tests assert.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world):
    """Contiguous split: [(start, stop)] per rank; the first n % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < extra else 0)
        out.append((s, e))
        s = e
    return out


def _p2p(ops):
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def scatter_rows(full, shape_tail, dtype, device, src=0, group=None, n_total=None):
    """Rank `src` holds `full` [B, *tail]; every rank returns its contiguous row shard.
    B is broadcast first so that the other ranks can size their buffers, unless every rank already knows it
    (`n_total`: no collective and no host sync on the data path then)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if n_total is None:
        nb = torch.tensor([full.shape[0] if rank == src else 0], dtype=torch.int64, device=device)
        dist.broadcast(nb, src, group=group)
        n_total = int(nb.item())
    bounds = shard_bounds(int(n_total), world)
    s, e = bounds[rank]
    if rank == src:
        ops = [dist.P2POp(dist.isend, full[a:b].contiguous(), r, group) for r, (a, b) in enumerate(bounds)
               if r != src and b > a]
        _p2p(ops)
        return full[s:e].contiguous(), bounds
    mine = torch.empty((e - s,) + tuple(shape_tail), dtype=dtype, device=device)
    if e > s:
        _p2p([dist.P2POp(dist.irecv, mine, src, group)])
    return mine, bounds


def gather_rows(mine, bounds, dst=0, group=None):
    """Inverse of scatter_rows: rank `dst` returns [B, *tail], the others None."""
    rank = dist.get_rank(group)
    if rank != dst:
        if mine.shape[0]:
            _p2p([dist.P2POp(dist.isend, mine.contiguous(), dst, group)])
        return None
    total = bounds[-1][1]
    out = torch.empty((total,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    s, e = bounds[rank]
    out[s:e] = mine
    ops = [dist.P2POp(dist.irecv, out[a:b], r, group) for r, (a, b) in enumerate(bounds) if r != dst and b > a]
    _p2p(ops)
    return out


def generate_sharded(generate_fn, x, noise, n_in, n_frames, n_mels=256, src=0, group=None, device=None,
                     n_total=None, t48=None):
    """x [B, n_in] low-rate clips and noise [B, n_frames, n_mels] live on rank `src` (None elsewhere).
    `generate_fn(x_shard, noise_shard) -> [b, T48]` runs on every rank (e.g. FlowHighSR.generate_from_device).
    Returns [B, T48] on rank `src`, None on the others.
    n_total (= B) and t48 (output samples per clip), when every rank knows them, remove the two size exchanges
    and their host syncs: the step is then P2P scatter -> generate -> P2P gather, nothing else.
    Without an initialised process group (one GPU) the scatter / gather are the identity."""
    device = device if device is not None else (x.device if x is not None else torch.device("cpu"))
    if not (dist.is_available() and dist.is_initialized()):
        return generate_fn(x, noise)
    rank = dist.get_rank(group)
    xs, bounds = scatter_rows(x if rank == src else None, (n_in,), torch.float32, device, src, group, n_total)
    ns, _ = scatter_rows(noise if rank == src else None, (n_frames, n_mels), torch.float32, device, src, group, n_total)
    out = generate_fn(xs, ns) if xs.shape[0] else torch.empty(0, 0, device=device)
    if t48 is None:
        if not xs.shape[0]:                  # ranks without clips still take part in the gather
            t = torch.tensor([0], dtype=torch.int64, device=device)
        else:
            t = torch.tensor([out.shape[1]], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        t48 = int(t.item())
    if not xs.shape[0]:
        out = torch.empty(0, int(t48), dtype=torch.float32, device=device)
    return gather_rows(out, bounds, src, group)
