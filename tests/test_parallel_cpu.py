"""CPU (gloo, world_size 2 and 3): the clip scatter / gather used for multi-GPU runs reproduces the
unsharded result bit for bit, including ragged splits and ranks that receive no clip."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flowhigh_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_generate(x, noise):
    """Stand-in for generate(): per-clip, deterministic, non-linear, depends on both inputs."""
    y = torch.tanh(x.repeat_interleave(4, dim=1) * 3.0)
    y = y / y.abs().amax(dim=1, keepdim=True) * 0.99
    return y + noise.mean(dim=(1, 2))[:, None] * 1e-3


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(n_clips, 50, generator=g) if rank == 0 else None
        noise = torch.randn(n_clips, 5, 8, generator=g) if rank == 0 else None
        out = parallel.generate_sharded(_fake_generate, x, noise, 50, 5, n_mels=8, device=torch.device("cpu"))
        if rank == 0:
            q.put(torch.equal(out, _fake_generate(x, noise)) and out.shape == (n_clips, 200))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_clips", [(2, 8), (2, 5), (3, 2), (2, 1)])
def test_sharded_generate_equals_unsharded(world, n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_bounds():
    assert parallel.shard_bounds(256, 8) == [(32 * i, 32 * i + 32) for i in range(8)]
    assert parallel.shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert parallel.shard_bounds(2, 3) == [(0, 1), (1, 2), (2, 2)]
