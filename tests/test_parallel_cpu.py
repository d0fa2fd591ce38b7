"""CPU (gloo, world_size 2 and 3): the clip scatter / gather used for multi-GPU runs reproduces the
unsharded result bit for bit, including ragged splits and ranks that receive no clip."""
import os

import numpy as np
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
        # ... and with the sizes known to every rank (bench.py --config 4: no size exchange, no host sync)
        out2 = parallel.generate_sharded(_fake_generate, x, noise, 50, 5, n_mels=8, device=torch.device("cpu"),
                                         n_total=n_clips, t48=200)
        if rank == 0:
            q.put(torch.equal(out, _fake_generate(x, noise)) and out.shape == (n_clips, 200) and torch.equal(out, out2))
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


def test_sharded_generate_without_process_group_is_the_identity():
    g = torch.Generator().manual_seed(1)
    x, noise = torch.randn(3, 50, generator=g), torch.randn(3, 5, 8, generator=g)
    assert not dist.is_initialized()
    assert torch.equal(parallel.generate_sharded(_fake_generate, x, noise, 50, 5, n_mels=8), _fake_generate(x, noise))


def test_shard_bounds():
    assert parallel.shard_bounds(256, 8) == [(32 * i, 32 * i + 32) for i in range(8)]
    assert parallel.shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert parallel.shard_bounds(2, 3) == [(0, 1), (1, 2), (2, 2)]


# ------------------------------------------------------------------------------------------
# serving front (host logic only: a stub model records what the batcher hands it)
# ------------------------------------------------------------------------------------------
class _StubModel:
    def __init__(self):
        self.calls = []

    def _draw_noise(self, batch, n_frames, generator):
        return torch.randn(batch, n_frames, 4, generator=generator)

    def generate_many(self, clips, sr, target, steps, noise=None, max_batch=64):
        self.calls.append((len(clips), sr, steps, None if noise is None else [tuple(n.shape) for n in noise]))
        return [torch.from_numpy(np.asarray(c, dtype=np.float32))[None] * 2 for c in clips]


def test_batching_server_groups_and_returns_in_order():
    import threading
    import numpy as np
    from flowhigh_amd.serve import BatchingServer
    m = _StubModel()
    srv = BatchingServer(m, max_batch=8, max_wait_ms=200)
    clips = [np.full(120 + 10 * (i % 3), float(i), dtype=np.float32) for i in range(9)]
    futs = [None] * len(clips)

    def client(i):
        futs[i] = srv.submit(clips[i], 12000 if i % 2 == 0 else 16000, timestep=1, seed=i)
    threads = [threading.Thread(target=client, args=(i,)) for i in range(len(clips))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    outs = [f.result(timeout=30) for f in futs]
    for i, o in enumerate(outs):                         # every caller gets ITS clip back
        assert o.dtype == np.float32 and o.shape == clips[i].shape and np.all(o == 2 * clips[i])
    assert sum(c[0] for c in m.calls) == len(clips)
    assert {c[1] for c in m.calls} == {12000, 16000}     # never mixes input rates in one call
    assert len(m.calls) <= 4                             # ... and did batch the concurrent requests
    sr, y = srv.generate((12000, (clips[0] * 100).astype(np.int16)), 48000, 1)     # app.py's signature
    assert sr == 48000 and y.shape == clips[0].shape
    with pytest.raises(NotImplementedError):
        srv.generate((12000, clips[0]), 44100, 1)
    srv.close()
    with pytest.raises(RuntimeError):
        srv.submit(clips[0], 12000)


def _calib_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flowhigh_amd import vocoder as V
        calls = []

        def measure():              # rank 0's box has the clock dip, the others' measurement would say it has not
            calls.append(rank)
            return [{0: 615.0, 3: 560.0}, {0: 610.0, 3: 570.0}] if rank == 0 else [{0: 540.0, 3: 560.0}] * 2
        choice, passes = V.decide_act_blocks(measure, collective=True)
        # a failing rank 0 reaches every rank as an error instead of leaving them in the broadcast
        def broken():
            raise ValueError("no device")
        try:
            V.decide_act_blocks(broken if rank == 0 else measure, collective=True)
            failed = False
        except RuntimeError as e:
            failed = "ValueError: no device" in str(e)
        # and without collective=True nothing is exchanged: every rank's own measurement
        own = V.decide_act_blocks(measure)[0]
        q.put((rank, choice, len(calls), passes[0][3], failed, own))
    finally:
        dist.destroy_process_group()


def test_act_occupancy_choice_is_measured_on_rank_0_and_broadcast():
    """vocoder.decide_act_blocks(collective=True) (sync_act_blocks: an explicit call every rank makes): only rank 0 times launch
    pairs, every rank takes its choice (eight ranks calibrating concurrently under one power budget would each measure something
    else); rank 0's failure raises on every rank; the constructors' form (collective=False) never communicates."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_calib_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, 3, 2, 560.0, True, 3), (1, 3, 1, 560.0, True, 0)]
    # without a process group: measured locally
    from flowhigh_amd import vocoder as V
    assert V.decide_act_blocks(lambda: [{0: 540.0, 3: 560.0}] * 2)[0] == 0
