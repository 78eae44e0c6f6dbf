"""world_size-2 gloo test of the multi-GPU exchange step (runs on CPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from veloslam_amd.dist import exchange_increments, shard_units


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        for rnd in range(3):
            rng = np.random.default_rng(100 * rnd + rank)
            n = [0, 5, 37][(rank + rnd) % 3]  # ragged, including an empty contribution
            buf = torch.zeros((3, 64), dtype=torch.float32)
            buf[:, :n] = torch.from_numpy(rng.uniform(-1, 1, (3, n)).astype(np.float32))
            blocks, counts = exchange_increments(buf, n)
            out.append(([b.numpy().copy() for b in blocks], counts))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_exchange_increments_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rnd in range(3):
        b0, c0 = res[0][rnd]
        b1, c1 = res[1][rnd]
        assert c0 == c1  # every rank sees the same counts ...
        for r in range(world):  # ... and the same blocks, in rank order
            assert np.array_equal(b0[r], b1[r])
            rng = np.random.default_rng(100 * rnd + r)
            n = [0, 5, 37][(r + rnd) % 3]
            assert c0[r] == n
            assert np.array_equal(b0[r], rng.uniform(-1, 1, (3, n)).astype(np.float32))


def test_exchange_single_process_passthrough():
    buf = torch.arange(30, dtype=torch.float32).view(3, 10)
    blocks, counts = exchange_increments(buf, 4)
    assert counts == [4] and torch.equal(blocks[0], buf[:, :4])


def test_shard_units_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(u for r in range(world) for u in shard_units(21, r, world))
        assert seen == list(range(21))
