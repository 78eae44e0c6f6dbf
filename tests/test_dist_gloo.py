"""world_size-2 gloo tests (CPU) of the all-torch exchange path (veloslam_amd/dist.py: bench.py
--exchange torch, and the recorded fallback), and of the C library's host-side plan
(velo_exchange_plan) on counts that really came out of a two-process all-gather.  The default
transport of bench.py --gpus N -- RCCL behind velo_exchange_increments -- is a different
implementation: its pack kernel is held to numpy in tests/test_gpu_comm.py."""
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
            # the C library's plan for the same gathered counts places the same blocks
            from veloslam_amd import capi
            offs, pad, total = capi.exchange_plan(counts)
            cat = np.concatenate([b.numpy() for b in blocks], axis=1)
            assert cat.shape[1] == total and pad == max(max(counts), 1)
            for r, b in enumerate(blocks):
                assert np.array_equal(cat[:, offs[r]:offs[r + 1]], b.numpy())
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


def _replica_worker(rank, world, port, q):
    """Every rank holds a replica of the map; each contributes a different increment; all append
    the gathered blocks in rank order -> the replicas must stay IDENTICAL (sorted order, cell
    table), which is what makes the frame-parallel results independent of the rank count."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        base = np.random.default_rng(7).uniform(0, 12, (3, 4000)).astype(np.float32)
        roll = orc.RollingMap(*base, 1.0, 8, 3, margin=2)
        digests = []
        for rnd in range(3):
            rng = np.random.default_rng(1000 * rnd + rank)
            n = [2600, 0, 2041][(rank + rnd) % 3]      # (mapping volume: the mapping stream accepts 2 300 points per frame)
            mine = rng.uniform(1, 11, (3, n)).astype(np.float32)
            buf = torch.zeros((3, 4096), dtype=torch.float32)
            buf[:, :n] = torch.from_numpy(mine)
            blocks, counts = exchange_increments(buf, n)
            for b in blocks:                      # rank order
                if b.shape[1]:
                    roll.append(*b.numpy())
            m = roll.map
            digests.append((roll.n, m.perm().tobytes(), m.cell_start().tobytes(),
                            m.normals()[0].tobytes(), counts))
        q.put((rank, digests))
    finally:
        dist.destroy_process_group()


def test_replicas_stay_identical_after_rank_order_append():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_replica_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rnd in range(3):
        assert res[0][rnd] == res[1][rnd]       # size, permutation, cell table, normals, counts
    assert res[0][2][0] == 4000 + sum(sum(res[0][r][4]) for r in range(3))


def test_exchange_single_process_passthrough():
    buf = torch.arange(30, dtype=torch.float32).view(3, 10)
    blocks, counts = exchange_increments(buf, 4)
    assert counts == [4] and torch.equal(blocks[0], buf[:, :4])


def test_shard_units_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(u for r in range(world) for u in shard_units(21, r, world))
        assert seen == list(range(21))
