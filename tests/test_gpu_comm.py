"""The exchange step behind the C ABI (velo_comm_*, velo_exchange_increments): RCCL opened at
run time, counts + max-padded blocks, rank-order packing.  One GPU here, so the communicator has
one rank (RCCL refuses two ranks on one device); the N > 1 packing logic is the same code path
with W > 1 and is covered on CPU tensors by tests/test_dist_gloo.py."""
import numpy as np
import pytest
import torch

from veloslam_amd import capi

pytestmark = pytest.mark.gpu


def test_exchange_requires_a_communicator():
    c = capi.Context(0, max_batch=2)
    try:
        assert c.comm_info() == (0, 0)
        buf = torch.zeros((3, 8), dtype=torch.float32, device="cuda")
        with pytest.raises(capi.VeloError):
            c.exchange_increments(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), 4,
                                  buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), 8)
    finally:
        c.close()


def test_single_rank_exchange_roundtrip_and_append(oracle):
    rng = np.random.default_rng(3)
    base = rng.uniform(0, 10, (3, 5000)).astype(np.float32)
    inc = rng.uniform(2, 8, (3, 700)).astype(np.float32)
    c = capi.Context(0, max_batch=2, map_margin=2)
    try:
        c.comm_init(capi.comm_unique_id(), 0, 1)
        assert c.comm_info() == (0, 1)
        with pytest.raises(capi.VeloError):
            c.comm_init(capi.comm_unique_id(), 0, 1)   # once per ctx
        c.map_reset(*base, 1.0, 8)
        src = torch.from_numpy(inc).cuda()
        dst = torch.full((3, 1024), -1.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        counts, total = c.exchange_increments(src[0].data_ptr(), src[1].data_ptr(), src[2].data_ptr(), 700,
                                              dst[0].data_ptr(), dst[1].data_ptr(), dst[2].data_ptr(), 1024,
                                              after_async_increment=False)
        assert counts == [700] and total == 700
        # the append is stream-ordered behind the exchange
        c.map_append_dev(dst[0].data_ptr(), dst[1].data_ptr(), dst[2].data_ptr(), total)
        c.synchronize()
        assert np.array_equal(dst[:, :700].cpu().numpy(), inc)
        assert np.all(dst[:, 700:].cpu().numpy() == -1.0)
        roll = oracle.RollingMap(*base, 1.0, 8, 3, margin=2)
        roll.append(*inc)
        g = c.map_download()
        assert np.array_equal(g["cell_start"], roll.map.cell_start()) and np.array_equal(g["perm"], roll.map.perm())
        # an empty contribution and a capacity overflow
        counts, total = c.exchange_increments(0, 0, 0, 0, dst[0].data_ptr(), dst[1].data_ptr(), dst[2].data_ptr(),
                                              1024, after_async_increment=False)
        assert counts == [0] and total == 0
        with pytest.raises(capi.VeloError):
            c.exchange_increments(src[0].data_ptr(), src[1].data_ptr(), src[2].data_ptr(), 700,
                                  dst[0].data_ptr(), dst[1].data_ptr(), dst[2].data_ptr(), 100,
                                  after_async_increment=False)
    finally:
        c.close()
