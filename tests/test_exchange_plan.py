"""Host half of the multi-GPU exchange step (velo_exchange_plan: per-rank counts -> offsets, pad,
total), on CPU: it needs no ctx and no GPU.  The device half (the pack kernel) is held to numpy
for W = 1..64 in tests/test_gpu_comm.py; the transport (RCCL all-gather with world > 1) can only
run on a multi-GPU node -- bench.py --gpus N records exchange.ranks from velo_comm_info there."""
import numpy as np
import pytest

from veloslam_amd import capi


def test_plan_is_the_exclusive_prefix_sum():
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 7, 8, 33, capi.VELO_MAX_RANKS):
        for _ in range(20):
            counts = rng.integers(0, 50_000, world) * (rng.random(world) < 0.7)
            offs, pad, total = capi.exchange_plan(counts)
            assert offs.dtype == np.uint32 and offs.size == world + 1
            assert np.array_equal(offs, np.concatenate([[0], np.cumsum(counts)]))
            assert total == counts.sum() and pad == max(int(counts.max()), 1)


def test_plan_of_nothing_still_has_a_block():
    offs, pad, total = capi.exchange_plan([0, 0, 0])
    assert list(offs) == [0, 0, 0, 0] and total == 0 and pad == 1   # no zero-length collective


def test_plan_refuses_what_it_cannot_express():
    with pytest.raises(capi.VeloError) as e:
        capi.exchange_plan([3, -1])
    assert e.value.code == -1
    with pytest.raises(capi.VeloError):
        capi.exchange_plan([])
    with pytest.raises(capi.VeloError):
        capi.exchange_plan([1] * (capi.VELO_MAX_RANKS + 1))
    big = [2 ** 31 - 1, 2 ** 31 - 1, 2]                         # 2^32: one past the 32-bit index
    with pytest.raises(capi.VeloError) as e:
        capi.exchange_plan(big)
    assert e.value.code == -5
    offs, pad, total = capi.exchange_plan([2 ** 31 - 1, 2 ** 31 - 1, 1])
    assert total == 2 ** 32 - 1 and offs[-1] == 2 ** 32 - 1


def test_create_refuses_a_cfg_from_another_header():
    """the ABI version is checked where a consumer hands its structs over (no GPU is
    needed to be refused... but device discovery comes first, so only the message is checked when
    there is a GPU)."""
    import ctypes as C
    L = capi.lib()
    cfg = capi.Cfg()
    cfg.struct_size = C.sizeof(capi.Cfg)
    cfg.abi_version = 1
    h = L.velo_create(0, C.byref(cfg))
    assert not h
    msg = L.velo_last_error(None).decode()
    assert "abi_version" in msg or "no HIP device" in msg
