"""Randomised GPU-vs-oracle parity cases (tools/fuzz_parity.py): lattice / duplicate / clustered
maps, random voxel, sub-division, k, d_max, hinted pose sequences and rolling-map operations,
everything compared bit for bit.  Seeds 1002 and 1019 are regressions: a miscompiled normal
sign flip (vz == 0 branch) and a neighbour at exactly one voxel distance that made a carried
normal depend on the grid anchoring."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1002, 1019] + list(range(2000, 2012)))
def test_fuzz_case(seed):
    """both linearise kernels on every seed (the regression seeds were found on what is now the
    throughput kernel); the table kind, auto sub-division and the round-2 map operations follow
    the seed"""
    from tools import fuzz_parity
    from veloslam_amd import capi
    fuzz_parity.one_case(seed, kernel=capi.KERNEL_THROUGHPUT)
    fuzz_parity.one_case(seed, kernel=capi.KERNEL_LATENCY)


@pytest.mark.parametrize("seed", range(3000, 3006))
def test_fuzz_case_hash_table(seed):
    from tools import fuzz_parity
    fuzz_parity.one_case(seed, hash_load=50)
