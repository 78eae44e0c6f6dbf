"""Pin the oracle's CoordiTran restatement bit-for-bit against
vectors cut from the reference's own object code
(tests/golden/make_coorditran_golden.py -> tests/golden/coorditran.json)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "coorditran.json")


def unhex(v):
    return np.array([float.fromhex(s) for s in v], dtype=np.float64)


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_geodesy_bit_exact(oracle, gold):
    assert len(gold["llh_cases"]) == 300
    for c in gold["llh_cases"]:
        llh, org = unhex(c["llh"]), unhex(c["org"])
        xyz = oracle.llh2xyz(llh)
        assert same_bits(xyz, unhex(c["llh2xyz"]))
        assert same_bits(oracle.xyz2llh(xyz), unhex(c["xyz2llh"]))
        enu = oracle.llh2enu(llh, org)
        assert same_bits(enu, unhex(c["llh2enu"]))
        assert same_bits(oracle.xyz2enu(xyz, org), unhex(c["xyz2enu"]))
        assert same_bits(oracle.enu2xyz(enu, org), unhex(c["enu2xyz"]))
        assert same_bits(oracle.enu2llh(enu, org), unhex(c["enu2llh"]))


def test_eulr2dcm_bit_exact(oracle, gold):
    for c in gold["eulr2dcm"]:
        assert same_bits(oracle.eulr2dcm(unhex(c["eul"])).ravel(), unhex(c["dcm"]))


def test_mapping_angle_bit_exact(oracle, gold):
    for c in gold["mapping_angle"]:
        assert same_bits([oracle.mapping_angle(float.fromhex(c["angle"]))],
                         [float.fromhex(c["out"])])


def test_survey_known_answers(oracle):
    """The decimal known answers quoted in SURVEY.md 8(c)."""
    llh = np.array([np.radians(39.8569901), np.radians(116.1736406), 89.09288895])
    xyz = oracle.llh2xyz(llh)
    np.testing.assert_allclose(xyz, [-2162664.730803801, 4400224.072115005, 4065866.035327236],
                               rtol=0, atol=1e-8)
    back = oracle.xyz2llh(xyz)
    np.testing.assert_allclose(np.degrees(back[:2]), [39.8569901, 116.1736406], atol=1e-11)
    assert abs(back[2] - 89.092888949) < 1e-6
    llh2 = np.array([np.radians(39.8579901), np.radians(116.1746406), 90.0])
    np.testing.assert_allclose(oracle.llh2enu(llh2, xyz),
                               [85.571696745, 111.033944559, 0.905568857], atol=1e-8)
    np.testing.assert_allclose(oracle.eulr2dcm([0.1, -0.2, 0.3])[0],
                               [0.936293363584199, 0.275095847318244, 0.218350663146334],
                               atol=1e-15)
    np.testing.assert_allclose([oracle.mapping_angle(a) for a in (45, 180, 300)],
                               [0.785398163397, -1.570796326795, 2.617993877991], atol=1e-12)
