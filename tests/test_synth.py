"""The synthetic input generator (veloslam_amd/synth.py) is only a manufacturer of inputs, but
its numpy packet decode is what the benchmark's "decoded HDLFrame" comes from: hold it to the
oracle's restatement of HDLParser (no pose store: frames stay in the sensor frame)."""
import numpy as np
import pytest

from veloslam_amd import synth


@pytest.mark.parametrize("azcorr", [False, True])
def test_numpy_decode_equals_oracle_parser(oracle, azcorr):
    sc, mo = synth.Scene(), synth.Motion()
    cal = synth.hdl64_calibration(azcorr)
    pk, ts, _ = synth.make_frame_packets(sc, mo, 5, cal, seed=42)
    assert len(pk) == synth.PKTS_PER_FRAME and all(len(p) == 1206 for p in pk)
    fr = synth.decode_sensor_frame(pk, cal)
    dec = oracle.Decoder(cal, 64, None)
    for p, t in zip(pk, ts):
        dec.packet(p, t)
    dec.flush()
    assert dec.num_frames == 1
    n = 0
    for b in range(64):
        x, y, z, it, az, dist = dec.beam(0, b)
        lo, hi = fr["beam_start"][b], fr["beam_start"][b + 1]
        assert hi - lo == x.size
        assert np.array_equal(fr["x"][lo:hi].view(np.uint32), x.view(np.uint32))
        assert np.array_equal(fr["y"][lo:hi].view(np.uint32), y.view(np.uint32))
        assert np.array_equal(fr["z"][lo:hi].view(np.uint32), z.view(np.uint32))
        assert np.array_equal(fr["intensity"][lo:hi], it)
        assert np.array_equal(fr["azimuth"][lo:hi], az)
        assert np.array_equal(fr["distance"][lo:hi].view(np.uint32), dist.view(np.uint32))
        n += x.size
    assert n == fr["x"].size and n > 100_000


def test_scene_and_motion_are_seeded():
    a = synth.Scene().sample_map(5000)
    b = synth.Scene().sample_map(5000)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    mo = synth.Motion()
    tr = mo.ins_track(1_000_000, 1_100_000)
    assert len(tr) >= 10 and tr[0][3] <= 1_000_000 and tr[-1][3] >= 1_100_000
    T = np.array([1, 0, 0, 1.0, 0, 1, 0, 2.0, 0, 0, 1, 3.0])
    g = synth.perturbed_guess(T)
    assert np.allclose(g.reshape(3, 4)[:, 3], [1.30, 1.80, 3.05])
