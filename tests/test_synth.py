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


def test_long_scene_rays_agree_between_numpy_and_torch():
    """synth.LongScene (the street of the mapping stream): one ray caster written for numpy and torch -- the same hit
    distances bit for bit on the CPU, hits on every kind of surface, and nothing beyond the sensor's culling range."""
    import torch
    sc = synth.LongScene(300.0)
    rng = np.random.default_rng(11)
    n = 20_000
    o = np.column_stack([rng.uniform(48.0, 52.0, n), rng.uniform(-1.0, 1.0, n), np.full(n, synth.SENSOR_HEIGHT)])
    d = rng.normal(size=(n, 3))
    d[:, 2] = -np.abs(d[:, 2]) * 0.15 + rng.uniform(-0.02, 0.05, n)
    d /= np.linalg.norm(d, axis=1)[:, None]
    a = sc.raycast(o, d)
    b = sc.raycast(torch.as_tensor(o), torch.as_tensor(d), xp=torch).numpy()
    assert np.array_equal(a, b)
    hit = np.isfinite(a)
    assert 0.7 < hit.mean() <= 1.0
    p = o + a[:, None] * d
    ph = p[hit]
    ground = np.abs(ph[:, 2]) < 1e-6
    wall = np.abs(np.abs(ph[:, 1]) - sc.wall_y) < 1e-6
    fin = np.any(np.abs(ph[:, 0][:, None] - sc.fin_x[None, :]) < 1e-6, axis=1) & ~wall
    assert ground.sum() > 1000 and wall.sum() > 100 and fin.sum() > 10
    assert (~(ground | wall | fin)).sum() > 10                      # poles and boxes
    assert np.all(ph[:, 2] > -1e-6) and np.all(np.abs(ph[:, 1]) <= sc.ground_half + 1e-6)


def test_device_frame_generator_makes_packets_the_parser_reads(oracle):
    """make_frame_packets_device (ray casting + packet assembly in torch; here on the CPU device): 1206-byte packets in the
    wire layout of HDLParser.cxx:67-87 -- the oracle's parser and the numpy decode read the same frame out of them, the
    frame is a full revolution, and consecutive frames differ (the car moved)."""
    import torch
    sc = synth.LongScene(200.0)
    mo = synth.Motion(p0=(0.0, 0.0, synth.SENSOR_HEIGHT))
    cal = synth.hdl64_calibration()
    pk, ts = synth.make_frame_packets_device(sc, mo, [0, 1], cal, torch.device("cpu"))
    assert tuple(pk.shape) == (2, synth.PKTS_PER_FRAME, 1206) and ts.shape == (2, synth.PKTS_PER_FRAME)
    assert ts[0, 0] == mo.t0_us and ts[1, 0] == mo.t0_us + synth.FRAME_US and np.all(np.diff(ts.reshape(-1)) > 0)
    sizes = []
    for k in range(2):
        packets = [bytes(p) for p in pk[k].numpy()]
        fr = synth.decode_sensor_frame(packets, cal)
        dec = oracle.Decoder(cal, 64, None)
        for p, t in zip(packets, ts[k]):
            dec.packet(p, int(t))
        dec.flush()
        assert dec.num_frames == 1
        x, y, z = dec.frame_cloud(0)[:3]
        assert x.size == fr["x"].size > 100_000
        assert np.array_equal(x.view(np.uint32), fr["x"].view(np.uint32)) and np.array_equal(z.view(np.uint32), fr["z"].view(np.uint32))
        r = np.sqrt(fr["x"].astype(np.float64) ** 2 + fr["y"] ** 2 + fr["z"] ** 2)
        assert 0.9 < r.min() and r.max() < synth.MAX_RANGE + 1.0
        sizes.append(fr["x"].size)
    assert not np.array_equal(pk[0].numpy(), pk[1].numpy())
