"""Shared seeded workload builder for tests, smoke() and bench.py (inputs only)."""
import numpy as np

from veloslam_amd import capi, synth


def make_workload(map_points=1_000_000, n_frames=1, first_frame=3, seed=42, azimuth_correction=False):
    """-> dict(map=(x,y,z), frames=[dict(sensor=..., pkt=..., table=..., T_true, T0, times)])
    The per-packet transform tables come from the PRODUCT's host code
    (velo_packet_transforms); tests check that against the oracle separately."""
    sc = synth.Scene()
    mx, my, mz = sc.sample_map(map_points)
    mo = synth.Motion()
    cal = synth.hdl64_calibration(azimuth_correction)
    frames = []
    for k in range(n_frames):
        fi = first_frame + k
        pk, ts, _ = synth.make_frame_packets(sc, mo, fi, cal, seed=seed)
        fr = synth.decode_sensor_frame(pk, cal)
        track = mo.ins_track(ts[0], ts[-1])
        poses, n = capi.make_poses(track)
        tab, valid, car = capi.packet_transforms(poses, n, ts)
        T_true = np.array([1, 0, 0, car.T[0], 0, 1, 0, car.T[1], 0, 0, 1, car.T[2]], dtype=np.float64)
        frames.append(dict(sensor=fr, table=tab, T_true=T_true, T0=synth.perturbed_guess(T_true),
                           times=ts, track=track, packets=pk))
    return dict(map=(mx, my, mz), frames=frames, calib=cal, scene=sc, motion=mo)


def rot_angle(Ra, Rb):
    """angle (rad) of Ra^T Rb for two 3x3 rotations."""
    R = Ra.T @ Rb
    return float(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))


def pose_delta(Ta, Tb):
    A = np.asarray(Ta, dtype=np.float64).reshape(3, 4)
    B = np.asarray(Tb, dtype=np.float64).reshape(3, 4)
    return float(np.linalg.norm(A[:, 3] - B[:, 3])), rot_angle(A[:, :3], B[:, :3])
