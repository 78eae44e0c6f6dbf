"""A recorded drive on disk (veloslam_amd/drive.py): the reference's file formats either side of
the path -- pcap + frame index, carposes.txt, db.xml, MapManager's tile file -- written by the
exporter, read back through the C ABI / the C++ classes; and (GPU) the replay of such a drive
against a rolling device map from C++ (tools/stream_driver.cpp -> veloslam::MapManager) and from
Python (bench.py --workload stream --drive), which must agree."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from veloslam_amd import capi, drive, synth
from tests.test_cpp_api import build_exe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tools", "stream_driver")


def build_driver():
    src = os.path.join(ROOT, "tools", "stream_driver.cpp")
    csrc = os.path.join(ROOT, "veloslam_amd", "csrc")
    import glob
    deps = [src, os.path.join(csrc, "libveloslam_amd.so")] + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        glob.glob(os.path.join(ROOT, "include", "veloslam", "*.hpp"))
    if (not os.path.exists(DRIVER)) or os.path.getmtime(DRIVER) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["hipcc", "-std=c++17", "-O2", "-x", "c++", src, "-I", os.path.join(ROOT, "include"),
                               "-L", csrc, "-lveloslam_amd", "-Wl,-rpath," + csrc, "-o", DRIVER])
    return DRIVER


def test_map_file_written_by_python_is_what_mapmanager_loads(tmp_path):
    """world.map: veloslam::MapManager::load reads the exporter's file -- same tile count, same
    points, the same tile for a given position (getPatchIdx on both sides), tilesInRange = the
    tiles overlapping the +-ROI_RANGE square."""
    rng = np.random.default_rng(4)
    n = 30_000
    x = rng.uniform(-260, 240, n).astype(np.float32)
    y = rng.uniform(-130, 170, n).astype(np.float32)
    z = rng.normal(0, 1, n).astype(np.float32)
    x[:4] = [-5.0, 5.0, 4.9999995, -15.0]          # on and next to tile edges (10 m tiles: edges at +-5, +-15 ...)
    path = str(tmp_path / "world.map")
    n_tiles = drive.write_map_file(path, x, y, z, 10.0)
    pr, tiles = drive.read_map_file(path)
    assert pr == 10.0 and len(tiles) == n_tiles and sum(t[2].size for t in tiles) == n
    ti, tj = drive.tile_index(x, y, 10.0)
    for cx, cy, tx, ty, tz in tiles[::37]:
        i, j = int(round(cx / 10)), int(round(cy / 10))
        sel = (ti == i) & (tj == j)
        assert np.array_equal(tx, x[sel]) and np.array_equal(tz, z[sel])     # input order inside a tile
    exe = build_exe()
    qx, qy = 12.3, -41.0
    out = subprocess.run([exe, "--load", path, str(qx), str(qy)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    assert lines["loaded"] == "%d %d" % (n_tiles, n)
    i0, j0 = drive.tile_index(qx - 100, qy - 100, 10.0)
    i1, j1 = drive.tile_index(qx + 100, qy + 100, 10.0)
    inside = (ti >= i0) & (ti <= i1) & (tj >= j0) & (tj <= j1)
    n_in_tiles = len({(a, b) for a, b in zip(ti[inside], tj[inside])})
    assert lines["inrange"] == "%d %d" % (n_in_tiles, int(inside.sum()))
    qi, qj = drive.tile_index(qx, qy, 10.0)
    cnt, cx, cy = lines["tile"].split()
    assert int(cnt) == int(((ti == qi) & (tj == qj)).sum()) and float(cx) == qi * 10.0 and float(cy) == qj * 10.0


def test_carposes_roundtrip_through_the_reference_format(tmp_path):
    """carposes.txt (TransformManager.cxx:95-125): x, y and the angles survive (radians in the
    file, yaw sign flipped, 8 h added to the stamp); z and the velocity vector do not exist in
    the format."""
    import ctypes as C
    mo = synth.Motion()
    samples = mo.ins_track(mo.t0_us, mo.t0_us + 300_000)
    samples = [(T, np.array([1.5, -0.7, R[2]]), V, t) for T, R, V, t in samples]
    path = str(tmp_path / "carposes.txt")
    drive.write_carposes(path, samples)
    n = C.c_size_t()
    L = capi.lib()
    L.velo_carposes_read.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    assert L.velo_carposes_read(path.encode(), None, 0, C.byref(n)) == 0 and n.value == len(samples)
    poses = (capi.Pose * n.value)()
    assert L.velo_carposes_read(path.encode(), poses, n.value, C.byref(n)) == 0
    for p, (T, R, V, t) in zip(poses, samples):
        assert p.T[0] == T[0] and p.T[1] == T[1] and p.T[2] == 0.0
        assert abs(p.R[0] - R[0]) < 1e-12 and abs(p.R[1] - R[1]) < 1e-12 and abs(p.R[2] - R[2]) < 1e-12
        assert p.t_us == t + drive.EIGHT_H_US and list(p.V) == [0.0, 0.0, 0.0]
    small = (capi.Pose * 2)()
    assert L.velo_carposes_read(path.encode(), small, 2, C.byref(n)) == -5
    assert L.velo_carposes_read(str(tmp_path / "none.txt").encode(), None, 0, C.byref(n)) == -6


def test_export_and_load_a_small_drive(tmp_path):
    rng = np.random.default_rng(1)
    world = tuple(rng.uniform(-50, 50, 5000).astype(np.float32) for _ in range(3))
    meta = drive.export_synthetic(str(tmp_path), n_frames=2, patch_range=25.0, world_xyz=world)
    d = drive.load(str(tmp_path))
    assert meta["n_frames"] == 2 and len(d["index"]) == 2 and d["packets"].size == 600 * 1206
    assert [e.first_packet for e in d["index"]] == [0, 300] and d["index"][1].firing_skip == 0
    assert d["n_poses"] > 10 and d["calib"].shape == (64, 9)
    # packet stamps and pose stamps share a clock: every packet has a pose bracket
    assert d["poses"][0].t_us <= d["times"][0] and d["times"][-1] <= d["poses"][d["n_poses"] - 1].t_us
    cal = synth.hdl64_calibration()
    assert np.allclose(d["calib"][:, :5], np.asarray(cal)[:, :5], rtol=0, atol=1e-12)


def test_packet_file_reader_and_writer_classes(tmp_path):
    """veloslam::PacketFileWriter / PacketFileReader -- the reference's vtkPacketFileWriter /
    vtkPacketFileReader without libpcap (vtkPacketFileWriter.cxx:118-161, vtkPacketFileReader.h:166-197):
    lidar (1206) and position (512) packets get their 42-byte prefixes, any other length is refused;
    the reader returns every UDP payload with its record time, closes itself at the end, re-reads
    from a remembered position; the bulk C functions see the same capture, and the class writes
    byte for byte what velo_pcap_write writes."""
    exe = build_exe()
    out = subprocess.run([exe, "--pcapfile", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    assert lines["written"] == "1 refused 1 open 0"
    # size @ file position : first payload byte / microseconds after the first record
    assert lines["records"] == "1206@24:0/0 1206@1288:1/553 512@2552:255/600 1206@3122:2/1106"
    assert lines["closed"] == "1" and lines["reread"] == "512 1206:2"
    assert lines["bulk"] == "3 0 1 2 1106"
    assert lines["same_bytes"] == "1 %d" % (24 + 3 * 1264) and lines["missing"] == "0 1"
    raw = open(os.path.join(str(tmp_path), "w.pcap"), "rb").read()
    assert len(raw) == 24 + 3 * 1264 + (16 + 42 + 512)
    pos_rec = raw[24 + 2 * 1264:24 + 2 * 1264 + 16 + 42]
    assert pos_rec[8:12] == (554).to_bytes(4, "little")                       # caplen of a position packet
    assert pos_rec[16 + 34:16 + 38] == bytes([0x20, 0x74, 0x20, 0x74])        # ports 8308 -> 8308
    assert pos_rec[16 + 38:16 + 40] == bytes([0x02, 0x08])                    # UDP length 520


def test_hdlmanager_frame_store_semantics(tmp_path):
    """veloslam::HDLManager as the store a consumer pulls from (HDLManager.cxx:226-260 over
    TimeLine.h): sorted by stamp whatever the arrival order, exact / nearest lookups (a tie goes to
    the later frame, the ends clamp), the inclusive range, a repeated stamp overwrites, the cache
    clears the oldest unreferenced arrivals and puts held ones back, waitForFrame times out or hands
    over what a producer thread added.  No GPU: frames are in memory already."""
    exe = build_exe()
    out = subprocess.run([exe, "--hdl-store"], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, VELO_TMP=str(tmp_path)))
    assert out.returncode == 0, out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    assert lines["empty"] == "0 0 0" and lines["count"] == "4" and lines["order"] == "100 200 300 500"
    assert lines["recent"] == "500" and lines["at"] == "1 0"
    #                        t = -50 100 149 150 151 349 400 401 9000
    assert lines["near"] == "100 100 100 200 200 300 500 500 500"
    assert lines["range"] == "200 300 500"                  # nearest(160) .. nearest(420), both included
    assert lines["overwrite"] == "4 7"
    assert lines["cached"] == "3 in_memory 0 1 0 1 1"       # arrivals 30 10 20 50 40, capacity 3: 30 and 10 cleared
    assert lines["held"] == "1 0 0 count 1" and lines["released"] == "1 0 count 0" and lines["gone"] == "0"
    assert lines["wait"] == "1 42 0"
    # .hdlmeta (HDLFrame.cxx:160-190) / .insmeta (type_defs.cxx:4-33) round trip: 132 bytes per frame
    m = lines["meta"].split()
    assert m[:4] == ["1", str(3 * 132), "3", "3"]
    assert m[4:] == ["%d/1000/%d/%d/1/%g/-7.25/0.125/%d/0.5" % (1000 + 100 * k, 24 + 1264 * 300 * k, 3 * k, 1.5 * k, 999 + k)
                     for k in range(3)]
    assert lines["meta_missing"] == "0 0" and lines["meta_unbound"] == "0"
    raw = open(os.path.join(str(tmp_path), "s.hdlmeta"), "rb").read()
    assert int.from_bytes(raw[132 + 16:132 + 24], "little") == 24 + 1264 * 300 and raw[132 + 24:132 + 32] == bytes(8)
    assert raw[132 + 32] == 3 and raw[132 + 33] == 1
    assert os.path.getsize(os.path.join(str(tmp_path), "s.insmeta")) == 3 * 98
    # the debug dumps (HDLFrame.cxx:36-125): names carry boost's to_iso_string of the stamp
    assert lines["dump"] == "1 20160704T080000.123456 19700101T000000 20000229T000000"
    d = str(tmp_path)
    pts = open(os.path.join(d, "20160704T080000.123456-points.txt")).read().split("\n")
    assert pts[0].split("\t") == ["1.5", "1.5", "1.5", "10"] and len(pts) == 6
    meta = open(os.path.join(d, "20160704T080000.123456-pointsMeta.txt")).read().split("\n")
    assert meta[1] == "35999\t12.5\t0\t0\t0"
    pcd = open(os.path.join(d, "20160704T080000.123456-2.pcd")).read().split("\n")
    assert pcd[2] == "FIELDS x y z intensity" and pcd[6] == "WIDTH 3" and pcd[9] == "POINTS 3" and pcd[10] == "DATA ascii"
    assert pcd[11] == "3 3 3 30" and pcd[13] == "5 5 5 50"
    allb = open(os.path.join(d, "20160704T080000.123456--1.pcd")).read().split("\n")
    assert allb[6] == "WIDTH 5"           # (3 beams here: "all" = beams 0..62 of however many there are)


@pytest.mark.gpu
def test_hdlmanager_offline_frames_equal_the_parser_reread(tmp_path, oracle):
    """HDLManager::loadOffline + getFrameAt (HDLManager.cxx:98-112, 207-224, 246-249): every frame
    of a capture whose revolutions split in the MIDDLE of a packet, prepared through the C++ class
    (GPU decode + compensation behind it), equals the parser restatement's re-read from that index
    entry with its skip (HDLParser::getFrame, HDLParser.cxx:505-544) bit for bit -- points,
    intensity, azimuth, distance, beam offsets, the car pose; the stub's pose is the track
    interpolated at the frame's stamp."""
    sc, mo, cal = synth.Scene(), synth.Motion(), synth.hdl64_calibration()
    pk, ts = [], []
    for f in range(3):                      # az_start 1.4 deg: the wrap falls on block 10 of a packet
        p, t, _ = synth.make_frame_packets(sc, mo, 3 + f, cal, az_start=140, seed=42)
        pk += p
        ts += t
    pk, ts = pk[:760], ts[:760]
    d = str(tmp_path)
    capi.pcap_write(os.path.join(d, "drive.pcap"), pk, ts)
    drive.write_carposes(os.path.join(d, "carposes.txt"), mo.ins_track(ts[0], ts[-1]))
    drive.write_db_xml(os.path.join(d, "db.xml"), cal)
    exe = build_exe()
    out = subprocess.run([exe, "--hdl", d, os.path.join(d, "frames.bin")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    assert lines["nocalib"] == "Corrections have not been set"
    dd = drive.load(d)
    idx = dd["index"]
    assert lines["frames"].split()[:2] == ["0", str(len(idx))] and int(lines["frames"].split()[3]) == dd["n_poses"]
    assert len(idx) == 3 and idx[1].firing_skip == 10
    tl = oracle.Timeline()
    for i in range(dd["n_poses"]):
        q = dd["poses"][i]
        tl.add(list(q.T), list(q.R), list(q.V), q.t_us, q.seconds_pos)
    raw = open(os.path.join(d, "frames.bin"), "rb").read()
    off = 0
    times = dd["times"]
    for k, e in enumerate(idx):
        head = np.frombuffer(raw, np.int64, 6, off); off += 48
        beams = np.frombuffer(raw, np.int32, 65, off); off += 260
        pose = np.frombuffer(raw, np.float64, 12, off); off += 96
        n = int(head[5])
        arr = []
        for dt in (np.float32, np.float32, np.float32, np.float32, np.uint16, np.float32):
            arr.append(np.frombuffer(raw, dt, n, off)); off += n * np.dtype(dt).itemsize
        end = idx[k + 1].first_packet + 1 if k + 1 < len(idx) else len(pk)
        assert list(head[:5]) == [e.t_us + drive.EIGHT_H_US, e.file_pos, e.firing_skip, e.first_packet, end - e.first_packet]
        re = oracle.Decoder(cal, timeline=tl)
        re.set_skip(e.firing_skip)
        for p, t in zip(pk[e.first_packet:], times[e.first_packet:]):
            re.packet(p, int(t))
            if re.num_frames:
                break
        if not re.num_frames:
            re.flush()
        want = re.frame_cloud(0)
        assert n == want[0].size and n > 50_000
        for got, w in zip(arr, want):
            assert np.array_equal(got.view(np.uint8), np.ascontiguousarray(w).view(np.uint8))
        sizes = [re.beam(0, b)[0].size for b in range(64)]
        assert list(np.diff(beams)) == sizes
        car, _, _ = re.carpose(0)
        assert list(pose[6:9]) == list(car.T) and list(pose[9:12]) == list(car.R)
        ok, stub = tl.interpolate(int(e.t_us + drive.EIGHT_H_US))
        assert ok and list(pose[0:3]) == list(stub.T) and list(pose[3:6]) == list(stub.R)
    assert off == len(raw)
    assert lines["in_memory"].split()[0] == "2"               # capacity 2: the first frame's points were cleared ...
    assert int(lines["in_memory"].split()[2]) > 50_000        # ... and come back by decoding it again
    assert int(lines["resident"]) > 50_000
    assert lines["meta_reload"].split()[0] == "3" and int(lines["meta_reload"].split()[1]) == n   # the last frame again


@pytest.mark.gpu
def test_cpp_stream_driver_rolls_the_device_map_and_matches_python(tmp_path):
    """MapManager::registerResident on the rolling device map.  A drive whose ROI
    rectangle loses a tile column on the way: the C++ driver (include/veloslam/*.hpp) must apply it
    incrementally -- no full build after the first, last_update == 1 -- and register every frame
    where the Python replay (same C ABI calls) does."""
    sc = synth.Scene()
    world = sc.sample_map(400_000)
    # frames 70..77: the car passes x = 45 m, where the tile column [-65, -55) leaves the +-100 m square
    drive.export_synthetic(str(tmp_path), n_frames=8, patch_range=10.0, first_frame=70, world_xyz=world)
    exe = build_driver()
    out = subprocess.run([exe, str(tmp_path), "--steps", "12", "--warmup", "1", "--threshold", "64"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    cpp = json.loads(out.stdout.strip().splitlines()[-1])
    assert cpp["worst_pose_error_m"] < 0.02
    assert cpp["map"]["full_builds"] == 0 and cpp["map"]["rolls"] >= 1 and cpp["map"]["points_evicted"] > 0
    assert cpp["last_update"] == 1 and cpp["map"]["increment_flushes"] >= 1
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    py = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stream", "--drive", str(tmp_path),
                         "--steps", "12", "--warmup", "1", "--append-threshold", "64", "--roll-lead", "4"],
                        capture_output=True, text=True, timeout=600, env=env)   # (the C++ driver's default lead)
    assert py.returncode == 0, py.stderr[-2000:]
    rec = json.loads(py.stdout.strip().splitlines()[-1])
    assert rec["worst_pose_error_m"] < 0.02 and rec["map"]["full_builds"] == 0 and rec["map"]["rolls"] == cpp["map"]["rolls"]
    assert rec["map"]["points_evicted"] == cpp["map"]["points_evicted"]
    assert rec["map_points_mean"] == cpp["map_points"]            # the two hosts left the same map behind
    assert abs(rec["worst_pose_error_m"] - cpp["worst_pose_error_m"]) < 1e-9
    # the pipelining changes WHEN things run, not what comes out: with the next frame decoded on the
    # second stream but the map rolled when due, the drive replays exactly as with nothing overlapped
    # (same poses to the last digit printed, same map); rolling ahead as well only defers the pending
    # increments past the roll (they join at the next flush), which moves the poses by micrometres
    runs = {}
    for flags in (["--no-roll-ahead"], ["--no-overlap"]):
        o = subprocess.run([exe, str(tmp_path), "--steps", "12", "--warmup", "1", "--threshold", "64"] + flags,
                           capture_output=True, text=True, timeout=600)
        assert o.returncode == 0, o.stderr[-2000:]
        runs[flags[0]] = json.loads(o.stdout.strip().splitlines()[-1])
    a, b = runs["--no-roll-ahead"], runs["--no-overlap"]
    assert a["worst_pose_error_m"] == b["worst_pose_error_m"] and a["map_points"] == b["map_points"]
    assert a["map"]["points_evicted"] == b["map"]["points_evicted"] and a["map"]["rolls"] == b["map"]["rolls"]
    assert a["decode_planned_ahead"] and not b["decode_planned_ahead"] and a["map"]["rolls_ahead"] == 0
    assert cpp["roll_ahead"] and cpp["map"]["rolls_ahead"] + cpp["map"]["rolls_refused"] >= 1
    assert abs(cpp["worst_pose_error_m"] - b["worst_pose_error_m"]) < 1e-4 and cpp["map"]["rolls"] == b["map"]["rolls"]


@pytest.mark.gpu
def test_cpp_stream_driver_maps_a_drive_from_its_own_frames(tmp_path):
    """Round 6, configs[2] as SLAM from the C++ host (veloslam::MapManager::seedFromFrame + RegisterOptions::integrate /
    increments_in_roi_only / pipeline_increments): no world.map -- the map is seeded with frame 0 and grows from the accepted
    increments.  Every frame updates the map; the pipelined schedule gives the same poses, map and increments whichever
    way its host calls are ordered and whichever stream the roll runs on (nothing depends on timing); the synchronous
    integration is another schedule (each registration sees one frame's increment more) and stays within centimetres of it."""
    import torch
    drive.export_mapping_drive(str(tmp_path), n_frames=26, device=torch.device("cuda:0"))
    torch.cuda.synchronize()
    exe = build_driver()

    def run(extra=(), env_extra=None):
        env = dict(os.environ)
        for k in ("VELO_UPDATE_BEFORE_START", "VELO_ROLL_LIGHT_MAX", "VELO_NO_PAIR_CERT"):
            env.pop(k, None)
        env.update(env_extra or {})
        o = subprocess.run([exe, str(tmp_path), "--mapping", "--steps", "20", "--warmup", "4", "--threshold", "1"] + list(extra),
                           capture_output=True, text=True, timeout=600, env=env)
        assert o.returncode == 0, o.stderr[-2000:]
        return json.loads(o.stdout.strip().splitlines()[-1])

    def sig(r):
        return (r["worst_pose_error_m"], r["mean_pose_error_m"], r["map_points"], r["increment_points_per_frame"],
                r["map_updates"], r["map"]["points_evicted"], r["map"]["increment_points"])

    a = run()
    assert a["mode"].startswith("mapping, increments integrated in pipeline")
    assert a["map_updates"] >= 20 and a["map_updates_beside_registration"] >= 19
    assert a["increment_points_per_frame"] > 2000 and a["worst_pose_error_m"] < 0.05
    assert a["map"]["full_builds"] == 0 and a["map"]["rolls_refused"] == 0
    assert a["map_points"] > 150_000                                  # the seed frame and what 24 frames added to it
    for env_extra in ({"VELO_UPDATE_BEFORE_START": "1"}, {"VELO_ROLL_LIGHT_MAX": "-1"}, {"VELO_NO_PAIR_CERT": "1"}):
        assert sig(run(env_extra=env_extra)) == sig(a), env_extra
    s = run(extra=["--no-pipeline"])
    assert s["mode"].startswith("mapping, increments integrated synchronously")
    assert s["worst_pose_error_m"] < 0.05 and s["increment_points_per_frame"] > 2000
    assert abs(s["worst_pose_error_m"] - a["worst_pose_error_m"]) < 0.03
