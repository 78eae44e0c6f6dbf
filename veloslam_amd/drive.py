"""A recorded drive on disk, in the reference's own file formats, and how to get one.

    drive.pcap     the LiDAR packets (vtkPacketFileWriter.cxx:118-161; velo_pcap_write / _read / _index)
    carposes.txt   the pose track of HDLManager::loadOffline (HDLManager.cxx:103-117), rows
                   "x y yaw roll pitch v sec usec" (TransformManager.cxx:95-125: radians, yaw
                   clockwise, NO z and no velocity vector); velo_carposes_read
    db.xml         the sensor calibration (HDLParser.cxx:771-858); velo_load_corrections
    world.map      MapManager's stream format (MapManager.cxx:81-110 + MapPatch.cxx:3-69) with the
                   point payload of veloslam::MapManager::save / load
    drive.json     what the formats cannot carry: the z of the first pose, ROI / voxel parameters,
                   and (synthetic drives only) the true pose of every frame

`export_synthetic` writes such a directory from the synthetic generator (veloslam_amd/synth.py);
a real capture + its carposes.txt drop in the same way.  `load` reads one back through the C ABI
only -- the same calls tools/stream_driver.cpp makes from C++.
"""
import json
import math
import os
import struct

import numpy as np

from . import capi, synth

EIGHT_H_US = 8 * 3600 * 1_000_000  # timevalToPtime (type_defs.cxx:69-72): added to pose AND packet stamps


def write_db_xml(path, cal, enabled=64, shuffle=False):
    """A calibration file in Velodyne's boost-serialisation layout (the element names are all
    HDLParser::loadCorrectionsFile looks at, HDLParser.cxx:771-858); distances in centimetres."""
    ids = list(range(64))
    if shuffle:
        ids = ids[::-1]
    px = []
    for i in ids:
        r = cal[i]
        px.append("""\t\t<item class_id="2" tracking_level="0" version="1">
\t\t\t<px class_id="3" tracking_level="1" version="1" object_id="_%d">
\t\t\t\t<id_>%d</id_>
\t\t\t\t<rotCorrection_>%r</rotCorrection_>
\t\t\t\t<vertCorrection_>%r</vertCorrection_>
\t\t\t\t<distCorrection_>%r</distCorrection_>
\t\t\t\t<distCorrectionX_>0</distCorrectionX_>
\t\t\t\t<vertOffsetCorrection_>%r</vertOffsetCorrection_>
\t\t\t\t<horizOffsetCorrection_>%r</horizOffsetCorrection_>
\t\t\t\t<focalDistance_>0</focalDistance_>
\t\t\t</px>
\t\t</item>""" % (i, i, float(r[0]), float(r[1]), float(r[2]) * 100.0, float(r[3]) * 100.0, float(r[4]) * 100.0))
    en = "\n".join("\t\t<item>%d</item>" % (1 if i < enabled else 0) for i in range(64))
    with open(path, "w") as f:
        f.write("""<?xml version="1.0" encoding="UTF-8" standalone="yes" ?>
<!DOCTYPE boost_serialization>
<boost_serialization signature="serialization::archive" version="4">
<DB class_id="0" tracking_level="1" version="0" object_id="_0">
\t<distLSB_>0.2</distLSB_>
\t<enabled_ class_id="4" tracking_level="0" version="0">
\t\t<count>64</count>
%s
\t</enabled_>
\t<points_ class_id="1" tracking_level="0" version="0">
\t\t<count>64</count>
\t\t<item_version>1</item_version>
%s
\t</points_>
</DB>
</boost_serialization>
""" % (en, "\n".join(px)))


def tile_index(x, y, patch_range):
    """MapManager::getPatchIdx of the build (tiles centred on multiples of the edge,
    MapManager.cxx:47-52 getMapCenter): doubles, as the C++ side computes it"""
    r = float(patch_range)
    return (np.floor((np.asarray(x, np.float64) + r / 2) / r).astype(np.int64),
            np.floor((np.asarray(y, np.float64) + r / 2) / r).astype(np.int64))


def write_map_file(path, x, y, z, patch_range):
    """veloslam::MapManager::save's layout: points binned into square tiles, tiles in (i, j) key
    order (std::map<pair<int,int>> iterates i-major), each tile's points in input order."""
    x, y, z = (np.ascontiguousarray(a, np.float32) for a in (x, y, z))
    ti, tj = tile_index(x, y, patch_range)
    order = np.lexsort((np.arange(x.size), tj, ti))          # i-major, then j, stable in input order
    ti, tj = ti[order], tj[order]
    key = np.stack([ti, tj], 1)
    starts = np.flatnonzero(np.r_[True, np.any(key[1:] != key[:-1], axis=1)])
    ends = np.r_[starts[1:], x.size]
    if starts.size > 65535:
        raise ValueError("more than 65535 tiles: the reference header counts them in a u16")
    r = float(patch_range)
    cx = ti[starts].astype(np.float64) * r
    cy = tj[starts].astype(np.float64) * r
    x0, x1, y0, y1 = (cx - r / 2).min(), (cx + r / 2).max(), (cy - r / 2).min(), (cy + r / 2).max()
    with open(path, "wb") as f:
        f.write(struct.pack("<ddffH", 0.5 * (x0 + x1), 0.5 * (y0 + y1), np.float32(max(x1 - x0, y1 - y0)),
                            np.float32(r), starts.size))
        for a, b, px, py in zip(starts, ends, cx, cy):
            sel = order[a:b]
            f.write(struct.pack("<ddf4HQ", px, py, np.float32(r), 0, 0, 0, 0, b - a))
            f.write(x[sel].tobytes())
            f.write(y[sel].tobytes())
            f.write(z[sel].tobytes())
    return int(starts.size)


def read_map_file(path):
    """-> (patch_range, list of (centerX, centerY, x, y, z)) in file order"""
    with open(path, "rb") as f:
        raw = f.read()
    cx, cy, rng, pr, n = struct.unpack_from("<ddffH", raw, 0)
    off = struct.calcsize("<ddffH")
    tiles = []
    for _ in range(n):
        px, py, r, a, b, c, d, m = struct.unpack_from("<ddf4HQ", raw, off)
        off += struct.calcsize("<ddf4HQ")
        arrs = []
        for _k in range(3):
            arrs.append(np.frombuffer(raw, np.float32, m, off).copy())
            off += 4 * m
        tiles.append((px, py, arrs[0], arrs[1], arrs[2]))
    return float(pr), tiles


def write_carposes(path, samples):
    """samples: (T[3], Rdeg[3], V[3], t_us) as synth.Motion.ins_track yields them.  The format keeps
    x, y, the three angles (radians, yaw sign flipped) and a scalar speed; 17 significant digits so
    that the doubles survive the text."""
    with open(path, "w") as f:
        for T, R, V, t in samples:
            f.write("%.17g %.17g %.17g %.17g %.17g %.17g %d %d\n" % (
                T[0], T[1], -math.radians(R[2]), math.radians(R[0]), math.radians(R[1]),
                float(np.linalg.norm(V)), t // 1_000_000, t % 1_000_000))


def export_synthetic(out_dir, n_frames=64, world_points=12_000_000, patch_range=10.0, first_frame=3,
                     world_xyz=None, voxel=1.0, k_normals=16):
    """A synthetic drive of n_frames consecutive revolutions (1 m apart) through the synthetic scene."""
    os.makedirs(out_dir, exist_ok=True)
    sc, mo, cal = synth.Scene(), synth.Motion(), synth.hdl64_calibration()
    packets, times, truth = [], [], []
    for k in range(n_frames):
        pk, ts, _ = synth.make_frame_packets(sc, mo, first_frame + k, cal, seed=42)
        packets += pk
        times += ts
        T, R, V = mo.pose(ts[0])
        truth.append([float(T[0]), float(T[1]), float(T[2])])
    capi.pcap_write(os.path.join(out_dir, "drive.pcap"), packets, times)
    write_carposes(os.path.join(out_dir, "carposes.txt"), mo.ins_track(times[0], times[-1]))
    write_db_xml(os.path.join(out_dir, "db.xml"), cal)
    if world_xyz is None:
        world_xyz = sc.sample_map(world_points)
    n_tiles = write_map_file(os.path.join(out_dir, "world.map"), *world_xyz, patch_range)
    meta = dict(n_frames=n_frames, z0=truth[0][2], patch_range=patch_range, voxel=voxel, k_normals=k_normals,
                world_points=int(np.asarray(world_xyz[0]).size), tiles=n_tiles, true_positions=truth,
                note="synthetic: veloslam_amd/synth.py scene, constant 10 m/s, 5 deg/s yaw; carposes.txt has no z")
    with open(os.path.join(out_dir, "drive.json"), "w") as f:
        json.dump(meta, f)
    with open(os.path.join(out_dir, "truth.txt"), "w") as f:   # (for hosts without a JSON parser)
        f.write("%.17g %g %g %d %g\n" % (truth[0][2], patch_range, voxel, k_normals, 0.0))
        for t in truth:
            f.write("%.17g %.17g %.17g\n" % tuple(t))
    return meta


def export_mapping_drive(out_dir, n_frames=640, device="cpu", patch_range=10.0, voxel=1.0, k_normals=16, speed=10.0,
                         scene_length=None):
    """A drive to MAP (BASELINE configs[2] as SLAM): n_frames consecutive revolutions, `speed` / 10 m apart, straight
    down synth.LongScene -- a street longer than the drive, so that every frame sees ground no frame saw before.  No
    world.map is written: the map starts from the first frame and grows from accepted increments only
    (tools/stream_driver --mapping).  The ray casting and the packet assembly run in torch on `device`."""
    os.makedirs(out_dir, exist_ok=True)
    length = float(scene_length) if scene_length else speed * 0.1 * n_frames + 150.0
    sc = synth.LongScene(length)
    mo = synth.Motion(p0=(0.0, 0.0, synth.SENSOR_HEIGHT), speed=speed)
    cal = synth.hdl64_calibration()
    pk, ts = synth.make_frame_packets_device(sc, mo, list(range(n_frames)), cal, device)
    buf = np.ascontiguousarray(pk.reshape(-1, 1206).cpu().numpy())
    times = np.ascontiguousarray(ts.reshape(-1), dtype=np.int64)
    rc = capi.lib().velo_pcap_write(os.path.join(out_dir, "drive.pcap").encode(), capi._p(buf), capi._p(times), buf.shape[0])
    if rc:
        raise capi.VeloError(rc, "velo_pcap_write")
    write_carposes(os.path.join(out_dir, "carposes.txt"), mo.ins_track(int(times[0]), int(times[-1])))
    write_db_xml(os.path.join(out_dir, "db.xml"), cal)
    truth = [[float(v) for v in mo.pose(int(ts[k][0]))[0]] for k in range(n_frames)]
    meta = dict(n_frames=n_frames, z0=truth[0][2], patch_range=patch_range, voxel=voxel, k_normals=k_normals,
                world_points=0, tiles=0, true_positions=truth, scene_length=length,
                note="synthetic, to be mapped: veloslam_amd/synth.py LongScene, constant %g m/s, 5 deg/s yaw; no world.map: "
                     "the map is seeded with frame 0 and grown from accepted increments" % speed)
    with open(os.path.join(out_dir, "drive.json"), "w") as f:
        json.dump(meta, f)
    with open(os.path.join(out_dir, "truth.txt"), "w") as f:
        f.write("%.17g %g %g %d %g\n" % (truth[0][2], patch_range, voxel, k_normals, 0.0))
        for t in truth:
            f.write("%.17g %.17g %.17g\n" % tuple(t))
    return meta


def load(drive_dir):
    """Everything a replay needs, through the C ABI (velo_pcap_read, velo_pcap_index,
    velo_carposes_read, velo_load_corrections).  Packet stamps get the reference's + 8 h so that
    they live on the pose track's clock (TransformManager.cxx:117: timevalToPtime on both)."""
    import ctypes as C
    p = lambda n: os.path.join(drive_dir, n)  # noqa: E731
    pk, t = capi.pcap_read(p("drive.pcap"))
    idx = capi.pcap_index(p("drive.pcap"))
    n = C.c_size_t()
    L = capi.lib()
    L.velo_carposes_read.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    if L.velo_carposes_read(p("carposes.txt").encode(), None, 0, C.byref(n)):
        raise RuntimeError("cannot read carposes.txt")
    poses = (capi.Pose * max(n.value, 1))()
    if L.velo_carposes_read(p("carposes.txt").encode(), poses, n.value, C.byref(n)):
        raise RuntimeError("cannot read carposes.txt")
    corr, n_enabled = capi.load_corrections(p("db.xml"))
    meta = json.load(open(p("drive.json"))) if os.path.exists(p("drive.json")) else {}
    return dict(packets=np.frombuffer(b"".join(pk), np.uint8).copy(), times=np.asarray(t, np.int64) + EIGHT_H_US,
                index=idx, poses=poses, n_poses=n.value, calib=corr, meta=meta, dir=drive_dir)
