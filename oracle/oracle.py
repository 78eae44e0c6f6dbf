"""ctypes doorway to oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  The product package (veloslam_amd) never
does.  See oracle/velo_oracle.h for what each entry point restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

VO_TIME_INVALID = -(2 ** 63)


def build(force=False):
    """Compile the C restatement (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
        for f in ("geodesy.c", "pose.c", "decode.c", "icp.c", "velo_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _LIB


class Pose(C.Structure):
    _fields_ = [
        ("T", C.c_double * 3),
        ("R", C.c_double * 3),
        ("V", C.c_double * 3),
        ("t_us", C.c_int64),
        ("week_number", C.c_uint16),
        ("milliseconds", C.c_uint32),
        ("week_number_pos", C.c_uint32),
        ("seconds_pos", C.c_double),
    ]

    def as_tuple(self):
        return (tuple(self.T), tuple(self.R), tuple(self.V), self.t_us, self.seconds_pos)


class LaserCorr(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "azimuthCorrection", "verticalCorrection", "distanceCorrection",
        "verticalOffsetCorrection", "horizontalOffsetCorrection",
        "sinVertCorrection", "cosVertCorrection",
        "sinVertOffsetCorrection", "cosVertOffsetCorrection")]


class IcpStat(C.Structure):
    _fields_ = [("n_pairs", C.c_uint32), ("rmse", C.c_double), ("candidates", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB)
    dp = C.POINTER(C.c_double)
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int32)
    for name in ("vo_llh2xyz", "vo_xyz2llh"):
        getattr(L, name).argtypes = [dp, dp]
    for name in ("vo_xyz2enu", "vo_enu2xyz", "vo_enu2llh", "vo_llh2enu"):
        getattr(L, name).argtypes = [dp, dp, dp]
    L.vo_eulr2dcm.argtypes = [dp, dp]
    L.vo_mapping_angle.argtypes = [C.c_double]
    L.vo_mapping_angle.restype = C.c_double
    pp = C.POINTER(Pose)
    L.vo_pose_init.argtypes = [pp]
    L.vo_pose_matrix.argtypes = [pp, dp]
    L.vo_matrix_to_TRdeg.argtypes = [dp, dp]
    L.vo_transform_point.argtypes = [dp, dp]
    L.vo_timeline_new.restype = C.c_void_p
    L.vo_timeline_free.argtypes = [C.c_void_p]
    L.vo_timeline_size.argtypes = [C.c_void_p]
    L.vo_timeline_size.restype = C.c_size_t
    L.vo_timeline_add.argtypes = [C.c_void_p, pp]
    L.vo_timeline_boundary.argtypes = [C.c_void_p, C.c_int64, pp, pp]
    L.vo_interpolate_transform.argtypes = [C.c_void_p, C.c_int64, pp]
    L.vo_time_to_week_milli.argtypes = [C.c_int64, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)]
    L.vo_time_to_week_milli.restype = None
    L.vo_compensate.argtypes = [fp, fp, fp, C.POINTER(C.c_uint16), C.c_size_t, dp, C.c_size_t,
                                fp, fp, fp]
    L.vo_load_corrections.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_int)]
    L.vo_decoder_new.argtypes = [C.POINTER(LaserCorr), C.c_int, C.c_void_p]
    L.vo_decoder_new.restype = C.c_void_p
    L.vo_decoder_free.argtypes = [C.c_void_p]
    L.vo_decoder_set_crop.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
    L.vo_decoder_set_skip.argtypes = [C.c_void_p, C.c_int]
    L.vo_decoder_set_laser_selection.argtypes = [C.c_void_p, C.c_char_p]
    L.vo_decoder_set_points_skip.argtypes = [C.c_void_p, C.c_int]
    L.vo_decoder_packet.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64]
    L.vo_decoder_flush.argtypes = [C.c_void_p]
    L.vo_decoder_num_frames.argtypes = [C.c_void_p]
    L.vo_frame_beam_size.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.vo_frame_beam_size.restype = C.c_size_t
    L.vo_frame_beam_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, fp, fp, fp,
                                     C.POINTER(C.c_uint16), fp]
    L.vo_frame_carpose.argtypes = [C.c_void_p, C.c_int, pp, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int)]
    L.vo_frame_num_packets.argtypes = [C.c_void_p, C.c_int]
    L.vo_frame_num_packets.restype = C.c_size_t
    L.vo_map_build.argtypes = [fp, fp, fp, C.c_size_t, C.c_float, C.c_int]
    L.vo_map_build.restype = C.c_void_p
    L.vo_map_build_ex.argtypes = [fp, fp, fp, C.c_size_t, C.c_float, C.c_int, C.c_int]
    L.vo_map_build_ex.restype = C.c_void_p
    L.vo_auto_subdiv.argtypes = [fp, fp, fp, C.c_size_t, C.c_float]
    L.vo_map_build_grid.argtypes = [fp, fp, fp, C.c_size_t, C.c_float, C.c_int, C.c_int, fp, ip]
    L.vo_map_build_grid.restype = C.c_void_p
    L.vo_roll_new.argtypes = [fp, fp, fp, C.c_size_t, C.c_float, C.c_int, C.c_int, C.c_int]
    L.vo_roll_new.restype = C.c_void_p
    L.vo_roll_new3.argtypes = [fp, fp, fp, C.c_size_t, C.c_float, C.c_int, C.c_int, ip]
    L.vo_roll_new3.restype = C.c_void_p
    L.vo_roll_free.argtypes = [C.c_void_p]
    L.vo_roll_map.argtypes = [C.c_void_p]
    L.vo_roll_map.restype = C.c_void_p
    L.vo_roll_size.argtypes = [C.c_void_p]
    L.vo_roll_size.restype = C.c_size_t
    L.vo_roll_append.argtypes = [C.c_void_p, fp, fp, fp, C.c_size_t]
    L.vo_roll_evict_outside.argtypes = [C.c_void_p, fp, fp]
    L.vo_roll_filter_sparse.argtypes = [C.c_void_p, fp, fp, fp, C.c_size_t, C.c_int, C.c_void_p]
    L.vo_roll_filter_sparse.restype = C.c_size_t
    L.vo_roll_append_sparse.argtypes = [C.c_void_p, fp, fp, fp, C.c_size_t, C.c_int]
    L.vo_roll_append_sparse.restype = C.c_long
    L.vo_roll_evict_region.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float, C.c_float]
    L.vo_map_free.argtypes = [C.c_void_p]
    L.vo_map_size.argtypes = [C.c_void_p]
    L.vo_map_subdiv.argtypes = [C.c_void_p]
    L.vo_map_size.restype = C.c_size_t
    L.vo_map_num_cells.argtypes = [C.c_void_p]
    L.vo_map_num_cells.restype = C.c_size_t
    L.vo_map_grid.argtypes = [C.c_void_p, fp, ip, fp]
    for nm in ("x", "y", "z", "nx", "ny", "nz"):
        f = getattr(L, "vo_map_" + nm)
        f.argtypes = [C.c_void_p]
        f.restype = fp
    for nm in ("perm", "cell_start"):
        f = getattr(L, "vo_map_" + nm)
        f.argtypes = [C.c_void_p]
        f.restype = ip
    L.vo_correspond.argtypes = [C.c_void_p, fp, fp, fp, C.c_size_t, dp, C.c_float, ip, fp]
    L.vo_correspond.restype = C.c_uint64
    L.vo_knn.argtypes = [C.c_void_p, fp, fp, fp, C.c_size_t, dp, C.c_float, C.c_int, ip, fp, ip]
    L.vo_accumulate.argtypes = [C.c_void_p, fp, fp, fp, C.c_size_t, dp, ip, dp]
    L.vo_solve_update.argtypes = [dp, dp, dp]
    L.vo_icp.argtypes = [C.c_void_p, fp, fp, fp, C.c_size_t, dp, C.c_int, C.c_float, dp,
                         C.POINTER(IcpStat), dp, C.c_int]
    L.vo_increment.argtypes = [C.c_void_p, fp, fp, fp, C.c_size_t, dp, C.c_int, fp, fp, fp]
    L.vo_increment.restype = C.c_size_t
    _lib = L
    return L


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ geodesy
def _g2(name, a):
    a = np.ascontiguousarray(a, dtype=np.float64).copy()
    o = np.zeros(3)
    getattr(lib(), name)(_d(a), _d(o))
    return o


def _g3(name, a, org):
    a = np.ascontiguousarray(a, dtype=np.float64).copy()
    org = np.ascontiguousarray(org, dtype=np.float64).copy()
    o = np.zeros(3)
    getattr(lib(), name)(_d(a), _d(org), _d(o))
    return o


def llh2xyz(a): return _g2("vo_llh2xyz", a)
def xyz2llh(a): return _g2("vo_xyz2llh", a)
def xyz2enu(a, org): return _g3("vo_xyz2enu", a, org)
def enu2xyz(a, org): return _g3("vo_enu2xyz", a, org)
def enu2llh(a, org): return _g3("vo_enu2llh", a, org)
def llh2enu(a, org): return _g3("vo_llh2enu", a, org)


def eulr2dcm(e):
    e = np.ascontiguousarray(e, dtype=np.float64).copy()
    o = np.zeros(9)
    lib().vo_eulr2dcm(_d(e), _d(o))
    return o.reshape(3, 3)


def mapping_angle(a):
    return lib().vo_mapping_angle(float(a))


# --------------------------------------------------------------------- poses
def make_pose(T=(0, 0, 0), R=(0, 0, 0), V=(0, 0, 0), t_us=VO_TIME_INVALID, seconds_pos=-1.0):
    p = Pose()
    lib().vo_pose_init(C.byref(p))
    for i in range(3):
        p.T[i], p.R[i], p.V[i] = float(T[i]), float(R[i]), float(V[i])
    p.t_us = int(t_us)
    p.seconds_pos = float(seconds_pos)
    return p


def pose_matrix(T, Rdeg):
    p = make_pose(T, Rdeg)
    M = np.zeros(12)
    lib().vo_pose_matrix(C.byref(p), _d(M))
    return M


def matrix_to_TRdeg(M):
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(12).copy()
    o = np.zeros(6)
    lib().vo_matrix_to_TRdeg(_d(M), _d(o))
    return o


class Timeline:
    def __init__(self):
        self.h = lib().vo_timeline_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().vo_timeline_free(self.h)
            self.h = None

    def __len__(self):
        return lib().vo_timeline_size(self.h)

    def add(self, T, R, V, t_us, seconds_pos=0.0):
        p = make_pose(T, R, V, t_us, seconds_pos)
        lib().vo_timeline_add(self.h, C.byref(p))

    def boundary(self, t_us):
        f, b = Pose(), Pose()
        n = lib().vo_timeline_boundary(self.h, int(t_us), C.byref(f), C.byref(b))
        return n, f, b

    def interpolate(self, t_us):
        out = Pose()
        lib().vo_pose_init(C.byref(out))
        ok = lib().vo_interpolate_transform(self.h, int(t_us), C.byref(out))
        return bool(ok), out


def time_to_week_milli(t_us):
    w, m = C.c_uint16(0), C.c_uint32(0)
    lib().vo_time_to_week_milli(int(t_us), C.byref(w), C.byref(m))
    return w.value, m.value


def compensate(x, y, z, pkt, table):
    x, y, z = _f32(x), _f32(y), _f32(z)
    pkt = np.ascontiguousarray(pkt, dtype=np.uint16)
    table = np.ascontiguousarray(table, dtype=np.float64).reshape(-1, 12)
    n = x.size
    ox, oy, oz = (np.empty(n, np.float32) for _ in range(3))
    lib().vo_compensate(_f(x), _f(y), _f(z), pkt.ctypes.data_as(C.POINTER(C.c_uint16)), n,
                        _d(table), table.shape[0], _f(ox), _f(oy), _f(oz))
    return ox, oy, oz


# -------------------------------------------------------------------- decode
class Decoder:
    def __init__(self, corr, n_lasers=64, timeline=None):
        """corr: (64, 9) float64 rows in LaserCorr field order."""
        corr = np.ascontiguousarray(corr, dtype=np.float64).reshape(64, 9)
        self._tl = timeline
        self.h = lib().vo_decoder_new(corr.ctypes.data_as(C.POINTER(LaserCorr)), n_lasers,
                                      timeline.h if timeline is not None else None)

    def __del__(self):
        if getattr(self, "h", None):
            lib().vo_decoder_free(self.h)
            self.h = None

    def set_crop(self, enable, inside, region):
        r = np.ascontiguousarray(region, dtype=np.float64)
        lib().vo_decoder_set_crop(self.h, int(enable), int(inside), _d(r))

    def set_skip(self, s):
        lib().vo_decoder_set_skip(self.h, int(s))

    def set_laser_selection(self, sel):
        lib().vo_decoder_set_laser_selection(self.h, bytes(bytearray(int(bool(v)) for v in sel)))

    def set_points_skip(self, s):
        lib().vo_decoder_set_points_skip(self.h, int(s))

    def packet(self, data, t_us):
        return lib().vo_decoder_packet(self.h, bytes(data), len(data), int(t_us))

    def flush(self):
        return lib().vo_decoder_flush(self.h)

    @property
    def num_frames(self):
        return lib().vo_decoder_num_frames(self.h)

    def beam(self, frame, b):
        n = lib().vo_frame_beam_size(self.h, frame, b)
        x, y, z, it, dist = (np.empty(n, np.float32) for _ in range(5))
        az = np.empty(n, np.uint16)
        if n:
            lib().vo_frame_beam_copy(self.h, frame, b, _f(x), _f(y), _f(z), _f(it),
                                     az.ctypes.data_as(C.POINTER(C.c_uint16)), _f(dist))
        return x, y, z, it, az, dist

    def frame_cloud(self, frame, start=0, end=64):
        """HDLFrame::getPointsAsOneCloud ordering (HDLFrame.cxx:127-144)."""
        parts = [self.beam(frame, b) for b in range(start, end)]
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(6))

    def carpose(self, frame):
        p = Pose()
        t = C.c_int64()
        s = C.c_int()
        lib().vo_frame_carpose(self.h, frame, C.byref(p), C.byref(t), C.byref(s))
        return p, t.value, s.value

    def num_packets(self, frame):
        return lib().vo_frame_num_packets(self.h, frame)


# ----------------------------------------------------------------------- ICP
class Map:
    def __init__(self, x, y, z, voxel=1.0, k_normals=16, subdiv=3, origin=None, dims_min=None):
        x, y, z = _f32(x), _f32(y), _f32(z)
        self._owned = True
        if int(subdiv) == 0:  # chosen from the density
            subdiv = lib().vo_auto_subdiv(_f(x), _f(y), _f(z), x.size, float(voxel))
        if origin is None and dims_min is None:
            self.h = lib().vo_map_build_ex(_f(x), _f(y), _f(z), x.size, float(voxel),
                                           int(k_normals), int(subdiv))
        else:
            o = None if origin is None else _f(np.ascontiguousarray(origin, np.float32))
            dm = None if dims_min is None else _i(np.ascontiguousarray(dims_min, np.int32))
            self.h = lib().vo_map_build_grid(_f(x), _f(y), _f(z), x.size, float(voxel),
                                             int(k_normals), int(subdiv), o, dm)
        self.subdiv = int(subdiv)
        if not self.h:
            raise ValueError("vo_map_build failed")
        self.n = x.size
        self.voxel = float(voxel)

    @classmethod
    def _view(cls, handle, owner):
        """Non-owning view of a vo_map that lives inside `owner` (a RollingMap)."""
        m = cls.__new__(cls)
        m.h = handle
        m._owned = False
        m._owner = owner
        m.n = lib().vo_map_size(handle)
        m.subdiv = lib().vo_map_subdiv(handle)
        o, d, ih = m.grid()
        m.voxel = 1.0 / ih
        return m

    def __del__(self):
        if getattr(self, "h", None) and getattr(self, "_owned", False):
            lib().vo_map_free(self.h)
        self.h = None

    def _arr(self, nm, n, dt):
        p = getattr(lib(), "vo_map_" + nm)(self.h)
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True)

    @property
    def ncell(self):
        return lib().vo_map_num_cells(self.h)

    def grid(self):
        o = np.zeros(3, np.float32)
        d = np.zeros(3, np.int32)
        ih = C.c_float()
        lib().vo_map_grid(self.h, _f(o), _i(d), C.byref(ih))
        return o, d, ih.value

    def sorted_xyz(self):
        return tuple(self._arr(k, self.n, np.float32) for k in "xyz")

    def normals(self):
        return tuple(self._arr(k, self.n, np.float32) for k in ("nx", "ny", "nz"))

    def perm(self):
        return self._arr("perm", self.n, np.int32)

    def cell_start(self):
        return self._arr("cell_start", self.ncell + 1, np.int32)

    def correspond(self, x, y, z, T, d_max):
        x, y, z = _f32(x), _f32(y), _f32(z)
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
        corr = np.empty(x.size, np.int32)
        d2 = np.empty(x.size, np.float32)
        cand = lib().vo_correspond(self.h, _f(x), _f(y), _f(z), x.size, _d(T), float(d_max),
                                   _i(corr), _f(d2))
        return corr, d2, int(cand)

    def knn(self, x, y, z, T, d_max, k):
        x, y, z = _f32(x), _f32(y), _f32(z)
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
        idx = np.empty((x.size, k), np.int32)
        d2 = np.empty((x.size, k), np.float32)
        cnt = np.empty(x.size, np.int32)
        lib().vo_knn(self.h, _f(x), _f(y), _f(z), x.size, _d(T), float(d_max), int(k), _i(idx),
                     _f(d2), _i(cnt))
        return idx, d2, cnt

    def accumulate(self, x, y, z, T, corr):
        x, y, z = _f32(x), _f32(y), _f32(z)
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
        corr = np.ascontiguousarray(corr, dtype=np.int32)
        acc = np.zeros(29)
        lib().vo_accumulate(self.h, _f(x), _f(y), _f(z), x.size, _d(T), _i(corr), _d(acc))
        return acc

    def icp(self, x, y, z, T0, iters=20, d_max=1.0, threads=1):
        x, y, z = _f32(x), _f32(y), _f32(z)
        T0 = np.ascontiguousarray(T0, dtype=np.float64).reshape(12)
        T = np.zeros(12)
        stats = (IcpStat * iters)()
        trace = np.zeros((iters, 12))
        rc = lib().vo_icp(self.h, _f(x), _f(y), _f(z), x.size, _d(T0), iters, float(d_max),
                          _d(T), stats, _d(trace), int(threads))
        if rc != 0:
            raise ValueError("vo_icp rc=%d" % rc)
        st = [dict(n_pairs=s.n_pairs, rmse=s.rmse, candidates=s.candidates) for s in stats]
        return T, st, trace

    def increment(self, x, y, z, T, min_count):
        x, y, z = _f32(x), _f32(y), _f32(z)
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
        ox, oy, oz = (np.empty(x.size, np.float32) for _ in range(3))
        n = lib().vo_increment(self.h, _f(x), _f(y), _f(z), x.size, _d(T), int(min_count),
                               _f(ox), _f(oy), _f(oz))
        return ox[:n].copy(), oy[:n].copy(), oz[:n].copy()


class RollingMap:
    """oracle/icp.c vo_roll: raw list + sticky grid + margin; `.map` is the fresh build."""

    def __init__(self, x, y, z, voxel=1.0, k_normals=16, subdiv=3, margin=0):
        x, y, z = _f32(x), _f32(y), _f32(z)
        m3 = np.ascontiguousarray(np.broadcast_to(np.asarray(margin, np.int32), (3,)))
        self.r = lib().vo_roll_new3(_f(x), _f(y), _f(z), x.size, float(voxel), int(k_normals),
                                    int(subdiv), _i(m3))
        if not self.r:
            raise ValueError("vo_roll_new failed")

    def __del__(self):
        if getattr(self, "r", None):
            lib().vo_roll_free(self.r)
            self.r = None

    @property
    def map(self):
        return Map._view(lib().vo_roll_map(self.r), self)

    @property
    def n(self):
        return lib().vo_roll_size(self.r)

    def append(self, x, y, z):
        x, y, z = _f32(x), _f32(y), _f32(z)
        return lib().vo_roll_append(self.r, _f(x), _f(y), _f(z), x.size)

    def evict_outside(self, lo, hi):
        lo = np.ascontiguousarray(lo, np.float32)
        hi = np.ascontiguousarray(hi, np.float32)
        return lib().vo_roll_evict_outside(self.r, _f(lo), _f(hi))


def _filter_sparse(self, x, y, z, min_count):
    x, y, z = _f32(x), _f32(y), _f32(z)
    acc = np.zeros(x.size, np.uint8)
    lib().vo_roll_filter_sparse(self.r, _f(x), _f(y), _f(z), x.size, int(min_count), acc.ctypes.data_as(C.c_void_p))
    return acc.astype(bool)


def _append_sparse(self, x, y, z, min_count):
    x, y, z = _f32(x), _f32(y), _f32(z)
    return lib().vo_roll_append_sparse(self.r, _f(x), _f(y), _f(z), x.size, int(min_count))


RollingMap.filter_sparse = _filter_sparse
RollingMap.append_sparse = _append_sparse


def _evict_radius(self, cx, cy, radius):
    big = np.float32(3.0e38)
    lo = np.array([-big, -big, -big], np.float32)
    hi = np.array([big, big, big], np.float32)
    return lib().vo_roll_evict_region(self.r, _f(lo), _f(hi), float(cx), float(cy), float(radius))


RollingMap.evict_radius = _evict_radius


def load_corrections(path):
    """HDLParser.cxx:771-858 -> ((64, 9) float64 in LaserCorr field order, enabled count)."""
    corr = np.zeros((64, 9))
    n = C.c_int()
    if lib().vo_load_corrections(str(path).encode(), corr.ctypes.data_as(C.c_void_p), C.byref(n)):
        raise ValueError("cannot read " + str(path))
    return corr, n.value


def solve_update(acc, T):
    acc = np.ascontiguousarray(acc, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(12).copy()
    xi = np.zeros(6)
    rc = lib().vo_solve_update(_d(acc), _d(T), _d(xi))
    return rc, T, xi
