"""ctypes doorway to libveloslam_amd.so (include/velo.h).

This is plumbing only: it marshals numpy arrays / device pointers into the C ABI.
There is no Python or CPU implementation behind it -- if the shared library has
not been built, or no MI355X is visible, the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("VELO_LIB") or os.path.join(CSRC, "libveloslam_amd.so")  # VELO_LIB: A/B builds

VELO_MAX_ITERS = 64
VELO_ABI_VERSION = 3   # include/velo.h; lib() refuses a library that reports another
VELO_MAX_RANKS = 64
VARIANT_BALL, VARIANT_SCAN = 1, 100  # velo_cfg.linearize_variant (0 = default = BALL)
KERNEL_AUTO, KERNEL_THROUGHPUT, KERNEL_LATENCY = 0, 1, 2  # velo_cfg.force_kernel
VELO_TIME_INVALID = -(2 ** 63)


# include/velo.h, "STREAM CONTRACT": a ctx runs on its own non-blocking stream, so whatever produced a device
# buffer handed to a *_dev / *_async / exchange call must have completed before the call.  A caller that makes
# its buffers with torch can install a hook here (tests/conftest.py does: torch.cuda.current_stream().synchronize)
# and every call below that takes device pointers runs it first.  None (default) = the caller orders its
# producers itself, which is what bench.py's timed loops do.
_producer_sync = None


def set_producer_sync(fn):
    """fn() is called before every entry point that is handed device pointers (None: no hook)."""
    global _producer_sync
    _producer_sync = fn


def _order_producers():
    if _producer_sync is not None:
        _producer_sync()


class VeloError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("velo error %d: %s" % (code, msg))
        self.code = code


class Cfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_batch", C.c_int32),
                ("linearize_variant", C.c_int32), ("sort_frames", C.c_int32),
                ("use_graph", C.c_int32), ("map_subdiv", C.c_int32), ("use_hints", C.c_int32),
                ("rounds_per_block", C.c_int32), ("map_margin", C.c_int32),
                ("map_full_rebuild", C.c_int32), ("map_hash_load", C.c_int32), ("force_kernel", C.c_int32), ("plan_wave_slots", C.c_int32), ("abi_version", C.c_uint32),
                ("reserved", C.c_int32 * 2),
                # ABI 3: the tuning knobs (include/velo.h)
                ("split_iterations", C.c_int32), ("split_batches", C.c_int32), ("split_per_wave_max", C.c_int32),
                ("solve_threads", C.c_int32), ("roll_cus", C.c_int32), ("pair_certificates", C.c_int32),
                ("reserved2", C.c_int32 * 2)]


class Pose(C.Structure):
    _fields_ = [("T", C.c_double * 3), ("R", C.c_double * 3), ("V", C.c_double * 3),
                ("t_us", C.c_int64), ("week_number", C.c_uint16), ("milliseconds", C.c_uint32),
                ("week_number_pos", C.c_uint32), ("seconds_pos", C.c_double)]


class InsPVA(C.Structure):
    _fields_ = [("message_id", C.c_uint16), ("week_number", C.c_uint16), ("milliseconds", C.c_uint32),
                ("week_number_pos", C.c_uint32), ("seconds_pos", C.c_double), ("LLH", C.c_double * 3),
                ("V", C.c_double * 3), ("Eulr", C.c_double * 3), ("ins_status", C.c_int32)]


class DecodeOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("points_skip", C.c_int32), ("initial_firing_skip", C.c_int32),
                ("laser_selection", C.c_uint8 * 64)]


class FrameIndex(C.Structure):
    _fields_ = [("file_pos", C.c_int64), ("firing_skip", C.c_int32), ("reserved", C.c_int32),
                ("first_packet", C.c_int64), ("t_us", C.c_int64)]


class IcpIter(C.Structure):
    _fields_ = [("n_pairs", C.c_uint32), ("solve_flag", C.c_uint32), ("rmse", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("T", C.c_double * 12), ("TRdeg", C.c_double * 6), ("iters", C.c_int32),
                ("reserved", C.c_int32), ("total_pairs", C.c_uint64),
                ("iter", IcpIter * VELO_MAX_ITERS)]


class MapInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("n_points", C.c_uint64), ("n_cells", C.c_uint64), ("origin", C.c_float * 3),
                ("voxel", C.c_float), ("inv_voxel", C.c_float), ("dims", C.c_int32 * 3),
                ("k_normals", C.c_int32), ("n_invalid_normals", C.c_uint64),
                ("subdiv", C.c_int32), ("last_update", C.c_int32),
                ("n_normals_recomputed", C.c_uint64), ("table_kind", C.c_int32), ("reserved", C.c_int32),
                ("table_slots", C.c_uint64), ("table_occupied", C.c_uint64)]


# every symbol include/velo.h declares (tests check the library exports them all)
EXPORTS = [
    "velo_create", "velo_destroy", "velo_last_error", "velo_abi_version", "velo_cfg_get", "velo_set_stream",
    "velo_synchronize", "velo_map_reset", "velo_map_reset_dev", "velo_map_append",
    "velo_map_append_dev", "velo_map_append_sparse", "velo_map_append_sparse_dev", "velo_map_evict_outside", "velo_map_roll_overlapped", "velo_map_roll_begin", "velo_map_roll_publish", "velo_map_evict_radius", "velo_map_set_margins", "velo_map_info_get", "velo_map_size", "velo_map_download", "velo_compensate",
    "velo_compensate_dev", "velo_icp", "velo_frames_upload", "velo_frames_adopt_dev",
    "velo_icp_batch", "velo_icp_batch_async", "velo_icp_batch_fetch", "velo_icp_batch_start", "velo_icp_batch_finish", "velo_linearize",
    "velo_linearize_hints", "velo_solve_update", "velo_knn", "velo_knn_dev", "velo_decode", "velo_decode_stream", "velo_decode_stream_reset", "velo_decode_set_options", "velo_decode_fetch", "velo_decode_to_frames", "velo_decode_plan_create", "velo_decode_plan_destroy", "velo_decode_plan_fill", "velo_decode_submit", "velo_decode_submit_overlapped", "velo_decode_plan_error", "velo_increment", "velo_increment_dev", "velo_increment_registered_async", "velo_increment_all_registered_async", "velo_increment_wait", "velo_increment_pending", "velo_pending_count", "velo_pending_fetch", "velo_map_append_pending", "velo_pending_clear", "velo_comm_unique_id", "velo_comm_init", "velo_comm_destroy", "velo_comm_info",
    "velo_exchange_increments", "velo_exchange_plan", "velo_exchange_pack_dev", "velo_last_timing", "velo_last_linearize_us", "velo_set_timing", "velo_debug_search_stats", "velo_set_stats", "velo_pairs_total", "velo_search_stats",
    "velo_matrix_from_pose", "velo_pose_from_matrix", "velo_interp_pose",
    "velo_packet_transforms", "velo_pcap_write", "velo_pcap_read", "velo_pcap_index", "velo_ins_to_pose",
    "velo_insmeta_write", "velo_insmeta_read", "velo_carposes_read", "velo_time_to_week_milli", "velo_load_corrections", "eulr2dcm", "llh2xyz", "xyz2llh", "xyz2enu", "enu2xyz", "enu2llh",
    "llh2enu", "MappingAngle",
]


def build(verbose=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libveloslam_amd.so is not built (%s). Run `make -C veloslam_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback."
            % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    if L.velo_abi_version() != VELO_ABI_VERSION:
        raise RuntimeError("%s implements VELO_ABI_VERSION %d, this binding was written for %d: rebuild it "
                           "(make -C veloslam_amd/csrc)" % (LIB_PATH, L.velo_abi_version(), VELO_ABI_VERSION))
    fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.velo_create.restype = vp
    L.velo_create.argtypes = [C.c_int, C.POINTER(Cfg)]
    L.velo_destroy.argtypes = [vp]
    L.velo_last_error.restype = C.c_char_p
    L.velo_last_error.argtypes = [vp]
    L.velo_set_stream.argtypes = [vp, vp]
    L.velo_cfg_get.argtypes = [vp, C.POINTER(Cfg)]
    L.velo_synchronize.argtypes = [vp]
    L.velo_map_reset.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_float, C.c_int]
    L.velo_map_reset_dev.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_float, C.c_int]
    L.velo_map_append.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.velo_map_append_dev.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.velo_map_append_sparse.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]
    L.velo_map_append_sparse_dev.argtypes = L.velo_map_append_sparse.argtypes
    L.velo_map_evict_outside.argtypes = [vp, vp, vp]
    L.velo_map_roll_overlapped.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t]
    L.velo_map_roll_begin.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t]
    L.velo_map_roll_publish.argtypes = [vp]
    L.velo_map_evict_radius.argtypes = [vp, vp, C.c_float]
    L.velo_map_set_margins.argtypes = [vp, vp]
    L.velo_debug_search_stats.argtypes = [vp, vp, C.c_int]
    L.velo_search_stats.argtypes = [vp, vp, C.c_int]
    L.velo_set_stats.argtypes = [vp, C.c_int]
    L.velo_pairs_total.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.velo_map_info_get.argtypes = [vp, C.POINTER(MapInfo)]
    L.velo_map_size.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.velo_map_download.argtypes = [vp] + [vp] * 8
    L.velo_compensate.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp]
    L.velo_compensate_dev.argtypes = L.velo_compensate.argtypes
    L.velo_icp.argtypes = [vp, vp, vp, vp, C.c_size_t, dp, C.c_int, C.c_float, C.c_int,
                           C.POINTER(IcpResult)]
    L.velo_frames_upload.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.velo_frames_adopt_dev.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.velo_icp_batch.argtypes = [vp, dp, C.c_int, C.c_float, C.POINTER(IcpResult)]
    L.velo_icp_batch_async.argtypes = [vp, dp, C.c_int, C.c_float]
    L.velo_icp_batch_fetch.argtypes = [vp, C.POINTER(IcpResult)]
    L.velo_icp_batch_start.argtypes = [vp, dp, C.c_int, C.c_float]
    L.velo_icp_batch_finish.argtypes = [vp, C.POINTER(IcpResult)]
    L.velo_linearize.argtypes = [vp, C.c_int, dp, C.c_float, vp, vp, dp]
    L.velo_linearize_hints.argtypes = [vp, C.c_int]
    L.velo_solve_update.argtypes = [vp, dp, dp, C.POINTER(C.c_int32)]
    L.velo_knn.argtypes = [vp, C.c_int, dp, C.c_float, C.c_int, vp, vp, vp]
    L.velo_knn_dev.argtypes = [vp, C.c_int, dp, C.c_float, C.c_int, vp, vp, vp, vp]
    L.velo_decode.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_int, C.POINTER(Pose), C.c_size_t, C.c_int,
                              vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]
    L.velo_decode_stream.argtypes = L.velo_decode.argtypes
    L.velo_decode_stream_reset.argtypes = [vp]
    L.velo_decode_set_options.argtypes = [vp, C.POINTER(DecodeOpts)]
    L.velo_decode_fetch.argtypes = [vp] * 13
    L.velo_decode_to_frames.argtypes = [vp]
    L.velo_decode_plan_create.argtypes = [vp, C.POINTER(vp)]
    L.velo_decode_plan_destroy.argtypes = [vp]
    L.velo_decode_plan_destroy.restype = None
    L.velo_decode_plan_fill.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, C.c_int, vp, C.c_size_t, C.c_int, vp, C.c_int]
    L.velo_decode_submit.argtypes = [vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]
    L.velo_decode_submit_overlapped.argtypes = L.velo_decode_submit.argtypes
    L.velo_decode_plan_error.argtypes = [vp]
    L.velo_decode_plan_error.restype = C.c_char_p
    L.velo_increment.argtypes = [vp, C.c_int, dp, C.c_int, vp, vp, vp, C.POINTER(C.c_size_t)]
    L.velo_increment_dev.argtypes = L.velo_increment.argtypes
    L.velo_increment_registered_async.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.velo_increment_all_registered_async.argtypes = [vp, C.c_int, vp, vp, vp]
    L.velo_increment_wait.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.velo_comm_unique_id.argtypes = [vp]
    L.velo_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.velo_comm_destroy.argtypes = [vp]
    L.velo_comm_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.velo_exchange_increments.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, vp, vp, vp, C.c_size_t, vp,
                                           C.POINTER(C.c_size_t)]
    L.velo_increment_pending.argtypes = [vp, C.c_int, vp, C.c_int]
    L.velo_pending_count.argtypes = [vp, C.POINTER(C.c_size_t), C.c_int]
    L.velo_pending_fetch.argtypes = [vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.velo_map_append_pending.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.velo_pending_clear.argtypes = [vp]
    L.velo_exchange_plan.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.velo_exchange_pack_dev.argtypes = [vp, vp, vp, C.c_int, C.c_size_t, vp, vp, vp, C.c_size_t,
                                         C.POINTER(C.c_size_t)]
    L.velo_last_timing.argtypes = [vp, dp]
    L.velo_set_timing.argtypes = [vp, C.c_int]
    L.velo_last_linearize_us.argtypes = [vp, vp, C.c_int]
    L.velo_matrix_from_pose.argtypes = [dp, dp]
    L.velo_pose_from_matrix.argtypes = [dp, dp]
    L.velo_interp_pose.argtypes = [C.POINTER(Pose), C.c_size_t, C.c_int64, C.POINTER(Pose)]
    L.velo_packet_transforms.argtypes = [C.POINTER(Pose), C.c_size_t, C.POINTER(C.c_int64),
                                         C.c_size_t, dp, C.POINTER(C.c_uint8), C.POINTER(Pose)]
    L.velo_pcap_write.argtypes = [C.c_char_p, vp, vp, C.c_size_t]
    L.velo_pcap_read.argtypes = [C.c_char_p, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.velo_ins_to_pose.argtypes = [C.POINTER(InsPVA), dp, C.c_int64, C.POINTER(Pose)]
    L.velo_insmeta_write.argtypes = [C.c_char_p, C.POINTER(Pose), C.c_size_t]
    L.velo_insmeta_read.argtypes = [C.c_char_p, C.POINTER(Pose), C.c_size_t, C.POINTER(C.c_size_t)]
    L.velo_load_corrections.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_int32)]
    for nm in ("llh2xyz", "xyz2llh"):
        getattr(L, nm).argtypes = [dp, dp]
    for nm in ("xyz2enu", "enu2xyz", "enu2llh", "llh2enu"):
        getattr(L, nm).argtypes = [dp, dp, dp]
    L.eulr2dcm.argtypes = [dp, dp]
    L.MappingAngle.argtypes = [C.c_double]
    L.MappingAngle.restype = C.c_double
    _lib = L
    return L


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------- host helpers
def comm_unique_id():
    """128 opaque bytes from rank 0, to be carried to the other ranks (velo_comm_init)."""
    buf = np.zeros(128, np.uint8)
    rc = lib().velo_comm_unique_id(_p(buf))
    if rc:
        raise VeloError(rc, lib().velo_last_error(None).decode())
    return buf.tobytes()


def exchange_plan(counts):
    """velo_exchange_plan (host only, no GPU): per-rank counts -> (offsets[world+1], pad, total)"""
    cnt = np.ascontiguousarray(counts, np.int32)
    offs = np.zeros(cnt.size + 1, np.uint32)
    pad, tot = C.c_size_t(), C.c_size_t()
    rc = lib().velo_exchange_plan(_p(cnt), cnt.size, _p(offs), C.byref(pad), C.byref(tot))
    if rc:
        raise VeloError(rc, "velo_exchange_plan")
    return offs, pad.value, tot.value


def matrix_from_pose(T, Rdeg):
    tr = np.array(list(T) + list(Rdeg), dtype=np.float64)
    M = np.zeros(12)
    lib().velo_matrix_from_pose(_d(tr), _d(M))
    return M


def pose_from_matrix(M):
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(12)
    tr = np.zeros(6)
    lib().velo_pose_from_matrix(_d(M), _d(tr))
    return tr


def make_poses(samples):
    """samples: iterable of (T, Rdeg, V, t_us[, seconds_pos]) -> ctypes array of Pose."""
    samples = list(samples)
    arr = (Pose * max(len(samples), 1))()
    for i, s in enumerate(samples):
        T, R, V, t = s[:4]
        for k in range(3):
            arr[i].T[k], arr[i].R[k], arr[i].V[k] = float(T[k]), float(R[k]), float(V[k])
        arr[i].t_us = int(t)
        arr[i].seconds_pos = float(s[4]) if len(s) > 4 else 0.0
    return arr, len(samples)


def interp_pose(poses, n, t_us):
    out = Pose()
    rc = lib().velo_interp_pose(poses, n, int(t_us), C.byref(out))
    return rc == 0, out


def packet_transforms(poses, n, pkt_times):
    t = np.ascontiguousarray(pkt_times, dtype=np.int64)
    tab = np.zeros((t.size, 12))
    valid = np.zeros(t.size, dtype=np.uint8)
    car = Pose()
    rc = lib().velo_packet_transforms(poses, n, t.ctypes.data_as(C.POINTER(C.c_int64)), t.size,
                                      _d(tab), valid.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      C.byref(car))
    if rc:
        raise VeloError(rc, "velo_packet_transforms")
    return tab, valid, car


def time_to_week_milli(t_us):
    """ptimeToWeekMilli (type_defs.cxx:74-79): (ISO week of the date, ms since Sunday 00:00)."""
    L = lib()
    L.velo_time_to_week_milli.argtypes = [C.c_int64, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)]
    L.velo_time_to_week_milli.restype = None
    w, m = C.c_uint16(0), C.c_uint32(0)
    L.velo_time_to_week_milli(int(t_us), C.byref(w), C.byref(m))
    return w.value, m.value


def load_corrections(path):
    """Velodyne db.xml -> ((64, 9) float64 laser corrections in velo_laser_corr field order,
    number of enabled lasers)."""
    corr = np.zeros((64, 9))
    n = C.c_int32()
    rc = lib().velo_load_corrections(str(path).encode(), corr.ctypes.data_as(C.c_void_p), C.byref(n))
    if rc:
        raise VeloError(rc, "velo_load_corrections: cannot read %s" % path)
    return corr, n.value


def pcap_write(path, packets, times_us):
    buf = np.frombuffer(b"".join(packets), dtype=np.uint8)
    t = np.ascontiguousarray(times_us, dtype=np.int64)
    rc = lib().velo_pcap_write(path.encode(), _p(buf), _p(t), len(packets))
    if rc:
        raise VeloError(rc, "velo_pcap_write")


def pcap_read(path):
    n = C.c_size_t()
    rc = lib().velo_pcap_read(path.encode(), None, None, 0, C.byref(n))
    if rc:
        raise VeloError(rc, "velo_pcap_read")
    buf = np.empty(n.value * 1206, np.uint8)
    t = np.empty(n.value, np.int64)
    rc = lib().velo_pcap_read(path.encode(), _p(buf), _p(t), n.value, C.byref(n))
    if rc:
        raise VeloError(rc, "velo_pcap_read")
    return [bytes(buf[i * 1206:(i + 1) * 1206]) for i in range(n.value)], t


def pcap_index(path):
    """velo_pcap_index -> list of FrameIndex (file_pos, firing_skip, first_packet, t_us)"""
    L = lib()
    L.velo_pcap_index.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    n = C.c_size_t()
    rc = L.velo_pcap_index(path.encode(), None, 0, C.byref(n))
    if rc:
        raise VeloError(rc, "velo_pcap_index")
    arr = (FrameIndex * max(n.value, 1))()
    rc = L.velo_pcap_index(path.encode(), arr, n.value, C.byref(n))
    if rc:
        raise VeloError(rc, "velo_pcap_index")
    return list(arr[:n.value])


def _geo2(name, a):
    a = np.ascontiguousarray(a, dtype=np.float64).copy()
    o = np.zeros(3)
    getattr(lib(), name)(_d(a), _d(o))
    return o


def _geo3(name, a, org):
    a = np.ascontiguousarray(a, dtype=np.float64).copy()
    org = np.ascontiguousarray(org, dtype=np.float64).copy()
    o = np.zeros(3)
    getattr(lib(), name)(_d(a), _d(org), _d(o))
    return o


def llh2xyz(a): return _geo2("llh2xyz", a)
def xyz2llh(a): return _geo2("xyz2llh", a)
def xyz2enu(a, org): return _geo3("xyz2enu", a, org)
def enu2xyz(a, org): return _geo3("enu2xyz", a, org)
def enu2llh(a, org): return _geo3("enu2llh", a, org)
def llh2enu(a, org): return _geo3("llh2enu", a, org)


def eulr2dcm(e):
    e = np.ascontiguousarray(e, dtype=np.float64).copy()
    o = np.zeros(9)
    lib().eulr2dcm(_d(e), _d(o))
    return o.reshape(3, 3)


def mapping_angle(a):
    return lib().MappingAngle(float(a))


# ----------------------------------------------------------------- GPU context
class Context:
    """One velo_ctx: one GPU, one stream, single-threaded."""

    def __init__(self, device=0, max_batch=64, sort_frames=0, linearize_variant=1, map_subdiv=3,
                 use_hints=2, use_graph=1, rounds_per_block=0, map_margin=0, map_full_rebuild=0,
                 map_hash_load=0, force_kernel=0, plan_wave_slots=0, split_iterations=0, split_batches=0,
                 split_per_wave_max=0, solve_threads=0, roll_cus=0, pair_certificates=0):
        L = lib()
        cfg = Cfg()
        cfg.struct_size = C.sizeof(Cfg)
        cfg.abi_version = VELO_ABI_VERSION
        cfg.max_batch = max_batch
        cfg.sort_frames = sort_frames
        cfg.linearize_variant = linearize_variant
        cfg.map_subdiv = map_subdiv
        cfg.use_hints = use_hints
        cfg.use_graph = use_graph
        cfg.rounds_per_block = rounds_per_block
        cfg.map_margin = map_margin
        cfg.map_full_rebuild = map_full_rebuild
        cfg.map_hash_load = map_hash_load
        cfg.force_kernel = force_kernel
        cfg.plan_wave_slots = plan_wave_slots
        cfg.split_iterations = split_iterations
        cfg.split_batches = split_batches
        cfg.split_per_wave_max = split_per_wave_max
        cfg.solve_threads = solve_threads
        cfg.roll_cus = roll_cus
        cfg.pair_certificates = pair_certificates
        self.h = L.velo_create(device, C.byref(cfg))
        if not self.h:
            raise VeloError(-3, L.velo_last_error(None).decode())
        self.max_batch = max_batch
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            lib().velo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise VeloError(rc, lib().velo_last_error(self.h).decode())

    def cfg(self):
        out = Cfg()
        self._chk(lib().velo_cfg_get(self.h, C.byref(out)))
        return out

    def set_stream(self, stream_ptr):
        self._chk(lib().velo_set_stream(self.h, C.c_void_p(stream_ptr or 0)))

    def pairs_total(self, reset=False):
        v = C.c_uint64()
        self._chk(lib().velo_pairs_total(self.h, C.byref(v), int(reset)))
        return v.value

    def set_stats(self, on):
        """1: launch the counting instantiation of the linearise kernel (same results)."""
        self._chk(lib().velo_set_stats(self.h, int(on)))

    def search_stats(self, reset=True):
        out = (C.c_uint64 * 16)()
        self._chk(lib().velo_search_stats(self.h, out, int(reset)))
        return dict(zip(("live", "certified", "searched", "empty_skips", "stage_a_final",
                         "stage_b_per_lane", "stage_b", "valid_pairs", "bytes", "candidates",
                         "table_requests", "launches", "query_bytes"), [int(v) for v in out[:13]]))

    def synchronize(self):
        self._chk(lib().velo_synchronize(self.h))

    def set_timing(self, on):
        self._chk(lib().velo_set_timing(self.h, int(on)))

    def last_timing(self):
        t = np.zeros(8)
        self._chk(lib().velo_last_timing(self.h, _d(t)))
        return dict(linearize_ms=t[0], linearize_launches=int(t[1]), solve_ms=t[2],
                    solve_launches=int(t[3]), call_ms=t[4], linearize_first_ms=t[5],
                    linearize_min_ms=t[6])

    def last_linearize_us(self):
        buf = np.zeros(VELO_MAX_ITERS, np.float32)
        n = lib().velo_last_linearize_us(self.h, _p(buf), buf.size)
        return buf[:max(min(n, buf.size), 0)].copy()

    # ---- map
    def map_reset(self, x, y, z, voxel=1.0, k_normals=16):
        x, y, z = _f32(x), _f32(y), _f32(z)
        self._chk(lib().velo_map_reset(self.h, _p(x), _p(y), _p(z), x.size, voxel, k_normals))

    def map_reset_dev(self, px, py, pz, n, voxel=1.0, k_normals=16):
        _order_producers()
        self._chk(lib().velo_map_reset_dev(self.h, px, py, pz, n, voxel, k_normals))

    def map_append(self, x, y, z):
        x, y, z = _f32(x), _f32(y), _f32(z)
        self._chk(lib().velo_map_append(self.h, _p(x), _p(y), _p(z), x.size))

    def map_append_dev(self, px, py, pz, n):
        _order_producers()
        self._chk(lib().velo_map_append_dev(self.h, px, py, pz, n))

    def map_append_sparse(self, x, y, z, min_count):
        x, y, z = _f32(x), _f32(y), _f32(z)
        k = C.c_size_t()
        self._chk(lib().velo_map_append_sparse(self.h, _p(x), _p(y), _p(z), x.size, min_count, C.byref(k)))
        return k.value

    def map_append_sparse_dev(self, px, py, pz, n, min_count):
        _order_producers()
        k = C.c_size_t()
        self._chk(lib().velo_map_append_sparse_dev(self.h, px, py, pz, n, min_count, C.byref(k)))
        return k.value

    def map_set_margins(self, mx, my, mz):
        m = np.array([mx, my, mz], np.int32)
        self._chk(lib().velo_map_set_margins(self.h, _p(m)))

    def map_roll_overlapped(self, lo, hi, x, y, z):
        """evict (lo / hi None: no eviction) + append beside the registration in flight; -> False when the
        library refuses (VELO_E_AGAIN: do it with the plain calls after icp_batch_finish)"""
        x, y, z = (np.ascontiguousarray(a, np.float32) for a in (x, y, z))
        plo = None if lo is None else _p(np.ascontiguousarray(lo, np.float32))
        phi = None if hi is None else _p(np.ascontiguousarray(hi, np.float32))
        rc = lib().velo_map_roll_overlapped(self.h, plo, phi, _p(x), _p(y), _p(z), x.size)
        if rc == -7:
            return False
        self._chk(rc)
        return True

    def map_roll_begin(self, lo, hi, x, y, z):
        """the same roll begun ahead (velo_map_roll_begin): enqueued on a stream of its own, readers keep the map
        as it was until map_roll_publish(); -> False when the library refuses (VELO_E_AGAIN)"""
        x, y, z = (np.ascontiguousarray(a, np.float32) for a in (x, y, z))
        plo = None if lo is None else _p(np.ascontiguousarray(lo, np.float32))
        phi = None if hi is None else _p(np.ascontiguousarray(hi, np.float32))
        rc = lib().velo_map_roll_begin(self.h, plo, phi, _p(x), _p(y), _p(z), x.size)
        if rc == -7:
            return False
        self._chk(rc)
        return True

    def map_roll_publish(self):
        self._chk(lib().velo_map_roll_publish(self.h))

    def map_evict_outside(self, lo, hi):
        lo = np.ascontiguousarray(lo, np.float32)
        hi = np.ascontiguousarray(hi, np.float32)
        self._chk(lib().velo_map_evict_outside(self.h, _p(lo), _p(hi)))

    def map_evict_radius(self, cx, cy, radius):
        c = np.array([cx, cy], np.float32)
        self._chk(lib().velo_map_evict_radius(self.h, _p(c), float(radius)))

    def map_info(self):
        mi = MapInfo()
        mi.struct_size = C.sizeof(MapInfo)
        self._chk(lib().velo_map_info_get(self.h, C.byref(mi)))
        return mi

    def map_size(self):
        """points of the map as of the last update (begun rolls included); never waits"""
        n = C.c_uint64()
        self._chk(lib().velo_map_size(self.h, C.byref(n)))
        return int(n.value)

    def map_download(self):
        mi = self.map_info()
        n, nc = mi.n_points, mi.n_cells
        out = {k: np.empty(n, np.float32) for k in ("x", "y", "z", "nx", "ny", "nz")}
        out["perm"] = np.empty(n, np.int32)
        out["cell_start"] = np.empty(nc + 1, np.int32)
        self._chk(lib().velo_map_download(self.h, *[_p(out[k]) for k in
                                                    ("x", "y", "z", "nx", "ny", "nz", "perm",
                                                     "cell_start")]))
        return out

    def map_download_perm(self):
        """only the sort permutation (maps whose dense table is too large to copy back)"""
        n = self.map_info().n_points
        perm = np.empty(n, np.int32)
        self._chk(lib().velo_map_download(self.h, None, None, None, None, None, None, _p(perm), None))
        return perm

    # ---- K1
    def compensate(self, x, y, z, pkt, table):
        x, y, z = _f32(x), _f32(y), _f32(z)
        pkt = np.ascontiguousarray(pkt, dtype=np.uint16)
        table = np.ascontiguousarray(table, dtype=np.float64).reshape(-1, 12)
        ox, oy, oz = (np.empty(x.size, np.float32) for _ in range(3))
        self._chk(lib().velo_compensate(self.h, _p(x), _p(y), _p(z), _p(pkt), x.size, _p(table),
                                        table.shape[0], _p(ox), _p(oy), _p(oz)))
        return ox, oy, oz

    def compensate_dev(self, px, py, pz, ppkt, n, ptab, n_pkt, pox, poy, poz):
        _order_producers()
        self._chk(lib().velo_compensate_dev(self.h, px, py, pz, ppkt, n, ptab, n_pkt, pox, poy, poz))

    # ---- ICP
    def icp(self, x, y, z, T0, iters=20, d_max=1.0):
        x, y, z = _f32(x), _f32(y), _f32(z)
        T0 = np.ascontiguousarray(T0, dtype=np.float64).reshape(12)
        res = IcpResult()
        self._chk(lib().velo_icp(self.h, _p(x), _p(y), _p(z), x.size, _d(T0), iters, d_max, 1,
                                 C.byref(res)))
        return res

    def frames_upload(self, frames):
        """frames: list of (x, y, z) float arrays."""
        xs = _f32(np.concatenate([f[0] for f in frames]))
        ys = _f32(np.concatenate([f[1] for f in frames]))
        zs = _f32(np.concatenate([f[2] for f in frames]))
        fs = np.zeros(len(frames) + 1, dtype=np.int64)
        fs[1:] = np.cumsum([len(f[0]) for f in frames])
        self._chk(lib().velo_frames_upload(self.h, len(frames), _p(xs), _p(ys), _p(zs), _p(fs)))
        self.n_frames = len(frames)

    def frames_adopt_dev(self, px, py, pz, frame_start):
        _order_producers()
        fs = np.ascontiguousarray(frame_start, dtype=np.int64)
        self._chk(lib().velo_frames_adopt_dev(self.h, fs.size - 1, px, py, pz, _p(fs)))
        self.n_frames = fs.size - 1

    def icp_batch(self, T0, iters=20, d_max=1.0):
        T0 = np.ascontiguousarray(T0, dtype=np.float64).reshape(self.n_frames, 12)
        res = (IcpResult * self.n_frames)()
        self._chk(lib().velo_icp_batch(self.h, _d(T0), iters, d_max, res))
        return res

    def icp_batch_async(self, T0, iters=20, d_max=1.0):
        T0 = np.ascontiguousarray(T0, dtype=np.float64).reshape(self.n_frames, 12)
        self._chk(lib().velo_icp_batch_async(self.h, _d(T0), iters, d_max))

    def icp_batch_start(self, T0, iters=20, d_max=1.0):
        T0 = np.ascontiguousarray(T0, dtype=np.float64).reshape(self.n_frames, 12)
        self._started_frames = self.n_frames
        self._chk(lib().velo_icp_batch_start(self.h, _d(T0), iters, d_max))

    def icp_batch_finish(self):
        res = (IcpResult * max(getattr(self, "_started_frames", 1), 1))()
        self._chk(lib().velo_icp_batch_finish(self.h, res))
        return res

    def icp_batch_fetch(self):
        res = (IcpResult * self.n_frames)()
        self._chk(lib().velo_icp_batch_fetch(self.h, res))
        return res

    def linearize(self, frame, T, d_max, n):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
        corr = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        acc = np.zeros(29)
        self._chk(lib().velo_linearize(self.h, frame, _d(T), d_max, _p(corr), _p(d2), _d(acc)))
        return corr, d2, acc

    def solve_update(self, acc, T):
        """a12 on the device: (29 sums, pose) -> (solve_flag, updated pose)."""
        acc = np.ascontiguousarray(acc, dtype=np.float64).reshape(29)
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12).copy()
        flag = C.c_int32()
        self._chk(lib().velo_solve_update(self.h, _d(acc), _d(T), C.byref(flag)))
        return flag.value, T

    def knn(self, frame, T, d_max, k, n):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
        idx = np.empty((n, k), np.int32)
        d2 = np.empty((n, k), np.float32)
        cnt = np.empty(n, np.int32)
        self._chk(lib().velo_knn(self.h, frame, _d(T), d_max, k, _p(idx), _p(d2), _p(cnt)))
        return idx, d2, cnt

    def knn_dev(self, frame, T, d_max, k, pidx, pd2, pcount=None, stats=False):
        """velo_knn_dev: results stay on the device; stats=True -> the counting instantiation's dict"""
        _order_producers()
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
        st = (C.c_uint64 * 4)() if stats else None
        self._chk(lib().velo_knn_dev(self.h, frame, _d(T), d_max, k, pidx, pd2, pcount, st))
        if stats:
            return dict(queries=int(st[0]), candidates=int(st[1]), rows=int(st[2]), cells=int(st[3]))
        return None

    def decode(self, packets, times_us, calib, n_lasers=64, poses=None, n_poses=0, flush=True,
               crop_region=None, crop_inside=False, stream=False):
        """packets: list of 1206-byte strings; calib: (64, 9) float64.  Returns a dict with the
        decoded frames (beam-major SoA) fetched back to the host.  stream=True keeps the parser
        state inside the ctx between calls (velo_decode_stream)."""
        buf = np.frombuffer(b"".join(packets), dtype=np.uint8) if len(packets) else np.zeros(0, np.uint8)
        t = np.ascontiguousarray(times_us, dtype=np.int64)
        cal = np.ascontiguousarray(calib, dtype=np.float64).reshape(64, 9)
        crop = None if crop_region is None else np.ascontiguousarray(crop_region, dtype=np.float64)
        nf = C.c_int32()
        npts = C.c_size_t()
        fn = lib().velo_decode_stream if stream else lib().velo_decode
        self._chk(fn(self.h, _p(buf), _p(t), len(packets), _p(cal), n_lasers,
                     poses, n_poses, int(bool(flush)), _p(crop), int(bool(crop_inside)),
                     C.byref(nf), C.byref(npts)))
        self._decoded_frames = nf.value
        return self.decode_fetch(nf.value, npts.value)

    def decode_fetch(self, F, n):
        """the last decode (F frames, n points) copied back to the host"""
        out = dict(n_frames=F, n_points=n,
                   x=np.empty(n, np.float32), y=np.empty(n, np.float32), z=np.empty(n, np.float32),
                   intensity=np.empty(n, np.float32), azimuth=np.empty(n, np.uint16),
                   distance=np.empty(n, np.float32), packet_index=np.empty(n, np.uint16),
                   frame_start=np.zeros(F + 1, np.int64), beam_start=np.zeros((F, 65), np.int32),
                   frame_t_us=np.zeros(F, np.int64), frame_packets=np.zeros(F, np.int32))
        car = (Pose * max(F, 1))()
        self._chk(lib().velo_decode_fetch(
            self.h, _p(out["x"]), _p(out["y"]), _p(out["z"]), _p(out["intensity"]), _p(out["azimuth"]),
            _p(out["distance"]), _p(out["packet_index"]), _p(out["frame_start"]), _p(out["beam_start"]),
            C.cast(car, C.c_void_p), _p(out["frame_t_us"]), _p(out["frame_packets"])))
        out["carposes"] = car
        return out

    def decode_set_options(self, laser_selection=None, points_skip=0, initial_firing_skip=0):
        o = DecodeOpts()
        o.struct_size = C.sizeof(DecodeOpts)
        o.points_skip = points_skip
        o.initial_firing_skip = initial_firing_skip
        for i in range(64):
            o.laser_selection[i] = 1 if laser_selection is None else int(bool(laser_selection[i]))
        self._chk(lib().velo_decode_set_options(self.h, C.byref(o)))

    def decode_stream_reset(self):
        self._chk(lib().velo_decode_stream_reset(self.h))

    def decode_resident(self, buf, times_us, calib, poses, n_poses, n_lasers=64, flush=True):
        """velo_decode without the host fetch: buf = contiguous uint8 packets (n x 1206),
        the decoded frames stay on the device (follow with decode_to_frames).
        Returns (n_frames, n_points)."""
        nf = C.c_int32()
        npts = C.c_size_t()
        self._chk(lib().velo_decode(self.h, _p(buf), _p(times_us), times_us.size, _p(calib),
                                    n_lasers, poses, n_poses, int(bool(flush)), None, 0,
                                    C.byref(nf), C.byref(npts)))
        self._decoded_frames = nf.value
        return nf.value, npts.value

    # ---- the decode in its two halves (host plan / device submit)
    def decode_plan_create(self):
        h = C.c_void_p()
        self._chk(lib().velo_decode_plan_create(self.h, C.byref(h)))
        return h

    @staticmethod
    def decode_plan_destroy(plan):
        lib().velo_decode_plan_destroy(plan)

    @staticmethod
    def decode_plan_fill(plan, buf, times_us, calib, poses, n_poses, n_lasers=64, flush=True,
                         initial_firing_skip=0, points_skip=0, laser_selection=None):
        """host half of velo_decode into `plan` (no GPU work, the ctx is not touched)"""
        o = DecodeOpts()
        o.struct_size = C.sizeof(DecodeOpts)
        o.points_skip = points_skip
        o.initial_firing_skip = initial_firing_skip
        for i in range(64):
            o.laser_selection[i] = 1 if laser_selection is None else int(bool(laser_selection[i]))
        rc = lib().velo_decode_plan_fill(plan, C.byref(o), _p(buf), _p(times_us), times_us.size, _p(calib), n_lasers,
                                         poses, n_poses, int(bool(flush)), None, 0)
        if rc:
            raise RuntimeError("velo_decode_plan_fill: %d %s" % (rc, lib().velo_decode_plan_error(plan).decode()))

    def decode_submit(self, plan):
        """device half: -> (n_frames, n_points), frames left on the device like decode_resident"""
        nf = C.c_int32()
        npts = C.c_size_t()
        self._chk(lib().velo_decode_submit(self.h, plan, C.byref(nf), C.byref(npts)))
        self._decoded_frames = nf.value
        return nf.value, npts.value

    def decode_submit_overlapped(self, plan):
        """device half + adoption on the side stream, during a registration begun with icp_batch_start"""
        nf = C.c_int32()
        npts = C.c_size_t()
        self._chk(lib().velo_decode_submit_overlapped(self.h, plan, C.byref(nf), C.byref(npts)))
        self._decoded_frames = nf.value
        self.n_frames = nf.value
        return nf.value, npts.value

    def decode_to_frames(self):
        self._chk(lib().velo_decode_to_frames(self.h))
        self.n_frames = self._decoded_frames

    def linearize_hints(self, mode):
        self._chk(lib().velo_linearize_hints(self.h, int(mode)))

    def increment(self, frame, T, min_count, n):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
        ox, oy, oz = (np.empty(n, np.float32) for _ in range(3))
        cnt = C.c_size_t()
        self._chk(lib().velo_increment(self.h, frame, _d(T), min_count, _p(ox), _p(oy), _p(oz),
                                       C.byref(cnt)))
        k = cnt.value
        return ox[:k].copy(), oy[:k].copy(), oz[:k].copy()

    def increment_registered_async(self, frame, min_count, pox, poy, poz):
        _order_producers()
        self._chk(lib().velo_increment_registered_async(self.h, frame, min_count, pox, poy, poz))

    def increment_all_registered_async(self, min_count, pox, poy, poz):
        _order_producers()
        self._chk(lib().velo_increment_all_registered_async(self.h, min_count, pox, poy, poz))

    # ---- multi-GPU exchange (RCCL behind the C ABI)
    def comm_init(self, unique_id, rank, world):
        buf = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        assert buf.size == 128
        self._chk(lib().velo_comm_init(self.h, _p(buf), rank, world))

    def comm_info(self):
        r, w = C.c_int32(), C.c_int32()
        self._chk(lib().velo_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def exchange_increments(self, px, py, pz, n_local, pox, poy, poz, cap, after_async_increment=True):
        """-> (counts per rank, total); blocks land in rank order in the output device arrays"""
        _order_producers()
        _, world = self.comm_info()
        counts = np.zeros(max(world, 1), np.int32)
        tot = C.c_size_t()
        self._chk(lib().velo_exchange_increments(self.h, px, py, pz, n_local, int(bool(after_async_increment)),
                                                 pox, poy, poz, cap, _p(counts), C.byref(tot)))
        return counts.tolist(), tot.value

    def exchange_pack_dev(self, precv, counts, pad, pox, poy, poz, cap):
        """the rank-order pack of velo_exchange_increments alone (any world size on one GPU)"""
        _order_producers()
        cnt = np.ascontiguousarray(counts, np.int32)
        tot = C.c_size_t()
        self._chk(lib().velo_exchange_pack_dev(self.h, precv, _p(cnt), cnt.size, pad, pox, poy, poz, cap,
                                               C.byref(tot)))
        return tot.value

    # ---- pending increments (device-side list inside the ctx)
    def increment_pending(self, frame, T=None, min_count=3):
        Tp = None if T is None else _d(np.ascontiguousarray(T, dtype=np.float64).reshape(12))
        self._chk(lib().velo_increment_pending(self.h, frame, Tp, min_count))

    def pending_count(self, wait=True):
        n = C.c_size_t()
        self._chk(lib().velo_pending_count(self.h, C.byref(n), int(bool(wait))))
        return n.value

    def pending_clear(self):
        self._chk(lib().velo_pending_clear(self.h))

    def pending_fetch(self):
        n = self.pending_count(True)
        x, y, z = (np.empty(n, np.float32) for _ in range(3))
        k = C.c_size_t()
        self._chk(lib().velo_pending_fetch(self.h, _p(x), _p(y), _p(z), n, C.byref(k)))
        return x, y, z

    def map_append_pending(self):
        n = C.c_size_t()
        self._chk(lib().velo_map_append_pending(self.h, C.byref(n)))
        return n.value

    def increment_wait(self):
        cnt = C.c_size_t()
        self._chk(lib().velo_increment_wait(self.h, C.byref(cnt)))
        return cnt.value

    def increment_dev(self, frame, T, min_count, pox, poy, poz):
        _order_producers()
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(12)
        cnt = C.c_size_t()
        self._chk(lib().velo_increment_dev(self.h, frame, _d(T), min_count, pox, poy, poz,
                                           C.byref(cnt)))
        return cnt.value
