/*
 * velo.h -- C ABI of libveloslam_amd.so: the MI355X-native scan-to-map
 * registration path for VeloSLAM-style LiDAR frames.
 *
 * The reference (victl/VeloSLAM) has no plugin / FFI layer; the seam this
 * library drops into is the in-process consumer side of HDLManager
 * (HDLManager.h:137-148: waitForFrame/getRecentFrame -> intrusive_ptr<HDLFrame>)
 * plus TransformManager::interpolateTransform (TransformManager.h:108).  Each
 * entry point below names the reference interface it replaces (file:line under
 * the reference tree) or says that the reference has none.  INTEGRATION.md
 * shows the few lines a maintainer adds on the reference side.
 *
 * Conventions: plain pointers and sizes only; every function
 * returns 0 on success or a negative VELO_E_* code, with a message available
 * from velo_last_error(); nothing throws.  A velo_ctx is bound to one GPU and
 * one HIP stream and is single-threaded: use one ctx per GPU / thread.
 * Point clouds are struct-of-arrays float32 (x[], y[], z[]); poses are
 * row-major 3x4 double matrices [R|t] or the reference's PoseTransform fields
 * (metres, roll/pitch/yaw DEGREES, rotation = Ry(roll) Rx(pitch) Rz(yaw),
 * type_defs.h:134-146).
 *
 * Entry points whose name ends in _dev take DEVICE pointers (already resident in
 * HBM, e.g. torch tensors); the others take host pointers and stage them.
 * There is no CPU fallback anywhere: without a GPU velo_create fails.
 *
 * STREAM CONTRACT of the _dev entry points (and of every call that is handed a
 * device pointer: velo_frames_adopt_dev, velo_increment_*_async, velo_exchange_*).
 * The ctx enqueues its work on its OWN stream, created hipStreamNonBlocking: it does
 * NOT synchronise with the NULL stream or with any stream of the caller.  Therefore
 *   (1) whatever PRODUCED the buffers passed in (a fill, a copy, a kernel on the
 *       caller's stream -- e.g. torch.full / tensor.cuda() on torch's stream) must
 *       have COMPLETED, or be ordered before the ctx stream, when the call is made:
 *       synchronise the producing stream first (hipStreamSynchronize /
 *       torch.cuda.synchronize()), or make both sides one stream with
 *       velo_set_stream(ctx, producer_stream);
 *   (2) results written into caller buffers are complete after velo_synchronize(ctx)
 *       (or for work enqueued later on the ctx stream); a consumer on another stream
 *       must wait for that, the ctx does not signal foreign streams;
 *   (3) the buffers must stay allocated and unmodified until then.
 * Host-pointer entry points have no such requirement: they stage and order themselves.
 */
#ifndef VELO_H
#define VELO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: velo_map_info carries a caller-filled struct_size (and grew); velo_cfg carries abi_version,
 *    checked by velo_create; cfg zero values map_subdiv = automatic, linearize_variant = ball */
/* 3: velo_cfg grew the tuning knobs that used to be environment variables (split_iterations ... pair_certificates);
 *    velo_map_size; velo_map_roll_begin may be called with no registration outstanding */
#define VELO_ABI_VERSION 3
#define VELO_MAX_ITERS 64
#define VELO_MAX_RANKS 64   /* ranks of one exchange communicator */
#define VELO_MAX_KNORMALS 32
#define VELO_VARIANT_BALL 1
#define VELO_VARIANT_SCAN 100

enum {
    VELO_OK = 0,
    VELO_E_INVALID = -1,  /* bad argument */
    VELO_E_NOMAP = -2,    /* registration asked before velo_map_reset */
    VELO_E_DEVICE = -3,   /* HIP runtime error (see velo_last_error) */
    VELO_E_NOMEM = -4,
    VELO_E_RANGE = -5,    /* grid beyond the 32-bit fine key / d_max > voxel / too many frames */
    VELO_E_NODATA = -6,   /* empty pose store etc. */
    VELO_E_AGAIN = -7     /* not in this mode: call the plain form (velo_map_roll_overlapped) */
};

typedef struct velo_ctx velo_ctx;

typedef struct velo_cfg {
    uint32_t struct_size;   /* = sizeof(velo_cfg); velo_create refuses a cfg whose abi_version (below)
                               is not VELO_ABI_VERSION: a consumer compiled against another header
                               fails loudly instead of being misread */
    int32_t max_batch;      /* frames registered per launch (default 64) */
    int32_t linearize_variant; /* 0 = default = VELO_VARIANT_BALL (exact pruned ball search);
                               VELO_VARIANT_SCAN = the exhaustive 27-voxel validation kernel
                               (same results, ~10x slower) */
    int32_t sort_frames;    /* 1: order each frame's queries by map cell once per registration */
    int32_t use_graph;      /* 1: replay a registration's launch sequence as one hipGraph
                               (cfg == NULL enables it) */
    int32_t map_subdiv;     /* sub-cells per voxel edge of the map sort order; 0 (default) = chosen
                               at velo_map_reset from the map's density: round(max(1.6 rho^0.2,
                               1.137 rho^0.314)), rho = points per occupied voxel, clamped to [2, 8]; velo_map_info.subdiv
                               reports the value in use */
    int32_t use_hints;      /* temporal coherence, exact either way.  1: bound each query's search by
                               its previous correspondence; 2: also skip the search when last
                               iteration certified that correspondence as the unique nearest
                               point within a radius the query has not left (cfg == NULL: 2) */
    int32_t rounds_per_block; /* tuning: rounds of 256 queries per work item, in every iteration
                               (0 = planned per iteration, see plan_wave_slots); rounded down to 1, 2 or 4
                               rounds per wavefront: a work item is an aligned node of the frame's
                               summation tree.  Whatever is chosen here, or by the planner, or by the
                               batch a frame shares a launch with: the frame's sums, pose and statistics
                               are the same bits ("Reduction") */
    int32_t map_margin;     /* rolling map: the grid is anchored this many voxels below the lowest
                               point and padded as many above, so appends/evictions inside the
                               slack update the sorted map incrementally (default 0 = tight) */
    int32_t map_full_rebuild; /* 1: every append/evict re-sorts and re-estimates the whole map
                               (same result; A/B switch for the incremental update) */
    int32_t map_hash_load;  /* fine-cell table.  0 (default): the dense prefix table while it has
                               < 2^31 entries, an open-addressing hash over the occupied cells
                               (load factor 0.5) beyond that; 5..90: always the hash, at that load
                               factor in percent.  Same sorted order, same results either way. */
    int32_t force_kernel;   /* (also pins velo_knn's kernel and the full-build normals': 1 = one lane per
                               query, 2 = wavefront-cooperative (velo_knn: two queries per wavefront),
                               0 = by the map's density.)
                               0: the linearise kernel is chosen by the size of the registration
                               (latency kernel below 2048 x 256 queries, ~4 frames; throughput kernel
                               above); 1: always the throughput kernel, 2: always the latency kernel
                               (tests hold both to the oracle) */
    int32_t plan_wave_slots; /* tuning / tests: wavefront slots the work-item planner assumes
                               (0 = the device's: CUs x 4 SIMDs x 7).  A batch is cut as coarse as
                               4 rounds per wavefront (1 / 2 / 4: powers of two) while it keeps 2.5
                               wavefronts per slot; a small value makes a small batch take the
                               decompositions of a large one.  Speed only: results do not depend on it */
    uint32_t abi_version;   /* = VELO_ABI_VERSION (the header the caller was compiled against) */
    int32_t reserved[2];
    /* ---- ABI 3: the tuning knobs that used to be read from the environment at velo_create.
     * Speed only, every one of them: results never depend on them (tests hold each setting to the oracle).  0 = the
     * default everywhere.  The old environment variables are still read as MEASUREMENT OVERRIDES of a zero field
     * (tools/ab_*.sh), documented here and nowhere consulted when the field is set. */
    int32_t split_iterations;   /* first iterations run as three launches (stage A of every query / the launch's
                                   stragglers together / the ordinary kernel on certified hints): 0 = default (1, on
                                   the latency path), n > 0 = n, -1 = never.  [VELO_SPLIT_ITERS] */
    int32_t split_batches;      /* 1: ... on the throughput path too (measured slower there).  [VELO_SPLIT_BATCH] */
    int32_t split_per_wave_max; /* stragglers up to which the second launch gives each a wavefront of its own:
                                   0 = default (131072), n > 0 = n, -1 = none (always 64 per wavefront).
                                   [VELO_SPLIT_PER_WAVE_MAX] */
    int32_t solve_threads;      /* workgroup size of k_reduce_solve: 0 = default (1024), 256 / 512 / 1024.
                                   [VELO_SOLVE_THREADS] */
    int32_t roll_cus;           /* CUs the stream of a roll begun ahead may use: 0 = default (three quarters of the
                                   device, the last two CUs of every shader engine left free), n >= 32 = n (rounded
                                   down to whole engines' worth), -1 = no CU mask.  [VELO_ROLL_CUS, VELO_ROLL_NO_CU_MASK] */
    int32_t pair_certificates;  /* latency kernels: pair certificates + no-match certificates with a radius
                                   (round 6): 0 = default (on), -1 = off.  [VELO_NO_PAIR_CERT] */
    int32_t reserved2[2];
} velo_cfg;
/* Environment variables the library still reads, ALL of them measurement aids (A/B scripts under tools/; results never
 * depend on them, and none is needed for any documented behaviour):
 *   tracing            VELO_TRACE_ROLL, VELO_TRACE_REGISTER (MapManager), VELO_KNN_TRACE, VELO_SPLIT_DEBUG
 *   overrides of a ZERO cfg field (see above)   VELO_SPLIT_ITERS, VELO_SPLIT_BATCH, VELO_SPLIT_PER_WAVE_MAX,
 *                      VELO_SOLVE_THREADS, VELO_ROLL_CUS, VELO_ROLL_NO_CU_MASK, VELO_NO_PAIR_CERT
 *   A/B of round 6     VELO_ROLL_LIGHT_MAX (entering points up to which a roll begun ahead runs on the plain stream
 *                      instead of the CU-masked one: default 32768, -1 = never), VELO_NRM_SUBSET_WAVE (the cooperative
 *                      re-estimation kernel for incremental updates), VELO_SORT_MERGE (rocPRIM's merge-sort path),
 *                      VELO_UPDATE_BEFORE_START (MapManager: the pipelined roll begun before the registration's start),
 *                      VELO_KNN_ONE_PER_WAVE (velo_knn: one query per wavefront, the kernel of round 5) */

/* PoseTransform (type_defs.h:86-147) with ptime flattened to microseconds. */
#define VELO_TIME_INVALID INT64_MIN
typedef struct velo_pose {
    double T[3];
    double R[3]; /* roll, pitch, yaw in degrees */
    double V[3];
    int64_t t_us;
    uint16_t week_number;
    uint32_t milliseconds;
    uint32_t week_number_pos;
    double seconds_pos; /* -1 marks "not a valid transform" (type_defs.cxx:56) */
} velo_pose;

typedef struct velo_icp_iter {
    uint32_t n_pairs; /* valid correspondences at the pose BEFORE this iteration's update */
    uint32_t solve_flag; /* 0 ok, 1 diagonal guard used, 2 update skipped */
    double rmse;      /* sqrt(sum r^2 / n_pairs) at that pose */
} velo_icp_iter;

typedef struct velo_icp_result {
    double T[12];     /* final frame->map transform, row-major 3x4 */
    double TRdeg[6];  /* same as T[3] + roll/pitch/yaw degrees (a5 convention) */
    int32_t iters;
    int32_t reserved;
    uint64_t total_pairs; /* sum of n_pairs over the iterations */
    velo_icp_iter iter[VELO_MAX_ITERS];
} velo_icp_result;

typedef struct velo_map_info {
    uint32_t struct_size; /* IN: = sizeof(velo_map_info) of the caller's header; velo_map_info_get
                             writes at most that many bytes (0 is refused) */
    uint32_t reserved0;
    uint64_t n_points;
    uint64_t n_cells;     /* fine cells; cell_start has n_cells + 1 entries */
    float origin[3];
    float voxel;
    float inv_voxel;
    int32_t dims[3];
    int32_t k_normals;
    uint64_t n_invalid_normals;
    int32_t subdiv;       /* n_cells counts FINE cells: prod(dims) * subdiv^3 */
    int32_t last_update;  /* how the last reset/append/evict was applied: 0 = full build on a
                             freshly anchored grid, 1 = incremental on the kept grid */
    uint64_t n_normals_recomputed; /* normals estimated by that update */
    int32_t table_kind;   /* 0 = dense prefix table, 1 = hash over the occupied fine cells */
    int32_t reserved;
    uint64_t table_slots; /* entries of the dense table / slots of the hash */
    uint64_t table_occupied; /* hash: occupied ROW PIECES -- the S fine cells of a voxel along one fine row, the unit the
                                sparse table is keyed by since round 6 (load = occupied / slots; 16 or 32 bytes per slot);
                                dense: 0 */
} velo_map_info;

/* ---- lifetime -------------------------------------------------------------- */
/* No reference counterpart (the reference is CPU-only).  cfg may be NULL. */
velo_ctx* velo_create(int device_id, const velo_cfg* cfg);
void velo_destroy(velo_ctx*);
const char* velo_last_error(const velo_ctx*); /* ctx may be NULL: creation errors */
int velo_abi_version(void);
/* The configuration in effect (defaults filled in: e.g. linearize_variant 0 reads back as
 * VELO_VARIANT_BALL; map_subdiv 0 stays 0 = automatic, see velo_map_info.subdiv). */
int velo_cfg_get(const velo_ctx*, velo_cfg* out);
/* Run all work of this ctx on `hip_stream` (a hipStream_t, e.g. torch's current
 * stream).  NULL = the ctx's own stream. */
int velo_set_stream(velo_ctx*, void* hip_stream);
int velo_synchronize(velo_ctx*);

/* ---- map (the "accumulated MapPatch cloud" of the north star) ------------------
 * The reference's MapPatch holds vector features, not points (MapPatch.h:7-17),
 * and MapManager is unreachable (MapManager.h:15-48, all private): there is no
 * reference call to replace.  These build the voxel-sorted point map that
 * MapManager::registerFrame (include/veloslam/MapManager.hpp) registers against. */
int velo_map_reset(velo_ctx*, const float* x, const float* y, const float* z, size_t n,
                   float voxel, int k_normals);
int velo_map_reset_dev(velo_ctx*, const float* dx, const float* dy, const float* dz, size_t n,
                       float voxel, int k_normals);
/* accepted increment: appended in call order.  The map afterwards equals a fresh
 * build of the whole point list on the map's grid.  The grid (origin, dims) is sticky: it is
 * re-anchored (origin = min - margin*voxel, full rebuild) only when a new point lies below the
 * origin; otherwise the new points are merged into the sorted order, the cell table is shifted
 * and only normals within one voxel of a new point are re-estimated. */
int velo_map_append(velo_ctx*, const float* x, const float* y, const float* z, size_t n);
int velo_map_append_dev(velo_ctx*, const float* dx, const float* dy, const float* dz, size_t n);
/* Voxel-downsampled insertion: of the n points, taken in order, one is appended iff
 * its voxel (on the map's current grid, extended to any integer coordinate) holds fewer than
 * min_count points counting the map's and the points accepted before it -- what integrating the
 * frames one after another would leave.  F frames x W ranks that all see the same under-filled
 * voxel then add min_count points to it, not F*W near-duplicates.  *n_accepted (may be NULL) gets
 * the number appended; the rest is velo_map_append. */
int velo_map_append_sparse(velo_ctx*, const float* x, const float* y, const float* z, size_t n,
                           int min_count, size_t* n_accepted);
int velo_map_append_sparse_dev(velo_ctx*, const float* dx, const float* dy, const float* dz, size_t n,
                               int min_count, size_t* n_accepted);
/* Rolling map (BASELINE configs[2]; the reference's patch eviction policy is unimplemented,
 * MapManager.h:43): drop every map point outside the closed box [lo, hi]; append order of the
 * survivors is kept.  Refused (VELO_E_INVALID, map unchanged) if nothing would remain.  The
 * grid is re-anchored when the lowest survivor is >= 2*margin+2 voxels above the origin. */
int velo_map_evict_outside(velo_ctx*, const float lo[3], const float hi[3]);
/* The same with the region the reference names: keep what lies within `radius` of (x, y) in the
 * ground plane (z free) -- ROI_RANGE, "sensor detecting range" (MapManager.h:13) around the
 * current pose.  Same grid rules and refusal as velo_map_evict_outside. */
int velo_map_evict_radius(velo_ctx*, const float center_xy[2], float radius);
/* A roll of the map -- velo_map_evict_outside(lo, hi) (lo = hi = NULL: none) followed by
 * velo_map_append(x, y, z, n) (n = 0: none) -- on a second stream of the ctx, CONCURRENTLY with the
 * registration a velo_icp_batch_start put on the main stream (valid between that start and its finish,
 * once per registration).  The registration keeps reading the map as it was: the updates write other
 * copies of the sorted arrays, the fine table and the near-voxel flags, and wait on the device for
 * everything older than the registration; whatever is enqueued afterwards sees the new map.  Same map
 * as the two plain calls, bit for bit -- also when the update re-anchors the grid or grows it (the map is then
 * rebuilt into the other copies; round 5) and with a hashed table (its other copy; round 6).  VELO_E_AGAIN (refused before
 * anything changed): a map without normals (k = 0) whose grid would have to move, cfg.map_full_rebuild -- do those with
 * the plain calls after velo_icp_batch_finish.  The call waits for the side stream. */
int velo_map_roll_overlapped(velo_ctx*, const float lo[3], const float hi[3], const float* x, const float* y,
                             const float* z, size_t n);
/* The same roll BEGUN AHEAD of the frame that needs it.  The tile rectangle of a frame comes from the pose track
 * (ROI_RANGE, MapManager.h:13,43), so the host knows it frames before: velo_map_roll_begin -- called, like
 * velo_map_roll_overlapped, between velo_icp_batch_start and _finish, or (round 6) with no registration outstanding at
 * all: the registration started next, while the roll is begun, reads the map as it was -- waits once for the first count (points
 * kept and their bounds, ~0.2 ms) and enqueues the rest of the eviction and the append on a stream of its own
 * without waiting; the registrations that follow keep reading the map AS IT WAS.  velo_map_roll_publish, called at
 * the frame the new rectangle is due (outside a registration or inside one: the main stream waits on the
 * device), switches them to the rolled map -- bit for bit what velo_map_evict_outside + velo_map_append
 * leave.  One roll may be begun at a time; any other map operation (append, evict, reset, download) publishes a
 * begun roll first; velo_map_info_get reports the rolled map at once (its normal counts once the roll is
 * through).  Refusals as for velo_map_roll_overlapped (VELO_E_AGAIN before anything changed). */
int velo_map_roll_begin(velo_ctx*, const float lo[3], const float hi[3], const float* x, const float* y,
                        const float* z, size_t n);
int velo_map_roll_publish(velo_ctx*);
/* Per-axis grid slack in voxels (cfg.map_margin sets all three): a vehicle wants tens of
 * voxels in x/y and one or two in z -- the dense fine-cell table grows with the product.
 * Takes effect at the next (re-)anchoring: velo_map_reset, or the rules above. */
int velo_map_set_margins(velo_ctx*, const int32_t margin[3]);
int velo_map_info_get(velo_ctx*, velo_map_info* out);
/* Points of the map as of the last update -- begun rolls included -- and nothing else: host arithmetic, never waits
 * (velo_map_info_get waits for the normal counts of a roll begun ahead that is still running). */
int velo_map_size(velo_ctx*, uint64_t* n_points);
/* Test / inspection: copy the voxel-sorted map back.  Any pointer may be NULL.
 * perm[s] = index of sorted point s in append order; cell_start has n_cells+1 entries. */
int velo_map_download(velo_ctx*, float* x, float* y, float* z, float* nx, float* ny, float* nz,
                      int32_t* perm, int32_t* cell_start);

/* ---- K1: per-point SE(3) motion compensation --------------------------------------
 * Replaces transformPoint<double>() (type_defs.h:160-166) as applied by
 * HDLParser::vsInternal::pushFiringData (HDLParser.cxx:731-737): point i is
 * multiplied by the affine of its packet, T3x4[pkt[i]], in double, and rounded
 * once to float.  T3x4 comes from velo_packet_transforms (below). */
int velo_compensate(velo_ctx*, const float* x, const float* y, const float* z,
                    const uint16_t* pkt, size_t n, const double* T3x4, size_t n_pkt, float* ox,
                    float* oy, float* oz);
int velo_compensate_dev(velo_ctx*, const float* dx, const float* dy, const float* dz,
                        const uint16_t* dpkt, size_t n, const double* dT3x4, size_t n_pkt,
                        float* dox, float* doy, float* doz);

/* ---- f1: packet decode + calibration + frame split + compensation on the GPU ----------
 * Replaces HDLParser::vsInternal::processHDLPacket / processFiring / pushFiringData /
 * splitFrame (HDLParser.cxx:980-1055, 900-977, 587-752, 867-897) for a packet sequence that
 * starts with fresh parser state: raw 1206-byte packets in (HDLParser.cxx:67-87), frames out as
 * beam-major SoA (HDLFrame::getPointsAsOneCloud order, HDLFrame.cxx:127-144; HDL-64 beams
 * re-ordered by the LUT of HDLParser.cxx:179-187).  Quirks are reproduced: a frame's first
 * packet counts twice, the rest of a packet after a mid-packet split keeps the previous
 * frame's origin, and the packet after a split starts at the split's firing block. */
typedef struct velo_laser_corr { /* HDLLaserCorrection, HDLParser.cxx:89-100, metres/degrees */
    double azimuthCorrection, verticalCorrection, distanceCorrection;
    double verticalOffsetCorrection, horizontalOffsetCorrection;
    double sinVertCorrection, cosVertCorrection;
    double sinVertOffsetCorrection, cosVertOffsetCorrection;
} velo_laser_corr;
/* packets: n_pkt x 1206 bytes (host).  poses: sorted pose store (n_poses may be 0: frames stay
 * in the sensor frame, as when the reference has no valid transform).  flush != 0 also emits the
 * unfinished last frame (HDLParser::getFrame's tail, HDLParser.cxx:541).  crop_region: NULL or
 * {xmin,xmax,ymin,ymax,zmin,zmax} (HDLParser.cxx:629-639).  Results stay on the device. */
int velo_decode(velo_ctx*, const uint8_t* packets, const int64_t* pkt_t_us, size_t n_pkt,
                const velo_laser_corr corr[64], int n_lasers, const velo_pose* poses,
                size_t n_poses, int flush, const double* crop_region, int crop_inside,
                int32_t* n_frames, size_t* n_points);
/* The same parser, stateful across calls like the reference's (HDLSource feeds HDLParser one
 * packet at a time, HDLSource.cxx:209-225): packets may arrive in chunks of any size, a call
 * emits the frames whose closing split lies in its packets, and the packets that still hold
 * firing blocks of the unfinished frame are kept inside the ctx for the next call (n_pkt may be
 * 0 with flush != 0 to drain).  Emitted frames are identical to one velo_decode over the whole
 * sequence; packet_index counts from the first packet still in flight.  flush or
 * velo_decode_stream_reset() return the parser to its initial state. */
int velo_decode_stream(velo_ctx*, const uint8_t* packets, const int64_t* pkt_t_us, size_t n_pkt,
                       const velo_laser_corr corr[64], int n_lasers, const velo_pose* poses,
                       size_t n_poses, int flush, const double* crop_region, int crop_inside,
                       int32_t* n_frames, size_t* n_points);
int velo_decode_stream_reset(velo_ctx*);
/* The parser's remaining knobs, sticky on the ctx like the reference's setters (defaults: every
 * laser, no skipping, start at block 0):
 *   laser_selection[i] != 0  laser i is kept -- setLaserSelection (HDLParser.h:106-110, consumed at
 *                            HDLParser.cxx:964; i = the laser id inside the frame: 0..63, or 0..15
 *                            for a 16-laser sensor);
 *   points_skip = k          only firing blocks with block % (k+1) == 0 are decoded --
 *                            setPointsSkip (HDLParser.h:119, HDLParser.cxx:1042); frame splitting
 *                            still looks at every block;
 *   initial_firing_skip      the first packet of a parse that starts from fresh state begins at this
 *                            block -- the `skip` argument of the offline re-read HDLParser::getFrame
 *                            (HDLParser.cxx:505-544, consumed at :1013). */
typedef struct velo_decode_opts {
    uint32_t struct_size; /* = sizeof(velo_decode_opts) */
    int32_t points_skip;
    int32_t initial_firing_skip;
    uint8_t laser_selection[64];
} velo_decode_opts;
int velo_decode_set_options(velo_ctx*, const velo_decode_opts* opts); /* NULL = defaults */
/* velo_decode in its two halves.  The HOST half (the sequential part of the parser: pose
 * interpolation and the 3x4 table per packet, frame splits, the 12 block owners per packet; staged
 * in pinned memory the plan owns) touches neither the ctx nor the GPU: it may run while the ctx is
 * busy registering the previous frame -- from the same thread between an asynchronous registration
 * and its fetch, or from another thread (one thread per plan).  The DEVICE half consumes the plan.
 * fill + submit == velo_decode of the same arguments with `opts` in place of the ctx's sticky
 * velo_decode_set_options (NULL = defaults); always a parse from fresh state.  A plan is reusable
 * (fill, submit, fill, ...); filling twice overwrites; submitting an unfilled plan is an error.
 * velo_decode_plan_fill reports through velo_decode_plan_error, not velo_last_error. */
typedef struct velo_decode_plan velo_decode_plan;
int velo_decode_plan_create(velo_ctx*, velo_decode_plan** out);
void velo_decode_plan_destroy(velo_decode_plan*);
int velo_decode_plan_fill(velo_decode_plan*, const velo_decode_opts* opts, const uint8_t* packets,
                          const int64_t* pkt_t_us, size_t n_pkt, const velo_laser_corr corr[64],
                          int n_lasers, const velo_pose* poses, size_t n_poses, int flush,
                          const double* crop_region, int crop_inside);
int velo_decode_submit(velo_ctx*, velo_decode_plan*, int32_t* n_frames, size_t* n_points);
/* velo_decode_submit + velo_decode_to_frames on a second stream of the ctx, CONCURRENTLY with the
 * registration a velo_icp_batch_start put on the main stream (valid only between that start and its
 * finish, once per registration; call velo_increment_pending for the frames being registered first:
 * afterwards the resident frames are the new ones).  The decode writes the output set and the plan
 * copy the running registration does not use (both alternate from call to call), waits on the device
 * for everything older than that registration, and the main stream is held behind it, so whatever is
 * enqueued next sees the new frames complete.  The call waits for the side stream only. */
int velo_decode_submit_overlapped(velo_ctx*, velo_decode_plan*, int32_t* n_frames, size_t* n_points);
const char* velo_decode_plan_error(const velo_decode_plan*);
/* Copy the last decode back; any pointer may be NULL.  frame_start: n_frames+1; beam_start:
 * n_frames x 65 (absolute offsets); packet_index: index of the source packet of each point. */
int velo_decode_fetch(velo_ctx*, float* x, float* y, float* z, float* intensity, uint16_t* azimuth,
                      float* distance, uint16_t* packet_index, int64_t* frame_start,
                      int32_t* beam_start, velo_pose* carposes, int64_t* frame_t_us,
                      int32_t* frame_packets);
/* The decoded frames become the resident frames of velo_icp_batch (no host round-trip).  They
 * stay resident until the next velo_decode / velo_decode_stream on this ctx, which rewrites
 * those buffers: that call ends the adoption (registration then fails with VELO_E_INVALID until
 * velo_decode_to_frames / velo_frames_upload / velo_frames_adopt_dev is called again). */
int velo_decode_to_frames(velo_ctx*);

/* ---- K2+K3+solve: scan-to-map ICP (no reference counterpart) -----------
 * Point-to-plane Gauss-Newton, exactly `iters` iterations, nearest neighbour
 * within d_max over the 27 map cells around the transformed point.  k must be 1
 * in this ABI version (k-NN output is a later row).  Semantics: DESIGN.md
 * "ICP semantics" == oracle/icp.c. */
int velo_icp(velo_ctx*, const float* x, const float* y, const float* z, size_t n,
             const double T0[12], int iters, float d_max, int k, velo_icp_result* out);

/* Batched / resident form used for throughput: upload (or adopt) `n_frames`
 * frames concatenated SoA, frame f = [frame_start[f], frame_start[f+1]). */
int velo_frames_upload(velo_ctx*, int n_frames, const float* x, const float* y, const float* z,
                       const int64_t* frame_start);
int velo_frames_adopt_dev(velo_ctx*, int n_frames, const float* dx, const float* dy,
                          const float* dz, const int64_t* frame_start /* host */);
/* Register every resident frame against the current map snapshot.
 * T0: n_frames x 12 ; out: n_frames results. */
int velo_icp_batch(velo_ctx*, const double* T0, int iters, float d_max, velo_icp_result* out);
/* Same, but leaves poses/statistics on the device and does not synchronise
 * (bench inner loop); fetch with velo_icp_batch_fetch. */
int velo_icp_batch_async(velo_ctx*, const double* T0, int iters, float d_max);
int velo_icp_batch_fetch(velo_ctx*, velo_icp_result* out);
/* The pipelined pair for a stream of frames: start = velo_icp_batch_async + the result's copy to
 * pinned host memory + an event; finish waits for that event ONLY and hands the results out.  Work
 * enqueued on the ctx between the two -- velo_increment_pending of the frame, the NEXT frame's
 * velo_decode_submit + velo_decode_to_frames -- is not waited for by finish and does not disturb it
 * (the result belongs to the frames that were resident at start).  One start may be outstanding. */
int velo_icp_batch_start(velo_ctx*, const double* T0, int iters, float d_max);
int velo_icp_batch_finish(velo_ctx*, velo_icp_result* out);

/* Diagnostics for parity tests on resident frame `frame`: one linearisation at
 * pose T.  corr (sorted map index or -1) and d2 may be NULL; acc = the 29
 * doubles (21 upper-triangular JtJ, 6 Jtr, sum r^2, count). */
int velo_linearize(velo_ctx*, int frame, const double T[12], float d_max, int32_t* corr,
                   float* d2, double acc[29]);

/* a10 with k neighbours (k <= VELO_MAX_KNORMALS): for every point of resident frame `frame`,
 * transformed by T, the k nearest map points within d_max, ascending by (d2, sorted map
 * index).  idx and d2 are n x k row-major, padded with -1 / +inf; count (may be NULL) gets the
 * number found per query.  No reference counterpart. */
int velo_knn(velo_ctx*, int frame, const double T[12], float d_max, int k, int32_t* idx, float* d2,
             int32_t* count);
/* The same with the results left on the DEVICE (d_idx, d_d2: n x k; d_count may be NULL), enqueued on the ctx
 * stream, nothing fetched and nothing waited for (the pose is a kernel argument): what a caller that consumes the
 * neighbours on the GPU uses, and what bench.py times
 * for BASELINE configs[4] (100 M-point map, k = 32).  stats (may be NULL) switches to the COUNTING
 * instantiation of the kernel and waits for it: [0] queries, [1] candidate points fetched (16 B each),
 * [2] fine rows looked up in the table, [3] fine cells those rows span -- the kernel's byte accounting. */
int velo_knn_dev(velo_ctx*, int frame, const double T[12], float d_max, int k, int32_t* d_idx, float* d_d2,
                 int32_t* d_count, uint64_t stats[4]);

/* a12 alone, on the device (the kernel every registration iteration runs): the 29 sums of one
 * linearisation -> 6x6 LDLt (diagonal guard on a non-positive pivot) -> T <- exp(xi^) T.
 * T is updated in place; solve_flag (may be NULL): 0 ok, 1 guard used, 2 no update (fewer than
 * 6 pairs or singular).  No reference counterpart. */
int velo_solve_update(velo_ctx*, const double acc[29], double T[12], int32_t* solve_flag);

/* mode 1: velo_linearize remembers each query's correspondence and uses it as the search-
 * radius hint of the next call (what velo_icp_batch does between iterations); mode 0: every
 * call searches from scratch.  Either way the results are identical -- this exists so tests
 * can hold the hinted path to the oracle too.  Calling it forgets the stored hints. */
int velo_linearize_hints(velo_ctx*, int mode);

/* Accepted map increment of resident frame `frame` under pose T: points that land
 * in a map cell holding fewer than min_count points, order preserving.  Outputs
 * have room for the whole frame.  _dev writes device buffers (for the RCCL
 * all-gather) and returns the count in *n_out. */
int velo_increment(velo_ctx*, int frame, const double T[12], int min_count, float* ox, float* oy,
                   float* oz, size_t* n_out);
int velo_increment_dev(velo_ctx*, int frame, const double T[12], int min_count, float* dox,
                       float* doy, float* doz, size_t* n_out);
/* Pipelined form for the exchange step: the increment of a frame of the LAST
 * registration at the pose that registration left on the device -- nothing is fetched, nothing
 * blocks; outputs are device arrays of at least that frame's size.  velo_increment_wait()
 * blocks only until the increment itself is done (not for work enqueued after it, e.g. the
 * next batch) and returns the count. */
int velo_increment_registered_async(velo_ctx*, int frame, int min_count, float* dox, float* doy,
                                    float* doz);
/* The same for EVERY resident frame of the last registration in one pass: the concatenation, in
 * frame order, of the per-frame increments (each frame at its own registered pose, all against
 * the same map snapshot).  Outputs: device arrays with room for all resident points. */
int velo_increment_all_registered_async(velo_ctx*, int min_count, float* dox, float* doy, float* doz);
int velo_increment_wait(velo_ctx*, size_t* n_out);

/* Pending increments: the accepted points of registered frames, collected in a device-side list
 * inside the ctx until they are worth a map update -- what MapManager::registerFrame's integrate
 * step and a streaming host use.  Nothing is fetched and nothing blocks per frame:
 *   velo_increment_pending    the increment of resident frame `frame` (pose T, or NULL = the pose
 *                             the last registration left on the device) is appended to the list;
 *                             returns as soon as the work is enqueued;
 *   velo_pending_count        points in the list; wait != 0 waits for the increment in flight and
 *                             counts it, wait = 0 never blocks and never counts it (deterministic:
 *                             the answer does not depend on how far the GPU has got);
 *   velo_pending_fetch        host copy of the list (for a host-side tile store); VELO_E_RANGE with
 *                             *n_out = the size needed beyond cap;
 *   velo_map_append_pending   merges the list into the map (velo_map_append_dev) and empties it;
 *   velo_pending_clear        empties it without touching the map -- for a host that fetched the
 *                             list and decides itself what goes back up (MapManager: the points
 *                             inside the resident tiles, folded into the next roll's one append). */
int velo_increment_pending(velo_ctx*, int frame, const double* T, int min_count);
int velo_pending_count(velo_ctx*, size_t* n, int wait);
int velo_pending_fetch(velo_ctx*, float* x, float* y, float* z, size_t cap, size_t* n_out);
int velo_map_append_pending(velo_ctx*, size_t* n_appended);
int velo_pending_clear(velo_ctx*);

/* Valid correspondence pairs processed by every registration iteration of this ctx since the
 * last reset, counted on the device (exact; lets a caller that never fetches per-batch results
 * -- the bench loop -- report pairs/s).  Synchronises the ctx stream. */
int velo_pairs_total(velo_ctx*, uint64_t* out, int reset);

/* ---- multi-GPU exchange step (no reference counterpart: the reference is one
 * process, one CPU) ---------------------------------------------------------------------------
 * One process per GPU, one ctx per process.  Rank 0 makes an id (velo_comm_unique_id), the host
 * application carries its 128 bytes to the other ranks by whatever it has (MPI, a socket, a
 * file, torch.distributed), every rank calls velo_comm_init: an RCCL communicator over xGMI,
 * opened with dlopen at that moment -- the library does not link against RCCL. */
#define VELO_COMM_ID_BYTES 128
int velo_comm_unique_id(uint8_t id[VELO_COMM_ID_BYTES]);
int velo_comm_init(velo_ctx*, const uint8_t id[VELO_COMM_ID_BYTES], int rank, int world);
int velo_comm_destroy(velo_ctx*);
int velo_comm_info(velo_ctx*, int32_t* rank, int32_t* world); /* world = 0: no communicator */
/* The host-side half of the exchange, on its own (no ctx, no GPU): the per-rank counts the first
 * all-gather returned -> offsets[r] = first output index of rank r's block (offsets has world + 1
 * entries, offsets[world] = total), *pad = floats per axis of every padded block (the largest
 * count, at least 1), *total = points of all ranks.  VELO_E_INVALID: world outside
 * 1..VELO_MAX_RANKS or a negative count; VELO_E_RANGE: total beyond 2^32 - 1. */
int velo_exchange_plan(const int32_t* counts, int world, uint32_t* offsets, size_t* pad, size_t* total);
/* The device-side half on its own: `recv` (device) laid out as the padded all-gather leaves it --
 * rank r's block at recv + r*3*pad floats, [x | y | z] each padded to pad -- is packed in RANK
 * ORDER into ox/oy/oz (one kernel, stream-ordered on the ctx stream).  This is the step
 * velo_exchange_increments runs after its second all-gather; exported so that it can be held to a
 * reference for any world size on one GPU. */
int velo_exchange_pack_dev(velo_ctx*, const float* recv, const int32_t* counts, int world, size_t pad,
                           float* ox, float* oy, float* oz, size_t cap, size_t* n_total);
/* All-gather of every rank's accepted increment (device SoA, n_local points; may be 0): counts
 * first, then max-padded blocks, packed in RANK ORDER into the device arrays ox/oy/oz (capacity
 * cap points) -- appending that list with velo_map_append_dev keeps every replica of the map
 * bit-identical.  Every rank must pass the same cap (a rank that returned early would leave the
 * others inside the second collective: the capacity is checked after it, VELO_E_RANGE on every
 * rank alike).  counts (host, world entries, may be NULL) and *n_total are valid on return; the
 * blocks are complete for work enqueued on the ctx stream after the call.  The collectives run on
 * the ctx's own communication stream: with after_async_increment != 0 they wait only for the
 * last velo_increment_*_async, not for a batch enqueued behind it (overlap with the next batch);
 * with 0 they wait for everything enqueued on the ctx stream so far. */
int velo_exchange_increments(velo_ctx*, const float* dx, const float* dy, const float* dz, size_t n_local,
                             int after_async_increment, float* ox, float* oy, float* oz, size_t cap,
                             int32_t* counts, size_t* n_total);

/* per-kernel device time of the last velo_icp_batch* call, HIP events on the ctx
 * stream: [0] linearise kernel total ms, [1] its launch count, [2] solve total ms,
 * [3] solve launches, [4] whole call ms, [5] first linearise launch ms (no hints yet),
 * [6] fastest linearise launch ms */
int velo_last_timing(velo_ctx*, double out[8]);
/* Device time (microseconds) of every linearise launch of that call, in launch order; returns
 * the number of launches (>= 0; may exceed cap, only cap values are written). */
int velo_last_linearize_us(velo_ctx*, float* out, int cap);
/* enable (1) / disable (0) per-launch event timing (adds event records) */
int velo_set_timing(velo_ctx*, int on);
/* Search statistics and byte accounting of the linearise kernel.  velo_set_stats(ctx, 1) makes
 * every following linearise launch use the COUNTING instantiation of the kernel (same results,
 * slower; the production instantiation carries no counters); velo_search_stats reads the
 * process-wide totals since the last reset: [0] live queries, [1] certified without search,
 * [2] searched, [3] empty-neighbourhood skips, [4] stage-A final, [5] stragglers searched per
 * lane, [6] stage-B stragglers (all), [7] valid pairs, [8] BYTES the kernel requested from
 * memory (loads + stores: the roofline numerator bench.py uses), [9] candidate points examined,
 * [10] fine-table requests, [11] launches, [12] the query-side part of [8] (per-query stream and
 * per-workgroup rows; the rest of [8] are gathers from the map: points, normals, fine table).
 * velo_debug_search_stats = the first 8. */
int velo_set_stats(velo_ctx*, int on);
int velo_search_stats(velo_ctx*, uint64_t out[16], int reset);
int velo_debug_search_stats(velo_ctx*, uint64_t out[8], int reset);

/* ---- host-side pose plumbing -------------------------------------------------------- */
/* PoseTransform::getMatrix (type_defs.h:134-146) and its inverse. TRdeg = T[3] + R[3]. */
int velo_matrix_from_pose(const double TRdeg[6], double T[12]);
int velo_pose_from_matrix(const double T[12], double TRdeg[6]);
/* TransformManager::interpolateTransform (TransformManager.cxx:149-177) over a
 * caller-held array sorted by t_us.  Returns 0 and fills *out on "true", VELO_E_NODATA on
 * "false" (empty store). out->seconds_pos keeps the reference's validity signal.
 * PRECONDITION of every entry point that takes a pose array (this one, velo_packet_transforms, velo_decode*,
 * velo_decode_plan_fill): `sorted` is STRICTLY ascending in t_us over its whole length, as
 * TransformManager's store is by construction (a map keyed by time).  The library spot-checks the ends and the
 * samples around the bracket it uses (VELO_E_INVALID on a violation it sees); an unsorted or duplicated stretch
 * elsewhere is NOT detected and yields the bracket a binary search happens to land on.  Sort and de-duplicate
 * before the call (veloslam::TransformManager::snapshot() does). */
int velo_interp_pose(const velo_pose* sorted, size_t n, int64_t t_us, velo_pose* out);
/* What HDLParser::processHDLPacket does per packet before the firing loop
 * (HDLParser.cxx:988-1007): interpolate at each packet time, take packet 0's pose as
 * carpose, subtract carpose.T (reprojectToFrameBeginning, :1057-1062), getMatrix.
 * T3x4: n_pkt x 12.  valid[i]=0 where the reference would have left geotransform null
 * (identity is written there).  carpose may be NULL. */
int velo_packet_transforms(const velo_pose* sorted, size_t n, const int64_t* pkt_t_us,
                           size_t n_pkt, double* T3x4, uint8_t* valid, velo_pose* carpose);

/* ---- f2 / f4: the formats on either side of the path (host only) ---------------------- */
/* pcap of 1206-byte lidar packets, no libpcap: what vtkPacketFileWriter::writePacket
 * (vtkPacketFileWriter.cxx:118-161) produces and vtkPacketFileReader::nextPacket
 * (vtkPacketFileReader.h:166-197) consumes -- 24 B global header, per packet 16 B record header
 * + 42 B Ethernet/IPv4/UDP prefix + payload (PCAP_PACKET_LEN 1264, vtkPacketFileReader.h:66).
 * Time stamps are epoch microseconds (the reference's +8 h shift, type_defs.cxx:69-72, is not
 * applied).  velo_pcap_read: packets/t_us may be NULL to count; returns VELO_E_RANGE (with
 * *n_out = cap) when the file holds more than cap packets. */
int velo_pcap_write(const char* path, const uint8_t* packets, const int64_t* t_us, size_t n_pkt);
int velo_pcap_read(const char* path, uint8_t* packets, int64_t* t_us, size_t cap, size_t* n_out);
/* Frame index of a capture -- HDLParser::readFrameInformation (HDLParser.cxx:1065-1160), consumed
 * by HDLManager::loadOffline (HDLManager.cxx:103-117: fileStartPos / skips / timestamp per frame)
 * and by the offline re-read HDLParser::getFrame (HDLParser.cxx:505-544).  One pass over the file
 * that looks only at the rotational position of every firing block: frame 0 starts at the first
 * record (position 24, skip 0, time of the first lidar packet; present even for an empty capture,
 * with t_us = VELO_TIME_INVALID, as the reference pushes it before reading); every block whose
 * raw azimuth is below the previous block's opens a frame at { position of the record holding it,
 * block index, that packet's time }.  first_packet (not in the reference) = ordinal of that packet
 * among the 1206-byte packets, i.e. the index into what velo_pcap_read returns: decoding
 * packets [first_packet, ...) with velo_decode_opts.initial_firing_skip = firing_skip reproduces
 * the frame.  frames may be NULL to count; VELO_E_RANGE (with *n_out = frames found) beyond cap. */
typedef struct velo_frame_index {
    int64_t file_pos;       /* HDLFrame::fileStartPos */
    int32_t firing_skip;    /* HDLFrame::skips */
    int32_t reserved;
    int64_t first_packet;
    int64_t t_us;           /* HDLFrame::timestamp */
} velo_frame_index;
int velo_pcap_index(const char* path, velo_frame_index* frames, size_t cap, size_t* n_out);
/* InsPVA wire struct (type_defs.h:39-58, natural alignment: 104 bytes) */
typedef struct velo_inspva {
    uint16_t message_id;
    uint16_t week_number;
    uint32_t milliseconds;
    uint32_t week_number_pos;
    double seconds_pos;
    double LLH[3];  /* degrees, degrees, metres */
    double V[3];
    double Eulr[3]; /* degrees */
    int32_t ins_status;
} velo_inspva;
/* INSSource::calcTransform (INSSource.cxx:305-326) with the origin as a parameter */
int velo_ins_to_pose(const velo_inspva* ins, const double orig_xyz[3], int64_t t_us, velo_pose* out);
/* pose-store persistence: the record layout of type_defs.cxx:4-33 (ptime -> int64 us) */
int velo_insmeta_write(const char* path, const velo_pose* poses, size_t n);
int velo_insmeta_read(const char* path, velo_pose* poses, size_t cap, size_t* n_out);
/* carposes.txt of an offline drive -- TransformManager::loadFromTxtFile (TransformManager.cxx:95-125),
 * consumed by HDLManager::loadOffline (HDLManager.cxx:103-117): rows "x y yaw roll pitch v sec usec",
 * angles in radians (converted to degrees, yaw sign flipped), no z and no velocity vector (both 0),
 * time = sec * 1e6 + usec + 8 h (the reference's timevalToPtime, type_defs.cxx:69-72, adds the 8 h to
 * pose AND packet stamps; a replay adds them to the packet stamps of velo_pcap_read, which does not).
 * Poses come back sorted by time; poses may be NULL to count; VELO_E_RANGE beyond cap. */
int velo_carposes_read(const char* path, velo_pose* poses, size_t cap, size_t* n_out);
/* ptimeToWeekMilli (type_defs.cxx:74-79; time in microseconds, the reference's +8 h already in it):
 * *week = ISO 8601 week number of the date (boost::gregorian::date::week_number), *milli =
 * milliseconds since the preceding Sunday 00:00.  Either pointer may be NULL.  velo_carposes_read
 * fills week_number, milliseconds, week_number_pos (= week) and seconds_pos (= milli / 1000.0f,
 * divided in float) from it, as TransformManager.cxx:116-119 does. */
void velo_time_to_week_milli(int64_t t_us, uint16_t* week, uint32_t* milli);
/* Velodyne calibration file (db.xml) -> the 64 laser corrections velo_decode takes.  Replaces
 * HDLParser::vsInternal::loadCorrectionsFile (HDLParser.cxx:771-858): same element names, same
 * units (centimetres in the file, metres afterwards) and derived sin/cos fields; n_enabled
 * (optional) = number of `enabled_` items equal to 1 (calibFileReportedNumLasers). */
int velo_load_corrections(const char* path, velo_laser_corr corr[64], int32_t* n_enabled);

/* ---- CoordiTran (CoordiTran.h:7-15): reference names and signatures verbatim -------- */
void eulr2dcm(double eul_vect[3], double DCMbn[3][3]);
void llh2xyz(double llh[3], double xyz[3]);
void xyz2llh(double xyz[3], double llh[3]);
void xyz2enu(double xyz[3], double orgxyz[3], double enu[3]);
void enu2xyz(double enu[3], double orgxyz[3], double xyz[3]);
void enu2llh(double enu[3], double orgxyz[3], double llh[3]);
void llh2enu(double llh[3], double orgxyz[3], double enu[3]);
double MappingAngle(double angle);
/* HDL2enu (CoordiTran.h:12) is intentionally absent: the reference body reads an
 * uninitialised array (CoordiTran.cpp:232,251) and nothing calls it. */

#ifdef __cplusplus
}
#endif
#endif /* VELO_H */
