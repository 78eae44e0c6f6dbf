/*
 * velo_oracle.h -- CPU oracle for the scan-to-map registration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under veloslam_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Two halves:
 *   (A) a1..a8, a13: plain-C restatement of what the reference computes on the
 *       CPU.  Every function cites the reference file:line it follows.
 *       - CoordiTran (a1, a2) is PINNED: checked bit-for-bit against the
 *         reference's own object code (oracle/_ref, built from
 *         /root/reference/CoordiTran.cpp) through tests/golden/coorditran.json.
 *       - type_defs.h / TransformManager / TimeLine / HDLParser (a3..a8) need
 *         Eigen, Boost, PCL, glog, pcap -- none installed, none vendored -- so
 *         the reference cannot be built here.  PARITY UNPINNED for these rows
 *         beyond restatement-by-reading plus property tests.  Third-party
 *         algorithm restated: Eigen (version unpinned by the reference's
 *         CMakeLists.txt:78) AngleAxis::toRotationMatrix and
 *         Transform::rotate (right-multiply).
 *   (B) a9..a12: the reference has NO ICP.  The fp64 ICP defined
 *       here IS the specification the HIP kernels are held to.  PARITY
 *       UNPINNED with respect to the reference by construction.
 *
 * All arithmetic that must be bit-identical on CPU and GPU is written with
 * explicit fma()/fmaf() and compiled with -ffp-contract=off.
 */
#ifndef VELO_ORACLE_H
#define VELO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ a1, a2 */
/* CoordiTran.cpp:4-49, 51-81, 82-150, 152-187, 189-219, 264-269, 271-276, 278-293 */
void vo_eulr2dcm(const double eul[3], double dcm_bn[9]);
void vo_llh2xyz(const double llh[3], double xyz[3]);
void vo_xyz2llh(const double xyz[3], double llh[3]);
void vo_xyz2enu(const double xyz[3], const double orgxyz[3], double enu[3]);
void vo_enu2xyz(const double enu[3], const double orgxyz[3], double xyz[3]);
void vo_enu2llh(const double enu[3], const double orgxyz[3], double llh[3]);
void vo_llh2enu(const double llh[3], const double orgxyz[3], double enu[3]);
double vo_mapping_angle(double angle);

/* ------------------------------------------------------------------ a3..a7 */
#define VO_TIME_INVALID INT64_MIN /* boost not_a_date_time stand-in */

/* type_defs.h:86-147 ; ptime -> int64 microseconds */
typedef struct vo_pose {
    double T[3];
    double R[3]; /* roll, pitch, yaw in DEGREES */
    double V[3];
    int64_t t_us;
    uint16_t week_number;
    uint32_t milliseconds;
    uint32_t week_number_pos;
    double seconds_pos; /* -1 == invalid sentinel (type_defs.cxx:56) */
} vo_pose;

void vo_pose_init(vo_pose* p);                                   /* type_defs.cxx:47-57 */
void vo_pose_add(const vo_pose* a, const vo_pose* b, vo_pose* o); /* type_defs.h:102-114 */
void vo_pose_sub(const vo_pose* a, const vo_pose* b, vo_pose* o); /* type_defs.h:124-131 */
void vo_pose_scale(const vo_pose* a, double r, vo_pose* o);       /* type_defs.h:115-123 */
void vo_pose_matrix(const vo_pose* p, double M[12]);              /* type_defs.h:134-146 */
void vo_matrix_to_TRdeg(const double M[12], double TRdeg[6]);     /* inverse of a5 */
void vo_transform_point(double pt[3], const double M[12]);        /* type_defs.h:160-166 */

/* TimeLine.h (bucketed index + 5-slot ring), literal restatement */
typedef struct vo_timeline vo_timeline;
vo_timeline* vo_timeline_new(void);
void vo_timeline_free(vo_timeline*);
size_t vo_timeline_size(const vo_timeline*);
void vo_timeline_add(vo_timeline*, const vo_pose* p);             /* TimeLine.h:140-226 */
/* TimeLine.h:384-468 ; returns number of valid ends (0,1,2) */
int vo_timeline_boundary(const vo_timeline*, int64_t t_us, vo_pose* fore, vo_pose* back);
/* TransformManager.cxx:149-177 ; returns 1 (true) / 0 (false) */
int vo_interpolate_transform(const vo_timeline*, int64_t t_us, vo_pose* out);
/* ptimeToWeekMilli, type_defs.cxx:74-79 (Boost.DateTime week_number restated) */
void vo_time_to_week_milli(int64_t t_us, uint16_t* week, uint32_t* milli);

/* a7: K1 restated.  Per point i: M = table[pkt[i]] ; out = (float)(M*p). */
void vo_compensate(const float* x, const float* y, const float* z, const uint16_t* pkt,
                   size_t n, const double* T3x4, size_t n_pkt,
                   float* ox, float* oy, float* oz);

/* ---------------------------------------------------------------------- a8 */
typedef struct vo_laser_corr { /* HDLParser.cxx:89-100 */
    double azimuthCorrection, verticalCorrection, distanceCorrection;
    double verticalOffsetCorrection, horizontalOffsetCorrection;
    double sinVertCorrection, cosVertCorrection;
    double sinVertOffsetCorrection, cosVertOffsetCorrection;
} vo_laser_corr;

typedef struct vo_decoder vo_decoder;
/* n_lasers: calibFileReportedNumLasers (64/32/16) */
/* HDLParser.cxx:771-858 loadCorrectionsFile; 0 ok, -1 unreadable */
int vo_load_corrections(const char* path, vo_laser_corr corr[64], int* n_enabled);
vo_decoder* vo_decoder_new(const vo_laser_corr corr[64], int n_lasers, const vo_timeline* tl);
void vo_decoder_free(vo_decoder*);
void vo_decoder_set_crop(vo_decoder*, int enable, int crop_inside, const double region[6]);
void vo_decoder_set_skip(vo_decoder*, int firing_skip);
void vo_decoder_set_laser_selection(vo_decoder*, const unsigned char sel[64]);
void vo_decoder_set_points_skip(vo_decoder*, int points_skip);
/* HDLParser.cxx:980-1055.  Returns number of frames completed so far. */
int vo_decoder_packet(vo_decoder*, const unsigned char* data, size_t len, int64_t t_us);
int vo_decoder_flush(vo_decoder*); /* splitFrame(force) like getFrame's tail, :541 */
int vo_decoder_num_frames(const vo_decoder*);
/* Frame accessors; beam index AFTER the HDL64BeamLUT permutation (:880-893). */
size_t vo_frame_beam_size(const vo_decoder*, int frame, int beam);
/* copies xyzi (float) + azimuth(u16) + distance(float) for one beam */
void vo_frame_beam_copy(const vo_decoder*, int frame, int beam, float* x, float* y, float* z,
                        float* intensity, uint16_t* azimuth, float* distance);
void vo_frame_carpose(const vo_decoder*, int frame, vo_pose* out, int64_t* frame_t_us, int* skips);
size_t vo_frame_num_packets(const vo_decoder*, int frame);

/* ------------------------------------------------------------------ a9..a12 */
typedef struct vo_map vo_map;
/* voxel: cell edge h.  k_normals: neighbours for PCA (<=32). */
vo_map* vo_map_build(const float* x, const float* y, const float* z, size_t n, float voxel,
                     int k_normals);
/* subdiv: sub-cells per voxel edge (part of the sort order, hence of the spec); 3 above */
/* density-chosen sub-division (subdiv == 0 in vo_map_build_ex / vo_roll_new*): see oracle/icp.c */
int vo_auto_subdiv(const float* x, const float* y, const float* z, size_t n, float voxel);
vo_map* vo_map_build_ex(const float* x, const float* y, const float* z, size_t n, float voxel,
                        int k_normals, int subdiv);
/* fresh build on an explicit grid: origin (<= min of the points, NULL = min) and a lower
 * bound on the voxel dims (NULL = tight) */
vo_map* vo_map_build_grid(const float* x, const float* y, const float* z, size_t n, float voxel,
                          int k_normals, int subdiv, const float* origin, const int* dims_min);
int vo_map_subdiv(const vo_map*);
void vo_map_free(vo_map*);

/* f3 / configs[2]: rolling map = raw list + sticky grid + margin; after every operation the
 * map equals vo_map_build_grid(raw list, grid).  Rules in icp.c.
 * append: 0 = grid kept, 1 = re-anchored, -1 = failed.
 * evict_outside: 0 = nothing removed, 1 = re-anchored, 2 = removed with the grid kept,
 * -1 = refused (nothing would remain). */
typedef struct vo_roll vo_roll;
vo_roll* vo_roll_new(const float* x, const float* y, const float* z, size_t n, float voxel,
                     int k_normals, int subdiv, int margin);
vo_roll* vo_roll_new3(const float* x, const float* y, const float* z, size_t n, float voxel,
                      int k_normals, int subdiv, const int margin[3]);
void vo_roll_free(vo_roll*);
const vo_map* vo_roll_map(const vo_roll*);
size_t vo_roll_size(const vo_roll*);
int vo_roll_append(vo_roll*, const float* x, const float* y, const float* z, size_t m);
int vo_roll_evict_outside(vo_roll*, const float lo[3], const float hi[3]);
/* voxel-downsampled insertion: accept[i] = 1 iff new point i's voxel (current grid) holds fewer
 * than min_count points counting the map's and the new points accepted before it */
size_t vo_roll_filter_sparse(const vo_roll* r, const float* x, const float* y, const float* z,
                             size_t m, int min_count, unsigned char* accept);
long vo_roll_append_sparse(vo_roll* r, const float* x, const float* y, const float* z, size_t m,
                           int min_count);
/* box intersected with the cylinder of `radius` around (cx, cy) in the ground plane (radius < 0:
 * box only): eviction by ROI_RANGE (MapManager.h:13) */
int vo_roll_evict_region(vo_roll* r, const float lo[3], const float hi[3], float cx, float cy,
                         float radius);
size_t vo_map_size(const vo_map*);
void vo_map_grid(const vo_map*, float origin[3], int dims[3], float* inv_h);
/* sorted SoA arrays (length n), perm[s] = original index; cell_start = the FINE cell
 * table (vo_map_num_cells()+1 entries, vo_map_num_cells = product of subdiv*dims) */
const float* vo_map_x(const vo_map*);
const float* vo_map_y(const vo_map*);
const float* vo_map_z(const vo_map*);
const float* vo_map_nx(const vo_map*);
const float* vo_map_ny(const vo_map*);
const float* vo_map_nz(const vo_map*);
const int32_t* vo_map_perm(const vo_map*);
const int32_t* vo_map_cell_start(const vo_map*);
size_t vo_map_num_cells(const vo_map*);

/* a10: exhaustive 27-cell 1-NN.  corr[i] = sorted map index or -1.
 * d2[i] = best squared distance (float) or +inf.  Returns total number of
 * candidates scanned (sum over queries) -> Cbar = ret / n. */
uint64_t vo_correspond(const vo_map*, const float* x, const float* y, const float* z, size_t n,
                       const double T[12], float d_max, int32_t* corr, float* d2);

/* a10 with k neighbours (k <= 32): n x k row-major, ascending (d2, sorted index), padded
 * with -1 / +inf */
void vo_knn(const vo_map*, const float* x, const float* y, const float* z, size_t n,
            const double T[12], float d_max, int k, int32_t* idx, float* d2, int32_t* count);

/* a11: accumulate the 29 doubles: H upper triangle (21, row-major a<=b),
 * g (6), sum r^2, count. */
void vo_accumulate(const vo_map*, const float* x, const float* y, const float* z, size_t n,
                   const double T[12], const int32_t* corr, double acc[29]);

/* a12: solve H xi = -g (LDLt), T <- exp(xi^) T.  Returns 0 ok, 1 if the
 * guard (+1e-9 I) was needed, 2 if skipped (count < 6). */
int vo_solve_update(const double acc[29], double T[12], double xi[6]);

typedef struct vo_icp_stat {
    uint32_t n_pairs;
    double rmse;
    uint64_t candidates;
} vo_icp_stat;

/* a9: exactly `iters` Gauss-Newton steps, no early exit.  trace (optional):
 * iters x 12 doubles, pose AFTER each iteration.  threads<=1: serial. */
int vo_icp(const vo_map*, const float* x, const float* y, const float* z, size_t n,
           const double T0[12], int iters, float d_max, double T_out[12],
           vo_icp_stat* stats, double* trace, int threads);

/* accepted map increment: frame points, transformed by T, that land
 * in a map cell holding fewer than `min_count` points.  Returns count; order
 * preserving.  Outputs may be NULL to just count. */
size_t vo_increment(const vo_map*, const float* x, const float* y, const float* z, size_t n,
                    const double T[12], int min_count, float* ox, float* oy, float* oz);

#ifdef __cplusplus
}
#endif
#endif
