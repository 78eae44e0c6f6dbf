// velo_internal.hpp -- shared between the C-ABI glue (capi.cpp) and the HIP
// kernel translation units.  Not installed; the public surface is include/velo.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/velo.h"

namespace velo {

// ---------------------------------------------------------------- device views
// Voxel-sorted map as the kernels see it.  Points are float4 {x,y,z,0} so a
// candidate is one 16-byte load; normals likewise ({0,0,0,0} = invalid normal).
// a pose as a kernel ARGUMENT (96 bytes of kernarg, read by scalar loads): nothing to stage, nothing to wait for
struct Pose12 {
    double t[12];
};

struct MapView {
    const float4* pts;     // [n] sorted by FINE cell key (stable)
    const float4* nrm;     // [n]
    const int32_t* cell_start;  // [fx*fy*fz + 1] fine-cell table: number of keys < k (dense mode)
    // sparse mode (cell_start == nullptr): open-addressing hash over the OCCUPIED ROW PIECES (round 6; until then: over
    // the occupied fine cells, one probe per cell -- 27 per query of stage A, most of them unsuccessful).  A row piece
    // = the S fine cells of one voxel along one fine row: piece key = fine key / S = row * nx + voxel-x, and the points
    // of a piece are consecutive in the sorted order.  entry = {piece key, first sorted index, o1 | o2 << 16,
    // o3 | o4 << 16} (+ a second int4 {o5 | o6 << 16, o7 | o8 << 16, 0, 0} when S > 4: hash_stride = 2): o_c = points
    // of the piece in sub-cells < c (16 bits each: a piece holds fewer than 65 536 points, checked at build).
    // key 0xffffffff = empty slot.  slot = ((key * 0x9E3779B1) * capacity) >> 32, linear probing.  Same sorted order,
    // same answers.
    const int4* hash;
    uint32_t hash_cap;
    uint32_t hash_stride;  // int4 per slot: 1 (S <= 4) or 2
    uint64_t s_magic;      // ceil(2^64 / S) (S >= 2): v / S = (v * s_magic) >> 64 exactly for every 32-bit v
    const uint8_t* vox_near;    // [nx*ny*nz] 0 = no map point in the 27 voxels around (or nullptr)
    float ox, oy, oz, inv_h, h;
    int nx, ny, nz;        // voxels per axis
    int S;                 // sub-cells per voxel edge
    int fx, fy, fz;        // fine cells per axis = S * voxels
    int n;
};

// Frames resident on the device, concatenated SoA.
struct FrameView {
    const float* x;
    const float* y;
    const float* z;
    // When the frames were re-ordered by map cell (cfg.sort_frames) x/y/z are the permuted
    // copies and order[q] is the ORIGINAL index of slot q (used only to write the per-query
    // diagnostics back in caller order); nullptr = identity.
    const int32_t* order;
    // Round 6, latency kernels only (nullptr = off): PAIR CERTIFICATES.  hint2[q] = the runner-up of query q's last
    // search, rho3[q] = a radius around the query's previous position inside which the winner and the runner-up are
    // the ONLY map points -- a query whose two nearest candidates are nearly equidistant (consecutive returns of one
    // scan line in a map made of scans) has no uniqueness radius to speak of, and was searched at every iteration.
    int32_t* hint2;
    float* rho3;
};

// One work item of the linearise kernel: a run of consecutive queries of one frame.
struct BlockItem {
    int32_t frame;
    int32_t q0;  // first query (global index into the concatenated arrays)
    int32_t q1;  // one past last
    int32_t slot;  // bits 0..27: row of the partials buffer this block writes (frame-major, fixed);
                   // bits 28..30: log2 of the rounds per wavefront the item is cut for (its NOMINAL size is
                   // threads x rounds queries, an aligned block of the frame -- see linearize_body)
};
constexpr int kItemRowBits = 28;

constexpr int kAccN = 29;      // 21 + 6 + 1 + 1
constexpr int kAccStride = 32; // padded row of the partials buffer (doubles)
constexpr int kLinThreads = 256;
// threads per workgroup of the THROUGHPUT linearise kernel (the latency kernel and the first
// decomposition's work items stay at kLinThreads: a smaller workgroup takes more rounds per item)
#ifndef VELO_LIN_NT
#define VELO_LIN_NT 128
#endif
constexpr int kLinNT = VELO_LIN_NT;
// registrations of fewer queries than this (~4 frames) run on the latency kernel
constexpr int64_t kLatQueries = 2048 * 256;
// queries per wavefront of the latency kernel in the first (unhinted) iteration of a registration
#ifndef VELO_LAT_FIRST_LANES
#define VELO_LAT_FIRST_LANES 8
#endif
constexpr int kLatFirstLanes = VELO_LAT_FIRST_LANES;

// ---------------------------------------------------------------- launchers (kernels/*.hip)
struct MapBuild;  // opaque scratch owned by the ctx

hipError_t launch_compensate(const float* x, const float* y, const float* z, const uint16_t* pkt,
                             size_t n, const double* T3x4, size_t n_pkt, float* ox, float* oy,
                             float* oz, hipStream_t s);

struct MinMax {
    float mn[3], mx[3];
};
hipError_t launch_minmax(const float* x, const float* y, const float* z, size_t n,
                         unsigned* d_scratch6, MinMax* out_host, hipStream_t s);
hipError_t launch_keys(const float* x, const float* y, const float* z, size_t n, float ox, float oy,
                       float oz, float inv_h, int S, int fx, int fy, uint32_t* keys, uint32_t* idx,
                       hipStream_t s);
hipError_t sort_pairs(void* temp, size_t& temp_bytes, const uint32_t* k_in, uint32_t* k_out,
                      const uint32_t* v_in, uint32_t* v_out, size_t n, int end_bit, hipStream_t s);
hipError_t sort_pairs64(void* temp, size_t& temp_bytes, const uint64_t* k_in, uint64_t* k_out,
                        const uint32_t* v_in, uint32_t* v_out, size_t n, hipStream_t s);
hipError_t launch_gather(const float* x, const float* y, const float* z, const uint32_t* perm,
                         size_t n, float4* pts, hipStream_t s);
// (tile_scratch: cell_start_bounds(ncell) words, filled by the call)
size_t cell_start_bounds(size_t ncell);
hipError_t launch_cell_start(const uint32_t* sorted_keys, size_t n, size_t ncell,
                             int32_t* cell_start, uint32_t* tile_scratch, hipStream_t s);
// sparse table: number of occupied fine cells, then the hash itself (cap slots)
hipError_t launch_count_runs(const uint32_t* sorted_keys, size_t n, int S, unsigned long long* d_count, hipStream_t s);
// (runs of equal PIECE keys = fine key / S: the occupied row pieces of the sparse table)
hipError_t launch_hash_build(const uint32_t* sorted_keys, size_t n, int4* hash, uint32_t cap, int S, unsigned* d_overflow,
                             hipStream_t s);
// (row-piece hash: MapView.  *d_overflow != 0 afterwards: a piece holds 65 536 points or more -- its offsets do not fit)
hipError_t launch_normals(const MapView& mv, const uint32_t* perm, int k, float4* nrm,
                          unsigned long long* d_invalid, hipStream_t s, int mode = 0);
// (mode as launch_knn's: the full-map normals of a dense map go through the cooperative search)
hipError_t launch_normals_wave(const MapView& mv, const uint32_t* perm, int k, float4* nrm,
                               unsigned long long* d_invalid, hipStream_t s);
hipError_t launch_knn_wave(const MapView& mv, const float* x, const float* y, const float* z, size_t n,
                           const struct Pose12& T, float dmax2, int k, int32_t* idx, float* d2, int32_t* count,
                           hipStream_t s, unsigned long long* stats_out);

// incremental map update (f3): see kernels/map_build.hip
hipError_t launch_keys4(const float4* pts, size_t n, const MapView& grid, uint32_t* keys,
                        hipStream_t s);
hipError_t launch_merge(const float4* pts, const float4* nrm, const uint32_t* perm,
                        const uint32_t* keys, uint32_t n, const float* rx, const float* ry,
                        const float* rz, uint32_t raw_base, const uint32_t* nk,
                        const uint32_t* nidx, uint32_t m, float4* pts2, float4* nrm2,
                        uint32_t* perm2, uint32_t* keys2, hipStream_t s);
// (tile_scratch: table_tile_bounds(n_entries) words, filled by the call)
size_t table_tile_bounds(size_t n_entries);
hipError_t launch_table_shift(const int32_t* src, int32_t* dst, size_t n_entries, const uint32_t* nk, uint32_t m,
                              uint32_t* tile_scratch, hipStream_t s);
hipError_t launch_mark_dirty(const uint32_t* keys, uint32_t m, const uint32_t* sel,
                             const MapView& grid, uint8_t* dirty, hipStream_t s);
hipError_t launch_select_dirty(const uint32_t* keys, uint32_t n, const MapView& grid, const uint8_t* dirty,
                               const float4* nrm, const uint32_t* chg_keys, uint32_t n_chg, int32_t* work,
                               unsigned* count, hipStream_t s);
// (grid = the map AFTER the update: its pts / nrm are read for the reach test when chg_keys != nullptr)
hipError_t launch_normals_subset(const MapView& mv, const uint32_t* perm, int k,
                                 const int32_t* work, int n_work, const uint32_t* chg_keys,
                                 uint32_t n_chg, float4* nrm, unsigned long long* d_invalid,
                                 unsigned* d_done, hipStream_t s, const unsigned* n_work_dev = nullptr, int mode = 0);
// (n_work_dev != nullptr: n_work is an upper bound, the length of the list is read on the device;
//  mode = cfg.force_kernel: 1 = one lane per listed point, 2 = one wavefront per listed point, 0 = the default)
bool normals_subset_use_wave(const MapView& mv, int n_work, int mode);
hipError_t launch_normals_wave_subset(const MapView& mv, const uint32_t* perm, int k, const int32_t* work, int n_work,
                                      float4* nrm, unsigned long long* d_invalid, unsigned* d_done, hipStream_t s,
                                      const unsigned* n_work_dev);
hipError_t launch_removed_keys(const uint32_t* keys, const uint32_t* keep, const uint32_t* offs,
                               uint32_t n, uint32_t* out, hipStream_t s);
hipError_t launch_count_invalid(const float4* nrm, uint32_t n, unsigned long long* d_invalid,
                                hipStream_t s);
struct KeepRegion {  // closed box, optionally intersected with a vertical cylinder
    float lo[3], hi[3];
    float cx, cy, r2;
    int use_radius;
};
hipError_t launch_keep_flags(const float4* pts, const float* x, const float* y, const float* z,
                             uint32_t n, const KeepRegion& g, uint32_t* flags, hipStream_t s);
hipError_t launch_keep_flags_minmax(const float* x, const float* y, const float* z, uint32_t n,
                                    const KeepRegion& g, uint32_t* flags, unsigned* d_scratch6, hipStream_t s);
hipError_t launch_compact_sorted(const float4* pts, const float4* nrm, const uint32_t* perm,
                                 const uint32_t* keys, uint32_t n, const uint32_t* flags,
                                 const uint32_t* offs, const uint32_t* raw_offs, float4* pts2,
                                 float4* nrm2, uint32_t* perm2, uint32_t* keys2,
                                 unsigned long long* d_invalid, hipStream_t s);
hipError_t launch_compact_raw(const float* x, const float* y, const float* z, uint32_t n,
                              const uint32_t* flags, const uint32_t* offs, float* x2, float* y2,
                              float* z2, hipStream_t s);
hipError_t launch_table_remap(const int32_t* src, int32_t* dst, size_t n_entries, const uint32_t* offs, uint32_t n,
                              uint32_t kept, uint32_t* tile_scratch, hipStream_t s);

hipError_t launch_scatter_nrm_raw(const float4* nrm, const uint32_t* perm, uint32_t n,
                                  const uint32_t* keep, const uint32_t* raw_offs, float4* nrm_raw,
                                  hipStream_t s);
hipError_t launch_fill_fresh(float4* nrm_raw, uint32_t n, hipStream_t s);
hipError_t launch_gather_nrm(const float4* nrm_raw, const uint32_t* perm, uint32_t n, float4* nrm,
                             hipStream_t s);
hipError_t launch_mark_dirty_pts(const float4* pts, const float4* nrm, const uint32_t* keep,
                                 uint32_t n, const MapView& grid, uint8_t* dirty, hipStream_t s);

hipError_t launch_vox_near(const MapView& mv, const uint32_t* keys_sorted, uint8_t* occ, uint8_t* near,
                           hipStream_t s);
hipError_t launch_count_occupied_voxels(const float* x, const float* y, const float* z, size_t n,
                                        const float mn[3], float inv_h, const size_t dims[3],
                                        uint8_t* occ, unsigned long long* d_count, hipStream_t s);

hipError_t launch_knn(const MapView& mv, const float* x, const float* y, const float* z, size_t n,
                      const Pose12& T, float dmax2, int k, int32_t* idx, float* d2, int32_t* count,
                      hipStream_t s, unsigned long long* stats_out = nullptr, int mode = 0);
// (mode: 0 = by the map's density, 1 = one lane per query, 2 = one wavefront per query; same results)
bool knn_use_wave(const MapView& mv, int mode);
bool normals_use_wave(const MapView& mv, int mode);
// (stats_out != nullptr: the counting instantiation; waits for the stream; [0] queries, [1] candidate points
//  fetched, [2] fine rows looked up, [3] fine cells those rows span)
hipError_t launch_linearize(int variant, const BlockItem* items, int n_items, const FrameView& fv,
                            const MapView& mv, const double* poses, float dmax2, double* partials,
                            int32_t* corr, float* d2, int32_t* hint, float* rho,
                            const double* poses_prev, bool stats, int force_kernel, hipStream_t s, int lat_lanes = 64);
// the split iteration (kernels/icp.hip): phase A (certificate test + stage A of every query, stragglers to the queue
// sq, *sq_count of them) and phase B (the queue: one wavefront per straggler up to per_wave_max of them, 64 per
// wavefront beyond) of an iteration whose third launch is launch_linearize with poses_prev == poses
hipError_t launch_search_split(const BlockItem* items, int n_items, const FrameView& fv, const MapView& mv,
                               const double* poses, float dmax2, int32_t* hint, float* rho, const double* poses_prev,
                               int force_kernel, int lat_lanes, int4* sq, unsigned* sq_count, unsigned per_wave_max,
                               int grid_b, hipStream_t s);
hipError_t read_lin_stats(unsigned long long out[16], bool reset, hipStream_t s);
// How the rows of one frame tile its canonical summation tree (kernels/icp.hip, k_reduce_solve): `head` rows
// of the small size, `nbig` rows 2^mlog times as large, small rows to the end; nslots = the frame's length in
// small rows.  One per frame and decomposition, made by the planner.
struct RowLayout {
    int32_t head, nbig, mlog, nslots;
};
// k_reduce_solve joins a frame's slots in 32 groups of 16-slot trips through a carry stack of 8 levels
constexpr int32_t kMaxRowSlots = 32 * 32 * 256;
hipError_t launch_reduce_solve(const double* partials, const int32_t* frame_block_start, const RowLayout* layout,
                               int n_frames, double* poses, velo_icp_iter* stats, int iter,
                               int solve_threads /* 0 = default, 256 / 512 / 1024 */, double* acc_out, int do_update, double* poses_prev,
                               unsigned long long* pairs_total, hipStream_t s, int spec_rows = 0,
                               const RowLayout* layout0 = nullptr, bool mixed = true);
// (mixed = false: the caller vouches that every frame's layout has nbig == 0 -- rows of one size)
// (spec_rows > 0 with layout0 = HOST copy of frame 0's layout: frame_block_start[0] == 0 and `partials`
//  holds at least spec_rows rows -- frame 0 then starts its loads without fetching anything first)
hipError_t launch_frame_cellkeys(const FrameView& fv, const int64_t* d_frame_start, int n_frames,
                                 size_t n_total, const MapView& mv, const double* poses,
                                 uint32_t* keys, uint32_t* idx, hipStream_t s);
hipError_t launch_permute3(const float* x, const float* y, const float* z, const uint32_t* order,
                           size_t n, float* ox, float* oy, float* oz, hipStream_t s);
hipError_t launch_increment_flags(const float* x, const float* y, const float* z, size_t n,
                                  const MapView& mv, const double* pose, int min_count,
                                  uint32_t* flags, hipStream_t s);
constexpr uint32_t kIncTilePoints = 256;      // launch_increment_fused: points per workgroup ...
constexpr uint32_t kIncFusedMaxTiles = 4096;  // ... and above this many workgroups (1 M points) the scan path
hipError_t launch_increment_fused(const float* x, const float* y, const float* z, uint32_t n, const MapView& mv,
                                  const double* pose, int min_count, uint32_t* flags, uint32_t* d_block_cnt,
                                  float* ox, float* oy, float* oz, uint32_t* d_total, hipStream_t s);
hipError_t exclusive_scan_u32(void* temp, size_t& temp_bytes, const uint32_t* in, uint32_t* out,
                              size_t n, hipStream_t s);
hipError_t launch_increment_scatter(const float* x, const float* y, const float* z, size_t n,
                                    const double* pose, const uint32_t* flags,
                                    const uint32_t* offs, float* ox, float* oy, float* oz,
                                    hipStream_t s);

hipError_t launch_increment_flags_items(const BlockItem* items, int n_items, const FrameView& fv,
                                        const MapView& mv, const double* poses, int min_count,
                                        uint32_t* flags, hipStream_t s);
hipError_t launch_increment_scatter_items(const BlockItem* items, int n_items, const FrameView& fv,
                                          const double* poses, const uint32_t* flags,
                                          const uint32_t* offs, float* ox, float* oy, float* oz,
                                          hipStream_t s);

// voxel-downsampled insertion (oracle/icp.c vo_roll_filter_sparse)
hipError_t launch_sparse_keys(const float* x, const float* y, const float* z, size_t n, const MapView& mv,
                              uint64_t* keys, uint32_t* idx, hipStream_t s);
hipError_t launch_sparse_accept(const uint64_t* keys_sorted, const uint32_t* idx_sorted, size_t n,
                                const MapView& mv, int min_count, uint32_t* accept, hipStream_t s);
hipError_t launch_compact3(const float* x, const float* y, const float* z, size_t n, const uint32_t* flags,
                           const uint32_t* offs, float* ox, float* oy, float* oz, hipStream_t s);

// ---- multi-GPU exchange step: pack kernels either side of the all-gather (kernels/exchange.hip)
struct RankOffsets {
    uint32_t off[VELO_MAX_RANKS + 1];  // off[r] = first output index of rank r's block, off[world] = total
    int32_t world;
};
hipError_t launch_pack_send(const float* x, const float* y, const float* z, uint32_t n, uint32_t pad,
                            float* send, hipStream_t s);
hipError_t launch_pack_rank_blocks(const float* recv, const RankOffsets& ro, uint32_t pad, float* ox,
                                   float* oy, float* oz, hipStream_t s);

// ---- f1: packet decode (kernels/decode.hip)
struct DecodeView {
    const uint8_t* pkts;        // n_pkt * 1206
    const int16_t* blk_frame;   // n_pkt * 12: frame of the firing block, -1 = not processed
    const uint8_t* frame_perm;  // per frame: 1 = apply the HDL-64 beam LUT
    const double* table;        // n_pkt * 12 affine
    const uint8_t* tvalid;      // n_pkt: 1 = transform present
    const int32_t* az_diff;     // n_pkt
    const double* corr;         // 64 * 9
    const double* lut_cos;      // 36001
    const double* lut_sin;
    const double* az_cos;       // 64 * 36000 (rows of lasers with azimuthCorrection != 0)
    const double* az_sin;
    const uint8_t* inv_lut;     // 64
    unsigned long long laser_mask;  // bit i: laser id i is selected (HDLParser.cxx:964)
    int n_pkt, n_lasers;
    int crop, crop_inside;
    double region[6];
};
hipError_t launch_decode_keys(const DecodeView& v, size_t n_ret, uint32_t* keys, uint32_t* idx,
                              hipStream_t s);
hipError_t launch_decode_emit(const DecodeView& v, const uint32_t* order, size_t n_valid, float* ox,
                              float* oy, float* oz, float* oi, uint16_t* oaz, float* odist,
                              uint16_t* opkt, hipStream_t s);
hipError_t launch_key_starts(const uint32_t* keys, size_t n, uint32_t n_keys, int32_t* starts,
                             hipStream_t s);

}  // namespace velo
