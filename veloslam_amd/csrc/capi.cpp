// capi.cpp -- C-ABI glue of libveloslam_amd.so (include/velo.h): device memory,
// stream-ordered launches of the gfx950 kernels, result marshalling.  No compute
// happens on the host here and there is no CPU fallback: a missing GPU or a HIP
// error surfaces as a negative return code plus velo_last_error().
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstring>
#include <memory>
#include <dlfcn.h>
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif
#include "velo_internal.hpp"
#include "host/decode_plan.hpp"
#include "../../include/veloslam/TransformManager.hpp"

// The handful of RCCL declarations the exchange needs, stated here so that the library builds
// without the RCCL headers too (values as in rccl.h of ROCm 7.x == nccl.h: the C ABI of the
// collectives library is stable across releases).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[VELO_COMM_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt32 = 2, ncclFloat32 = 7 } ncclDataType_t;
}

// launch order of the work items (launch_order below): measured on 64 frames, first iteration
// 486 us frame-major, 465 item-major, 448 item-major backwards (the ends of the frames hold the
// far, sparse returns -- the expensive searches -- and now start first), 439 with the item
// indices dealt over the 8 XCDs on top (each L2 then sees an eighth of the map regions at a
// time); hinted iterations gain in one place what they lose in another and stay frame-major
#ifndef VELO_ORDER_FIRST
#define VELO_ORDER_FIRST 4
#endif
#ifndef VELO_ORDER_LATE
#define VELO_ORDER_LATE 0
#endif
// (order 5, eighth x of every frame on XCD x, takes the converged launch from 69 to 67 us but
// costs as much in iterations 5-6, which still search a little; in the searching iterations it
// piles the expensive ends of the frames on one XCD: iteration 1 260 -> 388 us.  Not used.)
#ifndef VELO_ORDER_CONV
#define VELO_ORDER_CONV 0
#endif
#ifndef VELO_LATE_ROUNDS
#define VELO_LATE_ROUNDS 4  // (a power of two <= 4 since round 4: items are aligned nodes of the frame's summation tree)
#endif
#ifndef VELO_LAT_SPARSE_MIN_S
#define VELO_LAT_SPARSE_MIN_S 4
#endif
#ifndef VELO_FIRST_HEAD_PCT
#define VELO_FIRST_HEAD_PCT 10
#endif
#ifndef VELO_FIRST_ROUNDS
#define VELO_FIRST_ROUNDS 4
#endif
#ifndef VELO_MIN_SLOT_ROUNDS
#define VELO_MIN_SLOT_ROUNDS 2.5
#endif
#ifndef VELO_CONV_ROUNDS
#define VELO_CONV_ROUNDS 4  // (was 6: converged launch 68 us against ~70; see VELO_LATE_ROUNDS)
#endif
#ifndef VELO_CONV_FROM
#define VELO_CONV_FROM 5
#endif
#ifndef VELO_CONV_TAIL_PCT
#define VELO_CONV_TAIL_PCT 0  // (round 4: rows of one size let k_reduce_solve take its short path in 15 of 20 iterations: 2.17 -> 2.15 ms per step; the tail still pays in iterations 1-4: 0 there costs 2.5 %)
#endif
#ifndef VELO_LATE_TAIL_PCT
#define VELO_LATE_TAIL_PCT 10
#endif

using namespace velo;

namespace {

std::string g_create_error;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    bool owned = true;
    ~DevBuf() { release(); }
    void release()
    {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        owned = true;
    }
    // a view into somebody else's allocation (PlanUpload: the slices of one staging buffer)
    void alias(T* q)
    {
        release();
        p = q;
        owned = false;
    }
    hipError_t reserve(size_t n, bool keep = false, hipStream_t s = nullptr)
    {
        if (n <= cap) return hipSuccess;
        size_t want = keep ? std::max(n, cap + cap / 2) : n;
        T* q = nullptr;
        hipError_t e = hipMalloc((void**)&q, want * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep && p && cap) {
            e = hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                (void)hipFree(q);
                return e;
            }
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return hipSuccess;
    }
};

}  // namespace

struct velo_ctx {
    int device = 0;
    velo_cfg cfg{};
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;

    // ---- map
    DevBuf<float> raw_x, raw_y, raw_z;  // append order
    size_t raw_n = 0;
    DevBuf<uint32_t> keys, keys_sorted, idx, perm;
    DevBuf<float4> pts, nrm;
    DevBuf<int32_t> cell_start;
    DevBuf<int4> hash_tab;     // sparse fine-cell table (cfg.map_hash_load / extents beyond 2^31 cells)
    DevBuf<int4> hash_tab_alt; // ... its other copy: an update beside a registration builds the table the registration does not read
    DevBuf<unsigned long long> run_cnt;  // scratch: occupied fine cells
    bool use_hash = false;
    DevBuf<unsigned> mm_scratch;
    DevBuf<unsigned long long> invalid_cnt;
    // incremental update (f3): second set of sorted arrays (swapped in), new-point keys,
    // dirty voxels and the normal work list
    DevBuf<float> raw_x2, raw_y2, raw_z2;
    DevBuf<uint32_t> keys_alt, perm_alt, nk, nk_sorted, nidx, nidx_sorted, rflags, roffs;
    DevBuf<float4> pts_alt, nrm_alt, nrm_raw;
    DevBuf<uint8_t> dirty, vox_occ, vox_near;
    // second copies of what a registration reads and a rolling update would rewrite in place: with
    // them an update writes the copy the running registration does not read (overlap_update)
    DevBuf<int32_t> cell_start_alt;
    DevBuf<uint32_t> tile_bounds;   // per 1 024-entry tile of the table: launch_table_shift / _remap
    DevBuf<int4> sq;                // the split iteration's straggler queue (one slot per query) ...
    DevBuf<unsigned> sq_count;      // ... and its length, one counter per iteration of a registration
    bool pair_certs = true;         // latency kernels: pair + no-match certificates (cfg.pair_certificates)
    int split_iters = 1;            // iterations 0 .. split_iters - 1 run as three launches (cfg.split_iterations)
    bool split_batch = false;       // ... on the throughput path too (VELO_SPLIT_BATCH: measurements and tests)
    unsigned split_per_wave_max = 131072;  // stragglers up to which phase B gives each a wavefront of its own (a latency-path launch: two or three frames at most)
    DevBuf<uint8_t> vox_near_alt;
    bool overlap_update = false;        // inside velo_map_roll_overlapped
    DevBuf<int32_t> knn_idx, knn_cnt;   // velo_knn: device-side results before the copy back
    DevBuf<float> knn_d2;
    int overlap_done = 0;               // ... updates published so far in this call
    // a roll is TWO updates (evict, append): the second must not write the arrays the registration is
    // still reading either -- which after the first swap are the "alt" ones -- so a third set steps in
    DevBuf<float4> pts_3, nrm_3;
    DevBuf<uint32_t> perm_3, keys_3;
    hipStream_t overlap_main = nullptr;  // ... the main stream meanwhile
    DevBuf<int32_t> work;
    DevBuf<unsigned> work_cnt;  // [0] work-list length, [1] normals re-estimated
    unsigned n_done_host = 0;
    float map_mx[3] = {0, 0, 0};  // component-wise max of the map points
    int margin[3] = {0, 0, 0};    // grid slack per axis, voxels (cfg.map_margin / velo_map_set_margins)
    int map_S = 3;                // sub-division of the current map: cfg.map_subdiv, or chosen from
                                  // the density when that is 0; resolved at velo_map_reset
    DevBuf<char> temp;
    MapView mv{};          // the map as of the last update (what the next update starts from)
    MapView mv_read{};     // the map registrations / increments / k-NN READ: == mv, except between
                           // velo_map_roll_begin and velo_map_roll_publish, when it is still the map before the roll
    // ---- a roll begun ahead of the frame that needs it (velo_map_roll_begin .. velo_map_roll_publish)
    hipStream_t roll_stream = nullptr;  // a stream of its own: a decode on the side stream must not queue behind 2 ms of roll
    // Round 6: ... and a second one WITHOUT the CU mask for LIGHT rolls (a mapping stream's per-frame update: a few thousand
    // points into a map of a few hundred thousand).  The mask keeps a registration's 1 024-thread solve from waiting behind
    // a roll that fills the chip for milliseconds (tile columns of 500 k points into 11 M); a light roll never fills it, and
    // on the masked queue it ran -- and made the registration beside it run -- 20 % slower (profiles/r06/ab_roll_cus.txt).
    hipStream_t roll_stream_light = nullptr;
    hipStream_t roll_last = nullptr;    // the stream the last roll ran on (the next one waits for it: they share buffers)
    bool ev_roll_recorded = false;
    hipEvent_t ev_roll = nullptr;       // the roll's last kernel
    // Round 6: the rolled map's GEOMETRY (points, fine table) is complete long before its normals are (the re-estimation of
    // the dirty neighbourhoods is half of an append).  An increment reads geometry only: the publish makes the main stream
    // wait for this event, and whatever reads normals / vox_near next (a registration) waits for ev_roll then.
    hipEvent_t ev_roll_geom = nullptr;
    bool roll_geom_recorded = false;    // the LAST update of the begun roll recorded it (an incremental update with normals)
    bool normals_wait_owed = false;     // published on the geometry event: the main stream has not yet waited for ev_roll
    bool roll_staged = false;           // begun, not yet published
    bool defer_counts = false;          // inside velo_map_roll_begin: nothing waits for the device after the first count
    struct RollResults {                // pinned: what the device reports when the roll is through
        unsigned long long invalid;
        unsigned n_done;
    }* h_roll = nullptr;
    bool roll_counts_pending = false;   // info.n_invalid_normals / n_normals_recomputed still to be read from h_roll
    float* h_enter = nullptr;           // pinned staging of the entering points (x | y | z)
    size_t h_enter_cap = 0;
    bool have_enter_mm = false;         // bounds of the entering points, known on the host (no device min/max, no wait)
    float enter_mn[3] = {0, 0, 0}, enter_mx[3] = {0, 0, 0};
    size_t roll_extra = 0;              // points the roll's append will add: the eviction sizes the raw arrays for both
    DevBuf<char> roll_temp;             // sort / scan scratch of a roll on roll_stream
    bool has_map = false;
    velo_map_info info{};

    // ---- frames
    DevBuf<float> fx, fy, fz;
    const float *ax = nullptr, *ay = nullptr, *az = nullptr;  // active (owned or adopted)
    int n_frames = 0;
    std::vector<int64_t> frame_start;
    // Every small array a frame plan sends to the device (work items in their launch orders, block
    // ranges, frame offsets) goes through ONE pinned staging buffer and ONE copy; the DevBufs below
    // that hold them are views into `d` (six copies from pageable vectors per frame of a stream,
    // each staged and launched on its own, were ~50 us of every frame).
    struct PlanUpload {
        uint8_t* h = nullptr;  // pinned
        size_t h_cap = 0, used = 0;
        DevBuf<uint8_t> d[2];  // alternating: the plan of the frames in flight stays intact while the next goes up
        int sel = 0;
        struct Slice {
            void (*bind)(void* buf, uint8_t* base, size_t off);
            void* buf;
            size_t off;
        };
        std::vector<Slice> slices;
    } plan_up;
    DevBuf<int64_t> d_frame_start;
    std::vector<BlockItem> items_h;   // frame-major
    std::vector<int32_t> fbs_h;
    // host copies of the other work-item lists: kept alive here so that their uploads need no
    // synchronisation (plan_frames runs once per FRAME in a stream; ev_plan = last upload)
    std::vector<BlockItem> plan_late_h, plan_lo_h, plan_xcd_h;
    std::vector<int32_t> plan_fbl_h;
    hipEvent_t ev_plan = nullptr;
    hipEvent_t ev_res = nullptr;      // velo_icp_batch_start .. _finish
    int res_frames = 0, res_iters = 0;
    bool res_pending = false;
    DevBuf<BlockItem> items;          // frame-major (single-frame entry points index into it)
    DevBuf<BlockItem> items_first;    // the same items in launch order (launch_order)
    DevBuf<BlockItem> items_xcd;      // same blocks, dealt so that XCD r works on spatial slab r
    // second decomposition for the hinted iterations of a batch: three rounds of 256 queries per
    // workgroup (plan_frames); ni_late == 0 = not in use
    DevBuf<BlockItem> items_late;
    DevBuf<int32_t> fbs_late;
    DevBuf<RowLayout> lay, lay_late, lay_conv;   // per frame: how its rows tile the summation tree (k_reduce_solve)
    RowLayout lay0_h{}, lay0_late_h{}, lay0_conv_h{};  // frame 0's, by value with the launch
    std::vector<RowLayout> plan_lay_h, plan_lay2_h;
    int ni_late = 0;
    // third decomposition, for the iterations from VELO_CONV_FROM on (hardly any query searches
    // any more: coarser items)
    bool plan_lat = false;            // the resident frames are cut for the latency kernel
    int lat_first_lanes = 64;         // ... with this many queries per wavefront in the first decomposition
    int wave_slots = 256 * 28;        // wavefronts the device holds at 7 per SIMD (velo_create)
    DevBuf<BlockItem> items_conv;
    DevBuf<int32_t> fbs_conv;
    int ni_conv = 0;
    DevBuf<float> sx, sy, sz;         // cell-sorted copies of the frames (cfg.sort_frames)
    DevBuf<int32_t> fbs;
    DevBuf<double> poses, partials, acc;
    DevBuf<velo_icp_iter> stats;
    DevBuf<uint32_t> order_keys, order_keys2, order_idx, order;
    DevBuf<int32_t> corr;
    DevBuf<int32_t> hint;  // last correspondence per query slot (search-radius hint), -1 = none
    DevBuf<float> rho;     // certified uniqueness radius per query slot (valid with hint >= 0); < 0 with hint < 0: no candidate within -rho
    DevBuf<int32_t> hint2; // latency path (round 6): runner-up of the last search per query slot ...
    DevBuf<float> rho3;    // ... and the radius inside which winner and runner-up are the only map points (FrameView)
    DevBuf<double> poses_prev;  // pose each frame was linearised at in the previous iteration
    DevBuf<unsigned long long> pairs_total;  // pairs processed by every registration iteration so far
    DevBuf<float> d2;
    DevBuf<uint32_t> flags, offs;
    DevBuf<uint32_t> inc_flags, inc_offs;  // the increment's own: a map update on the side stream uses flags / offs
    DevBuf<float> inc_x, inc_y, inc_z;
    DevBuf<uint64_t> sp_keys, sp_keys2;   // sparse insertion: voxel keys of the new points
    DevBuf<uint32_t> sp_idx, sp_idx2;
    DevBuf<float> sp_x, sp_y, sp_z;
    DevBuf<double> inc_pose;
    // device-side list of accepted increments not yet merged into the map (velo_increment_pending)
    DevBuf<float> pend_x, pend_y, pend_z;
    size_t pend_n = 0;                 // points in the list whose count has reached the host
    uint32_t* h_pend_total = nullptr;  // pinned: count of the increment in flight
    hipEvent_t ev_pend = nullptr;
    bool pend_outstanding = false;
    hipStream_t copy_stream = nullptr;  // velo_pending_fetch: the list is complete once ev_pend has fired, its copy must
    float* h_pend_stage = nullptr;      // not queue behind the registration the main stream is busy with (pinned staging)
    size_t h_pend_stage_cap = 0;
    uint8_t* h_result = nullptr;      // pinned: poses + per-iteration statistics of a fetch
    int32_t* h_starts = nullptr;      // pinned: beam offsets of a decode
    size_t h_starts_cap = 0;
    uint32_t* h_inc_total = nullptr;  // pinned: count of the asynchronous increment
    DevBuf<uint8_t> dk_stage;         // ... and their one device-side copy
    hipEvent_t ev_inc = nullptr;
    bool inc_pending = false;
    int last_iters = 0;
    // hipGraph replay of the per-registration launch sequence (cfg.use_graph)
    hipGraphExec_t graph_exec = nullptr;
    struct GraphKey {
        int iters = 0, ni = 0, n_frames = 0, variant = 0;
        float dmax2 = 0;
        const void *hint = nullptr, *rho = nullptr, *items = nullptr, *stream = nullptr, *sq = nullptr, *hint2 = nullptr;
        uint64_t map_gen = 0, frames_gen = 0;
        int n_split = 0;
        bool operator==(const GraphKey& o) const
        {
            return iters == o.iters && ni == o.ni && n_frames == o.n_frames && variant == o.variant &&
                   dmax2 == o.dmax2 && hint == o.hint && rho == o.rho && items == o.items && stream == o.stream &&
                   map_gen == o.map_gen && frames_gen == o.frames_gen && n_split == o.n_split && sq == o.sq && hint2 == o.hint2;
        }
    } graph_key;
    uint64_t map_gen = 0, frames_gen = 0;
    uint64_t seen_map_gen = ~0ull, seen_frames_gen = ~0ull;  // of the previous registration
    // pinned staging of the initial poses, double-buffered: buffer b is reused only after the
    // upload that read it has completed (its event), so an _async call never waits for the
    // batch in flight
    double* h_T0[2] = {nullptr, nullptr};
    hipEvent_t ev_T0[2] = {nullptr, nullptr};
    int t0_next = 0;
    bool lin_hints = false;  // velo_linearize keeps/uses hints across calls (tests)
    bool stats_on = false;   // launch the counting instantiation of the linearise kernel

    // ---- multi-GPU exchange: RCCL communicator + its own stream
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_comm = nullptr, ev_comm_in = nullptr;
    DevBuf<int32_t> comm_counts;   // [world]
    DevBuf<float> comm_send, comm_recv;
    int32_t* h_comm_counts = nullptr;  // pinned [world]

    // ---- f1 decode
    DevBuf<uint8_t> dk_invlut;
    DevBuf<double> dk_corr, dk_lutc, dk_luts, dk_azc, dk_azs;
    DevBuf<int32_t> dk_starts;
    DevBuf<uint32_t> dk_keys, dk_keys2, dk_idx, dk_order;
    // decode outputs, two sets: a decode writes the set the PREVIOUS decode did not, so the frames of
    // the previous decode stay intact (and may be being registered) while the next ones are produced
    struct DecodeOut {
        DevBuf<float> x, y, z, i, dist;
        DevBuf<uint16_t> az, pidx;
    } dk_out[2];
    int dk_sel = 0;  // the set the last decode wrote
    DevBuf<char> dk_temp;                // sort scratch of a decode on the side stream
    hipStream_t side_stream = nullptr;   // velo_decode_submit_overlapped
    hipEvent_t ev_mark = nullptr, ev_side = nullptr;
    bool mark_valid = false;
    bool roll_overlapped_done = false;  // one velo_map_roll_overlapped per registration
    std::vector<double> dk_corr_host;      // calibration the device tables were built for
    int dk_frames = 0;
    size_t dk_points = 0;
    std::vector<int64_t> dk_frame_start, dk_frame_t;
    std::vector<int32_t> dk_beam_start, dk_frame_packets;
    std::vector<velo_pose> dk_carposes;

    velo_decode_opts dopts{};  // laser selection, points skip, initial firing skip (sticky)
    // parser state carried across velo_decode_stream calls, and what the host half of a decode leaves
    // for the device half (host/decode_plan.hpp: plain host code, no GPU types)
    using DecodeStream = velo::DecodeStream;
    using DecodePlan = velo::DecodePlan;
    DecodeStream dstream;
    DecodePlan dplan;

    // ---- timing
    bool timing = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    std::vector<int> ev_kind;  // per pair: 0 linearise, 1 solve
    hipEvent_t ev_call0 = nullptr, ev_call1 = nullptr;
    double last_timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<float> last_lin_us;  // every linearise launch of the last timed registration

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return (ctx)->fail(VELO_E_DEVICE, "%s failed: %s (%s:%d)", #expr,                  \
                               hipGetErrorString(e__), __FILE__, __LINE__);                    \
    } while (0)

namespace {

// A map update rewrites the live sorted arrays / the fine-cell table in place.  If it fails after
// it has started doing so (a HIP error, out of memory), the ctx must not keep pointing at a
// half-updated map: the guard turns such a failure into "no map" (VELO_E_NOMAP on the next
// registration, velo_map_reset recovers) -- never a silently inconsistent one.
struct MapTxn {
    velo_ctx* c;
    bool touched = false, ok = false;
    explicit MapTxn(velo_ctx* ctx) : c(ctx) {}
    ~MapTxn()
    {
        if (touched && !ok) c->has_map = false;
    }
    int done(int rc)
    {
        ok = rc == VELO_OK;
        return rc;
    }
};

int ensure_temp(velo_ctx* c, size_t bytes)
{
    HIP_TRY(c, c->temp.reserve(bytes));
    return VELO_OK;
}

template <typename T>
hipError_t reserve_slack(DevBuf<T>& b, size_t n)
{
    if (n <= b.cap) return hipSuccess;
    return b.reserve(n + n / 8 + 4096);  // a rolling map grows a little every frame
}

// overlapped update, second and later of a call: what "alt" points at is what the running
// registration reads (the first update swapped it there); write a third set instead
void overlap_rotate_alt(velo_ctx* c)
{
    if (!c->overlap_update || c->overlap_done == 0) return;
    std::swap(c->pts_alt.p, c->pts_3.p);
    std::swap(c->pts_alt.cap, c->pts_3.cap);
    std::swap(c->nrm_alt.p, c->nrm_3.p);
    std::swap(c->nrm_alt.cap, c->nrm_3.cap);
    std::swap(c->perm_alt.p, c->perm_3.p);
    std::swap(c->perm_alt.cap, c->perm_3.cap);
    std::swap(c->keys_alt.p, c->keys_3.p);
    std::swap(c->keys_alt.cap, c->keys_3.cap);
}

// normals of the points in dirty voxels, after the sorted arrays were updated in place
// chg_keys: sorted fine keys of the added / removed points (nullptr: re-estimate every point of
// the dirty voxels).  The number of normals really re-estimated lands in c->n_done_host once
// the stream has been synchronised.
int refresh_dirty_normals(velo_ctx* c, const MapView& mv, int k, const uint32_t* chg_keys,
                          uint32_t n_chg)
{
    hipStream_t s = c->stream;
    HIP_TRY(c, reserve_slack(c->work, (size_t)mv.n));
    HIP_TRY(c, c->work_cnt.reserve(2));
    HIP_TRY(c, hipMemsetAsync(c->work_cnt.p, 0, 2 * sizeof(unsigned), s));
    // (the work list holds only the points a changed point can reach: the test is part of the selection -- the
    //  normals kernel below is given no changed keys and re-estimates everything it is handed)
    HIP_TRY(c, launch_select_dirty(c->keys_sorted.p, (uint32_t)mv.n, mv, c->dirty.p, c->nrm.p, chg_keys, n_chg,
                                   c->work.p, c->work_cnt.p, s));
    chg_keys = nullptr;
    n_chg = 0;
    if (c->defer_counts) {
        // a roll begun ahead: the length of the work list stays on the device -- the launch covers its upper
        // bound (every point), the surplus workgroups leave at once -- and the count of normals re-estimated
        // goes to pinned memory for whoever asks after the roll (velo_map_info_get)
        HIP_TRY(c, launch_normals_subset(mv, c->perm.p, k, c->work.p, mv.n, chg_keys, n_chg, c->nrm.p,
                                         c->invalid_cnt.p, c->work_cnt.p + 1, s, c->work_cnt.p, c->cfg.force_kernel));
        HIP_TRY(c, hipMemcpyAsync(&c->h_roll->n_done, c->work_cnt.p + 1, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        return VELO_OK;
    }
    unsigned n_work = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_work, c->work_cnt.p, sizeof n_work, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, launch_normals_subset(mv, c->perm.p, k, c->work.p, (int)n_work, chg_keys, n_chg,
                                     c->nrm.p, c->invalid_cnt.p, c->work_cnt.p + 1, s, nullptr, c->cfg.force_kernel));
    HIP_TRY(c, hipMemcpyAsync(&c->n_done_host, c->work_cnt.p + 1, sizeof(unsigned),
                              hipMemcpyDeviceToHost, s));
    return VELO_OK;
}

// make `mv` the ctx's map: bookkeeping shared by the full build and the incremental updates
// A roll begun ahead and not yet published is published now: every entry point that changes or reads the map
// as a whole starts here.  (The registrations do not: they keep reading the map before the roll -- mv_read.)
static int resolve_roll_counts(velo_ctx* c);
static int settle_normals_fwd(velo_ctx* c);
static int settle_roll(velo_ctx* c)
{
    // (the counts too: a plain update that follows publishes its own, and must not be overwritten by a late read)
    if (int rc = resolve_roll_counts(c)) return rc;
    if (c->roll_staged)
        if (int rc = velo_map_roll_publish(c)) return rc;
    return settle_normals_fwd(c);
}
// info.n_invalid_normals / n_normals_recomputed of a roll begun ahead: known when its last kernel is through
static int resolve_roll_counts(velo_ctx* c)
{
    if (!c->roll_counts_pending) return VELO_OK;
    HIP_TRY(c, hipEventSynchronize(c->ev_roll));
    c->info.n_invalid_normals = c->h_roll->invalid;
    c->info.n_normals_recomputed = c->h_roll->n_done;
    c->n_done_host = c->h_roll->n_done;
    c->roll_counts_pending = false;
    return VELO_OK;
}

// hints / certificates are indices and radii in the OLD map: forget them (overlapped update: the
// running registration is using them -- the reset queues behind it on the main stream)
static int reset_hints(velo_ctx* c, hipStream_t hs)
{
    // A registration never reads what an earlier one left (its first iteration passes poses_prev = nullptr: every stored
    // hint and certificate is stale by definition and is overwritten).  Only velo_linearize in hinted mode carries state
    // from call to call: the four fills per map update are its alone (a stream publishes a map update every frame).
    if (!c->lin_hints) return VELO_OK;
    if (c->hint.p && c->hint.cap)
        HIP_TRY(c, hipMemsetAsync(c->hint.p, 0xFF, c->hint.cap * sizeof(int32_t), hs));
    if (c->rho.p && c->rho.cap)  // 0 = no certificate (negative values certify "no match")
        HIP_TRY(c, hipMemsetAsync(c->rho.p, 0, c->rho.cap * sizeof(float), hs));
    if (c->hint2.p && c->hint2.cap) HIP_TRY(c, hipMemsetAsync(c->hint2.p, 0xFF, c->hint2.cap * sizeof(int32_t), hs));
    if (c->rho3.p && c->rho3.cap) HIP_TRY(c, hipMemsetAsync(c->rho3.p, 0, c->rho3.cap * sizeof(float), hs));
    return VELO_OK;
}

static int settle_normals(velo_ctx* c);
static int settle_normals_fwd(velo_ctx* c) { return settle_normals(c); }
// main-stream readers of normals / vox_near (registrations, linearise, map updates, downloads) after a publish on the
// geometry event: the rest of the roll
static int settle_normals(velo_ctx* c)
{
    if (!c->normals_wait_owed) return VELO_OK;
    c->normals_wait_owed = false;
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_roll, 0));
    return VELO_OK;
}
// inside an update of a roll begun ahead, after the sorted arrays and the table are written and before the normals are
static int mark_roll_geometry(velo_ctx* c)
{
    if (!c->defer_counts) return VELO_OK;
    if (!c->ev_roll_geom) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_roll_geom, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_roll_geom, c->stream));   // (c->stream is the roll's stream here: roll_run)
    c->roll_geom_recorded = true;
    return VELO_OK;
}

int publish_map(velo_ctx* c, MapView mv, int k_normals, unsigned long long invalid,
                int last_update, uint64_t n_recomputed)
{
    hipStream_t s = c->stream;
    const size_t nvox = (size_t)mv.nx * mv.ny * mv.nz;
    HIP_TRY(c, reserve_slack(c->vox_occ, nvox));
    if (c->overlap_update && c->overlap_done == 0) {  // the running registration reads vox_near: build the other copy
        std::swap(c->vox_near.p, c->vox_near_alt.p);
        std::swap(c->vox_near.cap, c->vox_near_alt.cap);
    }
    if (c->overlap_update) ++c->overlap_done;
    HIP_TRY(c, reserve_slack(c->vox_near, nvox));
    HIP_TRY(c, launch_vox_near(mv, c->keys_sorted.p, c->vox_occ.p, c->vox_near.p, s));
    mv.vox_near = c->vox_near.p;
    c->mv = mv;
    c->has_map = true;
    ++c->map_gen;
    if (!c->defer_counts) {   // (a roll begun ahead: readers keep the old map until velo_map_roll_publish)
        c->mv_read = mv;
        if (int rc = reset_hints(c, c->overlap_update ? c->overlap_main : s)) return rc;
    }
    c->info.n_points = (uint64_t)mv.n;
    c->info.n_cells = (uint64_t)mv.fx * mv.fy * mv.fz;
    c->info.origin[0] = mv.ox;
    c->info.origin[1] = mv.oy;
    c->info.origin[2] = mv.oz;
    c->info.voxel = mv.h;
    c->info.inv_voxel = mv.inv_h;
    c->info.dims[0] = mv.nx;
    c->info.dims[1] = mv.ny;
    c->info.dims[2] = mv.nz;
    c->info.k_normals = k_normals;
    c->info.subdiv = mv.S;
    c->info.table_kind = mv.cell_start ? 0 : 1;
    if (mv.cell_start) {
        c->info.table_slots = (uint64_t)mv.fx * mv.fy * mv.fz + 1;
        c->info.table_occupied = 0;
    }
    c->info.n_invalid_normals = invalid;
    c->info.last_update = last_update;
    c->info.n_normals_recomputed = n_recomputed;
    return VELO_OK;
}

// (re)build the voxel grid over raw_x/y/z[0..raw_n)
// grid_org/grid_dims == nullptr: anchor the grid on the points (origin = min - margin*h);
// otherwise build on that explicit grid (which must contain every point)
//
// carry: c->nrm_raw holds the normals of the raw list in append order (w == 1 marks points
// without one).  A normal depends on the point list only, so they are permuted into the new
// order and only the neighbourhoods of fresh points -- and of the removed points
// old_pts[i] with removed_keep[i] == 0 -- are re-estimated.
// Fine-cell table over the sorted keys: the dense prefix table, or -- when it would pass 2^31
// entries, or when cfg.map_hash_load asks for it -- an open-addressing hash over the occupied
// cells only (capacity = occupied / load factor, rounded up to a power of two).
int build_table(velo_ctx* c, MapView& mv, const uint32_t* keys_sorted, size_t n, size_t ncell)
{
    hipStream_t s = c->stream;
    if (!c->use_hash) {
        if (c->overlap_update && c->overlap_done == 0) {
            // a rebuild beside a registration (a re-anchoring roll begun ahead): the registration reads cell_start --
            // the new table goes into the other copy, as the incremental updates' shift / remap do
            std::swap(c->cell_start.p, c->cell_start_alt.p);
            std::swap(c->cell_start.cap, c->cell_start_alt.cap);
        } else if (ncell + 8 > c->cell_start.cap) {
            HIP_TRY(c, hipStreamSynchronize(s));  // queued readers
        }
        // +1 entry, padded: rows are read 4 entries at a time; with slack, because a rolling map's
        // grid grows a margin at a time and a fresh 0.3-0.6 GB allocation costs milliseconds
        HIP_TRY(c, reserve_slack(c->cell_start, ncell + 8));
        HIP_TRY(c, reserve_slack(c->tile_bounds, cell_start_bounds(ncell)));
        HIP_TRY(c, launch_cell_start(keys_sorted, n, ncell, c->cell_start.p, c->tile_bounds.p, s));
        mv.cell_start = c->cell_start.p;
        mv.hash = nullptr;
        mv.hash_cap = 0;
        mv.hash_stride = 1;
        mv.s_magic = 0;
        return VELO_OK;
    }
    // round 6: keyed by ROW PIECE (the S fine cells of a voxel along a fine row: velo_internal.hpp MapView), one probe per
    // voxel a row window touches instead of one per fine cell
    const int S = mv.S > 0 ? mv.S : c->map_S;
    HIP_TRY(c, c->run_cnt.reserve(2));
    HIP_TRY(c, launch_count_runs(keys_sorted, n, S, c->run_cnt.p, s));
    unsigned long long occ = 0;
    HIP_TRY(c, hipMemcpyAsync(&occ, c->run_cnt.p, sizeof occ, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const int pct = c->cfg.map_hash_load > 0 ? std::min(std::max(c->cfg.map_hash_load, 5), 90) : 50;
    const double want = std::ceil((double)std::max<unsigned long long>(occ, 1) * 100.0 / (double)pct) + 1.0;
    if (want >= 4294967295.0) return c->fail(VELO_E_RANGE, "hash table of %.3g slots", want);
    const size_t cap = (size_t)std::max(want, 16.0);
    const uint32_t stride = S > 4 ? 2u : 1u;
    if (c->overlap_update && c->overlap_done == 0) {
        // beside a registration (round 6: a hashed table is no longer a reason to refuse a roll begun ahead): the
        // registration probes hash_tab -- the new table goes into the other copy, as the dense table's does
        std::swap(c->hash_tab.p, c->hash_tab_alt.p);
        std::swap(c->hash_tab.cap, c->hash_tab_alt.cap);
    }
    HIP_TRY(c, c->hash_tab.reserve(cap * stride));
    unsigned* d_over = reinterpret_cast<unsigned*>(c->run_cnt.p + 1);
    HIP_TRY(c, launch_hash_build(keys_sorted, n, c->hash_tab.p, (uint32_t)cap, S, d_over, s));
    unsigned over = 0;
    HIP_TRY(c, hipMemcpyAsync(&over, d_over, sizeof over, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (over)
        return c->fail(VELO_E_RANGE, "a row piece of the sparse table holds 65 536 points or more (its 16-bit offsets do not fit): "
                       "use a smaller voxel or the dense table");
    mv.cell_start = nullptr;
    mv.hash = c->hash_tab.p;
    mv.hash_cap = (uint32_t)cap;
    mv.hash_stride = stride;
    // ceil(2^64 / S): 2^64 / S is whole for S = 2, 4, 8 (then ~0ull / S + 1 is exactly it), otherwise floor + 1
    mv.s_magic = S >= 2 ? (~0ull / (unsigned long long)S + 1ull) : 0ull;
    c->info.table_slots = cap;
    c->info.table_occupied = occ;
    return VELO_OK;
}

struct CarryNormals {
    const uint32_t* removed_keep = nullptr;  // flags over the OLD sorted order (0 = removed)
    uint32_t removed_n = 0;
};
int rebuild_map(velo_ctx* c, float voxel, int k_normals, const float* grid_org = nullptr,
                const int* grid_dims = nullptr, const CarryNormals* carry = nullptr)
{
    const size_t n = c->raw_n;
    hipStream_t s = c->stream;
    if (n == 0) return c->fail(VELO_E_INVALID, "map needs at least one point");
    if (!(voxel > 0.0f)) return c->fail(VELO_E_INVALID, "voxel must be > 0");
    if (k_normals < 0 || k_normals > VELO_MAX_KNORMALS)
        return c->fail(VELO_E_INVALID, "k_normals must be in [0,%d]", VELO_MAX_KNORMALS);
    if (n >= (size_t)INT32_MAX) return c->fail(VELO_E_RANGE, "map larger than 2^31-1 points");
    HIP_TRY(c, c->mm_scratch.reserve(8));
    MinMax mm;
    HIP_TRY(c, launch_minmax(c->raw_x.p, c->raw_y.p, c->raw_z.p, n, c->mm_scratch.p, &mm, s));
    const float inv_h = 1.0f / voxel;
    int dims[3];
    float org[3];
    double ncell_d = 1.0;
    for (int a = 0; a < 3; ++a) {
        const int M = c->margin[a];
        if (!std::isfinite(mm.mn[a]) || !std::isfinite(mm.mx[a]))
            return c->fail(VELO_E_INVALID, "map points must be finite");
        // anchor (oracle/icp.c roll_anchor): M voxels of slack below the lowest point and above
        // the highest, so that a rolling map keeps its grid across appends and evictions
        org[a] = grid_org ? grid_org[a] : mm.mn[a] - (float)M * voxel;
        const float ext = floorf((mm.mx[a] - org[a]) * inv_h);
        if (!(ext < 2.0e9f)) return c->fail(VELO_E_RANGE, "map extent / voxel too large");
        dims[a] = grid_dims ? grid_dims[a] : (int)ext + 1 + M;
        if (mm.mn[a] < org[a] || (int)ext + 1 > dims[a])
            return c->fail(VELO_E_INVALID, "explicit grid does not contain the map points");
        ncell_d *= (double)dims[a];
        c->map_mx[a] = mm.mx[a];
    }
    // sub-division: as configured, lowered (never raised) until the dense fine-cell table fits
    // 2^31 entries; the value actually used is reported in velo_map_info.subdiv
    // The dense table holds < 2^31 entries; past that (or on request) the table is a hash over
    // the occupied cells and only the 32-bit fine KEY limits the grid: < 2^32 - 1 cells.  The
    // sub-division is lowered (never raised) only when even that does not fit.
    int S = c->map_S;
    const double key_limit = 4294967295.0;
    while (S > 1 && ncell_d * (double)S * S * S >= key_limit) --S;
    ncell_d *= (double)S * S * S;
    if (ncell_d >= key_limit)
        return c->fail(VELO_E_RANGE, "voxel grid of %.3g cells exceeds the 32-bit fine key even "
                       "without sub-division", ncell_d);
    c->use_hash = c->cfg.map_hash_load > 0 || ncell_d >= 2147483648.0;
    const int fdims[3] = {dims[0] * S, dims[1] * S, dims[2] * S};
    const size_t ncell = (size_t)fdims[0] * fdims[1] * fdims[2];
    // (with slack: a rolling map re-anchors with a few more points every time -- exact sizes were four to six
    //  hipFree + hipMalloc pairs, 0.8 ms of host time, per re-anchoring roll)
    HIP_TRY(c, reserve_slack(c->keys, n));
    HIP_TRY(c, reserve_slack(c->keys_sorted, n));
    HIP_TRY(c, reserve_slack(c->idx, n));
    HIP_TRY(c, reserve_slack(c->perm, n));
    if (carry) {  // build into the second set of arrays: the old points are still needed
        overlap_rotate_alt(c);  // (beside a registration, second update of the call: the third set)
        HIP_TRY(c, reserve_slack(c->pts_alt, n));
        HIP_TRY(c, reserve_slack(c->nrm_alt, n));
    } else {
        HIP_TRY(c, reserve_slack(c->pts, n));
        HIP_TRY(c, reserve_slack(c->nrm, n));
    }
    HIP_TRY(c, c->invalid_cnt.reserve(1));
    HIP_TRY(c, launch_keys(c->raw_x.p, c->raw_y.p, c->raw_z.p, n, org[0], org[1], org[2],
                           inv_h, S, fdims[0], fdims[1], c->keys.p, c->idx.p, s));
    int bits = 1;
    while (bits < 32 && ((size_t)1 << bits) < ncell) ++bits;
    size_t tb = 0;
    HIP_TRY(c, sort_pairs(nullptr, tb, c->keys.p, c->keys_sorted.p, c->idx.p, c->perm.p, n, bits, s));
    if (int rc = ensure_temp(c, tb)) return rc;
    MapTxn txn(c);
    txn.touched = true;  // from here on the live arrays (keys_sorted, perm, pts, the table) are rewritten
    HIP_TRY(c, sort_pairs(c->temp.p, tb, c->keys.p, c->keys_sorted.p, c->idx.p, c->perm.p, n, bits, s));
    const float4* old_pts = c->pts.p;
    if (carry) {
        HIP_TRY(c, launch_gather(c->raw_x.p, c->raw_y.p, c->raw_z.p, c->perm.p, n, c->pts_alt.p, s));
        HIP_TRY(c, launch_gather_nrm(c->nrm_raw.p, c->perm.p, (uint32_t)n, c->nrm_alt.p, s));
        std::swap(c->pts.p, c->pts_alt.p);
        std::swap(c->pts.cap, c->pts_alt.cap);
        std::swap(c->nrm.p, c->nrm_alt.p);
        std::swap(c->nrm.cap, c->nrm_alt.cap);
    } else {
        HIP_TRY(c, launch_gather(c->raw_x.p, c->raw_y.p, c->raw_z.p, c->perm.p, n, c->pts.p, s));
    }
    MapView mv{};
    mv.S = S;   // (the sparse table is keyed by row piece = fine key / S)
    if (int rc = build_table(c, mv, c->keys_sorted.p, n, ncell)) return rc;
    mv.pts = c->pts.p;
    mv.nrm = c->nrm.p;
    mv.ox = org[0];
    mv.oy = org[1];
    mv.oz = org[2];
    mv.inv_h = inv_h;
    mv.h = voxel;
    mv.nx = dims[0];
    mv.ny = dims[1];
    mv.nz = dims[2];
    mv.S = S;
    mv.fx = fdims[0];
    mv.fy = fdims[1];
    mv.fz = fdims[2];
    mv.n = (int)n;
    unsigned long long invalid = n;
    if (carry && k_normals > 0) {
        const size_t nvox = (size_t)dims[0] * dims[1] * dims[2];
        HIP_TRY(c, reserve_slack(c->dirty, nvox));
        HIP_TRY(c, hipMemsetAsync(c->dirty.p, 0, nvox, s));
        HIP_TRY(c, launch_mark_dirty_pts(c->pts.p, c->nrm.p, nullptr, (uint32_t)n, mv, c->dirty.p, s));
        if (carry->removed_keep)
            HIP_TRY(c, launch_mark_dirty_pts(old_pts, nullptr, carry->removed_keep, carry->removed_n,
                                             mv, c->dirty.p, s));
        if (int rc = refresh_dirty_normals(c, mv, k_normals, nullptr, 0)) return rc;
        HIP_TRY(c, launch_count_invalid(c->nrm.p, (uint32_t)n, c->invalid_cnt.p, s));
        if (c->defer_counts) {   // (a roll begun ahead: the counts arrive in pinned memory, nobody waits here)
            HIP_TRY(c, hipMemcpyAsync(&c->h_roll->invalid, c->invalid_cnt.p, sizeof invalid, hipMemcpyDeviceToHost, s));
            c->roll_counts_pending = true;
        } else {
            HIP_TRY(c, hipMemcpyAsync(&invalid, c->invalid_cnt.p, sizeof invalid, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
        }
        return txn.done(publish_map(c, mv, k_normals, invalid, 0, c->n_done_host));
    }
    if (k_normals > 0) {
        HIP_TRY(c, launch_normals(mv, c->perm.p, k_normals, c->nrm.p, c->invalid_cnt.p, s, c->cfg.force_kernel));
        HIP_TRY(c, hipMemcpyAsync(&invalid, c->invalid_cnt.p, sizeof invalid, hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(c, hipMemsetAsync(c->nrm.p, 0, n * sizeof(float4), s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    return txn.done(publish_map(c, mv, k_normals, invalid, 0, k_normals > 0 ? n : 0));
}

// Sub-division of a freshly reset map: as configured, or (cfg.map_subdiv == 0) chosen from the
// density exactly as oracle/icp.c vo_auto_subdiv does: rho = points per occupied voxel,
// S = round(max(1.6 rho^0.2, 1.137 rho^0.314)) clamped to [2, 8].  Kept until the next velo_map_reset.
int resolve_subdiv(velo_ctx* c, float voxel)
{
    if (c->cfg.map_subdiv > 0) {
        c->map_S = c->cfg.map_subdiv;
        return VELO_OK;
    }
    c->map_S = 3;
    if (c->raw_n == 0 || !(voxel > 0.0f)) return VELO_OK;  // rebuild_map reports these
    hipStream_t s = c->stream;
    HIP_TRY(c, c->mm_scratch.reserve(8));
    MinMax mm;
    HIP_TRY(c, launch_minmax(c->raw_x.p, c->raw_y.p, c->raw_z.p, c->raw_n, c->mm_scratch.p, &mm, s));
    const float inv_h = 1.0f / voxel;
    size_t d[3];
    double nv = 1.0;
    for (int a = 0; a < 3; ++a) {
        if (!std::isfinite(mm.mn[a]) || !std::isfinite(mm.mx[a])) return VELO_OK;
        const float ext = floorf((mm.mx[a] - mm.mn[a]) * inv_h);
        if (!(ext < 2.0e9f)) return VELO_OK;
        d[a] = (size_t)ext + 1;
        nv *= (double)d[a];
    }
    if (nv >= 2147483648.0) {
        c->map_S = 1;
        return VELO_OK;
    }
    HIP_TRY(c, reserve_slack(c->vox_occ, (size_t)nv));
    HIP_TRY(c, c->invalid_cnt.reserve(1));
    HIP_TRY(c, launch_count_occupied_voxels(c->raw_x.p, c->raw_y.p, c->raw_z.p, c->raw_n, mm.mn, inv_h, d,
                                            c->vox_occ.p, c->invalid_cnt.p, s));
    unsigned long long occ = 0;
    HIP_TRY(c, hipMemcpyAsync(&occ, c->invalid_cnt.p, sizeof occ, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (occ == 0) return VELO_OK;
    const double rho = (double)c->raw_n / (double)occ;
    const double a = 1.6 * std::pow(rho, 0.2), b = 1.137 * std::pow(rho, 0.314);
    const int S = (int)std::floor((a > b ? a : b) + 0.5);
    c->map_S = S < 2 ? 2 : (S > 8 ? 8 : S);
    return VELO_OK;
}

int stage_raw(velo_ctx* c, const float* x, const float* y, const float* z, size_t n, bool dev,
              bool append)
{
    if (!x || !y || !z) return c->fail(VELO_E_INVALID, "null point array");
    const size_t base = append ? c->raw_n : 0;
    const size_t total = base + n;
    HIP_TRY(c, c->raw_x.reserve(total, append, c->stream));
    HIP_TRY(c, c->raw_y.reserve(total, append, c->stream));
    HIP_TRY(c, c->raw_z.reserve(total, append, c->stream));
    const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    HIP_TRY(c, hipMemcpyAsync(c->raw_x.p + base, x, n * sizeof(float), kind, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->raw_y.p + base, y, n * sizeof(float), kind, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->raw_z.p + base, z, n * sizeof(float), kind, c->stream));
    c->raw_n = total;
    return VELO_OK;
}

// Append raw points [n_old, n_old+m) (already staged) without re-sorting the map.  Returns
// 1 if the grid cannot be kept (caller does the full, re-anchoring rebuild), 0 when done.
int append_incremental(velo_ctx* c, size_t n_old, size_t m, int* done, int* grown_dims = nullptr)
{
    *done = 0;
    if (grown_dims) grown_dims[0] = grown_dims[1] = grown_dims[2] = 0;
    hipStream_t s = c->stream;
    const MapView old = c->mv;
    const int k = c->info.k_normals;
    if (n_old + m >= (size_t)INT32_MAX) return c->fail(VELO_E_RANGE, "map larger than 2^31-1 points");
    HIP_TRY(c, c->mm_scratch.reserve(8));
    MinMax mm;
    if (c->defer_counts && c->have_enter_mm) {  // (host arrays: their bounds were taken on the host, no device pass, no wait)
        for (int a = 0; a < 3; ++a) {
            mm.mn[a] = c->enter_mn[a];
            mm.mx[a] = c->enter_mx[a];
        }
    } else {
        HIP_TRY(c, launch_minmax(c->raw_x.p + n_old, c->raw_y.p + n_old, c->raw_z.p + n_old, m,
                                 c->mm_scratch.p, &mm, s));
    }
    const float org[3] = {old.ox, old.oy, old.oz};
    int dims[3] = {old.nx, old.ny, old.nz};
    bool grew = false;
    float mx[3];
    for (int a = 0; a < 3; ++a) {
        if (!std::isfinite(mm.mn[a]) || !std::isfinite(mm.mx[a]))
            return c->fail(VELO_E_INVALID, "map points must be finite");
        if (mm.mn[a] < org[a]) return VELO_OK;  // below the origin: re-anchor (done == 0)
        mx[a] = std::max(c->map_mx[a], mm.mx[a]);
        const float ext = floorf((mx[a] - org[a]) * old.inv_h);
        if (!(ext < 2.0e9f)) return c->fail(VELO_E_RANGE, "map extent / voxel too large");
        const int need = (int)ext + 1;
        if (need > dims[a]) {
            dims[a] = need + c->margin[a];
            grew = true;
        }
    }
    const int S = old.S;
    const double ncell_d = (double)dims[0] * dims[1] * dims[2] * (double)S * S * S;
    // past the table's limit: let the full path switch to the sparse table, lower S or refuse
    if (ncell_d >= (c->use_hash ? 4294967295.0 : 2147483648.0)) return VELO_OK;
    MapView g = old;
    g.nx = dims[0];
    g.ny = dims[1];
    g.nz = dims[2];
    g.fx = dims[0] * S;
    g.fy = dims[1] * S;
    g.fz = dims[2] * S;
    const size_t ncell = (size_t)g.fx * g.fy * g.fz;
    const size_t total = n_old + m;
    if (c->overlap_update && (grew || c->cfg.map_full_rebuild)) {
        // (done == 0.  A grown grid beside a registration: the caller rebuilds on THIS grid -- same origin, grown dims --
        //  into the other copies, which is what the in-place re-encoding below leaves, bit for bit)
        if (grew && grown_dims)
            for (int a = 0; a < 3; ++a) grown_dims[a] = dims[a];
        return VELO_OK;
    }
    if (c->cfg.map_full_rebuild) {  // A/B switch: same grid, everything recomputed
        *done = 1;
        return rebuild_map(c, old.h, k, org, dims);
    }
    // the dims moved: every old key is re-encoded (the ORDER does not change -- keys are
    // lexicographic in (Fz, Fy, Fx) whatever the row lengths are)
    if (grew) HIP_TRY(c, launch_keys4(c->pts.p, n_old, g, c->keys_sorted.p, s));
    HIP_TRY(c, reserve_slack(c->nk, m));
    HIP_TRY(c, reserve_slack(c->nk_sorted, m));
    HIP_TRY(c, reserve_slack(c->nidx, m));
    HIP_TRY(c, reserve_slack(c->nidx_sorted, m));
    HIP_TRY(c, launch_keys(c->raw_x.p + n_old, c->raw_y.p + n_old, c->raw_z.p + n_old, m, org[0],
                           org[1], org[2], old.inv_h, S, g.fx, g.fy, c->nk.p, c->nidx.p, s));
    int bits = 1;
    while (bits < 32 && ((size_t)1 << bits) < ncell) ++bits;
    size_t tb = 0;
    HIP_TRY(c, sort_pairs(nullptr, tb, c->nk.p, c->nk_sorted.p, c->nidx.p, c->nidx_sorted.p, m, bits, s));
    if (int rc = ensure_temp(c, tb)) return rc;
    HIP_TRY(c, sort_pairs(c->temp.p, tb, c->nk.p, c->nk_sorted.p, c->nidx.p, c->nidx_sorted.p, m, bits, s));
    overlap_rotate_alt(c);
    HIP_TRY(c, reserve_slack(c->pts_alt, total));
    HIP_TRY(c, reserve_slack(c->nrm_alt, total));
    HIP_TRY(c, reserve_slack(c->perm_alt, total));
    HIP_TRY(c, reserve_slack(c->keys_alt, total));
    MapTxn txn(c);
    txn.touched = true;  // (the merge writes the second set of arrays, but the table is updated in place)
    HIP_TRY(c, launch_merge(c->pts.p, c->nrm.p, c->perm.p, c->keys_sorted.p, (uint32_t)n_old,
                            c->raw_x.p, c->raw_y.p, c->raw_z.p, (uint32_t)n_old, c->nk_sorted.p,
                            c->nidx_sorted.p, (uint32_t)m, c->pts_alt.p, c->nrm_alt.p,
                            c->perm_alt.p, c->keys_alt.p, s));
    if (c->use_hash) {  // sparse table: re-hashed from the merged keys (O(points), no table pass)
        if (int rc = build_table(c, g, c->keys_alt.p, total, ncell)) return rc;
    } else if (grew) {
        HIP_TRY(c, hipStreamSynchronize(s));  // the old table may still be read by queued work
        HIP_TRY(c, reserve_slack(c->cell_start, ncell + 8));
        HIP_TRY(c, reserve_slack(c->tile_bounds, cell_start_bounds(ncell)));
        HIP_TRY(c, launch_cell_start(c->keys_alt.p, total, ncell, c->cell_start.p, c->tile_bounds.p, s));
    } else if (c->overlap_update && c->overlap_done == 0) {  // the running registration reads cell_start: write the other copy
        HIP_TRY(c, reserve_slack(c->cell_start_alt, ncell + 8));
        HIP_TRY(c, reserve_slack(c->tile_bounds, table_tile_bounds(ncell + 1)));
        HIP_TRY(c, launch_table_shift(c->cell_start.p, c->cell_start_alt.p, ncell + 1, c->nk_sorted.p, (uint32_t)m, c->tile_bounds.p, s));
        std::swap(c->cell_start.p, c->cell_start_alt.p);
        std::swap(c->cell_start.cap, c->cell_start_alt.cap);
    } else {
        HIP_TRY(c, reserve_slack(c->tile_bounds, table_tile_bounds(ncell + 1)));
        HIP_TRY(c, launch_table_shift(c->cell_start.p, c->cell_start.p, ncell + 1, c->nk_sorted.p, (uint32_t)m, c->tile_bounds.p, s));
    }
    std::swap(c->pts.p, c->pts_alt.p);
    std::swap(c->pts.cap, c->pts_alt.cap);
    std::swap(c->nrm.p, c->nrm_alt.p);
    std::swap(c->nrm.cap, c->nrm_alt.cap);
    std::swap(c->perm.p, c->perm_alt.p);
    std::swap(c->perm.cap, c->perm_alt.cap);
    std::swap(c->keys_sorted.p, c->keys_alt.p);
    std::swap(c->keys_sorted.cap, c->keys_alt.cap);
    g.pts = c->pts.p;
    g.nrm = c->nrm.p;
    if (!c->use_hash) g.cell_start = c->cell_start.p;
    g.n = (int)total;
    unsigned long long invalid = total;
    c->n_done_host = 0;
    if (k > 0) {
        const size_t nvox = (size_t)dims[0] * dims[1] * dims[2];
        HIP_TRY(c, reserve_slack(c->dirty, nvox));
        HIP_TRY(c, hipMemsetAsync(c->dirty.p, 0, nvox, s));
        if (int rc = mark_roll_geometry(c)) return rc;   // (points + table of the updated map are enqueued: the normals follow)
        HIP_TRY(c, launch_mark_dirty(c->nk_sorted.p, (uint32_t)m, nullptr, g, c->dirty.p, s));
        if (int rc = refresh_dirty_normals(c, g, k, c->nk_sorted.p, (uint32_t)m)) return rc;
        if (c->defer_counts)
            HIP_TRY(c, hipMemcpyAsync(&c->h_roll->invalid, c->invalid_cnt.p, sizeof invalid, hipMemcpyDeviceToHost, s));
        else
            HIP_TRY(c, hipMemcpyAsync(&invalid, c->invalid_cnt.p, sizeof invalid, hipMemcpyDeviceToHost, s));
    }
    if (!c->defer_counts) HIP_TRY(c, hipStreamSynchronize(s));
    for (int a = 0; a < 3; ++a) c->map_mx[a] = mx[a];
    *done = 1;
    if (c->defer_counts) c->roll_counts_pending = k > 0;
    return txn.done(publish_map(c, g, k, invalid, 1, c->n_done_host));
}


// work decomposition of the linearise kernel over the resident frames
// Launch order of a frame-major item list (rows of the partial-sum buffer -- `slot` -- are not
// affected).  0 = as is; 1 = item-major: item i of every frame, then item i+1 ... -- the
// workgroups in flight together then look at the same region of the map; 2 = item-major from
// the last item of the frames backwards.
std::vector<BlockItem> launch_order(const std::vector<BlockItem>& items, int mode)
{
    if (mode == 0 || items.empty()) return items;
    std::vector<std::vector<BlockItem>> per_frame;
    for (const BlockItem& it : items) {
        if (per_frame.empty() || per_frame.back().back().frame != it.frame) per_frame.emplace_back();
        per_frame.back().push_back(it);
    }
    size_t longest = 0;
    for (auto& f : per_frame) longest = std::max(longest, f.size());
    std::vector<BlockItem> out;
    out.reserve(items.size());
    if (mode == 5) {
        // frame-major inside each XCD, eighth x of every frame on XCD x (slabs)
        std::vector<std::vector<BlockItem>> q(8);
        for (auto& f : per_frame)
            for (size_t i = 0; i < f.size(); ++i) q[i * 8 / f.size()].push_back(f[i]);
        size_t most = 0;
        for (auto& v : q) most = std::max(most, v.size());
        for (size_t t = 0; t < most; ++t)
            for (int x = 0; x < 8; ++x)
                if (t < q[(size_t)x].size()) out.push_back(q[(size_t)x][t]);
        return out;
    }
    if (mode == 3 || mode == 4) {
        // item-major, the item indices dealt over the 8 XCDs (workgroup b runs on XCD b % 8, each
        // with its own L2): XCD x gets the items i with i % 8 == x of every frame, so its L2 holds
        // an eighth of the map regions at a time instead of all of them.  4: from the ends backwards.
        std::vector<std::vector<BlockItem>> q(8);
        for (size_t k = 0; k < longest; ++k) {
            const size_t i = mode == 4 ? longest - 1 - k : k;
            for (auto& f : per_frame)
                if (i < f.size()) q[k % 8].push_back(f[i]);
        }
        size_t most = 0;
        for (auto& v : q) most = std::max(most, v.size());
        for (size_t t = 0; t < most; ++t)
            for (int x = 0; x < 8; ++x)
                if (t < q[(size_t)x].size()) out.push_back(q[(size_t)x][t]);
        return out;
    }
    for (size_t k = 0; k < longest; ++k) {
        const size_t i = mode == 2 ? longest - 1 - k : k;
        for (auto& f : per_frame)
            if (i < f.size()) out.push_back(f[i]);
    }
    return out;
}

// A decomposition of the resident frames for the hinted iterations of a BATCH: `rounds` rounds of
// a workgroup per item.  The last VELO_LATE_TAIL_PCT % of every frame's queries are cut into
// one-round items and all of those are launched AFTER the big ones: the launch then drains in
// small pieces instead of ending on a row of long workgroups.  (Rows of the partial-sum buffer --
// `slot` -- stay frame-major whatever the launch order.)
// rounds per wavefront of a decomposition: as many as `r_max`, as long as the launch keeps
// VELO_MIN_SLOT_ROUNDS wavefronts per wavefront slot of the device (measured on 16 / 32 / 64
// frames: a launch cut coarser than that loses more in its tail than it saves in fixed cost)
int rounds_per_wave(const velo_ctx* c, int64_t n_queries, int r_max)
{
    const double wave_rounds = (double)n_queries / 64.0;
    int r = 1;
    while (r * 2 <= std::min(r_max, 4)) r *= 2;  // 1 / 2 / 4: an item is an aligned node of the summation tree
    while (r > 1 && wave_rounds / r < VELO_MIN_SLOT_ROUNDS * (double)c->wave_slots) r >>= 1;
    return r;
}

// ---- PlanUpload: stage now, copy once, bind the views (plan_flush)
template <typename T>
int plan_add(velo_ctx* c, DevBuf<T>& dst, const T* src, size_t n)
{
    velo_ctx::PlanUpload& u = c->plan_up;
    const size_t off = (u.used + 255) & ~(size_t)255, bytes = n * sizeof(T);
    if (off + bytes > u.h_cap) {
        const size_t want = std::max<size_t>((off + bytes) * 2, 1 << 16);
        uint8_t* q = nullptr;
        HIP_TRY(c, hipHostMalloc((void**)&q, want, 0));
        if (u.h) {
            std::memcpy(q, u.h, u.used);
            (void)hipHostFree(u.h);
        }
        u.h = q;
        u.h_cap = want;
    }
    if (bytes) std::memcpy(u.h + off, src, bytes);
    u.used = off + bytes;
    u.slices.push_back({[](void* buf, uint8_t* base, size_t o) { static_cast<DevBuf<T>*>(buf)->alias(reinterpret_cast<T*>(base + o)); },
                        &dst, off});
    return VELO_OK;
}

int plan_flush(velo_ctx* c)
{
    velo_ctx::PlanUpload& u = c->plan_up;
    u.sel ^= 1;
    DevBuf<uint8_t>& d = u.d[u.sel];
    if (u.used) {
        HIP_TRY(c, d.reserve(u.used + u.used / 2));
        HIP_TRY(c, hipMemcpyAsync(d.p, u.h, u.used, hipMemcpyHostToDevice, c->stream));
    }
    for (const auto& sl : u.slices) sl.bind(sl.buf, d.p, sl.off);
    u.slices.clear();
    u.used = 0;
    return VELO_OK;
}

static int log2i(int v)
{
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}

int plan_decomposition(velo_ctx* c, const int64_t* frame_start, int n_frames, int rounds, int tail_pct,
                       int order, DevBuf<BlockItem>& d_items, DevBuf<int32_t>& d_fbs, DevBuf<RowLayout>& d_lay,
                       RowLayout& lay0, std::vector<RowLayout>& lay_h, int& n_out)
{
    std::vector<BlockItem> big, tail;
    std::vector<int32_t> fbl((size_t)n_frames + 1, 0);
    lay_h.assign((size_t)n_frames, RowLayout{0, 0, 0, 0});
    const int per_big = kLinNT * rounds;   // rounds: 1 / 2 / 4 (rounds_per_wave)
    const int32_t rbits = (int32_t)((uint32_t)log2i(rounds) << kItemRowBits);
    int32_t slot = 0;
    for (int f = 0; f < n_frames; ++f) {
        fbl[f] = slot;
        const int64_t nqf = frame_start[f + 1] - frame_start[f];
        // (both regions start at multiples of their item size from the frame's first query: every item is an
        // aligned node of the frame's summation tree)
        const int64_t big_end = tail_pct <= 0 ? frame_start[f + 1]   // (no tail: the last large item is a partial one)
                                              : frame_start[f] + (nqf * (100 - tail_pct) / 100) / per_big * per_big;
        int nbig = 0;
        for (int64_t q = frame_start[f]; q < frame_start[f + 1];) {
            const int64_t step = q < big_end ? per_big : kLinNT;
            BlockItem it;
            it.frame = f;
            it.q0 = (int32_t)q;
            it.q1 = (int32_t)std::min<int64_t>(q + step, frame_start[f + 1]);
            it.slot = slot++ | (q < big_end ? rbits : 0);
            nbig += q < big_end ? 1 : 0;
            (q < big_end ? big : tail).push_back(it);
            q += step;
        }
        // (no one-round tail for this frame: its rows are of one size, one slot each -- the solve's short path)
        if (big_end >= frame_start[f + 1] || rounds == 1)
            lay_h[(size_t)f] = RowLayout{0, 0, 0, slot - fbl[f]};
        else
            lay_h[(size_t)f] = RowLayout{0, nbig, log2i(rounds), (int32_t)((nqf + kLinNT - 1) / kLinNT)};
    }
    fbl[n_frames] = slot;
    if (slot >= (1 << kItemRowBits)) return c->fail(VELO_E_RANGE, "too many work items");
    big = launch_order(big, order);
    tail = launch_order(tail, order);
    big.insert(big.end(), tail.begin(), tail.end());
    if (int rc = plan_add(c, d_items, big.data(), big.size())) return rc;
    if (int rc = plan_add(c, d_fbs, fbl.data(), fbl.size())) return rc;
    if (int rc = plan_add(c, d_lay, lay_h.data(), lay_h.size())) return rc;
    lay0 = lay_h[0];
    n_out = (int)big.size();
    return VELO_OK;
}

// which work items iteration `it` of a registration runs on (plan_frames)
struct Decomposition {
    const BlockItem* items;
    const int32_t* fbs;
    int n;
    int lat_lanes;  // latency kernel: queries per wavefront the items are cut for
    const RowLayout* lay;   // device, per frame
    const RowLayout* lay0;  // host, frame 0's
};

Decomposition decomposition_for(velo_ctx* c, int it, bool hinted, bool sorted_queries)
{
    if (sorted_queries)
        return {c->cfg.sort_frames == 1 ? c->items_xcd.p : c->items_first.p, c->fbs.p, (int)c->items_h.size(),
                c->lat_first_lanes, c->lay.p, &c->lay0_h};
    if (it >= VELO_CONV_FROM && c->ni_conv > 0 && hinted)
        return {c->items_conv.p, c->fbs_conv.p, c->ni_conv, 64, c->lay_conv.p, &c->lay0_conv_h};
    if (it > 0 && c->ni_late > 0 && hinted) return {c->items_late.p, c->fbs_late.p, c->ni_late, 64, c->lay_late.p, &c->lay0_late_h};
    // latency kernel, sparse first iteration: it pays where stragglers are expensive -- fine cells
    // full of points (S >= 4: first launch 304 -> 132 us on a 9 M-point map); on a light map (S = 3,
    // 1 M points) the 13 us it saves are less than the eight times as many partial rows cost the
    // solve: there the 256-query items serve every iteration
    if (c->plan_lat && c->lat_first_lanes < 64 && c->ni_late > 0 && c->mv_read.S < VELO_LAT_SPARSE_MIN_S)
        return {c->items_late.p, c->fbs_late.p, c->ni_late, 64, c->lay_late.p, &c->lay0_late_h};
    return {c->items_first.p, c->fbs.p, (int)c->items_h.size(), c->lat_first_lanes, c->lay.p, &c->lay0_h};
}

int plan_frames(velo_ctx* c, int n_frames, const int64_t* frame_start)
{
    if (n_frames < 1) return c->fail(VELO_E_INVALID, "n_frames must be >= 1");
    const int maxb = c->cfg.max_batch;
    if (n_frames > maxb) return c->fail(VELO_E_RANGE, "n_frames %d > cfg.max_batch %d", n_frames, maxb);
    if (!frame_start || frame_start[0] != 0) return c->fail(VELO_E_INVALID, "frame_start[0] must be 0");
    for (int f = 0; f < n_frames; ++f)
        if (frame_start[f + 1] < frame_start[f]) return c->fail(VELO_E_INVALID, "frame_start must ascend");
    if (frame_start[n_frames] >= INT32_MAX) return c->fail(VELO_E_RANGE, "too many query points");
    // the previous plan's uploads read the host vectors rewritten below: normally long complete
    if (c->ev_plan)
        HIP_TRY(c, hipEventSynchronize(c->ev_plan));
    else
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming));
    c->plan_up.slices.clear();  // (a plan that failed half way leaves nothing behind)
    c->plan_up.used = 0;
    c->n_frames = n_frames;
    ++c->frames_gen;
    c->frame_start.assign(frame_start, frame_start + n_frames + 1);
    c->items_h.clear();
    c->fbs_h.assign((size_t)n_frames + 1, 0);
    // Which kernel the registrations of these frames run on: fewer than kLatQueries queries (~4
    // frames) leave most of the chip idle -- a latency problem, 256-query items for the latency
    // kernel (which exists for the ball search only).  Otherwise the throughput kernel (workgroups of
    // kLinNT threads), whose items are sized in rounds per wavefront (rounds_per_wave): up to 4 in
    // the first, unhinted iteration.
    const int64_t total_q = frame_start[n_frames];
    c->plan_lat = c->cfg.linearize_variant == VELO_VARIANT_BALL &&
                  (c->cfg.force_kernel == 2 || (c->cfg.force_kernel != 1 && total_q < kLatQueries));
    const int nt = c->plan_lat ? kLinThreads : kLinNT;  // threads of the workgroups these items are cut for
    // Every item is an ALIGNED NODE of its frame's summation tree (kernels/icp.hip, linearize_body): nt x R
    // queries, R = rounds per wavefront in {1, 2, 4}, starting a multiple of that from the frame's first query.
    // cfg.rounds_per_block (rounds of 256 queries per item, tuning / tests) is rounded down to such a size.
    int R = 1;
    if (c->cfg.rounds_per_block > 0) {
        const int want = std::min(c->cfg.rounds_per_block, 64) * kLinThreads / nt;
        while (R * 2 <= std::min(want, 4)) R *= 2;
    }
    const bool planned = c->cfg.rounds_per_block <= 0 && !c->plan_lat;
    if (planned) R = rounds_per_wave(c, total_q, VELO_FIRST_ROUNDS);
    int per_block = nt * R;
    // latency kernel: the first iteration runs kLatFirstLanes queries per wavefront (one round per
    // item), the hinted ones all 64 -- a second, 256-query decomposition below
    const bool lat_sparse = c->plan_lat && c->cfg.rounds_per_block <= 0 && kLatFirstLanes < 64;
    c->lat_first_lanes = lat_sparse ? kLatFirstLanes : 64;
    if (lat_sparse) {
        R = 1;
        per_block = (kLinThreads / 64) * kLatFirstLanes;
    }
    const int32_t rbits = (int32_t)((uint32_t)log2i(R) << kItemRowBits);
    std::vector<RowLayout>& lay_h = c->plan_lay_h;
    lay_h.assign((size_t)n_frames, RowLayout{0, 0, 0, 0});
    for (int f = 0; f < n_frames; ++f) {
        c->fbs_h[f] = (int32_t)c->items_h.size();
        // (planned: the head of every frame in one-round items -- launched item-major from the
        // ends of the frames backwards, those are the last workgroups of the launch; the head ends on a
        // multiple of the large item size, so that the large items are aligned too)
        const int64_t nqf = frame_start[f + 1] - frame_start[f];
        const int64_t head_end = planned ? frame_start[f] + nqf * VELO_FIRST_HEAD_PCT / 100 / per_block * per_block
                                         : frame_start[f];
        int nbig = 0;
        for (int64_t q = frame_start[f]; q < frame_start[f + 1];) {
            const int64_t step = q < head_end ? kLinNT : per_block;
            BlockItem it;
            it.frame = f;
            it.q0 = (int32_t)q;
            it.q1 = (int32_t)std::min<int64_t>(q + step, frame_start[f + 1]);
            it.slot = (int32_t)c->items_h.size() | (q < head_end ? 0 : rbits);
            nbig += q < head_end ? 0 : 1;
            c->items_h.push_back(it);
            q += step;
        }
        if (planned && R > 1)   // head small rows (kLinNT queries), then large ones: slots of kLinNT queries
            lay_h[(size_t)f] = RowLayout{(int32_t)((head_end - frame_start[f]) / kLinNT), nbig, log2i(R),
                                         (int32_t)((nqf + kLinNT - 1) / kLinNT)};
        else           // uniform rows, one slot each -- also a planned batch at one round per wavefront, whose "large"
                       // rows are the small size: the mixed layout sent every one of its 16-slot spans down the
                       // solve's per-slot path (plan_decomposition already collapses this case)
            lay_h[(size_t)f] = RowLayout{0, 0, 0, (int32_t)c->items_h.size() - c->fbs_h[f]};
    }
    if (c->items_h.size() >= ((size_t)1 << kItemRowBits)) return c->fail(VELO_E_RANGE, "too many work items");
    for (int f = 0; f < n_frames; ++f)
        if (lay_h[(size_t)f].nslots > kMaxRowSlots)
            return c->fail(VELO_E_RANGE, "frame %d has %lld points: more than %lld per frame", f,
                           (long long)(frame_start[f + 1] - frame_start[f]), (long long)kMaxRowSlots * (planned && R > 1 ? kLinNT : per_block));
    c->lay0_h = lay_h[0];
    c->fbs_h[n_frames] = (int32_t)c->items_h.size();
    const size_t ni = c->items_h.size();
    // The hinted iterations of a batch run on coarser items: their work is even (hinted /
    // certified queries), so fewer workgroups -- fewer item / pose loads, block reductions,
    // partial rows for the solve to read -- are cheaper: up to VELO_LATE_ROUNDS rounds per
    // wavefront while a good share of the queries still searches (iterations 1 .. VELO_CONV_FROM-1),
    // up to VELO_CONV_ROUNDS afterwards (64 frames: converged launch 84 us on the first
    // decomposition, 71 at three rounds, 68 at six, 76 at twelve).
    c->ni_late = 0;
    c->ni_conv = 0;
    size_t max_rows = std::max<size_t>(ni, 1);
    if (planned) {
        if (int rc = plan_decomposition(c, frame_start, n_frames, rounds_per_wave(c, total_q, VELO_LATE_ROUNDS),
                                        VELO_LATE_TAIL_PCT, VELO_ORDER_LATE, c->items_late, c->fbs_late, c->lay_late,
                                        c->lay0_late_h, c->plan_lay2_h, c->ni_late))
            return rc;
        if (int rc = plan_decomposition(c, frame_start, n_frames, rounds_per_wave(c, total_q, VELO_CONV_ROUNDS),
                                        VELO_CONV_TAIL_PCT, VELO_ORDER_CONV, c->items_conv, c->fbs_conv, c->lay_conv,
                                        c->lay0_conv_h, c->plan_lay2_h, c->ni_conv))
            return rc;
        max_rows = std::max(max_rows, (size_t)std::max(c->ni_late, c->ni_conv));
    }
    if (lat_sparse) {  // plain 256-query items, frame-major, for the hinted iterations
        std::vector<BlockItem>& late = c->plan_late_h;
        std::vector<int32_t>& fbl = c->plan_fbl_h;
        late.clear();
        fbl.assign((size_t)n_frames + 1, 0);
        for (int f = 0; f < n_frames; ++f) {
            fbl[f] = (int32_t)late.size();
            for (int64_t q = frame_start[f]; q < frame_start[f + 1]; q += kLinThreads) {
                BlockItem it;
                it.frame = f;
                it.q0 = (int32_t)q;
                it.q1 = (int32_t)std::min<int64_t>(q + kLinThreads, frame_start[f + 1]);
                it.slot = (int32_t)late.size();
                late.push_back(it);
            }
        }
        fbl[n_frames] = (int32_t)late.size();
        if (!late.empty()) {
            std::vector<RowLayout>& l2 = c->plan_lay2_h;   // uniform 256-query rows: one slot each
            l2.assign((size_t)n_frames, RowLayout{0, 0, 0, 0});
            for (int f = 0; f < n_frames; ++f) l2[(size_t)f].nslots = fbl[f + 1] - fbl[f];
            if (int rc = plan_add(c, c->items_late, late.data(), late.size())) return rc;
            if (int rc = plan_add(c, c->fbs_late, fbl.data(), fbl.size())) return rc;
            if (int rc = plan_add(c, c->lay_late, l2.data(), l2.size())) return rc;
            c->lay0_late_h = l2[0];
            c->ni_late = (int)late.size();
        }
    }
    // (with slack: in the pipelined stream plan_frames runs on the side stream while a registration still reads
    //  these rows -- a frame a few points larger than every one before must not reallocate under it)
    //  Sized ONCE for the largest batch of real frames -- max_batch revolutions of 64 lasers x HDL_MAX_PTS_PER_LASER
    //  2200 returns (type_defs.h:20), cut into the smallest items -- so that a short first sweep followed by full
    //  revolutions never reallocates (a hipFree under a running registration is safe only through its device-wide
    //  wait, which is the stall the pipeline exists to avoid); larger synthetic frames still grow it.
    const size_t rows_bound = (size_t)std::max(maxb, n_frames) * ((size_t)(64 * 2200 + kLinNT - 1) / kLinNT + 2);
    HIP_TRY(c, reserve_slack(c->partials, std::max(rows_bound, max_rows + max_rows / 4 + 64) * kAccStride));
    HIP_TRY(c, c->poses.reserve((size_t)maxb * 12));
    HIP_TRY(c, c->acc.reserve((size_t)maxb * kAccStride));
    HIP_TRY(c, c->stats.reserve((size_t)maxb * VELO_MAX_ITERS));
    std::vector<BlockItem>& lo = c->plan_lo_h;  // the same items in launch order (registrations)
    lo.clear();
    if (int rc = plan_add(c, c->items, c->items_h.data(), ni)) return rc;
    if (ni) lo = launch_order(c->items_h, VELO_ORDER_FIRST);
    if (int rc = plan_add(c, c->items_first, lo.data(), lo.size())) return rc;
    std::vector<BlockItem>& xcd = c->plan_xcd_h;
    xcd.clear();
    if (c->cfg.sort_frames == 1 && ni) {
        // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share an XCD, each
        // with its own 4 MiB L2).  With cell-sorted queries, eighth r of every frame covers
        // roughly the same slab of the map, so give all of slab r to one XCD: its L2 then
        // holds 1/8 of the map instead of all of it.  Speed only -- never correctness.
        std::vector<std::vector<BlockItem>> region(8);
        for (int f = 0; f < n_frames; ++f) {
            const int b0 = c->fbs_h[f], nb = c->fbs_h[f + 1] - b0;
            for (int b = 0; b < nb; ++b) region[(size_t)((int64_t)b * 8 / std::max(nb, 1))].push_back(c->items_h[(size_t)b0 + b]);
        }
        size_t longest = 0;
        for (auto& r : region) longest = std::max(longest, r.size());
        xcd.reserve(ni);
        for (size_t t = 0; t < longest; ++t)
            for (int r = 0; r < 8; ++r)
                if (t < region[r].size()) xcd.push_back(region[r][t]);
        if (int rc = plan_add(c, c->items_xcd, xcd.data(), ni)) return rc;
    }
    if (int rc = plan_add(c, c->fbs, c->fbs_h.data(), (size_t)n_frames + 1)) return rc;
    if (int rc = plan_add(c, c->lay, lay_h.data(), lay_h.size())) return rc;
    if (int rc = plan_add(c, c->d_frame_start, c->frame_start.data(), (size_t)n_frames + 1)) return rc;
    // one copy for all of the above; its source is the pinned stage, free again at ev_plan
    if (int rc = plan_flush(c)) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_plan, c->stream));
    return VELO_OK;
}

hipEvent_t next_event(velo_ctx* c)
{
    if (c->ev_used == c->ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev.push_back(e);
    }
    return c->ev[c->ev_used++];
}

struct Timed {  // brackets one launch with events when timing is on
    velo_ctx* c;
    Timed(velo_ctx* ctx, int kind) : c(ctx)
    {
        if (!c->timing) return;
        hipEvent_t e = next_event(c);
        if (e) (void)hipEventRecord(e, c->stream);
        c->ev_kind.push_back(kind);
    }
    ~Timed()
    {
        if (!c->timing) return;
        hipEvent_t e = next_event(c);
        if (e) (void)hipEventRecord(e, c->stream);
    }
};

int maybe_sort_frames(velo_ctx* c, FrameView& fv)
{
    fv.order = nullptr;
    if (!c->cfg.sort_frames) return VELO_OK;
    const size_t n = (size_t)c->frame_start[c->n_frames];
    const double span = (double)c->n_frames * ((double)c->mv_read.nx * c->mv_read.ny * c->mv_read.nz + 1.0);
    if (n == 0 || span >= 4294967296.0) return VELO_OK;  // composite key would not fit 32 bits
    HIP_TRY(c, c->order_keys.reserve(n));
    HIP_TRY(c, c->order_keys2.reserve(n));
    HIP_TRY(c, c->order_idx.reserve(n));
    HIP_TRY(c, c->order.reserve(n));
    HIP_TRY(c, launch_frame_cellkeys(fv, c->d_frame_start.p, c->n_frames, n, c->mv_read, c->poses.p,
                                     c->order_keys.p, c->order_idx.p, c->stream));
    int bits = 1;
    while (bits < 32 && (double)((uint64_t)1 << bits) < span) ++bits;
    size_t tb = 0;
    HIP_TRY(c, sort_pairs(nullptr, tb, c->order_keys.p, c->order_keys2.p, c->order_idx.p,
                          c->order.p, n, bits, c->stream));
    if (int rc = ensure_temp(c, tb)) return rc;
    HIP_TRY(c, sort_pairs(c->temp.p, tb, c->order_keys.p, c->order_keys2.p, c->order_idx.p,
                          c->order.p, n, bits, c->stream));
    HIP_TRY(c, c->sx.reserve(n));
    HIP_TRY(c, c->sy.reserve(n));
    HIP_TRY(c, c->sz.reserve(n));
    HIP_TRY(c, launch_permute3(fv.x, fv.y, fv.z, c->order.p, n, c->sx.p, c->sy.p, c->sz.p, c->stream));
    fv.x = c->sx.p;
    fv.y = c->sy.p;
    fv.z = c->sz.p;
    fv.order = reinterpret_cast<const int32_t*>(c->order.p);
    return VELO_OK;
}

// initial poses -> device through the pinned double buffer (never blocks on queued work
// younger than two uploads ago; a pageable-source hipMemcpyAsync would block every time)
int upload_T0(velo_ctx* c, const double* T0, size_t pose_bytes)
{
    const int b = c->t0_next;
    if (!c->h_T0[b]) {
        HIP_TRY(c, hipHostMalloc((void**)&c->h_T0[b], (size_t)c->cfg.max_batch * 12 * sizeof(double), 0));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_T0[b], hipEventDisableTiming));
    } else {
        HIP_TRY(c, hipEventSynchronize(c->ev_T0[b]));
    }
    std::memcpy(c->h_T0[b], T0, pose_bytes);
    HIP_TRY(c, hipMemcpyAsync(c->poses.p, c->h_T0[b], pose_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev_T0[b], c->stream));
    c->t0_next = b ^ 1;
    return VELO_OK;
}

// A registration leaves its last iteration's hints and certificates in the arrays (the next one
// ignores them).  velo_linearize in hinted mode (velo_linearize_hints, a test hook) would read them
// against a different "previous pose": in that mode they are dropped instead.
int forget_hints_for_linearize(velo_ctx* c, int32_t* hint, size_t n_all, hipStream_t s)
{
    if (c->lin_hints && hint) {
        HIP_TRY(c, hipMemsetAsync(hint, 0xFF, n_all * sizeof(int32_t), s));
        // ... and the certificates with them: rho < 0 at hint < 0 certifies "no match" for an unmoved query (the split
        // first iteration writes those), whatever d_max the next call passes -- it covered this registration's only
        if (c->rho.p && c->rho.cap >= n_all) HIP_TRY(c, hipMemsetAsync(c->rho.p, 0, n_all * sizeof(float), s));
    }
    return VELO_OK;
}

// One iteration's linearise launch(es).  Split (round 5): the searching iterations -- unhinted, or hinted from a pose
// still far off -- spend most of their time on the stragglers of stage A, a fifth of the queries on a dense map, each
// searched inside the wavefront it happens to sit in.  Cut in three launches the stragglers of the WHOLE launch are
// searched together (launch_search_split) and the sums are made by the ordinary kernel from certified hints.
// Same correspondences bit for bit (the search functions are the same); the sums are canonical.
static hipError_t launch_iteration(velo_ctx* c, int it, bool split, const FrameView& fv, float dmax2, int32_t* hint, float* rho,
                                   bool stats, bool sorted, hipStream_t s)
{
    const Decomposition dc = decomposition_for(c, split ? std::max(it, 1) : it, hint != nullptr, sorted);
    const double* prev = it == 0 ? nullptr : c->poses_prev.p;  // (first iteration: stale hints)
    if (split) {
        hipError_t e = launch_search_split(dc.items, dc.n, fv, c->mv_read, c->poses.p, dmax2, hint, rho, prev,
                                           c->plan_lat ? 2 : 1, dc.lat_lanes, c->sq.p, c->sq_count.p + it,
                                           c->split_per_wave_max, c->wave_slots > 0 ? std::max(256, c->wave_slots / 4) : 2048, s);
        if (e != hipSuccess) return e;
        prev = c->poses.p;   // the third launch: every query holds its certified result at THIS pose
    }
    static const bool split_debug = getenv("VELO_SPLIT_DEBUG") != nullptr;   // (count the third launch's searches: velo_search_stats)
    return launch_linearize(c->cfg.linearize_variant, dc.items, dc.n, fv, c->mv_read, c->poses.p, dmax2, c->partials.p, nullptr,
                            nullptr, hint, rho, prev, stats || (split && split_debug), c->plan_lat ? 2 : 1, s, dc.lat_lanes);
}

int run_icp(velo_ctx* c, const double* T0, int iters, float d_max)
{
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map: call velo_map_reset first");
    if (c->n_frames < 1) return c->fail(VELO_E_INVALID, "no resident frames: call velo_frames_upload");
    if (!T0) return c->fail(VELO_E_INVALID, "T0 is null");
    if (iters < 1 || iters > VELO_MAX_ITERS)
        return c->fail(VELO_E_INVALID, "iters must be in [1,%d]", VELO_MAX_ITERS);
    if (!(d_max > 0.0f) || !(d_max <= c->mv_read.h))
        return c->fail(VELO_E_RANGE, "d_max must be in (0, voxel=%g]", (double)c->mv_read.h);
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_normals(c)) return rc;
    hipStream_t s = c->stream;
    c->ev_used = 0;
    c->ev_kind.clear();
    if (c->timing) {
        if (!c->ev_call0) {
            HIP_TRY(c, hipEventCreate(&c->ev_call0));
            HIP_TRY(c, hipEventCreate(&c->ev_call1));
        }
        HIP_TRY(c, hipEventRecord(c->ev_call0, s));
    }
    const size_t n_all = (size_t)c->frame_start[c->n_frames];
    const float dmax2 = d_max * d_max;
    const int ni = (int)c->items_h.size();
    const size_t pose_bytes = (size_t)c->n_frames * 12 * sizeof(double);
    int32_t* hint = nullptr;
    float* rho = nullptr;
    if (c->cfg.use_hints && n_all) {
        HIP_TRY(c, c->hint.reserve(n_all));
        hint = c->hint.p;
        if (c->cfg.use_hints >= 2) {
            HIP_TRY(c, c->rho.reserve(n_all));
            rho = c->rho.p;
        }
    }
    // pair certificates: the latency kernels' (FrameView::hint2 / rho3; the throughput kernel never looks at them)
    int32_t* hint2 = nullptr;
    float* rho3 = nullptr;
    if (hint && rho && c->plan_lat && c->cfg.linearize_variant == VELO_VARIANT_BALL && c->pair_certs) {
        HIP_TRY(c, c->hint2.reserve(n_all));
        HIP_TRY(c, c->rho3.reserve(n_all));
        hint2 = c->hint2.p;
        rho3 = c->rho3.p;
    }
    HIP_TRY(c, c->poses_prev.reserve((size_t)c->cfg.max_batch * 12));
    // the split iterations need hints AND certificates (the third launch lives on them), the pruned search, plain
    // query order; the counting instantiation keeps the one-launch form (its byte model is per launch)
    // ... and it pays on the LATENCY path only (a frame or two: a wavefront's stragglers wait for each other while the
    // chip idles -- first launch of a single frame 70 -> 40 us); a batch fills the chip either way, and its stragglers
    // packed 64 to a wavefront walk for as long as the slowest of them (headline 339 -> 630 us, dense 363 -> 320:
    // profiles/r05/split_iteration_ab.txt).  VELO_SPLIT_BATCH=1 forces it there for measurements.
    const bool split_batch = c->split_batch;
    const int n_split = (hint && rho && c->cfg.linearize_variant == VELO_VARIANT_BALL && !c->stats_on && !c->cfg.sort_frames && n_all &&
                         (c->plan_lat || split_batch))
                            ? std::min(c->split_iters, iters) : 0;
    if (n_split > 0) {
        HIP_TRY(c, reserve_slack(c->sq, n_all));
        HIP_TRY(c, c->sq_count.reserve(VELO_MAX_ITERS));
    }
    if (!c->pairs_total.p) {
        HIP_TRY(c, c->pairs_total.reserve(1));
        HIP_TRY(c, hipMemsetAsync(c->pairs_total.p, 0, sizeof(unsigned long long), s));
    }
    // A graph is only worth capturing for a launch sequence that will be replayed: when the map
    // or the frames changed since the previous registration (a stream: every frame appends to the
    // map) the sequence is launched directly, and captured only once the same map and frames come
    // back (capture + instantiate cost more than the launches they would save once).
    const bool stable = c->seen_map_gen == c->map_gen && c->seen_frames_gen == c->frames_gen;
    c->seen_map_gen = c->map_gen;
    c->seen_frames_gen = c->frames_gen;
    const bool graph_ok = c->cfg.use_graph && !c->timing && !c->stats_on && !c->cfg.sort_frames && stable;
    if (graph_ok) {
        // Replay the whole registration (hint reset, iters x (linearise, solve)) as
        // one hipGraph: the kernels are tens of microseconds long, so per-launch host cost and
        // inter-kernel gaps are a visible share of an iteration.
        velo_ctx::GraphKey key;
        key.iters = iters;
        key.ni = ni;
        key.n_frames = c->n_frames;
        key.variant = c->cfg.linearize_variant;
        key.dmax2 = dmax2;
        key.hint = hint;
        key.rho = rho;
        key.items = c->items_first.p;
        key.stream = s;
        key.map_gen = c->map_gen;
        key.frames_gen = c->frames_gen;
        key.n_split = n_split;
        key.sq = c->sq.p;
        key.hint2 = hint2;
        if (!c->graph_exec || !(key == c->graph_key)) {
            if (c->graph_exec) {
                (void)hipGraphExecDestroy(c->graph_exec);
                c->graph_exec = nullptr;
            }
            HIP_TRY(c, hipStreamSynchronize(s));
            HIP_TRY(c, hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            hipError_t e = hipSuccess;  // (the pose upload stays outside the graph: its source alternates)
            FrameView fv{c->ax, c->ay, c->az, nullptr, hint2, rho3};
            if (n_split > 0) e = hipMemsetAsync(c->sq_count.p, 0, VELO_MAX_ITERS * sizeof(unsigned), s);
            for (int it = 0; it < iters && e == hipSuccess; ++it) {
                const bool split = it < n_split;
                const Decomposition dc = decomposition_for(c, split ? std::max(it, 1) : it, hint != nullptr, false);
                e = launch_iteration(c, it, split, fv, dmax2, hint, rho, false, false, s);
                if (e == hipSuccess)
                    e = launch_reduce_solve(c->partials.p, dc.fbs, dc.lay, c->n_frames, c->poses.p,
                                            c->stats.p, it, c->cfg.solve_threads, nullptr, 1, c->poses_prev.p,
                                            c->pairs_total.p, s, (int)(c->partials.cap / kAccStride), dc.lay0,
                                            dc.lay0->nbig != 0 || c->n_frames > 1);
            }
            hipGraph_t g = nullptr;
            hipError_t e2 = hipStreamEndCapture(s, &g);
            if (e != hipSuccess || e2 != hipSuccess || !g) {
                if (g) (void)hipGraphDestroy(g);
                return c->fail(VELO_E_DEVICE, "graph capture failed: %s",
                               hipGetErrorString(e != hipSuccess ? e : e2));
            }
            e = hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) {
                c->graph_exec = nullptr;
                return c->fail(VELO_E_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(e));
            }
            c->graph_key = key;
        }
        if (int rc = upload_T0(c, T0, pose_bytes)) return rc;
        HIP_TRY(c, hipGraphLaunch(c->graph_exec, s));
        c->last_iters = iters;
        return forget_hints_for_linearize(c, hint, n_all, s);
    }
    if (int rc = upload_T0(c, T0, pose_bytes)) return rc;
    FrameView fv{c->ax, c->ay, c->az, nullptr, hint2, rho3};
    if (int rc = maybe_sort_frames(c, fv)) return rc;
    // hints never outlive a registration: results do not depend on earlier calls
    if (n_split > 0) HIP_TRY(c, hipMemsetAsync(c->sq_count.p, 0, VELO_MAX_ITERS * sizeof(unsigned), s));
    for (int it = 0; it < iters; ++it) {
        const bool split = it < n_split && fv.order == nullptr;
        const Decomposition dc = decomposition_for(c, split ? std::max(it, 1) : it, hint != nullptr, fv.order != nullptr);
        {
            Timed t(c, 0);   // (a split iteration's three launches are ONE linearise step of the timing record)
            HIP_TRY(c, launch_iteration(c, it, split, fv, dmax2, hint, rho, c->stats_on, fv.order != nullptr, s));
        }
        {
            Timed t(c, 1);
            HIP_TRY(c, launch_reduce_solve(c->partials.p, dc.fbs, dc.lay, c->n_frames, c->poses.p,
                                           c->stats.p, it, c->cfg.solve_threads, nullptr, 1, c->poses_prev.p,
                                           c->pairs_total.p, s, (int)(c->partials.cap / kAccStride), dc.lay0,
                                           dc.lay0->nbig != 0 || c->n_frames > 1));
        }
    }
    if (c->timing) HIP_TRY(c, hipEventRecord(c->ev_call1, s));
    c->last_iters = iters;
    return forget_hints_for_linearize(c, hint, n_all, s);
}

// the poses and per-iteration statistics of the registration just enqueued, on their way to pinned
// host memory (a copy into pageable memory is staged and waited for inside the call, twice per
// registration -- a visible share of a single frame's 0.4 ms)
int enqueue_result_copies(velo_ctx* c, int F)
{
    const size_t t_bytes = (size_t)c->cfg.max_batch * 12 * sizeof(double);
    const size_t s_bytes = (size_t)c->cfg.max_batch * VELO_MAX_ITERS * sizeof(velo_icp_iter);
    if (!c->h_result) HIP_TRY(c, hipHostMalloc((void**)&c->h_result, t_bytes + s_bytes, 0));
    HIP_TRY(c, hipMemcpyAsync(c->h_result, c->poses.p, (size_t)F * 12 * sizeof(double), hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_result + t_bytes, c->stats.p, (size_t)F * VELO_MAX_ITERS * sizeof(velo_icp_iter),
                              hipMemcpyDeviceToHost, c->stream));
    return VELO_OK;
}

void parse_results(velo_ctx* c, int F, int iters, velo_icp_result* out)
{
    const size_t t_bytes = (size_t)c->cfg.max_batch * 12 * sizeof(double);
    const double* T = reinterpret_cast<const double*>(c->h_result);
    const velo_icp_iter* st = reinterpret_cast<const velo_icp_iter*>(c->h_result + t_bytes);
    for (int f = 0; f < F; ++f) {
        velo_icp_result& r = out[f];
        std::memset(&r, 0, sizeof r);
        std::memcpy(r.T, &T[(size_t)f * 12], sizeof r.T);
        velo_pose_from_matrix(r.T, r.TRdeg);
        r.iters = iters;
        for (int i = 0; i < iters; ++i) {
            r.iter[i] = st[(size_t)f * VELO_MAX_ITERS + i];
            r.total_pairs += r.iter[i].n_pairs;
        }
    }
}

int fetch_icp(velo_ctx* c, velo_icp_result* out)
{
    if (!out) return c->fail(VELO_E_INVALID, "out is null");
    if (c->last_iters < 1) return c->fail(VELO_E_INVALID, "no registration has run");
    HIP_TRY(c, hipSetDevice(c->device));
    const int F = c->n_frames, iters = c->last_iters;
    if (int rc = enqueue_result_copies(c, F)) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    parse_results(c, F, iters, out);
    if (c->timing) {
        double lin = 0, sol = 0, lin_first = 0, lin_min = 1e30;
        int nl = 0, ns = 0;
        c->last_lin_us.clear();
        for (size_t k = 0; k < c->ev_kind.size(); ++k) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, c->ev[2 * k], c->ev[2 * k + 1]) != hipSuccess) continue;
            if (c->ev_kind[k] == 0) {
                c->last_lin_us.push_back(1e3f * ms);
                lin += ms;
                if (nl == 0) lin_first = ms;
                if (ms < lin_min) lin_min = ms;
                ++nl;
            } else {
                sol += ms;
                ++ns;
            }
        }
        float all = 0;
        (void)hipEventElapsedTime(&all, c->ev_call0, c->ev_call1);
        c->last_timing[0] = lin;
        c->last_timing[1] = nl;
        c->last_timing[2] = sol;
        c->last_timing[3] = ns;
        c->last_timing[4] = all;
        c->last_timing[5] = lin_first;
        c->last_timing[6] = nl ? lin_min : 0.0;
    }
    return VELO_OK;
}

}  // namespace

extern "C" {

int velo_abi_version(void) { return VELO_ABI_VERSION; }

const char* velo_last_error(const velo_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

velo_ctx* velo_create(int device_id, const velo_cfg* cfg)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no HIP device available: ") +
                         (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                         " (libveloslam_amd has no CPU fallback)";
        return nullptr;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_create_error = "device_id out of range";
        return nullptr;
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return nullptr;
    }
    std::unique_ptr<velo_ctx> c(new velo_ctx);
    c->device = device_id;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
            c->wave_slots = cus * 4 * 7;
    }
    std::memset(&c->cfg, 0, sizeof c->cfg);
    if (cfg) {
        // the meaning of fields (and of their zero values) changes with VELO_ABI_VERSION: a cfg
        // filled in against another header is refused, not reinterpreted
        const bool has_abi = cfg->struct_size >= offsetof(velo_cfg, abi_version) + sizeof(uint32_t);
        if (!has_abi || cfg->abi_version != VELO_ABI_VERSION) {
            char buf[160];
            snprintf(buf, sizeof buf, "velo_cfg.abi_version is %u, this library implements VELO_ABI_VERSION %d "
                     "(set cfg.struct_size and cfg.abi_version from the header you compile against)",
                     has_abi ? cfg->abi_version : 0u, VELO_ABI_VERSION);
            g_create_error = buf;
            return nullptr;
        }
        std::memcpy(&c->cfg, cfg, std::min<size_t>(cfg->struct_size, sizeof(velo_cfg)));
    }
    c->cfg.abi_version = VELO_ABI_VERSION;
    c->cfg.struct_size = sizeof(velo_cfg);
    if (c->cfg.max_batch <= 0) c->cfg.max_batch = 64;
    if (c->cfg.plan_wave_slots > 0) c->wave_slots = c->cfg.plan_wave_slots;
    // the tuning knobs: velo_cfg (ABI 3); a ZERO field may be overridden by the old environment variable (A/B scripts)
    if (c->cfg.split_iterations > 0) c->split_iters = std::min(c->cfg.split_iterations, VELO_MAX_ITERS);
    else if (c->cfg.split_iterations < 0) c->split_iters = 0;
    else if (const char* e = getenv("VELO_SPLIT_ITERS")) c->split_iters = std::max(0, std::min(atoi(e), VELO_MAX_ITERS));
    c->split_batch = c->cfg.split_batches != 0 || getenv("VELO_SPLIT_BATCH") != nullptr;
    if (c->cfg.split_per_wave_max > 0) c->split_per_wave_max = (unsigned)c->cfg.split_per_wave_max;
    else if (c->cfg.split_per_wave_max < 0) c->split_per_wave_max = 0;
    else if (const char* e = getenv("VELO_SPLIT_PER_WAVE_MAX")) c->split_per_wave_max = (unsigned)std::max(0, atoi(e));
    if (c->cfg.solve_threads != 0 && c->cfg.solve_threads != 256 && c->cfg.solve_threads != 512 && c->cfg.solve_threads != 1024) {
        g_create_error = "velo_cfg.solve_threads must be 0, 256, 512 or 1024";
        return nullptr;
    }
    c->pair_certs = c->cfg.pair_certificates >= 0 && !(c->cfg.pair_certificates == 0 && getenv("VELO_NO_PAIR_CERT"));
    // 0 = the default (fast, pruned) kernel for every consumer -- C, C++ MapManager and Python
    // alike; the exhaustive validation kernel has to be asked for by name
    if (c->cfg.linearize_variant == 0) c->cfg.linearize_variant = VELO_VARIANT_BALL;
    {
        const int v = c->cfg.linearize_variant;
        bool known = v == VELO_VARIANT_BALL || v == VELO_VARIANT_SCAN;
#ifdef VELO_ABLATIONS  // private timing builds (tools/build_variant.sh -DVELO_ABLATIONS): wrong results by design
        known = known || (v >= 11 && v <= 13);
#endif
        if (!known) {
            g_create_error = "velo_cfg.linearize_variant: unknown value (0 / VELO_VARIANT_BALL / VELO_VARIANT_SCAN)";
            return nullptr;
        }
    }
    if (!cfg) {
        c->cfg.use_hints = 2;
        c->cfg.use_graph = 1;
    }
    if (c->cfg.map_margin < 0) c->cfg.map_margin = 0;
    for (int a = 0; a < 3; ++a) c->margin[a] = c->cfg.map_margin;
    if (c->cfg.map_subdiv < 0) c->cfg.map_subdiv = 0;  // 0 = chosen from the map's density
    if (c->cfg.map_subdiv > 16) c->cfg.map_subdiv = 16;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        return nullptr;
    }
    c->stream = c->own_stream;
    c->dopts.struct_size = sizeof c->dopts;
    std::memset(c->dopts.laser_selection, 1, sizeof c->dopts.laser_selection);
    return c.release();
}

void velo_destroy(velo_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);  // (pinned buffers below may still be its sources)
    if (c->roll_stream) (void)hipStreamSynchronize(c->roll_stream);
    if (c->roll_stream_light) (void)hipStreamSynchronize(c->roll_stream_light);
    if (c->ev_roll) (void)hipEventDestroy(c->ev_roll);
    if (c->ev_roll_geom) (void)hipEventDestroy(c->ev_roll_geom);
    if (c->roll_stream) (void)hipStreamDestroy(c->roll_stream);
    if (c->roll_stream_light) (void)hipStreamDestroy(c->roll_stream_light);
    if (c->h_roll) (void)hipHostFree(c->h_roll);
    if (c->h_enter) (void)hipHostFree(c->h_enter);
    (void)velo_comm_destroy(c);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    for (int b = 0; b < 2; ++b) {
        if (c->h_T0[b]) (void)hipHostFree(c->h_T0[b]);
        if (c->ev_T0[b]) (void)hipEventDestroy(c->ev_T0[b]);
    }
    if (c->h_inc_total) (void)hipHostFree(c->h_inc_total);
    if (c->h_result) (void)hipHostFree(c->h_result);
    if (c->h_starts) (void)hipHostFree(c->h_starts);
    if (c->h_pend_total) (void)hipHostFree(c->h_pend_total);
    if (c->h_pend_stage) (void)hipHostFree(c->h_pend_stage);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->ev_pend) (void)hipEventDestroy(c->ev_pend);
    if (c->ev_inc) (void)hipEventDestroy(c->ev_inc);
    if (c->ev_plan) (void)hipEventDestroy(c->ev_plan);
    if (c->ev_res) (void)hipEventDestroy(c->ev_res);
    if (c->ev_mark) (void)hipEventDestroy(c->ev_mark);
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->plan_up.h) (void)hipHostFree(c->plan_up.h);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->ev_call0) (void)hipEventDestroy(c->ev_call0);
    if (c->ev_call1) (void)hipEventDestroy(c->ev_call1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int velo_cfg_get(const velo_ctx* c, velo_cfg* out)
{
    if (!c || !out) return VELO_E_INVALID;
    *out = c->cfg;
    return VELO_OK;
}

int velo_set_stream(velo_ctx* c, void* hip_stream)
{
    if (!c) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return VELO_OK;
}

int velo_synchronize(velo_ctx* c)
{
    if (!c) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->roll_stream) HIP_TRY(c, hipStreamSynchronize(c->roll_stream));  // (a roll begun ahead runs on its own stream)
    if (c->roll_stream_light) HIP_TRY(c, hipStreamSynchronize(c->roll_stream_light));
    return VELO_OK;
}

int velo_linearize_hints(velo_ctx* c, int mode)
{
    if (!c) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    c->lin_hints = mode != 0;
    if (c->hint.p && c->hint.cap)  // (re)start from "no hint"
        HIP_TRY(c, hipMemsetAsync(c->hint.p, 0xFF, c->hint.cap * sizeof(int32_t), c->stream));
    if (c->rho.p && c->rho.cap)
        HIP_TRY(c, hipMemsetAsync(c->rho.p, 0, c->rho.cap * sizeof(float), c->stream));
    return VELO_OK;
}

int velo_last_linearize_us(velo_ctx* c, float* out, int cap)
{
    if (!c || (cap > 0 && !out)) return VELO_E_INVALID;
    const int n = (int)c->last_lin_us.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = c->last_lin_us[(size_t)i];
    return n;
}

int velo_pairs_total(velo_ctx* c, uint64_t* out, int reset)
{
    if (!c || !out) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long v = 0;
    if (c->pairs_total.p) {
        HIP_TRY(c, hipMemcpyAsync(&v, c->pairs_total.p, sizeof v, hipMemcpyDeviceToHost, c->stream));
        if (reset) HIP_TRY(c, hipMemsetAsync(c->pairs_total.p, 0, sizeof v, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    *out = v;
    return VELO_OK;
}

int velo_set_stats(velo_ctx* c, int on)
{
    if (!c) return VELO_E_INVALID;
    c->stats_on = on != 0;
    return VELO_OK;
}

int velo_search_stats(velo_ctx* c, uint64_t out[16], int reset)
{
    if (!c || !out) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long v[16];
    HIP_TRY(c, read_lin_stats(v, reset != 0, c->stream));
    for (int i = 0; i < 16; ++i) out[i] = v[i];
    return VELO_OK;
}

int velo_debug_search_stats(velo_ctx* c, uint64_t out[8], int reset)
{
    uint64_t v[16];
    if (!out) return VELO_E_INVALID;
    if (int rc = velo_search_stats(c, v, reset)) return rc;
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return VELO_OK;
}

int velo_set_timing(velo_ctx* c, int on)
{
    if (!c) return VELO_E_INVALID;
    c->timing = on != 0;
    return VELO_OK;
}

int velo_last_timing(velo_ctx* c, double out[8])
{
    if (!c || !out) return VELO_E_INVALID;
    std::memcpy(out, c->last_timing, sizeof c->last_timing);
    return VELO_OK;
}

// ------------------------------------------------------------------------ map
static int map_reset_impl(velo_ctx* c, const float* x, const float* y, const float* z, size_t n,
                          float voxel, int k, bool dev)
{
    if (!c) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_roll(c)) return rc;
    c->has_map = false;
    if (n == 0) return c->fail(VELO_E_INVALID, "map needs at least one point");
    if (int rc = stage_raw(c, x, y, z, n, dev, false)) return rc;
    if (int rc = resolve_subdiv(c, voxel)) return rc;
    return rebuild_map(c, voxel, k);
}
int velo_map_reset(velo_ctx* c, const float* x, const float* y, const float* z, size_t n,
                   float voxel, int k)
{
    return map_reset_impl(c, x, y, z, n, voxel, k, false);
}
int velo_map_reset_dev(velo_ctx* c, const float* x, const float* y, const float* z, size_t n,
                       float voxel, int k)
{
    return map_reset_impl(c, x, y, z, n, voxel, k, true);
}
static int map_append_impl(velo_ctx* c, const float* x, const float* y, const float* z, size_t n,
                           bool dev)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "velo_map_append before velo_map_reset");
    if (n == 0) return VELO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->overlap_update)
        if (int rc = settle_roll(c)) return rc;
    const size_t n_old = c->raw_n;
    if (int rc = stage_raw(c, x, y, z, n, dev, true)) return rc;
    int done = 0, grown[3] = {0, 0, 0};
    if (int rc = append_incremental(c, n_old, n, &done, grown)) {
        c->raw_n = n_old;  // refused: the map is unchanged
        return rc;
    }
    if (done) return VELO_OK;
    const int k = c->info.k_normals;
    // beside a registration the rebuild needs the carried normals (it then writes the other copies of everything the
    // registration reads); without them, with the A/B switch or a hashed table it is refused, nothing changed
    if (c->overlap_update && (k <= 0 || c->cfg.map_full_rebuild)) {
        c->raw_n = n_old;
        return VELO_E_AGAIN;
    }
    // re-anchor: new origin, full re-sort; the normals of the old points travel with them
    if (k > 0 && !c->cfg.map_full_rebuild) {
        hipStream_t s = c->stream;
        HIP_TRY(c, reserve_slack(c->nrm_raw, n_old + n));
        HIP_TRY(c, launch_scatter_nrm_raw(c->nrm.p, c->perm.p, (uint32_t)n_old, nullptr, nullptr,
                                          c->nrm_raw.p, s));
        HIP_TRY(c, launch_fill_fresh(c->nrm_raw.p + n_old, (uint32_t)n, s));
        CarryNormals cr;
        int rc;
        if (grown[0] > 0) {  // (beside a registration only: the grid keeps its origin and grows, see append_incremental)
            const float org[3] = {c->mv.ox, c->mv.oy, c->mv.oz};
            rc = rebuild_map(c, c->info.voxel, k, org, grown, &cr);
        } else {
            rc = rebuild_map(c, c->info.voxel, k, nullptr, nullptr, &cr);
        }
        if (rc) c->raw_n = n_old;
        return rc;
    }
    int rc = rebuild_map(c, c->info.voxel, k);
    if (rc) c->raw_n = n_old;
    return rc;
}
int velo_map_append(velo_ctx* c, const float* x, const float* y, const float* z, size_t n)
{
    return map_append_impl(c, x, y, z, n, false);
}

// voxel-downsampled insertion (oracle/icp.c vo_roll_filter_sparse + vo_roll_append)
static int map_append_sparse_impl(velo_ctx* c, const float* x, const float* y, const float* z, size_t n,
                                  int min_count, size_t* n_accepted, bool dev)
{
    if (!c) return VELO_E_INVALID;
    if (n_accepted) *n_accepted = 0;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "velo_map_append_sparse before velo_map_reset");
    if (n == 0) return VELO_OK;
    if (!x || !y || !z) return c->fail(VELO_E_INVALID, "null point array");
    if (min_count < 1) return c->fail(VELO_E_INVALID, "min_count must be >= 1");
    if (n >= (size_t)INT32_MAX) return c->fail(VELO_E_RANGE, "too many points");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_roll(c)) return rc;
    hipStream_t s = c->stream;
    HIP_TRY(c, c->sp_x.reserve(2 * n));  // [0,n): staged input (host entry), [n,2n): survivors
    HIP_TRY(c, c->sp_y.reserve(2 * n));
    HIP_TRY(c, c->sp_z.reserve(2 * n));
    const float *dx = x, *dy = y, *dz = z;
    if (!dev) {
        HIP_TRY(c, hipMemcpyAsync(c->sp_x.p, x, n * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->sp_y.p, y, n * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->sp_z.p, z, n * sizeof(float), hipMemcpyHostToDevice, s));
        dx = c->sp_x.p;
        dy = c->sp_y.p;
        dz = c->sp_z.p;
    }
    HIP_TRY(c, c->sp_keys.reserve(n));
    HIP_TRY(c, c->sp_keys2.reserve(n));
    HIP_TRY(c, c->sp_idx.reserve(n));
    HIP_TRY(c, c->sp_idx2.reserve(n));
    HIP_TRY(c, c->flags.reserve(n + 1));
    HIP_TRY(c, c->offs.reserve(n + 1));
    HIP_TRY(c, launch_sparse_keys(dx, dy, dz, n, c->mv, c->sp_keys.p, c->sp_idx.p, s));
    size_t tb = 0, tb2 = 0;
    HIP_TRY(c, sort_pairs64(nullptr, tb, c->sp_keys.p, c->sp_keys2.p, c->sp_idx.p, c->sp_idx2.p, n, s));
    HIP_TRY(c, exclusive_scan_u32(nullptr, tb2, c->flags.p, c->offs.p, n + 1, s));
    if (int rc = ensure_temp(c, std::max(tb, tb2))) return rc;
    HIP_TRY(c, sort_pairs64(c->temp.p, tb, c->sp_keys.p, c->sp_keys2.p, c->sp_idx.p, c->sp_idx2.p, n, s));
    HIP_TRY(c, launch_sparse_accept(c->sp_keys2.p, c->sp_idx2.p, n, c->mv, min_count, c->flags.p, s));
    HIP_TRY(c, hipMemsetAsync(c->flags.p + n, 0, sizeof(uint32_t), s));
    HIP_TRY(c, exclusive_scan_u32(c->temp.p, tb2, c->flags.p, c->offs.p, n + 1, s));
    uint32_t kept = 0;
    HIP_TRY(c, hipMemcpyAsync(&kept, c->offs.p + n, sizeof kept, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, launch_compact3(dx, dy, dz, n, c->flags.p, c->offs.p, c->sp_x.p + n, c->sp_y.p + n,
                               c->sp_z.p + n, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (n_accepted) *n_accepted = kept;
    if (kept == 0) return VELO_OK;
    return map_append_impl(c, c->sp_x.p + n, c->sp_y.p + n, c->sp_z.p + n, kept, true);
}
int velo_map_append_sparse(velo_ctx* c, const float* x, const float* y, const float* z, size_t n,
                           int min_count, size_t* n_accepted)
{
    return map_append_sparse_impl(c, x, y, z, n, min_count, n_accepted, false);
}
int velo_map_append_sparse_dev(velo_ctx* c, const float* dx, const float* dy, const float* dz, size_t n,
                               int min_count, size_t* n_accepted)
{
    return map_append_sparse_impl(c, dx, dy, dz, n, min_count, n_accepted, true);
}
int velo_map_append_dev(velo_ctx* c, const float* x, const float* y, const float* z, size_t n)
{
    return map_append_impl(c, x, y, z, n, true);
}

static int evict_impl(velo_ctx* c, const KeepRegion& region);

int velo_map_evict_outside(velo_ctx* c, const float lo[3], const float hi[3])
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "velo_map_evict_outside before velo_map_reset");
    if (int rc = settle_roll(c)) return rc;
    if (!lo || !hi) return c->fail(VELO_E_INVALID, "null box");
    KeepRegion g{};
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = lo[a];
        g.hi[a] = hi[a];
    }
    return evict_impl(c, g);
}

int velo_map_evict_radius(velo_ctx* c, const float center_xy[2], float radius)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "velo_map_evict_radius before velo_map_reset");
    if (int rc = settle_roll(c)) return rc;
    if (!center_xy || !(radius >= 0.0f)) return c->fail(VELO_E_INVALID, "null centre or negative radius");
    KeepRegion g{};
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = -3.0e38f;
        g.hi[a] = 3.0e38f;
    }
    g.cx = center_xy[0];
    g.cy = center_xy[1];
    g.r2 = radius * radius;
    g.use_radius = 1;
    return evict_impl(c, g);
}

static int evict_impl(velo_ctx* c, const KeepRegion& region)
{
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const MapView old = c->mv;
    const uint32_t n = (uint32_t)c->raw_n;
    const int k = c->info.k_normals;
    // keep flags + exclusive scans, in sorted order and in append order
    HIP_TRY(c, reserve_slack(c->flags, n));
    HIP_TRY(c, reserve_slack(c->offs, n));
    HIP_TRY(c, reserve_slack(c->rflags, n));
    HIP_TRY(c, reserve_slack(c->roffs, n));
    HIP_TRY(c, launch_keep_flags(c->pts.p, nullptr, nullptr, nullptr, n, region, c->flags.p, s));
    HIP_TRY(c, c->mm_scratch.reserve(8));
    HIP_TRY(c, launch_keep_flags_minmax(c->raw_x.p, c->raw_y.p, c->raw_z.p, n, region, c->rflags.p, c->mm_scratch.p, s));
    size_t tb = 0;
    HIP_TRY(c, exclusive_scan_u32(nullptr, tb, c->flags.p, c->offs.p, n, s));
    if (int rc = ensure_temp(c, tb)) return rc;
    HIP_TRY(c, exclusive_scan_u32(c->temp.p, tb, c->flags.p, c->offs.p, n, s));
    HIP_TRY(c, exclusive_scan_u32(c->temp.p, tb, c->rflags.p, c->roffs.p, n, s));
    uint32_t last[2] = {0, 0};
    unsigned mm6[6];
    HIP_TRY(c, hipMemcpyAsync(&last[0], c->offs.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&last[1], c->flags.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(mm6, c->mm_scratch.p, sizeof mm6, hipMemcpyDeviceToHost, s));
    static const bool trace_roll = getenv("VELO_TRACE_ROLL") != nullptr;
    const auto te0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipStreamSynchronize(s));
    if (trace_roll)
        std::fprintf(stderr, "evict: waited %.0f us for the keep counts\n",
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - te0).count());
    const uint32_t kept = last[0] + last[1];
    if (kept == 0) return c->fail(VELO_E_INVALID, "eviction region would remove every map point");
    if (kept == n) return VELO_OK;
    // append-order arrays first: they decide whether the grid keeps
    // (a roll: room for the points its append will add, so that staging them does not reallocate -- and wait)
    HIP_TRY(c, reserve_slack(c->raw_x2, kept + c->roll_extra));
    HIP_TRY(c, reserve_slack(c->raw_y2, kept + c->roll_extra));
    HIP_TRY(c, reserve_slack(c->raw_z2, kept + c->roll_extra));
    HIP_TRY(c, launch_compact_raw(c->raw_x.p, c->raw_y.p, c->raw_z.p, n, c->rflags.p, c->roffs.p,
                                  c->raw_x2.p, c->raw_y2.p, c->raw_z2.p, s));
    MinMax mm;  // of the kept points: came back with the counts above
    auto dec = [](unsigned e) {  // (device_math.hpp enc_f32's inverse)
        const unsigned u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    for (int a = 0; a < 3; ++a) {
        mm.mn[a] = dec(mm6[a]);
        mm.mx[a] = dec(mm6[3 + a]);
    }
    const float org[3] = {old.ox, old.oy, old.oz};
    bool anchor = false;
    for (int a = 0; a < 3; ++a)
        if (floorf((mm.mn[a] - org[a]) * old.inv_h) >= (float)(2 * c->margin[a] + 2)) anchor = true;
    auto swap_raw = [&]() {
        std::swap(c->raw_x.p, c->raw_x2.p);
        std::swap(c->raw_x.cap, c->raw_x2.cap);
        std::swap(c->raw_y.p, c->raw_y2.p);
        std::swap(c->raw_y.cap, c->raw_y2.cap);
        std::swap(c->raw_z.p, c->raw_z2.p);
        std::swap(c->raw_z.cap, c->raw_z2.cap);
    };
    // beside a registration: the re-anchoring rebuild goes into the other copies of the arrays and of the table (round 5:
    // rebuild_map / build_table), like an incremental update; what is still refused -- before anything changed -- is a
    // rebuild in place (A/B switch), a hashed table, a map without normals (its rebuild rewrites pts / nrm in place)
    if (c->overlap_update && (c->cfg.map_full_rebuild || (anchor && k <= 0)))
        return VELO_E_AGAIN;
    if (anchor || c->cfg.map_full_rebuild) {
        CarryNormals cr;
        const bool carry = anchor && k > 0 && !c->cfg.map_full_rebuild;
        if (carry) {  // surviving normals into append order of the compacted list
            HIP_TRY(c, reserve_slack(c->nrm_raw, kept));
            HIP_TRY(c, launch_scatter_nrm_raw(c->nrm.p, c->perm.p, n, c->flags.p, c->roffs.p,
                                              c->nrm_raw.p, s));
            cr.removed_keep = c->flags.p;
            cr.removed_n = n;
        }
        swap_raw();
        c->raw_n = kept;
        const int dims[3] = {old.nx, old.ny, old.nz};
        int rc = anchor ? rebuild_map(c, old.h, k, nullptr, nullptr, carry ? &cr : nullptr)
                        : rebuild_map(c, old.h, k, org, dims);
        if (rc) {  // restore the old point list (if the rebuild had started rewriting the sorted
            swap_raw();  // arrays the ctx now has NO map, see MapTxn: velo_map_reset recovers)
            c->raw_n = n;
        }
        return rc;
    }
    overlap_rotate_alt(c);
    HIP_TRY(c, reserve_slack(c->pts_alt, kept));
    HIP_TRY(c, reserve_slack(c->nrm_alt, kept));
    HIP_TRY(c, reserve_slack(c->perm_alt, kept));
    HIP_TRY(c, reserve_slack(c->keys_alt, kept));
    const size_t nvox = (size_t)old.nx * old.ny * old.nz;
    if (k > 0) {  // voxels that lose a point, marked from the OLD arrays
        HIP_TRY(c, reserve_slack(c->dirty, nvox));
        HIP_TRY(c, hipMemsetAsync(c->dirty.p, 0, nvox, s));
        HIP_TRY(c, launch_mark_dirty(c->keys_sorted.p, n, c->flags.p, old, c->dirty.p, s));
        HIP_TRY(c, reserve_slack(c->nk_sorted, n - kept));  // sorted keys of the removed points
        HIP_TRY(c, launch_removed_keys(c->keys_sorted.p, c->flags.p, c->offs.p, n, c->nk_sorted.p, s));
    }
    MapTxn txn(c);
    txn.touched = true;  // the table is remapped in place below
    HIP_TRY(c, launch_compact_sorted(c->pts.p, c->nrm.p, c->perm.p, c->keys_sorted.p, n, c->flags.p,
                                     c->offs.p, c->roffs.p, c->pts_alt.p, c->nrm_alt.p,
                                     c->perm_alt.p, c->keys_alt.p, k > 0 ? c->invalid_cnt.p : nullptr, s));
    const size_t ncell = (size_t)old.fx * old.fy * old.fz;
    MapView g = old;
    if (c->use_hash) {
        if (int rc = build_table(c, g, c->keys_alt.p, kept, ncell)) return rc;
    } else if (c->overlap_update && c->overlap_done == 0) {
        HIP_TRY(c, reserve_slack(c->cell_start_alt, ncell + 8));
        HIP_TRY(c, reserve_slack(c->tile_bounds, table_tile_bounds(ncell + 1)));
        HIP_TRY(c, launch_table_remap(c->cell_start.p, c->cell_start_alt.p, ncell + 1, c->offs.p, n, kept, c->tile_bounds.p, s));
        std::swap(c->cell_start.p, c->cell_start_alt.p);
        std::swap(c->cell_start.cap, c->cell_start_alt.cap);
    } else {
        HIP_TRY(c, reserve_slack(c->tile_bounds, table_tile_bounds(ncell + 1)));
        HIP_TRY(c, launch_table_remap(c->cell_start.p, c->cell_start.p, ncell + 1, c->offs.p, n, kept, c->tile_bounds.p, s));
    }
    swap_raw();
    c->raw_n = kept;
    std::swap(c->pts.p, c->pts_alt.p);
    std::swap(c->pts.cap, c->pts_alt.cap);
    std::swap(c->nrm.p, c->nrm_alt.p);
    std::swap(c->nrm.cap, c->nrm_alt.cap);
    std::swap(c->perm.p, c->perm_alt.p);
    std::swap(c->perm.cap, c->perm_alt.cap);
    std::swap(c->keys_sorted.p, c->keys_alt.p);
    std::swap(c->keys_sorted.cap, c->keys_alt.cap);
    g.pts = c->pts.p;
    g.nrm = c->nrm.p;
    if (!c->use_hash) g.cell_start = c->cell_start.p;
    g.n = (int)kept;
    unsigned long long invalid = kept;
    c->n_done_host = 0;
    if (k > 0) {
        if (int rc = mark_roll_geometry(c)) return rc;
        if (int rc = refresh_dirty_normals(c, g, k, c->nk_sorted.p, n - kept)) return rc;
        // (the running count: compact_sorted took the leavers off, the re-estimation adjusted the rest)
        if (c->defer_counts)
            HIP_TRY(c, hipMemcpyAsync(&c->h_roll->invalid, c->invalid_cnt.p, sizeof invalid, hipMemcpyDeviceToHost, s));
        else
            HIP_TRY(c, hipMemcpyAsync(&invalid, c->invalid_cnt.p, sizeof invalid, hipMemcpyDeviceToHost, s));
    }
    if (!c->defer_counts) HIP_TRY(c, hipStreamSynchronize(s));
    for (int a = 0; a < 3; ++a) c->map_mx[a] = mm.mx[a];
    if (c->defer_counts) c->roll_counts_pending = k > 0;
    return txn.done(publish_map(c, g, k, invalid, 1, c->n_done_host));
}

// evict + append on the SIDE stream while the registration begun with velo_icp_batch_start runs on
// the main one.  The registration reads pts / nrm / the fine table / vox_near of the map as it was;
// the updates write other copies of all four (the sorted arrays have a third set for the second
// update of the call), scratch of their own (the increment got its own flags / offsets), and wait on
// the device for everything older than the registration.  What cannot be done that way -- a
// re-anchor, a grown or hashed table -- is refused with VELO_E_AGAIN before anything changed.
// what both forms of the roll check before anything changes, and the bounds of the entering points
// min / max of p[0..n) (n >= 1); false if a value is NaN or infinite
static bool bounds_finite(const float* p, size_t n, float& lo_a, float& hi_a)
{
    size_t i = 0;
    bool bad = false;
    lo_a = hi_a = p[0];
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
    {
        __m128 lo0 = _mm_set1_ps(p[0]), hi0 = lo0, lo1 = lo0, hi1 = lo0;
        const __m128i ex = _mm_set1_epi32(0x7f800000);
        __m128i b4 = _mm_setzero_si128();
        for (; i + 8 <= n; i += 8) {
            const __m128 a = _mm_loadu_ps(p + i), b = _mm_loadu_ps(p + i + 4);
            lo0 = _mm_min_ps(lo0, a), hi0 = _mm_max_ps(hi0, a);
            lo1 = _mm_min_ps(lo1, b), hi1 = _mm_max_ps(hi1, b);
            b4 = _mm_or_si128(b4, _mm_cmpeq_epi32(_mm_and_si128(_mm_castps_si128(a), ex), ex));  // exponent all ones
            b4 = _mm_or_si128(b4, _mm_cmpeq_epi32(_mm_and_si128(_mm_castps_si128(b), ex), ex));
        }
        float l4[4], h4[4];
        _mm_storeu_ps(l4, _mm_min_ps(lo0, lo1));
        _mm_storeu_ps(h4, _mm_max_ps(hi0, hi1));
        for (int k = 0; k < 4; ++k) {
            lo_a = l4[k] < lo_a ? l4[k] : lo_a;
            hi_a = h4[k] > hi_a ? h4[k] : hi_a;
        }
        bad = _mm_movemask_epi8(b4) != 0;
    }
#endif
    for (; i < n; ++i) {
        const float v = p[i];
        uint32_t u;
        std::memcpy(&u, &v, 4);
        bad |= (u & 0x7f800000u) == 0x7f800000u;
        lo_a = v < lo_a ? v : lo_a;
        hi_a = v > hi_a ? v : hi_a;
    }
    return !bad;
}

static int roll_precheck(velo_ctx* c, const char* who, const float lo[3], const float hi[3], const float* x,
                         const float* y, const float* z, size_t n, float mn[3], float mx[3], bool allow_idle = false)
{
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "%s before velo_map_reset", who);
    // velo_map_roll_overlapped: beside ONE registration, between its start and its finish.  velo_map_roll_begin: there,
    // or with no registration outstanding at all (round 6: the roll then starts before the next registration is even
    // enqueued; that registration, started while the roll is begun, reads the map as it was and counts as the roll's)
    if (!c->res_pending && !allow_idle)
        return c->fail(VELO_E_INVALID, "%s needs a registration started with velo_icp_batch_start and not yet finished", who);
    if (c->res_pending && c->roll_overlapped_done) return c->fail(VELO_E_INVALID, "one overlapped roll per registration");
    if ((lo == nullptr) != (hi == nullptr)) return c->fail(VELO_E_INVALID, "lo and hi go together");
    if (n && (!x || !y || !z)) return c->fail(VELO_E_INVALID, "null point array");
    // "Refused before anything changed" has to hold for the PAIR: the eviction publishes its map
    // before the append can find out that the entering points need a re-anchor or a larger table -- the
    // caller then saw n_points move, took the device map for dirty and rebuilt it from the host tiles.
    // Everything the append's refusal depends on is known here: the entering points are host arrays, and an
    // eviction that goes ahead beside a registration keeps the grid (origin, dims) -- one that would
    // re-anchor is refused by evict_impl itself before it changes anything.
    if (c->cfg.map_full_rebuild)
        return c->fail(VELO_E_AGAIN, "this update needs the map rebuilt: not beside a registration");
    if (n) {
        const float* src[3] = {x, y, z};
        for (int a = 0; a < 3; ++a) {
            // (on the host's critical path once per roll: 0.85 - 1.05 ms for the 547 k points of a tile column as a scalar
            //  loop -- neither clang nor gcc vectorises the min / max / finiteness pass whichever way it is written --
            //  0.25 ms with SSE2, profiles/r05/roll_begin_host.txt)
            float lo_a, hi_a;
            const bool bad = !bounds_finite(src[a], n, lo_a, hi_a);
            const float acc = bad ? 1.0f : 0.0f;
            if (!(acc == 0.0f) || !std::isfinite(lo_a) || !std::isfinite(hi_a))
                return c->fail(VELO_E_INVALID, "map points must be finite");
            mn[a] = lo_a;
            mx[a] = hi_a;
        }
        const float org[3] = {c->mv.ox, c->mv.oy, c->mv.oz};
        const int dims[3] = {c->mv.nx, c->mv.ny, c->mv.nz};
        for (int a = 0; a < 3; ++a) {
            // (the same float expressions append_incremental decides with.  With normals to carry a re-anchor or a
            //  grown grid is built into the other copies beside the registration; without them it is refused here)
            const float ext = floorf((mx[a] - org[a]) * c->mv.inv_h);
            if (!(ext < 2.0e9f)) return c->fail(VELO_E_RANGE, "map extent / voxel too large");
            if ((mn[a] < org[a] || (int)ext + 1 > dims[a]) && c->info.k_normals <= 0)
                return c->fail(VELO_E_AGAIN, "the entering points need the grid re-anchored or grown: not beside a registration");
        }
    }
    return VELO_OK;
}

// evict + append on stream `rs` with the ctx's update machinery pointed at it
static int roll_run(velo_ctx* c, hipStream_t rs, DevBuf<char>& scratch, const float lo[3], const float hi[3],
                    const float* x, const float* y, const float* z, size_t n)
{
    hipStream_t main_stream = c->stream;
    c->stream = rs;
    // (sort / scan scratch of its own: `temp` belongs to the main stream's work, and a roll on its own stream may run
    //  beside a decode on the side stream, whose scratch dk_temp is)
    std::swap(c->temp.p, scratch.p);
    std::swap(c->temp.cap, scratch.cap);
    c->overlap_update = true;
    c->overlap_done = 0;
    c->overlap_main = main_stream;
    int rc = VELO_OK;
    c->roll_geom_recorded = false;
    if (lo) {
        KeepRegion g{};
        for (int a = 0; a < 3; ++a) {
            g.lo[a] = lo[a];
            g.hi[a] = hi[a];
        }
        rc = evict_impl(c, g);
    }
    if (rc == VELO_OK && n) {
        c->roll_geom_recorded = false;   // (the last update's geometry is the rolled map's)
        rc = map_append_impl(c, x, y, z, n, false);
    }
    c->overlap_update = false;
    std::swap(c->temp.p, scratch.p);
    std::swap(c->temp.cap, scratch.cap);
    c->stream = main_stream;
    return rc;
}

int velo_map_roll_overlapped(velo_ctx* c, const float lo[3], const float hi[3], const float* x, const float* y,
                             const float* z, size_t n)
{
    if (!c) return VELO_E_INVALID;
    if (int rc = settle_roll(c)) return rc;
    float mn[3], mx[3];
    if (int rc = roll_precheck(c, "velo_map_roll_overlapped", lo, hi, x, y, z, n, mn, mx)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->side_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    if (!c->ev_side) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    HIP_TRY(c, hipStreamWaitEvent(c->side_stream, c->ev_mark, 0));
    c->roll_overlapped_done = true;
    const int rc = roll_run(c, c->side_stream, c->dk_temp, lo, hi, x, y, z, n);
    hipError_t e = hipEventRecord(c->ev_side, c->side_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_side, 0);
    if (rc == VELO_E_AGAIN) return c->fail(VELO_E_AGAIN, "this update needs the map rebuilt: not beside a registration");
    if (rc) return rc;
    if (e != hipSuccess) return c->fail(VELO_E_DEVICE, "side stream: %s", hipGetErrorString(e));
    return VELO_OK;
}

// The same roll BEGUN AHEAD of the frame that needs it.  The tile rectangle of a frame comes
// from the pose track, so it is known frames before; the roll is 2 ms of GPU work against 0.5 ms of registration,
// and velo_map_roll_overlapped holds the host until it is through (its stages hand counts over through the
// host) while the main stream idles behind the one registration it overlaps.  Here:
//   * one wait, for the first count (points kept, their bounds: what decides whether the grid keeps -- ~0.2 ms
//     of flags and scans); everything after it is enqueued on a stream of its own without waiting: buffer
//     sizes are host arithmetic from that count, the one length that stays on the device (the work list of the
//     normals) is read there, the counts the host only reports (invalid normals, normals re-estimated) land in
//     pinned memory;
//   * the registrations that follow keep reading the map as it was (mv_read) until velo_map_roll_publish --
//     called by the host at the frame the new rectangle is due -- makes the main stream wait for the roll's last
//     kernel (a device-side wait) and switches the readers over.  Any other map operation publishes first.
// The map that results is the plain evict + append, bit for bit (tests/test_gpu_parity.py).
int velo_map_roll_begin(velo_ctx* c, const float lo[3], const float hi[3], const float* x, const float* y,
                        const float* z, size_t n)
{
    if (!c) return VELO_E_INVALID;
    if (c->roll_staged) return c->fail(VELO_E_INVALID, "a roll is begun already: velo_map_roll_publish first");
    static const bool trace_roll = getenv("VELO_TRACE_ROLL") != nullptr;   // (where the host's time in this call goes)
    const auto tr0 = std::chrono::steady_clock::now();
    auto tr_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(); };
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    if (int rc = roll_precheck(c, "velo_map_roll_begin", lo, hi, x, y, z, n, mn, mx, true)) return rc;
    // (the previous roll was published on its geometry event and the main stream still owes the wait for its normals: enqueued
    //  NOW, before ev_roll is recorded again for this roll -- afterwards the wait would be for the roll begun here)
    if (int rc = settle_normals(c)) return rc;
    const double t_pre = tr_us();
    HIP_TRY(c, hipSetDevice(c->device));
    // which stream: the CU-masked one for a roll that can fill the chip for milliseconds, the plain one for a light roll
    // (cfg.roll_cus = -1 / VELO_ROLL_NO_CU_MASK: the first is unmasked too; VELO_ROLL_LIGHT_MAX: the threshold, measurement)
    static const long light_max = [] { const char* e = getenv("VELO_ROLL_LIGHT_MAX"); return e ? atol(e) : 32768L; }();
    const bool heavy = (long)n > light_max || (lo != nullptr && c->info.n_points > 2000000ull);
    if (heavy && !c->roll_stream) {   // (made when first needed: a mapping stream never creates it)
        // A stream that may use THREE QUARTERS of the CUs (8 of every XCD's 32 are masked out).  The roll's kernels
        // fill whatever they may run on for 2 ms; the registration on the main stream is a chain of small dependent
        // kernels, one of them (k_reduce_solve) a single 1 024-thread workgroup that needs a nearly empty CU -- beside
        // an unmasked roll it waited 0.9 - 4.7 ms for one, whatever the stream priorities and however the roll's
        // grids were cut (profiles/r05/roll_begin_trace_*.txt).  The masked quarter is always free of roll work.
        // Which bits: measured with tools/cu_mask_probe (profiles/r05/cu_mask_probe.txt) -- bit i is XCD i % 8, shader
        // engine (i / 8) % 4, CU (i / 32) of that engine.  The free CUs must be spread over ALL shader engines: a
        // workgroup is handed to an engine before anyone asks whether that engine has room (with engine 3 of every
        // XCD masked out instead, the solve still waited for 1.3 ms three times out of four).  So: the last two CUs
        // of every engine, i.e. bits 3/4 x n_CUs and up.  No such stream (older runtime): a plain one, still correct.
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        int roll_cus = ncu - ncu / 4;
        int want_cus = c->cfg.roll_cus;
        if (want_cus == 0) {   // (measurement overrides of the default)
            if (const char* e = getenv("VELO_ROLL_CUS")) want_cus = atoi(e);
            if (getenv("VELO_ROLL_NO_CU_MASK")) want_cus = -1;
        }
        if (want_cus >= 32 && want_cus <= ncu) roll_cus = want_cus / 32 * 32;
        for (int i = 0; i < roll_cus; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
        if (want_cus < 0 || hipExtStreamCreateWithCUMask(&c->roll_stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            if (getenv("VELO_TRACE_ROLL")) std::fprintf(stderr, "velo: the roll's stream has no CU mask\n");
            HIP_TRY(c, hipStreamCreateWithFlags(&c->roll_stream, hipStreamNonBlocking));
        }
    }
    if (!c->ev_roll) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_roll, hipEventDisableTiming));
    if (!c->h_roll) HIP_TRY(c, hipHostMalloc((void**)&c->h_roll, sizeof *c->h_roll, 0));
    if (int rc = resolve_roll_counts(c)) return rc;  // (the previous roll's, before its pinned slots are reused)
    // the entering points through pinned memory: a copy from pageable memory would hold the host
    const float* px = x;
    const float* py = y;
    const float* pz = z;
    if (n) {
        if (c->h_enter_cap < 3 * n) {
            if (c->h_enter) {
                if (c->roll_last) HIP_TRY(c, hipStreamSynchronize(c->roll_last));
                (void)hipHostFree(c->h_enter);
                c->h_enter = nullptr;
                c->h_enter_cap = 0;
            }
            const size_t want = 3 * n + 3 * n / 4 + 4096;
            HIP_TRY(c, hipHostMalloc((void**)&c->h_enter, want * sizeof(float), 0));
            c->h_enter_cap = want;
        } else {
            if (c->roll_last) HIP_TRY(c, hipStreamSynchronize(c->roll_last));  // (the previous roll's copy out of this buffer: long done)
        }
        std::memcpy(c->h_enter, x, n * sizeof(float));
        std::memcpy(c->h_enter + n, y, n * sizeof(float));
        std::memcpy(c->h_enter + 2 * n, z, n * sizeof(float));
        px = c->h_enter;
        py = c->h_enter + n;
        pz = c->h_enter + 2 * n;
    }
    const double t_copy = tr_us();
    if (!c->res_pending) {
        // no registration outstanding: everything enqueued so far (the last frame's increment, which reads the arrays
        // this roll's second update may rewrite) lies before this mark
        if (!c->ev_mark) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_mark, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->ev_mark, c->stream));
        c->mark_valid = true;
    }
    if (!heavy && !c->roll_stream_light) HIP_TRY(c, hipStreamCreateWithFlags(&c->roll_stream_light, hipStreamNonBlocking));
    hipStream_t rs = heavy ? c->roll_stream : c->roll_stream_light;
    if (c->ev_roll_recorded && c->roll_last && c->roll_last != rs)
        HIP_TRY(c, hipStreamWaitEvent(rs, c->ev_roll, 0));   // (the previous roll, on the other stream: same buffers)
    c->roll_last = rs;
    HIP_TRY(c, hipStreamWaitEvent(rs, c->ev_mark, 0));
    c->roll_overlapped_done = true;
    const uint64_t n_before = c->info.n_points;
    c->defer_counts = true;
    c->have_enter_mm = n > 0;
    for (int a = 0; a < 3; ++a) {
        c->enter_mn[a] = mn[a];
        c->enter_mx[a] = mx[a];
    }
    c->roll_extra = n;
    c->h_roll->invalid = c->info.n_invalid_normals;
    c->h_roll->n_done = 0;
    c->roll_counts_pending = false;
    const uint64_t gen0 = c->map_gen;
    const int rc = roll_run(c, rs, c->roll_temp, lo, hi, px, py, pz, n);
    if (trace_roll)
        std::fprintf(stderr, "roll_begin: precheck %.0f us, pinned copy %.0f, evict%s + append of %zu enqueued %.0f (rc %d)\n", t_pre,
                     t_copy - t_pre, lo ? "" : " (none)", n, tr_us() - t_copy, rc);
    c->defer_counts = false;
    c->have_enter_mm = false;
    c->roll_extra = 0;
    const hipError_t e = hipEventRecord(c->ev_roll, rs);
    c->ev_roll_recorded = e == hipSuccess;
    if (c->map_gen != gen0) c->roll_staged = true;  // (something was published into mv: the readers are behind it now)
    (void)n_before;
    if (rc == VELO_E_AGAIN) return c->fail(VELO_E_AGAIN, "this update needs the map rebuilt: not beside a registration");
    if (rc) return rc;
    if (e != hipSuccess) return c->fail(VELO_E_DEVICE, "roll stream: %s", hipGetErrorString(e));
    return VELO_OK;
}

int velo_map_roll_publish(velo_ctx* c)
{
    if (!c) return VELO_E_INVALID;
    if (!c->roll_staged) return VELO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // the main stream waits ON THE DEVICE for the roll; what it runs from here on reads the new map.  For the rolled
    // map's geometry only where the roll says when that is complete: an increment enqueued next reads points and table,
    // the first reader of normals (the next registration: settle_normals) waits for the roll's last kernel
    if (c->roll_geom_recorded && c->ev_roll_geom) {
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_roll_geom, 0));
        c->normals_wait_owed = true;
    } else {
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_roll, 0));
    }
    c->roll_geom_recorded = false;
    c->mv_read = c->mv;
    c->roll_staged = false;
    ++c->map_gen;  // (captured graphs and hints of the old map are stale)
    return reset_hints(c, c->stream);
}

int velo_map_set_margins(velo_ctx* c, const int32_t margin[3])
{
    if (!c || !margin) return VELO_E_INVALID;
    for (int a = 0; a < 3; ++a)
        if (margin[a] < 0 || margin[a] > 100000) return c->fail(VELO_E_INVALID, "margin out of range");
    for (int a = 0; a < 3; ++a) c->margin[a] = margin[a];
    return VELO_OK;
}

int velo_map_info_get(velo_ctx* c, velo_map_info* out)
{
    if (!c || !out) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    // the caller says how large ITS velo_map_info is; never write past that (the struct has grown
    // before and will again)
    if (int rc = resolve_roll_counts(c)) return rc;
    const uint32_t have = out->struct_size;
    if (have < offsetof(velo_map_info, n_points) + sizeof(uint64_t))
        return c->fail(VELO_E_INVALID, "velo_map_info.struct_size must be set to sizeof(velo_map_info) by the caller");
    velo_map_info full = c->info;
    full.struct_size = (uint32_t)std::min<size_t>(have, sizeof full);
    full.reserved0 = 0;
    std::memcpy(out, &full, full.struct_size);
    return VELO_OK;
}

int velo_map_size(velo_ctx* c, uint64_t* n_points)
{
    if (!c || !n_points) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    *n_points = c->info.n_points;
    return VELO_OK;
}

int velo_map_download(velo_ctx* c, float* x, float* y, float* z, float* nx, float* ny, float* nz,
                      int32_t* perm, int32_t* cell_start)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_roll(c)) return rc;
    const size_t n = c->info.n_points;
    std::vector<float4> h;
    if (x || y || z || nx || ny || nz) h.resize(n);
    if (x || y || z) {
        HIP_TRY(c, hipMemcpyAsync(h.data(), c->pts.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < n; ++i) {
            if (x) x[i] = h[i].x;
            if (y) y[i] = h[i].y;
            if (z) z[i] = h[i].z;
        }
    }
    if (nx || ny || nz) {
        HIP_TRY(c, hipMemcpyAsync(h.data(), c->nrm.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < n; ++i) {
            if (nx) nx[i] = h[i].x;
            if (ny) ny[i] = h[i].y;
            if (nz) nz[i] = h[i].z;
        }
    }
    if (perm) HIP_TRY(c, hipMemcpyAsync(perm, c->perm.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (cell_start) {
        if (c->use_hash) {  // the dense prefix table the sparse one stands for, built for the copy
            if (c->info.n_cells + 8 >= 2147483648ull)
                return c->fail(VELO_E_RANGE, "the dense cell table of this map has more than 2^31 entries");
            HIP_TRY(c, c->cell_start.reserve(c->info.n_cells + 8));
            HIP_TRY(c, reserve_slack(c->tile_bounds, cell_start_bounds(c->info.n_cells)));
            HIP_TRY(c, launch_cell_start(c->keys_sorted.p, n, c->info.n_cells, c->cell_start.p, c->tile_bounds.p, c->stream));
        }
        HIP_TRY(c, hipMemcpyAsync(cell_start, c->cell_start.p, (c->info.n_cells + 1) * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VELO_OK;
}

// ---------------------------------------------------------------- compensate
int velo_compensate_dev(velo_ctx* c, const float* dx, const float* dy, const float* dz,
                        const uint16_t* dpkt, size_t n, const double* dT, size_t n_pkt, float* dox,
                        float* doy, float* doz)
{
    if (!c) return VELO_E_INVALID;
    if (n == 0) return VELO_OK;
    if (!dx || !dy || !dz || !dpkt || !dT || !dox || !doy || !doz || n_pkt == 0)
        return c->fail(VELO_E_INVALID, "velo_compensate: null argument or empty transform table");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_compensate(dx, dy, dz, dpkt, n, dT, n_pkt, dox, doy, doz, c->stream));
    return VELO_OK;
}

int velo_compensate(velo_ctx* c, const float* x, const float* y, const float* z,
                    const uint16_t* pkt, size_t n, const double* T3x4, size_t n_pkt, float* ox,
                    float* oy, float* oz)
{
    if (!c) return VELO_E_INVALID;
    if (n == 0) return VELO_OK;
    if (!x || !y || !z || !pkt || !T3x4 || !ox || !oy || !oz || n_pkt == 0)
        return c->fail(VELO_E_INVALID, "velo_compensate: null argument or empty transform table");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<float> in, out;
    DevBuf<uint16_t> dp;
    DevBuf<double> dt;
    const size_t n4 = (n + 3) & ~(size_t)3;  // keep the three planes 16-byte aligned
    HIP_TRY(c, in.reserve(3 * n4));
    HIP_TRY(c, out.reserve(3 * n4));
    HIP_TRY(c, dp.reserve(n4));
    HIP_TRY(c, dt.reserve(12 * n_pkt));
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(in.p, x, n * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(in.p + n4, y, n * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(in.p + 2 * n4, z, n * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dp.p, pkt, n * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dt.p, T3x4, 12 * n_pkt * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, launch_compensate(in.p, in.p + n4, in.p + 2 * n4, dp.p, n, dt.p, n_pkt, out.p,
                                 out.p + n4, out.p + 2 * n4, s));
    HIP_TRY(c, hipMemcpyAsync(ox, out.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(oy, out.p + n4, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(oz, out.p + 2 * n4, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return VELO_OK;
}

// -------------------------------------------------------------------- frames
int velo_frames_upload(velo_ctx* c, int n_frames, const float* x, const float* y, const float* z,
                       const int64_t* frame_start)
{
    if (!c) return VELO_E_INVALID;
    if (!x || !y || !z) return c->fail(VELO_E_INVALID, "null point array");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = plan_frames(c, n_frames, frame_start)) return rc;
    const size_t n = (size_t)frame_start[n_frames];
    HIP_TRY(c, c->fx.reserve(std::max<size_t>(n, 1)));
    HIP_TRY(c, c->fy.reserve(std::max<size_t>(n, 1)));
    HIP_TRY(c, c->fz.reserve(std::max<size_t>(n, 1)));
    HIP_TRY(c, hipMemcpyAsync(c->fx.p, x, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->fy.p, y, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->fz.p, z, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->ax = c->fx.p;
    c->ay = c->fy.p;
    c->az = c->fz.p;
    c->last_iters = 0;
    return VELO_OK;
}

int velo_frames_adopt_dev(velo_ctx* c, int n_frames, const float* dx, const float* dy,
                          const float* dz, const int64_t* frame_start)
{
    if (!c) return VELO_E_INVALID;
    if (!dx || !dy || !dz) return c->fail(VELO_E_INVALID, "null point array");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = plan_frames(c, n_frames, frame_start)) return rc;
    c->ax = dx;
    c->ay = dy;
    c->az = dz;
    c->last_iters = 0;
    return VELO_OK;
}

int velo_icp_batch_async(velo_ctx* c, const double* T0, int iters, float d_max)
{
    if (!c) return VELO_E_INVALID;
    return run_icp(c, T0, iters, d_max);
}

// start = async + the result copies + an event; finish waits for THAT event only: work enqueued on
// the ctx stream after the start (the increment, the next frame's decode) is not waited for, and the
// resident frames may already be the next ones
int velo_icp_batch_start(velo_ctx* c, const double* T0, int iters, float d_max)
{
    if (!c) return VELO_E_INVALID;
    c->res_pending = false;
    HIP_TRY(c, hipSetDevice(c->device));
    // everything enqueued so far -- the previous frame's registration and increment included -- lies
    // before this mark: what a decode on the side stream has to wait for before it may write the
    // buffers those read (velo_decode_submit_overlapped)
    if (!c->ev_mark) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_mark, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_mark, c->stream));
    c->mark_valid = true;
    // (a roll begun before this registration and not yet published is this registration's roll: the registration reads
    //  the arrays the roll left behind, a second roll beside it would rewrite them)
    c->roll_overlapped_done = c->roll_staged;
    if (int rc = run_icp(c, T0, iters, d_max)) return rc;
    if (int rc = enqueue_result_copies(c, c->n_frames)) return rc;
    if (!c->ev_res) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_res, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_res, c->stream));
    c->res_frames = c->n_frames;
    c->res_iters = c->last_iters;
    c->res_pending = true;
    return VELO_OK;
}

int velo_icp_batch_finish(velo_ctx* c, velo_icp_result* out)
{
    if (!c) return VELO_E_INVALID;
    if (!out) return c->fail(VELO_E_INVALID, "out is null");
    if (!c->res_pending) return c->fail(VELO_E_INVALID, "no registration started with velo_icp_batch_start is outstanding");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev_res));
    c->res_pending = false;
    parse_results(c, c->res_frames, c->res_iters, out);
    return VELO_OK;
}

int velo_icp_batch_fetch(velo_ctx* c, velo_icp_result* out)
{
    if (!c) return VELO_E_INVALID;
    return fetch_icp(c, out);
}

int velo_icp_batch(velo_ctx* c, const double* T0, int iters, float d_max, velo_icp_result* out)
{
    if (!c) return VELO_E_INVALID;
    if (int rc = run_icp(c, T0, iters, d_max)) return rc;
    return fetch_icp(c, out);
}

int velo_icp(velo_ctx* c, const float* x, const float* y, const float* z, size_t n,
             const double T0[12], int iters, float d_max, int k, velo_icp_result* out)
{
    if (!c) return VELO_E_INVALID;
    if (k != 1) return c->fail(VELO_E_INVALID, "k must be 1 in ABI version %d", VELO_ABI_VERSION);
    if (n == 0) return c->fail(VELO_E_INVALID, "empty frame");
    const int64_t fs[2] = {0, (int64_t)n};
    if (int rc = velo_frames_upload(c, 1, x, y, z, fs)) return rc;
    return velo_icp_batch(c, T0, iters, d_max, out);
}

int velo_linearize(velo_ctx* c, int frame, const double T[12], float d_max, int32_t* corr,
                   float* d2, double acc[29])
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    if (frame < 0 || frame >= c->n_frames) return c->fail(VELO_E_INVALID, "frame index out of range");
    if (!T) return c->fail(VELO_E_INVALID, "T is null");
    if (!(d_max > 0.0f) || !(d_max <= c->mv_read.h))
        return c->fail(VELO_E_RANGE, "d_max must be in (0, voxel=%g]", (double)c->mv_read.h);
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_normals(c)) return rc;
    hipStream_t s = c->stream;
    const size_t n_all = (size_t)c->frame_start[c->n_frames];
    const size_t q0 = (size_t)c->frame_start[frame], q1 = (size_t)c->frame_start[frame + 1];
    HIP_TRY(c, c->corr.reserve(std::max<size_t>(n_all, 1)));
    HIP_TRY(c, c->d2.reserve(std::max<size_t>(n_all, 1)));
    if (c->lin_hints && c->hint.cap < n_all) {
        HIP_TRY(c, c->hint.reserve(n_all));
        HIP_TRY(c, hipMemsetAsync(c->hint.p, 0xFF, n_all * sizeof(int32_t), s));
    }
    HIP_TRY(c, c->poses_prev.reserve((size_t)c->cfg.max_batch * 12));
    if (c->lin_hints) {
        if (c->rho.cap < n_all) {
            HIP_TRY(c, c->rho.reserve(n_all));
            HIP_TRY(c, hipMemsetAsync(c->rho.p, 0, n_all * sizeof(float), s));
        }
        // the pose of the previous call is this call's "previous iteration"
        HIP_TRY(c, hipMemcpyAsync(c->poses_prev.p + 12 * (size_t)frame, c->poses.p + 12 * (size_t)frame,
                                  12 * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    // poses of the other frames are irrelevant here: only this frame's blocks are launched
    HIP_TRY(c, hipMemcpyAsync(c->poses.p + 12 * (size_t)frame, T, 12 * sizeof(double),
                              hipMemcpyHostToDevice, s));
    FrameView fv{c->ax, c->ay, c->az, nullptr};
    const int b0 = c->fbs_h[frame], b1 = c->fbs_h[frame + 1];
    HIP_TRY(c, launch_linearize(c->cfg.linearize_variant, c->items.p + b0, b1 - b0, fv, c->mv_read,
                                c->poses.p, d_max * d_max, c->partials.p, c->corr.p, c->d2.p,
                                c->lin_hints ? c->hint.p : nullptr,
                                (c->lin_hints && c->cfg.use_hints >= 2) ? c->rho.p : nullptr,
                                c->poses_prev.p, c->stats_on,
                                // (the frames are cut for one kernel: say which, item counts no longer do)
                                c->cfg.linearize_variant == VELO_VARIANT_BALL ? (c->plan_lat ? 2 : 1) : c->cfg.force_kernel,
                                s, c->lat_first_lanes));
    HIP_TRY(c, launch_reduce_solve(c->partials.p, c->fbs.p + frame, c->lay.p + frame, 1, c->poses.p, nullptr, 0, c->cfg.solve_threads,
                                   c->acc.p, 0, nullptr, nullptr, s));
    if (corr)
        HIP_TRY(c, hipMemcpyAsync(corr, c->corr.p + q0, (q1 - q0) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (d2)
        HIP_TRY(c, hipMemcpyAsync(d2, c->d2.p + q0, (q1 - q0) * sizeof(float), hipMemcpyDeviceToHost, s));
    double a[kAccStride];
    HIP_TRY(c, hipMemcpyAsync(a, c->acc.p, sizeof a, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (acc) std::memcpy(acc, a, kAccN * sizeof(double));
    return VELO_OK;
}

// a12 on its own (parity tests): the 29 sums -> LDLt solve -> T <- exp(xi^) T, on the device
// (k_reduce_solve over one partial row, the kernel every registration iteration runs)
int velo_solve_update(velo_ctx* c, const double acc[29], double T[12], int32_t* solve_flag)
{
    if (!c) return VELO_E_INVALID;
    if (!acc || !T) return c->fail(VELO_E_INVALID, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    DevBuf<double> part, pose;
    DevBuf<int32_t> fbs;
    DevBuf<RowLayout> lay;
    DevBuf<velo_icp_iter> st;
    HIP_TRY(c, part.reserve(kAccStride));
    HIP_TRY(c, pose.reserve(12));
    HIP_TRY(c, fbs.reserve(2));
    HIP_TRY(c, lay.reserve(1));
    const RowLayout one{0, 0, 0, 1};  // one row: the root
    HIP_TRY(c, hipMemcpyAsync(lay.p, &one, sizeof one, hipMemcpyHostToDevice, s));
    HIP_TRY(c, st.reserve(VELO_MAX_ITERS));
    double row[kAccStride] = {0};
    std::memcpy(row, acc, kAccN * sizeof(double));
    const int32_t range[2] = {0, 1};
    HIP_TRY(c, hipMemcpyAsync(part.p, row, sizeof row, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(pose.p, T, 12 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(fbs.p, range, sizeof range, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));  // the sources are stack arrays
    HIP_TRY(c, launch_reduce_solve(part.p, fbs.p, lay.p, 1, pose.p, st.p, 0, c->cfg.solve_threads, nullptr, 1, nullptr, nullptr, s));
    velo_icp_iter it0;
    HIP_TRY(c, hipMemcpyAsync(T, pose.p, 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&it0, st.p, sizeof it0, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (solve_flag) *solve_flag = (int32_t)it0.solve_flag;
    return VELO_OK;
}

// -------------------------------------------------------------------- decode (f1)
namespace {

// HDLParser.cxx:179-181: vertical-order index -> raw laser id
const int kHdl64BeamLut[64] = {38, 39, 42, 43, 32, 33, 36, 37, 40, 41, 46, 47, 50, 51, 54, 55,
                               44, 45, 48, 49, 52, 53, 58, 59, 62, 63, 34, 35, 56, 57, 60, 61,
                               6,  7,  10, 11, 0,  1,  4,  5,  8,  9,  14, 15, 18, 19, 22, 23,
                               12, 13, 16, 17, 20, 21, 26, 27, 30, 31, 2,  3,  24, 25, 28, 29};

// sin/cos tables exactly as the reference builds / evaluates them on the host (libm), so the
// device results are bit-identical: 36001-entry LUT (HDLParser.cxx:754-768) and, for lasers
// with a non-zero azimuth correction, cos/sin((az/100 - corr) deg) per azimuth (:607-611)
int upload_calibration(velo_ctx* c, const velo_laser_corr corr[64])
{
    const double* cd = reinterpret_cast<const double*>(corr);
    if (c->dk_corr_host.size() == 64 * 9 && std::memcmp(c->dk_corr_host.data(), cd, 64 * 9 * sizeof(double)) == 0)
        return VELO_OK;
    std::vector<double> lc(36001), ls(36001);
    for (unsigned i = 0; i < 36001; ++i) {
        const double rad = (i / 100.0) * M_PI / 180.0;
        lc[i] = std::cos(rad);
        ls[i] = std::sin(rad);
    }
    std::vector<double> ac((size_t)64 * 36000, 0.0), as((size_t)64 * 36000, 0.0);
    for (int l = 0; l < 64; ++l) {
        if (corr[l].azimuthCorrection == 0) continue;
        for (unsigned a = 0; a < 36000; ++a) {
            const double rad = ((static_cast<double>(a) / 100.0) - corr[l].azimuthCorrection) * M_PI / 180.0;
            ac[(size_t)l * 36000 + a] = std::cos(rad);
            as[(size_t)l * 36000 + a] = std::sin(rad);
        }
    }
    uint8_t inv[64];
    for (int i = 0; i < 64; ++i) inv[kHdl64BeamLut[i]] = (uint8_t)i;
    HIP_TRY(c, c->dk_corr.reserve(64 * 9));
    HIP_TRY(c, c->dk_lutc.reserve(36001));
    HIP_TRY(c, c->dk_luts.reserve(36001));
    HIP_TRY(c, c->dk_azc.reserve((size_t)64 * 36000));
    HIP_TRY(c, c->dk_azs.reserve((size_t)64 * 36000));
    HIP_TRY(c, c->dk_invlut.reserve(64));
    HIP_TRY(c, hipMemcpy(c->dk_corr.p, cd, 64 * 9 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->dk_lutc.p, lc.data(), 36001 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->dk_luts.p, ls.data(), 36001 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->dk_azc.p, ac.data(), ac.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->dk_azs.p, as.data(), as.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->dk_invlut.p, inv, 64, hipMemcpyHostToDevice));
    c->dk_corr_host.assign(cd, cd + 64 * 9);
    return VELO_OK;
}

}  // namespace

// staging memory of a decode plan: pinned (the one copy up is then asynchronous)
static void* plan_stage_alloc(size_t n)
{
    void* p = nullptr;
    return hipHostMalloc(&p, n, 0) == hipSuccess ? p : nullptr;
}
static void plan_stage_free(void* p) { (void)hipHostFree(p); }

// Device half: the planned packet set through the decode kernels.  `st` (optional) receives the
// parser state the plan carries on.
static int decode_submit(velo_ctx* c, velo_ctx::DecodePlan& P, velo_ctx::DecodeStream* st, int32_t* n_frames,
                         size_t* n_points)
{
    if (!P.filled) return c->fail(VELO_E_INVALID, "velo_decode: nothing planned");
    P.filled = false;  // (the staging buffer is free again at the synchronisation below)
    HIP_TRY(c, hipSetDevice(c->device));
    c->dk_sel ^= 1;  // write the other set
    if (c->ax && c->ax == c->dk_out[c->dk_sel].x.p) {
        // the resident frames ARE the previous decode's output (velo_decode_to_frames): this
        // decode rewrites (and may reallocate) those buffers, so the adoption ends here --
        // registration calls fail with VELO_E_INVALID until frames are uploaded/adopted again
        c->ax = c->ay = c->az = nullptr;
        c->n_frames = 0;
        c->last_iters = 0;
        ++c->frames_gen;
    }
    if (int rc = upload_calibration(c, P.corr)) return rc;
    hipStream_t s = c->stream;
    const size_t n_pkt = P.n_pkt;
    const int nfr = P.nfr;
    c->dk_carposes = P.carposes;
    c->dk_frame_t = P.frame_t;
    c->dk_frame_packets = P.frame_packets;
    const size_t n_ret = n_pkt * 384;
    HIP_TRY(c, c->dk_keys.reserve(n_ret));
    HIP_TRY(c, c->dk_keys2.reserve(n_ret));
    HIP_TRY(c, c->dk_idx.reserve(n_ret));
    HIP_TRY(c, c->dk_order.reserve(n_ret));
    const uint32_t n_keys = (uint32_t)std::max(nfr, 1) * 64u;
    HIP_TRY(c, c->dk_starts.reserve((size_t)n_keys + 1));
    HIP_TRY(c, c->dk_stage.reserve(P.stage_bytes));
    HIP_TRY(c, hipMemcpyAsync(c->dk_stage.p, P.stage, P.stage_bytes, hipMemcpyHostToDevice, s));
    DecodeView v;
    v.pkts = c->dk_stage.p + P.o_pk;
    v.blk_frame = reinterpret_cast<const int16_t*>(c->dk_stage.p + P.o_blk);
    v.frame_perm = c->dk_stage.p + P.o_perm;
    v.table = reinterpret_cast<const double*>(c->dk_stage.p + P.o_tab);
    v.tvalid = c->dk_stage.p + P.o_tv;
    v.az_diff = reinterpret_cast<const int32_t*>(c->dk_stage.p + P.o_az);
    v.corr = c->dk_corr.p;
    v.lut_cos = c->dk_lutc.p;
    v.lut_sin = c->dk_luts.p;
    v.az_cos = c->dk_azc.p;
    v.az_sin = c->dk_azs.p;
    v.inv_lut = c->dk_invlut.p;
    v.laser_mask = P.laser_mask;
    v.n_pkt = (int)n_pkt;
    v.n_lasers = P.n_lasers;
    v.crop = P.crop;
    v.crop_inside = P.crop_inside;
    for (int i = 0; i < 6; ++i) v.region[i] = P.region[i];
    HIP_TRY(c, launch_decode_keys(v, n_ret, c->dk_keys.p, c->dk_idx.p, s));
    // valid keys are < n_keys, the key of a dropped return is all ones: the low bits that cover
    // n_keys order both (one radix pass for a frame or two instead of four over 32 bits)
    int key_bits = 1;
    while (key_bits < 32 && (1u << key_bits) <= n_keys) ++key_bits;
    size_t tb = 0;
    HIP_TRY(c, sort_pairs(nullptr, tb, c->dk_keys.p, c->dk_keys2.p, c->dk_idx.p, c->dk_order.p, n_ret, key_bits, s));
    if (int rc = ensure_temp(c, tb)) return rc;
    HIP_TRY(c, sort_pairs(c->temp.p, tb, c->dk_keys.p, c->dk_keys2.p, c->dk_idx.p, c->dk_order.p, n_ret, key_bits,
                          s));
    HIP_TRY(c, launch_key_starts(c->dk_keys2.p, n_ret, n_keys, c->dk_starts.p, s));
    // (pinned landing zone: a copy into pageable memory is staged and waited for inside the call)
    const size_t n_starts = (size_t)n_keys + 1;
    if (n_starts > c->h_starts_cap) {
        if (c->h_starts) (void)hipHostFree(c->h_starts);
        c->h_starts = nullptr;
        c->h_starts_cap = 0;
        HIP_TRY(c, hipHostMalloc((void**)&c->h_starts, (n_starts + 256) * sizeof(int32_t), 0));
        c->h_starts_cap = n_starts + 256;
    }
    int32_t* const starts = c->h_starts;
    HIP_TRY(c, hipMemcpyAsync(starts, c->dk_starts.p, n_starts * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const size_t n_valid = nfr > 0 ? (size_t)starts[(size_t)nfr * 64] : 0;
    HIP_TRY(c, reserve_slack(c->dk_out[c->dk_sel].x, std::max<size_t>(n_valid, 1)));
    HIP_TRY(c, reserve_slack(c->dk_out[c->dk_sel].y, std::max<size_t>(n_valid, 1)));
    HIP_TRY(c, reserve_slack(c->dk_out[c->dk_sel].z, std::max<size_t>(n_valid, 1)));
    HIP_TRY(c, reserve_slack(c->dk_out[c->dk_sel].i, std::max<size_t>(n_valid, 1)));
    HIP_TRY(c, reserve_slack(c->dk_out[c->dk_sel].dist, std::max<size_t>(n_valid, 1)));
    HIP_TRY(c, reserve_slack(c->dk_out[c->dk_sel].az, std::max<size_t>(n_valid, 1)));
    HIP_TRY(c, reserve_slack(c->dk_out[c->dk_sel].pidx, std::max<size_t>(n_valid, 1)));
    HIP_TRY(c, launch_decode_emit(v, c->dk_order.p, n_valid, c->dk_out[c->dk_sel].x.p, c->dk_out[c->dk_sel].y.p, c->dk_out[c->dk_sel].z.p, c->dk_out[c->dk_sel].i.p,
                                  c->dk_out[c->dk_sel].az.p, c->dk_out[c->dk_sel].dist.p, c->dk_out[c->dk_sel].pidx.p, s));
    // (no wait for the emit: whoever reads the frames -- velo_decode_fetch, the registration -- is
    // ordered behind it on the ctx stream; launching it over an upper bound BEFORE the count is
    // known, so that the wait overlaps with it, measured no different: 0.120-0.151 ms either way)
    c->dk_frames = nfr;
    c->dk_points = n_valid;
    c->dk_frame_start.assign((size_t)nfr + 1, 0);
    c->dk_beam_start.assign((size_t)nfr * 65, 0);
    for (int f = 0; f < nfr; ++f) {
        c->dk_frame_start[f] = starts[(size_t)f * 64];
        for (int b = 0; b <= 64; ++b) c->dk_beam_start[(size_t)f * 65 + b] = starts[(size_t)f * 64 + b];
    }
    c->dk_frame_start[nfr] = (int64_t)n_valid;
    if (n_frames) *n_frames = nfr;
    if (n_points) *n_points = n_valid;
    if (P.keep_state && st) {
        if (P.flush)
            *st = velo_ctx::DecodeStream();  // everything emitted: the next packet starts afresh
        else
            *st = std::move(P.st_next);
    }
    return VELO_OK;
}

static int decode_impl(velo_ctx* c, velo_ctx::DecodeStream& st, const uint8_t* packets,
                       const int64_t* pkt_t_us, size_t n_new, const velo_laser_corr corr[64],
                       int n_lasers, const velo_pose* poses, size_t n_poses, int flush,
                       const double* crop_region, int crop_inside, bool keep_state,
                       int32_t* n_frames, size_t* n_points)
{
    HIP_TRY(c, hipSetDevice(c->device));
    c->dplan.alloc_fn = plan_stage_alloc;
    c->dplan.free_fn = plan_stage_free;
    if (velo::decode_plan_host(c->dplan, st, c->dopts, packets, pkt_t_us, n_new, corr, n_lasers, poses, n_poses, flush,
                         crop_region, crop_inside, keep_state))
        return c->fail(c->dplan.code, "%s", c->dplan.err);
    return decode_submit(c, c->dplan, &st, n_frames, n_points);
}

// ---- the two halves, separately: plan frame k + 1 on the host while the GPU registers frame k
struct velo_decode_plan {
    velo_ctx::DecodePlan P;
    int device = 0;
};

int velo_decode_plan_create(velo_ctx* c, velo_decode_plan** out)
{
    if (!c || !out) return VELO_E_INVALID;
    velo_decode_plan* p = new (std::nothrow) velo_decode_plan;
    if (!p) return c->fail(VELO_E_NOMEM, "out of memory");
    p->device = c->device;
    p->P.alloc_fn = plan_stage_alloc;
    p->P.free_fn = plan_stage_free;
    *out = p;
    return VELO_OK;
}

void velo_decode_plan_destroy(velo_decode_plan* p) { delete p; }

const char* velo_decode_plan_error(const velo_decode_plan* p) { return p ? p->P.err : "null plan"; }

int velo_decode_plan_fill(velo_decode_plan* p, const velo_decode_opts* opts, const uint8_t* packets,
                          const int64_t* pkt_t_us, size_t n_pkt, const velo_laser_corr corr[64], int n_lasers,
                          const velo_pose* poses, size_t n_poses, int flush, const double* crop_region,
                          int crop_inside)
{
    if (!p) return VELO_E_INVALID;
    velo_decode_opts d{};
    d.struct_size = sizeof d;
    std::memset(d.laser_selection, 1, sizeof d.laser_selection);
    if (opts) {
        if (opts->struct_size < sizeof d) return p->P.fail(VELO_E_INVALID, "velo_decode_opts.struct_size too small");
        if (opts->points_skip < 0 || opts->initial_firing_skip < 0 || opts->initial_firing_skip > 12)
            return p->P.fail(VELO_E_INVALID, "points_skip must be >= 0, initial_firing_skip in [0,12]");
        d = *opts;
        d.struct_size = sizeof d;
    }
    if (n_pkt == 0) return p->P.fail(VELO_E_RANGE, "n_pkt must be in [1, 60000]");
    (void)hipSetDevice(p->device);  // (the pinned staging buffer may be allocated from this thread)
    const velo_ctx::DecodeStream fresh;
    return velo::decode_plan_host(p->P, fresh, d, packets, pkt_t_us, n_pkt, corr, n_lasers, poses, n_poses, flush,
                            crop_region, crop_inside, false);
}

int velo_decode_submit(velo_ctx* c, velo_decode_plan* p, int32_t* n_frames, size_t* n_points)
{
    if (!c || !p) return VELO_E_INVALID;
    c->mark_valid = false;
    return decode_submit(c, p->P, nullptr, n_frames, n_points);
}

// velo_decode_submit + velo_decode_to_frames on the ctx's SIDE stream, concurrently with the
// registration velo_icp_batch_start put on the main stream.  Safe because (a) the decode writes the
// output set, and the frame plan the device copy, that the frames being registered do NOT use (both
// alternate), (b) the side stream first waits for the mark velo_icp_batch_start recorded -- everything
// older than the running registration, i.e. whatever still read those buffers -- and (c) the main
// stream is made to wait for the side stream's end before anything enqueued LATER runs (the next
// registration, an increment, a fetch).  The call itself waits only for the side stream.
int velo_decode_submit_overlapped(velo_ctx* c, velo_decode_plan* p, int32_t* n_frames, size_t* n_points)
{
    if (!c || !p) return VELO_E_INVALID;
    if (!c->mark_valid || !c->res_pending)
        return c->fail(VELO_E_INVALID, "velo_decode_submit_overlapped needs a registration started with velo_icp_batch_start "
                                       "and not yet finished");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->side_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    if (!c->ev_side) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    c->mark_valid = false;  // one overlapped decode per registration
    HIP_TRY(c, hipStreamWaitEvent(c->side_stream, c->ev_mark, 0));
    hipStream_t main_stream = c->stream;
    c->stream = c->side_stream;  // (decode_submit and plan_frames enqueue on c->stream)
    std::swap(c->temp.p, c->dk_temp.p);
    std::swap(c->temp.cap, c->dk_temp.cap);
    int32_t nf = 0;
    int rc = decode_submit(c, p->P, nullptr, &nf, n_points);
    if (rc == VELO_OK && nf >= 1) rc = velo_decode_to_frames(c);
    std::swap(c->temp.p, c->dk_temp.p);
    std::swap(c->temp.cap, c->dk_temp.cap);
    c->stream = main_stream;
    // whatever happened: nothing enqueued on the main stream from now on may overtake the side stream
    hipError_t e = hipEventRecord(c->ev_side, c->side_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(main_stream, c->ev_side, 0);
    if (rc) return rc;
    if (e != hipSuccess) return c->fail(VELO_E_DEVICE, "side stream: %s", hipGetErrorString(e));
    if (n_frames) *n_frames = nf;
    if (nf < 1) return c->fail(VELO_E_NODATA, "the packets hold no complete revolution");
    return VELO_OK;
}

int velo_decode(velo_ctx* c, const uint8_t* packets, const int64_t* pkt_t_us, size_t n_pkt,
                const velo_laser_corr corr[64], int n_lasers, const velo_pose* poses,
                size_t n_poses, int flush, const double* crop_region, int crop_inside,
                int32_t* n_frames, size_t* n_points)
{
    if (!c) return VELO_E_INVALID;
    if (n_pkt == 0) return c->fail(VELO_E_RANGE, "n_pkt must be in [1, 60000]");
    velo_ctx::DecodeStream fresh;
    return decode_impl(c, fresh, packets, pkt_t_us, n_pkt, corr, n_lasers, poses, n_poses, flush,
                       crop_region, crop_inside, false, n_frames, n_points);
}

int velo_decode_stream(velo_ctx* c, const uint8_t* packets, const int64_t* pkt_t_us, size_t n_pkt,
                       const velo_laser_corr corr[64], int n_lasers, const velo_pose* poses,
                       size_t n_poses, int flush, const double* crop_region, int crop_inside,
                       int32_t* n_frames, size_t* n_points)
{
    if (!c) return VELO_E_INVALID;
    if (n_pkt == 0 && c->dstream.t.empty()) {  // nothing in flight: nothing to emit
        c->dk_frames = 0;
        c->dk_points = 0;
        if (n_frames) *n_frames = 0;
        if (n_points) *n_points = 0;
        return VELO_OK;
    }
    return decode_impl(c, c->dstream, packets, pkt_t_us, n_pkt, corr, n_lasers, poses, n_poses, flush,
                       crop_region, crop_inside, true, n_frames, n_points);
}

int velo_decode_set_options(velo_ctx* c, const velo_decode_opts* o)
{
    if (!c) return VELO_E_INVALID;
    velo_decode_opts d{};
    d.struct_size = sizeof d;
    std::memset(d.laser_selection, 1, sizeof d.laser_selection);
    if (o) {
        if (o->struct_size < sizeof d) return c->fail(VELO_E_INVALID, "velo_decode_opts.struct_size too small");
        if (o->points_skip < 0 || o->initial_firing_skip < 0 || o->initial_firing_skip > 12)
            return c->fail(VELO_E_INVALID, "points_skip must be >= 0, initial_firing_skip in [0,12]");
        d = *o;
        d.struct_size = sizeof d;
    }
    c->dopts = d;
    return VELO_OK;
}

int velo_decode_stream_reset(velo_ctx* c)
{
    if (!c) return VELO_E_INVALID;
    c->dstream = velo_ctx::DecodeStream();
    return VELO_OK;
}

int velo_decode_fetch(velo_ctx* c, float* x, float* y, float* z, float* intensity, uint16_t* azimuth,
                      float* distance, uint16_t* packet_index, int64_t* frame_start,
                      int32_t* beam_start, velo_pose* carposes, int64_t* frame_t_us,
                      int32_t* frame_packets)
{
    if (!c) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->dk_points;
    hipStream_t s = c->stream;
    if (n) {
        if (x) HIP_TRY(c, hipMemcpyAsync(x, c->dk_out[c->dk_sel].x.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
        if (y) HIP_TRY(c, hipMemcpyAsync(y, c->dk_out[c->dk_sel].y.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
        if (z) HIP_TRY(c, hipMemcpyAsync(z, c->dk_out[c->dk_sel].z.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
        if (intensity) HIP_TRY(c, hipMemcpyAsync(intensity, c->dk_out[c->dk_sel].i.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
        if (azimuth) HIP_TRY(c, hipMemcpyAsync(azimuth, c->dk_out[c->dk_sel].az.p, n * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
        if (distance) HIP_TRY(c, hipMemcpyAsync(distance, c->dk_out[c->dk_sel].dist.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
        if (packet_index) HIP_TRY(c, hipMemcpyAsync(packet_index, c->dk_out[c->dk_sel].pidx.p, n * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    const size_t F = (size_t)c->dk_frames;
    if (frame_start) std::memcpy(frame_start, c->dk_frame_start.data(), (F + 1) * sizeof(int64_t));
    if (beam_start && F) std::memcpy(beam_start, c->dk_beam_start.data(), F * 65 * sizeof(int32_t));
    if (carposes && F) std::memcpy(carposes, c->dk_carposes.data(), F * sizeof(velo_pose));
    if (frame_t_us && F) std::memcpy(frame_t_us, c->dk_frame_t.data(), F * sizeof(int64_t));
    if (frame_packets && F) std::memcpy(frame_packets, c->dk_frame_packets.data(), F * sizeof(int32_t));
    return VELO_OK;
}

int velo_decode_to_frames(velo_ctx* c)
{
    if (!c) return VELO_E_INVALID;
    if (c->dk_frames < 1) return c->fail(VELO_E_INVALID, "nothing decoded");
    return velo_frames_adopt_dev(c, c->dk_frames, c->dk_out[c->dk_sel].x.p, c->dk_out[c->dk_sel].y.p, c->dk_out[c->dk_sel].z.p,
                                 c->dk_frame_start.data());
}

// ----------------------------------------------------------------------- kNN
int velo_knn(velo_ctx* c, int frame, const double T[12], float d_max, int k, int32_t* idx,
             float* d2, int32_t* count)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    if (frame < 0 || frame >= c->n_frames) return c->fail(VELO_E_INVALID, "frame index out of range");
    if (!T || !idx || !d2) return c->fail(VELO_E_INVALID, "null argument");
    if (k < 1 || k > VELO_MAX_KNORMALS) return c->fail(VELO_E_INVALID, "k must be in [1,%d]", VELO_MAX_KNORMALS);
    if (!(d_max > 0.0f) || !(d_max <= c->mv_read.h))
        return c->fail(VELO_E_RANGE, "d_max must be in (0, voxel=%g]", (double)c->mv_read.h);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t q0 = (size_t)c->frame_start[frame], n = (size_t)c->frame_start[frame + 1] - q0;
    if (n == 0) return VELO_OK;
    // (buffers of the ctx: a hipMalloc / hipFree pair per call is a device-wide synchronisation each)
    DevBuf<int32_t>&di = c->knn_idx, &dc = c->knn_cnt;
    DevBuf<float>& dd = c->knn_d2;
    HIP_TRY(c, reserve_slack(di, n * (size_t)k));
    HIP_TRY(c, reserve_slack(dd, n * (size_t)k));
    HIP_TRY(c, reserve_slack(dc, n));
    Pose12 P;
    std::memcpy(P.t, T, sizeof P.t);
    HIP_TRY(c, launch_knn(c->mv_read, c->ax + q0, c->ay + q0, c->az + q0, n, P, d_max * d_max, k, di.p,
                          dd.p, dc.p, s, nullptr, c->cfg.force_kernel));
    HIP_TRY(c, hipMemcpyAsync(idx, di.p, n * (size_t)k * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(d2, dd.p, n * (size_t)k * sizeof(float), hipMemcpyDeviceToHost, s));
    if (count) HIP_TRY(c, hipMemcpyAsync(count, dc.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return VELO_OK;
}

int velo_knn_dev(velo_ctx* c, int frame, const double T[12], float d_max, int k, int32_t* d_idx, float* d_d2,
                 int32_t* d_count, uint64_t stats[4])
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    if (frame < 0 || frame >= c->n_frames) return c->fail(VELO_E_INVALID, "frame index out of range");
    if (!T || !d_idx || !d_d2) return c->fail(VELO_E_INVALID, "null argument");
    if (k < 1 || k > VELO_MAX_KNORMALS) return c->fail(VELO_E_INVALID, "k must be in [1,%d]", VELO_MAX_KNORMALS);
    if (!(d_max > 0.0f) || !(d_max <= c->mv_read.h))
        return c->fail(VELO_E_RANGE, "d_max must be in (0, voxel=%g]", (double)c->mv_read.h);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t q0 = (size_t)c->frame_start[frame], n = (size_t)c->frame_start[frame + 1] - q0;
    if (n == 0) return VELO_OK;
    // the pose travels as a kernel argument: nothing staged, nothing waited for -- the call only enqueues
    // (it used to drain the stream to reuse one pinned staging buffer)
    Pose12 P;
    std::memcpy(P.t, T, sizeof P.t);
    unsigned long long st[4] = {0, 0, 0, 0};
    HIP_TRY(c, launch_knn(c->mv_read, c->ax + q0, c->ay + q0, c->az + q0, n, P, d_max * d_max, k, d_idx, d_d2,
                          d_count, s, stats ? st : nullptr, c->cfg.force_kernel));
    if (stats)
        for (int i = 0; i < 4; ++i) stats[i] = st[i];
    return VELO_OK;
}

// ----------------------------------------------------------------- increment
// flags -> exclusive scan -> order-preserving scatter, all enqueued on the ctx stream; the
// count lands in *h_total (host memory the caller keeps alive until the stream reaches it)
static int enqueue_increment(velo_ctx* c, int frame, const double* d_pose, int min_count, float* tx,
                             float* ty, float* tz, uint32_t* h_total)
{
    hipStream_t s = c->stream;
    const size_t q0 = (size_t)c->frame_start[frame], n = (size_t)c->frame_start[frame + 1] - q0;
    HIP_TRY(c, c->inc_flags.reserve(n + 1));
    HIP_TRY(c, c->inc_offs.reserve(n + 1));
#ifndef VELO_INC_FUSED
#define VELO_INC_FUSED 1
#endif
    const size_t tiles = (n + kIncTilePoints - 1) / kIncTilePoints;
    if (VELO_INC_FUSED && tiles <= kIncFusedMaxTiles) {
        // a frame: two launches (flags + per-tile counts; bases + scatter) and the count's copy, instead of
        // flags, memset, two scan launches, count copy and scatter; offs doubles as [tile counts | total]
        HIP_TRY(c, c->inc_offs.reserve(tiles + 1));
        HIP_TRY(c, launch_increment_fused(c->ax + q0, c->ay + q0, c->az + q0, (uint32_t)n, c->mv_read, d_pose, min_count,
                                          c->inc_flags.p, c->inc_offs.p, tx, ty, tz, c->inc_offs.p + tiles, s));
        HIP_TRY(c, hipMemcpyAsync(h_total, c->inc_offs.p + tiles, sizeof *h_total, hipMemcpyDeviceToHost, s));
        return VELO_OK;
    }
    HIP_TRY(c, launch_increment_flags(c->ax + q0, c->ay + q0, c->az + q0, n, c->mv_read, d_pose, min_count,
                                      c->inc_flags.p, s));
    HIP_TRY(c, hipMemsetAsync(c->inc_flags.p + n, 0, sizeof(uint32_t), s));
    size_t tb = 0;
    HIP_TRY(c, exclusive_scan_u32(nullptr, tb, c->inc_flags.p, c->inc_offs.p, n + 1, s));
    if (int rc = ensure_temp(c, tb)) return rc;
    HIP_TRY(c, exclusive_scan_u32(c->temp.p, tb, c->inc_flags.p, c->inc_offs.p, n + 1, s));
    HIP_TRY(c, hipMemcpyAsync(h_total, c->inc_offs.p + n, sizeof *h_total, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, launch_increment_scatter(c->ax + q0, c->ay + q0, c->az + q0, n, d_pose, c->inc_flags.p,
                                        c->inc_offs.p, tx, ty, tz, s));
    return VELO_OK;
}

static int increment_impl(velo_ctx* c, int frame, const double T[12], int min_count, float* ox,
                          float* oy, float* oz, size_t* n_out, bool dev)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    if (frame < 0 || frame >= c->n_frames) return c->fail(VELO_E_INVALID, "frame index out of range");
    if (!T || !n_out) return c->fail(VELO_E_INVALID, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t n = (size_t)(c->frame_start[frame + 1] - c->frame_start[frame]);
    *n_out = 0;
    if (n == 0) return VELO_OK;
    HIP_TRY(c, c->inc_pose.reserve(12));
    HIP_TRY(c, hipMemcpyAsync(c->inc_pose.p, T, 12 * sizeof(double), hipMemcpyHostToDevice, s));
    float *tx = ox, *ty = oy, *tz = oz;
    if (!dev) {
        HIP_TRY(c, c->inc_x.reserve(n));
        HIP_TRY(c, c->inc_y.reserve(n));
        HIP_TRY(c, c->inc_z.reserve(n));
        tx = c->inc_x.p;
        ty = c->inc_y.p;
        tz = c->inc_z.p;
    }
    if (!tx || !ty || !tz) return c->fail(VELO_E_INVALID, "null output array");
    uint32_t total = 0;
    if (int rc = enqueue_increment(c, frame, c->inc_pose.p, min_count, tx, ty, tz, &total)) return rc;
    HIP_TRY(c, hipStreamSynchronize(s));
    if (!dev && total) {
        if (!ox || !oy || !oz) return c->fail(VELO_E_INVALID, "null output array");
        HIP_TRY(c, hipMemcpy(ox, tx, total * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(oy, ty, total * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(oz, tz, total * sizeof(float), hipMemcpyDeviceToHost));
    }
    *n_out = total;
    return VELO_OK;
}

// ---- pending increments: the accepted points of registered frames collected ON THE DEVICE until
// they are worth a map update (what MapManager::registerFrame's integrate step and a streaming
// host use: no host round trip per frame, no synchronisation per frame)
static int pending_resolve(velo_ctx* c, bool wait)
{
    if (!c->pend_outstanding) return VELO_OK;
    // wait == false never looks at the event: whether the increment in flight has landed yet must
    // not change what a caller does next (a flush decided by timing would make the map, and with
    // it every later registration, depend on the host's speed)
    if (!wait) return VELO_OK;
    HIP_TRY(c, hipEventSynchronize(c->ev_pend));
    c->pend_n += *c->h_pend_total;
    c->pend_outstanding = false;
    return VELO_OK;
}

int velo_increment_pending(velo_ctx* c, int frame, const double* T, int min_count)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    if (frame < 0 || frame >= c->n_frames) return c->fail(VELO_E_INVALID, "frame index out of range");
    if (!T && c->last_iters <= 0) return c->fail(VELO_E_INVALID, "no registration has run on the resident frames");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (int rc = pending_resolve(c, true)) return rc;  // the list's length decides where this one lands
    const size_t n = (size_t)(c->frame_start[frame + 1] - c->frame_start[frame]);
    if (n == 0) return VELO_OK;
    if (!c->h_pend_total) HIP_TRY(c, hipHostMalloc((void**)&c->h_pend_total, sizeof(uint32_t), 0));
    if (!c->ev_pend) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pend, hipEventDisableTiming));
    HIP_TRY(c, c->pend_x.reserve(c->pend_n + n, true, s));
    HIP_TRY(c, c->pend_y.reserve(c->pend_n + n, true, s));
    HIP_TRY(c, c->pend_z.reserve(c->pend_n + n, true, s));
    const double* d_pose = c->poses.p + 12 * (size_t)frame;  // where the last registration left it
    if (T) {
        HIP_TRY(c, c->inc_pose.reserve(12));
        // (pageable source: the copy is staged before the call returns)
        HIP_TRY(c, hipMemcpyAsync(c->inc_pose.p, T, 12 * sizeof(double), hipMemcpyHostToDevice, s));
        d_pose = c->inc_pose.p;
    }
    *c->h_pend_total = 0;
    if (int rc = enqueue_increment(c, frame, d_pose, min_count, c->pend_x.p + c->pend_n, c->pend_y.p + c->pend_n,
                                   c->pend_z.p + c->pend_n, c->h_pend_total))
        return rc;
    HIP_TRY(c, hipEventRecord(c->ev_pend, s));
    c->pend_outstanding = true;
    return VELO_OK;
}

int velo_pending_count(velo_ctx* c, size_t* n, int wait)
{
    if (!c || !n) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = pending_resolve(c, wait != 0)) return rc;
    *n = c->pend_n;
    return VELO_OK;
}

int velo_pending_fetch(velo_ctx* c, float* x, float* y, float* z, size_t cap, size_t* n_out)
{
    if (!c || !n_out) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = pending_resolve(c, true)) return rc;
    *n_out = c->pend_n;
    if (c->pend_n > cap) return c->fail(VELO_E_RANGE, "%zu pending points exceed the capacity %zu", c->pend_n, cap);
    if (c->pend_n == 0) return VELO_OK;
    if (!x || !y || !z) return c->fail(VELO_E_INVALID, "null output array");
    // Every entry of the list is complete (pending_resolve waited for the last increment's event): the copy runs on a
    // stream of its own through pinned memory.  On the ctx stream it would wait for whatever was enqueued since -- a
    // streaming host fetches frame k's increment while frame k + 1 registers (MapManager, pipelined integration).
    if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    const size_t n = c->pend_n;
    if (c->h_pend_stage_cap < 3 * n) {
        if (c->h_pend_stage) (void)hipHostFree(c->h_pend_stage);
        c->h_pend_stage = nullptr;
        c->h_pend_stage_cap = 0;
        const size_t want = 3 * n + 3 * n / 2 + 4096;
        HIP_TRY(c, hipHostMalloc((void**)&c->h_pend_stage, want * sizeof(float), 0));
        c->h_pend_stage_cap = want;
    }
    hipStream_t s = c->copy_stream;
    HIP_TRY(c, hipMemcpyAsync(c->h_pend_stage, c->pend_x.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(c->h_pend_stage + n, c->pend_y.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(c->h_pend_stage + 2 * n, c->pend_z.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    std::memcpy(x, c->h_pend_stage, n * sizeof(float));
    std::memcpy(y, c->h_pend_stage + n, n * sizeof(float));
    std::memcpy(z, c->h_pend_stage + 2 * n, n * sizeof(float));
    return VELO_OK;
}

int velo_pending_clear(velo_ctx* c)
{
    if (!c) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = pending_resolve(c, true)) return rc;  // (the increment in flight belongs to the list being dropped)
    c->pend_n = 0;
    return VELO_OK;
}

int velo_map_append_pending(velo_ctx* c, size_t* n_appended)
{
    if (!c) return VELO_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = pending_resolve(c, true)) return rc;
    if (n_appended) *n_appended = c->pend_n;
    if (c->pend_n == 0) return VELO_OK;
    const size_t n = c->pend_n;
    c->pend_n = 0;  // consumed either way: a failed append leaves the map as it was, the list is dropped
    return velo_map_append_dev(c, c->pend_x.p, c->pend_y.p, c->pend_z.p, n);
}

int velo_increment_registered_async(velo_ctx* c, int frame, int min_count, float* dox, float* doy,
                                    float* doz)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    if (frame < 0 || frame >= c->n_frames) return c->fail(VELO_E_INVALID, "frame index out of range");
    if (c->last_iters < 1) return c->fail(VELO_E_INVALID, "no registration has run");
    if (!dox || !doy || !doz) return c->fail(VELO_E_INVALID, "null output array");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->h_inc_total) {
        HIP_TRY(c, hipHostMalloc((void**)&c->h_inc_total, sizeof(uint32_t), 0));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_inc, hipEventDisableTiming));
    }
    *c->h_inc_total = 0;
    c->inc_pending = false;
    if (c->frame_start[frame + 1] > c->frame_start[frame]) {
        // the pose the registration left on the device: no host round trip before the increment
        if (int rc = enqueue_increment(c, frame, c->poses.p + 12 * (size_t)frame, min_count, dox, doy,
                                       doz, c->h_inc_total))
            return rc;
    }
    HIP_TRY(c, hipEventRecord(c->ev_inc, c->stream));
    c->inc_pending = true;
    return VELO_OK;
}

// every resident frame of the last registration at once ("every GPU contributes its
// accepted map increment"): concatenation, in frame order, of the per-frame increments
int velo_increment_all_registered_async(velo_ctx* c, int min_count, float* dox, float* doy, float* doz)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    if (c->n_frames < 1 || c->last_iters < 1) return c->fail(VELO_E_INVALID, "no registration has run");
    if (!dox || !doy || !doz) return c->fail(VELO_E_INVALID, "null output array");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->h_inc_total) {
        HIP_TRY(c, hipHostMalloc((void**)&c->h_inc_total, sizeof(uint32_t), 0));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_inc, hipEventDisableTiming));
    }
    *c->h_inc_total = 0;
    c->inc_pending = false;
    hipStream_t s = c->stream;
    const size_t n = (size_t)c->frame_start[c->n_frames];
    const int ni = (int)c->items_h.size();
    if (n) {
        HIP_TRY(c, c->inc_flags.reserve(n + 1));
        HIP_TRY(c, c->inc_offs.reserve(n + 1));
        FrameView fv{c->ax, c->ay, c->az, nullptr};
        HIP_TRY(c, launch_increment_flags_items(c->items.p, ni, fv, c->mv_read, c->poses.p, min_count, c->inc_flags.p, s));
        HIP_TRY(c, hipMemsetAsync(c->inc_flags.p + n, 0, sizeof(uint32_t), s));
        size_t tb = 0;
        HIP_TRY(c, exclusive_scan_u32(nullptr, tb, c->inc_flags.p, c->inc_offs.p, n + 1, s));
        if (int rc = ensure_temp(c, tb)) return rc;
        HIP_TRY(c, exclusive_scan_u32(c->temp.p, tb, c->inc_flags.p, c->inc_offs.p, n + 1, s));
        HIP_TRY(c, hipMemcpyAsync(c->h_inc_total, c->inc_offs.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, launch_increment_scatter_items(c->items.p, ni, fv, c->poses.p, c->inc_flags.p, c->inc_offs.p,
                                                  dox, doy, doz, s));
    }
    HIP_TRY(c, hipEventRecord(c->ev_inc, s));
    c->inc_pending = true;
    return VELO_OK;
}

int velo_increment_wait(velo_ctx* c, size_t* n_out)
{
    if (!c || !n_out) return VELO_E_INVALID;
    if (!c->inc_pending) return c->fail(VELO_E_INVALID, "no asynchronous increment is pending");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev_inc));  // only this event: later work keeps running
    c->inc_pending = false;
    *n_out = *c->h_inc_total;
    return VELO_OK;
}

int velo_increment(velo_ctx* c, int frame, const double T[12], int min_count, float* ox, float* oy,
                   float* oz, size_t* n_out)
{
    return increment_impl(c, frame, T, min_count, ox, oy, oz, n_out, false);
}
int velo_increment_dev(velo_ctx* c, int frame, const double T[12], int min_count, float* dox,
                       float* doy, float* doz, size_t* n_out)
{
    return increment_impl(c, frame, T, min_count, dox, doy, doz, n_out, true);
}

// ------------------------------------------------------------- multi-GPU exchange
// RCCL is opened at run time: the library has no link-time dependency on it (a single-GPU
// consumer never needs it), and inside a process that already carries an RCCL -- torch ships
// one under the same soname -- the loader hands back that copy.
namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load()
    {
        if (h) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            err = std::string("cannot open librccl: ") + (dlerror() ? dlerror() : "?");
            return false;
        }
        GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather || !GetErrorString) {
            err = "librccl lacks an expected symbol";
            h = nullptr;
            return false;
        }
        return true;
    }
} g_rccl;

#define RCCL_TRY(ctx, expr)                                                                    \
    do {                                                                                       \
        ncclResult_t r__ = (expr);                                                             \
        if (r__ != ncclSuccess)                                                                \
            return (ctx)->fail(VELO_E_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(r__)); \
    } while (0)
}  // namespace

int velo_comm_unique_id(uint8_t id[VELO_COMM_ID_BYTES])
{
    if (!id) return VELO_E_INVALID;
    if (!g_rccl.load()) {
        g_create_error = g_rccl.err;
        return VELO_E_DEVICE;
    }
    ncclUniqueId u;
    static_assert(sizeof u == VELO_COMM_ID_BYTES, "ncclUniqueId size");
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) {
        g_create_error = "ncclGetUniqueId failed";
        return VELO_E_DEVICE;
    }
    std::memcpy(id, &u, sizeof u);
    return VELO_OK;
}

int velo_comm_init(velo_ctx* c, const uint8_t id[VELO_COMM_ID_BYTES], int rank, int world)
{
    if (!c) return VELO_E_INVALID;
    if (!id || world < 1 || world > VELO_MAX_RANKS || rank < 0 || rank >= world)
        return c->fail(VELO_E_INVALID, "bad rank / world (1..%d) / id", VELO_MAX_RANKS);
    if (c->comm) return c->fail(VELO_E_INVALID, "communicator already initialised");
    if (!g_rccl.load()) return c->fail(VELO_E_DEVICE, "%s", g_rccl.err.c_str());
    HIP_TRY(c, hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    RCCL_TRY(c, g_rccl.CommInitRank(&c->comm, world, u, rank));
    c->comm_rank = rank;
    c->comm_world = world;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_comm_in, hipEventDisableTiming));
    HIP_TRY(c, c->comm_counts.reserve((size_t)world + 1));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_comm_counts, ((size_t)world + 1) * sizeof(int32_t), 0));
    return VELO_OK;
}

int velo_comm_destroy(velo_ctx* c)
{
    if (!c) return VELO_E_INVALID;
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    c->comm_stream = nullptr;
    if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
    if (c->ev_comm_in) (void)hipEventDestroy(c->ev_comm_in);
    c->ev_comm = c->ev_comm_in = nullptr;
    if (c->h_comm_counts) (void)hipHostFree(c->h_comm_counts);
    c->h_comm_counts = nullptr;
    c->comm_world = 1;
    c->comm_rank = 0;
    return VELO_OK;
}

// Host half of the exchange: counts -> offsets / pad / total (pure, no ctx: CPU-testable).
int velo_exchange_plan(const int32_t* counts, int world, uint32_t* offsets, size_t* pad, size_t* total)
{
    if (!counts || !offsets || world < 1 || world > VELO_MAX_RANKS) return VELO_E_INVALID;
    uint64_t run = 0;
    size_t widest = 1;  // a zero-length all-gather is not a collective every RCCL accepts
    for (int r = 0; r < world; ++r) {
        if (counts[r] < 0) return VELO_E_INVALID;
        offsets[r] = (uint32_t)run;
        run += (uint64_t)counts[r];
        if (run > 0xFFFFFFFFull) return VELO_E_RANGE;
        widest = std::max(widest, (size_t)counts[r]);
    }
    offsets[world] = (uint32_t)run;
    if (pad) *pad = widest;
    if (total) *total = (size_t)run;
    return VELO_OK;
}

namespace {
// device half: the W padded blocks -> ox/oy/oz in rank order, one launch on stream s
int pack_rank_blocks(velo_ctx* c, const float* recv, const int32_t* counts, int world, size_t pad,
                     float* ox, float* oy, float* oz, size_t cap, size_t* n_total, hipStream_t s)
{
    RankOffsets ro{};
    size_t need_pad = 0, total = 0;
    const int rc = velo_exchange_plan(counts, world, ro.off, &need_pad, &total);
    if (rc == VELO_E_RANGE) return c->fail(rc, "exchange: more than 2^32 - 1 points in one round");
    if (rc) return c->fail(rc, "exchange: bad counts / world (1..%d ranks, counts >= 0)", VELO_MAX_RANKS);
    ro.world = world;
    if (n_total) *n_total = total;
    if (pad < need_pad) return c->fail(VELO_E_INVALID, "exchange: blocks of %zu floats cannot hold a count of %zu", pad, need_pad);
    if (pad > 0xFFFFFFFFull / 3) return c->fail(VELO_E_RANGE, "exchange: block too large");
    if (total > cap) return c->fail(VELO_E_RANGE, "exchange: %zu points exceed the output capacity %zu", total, cap);
    if (total == 0) return VELO_OK;
    if (!recv || !ox || !oy || !oz) return c->fail(VELO_E_INVALID, "null argument");
    HIP_TRY(c, launch_pack_rank_blocks(recv, ro, (uint32_t)pad, ox, oy, oz, s));
    return VELO_OK;
}
}  // namespace

int velo_exchange_pack_dev(velo_ctx* c, const float* recv, const int32_t* counts, int world, size_t pad,
                           float* ox, float* oy, float* oz, size_t cap, size_t* n_total)
{
    if (!c) return VELO_E_INVALID;
    if (!counts) return c->fail(VELO_E_INVALID, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    return pack_rank_blocks(c, recv, counts, world, pad, ox, oy, oz, cap, n_total, c->stream);
}

// All-gather-v of the accepted map increments: counts first (one int per rank), then blocks
// padded to the largest count (a few hundred KB at most: latency-bound on the direct xGMI
// links, one padded all-gather beats world-1 ring steps of exact sizes), then the blocks are
// packed in RANK ORDER into ox/oy/oz so that every replica appends the same list.  Device work
// per call: one pack kernel before the second collective, one after it.
int velo_exchange_increments(velo_ctx* c, const float* dx, const float* dy, const float* dz, size_t n_local,
                             int after_async_increment, float* ox, float* oy, float* oz, size_t cap,
                             int32_t* counts, size_t* n_total)
{
    if (!c) return VELO_E_INVALID;
    if (!c->comm) return c->fail(VELO_E_INVALID, "velo_comm_init has not been called");
    if ((n_local && (!dx || !dy || !dz)) || !ox || !oy || !oz || !n_total)
        return c->fail(VELO_E_INVALID, "null argument");
    if (n_local >= (size_t)INT32_MAX) return c->fail(VELO_E_RANGE, "increment too large");
    HIP_TRY(c, hipSetDevice(c->device));
    const int W = c->comm_world;
    hipStream_t cs = c->comm_stream;
    // the inputs are ready once the producing work on the ctx stream is: either the last
    // asynchronous increment (its event), or everything enqueued so far
    if (after_async_increment && c->ev_inc) {
        HIP_TRY(c, hipStreamWaitEvent(cs, c->ev_inc, 0));
    } else {
        HIP_TRY(c, hipEventRecord(c->ev_comm_in, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(cs, c->ev_comm_in, 0));
    }
    c->h_comm_counts[W] = (int32_t)n_local;
    HIP_TRY(c, hipMemcpyAsync(c->comm_counts.p + W, c->h_comm_counts + W, sizeof(int32_t),
                              hipMemcpyHostToDevice, cs));
    RCCL_TRY(c, g_rccl.AllGather(c->comm_counts.p + W, c->comm_counts.p, 1, ncclInt32, c->comm, cs));
    HIP_TRY(c, hipMemcpyAsync(c->h_comm_counts, c->comm_counts.p, (size_t)W * sizeof(int32_t),
                              hipMemcpyDeviceToHost, cs));
    HIP_TRY(c, hipStreamSynchronize(cs));  // the counts size the second phase
    uint32_t offs[VELO_MAX_RANKS + 1];
    size_t pad = 1, total = 0;
    if (int rc = velo_exchange_plan(c->h_comm_counts, W, offs, &pad, &total))
        return c->fail(rc == VELO_E_RANGE ? VELO_E_RANGE : VELO_E_DEVICE,
                       "exchange: the gathered counts are not a valid plan (negative, or beyond 2^32 - 1 points)");
    for (int r = 0; r < W && counts; ++r) counts[r] = c->h_comm_counts[r];
    *n_total = total;
    if (total == 0) return VELO_OK;  // the same decision on every rank: the counts are identical
    // From here to the second collective nothing may depend on rank-local state (cap, the
    // output pointers): a rank that returned early would leave the others inside the all-gather.
    HIP_TRY(c, c->comm_send.reserve(3 * pad));
    HIP_TRY(c, c->comm_recv.reserve(3 * pad * (size_t)W));
    HIP_TRY(c, launch_pack_send(dx, dy, dz, (uint32_t)n_local, (uint32_t)pad, c->comm_send.p, cs));
    RCCL_TRY(c, g_rccl.AllGather(c->comm_send.p, c->comm_recv.p, 3 * pad, ncclFloat32, c->comm, cs));
    if (int rc = pack_rank_blocks(c, c->comm_recv.p, c->h_comm_counts, W, pad, ox, oy, oz, cap, nullptr, cs))
        return rc;
    // work enqueued on the ctx stream from now on (velo_map_append_dev) sees the blocks
    HIP_TRY(c, hipEventRecord(c->ev_comm, cs));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_comm, 0));
    return VELO_OK;
}

int velo_comm_info(velo_ctx* c, int32_t* rank, int32_t* world)
{
    if (!c) return VELO_E_INVALID;
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm ? c->comm_world : 0;
    return VELO_OK;
}

}  // extern "C"
