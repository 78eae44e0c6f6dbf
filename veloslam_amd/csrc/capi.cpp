// capi.cpp -- C-ABI glue of libveloslam_amd.so (include/velo.h): device memory,
// stream-ordered launches of the gfx950 kernels, result marshalling.  No compute
// happens on the host here and there is no CPU fallback: a missing GPU or a HIP
// error surfaces as a negative return code plus velo_last_error().
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <memory>
#include "velo_internal.hpp"

using namespace velo;

namespace {

std::string g_create_error;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
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
    DevBuf<unsigned> mm_scratch;
    DevBuf<unsigned long long> invalid_cnt;
    DevBuf<char> temp;
    MapView mv{};
    bool has_map = false;
    velo_map_info info{};

    // ---- frames
    DevBuf<float> fx, fy, fz;
    const float *ax = nullptr, *ay = nullptr, *az = nullptr;  // active (owned or adopted)
    int n_frames = 0;
    std::vector<int64_t> frame_start;
    DevBuf<int64_t> d_frame_start;
    std::vector<BlockItem> items_h;   // frame-major
    std::vector<int32_t> fbs_h;
    DevBuf<BlockItem> items;
    DevBuf<BlockItem> items_xcd;      // same blocks, dealt so that XCD r works on spatial slab r
    DevBuf<float> sx, sy, sz;         // cell-sorted copies of the frames (cfg.sort_frames)
    DevBuf<int32_t> fbs;
    DevBuf<double> poses, partials, acc;
    DevBuf<velo_icp_iter> stats;
    DevBuf<uint32_t> order_keys, order_keys2, order_idx, order;
    DevBuf<int32_t> corr;
    DevBuf<int32_t> hint;  // last correspondence per query slot (search-radius hint), -1 = none
    DevBuf<float> d2;
    DevBuf<uint32_t> flags, offs;
    DevBuf<float> inc_x, inc_y, inc_z;
    int last_iters = 0;
    // hipGraph replay of the per-registration launch sequence (cfg.use_graph)
    hipGraphExec_t graph_exec = nullptr;
    struct GraphKey {
        int iters = 0, ni = 0, n_frames = 0, variant = 0;
        float dmax2 = 0;
        const void *hint = nullptr, *items = nullptr, *stream = nullptr;
        uint64_t map_gen = 0, frames_gen = 0;
        bool operator==(const GraphKey& o) const
        {
            return iters == o.iters && ni == o.ni && n_frames == o.n_frames && variant == o.variant &&
                   dmax2 == o.dmax2 && hint == o.hint && items == o.items && stream == o.stream &&
                   map_gen == o.map_gen && frames_gen == o.frames_gen;
        }
    } graph_key;
    uint64_t map_gen = 0, frames_gen = 0;
    double* h_T0 = nullptr;  // pinned staging of the initial poses (stable address for the graph)
    bool lin_hints = false;  // velo_linearize keeps/uses hints across calls (tests)

    // ---- timing
    bool timing = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    std::vector<int> ev_kind;  // per pair: 0 linearise, 1 solve
    hipEvent_t ev_call0 = nullptr, ev_call1 = nullptr;
    double last_timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};

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

int ensure_temp(velo_ctx* c, size_t bytes)
{
    HIP_TRY(c, c->temp.reserve(bytes));
    return VELO_OK;
}

// (re)build the voxel grid over raw_x/y/z[0..raw_n)
int rebuild_map(velo_ctx* c, float voxel, int k_normals)
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
    double ncell_d = 1.0;
    for (int a = 0; a < 3; ++a) {
        if (!std::isfinite(mm.mn[a]) || !std::isfinite(mm.mx[a]))
            return c->fail(VELO_E_INVALID, "map points must be finite");
        const float ext = floorf((mm.mx[a] - mm.mn[a]) * inv_h);
        if (!(ext < 2.0e9f)) return c->fail(VELO_E_RANGE, "map extent / voxel too large");
        dims[a] = (int)ext + 1;
        ncell_d *= (double)dims[a];
    }
    const int S = c->cfg.map_subdiv;
    ncell_d *= (double)S * S * S;
    if (ncell_d >= 2147483648.0)
        return c->fail(VELO_E_RANGE, "dense fine-cell grid of %.3g cells (voxels x %d^3) exceeds 2^31",
                       ncell_d, S);
    const int fdims[3] = {dims[0] * S, dims[1] * S, dims[2] * S};
    const size_t ncell = (size_t)fdims[0] * fdims[1] * fdims[2];
    HIP_TRY(c, c->keys.reserve(n));
    HIP_TRY(c, c->keys_sorted.reserve(n));
    HIP_TRY(c, c->idx.reserve(n));
    HIP_TRY(c, c->perm.reserve(n));
    HIP_TRY(c, c->pts.reserve(n));
    HIP_TRY(c, c->nrm.reserve(n));
    HIP_TRY(c, c->cell_start.reserve(ncell + 8));  // +1 entry, padded: rows are read 4 entries at a time
    HIP_TRY(c, c->invalid_cnt.reserve(1));
    HIP_TRY(c, launch_keys(c->raw_x.p, c->raw_y.p, c->raw_z.p, n, mm.mn[0], mm.mn[1], mm.mn[2],
                           inv_h, S, fdims[0], fdims[1], c->keys.p, c->idx.p, s));
    int bits = 1;
    while (bits < 32 && ((size_t)1 << bits) < ncell) ++bits;
    size_t tb = 0;
    HIP_TRY(c, sort_pairs(nullptr, tb, c->keys.p, c->keys_sorted.p, c->idx.p, c->perm.p, n, bits, s));
    if (int rc = ensure_temp(c, tb)) return rc;
    HIP_TRY(c, sort_pairs(c->temp.p, tb, c->keys.p, c->keys_sorted.p, c->idx.p, c->perm.p, n, bits, s));
    HIP_TRY(c, launch_gather(c->raw_x.p, c->raw_y.p, c->raw_z.p, c->perm.p, n, c->pts.p, s));
    HIP_TRY(c, launch_cell_start(c->keys_sorted.p, n, ncell, c->cell_start.p, s));
    MapView mv;
    mv.pts = c->pts.p;
    mv.nrm = c->nrm.p;
    mv.cell_start = c->cell_start.p;
    mv.ox = mm.mn[0];
    mv.oy = mm.mn[1];
    mv.oz = mm.mn[2];
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
    if (k_normals > 0) {
        HIP_TRY(c, launch_normals(mv, k_normals, c->nrm.p, c->invalid_cnt.p, s));
        HIP_TRY(c, hipMemcpyAsync(&invalid, c->invalid_cnt.p, sizeof invalid, hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(c, hipMemsetAsync(c->nrm.p, 0, n * sizeof(float4), s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    c->mv = mv;
    c->has_map = true;
    ++c->map_gen;
    c->info.n_points = n;
    c->info.n_cells = ncell;
    c->info.origin[0] = mv.ox;
    c->info.origin[1] = mv.oy;
    c->info.origin[2] = mv.oz;
    c->info.voxel = voxel;
    c->info.inv_voxel = inv_h;
    c->info.dims[0] = dims[0];
    c->info.dims[1] = dims[1];
    c->info.dims[2] = dims[2];
    c->info.k_normals = k_normals;
    c->info.subdiv = S;
    c->info.n_invalid_normals = invalid;
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

// work decomposition of the linearise kernel over the resident frames
int plan_frames(velo_ctx* c, int n_frames, const int64_t* frame_start)
{
    if (n_frames < 1) return c->fail(VELO_E_INVALID, "n_frames must be >= 1");
    const int maxb = c->cfg.max_batch;
    if (n_frames > maxb) return c->fail(VELO_E_RANGE, "n_frames %d > cfg.max_batch %d", n_frames, maxb);
    if (!frame_start || frame_start[0] != 0) return c->fail(VELO_E_INVALID, "frame_start[0] must be 0");
    for (int f = 0; f < n_frames; ++f)
        if (frame_start[f + 1] < frame_start[f]) return c->fail(VELO_E_INVALID, "frame_start must ascend");
    if (frame_start[n_frames] >= INT32_MAX) return c->fail(VELO_E_RANGE, "too many query points");
    c->n_frames = n_frames;
    ++c->frames_gen;
    c->frame_start.assign(frame_start, frame_start + n_frames + 1);
    c->items_h.clear();
    c->fbs_h.assign((size_t)n_frames + 1, 0);
    // rounds of 256 queries per workgroup.  Measured on config 2 (16 x 115 200 queries): 1 round
    // = 60 us per launch, 2 = 66, 4 = 70 -- more, smaller workgroups balance better than the
    // start-up they cost, so the default is one round; cfg.rounds_per_block overrides.
    int rounds = 1;
    if (c->cfg.rounds_per_block > 0) rounds = std::min(c->cfg.rounds_per_block, 64);
    const int per_block = kLinThreads * rounds;
    for (int f = 0; f < n_frames; ++f) {
        c->fbs_h[f] = (int32_t)c->items_h.size();
        for (int64_t q = frame_start[f]; q < frame_start[f + 1]; q += per_block) {
            BlockItem it;
            it.frame = f;
            it.q0 = (int32_t)q;
            it.q1 = (int32_t)std::min<int64_t>(q + per_block, frame_start[f + 1]);
            it.slot = (int32_t)c->items_h.size();
            c->items_h.push_back(it);
        }
    }
    c->fbs_h[n_frames] = (int32_t)c->items_h.size();
    const size_t ni = c->items_h.size();
    HIP_TRY(c, c->items.reserve(std::max<size_t>(ni, 1)));
    HIP_TRY(c, c->fbs.reserve((size_t)n_frames + 1));
    HIP_TRY(c, c->d_frame_start.reserve((size_t)n_frames + 1));
    HIP_TRY(c, c->partials.reserve(std::max<size_t>(ni, 1) * kAccStride));
    HIP_TRY(c, c->poses.reserve((size_t)maxb * 12));
    HIP_TRY(c, c->acc.reserve((size_t)maxb * kAccStride));
    HIP_TRY(c, c->stats.reserve((size_t)maxb * VELO_MAX_ITERS));
    if (ni)
        HIP_TRY(c, hipMemcpyAsync(c->items.p, c->items_h.data(), ni * sizeof(BlockItem),
                                  hipMemcpyHostToDevice, c->stream));
    std::vector<BlockItem> xcd;
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
        HIP_TRY(c, c->items_xcd.reserve(ni));
        HIP_TRY(c, hipMemcpyAsync(c->items_xcd.p, xcd.data(), ni * sizeof(BlockItem),
                                  hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(c->fbs.p, c->fbs_h.data(), ((size_t)n_frames + 1) * sizeof(int32_t),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_frame_start.p, c->frame_start.data(),
                              ((size_t)n_frames + 1) * sizeof(int64_t), hipMemcpyHostToDevice,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // host vectors may be reused
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
    const double span = (double)c->n_frames * ((double)c->mv.nx * c->mv.ny * c->mv.nz + 1.0);
    if (n == 0 || span >= 4294967296.0) return VELO_OK;  // composite key would not fit 32 bits
    HIP_TRY(c, c->order_keys.reserve(n));
    HIP_TRY(c, c->order_keys2.reserve(n));
    HIP_TRY(c, c->order_idx.reserve(n));
    HIP_TRY(c, c->order.reserve(n));
    HIP_TRY(c, launch_frame_cellkeys(fv, c->d_frame_start.p, c->n_frames, n, c->mv, c->poses.p,
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

int run_icp(velo_ctx* c, const double* T0, int iters, float d_max)
{
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map: call velo_map_reset first");
    if (c->n_frames < 1) return c->fail(VELO_E_INVALID, "no resident frames: call velo_frames_upload");
    if (!T0) return c->fail(VELO_E_INVALID, "T0 is null");
    if (iters < 1 || iters > VELO_MAX_ITERS)
        return c->fail(VELO_E_INVALID, "iters must be in [1,%d]", VELO_MAX_ITERS);
    if (!(d_max > 0.0f) || !(d_max <= c->mv.h))
        return c->fail(VELO_E_RANGE, "d_max must be in (0, voxel=%g]", (double)c->mv.h);
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
    if (c->cfg.use_hints && n_all) {
        HIP_TRY(c, c->hint.reserve(n_all));
        hint = c->hint.p;
    }
    const bool graph_ok = c->cfg.use_graph && !c->timing && !c->cfg.sort_frames;
    if (graph_ok) {
        // Replay the whole registration (pose upload, hint reset, iters x (linearise, solve)) as
        // one hipGraph: the kernels are tens of microseconds long, so per-launch host cost and
        // inter-kernel gaps are a visible share of an iteration.
        if (!c->h_T0) HIP_TRY(c, hipHostMalloc((void**)&c->h_T0, (size_t)c->cfg.max_batch * 12 * sizeof(double), 0));
        velo_ctx::GraphKey key;
        key.iters = iters;
        key.ni = ni;
        key.n_frames = c->n_frames;
        key.variant = c->cfg.linearize_variant;
        key.dmax2 = dmax2;
        key.hint = hint;
        key.items = c->items.p;
        key.stream = s;
        key.map_gen = c->map_gen;
        key.frames_gen = c->frames_gen;
        if (!c->graph_exec || !(key == c->graph_key)) {
            if (c->graph_exec) {
                (void)hipGraphExecDestroy(c->graph_exec);
                c->graph_exec = nullptr;
            }
            HIP_TRY(c, hipStreamSynchronize(s));
            HIP_TRY(c, hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            hipError_t e = hipMemcpyAsync(c->poses.p, c->h_T0, pose_bytes, hipMemcpyHostToDevice, s);
            if (e == hipSuccess && hint) e = hipMemsetAsync(hint, 0xFF, n_all * sizeof(int32_t), s);
            FrameView fv{c->ax, c->ay, c->az, nullptr};
            for (int it = 0; it < iters && e == hipSuccess; ++it) {
                e = launch_linearize(c->cfg.linearize_variant, c->items.p, ni, fv, c->mv, c->poses.p,
                                     dmax2, c->partials.p, nullptr, nullptr, hint, s);
                if (e == hipSuccess)
                    e = launch_reduce_solve(c->partials.p, c->fbs.p, c->n_frames, c->poses.p,
                                            c->stats.p, it, iters, nullptr, 1, s);
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
        // the previous replay may still be reading h_T0
        HIP_TRY(c, hipStreamSynchronize(s));
        std::memcpy(c->h_T0, T0, pose_bytes);
        HIP_TRY(c, hipGraphLaunch(c->graph_exec, s));
        c->last_iters = iters;
        return VELO_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(c->poses.p, T0, pose_bytes, hipMemcpyHostToDevice, s));
    FrameView fv{c->ax, c->ay, c->az, nullptr};
    if (int rc = maybe_sort_frames(c, fv)) return rc;
    // hints never outlive a registration: results do not depend on earlier calls
    if (hint) HIP_TRY(c, hipMemsetAsync(hint, 0xFF, n_all * sizeof(int32_t), s));
    for (int it = 0; it < iters; ++it) {
        {
            Timed t(c, 0);
            HIP_TRY(c, launch_linearize(c->cfg.linearize_variant,
                                        (fv.order && c->cfg.sort_frames == 1) ? c->items_xcd.p : c->items.p, ni, fv, c->mv,
                                        c->poses.p, dmax2, c->partials.p, nullptr, nullptr, hint, s));
        }
        {
            Timed t(c, 1);
            HIP_TRY(c, launch_reduce_solve(c->partials.p, c->fbs.p, c->n_frames, c->poses.p,
                                           c->stats.p, it, iters, nullptr, 1, s));
        }
    }
    if (c->timing) HIP_TRY(c, hipEventRecord(c->ev_call1, s));
    c->last_iters = iters;
    return VELO_OK;
}

int fetch_icp(velo_ctx* c, velo_icp_result* out)
{
    if (!out) return c->fail(VELO_E_INVALID, "out is null");
    if (c->last_iters < 1) return c->fail(VELO_E_INVALID, "no registration has run");
    const int F = c->n_frames, iters = c->last_iters;
    std::vector<double> T((size_t)F * 12);
    std::vector<velo_icp_iter> st((size_t)F * VELO_MAX_ITERS);
    HIP_TRY(c, hipMemcpyAsync(T.data(), c->poses.p, T.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(st.data(), c->stats.p, st.size() * sizeof(velo_icp_iter),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
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
    if (c->timing) {
        double lin = 0, sol = 0, lin_first = 0, lin_min = 1e30;
        int nl = 0, ns = 0;
        for (size_t k = 0; k < c->ev_kind.size(); ++k) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, c->ev[2 * k], c->ev[2 * k + 1]) != hipSuccess) continue;
            if (c->ev_kind[k] == 0) {
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
    std::memset(&c->cfg, 0, sizeof c->cfg);
    if (cfg) std::memcpy(&c->cfg, cfg, std::min<size_t>(cfg->struct_size, sizeof(velo_cfg)));
    c->cfg.struct_size = sizeof(velo_cfg);
    if (c->cfg.max_batch <= 0) c->cfg.max_batch = 64;
    if (!cfg) {
        c->cfg.use_hints = 1;
        c->cfg.use_graph = 1;
    }
    if (c->cfg.map_subdiv <= 0) c->cfg.map_subdiv = 3;
    if (c->cfg.map_subdiv > 16) c->cfg.map_subdiv = 16;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        return nullptr;
    }
    c->stream = c->own_stream;
    return c.release();
}

void velo_destroy(velo_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->h_T0) (void)hipHostFree(c->h_T0);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->ev_call0) (void)hipEventDestroy(c->ev_call0);
    if (c->ev_call1) (void)hipEventDestroy(c->ev_call1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int velo_set_stream(velo_ctx* c, void* hip_stream)
{
    if (!c) return VELO_E_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return VELO_OK;
}

int velo_synchronize(velo_ctx* c)
{
    if (!c) return VELO_E_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VELO_OK;
}

int velo_linearize_hints(velo_ctx* c, int mode)
{
    if (!c) return VELO_E_INVALID;
    c->lin_hints = mode != 0;
    if (c->hint.p && c->hint.cap)  // (re)start from "no hint"
        HIP_TRY(c, hipMemsetAsync(c->hint.p, 0xFF, c->hint.cap * sizeof(int32_t), c->stream));
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
    c->has_map = false;
    if (n == 0) return c->fail(VELO_E_INVALID, "map needs at least one point");
    if (int rc = stage_raw(c, x, y, z, n, dev, false)) return rc;
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
    if (int rc = stage_raw(c, x, y, z, n, dev, true)) return rc;
    return rebuild_map(c, c->info.voxel, c->info.k_normals);
}
int velo_map_append(velo_ctx* c, const float* x, const float* y, const float* z, size_t n)
{
    return map_append_impl(c, x, y, z, n, false);
}
int velo_map_append_dev(velo_ctx* c, const float* x, const float* y, const float* z, size_t n)
{
    return map_append_impl(c, x, y, z, n, true);
}

int velo_map_info_get(velo_ctx* c, velo_map_info* out)
{
    if (!c || !out) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    *out = c->info;
    return VELO_OK;
}

int velo_map_download(velo_ctx* c, float* x, float* y, float* z, float* nx, float* ny, float* nz,
                      int32_t* perm, int32_t* cell_start)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    const size_t n = c->info.n_points;
    std::vector<float4> h(n);
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
    if (cell_start)
        HIP_TRY(c, hipMemcpyAsync(cell_start, c->cell_start.p, (c->info.n_cells + 1) * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, c->stream));
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
    if (!(d_max > 0.0f) || !(d_max <= c->mv.h))
        return c->fail(VELO_E_RANGE, "d_max must be in (0, voxel=%g]", (double)c->mv.h);
    hipStream_t s = c->stream;
    const size_t n_all = (size_t)c->frame_start[c->n_frames];
    const size_t q0 = (size_t)c->frame_start[frame], q1 = (size_t)c->frame_start[frame + 1];
    HIP_TRY(c, c->corr.reserve(std::max<size_t>(n_all, 1)));
    HIP_TRY(c, c->d2.reserve(std::max<size_t>(n_all, 1)));
    if (c->lin_hints && c->hint.cap < n_all) {
        HIP_TRY(c, c->hint.reserve(n_all));
        HIP_TRY(c, hipMemsetAsync(c->hint.p, 0xFF, n_all * sizeof(int32_t), s));
    }
    // poses of the other frames are irrelevant here: only this frame's blocks are launched
    HIP_TRY(c, hipMemcpyAsync(c->poses.p + 12 * (size_t)frame, T, 12 * sizeof(double),
                              hipMemcpyHostToDevice, s));
    FrameView fv{c->ax, c->ay, c->az, nullptr};
    const int b0 = c->fbs_h[frame], b1 = c->fbs_h[frame + 1];
    HIP_TRY(c, launch_linearize(c->cfg.linearize_variant, c->items.p + b0, b1 - b0, fv, c->mv,
                                c->poses.p, d_max * d_max, c->partials.p, c->corr.p, c->d2.p,
                                c->lin_hints ? c->hint.p : nullptr, s));
    HIP_TRY(c, launch_reduce_solve(c->partials.p, c->fbs.p + frame, 1, c->poses.p, nullptr, 0, 1,
                                   c->acc.p, 0, s));
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

// ----------------------------------------------------------------------- kNN
int velo_knn(velo_ctx* c, int frame, const double T[12], float d_max, int k, int32_t* idx,
             float* d2, int32_t* count)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    if (frame < 0 || frame >= c->n_frames) return c->fail(VELO_E_INVALID, "frame index out of range");
    if (!T || !idx || !d2) return c->fail(VELO_E_INVALID, "null argument");
    if (k < 1 || k > VELO_MAX_KNORMALS) return c->fail(VELO_E_INVALID, "k must be in [1,%d]", VELO_MAX_KNORMALS);
    if (!(d_max > 0.0f) || !(d_max <= c->mv.h))
        return c->fail(VELO_E_RANGE, "d_max must be in (0, voxel=%g]", (double)c->mv.h);
    hipStream_t s = c->stream;
    const size_t q0 = (size_t)c->frame_start[frame], n = (size_t)c->frame_start[frame + 1] - q0;
    if (n == 0) return VELO_OK;
    DevBuf<int32_t> di, dc;
    DevBuf<float> dd;
    DevBuf<double> dT;
    HIP_TRY(c, di.reserve(n * (size_t)k));
    HIP_TRY(c, dd.reserve(n * (size_t)k));
    HIP_TRY(c, dc.reserve(n));
    HIP_TRY(c, dT.reserve(12));
    HIP_TRY(c, hipMemcpyAsync(dT.p, T, 12 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, launch_knn(c->mv, c->ax + q0, c->ay + q0, c->az + q0, n, dT.p, d_max * d_max, k, di.p,
                          dd.p, dc.p, s));
    HIP_TRY(c, hipMemcpyAsync(idx, di.p, n * (size_t)k * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(d2, dd.p, n * (size_t)k * sizeof(float), hipMemcpyDeviceToHost, s));
    if (count) HIP_TRY(c, hipMemcpyAsync(count, dc.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return VELO_OK;
}

// ----------------------------------------------------------------- increment
static int increment_impl(velo_ctx* c, int frame, const double T[12], int min_count, float* ox,
                          float* oy, float* oz, size_t* n_out, bool dev)
{
    if (!c) return VELO_E_INVALID;
    if (!c->has_map) return c->fail(VELO_E_NOMAP, "no map");
    if (frame < 0 || frame >= c->n_frames) return c->fail(VELO_E_INVALID, "frame index out of range");
    if (!T || !n_out) return c->fail(VELO_E_INVALID, "null argument");
    hipStream_t s = c->stream;
    const size_t q0 = (size_t)c->frame_start[frame], n = (size_t)c->frame_start[frame + 1] - q0;
    *n_out = 0;
    if (n == 0) return VELO_OK;
    HIP_TRY(c, c->flags.reserve(n + 1));
    HIP_TRY(c, c->offs.reserve(n + 1));
    DevBuf<double> dT;
    HIP_TRY(c, dT.reserve(12));
    HIP_TRY(c, hipMemcpyAsync(dT.p, T, 12 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, launch_increment_flags(c->ax + q0, c->ay + q0, c->az + q0, n, c->mv, dT.p, min_count,
                                      c->flags.p, s));
    HIP_TRY(c, hipMemsetAsync(c->flags.p + n, 0, sizeof(uint32_t), s));
    size_t tb = 0;
    HIP_TRY(c, exclusive_scan_u32(nullptr, tb, c->flags.p, c->offs.p, n + 1, s));
    if (int rc = ensure_temp(c, tb)) return rc;
    HIP_TRY(c, exclusive_scan_u32(c->temp.p, tb, c->flags.p, c->offs.p, n + 1, s));
    uint32_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->offs.p + n, sizeof total, hipMemcpyDeviceToHost, s));
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
    HIP_TRY(c, launch_increment_scatter(c->ax + q0, c->ay + q0, c->az + q0, n, dT.p, c->flags.p,
                                        c->offs.p, tx, ty, tz, s));
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

}  // extern "C"
