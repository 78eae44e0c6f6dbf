// stream_driver.cpp -- BASELINE configs[2] from C++, through include/veloslam/*.hpp and the C ABI
// they sit on: a recorded drive (veloslam_amd/drive.py layout: drive.pcap, carposes.txt, db.xml,
// world.map, truth.txt) is replayed frame by frame against a rolling device map --
//
//   HDLManager::setCalibFile + loadOffline    db.xml; the pose track, the capture and its frame index -> one
//                                             stub per revolution (HDLManager.cxx:98-112)
//   MapManager::load                          the accumulated map, as tiles
//   per frame: HDLManager::prepareResident (the frame's packets in, decoded + compensated frame
//              resident in HBM) -> MapManager::registerResident (rolls the device map to the
//              prior's ROI: evict the tiles that left, append the ones that entered; 20 ICP
//              iterations; accepted increment to the device-side pending list, merged every
//              append_threshold points); the NEXT frame's decode (HDLManager::prepareResidentDuringRegistration)
//              is issued inside registerResident (RegisterOptions::while_registering) and runs on the context's
//              second stream, concurrently with this frame's registration
//
// The prior is what the reference's INS would give: the interpolated car pose (x, y, angles from
// carposes.txt -- the format has no z: z is carried from the previous registration) plus the
// perturbation the bench uses.  Frames are played forwards, then backwards (like bench.py's stream
// record), so that the pose never jumps and any number of steps can be timed.
//
//   hipcc -std=c++17 -O2 -x c++ tools/stream_driver.cpp -Iinclude -Lveloslam_amd/csrc -lveloslam_amd \
//         -Wl,-rpath,$PWD/veloslam_amd/csrc -o tools/stream_driver
//   tools/stream_driver DIR [--steps 100] [--warmup 10] [--threshold 512] [--roll-lead 4] [--no-integrate] [--no-overlap] [--no-roll-ahead]
//
// --mapping: configs[2] as SLAM (README.md:25 "[ ] Implement various SLAM algorithms"; MapManager.h:13,43).  No world.map is
// read: the map is SEEDED with the first frame at its prior and grows only from the accepted increments of the frames
// registered against it (RegisterOptions::integrate, increments_in_roi_only); tiles further than ROI_RANGE behind the car
// leave the device for the host tiles and come back when the car does.  By default the increments are integrated in
// pipeline (RegisterOptions::pipeline_increments: frame k's increment joins the device map beside frame k + 1's
// registration, together with the move of the tile rectangle to frame k + 2's prior); --no-pipeline integrates every
// frame's increment before the next frame is registered (host-synchronous append).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <unistd.h>
#include "veloslam/HDLManager.hpp"
#include "veloslam/MapManager.hpp"
#include "veloslam/TransformManager.hpp"

using namespace veloslam;
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }

// ---- watchdog (--watchdog SECONDS, default 30; 0 = none): a stream host that makes no progress for that long says where
// it stands -- frame, phase, and the main thread's call stack (the velo_* entry point and the HIP call under it) -- and
// exits with code 7 instead of hanging until somebody's timeout kills it silently.
static std::atomic<unsigned> g_beat{0};
static std::atomic<const char*> g_phase{"setup"};
static std::atomic<int> g_frame{-1};
static void beat(const char* phase, int frame = -2)
{
    g_phase.store(phase, std::memory_order_relaxed);
    if (frame != -2) g_frame.store(frame, std::memory_order_relaxed);
    g_beat.fetch_add(1, std::memory_order_relaxed);
}
static void on_stall_signal(int)
{
    void* bt[64];
    const int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    _exit(7);
}
static void start_watchdog(int seconds)
{
    if (seconds <= 0) return;
    struct sigaction sa;
    std::memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_stall_signal;
    sigaction(SIGUSR1, &sa, nullptr);
    const pthread_t main_thread = pthread_self();
    std::thread([seconds, main_thread] {
        unsigned last = g_beat.load();
        int still = 0;
        for (;;) {
            std::this_thread::sleep_for(std::chrono::milliseconds(500));
            const unsigned now = g_beat.load();
            still = now == last ? still + 1 : 0;
            last = now;
            if (still >= 2 * seconds) {
                std::fprintf(stderr, "stream_driver: NO PROGRESS for %d s at frame %d in %s; the main thread's stack:\n", seconds,
                             g_frame.load(), g_phase.load());
                pthread_kill(main_thread, SIGUSR1);
                std::this_thread::sleep_for(std::chrono::seconds(3));
                _exit(7);
            }
        }
    }).detach();
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: stream_driver DIR [--steps N] [--warmup W] [--threshold T] [--no-integrate]\n");
        return 2;
    }
    const std::string dir = argv[1];
    int steps = 100, warmup = 10, threshold = 512, roll_lead = 4;
    bool integrate = true, overlap = true, roll_ahead = true, mapping = false, pipeline = true;
    std::string per_frame;
    int margin = -1, min_count = -1, watchdog = 30;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--steps" && i + 1 < argc) steps = std::atoi(argv[++i]);
        else if (a == "--warmup" && i + 1 < argc) warmup = std::atoi(argv[++i]);
        else if (a == "--threshold" && i + 1 < argc) threshold = std::atoi(argv[++i]);
        else if (a == "--roll-lead" && i + 1 < argc) roll_lead = std::atoi(argv[++i]);  // frames a roll is begun ahead (0: beside the previous frame only)
        else if (a == "--margin" && i + 1 < argc) margin = std::atoi(argv[++i]);  // grid slack in x / y, voxels (MapManager's default: 16)
        else if (a == "--per-frame" && i + 1 < argc) per_frame = argv[++i];  // per-frame wall time + what the map did, one line each
        else if (a == "--no-integrate") integrate = false;
        else if (a == "--watchdog" && i + 1 < argc) watchdog = std::atoi(argv[++i]);
        else if (a == "--mapping") mapping = true;
        else if (a == "--min-count" && i + 1 < argc) min_count = std::atoi(argv[++i]);  // a voxel accepts new points while it holds fewer (default 3; --mapping: 20)
        else if (a == "--no-pipeline") pipeline = false;
        else if (a == "--no-overlap") overlap = false;
        else if (a == "--no-roll-ahead") roll_ahead = false;  // roll the map when the frame is due, not beside the previous registration  // decode every frame when it is due, not during the previous registration
    }
    double z0 = 0, patch = 10, voxel = 1, zero = 0;
    int k_normals = 16;
    std::vector<double> truth;
    {
        std::ifstream tf(dir + "/truth.txt");
        if (tf >> z0 >> patch >> voxel >> k_normals >> zero) {
            double a, b, c;
            while (tf >> a >> b >> c) truth.insert(truth.end(), {a, b, c});
        }
    }
    MapManager mgr((float)patch, 0);
    if (!mgr.context()) {
        std::fprintf(stderr, "no context: %s\n", mgr.lastError());
        return 4;
    }
    if (!mapping && !mgr.load(dir + "/world.map")) {
        std::fprintf(stderr, "cannot load world.map: %s\n", mgr.lastError());
        return 3;
    }
    velo_ctx* ctx = mgr.context();
    if (margin >= 0) {
        const int32_t mg[3] = {margin, margin, 2};
        velo_map_set_margins(ctx, mg);
    }
    start_watchdog(watchdog);
    // ---- the drive: one stub per revolution, points decoded on the GPU when asked for
    HDLManager hdl(ctx);
    if (!hdl.setCalibFile(dir + "/db.xml") || !hdl.loadOffline(dir + "/carposes.txt", dir + "/drive.pcap") ||
        hdl.getNumberOfFrames() == 0) {
        std::fprintf(stderr, "cannot load the drive: %s\n", hdl.lastError());
        return 3;
    }
    const auto frames = hdl.getAllFrameMeta();
    const int n_frames = (int)frames.size();
    RegisterOptions opt;
    opt.iters = 20;
    opt.d_max = 1.0f;
    opt.voxel = (float)voxel;
    opt.k_normals = k_normals;
    opt.integrate = integrate;
    opt.append_threshold = threshold;
    if (min_count > 0) opt.increment_min_count = min_count;
    else if (mapping) opt.increment_min_count = 20;   // (a map made of increments alone needs voxels dense enough for a normal: k = 16)
    if (mapping) {
        opt.increments_in_roi_only = true;
        opt.pipeline_increments = pipeline;
        if (pipeline) roll_ahead = false;   // (the pipelined update moves the tile rectangle itself, one frame ahead)
        // the seed: frame 0's points at its prior without the perturbation (the pose track's x, y; z from truth.txt)
        FrameRef f0 = hdl.prepareFrame(frames[0]);
        if (!f0) {
            std::fprintf(stderr, "cannot decode the seed frame: %s\n", hdl.lastError());
            return 5;
        }
        PoseTransform seed;
        seed.T[0] = f0->carpose->T[0], seed.T[1] = f0->carpose->T[1], seed.T[2] = z0;
        if (!mgr.seedFromFrame(*f0, seed)) {
            std::fprintf(stderr, "seed: %s\n", mgr.lastError());
            return 5;
        }
    }

    if (std::getenv("VELO_TRACE_REGISTER"))
        std::fprintf(stderr, "stream_driver: mapping %d pipeline %d min_count %d threshold %d roll_lead %d\n", (int)mapping, (int)opt.pipeline_increments,
                     opt.increment_min_count, opt.append_threshold, roll_lead);
    double err_vec[3] = {0, 0, 0};
    double z_prev = z0, worst = 0, t_decode = 0, t_register = 0, err_sum = 0, err_last = 0;
    int prepared = -1;  // the frame resident in HBM already (decoded during the previous registration)
    uint64_t pairs = 0;
    const int period = std::max(2 * n_frames - 2, 1);
    // (mapping: frame 0 is the seed, the drive starts at frame 1)
    auto frame_at = [&](int k) { const int j = (k + (mapping ? 1 : 0)) % period; return j < n_frames ? j : period - j; };
    int k_now = 0;   // play position of the frame being registered
    const int k_last = warmup + steps - 1;
    auto one = [&](int f, int f_next, bool timed) -> bool {
        const std::shared_ptr<HDLFrame>& fr = frames[(size_t)f];
        // the NEXT frame is decoded while the GPU registers this one: its packets go up and through the
        // decode kernels right behind this frame's registration and increment
        bool next_ok = true;
        double t_next = 0;
        opt.while_registering = nullptr;
        if (overlap && f_next >= 0)
            opt.while_registering = [&hdl, &mgr, &opt, &frames, f_next, &next_ok, &t_next, roll_ahead, roll_lead, &frame_at, &k_now, k_last] {
                const auto a0 = clk::now();
                beat("registerResident > while_registering: decode of the next frame");
                next_ok = hdl.prepareResidentDuringRegistration(frames[(size_t)f_next]);
                beat("registerResident > while_registering: roll ahead");
                // ... and the map is rolled towards the ROI of a frame to come beside the registration as well (the
                // prior's x, y come from the pose track, not from this registration's result): begun as soon as one
                // of the next roll_lead frames names another tile rectangle, published when that frame is due
                if (next_ok && roll_ahead && roll_lead > 0) {
                    for (int d = 1; d <= roll_lead && k_now + d <= k_last; ++d) {   // (no later frame of this run needs it)
                        const PoseTransform& cd = *frames[(size_t)frame_at(k_now + d)]->carpose;
                        if (mgr.needsRoll(cd.T[0] + 0.15, cd.T[1] - 0.10)) {
                            mgr.rollBegin(cd.T[0] + 0.15, cd.T[1] - 0.10, opt);
                            break;
                        }
                    }
                } else if (next_ok && roll_ahead) {
                    const PoseTransform& c2 = *frames[(size_t)f_next]->carpose;
                    mgr.rollAhead(c2.T[0] + 0.15, c2.T[1] - 0.10, opt);
                }
                t_next = ms_since(a0);
                beat("registerResident (after while_registering)");
            };
        opt.have_next_prior = false;
        if (mapping && f_next >= 0) {   // where the NEXT frame's prior will be: the pipelined update rolls the tiles there
            const PoseTransform& cn = *frames[(size_t)f_next]->carpose;
            opt.have_next_prior = true;
            opt.next_prior_x = cn.T[0] + 0.15, opt.next_prior_y = cn.T[1] - 0.10;
        }
        const auto a = clk::now();
        beat("prepareResident", f);
        if (prepared != f && !hdl.prepareResident(fr)) {
            std::fprintf(stderr, "frame %d: %s\n", f, hdl.lastError());
            return false;
        }
        prepared = -1;
        const double td = ms_since(a);
        const PoseTransform& car = *fr->carpose;  // the track interpolated at the frame's stamp (HDLManager.cxx:104-109)
        PoseTransform init;
        init.T[0] = car.T[0] + 0.15, init.T[1] = car.T[1] - 0.10, init.T[2] = z_prev + 0.03;
        init.R[0] = 0.2, init.R[1] = -0.1, init.R[2] = 0.4;  // frames keep ENU axes: the true rotation is the identity
        PoseTransform out;
        velo_icp_result res;
        const auto b = clk::now();
        beat("registerResident");
        if (!mgr.registerResident(0, fr->timestamp, init, opt, &out, &res)) {
            std::fprintf(stderr, "registerResident failed at frame %d: %s\n", f, mgr.lastError());
            return false;
        }
        const double tr = ms_since(b) - t_next;
        if (opt.while_registering) {
            if (!next_ok) {
                std::fprintf(stderr, "frame %d: %s\n", f_next, hdl.lastError());
                return false;
            }
            prepared = f_next;
        }
        z_prev = out.T[2];
        if (timed) {
            t_decode += td + t_next;
            t_register += tr;
            pairs += res.total_pairs;
            if ((size_t)f * 3 + 2 < truth.size()) {
                const double dx = out.T[0] - truth[(size_t)f * 3], dy = out.T[1] - truth[(size_t)f * 3 + 1],
                             dz = out.T[2] - truth[(size_t)f * 3 + 2];
                err_last = std::sqrt(dx * dx + dy * dy + dz * dz);
                err_vec[0] = dx, err_vec[1] = dy, err_vec[2] = dz;
                err_sum += err_last;
                worst = std::max(worst, err_last);
            }
        }
        return true;
    };
    for (int k = 0; k < warmup; ++k) {
        k_now = k;
        beat("warm-up");
        if (!one(frame_at(k), frame_at(k + 1), false)) return 5;
    }
    beat("velo_synchronize after the warm-up");
    velo_synchronize(ctx);
    const MapStats s0 = mgr.stats();
    const auto t0 = clk::now();
    std::FILE* pf = per_frame.empty() ? nullptr : std::fopen(per_frame.c_str(), "w");
    MapStats sp = s0;
    for (int k = 0; k < steps; ++k) {
        k_now = warmup + k;
        const auto tk = clk::now();
        if (!one(frame_at(warmup + k), k + 1 < steps ? frame_at(warmup + k + 1) : -1, true)) return 5;
        if (pf) {
            const MapStats sn = mgr.stats();
            velo_map_info m2{};      // (velo_map_info_get waits for a roll begun ahead: only on request)
            m2.struct_size = sizeof m2;
            if (std::getenv("VELO_PER_FRAME_INFO")) velo_map_info_get(ctx, &m2);
            std::fprintf(pf, "%d frame %d ms %.4f rolls %llu ahead %llu begun %llu refused %llu full %llu flush %llu up %llu ev %llu last_update %d dims %d %d %d n %llu\n",
                         k, frame_at(warmup + k), ms_since(tk), (unsigned long long)(sn.rolls - sp.rolls),
                         (unsigned long long)(sn.rolls_ahead - sp.rolls_ahead), (unsigned long long)(sn.rolls_begun - sp.rolls_begun),
                         (unsigned long long)(sn.rolls_refused - sp.rolls_refused), (unsigned long long)(sn.full_builds - sp.full_builds),
                         (unsigned long long)(sn.increment_flushes - sp.increment_flushes),
                         (unsigned long long)(sn.points_uploaded - sp.points_uploaded), (unsigned long long)(sn.points_evicted - sp.points_evicted),
                         m2.last_update, m2.dims[0], m2.dims[1], m2.dims[2], (unsigned long long)m2.n_points);
            sp = sn;
        }
    }
    if (pf) std::fclose(pf);
    const bool dbg_end = std::getenv("VELO_TRACE_END") != nullptr;
    if (dbg_end) std::fprintf(stderr, "loop done\n");
    mgr.flushIncrements();
    if (dbg_end) std::fprintf(stderr, "flushed\n");
    velo_synchronize(ctx);
    if (dbg_end) std::fprintf(stderr, "synchronized\n");
    const double total_ms = ms_since(t0);
    const MapStats s1 = mgr.stats();
    velo_map_info mi;
    mi.struct_size = sizeof mi;
    velo_map_info_get(ctx, &mi);
    std::printf("{\"host\": \"C++ (include/veloslam/*.hpp)\", \"mode\": \"%s\", \"frames\": %d, \"distinct_frames\": %d, \"frames_per_s\": %.2f, \"ms_per_frame\": %.4f, "
                "\"mean_pose_error_m\": %.6g, \"last_pose_error_m\": %.6g, \"last_pose_error_xyz_m\": [%.4g, %.4g, %.4g], \"increment_points_per_frame\": %.1f, \"map_updates\": %llu, "
                "\"map_updates_beside_registration\": %llu, \"increment_points_dropped_outside_roi\": %llu, \"increment_min_count\": %d, "
                "\"stage_ms_per_frame\": {\"decode\": %.4f, \"register_roll_icp_increment\": %.4f}, "
                "\"pairs_per_s\": %.4g, \"worst_pose_error_m\": %.15g, \"map_points\": %llu, \"map_subdiv\": %d, "
                "\"last_update\": %d, \"tile_edge_m\": %g, \"decode_planned_ahead\": %s, \"roll_ahead\": %s, "
                "\"roll_lead\": %d, \"map\": {\"full_builds\": %llu, \"rolls\": %llu, \"rolls_ahead\": %llu, \"rolls_begun\": %llu, \"rolls_refused\": %llu, \"tiles_entered\": %llu, \"tiles_left\": %llu, "
                "\"points_uploaded\": %llu, \"points_evicted\": %llu, \"increment_flushes\": %llu, \"increment_points\": %llu}}\n",
                mapping ? (pipeline ? "mapping, increments integrated in pipeline" : "mapping, increments integrated synchronously") : "localisation in a pre-mapped world",
                steps, n_frames, 1e3 * steps / total_ms, total_ms / steps,
                err_sum / std::max(steps, 1), err_last, err_vec[0], err_vec[1], err_vec[2], (double)(s1.increment_points - s0.increment_points) / std::max(steps, 1),
                (unsigned long long)(s1.map_updates - s0.map_updates), (unsigned long long)(s1.updates_beside - s0.updates_beside),
                (unsigned long long)(s1.increment_dropped - s0.increment_dropped), opt.increment_min_count,
                t_decode / steps, t_register / steps,
                (double)pairs / (total_ms * 1e-3), worst, (unsigned long long)mi.n_points, mi.subdiv, mi.last_update, patch, overlap ? "true" : "false", (overlap && roll_ahead) ? "true" : "false",
                (overlap && roll_ahead) ? roll_lead : 0,
                (unsigned long long)(s1.full_builds - s0.full_builds), (unsigned long long)(s1.rolls - s0.rolls),
                (unsigned long long)(s1.rolls_ahead - s0.rolls_ahead), (unsigned long long)(s1.rolls_begun - s0.rolls_begun),
                (unsigned long long)(s1.rolls_refused - s0.rolls_refused),
                (unsigned long long)(s1.tiles_entered - s0.tiles_entered), (unsigned long long)(s1.tiles_left - s0.tiles_left),
                (unsigned long long)(s1.points_uploaded - s0.points_uploaded),
                (unsigned long long)(s1.points_evicted - s0.points_evicted),
                (unsigned long long)(s1.increment_flushes - s0.increment_flushes),
                (unsigned long long)(s1.increment_points - s0.increment_points));
    // (a map grown from its own registrations drifts -- decimetres over hundreds of metres here: the far sides of thin poles
    //  are never mapped; reported, and only a registration that lost the map fails the run)
    return worst > (mapping ? 1.0 : 0.05) ? 6 : 0;
}
