// stream_driver.cpp -- BASELINE configs[2] from C++, through include/veloslam/*.hpp and the C ABI
// they sit on: a recorded drive (veloslam_amd/drive.py layout: drive.pcap, carposes.txt, db.xml,
// world.map, truth.txt) is replayed frame by frame against a rolling device map --
//
//   velo_pcap_read + velo_pcap_index          the capture and its frame index (readFrameInformation)
//   TransformManager::loadFromTxtFile         the pose track (HDLManager::loadOffline, HDLManager.cxx:103-117)
//   velo_load_corrections                     db.xml
//   MapManager::load                          the accumulated map, as tiles
//   per frame: velo_decode (packets in, compensated frame resident in HBM) -> velo_decode_to_frames
//              -> MapManager::registerResident (rolls the device map to the prior's ROI: evict the
//                 tiles that left, append the ones that entered; 20 ICP iterations; accepted
//                 increment to the device-side pending list, merged every append_threshold points)
//
// The prior is what the reference's INS would give: the interpolated car pose (x, y, angles from
// carposes.txt -- the format has no z: z is carried from the previous registration) plus the
// perturbation the bench uses.  Frames are played forwards, then backwards (like bench.py's stream
// record), so that the pose never jumps and any number of steps can be timed.
//
//   hipcc -std=c++17 -O2 -x c++ tools/stream_driver.cpp -Iinclude -Lveloslam_amd/csrc -lveloslam_amd \
//         -Wl,-rpath,$PWD/veloslam_amd/csrc -o tools/stream_driver
//   tools/stream_driver DIR [--steps 100] [--warmup 10] [--threshold 512] [--no-integrate]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "veloslam/MapManager.hpp"
#include "veloslam/TransformManager.hpp"

using namespace veloslam;
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: stream_driver DIR [--steps N] [--warmup W] [--threshold T] [--no-integrate]\n");
        return 2;
    }
    const std::string dir = argv[1];
    int steps = 100, warmup = 10, threshold = 512;
    bool integrate = true;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--steps" && i + 1 < argc) steps = std::atoi(argv[++i]);
        else if (a == "--warmup" && i + 1 < argc) warmup = std::atoi(argv[++i]);
        else if (a == "--threshold" && i + 1 < argc) threshold = std::atoi(argv[++i]);
        else if (a == "--no-integrate") integrate = false;
    }
    // ---- the drive
    size_t n_pkt = 0;
    const std::string pcap = dir + "/drive.pcap";
    if (velo_pcap_read(pcap.c_str(), nullptr, nullptr, 0, &n_pkt) || n_pkt == 0) {
        std::fprintf(stderr, "cannot read %s\n", pcap.c_str());
        return 3;
    }
    std::vector<uint8_t> packets(n_pkt * 1206);
    std::vector<int64_t> times(n_pkt);
    velo_pcap_read(pcap.c_str(), packets.data(), times.data(), n_pkt, &n_pkt);
    const int64_t eight_h = 8LL * 3600 * 1000000;  // timevalToPtime (type_defs.cxx:69-72): the pose track's clock
    for (auto& t : times) t += eight_h;
    size_t n_idx = 0;
    velo_pcap_index(pcap.c_str(), nullptr, 0, &n_idx);
    std::vector<velo_frame_index> index(n_idx);
    velo_pcap_index(pcap.c_str(), index.data(), n_idx, &n_idx);
    TransformManager tm;
    if (!tm.loadFromTxtFile(dir + "/carposes.txt", true)) {
        std::fprintf(stderr, "cannot read carposes.txt\n");
        return 3;
    }
    const std::vector<velo_pose> poses = tm.snapshot();
    velo_laser_corr corr[64];
    int32_t n_enabled = 0;
    if (velo_load_corrections((dir + "/db.xml").c_str(), corr, &n_enabled)) {
        std::fprintf(stderr, "cannot read db.xml\n");
        return 3;
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
    if (!mgr.load(dir + "/world.map")) {
        std::fprintf(stderr, "cannot load world.map: %s\n", mgr.lastError());
        return 3;
    }
    velo_ctx* ctx = mgr.context();
    const int n_frames = (int)n_idx;
    RegisterOptions opt;
    opt.iters = 20;
    opt.d_max = 1.0f;
    opt.voxel = (float)voxel;
    opt.k_normals = k_normals;
    opt.integrate = integrate;
    opt.append_threshold = threshold;

    double z_prev = z0, worst = 0, t_decode = 0, t_register = 0;
    uint64_t pairs = 0;
    auto one = [&](int f, bool timed) -> bool {
        const velo_frame_index& e = index[(size_t)f];
        const bool last = f + 1 >= n_frames;
        const size_t p0 = (size_t)e.first_packet;
        const size_t p1 = last ? n_pkt : (size_t)index[(size_t)f + 1].first_packet + 1;  // incl. the packet that closes it
        velo_decode_opts dop;
        std::memset(&dop, 0, sizeof dop);
        dop.struct_size = sizeof dop;
        dop.initial_firing_skip = e.firing_skip;
        std::memset(dop.laser_selection, 1, sizeof dop.laser_selection);
        const auto a = clk::now();
        int32_t nf = 0;
        size_t npts = 0;
        if (velo_decode_set_options(ctx, &dop) ||
            velo_decode(ctx, packets.data() + p0 * 1206, times.data() + p0, p1 - p0, corr, 64, poses.data(), poses.size(),
                        last ? 1 : 0, nullptr, 0, &nf, &npts) ||
            nf < 1 || velo_decode_to_frames(ctx)) {
            std::fprintf(stderr, "decode of frame %d failed: %s (frames %d)\n", f, velo_last_error(ctx), nf);
            return false;
        }
        const double td = ms_since(a);
        PoseTransform car;
        tm.interpolateTransform(times[p0], &car);  // what the parser takes as the frame's carpose (HDLParser.cxx:993-1001)
        PoseTransform init;
        init.T[0] = car.T[0] + 0.15, init.T[1] = car.T[1] - 0.10, init.T[2] = z_prev + 0.03;
        init.R[0] = 0.2, init.R[1] = -0.1, init.R[2] = 0.4;  // frames keep ENU axes: the true rotation is the identity
        PoseTransform out;
        velo_icp_result res;
        const auto b = clk::now();
        if (!mgr.registerResident(0, times[p0], init, opt, &out, &res)) {
            std::fprintf(stderr, "registerResident failed at frame %d: %s\n", f, mgr.lastError());
            return false;
        }
        const double tr = ms_since(b);
        z_prev = out.T[2];
        if (timed) {
            t_decode += td;
            t_register += tr;
            pairs += res.total_pairs;
            if ((size_t)f * 3 + 2 < truth.size()) {
                const double dx = out.T[0] - truth[(size_t)f * 3], dy = out.T[1] - truth[(size_t)f * 3 + 1],
                             dz = out.T[2] - truth[(size_t)f * 3 + 2];
                worst = std::max(worst, std::sqrt(dx * dx + dy * dy + dz * dz));
            }
        }
        return true;
    };
    const int period = std::max(2 * n_frames - 2, 1);
    auto frame_at = [&](int k) { const int j = k % period; return j < n_frames ? j : period - j; };
    for (int k = 0; k < warmup; ++k)
        if (!one(frame_at(k), false)) return 5;
    velo_synchronize(ctx);
    const MapStats s0 = mgr.stats();
    const auto t0 = clk::now();
    for (int k = 0; k < steps; ++k)
        if (!one(frame_at(warmup + k), true)) return 5;
    mgr.flushIncrements();
    velo_synchronize(ctx);
    const double total_ms = ms_since(t0);
    const MapStats s1 = mgr.stats();
    velo_map_info mi;
    mi.struct_size = sizeof mi;
    velo_map_info_get(ctx, &mi);
    std::printf("{\"host\": \"C++ (include/veloslam/*.hpp)\", \"frames\": %d, \"frames_per_s\": %.2f, \"ms_per_frame\": %.4f, "
                "\"stage_ms_per_frame\": {\"decode\": %.4f, \"register_roll_icp_increment\": %.4f}, "
                "\"pairs_per_s\": %.4g, \"worst_pose_error_m\": %.15g, \"map_points\": %llu, \"map_subdiv\": %d, "
                "\"last_update\": %d, \"tile_edge_m\": %g, "
                "\"map\": {\"full_builds\": %llu, \"rolls\": %llu, \"tiles_entered\": %llu, \"tiles_left\": %llu, "
                "\"points_uploaded\": %llu, \"points_evicted\": %llu, \"increment_flushes\": %llu, \"increment_points\": %llu}}\n",
                steps, 1e3 * steps / total_ms, total_ms / steps, t_decode / steps, t_register / steps,
                (double)pairs / (total_ms * 1e-3), worst, (unsigned long long)mi.n_points, mi.subdiv, mi.last_update, patch,
                (unsigned long long)(s1.full_builds - s0.full_builds), (unsigned long long)(s1.rolls - s0.rolls),
                (unsigned long long)(s1.tiles_entered - s0.tiles_entered), (unsigned long long)(s1.tiles_left - s0.tiles_left),
                (unsigned long long)(s1.points_uploaded - s0.points_uploaded),
                (unsigned long long)(s1.points_evicted - s0.points_evicted),
                (unsigned long long)(s1.increment_flushes - s0.increment_flushes),
                (unsigned long long)(s1.increment_points - s0.increment_points));
    return worst > 0.05 ? 6 : 0;
}
