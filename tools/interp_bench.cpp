// interp_bench.cpp -- the one number the reference publishes for this path, measured on the
// replacement: TransformManager::interpolateTransform on a 1e6-pose timeline.
//
// Reference: "random access ... around 23 microsecs, while sequential access gave 3-4 microsecs.
// Testing platform is a MacBook Air, early 2015 model" (TransformManager.cxx:143-146), produced by
// Test/InterpolateTransformMeasure.cxx:16-58 -- 1e6 poses 10 ms apart with N(0, 1 ms) jitter
// (mt19937), then queries that step 80 us at a time from a random start.  Same generator here
// (std::mt19937 / std::normal_distribution stand in for boost::random: same distributions, other
// streams), two access patterns, both through the reference-shaped C++ class (mutex included, as
// the reference's has one) and through the C entry point velo_interp_pose:
//   sequential  N queries, each 80 us after the previous one, from a random start
//   random      N queries at independent uniform times over the whole timeline
// Host-only (no GPU is touched).  Different hardware than the reference's figure: stated, not compared.
//
//   g++ -O2 -std=c++17 tools/interp_bench.cpp -Iinclude -Lveloslam_amd/csrc -lveloslam_amd \
//       -Wl,-rpath,$PWD/veloslam_amd/csrc -o tools/interp_bench && tools/interp_bench
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "veloslam/TransformManager.hpp"

using clk = std::chrono::steady_clock;

int main(int argc, char** argv)
{
    const size_t n_poses = 1000000;
    const size_t n_q = argc > 1 ? (size_t)atoll(argv[1]) : 1000000;
    std::mt19937 rng(7);
    std::normal_distribution<double> jitter(0.0, 1e3);
    veloslam::TransformManager mgr;
    std::vector<velo_pose> flat(n_poses);
    double last = 0;
    for (size_t i = 0; i < n_poses; ++i) {
        const double now = last + 1e4 + jitter(rng);
        veloslam::PoseTransform p;
        p.timestamp = (int64_t)now;
        p.T[0] = 1e-3 * (double)i;
        p.R[2] = 1e-4 * (double)i;
        p.seconds_pos = 0;
        mgr.addTransform(p);
        last = now;
    }
    flat = mgr.snapshot();
    const int64_t t_end = flat.back().t_us;
    std::uniform_real_distribution<double> u(0.0, (double)t_end);
    double sink = 0;
    auto run = [&](bool sequential, bool c_abi) {
        std::mt19937 r2(11);
        int64_t t = (int64_t)u(r2) / 2;
        std::vector<int64_t> times(n_q);
        for (size_t i = 0; i < n_q; ++i) {
            t = sequential ? t + 80 : (int64_t)u(r2);
            times[i] = t;
        }
        veloslam::PoseTransform out;
        velo_pose po;
        const auto a = clk::now();
        if (c_abi)
            for (size_t i = 0; i < n_q; ++i) {
                velo_interp_pose(flat.data(), flat.size(), times[i], &po);
                sink += po.T[0];
            }
        else
            for (size_t i = 0; i < n_q; ++i) {
                mgr.interpolateTransform(times[i], &out);
                sink += out.T[0];
            }
        const auto b = clk::now();
        return std::chrono::duration<double, std::nano>(b - a).count() / (double)n_q;
    };
    run(true, false);  // warm the caches
    const double seq = run(true, false), rnd = run(false, false);
    const double seq_c = run(true, true), rnd_c = run(false, true);
    std::printf("{\"poses\": %zu, \"queries\": %zu, "
                "\"TransformManager_interpolateTransform_ns\": {\"sequential\": %.1f, \"random\": %.1f}, "
                "\"velo_interp_pose_ns\": {\"sequential\": %.1f, \"random\": %.1f}, "
                "\"reference_published_ns\": {\"sequential\": \"3000-4000\", \"random\": 23000, "
                "\"where\": \"TransformManager.cxx:143-146, MacBook Air early 2015 (other hardware)\"}, "
                "\"checksum\": %.6f}\n",
                n_poses, n_q, seq, rnd, seq_c, rnd_c, sink);
    return 0;
}
