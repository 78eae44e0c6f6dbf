// api_smoke.cpp -- exercises the reference-shaped C++ surface (include/veloslam/*.hpp) the way a
// VeloSLAM maintainer would: TransformManager -> HDLFrame -> MapManager::registerFrame.
// Inputs are raw little-endian arrays written by tests/test_cpp_api.py; the result is printed
// for the test to compare with the C-ABI path.  Build: hipcc -std=c++17 (host code only).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>
#include <veloslam/HDLFrame.hpp>
#include <veloslam/MapManager.hpp>
#include <veloslam/TransformManager.hpp>

template <typename T>
static std::vector<T> slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) {
        std::cerr << "cannot open " << path << std::endl;
        std::exit(2);
    }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v((size_t)n / sizeof(T));
    f.read(reinterpret_cast<char*>(v.data()), n);
    return v;
}

// f3 on the CPU: tiles, eviction, save/load round trip (no GPU needed: the tile index is host code)
static int tiles_mode(const std::string& dir)
{
    using namespace veloslam;
    MapManager mgr(50.0f, 0);  // without a GPU the context is null; tile operations still work
    std::vector<float> x, y, z;
    for (int i = 0; i < 4000; ++i) {
        x.push_back((float)((i * 37) % 400) - 200.0f);
        y.push_back((float)((i * 91) % 300) - 150.0f);
        z.push_back((float)(i % 7) * 0.1f);
    }
    mgr.addPoints(x.data(), y.data(), z.data(), x.size());
    std::printf("patches %zu points %zu\n", mgr.numPatches(), mgr.numPoints());
    const std::string path = dir + "/map.bin";
    if (!mgr.save(path)) return 6;
    MapManager back(1.0f, 0);
    if (!back.load(path)) return 7;
    std::printf("loaded %zu %zu\n", back.numPatches(), back.numPoints());
    auto p = back.findPatch(x[5], y[5]);
    std::printf("tile %d\n", p ? (int)p->size() : -1);
    const size_t roi = back.getROI(0, 0).size();
    const size_t dropped = back.evictOutside(0.0, 0.0, 100.0);
    std::printf("roi %zu dropped %zu left %zu %zu\n", roi, dropped, back.numPatches(), back.numPoints());
    return 0;
}

// a world.map written elsewhere (veloslam_amd/drive.py) read by MapManager::load: tiles, points,
// tilesInRange around a query, and the size of the tile a given point falls into
static int load_mode(const std::string& path, double qx, double qy)
{
    using namespace veloslam;
    MapManager mgr(1.0f, 0);
    if (!mgr.load(path)) {
        std::cerr << mgr.lastError() << std::endl;
        return 7;
    }
    size_t in_range = 0;
    const auto tiles = mgr.tilesInRange(qx, qy);
    for (const auto& t : tiles) in_range += t->size();
    auto p = mgr.findPatch(qx, qy);
    std::printf("loaded %zu %zu\ninrange %zu %zu\ntile %d %.17g %.17g\n", mgr.numPatches(), mgr.numPoints(),
                tiles.size(), in_range, p ? (int)p->size() : -1, p ? p->centerX : 0.0, p ? p->centerY : 0.0);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    if (std::string(argv[1]) == "--tiles") return argc < 3 ? 2 : tiles_mode(argv[2]);
    if (std::string(argv[1]) == "--load") return argc < 5 ? 2 : load_mode(argv[2], std::atof(argv[3]), std::atof(argv[4]));
    const std::string dir = argv[1];
    using namespace veloslam;
    // pose store: rows of 10 doubles (T, Rdeg, V, t_us)
    TransformManager tm;
    const auto poses = slurp<double>(dir + "/poses.f64");
    for (size_t i = 0; i + 9 < poses.size(); i += 10) {
        PoseTransform p;
        for (int k = 0; k < 3; ++k) {
            p.T[k] = poses[i + k];
            p.R[k] = poses[i + 3 + k];
            p.V[k] = poses[i + 6 + k];
        }
        p.timestamp = (int64_t)poses[i + 9];
        p.seconds_pos = 0;
        tm.addTransform(p);
    }
    const auto t = slurp<int64_t>(dir + "/query_t.i64");
    PoseTransform q;
    if (!tm.interpolateTransform(t[0], &q) || !q.valid()) {
        std::cerr << "interpolateTransform failed" << std::endl;
        return 3;
    }
    std::printf("interp %.17g %.17g %.17g %.17g\n", q.T[0], q.T[1], q.T[2], q.R[2]);

    // frame (already motion compensated) and map
    HDLFrame frame;
    const auto fx = slurp<float>(dir + "/fx.f32"), fy = slurp<float>(dir + "/fy.f32"),
               fz = slurp<float>(dir + "/fz.f32");
    const auto bs = slurp<int32_t>(dir + "/beam_start.i32");
    frame.setPoints(fx.data(), fy.data(), fz.data(), nullptr, nullptr, bs.data(), (int)bs.size() - 1);
    frame.timestamp = t[0];
    const CloudView one = frame.getPointsAsOneCloud(3, 4);
    std::printf("beam3 %zu\n", one.size);

    MapManager mgr(400.0f, 0);
    if (!mgr.context()) {
        std::cerr << "no context: " << mgr.lastError() << std::endl;
        return 4;
    }
    // MapManager creates its ctx with cfg == NULL: that must select the fast (pruned) kernel,
    // hints + certificates and graph replay -- not the validation kernel
    velo_cfg eff;
    velo_cfg_get(mgr.context(), &eff);
    std::printf("cfg %d %d %d %d\n", eff.linearize_variant, eff.use_hints, eff.use_graph, eff.map_subdiv);
    const auto mx = slurp<float>(dir + "/mx.f32"), my = slurp<float>(dir + "/my.f32"),
               mz = slurp<float>(dir + "/mz.f32");
    mgr.addPoints(mx.data(), my.data(), mz.data(), mx.size());
    std::printf("patches %zu\n", mgr.numPatches());

    const auto init = slurp<double>(dir + "/init.f64");  // T[3], Rdeg[3]
    PoseTransform prior;
    for (int k = 0; k < 3; ++k) {
        prior.T[k] = init[k];
        prior.R[k] = init[3 + k];
    }
    RegisterOptions opt;
    opt.iters = 10;
    PoseTransform out;
    velo_icp_result res;
    if (!mgr.registerFrame(frame, prior, opt, &out, &res)) {
        std::cerr << "registerFrame failed: " << mgr.lastError() << std::endl;
        return 5;
    }
    std::printf("pose");
    for (int k = 0; k < 12; ++k) std::printf(" %.17g", res.T[k]);
    std::printf("\npairs %u\n", res.iter[9].n_pairs);
    std::printf("trdeg %.12g %.12g %.12g %.12g %.12g %.12g\n", out.T[0], out.T[1], out.T[2], out.R[0], out.R[1], out.R[2]);
    // integrate: the accepted increment goes into the host tiles AND is merged into the device
    // map in place; the next registration must not re-upload the ROI
    velo_map_info mi0, mi1, mi2;
    mi0.struct_size = mi1.struct_size = mi2.struct_size = sizeof(velo_map_info);
    velo_map_info_get(mgr.context(), &mi0);
    opt.integrate = true;
    opt.append_threshold = 1;  // merge after every frame (the default collects 512 points first)
    PoseTransform out2, out3;
    if (!mgr.registerFrame(frame, prior, opt, &out2)) return 6;
    velo_map_info_get(mgr.context(), &mi1);
    if (!mgr.registerFrame(frame, prior, opt, &out3)) return 7;
    velo_map_info_get(mgr.context(), &mi2);
    std::printf("integrate %llu %llu %llu %d %d %zu\n", (unsigned long long)mi0.n_points,
                (unsigned long long)mi1.n_points, (unsigned long long)mi2.n_points, mi1.last_update,
                mi2.last_update, mgr.numPoints());
    return 0;
}
