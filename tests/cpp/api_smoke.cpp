// api_smoke.cpp -- exercises the reference-shaped C++ surface (include/veloslam/*.hpp) the way a
// VeloSLAM maintainer would: TransformManager -> HDLFrame -> MapManager::registerFrame.
// Inputs are raw little-endian arrays written by tests/test_cpp_api.py; the result is printed
// for the test to compare with the C-ABI path.  Build: hipcc -std=c++17 (host code only).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <vector>
#include <thread>
#include <veloslam/HDLFrame.hpp>
#include <veloslam/HDLManager.hpp>
#include <veloslam/MapManager.hpp>
#include <veloslam/PacketFile.hpp>
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

// HDLManager as a store of frames that are already in memory (no GPU, no context): the time
// queries of TimeLine, the overwrite rule of addData, waitForFrame, and the cache that clears
// the points of unreferenced frames
static int hdl_store_mode()
{
    using namespace veloslam;
    HDLManager hm(nullptr, 8);
    auto mk = [](int64_t t, float v) {
        auto f = std::make_shared<HDLFrame>();
        f->timestamp = t;
        const int32_t bs[2] = {0, 2};
        const float p[2] = {v, v + 1};
        f->setPoints(p, p, p, nullptr, nullptr, bs, 1);
        return f;
    };
    std::printf("empty %d %d %d\n", (int)(bool)hm.getRecentFrame(), (int)(bool)hm.getFrameNear(5), (int)hm.getRangeBetween(0, 9).size());
    for (int64_t t : {300, 100, 200, 500}) hm.addFrame(mk(t, (float)t));   // out of order on purpose
    std::printf("count %d\n", hm.getNumberOfFrames());
    std::printf("order");
    for (auto& f : hm.getAllFrameMeta()) std::printf(" %lld", (long long)f->timestamp);
    std::printf("\nrecent %lld\n", (long long)hm.getRecentFrame()->timestamp);
    std::printf("at %d %d\n", (int)(bool)hm.getFrameAt(200), (int)(bool)hm.getFrameAt(201));
    std::printf("near");
    for (int64_t t : {-50, 100, 149, 150, 151, 349, 400, 401, 9000}) std::printf(" %lld", (long long)hm.getFrameNear(t)->timestamp);
    std::printf("\nrange");
    for (auto& f : hm.getRangeBetween(160, 420)) std::printf(" %lld", (long long)f->timestamp);
    std::printf("\n");
    // a second frame with a stamp already present replaces the first (TimeLine::addData)
    hm.addFrame(mk(200, 7.0f));
    std::printf("overwrite %d %g\n", hm.getNumberOfFrames(), hm.getFrameAt(200)->x[0]);
    // cache: capacity 3, five frames go in -> the two oldest arrivals lose their points ...
    {
        HDLManager h1(nullptr, 3);
        for (int64_t t : {30, 10, 20, 50, 40}) h1.addFrame(mk(t, (float)t));
        std::printf("cached %d in_memory", h1.cachedFrames());
        for (auto& f : h1.getAllFrameMeta()) std::printf(" %d", (int)f->isInMemory);
        std::printf("\n");
    }
    // ... unless somebody holds them: a held frame is put back, the next unreferenced one goes
    {
        HDLManager h2(nullptr, 1);
        auto a = mk(1, 1.0f), b = mk(2, 2.0f), c = mk(3, 3.0f);
        h2.addFrame(a);
        FrameRef hold = h2.getFrameAt(1);
        h2.addFrame(b);
        h2.addFrame(c);
        std::printf("held %d %d %d count %d\n", (int)a->isInMemory, (int)b->isInMemory, (int)c->isInMemory, (int)a->count.load());
        hold = FrameRef();
        h2.cleanCache();
        std::printf("released %d %d count %d\n", (int)a->isInMemory, (int)c->isInMemory, (int)a->count.load());
        std::printf("gone %d\n", (int)(bool)h2.getFrameAt(2));   // neither in memory nor on a capture
    }
    // .hdlmeta / .insmeta round trip (stubs only: no capture is loaded, nothing can be prepared)
    {
        const char* tmp = std::getenv("VELO_TMP");
        const std::string d = tmp ? tmp : "/tmp";
        HDLManager a(nullptr, 8), b(nullptr, 8);
        for (int k = 0; k < 3; ++k) {
            auto f = std::make_shared<HDLFrame>();
            f->timestamp = 1000 + 100 * k;
            f->filenameTime = 1000;
            f->fileStartPos = 24 + 1264 * 300 * k;
            f->skips = (uint8_t)(3 * k);
            f->isOnHardDrive = true;
            f->carpose->T[0] = 1.5 * k, f->carpose->R[2] = -7.25, f->carpose->V[1] = 0.125, f->carpose->timestamp = 999 + k;
            f->carpose->seconds_pos = 0.5;
            a.addFrame(f);
            PoseTransform p = *f->carpose;
            a.transformManager()->addTransform(p);
        }
        const bool ok = a.saveHDLMeta(d + "/s.hdlmeta") && a.saveINSMeta(d + "/s.insmeta") && b.loadHDLMeta(d + "/s.hdlmeta") &&
                        b.loadINSMeta(d + "/s.insmeta");
        std::ifstream sz(d + "/s.hdlmeta", std::ios::binary | std::ios::ate);
        std::printf("meta %d %lld %d %d", (int)ok, (long long)sz.tellg(), b.getNumberOfFrames(), b.getNumberOfTransforms());
        for (auto& f : b.getAllFrameMeta())
            std::printf(" %lld/%lld/%lld/%d/%d/%g/%g/%g/%lld/%g", (long long)f->timestamp, (long long)f->filenameTime,
                        (long long)f->fileStartPos, (int)f->skips, (int)f->isOnHardDrive, f->carpose->T[0], f->carpose->R[2],
                        f->carpose->V[1], (long long)f->carpose->timestamp, f->carpose->seconds_pos);
        std::printf("\nmeta_missing %d %d\n", (int)b.loadHDLMeta(d + "/none.hdlmeta"), (int)b.loadINSMeta(d + "/none.insmeta"));
        std::printf("meta_unbound %d\n", (int)(bool)b.getRecentFrame());   // a stub of a capture that is not loaded
    }
    // debug dumps (HDLFrame.cxx:36-125)
    {
        const char* tmp = std::getenv("VELO_TMP");
        const std::string d = tmp ? tmp : "/tmp";
        HDLFrame f;
        f.timestamp = 1467590400123456LL + 8LL * 3600 * 1000000;   // 2016-07-04 08:00:00.123456 on the +8 h clock
        const int32_t bs[4] = {0, 2, 2, 5};
        const float px[5] = {1.5f, -2.25f, 3.f, 4.f, 5.f}, pi[5] = {10, 20, 30, 40, 50};
        f.setPoints(px, px, px, pi, nullptr, bs, 3);
        f.pointsMeta.resize(5);
        f.pointsMeta[1].azimuth = 35999, f.pointsMeta[1].distance = 12.5f;
        const bool ok = f.dumpToFiles(d) && f.dumpToPCD(d, 2) && f.dumpToPCD(d, -1);
        std::printf("dump %d %s %s %s\n", (int)ok, HDLFrame::isoString(f.timestamp).c_str(), HDLFrame::isoString(0).c_str(),
                    HDLFrame::isoString(951782400000000LL).c_str());   // 2000-02-29
    }
    // waitForFrame: times out without data, returns the newest frame once a producer adds one
    HDLManager h3(nullptr, 8);
    const bool none = !(bool)h3.waitForFrame(std::chrono::microseconds(2000));
    std::thread producer([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        h3.addFrame(mk(42, 4.0f));
    });
    FrameRef got = h3.waitForFrame(std::chrono::microseconds(5000000));
    producer.join();
    const bool again = (bool)h3.waitForFrame(std::chrono::microseconds(2000));
    std::printf("wait %d %lld %d\n", (int)none, got ? (long long)got->timestamp : -1LL, (int)again);
    return 0;
}

// HDLManager::loadOffline over a recorded drive, every frame prepared (GPU decode) and written out
// for the test to compare with the one-pass decode of the same capture
static int hdl_mode(const std::string& dir, const std::string& out)
{
    using namespace veloslam;
    MapManager mgr(10.0f, 0);
    if (!mgr.context()) {
        std::cerr << "no context: " << mgr.lastError() << std::endl;
        return 4;
    }
    HDLManager hm(mgr.context(), 2);
    auto stubs_before = hm.getNumberOfFrames();
    if (hm.loadOffline(dir + "/carposes.txt", dir + "/drive.pcap") && hm.getNumberOfFrames() > 0 &&
        hm.getRecentFrame()) {
        std::cerr << "a frame was prepared without a calibration" << std::endl;
        return 5;
    }
    std::printf("nocalib %s\n", hm.lastError());
    if (!hm.setCalibFile(dir + "/db.xml") || !hm.loadOffline(dir + "/carposes.txt", dir + "/drive.pcap")) {
        std::cerr << hm.lastError() << std::endl;
        return 3;
    }
    std::printf("frames %d %d transforms %d\n", stubs_before, hm.getNumberOfFrames(), hm.getNumberOfTransforms());
    std::ofstream os(out, std::ios::binary);
    const auto metas = hm.getAllFrameMeta();
    for (size_t k = 0; k < metas.size(); ++k) {
        const auto& m = metas[k];
        const PoseTransform stub_pose = *m->carpose;
        if (m->isInMemory) return 8;
        FrameRef f = hm.getFrameAt(m->timestamp);
        if (!f || !f->isInMemory) {
            std::cerr << "frame " << k << ": " << hm.lastError() << std::endl;
            return 6;
        }
        const int64_t head[6] = {f->timestamp, f->fileStartPos, f->skips, f->firstPacket, f->numPackets, (int64_t)f->numPoints()};
        os.write(reinterpret_cast<const char*>(head), sizeof head);
        os.write(reinterpret_cast<const char*>(f->beamStart.data()), 65 * 4);
        const double pose[12] = {stub_pose.T[0], stub_pose.T[1], stub_pose.T[2], stub_pose.R[0], stub_pose.R[1], stub_pose.R[2],
                                 f->carpose->T[0], f->carpose->T[1], f->carpose->T[2], f->carpose->R[0], f->carpose->R[1], f->carpose->R[2]};
        os.write(reinterpret_cast<const char*>(pose), sizeof pose);
        const size_t n = f->numPoints();
        os.write(reinterpret_cast<const char*>(f->x.data()), 4 * n);
        os.write(reinterpret_cast<const char*>(f->y.data()), 4 * n);
        os.write(reinterpret_cast<const char*>(f->z.data()), 4 * n);
        os.write(reinterpret_cast<const char*>(f->intensity.data()), 4 * n);
        std::vector<uint16_t> az(n);
        std::vector<float> dist(n);
        for (size_t i = 0; i < n; ++i) az[i] = f->pointsMeta[i].azimuth, dist[i] = f->pointsMeta[i].distance;
        os.write(reinterpret_cast<const char*>(az.data()), 2 * n);
        os.write(reinterpret_cast<const char*>(dist.data()), 4 * n);
    }
    // capacity 2: only the frames prepared last still hold points; preparing the first again decodes it again
    int in_mem = 0;
    for (auto& m : metas) in_mem += m->isInMemory ? 1 : 0;
    FrameRef again = hm.getFrameNear(metas.front()->timestamp - 5);
    std::printf("in_memory %d again %zu\n", in_mem, again ? again->numPoints() : (size_t)0);
    // the device-side half: resident in HBM, registered from there
    size_t npts = 0;
    if (!hm.prepareResident(metas.back(), &npts)) {
        std::cerr << hm.lastError() << std::endl;
        return 7;
    }
    std::printf("resident %zu\n", npts);
    // the stubs saved as .hdlmeta and read into a second manager that holds the same capture: same
    // store, and the stubs can be prepared (they were matched to the capture by position and skip)
    HDLManager h2(mgr.context(), 4);
    if (!hm.saveHDLMeta(dir + "/s.hdlmeta") || !h2.setCalibFile(dir + "/db.xml") ||
        !h2.loadOffline(dir + "/carposes.txt", dir + "/drive.pcap") || !h2.loadHDLMeta(dir + "/s.hdlmeta"))
        return 9;
    FrameRef r = h2.getRecentFrame();
    std::printf("meta_reload %d %zu\n", h2.getNumberOfFrames(), r ? r->numPoints() : (size_t)0);
    return 0;
}

// PacketFileWriter / PacketFileReader (the reference's vtkPacketFileWriter / vtkPacketFileReader):
// a capture written packet by packet -- lidar and position packets mixed, a length the writer must
// refuse -- read back record by record, re-read from a remembered position, and held to the bulk
// C functions
static int pcapfile_mode(const std::string& dir)
{
    using namespace veloslam;
    const std::string path = dir + "/w.pcap";
    std::vector<unsigned char> lidar(1206), pos(512), odd(100);
    for (size_t i = 0; i < lidar.size(); ++i) lidar[i] = (unsigned char)(i * 7);
    for (size_t i = 0; i < pos.size(); ++i) pos[i] = (unsigned char)(255 - i);
    PacketFileWriter w;
    if (w.writePacket(lidar.data(), 1206, 5)) return 3;            // not open
    if (!w.open(path)) return 3;
    const int64_t t0 = 1467590400000000LL;
    bool ok = w.writePacket(lidar.data(), 1206, t0);
    lidar[0] = 1;
    ok = ok && w.writePacket(lidar.data(), 1206, t0 + 553);
    ok = ok && w.writePacket(pos.data(), 512, t0 + 600);
    const bool refused = !w.writePacket(odd.data(), 100, t0 + 700);
    lidar[0] = 2;
    ok = ok && w.writePacket(lidar.data(), 1206, t0 + 1106);
    w.close();
    std::printf("written %d refused %d open %d\n", (int)ok, (int)refused, (int)w.isOpen());
    PacketFileReader r;
    if (!r.open(path) || !r.open(path)) return 4;                  // a second open of the same file is a no-op
    const unsigned char* d = nullptr;
    unsigned int n = 0;
    int64_t t = 0, posn[8];
    int k = 0;
    std::printf("records");
    for (;;) {
        r.getFilePosition(&posn[k]);
        if (!r.nextPacket(d, n, t)) break;
        std::printf(" %u@%lld:%u/%lld", n, (long long)posn[k], (unsigned)d[0], (long long)(t - t0));
        ++k;
    }
    std::printf("\nclosed %d\n", (int)!r.isOpen());                // the end of the file closes the reader
    if (!r.open(path)) return 4;
    r.setFilePosition(&posn[2]);                                   // the position packet again, then the last lidar packet
    r.nextPacket(d, n, t);
    const unsigned n2 = n;
    r.nextPacket(d, n, t);
    std::printf("reread %u %u:%u\n", n2, n, (unsigned)d[0]);
    // the bulk reader sees the three lidar packets; the bulk writer writes byte for byte the same file
    size_t cnt = 0;
    std::vector<uint8_t> pk(4 * 1206);
    std::vector<int64_t> ts(4);
    if (velo_pcap_read(path.c_str(), pk.data(), ts.data(), 4, &cnt)) return 5;
    std::printf("bulk %zu %u %u %u %lld\n", cnt, (unsigned)pk[0], (unsigned)pk[1206], (unsigned)pk[2412], (long long)(ts[2] - t0));
    const std::string p2 = dir + "/w2.pcap", p3 = dir + "/w3.pcap";
    if (velo_pcap_write(p2.c_str(), pk.data(), ts.data(), 3)) return 5;
    PacketFileWriter w3;
    if (!w3.open(p3)) return 5;
    for (int i = 0; i < 3; ++i) w3.writePacket(pk.data() + (size_t)i * 1206, 1206, ts[(size_t)i]);
    w3.close();
    std::ifstream a(p2, std::ios::binary), b(p3, std::ios::binary);
    const std::string sa((std::istreambuf_iterator<char>(a)), std::istreambuf_iterator<char>()),
        sb((std::istreambuf_iterator<char>(b)), std::istreambuf_iterator<char>());
    std::printf("same_bytes %d %zu\n", (int)(sa == sb), sa.size());
    PacketFileReader bad;
    std::printf("missing %d %d\n", (int)bad.open(dir + "/none.pcap"), (int)!bad.getLastError().empty());
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    if (std::string(argv[1]) == "--pcapfile") return argc < 3 ? 2 : pcapfile_mode(argv[2]);
    if (std::string(argv[1]) == "--hdl-store") return hdl_store_mode();
    if (std::string(argv[1]) == "--hdl") return argc < 4 ? 2 : hdl_mode(argv[2], argv[3]);
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
