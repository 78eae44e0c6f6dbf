// frame_map.cpp -- HDLFrame container and the MapManager that hands
// frames to the registration kernels through the C ABI.  Host-only
// plumbing: no arithmetic beyond tile indexing lives here.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include "../../../include/veloslam/MapManager.hpp"

namespace veloslam {

// ------------------------------------------------------------------ HDLFrame
HDLFrame::HDLFrame()
    : timestamp(VELO_TIME_INVALID), carpose(new PoseTransform), isInMemory(false),
      isOnHardDrive(false), count(0), filenameTime(VELO_TIME_INVALID), fileStartPos(0), firstPacket(-1),
      numPackets(0), skips(0)
{
}

CloudView HDLFrame::getPointsAsOneCloud(int startBeam, int endBeam) const
{
    CloudView v{nullptr, nullptr, nullptr, nullptr, 0};
    const int nb = numBeams();
    if (nb == 0) return v;
    if (startBeam < 0) startBeam = 0;
    if (startBeam >= nb) return v;
    if (endBeam <= startBeam + 1) endBeam = startBeam + 1;  // HDLFrame.cxx:133-136
    if (endBeam > nb) endBeam = nb;
    const size_t a = (size_t)beamStart[startBeam], b = (size_t)beamStart[endBeam];
    v.x = x.data() + a;
    v.y = y.data() + a;
    v.z = z.data() + a;
    v.intensity = intensity.empty() ? nullptr : intensity.data() + a;
    v.size = b - a;
    return v;
}

void HDLFrame::setPoints(const float* px, const float* py, const float* pz, const float* pi,
                         const uint16_t* pkt, const int32_t* beam_start, int n_beams)
{
    const size_t n = (size_t)beam_start[n_beams];
    x.assign(px, px + n);
    y.assign(py, py + n);
    z.assign(pz, pz + n);
    if (pi) intensity.assign(pi, pi + n); else intensity.assign(n, 0.0f);
    if (pkt) packetIndex.assign(pkt, pkt + n); else packetIndex.clear();
    beamStart.assign(beam_start, beam_start + n_beams + 1);
    isInMemory = true;
}

void HDLFrame::clear()
{
    std::vector<float>().swap(x);
    std::vector<float>().swap(y);
    std::vector<float>().swap(z);
    std::vector<float>().swap(intensity);
    std::vector<uint16_t>().swap(packetIndex);
    std::vector<PointMeta>().swap(pointsMeta);
    std::vector<std::pair<int64_t, std::string>>().swap(packets);
    beamStart.clear();
    isInMemory = false;
}

namespace {
void put_pose(std::ostream& os, const PoseTransform& p)
{
    for (int i = 0; i < 3; ++i) {
        os.write(reinterpret_cast<const char*>(&p.T[i]), 8);
        os.write(reinterpret_cast<const char*>(&p.R[i]), 8);
        os.write(reinterpret_cast<const char*>(&p.V[i]), 8);
    }
    os.write(reinterpret_cast<const char*>(&p.timestamp), 8);
    os.write(reinterpret_cast<const char*>(&p.week_number), 2);
    os.write(reinterpret_cast<const char*>(&p.milliseconds), 4);
    os.write(reinterpret_cast<const char*>(&p.week_number_pos), 4);
    os.write(reinterpret_cast<const char*>(&p.seconds_pos), 8);
}
void get_pose(std::istream& is, PoseTransform& p)
{
    for (int i = 0; i < 3; ++i) {
        is.read(reinterpret_cast<char*>(&p.T[i]), 8);
        is.read(reinterpret_cast<char*>(&p.R[i]), 8);
        is.read(reinterpret_cast<char*>(&p.V[i]), 8);
    }
    is.read(reinterpret_cast<char*>(&p.timestamp), 8);
    is.read(reinterpret_cast<char*>(&p.week_number), 2);
    is.read(reinterpret_cast<char*>(&p.milliseconds), 4);
    is.read(reinterpret_cast<char*>(&p.week_number_pos), 4);
    is.read(reinterpret_cast<char*>(&p.seconds_pos), 8);
}
}  // namespace

bool HDLFrame::writeMeta(std::ostream& os) const
{
    const int64_t pos[2] = {fileStartPos, 0};
    const uint8_t flags[2] = {skips, (uint8_t)(isOnHardDrive ? 1 : 0)};
    os.write(reinterpret_cast<const char*>(&timestamp), 8);
    os.write(reinterpret_cast<const char*>(&filenameTime), 8);
    os.write(reinterpret_cast<const char*>(pos), 16);
    os.write(reinterpret_cast<const char*>(flags), 2);
    put_pose(os, *carpose);
    return (bool)os;
}

bool HDLFrame::readMeta(std::istream& is)
{
    int64_t pos[2] = {0, 0};
    uint8_t flags[2] = {0, 0};
    int64_t t = 0, ft = 0;
    PoseTransform car;
    is.read(reinterpret_cast<char*>(&t), 8);
    is.read(reinterpret_cast<char*>(&ft), 8);
    is.read(reinterpret_cast<char*>(pos), 16);
    is.read(reinterpret_cast<char*>(flags), 2);
    get_pose(is, car);
    if (!is) return false;  // a truncated record leaves the frame untouched
    timestamp = t;
    filenameTime = ft;
    fileStartPos = pos[0];
    skips = flags[0];
    isOnHardDrive = flags[1] != 0;
    *carpose = car;
    return true;
}

std::string HDLFrame::isoString(int64_t t_us)
{
    if (t_us == VELO_TIME_INVALID) return "not-a-date-time";
    int64_t sec = t_us / 1000000, frac = t_us % 1000000;
    if (frac < 0) {
        frac += 1000000;
        --sec;
    }
    int64_t days = sec / 86400, rem = sec % 86400;
    if (rem < 0) {
        rem += 86400;
        --days;
    }
    // civil date from days since 1970-01-01 (proleptic Gregorian)
    const int64_t z = days + 719468, era = (z >= 0 ? z : z - 146096) / 146097;
    const int64_t doe = z - era * 146097, yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100), mp = (5 * doy + 2) / 153;
    const int d = (int)(doy - (153 * mp + 2) / 5 + 1), m = (int)(mp < 10 ? mp + 3 : mp - 9);
    const long long y = (long long)(yoe + era * 400 + (m <= 2 ? 1 : 0));
    char buf[64];
    if (frac)
        std::snprintf(buf, sizeof buf, "%04lld%02d%02dT%02d%02d%02d.%06d", y, m, d, (int)(rem / 3600), (int)(rem / 60 % 60),
                      (int)(rem % 60), (int)frac);
    else
        std::snprintf(buf, sizeof buf, "%04lld%02d%02dT%02d%02d%02d", y, m, d, (int)(rem / 3600), (int)(rem / 60 % 60),
                      (int)(rem % 60));
    return buf;
}

bool HDLFrame::dumpToFiles(const std::string& dirname) const
{
    const std::string base = dirname + "/" + isoString(timestamp);
    {
        std::ofstream ofs(base + "-points.txt");
        if (!ofs) return false;
        ofs.precision(9);
        for (size_t i = 0; i < x.size(); ++i)
            ofs << x[i] << '\t' << y[i] << '\t' << z[i] << '\t' << (i < intensity.size() ? intensity[i] : 0.0f) << '\n';
    }
    {
        std::ofstream ofs(base + "-pointsMeta.txt");
        if (!ofs) return false;
        for (const PointMeta& m : pointsMeta)
            ofs << m.azimuth << '\t' << m.distance << '\t' << (int)m.intensityFlag << '\t' << (int)m.distanceFlag << '\t'
                << (int)m.flags << '\n';
    }
    std::ofstream ofs(base + "-others.txt");
    if (!ofs) return false;
    ofs.precision(17);
    if (carpose) {
        ofs << "T " << carpose->T[0] << ' ' << carpose->T[1] << ' ' << carpose->T[2] << "\nR " << carpose->R[0] << ' '
            << carpose->R[1] << ' ' << carpose->R[2] << "\nV " << carpose->V[0] << ' ' << carpose->V[1] << ' '
            << carpose->V[2] << "\nt " << isoString(carpose->timestamp) << '\n';
    }
    ofs << '\n' << isInMemory << '\t' << isOnHardDrive << '\t' << (int)count.load() << '\n' << isoString(filenameTime) << '\n'
        << (long long)fileStartPos << '\n' << (int)skips << std::endl;
    return (bool)ofs;
}

bool HDLFrame::dumpToPCD(const std::string& dirname, int beamId) const
{
    if (numBeams() == 0) return false;
    int startBeam, endBeam;
    if (beamId < 0 || beamId > 63) {  // HDLFrame.cxx:110-116: "all" stops one beam short there too
        startBeam = 0;
        endBeam = 63;
    } else {
        startBeam = beamId;
        endBeam = beamId + 1;
    }
    const CloudView c = getPointsAsOneCloud(startBeam, endBeam);
    std::ofstream ofs(dirname + "/" + isoString(timestamp) + "-" + std::to_string(beamId) + ".pcd");
    if (!ofs) return false;
    ofs << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
           "COUNT 1 1 1 1\nWIDTH "
        << c.size << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << c.size << "\nDATA ascii\n";
    ofs.precision(8);  // (pcl writes floats with 8 significant digits)
    for (size_t i = 0; i < c.size; ++i)
        ofs << c.x[i] << ' ' << c.y[i] << ' ' << c.z[i] << ' ' << (c.intensity ? c.intensity[i] : 0.0f) << '\n';
    return (bool)ofs;
}

void intrusive_ptr_add_ref(HDLFrame* p) { ++p->count; }
void intrusive_ptr_release(HDLFrame* p)
{
    unsigned char c = p->count.load();
    while (c != 0 && !p->count.compare_exchange_weak(c, (unsigned char)(c - 1))) {
    }
}

// ---------------------------------------------------------------- MapManager
MapManager::MapManager(float patchRange, int device_id)
    : patchRange_(patchRange), ctx_(nullptr), dirty_(true), haveDevice_(false), res_i0_(0), res_i1_(-1),
      res_j0_(0), res_j1_(-1), residentVoxel_(0), residentK_(0)
{
    // the defaults of cfg == NULL (fast kernel, hints + certificates, graph replay) plus what a
    // rolling map needs: slack around the grid so that rolls and increments update the sorted map
    // in place (tens of voxels in x / y, two in z: the dense table grows with the product)
    velo_cfg cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.abi_version = VELO_ABI_VERSION;
    cfg.max_batch = 4;
    cfg.use_hints = 2;
    cfg.use_graph = 1;
    cfg.map_margin = 16;
    ctx_ = velo_create(device_id, &cfg);
    if (!ctx_) {
        err_ = velo_last_error(nullptr);
        return;
    }
    const int32_t margins[3] = {16, 16, 2};
    velo_map_set_margins(ctx_, margins);
}
MapManager::~MapManager()
{
    if (ctx_) velo_destroy(ctx_);
}
const char* MapManager::lastError() const { return err_.c_str(); }

std::pair<int, int> MapManager::getPatchIdx(double x, double y) const
{
    // tile (i,j) covers [i*r - r/2, i*r + r/2): centres on multiples of the tile
    // edge, like MapManager::getMapCenter (MapManager.cxx:47-52)
    return {(int)std::floor((x + patchRange_ / 2) / patchRange_),
            (int)std::floor((y + patchRange_ / 2) / patchRange_)};
}

std::shared_ptr<MapPatch> MapManager::findPatch(double x, double y)
{
    auto it = patches_.find(getPatchIdx(x, y));
    return it == patches_.end() ? std::shared_ptr<MapPatch>() : it->second;
}

std::shared_ptr<MapPatch> MapManager::getPatch(double x, double y)
{
    auto idx = getPatchIdx(x, y);
    auto it = patches_.find(idx);
    if (it != patches_.end()) return it->second;
    auto mp = std::make_shared<MapPatch>(idx.first * (double)patchRange_,
                                         idx.second * (double)patchRange_, patchRange_);
    patches_[idx] = mp;
    return mp;
}

std::set<std::shared_ptr<MapPatch>> MapManager::getROI(double x, double y)
{
    std::set<std::shared_ptr<MapPatch>> result;  // MapManager.cxx:33-45: the four corners
    for (double cx : {x + ROI_RANGE, x - ROI_RANGE})
        for (double cy : {y + ROI_RANGE, y - ROI_RANGE}) {
            auto p = findPatch(cx, cy);
            if (p) result.insert(p);
        }
    return result;
}

void MapManager::tileRange(double x, double y, int& i0, int& i1, int& j0, int& j1) const
{
    const auto lo = getPatchIdx(x - ROI_RANGE, y - ROI_RANGE), hi = getPatchIdx(x + ROI_RANGE, y + ROI_RANGE);
    i0 = lo.first, i1 = hi.first, j0 = lo.second, j1 = hi.second;
}

std::vector<std::shared_ptr<MapPatch>> MapManager::tilesInRange(double x, double y)
{
    int i0, i1, j0, j1;
    tileRange(x, y, i0, i1, j0, j1);
    std::vector<std::shared_ptr<MapPatch>> out;
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            auto it = patches_.find({i, j});
            if (it != patches_.end()) out.push_back(it->second);
        }
    return out;
}

void MapManager::addPoints(const float* x, const float* y, const float* z, size_t n)
{
    // (consecutive points of a frame or of an increment mostly share a tile: one map lookup per run, not per point --
    //  a frame's increment is thousands of points, every frame, on the host's critical path)
    MapPatch* last = nullptr;
    std::pair<int, int> last_idx{0, 0};
    for (size_t i = 0; i < n; ++i) {
        const auto idx = getPatchIdx(x[i], y[i]);
        if (!last || idx != last_idx) {
            last = getPatch(x[i], y[i]).get();
            last_idx = idx;
        }
        last->x.push_back(x[i]);
        last->y.push_back(y[i]);
        last->z.push_back(z[i]);
    }
    dirty_ = true;
}

bool MapManager::seedFromFrame(const HDLFrame& frame, const PoseTransform& pose)
{
    const size_t n = frame.numPoints();
    if (n == 0) {
        err_ = "seedFromFrame: the frame holds no points (HDLManager::prepareFrame first)";
        return false;
    }
    const Affine3x4 T = pose.getMatrix();
    const double* t = T.data();
    std::vector<float> sx(n), sy(n), sz(n);
    for (size_t i = 0; i < n; ++i) {
        const double x = frame.x[i], y = frame.y[i], z = frame.z[i];
        sx[i] = (float)std::fma(t[0], x, std::fma(t[1], y, std::fma(t[2], z, t[3])));
        sy[i] = (float)std::fma(t[4], x, std::fma(t[5], y, std::fma(t[6], z, t[7])));
        sz[i] = (float)std::fma(t[8], x, std::fma(t[9], y, std::fma(t[10], z, t[11])));
    }
    addPoints(sx.data(), sy.data(), sz.data(), n);
    return true;
}

bool MapManager::boxOf(int i0, int i1, int j0, int j1, float lo[3], float hi[3]) const
{
    // keep region of a tile rectangle: closed box on float coordinates; a tile covers [c - r/2, c + r/2), so the upper
    // edge is the largest float below it
    const float big = 3.0e38f;
    lo[0] = (float)(i0 * (double)patchRange_ - patchRange_ / 2.0);
    lo[1] = (float)(j0 * (double)patchRange_ - patchRange_ / 2.0);
    lo[2] = -big;
    hi[0] = std::nextafter((float)(i1 * (double)patchRange_ + patchRange_ / 2.0), -big);
    hi[1] = std::nextafter((float)(j1 * (double)patchRange_ + patchRange_ / 2.0), -big);
    hi[2] = big;
    return true;
}

size_t MapManager::numPoints() const
{
    size_t n = 0;
    for (const auto& kv : patches_) n += kv.second->size();
    return n;
}

size_t MapManager::evictOutside(double x, double y, double radius)
{
    const double reach = radius + 0.70710678118654752 * patchRange_;
    size_t dropped = 0;
    for (auto it = patches_.begin(); it != patches_.end();) {
        const double dx = it->second->centerX - x, dy = it->second->centerY - y;
        if (dx * dx + dy * dy > reach * reach) {
            dropped += it->second->size();
            // a tile the device map holds is gone from the host store: the next roll rebuilds
            if (haveDevice_ && it->first.first >= res_i0_ && it->first.first <= res_i1_ &&
                it->first.second >= res_j0_ && it->first.second <= res_j1_)
                dirty_ = true;
            it = patches_.erase(it);
        } else {
            ++it;
        }
    }
    return dropped;
}

bool MapManager::save(const std::string& filename) const
{
    // the reference header counts tiles in an unsigned short (MapManager.cxx:81-96)
    if (patches_.size() > 65535) {
        err_ = "map has more than 65535 tiles: the reference record layout cannot count them";
        return false;
    }
    std::ofstream os(filename, std::ios::binary);
    if (!os) {
        err_ = "cannot open " + filename;
        return false;
    }
    // centre and half-extent of the tiled area (the reference writes its own centerX/Y, range)
    double x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    bool first = true;
    for (const auto& kv : patches_) {
        const MapPatch& p = *kv.second;
        const double h = 0.5 * p.range;
        if (first || p.centerX - h < x0) x0 = p.centerX - h;
        if (first || p.centerX + h > x1) x1 = p.centerX + h;
        if (first || p.centerY - h < y0) y0 = p.centerY - h;
        if (first || p.centerY + h > y1) y1 = p.centerY + h;
        first = false;
    }
    const double cx = 0.5 * (x0 + x1), cy = 0.5 * (y0 + y1);
    const float range = (float)std::max(x1 - x0, y1 - y0);
    os.write(reinterpret_cast<const char*>(&cx), sizeof(double));
    os.write(reinterpret_cast<const char*>(&cy), sizeof(double));
    os.write(reinterpret_cast<const char*>(&range), sizeof(float));
    os.write(reinterpret_cast<const char*>(&patchRange_), sizeof(float));
    const unsigned short sz = (unsigned short)patches_.size();
    os.write(reinterpret_cast<const char*>(&sz), sizeof(unsigned short));
    for (const auto& kv : patches_) {
        const MapPatch& p = *kv.second;
        os.write(reinterpret_cast<const char*>(&p.centerX), sizeof(double));
        os.write(reinterpret_cast<const char*>(&p.centerY), sizeof(double));
        os.write(reinterpret_cast<const char*>(&p.range), sizeof(float));
        const unsigned short zero = 0;  // posts, planes, marks, cplxes: not carried
        for (int k = 0; k < 4; ++k) os.write(reinterpret_cast<const char*>(&zero), sizeof(unsigned short));
        const uint64_t n = p.size();
        os.write(reinterpret_cast<const char*>(&n), sizeof n);
        os.write(reinterpret_cast<const char*>(p.x.data()), (std::streamsize)(n * sizeof(float)));
        os.write(reinterpret_cast<const char*>(p.y.data()), (std::streamsize)(n * sizeof(float)));
        os.write(reinterpret_cast<const char*>(p.z.data()), (std::streamsize)(n * sizeof(float)));
    }
    return (bool)os;
}

bool MapManager::load(const std::string& filename)
{
    std::ifstream is(filename, std::ios::binary);
    if (!is) {
        err_ = "cannot open " + filename;
        return false;
    }
    double cx, cy;
    float range, pr;
    unsigned short sz = 0;
    is.read(reinterpret_cast<char*>(&cx), sizeof(double));
    is.read(reinterpret_cast<char*>(&cy), sizeof(double));
    is.read(reinterpret_cast<char*>(&range), sizeof(float));
    is.read(reinterpret_cast<char*>(&pr), sizeof(float));
    is.read(reinterpret_cast<char*>(&sz), sizeof(unsigned short));
    if (!is || !(pr > 0)) {
        err_ = "bad map file header";
        return false;
    }
    patches_.clear();
    patchRange_ = pr;
    for (unsigned short i = 0; i < sz; ++i) {
        auto p = std::make_shared<MapPatch>();
        unsigned short counts[4];
        uint64_t n = 0;
        is.read(reinterpret_cast<char*>(&p->centerX), sizeof(double));
        is.read(reinterpret_cast<char*>(&p->centerY), sizeof(double));
        is.read(reinterpret_cast<char*>(&p->range), sizeof(float));
        is.read(reinterpret_cast<char*>(counts), sizeof counts);
        is.read(reinterpret_cast<char*>(&n), sizeof n);
        if (!is || counts[0] || counts[1] || counts[2] || counts[3] || n > (1ull << 33)) {
            err_ = "bad map patch record";
            return false;
        }
        p->x.resize(n);
        p->y.resize(n);
        p->z.resize(n);
        is.read(reinterpret_cast<char*>(p->x.data()), (std::streamsize)(n * sizeof(float)));
        is.read(reinterpret_cast<char*>(p->y.data()), (std::streamsize)(n * sizeof(float)));
        is.read(reinterpret_cast<char*>(p->z.data()), (std::streamsize)(n * sizeof(float)));
        if (!is) {
            err_ = "truncated map file";
            return false;
        }
        patches_[getPatchIdx(p->centerX, p->centerY)] = p;
    }
    dirty_ = true;
    return true;
}

// Rolling device map.  The resident set is a rectangle of tile indices, so "what stays" is a box
// and the C ABI's box eviction applies it exactly; entering tiles go up in ONE append (row-major
// tile order, each tile's points in their stored order): the device map then equals a fresh build of
// [survivors in their old order, entering points] on the kept grid, by velo_map_append's definition.
bool MapManager::leavingTilesHoldPoints(int i0, int i1, int j0, int j1) const
{
    for (int j = res_j0_; j <= res_j1_; ++j)
        for (int i = res_i0_; i <= res_i1_; ++i) {
            if (i >= i0 && i <= i1 && j >= j0 && j <= j1) continue;
            const auto it = patches_.find({i, j});
            if (it != patches_.end() && it->second->size() != 0) return true;
        }
    return false;
}

bool MapManager::enteringTilesHoldPoints(int i0, int i1, int j0, int j1) const
{
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            if (i >= res_i0_ && i <= res_i1_ && j >= res_j0_ && j <= res_j1_) continue;
            const auto it = patches_.find({i, j});
            if (it != patches_.end() && it->second->size() != 0) return true;
        }
    return false;
}

// device points without waiting for anything (velo_map_info_get waits for the counts of a roll begun ahead)
static uint64_t devicePoints(velo_ctx* ctx)
{
    uint64_t n = 0;
    velo_map_size(ctx, &n);
    return n;
}

bool MapManager::rollTo(double x, double y, const RegisterOptions& o)
{
    if (!ctx_) return false;
    int i0, i1, j0, j1;
    tileRange(x, y, i0, i1, j0, j1);
    const bool same_cfg = o.voxel == residentVoxel_ && o.k_normals == residentK_;
    if (haveDevice_ && !dirty_ && same_cfg && i0 == res_i0_ && i1 == res_i1_ && j0 == res_j0_ && j1 == res_j1_)
        return true;   // (a roll begun ahead is not due yet: the readers keep the resident rectangle)
    // this prior leaves the resident rectangle: a roll begun ahead is due now -- or, if it went elsewhere,
    // is published all the same and the plain roll below goes on from its rectangle
    if (staged_) {
        if (!publishBegun()) return false;
        if (haveDevice_ && !dirty_ && same_cfg && i0 == res_i0_ && i1 == res_i1_ && j0 == res_j0_ && j1 == res_j1_) {
            // While a roll is begun the increments stay pending (registerCore): with a driver that begins the next roll
            // inside every frame's while_registering they would never reach the map.  The due frame is where
            // no roll is begun: what has been collected meanwhile goes in now.
            if (o.integrate) {
                size_t pending = 0;
                velo_pending_count(ctx_, &pending, 0);
                if (pending + pend_x_.size() >= (size_t)std::max(o.append_threshold, 1) && !flushIncrements()) return false;
            }
            return true;
        }
    }
    auto gather = [&](int a0, int a1, int b0, int b1, bool only_new, size_t* tiles) {
        stage_x_.clear(), stage_y_.clear(), stage_z_.clear();
        for (int j = b0; j <= b1; ++j)
            for (int i = a0; i <= a1; ++i) {
                if (only_new && i >= res_i0_ && i <= res_i1_ && j >= res_j0_ && j <= res_j1_) continue;
                auto it = patches_.find({i, j});
                if (it == patches_.end() || it->second->size() == 0) continue;
                const MapPatch& p = *it->second;
                stage_x_.insert(stage_x_.end(), p.x.begin(), p.x.end());
                stage_y_.insert(stage_y_.end(), p.y.begin(), p.y.end());
                stage_z_.insert(stage_z_.end(), p.z.begin(), p.z.end());
                if (tiles) ++*tiles;
            }
    };
    const bool overlap = haveDevice_ && !dirty_ && same_cfg && i0 <= res_i1_ && i1 >= res_i0_ && j0 <= res_j1_ &&
                         j1 >= res_j0_;
    // increments accepted so far go to the host tiles now; what of them lies in tiles that stay
    // resident goes back up with the entering tiles, in the roll's ONE append (a flush of its own
    // would be a second pass over the whole sorted map for a few dozen points)
    static const bool trace_host = std::getenv("VELO_TRACE_REGISTER") != nullptr;
    using tclk = std::chrono::steady_clock;
    const auto tt0 = tclk::now();
    auto us_since = [&](tclk::time_point a) { return std::chrono::duration<double, std::micro>(tclk::now() - a).count(); };
    // (a roll that only evicts leaves the list pending: folding a dozen points into the map is a pass over all of
    //  it -- 0.85 ms on the stream's 11 M points -- that the next flush or the next entering column does anyway.
    //  pend_ = increments already in the host tiles and not yet on the device: they go up with whatever goes up)
    const bool enters = !overlap || enteringTilesHoldPoints(i0, i1, j0, j1) || !pend_x_.empty();
    if (haveDevice_ && enters && !takeIncrements(o.increments_in_roi_only)) return false;
    const double t_take = us_since(tt0);
    auto keepPendInside = [&](int a0, int a1, int b0, int b1) {   // (what left with its tile waits there, in the host tile)
        size_t w = 0;
        for (size_t k = 0; k < pend_x_.size(); ++k) {
            const auto t = getPatchIdx(pend_x_[k], pend_y_[k]);
            if (t.first < a0 || t.first > a1 || t.second < b0 || t.second > b1) continue;
            pend_x_[w] = pend_x_[k], pend_y_[w] = pend_y_[k], pend_z_[w] = pend_z_[k];
            ++w;
        }
        pend_x_.resize(w), pend_y_.resize(w), pend_z_.resize(w);
    };
    if (overlap) {
        const uint64_t n_before = devicePoints(ctx_);
        float lo[3], hi[3];
        boxOf(i0, i1, j0, j1, lo, hi);
        bool kept_something = true;
        if ((i0 > res_i0_ || i1 < res_i1_ || j0 > res_j0_ || j1 < res_j1_) && leavingTilesHoldPoints(i0, i1, j0, j1)) {
            const int rc = velo_map_evict_outside(ctx_, lo, hi);
            if (rc == VELO_E_INVALID) {
                kept_something = false;  // nothing of the device map lies in the new rectangle: rebuild
            } else if (rc) {
                err_ = velo_last_error(ctx_);
                return false;
            }
        }
        if (kept_something) {
            const double t_evict = us_since(tt0);
            stats_.points_evicted += n_before - devicePoints(ctx_);
            size_t tiles = 0;
            gather(i0, i1, j0, j1, true, &tiles);
            const double t_gather = us_since(tt0);
            const size_t n_tile_points = stage_x_.size();
            size_t n_inc = 0;
            for (size_t k = 0; k < pend_x_.size(); ++k) {
                // (entering tiles already hold their share: takeIncrements put it into the host tiles)
                const auto t = getPatchIdx(pend_x_[k], pend_y_[k]);
                const bool stays = t.first >= std::max(i0, res_i0_) && t.first <= std::min(i1, res_i1_) &&
                                   t.second >= std::max(j0, res_j0_) && t.second <= std::min(j1, res_j1_);
                if (!stays) continue;
                stage_x_.push_back(pend_x_[k]), stage_y_.push_back(pend_y_[k]), stage_z_.push_back(pend_z_[k]);
                ++n_inc;
            }
            if (!stage_x_.empty()) {
                if (velo_map_append(ctx_, stage_x_.data(), stage_y_.data(), stage_z_.data(), stage_x_.size())) {
                    err_ = velo_last_error(ctx_);
                    return false;
                }
                pend_x_.clear(), pend_y_.clear(), pend_z_.clear();   // (on the device now, or gone with a tile that left)
                if (n_inc) ++stats_.map_updates;
            } else {
                keepPendInside(i0, i1, j0, j1);
            }
            if (trace_host)
                std::fprintf(stderr, "rollTo: take %.0f us, evict %.0f, gather %.0f (%zu points), append %.0f\n", t_take, t_evict - t_take,
                             t_gather - t_evict, stage_x_.size(), us_since(tt0) - t_gather);
            stats_.points_uploaded += n_tile_points;
            stats_.tiles_entered += tiles;
            stats_.tiles_left += (uint64_t)std::max(0, (res_i1_ - res_i0_ + 1) * (res_j1_ - res_j0_ + 1) -
                                                           (std::min(i1, res_i1_) - std::max(i0, res_i0_) + 1) *
                                                               (std::min(j1, res_j1_) - std::max(j0, res_j0_) + 1));
            ++stats_.rolls;
            res_i0_ = i0, res_i1_ = i1, res_j0_ = j0, res_j1_ = j1;
            return true;
        }
    }
    // first ROI, a jump with no overlap, other map parameters, or host tiles edited behind the
    // device map's back (addPoints / load / evictOutside): build from the tiles
    size_t tiles = 0;
    gather(i0, i1, j0, j1, false, &tiles);
    if (stage_x_.empty()) {
        err_ = "no map points within ROI_RANGE of the prior";
        return false;
    }
    if (velo_map_reset(ctx_, stage_x_.data(), stage_y_.data(), stage_z_.data(), stage_x_.size(), o.voxel,
                       o.k_normals)) {
        err_ = velo_last_error(ctx_);
        return false;
    }
    pend_x_.clear(), pend_y_.clear(), pend_z_.clear();   // (the host tiles held them: they came up with the build)
    ++stats_.full_builds;
    stats_.points_uploaded += stage_x_.size();
    haveDevice_ = true;
    dirty_ = false;
    residentVoxel_ = o.voxel;
    residentK_ = o.k_normals;
    res_i0_ = i0, res_i1_ = i1, res_j0_ = j0, res_j1_ = j1;
    return true;
}

bool MapManager::rollAhead(double x, double y, const RegisterOptions& o)
{
    // (a roll begun ahead and not yet published: velo_map_roll_overlapped would publish it inside the library and this
    //  method would then work from the stale resident rectangle -- entering tiles appended twice)
    if (!ctx_ || !haveDevice_ || dirty_ || staged_) return false;
    int i0, i1, j0, j1;
    tileRange(x, y, i0, i1, j0, j1);
    if (o.voxel != residentVoxel_ || o.k_normals != residentK_) return false;
    if (i0 == res_i0_ && i1 == res_i1_ && j0 == res_j0_ && j1 == res_j1_) return true;  // nothing to do
    if (!(i0 <= res_i1_ && i1 >= res_i0_ && j0 <= res_j1_ && j1 >= res_j0_)) return false;  // a jump: plain rebuild
    if (refusedBefore(i0, i1, j0, j1)) return false;   // (the same question from the same rectangle: the same answer)
    // entering tiles (as rollTo gathers them; the pending increments are not taken here)
    stage_x_.clear(), stage_y_.clear(), stage_z_.clear();
    size_t tiles = 0;
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            if (i >= res_i0_ && i <= res_i1_ && j >= res_j0_ && j <= res_j1_) continue;
            auto it = patches_.find({i, j});
            if (it == patches_.end() || it->second->size() == 0) continue;
            const MapPatch& p = *it->second;
            stage_x_.insert(stage_x_.end(), p.x.begin(), p.x.end());
            stage_y_.insert(stage_y_.end(), p.y.begin(), p.y.end());
            stage_z_.insert(stage_z_.end(), p.z.begin(), p.z.end());
            ++tiles;
        }
    const bool evicts = (i0 > res_i0_ || i1 < res_i1_ || j0 > res_j0_ || j1 < res_j1_) && leavingTilesHoldPoints(i0, i1, j0, j1);
    float lo[3], hi[3];
    boxOf(i0, i1, j0, j1, lo, hi);
    const uint64_t n_before = devicePoints(ctx_);
    const int rc = velo_map_roll_overlapped(ctx_, evicts ? lo : nullptr, evicts ? hi : nullptr, stage_x_.data(),
                                            stage_y_.data(), stage_z_.data(), stage_x_.size());
    const uint64_t n_after = devicePoints(ctx_);
    if (rc == VELO_E_AGAIN || rc == VELO_E_INVALID) {
        // refused: what the device holds no longer matches a tile rectangle for sure (an eviction may
        // have gone through) -- the plain roll rebuilds from the tiles
        if (n_after != n_before) dirty_ = true;
        ++stats_.rolls_refused;
        if (rc == VELO_E_AGAIN) noteRefused(i0, i1, j0, j1);
        return false;
    }
    if (rc) {
        err_ = velo_last_error(ctx_);
        dirty_ = true;
        return false;
    }
    stats_.points_evicted += n_before + stage_x_.size() - n_after;
    stats_.points_uploaded += stage_x_.size();
    stats_.tiles_entered += tiles;
    stats_.tiles_left += (uint64_t)std::max(0, (res_i1_ - res_i0_ + 1) * (res_j1_ - res_j0_ + 1) -
                                                   (std::min(i1, res_i1_) - std::max(i0, res_i0_) + 1) *
                                                       (std::min(j1, res_j1_) - std::max(j0, res_j0_) + 1));
    ++stats_.rolls;
    ++stats_.rolls_ahead;
    res_i0_ = i0, res_i1_ = i1, res_j0_ = j0, res_j1_ = j1;
    return true;
}

bool MapManager::needsRoll(double x, double y) const
{
    if (!ctx_ || !haveDevice_ || dirty_ || staged_) return false;
    int i0, i1, j0, j1;
    tileRange(x, y, i0, i1, j0, j1);
    return !(i0 == res_i0_ && i1 == res_i1_ && j0 == res_j0_ && j1 == res_j1_);
}

bool MapManager::rollBegin(double x, double y, const RegisterOptions& o)
{
    if (!ctx_ || !haveDevice_ || dirty_ || staged_) return false;
    int i0, i1, j0, j1;
    tileRange(x, y, i0, i1, j0, j1);
    if (o.voxel != residentVoxel_ || o.k_normals != residentK_) return false;
    if (i0 == res_i0_ && i1 == res_i1_ && j0 == res_j0_ && j1 == res_j1_) return true;  // nothing to do
    if (!(i0 <= res_i1_ && i1 >= res_i0_ && j0 <= res_j1_ && j1 >= res_j0_)) return false;  // a jump: plain rebuild
    if (refusedBefore(i0, i1, j0, j1)) return false;   // (the same question from the same rectangle: the same answer)
    stage_x_.clear(), stage_y_.clear(), stage_z_.clear();
    size_t tiles = 0;
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            if (i >= res_i0_ && i <= res_i1_ && j >= res_j0_ && j <= res_j1_) continue;
            auto it = patches_.find({i, j});
            if (it == patches_.end() || it->second->size() == 0) continue;
            const MapPatch& p = *it->second;
            stage_x_.insert(stage_x_.end(), p.x.begin(), p.x.end());
            stage_y_.insert(stage_y_.end(), p.y.begin(), p.y.end());
            stage_z_.insert(stage_z_.end(), p.z.begin(), p.z.end());
            ++tiles;
        }
    const bool evicts = (i0 > res_i0_ || i1 < res_i1_ || j0 > res_j0_ || j1 < res_j1_) && leavingTilesHoldPoints(i0, i1, j0, j1);
    float lo[3], hi[3];
    boxOf(i0, i1, j0, j1, lo, hi);
    const uint64_t n_before = devicePoints(ctx_);
    const int rc = velo_map_roll_begin(ctx_, evicts ? lo : nullptr, evicts ? hi : nullptr, stage_x_.data(),
                                       stage_y_.data(), stage_z_.data(), stage_x_.size());
    if (rc == VELO_E_AGAIN || rc == VELO_E_INVALID) {
        // Refused.  Normally before anything changed (the plain roll does it when due) -- but the roll is an eviction
        // FOLLOWED by an append, and the append can be refused after the eviction went into the library's map (a table
        // that passes its limit with the entering points): the device then holds no tile rectangle any more
        if (devicePoints(ctx_) != n_before) dirty_ = true;
        ++stats_.rolls_refused;
        if (rc == VELO_E_AGAIN) noteRefused(i0, i1, j0, j1);   // (E_INVALID: "one roll per registration" -- ask again next frame)
        return false;
    }
    if (rc) {
        err_ = velo_last_error(ctx_);
        dirty_ = true;
        return false;
    }
    staged_ = true;
    st_i0_ = i0, st_i1_ = i1, st_j0_ = j0, st_j1_ = j1;
    st_n_before_ = n_before;
    st_n_in_ = stage_x_.size();
    stats_.points_uploaded += stage_x_.size();
    stats_.tiles_entered += tiles;
    stats_.tiles_left += (uint64_t)std::max(0, (res_i1_ - res_i0_ + 1) * (res_j1_ - res_j0_ + 1) -
                                                   (std::min(i1, res_i1_) - std::max(i0, res_i0_) + 1) *
                                                       (std::min(j1, res_j1_) - std::max(j0, res_j0_) + 1));
    return true;
}

bool MapManager::publishBegun()
{
    if (!staged_) return true;
    if (velo_map_roll_publish(ctx_)) {
        err_ = velo_last_error(ctx_);
        dirty_ = true;
        return false;
    }
    // (host arithmetic inside the library: velo_map_info_get would wait for the roll's normal counts -- a host wait on
    //  the very path the roll begun ahead exists to clear, whenever the roll is not through yet)
    stats_.points_evicted += st_n_before_ + st_n_in_ - devicePoints(ctx_);
    ++stats_.rolls;
    ++stats_.rolls_ahead;
    ++stats_.rolls_begun;
    res_i0_ = st_i0_, res_i1_ = st_i1_, res_j0_ = st_j0_, res_j1_ = st_j1_;
    staged_ = false;
    // what was taken off the device while the roll was begun and lies in a tile that left: it waits in its host tile
    size_t w = 0;
    for (size_t k = 0; k < pend_x_.size(); ++k) {
        const auto t = getPatchIdx(pend_x_[k], pend_y_[k]);
        if (t.first < res_i0_ || t.first > res_i1_ || t.second < res_j0_ || t.second > res_j1_) continue;
        pend_x_[w] = pend_x_[k], pend_y_[w] = pend_y_[k], pend_z_[w] = pend_z_[k];
        ++w;
    }
    pend_x_.resize(w), pend_y_.resize(w), pend_z_.resize(w);
    return true;
}

bool MapManager::takeIncrements(bool roi_only)
{
    size_t n = 0;
    if (velo_pending_count(ctx_, &n, 1)) {
        err_ = velo_last_error(ctx_);
        return false;
    }
    if (n == 0) return true;
    take_x_.resize(n), take_y_.resize(n), take_z_.resize(n);
    if (velo_pending_fetch(ctx_, take_x_.data(), take_y_.data(), take_z_.data(), n, &n) || velo_pending_clear(ctx_)) {
        err_ = velo_last_error(ctx_);
        return false;
    }
    if (roi_only && haveDevice_) {   // what fell outside the resident rectangle is not integrated (RegisterOptions)
        size_t w = 0;
        for (size_t k = 0; k < n; ++k) {
            const auto t = getPatchIdx(take_x_[k], take_y_[k]);
            if (t.first < res_i0_ || t.first > res_i1_ || t.second < res_j0_ || t.second > res_j1_) continue;
            take_x_[w] = take_x_[k], take_y_[w] = take_y_[k], take_z_[w] = take_z_[k];
            ++w;
        }
        stats_.increment_dropped += n - w;
        n = w;
    }
    // the host tiles follow the device map (not the other way round: dirty_ stays as it was)
    const bool was_dirty = dirty_;
    addPoints(take_x_.data(), take_y_.data(), take_z_.data(), n);
    dirty_ = was_dirty;
    pend_x_.insert(pend_x_.end(), take_x_.begin(), take_x_.begin() + (std::ptrdiff_t)n);
    pend_y_.insert(pend_y_.end(), take_y_.begin(), take_y_.begin() + (std::ptrdiff_t)n);
    pend_z_.insert(pend_z_.end(), take_z_.begin(), take_z_.begin() + (std::ptrdiff_t)n);
    ++stats_.increment_flushes;
    stats_.increment_points += n;
    return true;
}

bool MapManager::flushIncrements()
{
    if (!ctx_) return false;
    // (an explicit flush while a roll is begun: the append below would publish it inside the library -- keep the
    //  resident rectangle in step)
    if (staged_ && !publishBegun()) return false;
    if (!takeIncrements(roiOnly_)) return false;
    if (pend_x_.empty() || !haveDevice_) return true;
    // back up: the points in resident tiles (anything else waits in its host tile until that tile enters)
    stage_x_.clear(), stage_y_.clear(), stage_z_.clear();
    for (size_t k = 0; k < pend_x_.size(); ++k) {
        const auto t = getPatchIdx(pend_x_[k], pend_y_[k]);
        if (t.first < res_i0_ || t.first > res_i1_ || t.second < res_j0_ || t.second > res_j1_) continue;
        stage_x_.push_back(pend_x_[k]), stage_y_.push_back(pend_y_[k]), stage_z_.push_back(pend_z_[k]);
    }
    pend_x_.clear(), pend_y_.clear(), pend_z_.clear();
    if (!stage_x_.empty()) {
        if (velo_map_append(ctx_, stage_x_.data(), stage_y_.data(), stage_z_.data(), stage_x_.size())) {
            err_ = velo_last_error(ctx_);
            return false;
        }
        ++stats_.map_updates;
    }
    return true;
}

// pipeline_increments: BEFORE velo_icp_batch_start (the publish follows it, registerCore).  The previous frame's increment is
// complete on the device (it ran right behind that frame's registration): it comes down (copy stream: nothing queues
// behind the registration just started), goes into the host tiles, and back up -- with the tiles that enter on the way
// to the next frame's rectangle, after the eviction of those that leave -- as ONE roll on the roll's own stream,
// beside the registration; the publish makes the main stream wait for it on the device, so the increment enqueued next
// is computed against the updated map.  The host waits for nothing but an eviction's first count.
bool MapManager::updateBesideRegistration(const RegisterOptions& o)
{
    if (staged_ && !publishBegun()) return false;
    if (!takeIncrements(o.increments_in_roi_only)) return false;
    int i0 = res_i0_, i1 = res_i1_, j0 = res_j0_, j1 = res_j1_;
    if (o.have_next_prior) tileRange(o.next_prior_x, o.next_prior_y, i0, i1, j0, j1);
    const bool moves = !(i0 == res_i0_ && i1 == res_i1_ && j0 == res_j0_ && j1 == res_j1_);
    if (!moves && pend_x_.empty()) return true;
    if (moves && !(i0 <= res_i1_ && i1 >= res_i0_ && j0 <= res_j1_ && j1 >= res_j0_)) return true;  // a jump: rollTo rebuilds
    if (moves && refusedBefore(i0, i1, j0, j1)) {   // (the library said no to this move: the increments alone, the move when due)
        i0 = res_i0_, i1 = res_i1_, j0 = res_j0_, j1 = res_j1_;
        if (pend_x_.empty()) return true;
    }
    stage_x_.clear(), stage_y_.clear(), stage_z_.clear();
    size_t tiles = 0;
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            if (i >= res_i0_ && i <= res_i1_ && j >= res_j0_ && j <= res_j1_) continue;
            auto it = patches_.find({i, j});
            if (it == patches_.end() || it->second->size() == 0) continue;
            const MapPatch& p = *it->second;
            stage_x_.insert(stage_x_.end(), p.x.begin(), p.x.end());
            stage_y_.insert(stage_y_.end(), p.y.begin(), p.y.end());
            stage_z_.insert(stage_z_.end(), p.z.begin(), p.z.end());
            ++tiles;
        }
    const size_t n_tile_points = stage_x_.size();
    size_t n_inc = 0;
    for (size_t k = 0; k < pend_x_.size(); ++k) {   // (entering tiles hold their share already)
        const auto t = getPatchIdx(pend_x_[k], pend_y_[k]);
        const bool stays = t.first >= std::max(i0, res_i0_) && t.first <= std::min(i1, res_i1_) &&
                           t.second >= std::max(j0, res_j0_) && t.second <= std::min(j1, res_j1_);
        if (!stays) continue;
        stage_x_.push_back(pend_x_[k]), stage_y_.push_back(pend_y_[k]), stage_z_.push_back(pend_z_[k]);
        ++n_inc;
    }
    const bool shrinks = i0 > res_i0_ || i1 < res_i1_ || j0 > res_j0_ || j1 < res_j1_;
    const bool evicts = shrinks && leavingTilesHoldPoints(i0, i1, j0, j1);
    if (!evicts && stage_x_.empty()) {   // (only empty tiles changed hands)
        res_i0_ = i0, res_i1_ = i1, res_j0_ = j0, res_j1_ = j1;
        pend_x_.clear(), pend_y_.clear(), pend_z_.clear();
        return true;
    }
    float lo[3], hi[3];
    boxOf(i0, i1, j0, j1, lo, hi);
    const uint64_t n_before = devicePoints(ctx_);
    int rc = velo_map_roll_begin(ctx_, evicts ? lo : nullptr, evicts ? hi : nullptr, stage_x_.data(), stage_y_.data(),
                                 stage_z_.data(), stage_x_.size());
    if (rc == VELO_OK) publishOwed_ = true;   // (registerCore: right after velo_icp_batch_start)
    if (rc == VELO_E_AGAIN || rc == VELO_E_INVALID) {
        // not beside a registration (a hashed table, a map without normals whose grid would move ...): the increments
        // stay in pend_ and the rectangle where it is -- the plain roll / flush of the next frame does both
        if (devicePoints(ctx_) != n_before) dirty_ = true;
        ++stats_.rolls_refused;
        if (rc == VELO_E_AGAIN && moves) noteRefused(i0, i1, j0, j1);
        forcePlainFlush_ = true;
        return true;
    }
    if (rc) {
        err_ = velo_last_error(ctx_);
        dirty_ = true;
        return false;
    }
    const uint64_t n_after = devicePoints(ctx_);
    stats_.points_evicted += n_before + stage_x_.size() - n_after;
    stats_.points_uploaded += n_tile_points;
    stats_.tiles_entered += tiles;
    if (moves) {
        stats_.tiles_left += (uint64_t)std::max(0, (res_i1_ - res_i0_ + 1) * (res_j1_ - res_j0_ + 1) -
                                                       (std::min(i1, res_i1_) - std::max(i0, res_i0_) + 1) *
                                                           (std::min(j1, res_j1_) - std::max(j0, res_j0_) + 1));
        ++stats_.rolls;
        ++stats_.rolls_ahead;
    }
    if (n_inc) {
        ++stats_.map_updates;
        ++stats_.updates_beside;
    }
    res_i0_ = i0, res_i1_ = i1, res_j0_ = j0, res_j1_ = j1;
    pend_x_.clear(), pend_y_.clear(), pend_z_.clear();
    return true;
}

bool MapManager::registerCore(int frame, int64_t timestamp, const PoseTransform& init, const RegisterOptions& o,
                              PoseTransform* out, velo_icp_result* result)
{
    static const bool trace_host = std::getenv("VELO_TRACE_REGISTER") != nullptr;   // (debugging aid: where a frame's host time goes)
    using tclk = std::chrono::steady_clock;
    const auto tt0 = tclk::now();
    auto us_since = [&](tclk::time_point a) { return std::chrono::duration<double, std::micro>(tclk::now() - a).count(); };
    roiOnly_ = o.increments_in_roi_only;
    const bool pipelined = o.integrate && o.pipeline_increments;
    if (!rollTo(init.T[0], init.T[1], o)) return false;
    if (forcePlainFlush_) {   // (the library refused the update beside the last registration: plainly, now)
        forcePlainFlush_ = false;
        if (!flushIncrements()) return false;
    }
    const double t_roll = us_since(tt0);
    const Affine3x4 T0 = init.getMatrix();
    // one registration of the resident frames; the other frames of a multi-frame decode keep the
    // prior they are given here only if they are registered by their own call
    velo_icp_result local[4];
    if (frame < 0 || frame >= 4) {
        err_ = "frame index out of range (the context holds up to 4 resident frames)";
        return false;
    }
    double T0s[4 * 12];
    for (int f = 0; f < 4; ++f) std::memcpy(T0s + 12 * f, T0.data(), sizeof(double) * 12);
    // pipelined integration: the map update with the PREVIOUS frame's increment runs on the roll's own stream beside this
    // registration, which reads the map as it was ...
    // The roll is begun right BEHIND the registration's start: the ~200 us of host work it takes (the increment's fetch, the
    // host tiles, ~25 launches) then run while the GPU registers.  (A/B, profiles/r06/ab_mapping_order.txt: while every roll
    // ran on the CU-masked stream, beginning it BEFORE the start won -- 830 against 720 frames/s; with the light roll on a
    // plain stream it is the other way round -- 1 115 - 1 145 against 970 - 1 050.  VELO_UPDATE_BEFORE_START=1: the other order.)
    static const bool update_after_start = std::getenv("VELO_UPDATE_BEFORE_START") == nullptr;
    if (pipelined && !update_after_start && !updateBesideRegistration(o)) return false;
    const double t_update = us_since(tt0);
    if (velo_icp_batch_start(ctx_, T0s, o.iters, o.d_max)) {
        err_ = velo_last_error(ctx_);
        return false;
    }
    if (pipelined && update_after_start && !updateBesideRegistration(o)) return false;
    // ... and published right behind the registration: the main stream waits for the roll ON THE DEVICE, whatever is
    // enqueued from here on -- this frame's increment first of all -- sees the updated map
    if (publishOwed_) {
        publishOwed_ = false;
        if (velo_map_roll_publish(ctx_)) {
            err_ = velo_last_error(ctx_);
            dirty_ = true;
            return false;
        }
    }
    // the accepted increment joins the device-side pending list at the pose the registration
    // leaves on the device: nothing is fetched, nothing blocks, so it is enqueued right behind
    if (o.integrate && velo_increment_pending(ctx_, frame, nullptr, o.increment_min_count)) {
        err_ = velo_last_error(ctx_);
        return false;
    }
    // the GPU is busy for ~0.5 ms: the caller's work goes here -- host work, or the NEXT frame's decode
    // on this very context (HDLManager::prepareResident: it queues behind the registration, and the
    // result below belongs to the frames that were resident when it started)
    const double t_start = us_since(tt0);
    if (o.while_registering) o.while_registering();
    const double t_while = us_since(tt0);
    if (velo_icp_batch_finish(ctx_, local)) {
        err_ = velo_last_error(ctx_);
        return false;
    }
    if (trace_host && us_since(tt0) > 900.0)
        std::fprintf(stderr, "registerCore: rollTo %.0f us, update begun %.0f, start + publish + increment %.0f, while_registering %.0f, finish %.0f\n",
                     t_roll, t_update - t_roll, t_start - t_update, t_while - t_start, us_since(tt0) - t_while);
    const velo_icp_result& r = local[frame];
    if (result) *result = r;
    Affine3x4 M;
    std::memcpy(M.data(), r.T, sizeof(double) * 12);
    PoseTransform p = PoseTransform::fromMatrix(M);
    p.timestamp = timestamp;
    for (int i = 0; i < 3; ++i) p.V[i] = init.V[i];
    *out = p;
    if (o.integrate && !pipelined) {
        if (o.append_threshold <= 1) {  // "after every frame": wait for this one
            if (!staged_ && !flushIncrements()) return false;
        } else {
            size_t pending = 0;
            velo_pending_count(ctx_, &pending, 0);  // without waiting: the frame in flight counts next time
            // (while a roll is begun the increments stay pending -- a flush would publish it before it is due: they join
            //  the map in the rollTo of the frame the roll IS due at, right after its publish and before the driver can
            //  begin the next one, so a driver that keeps a roll begun at every frame no longer starves the map: ADVICE r5)
            if (pending >= (size_t)o.append_threshold && !staged_ && !flushIncrements()) return false;
        }
    }
    return true;
}

bool MapManager::registerResident(int frame, int64_t timestamp, const PoseTransform& init, const RegisterOptions& o,
                                  PoseTransform* out, velo_icp_result* result)
{
    if (!ctx_) return false;
    if (!out) {
        err_ = "null output";
        return false;
    }
    return registerCore(frame, timestamp, init, o, out, result);
}

bool MapManager::registerFrame(const HDLFrame& frame, const PoseTransform& init,
                               const RegisterOptions& o, PoseTransform* out,
                               velo_icp_result* result)
{
    if (!ctx_) return false;
    if (!out || frame.numPoints() == 0) {
        err_ = "empty frame or null output";
        return false;
    }
    const CloudView c = frame.getPointsAsOneCloud(0, frame.numBeams());
    const int64_t fs[2] = {0, (int64_t)c.size};
    if (velo_frames_upload(ctx_, 1, c.x, c.y, c.z, fs)) {
        err_ = velo_last_error(ctx_);
        return false;
    }
    return registerCore(0, frame.timestamp, init, o, out, result);
}

}  // namespace veloslam
