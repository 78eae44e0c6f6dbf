// frame_map.cpp -- HDLFrame container and the MapManager that hands
// frames to the registration kernels through the C ABI.  Host-only
// plumbing: no arithmetic beyond tile indexing lives here.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include "../../../include/veloslam/MapManager.hpp"

namespace veloslam {

// ------------------------------------------------------------------ HDLFrame
HDLFrame::HDLFrame()
    : timestamp(VELO_TIME_INVALID), carpose(new PoseTransform), isInMemory(false),
      isOnHardDrive(false), count(0), skips(0)
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

void intrusive_ptr_add_ref(HDLFrame* p) { ++p->count; }
void intrusive_ptr_release(HDLFrame* p)
{
    unsigned char c = p->count.load();
    while (c != 0 && !p->count.compare_exchange_weak(c, (unsigned char)(c - 1))) {
    }
}

// ---------------------------------------------------------------- MapManager
MapManager::MapManager(float patchRange, int device_id)
    : patchRange_(patchRange), ctx_(nullptr), dirty_(true), residentVoxel_(0), residentK_(0)
{
    ctx_ = velo_create(device_id, nullptr);
    if (!ctx_) err_ = velo_last_error(nullptr);
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

void MapManager::addPoints(const float* x, const float* y, const float* z, size_t n)
{
    for (size_t i = 0; i < n; ++i) getPatch(x[i], y[i])->append(x + i, y + i, z + i, 1);
    dirty_ = true;
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
            resident_.erase(it->second);
            it = patches_.erase(it);
            dirty_ = true;
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
    resident_.clear();
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

bool MapManager::syncDeviceMap(double x, double y, const RegisterOptions& o)
{
    auto roi = getROI(x, y);
    if (roi.empty()) {
        err_ = "no map patch within ROI_RANGE of the prior";
        return false;
    }
    if (!dirty_ && roi == resident_ && o.voxel == residentVoxel_ && o.k_normals == residentK_)
        return true;
    std::vector<float> mx, my, mz;
    for (const auto& p : roi) {  // std::set iterates in pointer order; make it positional
        (void)p;
    }
    std::vector<std::shared_ptr<MapPatch>> ordered(roi.begin(), roi.end());
    std::sort(ordered.begin(), ordered.end(), [](const auto& a, const auto& b) {
        return a->centerY != b->centerY ? a->centerY < b->centerY : a->centerX < b->centerX;
    });
    for (const auto& p : ordered) {
        mx.insert(mx.end(), p->x.begin(), p->x.end());
        my.insert(my.end(), p->y.begin(), p->y.end());
        mz.insert(mz.end(), p->z.begin(), p->z.end());
    }
    if (mx.empty()) {
        err_ = "map patches in the ROI hold no points";
        return false;
    }
    if (velo_map_reset(ctx_, mx.data(), my.data(), mz.data(), mx.size(), o.voxel, o.k_normals)) {
        err_ = velo_last_error(ctx_);
        return false;
    }
    resident_ = roi;
    residentVoxel_ = o.voxel;
    residentK_ = o.k_normals;
    dirty_ = false;
    return true;
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
    if (!syncDeviceMap(init.T[0], init.T[1], o)) return false;
    const Affine3x4 T0 = init.getMatrix();
    velo_icp_result local;
    velo_icp_result* r = result ? result : &local;
    const CloudView c = frame.getPointsAsOneCloud(0, frame.numBeams());
    if (velo_icp(ctx_, c.x, c.y, c.z, c.size, T0.data(), o.iters, o.d_max, 1, r)) {
        err_ = velo_last_error(ctx_);
        return false;
    }
    Affine3x4 M;
    std::memcpy(M.data(), r->T, sizeof(double) * 12);
    PoseTransform p = PoseTransform::fromMatrix(M);
    p.timestamp = frame.timestamp;
    for (int i = 0; i < 3; ++i) p.V[i] = init.V[i];
    *out = p;
    if (o.integrate) {
        std::vector<float> ix(c.size), iy(c.size), iz(c.size);
        size_t n_inc = 0;
        if (velo_increment(ctx_, 0, r->T, o.increment_min_count, ix.data(), iy.data(), iz.data(),
                           &n_inc)) {
            err_ = velo_last_error(ctx_);
            return false;
        }
        addPoints(ix.data(), iy.data(), iz.data(), n_inc);
        // keep the device map in step instead of re-uploading the ROI at the next frame: the
        // increment is merged into the sorted map in place (velo_map_append is incremental).  If
        // the ROI's tile set changes, syncDeviceMap rebuilds from the tiles as before.
        if (n_inc == 0 || velo_map_append(ctx_, ix.data(), iy.data(), iz.data(), n_inc) == VELO_OK)
            dirty_ = false;
    }
    return true;
}

}  // namespace veloslam
