// hdl_manager.cpp -- veloslam::HDLManager: the frame store a SLAM front end pulls from
// (HDLManager.cxx:98-260, 383-409) on top of the C ABI.  Host-only plumbing; the parsing,
// calibration and motion compensation behind prepareFrame / prepareResident are velo_decode's
// kernels.  See include/veloslam/HDLManager.hpp for what is and is not carried over.
#include <algorithm>
#include <cstring>
#include <fstream>
#include "../../../include/veloslam/HDLManager.hpp"

namespace veloslam {

HDLManager::HDLManager(velo_ctx* ctx, int capacity)
    : ctx_(ctx), maxCacheSize_((size_t)std::max(capacity, 1)), transMgr_(new TransformManager)
{
    std::memset(corr_, 0, sizeof corr_);
}

HDLManager::~HDLManager()
{
    if (plan_) velo_decode_plan_destroy(plan_);
}

bool HDLManager::setCalibFile(const std::string& filename)
{
    int32_t n_enabled = 0;
    if (velo_load_corrections(filename.c_str(), corr_, &n_enabled) != VELO_OK) {
        err_ = "cannot read the calibration file " + filename;
        haveCalib_ = false;
        return false;
    }
    haveCalib_ = true;
    return true;
}

bool HDLManager::loadOffline(const std::string& insTxt, const std::string& pcapfile)
{
    {
        std::lock_guard<std::mutex> lock(framesMutex_);
        frames_.clear();
    }
    {
        std::lock_guard<std::mutex> lock(cacheMutex_);
        cache_.clear();
    }
    {
        std::lock_guard<std::mutex> lock(decodeMutex_);
        planned_ = nullptr;  // a plan made for the previous capture is void
    }
    if (!transMgr_->loadFromTxtFile(insTxt, true)) {
        err_ = "cannot read the pose track " + insTxt;
        return false;
    }
    poses_ = transMgr_->snapshot();
    size_t n_pkt = 0, n_idx = 0;
    if (velo_pcap_read(pcapfile.c_str(), nullptr, nullptr, 0, &n_pkt) != VELO_OK) {
        err_ = "cannot read the capture " + pcapfile;
        return false;
    }
    packets_.assign(n_pkt * 1206, 0);
    times_.assign(n_pkt, 0);
    if (n_pkt && velo_pcap_read(pcapfile.c_str(), packets_.data(), times_.data(), n_pkt, &n_pkt) != VELO_OK) {
        err_ = "cannot read the capture " + pcapfile;
        return false;
    }
    for (auto& t : times_) t += kClockShiftUs;
    if (velo_pcap_index(pcapfile.c_str(), nullptr, 0, &n_idx) != VELO_OK) {
        err_ = "cannot index the capture " + pcapfile;
        return false;
    }
    std::vector<velo_frame_index>& index = index_;
    index.assign(n_idx, velo_frame_index());
    if (n_idx && velo_pcap_index(pcapfile.c_str(), index.data(), n_idx, &n_idx) != VELO_OK) {
        err_ = "cannot index the capture " + pcapfile;
        return false;
    }
    nPackets_ = n_pkt;
    const int64_t name_time = n_pkt ? times_[0] : VELO_TIME_INVALID;
    for (size_t k = 0; k < n_idx; ++k) {
        const velo_frame_index& e = index[k];
        if (e.t_us == VELO_TIME_INVALID) continue;  // the stub of an empty capture: nothing to look up
        auto f = std::make_shared<HDLFrame>();
        f->timestamp = e.t_us + kClockShiftUs;
        f->fileStartPos = e.file_pos;
        f->skips = (uint8_t)e.firing_skip;
        f->filenameTime = name_time;
        f->isOnHardDrive = true;
        bindToCapture(*f);
        // "readFrameInformation() can't determine carpose for each frame" (HDLManager.cxx:104-109)
        transMgr_->interpolateTransform(f->timestamp, f->carpose.get());
        addFrame(f);
    }
    {
        // stubs hold no points: loading is not "new data" for a waiting consumer, and nothing to cache
        std::lock_guard<std::mutex> lock(framesMutex_);
        hasNewData_ = false;
    }
    return true;
}

void HDLManager::bindToCapture(HDLFrame& f) const
{
    f.firstPacket = -1;
    f.numPackets = 0;
    // the index is in capture order, i.e. ascending in file_pos: a binary search, not a scan per frame
    // (loadOffline / loadHDLMeta call this once per frame: 36 k frames for an hour's drive)
    size_t k = (size_t)(std::lower_bound(index_.begin(), index_.end(), f.fileStartPos,
                                         [](const velo_frame_index& e, int64_t pos) { return e.file_pos < pos; }) -
                        index_.begin());
    for (; k < index_.size() && index_[k].file_pos == f.fileStartPos; ++k) {
        const velo_frame_index& e = index_[k];
        if (e.firing_skip != (int32_t)f.skips) continue;
        f.firstPacket = e.first_packet;
        // up to and including the packet in which the next frame opens; the last frame runs to the end
        const int64_t end = k + 1 < index_.size() ? index_[k + 1].first_packet + 1 : (int64_t)nPackets_;
        f.numPackets = (int32_t)std::max<int64_t>(end - e.first_packet, 0);
        return;
    }
}

bool HDLManager::saveHDLMeta(const std::string& filename)
{
    std::ofstream os(filename, std::ios::binary);
    if (!os) return false;
    for (const auto& f : getAllFrameMeta())   // TimeLine's operator<<: every frame, in time order
        if (!f->writeMeta(os)) return false;
    return (bool)os;
}

bool HDLManager::loadHDLMeta(const std::string& filename)
{
    std::ifstream is(filename, std::ios::binary);
    if (!is) return false;
    for (;;) {
        auto f = std::make_shared<HDLFrame>();
        if (!f->readMeta(is)) break;
        bindToCapture(*f);
        addFrame(f);   // a stamp already in the store is overwritten, as TimeLine::addData does
    }
    std::lock_guard<std::mutex> lock(framesMutex_);
    hasNewData_ = false;
    return true;
}

bool HDLManager::loadINSMeta(const std::string& filename)
{
    if (!transMgr_->loadFromMetaFile(filename, false)) return false;
    poses_ = transMgr_->snapshot();
    return true;
}

int HDLManager::getNumberOfFrames()
{
    std::lock_guard<std::mutex> lock(framesMutex_);
    return (int)frames_.size();
}

int HDLManager::getNumberOfTransforms() { return transMgr_->getNumberOfTransforms(); }

size_t HDLManager::lowerBound(int64_t t) const
{
    return (size_t)(std::lower_bound(frames_.begin(), frames_.end(), t,
                                     [](const std::shared_ptr<HDLFrame>& f, int64_t v) { return f->timestamp < v; }) -
                    frames_.begin());
}

void HDLManager::addFrame(std::shared_ptr<HDLFrame> frame)
{
    if (!frame) return;
    {
        std::lock_guard<std::mutex> lock(framesMutex_);
        const size_t i = lowerBound(frame->timestamp);
        if (i < frames_.size() && frames_[i]->timestamp == frame->timestamp)
            frames_[i] = frame;  // TimeLine::addData: "Old data will be OVERWRITTEN" (TimeLine.h:151-155, 205-208)
        else
            frames_.insert(frames_.begin() + (ptrdiff_t)i, frame);
        hasNewData_ = true;
    }
    cond_.notify_one();
    if (frame->isInMemory) pushCache(frame);
}

bool HDLManager::fillPlan(const HDLFrame& f)
{
    planned_ = nullptr;
    if (!ctx_) {
        err_ = "no device context";
        return false;
    }
    if (!haveCalib_) {
        err_ = "Corrections have not been set";  // HDLParser.cxx:513-516
        return false;
    }
    if (f.firstPacket < 0 || f.numPackets <= 0 ||
        (size_t)(f.firstPacket + f.numPackets) > times_.size()) {
        err_ = "the frame is not part of the loaded capture";
        return false;
    }
    if (!plan_ && velo_decode_plan_create(ctx_, &plan_) != VELO_OK) {
        err_ = std::string("decode failed: ") + velo_last_error(ctx_);
        return false;
    }
    velo_decode_opts dop;
    std::memset(&dop, 0, sizeof dop);
    dop.struct_size = sizeof dop;
    dop.initial_firing_skip = f.skips;  // HDLParser::getFrame's `skip` (HDLParser.cxx:527)
    std::memset(dop.laser_selection, 1, sizeof dop.laser_selection);
    const size_t p0 = (size_t)f.firstPacket;
    const bool last = p0 + (size_t)f.numPackets >= times_.size();
    // a frame that closes inside its last packet is emitted by the split; the last frame of the
    // capture by the flush (HDLParser::getFrame's tail, HDLParser.cxx:539-543)
    if (velo_decode_plan_fill(plan_, &dop, packets_.data() + p0 * 1206, times_.data() + p0, (size_t)f.numPackets, corr_,
                              64, poses_.data(), poses_.size(), last ? 1 : 0, nullptr, 0) != VELO_OK) {
        err_ = std::string("decode failed: ") + velo_decode_plan_error(plan_);
        return false;
    }
    planned_ = &f;
    plannedFirst_ = f.firstPacket;
    plannedSkip_ = f.skips;
    return true;
}

bool HDLManager::planResident(const std::shared_ptr<HDLFrame>& frame)
{
    if (!frame || !frame->isOnHardDrive) {
        err_ = "not a frame of the capture";
        return false;
    }
    std::lock_guard<std::mutex> lock(decodeMutex_);
    return fillPlan(*frame);
}

// ---- the decode behind prepareFrame / prepareResident ------------------------------------
bool HDLManager::decodeFrame(const HDLFrame& f, bool to_frames, size_t* npts, int* n_decoded, bool overlapped)
{
    // (planned ahead: only the device half is left; the key is the frame AND where it sits in the
    // capture -- an address alone can come back with another frame after a reload)
    const bool planned = planned_ == &f && plannedFirst_ == f.firstPacket && plannedSkip_ == f.skips;
    if (!planned && !fillPlan(f)) return false;
    planned_ = nullptr;
    int32_t nf = 0;
    size_t n = 0;
    if (overlapped) {  // device half + adoption on the side stream, next to the running registration
        if (velo_decode_submit_overlapped(ctx_, plan_, &nf, &n) != VELO_OK) {
            err_ = std::string("decode failed: ") + velo_last_error(ctx_);
            return false;
        }
        if (npts) *npts = n;
        if (n_decoded) *n_decoded = nf;
        return true;
    }
    if (velo_decode_submit(ctx_, plan_, &nf, &n) != VELO_OK) {
        err_ = std::string("decode failed: ") + velo_last_error(ctx_);
        return false;
    }
    if (nf < 1) {
        err_ = "the packets of the frame hold no complete revolution";
        return false;
    }
    if (to_frames && velo_decode_to_frames(ctx_) != VELO_OK) {
        err_ = std::string("decode failed: ") + velo_last_error(ctx_);
        return false;
    }
    if (npts) *npts = n;
    if (n_decoded) *n_decoded = nf;
    return true;
}

bool HDLManager::prepareResident(const std::shared_ptr<HDLFrame>& frame, size_t* points)
{
    if (!frame || !frame->isOnHardDrive) {
        err_ = "not a frame of the capture";
        return false;
    }
    std::lock_guard<std::mutex> lock(decodeMutex_);
    size_t n = 0;
    int nf = 0;
    if (!decodeFrame(*frame, true, &n, &nf)) return false;
    if (nf != 1) {
        // the packets of one index entry hold one revolution: its closing split, or the flush
        err_ = "the packets of the frame decode to more than one revolution";
        return false;
    }
    if (points) *points = n;
    return true;
}

bool HDLManager::prepareResidentDuringRegistration(const std::shared_ptr<HDLFrame>& frame, size_t* points)
{
    if (!frame || !frame->isOnHardDrive) {
        err_ = "not a frame of the capture";
        return false;
    }
    std::lock_guard<std::mutex> lock(decodeMutex_);
    size_t n = 0;
    int nf = 0;
    if (!decodeFrame(*frame, true, &n, &nf, true)) return false;
    if (nf != 1) {
        err_ = "the packets of the frame decode to more than one revolution";
        return false;
    }
    if (points) *points = n;
    return true;
}

FrameRef HDLManager::prepareFrame(std::shared_ptr<HDLFrame> frame)
{
    if (!frame) return FrameRef();
    auto held_if_in_memory = [&]() {
        // the reference takes the pointer first and counts it later, so the cache may clear the frame
        // in between ("MIGHT BE safe", HDLManager.h:181-186); here the check and the count are one step
        // as far as updateCacheSize is concerned
        std::lock_guard<std::mutex> lock(cacheMutex_);
        return frame->isInMemory ? FrameRef(frame) : FrameRef();
    };
    if (FrameRef r = held_if_in_memory()) return r;
    if (!frame->isOnHardDrive) return FrameRef();
    {
        std::lock_guard<std::mutex> lock(decodeMutex_);
        if (FrameRef r = held_if_in_memory()) return r;  // another consumer got there first
        size_t n = 0;
        int nf = 0;
        if (!decodeFrame(*frame, false, &n, &nf)) return FrameRef();
        if (nf != 1) {
            err_ = "the packets of the frame decode to more than one revolution";
            return FrameRef();
        }
        std::vector<float> x(n), y(n), z(n), in(n), dist(n);
        std::vector<uint16_t> az(n), pk(n);
        std::vector<int64_t> fstart(2, 0), ft(1, 0);
        std::vector<int32_t> bstart(65, 0), fpk(1, 0);
        std::vector<velo_pose> car(1);
        if (velo_decode_fetch(ctx_, x.data(), y.data(), z.data(), in.data(), az.data(), dist.data(), pk.data(),
                              fstart.data(), bstart.data(), car.data(), ft.data(), fpk.data()) != VELO_OK) {
            err_ = std::string("fetch failed: ") + velo_last_error(ctx_);
            return FrameRef();
        }
        const size_t a = (size_t)fstart[0], b = (size_t)fstart[1];
        int32_t beams[65];
        for (int i = 0; i < 65; ++i) beams[i] = bstart[(size_t)i] - (int32_t)a;
        frame->setPoints(x.data() + a, y.data() + a, z.data() + a, in.data() + a, pk.data() + a, beams, 64);
        frame->pointsMeta.resize(b - a);
        for (size_t i = a; i < b; ++i) {
            PointMeta& m = frame->pointsMeta[i - a];
            m.azimuth = az[i];
            m.distance = dist[i];
            m.intensityFlag = m.distanceFlag = m.flags = 0;
        }
        // the pose the points were compensated to (HDLParser.cxx:993-1001); an empty track leaves the
        // stub's pose alone
        if (car[0].seconds_pos != -1) *frame->carpose = PoseTransform::fromC(car[0]);
    }
    FrameRef held(frame);  // counted before the cache sees it
    pushCache(frame);
    return held;
}

// ---- consumers ----------------------------------------------------------------------------
FrameRef HDLManager::waitForFrame(std::chrono::microseconds micro)
{
    std::shared_ptr<HDLFrame> f;
    {
        std::unique_lock<std::mutex> lock(framesMutex_);
        cond_.wait_for(lock, micro, [&] { return hasNewData_; });
        if (!hasNewData_ || frames_.empty()) return FrameRef();
        hasNewData_ = false;
        f = frames_.back();
    }
    return prepareFrame(f);
}

FrameRef HDLManager::getRecentFrame()
{
    std::shared_ptr<HDLFrame> f;
    {
        std::lock_guard<std::mutex> lock(framesMutex_);
        if (frames_.empty()) return FrameRef();
        f = frames_.back();
    }
    return prepareFrame(f);
}

FrameRef HDLManager::getFrameAt(int64_t t_us)
{
    std::shared_ptr<HDLFrame> f;
    {
        std::lock_guard<std::mutex> lock(framesMutex_);
        const size_t i = lowerBound(t_us);
        if (i >= frames_.size() || frames_[i]->timestamp != t_us) return FrameRef();
        f = frames_[i];
    }
    return prepareFrame(f);
}

size_t HDLManager::nearestIndex(int64_t t) const
{
    // TimeLine::getNearestData (TimeLine.h:284-375): the ends clamp, the earlier neighbour wins only
    // when it is STRICTLY nearer
    const size_t i = lowerBound(t);
    if (i == 0) return 0;
    if (i >= frames_.size()) return frames_.size() - 1;
    if (frames_[i]->timestamp == t) return i;
    return (t - frames_[i - 1]->timestamp) < (frames_[i]->timestamp - t) ? i - 1 : i;
}

FrameRef HDLManager::getFrameNear(int64_t t_us)
{
    std::shared_ptr<HDLFrame> f;
    {
        std::lock_guard<std::mutex> lock(framesMutex_);
        if (frames_.empty()) return FrameRef();
        f = frames_[nearestIndex(t_us)];
    }
    return prepareFrame(f);
}

std::vector<std::shared_ptr<HDLFrame>> HDLManager::getAllFrameMeta()
{
    std::lock_guard<std::mutex> lock(framesMutex_);
    return frames_;
}

std::vector<FrameRef> HDLManager::getRangeBetween(int64_t a_us, int64_t b_us)
{
    std::vector<std::shared_ptr<HDLFrame>> pick;
    {
        std::lock_guard<std::mutex> lock(framesMutex_);
        if (frames_.empty()) return {};
        const size_t ia = nearestIndex(a_us), ib = nearestIndex(b_us);
        for (size_t i = ia; i <= ib; ++i) pick.push_back(frames_[i]);
    }
    std::vector<FrameRef> out;
    out.reserve(pick.size());
    for (auto& f : pick) out.push_back(prepareFrame(f));
    return out;
}

// ---- the cache ----------------------------------------------------------------------------
void HDLManager::pushCache(const std::shared_ptr<HDLFrame>& frame)
{
    {
        std::lock_guard<std::mutex> lock(cacheMutex_);
        cache_.push_back(frame);
    }
    updateCacheSize();
}

void HDLManager::updateCacheSize()
{
    std::lock_guard<std::mutex> lock(cacheMutex_);
    int putBackTimes = 10;  // HDLManager.cxx:385
    while (cache_.size() > maxCacheSize_ && putBackTimes) {
        std::shared_ptr<HDLFrame> f = cache_.front();
        cache_.pop_front();
        if (f->count.load() != 0) {
            cache_.push_back(f);
            --putBackTimes;
        } else {
            // the stub stays in the store: a frame of the capture is decoded again by the next
            // prepareFrame; one that came in through addFrame and exists nowhere else is gone (as in
            // the reference: this is what bounds the memory of an online run without disk swap)
            f->clear();
        }
    }
}

int HDLManager::cachedFrames()
{
    std::lock_guard<std::mutex> lock(cacheMutex_);
    return (int)cache_.size();
}

}  // namespace veloslam
