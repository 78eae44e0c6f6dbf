// HDLManager.hpp -- the consumer side of the reference's HDLManager (HDLManager.h:104-147,
// HDLManager.cxx:98-260): the time-indexed store of LiDAR frames a SLAM front end pulls from
// (getFrameAt / getFrameNear / getRangeBetween / getRecentFrame / waitForFrame), its offline
// loader (loadOffline: carposes.txt + pcap -> one frame stub per revolution, points decoded on
// demand) and the frame cache that clears the points of frames nobody holds.
//
// What is different underneath: prepareFrame() does not re-open the capture and run a CPU parser
// (HDLParser::getFrame, HDLParser.cxx:505-544) -- the capture is read once into host memory, and
// a frame's packets go to the MI355X, which decodes, calibrates, splits and motion-compensates
// them (velo_decode).  prepareResident() stops there: the frame stays in HBM as
// resident frame 0 of the context, ready for MapManager::registerResident -- packets in, pose out,
// the points never visit the host.  prepareFrame() additionally copies them into the HDLFrame.
//
// Out of scope here as in the rest of this library: the UDP sources (HDLSource /
// INSSource), the disk swap of live captures (writePackets / switchBuffer) and the .hdlmeta
// scan.  An online producer hands finished frames in through addFrame(), which is what
// HDLSource's parser thread does (HDLSource.cxx:209-225 -> HDLManager::addFrame).
//
// Time stamps are microseconds on the reference's clock: loadOffline adds the 8 hours of
// timevalToPtime (type_defs.cxx:69-72) to packet and pose stamps alike.
#pragma once
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../velo.h"
#include "HDLFrame.hpp"
#include "TransformManager.hpp"

namespace veloslam {

// What boost::intrusive_ptr<HDLFrame> was to the reference's callers: holding one keeps the
// frame's points in memory (HDLFrame::count, looked at by updateCacheSize()).
class FrameRef {
public:
    FrameRef() = default;
    explicit FrameRef(std::shared_ptr<HDLFrame> f) : f_(std::move(f)) { if (f_) intrusive_ptr_add_ref(f_.get()); }
    FrameRef(const FrameRef& o) : f_(o.f_) { if (f_) intrusive_ptr_add_ref(f_.get()); }
    FrameRef(FrameRef&& o) noexcept : f_(std::move(o.f_)) { o.f_.reset(); }
    FrameRef& operator=(FrameRef o) noexcept { f_.swap(o.f_); return *this; }
    ~FrameRef() { if (f_) intrusive_ptr_release(f_.get()); }
    HDLFrame* get() const { return f_.get(); }
    HDLFrame* operator->() const { return f_.get(); }
    HDLFrame& operator*() const { return *f_; }
    explicit operator bool() const { return (bool)f_; }
    const std::shared_ptr<HDLFrame>& shared() const { return f_; }

private:
    std::shared_ptr<HDLFrame> f_;
};

class HDLManager {
public:
    static constexpr int64_t kClockShiftUs = 8LL * 3600 * 1000000;  // timevalToPtime

    // ctx: the context whose GPU decodes (borrowed; e.g. MapManager::context()).  NULL is allowed
    // for a store of frames that are already in memory (addFrame + the time queries).
    // capacity: frames whose points are kept in host memory (HDLManager.cxx:47: 200).
    explicit HDLManager(velo_ctx* ctx, int capacity = 200);
    ~HDLManager();

    // HDLManager::setCalibFile (HDLManager.cxx:183-187) -> HDLParser::setCorrectionsFile: db.xml
    bool setCalibFile(const std::string& filename);
    // HDLManager::loadOffline (HDLManager.cxx:98-112): pose track, frame index of the capture, one
    // stub per frame {timestamp, fileStartPos, skips, carpose = track interpolated at timestamp,
    // isOnHardDrive}.  The reference returns void and prints; this returns false and sets lastError().
    bool loadOffline(const std::string& insTxt, const std::string& pcapfile);

    // .hdlmeta / .insmeta (HDLManager.cxx:411-449; the reference names the files by the clock and
    // finds them by scanning its buffer directory -- here the caller names them): the frame stubs
    // and the pose track of a session.  Stubs read back are matched to the capture loadOffline
    // holds by (fileStartPos, skips), so that they can be prepared; a stub of some other file
    // stays a stub.
    bool saveHDLMeta(const std::string& filename);
    bool loadHDLMeta(const std::string& filename);
    bool saveINSMeta(const std::string& filename) { return transMgr_->writeToMetaFile(filename); }
    bool loadINSMeta(const std::string& filename);

    int getNumberOfFrames();
    int getNumberOfTransforms();
    void addFrame(std::shared_ptr<HDLFrame> frame);  // HDLManager.cxx:189-205 (no file-buffer mode)

    // HDLManager::prepareFrame (HDLManager.cxx:207-224): in memory -> as is; on the capture ->
    // decoded (on the GPU) into the frame: x / y / z / intensity beam-major, pointsMeta, packetIndex,
    // carpose = the pose the frame was compensated to; empty FrameRef when neither or on failure.
    FrameRef prepareFrame(std::shared_ptr<HDLFrame> frame);
    // The device-side half only: decode + compensate, frame left RESIDENT in HBM as frame 0 of the
    // context (until the next decode on it); `frame` gets no points.  points (optional) = its size.
    bool prepareResident(const std::shared_ptr<HDLFrame>& frame, size_t* points = nullptr);
    // The HOST half of prepareResident(frame), ahead of time: the sequential part of the parser (pose
    // per packet, frame split, block owners) staged in pinned memory (velo_decode_plan_fill).  It
    // touches neither the context nor the GPU, so it can run while the previous frame is being
    // registered (RegisterOptions::while_registering); the prepareResident / prepareFrame of the same
    // frame that follows then only submits the device half.  A plan for another frame is discarded.
    bool planResident(const std::shared_ptr<HDLFrame>& frame);
    // prepareResident for the NEXT frame while the context registers the current one (call it from
    // RegisterOptions::while_registering): host half, then the device half on the context's second
    // stream, concurrently with the registration (velo_decode_submit_overlapped).  The frame is
    // resident, and everything enqueued afterwards sees it complete, when the call returns.
    bool prepareResidentDuringRegistration(const std::shared_ptr<HDLFrame>& frame, size_t* points = nullptr);

    // HDLManager.cxx:226-260.  waitForFrame blocks up to `micro` for an addFrame().
    FrameRef waitForFrame(std::chrono::microseconds micro = std::chrono::microseconds(100000));
    FrameRef getRecentFrame();
    FrameRef getFrameAt(int64_t t_us);    // exact stamp or empty (TimeLine::getExactDataAt)
    FrameRef getFrameNear(int64_t t_us);  // nearest stamp, the later one on a tie (TimeLine::getNearestData)
    // meta only -- "DO NOT ACCESS POINTS DATA VIA THIS METHOD" (HDLManager.h:139-142)
    std::vector<std::shared_ptr<HDLFrame>> getAllFrameMeta();
    // every frame from the one nearest a to the one nearest b, both included: the contract
    // HDLManager.h:144 states ("inclusive on both end").  (TimeLine::getRangeBetween's body,
    // TimeLine.h:478-495, leaves the frame nearest b out unless a and b share a bucket, in which
    // case it runs on to the end of that bucket; the stated contract is what is built.)
    std::vector<FrameRef> getRangeBetween(int64_t a_us, int64_t b_us);

    // frame cache (HDLManager.cxx:383-409): at most `capacity` frames keep their points; the oldest
    // unreferenced ones are clear()ed.  A frame somebody holds (count != 0) is put back, ten times
    // at most per call, like the reference.
    void pushCache(const std::shared_ptr<HDLFrame>& frame);
    void updateCacheSize();
    void cleanCache() { updateCacheSize(); }
    int cachedFrames();

    std::shared_ptr<TransformManager> transformManager() { return transMgr_; }
    velo_ctx* context() { return ctx_; }
    const char* lastError() const { return err_.c_str(); }

    HDLManager(const HDLManager&) = delete;
    void operator=(const HDLManager&) = delete;

private:
    bool decodeFrame(const HDLFrame& f, bool to_frames, size_t* npts, int* n_decoded, bool overlapped = false);
    bool fillPlan(const HDLFrame& f);
    velo_decode_plan* plan_ = nullptr;
    const HDLFrame* planned_ = nullptr;   // the frame plan_ is filled for ...
    int64_t plannedFirst_ = -1;           // ... and where it sits in the capture
    int plannedSkip_ = 0;
    size_t lowerBound(int64_t t) const;       // first frame with timestamp >= t
    size_t nearestIndex(int64_t t) const;     // frames_ not empty

    velo_ctx* ctx_;
    size_t maxCacheSize_;
    std::shared_ptr<TransformManager> transMgr_;
    std::vector<std::shared_ptr<HDLFrame>> frames_;  // sorted by timestamp
    std::deque<std::shared_ptr<HDLFrame>> cache_;
    std::mutex framesMutex_, cacheMutex_, decodeMutex_;
    std::condition_variable cond_;
    bool hasNewData_ = false;
    // the capture (loadOffline)
    std::vector<uint8_t> packets_;
    std::vector<int64_t> times_;       // + kClockShiftUs
    std::vector<velo_pose> poses_;     // snapshot of the pose track for velo_decode
    std::vector<velo_frame_index> index_;
    size_t nPackets_ = 0;
    void bindToCapture(HDLFrame& f) const;  // (fileStartPos, skips) -> firstPacket / numPackets
    velo_laser_corr corr_[64];
    bool haveCalib_ = false;
    std::string err_;
};

}  // namespace veloslam
