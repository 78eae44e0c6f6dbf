// TransformManager.hpp -- time-indexed pose store with linear interpolation: the
// reference's TransformManager (TransformManager.h:82-123) over
// TimeLine<PoseTransform> (TimeLine.h), re-designed as one sorted flat array
// with a cached cursor (sequential queries are O(1), random ones O(log n);
// the reference measured 3-4 us / 23 us per call, TransformManager.cxx:143-146).
// Bracket selection reproduces TimeLine::getBoundaryData (TimeLine.h:384-468)
// including its behaviour at the ends (extrapolation from the first/last two
// samples) and at exact knots; interpolation reproduces
// TransformManager.cxx:149-177 (Euler angles lerped in degrees, no wrap).
#pragma once
#include <cstddef>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "PoseTransform.hpp"

namespace veloslam {

class TransformManager {
public:
    TransformManager();
    ~TransformManager();
    int getNumberOfTransforms();
    void clearTransforms();
    void addTransform(std::shared_ptr<PoseTransform> trans);
    void addTransform(const PoseTransform& trans);
    // returns false only when the store is empty; otherwise fills *xform.  With a
    // single sample the result extrapolates along V and xform->seconds_pos is
    // left untouched (the reference's "not valid" signal, TransformManager.cxx:159-167).
    bool interpolateTransform(int64_t t_us, PoseTransform* xform);
    // carposes.txt rows "x y yaw roll pitch v sec usec" (TransformManager.cxx:95-125)
    bool loadFromTxtFile(const std::string& filename, bool clearOldData = false);
    // .insmeta (TransformManager.cxx:81-93, 127-134): the store as a stream of pose records
    // (type_defs.cxx:4-33; ptime -> int64 microseconds), in time order
    bool loadFromMetaFile(const std::string& filename, bool clearOldData = false);
    bool writeToMetaFile(const std::string& filename);
    void setOriginLLH(const double LLH[3]);  // TransformManager.cxx:179-185
    const double* originXYZ() const { return originXYZ_; }
    // per-packet transform table for K1 (HDLParser.cxx:988-1007)
    bool packetTransforms(const int64_t* pkt_t_us, size_t n_pkt, double* T3x4, uint8_t* valid,
                          PoseTransform* carpose);
    // snapshot for the C ABI
    std::vector<velo_pose> snapshot();

    TransformManager(const TransformManager&) = delete;
    void operator=(const TransformManager&) = delete;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    std::mutex mutex_;
    double originLLH_[3], originXYZ_[3];
};

// A time-sorted velo_pose array read in place with the same bracket / interpolation rules: what the
// C entry points (velo_interp_pose, velo_packet_transforms, velo_decode*) are handed.  O(log n) per
// query, nothing copied; const and stateless, so one view may serve several threads.
class SortedPoseView {
public:
    SortedPoseView(const velo_pose* sorted, size_t n);
    bool interpolate(int64_t t_us, PoseTransform* xform) const;
    bool packetTransforms(const int64_t* pkt_t_us, size_t n_pkt, double* T3x4, uint8_t* valid,
                          PoseTransform* carpose) const;

private:
    int bucket(size_t i) const { return (int)((double)(p_[i].t_us - p_[0].t_us) / interval_); }
    const velo_pose* p_;
    size_t n_;
    double interval_ = 0;
    bool strict_ = true;
};

}  // namespace veloslam
