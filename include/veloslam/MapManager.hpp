// MapManager.hpp -- 2-D patch index over MapPatch tiles plus the one method the
// north star needs: registerFrame().  Keeps the reference's member names and
// signatures (MapManager.h:15-48: getPatch / findPatch / getROI, ROI_RANGE), made
// public -- in the reference the whole class is private and its cv::Mat index is
// never allocated, so nothing can call it.  The cv::Mat index is
// replaced by an ordered map keyed on the patch grid index.
#pragma once
#include <map>
#include <memory>
#include <set>
#include <string>
#include <utility>
#include "../velo.h"
#include "HDLFrame.hpp"
#include "MapPatch.hpp"
#include "PoseTransform.hpp"

#define ROI_RANGE 100  // MapManager.h:13 -- sensor detecting range, metres

namespace veloslam {

struct RegisterOptions {
    int iters = 20;
    float d_max = 1.0f;
    float voxel = 1.0f;
    int k_normals = 16;
    int increment_min_count = 3;  // cells with fewer points accept new points
    bool integrate = false;       // append the accepted increment to the map afterwards
};

class MapManager {
public:
    // patchRange: tile edge (m).  device_id: which MI355X runs the kernels.
    explicit MapManager(float patchRange = 2.0f * ROI_RANGE, int device_id = 0);
    ~MapManager();
    std::shared_ptr<MapPatch> getPatch(double x, double y);   // creates if missing
    std::shared_ptr<MapPatch> findPatch(double x, double y);  // null if missing
    std::set<std::shared_ptr<MapPatch>> getROI(double x, double y);  // 4-corner lookup
    // insert map-frame points into their tiles
    void addPoints(const float* x, const float* y, const float* z, size_t n);
    // Motion-compensated frame (frame-start origin, ENU axes) -> pose in the map.
    // init: prior (e.g. frame.carpose).  Returns false (and leaves *out untouched)
    // on error; lastError() says why.  result (optional) gets per-iteration stats.
    bool registerFrame(const HDLFrame& frame, const PoseTransform& init, const RegisterOptions& opts,
                       PoseTransform* out, velo_icp_result* result = nullptr);
    // f3: drop every tile whose centre is further than `radius` (+ half a tile diagonal) from
    // (x, y) -- the rolling-map policy behind ROI_RANGE; returns the number of points dropped
    size_t evictOutside(double x, double y, double radius);
    // f3: persistence.  Manager header and per-patch header follow the reference's stream
    // operators (MapManager.cxx:81-110, MapPatch.cxx:3-69: centerX, centerY, range[, patchRange],
    // u16 count; per patch centerX, centerY, range and four u16 feature counts, written as 0);
    // each patch is then followed by its point payload: u64 n, x[n], y[n], z[n] (f32).
    bool save(const std::string& filename) const;
    bool load(const std::string& filename);
    size_t numPoints() const;
    const char* lastError() const;
    size_t numPatches() const { return patches_.size(); }
    velo_ctx* context() { return ctx_; }

private:
    std::pair<int, int> getPatchIdx(double x, double y) const;
    bool syncDeviceMap(double x, double y, const RegisterOptions& opts);
    float patchRange_;
    std::map<std::pair<int, int>, std::shared_ptr<MapPatch>> patches_;
    velo_ctx* ctx_;
    bool dirty_;
    std::set<std::shared_ptr<MapPatch>> resident_;
    float residentVoxel_;
    int residentK_;
    mutable std::string err_;
};

}  // namespace veloslam
