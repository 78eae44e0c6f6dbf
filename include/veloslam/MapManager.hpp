// MapManager.hpp -- 2-D patch index over MapPatch tiles plus the one method the
// north star needs: registerFrame().  Keeps the reference's member names and
// signatures (MapManager.h:15-48: getPatch / findPatch / getROI, ROI_RANGE), made
// public -- in the reference the whole class is private and its cv::Mat index is
// never allocated, so nothing can call it.  The cv::Mat index is
// replaced by an ordered map keyed on the patch grid index.
#pragma once
#include <functional>
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
    bool integrate = false;       // collect the accepted increment (device-side pending list) and
                                  // merge it into the map once append_threshold points are pending
    int append_threshold = 512;   // (1 = after every frame)
    // Work to do while the GPU registers: called once, after the registration (and the increment)
    // have been enqueued and before their result is waited for.  It may use the context: e.g.
    // HDLManager::prepareResident(next frame) -- the next frame's packets go up and are decoded right
    // behind this frame's registration, with no idle GPU in between, and are the resident frames when
    // registerResident is called next (velo_icp_batch_start / _finish keep the result apart from that).
    // What it must not do is register, or change the map.
    std::function<void()> while_registering;
    // ---- mapping (BASELINE configs[2] as SLAM: the map GROWS from the frames; README.md:25 "[ ] Implement various SLAM
    // algorithms", MapManager.h:13,43) -- with integrate:
    // increments_in_roi_only: an accepted point outside the resident tile rectangle is dropped instead of waiting in its
    //   host tile (the device map holds nothing there, so EVERY frame would add its far returns to those tiles until they
    //   enter: the min_count rule only works where the map is resident; the point is seen again from nearer).
    // pipeline_increments: the previous frame's increment joins the device map BESIDE this frame's registration
    //   (velo_map_roll_begin on the roll's own stream, published before this frame's increment is taken): a frame is
    //   registered against the map up to the frame before last, its increment is computed against the map up to the last
    //   frame.  The same roll moves the tile rectangle to the NEXT frame's prior when next_prior says where that is (the
    //   pose track knows), so the rollTo of the next frame finds nothing to do.  Deterministic (nothing depends on how
    //   far the GPU has got); tests/test_gpu_parity.py runs the same schedule on the oracle.
    bool increments_in_roi_only = false;
    bool pipeline_increments = false;
    bool have_next_prior = false;
    double next_prior_x = 0, next_prior_y = 0;
};

// what the device map has been through (rolling-map bookkeeping; diagnostic)
struct MapStats {
    uint64_t full_builds = 0;        // velo_map_reset: first ROI, or voxel / k changed, or nothing could be kept
    uint64_t rolls = 0;              // ROI changes applied incrementally (evict + append entering tiles)
    uint64_t rolls_ahead = 0;        // ... of which beside the previous frame's registration (rollAhead)
    uint64_t rolls_refused = 0;      // rollAhead / rollBegin attempts the library refused (a map without normals whose grid would move; since round 5 a re-anchor is not one, since round 6 a hashed table neither): done by the plain roll
    uint64_t rolls_begun = 0;        // ... begun several frames ahead (rollBegin) and published when due
    uint64_t tiles_entered = 0, tiles_left = 0;
    uint64_t points_uploaded = 0;    // host tile points sent to the device by rolls
    uint64_t points_evicted = 0;
    uint64_t increment_flushes = 0, increment_points = 0;
    uint64_t increment_dropped = 0;  // accepted points outside the resident rectangle (increments_in_roi_only)
    uint64_t map_updates = 0;        // device-map updates that carried increments (appends, rolls with increments folded in)
    uint64_t updates_beside = 0;     // ... of which beside a registration (pipeline_increments)
};

class MapManager {
public:
    // patchRange: tile edge (m).  device_id: which MI355X runs the kernels.
    explicit MapManager(float patchRange = 2.0f * ROI_RANGE, int device_id = 0);
    ~MapManager();
    std::shared_ptr<MapPatch> getPatch(double x, double y);   // creates if missing
    std::shared_ptr<MapPatch> findPatch(double x, double y);  // null if missing
    std::set<std::shared_ptr<MapPatch>> getROI(double x, double y);  // 4-corner lookup
    // Every tile that overlaps the square of +-ROI_RANGE around (x, y), in (row, column) order: the
    // same region getROI's four corners name when patchRange >= 2 * ROI_RANGE, and the rule that
    // keeps covering "the current detecting range" (MapManager.h:22-25) when tiles are smaller.
    std::vector<std::shared_ptr<MapPatch>> tilesInRange(double x, double y);
    // insert map-frame points into their tiles
    void addPoints(const float* x, const float* y, const float* z, size_t n);
    // Motion-compensated frame (frame-start origin, ENU axes) -> pose in the map.
    // init: prior (e.g. frame.carpose).  Returns false (and leaves *out untouched)
    // on error; lastError() says why.  result (optional) gets per-iteration stats.
    bool registerFrame(const HDLFrame& frame, const PoseTransform& init, const RegisterOptions& opts,
                       PoseTransform* out, velo_icp_result* result = nullptr);
    // The same for the frame a velo_decode* + velo_decode_to_frames on context() left resident in
    // HBM (frame index `frame` of that decode): packets in, pose out, the points never visit the host.
    bool registerResident(int frame, int64_t timestamp, const PoseTransform& init, const RegisterOptions& opts,
                          PoseTransform* out, velo_icp_result* result = nullptr);
    // Rolling map (BASELINE configs[2]; the reference's eviction policy is a stub, MapManager.h:43):
    // bring the DEVICE map to the tiles in range of (x, y).  The first call builds it; afterwards a
    // changed tile set is applied in place -- velo_map_evict_outside for the tiles that left (the
    // tile range is a box), one velo_map_append of the tiles that entered -- never a re-upload.
    // registerFrame / registerResident call it with the prior's position.
    bool rollTo(double x, double y, const RegisterOptions& opts);
    // rollTo for the NEXT frame's prior while the context registers the current one (call it from
    // RegisterOptions::while_registering, after the next frame's decode): the eviction and the append
    // run on the context's second stream beside the registration (velo_map_roll_overlapped).  The
    // pending increments stay pending (they join the map at the next flush or plain roll).  Returns
    // false without having changed anything when the roll cannot be done that way (first ROI, a jump,
    // a re-anchor ...): the rollTo inside the next registerFrame / registerResident then does it.
    bool rollAhead(double x, double y, const RegisterOptions& opts);
    // The roll BEGUN SEVERAL FRAMES AHEAD (velo_map_roll_begin): call it from while_registering with the prior
    // of a LATER frame -- the first of the next few whose tile rectangle differs (needsRoll tells).  The roll is
    // enqueued on a stream of its own and holds the host for its first count only; the frames registered
    // meanwhile keep reading the map as it was; the rollTo of the frame whose prior names the begun rectangle
    // publishes it (velo_map_roll_publish: a device-side wait).  While a roll is begun the pending increments
    // stay pending (a flush would publish it early); a prior that asks for any other rectangle publishes it and
    // rolls on from there.  Returns false without having changed anything when it cannot be done that way.
    bool rollBegin(double x, double y, const RegisterOptions& opts);
    // does the prior (x, y) name another tile rectangle than the resident one -- and no roll is begun yet?
    bool needsRoll(double x, double y) const;
    bool rollBegun() const { return staged_; }
    // The pending increments (device-side list) -> the host tiles, and -> the device map for those
    // that lie in resident tiles (one velo_map_append).  A roll does not call this: it takes the list
    // and folds the points into the ONE append that brings the entering tiles up (rollTo).
    bool flushIncrements();
    // Every point of `frame` (host points: HDLManager::prepareFrame), transformed by `pose` (fp64 fma chain, rounded once
    // to float: the increment's arithmetic), into the host tiles: the SEED of a map that is then grown from accepted
    // increments only (the first frame of a drive at its prior).  The device map is built from the tiles at the next roll.
    bool seedFromFrame(const HDLFrame& frame, const PoseTransform& pose);
    const MapStats& stats() const { return stats_; }
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
    bool registerCore(int frame, int64_t timestamp, const PoseTransform& init, const RegisterOptions& opts,
                      PoseTransform* out, velo_icp_result* result);
    void tileRange(double x, double y, int& i0, int& i1, int& j0, int& j1) const;
    // Does a tile of the resident rectangle outside [i0..i1] x [j0..j1] hold points / does a tile of that rectangle
    // outside the resident one?  The device map holds exactly what the host tiles of its rectangle hold, so a roll
    // whose leaving tiles are all empty has nothing to evict (the eviction would be three passes over the map that
    // find nothing: 0.3 ms), and one whose entering tiles are all empty has nothing to append.
    bool leavingTilesHoldPoints(int i0, int i1, int j0, int j1) const;
    bool enteringTilesHoldPoints(int i0, int i1, int j0, int j1) const;
    float patchRange_;
    std::map<std::pair<int, int>, std::shared_ptr<MapPatch>> patches_;
    velo_ctx* ctx_;
    bool dirty_;                   // host tiles changed behind the device map's back: rebuild
    bool haveDevice_;
    int res_i0_, res_i1_, res_j0_, res_j1_;  // resident tile index range (inclusive)
    // the rectangle the library last refused to roll to ahead of time (until round 5: every re-anchor), from
    // the resident rectangle it was refused at -- asked again at every frame until it is due, each attempt gathered
    // the entering tiles and walked their points before hearing the same answer (0.2 - 0.8 ms, four times per re-anchor)
    bool refused_ = false;
    int rf_i0_ = 0, rf_i1_ = 0, rf_j0_ = 0, rf_j1_ = 0, rf_from_i0_ = 0, rf_from_i1_ = 0, rf_from_j0_ = 0, rf_from_j1_ = 0;
    bool refusedBefore(int i0, int i1, int j0, int j1) const
    {
        return refused_ && i0 == rf_i0_ && i1 == rf_i1_ && j0 == rf_j0_ && j1 == rf_j1_ && res_i0_ == rf_from_i0_ &&
               res_i1_ == rf_from_i1_ && res_j0_ == rf_from_j0_ && res_j1_ == rf_from_j1_;
    }
    void noteRefused(int i0, int i1, int j0, int j1)
    {
        refused_ = true;
        rf_i0_ = i0, rf_i1_ = i1, rf_j0_ = j0, rf_j1_ = j1;
        rf_from_i0_ = res_i0_, rf_from_i1_ = res_i1_, rf_from_j0_ = res_j0_, rf_from_j1_ = res_j1_;
    }
    bool staged_ = false;                    // a roll is begun (rollBegin), to the rectangle below
    int st_i0_ = 0, st_i1_ = 0, st_j0_ = 0, st_j1_ = 0;
    uint64_t st_n_before_ = 0, st_n_in_ = 0;  // device points before it / entering points (for points_evicted)
    bool publishBegun();                     // the begun roll becomes the resident rectangle
    float residentVoxel_;
    int residentK_;
    MapStats stats_;
    std::vector<float> stage_x_, stage_y_, stage_z_;
    std::vector<float> pend_x_, pend_y_, pend_z_;  // increments taken off the device (in the host tiles), not yet back on it
    std::vector<float> take_x_, take_y_, take_z_;  // (staging of one take)
    bool roiOnly_ = false;          // RegisterOptions::increments_in_roi_only of the last registration (flushIncrements)
    bool publishOwed_ = false;      // updateBesideRegistration began a roll: published right behind the registration's start
    bool forcePlainFlush_ = false;  // the update beside the last registration was refused: the next frame flushes plainly
    bool takeIncrements(bool roi_only = false);  // device list -> appended to pend_*_ and the host tiles; the list is emptied
    // pipeline_increments: takeIncrements + ONE roll begun and published beside the registration just started
    bool updateBesideRegistration(const RegisterOptions& opts);
    bool boxOf(int i0, int i1, int j0, int j1, float lo[3], float hi[3]) const;
    mutable std::string err_;
};

}  // namespace veloslam
