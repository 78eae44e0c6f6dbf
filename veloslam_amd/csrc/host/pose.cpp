// pose.cpp -- product host code for SURVEY rows a3..a6: PoseTransform algebra,
// Euler(deg) -> affine, the time-indexed pose store and its linear interpolation,
// and the per-packet transform table that feeds K1.  fp64, host-only: per frame
// this is ~300 interpolations + ~300 matrix builds, i.e. microseconds of CPU
// (the reference measured 3-4 us per interpolation, TransformManager.cxx:143-146).
//
// Design: instead of the reference's vector<vector<shared_ptr>> buckets plus a
// 5-slot ring (TimeLine.h:118-128) the samples live in ONE sorted flat array; a
// cached cursor makes the sequential access pattern of packet streams O(1).  To
// return the same bracket as TimeLine::getBoundaryData at an exact knot, the
// bucket width ("interval", TimeLine.h:166-177, 536-552) is tracked the same way
// and each sample remembers its bucket number.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include "../../../include/veloslam/TransformManager.hpp"

namespace veloslam {

// ------------------------------------------------------------- PoseTransform
PoseTransform::PoseTransform()
    : T{0, 0, 0}, R{0, 0, 0}, V{0, 0, 0}, timestamp(VELO_TIME_INVALID), week_number(0),
      milliseconds(0), week_number_pos(0), seconds_pos(-1)
{
}

PoseTransform PoseTransform::operator+(const PoseTransform& d) const
{
    PoseTransform r;
    for (int i = 0; i < 3; ++i) {
        r.T[i] = T[i] + d.T[i];
        r.R[i] = R[i] + d.R[i];
        r.V[i] = V[i] + d.V[i];
    }
    return r;
}
PoseTransform PoseTransform::operator-(const PoseTransform& d) const
{
    PoseTransform r;
    for (int i = 0; i < 3; ++i) {
        r.T[i] = T[i] - d.T[i];
        r.R[i] = R[i] - d.R[i];
        r.V[i] = V[i] - d.V[i];
    }
    return r;
}
PoseTransform PoseTransform::operator*(double ratio) const
{
    PoseTransform r;
    for (int i = 0; i < 3; ++i) {
        r.T[i] = T[i] * ratio;
        r.R[i] = R[i] * ratio;
        r.V[i] = V[i] * ratio;
    }
    return r;
}

namespace {
// Rotation by `angle` about a unit axis, with the operation order of Eigen's
// AngleAxis::toRotationMatrix (the reference's getMatrix goes through it): the
// diagonal is (1-c)*a_i*a_i + c, which for the rotation axis itself is
// (1-c)+c and not necessarily exactly 1.
struct Rot3 {
    double m[9];
    Rot3(double angle, int axis)
    {
        double a[3] = {0, 0, 0};
        a[axis] = 1.0;
        double s, c;
        ::sincos(angle, &s, &c);  // glibc: one entry point, so gcc- and clang-built callers agree bitwise
        const double sa[3] = {s * a[0], s * a[1], s * a[2]};
        const double ca[3] = {(1.0 - c) * a[0], (1.0 - c) * a[1], (1.0 - c) * a[2]};
        double t = ca[0] * a[1];
        m[1] = t - sa[2];
        m[3] = t + sa[2];
        t = ca[0] * a[2];
        m[2] = t + sa[1];
        m[6] = t - sa[1];
        t = ca[1] * a[2];
        m[5] = t - sa[0];
        m[7] = t + sa[0];
        m[0] = ca[0] * a[0] + c;
        m[4] = ca[1] * a[1] + c;
        m[8] = ca[2] * a[2] + c;
    }
};
void rmul(double L[9], const Rot3& R)
{
    double o[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            o[3 * i + j] = (L[3 * i] * R.m[j] + L[3 * i + 1] * R.m[3 + j]) + L[3 * i + 2] * R.m[6 + j];
    std::memcpy(L, o, sizeof o);
}
inline double to_radius(double deg) { return deg * M_PI / 180; }  // type_defs.h:25
}  // namespace

Affine3x4 PoseTransform::getMatrix() const
{
    double L[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    rmul(L, Rot3(to_radius(R[0]), 1));  // roll  about Y
    rmul(L, Rot3(to_radius(R[1]), 0));  // pitch about X
    rmul(L, Rot3(to_radius(R[2]), 2));  // yaw   about Z
    Affine3x4 M;
    for (int i = 0; i < 3; ++i) {
        M[4 * i] = L[3 * i];
        M[4 * i + 1] = L[3 * i + 1];
        M[4 * i + 2] = L[3 * i + 2];
        M[4 * i + 3] = T[i];
    }
    return M;
}

PoseTransform PoseTransform::fromMatrix(const Affine3x4& M)
{
    PoseTransform p;
    p.T[0] = M[3];
    p.T[1] = M[7];
    p.T[2] = M[11];
    double sb = -M[6];
    sb = sb > 1 ? 1 : (sb < -1 ? -1 : sb);
    p.R[0] = std::atan2(M[2], M[10]) * 180 / M_PI;
    p.R[1] = std::asin(sb) * 180 / M_PI;
    p.R[2] = std::atan2(M[4], M[5]) * 180 / M_PI;
    p.seconds_pos = 0;
    return p;
}

velo_pose PoseTransform::toC() const
{
    velo_pose c;
    std::memset(&c, 0, sizeof c);
    for (int i = 0; i < 3; ++i) {
        c.T[i] = T[i];
        c.R[i] = R[i];
        c.V[i] = V[i];
    }
    c.t_us = timestamp;
    c.week_number = week_number;
    c.milliseconds = milliseconds;
    c.week_number_pos = week_number_pos;
    c.seconds_pos = seconds_pos;
    return c;
}
PoseTransform PoseTransform::fromC(const velo_pose& c)
{
    PoseTransform p;
    for (int i = 0; i < 3; ++i) {
        p.T[i] = c.T[i];
        p.R[i] = c.R[i];
        p.V[i] = c.V[i];
    }
    p.timestamp = c.t_us;
    p.week_number = c.week_number;
    p.milliseconds = c.milliseconds;
    p.week_number_pos = c.week_number_pos;
    p.seconds_pos = c.seconds_pos;
    return p;
}

void transformPoint(double pt[3], const Affine3x4& M)
{
    const double x = pt[0], y = pt[1], z = pt[2];
    pt[0] = M[0] * x + M[1] * y + M[2] * z + M[3];
    pt[1] = M[4] * x + M[5] * y + M[6] * z + M[7];
    pt[2] = M[8] * x + M[9] * y + M[10] * z + M[11];
}

// ----------------------------------------------------------- sorted pose store
struct TransformManager::Impl {
    std::vector<PoseTransform> v;  // ascending timestamp
    std::vector<int> bucket;       // TimeLine bucket number of each sample
    double interval = 0;
    bool finalized = false;
    size_t cursor = 0;  // index of the last bracket's lower end

    int bucket_of(int64_t t) const { return (int)((double)(t - v.front().timestamp) / interval); }

    void add(const PoseTransform& p)
    {
        const int64_t t = p.timestamp;
        if (v.empty()) {
            v.push_back(p);
            bucket.push_back(0);
            return;
        }
        if (v.size() == 1) {
            if (t == v[0].timestamp) {
                v[0] = p;
                return;
            }
            interval = (double)(t > v[0].timestamp ? t - v[0].timestamp : v[0].timestamp - t) * 0.95;
            if (t < v[0].timestamp) {
                v.insert(v.begin(), p);
                bucket = {0, 1};
            } else {
                v.push_back(p);
                bucket.push_back(1);
            }
            return;
        }
        if (!finalized && v.size() == 10) {  // TimeLine.h:166, 536-552
            interval = (double)((uint64_t)(v.back().timestamp - v.front().timestamp) / (uint64_t)v.size());
            for (size_t i = 0; i < v.size(); ++i) bucket[i] = bucket_of(v[i].timestamp);
            finalized = true;
        }
        auto it = std::lower_bound(v.begin(), v.end(), t,
                                   [](const PoseTransform& a, int64_t tt) { return a.timestamp < tt; });
        const size_t pos = (size_t)(it - v.begin());
        if (it != v.end() && it->timestamp == t) {  // same instant: overwrite (TimeLine.h:197-200)
            *it = p;
            return;
        }
        if (pos == 0) {  // older than everything: the origin of the bucket grid moves
            v.insert(v.begin(), p);
            bucket.insert(bucket.begin(), 0);
            for (size_t i = 0; i < v.size(); ++i) bucket[i] = bucket_of(v[i].timestamp);
        } else {
            v.insert(it, p);
            bucket.insert(bucket.begin() + (long)pos, bucket_of(t));
        }
        cursor = 0;
    }

    // -> number of valid ends; indices of fore/back in v
    int boundary(int64_t t, size_t& fi, size_t& bi)
    {
        const size_t n = v.size();
        if (n == 0) return 0;
        if (n == 1) {
            fi = 0;
            return 1;
        }
        if (t <= v.front().timestamp) {  // TimeLine.h:394-402: first two (extrapolates)
            fi = 0;
            bi = 1;
            return 2;
        }
        if (t >= v.back().timestamp) {  // TimeLine.h:403-407: last two (extrapolates)
            fi = n - 2;
            bi = n - 1;
            return 2;
        }
        // first index with timestamp >= t, starting from the cached cursor
        size_t lo;
        if (cursor + 1 < n && v[cursor].timestamp < t && t <= v[cursor + 1].timestamp)
            lo = cursor + 1;
        else if (cursor + 2 < n && v[cursor + 1].timestamp < t && t <= v[cursor + 2].timestamp)
            lo = cursor + 2;
        else
            lo = (size_t)(std::lower_bound(v.begin(), v.end(), t,
                                           [](const PoseTransform& a, int64_t tt) {
                                               return a.timestamp < tt;
                                           }) -
                          v.begin());
        fi = lo - 1;
        bi = lo;
        if (v[lo].timestamp == t) {
            // exact knot.  Inside the last five samples the reference's ring search
            // (TimeLine.h:412-416) yields (previous, knot).  Further back it depends on
            // whether the knot opens its bucket (TimeLine.h:419-444).
            const size_t ring0 = n >= 5 ? n - 5 : 0;
            const bool in_ring = t > v[ring0].timestamp;
            if (!in_ring && bucket[lo] != bucket[lo - 1] && lo + 1 < n) {
                const int64_t gap_next = v[lo + 1].timestamp - t, gap_prev = t - v[lo - 1].timestamp;
                if (!(gap_next < gap_prev)) {  // keep (knot, next) unless next is closer than prev
                    fi = lo;
                    bi = lo + 1;
                }
            }
        }
        cursor = fi;
        return 2;
    }

    bool interpolate(int64_t t, PoseTransform* out)
    {
        out->timestamp = t;  // TransformManager.cxx:151
        size_t fi = 0, bi = 0;
        const int nb = boundary(t, fi, bi);
        if (nb == 0) return false;
        if (nb == 1) {
            const PoseTransform& fore = v[fi];
            // TransformManager.cxx:161: integer / 1e6f is a float division
            const double sec = (double)((float)(t - fore.timestamp) / 1e6f);
            for (int i = 0; i < 3; ++i) {
                out->V[i] = fore.V[i];
                out->R[i] = fore.R[i];
                out->T[i] = fore.T[i] + fore.V[i] * sec;
            }
            return true;
        }
        const PoseTransform& fore = v[fi];
        const PoseTransform& back = v[bi];
        const double ratio = double(t - fore.timestamp) / double(back.timestamp - fore.timestamp);
        *out = fore + ((back - fore) * ratio);  // timestamp becomes invalid, as in the reference
        out->seconds_pos = 0;                   // TransformManager.cxx:174
        return true;
    }

    bool packet_transforms(const int64_t* pkt_t, size_t n_pkt, double* T3x4, uint8_t* valid,
                           PoseTransform* carpose_out)
    {
        PoseTransform car;
        for (size_t i = 0; i < n_pkt; ++i) {
            PoseTransform tr;
            const bool got = interpolate(pkt_t[i], &tr);
            if (i == 0) {
                car = tr;  // HDLParser.cxx:993-1001 (memcpy before timestamp is restored)
                if (carpose_out) *carpose_out = car;
            }
            tr.timestamp = pkt_t[i];
            double* M = T3x4 + 12 * i;
            if (got && tr.seconds_pos != -1) {  // HDLParser.cxx:1004
                for (int a = 0; a < 3; ++a) tr.T[a] -= car.T[a];  // :1057-1062
                const Affine3x4 A = tr.getMatrix();
                std::memcpy(M, A.data(), sizeof(double) * 12);
                if (valid) valid[i] = 1;
            } else {
                static const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                std::memcpy(M, I, sizeof I);
                if (valid) valid[i] = 0;
            }
        }
        return n_pkt > 0;
    }
};

TransformManager::TransformManager() : impl_(new Impl), originLLH_{0, 0, 0}, originXYZ_{0, 0, 0} {}
TransformManager::~TransformManager() = default;

int TransformManager::getNumberOfTransforms()
{
    std::lock_guard<std::mutex> lock(mutex_);
    return (int)impl_->v.size();
}
void TransformManager::clearTransforms()
{
    std::lock_guard<std::mutex> lock(mutex_);
    impl_.reset(new Impl);
}
void TransformManager::addTransform(std::shared_ptr<PoseTransform> trans)
{
    if (trans) addTransform(*trans);
}
void TransformManager::addTransform(const PoseTransform& trans)
{
    std::lock_guard<std::mutex> lock(mutex_);
    impl_->add(trans);
}
bool TransformManager::interpolateTransform(int64_t t_us, PoseTransform* xform)
{
    std::lock_guard<std::mutex> lock(mutex_);
    return impl_->interpolate(t_us, xform);
}
bool TransformManager::packetTransforms(const int64_t* pkt_t_us, size_t n_pkt, double* T3x4,
                                        uint8_t* valid, PoseTransform* carpose)
{
    std::lock_guard<std::mutex> lock(mutex_);
    return impl_->packet_transforms(pkt_t_us, n_pkt, T3x4, valid, carpose);
}
std::vector<velo_pose> TransformManager::snapshot()
{
    std::lock_guard<std::mutex> lock(mutex_);
    std::vector<velo_pose> out;
    out.reserve(impl_->v.size());
    for (const auto& p : impl_->v) out.push_back(p.toC());
    return out;
}

bool TransformManager::loadFromTxtFile(const std::string& filename, bool clearOldData)
{
    std::ifstream ifs(filename);
    if (!ifs) return false;
    if (clearOldData) clearTransforms();
    PoseTransform tr;
    double v;
    long long sec, usec;
    // TransformManager.cxx:108-121: x y yaw roll pitch v sec usec ; angles rad -> deg,
    // yaw sign flipped.  The reference's timevalToPtime adds a fixed +8 h
    // (type_defs.cxx:69-72); timestamps here stay plain epoch microseconds + 8 h.
    while (ifs >> tr.T[0] >> tr.T[1] >> tr.R[2] >> tr.R[0] >> tr.R[1] >> v >> sec >> usec) {
        tr.R[0] = tr.R[0] * 180 / M_PI;
        tr.R[1] = tr.R[1] * 180 / M_PI;
        tr.R[2] = -(tr.R[2] * 180 / M_PI);
        tr.timestamp = (int64_t)sec * 1000000 + usec + 8LL * 3600 * 1000000;
        // TransformManager.cxx:116-119: the GPS-week fields are filled from the (+8 h) time, and
        // seconds_pos is the millisecond count divided IN FLOAT (`milliseconds / 1000.0f`:
        // the uint32 is first rounded to 24 bits), which also marks the sample valid (!= -1)
        velo_time_to_week_milli(tr.timestamp, &tr.week_number, &tr.milliseconds);
        tr.week_number_pos = tr.week_number;
        tr.seconds_pos = (double)((float)tr.milliseconds / 1000.0f);
        addTransform(tr);
        tr = PoseTransform();
    }
    return true;
}

bool TransformManager::loadFromMetaFile(const std::string& filename, bool clearOldData)
{
    size_t n = 0;
    if (velo_insmeta_read(filename.c_str(), nullptr, 0, &n) != VELO_OK) return false;  // "does not exist"
    std::vector<velo_pose> v(n);
    if (n && velo_insmeta_read(filename.c_str(), v.data(), n, &n) != VELO_OK) return false;
    if (clearOldData) clearTransforms();
    for (const velo_pose& p : v) addTransform(PoseTransform::fromC(p));
    return true;
}

bool TransformManager::writeToMetaFile(const std::string& filename)
{
    const std::vector<velo_pose> v = snapshot();
    return velo_insmeta_write(filename.c_str(), v.data(), v.size()) == VELO_OK;
}

void TransformManager::setOriginLLH(const double LLH[3])
{
    originLLH_[0] = to_radius(LLH[0]);
    originLLH_[1] = to_radius(LLH[1]);
    originLLH_[2] = LLH[2];
    llh2xyz(originLLH_, originXYZ_);
}

}  // namespace veloslam

// ------------------------------------------------------------------- C ABI
using veloslam::PoseTransform;

extern "C" {

int velo_matrix_from_pose(const double TRdeg[6], double T[12])
{
    if (!TRdeg || !T) return VELO_E_INVALID;
    PoseTransform p;
    for (int i = 0; i < 3; ++i) {
        p.T[i] = TRdeg[i];
        p.R[i] = TRdeg[3 + i];
    }
    const veloslam::Affine3x4 M = p.getMatrix();
    std::memcpy(T, M.data(), sizeof(double) * 12);
    return VELO_OK;
}

int velo_pose_from_matrix(const double T[12], double TRdeg[6])
{
    if (!TRdeg || !T) return VELO_E_INVALID;
    veloslam::Affine3x4 M;
    std::memcpy(M.data(), T, sizeof(double) * 12);
    const PoseTransform p = PoseTransform::fromMatrix(M);
    for (int i = 0; i < 3; ++i) {
        TRdeg[i] = p.T[i];
        TRdeg[3 + i] = p.R[i];
    }
    return VELO_OK;
}

}  // extern "C" (reopened below)

namespace veloslam {

// A caller-held array sorted by time, read in place: what the C entry points get.  Same bracket
// rules as TransformManager (TimeLine::getBoundaryData, TimeLine.h:384-468) in O(log n) per query
// and without copying the store -- the first version of these entry points rebuilt a
// TransformManager from the array on EVERY call, 100 MB of copies per frame on a 1e6-pose drive
// (found by tools/interp_bench).  The reference's bucket grid is a function of the first samples
// only when the samples arrive in time order (interval fixed when the 11th sample is added,
// TimeLine.h:166,536-552), which a sorted array is; equal time stamps (a later sample overwrites an
// earlier one, TimeLine.h:197-200) shift indices, so an array that shows one where it matters --
// the first 11 samples, the last 6, the neighbours of the bracket -- takes the slow path through a
// real TransformManager instead.
SortedPoseView::SortedPoseView(const velo_pose* sorted, size_t n) : p_(sorted), n_(n)
{
    strict_ = true;
    for (size_t i = 1; i < n_ && i < 11; ++i) strict_ = strict_ && p_[i - 1].t_us < p_[i].t_us;
    for (size_t i = n_ > 6 ? n_ - 6 : 1; i < n_; ++i) strict_ = strict_ && p_[i - 1].t_us < p_[i].t_us;
    if (n_ >= 11)
        interval_ = (double)((uint64_t)(p_[9].t_us - p_[0].t_us) / (uint64_t)10);
    else if (n_ >= 2)
        interval_ = (double)(p_[1].t_us - p_[0].t_us) * 0.95;
    if (n_ >= 2 && !(interval_ > 0)) strict_ = false;
}

bool SortedPoseView::interpolate(int64_t t, PoseTransform* out) const
{
    const size_t n = n_;
    if (n == 0) {
        out->timestamp = t;
        return false;
    }
    size_t fi = 0, bi = 0;
    bool two = n >= 2, slow = !strict_;
    if (two && !slow) {
        if (t <= p_[0].t_us) {
            fi = 0, bi = 1;
        } else if (t >= p_[n - 1].t_us) {
            fi = n - 2, bi = n - 1;
        } else {
            size_t lo = 0, hi = n;  // first index with t_us >= t
            while (lo < hi) {
                const size_t mid = lo + (hi - lo) / 2;
                if (p_[mid].t_us < t) lo = mid + 1; else hi = mid;
            }
            // equal stamps next to the bracket would have been merged by the store
            for (size_t i = lo >= 2 ? lo - 2 : 0; i + 1 < n && i <= lo + 1; ++i)
                slow = slow || !(p_[i].t_us < p_[i + 1].t_us);
            fi = lo - 1, bi = lo;
            if (!slow && p_[lo].t_us == t) {  // exact knot: see TransformManager::Impl::boundary
                const size_t ring0 = n >= 5 ? n - 5 : 0;
                const bool in_ring = t > p_[ring0].t_us;
                if (!in_ring && bucket(lo) != bucket(lo - 1) && lo + 1 < n) {
                    const int64_t gap_next = p_[lo + 1].t_us - t, gap_prev = t - p_[lo - 1].t_us;
                    if (!(gap_next < gap_prev)) fi = lo, bi = lo + 1;
                }
            }
        }
    }
    if (slow) {
        TransformManager tm;
        for (size_t i = 0; i < n; ++i) tm.addTransform(PoseTransform::fromC(p_[i]));
        return tm.interpolateTransform(t, out);
    }
    out->timestamp = t;  // TransformManager.cxx:151
    const PoseTransform fore = PoseTransform::fromC(p_[fi]);
    if (!two) {
        const double sec = (double)((float)(t - fore.timestamp) / 1e6f);  // TransformManager.cxx:161
        for (int i = 0; i < 3; ++i) {
            out->V[i] = fore.V[i];
            out->R[i] = fore.R[i];
            out->T[i] = fore.T[i] + fore.V[i] * sec;
        }
        return true;
    }
    const PoseTransform back = PoseTransform::fromC(p_[bi]);
    const double ratio = double(t - fore.timestamp) / double(back.timestamp - fore.timestamp);
    *out = fore + ((back - fore) * ratio);
    out->seconds_pos = 0;  // TransformManager.cxx:174
    return true;
}

bool SortedPoseView::packetTransforms(const int64_t* pkt_t, size_t n_pkt, double* T3x4, uint8_t* valid,
                                      PoseTransform* carpose_out) const
{
    PoseTransform car;
    for (size_t i = 0; i < n_pkt; ++i) {
        PoseTransform tr;
        const bool got = interpolate(pkt_t[i], &tr);
        if (i == 0) {
            car = tr;  // HDLParser.cxx:993-1001
            if (carpose_out) *carpose_out = car;
        }
        tr.timestamp = pkt_t[i];
        double* M = T3x4 + 12 * i;
        if (got && tr.seconds_pos != -1) {  // HDLParser.cxx:1004
            for (int a = 0; a < 3; ++a) tr.T[a] -= car.T[a];  // :1057-1062
            const Affine3x4 A = tr.getMatrix();
            std::memcpy(M, A.data(), sizeof(double) * 12);
            if (valid) valid[i] = 1;
        } else {
            static const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
            std::memcpy(M, I, sizeof I);
            if (valid) valid[i] = 0;
        }
    }
    return n_pkt > 0;
}

}  // namespace veloslam

extern "C" {

int velo_interp_pose(const velo_pose* sorted, size_t n, int64_t t_us, velo_pose* out)
{
    if (!out || (n && !sorted)) return VELO_E_INVALID;
    const veloslam::SortedPoseView view(sorted, n);
    PoseTransform p;  // fresh: seconds_pos = -1 unless a two-sample bracket sets it to 0
    if (!view.interpolate(t_us, &p)) return VELO_E_NODATA;
    *out = p.toC();
    return VELO_OK;
}

// ptimeToWeekMilli (type_defs.cxx:74-79): `week` = boost::gregorian::date::week_number() of the
// date (the ISO 8601 week, 1..53), `milli` = milliseconds since the last Sunday 00:00
// (`date - days(tm_wday)`), truncated into 32 bits.  Written here from the ISO rule itself -- the
// week of a date is the ordinal week of the Thursday of its Monday-based week -- the oracle
// (oracle/pose.c vo_time_to_week_milli) restates Boost.DateTime's julian-day formulation; the two
// and Python's date.isocalendar() agree on every day of 1970-2199 (tests/test_host_parity.py).
void velo_time_to_week_milli(int64_t t_us, uint16_t* week, uint32_t* milli)
{
    const int64_t kDayUs = 86400LL * 1000000;
    int64_t days = t_us / kDayUs;
    if (t_us % kDayUs < 0) --days;                 // floor: the date of a time before the epoch
    const int64_t wd_sun = ((days % 7) + 7 + 4) % 7;   // 1970-01-01 was a Thursday; 0 = Sunday
    if (milli) *milli = (uint32_t)((t_us - (days - wd_sun) * kDayUs) / 1000);
    if (!week) return;
    const int64_t wd_mon = (wd_sun + 6) % 7;           // 0 = Monday
    const int64_t thursday = days - wd_mon + 3;        // the Thursday that names the ISO week
    // civil year of `thursday` and its ordinal day (days-from-civil, proleptic Gregorian)
    int64_t z = thursday + 719468;
    const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    const int64_t doe = z - era * 146097;
    const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const int64_t doy_mar = doe - (365 * yoe + yoe / 4 - yoe / 100);  // day of the March-based year
    const int64_t month_mar = (5 * doy_mar + 2) / 153;
    int64_t year = yoe + era * 400 + (month_mar >= 10 ? 1 : 0);
    const bool leap = (year % 4 == 0 && year % 100 != 0) || year % 400 == 0;
    // ordinal day in the January-based year
    const int64_t doy = month_mar >= 10 ? doy_mar - 306 : doy_mar + 59 + (leap ? 1 : 0);
    *week = (uint16_t)(doy / 7 + 1);
}

int velo_carposes_read(const char* path, velo_pose* poses, size_t cap, size_t* n_out)
{
    if (!path || !n_out) return VELO_E_INVALID;
    veloslam::TransformManager tm;
    if (!tm.loadFromTxtFile(path, true)) return VELO_E_NODATA;
    const std::vector<velo_pose> v = tm.snapshot();
    *n_out = v.size();
    if (!poses) return VELO_OK;
    if (v.size() > cap) return VELO_E_RANGE;
    if (!v.empty()) std::memcpy(poses, v.data(), v.size() * sizeof(velo_pose));  // (memcpy(.., nullptr, 0) is UB)
    return VELO_OK;
}

int velo_packet_transforms(const velo_pose* sorted, size_t n, const int64_t* pkt_t_us, size_t n_pkt,
                           double* T3x4, uint8_t* valid, velo_pose* carpose)
{
    if (!pkt_t_us || !T3x4 || (n && !sorted)) return VELO_E_INVALID;
    const veloslam::SortedPoseView view(sorted, n);
    PoseTransform car;
    view.packetTransforms(pkt_t_us, n_pkt, T3x4, valid, &car);
    if (carpose) *carpose = car.toC();
    return VELO_OK;
}

}  // extern "C"
