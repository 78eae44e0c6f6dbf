// PoseTransform.hpp -- the reference's pose sample (type_defs.h:86-147) without
// Boost/Eigen: ptime -> int64 microseconds, Eigen::Affine3d -> row-major 3x4.
// Same field names, same operator semantics (T,R,V only; result otherwise
// default-constructed), same validity sentinel (seconds_pos == -1).
#pragma once
#include <array>
#include <cstdint>
#include "../velo.h"

namespace veloslam {

using Affine3x4 = std::array<double, 12>;  // [R|t], row-major

struct PoseTransform {
    double T[3];   // ENU metres
    double R[3];   // roll, pitch, yaw -- DEGREES
    double V[3];   // ENU m/s
    int64_t timestamp;  // microseconds; VELO_TIME_INVALID = not_a_date_time
    uint16_t week_number;
    uint32_t milliseconds;
    uint32_t week_number_pos;
    double seconds_pos;

    PoseTransform();  // type_defs.cxx:47-57
    PoseTransform operator+(const PoseTransform& d) const;  // type_defs.h:102-114
    PoseTransform operator-(const PoseTransform& d) const;  // type_defs.h:124-131
    PoseTransform operator*(double ratio) const;            // type_defs.h:115-123
    // linear = Ry(roll) Rx(pitch) Rz(yaw), translation = T (type_defs.h:134-146)
    Affine3x4 getMatrix() const;
    bool valid() const { return seconds_pos != -1; }

    velo_pose toC() const;
    static PoseTransform fromC(const velo_pose& c);
    static PoseTransform fromMatrix(const Affine3x4& M);  // inverse of getMatrix
};

// type_defs.h:160-166
void transformPoint(double pt[3], const Affine3x4& M);

}  // namespace veloslam
