// geodesy.cpp -- product-side WGS-84 LLH <-> ECEF <-> local ENU and Euler -> DCM
// (CoordiTran.h:7-15).  Host-only, fp64: ~15 transcendental calls per 100 Hz INS sample is
// not GPU work.
//
// Exported TWICE, from one set of bodies (namespace velo_geodesy below):
//   * with C linkage (`llh2xyz`, ...; declared in include/velo.h) -- the C ABI that ctypes,
//     cgo-style bindings and the tests call;
//   * with C++ linkage (`_Z7llh2xyzPdS_`, ...; host/geodesy_cxx.cpp, declared in
//     include/veloslam/CoordiTran.h) -- the reference's header declares these functions
//     WITHOUT extern "C", so its translation units (INSSource.cxx:305-326 calcTransform,
//     TransformManager.cxx:179-185 setOriginLLH, TestINSSender.cxx:52-76) reference the
//     mangled names: those are the symbols such a caller links against unchanged
//     (tests/test_host_parity.py::test_coorditran_cxx_linkage_links_reference_style_callers).
//
// The parity bar is bit-exact against vectors cut from the reference's own object
// code (tests/golden/coorditran.json, and live against oracle/_ref), and a closed
// form held to the last bit leaves no freedom: every expression keeps the
// reference's evaluation order, cited per function.  xyz2llh in particular follows
// CoordiTran.cpp:82-150 expression for expression (renamed variables); what is
// this file's own is the structure around the formulas (Ellipsoid, EnuBasis, mul3).
// Built with g++ -O2 -ffp-contract=off, the reference's compiler (csrc/Makefile).
#include <cmath>
#include "geodesy_impl.hpp"

namespace velo_geodesy {
namespace {

struct Ellipsoid {
    double a = 6378137.0000;  // CoordiTran.cpp:58
    double b = 6356752.3142;  // CoordiTran.cpp:59
    double ecc() const { return std::sqrt(1 - (b / a) * (b / a)); }
};

// rows of the ECEF->ENU rotation at a geodetic latitude/longitude
struct EnuBasis {
    double sp, cp, sl, cl;
    explicit EnuBasis(const double org_ecef[3])
    {
        double g[3];
        double tmp[3] = {org_ecef[0], org_ecef[1], org_ecef[2]};
        xyz2llh(tmp, g);  // the reference re-derives this on every call (CoordiTran.cpp:169,192)
        sp = std::sin(g[0]);
        cp = std::cos(g[0]);
        sl = std::sin(g[1]);
        cl = std::cos(g[1]);
    }
    void rows(double R[3][3]) const
    {
        R[0][0] = -sl;      R[0][1] = cl;       R[0][2] = 0;
        R[1][0] = -sp * cl; R[1][1] = -sp * sl; R[1][2] = cp;
        R[2][0] = cp * cl;  R[2][1] = cp * sl;  R[2][2] = sp;
    }
};

void mul3(const double A[3][3], const double B[3][3], double O[3][3])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += A[i][k] * B[k][j];
            O[i][j] = acc;
        }
}

}  // namespace

// CoordiTran.cpp:4-49 -- body->nav DCM = (C3(-phi) C2(-theta) C1(-psi))^T
void eulr2dcm(double eul_vect[3], double DCMbn[3][3])
{
    const double ph = -eul_vect[0], th = -eul_vect[1], ps = -eul_vect[2];
    const double cz = std::cos(ps), sz = std::sin(ps);
    const double cy = std::cos(th), sy = std::sin(th);
    const double cx = std::cos(ph), sx = std::sin(ph);
    const double Rz[3][3] = {{cz, sz, 0}, {-sz, cz, 0}, {0, 0, 1}};
    const double Ry[3][3] = {{cy, 0, -sy}, {0, 1, 0}, {sy, 0, cy}};
    const double Rx[3][3] = {{1, 0, 0}, {0, cx, sx}, {0, -sx, cx}};
    double yz[3][3], nb[3][3];
    mul3(Ry, Rz, yz);
    mul3(Rx, yz, nb);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) DCMbn[i][j] = nb[j][i];
}

// CoordiTran.cpp:51-81
void llh2xyz(double llh[3], double xyz[3])
{
    const Ellipsoid E;
    const double lat = llh[0], lon = llh[1], alt = llh[2];
    const double e = E.ecc();
    const double s_lat = std::sin(lat), c_lat = std::cos(lat);
    const double c_lon = std::cos(lon), s_lon = std::sin(lon);
    const double t2 = std::tan(lat) * std::tan(lat);
    const double k = 1 - e * e;
    const double den = std::sqrt(1 + k * t2);
    const double X = (E.a * c_lon) / den + alt * c_lon * c_lat;
    const double Y = (E.a * s_lon) / den + alt * s_lon * c_lat;
    const double den2 = std::sqrt(1 - e * e * s_lat * s_lat);
    const double Z = (E.a * k * s_lat) / den2 + alt * s_lat;
    xyz[0] = X;
    xyz[1] = Y;
    xyz[2] = Z;
}

// CoordiTran.cpp:82-150 -- closed-form (Zhu/Heikkinen style) ECEF -> geodetic
void xyz2llh(double xyz[3], double llh[3])
{
    const Ellipsoid E;
    const double kPi = 3.141592653589793;
    const double X = xyz[0], Y = xyz[1], Z = xyz[2];
    const double X2 = X * X, Y2 = Y * Y, Z2 = Z * Z;
    const double a = E.a, b = E.b;
    const double e = E.ecc();
    const double b2 = b * b, e2 = e * e;
    const double ep = e * (a / b);
    const double rho = std::sqrt(X2 + Y2);
    const double rho2 = rho * rho;
    const double Esq = a * a - b * b;
    const double F = 54 * b2 * Z2;
    const double G = rho2 + (1 - e2) * Z2 - e2 * Esq;
    const double c = (e2 * e2 * F * rho2) / (G * G * G);
    const double s = std::pow(double(1 + c + std::sqrt(c * c + 2 * c)), double(1.0 / 3.0));
    const double P = F / (3 * (s + 1 / s + 1) * (s + 1 / s + 1) * G * G);
    const double Q = std::sqrt(1 + 2 * e2 * e2 * P);
    const double r0 = -(P * e2 * rho) / (1 + Q) +
                      std::sqrt((a * a / 2) * (1 + 1 / Q) - (P * (1 - e2) * Z2) / (Q * (1 + Q)) -
                                P * rho2 / 2);
    const double w = (rho - e2 * r0) * (rho - e2 * r0);
    const double U = std::sqrt(w + Z2);
    const double V = std::sqrt(w + (1 - e2) * Z2);
    const double z0 = (b2 * Z) / (a * V);
    llh[2] = U * (a * V - b2) / (a * V);
    llh[0] = std::atan((Z + ep * ep * z0) / rho);
    const double base = std::atan(Y / X);
    // quadrant fix-up, CoordiTran.cpp:132-143
    llh[1] = (X >= 0) ? base : (((X < 0) & (Y >= 0)) ? kPi + base : base - kPi);
}

// CoordiTran.cpp:152-187
void xyz2enu(double xyz[3], double orgxyz[3], double enu[3])
{
    const double d[3] = {xyz[0] - orgxyz[0], xyz[1] - orgxyz[1], xyz[2] - orgxyz[2]};
    double R[3][3];
    EnuBasis(orgxyz).rows(R);
    double e = 0, n = 0, u = 0;
    for (int i = 0; i < 3; ++i) {
        e = e + R[0][i] * d[i];
        n = n + R[1][i] * d[i];
        u = u + R[2][i] * d[i];
    }
    enu[0] = e;
    enu[1] = n;
    enu[2] = u;
}

// CoordiTran.cpp:189-219
void enu2xyz(double enu[3], double orgxyz[3], double xyz[3])
{
    double R[3][3];
    EnuBasis(orgxyz).rows(R);
    for (int i = 0; i < 3; ++i) {
        double acc = 0;
        for (int j = 0; j < 3; ++j) acc = acc + R[j][i] * enu[j];  // transpose of rows()
        xyz[i] = orgxyz[i] + acc;
    }
}

// CoordiTran.cpp:264-269
void enu2llh(double enu[3], double orgxyz[3], double llh[3])
{
    double ecef[3] = {0, 0, 0};
    enu2xyz(enu, orgxyz, ecef);
    xyz2llh(ecef, llh);
}

// CoordiTran.cpp:271-276
void llh2enu(double llh[3], double orgxyz[3], double enu[3])
{
    double ecef[3] = {0, 0, 0};
    llh2xyz(llh, ecef);
    xyz2enu(ecef, orgxyz, enu);
}

// CoordiTran.cpp:278-293 -- compass bearing (deg, clockwise from north) -> math angle (rad)
double MappingAngle(double angle)
{
    const double kPi = 3.141592653589793;
    if (angle >= 0.0 && angle <= 90.0) return (90.0 - angle) * kPi / 180.0;
    if (angle > 90.0 && angle <= 270.0) return -(angle - 90.0) * kPi / 180.0;
    return (450.0 - angle) * kPi / 180.0;
}

}  // namespace velo_geodesy

// ---- C linkage (include/velo.h) ----------------------------------------------------------
#include "../../../include/velo.h"
extern "C" {
void eulr2dcm(double eul_vect[3], double DCMbn[3][3]) { velo_geodesy::eulr2dcm(eul_vect, DCMbn); }
void llh2xyz(double llh[3], double xyz[3]) { velo_geodesy::llh2xyz(llh, xyz); }
void xyz2llh(double xyz[3], double llh[3]) { velo_geodesy::xyz2llh(xyz, llh); }
void xyz2enu(double xyz[3], double orgxyz[3], double enu[3]) { velo_geodesy::xyz2enu(xyz, orgxyz, enu); }
void enu2xyz(double enu[3], double orgxyz[3], double xyz[3]) { velo_geodesy::enu2xyz(enu, orgxyz, xyz); }
void enu2llh(double enu[3], double orgxyz[3], double llh[3]) { velo_geodesy::enu2llh(enu, orgxyz, llh); }
void llh2enu(double llh[3], double orgxyz[3], double enu[3]) { velo_geodesy::llh2enu(llh, orgxyz, enu); }
double MappingAngle(double angle) { return velo_geodesy::MappingAngle(angle); }
}  // extern "C"
