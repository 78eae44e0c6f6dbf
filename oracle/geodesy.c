/*
 * geodesy.c -- oracle restatement of the reference's CoordiTran.cpp (a1, a2).
 * TEST INFRASTRUCTURE ONLY (see velo_oracle.h).
 *
 * PINNED: tests/test_oracle_golden.py checks every function here bit-for-bit
 * against tests/golden/coorditran.json, which was cut from the reference's own
 * object code (oracle/_ref/libcoorditran_ref.so, built from
 * /root/reference/CoordiTran.cpp by oracle/Makefile).
 *
 * The order of every floating-point operation follows the cited lines, because
 * the check is bit-exact.  HDL2enu (CoordiTran.cpp:220-261) is deliberately not
 * restated: it reads an uninitialised array (:232,:251) so it has no defined
 * result to pin.
 */
#include <math.h>
#include "velo_oracle.h"

static const double WGS_A = 6378137.0000; /* CoordiTran.cpp:58,91 */
static const double WGS_B = 6356752.3142; /* CoordiTran.cpp:59,92 */

static void mat3_mul(const double a[9], const double b[9], double o[9])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j];
            o[3 * i + j] = s;
        }
}

/* CoordiTran.cpp:4-49: angles negated, DCMnb = C3*(C2*C1), result = transpose. */
void vo_eulr2dcm(const double eul[3], double dcm_bn[9])
{
    const double phi = -eul[0], theta = -eul[1], psi = -eul[2];
    const double cpsi = cos(psi), spsi = sin(psi);
    const double cthe = cos(theta), sthe = sin(theta);
    const double cphi = cos(phi), sphi = sin(phi);
    const double C1[9] = {cpsi, spsi, 0, -spsi, cpsi, 0, 0, 0, 1};
    const double C2[9] = {cthe, 0, -sthe, 0, 1, 0, sthe, 0, cthe};
    const double C3[9] = {1, 0, 0, 0, cphi, sphi, 0, -sphi, cphi};
    double c21[9], nb[9];
    mat3_mul(C2, C1, c21);
    mat3_mul(C3, c21, nb);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) dcm_bn[3 * i + j] = nb[3 * j + i];
}

/* CoordiTran.cpp:51-81 */
void vo_llh2xyz(const double llh[3], double xyz[3])
{
    const double phi = llh[0], lambda = llh[1], h = llh[2];
    const double e = sqrt(1 - (WGS_B / WGS_A) * (WGS_B / WGS_A));
    const double sinphi = sin(phi), cosphi = cos(phi);
    const double coslam = cos(lambda), sinlam = sin(lambda);
    const double tan2phi = tan(phi) * tan(phi);
    const double one_me2 = 1 - e * e;
    const double den = sqrt(1 + one_me2 * tan2phi);
    xyz[0] = (WGS_A * coslam) / den + h * coslam * cosphi;
    xyz[1] = (WGS_A * sinlam) / den + h * sinlam * cosphi;
    const double den2 = sqrt(1 - e * e * sinphi * sinphi);
    xyz[2] = (WGS_A * one_me2 * sinphi) / den2 + h * sinphi;
}

/* CoordiTran.cpp:82-150: closed-form ECEF -> geodetic. */
void vo_xyz2llh(const double xyz[3], double llh[3])
{
    const double pi = 3.141592653589793; /* :84 */
    const double x = xyz[0], y = xyz[1], z = xyz[2];
    const double x2 = x * x, y2 = y * y, z2 = z * z;
    const double a = WGS_A, b = WGS_B;
    const double e = sqrt(1 - (b / a) * (b / a));
    const double b2 = b * b;
    const double e2 = e * e;
    const double ep = e * (a / b);
    const double r = sqrt(x2 + y2);
    const double r2 = r * r;
    const double E2 = a * a - b * b;
    const double F = 54 * b2 * z2;
    const double G = r2 + (1 - e2) * z2 - e2 * E2;
    const double c = (e2 * e2 * F * r2) / (G * G * G);
    const double s = pow((double)(1 + c + sqrt(c * c + 2 * c)), (double)(1.0 / 3.0));
    const double P = F / (3 * (s + 1 / s + 1) * (s + 1 / s + 1) * G * G);
    const double Q = sqrt(1 + 2 * e2 * e2 * P);
    const double ro = -(P * e2 * r) / (1 + Q) +
                      sqrt((a * a / 2) * (1 + 1 / Q) - (P * (1 - e2) * z2) / (Q * (1 + Q)) - P * r2 / 2);
    const double tmp = (r - e2 * ro) * (r - e2 * ro);
    const double U = sqrt(tmp + z2);
    const double V = sqrt(tmp + (1 - e2) * z2);
    const double zo = (b2 * z) / (a * V);
    const double height = U * (a * V - b2) / (a * V);
    const double lat = atan((z + ep * ep * zo) / r);
    const double at = atan(y / x);
    double lon;
    if (x >= 0)
        lon = at;
    else if ((x < 0) & (y >= 0)) /* :136, bitwise & as in the reference */
        lon = pi + at;
    else
        lon = at - pi;
    llh[0] = lat;
    llh[1] = lon;
    llh[2] = height;
}

/* CoordiTran.cpp:152-187.  The origin's lat/lon are re-derived on every call. */
void vo_xyz2enu(const double xyz[3], const double orgxyz[3], double enu[3])
{
    double d[3], orgllh[3];
    for (int i = 0; i < 3; ++i) d[i] = xyz[i] - orgxyz[i];
    vo_xyz2llh(orgxyz, orgllh);
    const double sinphi = sin(orgllh[0]), cosphi = cos(orgllh[0]);
    const double sinlam = sin(orgllh[1]), coslam = cos(orgllh[1]);
    const double R[9] = {-sinlam,          coslam,           0,
                         -sinphi * coslam, -sinphi * sinlam, cosphi,
                         cosphi * coslam,  cosphi * sinlam,  sinphi};
    enu[0] = enu[1] = enu[2] = 0;
    for (int i = 0; i < 3; ++i) { /* :180-185: column-by-column accumulation */
        enu[0] = enu[0] + R[0 + i] * d[i];
        enu[1] = enu[1] + R[3 + i] * d[i];
        enu[2] = enu[2] + R[6 + i] * d[i];
    }
}

/* CoordiTran.cpp:189-219 */
void vo_enu2xyz(const double enu[3], const double orgxyz[3], double xyz[3])
{
    double orgllh[3];
    vo_xyz2llh(orgxyz, orgllh);
    const double sinphi = sin(orgllh[0]), cosphi = cos(orgllh[0]);
    const double sinlam = sin(orgllh[1]), coslam = cos(orgllh[1]);
    const double Rt[9] = {-sinlam, -sinphi * coslam, cosphi * coslam,
                          coslam,  -sinphi * sinlam, cosphi * sinlam,
                          0,       cosphi,           sinphi};
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int j = 0; j < 3; ++j) s = s + Rt[3 * i + j] * enu[j];
        xyz[i] = orgxyz[i] + s;
    }
}

/* CoordiTran.cpp:264-269 */
void vo_enu2llh(const double enu[3], const double orgxyz[3], double llh[3])
{
    double xyz[3] = {0, 0, 0};
    vo_enu2xyz(enu, orgxyz, xyz);
    vo_xyz2llh(xyz, llh);
}

/* CoordiTran.cpp:271-276 */
void vo_llh2enu(const double llh[3], const double orgxyz[3], double enu[3])
{
    double xyz[3] = {0, 0, 0};
    vo_llh2xyz(llh, xyz);
    vo_xyz2enu(xyz, orgxyz, enu);
}

/* CoordiTran.cpp:278-293: compass degrees -> math radians. */
double vo_mapping_angle(double angle)
{
    const double pi = 3.141592653589793;
    if (angle >= 0.0 && angle <= 90.0) return (90.0 - angle) * pi / 180.0;
    if (angle > 90.0 && angle <= 270.0) return -(angle - 90.0) * pi / 180.0;
    return (450.0 - angle) * pi / 180.0;
}
