// ref_shim.cpp -- C-linkage doorway into the REFERENCE's own CoordiTran object
// code.  TEST INFRASTRUCTURE ONLY.
//
// oracle/Makefile compiles this file together with /root/reference/CoordiTran.cpp
// *where it lies* (nothing is copied into the repo) into
// oracle/_ref/libcoorditran_ref.so.  tests/golden/make_coorditran_golden.py then
// calls these entry points to cut the known-answer vectors that pin
// oracle/geodesy.c.  /root/reference does not exist on the GPU box: nothing at
// run time there needs this library.
#include "CoordiTran.h"  // resolved with -I/root/reference

extern "C" {
void ref_eulr2dcm(double e[3], double dcm[9])
{
    double m[3][3];
    eulr2dcm(e, m);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) dcm[3 * i + j] = m[i][j];
}
void ref_llh2xyz(double a[3], double o[3]) { llh2xyz(a, o); }
void ref_xyz2llh(double a[3], double o[3]) { xyz2llh(a, o); }
void ref_xyz2enu(double a[3], double org[3], double o[3]) { xyz2enu(a, org, o); }
void ref_enu2xyz(double a[3], double org[3], double o[3]) { enu2xyz(a, org, o); }
void ref_enu2llh(double a[3], double org[3], double o[3]) { enu2llh(a, org, o); }
void ref_llh2enu(double a[3], double org[3], double o[3]) { llh2enu(a, org, o); }
double ref_MappingAngle(double a) { return MappingAngle(a); }
}
