// geodesy_impl.hpp -- the one set of CoordiTran bodies (host/geodesy.cpp), behind both the
// C-linkage exports (include/velo.h) and the C++-linkage exports (host/geodesy_cxx.cpp).
#pragma once
namespace velo_geodesy __attribute__((visibility("hidden"))) {
void eulr2dcm(double eul_vect[3], double DCMbn[3][3]);
void llh2xyz(double llh[3], double xyz[3]);
void xyz2llh(double xyz[3], double llh[3]);
void xyz2enu(double xyz[3], double orgxyz[3], double enu[3]);
void enu2xyz(double enu[3], double orgxyz[3], double xyz[3]);
void enu2llh(double enu[3], double orgxyz[3], double llh[3]);
void llh2enu(double llh[3], double orgxyz[3], double enu[3]);
double MappingAngle(double angle);
}  // namespace velo_geodesy
