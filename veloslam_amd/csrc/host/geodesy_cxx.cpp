// geodesy_cxx.cpp -- the CoordiTran functions with C++ LINKAGE (CoordiTran.h:7-15 declares
// them without extern "C": the reference's object code exports and references
// _Z7llh2xyzPdS_ etc., `nm -D oracle/_ref/libcoorditran_ref.so`).  Same bodies as the
// C-linkage exports of geodesy.cpp (namespace velo_geodesy); this translation unit must not
// see velo.h, where the same names carry C linkage.
#include "../../../include/veloslam/CoordiTran.h"
#include "geodesy_impl.hpp"

#define VELO_EXPORT __attribute__((visibility("default")))

VELO_EXPORT void eulr2dcm(double eul_vect[3], double DCMbn[3][3]) { velo_geodesy::eulr2dcm(eul_vect, DCMbn); }
VELO_EXPORT void llh2xyz(double llh[3], double xyz[3]) { velo_geodesy::llh2xyz(llh, xyz); }
VELO_EXPORT void xyz2llh(double xyz[3], double llh[3]) { velo_geodesy::xyz2llh(xyz, llh); }
VELO_EXPORT void xyz2enu(double xyz[3], double orgxyz[3], double enu[3]) { velo_geodesy::xyz2enu(xyz, orgxyz, enu); }
VELO_EXPORT void enu2xyz(double enu[3], double orgxyz[3], double xyz[3]) { velo_geodesy::enu2xyz(enu, orgxyz, xyz); }
VELO_EXPORT void enu2llh(double enu[3], double orgxyz[3], double llh[3]) { velo_geodesy::enu2llh(enu, orgxyz, llh); }
VELO_EXPORT void llh2enu(double llh[3], double orgxyz[3], double enu[3]) { velo_geodesy::llh2enu(llh, orgxyz, enu); }
VELO_EXPORT double MappingAngle(double angle) { return velo_geodesy::MappingAngle(angle); }
