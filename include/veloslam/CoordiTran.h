/* veloslam/CoordiTran.h -- the reference's CoordiTran.h:7-15 declarations, C++ linkage.
 *
 * The reference declares these functions WITHOUT extern "C" (CoordiTran.h:7-15), so every
 * reference translation unit that includes its header (INSSource.cxx:305-326,
 * TransformManager.cxx:179-185, TestINSSender.cxx:52-76) references the C++-mangled names
 * (_Z7llh2xyzPdS_, _Z8eulr2dcmPdPA3_d, ...).  libveloslam_amd.so exports exactly those
 * symbols (host/geodesy_cxx.cpp) next to the C-linkage ones of velo.h: a reference object
 * file links against the library unchanged, and this header may replace the reference's.
 *
 * Do not include this header and <velo.h> in the same C++ translation unit: the same names
 * cannot carry both linkages in one scope.  C++ callers take this one; C callers and FFI
 * bindings take velo.h.
 *
 * HDL2enu (CoordiTran.h:12) is intentionally absent: the reference body reads an
 * uninitialised array (CoordiTran.cpp:232,251) and nothing calls it. */
#ifndef VELOSLAM_COORDITRAN_H
#define VELOSLAM_COORDITRAN_H
#ifndef __cplusplus
#error "veloslam/CoordiTran.h declares C++-linkage functions; C callers use velo.h"
#endif

void eulr2dcm(double eul_vect[3], double DCMbn[3][3]);
void llh2xyz(double llh[3], double xyz[3]);
void xyz2llh(double xyz[3], double llh[3]);
void xyz2enu(double xyz[3], double orgxyz[3], double enu[3]);
void enu2xyz(double enu[3], double orgxyz[3], double xyz[3]);
void enu2llh(double enu[3], double orgxyz[3], double llh[3]);
void llh2enu(double llh[3], double orgxyz[3], double enu[3]);
double MappingAngle(double angle);

#endif /* VELOSLAM_COORDITRAN_H */
