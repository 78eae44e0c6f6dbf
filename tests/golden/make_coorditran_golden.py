"""Cut known-answer vectors for CoordiTran from the REFERENCE's
own object code.

Run in the authoring container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_coorditran_golden.py
It loads oracle/_ref/libcoorditran_ref.so (= /root/reference/CoordiTran.cpp
compiled where it lies + oracle/ref_shim.cpp) and writes
tests/golden/coorditran.json.  Values are stored as C99 hex floats so the pin is
bit-exact.  The fixture is data only: inputs and the reference's outputs.
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "libcoorditran_ref.so")


def hx(v):
    return [float(x).hex() for x in np.asarray(v, dtype=np.float64).ravel()]


def main():
    L = C.CDLL(REF)
    dp = C.POINTER(C.c_double)
    L.ref_MappingAngle.restype = C.c_double
    L.ref_MappingAngle.argtypes = [C.c_double]

    def call2(name, a):
        a = np.array(a, dtype=np.float64)
        o = np.zeros(3)
        getattr(L, name)(a.ctypes.data_as(dp), o.ctypes.data_as(dp))
        return o

    def call3(name, a, org):
        a = np.array(a, dtype=np.float64)
        org = np.array(org, dtype=np.float64)
        o = np.zeros(3)
        getattr(L, name)(a.ctypes.data_as(dp), org.ctypes.data_as(dp), o.ctypes.data_as(dp))
        return o

    rng = np.random.default_rng(20161004)
    cases = []
    # origins used by the reference's own senders: TestINSSender.cxx:59 and the
    # INSSource default ORIG_XYZ (INSSource.cxx:334, note its odd z)
    org_llh_deg = np.array([39.8569901, 116.1736406, 89.09288895])
    org_llh = np.array([np.radians(org_llh_deg[0]), np.radians(org_llh_deg[1]), org_llh_deg[2]])
    org_xyz = call2("ref_llh2xyz", org_llh)
    origins = [org_xyz, np.array([-2781621.9891904, 4672106.75052387, 18.8910392])]
    # the SURVEY 8c known answers first
    fixed_llh = [org_llh, np.array([np.radians(39.8579901), np.radians(116.1746406), 90.0])]
    for k in range(300):
        if k < len(fixed_llh):
            llh = fixed_llh[k]
        elif k < 200:
            llh = org_llh + np.array([np.radians(rng.uniform(-0.1, 0.1)),
                                      np.radians(rng.uniform(-0.1, 0.1)),
                                      rng.uniform(-50, 200)])
        else:  # whole globe incl. all longitude quadrants
            llh = np.array([np.radians(rng.uniform(-89, 89)), np.radians(rng.uniform(-179.9, 179.9)),
                            rng.uniform(-100, 9000)])
        xyz = call2("ref_llh2xyz", llh)
        back = call2("ref_xyz2llh", xyz)
        org = origins[0] if k < 2 else origins[k % 2]
        enu = call3("ref_llh2enu", llh, org)
        enu_x = call3("ref_xyz2enu", xyz, org)
        xyz_b = call3("ref_enu2xyz", enu, org)
        llh_b = call3("ref_enu2llh", enu, org)
        cases.append(dict(llh=hx(llh), org=hx(org), llh2xyz=hx(xyz), xyz2llh=hx(back),
                          llh2enu=hx(enu), xyz2enu=hx(enu_x), enu2xyz=hx(xyz_b),
                          enu2llh=hx(llh_b)))
    eul = []
    for k in range(100):
        e = np.array([0.1, -0.2, 0.3]) if k == 0 else rng.uniform(-np.pi, np.pi, 3)
        o = np.zeros(9)
        ee = e.copy()
        L.ref_eulr2dcm(ee.ctypes.data_as(dp), o.ctypes.data_as(dp))
        eul.append(dict(eul=hx(e), dcm=hx(o)))
    ang = []
    for a in [45.0, 180.0, 300.0, 0.0, 90.0, 270.0, 360.0, -10.0] + list(rng.uniform(-30, 400, 40)):
        ang.append(dict(angle=float(a).hex(), out=float(L.ref_MappingAngle(float(a))).hex()))
    out = dict(source="/root/reference/CoordiTran.cpp compiled with g++ -O2 (oracle/Makefile ref)",
               llh_cases=cases, eulr2dcm=eul, mapping_angle=ang)
    with open(os.path.join(HERE, "coorditran.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(cases), "geodesy cases,", len(eul), "dcm cases,", len(ang), "angle cases")


if __name__ == "__main__":
    main()
