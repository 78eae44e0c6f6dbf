"""The library's boundary is include/velo.h: every environment variable the product's sources read (all of them
measurement aids -- A/B switches, traces, overrides of a zero velo_cfg field) has to be named there (VERDICT r5,
"What's weak" 10: no behaviour keyed on an undocumented getenv)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_getenv_of_the_library_is_named_in_velo_h():
    header = open(os.path.join(ROOT, "include", "velo.h")).read()
    names = set()
    for pat in ("*.cpp", "*.hpp", "*.hip", "host/*.cpp", "kernels/*.hip", "kernels/*.hpp"):
        for f in glob.glob(os.path.join(ROOT, "veloslam_amd", "csrc", pat)):
            names |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(f).read()))
    assert len(names) >= 10                                  # (the scan sees the sources)
    missing = sorted(n for n in names if n not in header)
    assert not missing, "environment variables read by the library but not documented in include/velo.h: %s" % missing
