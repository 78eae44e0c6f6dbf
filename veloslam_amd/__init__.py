"""veloslam_amd -- MI355X-native scan-to-map registration path for VeloSLAM-style
LiDAR frames.  The product is the C-ABI library veloslam_amd/csrc/libveloslam_amd.so
(HIP kernels for gfx950 + dependency-free C++ host code); this package is the thin
Python doorway to it (veloslam_amd.capi) plus seeded synthetic inputs
(veloslam_amd.synth).  Nothing here falls back to a CPU path: using the API
without the built library raises."""
__version__ = "0.1.0"
