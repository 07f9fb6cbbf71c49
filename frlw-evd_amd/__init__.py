"""frlw-evd_amd: MI355X-native event-stream encoders and YOLOX detector path.

Only the hot path of HarmoniaLeo/FRLW-EvD lives here:
``csrc/`` holds the hand-written gfx950 HIP kernels behind the C-ABI declared in
``include/frlw_evd.h``; the Python modules mirror the reference's function and
class signatures for that path and call the C-ABI through ctypes.
"""
__version__ = "0.1.0"
