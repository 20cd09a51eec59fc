"""raytracing_amd -- MI355X-native ray-intersection + shading path behind the reference's
Scene / Camera / Mesh interface.

The compute path is the HIP library raytracing_amd/csrc/librtamd.so (C-ABI: include/rtamd.h).
This package is the thin Python host layer used by the tests, bench.py and the multi-GPU driver:
ctypes bindings (api), scene / camera construction that mirrors the reference's host code
(host), and the tile-sharded render + gather (parallel_render).  There is no CPU fallback: if
the library is missing, or no GPU is present when a device call is made, the call raises.
"""
from . import api  # noqa: F401
from .api import RtError  # noqa: F401
