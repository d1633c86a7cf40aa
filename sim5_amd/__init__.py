"""sim5_amd -- MI355X-native (gfx950) Kerr-spacetime ray tracer with the SIM5 API surface.

The product is the HIP shared library ``sim5_amd/lib/libsim5gpu.so`` behind the C-ABI of
``include/sim5gpu.h``; this package is the thin Python host side over it (ctypes).  There is
no CPU implementation in this package: importing ``sim5_amd.capi`` raises if the library has
not been built, and every compute call raises ``Sim5GpuError`` when no GPU is usable.
"""
from .build import build  # noqa: F401

__all__ = ["build"]
