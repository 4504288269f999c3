"""rufus_amd -- MI355X (gfx950) implementation of the RUFUS k-mer count / set-difference / read-filter hot path.

``capi``  : ctypes binding of the C-ABI (include/rufus_hip.h) -- plumbing only.
``tools`` : host-side mirror of the reference's command-line operators for this path.
The compute lives in rufus_amd/csrc/*.hip; there is no CPU fallback.
"""
from . import capi, tools  # noqa: F401

__all__ = ["capi", "tools"]
