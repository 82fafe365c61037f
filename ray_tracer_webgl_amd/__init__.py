"""ray_tracer_webgl_amd — MI355X-native path tracer behind the C ABI of include/ptrace.h.

Python here is harness only (ctypes bindings, scene generators, torch.distributed plumbing); the
product is ray_tracer_webgl_amd/libptrace.so: hand-written HIP kernels for gfx950 + a C ABI.
"""
from . import abi  # noqa: F401
from .abi import (  # noqa: F401
    PT_BG_BLACK,
    PT_BG_SKY,
    PT_DIFFUSE,
    PT_EMISSIVE,
    PT_GLASS,
    PT_METAL,
    PtCameraIn,
    PtLookAtIn,
    PtParams,
    PtSphere,
    PtStats,
)

__all__ = ["abi", "load", "PathTracer", "PtError"]


def load():
    from ._lib import load as _load

    return _load()


def __getattr__(name):
    if name in ("PathTracer", "PtError"):
        from . import tracer

        return getattr(tracer, name)
    raise AttributeError(name)
