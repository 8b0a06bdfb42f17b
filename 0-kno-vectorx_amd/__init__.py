"""vectorx_amd: MI355X-native backend for VectorX's header_range proving path.

Directory name `0-kno-vectorx_amd` is not an identifier; import through
`vx_import.load()` (alias `vectorx_amd`).  The compute path is libvxprove.so
(hand-written HIP for gfx950 behind the C ABI of include/vx.h); this package is
the thin host side: ctypes binding, synthetic witness data, and the mirror of
the reference's `Circuit::prove` interface.  There is NO CPU fallback: every
entry point raises if the library or a GPU is missing.
"""
from . import air_library, air_program, lib, shard, synth  # noqa: F401
from .lib import Context, VxError, load_library  # noqa: F401

__all__ = ["air_library", "air_program", "lib", "shard", "synth", "Context", "VxError", "load_library"]
