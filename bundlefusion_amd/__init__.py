"""bundlefusion_amd — MI355X (gfx950) implementation of the BundleFusion hot path.

The product is `lib/libbf_hip.so` (hand-written HIP kernels behind the C ABI declared in
`include/bf_hip.h`) plus the header-only C++ classes in `include/bundlefusion/` that keep the
reference's operator names.  This Python package is only the ctypes view of that C ABI used by
the tests and bench.py; torch is used for device memory and streams, nothing else.

There is no CPU fallback: `bundlefusion_amd.capi` (loaded on first use) fails loudly with ImportError if the shared
library is missing, and every entry point returns an error without a GPU.
"""
import importlib


def __getattr__(name):
    # `capi` (and `lib`, `BFError` from it) load libbf_hip.so on first use and raise ImportError if it has not been built;
    # `build`, `synth`, `shard` are plain Python and importable before the library exists (build() needs exactly that).
    if name in ("capi", "sensordata"):
        return importlib.import_module("." + name, __name__)
    if name in ("lib", "BFError"):
        return getattr(importlib.import_module(".capi", __name__), name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
