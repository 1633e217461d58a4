"""bundlefusion_amd — MI355X (gfx950) implementation of the BundleFusion hot path.

The product is `lib/libbf_hip.so` (hand-written HIP kernels behind the C ABI declared in
`include/bf_hip.h`) plus the header-only C++ classes in `include/bundlefusion/` that keep the
reference's operator names.  This Python package is only the ctypes view of that C ABI used by
the tests and bench.py; torch is used for device memory and streams, nothing else.

There is no CPU fallback: importing `bundlefusion_amd.capi` fails loudly if the shared library
is missing, and every entry point returns BF_ERR_NO_DEVICE without a GPU.
"""
from . import capi  # noqa: F401
from .capi import lib, BFError  # noqa: F401
