import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library's default arithmetic contract of the voxel update is `fast` (the reference GPU build's own contract, and what bench.py measures).
# The bit-for-bit tests hold the product to the oracle, which is a host build's arithmetic: they run under `exact`, selected here for every scene
# a test creates; the tests of the fast contract (tests/test_tsdf_fast_gpu.py and the fast legs of the pipeline tests) switch explicitly.
os.environ.setdefault("BF_TSDF_ARITH", "exact")
# Likewise the schedule: the pipeline's default runs the chunk solves on their own thread, applied ten frames later (the reference's optimiser thread, made
# deterministic); the oracle loop and the compiled reference loop are the serial order, so the suite selects it.  The lagged schedule is tested against the oracle
# loop under the same lag (tests/test_pipeline_gpu.py::test_lagged_solve_mode_vs_oracle_loop_with_the_same_lag) and measured by bench.py.
os.environ.setdefault("BF_PIPELINE_SOLVE_LAG", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the product library and the oracle exist (cross-compiles without a GPU)."""
    from bundlefusion_amd import build
    build.build_lib()
    build.build_oracle()
    return build


@pytest.fixture(scope="session")
def oracle(built):
    from tests import oracle_api
    return oracle_api


@pytest.fixture(scope="session")
def gpu(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test is marked gpu but no GPU is visible (there is no CPU fallback)")
    import bundlefusion_amd
    return bundlefusion_amd
