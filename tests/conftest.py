"""Test configuration: registers the `gpu` marker and exposes the product binding as the module `ptgpu`."""
import importlib
import os
import sys

import pytest
import torch  # noqa: F401  before the library: PyTorch ships its own HIP runtime and RCCL under the same sonames, and a
#                            process must end up with one copy of each

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# The package directory has a hyphen in its name, so it is imported by string and aliased.
ptgpu = importlib.import_module("path-tracer_amd")
sys.modules.setdefault("ptgpu", ptgpu)
ptgpu.register_earthmap()  # the image texture of textured_sphere / final_book_two (tests/golden/make_earthmap.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """Default (fast-math) context."""
    ctx = ptgpu.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 20)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx_exact():
    """Exact-arithmetic context (PT_CFG_EXACT): IEEE division, no FMA, f64 sin/cos."""
    ctx = ptgpu.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 20, flags=ptgpu.PT_CFG_EXACT)
    yield ctx
    ctx.close()


@pytest.fixture(params=["fast", "exact"])
def any_ctx(request, gpu_ctx, gpu_ctx_exact):
    return (gpu_ctx, False) if request.param == "fast" else (gpu_ctx_exact, True)
