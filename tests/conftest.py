import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """The .so files are git-ignored build products: (re)build them when missing or stale
    (hipcc cross-compiles gfx950 without a GPU), so a fresh checkout can run the suite."""
    import shutil
    import pp_amd
    if os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) or shutil.which("hipcc"):
        pp_amd._lib.build()
        pp_amd._lib.build_pybind_module()
        pp_amd._lib.build_variant("strict")
    from oracle import oracle as O
    O.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this container")
    import pp_amd
    # fail loudly if the HIP extension is missing on a GPU box
    pp_amd._lib.lib()
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)
