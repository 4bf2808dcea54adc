import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle (the checker) works on small tensors: on a 256-core host torch's default thread count turns every
    # tiny op into a fork-join over hundreds of threads and the suite spends its time there.
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden_interp():
    return torch.load(os.path.join(GOLDEN, "interpolation.pt"))


@pytest.fixture(scope="session")
def golden_cde():
    return torch.load(os.path.join(GOLDEN, "cdeint.pt"))


@pytest.fixture(scope="session")
def native():
    """The built extension; GPU tests must fail loudly (not skip) if it is missing."""
    import torchcde_amd
    torchcde_amd.load()
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    return torchcde_amd


@pytest.fixture(autouse=True)
def _default_tuning():
    """Tests switch kernel forms through the library's tuning table (torchcde_amd.set_option); every test starts and
    ends with the production defaults."""
    yield
    import torchcde_amd._lib as lib
    if lib._lib is not None:
        lib._lib.cde_reset_options()
