import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def knob(monkeypatch):
    """``knob("DCARL_TRACE_KERNEL", "duo")``: override a launcher's own choice (run a kernel instance at a size where the library would
    pick another one, or one of the measurement-only instances).  Those overrides exist in the A/B variant of the library only
    (-DDCARL_AB_BUILD, libdcarl_hip_ab.so: the product library reads no environment variable), so the first call switches the
    rest of the test to that variant; estimators must be created AFTER it."""
    import contextlib
    from dcarl_amd import _lib
    stack = contextlib.ExitStack()
    on = []

    def set_knob(name, value):
        if not on:
            lib = stack.enter_context(_lib.use_variant("ab"))
            assert lib.dcarl_build_id().endswith(b"+ab")
            on.append(True)
        monkeypatch.setenv(name, value)

    yield set_knob
    stack.close()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def sim1_data():
    return (np.load(os.path.join(REPO, "Simulation_testing/Simulation_1/data_carla.npy")),
            np.load(os.path.join(REPO, "Simulation_testing/Simulation_1/action_value_carla.npy")))


@pytest.fixture(scope="session")
def sim2_data():
    return (np.load(os.path.join(REPO, "Simulation_testing/Simulation_2/data.npy")),
            np.load(os.path.join(REPO, "Simulation_testing/Simulation_2/action_value.npy")))
