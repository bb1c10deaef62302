import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_ready():
    """-m gpu tests need a visible GPU and the in-tree HIP library (there is no CPU fallback to run instead)."""
    lib = os.path.join(ROOT, "alore_legged_manipulator_amd", "libalore_nmpc.so")
    if not os.path.exists(lib):
        return False, "libalore_nmpc.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    try:
        import torch
        if not torch.cuda.is_available():
            return False, "no GPU visible"
    except Exception as e:  # pragma: no cover
        return False, f"torch unavailable: {e}"
    return True, ""


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items:
        return
    ok, why = _gpu_ready()
    if ok:
        return
    skip = pytest.mark.skip(reason=why)
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """The oracle is test infrastructure: build it (plain gcc) before any test."""
    from oracle import drivers
    drivers.build(ref=os.path.isdir("/root/reference"))
