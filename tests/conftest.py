import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """The .so files are git-ignored build outputs: build them when a fresh checkout has none (hipcc cross-compiles
    without a GPU; the oracle's C restatement needs only gcc)."""
    from geometric_aware_dense_matching_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    from oracle import knn as oknn
    if not os.path.exists(oknn._ORACLE_SO):
        oknn.build(ref=True)
    yield


@pytest.fixture(scope="session", autouse=True)
def _poison_uninitialised_memory():
    """GDM_TEST_POISON=1: every torch.empty() comes back filled with NaN (torch.utils.deterministic.fill_uninitialized_memory under
    use_deterministic_algorithms(warn_only)): an output element a kernel does not write, or a workspace word it reads before writing,
    then shows up as NaN in whatever the test compares.  An audit mode, off by default (the fills cost time)."""
    if os.environ.get("GDM_TEST_POISON") != "1":
        yield
        return
    import torch
    import warnings
    warnings.filterwarnings("ignore", message=".*does not have a deterministic implementation.*")
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
    yield
    torch.use_deterministic_algorithms(False)
