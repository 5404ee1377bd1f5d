import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU skips the `gpu` tests instead of failing them (the driver selects with
    -m gpu / -m "not gpu"; this is for everybody else)."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _no_groupnorm_team_exchange_gave_up(request):
    """every GPU test: the one-launch GroupNorm kernels exchange partial sums between workgroups inside a launch with a bounded wait;
    a wait that gave up leaves a sticky flag in the caller-owned state (controllora_amd.kernels.gn_team_errors) -- never, in any test"""
    yield
    if "gpu" in request.keywords:
        import torch
        from controllora_amd import kernels as K
        for d in range(torch.cuda.device_count()):
            assert K.gn_team_errors(f"cuda:{d}") == 0, "a GroupNorm team kernel's in-launch exchange gave up"
