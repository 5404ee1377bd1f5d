"""CPU end-to-end test of the HOST LOGIC (autograd functions, module plumbing, processor mapping, trainer)
with the kernels running under the host emulator: one reference-style train step on the seeded small
cases must reproduce the golden vectors produced by the reference's own models.py."""
import pytest
import torch

from tests import e2e_cases as E
from tests.emu_fixture import use_emulator


@pytest.fixture(autouse=True)
def _emu():
    with use_emulator():
        yield


@pytest.mark.parametrize("case", ["v1", "v2", "sketch", "lora"])
def test_train_step_matches_reference_golden(case, golden_dir):
    errs = E.check_against_golden(case, "cpu", golden_dir)
    print(case, errs)
