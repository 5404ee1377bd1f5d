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


@pytest.mark.parametrize("case", ["v1", "v2", "sketch", "lora", "postadd"])
def test_train_step_matches_reference_golden(case, golden_dir):
    errs = E.check_against_golden(case, "cpu", golden_dir)
    print(case, errs)


@pytest.mark.parametrize("kind", ["v1", "v2"])
def test_pre_post_lora_chain_matches_oracle(kind):
    """host logic of the chained-adapter path on the emulated kernels (the GPU suite runs the same check on the real
    library: tests/test_e2e_gpu.py)"""
    print(kind, E.check_pre_post_chain(kind, "cpu"))


def test_trainer_accumulation_lr_schedule_and_resume(golden_dir):
    """host logic of gradient accumulation / LR multiplier / checkpoint-resume with synthetic gradients (the full
    version with real forward/backward passes runs in the GPU suite: tests/test_e2e_gpu.py)"""
    E.check_trainer_features("cpu", golden_dir, real_backward=False)


def test_vae_encode_decode_matches_oracle():
    from tests import vae_cases
    print(vae_cases.check_vae("cpu", res=32, batch=1))


@pytest.mark.parametrize("case", ["v1", "v2"])
def test_inference_with_control_batch_broadcast(case):
    print(case, E.check_inference_broadcast(case, "cpu"))
