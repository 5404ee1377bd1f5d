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
    """pre_loras / post_loras chaining (reference models.py:232-243, 249-265, 276-282; mix_lora_and_control_lora.py):
    product processors (unfused generic path on the emulated kernels) vs the oracle restatement, outputs and grads."""
    from controllora_amd import models as M, unet as U
    from oracle import cases, controllora_ref as cr, unet_ref
    torch.manual_seed(0)
    for self_attn in (True, False):
        cad = None if self_attn else 48
        o_attn = unet_ref.CrossAttention(64, cad, heads=4, dim_head=16)
        cases.seeded_weights_(o_attn, seed=5)
        p_attn = U.CrossAttention(64, cad, heads=4, dim_head=16)
        with torch.no_grad():
            for k, v in p_attn.state_dict().items():
                v.copy_(o_attn.state_dict()[k].to(v.dtype))
        if kind == "v1":
            o_main, p_main = cr.ControlLoRAProcRef(64, cad, rank=4), M.ControlLoRACrossAttnProcessor(64, cad, rank=4)
        else:
            o_main = cr.ControlLoRAProcV2Ref(64, cad, rank=4, control_channels=32)
            p_main = M.ControlLoRACrossAttnProcessorV2(64, cad, rank=4, control_channels=32)
        o_pre, p_pre = cr.LoRAProcRef(64, cad, rank=4), M.LoRACrossAttnProcessor(64, cad, rank=4)
        o_post, p_post = cr.LoRAProcRef(64, cad, rank=8, post_add=True), M.LoRACrossAttnProcessor(64, cad, rank=8, post_add=True)
        for o, p_, sd in ((o_main, p_main, 1), (o_pre, p_pre, 2), (o_post, p_post, 3)):
            cases.seeded_weights_(o, seed=sd)
            p_.load_state_dict(o.state_dict())
        o_main.inject_pre_lora(o_pre); o_main.inject_post_lora(o_post)
        p_main.inject_pre_lora(p_pre); p_main.inject_post_lora(p_post)
        h = torch.randn(2, 16, 64).half()
        e = None if self_attn else torch.randn(2, 5, 48).half()
        ctrl = torch.randn(2, 64 if kind == "v1" else 32, 4, 4).half()
        go = torch.randn(2, 16, 64).half()
        # oracle (fp32, fp16-rounded inputs / frozen weights)
        for q in o_attn.parameters():
            q.data = q.data.half().float()
        ho = h.float().requires_grad_(True)
        co = ctrl.float().requires_grad_(True)
        o_main.inject_control_states(co)
        yo = o_main(o_attn, ho, None if e is None else e.float(), None, 0.7)
        yo.backward(go.float())
        # product
        hp = h.clone().requires_grad_(True)
        cp = ctrl.permute(0, 2, 3, 1).reshape(2, 16, -1).contiguous().requires_grad_(True)
        p_main.inject_control_states(cp)
        yp = p_main(p_attn, hp, e, None, 0.7)
        yp.backward(go)
        assert E.rel(yp, yo.detach()) < 4e-3
        assert E.rel(hp.grad, ho.grad) < 1e-2
        assert E.rel(cp.grad, co.grad.permute(0, 2, 3, 1).reshape(2, 16, -1)) < 2e-2
        for (n, a), (_, b_) in zip(list(p_main.named_parameters()) + list(p_pre.named_parameters()) + list(p_post.named_parameters()),
                                   list(o_main.named_parameters()) + list(o_pre.named_parameters()) + list(o_post.named_parameters())):
            if b_.grad is not None and float(b_.grad.norm()) > 0:
                assert E.rel(a.grad, b_.grad) < 3e-2, (n, E.rel(a.grad, b_.grad))



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
