"""Worker for tests/test_distributed_cpu.py: one rank of a world-size-2 `gloo` data-parallel train step
(the N>1 path of controllora_amd.train.ControlLoRATrainer) with the kernels running under the host emulator."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir = sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu_fixture import use_emulator
    from tests import e2e_cases as E
    from controllora_amd.train import ControlLoRATrainer
    from oracle import cases, unet_ref
    with use_emulator():
        unet, clora, _ = E.build_product_case(os.environ.get("CLORA_DIST_CASE", "v1"), "cpu")
        if rank >= 1:                      # perturb the other ranks: the trainer must broadcast rank 0's adapters
            with torch.no_grad():
                for p in clora.parameters():
                    p.add_(0.01 * rank)
        trainer = ControlLoRATrainer(unet, clora, init_scale=128.0, dynamic_scale=False, process_group=dist.group.WORLD,
                                     world_size=world, comm=os.environ.get("CLORA_DIST_COMM") or None)
        # the default ("auto") picks torch.distributed on a gloo group without comment; a REQUESTED "clora" cannot run here (no RCCL
        # behind gloo) and must say so instead of silently switching
        want = os.environ.get("CLORA_DIST_COMM") or "auto"
        assert trainer.comm == "torch" and trainer.comm_requested == want
        assert (trainer.comm_fallback is not None) == (want == "clora"), trainer.comm_fallback
        full = cases.seeded_inputs(batch=world)
        sl = slice(rank, rank + 1)           # rank r gets sample r of the global batch of `world`
        noisy = unet_ref.DDPMSchedule().add_noise(full["latents"], full["noise"], full["timesteps"])
        trainer.forward_backward(noisy[sl].half(), full["timesteps"][sl], full["ehs"][sl].half(), full["guide"][sl].half(),
                                 full["noise"][sl])
        local_grad = trainer.unscaled_grads().clone()
        trainer.optimizer_step()
        torch.save(dict(local_grad=local_grad, reduced_grad=trainer.unscaled_grads().clone(), params=trainer.flat.data.clone(),
                        loss=trainer.loss(noisy[sl].numel())), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
