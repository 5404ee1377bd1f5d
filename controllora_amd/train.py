"""The training step of the path (reference train_text_to_image_control_lora.py:751-796), device resident:

  hint-encode guide -> UNet(noisy latents) -> fp32 MSE -> scaled backward (dgrad through the frozen UNet,
  wgrad only for adapters + hint encoder) -> [RCCL all-reduce of ONE flat fp32 gradient buffer] ->
  fused unscale + clip_grad_norm_(1.0) + AdamW (+ GradScaler skip / growth) over ONE flat parameter buffer.

No host synchronisation happens inside a step (loss and grad-norm stay on the device until asked for).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import kernels as K

f16, f32 = torch.float16, torch.float32


class FlatParams:
    """Re-homes every trainable parameter (and its .grad) as a view into one contiguous fp32 buffer, so the
    optimizer is one kernel and data parallelism is one all-reduce (24.19 MB for v1)."""

    def __init__(self, module: nn.Module):
        params = self._ordered(module)
        assert params and all(p.dtype == f32 for p in params)
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.numel = n
        self.data = torch.empty(n, dtype=f32, device=dev)
        self.grad = torch.zeros(n, dtype=f32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=f32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=f32, device=dev)
        off = 0
        self.params = params
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.data[off:off + k].copy_(p.reshape(-1))
                p.data = self.data[off:off + k].view_as(p)
                p.grad = self.grad[off:off + k].view_as(p)
                off += k

    @staticmethod
    def _ordered(module):
        """Trainable parameters, with the adapter matrices that the fused projection concatenates (q|k|v up
        weights, k|v down weights of one processor) placed back to back so the concatenation is a view."""
        seen, out = set(), []

        def add(p):
            if p is not None and p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                out.append(p)

        for m in module.modules():
            if hasattr(m, "to_q_lora"):
                for part in ("up", "down"):
                    for name in ("to_q_lora", "to_k_lora", "to_v_lora"):
                        layer = getattr(m, name, None)
                        if layer is not None:
                            add(getattr(layer, part).weight)
        for p in module.parameters():
            add(p)
        return out

    def zero_grad(self):
        self.grad.zero_()


class ControlLoRATrainer:
    def __init__(self, unet, control_lora, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0,
                 init_scale=65536.0, dynamic_scale=True, growth_interval=2000, process_group=None, world_size=1,
                 gradient_accumulation_steps=1, lr_lambda=None, comm=None):
        self.unet, self.control_lora = unet, control_lora
        trainable = {id(p) for p in control_lora.parameters()}
        for p in unet.parameters():          # the installed processors are sub-modules of the UNet too: keep them trainable
            if id(p) not in trainable:
                p.requires_grad_(False)
        self.flat = FlatParams(control_lora)
        dev = self.flat.data.device
        self.hp = dict(lr=lr, beta1=betas[0], beta2=betas[1], wd=weight_decay, eps=eps, max_norm=max_grad_norm,
                       dynamic=dynamic_scale, interval=growth_interval)
        self.state = torch.zeros(16, dtype=f32, device=dev)
        self.state[3] = init_scale
        self.state[11] = float(world_size)       # the optimizer kernels turn the all-reduce SUM into the mean
        self.loss_sum = torch.zeros(1, dtype=f32, device=dev)
        self.pg, self.world, self._reduced = process_group, world_size, False
        # gradient accumulation (train...:174-178, `accelerator.accumulate`): micro-batch losses are divided by the
        # number of accumulation steps and the optimizer runs on the last micro-batch only
        self.accum, self._micro = max(1, int(gradient_accumulation_steps)), 0
        # LR schedule (train...:660-665 get_scheduler): multiplier written to state[10] before each optimizer step.  Units are
        # accelerate's: `accelerator.prepare(lr_scheduler)` steps the wrapped LambdaLR `num_processes` times per optimizer step, so
        # the reference's multiplier at optimizer step s is lambda(s * world) -- with N GPUs it warms up and decays N times faster
        # than the flag values read (the entry point builds lambda from warmup * accum / max_train_steps * accum like train...:675-680)
        self.lr_lambda, self.global_step = lr_lambda, 0
        self.sched_epoch = 0          # `last_epoch` of the reference's wrapped LambdaLR: += world per optimizer step
        # Exchange step: "torch" = torch.distributed.all_reduce on the process group (backend "nccl" IS RCCL on ROCm; gloo in
        # the CPU tests); "clora" = the C ABI's own RCCL communicator (clora_comm_init / clora_allreduce_flat_f32), created
        # from a unique id that rank 0 draws and the process group broadcasts.  Both enqueue the same ncclAllReduce on the
        # current stream; "clora" is what a non-torch host would call.  Default ("auto"): "clora" when there is something to
        # exchange over RCCL (world > 1 on an "nccl" process group), "torch" otherwise; CLORA_COMM / comm= force either.  A
        # requested "clora" that cannot be set up (no RCCL behind the group, communicator creation failed on ANY rank) falls
        # back to "torch" LOUDLY: a warning on stderr and `comm_fallback` (reported in the bench line).
        import os as _os
        import sys as _sys
        self.comm_requested = comm or _os.environ.get("CLORA_COMM", "auto")
        if self.comm_requested not in ("auto", "torch", "clora"):
            raise ValueError(f"comm={self.comm_requested!r}: expected 'auto', 'torch' or 'clora'")
        backend = torch.distributed.get_backend(process_group) if world_size > 1 else None
        self.comm_fallback = None
        if self.comm_requested == "auto":
            self.comm = "clora" if (world_size > 1 and backend == "nccl") else "torch"
        else:
            self.comm = self.comm_requested
        if self.comm == "clora" and world_size > 1 and backend != "nccl":
            self.comm_fallback = f'process group backend is {backend!r}: no RCCL behind it'
            self.comm = "torch"
        if world_size > 1:
            torch.distributed.broadcast(self.flat.data, src=0, group=process_group)   # identical adapter init
        if self.comm == "clora":
            # (1) can EVERY rank load librccl through the C ABI?  ncclCommInitRank is a collective: a rank that cannot even open the
            # library must be found BEFORE the others enter it, or they would wait for it forever
            err = None if self.comm_library(force=True) else "librccl could not be opened through the C ABI (clora_comm_library)"
            if world_size > 1:
                flag = torch.tensor([int(err is not None)], dtype=torch.int32, device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=process_group)
                if int(flag.item()) and err is None:
                    err = "another rank could not open librccl through the C ABI"
            # (2) the collective communicator creation, (3) agreed on by all ranks below
            if err is None:
                try:
                    self._init_clora_comm(process_group, world_size)
                except Exception as e:                               # noqa: BLE001 -- decided collectively below
                    err = repr(e)
            bad = int(err is not None)
            if world_size > 1:                                       # every rank must take the same exchange path
                flag = torch.tensor([bad], dtype=torch.int32, device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=process_group)
                bad = int(flag.item())
            if bad:
                if self.comm_requested == "clora" and world_size <= 1:
                    raise RuntimeError(f"comm='clora' requested but the C ABI communicator could not be created: {err}")
                self.comm_fallback = f"clora_comm_init failed on at least one rank ({err})"
                try:
                    from . import capi
                    capi.lib().call("clora_comm_destroy")
                except Exception:                                    # noqa: BLE001
                    pass
                self.comm = "torch"
        if self.comm_fallback:
            print(f"[controllora_amd] WARNING: exchange path 'clora' unavailable, using torch.distributed instead: {self.comm_fallback}",
                  file=_sys.stderr, flush=True)

    def comm_library(self, force=False):
        """path of the librccl whose ncclAllReduce the exchange step calls (C-ABI path: dladdr of the bound symbol)"""
        if self.comm != "clora" and not force:
            return None
        import ctypes
        from . import capi
        buf = ctypes.create_string_buffer(1024)
        try:
            capi.lib().call("clora_comm_library", buf, 1024)
        except Exception:                                            # noqa: BLE001
            return None
        return buf.value.decode(errors="replace")

    def _init_clora_comm(self, pg, world):
        """one RCCL communicator per process behind the C ABI: rank 0 draws the 128-byte unique id, the torch process group
        (or nothing, world 1) carries it to the other ranks, every rank joins"""
        import ctypes
        from . import capi
        L = capi.lib()
        rank = torch.distributed.get_rank(pg) if world > 1 else 0
        have = L.cdll.clora_comm_world()
        # one communicator per process: reuse it only if every rank of THIS group already holds one of the same shape (a second
        # trainer on the same group); a communicator built for another world size / rank layout is destroyed and rebuilt.  The
        # decision is made collectively -- ncclCommInitRank is a collective, a rank that skipped it would hang the others.
        same = int(have == world and L.cdll.clora_comm_rank() == rank)
        if world > 1:
            flag = torch.tensor([same], dtype=torch.int32, device=self.flat.data.device if pg is None or
                                torch.distributed.get_backend(pg) == "nccl" else "cpu")
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=pg)
            same = int(flag.item())
        if same:
            return
        if have:
            L.call("clora_comm_destroy")
        uid = (ctypes.c_char * 128)()
        if rank == 0:
            L.call("clora_comm_unique_id", uid)
        if world > 1:
            box = [bytes(uid)] if rank == 0 else [None]
            torch.distributed.broadcast_object_list(box, src=0, group=pg)
            uid = (ctypes.c_char * 128).from_buffer_copy(box[0])
        L.call("clora_comm_init", uid, rank, world)

    def comm_ranks(self) -> int:
        """ranks of the communicator the exchange step runs on (1 when the step has no exchange)"""
        if self.world <= 1 and self.comm != "clora":
            return 1
        if self.comm == "clora":
            from . import capi
            return int(capi.lib().cdll.clora_comm_world())
        return torch.distributed.get_world_size(self.pg)

    def close(self):
        """release the C ABI's RCCL communicator (a no-op for comm="torch"); call before the process group is destroyed"""
        if self.comm == "clora":
            from . import capi
            capi.lib().call("clora_comm_destroy")

    def _all_reduce_grads(self):
        g = self.flat.grad
        if self.comm == "clora":
            from . import capi
            capi.lib().call("clora_allreduce_flat_f32", capi.ptr(g, f32), g.numel(), capi.stream())
        else:
            torch.distributed.all_reduce(g, group=self.pg)           # RCCL over xGMI: one flat 24 MB buffer (SUM; the 1/N of
            #                                                          the mean is folded into optim_prep)

    # -- pieces (kept separate so tests can check each against the oracle)
    def forward_backward(self, noisy_latents, timesteps, encoder_hidden_states, guide, target):
        K.lora_wgrad_discard()                                      # nothing may be pending from an aborted step
        if self._micro == 0:
            self.flat.zero_grad()
            self._reduced = False
        self.loss_sum.zero_()
        self.control_lora(guide)                                    # injects control states into the 32 processors
        pred = self.unet(noisy_latents, timesteps, encoder_hidden_states).sample
        pred_c = pred.contiguous()
        dpred = torch.empty_like(pred_c)
        n = pred_c.numel()
        K.mse(pred_c, target.to(f16).contiguous(), self.loss_sum, dpred, 2.0 / n / self.accum, self.state[3:4])
        pred_c.backward(dpred)
        K.lora_wgrad_flush()                                        # queued adapter weight gradients (no-op if already flushed)
        return pred

    def optimizer_step(self):
        """all-reduce + clip + AdamW; with gradient accumulation only every `accum`-th call does anything"""
        self._micro += 1
        if self._micro < self.accum:
            return False
        self._micro = 0
        if self.lr_lambda is not None:
            self.state[10:11].fill_(max(float(self.lr_lambda(self.sched_epoch)), 1e-30))
        self.global_step += 1
        self.sched_epoch += self.world
        if self.world > 1:
            self._all_reduce_grads()
            self._reduced = True
        self._optimizer_kernels()
        return True

    def _optimizer_kernels(self):
        g = self.flat.grad
        K.grad_sumsq(g, self.state)
        K.optim_prep(self.state, self.hp["max_norm"], self.hp["beta1"], self.hp["beta2"], self.hp["dynamic"], 2.0, 0.5,
                     self.hp["interval"])
        K.adamw_flat(self.flat.data, g, self.flat.exp_avg, self.flat.exp_avg_sq, self.state, self.hp["lr"], self.hp["beta1"],
                     self.hp["beta2"], self.hp["eps"], self.hp["wd"])
        from . import ops
        ops.repack_adapters()       # the fp16 operand blocks of the adapters that ride in their projection GEMMs (one launch)

    def step(self, noisy_latents, timesteps, encoder_hidden_states, guide, target):
        pred = self.forward_backward(noisy_latents, timesteps, encoder_hidden_states, guide, target)
        self.optimizer_step()
        return pred

    # -- hipGraph capture -------------------------------------------------------------------------------------
    def capture(self, noisy_latents, timesteps, encoder_hidden_states, guide, target, warmup=2):
        """Capture the step into hipGraphs (the step issues ~2500 small launches; replaying a graph removes
        the Python / launch overhead that otherwise bounds it).  Shapes become static: ``step_graphed`` copies a
        new batch into the captured input buffers and replays.  With data parallelism the all-reduce stays an
        eager RCCL call between the forward/backward graph and the optimizer graph."""
        self._static = [t.clone() for t in (noisy_latents, timesteps, encoder_hidden_states, guide, target)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                       # warm-up on a side stream (allocator / lazy packs settle)
            for _ in range(warmup):
                self.forward_backward(*self._static)
                self.optimizer_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb):
            self._static_pred = self.forward_backward(*self._static)
        self._capture_optimizer_graph()
        return self

    def _capture_optimizer_graph(self):
        """exchange + clip + AdamW + repack as one hipGraph.  With the C ABI's communicator (comm "clora") the all-reduce is captured
        INSIDE it -- RCCL enqueues on the capturing stream -- so a rank's step is two graph replays and no host-issued collective
        (round 6; reference train...:683-685, 790: DDP's reduce inside `accelerator.backward`).  torch.distributed's all-reduce
        (comm "torch") stays an eager call between the two graphs; so does "clora" if RCCL refuses the capture (loud fallback)."""
        from . import ops
        import sys as _sys
        torch.cuda.synchronize()
        want = self.comm == "clora" and (self.world > 1 or getattr(self, "exchange_at_world_1", False)) and \
            getattr(self, "_exchange_in_graph", None) is not False
        self._exchange_in_graph = False
        if want:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self._g_fb.pool()):
                    self._all_reduce_grads()
                    self._optimizer_kernels()
                self._g_opt, self._exchange_in_graph = g, True
            except Exception as e:                                   # noqa: BLE001 -- RCCL refused the capture: eager exchange
                torch.cuda.synchronize()
                self.exchange_capture_error = repr(e)
                print(f"[controllora_amd] WARNING: the RCCL all-reduce could not be captured into the optimizer graph ({e!r}); "
                      "it stays an eager call between the two graphs", file=_sys.stderr, flush=True)
        if not self._exchange_in_graph:
            if want:
                self._exchange_in_graph = False                      # remembered: later re-captures do not try again
            self._g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_opt, pool=self._g_fb.pool()):
                self._optimizer_kernels()
        self._pack_epoch = ops.ADAPTER_PACKS.epoch       # the captured repack launch covers the adapter groups registered so far

    def step_graphed(self, noisy_latents=None, timesteps=None, encoder_hidden_states=None, guide=None, target=None):
        for dst, src in zip(self._static, (noisy_latents, timesteps, encoder_hidden_states, guide, target)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        assert self.accum == 1, "the captured step assumes one micro-batch per optimizer step"
        self._reduced = False
        self._g_fb.replay()
        if self._exchange_in_graph:
            self._reduced = self.world > 1                           # the all-reduce is the first node of the optimizer graph
        elif self.world > 1:
            self._all_reduce_grads()
            self._reduced = True
        if self.lr_lambda is not None:
            self.state[10:11].fill_(max(float(self.lr_lambda(self.sched_epoch)), 1e-30))
        self.global_step += 1
        self.sched_epoch += self.world
        self._g_opt.replay()
        from . import ops
        if ops.ADAPTER_PACKS.epoch != self._pack_epoch:
            # a group registered after the capture (e.g. run_validation at a batch size whose projections fuse differently): the
            # captured repack launch does not know it and the flat AdamW does not bump its parameters' _version -- repack eagerly
            # THIS step, then capture the optimizer graph again so that its repack launch covers the new table (once per new
            # group, not an extra eager launch on every later step: ADVICE r05)
            ops.repack_adapters()
            self._capture_optimizer_graph()
        return self._static_pred

    # -- checkpoint / resume (train...:713-735 `accelerator.save_state / load_state`: weights, optimizer moments,
    # GradScaler state and the step counters; one flat tensor each because the trainer state IS flat)
    def state_dict(self) -> Dict[str, torch.Tensor]:
        # a checkpoint is a host sync anyway: the place to notice that a one-launch GroupNorm gave up an in-launch exchange at some point
        # (sticky flag of the team state, kernels.gn_team_errors) -- the weights would then come from at least one step on invalid norms
        dev = self.flat.data.device
        if dev.type == "cuda" and K.gn_team_errors(dev):
            raise RuntimeError("a GroupNorm team kernel gave up an in-launch exchange during this run (kernels.gn_team_errors != 0): "
                               "the state is not trustworthy; set CLORA_GN_TEAM=0 and report the device / co-tenant situation")
        return {"params": self.flat.data.detach().cpu().clone(), "exp_avg": self.flat.exp_avg.cpu().clone(),
                "exp_avg_sq": self.flat.exp_avg_sq.cpu().clone(), "state": self.state.cpu().clone(),
                "global_step": torch.tensor([self.global_step], dtype=torch.int64),
                "sched_epoch": torch.tensor([self.sched_epoch], dtype=torch.int64)}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        if sd["params"].numel() != self.flat.numel:
            raise ValueError(f"checkpoint holds {sd['params'].numel()} trainable values, the model has {self.flat.numel}")
        with torch.no_grad():
            self.flat.data.copy_(sd["params"])
            self.flat.exp_avg.copy_(sd["exp_avg"])
            self.flat.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.state.copy_(sd["state"])
            self.state[11] = float(self.world)       # the divisor belongs to THIS run's world size, not the checkpoint's
        self.global_step, self._micro = int(sd["global_step"][0]), 0
        self.sched_epoch = int(sd["sched_epoch"][0]) if "sched_epoch" in sd else self.global_step * self.world
        from . import ops
        ops.repack_adapters()        # the copies above do not bump the parameters' _version: refresh the fp16 operand packs now

    def save_state(self, directory: str) -> None:
        import os
        from safetensors.torch import save_file
        os.makedirs(directory, exist_ok=True)
        save_file(self.state_dict(), os.path.join(directory, "trainer_state.safetensors"))
        self.control_lora.save_pretrained(directory)

    def load_state(self, directory: str) -> None:
        """`checkpoint-N` directory written by this trainer (trainer_state.safetensors) or by the REFERENCE's
        `accelerator.save_state` (reference train_text_to_image_control_lora.py:713-735 / test_...:705-722): model weights +
        torch AdamW state + GradScaler + LR-scheduler files, see load_accelerate_state."""
        import os
        from safetensors.torch import load_file
        own = os.path.join(directory, "trainer_state.safetensors")
        if os.path.exists(own):
            self.load_state_dict(load_file(own))
        else:
            self.load_accelerate_state(directory)

    def _module_order_views(self):
        """(parameter, offset into the flat buffers) in `control_lora.parameters()` order -- the order torch.optim.AdamW numbers
        its state in when it is built from `control_lora.parameters()` as the reference does (train...:607-613)"""
        base, out = self.flat.data.data_ptr(), []
        for p in self.control_lora.parameters():
            if p.requires_grad:
                off = (p.data_ptr() - base) // 4
                assert 0 <= off and off + p.numel() <= self.flat.numel
                out.append((p, off))
        return out

    def load_accelerate_state(self, directory: str) -> None:
        """Resume from a directory in accelerate's `save_state` layout: `pytorch_model.bin` or `model.safetensors` (the
        ControlLoRA state dict), `optimizer.bin` (torch AdamW state_dict: per-parameter step / exp_avg / exp_avg_sq numbered in
        `control_lora.parameters()` order), optional `scaler.pt` (GradScaler: scale, _growth_tracker) and `scheduler.bin`
        (LambdaLR: last_epoch = optimizer steps taken)."""
        import os
        f = lambda n: os.path.join(directory, n)
        if os.path.exists(f("model.safetensors")):
            from safetensors.torch import load_file
            sd = load_file(f("model.safetensors"))
        elif os.path.exists(f("pytorch_model.bin")):
            sd = torch.load(f("pytorch_model.bin"), map_location="cpu")
        else:
            raise FileNotFoundError(f"{directory}: neither trainer_state.safetensors nor an accelerate checkpoint (pytorch_model.bin / model.safetensors)")
        self.control_lora.load_state_dict(sd)                    # in-place copies: the parameters stay views of the flat buffer
        opt = torch.load(f("optimizer.bin"), map_location="cpu")
        views = self._module_order_views()
        ids = [i for g in opt["param_groups"] for i in g["params"]]
        if len(ids) != len(views):
            raise ValueError(f"optimizer.bin holds {len(ids)} parameters, the model has {len(views)} trainable ones")
        steps = set()
        with torch.no_grad():
            self.flat.exp_avg.zero_(); self.flat.exp_avg_sq.zero_()
            for i, (p, off) in zip(ids, views):
                st = opt["state"].get(i)
                if st is None:
                    continue                                      # a parameter that never received a gradient
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
                self.flat.exp_avg[off:off + p.numel()].copy_(st["exp_avg"].reshape(-1))
                self.flat.exp_avg_sq[off:off + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            step = max(steps) if steps else 0
            self.state.zero_()
            self.state[2] = float(step)                           # bias corrections are recomputed from it (clora_optim_prep_f32)
            self.state[3] = 65536.0
            if os.path.exists(f("scaler.pt")):
                sc = torch.load(f("scaler.pt"), map_location="cpu")
                self.state[3] = float(sc.get("scale", 65536.0))
                self.state[4] = float(sc.get("_growth_tracker", 0))
            self.state[11] = float(self.world)
        # optimizer steps taken = the AdamW `step` (parsed above).  scheduler.bin is only a cross-check: accelerate's
        # AcceleratedScheduler advances the wrapped scheduler num_processes times per optimizer step (train...:660-665 builds the
        # schedule in those units), so a multi-GPU checkpoint holds last_epoch = world_of_the_saving_run x steps
        self.global_step = step
        self.sched_epoch = step * self.world
        self.resumed_scheduler_ratio = None
        if os.path.exists(f("scheduler.bin")):
            last = int(torch.load(f("scheduler.bin"), map_location="cpu", weights_only=False).get("last_epoch", step))
            self.sched_epoch = last                          # the schedule continues from the saved position, like accelerate.load_state
            if step > 0:
                self.resumed_scheduler_ratio = last / step   # = the saving run's process count (1 for a single-GPU run)
        self._micro = 0
        from . import ops
        ops.repack_adapters()

    def save_accelerate_state(self, directory: str, lr: float = None) -> None:
        """the same checkpoint in accelerate's layout, so a reference run (`accelerator.load_state`) can resume from it"""
        import os
        os.makedirs(directory, exist_ok=True)
        torch.save({k: v.detach().cpu().clone() for k, v in self.control_lora.state_dict().items()}, os.path.join(directory, "pytorch_model.bin"))
        views = self._module_order_views()
        step = torch.tensor(float(self.state[2]))
        state = {i: {"step": step.clone(), "exp_avg": self.flat.exp_avg[off:off + p.numel()].reshape(p.shape).cpu().clone(),
                     "exp_avg_sq": self.flat.exp_avg_sq[off:off + p.numel()].reshape(p.shape).cpu().clone()}
                 for i, (p, off) in enumerate(views)}
        group = {"lr": self.hp["lr"] if lr is None else lr, "betas": (self.hp["beta1"], self.hp["beta2"]), "eps": self.hp["eps"],
                 "weight_decay": self.hp["wd"], "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "params": list(range(len(views)))}
        torch.save({"state": state, "param_groups": [group]}, os.path.join(directory, "optimizer.bin"))
        torch.save({"scale": float(self.state[3]), "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": self.hp["interval"],
                    "_growth_tracker": int(self.state[4])}, os.path.join(directory, "scaler.pt"))
        # accelerate's unit: the wrapped scheduler is stepped `num_processes` times per optimizer step.  The file is the state_dict of
        # a real torch LambdaLR (every key `LambdaLR.load_state_dict` of the installed torch expects, `lr_lambdas` = [None] for a
        # plain function as torch writes it), so `accelerator.load_state` of a reference run can take it
        base_lr = self.hp["lr"] if lr is None else lr
        last = self.sched_epoch
        mult = float(self.lr_lambda(last)) if self.lr_lambda is not None else 1.0
        dummy = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=base_lr)
        sched = torch.optim.lr_scheduler.LambdaLR(dummy, self.lr_lambda if self.lr_lambda is not None else (lambda _: 1.0))
        sd = sched.state_dict()
        sd.update(last_epoch=last, _step_count=last + 1, _last_lr=[base_lr * mult], base_lrs=[base_lr])
        torch.save(sd, os.path.join(directory, "scheduler.bin"))

    # -- host-visible scalars (each forces a sync; call outside the timed region)
    def loss(self, numel) -> float:
        return float(self.loss_sum) / numel

    def unscaled_grads(self) -> torch.Tensor:
        """flat gradient in the internal buffer order (see FlatParams._ordered)"""
        return self.flat.grad / (self.state[3] * (self.world if self._reduced else 1))

    def unscaled_grads_module_order(self) -> torch.Tensor:
        """flat gradient in ``control_lora.parameters()`` order (what the reference / oracle would concatenate)"""
        return torch.cat([p.grad.reshape(-1) for p in self.control_lora.parameters() if p.requires_grad]) / (
            self.state[3] * (self.world if self._reduced else 1))
