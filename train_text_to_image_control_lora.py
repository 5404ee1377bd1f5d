#!/usr/bin/env python
"""Drop-in training entry point of the MI355X-native ControlLoRA path: accepts the 44 flags of the reference's
`parse_args` (reference train_text_to_image_control_lora.py:84-326) and runs the same step (:751-796) -- VAE encode
x 0.18215, noise, timesteps, add_noise, text encode, hint-encode + inject, UNet, fp32 MSE, scaled backward, clip,
AdamW, LR schedule, `checkpoint-N` states, final `save_pretrained` (.bin and .safetensors) -- on the gfx950 kernels
(controllora_amd), one process per GPU under `python -m torch.distributed.run` with an RCCL all-reduce of the flat
adapter-gradient buffer instead of accelerate/DDP.

What differs, because this image has no network / weights / torchvision: `--pretrained_model_name_or_path` may be
`random:sd15` (seeded random SD-1.5-shaped UNet + VAE + CLIP) and `--dataset_name` may be `synthetic:fill50k`
(generated circles, tasks/make_dataset_fill50k.py semantics); flags that configure services absent here
(`--push_to_hub`, `--hub_*`, `--report_to`, `--use_8bit_adam`, `--enable_xformers_memory_efficient_attention`,
`--allow_tf32`, `--gradient_checkpointing`) are accepted and reported as no-ops.
"""
from __future__ import annotations

import argparse
import json
import logging
import math
import os
import time

import torch

logger = logging.getLogger("control_lora.train")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="ControlLoRA training on MI355X (reference-compatible flags).")
    p.add_argument("--pretrained_model_name_or_path", type=str, default=None, required=True)
    p.add_argument("--revision", type=str, default=None, required=False)
    p.add_argument("--dataset_name", type=str, default=None)
    p.add_argument("--dataset_config_name", type=str, default=None)
    p.add_argument("--train_data_dir", type=str, default=None)
    p.add_argument("--image_column", type=str, default="image")
    p.add_argument("--guide_column", type=str, default="guide")
    p.add_argument("--caption_column", type=str, default="text")
    p.add_argument("--validation_prompt", type=str, default=None)
    p.add_argument("--num_validation_images", type=int, default=4)
    p.add_argument("--validation_epochs", type=int, default=1)
    p.add_argument("--max_train_samples", type=int, default=None)
    p.add_argument("--output_dir", type=str, default="sd-fill50k-model-control-lora")
    p.add_argument("--cache_dir", type=str, default=None)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--resolution", type=int, default=512)
    p.add_argument("--train_batch_size", type=int, default=16)
    p.add_argument("--num_train_epochs", type=int, default=100)
    p.add_argument("--max_train_steps", type=int, default=None)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--gradient_checkpointing", action="store_true")
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--scale_lr", action="store_true", default=False)
    p.add_argument("--lr_scheduler", type=str, default="constant")
    p.add_argument("--lr_warmup_steps", type=int, default=500)
    p.add_argument("--use_8bit_adam", action="store_true")
    p.add_argument("--allow_tf32", action="store_true")
    p.add_argument("--dataloader_num_workers", type=int, default=0)
    p.add_argument("--adam_beta1", type=float, default=0.9)
    p.add_argument("--adam_beta2", type=float, default=0.999)
    p.add_argument("--adam_weight_decay", type=float, default=1e-2)
    p.add_argument("--adam_epsilon", type=float, default=1e-08)
    p.add_argument("--max_grad_norm", default=1.0, type=float)
    p.add_argument("--push_to_hub", action="store_true")
    p.add_argument("--hub_token", type=str, default=None)
    p.add_argument("--hub_model_id", type=str, default=None)
    p.add_argument("--logging_dir", type=str, default="logs")
    p.add_argument("--mixed_precision", type=str, default=None, choices=["no", "fp16", "bf16"])
    p.add_argument("--report_to", type=str, default="tensorboard")
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument("--checkpointing_steps", type=int, default=500)
    p.add_argument("--resume_from_checkpoint", type=str, default=None)
    p.add_argument("--enable_xformers_memory_efficient_attention", action="store_true")
    p.add_argument("--control_lora_config", type=str, required=True)
    # additions (not in the reference): eager launches instead of the captured hipGraph step
    p.add_argument("--no_hipgraph", action="store_true", help="do not capture the step into hipGraphs")
    args = p.parse_args(argv)
    env_local_rank = int(os.environ.get("LOCAL_RANK", -1))
    if env_local_rank != -1 and env_local_rank != args.local_rank:
        args.local_rank = env_local_rank
    if args.dataset_name is None and args.train_data_dir is None:
        raise ValueError("Need either a dataset name or a training folder.")
    return args


def build_dataset(args, tokenizer):
    from controllora_amd import data
    if args.dataset_name is not None and args.dataset_name.startswith("synthetic:"):
        n = args.max_train_samples or 50000
        return data.SyntheticFill50k(args.resolution, n, seed=args.seed if args.seed is not None else 42, tokenizer=tokenizer)
    if args.dataset_name is not None and args.dataset_name.startswith("process/"):       # reference train...:546-550
        from controllora_amd import process
        return process.Dataset.from_name(args.dataset_name)(tokenizer, resolution=args.resolution, use_crop=True)
    from datasets import load_dataset
    if args.dataset_name is not None:
        ds = load_dataset(args.dataset_name, args.dataset_config_name, cache_dir=args.cache_dir)
    else:
        ds = load_dataset("imagefolder", data_files={"train": os.path.join(args.train_data_dir, "**")}, cache_dir=args.cache_dir)
    rows = ds["train"]
    names = rows.column_names
    for flag, col in (("--image_column", args.image_column), ("--guide_column", args.guide_column),
                      ("--caption_column", args.caption_column)):
        if col not in names:
            raise ValueError(f"{flag}' value '{col}' needs to be one of: {', '.join(names)}")
    if args.max_train_samples is not None:
        rows = rows.shuffle(seed=args.seed).select(range(args.max_train_samples))
    return data.ImageGuideDataset(rows, args.image_column, args.guide_column, args.caption_column, args.resolution, tokenizer)


def latest_checkpoint(output_dir):
    if not os.path.isdir(output_dir):
        return None
    dirs = sorted((d for d in os.listdir(output_dir) if d.startswith("checkpoint-")), key=lambda d: int(d.split("-")[1]))
    return dirs[-1] if dirs else None


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", level=logging.INFO)
    from controllora_amd import data, loading, models as M, text
    from controllora_amd.schedulers import DDPMScheduler
    from controllora_amd.train import ControlLoRATrainer

    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    local = max(args.local_rank, 0)
    if not torch.cuda.is_available():
        raise RuntimeError("the ControlLoRA training path needs an MI355X (no CPU fallback is provided)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=dev)
    main_proc = rank == 0
    for flag in ("push_to_hub", "use_8bit_adam", "allow_tf32", "gradient_checkpointing", "enable_xformers_memory_efficient_attention"):
        if getattr(args, flag) and main_proc:
            logger.warning("--%s is accepted for compatibility and has no effect on this path", flag)
    if args.mixed_precision == "bf16":
        logger.warning("--mixed_precision=bf16: the gfx950 kernels compute in fp16 with fp32 accumulation; using fp16")
    elif args.mixed_precision in (None, "no") and main_proc:
        logger.warning("--mixed_precision=%s: this path always computes in fp16 (fp32 accumulation, fp32 master weights) "
                       "with dynamic loss scaling; the reference would train in fp32 here", args.mixed_precision)
    if args.seed is not None:
        torch.manual_seed(args.seed + rank)            # per-rank noise / timestep streams, identical model init below
    if main_proc and args.output_dir is not None:
        os.makedirs(args.output_dir, exist_ok=True)

    # ---- frozen base model + trainable ControlLoRA (reference :395-487)
    name = args.pretrained_model_name_or_path
    small = name.endswith("small")
    tokenizer = text.load_tokenizer(name)
    text_encoder = text.load_text_encoder(name, dev, small=small)
    vae = loading.load_vae(name, dev)
    unet = loading.load_unet(name, dev)
    noise_scheduler = DDPMScheduler()
    # Adapter init: with --seed every rank draws the same init from a forked stream (and the per-rank noise / timestep
    # stream seeded above is left untouched); without --seed the run stays unseeded like the reference -- rank 0's init
    # reaches the other ranks through the trainer's broadcast of the flat parameter buffer.
    # torch.manual_seed() also reseeds every HIP generator, so the device stream is forked too: otherwise all ranks would
    # draw identical noise / timesteps / VAE samples after this block.
    fork_devs = [torch.device(dev).index or 0] if torch.device(dev).type == "cuda" else []
    with torch.random.fork_rng(devices=fork_devs):
        if args.seed is not None:
            torch.manual_seed(args.seed)
        control_lora = M.ControlLoRA.from_config(args.control_lora_config).to(dev)
    unet.set_attn_processor(M.map_processors_to_unet(unet, control_lora))

    if args.scale_lr:
        args.learning_rate = args.learning_rate * args.gradient_accumulation_steps * args.train_batch_size * world

    # ---- data (reference :520-666); each rank reads its own shard of every global batch
    dataset = build_dataset(args, tokenizer)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, world, rank, shuffle=True, seed=args.seed or 0, drop_last=True) \
        if world > 1 else None
    loader = torch.utils.data.DataLoader(dataset, shuffle=sampler is None, sampler=sampler, collate_fn=data.collate,
                                         batch_size=args.train_batch_size, num_workers=args.dataloader_num_workers, drop_last=True)
    steps_per_epoch = math.ceil(len(loader) / args.gradient_accumulation_steps)
    if args.max_train_steps is None:
        args.max_train_steps = args.num_train_epochs * steps_per_epoch
    args.num_train_epochs = math.ceil(args.max_train_steps / steps_per_epoch)

    trainer = ControlLoRATrainer(
        unet, control_lora, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2), weight_decay=args.adam_weight_decay,
        eps=args.adam_epsilon, max_grad_norm=args.max_grad_norm, world_size=world,
        # The kernels compute in fp16 whatever --mixed_precision says (the reference's flag defaults to None because
        # precision normally comes from `accelerate launch --mixed_precision fp16`): a seed gradient of 2/n ~ 1e-5 is
        # subnormal in fp16, so loss scaling with the dynamic GradScaler is ALWAYS on for this path.
        init_scale=65536.0, dynamic_scale=True, gradient_accumulation_steps=args.gradient_accumulation_steps,
        # schedule units as the reference builds them (train...:675-680); the trainer evaluates it at optimizer step x world
        lr_lambda=data.lr_lambda(args.lr_scheduler, args.lr_warmup_steps * args.gradient_accumulation_steps,
                                 args.max_train_steps * args.gradient_accumulation_steps))

    global_step, first_epoch, resume_step = 0, 0, 0
    if args.resume_from_checkpoint:
        path = os.path.basename(args.resume_from_checkpoint) if args.resume_from_checkpoint != "latest" else latest_checkpoint(args.output_dir)
        if path is None or not os.path.isdir(os.path.join(args.output_dir, path)):
            logger.info("Checkpoint '%s' does not exist. Starting a new training run.", args.resume_from_checkpoint)
            args.resume_from_checkpoint = None
        else:
            logger.info("Resuming from checkpoint %s", path)
            trainer.load_state(os.path.join(args.output_dir, path))
            global_step = int(path.split("-")[1])
            first_epoch = global_step // steps_per_epoch
            resume_step = (global_step * args.gradient_accumulation_steps) % (steps_per_epoch * args.gradient_accumulation_steps)

    if main_proc:
        logger.info("***** Running training *****")
        logger.info("  Num examples = %d", len(dataset))
        logger.info("  Num Epochs = %d", args.num_train_epochs)
        logger.info("  Instantaneous batch size per device = %d", args.train_batch_size)
        logger.info("  Total train batch size (w. parallel, distributed & accumulation) = %d",
                    args.train_batch_size * world * args.gradient_accumulation_steps)
        logger.info("  Gradient Accumulation steps = %d", args.gradient_accumulation_steps)
        logger.info("  Total optimization steps = %d", args.max_train_steps)

    graphed = not args.no_hipgraph and args.gradient_accumulation_steps == 1
    captured = False
    log_path = os.path.join(args.output_dir, args.logging_dir, "train_log.jsonl")
    if main_proc:
        os.makedirs(os.path.dirname(log_path), exist_ok=True)
    t_last, imgs_last = time.perf_counter(), 0

    for epoch in range(first_epoch, args.num_train_epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        for step, batch in enumerate(loader):
            if args.resume_from_checkpoint and epoch == first_epoch and step < resume_step:
                continue
            with torch.no_grad():
                pixel = batch["pixel_values"].to(dev, non_blocking=True).half()
                guide = batch["guide_values"].to(dev, non_blocking=True).half()
                latents = vae.encode(pixel).latent_dist.sample() * vae.scaling_factor
                noise = torch.randn_like(latents)
                timesteps = torch.randint(0, noise_scheduler.num_train_timesteps, (latents.shape[0],), device=dev).long()
                noisy = noise_scheduler.add_noise(latents, noise, timesteps).half()
                ehs = text_encoder(batch["input_ids"].to(dev))[0].half()
                if noise_scheduler.prediction_type == "epsilon":
                    target = noise
                elif noise_scheduler.prediction_type == "v_prediction":
                    target = noise_scheduler.get_velocity(latents, noise, timesteps)
                else:
                    raise ValueError(f"Unknown prediction type {noise_scheduler.prediction_type}")
            if graphed:
                if not captured:
                    snap = trainer.state_dict()                 # the capture warm-up runs real steps: undo them
                    trainer.capture(noisy, timesteps, ehs, guide, target)
                    trainer.load_state_dict(snap)
                    captured = True
                pred = trainer.step_graphed(noisy, timesteps, ehs, guide, target)
                stepped = True
            else:
                pred = trainer.forward_backward(noisy, timesteps, ehs, guide, target)
                stepped = trainer.optimizer_step()
            if not stepped:
                continue
            global_step += 1
            imgs_last += args.train_batch_size * world * args.gradient_accumulation_steps
            if main_proc and (global_step % 10 == 0 or global_step == args.max_train_steps or global_step <= 3):
                loss = trainer.loss(pred.numel())               # host sync: only on logging steps
                now = time.perf_counter()
                rec = {"step": global_step, "epoch": epoch, "step_loss": loss, "lr": args.learning_rate * (float(trainer.state[10]) or 1.0),
                       "loss_scale": float(trainer.state[3]), "grad_norm": float(trainer.state[9]),
                       "images_per_s": imgs_last / (now - t_last)}
                t_last, imgs_last = now, 0
                logger.info(json.dumps(rec))
                with open(log_path, "a") as f:
                    f.write(json.dumps(rec) + "\n")
            if global_step % args.checkpointing_steps == 0 and main_proc:
                save_path = os.path.join(args.output_dir, f"checkpoint-{global_step}")
                trainer.save_state(save_path)
                logger.info("Saved state to %s", save_path)
                if args.validation_prompt is not None:
                    run_validation(args, unet, control_lora, vae, text_encoder, tokenizer, dataset, dev, global_step)
            if global_step >= args.max_train_steps:
                break
        if global_step >= args.max_train_steps:
            break

    if world > 1:
        torch.distributed.barrier()
    if main_proc:
        control_lora.save_config(args.output_dir)
        control_lora.save_pretrained(args.output_dir, safe_serialization=False)
        control_lora.save_pretrained(args.output_dir, safe_serialization=True)
        logger.info("Saved ControlLoRA to %s", args.output_dir)
    if world > 1:
        torch.distributed.destroy_process_group()
    return global_step


@torch.no_grad()
def run_validation(args, unet, control_lora, vae, text_encoder, tokenizer, dataset, dev, global_step):
    """sampling with the current adapters (reference :811-860): DPM-Solver++(2M) like the reference's
    `DPMSolverMultistepScheduler.from_config(...)` (:824), 30 steps (:842), the pipeline's default guidance 7.5,
    [sample | target | guide] strips (`dataset_cls.cat_input`, :843)"""
    from PIL import Image
    import numpy as np
    from controllora_amd.pipeline import ddim_sample
    gen = torch.Generator(device=dev).manual_seed(args.seed or 0)
    out_dir = os.path.join(args.output_dir, "validation")
    os.makedirs(out_dir, exist_ok=True)
    cond = text_encoder(tokenizer([args.validation_prompt]).to(dev))[0].half()
    uncond = text_encoder(tokenizer([""]).to(dev))[0].half()
    for i in range(args.num_validation_images):
        ex = dataset[i % len(dataset)]
        guide = ex["guide_values"][None].to(dev).half()
        lat = ddim_sample(unet, control_lora, guide, cond, uncond, steps=30, guidance_scale=7.5, generator=gen, sampler="dpm")
        img = vae.decode(lat.half() / vae.scaling_factor).sample.float().clamp(-1, 1)
        strip = torch.cat([img[0].cpu(), ex["pixel_values"], ex["guide_values"]], dim=2)
        arr = ((strip.permute(1, 2, 0).numpy() + 1.0) * 127.5).round().astype(np.uint8)
        Image.fromarray(arr).save(os.path.join(out_dir, f"step{global_step:06d}_{i}.png"))


if __name__ == "__main__":
    main()
