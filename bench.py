"""bench.py -- headline benchmark of the ControlLoRA hot path on MI355X (contract in the task brief).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one full training step of the hot path on a synthetic fill50k-like batch resident in HBM:
hint-encoder forward, SD-1.5 UNet forward with the 32 ControlLoRA processors, fp32 MSE, backward (dgrad
through the frozen fp16 UNet, wgrad for adapters + hint encoder), [RCCL all-reduce of the flat 24 MB
gradient], fused clip + AdamW.  Workload = BASELINE.json configs[1]: configs/fill50k.json, SD-1.5 topology
with seeded random weights (no checkpoint offline), 512x512, batch 4 per GPU, fp16.  VAE-encode / CLIP are
outside the named hot path: latents and text embeddings are synthetic inputs.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under `torch.distributed.run` with N
ranks (one per GPU, RCCL); launched BY `torch.distributed.run` (the driver's way) it just joins.  `n_gpus` in the
line is the number of ranks that actually joined (`torch.distributed.get_world_size()`).

Prints ONE JSON line (rank 0) with metric/value plus `roofline` (dominant kernel family: algorithmic flops of its
launches in one step / its kernel time per GRAPH-REPLAYED step, taken from a `rocprofv3 --kernel-trace` of this same
command run as a child process; HIP-event timings of an eager step are reported beside it) and `cpu_baseline` (the
CPU oracle, N=1 only).
"""
from __future__ import annotations

import argparse
import glob
import json
import math
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def hot_path_tflop_per_image(res):
    """algorithmic work of one trained image (roofline.py: shapes only, 2 x MACs; SURVEY.md section 8d: 1.81 @512^2)"""
    import roofline as R
    cfg = json.load(open(os.path.join(ROOT, "configs", "fill50k.json")))
    return R.train_step_flops_per_image(res, cfg, 1.51 * (res / 512) ** 2)["total"] / 1e12
MFMA_PEAK_TFLOPS = 2500.0                # dense fp16, MI355X_MICROARCH.md
LEG_SECONDS = {}                         # wall time of each leg of the run (printed on stderr)
# measured ceilings of this chip (stand-alone probes, profiles/r02_{mfma_rate,stream_rate}_probe.txt): reported beside `frac`,
# never instead of it
# (the stand-alone MFMA probe of round 2 -- 18.3 ns per 32x32x16 instruction = the guide's 32 cycles at ~1.75 GHz -- measured the clock the
# chip held under that loop, not a lower matrix-pipe ceiling: `frac` is against the guide's 2.5 PFLOP/s only)
FABRIC_TBS = 6.3                         # L2 <- MALL / HBM stream, all XCDs
HBM_PEAK_GBS = 8000.0


def synthetic_batch(B, res, dev, seed):
    """fill50k-like: filled disc -> latents stand-in ~N(0,1); guide = disc outline (+1 on -1), 3 channels."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    L = res // 8
    yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
    guide = torch.empty(B, 3, res, res)
    for b in range(B):
        cx, cy = torch.randint(res // 4, 3 * res // 4, (2,), generator=g)
        r = int(torch.randint(res // 16, res // 4, (1,), generator=g))
        d = ((xx - cx) ** 2 + (yy - cy) ** 2).float().sqrt()
        guide[b] = ((d - r).abs() < 2).float()[None] * 2 - 1
    return dict(
        guide=guide.to(dev).half(),
        latents=torch.randn(B, 4, L, L, generator=g).to(dev).half(),
        noise=torch.randn(B, 4, L, L, generator=g).to(dev).half(),
        timesteps=torch.randint(0, 1000, (B,), generator=g).to(dev),
        ehs=torch.randn(B, 77, 768, generator=g).to(dev).half(),
    )


def build_models(dev, seed=0, config="fill50k.json"):
    from controllora_amd import models as M, unet as U
    unet = U.UNet2DConditionModel()
    unet.to(dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if p.ndim >= 2:
                p.normal_(0.0, 1.0 / math.sqrt(p[0].numel()), generator=g)
            elif name.endswith("bias"):
                p.zero_()
            else:
                p.fill_(1.0)
    torch.manual_seed(seed)
    clora = M.ControlLoRA.from_config(os.path.join(ROOT, "configs", config)).to(dev)
    unet.set_attn_processor(M.map_processors_to_unet(unet, clora))
    return unet, clora


def _pick_cpu_threads(unet, b):
    """The oracle is timed at the thread count that serves it best on this host: torch's default (all hardware threads)
    loses to a smaller pool on many-core hosts (oversubscribed SMT / NUMA).  One UNet forward per candidate, ~2-3 s each."""
    import os as _os
    hw = _os.cpu_count() or 1
    cands = sorted({c for c in (hw // 2, hw // 4, hw // 8, 32, 16, min(hw, 8)) if 1 <= c <= hw}, reverse=True)   # all `hw` threads lost 5x on the GPU hosts
    best, best_t = torch.get_num_threads(), None
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            unet(b["latents"].float(), b["timesteps"], b["ehs"].float())        # page in / thread pool spin-up
            t0 = time.perf_counter()
            unet(b["latents"].float(), b["timesteps"], b["ehs"].float())
            dt = time.perf_counter() - t0
            if best_t is None or dt < best_t:
                best, best_t = c, dt
    torch.set_num_threads(best)
    return best


def _cpu_limits():
    """what the container may actually use of the host's hardware threads: the scheduler affinity mask and the cgroup CPU quota
    (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`) -- a quota of e.g. 8 CPUs on a 256-thread host is why the
    oracle is fastest on 8 threads there"""
    out = {"hardware_threads": os.cpu_count()}
    try:
        out["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        except (OSError, ValueError):
            pass
    out["cgroup_cpu_quota"] = round(quota, 2) if quota is not None else "none"
    return out


def cpu_baseline(steps=3, full=False):
    """The CPU oracle (pure-torch fp32 restatement of the reference path, materialised attention) on BASELINE
    configs[0]: fill50k.json, SD-1.5, 256x256, batch 1.  Bounded sample: 1 warm-up + `steps` timed steps (SURVEY.md
    section 8d asks for >= 3) plus one timed 512x512 bs1 step (train_512_bs1).  full=True (--cpu-baseline-full) adds the oracle's
    DDIM + CFG UNet evaluation (batch 2) at 512x512: minutes of CPU work, reported in extra keys."""
    from oracle import cases, unet_ref
    from oracle.controllora_ref import ControlLoRARef, map_processors_to_unet
    torch.manual_seed(0)
    unet = unet_ref.UNet2DConditionModel()
    unet_ref.init_unet_weights_(unet, seed=0)
    clora = ControlLoRARef.from_config(os.path.join(ROOT, "configs", "fill50k.json"))
    unet.set_attn_processor(map_processors_to_unet(unet, clora))
    opt = torch.optim.AdamW(clora.parameters(), lr=1e-4, weight_decay=1e-2)

    def batch_of(res):
        b = synthetic_batch(1, res, "cpu", 42)
        return dict(guide=b["guide"].float(), latents=b["latents"].float(), noise=b["noise"].float(),
                    timesteps=b["timesteps"], ehs=b["ehs"].float())

    inp = batch_of(256)
    with torch.no_grad():
        clora(inp["guide"])
    threads = _pick_cpu_threads(unet, inp)

    def timed_steps(inp, n):
        times = []
        for i in range(n + 1):
            t0 = time.perf_counter()
            cases.oracle_train_step(unet, clora, clora, inp)
            torch.nn.utils.clip_grad_norm_(clora.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
            if i > 0:
                times.append(time.perf_counter() - t0)
        return sum(times) / len(times)

    sec = timed_steps(inp, steps)
    out = {"value": round(1.0 / sec, 4), "unit": "images/s", "cores": threads, "kind": "port",
           "sample": f"oracle (pure-torch fp32, materialised attention) train step, fill50k.json SD-1.5 256x256 bs1, "
                     f"{steps} steps after 1 warm-up, {sec:.2f} s/step, {threads} of {os.cpu_count()} hardware threads "
                     f"(fastest of the candidates tried)"}
    out["host_cpu"] = _cpu_limits()
    # the like-for-like resolution ("512x512 too, rather than extrapolating"): one timed 512x512 bs 1 step after
    # one warm-up step, ~10 s of CPU work, part of the default line since round 6
    inp512 = batch_of(512)
    sec512 = timed_steps(inp512, 1)
    out["train_512_bs1"] = {"s_per_step": round(sec512, 2), "images_per_s": round(1.0 / sec512, 4), "steps": 1,
                            "what": "the same oracle step at BASELINE configs[1]'s resolution, batch 1 (1 timed step after 1 warm-up)"}
    if full:
        with torch.no_grad():
            clora(inp512["guide"])
            x = torch.cat([inp512["latents"]] * 2)
            ehs = torch.cat([inp512["ehs"]] * 2)
            unet(x, 981, ehs)
            t0 = time.perf_counter()
            for t in (961, 941):
                unet(x, t, ehs)
            dt = (time.perf_counter() - t0) / 2
        out["ddim_512_1image"] = {"s_per_step": round(dt, 2), "extrapolated_50_steps_s": round(50 * dt, 1),
                                  "what": "oracle UNet evaluation of one DDIM + CFG step (UNet batch 2), 512x512, 2 timed steps"}
    return out


def box_calibration(dev):
    """What THIS box can do, measured right before the timed windows, so that lines from different boxes can be normalised
    (the driver's box ran every kernel 5-28 % slower than the builder's and nothing in the line said why):
      * gemm8192_cfg1_us -- the 8192^3 fp16 GEMM on tile_cfg 1 (128x128 BK32: a kernel whose code has not changed since round 1;
        tools/box_calib.py and every profiles/r0N_box_calib*.txt use the same one) under tile_order "auto", random operands,
        best of 3 pairs of launches; gemm8192_cfg59_* -- the same GEMM on the eight-phase tile (gemm_8p_kernel)
      * copy256MB_GBps -- a 256 MB device copy (read + write bytes)
      * mfma_clock_MHz / mfma_probe_TFLOPs -- clora_clock_probe: shader cycles vs the 100 MHz wall clock around a dense MFMA stream on
        pseudo-random operands: the clock the chip holds under matrix load (DVFS), and the dense rate that goes with it
      * sclk / power strings as the driver exposes them (sysfs), when readable"""
    import ctypes
    from controllora_amd import capi, kernels as K
    out = {}
    try:
        g = torch.Generator(device=dev).manual_seed(7)
        n = 8192
        A = (torch.rand(n, n, device=dev, generator=g) - 0.5).half()
        B = (torch.rand(n, n, device=dev, generator=g) - 0.5).half()
        C_ = torch.empty(n, n, device=dev, dtype=torch.float16)
        run = lambda: K.gemm(A, B, n, n, n, out=C_, split_k=1, tile_cfg=1, _tuned=False)
        K.set_tile_order("auto")               # the order every profiles/r0N_box_calib*.txt was taken under (the library's default, "grid",
        try:                                   # runs this very GEMM 6 % faster: 1515 vs 1615 us on one box, profiles/r05_ab_lib_382a909_vs_head.txt)
            run(); torch.cuda.synchronize()
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 2
                best = us if best is None else min(best, us)
        finally:
            # back to the order in force before the calibration: an A/B run started with CLORA_TILE_ORDER=auto|m|n (forwarded to the
            # library at load, capi._ENV_OPTIONS) must keep that order for the timed windows
            env_order = {"a": "auto", "g": "grid"}.get(os.environ.get("CLORA_TILE_ORDER", ""), os.environ.get("CLORA_TILE_ORDER", ""))
            K.set_tile_order(env_order if env_order in K.TILE_ORDERS else K.DEFAULT_TILE_ORDER)
        out["gemm8192_cfg1_us"] = round(best, 1)
        out["gemm8192_cfg1_TFLOPs"] = round(2 * n ** 3 / best / 1e6, 1)
        # the same GEMM on the eight-phase 256x256 tile (tile_cfg 59, default tile order): what the GEMM core reaches on this box
        run59 = lambda: K.gemm(A, B, n, n, n, out=C_, split_k=1, tile_cfg=59, _tuned=False)
        run59(); torch.cuda.synchronize()
        best59 = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run59(); run59(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 2
            best59 = us if best59 is None else min(best59, us)
        out["gemm8192_cfg59_us"] = round(best59, 1)
        out["gemm8192_cfg59_TFLOPs"] = round(2 * n ** 3 / best59 / 1e6, 1)
        del A, B, C_
        x = torch.empty(128 << 20, dtype=torch.float16, device=dev)
        y = torch.empty_like(x)
        y.copy_(x); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y.copy_(x)
        e1.record(); torch.cuda.synchronize()
        out["copy256MB_GBps"] = round(2 * x.numel() * 2 / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e9, 0)
        del x, y
        blocks, iters = 1024, 20000                       # 4 blocks per CU (one wave per SIMD each), ~3 ms of MFMA issue
        buf = torch.zeros(3 * blocks, dtype=torch.int64, device=dev)
        L = capi.lib()
        L.call("clora_clock_probe", capi.ptr(buf), blocks, 2000, capi.stream())          # warm the clocks
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("clora_clock_probe", capi.ptr(buf), blocks, iters, capi.stream())
        e1.record(); torch.cuda.synchronize()
        v = buf.view(blocks, 3).double().cpu()
        cyc, tick = v[:, 0].median().item(), v[:, 1].median().item()
        if tick > 0:
            out["mfma_clock_MHz"] = round(cyc / tick * 100.0, 0)
        out["mfma_probe_TFLOPs"] = round(blocks * 4 * iters * 8 * 2 * 16 * 16 * 32 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 0)
    except Exception as e:                                   # noqa: BLE001 -- calibration must never take the bench line down
        out["error"] = repr(e)[:200]
    try:
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            f = os.path.join(card, "pp_dpm_sclk")
            if os.path.exists(f):
                cur = [ln.strip() for ln in open(f) if "*" in ln]
                out.setdefault("sysfs", {})["sclk_now"] = cur[0] if cur else None
                for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
                    for key in ("power1_cap", "power1_average", "power1_input"):
                        q = os.path.join(hw, key)
                        if os.path.exists(q):
                            out["sysfs"][key + "_W"] = round(int(open(q).read().strip()) / 1e6, 1)
                break
    except Exception:                                        # noqa: BLE001
        pass
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` (no launcher): re-exec under torch.distributed.run, one rank per GPU (driver contract)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """Exercise the N>1 control flow without GPUs (`--backend gloo --dry-run`): rendezvous, the flat-buffer all-reduce
    (sum of ranks, divisor folded into the optimizer), barrier + MAX-over-ranks timing, rank-0 JSON line."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(args.backend)
        world = torch.distributed.get_world_size()
    n = 6047040                                   # fill50k.json trainable parameters = the all-reduce payload
    g = torch.full((n,), float(rank + 1))
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            torch.distributed.all_reduce(g)
            g.div_(world)
    if world > 1:
        torch.distributed.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    comm_fallback = None
    if args.comm == "clora":
        comm_fallback = f"--comm clora requested but backend {args.backend!r} / --dry-run has no RCCL behind it: flat all-reduce through torch.distributed"
        if rank == 0:
            print("[bench] WARNING: " + comm_fallback, file=sys.stderr, flush=True)
    ok = True
    if world > 1 and args.steps > 0:
        # after one all-reduce(sum)/N every rank holds mean(1..N); further rounds keep it there
        ok = bool(torch.allclose(g, torch.full_like(g, (world + 1) / 2.0)))
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work): flat-buffer all-reduce control flow", "dry_run": True, "value": None,
                          "n_gpus": world, "rccl_ranks": world, "backend": args.backend, "steps": args.steps, "warmup": args.warmup,
                          "comm": "torch", "comm_requested": args.comm, "comm_fallback": comm_fallback,
                          "ms_per_step": round(float(el) / max(1, args.steps) * 1e3, 3), "allreduce_bytes": n * 4, "allreduce_ok": ok}))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if ok else 1


GEMM_FAMILY = ("gemm_dma_kernel", "conv3x3_patch_kernel", "splitk_finish_kernel", "gemm_")       # kernels behind clora_gemm_f16_ex


def rocprof_child_trace(args, steps=6, warmup=2):
    """`rocprofv3 --kernel-trace` over THIS command (same config / batch / res, hipGraph replay, nothing but the train
    steps) as a child process; returns per-kernel time per step or None when rocprofv3 is unavailable / fails."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="clora_kt_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "-d", out, "-o", "kt", "--", sys.executable, os.path.abspath(__file__), "--trace-child",
           "--steps", str(steps), "--warmup", str(warmup), "--batch", str(args.batch), "--res", str(args.res), "--config", args.config]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    except Exception as e:                                   # noqa: BLE001
        return {"error": repr(e)}
    dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
    if r.returncode != 0 or not dbs:
        return {"error": f"rocprofv3 rc={r.returncode}", "tail": r.stdout.decode(errors="replace")[-400:]}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rocprof_summary
    rows = rocprof_summary.load(dbs[0])
    # the child runs 2 eager warm-up steps inside capture() + `warmup` + `steps` replays: all issue the same kernels.  Kernels
    # whose launch count is not a multiple of the step count belong to the set-up (weight initialisation, operand packing, the
    # flat parameter buffer): they are reported apart instead of being averaged into the step.
    n_steps = 2 + warmup + steps
    step_rows = [r_ for r_ in rows if r_["calls"] % n_steps == 0]
    setup_rows = [r_ for r_ in rows if r_["calls"] % n_steps != 0]
    res = {"steps_in_trace": n_steps, "total_kernel_ms_per_step": sum(r_["total_ms"] for r_ in step_rows) / n_steps,
           "launches_per_step": round(sum(r_["calls"] for r_ in step_rows) / n_steps),
           "setup_kernel_ms_total": round(sum(r_["total_ms"] for r_ in setup_rows), 3),
           "kernels": step_rows, "db": dbs[0]}
    shutil.rmtree(out, ignore_errors=True)
    return res


def rocprof_child_pmc(args, counter, steps=1, warmup=1):
    """One `rocprofv3 --pmc <counter>` pass over THIS command issued eagerly (counter collection serialises the kernels; a
    hipGraph replay is not instrumented per node), --kernel-trace / --stats only as gpurun requires: per-kernel
    [(grid, bytes)] lists, or {"error": ...}.  FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots, MI355X_MICROARCH.md)."""
    import re
    import sqlite3
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = tempfile.mkdtemp(prefix="clora_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--trace-child", "--no-graph",
           "--steps", str(steps), "--warmup", str(warmup), "--batch", str(args.batch), "--res", str(args.res), "--config", args.config]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return {"error": f"rocprofv3 --pmc {counter} rc={r.returncode}", "tail": r.stdout.decode(errors="replace")[-300:]}
        res = {}
        cur = sqlite3.connect(dbs[0]).cursor()
        for name, grid, val in cur.execute("select kernel_name, grid_size, value from counters_collection"):
            res.setdefault(re.sub(r"\(anonymous namespace\)::", "", name), []).append((grid, val * 1024.0))   # counters are in KiB
        return res
    except Exception as e:                                   # noqa: BLE001
        return {"error": repr(e)}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measured_traffic(args, family):
    """HBM-side (L2 <-> fabric) bytes per launch of the kernel family, measured by THIS run: two PMC child passes.
    gfx950 correction as the guide prescribes (FETCH_SIZE counts 128-byte requests at 64 B: x2), checked in place on a
    kernel whose read volume is known exactly (the hint encoder's first GroupNorm statistics pass reads B*H*W*32 fp16
    values once)."""
    F = rocprof_child_pmc(args, "FETCH_SIZE")
    if "error" in F:
        return None, F
    W = rocprof_child_pmc(args, "WRITE_SIZE")
    if "error" in W:
        return None, W
    known = args.batch * args.res * args.res * 32 * 2
    cal = max((v for k, vs in F.items() if "gn_fwd_partial" in k for _, v in vs), default=0.0)
    factor = known / cal if cal > 0 else 2.0
    if not 1.5 < factor < 2.5:
        factor = 2.0                                         # calibration kernel changed shape: fall back to the guide's x2
    fb = sum(v for k, vs in F.items() if any(t in k for t in family) for _, v in vs)
    wb = sum(v for k, vs in W.items() if any(t in k for t in family) for _, v in vs)
    n = sum(len(vs) for k, vs in F.items() if any(t in k for t in family))
    if n == 0:
        return None, {"error": "no kernel of the family in the PMC pass"}
    return round((fb * factor + wb) / n), {
        "static": False, "how": "two rocprofv3 --pmc child passes of this command (FETCH_SIZE, WRITE_SIZE; eager step, "
                                "2 steps each), L2<->fabric bytes per launch = FETCH_SIZE x correction + WRITE_SIZE",
        "fetch_correction": round(factor, 3), "calibration": "hint-encoder gn_fwd_partial reads B*H*W*32*2 bytes exactly once",
        "launches_in_pass": n, "fetch_bytes_per_launch": round(fb * factor / n), "write_bytes_per_launch": round(wb / n)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch (BASELINE: 4)")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-ddim", action="store_true", help="skip the secondary inference measurement")
    ap.add_argument("--no-graph", action="store_true", help="issue the step eagerly instead of replaying hipGraphs")
    ap.add_argument("--ddim-batch", type=int, default=16)
    ap.add_argument("--mix-lora", default="none", choices=["none", "pre", "post", "both"],
                    help="after the DDIM line: inject a plain rank-4 LoRACrossAttnProcessor into every control processor as pre / post "
                         "LoRA (the arrangement of the reference's mix_lora_and_control_lora.py:109-123) and time the same sampler again")
    ap.add_argument("--config", default="fill50k.json", help="ControlLoRA config under configs/ (BASELINE configs[1] = fill50k.json; "
                    "mpii-pose-v2.json with --batch 8 is BASELINE configs[3])")
    ap.add_argument("--no-full-step", action="store_true", help="skip the secondary 'whole reference step' line (VAE + CLIP inside)")
    ap.add_argument("--backend", default="nccl", help='torch.distributed backend ("nccl" is RCCL on ROCm; "gloo" only for '
                    "exercising the N>1 control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--comm", default=os.environ.get("CLORA_COMM", "auto"), choices=["auto", "torch", "clora"],
                    help='exchange path of the data-parallel step: "torch" = torch.distributed.all_reduce on the process group, '
                         '"clora" = the C ABI\'s own RCCL communicator (clora_comm_init / clora_allreduce_flat_f32); same '
                         'ncclAllReduce either way; "auto" (default) = clora when world > 1 on an nccl group, a requested clora that '
                         'cannot be set up falls back to torch with a warning and `comm_fallback` in the line (also: CLORA_COMM)')
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps steps each; the line reports the median window")
    ap.add_argument("--no-calibration", action="store_true", help="skip the box calibration (8192^3 GEMM, 256 MB copy, MFMA clock probe)")
    ap.add_argument("--dry-run", action="store_true", help="with --backend gloo: exercise the multi-rank control flow (spawn, "
                    "rendezvous, flat all-reduce, rank-0 line) without touching a GPU")
    ap.add_argument("--trace-child", action="store_true", help=argparse.SUPPRESS)    # run by rocprof_child_trace()
    ap.add_argument("--no-rocprof", action="store_true", help="skip the rocprofv3 child run (roofline falls back to HIP events)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="also time the CPU oracle at 512x512 bs1 and its DDIM "
                    "UNet evaluation (SURVEY.md section 8d; minutes of CPU work, not part of the default run)")
    ap.add_argument("--trace-out", default=None, help="write the child run's full per-kernel table (json) here, e.g. profiles/r02_kernel_stats.json")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    if args.dry_run:
        sys.exit(dry_run(args))
    if args.trace_child:
        args.no_cpu_baseline = args.no_roofline = args.no_ddim = args.no_full_step = True
        args.windows = 1                       # the parent divides the child's launch counts by 2 + warmup + steps

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm
        else:
            torch.distributed.init_process_group(args.backend)
        pg = torch.distributed.group.WORLD
        world = torch.distributed.get_world_size()       # the ranks that actually joined

    from controllora_amd import kernels as K
    from controllora_amd.schedulers import DDPMScheduler
    from controllora_amd.train import ControlLoRATrainer

    unet, clora = build_models(dev, config=args.config)
    trainer = ControlLoRATrainer(unet, clora, process_group=pg, world_size=world, comm=args.comm)   # falls back loudly (comm_fallback)
    batch = synthetic_batch(args.batch, args.res, dev, 42 + rank)         # data-parallel: different samples per rank
    noisy = DDPMScheduler().add_noise(batch["latents"], batch["noise"], batch["timesteps"]).half()

    def eager_step():
        trainer.step(noisy, batch["timesteps"], batch["ehs"], batch["guide"], batch["noise"])

    graphed = not args.no_graph
    if graphed:
        trainer.capture(noisy, batch["timesteps"], batch["ehs"], batch["guide"], batch["noise"])
        step = trainer.step_graphed          # same kernels, same work: the launches are replayed from hipGraphs
    else:
        step = eager_step

    calib = box_calibration(dev) if (rank == 0 and world == 1 and not args.trace_child and not args.no_calibration) else None

    for _ in range(args.warmup):
        step()
    # Timed region: `--windows` (default 3) back-to-back windows of EXACTLY --steps steps, each bracketed by barrier +
    # synchronize on both sides and reduced with MAX over ranks as the contract says; the line reports the MEDIAN window as
    # value / ms_per_step and every window beside it, so a reader can see the spread (clock ramps, a noisy neighbour).
    windows = []
    for _ in range(max(1, args.windows)):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t)
        windows.append(el)
    elapsed = sorted(windows)[len(windows) // 2]
    window_ms = [round(w / args.steps * 1e3, 3) for w in windows]
    ms = elapsed / args.steps * 1e3
    images_per_s = args.batch * world / (ms * 1e-3)
    loss = trainer.loss(noisy.numel())
    skipped = float(trainer.state[6])

    allreduce_ms = None
    if world > 1 or trainer.comm == "clora":      # the exchange step on its own: 10 flat-buffer all-reduces, HIP events (with --comm clora also
        #                                            at one rank: the C ABI's communicator / all-reduce call path on hardware, no peer traffic)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        keep = trainer.flat.grad.clone()           # through the trainer's own exchange path (torch process group or the C ABI)
        trainer._all_reduce_grads()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            trainer._all_reduce_grads()
        e1.record()
        torch.cuda.synchronize()
        allreduce_ms = round(e0.elapsed_time(e1) / 10, 4)
        trainer.flat.grad.copy_(keep)
        del keep

    if args.trace_child:
        return

    # secondary line of BASELINE.json's metric: 50-step DDIM latency, 512^2, 16 images, CFG 9.0 (UNet batch 32),
    # control batch 1 (the inference call pattern of apps/gradio_canny2image.py:66-92); replicas only, rank 0, N=1
    _t_leg = time.perf_counter()
    ddim = None
    if world == 1 and rank == 0 and not args.no_ddim:
        from controllora_amd.pipeline import ddim_sample
        nb = args.ddim_batch
        g = torch.Generator(device=dev).manual_seed(1)
        cond = torch.randn(nb, 77, 768, device=dev, generator=g).half()
        uncond = torch.randn(nb, 77, 768, device=dev, generator=g).half()
        lat0 = torch.randn(nb, 4, args.res // 8, args.res // 8, device=dev, generator=g).half()
        ddim_sample(unet, clora, batch["guide"][:1], cond, uncond, steps=2, latents=lat0)      # warm-up
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = ddim_sample(unet, clora, batch["guide"][:1], cond, uncond, steps=50, guidance_scale=9.0, latents=lat0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        ddim = {"metric": f"50-step DDIM latency {args.res}^2 bs{nb} (CFG 9.0, UNet batch {2 * nb}, control batch 1)",
                "latency_s": round(dt, 3), "images_per_s": round(nb / dt, 3), "finite": bool(torch.isfinite(out.float()).all())}
        if args.mix_lora != "none":
            # reference mix_lora_and_control_lora.py:109-123: one more LoRA per attention site, chained before / after the control
            # adapter (models.py:232-243, 249-265, 276-282); the product runs chained sites on its generic (unfused) path
            from controllora_amd import models as M
            torch.manual_seed(11)
            n_sites = 0
            for procs in clora.lora_layers:
                for proc in procs:
                    other = M.LoRACrossAttnProcessor(proc.hidden_size, proc.cross_attention_dim, rank=4).to(dev)
                    for q in other.parameters():
                        q.requires_grad_(False)
                        torch.nn.init.normal_(q, std=0.02)        # a trained LoRA: non-zero up matrices
                    if args.mix_lora in ("pre", "both"):
                        proc.inject_pre_lora(other)
                    if args.mix_lora in ("post", "both"):
                        proc.inject_post_lora(other)
                    n_sites += 1
            ddim_sample(unet, clora, batch["guide"][:1], cond, uncond, steps=2, latents=lat0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out2 = ddim_sample(unet, clora, batch["guide"][:1], cond, uncond, steps=50, guidance_scale=9.0, latents=lat0)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            ddim["mix_lora"] = {"arrangement": args.mix_lora, "sites": n_sites, "latency_s": round(dt2, 3),
                                "images_per_s": round(nb / dt2, 3), "finite": bool(torch.isfinite(out2.float()).all()),
                                "differs_from_unmixed": bool((out2.float() - out.float()).abs().max() > 0)}
            for procs in clora.lora_layers:
                for proc in procs:
                    proc.pre_loras.clear(); proc.post_loras.clear()
            del out2
        del out, cond, uncond, lat0
        torch.cuda.empty_cache()

    # informational third line: the WHOLE reference step (train...:751-796) -- VAE encode x0.18215, noise / timesteps /
    # add_noise, CLIP text encode (stock transformers model: frozen glue outside the hot path), then the captured hot
    # path -- so the cost of what SURVEY section 8f ranks "next" is visible beside the headline number.  N=1, rank 0.
    LEG_SECONDS['ddim50'] = time.perf_counter() - _t_leg; _t_leg = time.perf_counter()
    full = None
    if world == 1 and rank == 0 and graphed and not args.no_full_step:
        from controllora_amd import loading, text
        vae = loading.load_vae("random:sd15", dev)
        enc = text.load_text_encoder("random:sd15", dev)
        gpix = torch.Generator(device=dev).manual_seed(3)
        pixel = (torch.rand(args.batch, 3, args.res, args.res, device=dev, generator=gpix) * 2 - 1).half()
        ids = torch.randint(0, 49408, (args.batch, 77), device=dev, generator=gpix)

        def full_step():
            with torch.no_grad():
                lat = vae.encode(pixel).latent_dist.sample() * vae.scaling_factor
                nz = torch.randn_like(lat)
                ts = torch.randint(0, 1000, (args.batch,), device=dev)
                ny = DDPMScheduler().add_noise(lat, nz, ts).half()
                ehs_ = enc(ids)[0].half()
            trainer.step_graphed(ny, ts, ehs_, batch["guide"], nz)

        for _ in range(2):
            full_step()
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        nfull = max(5, args.steps // 2)
        for _ in range(nfull):
            full_step()
        torch.cuda.synchronize()
        fms = (time.perf_counter() - tf0) / nfull * 1e3
        full = {"what": "VAE encode + CLIP text encode + noise/add_noise + hot path (random-init SD-1.5-shaped VAE / CLIP)",
                "ms_per_step": round(fms, 3), "images_per_s": round(args.batch / fms * 1e3, 3), "steps": nfull}
        del vae, enc
        torch.cuda.empty_cache()

    # (the roofline leg runs AFTER the secondary timings: its rocprofv3 --pmc child passes were followed by a 10 % slower DDIM line
    # on the same box -- counter collection leaves the device in a profiling power state for a while)
    LEG_SECONDS['full_step'] = time.perf_counter() - _t_leg; _t_leg = time.perf_counter()
    roofline = None
    if not args.no_roofline and rank != 0:
        eager_step()                           # the warm-up and the profiled step contain the all-reduce: every rank takes part in both
        eager_step()
    if not args.no_roofline and rank == 0:
        # (1) algorithmic flops / bytes per launch family + HIP-event durations of ONE eager step (events on the launch stream)
        eager_step()                           # untimed: the secondary legs above emptied the allocator cache, and the first eager step
        torch.cuda.synchronize()               # after that pays hipMalloc stalls inside whatever launch follows them
        K.PROFILER = K.KernelProfiler()
        eager_step()
        agg = K.PROFILER.summary()
        K.PROFILER = None
        ev_total_ms = sum(a["ms"] for a in agg.values())
        # the roofline object is about the GEMM / conv family (clora_gemm_f16_ex: 5.67 of the step's 7.24 algorithmic TFLOP): named, not
        # picked by event time -- a one-off stall in a small family must not redirect it
        dom_name = "clora_gemm_f16_ex" if "clora_gemm_f16_ex" in agg else max(agg.items(), key=lambda kv: kv[1]["ms"])[0]
        dom = agg[dom_name]
        fam = {k: {"calls": v["calls"], "event_ms": round(v["ms"], 3),
                   "event_TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None}
               for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:8]}
        ev_ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        # (2) the same family's kernel time per GRAPH-REPLAYED step: rocprofv3 --kernel-trace of this command as a child
        # process (N=1 only).  The bench line's roofline uses THIS duration; the event figures are kept beside it.
        trace = None
        if world == 1 and not args.no_rocprof and graphed:
            trace = rocprof_child_trace(args)
        src, fam_ms, fam_launches, kt = "hip_events(eager step)", dom["ms"], dom["calls"], None
        if trace and "kernels" in trace and dom_name.startswith("clora_gemm"):
            rows = [r_ for r_ in trace["kernels"] if any(t in r_["kernel"] for t in GEMM_FAMILY)]
            n_st = trace["steps_in_trace"]
            fam_ms = sum(r_["total_ms"] for r_ in rows) / n_st
            fam_launches = round(sum(r_["calls"] for r_ in rows) / n_st)
            src = f"rocprofv3 --kernel-trace child run ({n_st} steps, hipGraph replay), kernels matching {GEMM_FAMILY}"
            if args.trace_out:
                json.dump({"command": "rocprofv3 --kernel-trace -- python bench.py --trace-child ...", "steps_in_trace": n_st,
                           "total_kernel_ms_per_step": trace["total_kernel_ms_per_step"], "launches_per_step": trace["launches_per_step"],
                           "setup_kernel_ms_total": trace["setup_kernel_ms_total"],
                           "kernels": [dict(r_, ms_per_step=round(r_["total_ms"] / n_st, 4), calls_per_step=round(r_["calls"] / n_st, 2))
                                       for r_ in trace["kernels"]]}, open(args.trace_out, "w"), indent=1)
            kt = {"total_kernel_ms_per_step": round(trace["total_kernel_ms_per_step"], 3),
                  "launches_per_step": trace["launches_per_step"], "setup_kernel_ms_total": trace["setup_kernel_ms_total"],
                  "top": [{"kernel": r_["kernel"][:60], "calls_per_step": round(r_["calls"] / n_st, 1),
                           "ms_per_step": round(r_["total_ms"] / n_st, 3), "avg_us": r_["avg_us"]} for r_ in trace["kernels"][:12]]}
        elif trace:
            kt = {"error": trace.get("error"), "tail": trace.get("tail")}
        ach = dom["flops"] / (fam_ms * 1e-3) / 1e12
        # HBM-side traffic per launch of the dominant family: PMC counters need their own rocprofv3 --pmc passes
        # (tools/pmc_traffic.sh -> tools/pmc_summary.py); the committed result of the latest passes is reported with
        # "static": true and the commit / round it was measured at -- it does not move with this run.
        # measured by THIS run (two --pmc child passes) or null -- never read from a committed file
        traffic, traffic_src = None, None
        if dom_name.startswith("clora_gemm") and world == 1 and not args.no_rocprof and not args.no_pmc:
            traffic, traffic_src = measured_traffic(args, GEMM_FAMILY)
        roofline = {"bound": "mfma", "kernel": dom_name, "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "duration_source": src,
                    "family_ms_per_step": round(fam_ms, 3), "launches": fam_launches,
                    "avg_launch_us": round(fam_ms * 1e3 / max(1, fam_launches), 1),
                    "algorithmic_TFLOP_per_step": round(dom["flops"] / 1e12, 3),
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["calls"]),
                    "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                    "event_timed": {"achieved": round(ev_ach, 1), "frac": round(ev_ach / MFMA_PEAK_TFLOPS, 4),
                                    "family_ms": round(dom["ms"], 3), "all_kernels_ms": round(ev_total_ms, 2),
                                    "note": "HIP events around each eager launch: includes launch gaps, upper bound on kernel time"},
                    "measured_ceilings": {"fabric_TB_per_s": FABRIC_TBS,
                                          "traffic_time_share": (round(traffic / (FABRIC_TBS * 1e12) / (fam_ms * 1e-3 / max(1, fam_launches)), 3)
                                                                 if traffic else None),
                                          "source": "tools/probes/stream_rate_probe.hip, profiles/r02_stream_rate_probe.txt"},
                    "families": fam, "kernel_trace": kt,
                    "whole_step_frac_of_mfma_peak": round(
                        images_per_s / world * hot_path_tflop_per_image(args.res) / MFMA_PEAK_TFLOPS, 4),
                    "whole_step_algorithmic_TFLOPs": round(images_per_s / world * hot_path_tflop_per_image(args.res), 1)}

    LEG_SECONDS['roofline(trace+pmc children)'] = time.perf_counter() - _t_leg; _t_leg = time.perf_counter()
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(full=args.cpu_baseline_full)

    LEG_SECONDS['cpu_baseline'] = time.perf_counter() - _t_leg
    if rank == 0:
        print("bench legs (s):", {k: round(v, 1) for k, v in LEG_SECONDS.items()}, file=sys.stderr)
        print(json.dumps({
            "metric": "train images/sec SD-1.5+ControlLoRA 512^2 bs4/GPU", "value": round(images_per_s, 3),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"configs/{args.config} on SD-1.5 topology (seeded random weights), {args.res}x{args.res}, "
                                   f"bs={args.batch}/GPU, fp16; hot path = hint encoder + UNet fwd/bwd + adapter AdamW; "
                                   f"latents/text embeddings synthetic (VAE/CLIP outside the hot path)",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "hipgraph": graphed,
                       "allreduce_bytes": trainer.flat.numel * 4,
                       "rccl_ranks": trainer.comm_ranks(), "comm": trainer.comm, "comm_requested": trainer.comm_requested,
                       "comm_fallback": trainer.comm_fallback, "rccl_library": trainer.comm_library(),
                       "allreduce_ms": allreduce_ms,
                       "multi_gpu_note": ("measured at N = %d ranks" % world) if world > 1 else
                                         "N > 1 scaling curve UNMEASURED by the builder (gpurun exposes one GPU): the driver's scaling run is the only place it can be measured"},
            "timed_windows": {"n": len(window_ms), "steps_each": args.steps, "ms_per_step": window_ms, "reported": "median",
                              "spread_pct": round((max(window_ms) - min(window_ms)) / min(window_ms) * 100, 2)},
            "calibration": calib,
            "loss": round(loss, 5), "steps_skipped_by_scaler": skipped,
            # in-launch exchanges of the one-launch GroupNorm kernels that gave up (include/clora.h clora_groupnorm_*_team): must be 0
            "gn_team_errors": K.gn_team_errors(dev),
            # compensated residual trunk (kernels.TrunkLo): "infer" = the samplers / validation only (the train step above is unaffected)
            "trunk_lo": K.TRUNK_LO_MODE,
            "roofline": roofline, "ddim50": ddim, "full_step_with_vae_clip": full, "cpu_baseline": cpu}))
    trainer.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
