"""`python bench.py --gpus N` must start N ranks by itself: the spawn path (re-exec under
torch.distributed.run, rendezvous on 127.0.0.1, flat-buffer all-reduce, rank-0 JSON line with the number of ranks that
actually joined) exercised on CPU with `--backend gloo --dry-run`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--dry-run", "--steps", "2"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out                  # ONE line, from rank 0
    line = json.loads(lines[0])
    line["_output"] = out
    return line


def test_gpus_flag_spawns_that_many_ranks():
    line = _run(["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["allreduce_ok"] is True
    assert line["allreduce_bytes"] == 6047040 * 4


def test_single_rank_does_not_spawn():
    line = _run(["--gpus", "1"])
    assert line["n_gpus"] == 1


def test_requested_clora_exchange_falls_back_loudly_without_rccl():
    """`--comm clora` on a process group with no RCCL behind it (gloo) must not be silently replaced: the
    line says which path ran, which was asked for and why, and a warning goes to stderr"""
    line = _run(["--gpus", "2", "--comm", "clora"])
    assert line["n_gpus"] == 2 and line["allreduce_ok"] is True
    assert line["comm"] == "torch" and line["comm_requested"] == "clora" and "no RCCL" in line["comm_fallback"]
    assert "WARNING" in line["_output"] and "clora" in line["_output"]
    quiet = _run(["--gpus", "2"])
    assert quiet["comm_requested"] == "auto" and quiet["comm_fallback"] is None and "WARNING" not in quiet["_output"]
