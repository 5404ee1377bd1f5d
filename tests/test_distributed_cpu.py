"""world_size-2 `gloo` test of the data-parallel path: adapter broadcast at start,
one all-reduce(mean) of the flat gradient buffer per step, identical parameters on every rank afterwards,
and reduced gradient == mean of the per-rank gradients == gradient of the global batch."""
import os
import socket
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_data_parallel_step(tmp_path):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert torch.equal(r0["params"], r1["params"]), "ranks diverged after the step"
    assert torch.equal(r0["reduced_grad"], r1["reduced_grad"])
    mean = 0.5 * (r0["local_grad"] + r1["local_grad"])
    err = float((r0["reduced_grad"] - mean).norm() / mean.norm())
    assert err < 1e-6, err
    assert float((r0["local_grad"] - r1["local_grad"]).norm()) > 0       # the shards really differ
