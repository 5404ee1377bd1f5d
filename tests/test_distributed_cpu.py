"""world_size-2 and -4 `gloo` tests of the data-parallel path: adapter broadcast at start,
one all-reduce(mean) of the flat gradient buffer per step, identical parameters on every rank afterwards,
and reduced gradient == mean of the per-rank gradients == gradient of the global batch."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,case,comm", [(2, "v1", ""), (4, "v1", ""), (2, "v2", ""), (2, "v1", "clora")])
def test_data_parallel_step(tmp_path, world, case, comm):
    """case v2: configs[3]'s processor family (mpii-pose-v2.json geometry: concat adapters, the control map's share of their
    down-projections evaluated once per level, ops.control_down_parts)"""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", CLORA_DIST_CASE=case, CLORA_DIST_COMM=comm)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert ("exchange path 'clora' unavailable" in outs[0]) == (comm == "clora"), outs[0]      # loud on stderr, silent otherwise
    rs = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    for r in rs[1:]:
        assert torch.equal(rs[0]["params"], r["params"]), "ranks diverged after the step"
        assert torch.equal(rs[0]["reduced_grad"], r["reduced_grad"])
    mean = sum(r["local_grad"] for r in rs) / world
    err = float((rs[0]["reduced_grad"] - mean).norm() / mean.norm())
    assert err < 1e-6, err
    assert float((rs[0]["local_grad"] - rs[1]["local_grad"]).norm()) > 0       # the shards really differ
