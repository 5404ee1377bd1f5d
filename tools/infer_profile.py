"""Per-entry-point / per-shape time of ONE batch-32 UNet forward of the DDIM loop (512^2, 16 images x CFG), HIP events around
each eager launch (includes launch gaps: use for ranking, not for absolute kernel time).  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from controllora_amd import kernels as K
from controllora_amd import models as M

dev = torch.device("cuda", 0)
unet, clora = bench.build_models(dev)
b = bench.synthetic_batch(1, 512, dev, 42)
nb = 32
g = torch.Generator(device=dev).manual_seed(1)
ehs = torch.randn(nb, 77, 768, device=dev, generator=g).half()
lat = torch.randn(nb, 4, 64, 64, device=dev, generator=g).half()
with torch.no_grad(), M.text_kv_cache():
    clora(b["guide"][:1])
    for _ in range(2):
        unet(lat, 10, ehs)
    torch.cuda.synchronize()
    K.PROFILER = K.KernelProfiler(detail=True)
    unet(lat, 10, ehs)
    agg = K.PROFILER.summary()
    K.PROFILER = None
rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(a["ms"] for _, a in rows)
print(f"== one UNet forward at batch {nb}: {tot:.2f} ms of event-timed launches")
for name, a in rows[:int(os.environ.get("TOPN", "60"))]:
    tf = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["flops"] else 0
    gb = a["bytes"] / (a["ms"] * 1e-3) / 1e9 if a.get("bytes") else 0
    print(f"{a['ms']:8.3f} ms {a['calls']:4d}x {a['ms']*1e3/a['calls']:8.1f} us {tf:7.1f} TF {gb:8.0f} GB/s  {name}")
