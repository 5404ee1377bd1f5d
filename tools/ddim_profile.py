"""One hint-encode + N DDIM steps of the BASELINE inference configuration (512^2, 16 images, CFG -> UNet batch 32) for a
rocprofv3 --kernel-trace run:   rocprofv3 --kernel-trace -d /tmp/ddim -o kt -- python tools/ddim_profile.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from controllora_amd.pipeline import ddim_sample

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
unet, clora = bench.build_models(dev)
b = bench.synthetic_batch(1, 512, dev, 42)
g = torch.Generator(device=dev).manual_seed(1)
nb = 16
cond = torch.randn(nb, 77, 768, device=dev, generator=g).half()
uncond = torch.randn(nb, 77, 768, device=dev, generator=g).half()
lat0 = torch.randn(nb, 4, 64, 64, device=dev, generator=g).half()
ddim_sample(unet, clora, b["guide"][:1], cond, uncond, steps=2, latents=lat0)
torch.cuda.synchronize()
t0 = time.perf_counter()
ddim_sample(unet, clora, b["guide"][:1], cond, uncond, steps=steps, latents=lat0)
torch.cuda.synchronize()
print(f"{steps} steps: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step")
