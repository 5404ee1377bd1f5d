"""Per-shape GEMM breakdown + torch-op breakdown of one train step (run on the GPU box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from controllora_amd import kernels as K
from controllora_amd.schedulers import DDPMScheduler
from controllora_amd.train import ControlLoRATrainer

dev = torch.device("cuda", 0)
unet, clora = bench.build_models(dev)
trainer = ControlLoRATrainer(unet, clora)
batch = bench.synthetic_batch(4, 512, dev, 42)
noisy = DDPMScheduler().add_noise(batch["latents"], batch["noise"], batch["timesteps"]).half()
step = lambda: trainer.step(noisy, batch["timesteps"], batch["ehs"], batch["guide"], batch["noise"])
for _ in range(2):
    step()
K.PROFILER = K.KernelProfiler(detail=True)
step()
agg = K.PROFILER.summary()
K.PROFILER = None
rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
print("== kernel entry points by shape (top 40)")
for name, a in rows[:40]:
    tf = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["flops"] else 0
    print(f"{a['ms']:8.3f} ms {a['calls']:4d}x {a['ms']*1e3/a['calls']:8.1f} us {tf:7.1f} TF  {name}")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=50, max_shapes_column_width=60))
