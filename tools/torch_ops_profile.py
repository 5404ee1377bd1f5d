"""List the stock torch (aten) kernels still issued inside one train step, with input shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from controllora_amd.schedulers import DDPMScheduler
from controllora_amd.train import ControlLoRATrainer
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
unet, clora = bench.build_models(dev)
trainer = ControlLoRATrainer(unet, clora)
batch = bench.synthetic_batch(4, 512, dev, 42)
noisy = DDPMScheduler().add_noise(batch["latents"], batch["noise"], batch["timesteps"]).half()
step = lambda: trainer.step(noisy, batch["timesteps"], batch["ehs"], batch["guide"], batch["noise"])
for _ in range(2):
    step()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8) if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"stock torch kernels in one eager train step: {tot/1e3:.3f} ms over {sum(e.count for e in rows)} launches")
for e in rows[:int(os.environ.get('TOPN', '40'))]:
    where = [f for f in e.stack if "/root/repo" in f or "controllora_amd" in f or "bench.py" in f][:3]
    print(f"{e.self_device_time_total/1e3:8.3f} ms {e.count:5d}x  {e.key:24s} {str(e.input_shapes)[:90]:90s} <- {' | '.join(w.split('/')[-1] for w in where)}")
