"""VAE encode / decode timing at SD-1.5 shapes (random weights): B images of res^2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import vae as V, kernels as K
dev = torch.device("cuda", 0)
B, res = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = V.AutoencoderKL(**V.SD15_VAE); V.init_random_(m, 1); m.to(dev)
x = (torch.rand(B, 3, res, res, device=dev) * 2 - 1).half()
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
enc = t(lambda: m.encode(x).latent_dist.sample())
z = m.encode(x).latent_dist.sample()
print("latent finite", bool(torch.isfinite(z).all()), float(z.abs().mean()))
dec = t(lambda: m.decode(z.half()))
img = m.decode(z.half()).sample
print("image finite", bool(torch.isfinite(img.float()).all()))
print(f"encode {enc:.2f} ms ({B} x {res}^2: {1.1167*B*(res/512)**2/enc*1e3:.0f} TFLOP/s)   decode {dec:.2f} ms ({2.5145*B*(res/512)**2/dec*1e3:.0f} TFLOP/s)")
K.PROFILER = K.KernelProfiler()
m.encode(x)
agg = K.PROFILER.summary(); K.PROFILER = None
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:6]:
    print(f"  {k:28s} {v['calls']:4d}x {v['ms']:8.3f} ms")
