"""Box-speed indicator for comparing runs from different gpurun boxes (the fleet spreads by several per cent): two fixed
kernels whose code never changes -- the 8192^3 GEMM on tile_cfg 1 and a 256 MB HBM copy -- timed inside a hipGraph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import gemm_pilot as GP
from controllora_amd import kernels as K

K.set_tile_order("auto")          # the order every earlier calibration file was taken under (the library default is "grid" since round 5)
A, Bw, out, res, cd = GP.operands(8192, 8192, 8192, 0, 0)
us = GP.timeit(lambda: K.gemm(A, Bw, 8192, 8192, 8192, out=out, split_k=1, tile_cfg=1, _tuned=False), iters=4)
x = torch.empty(128 << 20, dtype=torch.float16, device="cuda")
y = torch.empty_like(x)
cp = GP.timeit(lambda: y.copy_(x), iters=10)
print(f"BOX_CALIB gemm8192_cfg1 {us:.1f} us ({2 * 8192**3 / us / 1e6:.0f} TF)   copy256MB {cp:.1f} us ({2 * x.numel() * 2 / cp / 1e3:.0f} GB/s)")
