"""Is the GEMM main loop bound by L2 -> LDS operand traffic?  Times each shape with the normal kernel and with the A and / or
B tile DMAs redirected to a 16-byte zero page (same instruction stream, no traffic; needs the probe build:
tools/dma_probe.sh, CLORA_LIB_PATH=controllora_amd/_build_probe/libclora.so).  10 launches per hipGraph replay."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import gemm_pilot as GP
from controllora_amd import kernels as K

rows = []
for (M, N, Kd, H, Cin) in [(16384, 320, 2880, 64, 320), (4096, 640, 5760, 32, 640), (1024, 1280, 11520, 16, 1280), (131072, 320, 2880, 64, 320),
                           (16384, 320, 1280, 0, 0), (16384, 2560, 320, 0, 0), (8192, 8192, 8192, 0, 0)]:
    A, Bw, out, res, cd = GP.operands(M, N, Kd, H, Cin)
    line = []
    for base, probes in ((21, (91, 92, 93)), (26, (94, 95, 96))):
        for cfg in (base,) + probes:
            sk = 1
            us = GP.timeit(lambda: K.gemm(A, Bw, M, N, Kd, conv=cd, out=out, split_k=sk, tile_cfg=cfg, _tuned=False))
            rows.append(dict(M=M, N=N, K=Kd, conv=bool(H), cfg=cfg, us=round(us, 2)))
            line.append(f"{cfg}:{us:.1f}")
    print(f"{M}x{N}x{Kd}{'c' if H else ' '}  [128x128: normal, A=0, B=0, both=0 | 128x64: ...]  " + " ".join(line), flush=True)
    del A, Bw, out, res
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
