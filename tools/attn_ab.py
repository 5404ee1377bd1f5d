"""Attention kernels at the SD-1.5 site shapes (B=4, 512^2): one line per shape; run once per build of libclora.so
(CLORA_LIB_PATH selects the build) for same-box A/B comparisons."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import kbench

torch.manual_seed(0)
for shape in ((4, 8, 4096, 4096, 40), (4, 8, 4096, 77, 40), (4, 8, 1024, 1024, 80), (4, 8, 1024, 77, 80), (4, 8, 256, 256, 160),
              (32, 8, 4096, 4096, 40)):
    kbench.bench_attn(*shape)
if len(sys.argv) > 1:
    json.dump(kbench.rows, open(sys.argv[1], "w"), indent=1)
