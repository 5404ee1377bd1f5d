"""tile_cfg 59 (gemm_8p_kernel: 256x256, eight-phase ping-pong) against the incumbent tiles on the plain GEMM shapes it is meant
for, timed inside a hipGraph (interleaved rounds, median), plus a race screen: N launches of one shape must be bit-identical and
match an fp32 matmul.   python tools/gemm8p_ab.py [rounds]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K
dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if "--pmc" in sys.argv:
    # counter collection (rocprofv3 --pmc ... -- python tools/gemm8p_ab.py 1 --pmc): a few eager launches of the calibration GEMM per tile
    import math as _m
    for n in (8192, 4096):
        g_ = torch.Generator(device=dev).manual_seed(1)
        A = (torch.rand(n, n, device=dev, generator=g_) - 0.5).half()
        Bw = ((torch.rand(n, n, device=dev, generator=g_) - 0.5) * (2.0 / _m.sqrt(n))).half()
        out = torch.empty(n, n, device=dev, dtype=torch.float16)
        for c in (1, 56, 58, 59):
            for _ in range(3):
                K.gemm(A, Bw, n, n, n, out=out, split_k=1, tile_cfg=c, _tuned=False)
        torch.cuda.synchronize()
    sys.exit(0)

def graph_of(fn, iters):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    return g

def t_graph(g, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

SHAPES = [("calib 8192^3", 8192, 8192, 8192), ("4096^3", 4096, 4096, 4096), ("ff1 L0 16384x2560x320", 16384, 2560, 320),
          ("ff2' L0 16384x1280x320", 16384, 1280, 320), ("qkv L0 16384x960x320", 16384, 960, 320), ("ff1 L1 4096x5120x640", 4096, 5120, 640),
          ("ff1 L2 1024x10240x1280", 1024, 10240, 1280), ("ff1 b32 131072x2560x320", 131072, 2560, 320), ("qkv b32 131072x960x320", 131072, 960, 320),
          ("ff1 b32 L1 32768x5120x640", 32768, 5120, 640), ("ff2 b32 131072x320x1280", 131072, 320, 1280), ("16384x2560x2560", 16384, 2560, 2560)]
CFGS = [0, 1, 21, 53, 56, 58, 59]
for name, M, N, Kd in SHAPES:
    g_ = torch.Generator(device=dev).manual_seed(1)
    A = (torch.rand(M, Kd, device=dev, generator=g_) - 0.5).half()
    Bw = ((torch.rand(N, Kd, device=dev, generator=g_) - 0.5) * (2.0 / math.sqrt(Kd))).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    iters = 3 if M * N * Kd > 2e11 else 10
    graphs = {}
    for c in CFGS:
        try:
            graphs[c] = graph_of(lambda c=c: K.gemm(A, Bw, M, N, Kd, out=out, split_k=(0 if c == 0 else 1), tile_cfg=c, _tuned=(c == 0)), iters)
        except Exception as e:                          # noqa: BLE001
            print(name, "cfg", c, "failed:", repr(e)[:80])
    ts = {c: [] for c in graphs}
    for _ in range(rounds):
        for c, g in graphs.items():
            ts[c].append(t_graph(g, iters))
    med = {c: sorted(v)[len(v) // 2] for c, v in ts.items()}
    best_other = min(v for c, v in med.items() if c != 59)
    line = "  ".join(f"{('table' if c == 0 else 'cfg%d' % c)} {med[c]:8.1f}us {2.0 * M * N * Kd / med[c] / 1e6:6.0f}TF" for c in med)
    print(f"{name:28s} {line}   | 59 vs best other: {best_other / med[59]:.3f}x", flush=True)
    del graphs

# race screen + accuracy: 12 launches of three shapes, bitwise equal, vs fp32 matmul on a row sample
for M, N, Kd, sk in [(4096, 4096, 4096, 1), (16384, 2560, 320, 1), (1000, 520, 1096, 2), (8192, 8192, 8192, 1)]:
    g_ = torch.Generator(device=dev).manual_seed(2)
    A = (torch.rand(M, Kd, device=dev, generator=g_) - 0.5).half()
    Bw = ((torch.rand(N, Kd, device=dev, generator=g_) - 0.5) * (2.0 / math.sqrt(Kd))).half()
    ref_rows = torch.arange(0, M, max(1, M // 512), device=dev)
    ref = A[ref_rows].float() @ Bw.float().t()
    first, bad = None, 0
    for it in range(12):
        out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        K.gemm(A, Bw, M, N, Kd, out=out, split_k=sk, tile_cfg=59, _tuned=False)
        if first is None:
            first = out.clone()
        elif not torch.equal(first, out):
            bad += 1
    rel = float((first[ref_rows].float() - ref).norm() / ref.norm())
    worst = float((first[ref_rows].float() - ref).abs().max() / ref.abs().max())
    print(f"race screen {M}x{N}x{Kd} sk{sk}: {bad} of 11 repeats differ, rel-L2 vs fp32 {rel:.2e}, worst |err| / max|ref| {worst:.2e}, finite {bool(torch.isfinite(first.float()).all())}", flush=True)
