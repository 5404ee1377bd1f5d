"""Forward attention block width (clora_set_option "attn_fwd_waves": 8 = 256 queries per block, 16 = 512) at the level-0 self-attention
shapes of the batch-32 sampler and of the train step, interleaved rounds inside hipGraphs.   python tools/attn_fwd_width_ab.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K
dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5

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

for name, B, H, Nq, Nk, D in [("b32 L0 self", 32, 8, 4096, 4096, 40), ("b4 L0 self", 4, 8, 4096, 4096, 40), ("b32 L0 cross", 32, 8, 4096, 77, 40),
                              ("b16 L0 self", 16, 8, 4096, 4096, 40), ("b8 L0 self", 8, 8, 4096, 4096, 40)]:
    g_ = torch.Generator(device=dev).manual_seed(1)
    qkv = (torch.randn(B * Nq, 3 * H * D, device=dev, generator=g_) * 0.5).half()
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    if Nk != Nq:
        kv = (torch.randn(B * Nk, 2 * H * D, device=dev, generator=g_) * 0.5).half()
        k, v = kv[:, :H * D], kv[:, H * D:]
    iters = 4 if B >= 16 else 10
    outs, graphs = {}, {}
    for w in (8, 16, 4):
        K.set_option("attn_fwd_waves", w)
        o, lse = K.attn_fwd(q, k, v, B, H, Nq, Nk, D, D ** -0.5)
        outs[w] = o.clone()
        graphs[w] = graph_of(lambda: K.attn_fwd(q, k, v, B, H, Nq, Nk, D, D ** -0.5), iters)
    K.set_option("attn_fwd_waves", 0)
    ts = {w: [] for w in graphs}
    for _ in range(rounds):
        for w, g in graphs.items():
            ts[w].append(t_graph(g, iters))
    med = {w: sorted(v_)[len(v_) // 2] for w, v_ in ts.items()}
    fl = 4.0 * B * H * Nq * Nk * D
    print(f"{name:14s} " + "  ".join(f"{w:2d} waves {med[w]:8.1f}us {fl / med[w] / 1e6:6.0f}TF(useful){'' if torch.equal(outs[w], outs[8]) else ' d=%.1e' % float((outs[w].float() - outs[8].float()).norm() / outs[8].float().norm())}" for w in med), flush=True)
