"""Per-layer error budget of ONE UNet evaluation (VERDICT r04 "missing" 6): which blocks contribute the ~1.6e-3 rel-L2 that separates
the product's prediction from the fp32 oracle's on BASELINE's inference geometry (512x512, CFG batch 2, fill50k adapters, control
batch 1; the first DDIM step's inputs).

For every ResnetBlock2D / Transformer2DModel / Downsample2D / Upsample2D of the SD-1.5 topology, in execution order:
  cumulative  rel-L2(product block output, fp32 oracle block output) with the product fed by ITS OWN upstream activations: the
              error that has accumulated up to this point of the network;
  local       the same block of the product re-run on the ORACLE's input (rounded to fp16) against the oracle's output: what this
              block adds by itself (its own arithmetic + one fp16 rounding of its input and output);
  floor       rel-L2(fp16(oracle output), oracle output): the cost of merely STORING this block's output in fp16.
The oracle (oracle/unet_ref.py + oracle/controllora_ref.py, test infrastructure) runs in fp32 with stock torch ops on the GPU.

    python tools/error_budget_layers.py [out.json] [res] [batch]          (GPU box; < 1 min)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KINDS = ("ResnetBlock2D", "Transformer2DModel", "Downsample2D", "Upsample2D")


def to_tokens(t):
    """oracle NCHW -> the product's [B, H*W, C]"""
    return t.permute(0, 2, 3, 1).reshape(t.shape[0], -1, t.shape[1]) if t.ndim == 4 else t


def rel(a, b):
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).norm() / b.norm())


@torch.no_grad()
def main():
    from controllora_amd import kernels as K
    from tests import full_cases as F
    dev = "cuda"
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    o_unet, o_clora, p_unet, p_clora = F.build_pair("fill50k.json", dev)
    o_unet.to(dev).float(); o_clora.to(dev).float()
    g = torch.Generator().manual_seed(5)
    L = res // 8
    guide = ((torch.rand(1, 3, res, res, generator=g) > 0.9).float() * 2 - 1).to(dev)
    lat = torch.randn(nb, 4, L, L, generator=g).half().float().to(dev)
    ehs = torch.randn(nb, 77, 768, generator=g).half().float().to(dev)
    t = 981

    o_rec, p_rec, p_args = {}, {}, {}
    names = [n for n, m in o_unet.named_modules() if type(m).__name__ in KINDS]
    p_mods = dict(p_unet.named_modules())
    order = []
    hooks = []
    for n, m in o_unet.named_modules():
        if type(m).__name__ in KINDS:
            def oh(mod, args, out, n=n):
                o_rec[n] = (args[0].detach(), (out[0] if isinstance(out, tuple) else out).detach())
                order.append(n)
            hooks.append(m.register_forward_hook(oh))
            pm = p_mods[n]
            def ph(mod, args, kwargs, out, n=n):
                p_rec[n] = (out[0] if isinstance(out, tuple) else out).detach()
                p_args[n] = (args, kwargs)
            hooks.append(pm.register_forward_hook(ph, with_kwargs=True))
    o_clora(guide)
    o_pred = o_unet(lat, t, ehs).sample
    p_clora(guide.half())
    p_pred = p_unet(lat.half(), t, ehs.half()).sample
    for h in hooks:
        h.remove()
    rows = []
    for n in order:
        oin, oout = o_rec[n]
        oin_t, oout_t = to_tokens(oin), to_tokens(oout)
        cum = rel(p_rec[n].reshape(oout_t.shape), oout_t)
        args, kwargs = p_args[n]
        x_p = args[0]
        if kwargs.get("skip") is not None:                # up-path resnets (round 6): the product reads cat([x, skip]) in place -- split the oracle's input
            Cx = x_p.shape[-1]
            kwargs = dict(kwargs, skip=oin_t[..., Cx:].reshape(kwargs["skip"].shape).half().contiguous())
            oin_t = oin_t[..., :Cx]
        x_o = oin_t.reshape(x_p.shape).half().contiguous()
        out_p = p_mods[n](x_o, *args[1:], **kwargs)
        K.flush_pending()                                 # (out_to_norm = True: a split-K producer may have left its finish to the next norm)
        local = rel(out_p.reshape(oout_t.shape), oout_t)
        floor = rel(oout_t.half(), oout_t)
        rows.append({"module": n, "kind": type(p_mods[n]).__name__, "cumulative": cum, "local": local, "fp16_storage_floor": floor,
                     "tokens": int(oout_t.shape[1]), "channels": int(oout_t.shape[2])})
    out = {"what": f"one UNet evaluation, {res}x{res}, batch {nb}, t = {t}, fill50k adapters (non-zero up matrices), product vs fp32 oracle",
           "prediction_rel_l2": rel(p_pred, o_pred), "blocks": rows}
    by_kind = {}
    for r in rows:
        by_kind.setdefault(r["kind"], []).append(r["local"])
    out["local_by_kind"] = {k: {"n": len(v), "mean": sum(v) / len(v), "max": max(v)} for k, v in by_kind.items()}
    # if the local contributions were independent they would add in quadrature along the residual trunk
    out["quadrature_of_locals"] = sum(r["local"] ** 2 for r in rows) ** 0.5
    print(f"prediction rel-L2 {out['prediction_rel_l2']:.2e}; sqrt(sum local^2) {out['quadrature_of_locals']:.2e}")
    print(f"{'block':44s} {'cumulative':>10s} {'local':>9s} {'fp16 floor':>10s}")
    for r in rows:
        print(f"{r['module']:44s} {r['cumulative']:10.2e} {r['local']:9.2e} {r['fp16_storage_floor']:10.2e}")
    print("local by kind:", json.dumps(out["local_by_kind"]))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
