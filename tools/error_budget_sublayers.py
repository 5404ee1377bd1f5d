"""One level below tools/error_budget_layers.py: WHICH fp16 roundings inside the worst block of the per-block budget
-- `down_blocks.0.attentions.0`, a Transformer2DModel at 64x64x320 whose local error is 5.7e-4 against a 2.07e-4 storage floor -- carry
that error.

The block's fp32 ORACLE (oracle/unet_ref.py + oracle/controllora_ref.py, test infrastructure, stock torch ops on the GPU) is re-run on
the oracle's own input with ONE named intermediate rounded to fp16 at a time -- exactly the tensors the product stores or feeds to an
fp16 MFMA -- then with all of them, and compared with the un-rounded run:

  gn_out      GroupNorm output (operand of proj_in)              proj_in     proj_in output (the transformer's residual stream enters in fp16)
  ln1/2/3     LayerNorm outputs (operands of the projections)    qkv1/2      q, k, v after the adapter update (operands of QK^T / PV)
  qscale1/2   q * scale*log2(e) rounded again (the forward kernel pre-multiplies Q, clora_attn.hip)
  p1/2        softmax probabilities as the fp16 operand of PV    ctx1/2      attention output (operand of to_out)
  sum1/2/3    the residual sums written by to_out / FF2          geglu       a * gelu(g) (operand of FF2)
  out_sum     proj_out + residual (the block's output)

and the PRODUCT's own local error on the same input is printed beside them (the block of the product UNet fed with the oracle's input
rounded to fp16, as in error_budget_layers.py).  If the roundings were independent their contributions would add in quadrature.

    python tools/error_budget_sublayers.py [out.json] [res] [batch] [module]          (GPU box; < 1 min)
"""
import json
import math
import os
import sys
import types

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

POINTS = ["gn_out", "proj_in", "ln1", "qkv1", "qscale1", "p1", "ctx1", "sum1", "ln2", "qkv2", "qscale2", "p2", "ctx2", "sum2", "ln3",
          "geglu", "sum3", "out_sum"]
ACTIVE = set()


def R(name, t):
    return t.half().float() if name in ACTIVE else t


def rel(a, b):
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).norm() / b.norm())


def instrument(tr):
    """rounding points on ONE oracle Transformer2DModel instance (instance-level patches: the class stays untouched)"""
    blk = tr.transformer_blocks[0]

    def attn_patches(attn, idx):
        h2b, b2h = attn.head_to_batch_dim, attn.batch_to_head_dim
        c = attn.scale * math.log2(math.e)

        def head_to_batch_dim(self, t):
            return h2b(R(f"qkv{idx}", t))

        def batch_to_head_dim(self, t):
            return R(f"ctx{idx}", b2h(t))

        def get_attention_scores(self, query, key, attention_mask=None):
            q = R(f"qscale{idx}", query * c) / c
            scores = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], key.shape[1], dtype=q.dtype, device=q.device), q,
                                   key.transpose(-1, -2), beta=0, alpha=self.scale)
            return R(f"p{idx}", scores.softmax(dim=-1))
        attn.head_to_batch_dim = types.MethodType(head_to_batch_dim, attn)
        attn.batch_to_head_dim = types.MethodType(batch_to_head_dim, attn)
        attn.get_attention_scores = types.MethodType(get_attention_scores, attn)
    attn_patches(blk.attn1, 1)
    attn_patches(blk.attn2, 2)

    def geglu_forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return R("geglu", a * F.gelu(g))
    blk.ff.net[0].forward = types.MethodType(geglu_forward, blk.ff.net[0])

    def block_forward(self, x, encoder_hidden_states=None, cross_attention_kwargs=None):
        kw = cross_attention_kwargs or {}
        x = R("sum1", self.attn1(R("ln1", self.norm1(x)), **kw) + x)
        x = R("sum2", self.attn2(R("ln2", self.norm2(x)), encoder_hidden_states=encoder_hidden_states, **kw) + x)
        return R("sum3", self.ff(R("ln3", self.norm3(x))) + x)
    blk.forward = types.MethodType(block_forward, blk)

    def tr_forward(self, x, encoder_hidden_states=None, cross_attention_kwargs=None):
        b, c, h, w = x.shape
        res = x
        x = R("proj_in", self.proj_in(R("gn_out", self.norm(x))))
        inner = x.shape[1]
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        for blk_ in self.transformer_blocks:
            x = blk_(x, encoder_hidden_states, cross_attention_kwargs)
        x = x.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
        return R("out_sum", self.proj_out(x) + res)
    tr.forward = types.MethodType(tr_forward, tr)


# ------------------------------------------------------------------------------------------ whole UNet: what a compensated trunk would buy
# `python tools/error_budget_sublayers.py out.json 512 2 unet`: the WHOLE fp32 oracle UNet re-run with the product's roundings at every
# stored tensor, in two trunk regimes:
#   "rounded"      every residual sum / trunk tensor (conv_in, resnet outputs, shortcut convs, proj_in, the three sums of a transformer
#                  block, proj_out + residual, down / up-sampler outputs) is rounded to fp16 where it is written -- the product today;
#   "compensated"  the same roundings as seen by every NON-ADD consumer (norms, GEMM / conv operands, concatenations), but the next
#                  residual add continues from the unrounded sum -- what an epilogue that writes the rounding remainder as a second fp16
#                  tensor (hi + lo) and adds it back in the next residual add would compute; norms and MFMA operands still read hi only;
#   "trunk32"      as "compensated", and the norms read hi + lo as well (only GEMM / conv operands see the rounded trunk).
TRUNK = {"mode": "rounded"}


def r16(t):
    return t.half().float()


def T(v):
    return r16(v) if TRUNK["mode"] == "rounded" else v


def NIN(x):
    """what a norm reads of a trunk tensor: hi only, or -- "trunk32" -- hi + lo"""
    return x if TRUNK["mode"] == "trunk32" else r16(x)


def instrument_unet(u):
    from oracle import unet_ref as UR

    def resnet_forward(self, x, temb):
        h = r16(self.conv1(r16(F.silu(self.norm1(NIN(x))))) + r16(self.time_emb_proj(F.silu(temb)))[:, :, None, None])
        b = r16(F.silu(self.norm2(h)))
        sc = T(self.conv_shortcut(r16(x))) if self.conv_shortcut is not None else x
        return T(self.conv2(b) + sc)

    def block_forward(self, x, encoder_hidden_states=None, cross_attention_kwargs=None):
        kw = cross_attention_kwargs or {}
        x = T(self.attn1(r16(self.norm1(NIN(x))), **kw) + x)
        x = T(self.attn2(r16(self.norm2(NIN(x))), encoder_hidden_states=encoder_hidden_states, **kw) + x)
        return T(self.ff(r16(self.norm3(NIN(x)))) + x)

    def tr_forward(self, x, encoder_hidden_states=None, cross_attention_kwargs=None):
        b, c, h, w = x.shape
        res = x
        x = T(self.proj_in(r16(self.norm(NIN(x)))))
        inner = x.shape[1]
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        for blk_ in self.transformer_blocks:
            x = blk_(x, encoder_hidden_states, cross_attention_kwargs)
        x = x.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
        return T(self.proj_out(r16(x)) + res)

    def down_forward(self, x):
        x = r16(x)
        if self.use_conv and self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return T(self.conv(x))

    def up_forward(self, x):
        x = F.interpolate(r16(x), scale_factor=2.0, mode="nearest")
        return T(self.conv(x)) if self.use_conv else x

    def geglu_forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return r16(a * F.gelu(g))

    def unet_forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, return_dict=True):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.long, device=sample.device)
        timestep = timestep.reshape(-1).expand(sample.shape[0])
        emb = r16(self.time_embedding(r16(UR.timestep_embedding(timestep, self.config.block_out_channels[0]))))
        x = T(self.conv_in(sample))
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states, cross_attention_kwargs)
            skips.extend(outs)
        x = self.mid_block(x, emb, encoder_hidden_states, cross_attention_kwargs)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, encoder_hidden_states, cross_attention_kwargs)
        x = r16(self.conv_out(r16(F.silu(self.conv_norm_out(NIN(x))))))
        return UR.UNetOutput(sample=x) if return_dict else (x,)

    patch = {UR.ResnetBlock2D: resnet_forward, UR.BasicTransformerBlock: block_forward, UR.Transformer2DModel: tr_forward,
             UR.Downsample2D: down_forward, UR.Upsample2D: up_forward, UR.GEGLU: geglu_forward, UR.UNet2DConditionModel: unet_forward}
    for m in u.modules():
        fn = patch.get(type(m))
        if fn is not None:
            m.forward = types.MethodType(fn, m)
        if isinstance(m, UR.CrossAttention):
            def mk(attn):
                h2b, b2h = attn.head_to_batch_dim, attn.batch_to_head_dim
                c = attn.scale * math.log2(math.e)

                def head_to_batch_dim(self, t):
                    return h2b(r16(t))

                def batch_to_head_dim(self, t):
                    return r16(b2h(t))

                def get_attention_scores(self, query, key, attention_mask=None):
                    q = r16(query * c) / c
                    scores = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], key.shape[1], dtype=q.dtype, device=q.device), q,
                                           key.transpose(-1, -2), beta=0, alpha=self.scale)
                    return r16(scores.softmax(dim=-1))
                attn.head_to_batch_dim = types.MethodType(head_to_batch_dim, attn)
                attn.batch_to_head_dim = types.MethodType(batch_to_head_dim, attn)
                attn.get_attention_scores = types.MethodType(get_attention_scores, attn)
            mk(m)


@torch.no_grad()
def main_unet(out_path, res, nb):
    import copy
    from oracle.controllora_ref import map_processors_to_unet
    from tests import full_cases as FC
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    o_unet, o_clora, p_unet, p_clora = FC.build_pair("fill50k.json", dev)
    o_unet.to(dev).float(); o_clora.to(dev).float()
    g = torch.Generator().manual_seed(5)
    L = res // 8
    guide = ((torch.rand(1, 3, res, res, generator=g) > 0.9).float() * 2 - 1).to(dev)
    lat = torch.randn(nb, 4, L, L, generator=g).half().float().to(dev)
    ehs = torch.randn(nb, 77, 768, generator=g).half().float().to(dev)
    t = 981
    o_clora(guide)
    ref = o_unet(lat, t, ehs).sample
    out = {"what": f"one UNet evaluation, {res}x{res}, batch {nb}, t = {t}, fill50k adapters: rel-L2 of the prediction against the fp32 oracle",
           "fp16_storage_floor_of_the_prediction": rel(ref.half(), ref)}
    if dev == "cuda":
        p_clora(guide.half())
        out["product"] = rel(p_unet(lat.half(), t, ehs.half()).sample, ref)
    u = copy.deepcopy(o_unet)
    c = copy.deepcopy(o_clora)
    u.set_attn_processor(map_processors_to_unet(u, c))
    instrument_unet(u)
    c(guide)                                                         # (the hint encoder stays in fp32: its maps are rounded where they are used)
    for mode in ("rounded", "compensated", "trunk32"):
        TRUNK["mode"] = mode
        out[f"oracle_with_the_products_roundings_trunk_{mode}"] = rel(u(lat, t, ehs).sample, ref)
    for k, v in out.items():
        print(f"{k:60s} {v if isinstance(v, str) else format(v, '.3e')}")
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)


@torch.no_grad()
def main():
    if len(sys.argv) > 4 and sys.argv[4] == "unet":
        return main_unet(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    from tests import full_cases as FC
    dev = "cuda" if torch.cuda.is_available() else "cpu"          # (CPU: the oracle half only -- a syntax / plumbing check at a small size)
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    name = sys.argv[4] if len(sys.argv) > 4 else "down_blocks.0.attentions.0"
    o_unet, o_clora, p_unet, p_clora = FC.build_pair("fill50k.json", dev)
    o_unet.to(dev).float(); o_clora.to(dev).float()
    g = torch.Generator().manual_seed(5)
    L = res // 8
    guide = ((torch.rand(1, 3, res, res, generator=g) > 0.9).float() * 2 - 1).to(dev)
    lat = torch.randn(nb, 4, L, L, generator=g).half().float().to(dev)
    ehs = torch.randn(nb, 77, 768, generator=g).half().float().to(dev)
    t = 981
    o_mod = dict(o_unet.named_modules())[name]
    p_mod = dict(p_unet.named_modules())[name]
    rec = {}

    def o_hook(mod, args, kwargs, out):
        rec["o_args"], rec["o_kwargs"] = args, kwargs
    hk = o_mod.register_forward_hook(o_hook, with_kwargs=True)

    def p_hook(mod, args, kwargs, out):
        rec["p_args"], rec["p_kwargs"] = args, kwargs
    hk2 = p_mod.register_forward_hook(p_hook, with_kwargs=True)
    o_clora(guide)
    o_unet(lat, t, ehs)
    if dev == "cuda":
        p_clora(guide.half())
        p_unet(lat.half(), t, ehs.half())
    hk.remove(); hk2.remove()

    instrument(o_mod)
    args, kwargs = rec["o_args"], rec["o_kwargs"]
    x16 = args[0].half().float()                                    # the block's input as the product receives it
    ref = o_mod(x16, *args[1:], **kwargs)
    ref = ref[0] if isinstance(ref, tuple) else ref
    floor = rel(ref.half(), ref)
    rows = []
    for pt in POINTS:
        ACTIVE.clear(); ACTIVE.add(pt)
        out = o_mod(x16, *args[1:], **kwargs)
        rows.append((pt, rel(out[0] if isinstance(out, tuple) else out, ref)))
    ACTIVE.clear(); ACTIVE.update(POINTS)
    out = o_mod(x16, *args[1:], **kwargs)
    all_on = rel(out[0] if isinstance(out, tuple) else out, ref)
    ACTIVE.clear(); ACTIVE.update(p for p in POINTS if not p.startswith("qscale"))
    out = o_mod(x16, *args[1:], **kwargs)
    all_but_qscale = rel(out[0] if isinstance(out, tuple) else out, ref)
    ACTIVE.clear()
    # the product's block on the same input
    local = float("nan")
    if dev == "cuda":
        pa, pk = rec["p_args"], rec["p_kwargs"]
        x_p = x16.permute(0, 2, 3, 1).reshape(pa[0].shape).half().contiguous()
        prod = p_mod(x_p, *pa[1:], **pk)
        prod = prod[0] if isinstance(prod, tuple) else prod
        local = rel(prod, ref.permute(0, 2, 3, 1).reshape(prod.shape))
    quad = sum(v * v for _, v in rows) ** 0.5
    res_ = {"module": name, "what": f"{res}x{res}, batch {nb}, t = {t}, fill50k adapters; rel-L2 of the block output against the un-rounded fp32 oracle block",
            "fp16_storage_floor_of_the_output": floor, "one_rounding_at_a_time": dict(rows), "quadrature_of_the_single_roundings": quad,
            "all_roundings_together": all_on, "all_but_the_q_scale_rounding": all_but_qscale, "product_local_error": local}
    print(f"{name}: product local error {local:.2e}; oracle with every listed rounding {all_on:.2e} (without the second rounding of scaled Q "
          f"{all_but_qscale:.2e}); quadrature of the single roundings {quad:.2e}; storage floor of the output {floor:.2e}")
    for pt, v in sorted(rows, key=lambda r: -r[1]):
        print(f"  {pt:10s} {v:9.2e}   {100 * v * v / (quad * quad):5.1f} % of the quadrature sum")
    if len(sys.argv) > 1:
        json.dump(res_, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
