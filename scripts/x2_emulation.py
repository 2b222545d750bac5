"""CPU emulation of the split-precision ("float32x2") arithmetic, run through the pinned oracle (round 5, design check).

Every transformer-block contraction (qkv / out / ff1 / ff2 GEMMs, q k^T, p v) is evaluated as the 16-bit matrix pipes would:
operands split into an fp16 head and an fp16 tail, x = hi + lo, the three products hi*hi + hi*lo + lo*hi accumulated in fp32
(a product of two fp16 values is exact in fp32, so a torch fp32 matmul of the fp16-valued planes is the MFMA up to summation
order).  LayerNorm, qk-norm, softmax state, GEGLU, residual stream, embedding, head, Euler and Procrustes stay fp32 -- as in the
HIP path.  Compared: fp64 oracle (truth), fp32 oracle (what the reference computes), single-plane fp16 / bf16, and the split
forms with / without the power-of-two weight scale and with fp16 subnormal tails flushed (the pessimistic hardware model).

    python scripts/x2_emulation.py [--layers 2] [--steps 5]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import rap_oracle as O  # noqa: E402
from rap_amd import synthetic  # noqa: E402


def split_planes(x: torch.Tensor, mode: str, flush: bool = False):
    if mode == "f16x2":
        hi = x.clamp(-65504.0, 65504.0).to(torch.float16)
        lo = (x - hi.float()).to(torch.float16)
        if flush:      # subnormal fp16 inputs flushed to zero by the matrix pipe (pessimistic model)
            lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
            hi = torch.where(hi.abs() < 2.0 ** -14, torch.zeros_like(hi), hi)
        return [hi.float(), lo.float()]
    if mode == "bf16x3":
        p0 = x.to(torch.bfloat16); r = x - p0.float()
        p1 = r.to(torch.bfloat16); r = r - p1.float()
        p2 = r.to(torch.bfloat16)
        return [p0.float(), p1.float(), p2.float()]
    if mode == "f16":
        return [x.to(torch.float16).float()]
    if mode == "bf16":
        return [x.to(torch.bfloat16).float()]
    raise ValueError(mode)


def split_matmul(a: torch.Tensor, b_t: torch.Tensor, mode: str, flush: bool = False, wscale: bool = False):
    """a (M,K) @ b_t (N,K)^T with split operands; b_t optionally pre-scaled by a power of two (weights)."""
    s = 1.0
    if wscale:
        s = 2.0 ** (12 - int(torch.ceil(torch.log2(b_t.abs().max())).item()))
        b_t = b_t * s
    ap, bp = split_planes(a, mode, flush), split_planes(b_t, mode, flush)
    n = len(ap)
    acc = None
    for i in range(n):
        for j in range(n):
            if i + j <= n - 1:
                term = ap[i] @ bp[j].transpose(-1, -2)
                acc = term if acc is None else acc + term
    return acc / s


def patched(mode: str, flush: bool, wscale: bool):
    """Returns (attention_block, feed_forward) replacements for the oracle's."""

    def lin(x, w, b=None):
        y = split_matmul(x, w, mode, flush, wscale)
        return y if b is None else y + b

    def attn(qkv, cu):
        T, _, H, D = qkv.shape
        out = torch.zeros((T, H, D), dtype=qkv.dtype)
        cu = cu.tolist()
        for s in range(len(cu) - 1):
            a, b = cu[s], cu[s + 1]
            if b == a:
                continue
            q, k, v = (qkv[a:b, i].transpose(0, 1) for i in range(3))     # (H, L, D)
            sc = split_matmul(q, k, mode, flush) * (D ** -0.5)
            sc = sc - sc.amax(dim=-1, keepdim=True)
            p = torch.exp(sc)
            l = p.sum(-1, keepdim=True)
            o = split_matmul(p, v.transpose(1, 2), mode, flush) / l
            out[a:b] = o.transpose(0, 1)
        return out

    def attention_block(sd, prefix, which, x, cu_seqlens, H):
        T, d = x.shape
        qkv = lin(x, sd[prefix + f"{which}_qkv_proj.weight"]).reshape(T, 3, H, d // H)
        q, k, v = qkv.unbind(dim=1)
        q = O.multi_head_rms_norm(q, sd[prefix + f"{which}_q_norm.gamma"])
        k = O.multi_head_rms_norm(k, sd[prefix + f"{which}_k_norm.gamma"])
        out = attn(torch.stack([q, k, v], dim=1), cu_seqlens).reshape(T, d)
        return lin(out, sd[prefix + f"{which}_out_proj.weight"], sd[prefix + f"{which}_out_proj.bias"])

    def feed_forward(sd, prefix, x):
        u = lin(x, sd[prefix + "ff.net.0.proj.weight"], sd[prefix + "ff.net.0.proj.bias"])
        h, g = u.chunk(2, dim=-1)
        return lin(h * F.gelu(g), sd[prefix + "ff.net.2.weight"], sd[prefix + "ff.net.2.bias"])

    return attention_block, feed_forward


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--points", type=int, default=384)
    a = ap.parse_args()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = dict(embed_dim=512, num_layers=a.layers, num_heads=8, local_feat_dim=32)
    sd = synthetic.make_weights(cfg, seed=0)
    inputs = synthetic.make_inputs([[a.points, a.points - 100], [a.points // 2, a.points // 3, 50]], seed=1234)
    ref64 = O.sample(sd, cfg, inputs, a.steps, True, dtype=torch.float64)
    ref32 = O.sample(sd, cfg, inputs, a.steps, True, dtype=torch.float32)

    def dev(res):
        x = res["end_point_trajectory"][-1].double()
        return {"cloud_vs_f64": float((x - ref64["end_point_trajectory"][-1]).abs().max()),
                "cloud_vs_f32": float((x - ref32["end_point_trajectory"][-1].double()).abs().max()),
                "R_frob_vs_f64": float((res["R"].double() - ref64["R"]).flatten(-2).norm(dim=-1).max())}

    rows = {"fp32_oracle": dev(ref32)}
    keep = (O.attention_block, O.feed_forward)
    for name, mode, flush, wscale in [("f16", "f16", False, False), ("bf16", "bf16", False, False),
                                      ("f16x2", "f16x2", False, False), ("f16x2_wscale", "f16x2", False, True),
                                      ("f16x2_flush", "f16x2", True, False), ("f16x2_flush_wscale", "f16x2", True, True),
                                      ("bf16x3", "bf16x3", False, False)]:
        O.attention_block, O.feed_forward = patched(mode, flush, wscale)
        try:
            rows[name] = dev(O.sample(sd, cfg, inputs, a.steps, True, dtype=torch.float32))
        finally:
            O.attention_block, O.feed_forward = keep
        print(json.dumps({name: rows[name]}), flush=True)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
