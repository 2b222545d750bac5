"""GPU parity tests, kernel by kernel, through the C ABI (librapflow.so) against the CPU oracle.

Tolerances (fp32 path): every comparison is against an fp64 evaluation of the oracle's formula on the
same fp32 inputs; the bound is a small multiple of fp32 round-off for the op (stated per test).
Integer/byte-exact where the op allows it (Euler update given v; segment tables)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from oracle import rap_oracle as O
from rap_amd import _lib, synthetic as S
from rap_amd.flow_model import workspace

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def stream(dev):
    return _lib.current_stream(dev)


def gemm(lib, dev, epi, A, W, C, M, N, K, bias=None, resid=None, anchor=None, emb=None, heads=0, ldc=None):
    rc = lib.rap_gemm_f32(epi, _lib.ptr(A), A.stride(0), _lib.ptr(W), W.stride(0), _lib.ptr(C), ldc if ldc else N, M, N, K,
                          _lib.ptr(bias), _lib.ptr(resid), resid.stride(0) if resid is not None else 0, _lib.ptr(anchor),
                          _lib.ptr(emb), heads, stream(dev))
    _lib.check(rc, "rap_gemm_f32")
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(1, 128, 32), (100, 128, 64), (128, 256, 128), (300, 512, 512), (1000, 512, 2048),
                                   (257, 1536, 512)])
def test_gemm_bias_matches_fp64(lib, dev, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    C = torch.full((M, N), float("nan"), device=dev)
    gemm(lib, dev, 0, A.to(dev), W.to(dev), C, M, N, K, bias=b.to(dev))
    err = (C.cpu().double() - ref).abs().max().item()
    # fp32 fma chain of length K on O(1) terms: error ~ sqrt(K) * 6e-8 * |a||w| ; bound 2e-5
    assert err < 2e-5, err
    # no bias
    C2 = torch.empty((M, N), device=dev)
    gemm(lib, dev, 0, A.to(dev), W.to(dev), C2, M, N, K)
    assert (C2.cpu().double() - (ref - b.double())).abs().max().item() < 2e-5


def test_gemm_is_transpose_safe_identity_check(lib, dev):
    """A = I with an asymmetric W catches swapped row/col in the MFMA C/D mapping."""
    K = N = 128
    A = torch.eye(128); W = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 1000.0
    C = torch.empty((128, N), device=dev)
    gemm(lib, dev, 0, A.to(dev), W.to(dev), C, 128, N, K)
    assert torch.equal(C.cpu(), W.T.contiguous())


def test_gemm_epilogues(lib, dev):
    g = torch.Generator().manual_seed(3)
    M, N, K = 333, 256, 128
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    base = A.double() @ W.double().T + b.double()
    Ad, Wd, bd = A.to(dev), W.to(dev), b.to(dev)
    # residual, in place (C aliases resid)
    h = torch.randn(M, N, generator=g)
    C = h.to(dev).clone()
    gemm(lib, dev, 1, Ad, Wd, C, M, N, K, bias=bd, resid=C)
    assert (C.cpu().double() - (h.double() + base)).abs().max().item() < 2e-5
    # SiLU
    C = torch.empty((M, N), device=dev)
    gemm(lib, dev, 2, Ad, Wd, C, M, N, K, bias=bd)
    assert (C.cpu().double() - F.silu(base)).abs().max().item() < 2e-5
    # anchor-embedding select
    anchor = (torch.rand(M, generator=g) < 0.4)
    emb = torch.randn(2, N, generator=g)
    C = torch.empty((M, N), device=dev)
    gemm(lib, dev, 5, Ad, Wd, C, M, N, K, bias=bd, anchor=anchor.to(dev).to(torch.uint8), emb=emb.to(dev))
    ref = base + torch.where(anchor[:, None], emb[1][None], emb[0][None]).double()
    assert (C.cpu().double() - ref).abs().max().item() < 2e-5


def test_gemm_geglu_with_interleaved_weights(lib, dev):
    g = torch.Generator().manual_seed(4)
    M, K, inner = 200, 128, 256
    A = torch.randn(M, K, generator=g); W = torch.randn(2 * inner, K, generator=g) / K ** 0.5; b = torch.randn(2 * inner, generator=g)
    u = A.double() @ W.double().T + b.double()
    ref = u[:, :inner] * F.gelu(u[:, inner:])
    Wd, bd = W.to(dev), b.to(dev)
    Wp, bp = torch.empty_like(Wd), torch.empty_like(bd)
    _lib.check(lib.rap_geglu_interleave(_lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(Wp), _lib.ptr(bp), inner, K, stream(dev)), "interleave")
    C = torch.empty((M, inner), device=dev)
    gemm(lib, dev, 3, A.to(dev), Wp, C, M, 2 * inner, K, bias=bp, ldc=inner)
    assert (C.cpu().double() - ref).abs().max().item() < 2e-5


def test_gemm_qkv_headmajor_scatter(lib, dev):
    g = torch.Generator().manual_seed(5)
    M, H, K = 150, 2, 128
    N = 3 * H * 64
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5
    ref = (A.double() @ W.double().T).reshape(M, 3, H, 64).permute(1, 2, 0, 3)   # [3][H][M][64]
    C = torch.empty((3, H, M, 64), device=dev)
    gemm(lib, dev, 4, A.to(dev), W.to(dev), C, M, N, K, heads=H)
    assert (C.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("epi,M,N,K", [(0, 16384, 2048, 512), (1, 65536, 512, 2048), (1, 65536, 512, 512), (2, 65536, 512, 512), (3, 16384, 2048, 512),
                                       (4, 32768, 1536, 512)])
def test_persistent_fp32_gemm_is_bit_identical_to_the_one_tile_per_block_kernel(lib, dev, epi, M, N, K):
    """Round 3: full-tile shapes (M % 256 == 0, at least 512 tiles of 256 x 256, K >= 256) run on the PERSISTENT 256 x 256 kernel;
    rap_set_tuning(12, 0) restores one tile per block.  Same MFMA order, same epilogue: bit-identical for every epilogue (0 bias,
    1 bias + residual in place, 2 bias + SiLU, 3 GEGLU, 4 head-major QKV with the fused qk-norm left to the model tests)."""
    g = torch.Generator(device=dev).manual_seed(200 + epi)
    A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    bias = torch.randn(N, device=dev, generator=g)
    outs = []
    try:
        for persistent in (1, 0):
            assert lib.rap_set_tuning(12, persistent) == 0
            if epi == 1:
                C = torch.randn(M, N, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
                gemm(lib, dev, 1, A, W, C, M, N, K, bias=bias, resid=C)
            elif epi == 3:
                C = torch.zeros(M, N // 2, device=dev); gemm(lib, dev, 3, A, W, C, M, N, K, bias=bias, ldc=N // 2)
            elif epi == 4:
                C = torch.zeros(3, 8, M, 64, device=dev); gemm(lib, dev, 4, A, W, C, M, N, K, bias=bias, heads=8)
            else:
                C = torch.zeros(M, N, device=dev); gemm(lib, dev, epi, A, W, C, M, N, K, bias=bias)
            outs.append(C.clone())
    finally:
        assert lib.rap_set_tuning(12, 1) == 0
    assert torch.equal(outs[0], outs[1])
    if epi in (0, 1):
        rows = torch.randint(0, M, (64,), generator=torch.Generator().manual_seed(3)).to(dev)
        ref = A[rows].double() @ W.double().T + bias.double()
        if epi == 1:
            ref = ref + torch.randn(M, N, device=dev, generator=torch.Generator(device=dev).manual_seed(5))[rows].double()
        assert (outs[0][rows].double() - ref).abs().max().item() < 2e-5 * max(1.0, K / 512)


def test_gemm_full_size_linearity_property(lib, dev):
    """BASELINE geometry (TP = 32*8192 rows, d = 512): GEMM(a1 + a2) == GEMM(a1) + GEMM(a2) to round-off,
    and a spot check of 64 random rows against fp64."""
    M, N, K = 32 * 8192, 512, 512
    g = torch.Generator(device=dev).manual_seed(0)
    A1 = torch.randn(M, K, device=dev, generator=g); A2 = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    C1, C2, C12 = (torch.empty((M, N), device=dev) for _ in range(3))
    gemm(lib, dev, 0, A1, W, C1, M, N, K); gemm(lib, dev, 0, A2, W, C2, M, N, K); gemm(lib, dev, 0, A1 + A2, W, C12, M, N, K)
    assert (C12 - (C1 + C2)).abs().max().item() < 5e-5
    rows = torch.randint(0, M, (64,), device=dev)
    ref = A1[rows].double() @ W.double().T
    assert (C1[rows].double() - ref).abs().max().item() < 2e-5


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def run_attention(lib, dev, qkv_thd, cu, bound=None):
    """qkv_thd: (T,3,H,64) fp32 CPU -> out (T,H,64) from the HIP kernel.  bound: optional (H,) logit bounds -> bounded softmax."""
    T, _, H, D = qkv_thd.shape
    bound_d = None if bound is None else bound.to(device=dev, dtype=torch.float32)
    hm = qkv_thd.permute(1, 2, 0, 3).contiguous().to(dev)     # [3][H][T][64]
    cu_d = cu.to(torch.int32).to(dev)
    out = torch.full((T, H * 64), float("nan"), device=dev)
    nseg = cu.numel() - 1
    ws = workspace(dev, lib.rap_attention_workspace_bytes(T, nseg))
    rc = lib.rap_attention_f32(_lib.ptr(hm), _lib.ptr(cu_d), nseg, _lib.ptr(out), T, H, _lib.ptr(bound_d), _lib.ptr(ws), ws.numel(), stream(dev))
    _lib.check(rc, "rap_attention_f32")
    torch.cuda.synchronize()
    return out.cpu().reshape(T, H, 64)


@pytest.mark.parametrize("bounded", [False, True], ids=["online-max", "bounded"])
@pytest.mark.parametrize("H", [1, 8])
def test_attention_ragged_segments_match_fp64(lib, dev, H, bounded):
    g = torch.Generator().manual_seed(11 + H)
    cu = torch.tensor([0, 1, 38, 38, 294, 600, 1624, 1657])     # lengths 1, 37, 0, 256, 306, 1024, 33
    T = int(cu[-1])
    qkv = torch.randn(T, 3, H, 64, generator=g)
    # qk-normalised magnitudes as in the network: |q| = |k| = 8
    qkv[:, 0] = F.normalize(qkv[:, 0], dim=-1) * 8
    qkv[:, 1] = F.normalize(qkv[:, 1], dim=-1) * 8
    ref = O.varlen_attention(qkv.double(), cu.to(torch.int32))
    # |q| = |k| = 8 -> q.k/8 <= 8: the bound the model derives from the qk-norm gains (bounded-softmax instantiation)
    out = run_attention(lib, dev, qkv, cu, bound=torch.full((H,), 8.01) if bounded else None)
    assert not torch.isnan(out).any()
    err = (out.double() - ref).abs().max().item()
    assert err < 5e-6, err     # softmax-weighted mean of |v| <~ 4: fp32 round-off is ~1e-6


def test_attention_single_token_segments_return_v(lib, dev):
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(5, 3, 8, 64, generator=g)
    cu = torch.tensor([0, 1, 2, 3, 4, 5])
    out = run_attention(lib, dev, qkv, cu)
    assert (out - qkv[:, 2]).abs().max().item() < 1e-6
    bound = (qkv[:, 0].norm(dim=-1).amax(0) * qkv[:, 1].norm(dim=-1).amax(0) / 8.0) * 1.01
    outb = run_attention(lib, dev, qkv, cu, bound=bound)
    assert (outb - qkv[:, 2]).abs().max().item() < 2e-6


def test_attention_bound_above_40_is_refused_loudly(lib, dev):
    """The bounded fp32 kernel drops the softmax offset, which is only safe for bounds <= 40: a head with a larger bound gets NaN
    outputs (never a silently overflowed softmax); the other heads are unaffected."""
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(300, 3, 2, 64, generator=g)
    cu = torch.tensor([0, 300])
    ref = O.varlen_attention(qkv.double(), cu.to(torch.int32))
    out = run_attention(lib, dev, qkv, cu, bound=torch.tensor([39.0, 41.0]))
    assert (out[:, 0].double() - ref[:, 0]).abs().max().item() < 5e-6
    assert torch.isnan(out[:, 1]).all()


def test_attention_sharp_softmax_and_late_maximum(lib, dev):
    """Forces the online-softmax rescale path: the dominant key sits in the LAST 64-key tile and logits are large."""
    g = torch.Generator().manual_seed(3)
    T, H = 700, 2
    qkv = torch.randn(T, 3, H, 64, generator=g)
    qkv[:, 0] = F.normalize(qkv[:, 0], dim=-1) * 30
    qkv[:, 1] = F.normalize(qkv[:, 1], dim=-1) * 30
    qkv[650, 1] = qkv[10, 0] / 30 * 60       # key 650 aligned with query 10, far larger logit than any other
    cu = torch.tensor([0, T])
    ref = O.varlen_attention(qkv.double(), cu.to(torch.int32))
    out = run_attention(lib, dev, qkv, cu)
    assert (out.double() - ref).abs().max().item() < 1e-4   # logits up to ~225: fp32 round-off of q.k scales with |q||k|
    assert (out[10].double() - qkv[650, 2].double()).abs().max().item() < 1e-4   # one-hot attention -> v of that key


def test_attention_full_size_properties(lib, dev):
    """BASELINE geometry: one sample of 2 x 4096 points, 8 heads, per-sample (L = 8192) and per-part (L = 4096)
    segmentations.  Size-independent properties: with v = 1 the output is exactly 1 (softmax rows sum to 1);
    and agreement with torch's fp32 SDPA on the GPU (independent implementation) at full size."""
    T, H = 8192, 8
    g = torch.Generator(device=dev).manual_seed(1)
    q = F.normalize(torch.randn(T, H, 64, device=dev, generator=g), dim=-1) * 8
    k = F.normalize(torch.randn(T, H, 64, device=dev, generator=g), dim=-1) * 8
    v = torch.randn(T, H, 64, device=dev, generator=g)
    for cu_list in ([0, 8192], [0, 4096, 8192]):
        cu = torch.tensor(cu_list, dtype=torch.int32, device=dev)
        nseg = len(cu_list) - 1
        ws = workspace(dev, lib.rap_attention_workspace_bytes(T, nseg))
        hm = torch.stack([q, k, v]).permute(0, 2, 1, 3).contiguous()     # [3][H][T][64]
        out = torch.empty((T, H * 64), device=dev)
        _lib.check(lib.rap_attention_f32(_lib.ptr(hm), _lib.ptr(cu), nseg, _lib.ptr(out), T, H, _lib.ptr(None), _lib.ptr(ws), ws.numel(),
                                         stream(dev)), "attn")
        ref = torch.empty((T, H, 64), device=dev)
        for s in range(nseg):
            a, b = cu_list[s], cu_list[s + 1]
            ref[a:b] = F.scaled_dot_product_attention(q[a:b].transpose(0, 1).double(), k[a:b].transpose(0, 1).double(),
                                                      v[a:b].transpose(0, 1).double()).transpose(0, 1).float()
        torch.cuda.synchronize()
        e_ref = (out.reshape(T, H, 64) - ref).abs().max().item()
        print(f"full-size attention {cu_list}: max abs err vs fp64 SDPA {e_ref:.2e}")
        assert e_ref < 5e-6, e_ref
        hm1 = torch.stack([q, k, torch.ones_like(v)]).permute(0, 2, 1, 3).contiguous()
        _lib.check(lib.rap_attention_f32(_lib.ptr(hm1), _lib.ptr(cu), nseg, _lib.ptr(out), T, H, _lib.ptr(None), _lib.ptr(ws), ws.numel(),
                                         stream(dev)), "attn")
        torch.cuda.synchronize()
        e_one = (out - 1.0).abs().max().item()
        print(f"full-size attention {cu_list}: v = 1 -> max |out - 1| {e_one:.2e}")
        # numerator (MFMA fma chain) and denominator (VALU sum) associate 8192 fp32 terms differently
        assert e_one < 2e-5, e_one


# ---------------------------------------------------------------------------------------------
# normalisation / embedding / adaLN
# ---------------------------------------------------------------------------------------------
def test_layernorm_modulate_and_affine(lib, dev):
    g = torch.Generator().manual_seed(6)
    TP, d, B = 1001, 512, 3
    x = torch.randn(TP, d, generator=g) * 3 + 0.5
    mod = torch.randn(B, 4, 2 * d, generator=g)            # rows = samples, 4 LNs per row
    cu = torch.tensor([0, 400, 401, 1001])
    tok = torch.repeat_interleave(torch.arange(B), cu[1:] - cu[:-1]).to(torch.int32)
    j = 2
    scale, shift = mod[:, j, :d], mod[:, j, d:]
    ref = F.layer_norm(x.double(), (d,), eps=1e-5) * (1 + scale.double()[tok.long()]) + shift.double()[tok.long()]
    out = torch.empty((TP, d), device=dev)
    modd, xd, tokd = mod.to(dev), x.to(dev), tok.to(dev)     # keep every device tensor alive across the async launch
    rc = lib.rap_layernorm_mod(_lib.ptr(xd), _lib.ptr(out), TP, d, ctypes.c_void_p(modd.data_ptr() + j * 2 * d * 4),
                               4 * 2 * d, _lib.ptr(tokd), stream(dev))
    _lib.check(rc, "ln_mod"); torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 1e-5
    # uniform row (token_row = NULL -> row 0)
    ref0 = F.layer_norm(x.double(), (d,), eps=1e-5) * (1 + scale.double()[0]) + shift.double()[0]
    rc = lib.rap_layernorm_mod(_lib.ptr(xd), _lib.ptr(out), TP, d, ctypes.c_void_p(modd.data_ptr() + j * 2 * d * 4),
                               0, _lib.ptr(None), stream(dev))
    _lib.check(rc, "ln_mod"); torch.cuda.synchronize()
    assert (out.cpu().double() - ref0).abs().max().item() < 1e-5
    # affine
    gain, bias = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g)
    ref = F.layer_norm(x.double(), (d,), gain.double(), bias.double(), eps=1e-5)
    gaind, biasd = gain.to(dev), bias.to(dev)
    rc = lib.rap_layernorm_affine(_lib.ptr(xd), _lib.ptr(out), TP, d, _lib.ptr(gaind), _lib.ptr(biasd), stream(dev))
    _lib.check(rc, "ln_affine"); torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 1e-5


def test_qknorm(lib, dev):
    g = torch.Generator().manual_seed(7)
    TP, H = 333, 8
    qkv = torch.randn(3, H, TP, 64, generator=g)
    gq, gk = torch.rand(H, 64, generator=g) + 0.5, torch.rand(H, 64, generator=g) + 0.5
    qkv[0, 3, 5] = 0.0    # an all-zero row exercises the eps clamp
    ref = qkv.double().clone()
    ref[0] = O.multi_head_rms_norm(qkv[0].double().permute(1, 0, 2), gq.double()).permute(1, 0, 2)
    ref[1] = O.multi_head_rms_norm(qkv[1].double().permute(1, 0, 2), gk.double()).permute(1, 0, 2)
    buf = qkv.to(dev).contiguous()
    gqd, gkd = gq.to(dev), gk.to(dev)
    _lib.check(lib.rap_qknorm(_lib.ptr(buf), TP, H, _lib.ptr(gqd), _lib.ptr(gkd), stream(dev)), "qknorm")
    torch.cuda.synchronize()
    assert (buf.cpu().double() - ref).abs().max().item() < 1e-5
    assert torch.equal(buf.cpu()[2], qkv[2])     # v untouched


def test_posenc_feature_builders(lib, dev):
    g = torch.Generator().manual_seed(8)
    TP, B, Fd = 777, 3, 32
    x = torch.randn(TP, 3, generator=g) * 1.5           # |x| up to ~6 -> arguments up to ~3000 rad
    cond = (torch.rand(TP, 3, generator=g) - 0.5) * 1.4
    feat = F.normalize(torch.randn(TP, Fd, generator=g), dim=1)
    scales = torch.rand(B, generator=g) * 45 + 5
    cu = torch.tensor([0, 300, 301, 777], dtype=torch.int32)
    tok = torch.empty(TP, dtype=torch.int32, device=dev)
    cud, xd, condd, scd, featd = cu.to(dev), x.to(dev), cond.to(dev), scales.to(dev), feat.to(dev)
    _lib.check(lib.rap_token_sample(_lib.ptr(cud), B, _lib.ptr(tok), stream(dev)), "token_sample")
    torch.cuda.synchronize()
    tok_ref = torch.repeat_interleave(torch.arange(B), (cu[1:] - cu[:-1]).long()).to(torch.int32)
    assert torch.equal(tok.cpu(), tok_ref)
    ax = torch.empty((TP, 64), device=dev)
    _lib.check(lib.rap_posenc_x(_lib.ptr(xd), _lib.ptr(ax), TP, stream(dev)), "posenc_x")
    ast = torch.empty((TP, 128), device=dev)
    _lib.check(lib.rap_posenc_static(_lib.ptr(condd), _lib.ptr(scd), _lib.ptr(tok), _lib.ptr(featd),
                                     Fd, _lib.ptr(ast), TP, stream(dev)), "posenc_static")
    torch.cuda.synchronize()
    ref_x = O.posenc(x.double())                       # fp64 sin/cos of the exact fp32 argument 2^k * x
    assert (ax.cpu()[:, :63].double() - ref_x).abs().max().item() < 1e-6
    assert torch.equal(ax.cpu()[:, 63], torch.zeros(TP))
    sc_pt = scales[tok_ref.long()]
    ref_s = torch.cat([O.posenc(cond.double()), O.posenc(sc_pt.double().unsqueeze(-1)), feat.double()], dim=-1)
    assert (ast.cpu()[:, :116].double() - ref_s).abs().max().item() < 1e-6
    assert torch.equal(ast.cpu()[:, 116:], torch.zeros(TP, 12))


@pytest.fixture(scope="module")
def small_model(dev):
    import rap_amd
    cfg = dict(S.RAP_12); cfg["num_layers"] = 2
    sd = S.make_weights(cfg, 0)
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=2, num_heads=8, local_feat_dim=32)
    m.load_state_dict(sd)
    m.to(dev)
    return cfg, sd, m


def test_adaln_table_matches_oracle(lib, dev, small_model):
    cfg, sd, m = small_model
    rows, L, d = 5, cfg["num_layers"], cfg["embed_dim"]
    t = torch.tensor([1.0, 0.95, 0.5, 0.05, 0.3])
    out = torch.empty((rows, 2 * L, 2 * d), device=dev)
    scratch = torch.empty(rows * (256 + 4 * L * d), device=dev)
    td = t.to(dev)
    _lib.check(lib.rap_adaln_table(m._handle, _lib.ptr(td), rows, _lib.ptr(scratch), _lib.ptr(out), stream(dev)), "adaln")
    torch.cuda.synchronize()
    sd64 = {k: v.double() for k, v in sd.items()}
    for i in range(L):
        for a, which in enumerate(("self", "global")):
            scale, shift = O.adaln_scale_shift(sd64, f"transformer_layers.{i}.{which}_prenorm.", t)
            ref = torch.cat([scale, shift], dim=-1)
            got = out.cpu()[:, 2 * i + a].double()
            assert (got - ref).abs().max().item() < 2e-6, (i, which)


# ---------------------------------------------------------------------------------------------
# sampler ring: Euler, Procrustes
# ---------------------------------------------------------------------------------------------
def test_euler_step_is_bit_exact(lib, dev):
    g = torch.Generator().manual_seed(9)
    n = 3 * 12345
    x, v = torch.randn(n, generator=g), torch.randn(n, generator=g)
    for step in (0, 3, 19):
        dt = 1.0 / 20
        t = 1 - step * dt
        x_next_ref, x0_ref = O.euler_step(x, t, dt, lambda a, b: v)    # fp32 tensor ops, python-double scalars
        x0 = torch.empty(n, device=dev); xn = torch.empty(n, device=dev); tr = torch.empty(n, device=dev)
        xd, vd = x.to(dev), v.to(dev)
        _lib.check(lib.rap_euler_step(_lib.ptr(xd), _lib.ptr(vd), t, dt, _lib.ptr(x0), _lib.ptr(xn), _lib.ptr(tr),
                                      n, stream(dev)), "euler")
        torch.cuda.synchronize()
        assert torch.equal(x0.cpu(), x0_ref) and torch.equal(xn.cpu(), x_next_ref) and torch.equal(tr.cpu(), x_next_ref)


def test_procrustes_fit_and_rigidify_match_oracle(dev):
    import rap_amd
    inp = S.make_inputs([[37, 64, 100], [50, 129], [4096, 3000]], seed=5)
    g = torch.Generator().manual_seed(10)
    pred = inp["pointclouds_gt"] + 0.05 * torch.randn(inp["pointclouds_gt"].shape, generator=g)
    cond, ppp, cu = inp["pointclouds"], inp["points_per_part"], inp["cu_seqlens"]
    R, t = rap_amd.fit_transformations(cond.to(dev), pred.to(dev), ppp, cu)
    Rr, tr = O.fit_transformations(cond.double(), pred.double(), ppp, cu)
    assert (R.cpu().double() - Rr).abs().max().item() < 2e-6
    assert (t.cpu().double() - tr).abs().max().item() < 2e-6
    assert torch.equal(R.cpu()[1, 2], torch.zeros(3, 3)) and torch.equal(t.cpu()[1, 2], torch.zeros(3))   # empty part
    rig = rap_amd.rigidify_prediction_with_procrustes(pred.to(dev), cond.to(dev), ppp, cu)
    rig_ref = O.rigidify_prediction_with_procrustes(pred.double(), cond.double(), ppp, cu)
    assert (rig.cpu().double() - rig_ref).abs().max().item() < 2e-6
    # exact recovery at the BASELINE part size: tgt = src R0^T + t0
    R0 = S._random_rotation(torch.Generator().manual_seed(1)).float()
    t0 = torch.tensor([0.3, -0.1, 0.2])
    src = inp["pointclouds"][-7096:-3000]
    Rs, ts = rap_amd.solve_procrustes(src.to(dev), (src @ R0.T + t0).to(dev))
    assert (Rs.cpu() - R0).abs().max().item() < 2e-6 and (ts.cpu() - t0).abs().max().item() < 2e-6


def test_procrustes_empty_part_in_the_middle(dev):
    """The reference mis-indexes when an empty part precedes a non-empty one (procrustes.py:79 indexes the compacted
    list); the kernels work from offsets.  Check against a direct per-part solve."""
    import rap_amd
    ppp = torch.tensor([[50, 0, 70]])
    g = torch.Generator().manual_seed(12)
    src = torch.randn(120, 3, generator=g); tgt = torch.randn(120, 3, generator=g)
    cu = torch.tensor([0, 120])
    R, t = rap_amd.fit_transformations(src.to(dev), tgt.to(dev), ppp, cu)
    Ra, ta = O.solve_procrustes(src[:50].double(), tgt[:50].double())
    Rb, tb = O.solve_procrustes(src[50:].double(), tgt[50:].double())
    assert (R.cpu()[0, 0].double() - Ra).abs().max().item() < 2e-6 and (R.cpu()[0, 2].double() - Rb).abs().max().item() < 2e-6
    assert (t.cpu()[0, 2].double() - tb).abs().max().item() < 2e-6 and torch.equal(R.cpu()[0, 1], torch.zeros(3, 3))


def test_attention_large_scan_geometry_one_head(lib, dev):
    """BASELINE configs[4] geometry: 2 x 32768 points -> per-part L = 32768, per-sample L = 65536 (the longest
    segments the path is specified for).  One head, checked in full against a chunked fp64 softmax on the GPU."""
    T, H = 65536, 1
    g = torch.Generator(device=dev).manual_seed(4)
    q = F.normalize(torch.randn(T, 64, device=dev, generator=g), dim=-1) * 8
    k = F.normalize(torch.randn(T, 64, device=dev, generator=g), dim=-1) * 8
    v = torch.randn(T, 64, device=dev, generator=g)
    hm = torch.stack([q, k, v]).unsqueeze(1).contiguous()          # [3][1][T][64]
    for cu_list in ([0, 65536], [0, 32768, 65536]):
        cu = torch.tensor(cu_list, dtype=torch.int32, device=dev)
        nseg = len(cu_list) - 1
        ws = workspace(dev, lib.rap_attention_workspace_bytes(T, nseg))
        out = torch.empty((T, 64), device=dev)
        _lib.check(lib.rap_attention_f32(_lib.ptr(hm), _lib.ptr(cu), nseg, _lib.ptr(out), T, H, _lib.ptr(None), _lib.ptr(ws), ws.numel(),
                                         stream(dev)), "attn")
        torch.cuda.synchronize()
        worst = 0.0
        for s in range(nseg):
            a, b = cu_list[s], cu_list[s + 1]
            kd, vd = k[a:b].double(), v[a:b].double()
            for r0 in range(a, b, 4096):
                sc = (q[r0:r0 + 4096].double() @ kd.T) * 0.125
                ref = torch.softmax(sc, dim=-1) @ vd
                worst = max(worst, (out[r0:r0 + 4096].double() - ref).abs().max().item())
        print(f"large-scan attention {cu_list}: max abs err vs fp64 {worst:.2e}")
        assert worst < 5e-6, worst


# ---------------------------------------------------------------------------------------------
# input side of the boundary: device-side _transform + collate
# ---------------------------------------------------------------------------------------------
def _collate_fixture():
    import numpy as np
    from test_oracle import COLLATE_KEYS, load_collate_golden
    z, samples = load_collate_golden()
    return z, samples, COLLATE_KEYS


def test_transform_and_collate_matches_reference_golden(dev):
    """rap_collate_transform vs the reference's own PointCloudDataset._transform + variable_collate_fn (fixture written by the
    unmodified modules): same numpy seed -> same within-part shuffles -> every key of the collated batch."""
    import numpy as np
    import rap_amd
    z, samples, keys = _collate_fixture()
    np.random.seed(int(z["numpy_seed"]))
    got = rap_amd.transform_and_collate(samples, int(z["max_parts"]), device=dev)
    for k in keys:
        ref = torch.from_numpy(z["ref_" + k])
        g = got[k].cpu()
        assert g.shape == ref.shape and g.dtype == ref.dtype, (k, g.shape, ref.shape, g.dtype, ref.dtype)
        if ref.dtype.is_floating_point:
            err = (g - ref).abs().max().item()
            assert err <= 2e-7 * max(1.0, ref.abs().max().item()), (k, err)      # fp64 arithmetic on both sides, one fp32 rounding
        else:
            assert torch.equal(g, ref), k
    assert got["num_parts"] == [3, 2, 4]


def test_transform_and_collate_fp32_input_large_batch_properties(dev):
    """BASELINE-size batch (32 pairs x 2 x 4096) from fp32 scans: the reference's own invariant (dataset.py:927-932:
    cond @ R^T + t == gt for non-anchor parts, anchor cond == gt + gt_trans), unit extent of the anchor, the collated offsets; and
    the output drives the sampler unchanged (keys of the boundary dict)."""
    import rap_amd
    g = torch.Generator().manual_seed(3)
    samples = []
    for b in range(32):
        c = torch.randn(3, generator=g) * 300
        samples.append({"parts": [(c + torch.randn(4096, 3, generator=g) * torch.tensor([9.0, 6.0, 2.0])).to(dev),
                                  (c + 5 + torch.randn(4096, 3, generator=g) * torch.tensor([9.0, 6.0, 2.0])).to(dev)],
                        "features": [torch.nn.functional.normalize(torch.randn(4096, 32, generator=g), dim=1).to(dev) for _ in range(2)]})
    out = rap_amd.transform_and_collate(samples, 2, shuffle=False)
    TP = 32 * 8192
    assert out["pointclouds"].shape == (TP, 3) and out["cu_seqlens"].tolist() == list(range(0, TP + 1, 8192))
    cond, gt = out["pointclouds"].view(32, 2, 4096, 3), out["pointclouds_gt"].view(32, 2, 4096, 3)
    t = out["translations"]
    anchor = out["anchor_parts"]
    assert anchor.sum(1).tolist() == [1] * 32 and bool(anchor[:, 0].all())                 # equal counts: the first part is the anchor
    rec = cond[:, 1] + t[:, 1, None, :]                                                      # R = I
    assert (rec - gt[:, 1]).abs().max().item() < 2e-6
    assert (cond[:, 0] - (gt[:, 0] - t[:, 0, None, :])).abs().max().item() < 2e-6          # anchor: t = -gt_trans
    assert (cond[:, 0].abs().amax(dim=(1, 2)) * 1.5 - 1.0).abs().max().item() < 1e-5       # anchor extent: max |coord| * 1.5 = 1
    assert cond[:, 0].mean(1).abs().max().item() < 1e-5 and cond[:, 1].mean(1).abs().max().item() < 1e-5
    assert gt.reshape(32, -1, 3).mean(1).abs().max().item() < 1e-5
    assert torch.equal(out["anchor_indices"].view(32, 2, 4096)[:, 0], torch.ones(32, 4096, dtype=torch.bool, device=dev))
    with pytest.raises(ValueError):
        rap_amd.transform_and_collate(samples[:2], 2, order=torch.full((2 * 8192,), 5000, dtype=torch.int64))


@pytest.mark.parametrize("nseg", [1, 7, 300, 1500, 5000])
def test_attention_worklist_is_longest_segment_first_and_complete(lib, dev, nseg):
    """Round 4: rap_sample's attention work lists are emitted longest-segment-first (LPT: a block's work is its segment's length).
    Every (segment, 256-query block) appears exactly once, in non-increasing segment length (ties in segment order), the tail is
    zero items, and without scratch the list is in segment order -- incl. empty segments and more segments than threads / LDS tile."""
    g = torch.Generator().manual_seed(nseg)
    lens = torch.randint(0, 3000, (nseg,), generator=g)
    lens[torch.rand(nseg, generator=g) < 0.1] = 0
    if nseg > 2:
        lens[1] = 40000; lens[nseg - 1] = 40000                      # a tie between the two longest
    cu = torch.zeros(nseg + 1, dtype=torch.int32); cu[1:] = torch.cumsum(lens, 0).to(torch.int32)
    n_items = int(((lens + 255) // 256).sum())
    max_items = n_items + 37
    cud = cu.to(dev)
    expect = {(int(cu[s]), int(lens[s]), q0) for s in range(nseg) for q0 in range(0, int(lens[s]), 256)}
    for sorted_ in (True, False):
        items = torch.full((max_items, 4), -1, dtype=torch.int32, device=dev)
        ws = torch.empty(max(nseg, 1), dtype=torch.int32, device=dev) if sorted_ else None
        _lib.check(lib.rap_build_attention_worklist(_lib.ptr(cud), nseg, 256, _lib.ptr(items), max_items, _lib.ptr(ws), stream(dev)), "worklist")
        torch.cuda.synchronize()
        it = items.cpu()
        assert bool((it[n_items:] == 0).all()) and bool((it[:, 3] == 0).all())
        got = [(int(a), int(b), int(c)) for a, b, c, _ in it[:n_items].tolist()]
        assert len(set(got)) == n_items and set(got) == expect
        seg_len = it[:n_items, 1]
        if sorted_:
            assert bool((seg_len[1:] <= seg_len[:-1]).all())
            starts = it[:n_items, 0]
            same = seg_len[1:] == seg_len[:-1]
            assert bool((starts[1:][same] >= starts[:-1][same]).all())       # ties: segment order, a segment's items adjacent
        else:
            assert [x[0] for x in got] == sorted(x[0] for x in got)
