"""GPU tests of the SPLIT-PRECISION kernels (compute dtype 3, "float32x2", round 5) through the C ABI: fp32-accurate contractions on
the fp16 matrix pipe -- every operand an fp16 head + an fp16 tail in the paired layout, three products per contraction.

What is compared with what:
  * pack / unpack: bit-exact against torch's own fp16 round-to-nearest (head = rn16(clip(x)), tail = rn16(clip(x) - head)), layout
    checked element by element;
  * GEMMs, attention, LayerNorm: against an fp64 evaluation of the formula on the ORIGINAL fp32 operands -- the claim of the mode is
    "fp32-accurate", so the yardstick is the error an fp32 evaluation (torch matmul / the exact-fp32 HIP kernel) makes on the same
    inputs, printed next to it; bounds are fp32-class (a few 1e-7 relative to the result's scale), three to four orders of magnitude
    below what one fp16 plane gives;
  * model level: tests/test_sample_gpu.py, test_headline_gpu.py and test_fullconfig_gpu.py run this mode against the reference's
    golden fixtures and the device-side checker under the SAME asserts as the exact-fp32 path.
"""
import pytest
import torch
import torch.nn.functional as F

from rap_amd import _lib
from rap_amd.flow_model import workspace

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _split_precision_on_small_calls():
    """By default a model in compute dtype "float32x2" runs calls below 1 024 token rows on the exact-fp32 kernels (tuning key 17: the
    few-token forms of the fp32 path are faster there).  The fixtures of this file ARE small: force the split-precision kernels so that
    they are what is tested (ragged row counts, one-tile launches); test_x2_small_calls_fall_back_to_exact_fp32 covers the default."""
    from rap_amd import _lib as _l
    lib = _l.load()
    assert lib.rap_set_tuning(17, 0) == 0
    yield
    assert lib.rap_set_tuning(17, 1024) == 0


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def stream(dev):
    return _lib.current_stream(dev)


def vt_pos(t):
    return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)


def x2_col(k):
    return ((k >> 5) << 6) | (k & 31)


def split_ref(x):
    """torch model of half.h x2_split*: (head, tail) fp16 tensors of an fp32 tensor."""
    s = torch.where(torch.isnan(x), x, x.clamp(-65504.0, 65504.0))
    hi = s.to(torch.float16)
    lo = (s - hi.float()).to(torch.float16)
    return hi, lo


def pack_ref(x):
    """(rows, K) fp32 -> (rows, 2K) fp16 in the paired layout, on the CPU."""
    rows, K = x.shape
    hi, lo = split_ref(x)
    out = torch.empty(rows, 2 * K, dtype=torch.float16)
    k = torch.arange(K)
    out[:, x2_col(k)] = hi
    out[:, x2_col(k) + 32] = lo
    return out


def pack_dev(lib, dev, x, scale=1.0):
    xd = x.to(dev).contiguous()
    out = torch.empty(x.shape[0], 2 * x.shape[1], dtype=torch.float16, device=dev)
    _lib.check(lib.rap_x2_pack(_lib.ptr(xd), x.shape[1], x.shape[0], x.shape[1], float(scale), _lib.ptr(out), stream(dev)), "rap_x2_pack")
    torch.cuda.synchronize()
    return out


def unpack_dev(lib, dev, p, cols, inv_scale=1.0):
    out = torch.empty(p.shape[0], cols, dtype=torch.float32, device=dev)
    _lib.check(lib.rap_x2_unpack(_lib.ptr(p), p.shape[0], cols, float(inv_scale), _lib.ptr(out), stream(dev)), "rap_x2_unpack")
    torch.cuda.synchronize()
    return out.cpu()


def unpack_ref(p, cols):
    k = torch.arange(cols)
    return p[..., x2_col(k)].double() + p[..., x2_col(k) + 32].double()


def weight_scale(W):
    """the library's per-tensor power of two: max|W| * 2^e in [2^11, 2^12)"""
    import math
    m = float(W.abs().max())
    e = 12 - math.frexp(m)[1] if m > 0 else 0
    return 2.0 ** e


def x2_gemm(lib, dev, epi, A, W, C, M, N, Kp, ldc, bias=None, resid=None, acc_scale=1.0, heads=0, gq=None, gk=None, q_mul=8.0, vt=None,
            vt_nblk=0):
    rc = lib.rap_x2_gemm(epi, _lib.ptr(A), A.stride(0), _lib.ptr(W), W.stride(0), _lib.ptr(C), ldc, M, N, Kp, _lib.ptr(bias), _lib.ptr(resid),
                         resid.stride(0) if resid is not None else 0, float(acc_scale), heads, _lib.ptr(gq), _lib.ptr(gk), float(q_mul),
                         _lib.ptr(vt), vt_nblk, stream(dev))
    _lib.check(rc, "rap_x2_gemm")
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------
# pack / unpack
# ---------------------------------------------------------------------------------------------
def test_pack_is_head_plus_tail_in_the_paired_layout(lib, dev):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(37, 96, generator=g) * 10.0 ** torch.randint(-6, 4, (37, 96), generator=g).float()
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, 1.00048828125, 65504.0, 1e6, -3e5, 6.1e-5])
    x[1, 0] = float("nan")
    got = pack_dev(lib, dev, x).cpu()
    want = pack_ref(x)
    assert torch.equal(got.view(torch.int16)[2:], want.view(torch.int16)[2:])
    assert torch.equal(got[0].view(torch.int16), want[0].view(torch.int16))
    assert torch.isnan(got[1, 0]) and torch.isnan(got[1, 32]) and torch.equal(got[1, 1:32], want[1, 1:32])      # NaN stays NaN, in both planes
    assert float(got[0, 5]) == 65504.0 and float(got[0, 37]) == 0.0          # |x| > 65504 clips to the fp16 range: finite head, zero tail
    # head + tail reproduces x to 2^-22 relative (normal tails) or 2^-25 absolute (subnormal tails)
    back = unpack_dev(lib, dev, got.to(dev), 96)
    ok = torch.isfinite(x) & (x.abs() <= 65504)
    err = (back.double() - x.double()).abs()[ok]
    assert bool((err <= x.double().abs()[ok] * 2.0 ** -22 + 2.0 ** -25).all()), err.max()
    # scaled pack (weights): split(x * 2^e)
    xs = torch.randn(64, 64, generator=g) * 0.02
    sc = weight_scale(xs)
    assert 2048 <= float(xs.abs().max()) * sc < 4096
    assert torch.equal(pack_dev(lib, dev, xs, sc).cpu().view(torch.int16), pack_ref(xs * sc).view(torch.int16))


# ---------------------------------------------------------------------------------------------
# GEMMs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 512, 512), (1024, 512, 2048), (256, 256, 64), (40001, 512, 512), (65536, 512, 512), (131072, 256, 128)],
                         ids=["ragged-M-128x128", "ff2-shape-128x128", "one-k-tile-pair", "ragged-M-one-256x256-tile-per-block", "persistent-512-tiles",
                              "persistent-short-K"])
def test_x2_gemm_residual_epilogue_is_fp32_accurate(lib, dev, M, N, K):
    """C = resid + A W^T + bias (out-projection / ff2 form) on paired operands with a scaled weight plane, against fp64 on the original
    fp32 operands; torch's own fp32 matmul on the same operands is the yardstick."""
    g = torch.Generator().manual_seed(2)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) * 0.02
    bias = torch.randn(N, generator=g) * 0.1; resid = torch.randn(M, N, generator=g)
    sc = weight_scale(W)
    Ap, Wp = pack_dev(lib, dev, A), pack_dev(lib, dev, W, sc)
    C = resid.to(dev).clone()
    x2_gemm(lib, dev, 1, Ap, Wp, C, M, N, 2 * K, N, bias=bias.to(dev), resid=C, acc_scale=1.0 / sc)
    rows = slice(0, M) if M <= 4096 else torch.randint(0, M, (2048,), generator=g)
    ref = A[rows].double() @ W.double().T + bias.double() + resid[rows].double()
    scale = float((A[rows].double().abs() @ W.double().abs().T).max())             # sum |a w|: what an fp32 rounding is relative to
    err = float((C.cpu()[rows].double() - ref).abs().max()) / scale
    err32 = float(((A[rows] @ W.T + bias + resid[rows]).double() - ref).abs().max()) / scale
    print(f"x2 gemm {M}x{N}x{K}: max err / sum|aw| = {err:.2e}  (torch fp32 matmul: {err32:.2e})")
    assert err < 4e-7, (err, err32)                 # fp32 class (2^-24 = 6e-8 per rounding); ONE fp16 plane would be ~2e-4
    assert not torch.isnan(C).any()


def test_x2_gemm_geglu_epilogue(lib, dev):
    g = torch.Generator().manual_seed(3)
    M, K, inner = 700, 512, 1024
    A = torch.randn(M, K, generator=g); W = torch.randn(2 * inner, K, generator=g) * 0.03; b = torch.randn(2 * inner, generator=g) * 0.1
    Wd, bd = W.to(dev), b.to(dev)
    Wi, bi = torch.empty_like(Wd), torch.empty_like(bd)
    _lib.check(lib.rap_geglu_interleave(_lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(Wi), _lib.ptr(bi), inner, K, stream(dev)), "interleave")
    sc = weight_scale(W)
    Ap, Wp = pack_dev(lib, dev, A), pack_dev(lib, dev, Wi.cpu(), sc)
    C = torch.full((M, 2 * inner), float("nan"), dtype=torch.float16, device=dev)
    x2_gemm(lib, dev, 3, Ap, Wp, C, M, 2 * inner, 2 * K, 2 * inner, bias=bi, acc_scale=1.0 / sc)
    u = A.double() @ W.double().T + b.double()
    ref = u[:, :inner] * F.gelu(u[:, inner:])
    got = unpack_ref(C.cpu(), inner)
    assert not torch.isnan(got).any()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    u32 = (A @ W.T + b)
    err32 = float(((u32[:, :inner] * F.gelu(u32[:, inner:])).double() - ref).abs().max()) / float(ref.abs().max())
    print(f"x2 GEGLU: max err / max|out| = {err:.2e}  (torch fp32: {err32:.2e})")
    assert err < 1e-6, (err, err32)                 # the 1.5e-7 erfc polynomial + the 2^-22 split of the output


@pytest.mark.parametrize("M,K", [(256, 128), (1000, 512), (37, 192), (65536, 512)])
def test_x2_gemm_qkv_with_fused_qknorm(lib, dev, M, K):
    """q, k = MultiHeadRMSNorm(x W^T) as head / tail planes [2][H][2 chunks][M][64]; v as the paired transposed image
    [H][blk][2 chunks][64 d][64]; against fp64 on the original operands."""
    g = torch.Generator().manual_seed(15)
    H = 4 if M < 60000 else 8
    N = 3 * H * 64
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5 * 0.3
    gq, gk = torch.rand(H, 64, generator=g) + 0.5, torch.rand(H, 64, generator=g) + 0.5
    sc = weight_scale(W)
    Ap, Wp = pack_dev(lib, dev, A), pack_dev(lib, dev, W, sc)
    nblk = (M + 255) // 256 * 256 // 64
    qk = torch.full((2, H, 2, M, 64), float("nan"), dtype=torch.float16, device=dev)
    vt = torch.full((H, nblk, 2, 64, 64), float("nan"), dtype=torch.float16, device=dev)
    x2_gemm(lib, dev, 5, Ap, Wp, qk, M, N, 2 * K, 0, acc_scale=1.0 / sc, heads=H, gq=gq.to(dev), gk=gk.to(dev), q_mul=8.0, vt=vt, vt_nblk=nblk)
    rows = torch.arange(M) if M <= 4096 else torch.randint(0, M, (1024,), generator=g)
    x = (A[rows].double() @ W.double().T).reshape(len(rows), 3, H, 64).permute(1, 2, 0, 3)        # [3][H][rows][64]
    nrm = x[:2].norm(dim=-1, keepdim=True).clamp_min(1e-12)
    ref_qk = x[:2] / nrm * torch.stack([gq, gk])[:, :, None, :].double() * 8.0
    qkc = qk.cpu()[:, :, :, rows]                                                           # [2][H][2][rows][64]
    got = torch.cat([qkc[:, :, c, :, :32].double() + qkc[:, :, c, :, 32:].double() for c in range(2)], dim=-1)   # dims 32c..32c+31
    err = float((got - ref_qk).abs().max()) / float(ref_qk.abs().max())
    print(f"x2 qkv+norm M={M} K={K}: q/k max err / max = {err:.2e}")
    assert err < 1e-6, err
    vtc = vt.cpu()
    t = rows
    vsum = vtc[..., :32].double() + vtc[..., 32:].double()                                  # [H][blk][2][64 d][32 in-chunk positions]
    pos = vt_pos(t & 63)
    gotv = vsum[:, t >> 6, pos >> 5, :, pos & 31]                                            # (rows, H, 64)
    want = x[2].permute(1, 0, 2)
    errv = float((gotv - want).abs().max()) / float(want.abs().max())
    assert errv < 1e-6, errv
    # filler rows of the last M tile that was touched are zeros: 128-row tiles for few-tile launches (fewer 256 x 256 tiles than CUs), else 256
    m_tile = 128 if ((M + 255) // 256) * (N // 256) < 256 else 256
    tp = torch.arange(M, (M + m_tile - 1) // m_tile * m_tile)
    if tp.numel():
        pp = vt_pos(tp & 63)
        z = torch.stack([vtc[:, int(a) >> 6, int(b) >> 5, :, int(b) & 31] for a, b in zip(tp, pp)])
        z2 = torch.stack([vtc[:, int(a) >> 6, int(b) >> 5, :, 32 + (int(b) & 31)] for a, b in zip(tp, pp)])
        assert torch.equal(z, torch.zeros_like(z)) and torch.equal(z2, torch.zeros_like(z2))


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def make_x2_attention_operands(q, k, v):
    """(H, TP, 64) fp32 q, k, v -> the paired planes rap_x2_attention reads, built with torch (independent of the GEMM epilogue)."""
    H, TP, _ = q.shape
    nblk = (TP + 255) // 256 * 256 // 64
    qk = torch.zeros(2, H, 2, TP, 64, dtype=torch.float16)
    for c, x in enumerate((q, k)):
        hi, lo = split_ref(x)
        for ch in range(2):
            qk[c, :, ch, :, :32] = hi[:, :, 32 * ch:32 * ch + 32]
            qk[c, :, ch, :, 32:] = lo[:, :, 32 * ch:32 * ch + 32]
    vt = torch.zeros(H, nblk, 2, 64, 64, dtype=torch.float16)
    hi, lo = split_ref(v)
    t = torch.arange(TP)
    pos = vt_pos(t & 63)
    vt[:, t >> 6, pos >> 5, :, pos & 31] = hi.permute(1, 0, 2)
    vt[:, t >> 6, pos >> 5, :, 32 + (pos & 31)] = lo.permute(1, 0, 2)
    return qk, vt, nblk


def run_x2_attention(lib, dev, q, k, v, cu):
    H, TP, _ = q.shape
    qk, vt, nblk = make_x2_attention_operands(q, k, v)
    qk, vt = qk.to(dev), vt.to(dev)
    cu_d = cu.to(device=dev, dtype=torch.int32)
    nseg = cu.numel() - 1
    ws = workspace(dev, lib.rap_attention_workspace_bytes(TP, nseg))
    out = torch.full((TP, 2 * H * 64), float("nan"), dtype=torch.float16, device=dev)
    rc = lib.rap_x2_attention(_lib.ptr(qk), _lib.ptr(vt), nblk, _lib.ptr(cu_d), nseg, _lib.ptr(out), TP, H, _lib.ptr(ws), ws.numel(), stream(dev))
    _lib.check(rc, "rap_x2_attention")
    torch.cuda.synchronize()
    return unpack_ref(out.cpu(), H * 64)


def attention_ref64(q, k, v, cu):
    H, TP, _ = q.shape
    qd, kd, vd = q.double(), k.double(), v.double()
    out = torch.zeros(TP, H * 64, dtype=torch.float64)
    for a, b in zip(cu[:-1].tolist(), cu[1:].tolist()):
        if b == a:
            continue
        p = torch.softmax(qd[:, a:b] @ kd[:, a:b].transpose(1, 2) / 8.0, dim=-1)
        out[a:b] = (p @ vd[:, a:b]).permute(1, 0, 2).reshape(b - a, H * 64)
    return out


@pytest.mark.parametrize("wpe", [2, 4], ids=["one-block-per-cu", "two-blocks-per-cu"])
@pytest.mark.parametrize("H", [1, 8])
def test_x2_attention_ragged_segments(lib, dev, H, wpe):
    g = torch.Generator().manual_seed(11 + H)
    lens = [1, 63, 64, 65, 300, 0, 257, 1000, 31, 512]          # unaligned starts, empty segment, multi-block segments
    cu = torch.tensor([0] + lens).cumsum(0)
    TP = int(cu[-1])
    q = F.normalize(torch.randn(H, TP, 64, generator=g), dim=-1) * 8 * (0.5 + torch.rand(H, 1, 64, generator=g))
    k = F.normalize(torch.randn(H, TP, 64, generator=g), dim=-1) * 8 * (0.5 + torch.rand(H, 1, 64, generator=g))
    v = torch.randn(H, TP, 64, generator=g)
    try:
        assert lib.rap_set_tuning(16, wpe) == 0
        out = run_x2_attention(lib, dev, q, k, v, cu)
    finally:
        assert lib.rap_set_tuning(16, 2) == 0
    ref = attention_ref64(q, k, v, cu)
    assert not torch.isnan(out).any()
    err = float((out - ref).abs().max())
    ref32 = torch.zeros_like(ref)
    for a, b in zip(cu[:-1].tolist(), cu[1:].tolist()):
        if b > a:
            ref32[a:b] = F.scaled_dot_product_attention(q[:, a:b], k[:, a:b], v[:, a:b]).permute(1, 0, 2).reshape(b - a, H * 64).double()
    err32 = float((ref32 - ref).abs().max())
    print(f"x2 attention H={H} wpe={wpe}: max abs err vs fp64 {err:.2e}  (torch fp32 SDPA: {err32:.2e}); max|v| ~ 4")
    assert err < 5e-6, (err, err32)                 # fp32 class on values of magnitude ~4; one fp16 plane gives ~5e-4


def test_x2_attention_single_token_segments_return_v(lib, dev):
    g = torch.Generator().manual_seed(3)
    TP, H = 130, 2
    q, k, v = (torch.randn(H, TP, 64, generator=g) for _ in range(3))
    out = run_x2_attention(lib, dev, q, k, v, torch.arange(TP + 1))
    hi, lo = split_ref(v)
    want = (hi.double() + lo.double()).permute(1, 0, 2).reshape(TP, H * 64)
    assert float((out - want).abs().max()) <= 2.0 ** -20 * float(want.abs().max())      # softmax over one key is exactly 1: out = split(v)


@pytest.mark.parametrize("variant", [2, 4], ids=["one-block-per-cu", "two-blocks-per-cu"])
def test_x2_attention_sharp_softmax_and_late_maximum(lib, dev, variant):
    """One key dominates, placed in the LAST tile (forces the running-max rescale on the final step), the first tile and the middle."""
    g = torch.Generator().manual_seed(9)
    H = 2
    try:
        assert lib.rap_set_tuning(16, variant) == 0
        for L in (700, 640, 65, 129):
            for spike_at in (L - 1, 0, L // 2):
                q = torch.randn(H, L, 64, generator=g); k = torch.randn(H, L, 64, generator=g) * 0.1; v = torch.randn(H, L, 64, generator=g)
                k[:, spike_at] = q[:, 5] * 4.0
                ref = attention_ref64(q, k, v, torch.tensor([0, L]))
                out = run_x2_attention(lib, dev, q, k, v, torch.tensor([0, L]))
                err = float((out - ref).abs().max())
                assert err < 5e-6, (L, spike_at, err)
    finally:
        assert lib.rap_set_tuning(16, 2) == 0


def test_x2_attention_full_size_agrees_with_fp64_on_sampled_rows(lib, dev):
    """BASELINE geometry: segments of 4096 and 8192 tokens, 8 heads (the shapes of a configs[1] pair), sampled query rows vs fp64."""
    g = torch.Generator().manual_seed(21)
    H = 8
    cu = torch.tensor([0, 4096, 8192, 16384])
    TP = int(cu[-1])
    q = F.normalize(torch.randn(H, TP, 64, generator=g), dim=-1) * 8 * (0.5 + torch.rand(H, 1, 64, generator=g))
    k = F.normalize(torch.randn(H, TP, 64, generator=g), dim=-1) * 8 * (0.5 + torch.rand(H, 1, 64, generator=g))
    v = torch.randn(H, TP, 64, generator=g)
    out = run_x2_attention(lib, dev, q, k, v, cu)
    assert not torch.isnan(out).any()
    worst = 0.0
    for a, b in zip(cu[:-1].tolist(), cu[1:].tolist()):
        rows = torch.randint(a, b, (64,), generator=g)
        p = torch.softmax(q[:, rows].double() @ k[:, a:b].double().transpose(1, 2) / 8.0, dim=-1)
        ref = (p @ v[:, a:b].double()).permute(1, 0, 2).reshape(64, H * 64)
        worst = max(worst, float((out[rows] - ref).abs().max()))
    print(f"x2 attention at L = 4096 / 8192: max abs err vs fp64 {worst:.2e}")
    assert worst < 2e-6, worst


# ---------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------
def test_x2_layernorm_writes_head_and_tail_planes(lib, dev):
    g = torch.Generator().manual_seed(4)
    TP, d, rows = 1001, 512, 3
    x = torch.randn(TP, d, generator=g) * 3 + 0.5
    mod = torch.randn(rows, 2 * d, generator=g) * 0.3
    tok = torch.randint(0, rows, (TP,), generator=g, dtype=torch.int32)
    out = torch.full((TP, 2 * d), float("nan"), dtype=torch.float16, device=dev)
    xd, md, td = x.to(dev), mod.to(dev), tok.to(dev)
    _lib.check(lib.rap_layernorm_mod_h16(3, _lib.ptr(xd), _lib.ptr(out), TP, d, _lib.ptr(md), 2 * d, _lib.ptr(td), stream(dev)), "ln")
    torch.cuda.synchronize()
    ref = F.layer_norm(x.double(), (d,), eps=1e-5) * (1 + mod.double()[tok.long(), :d]) + mod.double()[tok.long(), d:]
    got = unpack_ref(out.cpu(), d)
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    assert err < 5e-7, err
    gain, shift = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g)
    gd, sd_ = gain.to(dev), shift.to(dev)
    _lib.check(lib.rap_layernorm_affine_h16(3, _lib.ptr(xd), _lib.ptr(out), TP, d, _lib.ptr(gd), _lib.ptr(sd_), stream(dev)), "ln affine")
    torch.cuda.synchronize()
    ref = F.layer_norm(x.double(), (d,), gain.double(), shift.double(), eps=1e-5)
    err = float((unpack_ref(out.cpu(), d) - ref).abs().max()) / float(ref.abs().max())
    assert err < 5e-7, err


# ---------------------------------------------------------------------------------------------
# model level: the mode is selected like the others, the residual-dtype switch is ignored, bit-reproducible
# ---------------------------------------------------------------------------------------------
def test_x2_model_forward_is_deterministic_and_close_to_exact_fp32(dev):
    import rap_amd
    from oracle import rap_oracle as O
    from rap_amd import synthetic as S
    cfg = dict(S.RAP_12); cfg["num_layers"] = 2
    sd = S.make_weights(cfg, 3)
    inp = S.make_inputs([[700, 650], [300, 0, 129]], seed=42)
    cu_b, cu_p = O.prepare_cu_seqlens(inp)
    d = {k: v.to(dev) for k, v in inp.items()}
    ts = torch.tensor([0.7, 0.3], device=dev)
    outs = {}
    for mode in ("float32", "float32x2"):
        m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=2, num_heads=8, local_feat_dim=32, compute_dtype=mode,
                                  residual_dtype="float16")            # ignored by the split mode (fp32 stream)
        m.load_state_dict(sd); m.to(dev)
        f = lambda: m(x=d["x_1"], timesteps=ts, cond_coord=d["pointclouds"], local_features=d["features"], latent_features=None,
                      scales=d["scales"], anchor_indices=d["anchor_indices"], cu_seqlens_batch=cu_b.to(dev), cu_seqlens_part=cu_p.to(dev),
                      return_transformer_features=True)
        a, b = f(), f()
        assert torch.equal(a["velocity"], b["velocity"]) and torch.equal(a["transformer_features"], b["transformer_features"])
        outs[mode] = a
    ref = O.dit_forward({k: v.double() for k, v in sd.items()}, cfg, inp["x_1"].double(), ts.cpu().double(), inp["pointclouds"].double(),
                        inp["features"].double(), inp["scales"].double(), inp["anchor_indices"], cu_b, cu_p)
    e32 = float((outs["float32"]["velocity"].cpu().double() - ref).abs().max())
    ex2 = float((outs["float32x2"]["velocity"].cpu().double() - ref).abs().max())
    print(f"velocity vs fp64 oracle: exact-fp32 path {e32:.2e}, split precision {ex2:.2e} (max|v| {float(ref.abs().max()):.2f})")
    assert ex2 < 2e-5 and ex2 < 10 * max(e32, 2e-7), (ex2, e32)


def test_x2_small_calls_fall_back_to_exact_fp32(lib, dev):
    """Tuning key 17 (default 1 024 token rows): a SMALL call of a split-precision model runs the exact-fp32 kernels -- both are
    fp32-accurate, and below a few thousand tokens the split forms are not measured (at 1 024 tokens and above split precision wins: 46 vs 71 ms).  With the
    default the result is bit-identical to compute_dtype="float32"; forced (key 17 = 0) it is the split kernels' (different bits, same
    accuracy class); a call above the threshold takes the split kernels whatever the key says."""
    import rap_amd
    from rap_amd import synthetic as S
    cfg = dict(S.RAP_12); cfg["num_layers"] = 2
    sd = S.make_weights(cfg, 3)

    def run(mode, inp):
        m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=2, num_heads=8, local_feat_dim=32, compute_dtype=mode)
        m.load_state_dict(sd); m.to(dev)
        flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=2, rigidity_forcing=True)
        d = {k: v.to(dev) for k, v in inp.items()}
        return flow.sample_and_register(d, x_1=d["x_1"])["end_point_trajectory"]

    small = S.make_inputs([[400, 350]], seed=5)                   # 750 tokens -> 768 rows < 1 024
    big = S.make_inputs([[2500, 2400]], seed=6)                   # 4 900 tokens -> 5 120 rows
    ref_small, ref_big = run("float32", small), run("float32", big)
    try:
        assert lib.rap_set_tuning(17, 1024) == 0
        assert torch.equal(run("float32x2", small), ref_small)                       # the fp32 kernels ran
        xb = run("float32x2", big)
        assert not torch.equal(xb, ref_big) and float((xb - ref_big).abs().max()) < 2e-5   # the split kernels ran: fp32-accurate, other bits
        assert lib.rap_set_tuning(17, 0) == 0
        xs = run("float32x2", small)
        assert not torch.equal(xs, ref_small) and float((xs - ref_small).abs().max()) < 2e-5
    finally:
        assert lib.rap_set_tuning(17, 0) == 0                     # (this module's fixture restores the default at the end)


def test_x2_graph_replay_and_concurrent_shards_equal_the_eager_call(lib, dev):
    """The split-precision call never synchronises or allocates inside the library either: it replays as one HIP graph (bit-identical) and
    runs as concurrent batch shards on several streams, each with its own workspace (equal up to fp32 rounding: a shard is a smaller call, and
    smaller calls may take other tile shapes / split-K in the fp32 embedding and head GEMMs -- as in tests/test_sample_gpu.py)."""
    import rap_amd
    from rap_amd import synthetic as S
    cfg = dict(S.RAP_12); cfg["num_layers"] = 2
    sd = S.make_weights(cfg, 3)
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=2, num_heads=8, local_feat_dim=32, compute_dtype="float32x2")
    m.load_state_dict(sd); m.to(dev)
    kw = dict(flow_model=m, inference_sampling_steps=3, rigidity_forcing=True)
    eager, graph, shards = rap_amd.RectifiedPointFlow(**kw), rap_amd.RectifiedPointFlow(graph_replay=True, **kw), rap_amd.RectifiedPointFlow(num_streams=2, **kw)
    a = {k: v.to(dev) for k, v in S.make_inputs([[900, 1100], [1300, 700], [1000, 1000]], seed=1).items()}      # 6 000 tokens: the split kernels
    b = {k: v.to(dev) for k, v in S.make_inputs([[900, 1100], [1300, 700], [1000, 1000]], seed=2).items()}
    for src in (a, b, a):
        want = eager.sample_and_register(src, x_1=src["x_1"])
        got = graph.sample_and_register(src, x_1=src["x_1"])
        par = shards.sample_and_register(src, x_1=src["x_1"])
        for k in ("end_point_trajectory", "trajectory", "R", "t"):
            assert torch.equal(got[k], want[k]), ("graph", k)
            assert float((par[k] - want[k]).abs().max()) < 2e-5, ("shards", k)
    assert len(graph._graphs) == 1


def test_x2_few_token_forms_match_the_plain_ones(lib, dev):
    """Few-token split-precision calls: every attention work item over 2 / 4 key ranges (online-softmax partials merged by a combine pass)
    and the long-K feed-forward down-projection over 2 / 4 k ranges (fp32 partial tiles + the 16-bit path's combine pass), all GEMMs on
    128 x 128 tiles.  Same call with both splits disabled (tuning keys 5, 6): equal up to fp32 summation order; and against the oracle.
    The filler rows of the padded buffers stay finite (the NaN-prefilled workspace gives the same bits)."""
    import rap_amd
    from oracle import rap_oracle as O
    from rap_amd import flow_model as FM, synthetic as S
    cfg = dict(S.RAP_12); cfg["num_layers"] = 2
    sd = S.make_weights(cfg, 5)
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=2, num_heads=8, local_feat_dim=32, compute_dtype="float32x2")
    m.load_state_dict(sd); m.to(dev)
    inp = S.make_inputs([[700, 324], [130, 257]], seed=3)      # 1 411 tokens: 9 + 4 work items x 8 heads -> 2- / 4-way split
    flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=3, rigidity_forcing=True)
    d = {k: v.to(dev) for k, v in inp.items()}
    try:
        assert lib.rap_set_tuning(5, 0) == 0 and lib.rap_set_tuning(6, 0) == 0
        ref = flow.sample_and_register(d, x_1=d["x_1"])
    finally:
        assert lib.rap_set_tuning(5, 1) == 0 and lib.rap_set_tuning(6, 1) == 0
    out = flow.sample_and_register(d, x_1=d["x_1"])
    for k in ("end_point_trajectory", "trajectory", "R", "t"):
        assert float((out[k] - ref[k]).abs().max()) < 5e-6, k
    assert not torch.equal(out["trajectory"], ref["trajectory"])        # the split forms really ran (different summation order)
    gold = O.sample(sd, cfg, inp, 3, True)
    assert float((out["end_point_trajectory"].cpu() - gold["end_point_trajectory"]).abs().max()) < 5e-5
    with torch.inference_mode():
        for buf in FM._WORKSPACES.values():
            buf.fill_(0xFF)                                              # NaN bit patterns everywhere, filler rows included
    torch.cuda.synchronize()
    again = flow.sample_and_register(d, x_1=d["x_1"])
    for k in ("end_point_trajectory", "trajectory", "R", "t"):
        assert torch.isfinite(again[k]).all() and torch.equal(again[k], out[k]), k
