"""GPU tests of the round-6 few-token forms (the reference's everyday regime: batch_size: 1, config/RAP_inference.yaml:30-36), through the
C ABI and the model path:

  * the four-stage LDS-DMA ring of the 128 x 128 16-bit / split-precision GEMM (tuning key 18) keeps the k order of every accumulator:
    BIT-identical to the two-stage kernel;
  * the combine pass of every residual GEMM folded into the following LayerNorm (tuning key 19) forms the residual-stream value in the
    order of the stand-alone combine pass and normalises the STORED value: BIT-identical to the unfused launch sequence;
  * the 16-bit attention of few-token calls on 64 / 128-row work items with a four-stage K / V^T ring (tuning key 20 = 2 / 64 / 128): the same
    per-query arithmetic in the same order, BIT-identical to 256-row items on two stages;
  * the 16-bit attention with KEY GROUPS inside the block (tuning key 20 = 1, the default rule; 66 / 130 forced): the keys of a row are summed in
    another order, so not bit-identical -- kernel-level against fp64 on the rounded operands at the bound of the unsplit kernel (ragged
    segments, single-key segments, a dominant key in the first / last / a middle tile), model-level against the unsplit form and the oracle;
  * qk-norm inside the QKV epilogue on 128 x 128 tiles (the 16-bit modes; split precision had it) against the fp32 golden vectors of the
    reference, and kernel-level against fp64 on the rounded operands.
"""
import pytest
import torch

import rap_amd
from conftest import load_golden
from oracle import rap_oracle as O
from rap_amd import _lib, synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


_MODELS, _ORACLE = {}, {}      # one model per (mode, stream) and one oracle run per geometry for the whole module: the suite has a time budget


def _sample(dev, cdt, rdt, parts, layers=2, steps=3, seed=11):
    """-> (outputs on the host, (sd, cfg, inp)).  The tuning keys are read at launch time, so one model serves every variant of a test."""
    cfg = dict(S.RAP_12); cfg["num_layers"] = layers
    if (cdt, rdt, layers) not in _MODELS:
        sd = S.make_weights(cfg, 0)
        m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=layers, num_heads=8, local_feat_dim=32, compute_dtype=cdt,
                                  residual_dtype=rdt)
        m.load_state_dict(sd); m.to(dev)
        _MODELS[(cdt, rdt, layers)] = (sd, m)
    sd, m = _MODELS[(cdt, rdt, layers)]
    flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=steps, rigidity_forcing=True)
    inp = S.make_inputs(parts, seed=seed)
    out = flow.sample_and_register({k: v.to(dev) for k, v in inp.items()}, x_1=inp["x_1"].to(dev))
    torch.cuda.synchronize()
    return {k: out[k].cpu() for k in ("end_point_trajectory", "trajectory", "R", "t")}, (sd, cfg, inp)


def _oracle(ctx, parts, steps=3):
    key = repr(parts)
    if key not in _ORACLE:
        sd, cfg, inp = ctx
        _ORACLE[key] = O.sample(sd, cfg, inp, steps, True)
    return _ORACLE[key]


MODES = [("bfloat16", "float32"), ("bfloat16", "float16"), ("float16", "float32"), ("float16", "float16"), ("float32x2", "float32")]


GEOMS = {"c0-geometry": [[1024, 1024]], "ragged-2-samples": [[300, 211], [64, 500, 33]], "4000-tokens": [[2000, 1500, 500]]}
# every mode at the demo-pair size; the ragged and the 4 000-token geometries (other tile counts, split-K factors 4 / 2) in one 16-bit stream
# variant each and in split precision -- the GPU suite has a time budget
CASES = [(c, r, "c0-geometry") for c, r in MODES] + [("bfloat16", "float16", "ragged-2-samples"), ("float32x2", "float32", "ragged-2-samples"),
                                                      ("float16", "float32", "4000-tokens"), ("float32x2", "float32", "4000-tokens")]


@pytest.mark.parametrize("cdt,rdt,geom", CASES, ids=[f"{g}-{c}-{r}-stream" for c, r, g in CASES])
def test_ring_and_fused_combine_layernorm_are_bit_identical_to_the_round5_launch_sequence(dev, cdt, rdt, geom):
    parts = GEOMS[geom]
    lib = _lib.load()
    outs = {}
    try:
        assert lib.rap_set_tuning(17, 0) == 0                      # split precision at every size (small calls default to exact fp32)
        # tuning keys 18 (GEMM ring), 19 (combine + LayerNorm), 20 (16-bit attention: 64 / 128-row work items + four-stage K / V^T ring)
        # (key 20 = 2: small work items + ring WITHOUT key groups -- the bit-identical form; the default, 1, adds key groups: tests below)
        for tag, ring, fused, attn in (("r6", 256, 1, 2), ("ring-only", 256, 0, 0), ("fused-only", 0, 1, 0), ("attn-only", 0, 0, 2), ("r5", 0, 0, 0)):
            assert lib.rap_set_tuning(18, ring) == 0 and lib.rap_set_tuning(19, fused) == 0 and lib.rap_set_tuning(20, attn) == 0
            outs[tag], ctx = _sample(dev, cdt, rdt, parts)
    finally:
        assert lib.rap_set_tuning(18, 256) == 0 and lib.rap_set_tuning(19, 1) == 0 and lib.rap_set_tuning(20, 1) == 0 and lib.rap_set_tuning(17, 1024) == 0
    for tag in ("r6", "ring-only", "fused-only", "attn-only"):
        for k, v in outs["r5"].items():
            assert not torch.isnan(v).any()
            if cdt == "float32x2" and tag not in ("ring-only", "attn-only"):
                # split precision: the fused sequence ALSO splits K of the out-projection (physical K = 1024; round 5 split ff2 only), so the
                # k-sum is re-associated there: fp32-class agreement instead of bit identity
                assert float((outs[tag][k] - v).abs().max()) < 5e-6, (tag, k)
            else:
                assert torch.equal(outs[tag][k], v), (tag, k, float((outs[tag][k] - v).abs().max()))
    # ... and the result is the function the oracle computes (fp32 class for split precision, the 16-bit deviation class otherwise)
    ref = _oracle(ctx, parts)
    err = float((outs["r6"]["end_point_trajectory"] - ref["end_point_trajectory"]).abs().max())
    print(f"{cdt}/{rdt}: end points vs the oracle {err:.2e}")
    assert err < (5e-5 if cdt == "float32x2" else 2e-2)


@pytest.mark.parametrize("cdt", ["bfloat16", "float16"])
@pytest.mark.parametrize("name", ["l12_small_rigid", "l2_ragged_rigid", "l2_emptypart_rigid"])
def test_few_token_fused_qknorm_on_128_tiles_against_the_reference_golden(name, cdt, dev):
    """the whole velocity network of a few-token call: fused QKV + qk-norm (128 x 128 tiles, ring) vs the unfused projection + qk-norm
    kernels (tuning key 7), both against the reference's fp32 velocity."""
    lib = _lib.load()
    g, inp = load_golden(name)
    bound = {"bfloat16": 8e-3, "float16": 1e-3}[cdt]
    outs = {}
    try:
        for fused in (1, 0):
            assert lib.rap_set_tuning(7, fused) == 0
            cfg = dict(S.RAP_12); cfg["num_layers"] = int(g["num_layers"])
            sd = S.make_weights(cfg, int(g["weight_seed"]))
            model = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=cfg["num_layers"], num_heads=8, local_feat_dim=32,
                                          compute_dtype=cdt)
            model.load_state_dict(sd); model.to(dev)
            cu_b, cu_p = O.prepare_cu_seqlens(inp)
            d = {k: v.to(dev) for k, v in inp.items()}
            outs[fused] = model(x=d["x_1"], timesteps=torch.from_numpy(g["fwd_timesteps"]).to(dev), cond_coord=d["pointclouds"],
                                local_features=d["features"], latent_features=None, scales=d["scales"], anchor_indices=d["anchor_indices"],
                                cu_seqlens_batch=cu_b.to(dev), cu_seqlens_part=cu_p.to(dev)).cpu()
    finally:
        assert lib.rap_set_tuning(7, 1) == 0
    v_ref = torch.from_numpy(g["fwd_velocity"])
    vmax = v_ref.abs().max().item()
    e1, e0 = (outs[1] - v_ref).abs().max().item() / vmax, (outs[0] - v_ref).abs().max().item() / vmax
    print(f"{cdt} {name}: fused {e1:.2e}, unfused {e0:.2e} of max|v|")
    assert not torch.isnan(outs[1]).any()
    assert e1 < bound and e0 < bound and (outs[1] - outs[0]).abs().max().item() / vmax < bound


# ---------------------------------------------------------------------------------------------
# key groups inside the attention block (attention_h16_kgroup_kernel)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [66, 130], ids=["64-rows-x-4-key-groups", "128-rows-x-2-key-groups"])
@pytest.mark.parametrize("dt", [1, 2], ids=["bf16", "f16"])
def test_key_group_attention_kernel_against_fp64(dev, dt, mode):
    """The kernel-level checks of tests/test_h16_gpu.py (fp64 softmax attention on the ROUNDED operands, same bounds as the unsplit kernel)
    with tuning key 20 forcing the key-group kernel: ragged segments from 1 to 1000 keys incl. an empty one (groups without a tile,
    unaligned starts, masks in first and last tiles), online and bounded softmax; single-key segments return v; one dominant key in the
    last / first / a middle tile (the groups' running maxima differ by far more than the deferred-rescale threshold when they meet)."""
    import torch.nn.functional as F
    import test_h16_gpu as T
    lib = _lib.load()
    try:
        assert lib.rap_set_tuning(20, mode) == 0
        for H in (1, 8):
            g = torch.Generator().manual_seed(11 + H)
            lens = [1, 63, 64, 65, 300, 0, 257, 1000, 31, 512, 129, 191]
            cu = torch.tensor([0] + lens).cumsum(0)
            TP = int(cu[-1])
            q = F.normalize(torch.randn(H, TP, 64, generator=g), dim=-1) * 8 * (0.5 + torch.rand(H, 1, 64, generator=g))
            k = F.normalize(torch.randn(H, TP, 64, generator=g), dim=-1) * 8 * (0.5 + torch.rand(H, 1, 64, generator=g))
            v = torch.randn(H, TP, 64, generator=g)
            ref = T.attention_ref64(q, k, v, cu, dt)
            for bounded in (False, True):
                out = T.run_attention_h(lib, dev, dt, q, k, v, cu, bound=T.logit_bound(q, k) if bounded else None)
                assert not torch.isnan(out.float()).any()
                err = (out.double() - ref).abs().max().item()
                print(f"key groups, mode {mode} dt={dt} H={H} bounded={bounded}: max abs err vs fp64 {err:.2e}")
                assert err < 8 * T.ULP[dt], (H, bounded, err)
                again = T.run_attention_h(lib, dev, dt, q, k, v, cu, bound=T.logit_bound(q, k) if bounded else None)
                assert torch.equal(out.view(torch.int16), again.view(torch.int16))              # fixed merge order: run-to-run identical
        g = torch.Generator().manual_seed(3)
        TP, H = 130, 2
        q, k, v = (torch.randn(H, TP, 64, generator=g) for _ in range(3))
        out = T.run_attention_h(lib, dev, dt, q, k, v, torch.arange(TP + 1))
        assert torch.equal(out, T.to_h(v, dt).permute(1, 0, 2).reshape(TP, H * 64))
        g = torch.Generator().manual_seed(9)
        H, L = 2, 700
        for spike_at in (L - 1, 0, 350, 64, 255):
            q = torch.randn(H, L, 64, generator=g); k = torch.randn(H, L, 64, generator=g) * 0.1; v = torch.randn(H, L, 64, generator=g)
            k[:, spike_at] = q[:, 5] * 4.0
            ref = T.attention_ref64(q, k, v, torch.tensor([0, L]), dt)
            for bound in (None, T.logit_bound(q, k).clamp(max=40.0)):
                out = T.run_attention_h(lib, dev, dt, q, k, v, torch.tensor([0, L]), bound=bound)
                err = (out.double() - ref).abs().max().item()
                assert err < 8 * T.ULP[dt], (spike_at, bound is not None, err)
    finally:
        assert lib.rap_set_tuning(20, 1) == 0


KG_CASES = [("bfloat16", "float16", "c0-geometry"), ("float16", "float32", "c0-geometry"), ("bfloat16", "float32", "ragged-2-samples"),
            ("float16", "float16", "ragged-2-samples"), ("bfloat16", "float16", "4000-tokens")]


@pytest.mark.parametrize("cdt,rdt,geom", KG_CASES, ids=[f"{g}-{c}-{r}-stream" for c, r, g in KG_CASES])
def test_key_group_attention_in_the_model_path(dev, cdt, rdt, geom):
    """Whole sampling calls (2 layers, 3 steps, rigidity forcing) with the default rule of tuning key 20 (64-row items x 4 key groups up to
    2 048 token rows, 128 x 2 up to 4 096) and with both forced forms, against the unsplit few-token kernel (key 20 = 2) -- the two differ
    only in the fp32 summation order over a row's keys, i.e. by the occasional neighbouring 16-bit value of an attention output -- and
    against the oracle: no further from it than the unsplit form is (16-bit deviation class)."""
    parts = GEOMS[geom]
    lib = _lib.load()
    outs = {}
    try:
        for mode in (2, 1, 66, 130):
            assert lib.rap_set_tuning(20, mode) == 0
            outs[mode], ctx = _sample(dev, cdt, rdt, parts)
    finally:
        assert lib.rap_set_tuning(20, 1) == 0
    ref = _oracle(ctx, parts)
    e = {m: float((outs[m]["end_point_trajectory"] - ref["end_point_trajectory"]).abs().max()) for m in outs}
    d = {m: float((outs[m]["end_point_trajectory"] - outs[2]["end_point_trajectory"]).abs().max()) for m in (1, 66, 130)}
    print(f"{cdt}/{rdt} {geom}: end points vs the oracle {e}, vs the unsplit kernel {d}")
    for m in (1, 66, 130):
        assert all(torch.isfinite(v).all() for v in outs[m].values())
        assert e[m] < 2e-2 and e[m] < 2.0 * e[2] + 1e-3, (m, e)
        assert d[m] < 1e-2, (m, d)
