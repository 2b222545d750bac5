"""GPU tests of the round-6 few-token forms (the reference's everyday regime: batch_size: 1, config/RAP_inference.yaml:30-36), through the
C ABI and the model path:

  * the four-stage LDS-DMA ring of the 128 x 128 16-bit / split-precision GEMM (tuning key 18) keeps the k order of every accumulator:
    BIT-identical to the two-stage kernel;
  * the combine pass of every residual GEMM folded into the following LayerNorm (tuning key 19) forms the residual-stream value in the
    order of the stand-alone combine pass and normalises the STORED value: BIT-identical to the unfused launch sequence;
  * the 16-bit attention of few-token calls on 64 / 128-row work items with a four-stage K / V^T ring (tuning key 20): the same per-query
    arithmetic in the same order, BIT-identical to 256-row items on two stages;
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


def _sample(dev, cdt, rdt, parts, layers=2, steps=3, seed=11):
    cfg = dict(S.RAP_12); cfg["num_layers"] = layers
    sd = S.make_weights(cfg, 0)
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=layers, num_heads=8, local_feat_dim=32, compute_dtype=cdt,
                              residual_dtype=rdt)
    m.load_state_dict(sd); m.to(dev)
    flow = rap_amd.RectifiedPointFlow(flow_model=m, inference_sampling_steps=steps, rigidity_forcing=True)
    inp = S.make_inputs(parts, seed=seed)
    out = flow.sample_and_register({k: v.to(dev) for k, v in inp.items()}, x_1=inp["x_1"].to(dev))
    torch.cuda.synchronize()
    return {k: out[k].cpu() for k in ("end_point_trajectory", "trajectory", "R", "t")}, (sd, cfg, inp)


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
        for tag, ring, fused, attn in (("r6", 256, 1, 1), ("ring-only", 256, 0, 0), ("fused-only", 0, 1, 0), ("attn-only", 0, 0, 1), ("r5", 0, 0, 0)):
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
    sd, cfg, inp = ctx
    ref = O.sample(sd, cfg, inp, 3, True)
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
