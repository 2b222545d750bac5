"""Full-geometry, ALL-STEP parity at the BASELINE.json configurations (VERDICT r01 item 1 / SURVEY.md section 8d).

Fixtures `tests/golden/headline_*.npz` come from the reference's UNMODIFIED modules (oracle/make_golden.py
--headline-only: ref_loader.reference_sample on the CPU of the build container, fp32): they keep the seeds that regenerate
inputs and weights, the final registered cloud, the last x_t, the poses, per-step max-norms and every `stride`-th point of
EVERY flow step -- so error growth over the re-noised steps at L = 8192 / 16384 attention is checked step by step.

  headline_c1_rigid / _free : configs[1]: 1 pair x 2 x 4096, rap_12, 20 steps, rigidity forcing on / off  (= pair 0 of bench.py)
  headline_c3_rigid         : configs[3]: 1 sample x 8 x 2048, rap_12, 30 steps, rigidity on
  headline_c4_forward       : configs[4] geometry: 2 x 32768, one forward of a 2-layer model at t = 0.5
  headline_c4_steps         : configs[4] geometry: 2 x 32768, ALL 12 layers, two re-noised flow steps with rigidity forcing (round 3)
  headline_c2_rank1         : configs[2]: the first pair of RANK 1 of the 8-GPU job (input seed 1234 + 32), all 20 steps (round 3)
  headline_c1_ragged        : one RAGGED sample, parts of 4096 / 2500 / 1000 points + a trailing empty part, rap_12, 20 steps (round 3)
  headline_c1_rap16         : the configs[1] pair through rap_16, the reference's largest model, 20 steps (round 3)

Stated fp32 tolerances: end points / x_t 5e-4, |R - R_ref|_F 1e-3, |t - t_ref| 1e-3, velocity per
forward 1e-4 max|v|.  The 16-bit modes are compared with the SAME reference fixtures (not with the fp32 GPU path); their
deviation is recorded (printed + gpurun_out/headline_parity.jsonl) and bounded loosely.
"""
import json
import os

import numpy as np
import pytest
import torch

import rap_amd
from conftest import ROOT
from oracle import rap_oracle as O
from rap_amd import synthetic as S

pytestmark = pytest.mark.gpu

_SD = {}


def _weights(layers):
    if layers not in _SD:
        cfg = dict(S.RAP_12); cfg["num_layers"] = layers
        _SD[layers] = (cfg, S.make_weights(cfg, 0))
    return _SD[layers]


def _model(layers, dtype, dev, residual_dtype="float32"):
    cfg, sd = _weights(layers)
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=layers, num_heads=8, local_feat_dim=32,
                              attn_dtype=dtype, compute_dtype=dtype, residual_dtype=residual_dtype)
    m.load_state_dict(sd)
    return m.to(dev)


def _golden(name):
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not generated (python -m oracle.make_golden --headline-only)")
    z = np.load(path)
    return {k: z[k] for k in z.files}


def _record(row):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "headline_parity.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass
    print(row)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run_sample(g, dtype, dev, residual_dtype="float32", features=False):
    views, points, steps, layers = int(g["views"]), int(g["points"]), int(g["num_steps"]), int(g["num_layers"])
    cfg, sd = _weights(layers)
    assert abs(sum(v.double().sum().item() for v in sd.values()) - float(g["weights_checksum"])) < 1e-6
    if "parts" in g:                                            # ragged fixture: explicit part sizes (empty parts included)
        inp = S.make_inputs([[int(n) for n in row] for row in g["parts"]], seed=int(g["input_seed"]))
    else:
        inp = S.make_uniform_inputs(1, views, points, seed=int(g["input_seed"]))
    flow = rap_amd.RectifiedPointFlow(flow_model=_model(layers, dtype, dev, residual_dtype), inference_sampling_steps=steps,
                                      rigidity_forcing=bool(g["rigidity"]))
    d = {k: v.to(dev) for k, v in inp.items()}
    out = flow.sample_and_register(d, x_1=d["x_1"], return_transformer_features=features)
    torch.cuda.synchronize()
    return {k: out[k].cpu() for k in ("end_point_trajectory", "trajectory", "R", "t") + (("transformer_features",) if features else ())}


def _errors(out, g):
    stride = int(g["stride"])
    ep, tr = out["end_point_trajectory"], out["trajectory"]
    e = {
        "final_end_point": (ep[-1] - torch.from_numpy(g["final_end_point"])).abs().max().item(),
        "final_x_t": (tr[-1] - torch.from_numpy(g["final_x_t"])).abs().max().item(),
        "R_frob": torch.linalg.matrix_norm(out["R"] - torch.from_numpy(g["R"])).max().item(),
        "t": (out["t"] - torch.from_numpy(g["t"])).abs().max().item(),
        # error growth: every stride-th point of every step, both trajectories
        "per_step_end_point": (ep[:, ::stride] - torch.from_numpy(g["end_point_strided"])).abs().amax(dim=(1, 2)).tolist(),
        "per_step_x_t": (tr[:, ::stride] - torch.from_numpy(g["x_t_strided"])).abs().amax(dim=(1, 2)).tolist(),
        "step_max_end_point": (ep.abs().amax(dim=(1, 2)) - torch.from_numpy(g["end_point_step_max"])).abs().max().item(),
        "step_max_x_t": (tr.abs().amax(dim=(1, 2)) - torch.from_numpy(g["x_t_step_max"])).abs().max().item(),
    }
    ne = torch.from_numpy(g["R"]).abs().sum(dim=(-1, -2)) > 0           # empty parts carry R = 0 (procrustes.py:71-76): no angle
    e["rot_deg"] = O.rotation_error_deg(out["R"][ne], torch.from_numpy(g["R"])[ne]).max().item()
    return e


@pytest.mark.parametrize("dtype", ["float32", "float32x2"], ids=["exact-fp32", "split-precision"])
@pytest.mark.parametrize("name", ["headline_c1_rigid", "headline_c1_free", "headline_c3_rigid", "headline_c2_rank1", "headline_c1_ragged",
                                  "headline_c1_rap16"])
def test_fp32_all_steps_match_the_reference(name, dtype, dev):
    """Both fp32-accurate arithmetic modes against the SAME asserts: the exact-fp32 MFMA path (compute dtype 0) and, since round 5, the
    split-precision path (compute dtype 3: fp16 head + tail operands, three products per contraction on the 16-bit matrix pipe)."""
    g = _golden(name)
    e = _errors(_run_sample(g, dtype, dev), g)
    _record({"case": name, "dtype": "f32" if dtype == "float32" else "f32x2", **{k: v for k, v in e.items() if not k.startswith("per_step")},
             "per_step_end_point_max": max(e["per_step_end_point"]), "per_step_end_point_last": e["per_step_end_point"][-1]})
    # the stated tolerances, at every step
    assert max(e["per_step_end_point"]) <= 5e-4 and max(e["per_step_x_t"]) <= 5e-4, e
    assert e["final_end_point"] <= 5e-4 and e["final_x_t"] <= 5e-4, e
    assert e["step_max_end_point"] <= 5e-4 and e["step_max_x_t"] <= 5e-4, e
    if bool(g["rigidity"]):
        assert e["R_frob"] <= 1e-3 and e["t"] <= 1e-3, e
        # what an exact-fp32 path achieves over all steps (an order of magnitude inside the stated bound)
        assert e["final_end_point"] < 5e-5 and e["R_frob"] < 1e-4, e
    else:
        # without rigidity forcing the final per-view fit amplifies end-point noise: stated bound only
        assert e["R_frob"] <= 1e-3 and e["t"] <= 1e-3, e


# Bounds of the 16-bit modes: ~3 x the WORST value measured on MI355X for that fixture and operand type over the
# residual-stream variants (re-measured on the round-5 tree: profiles/r05_c4_headline_parity_all.jsonl) -- (final cloud, |dR|_F, |dt|).  The class bounds of rounds 1-4 (5e-2 / 1e-1 for bf16) were 10-100 x the measured values: a 10 x
# regression passed.  The companion file tests/golden/h16_deviation.json holds the measured values these were derived from.
TOL16 = {
    ("headline_c1_rigid", "bfloat16"): (1.6e-3, 4e-3, 1e-3), ("headline_c1_rigid", "float16"): (7e-4, 1.3e-3, 2.3e-4),
    ("headline_c3_rigid", "bfloat16"): (6e-3, 1.5e-2, 1.3e-3), ("headline_c3_rigid", "float16"): (2.1e-3, 5.6e-3, 3e-4),
    ("headline_c2_rank1", "bfloat16"): (4.3e-3, 5.8e-3, 1.5e-3), ("headline_c2_rank1", "float16"): (7.3e-4, 9e-4, 1.5e-4),
    ("headline_c1_ragged", "bfloat16"): (9e-3, 2.2e-2, 2.5e-3), ("headline_c1_ragged", "float16"): (4.6e-4, 9e-4, 2.5e-4),
    ("headline_c1_rap16", "bfloat16"): (4.4e-3, 7.5e-3, 3.3e-3), ("headline_c1_rap16", "float16"): (2.5e-4, 1.6e-4, 1.9e-4),
    ("headline_c4_steps", "bfloat16"): (3.7e-3, 5.8e-3, 1.7e-3), ("headline_c4_steps", "float16"): (5e-4, 6e-4, 2e-4),
}


@pytest.mark.parametrize("stream", ["float32", "auto"], ids=["fp32-stream", "shipped-default-stream"])
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("name", ["headline_c1_rigid", "headline_c3_rigid", "headline_c2_rank1", "headline_c1_ragged", "headline_c1_rap16"])
def test_16bit_all_steps_deviation_from_the_reference(name, dtype, stream, dev):
    """north_star: report the measured deviation of the reduced-precision modes -- against the reference's fp32 result.  Both with the
    fp32 residual stream and with the SHIPPED default ("auto": fp16 stream under bf16 operands, saturating at +-65504 since round 4;
    fp32 stream under fp16 operands) -- ADVICE r03: the headline suite must run the configuration users get."""
    g = _golden(name)
    out = _run_sample(g, dtype, dev, residual_dtype=stream)
    e = _errors(out, g)
    _record({"case": name, "dtype": dtype, "residual_stream": stream, **{k: v for k, v in e.items() if not k.startswith("per_step")},
             "per_step_end_point_max": max(e["per_step_end_point"]), "per_step_end_point_last": e["per_step_end_point"][-1]})
    cloud_tol, R_tol, t_tol = TOL16[(name, dtype)]
    assert e["final_end_point"] <= cloud_tol and e["R_frob"] <= R_tol and e["t"] <= t_tol, e
    nonempty = torch.from_numpy(g["R"]).abs().sum(dim=(-1, -2)) > 0      # an empty part has R = 0, t = 0 in the reference too (procrustes.py:71-76)
    assert torch.equal(out["R"].abs().sum(dim=(-1, -2)) > 0, nonempty)
    det = torch.linalg.det(out["R"].double())[nonempty]
    assert (det - 1).abs().max().item() < 1e-4          # proper rotations in every mode


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("name", ["headline_c1_rigid", "headline_c3_rigid"])
def test_16bit_residual_stream_all_steps_deviation_from_the_reference(name, dtype, dev):
    """Round 3: the residual stream itself held in fp16 (PointCloudDiT(residual_dtype="float16"), what the reference's own
    "16-mixed" inference holds) -- deviation from the reference's fp32 result over ALL flow steps, recorded next to the fp32-stream
    rows of the test above and bounded by the same loose class bounds."""
    g = _golden(name)
    out = _run_sample(g, dtype, dev, residual_dtype="float16")
    e = _errors(out, g)
    _record({"case": name, "dtype": dtype, "residual_stream": "fp16", **{k: v for k, v in e.items() if not k.startswith("per_step")},
             "per_step_end_point_max": max(e["per_step_end_point"]), "per_step_end_point_last": e["per_step_end_point"][-1]})
    cloud_tol, R_tol, t_tol = TOL16[(name, dtype)]
    assert e["final_end_point"] <= cloud_tol and e["R_frob"] <= R_tol and e["t"] <= t_tol, e
    det = torch.linalg.det(out["R"].double())
    assert (det - 1).abs().max().item() < 1e-4


def test_c4_geometry_all_layers_two_flow_steps_match_the_reference(dev):
    """configs[4] geometry through the WHOLE path: 2 x 32 768 points, rap_12 (12 layers), two re-noised flow
    steps with the per-step Procrustes rigidity projection, final pose fit -- attention at L = 65 536 / 32 768 in every layer and step --
    against the fixture the unmodified reference produced (oracle/make_golden.py --headline-only --c4_steps).  fp32 at the stated
    tolerances, incl. the transformer_features the sampling call captures on its last model call; the 16-bit
    modes' deviation from the reference is recorded and bounded by their class bounds."""
    g = _golden("headline_c4_steps")
    stride = int(g["stride"])
    f_ref = torch.from_numpy(g["sample_features_strided"]); fmax = float(g["sample_features_max"])
    for dtype in ("float32", "float32x2", "bfloat16", "float16"):
        cloud_tol, R_tol, t_tol = (5e-4, 1e-3, 1e-3) if dtype.startswith("float32") else TOL16[("headline_c4_steps", dtype)]
        out = _run_sample(g, dtype, dev, features=True)
        e = _errors(out, g)
        ef = (out["transformer_features"][::stride] - f_ref).abs().max().item()
        _record({"case": "headline_c4_steps", "dtype": dtype, **{k: v for k, v in e.items() if not k.startswith("per_step")},
                 "per_step_end_point": e["per_step_end_point"], "features_max_abs_err": ef, "features_max_abs": fmax})
        assert e["final_end_point"] <= cloud_tol and e["final_x_t"] <= cloud_tol and e["R_frob"] <= R_tol and e["t"] <= t_tol, (dtype, e)
        if dtype.startswith("float32"):
            assert max(e["per_step_end_point"]) <= 5e-4 and max(e["per_step_x_t"]) <= 5e-4, e
            assert ef <= 2e-4 * max(1.0, fmax), (ef, fmax)
            # what an exact-fp32 path achieves (an order of magnitude inside the stated bounds)
            assert e["final_end_point"] < 5e-5 and e["R_frob"] < 1e-4, e
        det = torch.linalg.det(out["R"].double())
        assert (det - 1).abs().max().item() < 1e-4


def test_c4_geometry_forward_matches_the_reference(dev):
    g = _golden("headline_c4_forward")
    cfg, sd = _weights(2)
    assert abs(sum(v.double().sum().item() for v in sd.values()) - float(g["weights_checksum"])) < 1e-6
    inp = S.make_uniform_inputs(1, int(g["views"]), int(g["points"]), seed=int(g["input_seed"]))
    cu_b, cu_p = O.prepare_cu_seqlens(inp)
    d = {k: v.to(dev) for k, v in inp.items()}
    v_ref = torch.from_numpy(g["velocity"])
    vmax = v_ref.abs().max().item()
    # 16-bit: ~3 x the measured deviation (r03: bf16 4.9e-4, fp16 5.1e-5 of max|v| 0.36)
    for dtype, tol in (("float32", 1e-4), ("float32x2", 1e-4), ("bfloat16", 4.2e-3), ("float16", 4.3e-4)):
        model = _model(2, dtype, dev)
        v = model(x=d["x_1"], timesteps=torch.tensor([float(g["timestep"])], device=dev), cond_coord=d["pointclouds"],
                  local_features=d["features"], latent_features=None, scales=d["scales"], anchor_indices=d["anchor_indices"],
                  cu_seqlens_batch=cu_b.to(dev), cu_seqlens_part=cu_p.to(dev))
        v = v["velocity"] if isinstance(v, dict) else v
        err = (v.cpu() - v_ref).abs().max().item()
        _record({"case": "headline_c4_forward", "dtype": dtype, "velocity_max_abs_err": err, "max_abs_v": vmax})
        assert err <= tol * vmax, (dtype, err, vmax)
        if dtype.startswith("float32"):
            assert err < 2e-5 * max(1.0, vmax), err
