#!/usr/bin/env python3
"""bench.py -- registered points/sec of the MI355X-native flow-matching registration sampler.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch: RectifiedPointFlow.sample_rectified_flow
(20 Euler flow steps of the rap_12 velocity network + per-step Procrustes rigidity projection + final
per-view SE(3) recovery) on BASELINE.json configs[1]: 32 scan pairs x 2 views x 4096 points, fp32,
inputs already resident in HBM.  Multi-GPU: independent pairs shard across ranks (32 pairs per rank, weak
scaling), one RCCL all-gather of the registered clouds and poses at the end of every step.

`python bench.py --gpus N` WITHOUT torchrun launches the N ranks itself (re-runs this file under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`), refuses (exit code != 0) when fewer than N GPUs are visible, and every rank
asserts WORLD_SIZE == --gpus before anything is timed: the line can never report an n_gpus that was not requested.

Rank 0 prints ONE COMPACT JSON line on stdout (<= 6 KB, `compact_line`: the contract's keys + `roofline` + `cpu_baseline` + one-number
summaries; tests/test_host_logic.py::test_bench_line_is_small guards the size) and writes the FULL record described below to
gpurun_out/bench_detail.json (--detail-out) and to stderr.  `--config N` selects a BASELINE.json configs[N] preset (geometry, flow steps,
arithmetic and the config.workload label); the default is configs[1].  The full record carries:
  roofline      -- the dominant kernel (attention_f32_kernel): algorithmic FLOPs / HIP-event time, vs the
                   157.3 TFLOP/s fp32 matrix peak of gfx950.  `value` / `ms_per_step` come from K UN-instrumented calls
                   (`instrumented: false`); the per-kernel HIP events ride in ONE extra call after the timed region
                   (`instrumented_over_clean` = its duration over a clean call's);
  hbm_kernels   -- the HBM-bound ring of the same instrumented call (LayerNorm, posenc(x_t), Euler, Procrustes moments, rigid apply):
                   algorithmic bytes / HIP-event time in GB/s against the 8 TB/s spec (6.3 TB/s achievable);
  emulated_fp32 -- (round 5) the same batch in SPLIT PRECISION (compute dtype "float32x2": fp16 head + tail operands, three products per
                   contraction on the 16-bit matrix pipe, fp32-accurate results) with its own roofline (peak = 2 500 / 3 TFLOP/s of
                   fp32-equivalent work) and its parity against the reference fixture; `reduced_precision` = bf16, `f16` = fp16
                   operands (the reference's shipped GPU precision); `points_per_s_by_mode` lists all four at the top level;
  cpu_baseline  -- the reference timed on this box's host cores in this run: the LIVE reference (unmodified modules under
                   /root/reference, kind "reference") when the mount exists, else the CPU oracle (restatement pinned to it, kind
                   "port" -- the GPU box has no mount); a bounded sample: 1 pair, the first 2 of the 20 flow steps with the time taken at every
                   step start (-> a per-step slope and a fixed cost, extrapolated to 20 steps), or all 20 with --cpu-full.
  ragged        -- the same path on a RAGGED packed batch in the reference's own regime (rap_amd.synthetic.ragged_regime_parts:
                   samples of 2 / 8 / 64 parts, 200 ... 20 000 points per part, ~262 k points, not a multiple of any tile), in fp32 and
                   bf16, with the same roofline object and the whole-call algorithmic TFLOP/s beside the uniform batch's.
  roofline_online_softmax -- the same dominant kernel in its ONLINE-softmax instantiation, measured on the same batch with the seeded
                   q/k-norm gains multiplied by --gamma-scale (default 3: every logit bound 8 max|gamma_q| max|gamma_k| > 40, so every
                   attention launch takes the online kernel): what a trained checkpoint with large gains runs; the kernel is chosen per
                   (layer, branch) launch, so one hot head costs 1 / (2 * layers) of the difference.
  parity_vs_reference_golden -- pair 0 of the timed batch against tests/golden/headline_c1_*.npz: the unmodified reference's
                   result for that pair over ALL flow steps (final cloud, last x_t, poses, every 32nd point of every step).
  parity_vs_device_checker_last_pair -- the LAST pair of the timed batch against the pinned oracle evaluated on this GPU through
                   PyTorch-ROCm in fp32 (test infrastructure, oracle/rap_oracle.py sample(device=...); all 32 pairs:
                   tests/test_fullconfig_gpu.py).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_16BIT_MATRIX_TFLOPS = 2500.0  # same table: "Peak BF16/FP16 MFMA ~2.5 PF dense"
DTYPE_TAG = {"float32": "f32", "bfloat16": "bf16", "float16": "f16", "float32x2": "f32x2"}
# split precision: three fp16 MFMAs per product term -> a third of the 16-bit peak in fp32-equivalent (algorithmic) FLOPs
PEAK_X2_MATRIX_TFLOPS = PEAK_16BIT_MATRIX_TFLOPS / 3.0
PEAK_HBM_GBS = 8000.0              # same guide: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
PROF_CLASSES = 8                   # rap_profile_collect_ex: 0/1 attention, 2 GEMMs, 3 LayerNorm, 4 posenc, 5 Euler, 6 Procrustes, 7 rigid apply
CPU_BASELINE_THREADS = 16


# The template instantiation rap_sample launches per (precision, softmax kind) -- attn_f32.hip: launch_attention_f32,
# attn_h16.hip: launch_attention_h16 -- as rocprofv3 prints it; the committed PMC traffic numbers are used only for the symbol they
# were measured on.
SHIPPED_ATTENTION_SYMBOL = {
    ("float32", True): "attention_f32_kernel<4, true, false>", ("float32", False): "attention_f32_kernel<4, false, false>",
    ("bfloat16", True): "attention_h16_kernel<1, 24, true, 2>",
    ("bfloat16", False): "attention_h16_kernel<1, 3, true, 2>",
    ("float16", True): "attention_h16_kernel<2, 3, true, 2>",
    ("float16", False): "attention_h16_kernel<2, 3, true, 2>",
    ("float32x2", True): "attention_x2_kernel<2, false>", ("float32x2", False): "attention_x2_kernel<2, false>",
}


def pmc_traffic(dtype, symbol, section="kernels"):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc cannot run inside this
    process; the passes are separate runs, scripts/evidence.sh stages pmc / ragged, summarised in profiles/pmc_traffic.json per kernel SYMBOL).
    Returned only when the passes measured the instantiation this run timed; otherwise (None, reason)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            j = json.load(f)
        for sym, k in j.get(section, {}).items():
            if sym == symbol and isinstance(k, dict):
                return k["hbm_bytes_per_launch"], f"{j['source']}; measured on: {k['measured_on']}"
        return None, f"no PMC pass for {symbol} in profiles/pmc_traffic.json"
    except (OSError, KeyError, ValueError):
        return None, "no PMC summary under profiles/"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=None, choices=sorted(PRESETS),
                    help="BASELINE.json configs[N] preset: sets --batch/--views/--points/--flow-steps/--dtype/--rigidity and the "
                         "config.workload label (default 1 = the metric's configuration; explicit flags override the preset's values)")
    ap.add_argument("--batch", type=int, default=None, help="samples (scan pairs) per GPU")
    ap.add_argument("--views", type=int, default=None)
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--flow-steps", type=int, default=None)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--rigidity", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--dtype", default=None, choices=list(DTYPE_TAG),
                    help="arithmetic of the transformer blocks for the HEADLINE number: float32 = BASELINE configs[1] (default); "
                         "bfloat16 = the per-GPU shard of configs[2]; float16 = the reference's shipped GPU precision")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE",
                    help="rap_set_tuning(KEY, VALUE) before the run (A/B experiments; recorded in config.tuning)")
    ap.add_argument("--residual-dtype", default=None, choices=["auto", "float32", "float16"],
                    help="storage of the residual stream in the 16-bit modes (default: rap_amd's default, see PointCloudDiT)")
    ap.add_argument("--gamma-scale", type=float, default=3.0,
                    help="multiplier on the seeded q/k-norm gains for the extra 'roofline_online_softmax' leg (1 warm-up + 1 timed sample "
                         "call per precision; 0 = skip the leg)")
    ap.add_argument("--workload", default="uniform", choices=["uniform", "ragged"],
                    help="uniform = BASELINE configs[1] (--batch pairs x --views x --points; the headline); ragged = the HEADLINE run "
                         "itself on the ragged reference-regime batch (the default run appends it as the 'ragged' object instead)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default; the driver's contract): every rank owns --batch pairs (or its own ragged batch); strong: ONE job of "
                         "--batch pairs in all (or one ragged batch of --ragged-points points) is sharded over the ranks by algorithmic "
                         "cost (rap_amd.parallel.shard_by_cost) and gathered back into the job's sample order")
    ap.add_argument("--no-ragged", action="store_true", help="skip the extra 'ragged' leg of a uniform run")
    ap.add_argument("--ragged-points", type=int, default=262144)
    ap.add_argument("--cpu-full", action="store_true", help="CPU baseline over ALL flow steps of the pair (minutes) instead of 1 + 3 steps")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra bf16 measurement of the same workload that a float32 run appends as 'reduced_precision'")
    ap.add_argument("--detail-out", default=None, metavar="PATH",
                    help="where the FULL record goes (default gpurun_out/bench_detail.json; it is also printed to stderr); stdout carries "
                         "only the compact line")
    ap.add_argument("--light", action="store_true",
                    help="headline only: --no-secondary --no-ragged --gamma-scale 0 --no-cpu-baseline (profiler-driven and per-config runs)")
    args = ap.parse_args()
    p = PRESETS[1 if args.config is None else args.config]
    for k in ("batch", "views", "points", "flow_steps", "dtype", "rigidity"):
        if getattr(args, k) is None:
            setattr(args, k, p[k])
    if args.light:      # (the CPU leg and the device checker of a 2 x 32 768-point pair take tens of minutes: they belong to the default run only)
        args.no_secondary, args.no_ragged, args.gamma_scale, args.no_cpu_baseline = True, True, 0.0, True
    return args


def attention_flops_per_forward(parts, heads=8, dh=64):
    """Algorithmic FLOPs of the two attention launches of one layer, summed over the batch:
    4 * H * Dh * L_seg per query token (QK^T and PV, 2 FLOP per MAC) -> 4 * H * Dh * sum(L_seg^2).
    parts: list over samples of part sizes."""
    per_part = 4 * heads * dh * sum(n * n for sizes in parts for n in sizes)
    per_sample = 4 * heads * dh * sum(sum(sizes) ** 2 for sizes in parts)
    return per_part, per_sample


DENSE_FLOPS_PER_TOKEN_LAYER = 10.486e6      # qkv + out (x2 branches) + GEGLU feed-forward, d = 512
EMBED_HEAD_FLOPS_PER_TOKEN = 0.97e6


def call_flops(parts, layers, flow_steps):
    """Algorithmic FLOPs of one sampling call."""
    tp = sum(sum(sizes) for sizes in parts)
    fp, fs = attention_flops_per_forward(parts)
    return flow_steps * (layers * (tp * DENSE_FLOPS_PER_TOKEN_LAYER + fp + fs) + tp * EMBED_HEAD_FLOPS_PER_TOKEN)


def cpu_baseline(cfg, sd, args, gpu_first_step, full=False):
    """The reference on the host cores of THIS box in THIS run: 1 pair of the workload's geometry.
    With /root/reference mounted: the unmodified reference modules (kind "reference"); otherwise the pinned restatement (kind
    "port").  Bounded sample: the first 2 flow steps, timed per step -> per-step slope + fixed cost -> 20-step figure; --cpu-full
    times all steps.  Needs no GPU (tests/test_host_logic.py runs it on a tiny configuration)."""
    from oracle import rap_oracle as O
    from oracle import ref_loader
    from rap_amd import synthetic as S
    # one process, 16 threads: the fastest setting on the 256-core GPU box (scripts/cpu_thread_sweep.py: 8/16/32/64/128/256
    # threads -> 3.5/2.7/3.0/3.4/5.1/35 s); more threads only add synchronisation overhead at these matrix sizes.
    nproc = os.cpu_count() or 1
    torch.set_num_threads(min(CPU_BASELINE_THREADS, nproc))
    inp = S.make_uniform_inputs(1, args.views, args.points, seed=1234)   # == pair 0 of rank 0's batch
    pts = args.views * args.points
    S_ = args.flow_steps
    live = ref_loader.reference_available()
    rigid = bool(args.rigidity)
    x_t0 = None           # x_t after flow step 0 (what both kinds can hand back)
    ref_first = None
    if live:
        k = S_ if full else min(2, S_)
        r = ref_loader.reference_time_steps(cfg, sd, inp, S_, rigid, max_steps=None if full else k)
        st = r["step_start"]
        t1 = st[1] - st[0]
        tk = st[k] - st[0] if len(st) > k else st[-1] - st[0]
        x_t0 = r["x_t_after_step"][0] if r["x_t_after_step"] else None
        if r["result"] is not None:
            ref_first = r["result"]
    else:
        k = S_ if full else min(2, S_)
        st = []
        refk = O.sample(sd, cfg, inp, S_, rigid, max_steps=k, step_hook=lambda: st.append(time.perf_counter()))
        st.append(time.perf_counter())                  # end of the last timed step (incl. the final pose fit)
        t1 = st[1] - st[0]
        tk = st[-1] - st[0]
        x_t0 = refk["trajectory"][0]
        ref_first = refk
    if full:
        total, how = tk, f"all {S_} flow steps timed"
        slope = (tk - t1) / max(1, k - 1)
    elif k > 1:
        slope = (tk - t1) / (k - 1)                 # seconds per additional flow step
        fixed = max(0.0, t1 - slope)                # model build / first-touch cost inside the first step
        total = fixed + slope * S_
        how = f"first 1 step {t1:.1f} s and first {k} steps {tk:.1f} s -> {slope:.2f} s per step + {fixed:.2f} s fixed, extrapolated to {S_} steps"
    else:
        slope, total, how = t1, t1 * S_, f"1 of {S_} steps, extrapolated linearly"
    out = {"value": pts / total, "unit": "points/s", "cores": torch.get_num_threads(), "nproc": nproc,
           "kind": "reference" if live else "port",
           "implementation": ("unmodified reference modules under /root/reference (oracle/ref_loader.py; flash-attn / diffusers stand-ins)"
                              if live else "oracle/rap_oracle.py, the restatement pinned to the reference (/root/reference is not mounted on this box)"),
           "sample": f"1 pair ({args.views}x{args.points} pts) incl. per-step rigidity projection{'' if not full else ' and pose fit'}: {how}",
           "steps_timed": k, "seconds_first_step": t1, "seconds_measured": tk,
           "seconds_per_flow_step": slope, "seconds_per_pair_all_steps": total, "extrapolated": not full}
    # SE(3) / cloud deviation of the GPU result from this CPU run, LIKE WITH LIKE (round 4 compared the oracle's poses,
    # fitted on its last COMPUTED step k - 1, with GPU poses fitted on step 0 and printed 2.2 degrees): `gpu_first_step` is
    # (x0 per step (S, n, 3), x_t per step (S, n, 3), fit(step) -> (R, t) of the GPU's end point of that step); the poses are compared
    # at the oracle's own step, the clouds at every step both sides computed.
    err = None
    if gpu_first_step is not None and x_t0 is not None:
        x0_gpu, xt_gpu, fit = gpu_first_step
        err = {"x_t_after_step0_max_abs": float((xt_gpu[0] - x_t0).abs().max())}
        if ref_first is not None and not live:
            ks = int(ref_first["end_point_trajectory"].shape[0])          # flow steps the CPU side computed (poses fitted on the last)
            R_gpu, t_gpu = fit(ks - 1)
            err.update({"steps_compared": ks, "pose_fit_on_step": ks - 1,
                        "x0_max_abs": float((x0_gpu[:ks] - ref_first["end_point_trajectory"]).abs().max()),
                        "x_t_max_abs": float((xt_gpu[:ks] - ref_first["trajectory"]).abs().max()),
                        "rot_err_deg_max": float(O.rotation_error_deg(R_gpu, ref_first["R"][0]).max()),
                        "R_frob_max": float(torch.linalg.matrix_norm(R_gpu - ref_first["R"][0]).max()),
                        "trans_abs_max": float((t_gpu - ref_first["t"][0]).abs().max())})
    return out, err


_CHECKER_CACHE: dict = {}


def device_checker_last_pair(cfg, sd, args, last, inp_cpu, dev):
    """The LAST pair of the timed batch vs the pinned oracle run on this GPU in fp32 through PyTorch-ROCm (test infrastructure)."""
    from oracle import rap_oracle as O
    from rap_amd import synthetic as S
    b = args.batch - 1
    one = S.make_inputs([[args.points] * args.views], seed=1234 + b)             # == sample b of rank 0's batch (per-sample seeds)
    n = args.views * args.points
    a0 = b * n
    assert torch.equal(one["pointclouds"], inp_cpu["pointclouds"][a0:a0 + n]) and torch.equal(one["x_1"], inp_cpu["x_1"][a0:a0 + n])
    key = (b, n, args.flow_steps, bool(args.rigidity))
    if _CHECKER_CACHE.get("key") != key:          # the checker's result for this pair is computed once per run (the split-precision leg reuses it)
        _CHECKER_CACHE["key"], _CHECKER_CACHE["ref"] = key, O.sample(sd, cfg, one, args.flow_steps, bool(args.rigidity), device=dev)
    ref = _CHECKER_CACHE["ref"]
    ep = last["end_point_trajectory"][:, a0:a0 + n]; tr = last["trajectory"][:, a0:a0 + n]
    per_step = (ep - ref["end_point_trajectory"]).abs().amax(dim=(1, 2))
    return {"pair": b, "checker": "oracle/rap_oracle.py sample(device=cuda), fp32 torch ops (pinned: tests/test_fullconfig_gpu.py)",
            "final_cloud_max_abs": float((ep[-1] - ref["end_point_trajectory"][-1]).abs().max()),
            "final_x_t_max_abs": float((tr[-1] - ref["trajectory"][-1]).abs().max()),
            "R_frob_max": float(torch.linalg.matrix_norm(last["R"][b:b + 1] - ref["R"]).max()),
            "t_max_abs": float((last["t"][b:b + 1] - ref["t"]).abs().max()),
            "per_step_max_abs": {"max": float(per_step.max()), "first": float(per_step[0]), "last": float(per_step[-1])}}


def golden_parity(args, last, data, fixture=None):
    """Pair 0 of THIS RANK's batch vs the all-step fixture the unmodified reference produced for exactly that pair (rank 0: seed 1234,
    `headline_c1_*`; rank 1 of a multi-GPU job with 32 pairs per rank: seed 1234 + 32, `headline_c2_rank1`; rap_12, 20 steps;
    oracle/make_golden.py --headline-only).  None when the configuration has no fixture."""
    import numpy as np
    if (args.views, args.points, args.flow_steps, args.layers) != (2, 4096, 20, 12):
        return None
    path = os.path.join(ROOT, "tests", "golden", (fixture or f"headline_c1_{'rigid' if args.rigidity else 'free'}") + ".npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    n0 = args.views * args.points
    stride = int(g["stride"])
    ep = last["end_point_trajectory"][:, :n0].cpu(); tr = last["trajectory"][:, :n0].cpu()
    R = last["R"][:1].cpu(); t = last["t"][:1].cpu()
    per_step = (ep[:, ::stride] - torch.from_numpy(g["end_point_strided"])).abs().amax(dim=(1, 2))
    return {"fixture": os.path.relpath(path, ROOT), "source": "unmodified reference modules, fp32 CPU, all 20 flow steps",
            "final_cloud_max_abs": float((ep[-1] - torch.from_numpy(g["final_end_point"])).abs().max()),
            "final_x_t_max_abs": float((tr[-1] - torch.from_numpy(g["final_x_t"])).abs().max()),
            "R_frob_max": float(torch.linalg.matrix_norm(R - torch.from_numpy(g["R"])).max()),
            "t_max_abs": float((t - torch.from_numpy(g["t"])).abs().max()),
            "per_step_max_abs": {"max": float(per_step.max()), "first": float(per_step[0]), "last": float(per_step[-1])}}


def reference_cpu_record():
    """The committed one-off timing of the LIVE reference on all 20 steps of one pair (build container, profiles/)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r02_cpu_reference_headline.json")) as f:      # the c1 fixtures were made in round 2
            j = json.load(f)["headline_c1_rigid"]
        return {"points_per_s": j["points_per_s"], "seconds": j["seconds"], "threads": j["threads"],
                "what": "unmodified reference modules, all 20 flow steps of one 2x4096 pair, CPU of the build container "
                        "(profiles/r02_cpu_reference_headline.json)"}
    except (OSError, KeyError, ValueError):
        return None


# ---- the ONE line on stdout (the round-5 line had grown to 28 KB and the driver could not parse it) -------------------
LINE_BUDGET_BYTES = 6144
REQUIRED_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "dtype", "data", "config")


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is a record, not a checkpoint); containers recursively"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full, detail_path=None):
    """The driver-facing JSON line: the contract's keys + `roofline` + `cpu_baseline` + a few one-number summaries, <= 6 KB.
    Everything else of `full` (per-mode legs, ragged batch, online-softmax leg, graph replay, prose) lives in the detail file and on
    stderr.  tests/test_host_logic.py::test_bench_line_is_small feeds this a committed full record."""
    line = {k: full.get(k) for k in REQUIRED_LINE_KEYS if k != "config"}
    cfg = full.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "preset", "pairs_per_gpu", "views", "points_per_view", "flow_steps", "num_layers",
                                 "rigidity_forcing", "sharding", "tuning"))
    for k in ("workload", "sharding"):
        if isinstance(line["config"].get(k), str) and len(line["config"][k]) > 260:
            line["config"][k] = line["config"][k][:257] + "..."
    roof = full.get("roofline")
    if roof:
        r = _pick(roof, ("kernel", "kernel_symbol", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                         "algorithmic_bytes_per_launch", "avg_launch_ms", "launches", "flops_per_launch_avg"))
        if isinstance(r.get("traffic_source"), str):
            r["traffic_source"] = r["traffic_source"][:120]
        g = roof.get("gemm") or {}
        if g.get("tflops") is not None:
            r["gemm_tflops"] = g["tflops"]
        line["roofline"] = r
    base = full.get("cpu_baseline")
    if base:
        b = _pick(base, ("value", "unit", "cores", "nproc", "kind", "sample", "steps_timed", "extrapolated", "gpu_over_cpu"))
        if isinstance(b.get("sample"), str):
            b["sample"] = b["sample"][:200]
        line["cpu_baseline"] = b
    if full.get("points_per_s_by_mode"):
        line["points_per_s_by_mode"] = full["points_per_s_by_mode"]
    three = ("final_cloud_max_abs", "R_frob_max", "t_max_abs")
    for k in ("parity_vs_reference_golden", "parity_vs_device_checker_last_pair", "parity_vs_reference_golden_rank1"):
        if full.get(k):
            line[k] = _pick(full[k], three + ("fixture",))
    if full.get("hbm_kernels"):
        line["hbm_kernels"] = {k: v.get("GB_per_s") for k, v in full["hbm_kernels"].items()}
    for k in ("rccl_ranks", "pairs_total", "achieved_tflops_whole_call", "instrumented", "all_gather_ms_per_step", "streams"):
        if k in full:
            line[k] = full[k]
    # optional one-number summaries, dropped first (in this order, last first) when the budget is short
    optional = []
    modes = {}
    for key in ("emulated_fp32", "reduced_precision", "f16"):
        leg = full.get(key)
        if leg and leg.get("roofline"):
            modes[leg["dtype"]] = _pick(leg["roofline"], ("kernel", "achieved", "peak", "frac", "avg_launch_ms"))
            g = leg["roofline"].get("gemm") or {}
            if g.get("tflops") is not None:
                modes[leg["dtype"]]["gemm_tflops"] = g["tflops"]
    if modes:
        line["roofline_by_mode"] = modes; optional.append("roofline_by_mode")
    ft = full.get("few_token_latency")
    if ft:
        line["few_token_ms_per_call"] = ft.get("ms_per_call"); optional.append("few_token_ms_per_call")
    rg = full.get("ragged")
    if rg:
        line["ragged_points_per_s"] = {t: rg[t].get("points_per_s") for t in ("f32", "f32x2", "bf16", "f16") if isinstance(rg.get(t), dict)}
        optional.append("ragged_points_per_s")
    err = full.get("se3_vs_cpu_oracle") or full.get("first_step_vs_cpu_reference")
    if err:
        line["se3_vs_cpu_baseline"] = _pick(err, ("rot_err_deg_max", "R_frob_max", "trans_abs_max", "x_t_after_step0_max_abs"))
        optional.append("se3_vs_cpu_baseline")
    if detail_path:
        line["detail"] = detail_path
    line = _r(line)
    while len(json.dumps(line)) > LINE_BUDGET_BYTES - 64 and optional:
        line.pop(optional.pop(0), None)
    if len(json.dumps(line)) > LINE_BUDGET_BYTES:          # never print an unparseable record: the contract keys alone always fit
        line = {k: line.get(k) for k in REQUIRED_LINE_KEYS + ("roofline", "cpu_baseline") if k in line}
        line["config"] = _pick(line["config"], ("workload",))
    return line


# BASELINE.json configs[] as presets (`--config N`): geometry, flow steps, arithmetic AND the label the line carries.  configs[2] is a
# 256-pair job over 8 GPUs = 32 pairs per GPU (weak scaling: every rank runs this preset); configs[3] / [4] name no batch: 16 samples
# and 4 pairs are what fills one GPU for seconds per call.  configs[0] is the reference's CPU demo; its geometry on the GPU is `--config 0`.
PRESETS = {
    0: dict(batch=1, views=2, points=1024, flow_steps=10, dtype="float32", rigidity=1,
            label="configs[0] geometry on the GPU (the reference's demo pair; configs[0] itself is the CPU plumbing case)"),
    1: dict(batch=32, views=2, points=4096, flow_steps=20, dtype="float32", rigidity=1, label="configs[1]"),
    2: dict(batch=32, views=2, points=4096, flow_steps=20, dtype="bfloat16", rigidity=1,
            label="configs[2] per-GPU shard (256 pairs / 8 GPUs = 32 pairs per rank)"),
    3: dict(batch=16, views=8, points=2048, flow_steps=30, dtype="float32", rigidity=1, label="configs[3] (16 samples per call)"),
    4: dict(batch=4, views=2, points=32768, flow_steps=50, dtype="bfloat16", rigidity=1, label="configs[4] (4 pairs per call)"),
}


def preset_of(args):
    """The BASELINE.json configs[] entry whose GEOMETRY the run has (the label used to be derived from the dtype
    alone), or None for a custom shape.  configs[1] and configs[2]'s per-GPU shard share a geometry: the arithmetic tells them apart
    (fp32-accurate -> 1, 16-bit -> 2)."""
    for n, p in PRESETS.items():
        if (args.batch, args.views, args.points, args.flow_steps, int(bool(args.rigidity))) != (p["batch"], p["views"], p["points"], p["flow_steps"], p["rigidity"]):
            continue
        if n in (1, 2) and (args.dtype in ("float32", "float32x2")) != (n == 1):
            continue
        return n
    return None


def workload_label(args, layers=12):
    n = preset_of(args)
    arith = {"float32": "fp32 (exact-fp32 MFMA)", "float32x2": "split precision (fp16 head + tail, fp32-accurate)",
             "bfloat16": "bf16 MFMA blocks, fp32 accumulate/head", "float16": "fp16 MFMA blocks, fp32 accumulate/head"}[args.dtype]
    head = PRESETS[n]["label"] if n is not None else "custom shape (no BASELINE.json configs[] entry)"
    if n == 4 and args.dtype != "bfloat16":
        head += " -- geometry only: BASELINE.json states configs[4] in bf16"
    return (f"{head}: batch={args.batch} samples/GPU x {args.views} views x {args.points} pts, {args.flow_steps} Euler flow steps, "
            f"rap_{layers} (d=512, H=8), {arith}, rigidity_forcing={'on' if args.rigidity else 'off'}, final per-view SE(3) fit")


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: launch the N ranks ourselves -- the whole command a driver needs.
    Fails hard when the box has fewer than N GPUs instead of quietly timing one."""
    import subprocess
    selftest = os.environ.get("RAP_BENCH_LAUNCHER_SELFTEST") == "1"
    if not selftest:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} requested but {n_dev} GPU(s) visible; refusing to run fewer ranks than requested")
    # rendezvous on 127.0.0.1 with port 0: the c10d store binds a free port itself (a port picked here by bind-and-close could be
    # taken by another process before torchrun binds it -- ADVICE r03)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", f"--rdzv-id=rapbench{os.getpid()}", "--local-addr=127.0.0.1",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def launcher_selftest(args, world, rank, json_out):
    """RAP_BENCH_LAUNCHER_SELFTEST=1 (tests/test_parallel_cpu.py only): the launch / rendezvous / shard / gather / timing / JSON
    plumbing of this file on CPU ranks with the gloo backend and a STUB in place of the sampler (rap_amd has no CPU path).  The
    line it prints is labelled as such and carries no throughput claim."""
    import torch.distributed as dist
    from rap_amd.parallel import gather_registrations, shard_range
    dist.init_process_group(backend="gloo")
    assert dist.get_world_size() == args.gpus == world
    strong = getattr(args, "scaling", "weak") == "strong"
    ragged = strong and args.workload == "ragged"
    if strong:
        from rap_amd import synthetic as S
        from rap_amd.parallel import shard_by_cost
        job = S.ragged_regime_parts(args.ragged_points, seed=4321) if ragged else [[args.points] * args.views for _ in range(args.batch)]
        job_points = [sum(p) for p in job]
        assignment = shard_by_cost(job, world)
        mine = assignment[rank]
        if not mine:
            raise SystemExit(f"rank {rank}: the job has {len(job)} samples for {world} ranks -- nothing to do on this rank")
        nb = len(mine)
    else:
        mine = shard_range(args.batch * world, world, rank)
        assert (mine.start, mine.stop) == (rank * args.batch, (rank + 1) * args.batch)
        nb = args.batch
    per = args.views * args.points
    n = nb * per
    t0 = time.perf_counter()
    pmax = max(len(p) for p in job) if strong else args.views          # parts per sample (pose rows are padded to it, as the collate does)
    for _ in range(args.warmup + args.steps):
        final = torch.cat([torch.full((job_points[i], 3), float(i)) for i in mine]) if strong else torch.full((n, 3), float(rank))
        R = torch.eye(3).repeat(nb, pmax, 1, 1) * (rank + 1)
        t = torch.full((nb, pmax, 3), float(rank))
        if strong:      # the shard plan is known on every rank: ONE collective, no size / id exchange
            g = gather_registrations(final, R, t, plan=(assignment, job_points))
        else:
            g = gather_registrations(final, R, t, equal_shapes=True)
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    allt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allt, elapsed)
    if strong:      # the gathered job is in SAMPLE order whatever the assignment: sample i's rows carry the value i
        starts = [0]
        for c in job_points:
            starts.append(starts[-1] + c)
        ok = g[0].shape[0] == starts[-1] and g[1].shape[0] == len(job) and all(bool((g[0][starts[i]:starts[i + 1]] == float(i)).all()) for i in range(len(job)))
    else:
        ok = all(bool((g[0][r * n:(r + 1) * n] == float(r)).all()) for r in range(world)) and g[0].shape[0] == n * world
    if rank == 0:
        print(json.dumps({"metric": "launcher-selftest (stub sampler, gloo, CPU): NOT a measurement", "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "stub": True, "rccl_ranks": world,
                          "pairs_total": len(job) if strong else args.batch * world, "scaling": "strong" if strong else "weak",
                          "workload": "ragged" if ragged else "uniform", "samples_per_rank": [len(a) for a in assignment] if strong else [args.batch] * world,
                          "gather_ok": ok, "per_rank": {"elapsed_s": [float(x) for x in allt]}}), file=json_out, flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # stdout carries exactly ONE line, the JSON result: RCCL prints a version banner to fd 1 when the first communicator is
    # created, so everything else this process (or a library in it) writes to fd 1 is sent to stderr instead.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RAP_BENCH_FORCE_DIST=1: take the torch.distributed / RCCL path with a single rank too (exercises it on a 1-GPU box)
    distributed = world > 1 or os.environ.get("RAP_BENCH_FORCE_DIST") == "1"
    if args.gpus != world:                                  # in EVERY mode: never time a rank count that was not requested
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("RAP_BENCH_LAUNCHER_SELFTEST") == "1":
        return launcher_selftest(args, world, rank, json_out)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (rap_amd has no CPU path)")
    if torch.cuda.device_count() < (world if distributed and world > 1 else 1) or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} / WORLD_SIZE {world} but {torch.cuda.device_count()} GPU(s) visible "
                         "(one rank per GPU; RCCL refuses two ranks on one device)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    import rap_amd
    from rap_amd import _lib, synthetic as S
    from rap_amd.parallel import gather_registrations

    cfg = dict(S.RAP_12); cfg["num_layers"] = args.layers
    sd = S.make_weights(cfg, 0)
    # weak scaling (default): rank r owns pairs [r*batch, (r+1)*batch) of the global job; synthetic, seeded per pair.
    # strong scaling: ONE job (--batch pairs in all, or one ragged batch) sharded over the ranks by algorithmic cost; `mine` = the global
    # sample indices of this rank, every sample drawn from its own seed, so the job's result does not depend on the rank count.
    strong = args.scaling == "strong"
    from rap_amd.parallel import cost_imbalance, shard_by_cost
    shard_info = {}

    def make_workload(kind):
        if kind == "ragged":
            if strong:
                job = S.ragged_regime_parts(args.ragged_points, seed=4321)
                assign = shard_by_cost(job, world, num_layers=args.layers)
                shard_info.update(job_parts=job, assignment=assign, imbalance=cost_imbalance(job, assign, args.layers), job_points=[sum(p) for p in job])
                mine = assign[rank]
                if not mine:
                    raise SystemExit(f"rank {rank}: the ragged job has {len(job)} samples for {world} ranks -- nothing to do on this rank")
                cpu = S.make_inputs_subset(job, mine, seed=98765)
                parts = [job[i] for i in mine]
            else:
                parts = S.ragged_regime_parts(args.ragged_points, seed=4321 + rank)
                cpu = S.make_inputs(parts, seed=98765 + 1000 * rank)
        elif strong:
            job = [[args.points] * args.views for _ in range(args.batch)]
            assign = shard_by_cost(job, world, num_layers=args.layers)
            shard_info.update(job_parts=job, assignment=assign, imbalance=cost_imbalance(job, assign, args.layers), job_points=[sum(p) for p in job])
            mine = assign[rank]
            if not mine:
                raise SystemExit(f"rank {rank}: --batch {args.batch} pairs for {world} ranks -- nothing to do on this rank")
            cpu = S.make_inputs_subset(job, mine, seed=1234)
            parts = [job[i] for i in mine]
        else:
            parts = [[args.points] * args.views for _ in range(args.batch)]
            cpu = S.make_inputs(parts, seed=1234 + rank * args.batch)
        return parts, cpu, {k: v.to(dev) for k, v in cpu.items()}

    parts, inp, data = make_workload(args.workload)
    x_1 = data["x_1"]
    pts_per_rank = int(inp["pointclouds"].shape[0])
    # the whole job: what `value` counts (weak: every rank's batch; strong: the one job all ranks share)
    job_parts = shard_info["job_parts"] if strong else None
    job_pts = sum(sum(p) for p in job_parts) if strong else pts_per_rank * world
    job_samples = len(job_parts) if strong else len(parts) * world
    job_flops_per_call = call_flops(job_parts, args.layers, args.flow_steps) if strong else call_flops(parts, args.layers, args.flow_steps) * world
    lib = _lib.load()
    tuning = {}
    for kv in args.tuning:
        key, val = (int(x) for x in kv.split("="))
        if lib.rap_set_tuning(key, val) != 0:
            raise SystemExit(f"rap_set_tuning({key}, {val}) refused")
        tuning[str(key)] = val
    profile = not args.no_profile

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def scaled_gains(scale):
        """seeded weights with every MultiHeadRMSNorm gain (flow_model/norm.py:15-33) multiplied by `scale`"""
        if scale == 1.0:
            return sd
        return {k: (v * scale if k.endswith("q_norm.gamma") or k.endswith("k_norm.gamma") else v) for k, v in sd.items()}

    def run_mode(dtype, steps, warmup, gamma_scale=1.0, data=data, x_1=x_1, idle_probe=False):
        """W untimed + K timed sample calls with the transformer blocks in `dtype`; returns (elapsed, prof, last, flow)."""
        model = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=cfg["embed_dim"], num_layers=cfg["num_layers"],
                                      num_heads=cfg["num_heads"], local_feat_dim=cfg["local_feat_dim"],
                                      attn_dtype=dtype, compute_dtype=dtype, residual_dtype=args.residual_dtype)
        run_mode.residual_dtype = "float32" if dtype in ("float32", "float32x2") else (
            model.residual_dtype if model.residual_dtype != "auto" else ("float16" if dtype == "bfloat16" else "float32"))
        model.load_state_dict(scaled_gains(gamma_scale))
        model.to(dev)
        run_mode.bounded_launches = lib.rap_model_bounded_attention_launches(model._handle)
        flow = rap_amd.RectifiedPointFlow(flow_model=model, inference_sampling_steps=args.flow_steps,
                                          rigidity_forcing=bool(args.rigidity))

        ev_pairs = []

        def one_step():
            out = flow.sample_and_register(data, x_1=x_1)
            final = out["end_point_trajectory"][-1]
            if distributed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if strong:      # ranks hold different sample sets (cost-balanced, not contiguous): gather back into the job's sample order
                    g = gather_registrations(final, out["R"], out["t"], plan=(shard_info["assignment"], shard_info["job_points"]))
                else:
                    g = gather_registrations(final, out["R"], out["t"], equal_shapes=True)
                e1.record()
                ev_pairs.append((e0, e1))
                return g, out
            return (final, out["R"], out["t"]), out

        n_streams = flow._resolved_streams()
        run_mode.streams = n_streams
        lib.rap_profile_enable(0)
        for _ in range(warmup):
            one_step()
        barrier()
        # The K timed calls run UN-instrumented (round 4 recorded 3 840 HIP events per call inside the timed stream, and
        # a profiled arm clocks lower than a clean one -- never mix them): `value` is what a user's call costs.
        t0 = time.perf_counter()
        enqueue = 0.0
        first_enqueue = None
        for _ in range(steps):
            te = time.perf_counter()
            gathered, last = one_step()
            enqueue += time.perf_counter() - te      # host time to ENQUEUE one sample call (nothing in it synchronises)
            if first_enqueue is None:
                first_enqueue = time.perf_counter() - te     # the queue was empty (barrier above): pure enqueue cost of one call
        torch.cuda.synchronize()
        flow.check_pending()                          # deferred input validation: surface it here, outside the enqueue path
        local_elapsed = time.perf_counter() - t0          # this rank's own time, before the closing barrier
        barrier()
        elapsed = time.perf_counter() - t0
        run_mode.host_enqueue_ms = 1e3 * enqueue / steps
        run_mode.host_first_enqueue_ms = 1e3 * first_enqueue
        run_mode.gather_ms = sum(a.elapsed_time(b) for a, b in ev_pairs[-steps:]) / steps if ev_pairs else 0.0
        run_mode.rank_elapsed = [local_elapsed]
        if distributed:
            allt = torch.zeros(world, dtype=torch.float64, device=dev)
            mine_t = torch.tensor([local_elapsed], dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(allt, mine_t)
            run_mode.rank_elapsed = allt.tolist()
        # ... and the per-kernel HIP events ride in ONE extra call after it (single stream: with concurrent batch shards a kernel's
        # event-to-event time would include the other shard's kernels), whose duration against a clean call's is reported
        prof_ms = (ctypes.c_float * PROF_CLASSES)(); prof_n = (ctypes.c_int64 * PROF_CLASSES)()
        run_mode.prof_region_s = elapsed / steps
        run_mode.instrumented_over_clean = None
        if profile:
            prev_streams = flow.num_streams
            flow.num_streams = 1
            if n_streams != 1:
                one_step(); torch.cuda.synchronize()                  # new workspace / allocator warm-up of the one-stream shape
            lib.rap_profile_reset(); lib.rap_profile_enable(1)
            tp0 = time.perf_counter()
            one_step(); torch.cuda.synchronize()
            run_mode.prof_region_s = time.perf_counter() - tp0
            lib.rap_profile_enable(0)
            flow.num_streams = prev_streams
            _lib.check(lib.rap_profile_collect_ex(prof_ms, prof_n, PROF_CLASSES), "rap_profile_collect_ex")
            if n_streams == 1:
                run_mode.instrumented_over_clean = run_mode.prof_region_s / (local_elapsed / steps)
        run_mode.host_idle_unprofiled_ms = None
        if idle_probe and not distributed:
            # one extra call on an idle device after everything above: what a call costs the host once the process is warm and the HIP
            # queue is empty (~2 900 launches; the call path has no synchronisation).  The FIRST timed call is not that figure: it follows
            # the warm-up calls, during which the host ran a full call ahead of the GPU, and reads 0.5-2.6 s for the fp32 batch.
            torch.cuda.synchronize()
            tq = time.perf_counter()
            one_step()
            run_mode.host_idle_unprofiled_ms = 1e3 * (time.perf_counter() - tq)
            torch.cuda.synchronize()
        if distributed:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, (list(prof_ms), list(prof_n)), last

    def hbm_kernels_of(dtype, prof, parts):
        """The HBM-bound ring of the instrumented call (SURVEY.md section 8d "ALGORITHMIC BYTES"): GB/s = algorithmic bytes per launch x
        launches / summed HIP-event time, per kernel class.  Event pairs bracket ONE launch each, so a 7 us kernel's figure includes the
        event-to-event launch gap (a few us): the LayerNorm (~0.2 ms per launch) is the figure that measures the memory system."""
        prof_ms, prof_n = prof
        tokens = sum(sum(x) for x in parts)
        rows = (tokens + 255) // 256 * 256                       # layer kernels run over align_up(TP, 256) rows
        ln_bytes = {"float32": 4096, "float32x2": 4096}.get(dtype, 2048 if run_mode.residual_dtype == "float16" else 3072)
        spec = {"layernorm": (3, rows * ln_bytes, f"{ln_bytes} B per 512-wide token (read the residual stream, write the GEMM operand)"),
                "posenc_x": (4, tokens * 268, "268 B per point (12 read, 64 floats written)"),
                "euler": (5, tokens * (48 if args.rigidity else 60), "x_t, v read; x0_hat -> trajectory, x_t written (+ trajectory slot without rigidity forcing)"),
                "procrustes_moments": (6, tokens * 24, "cond, x0_hat read; 3x3 fit per part in fp64"),
                "rigid_apply": (7, tokens * 48, "cond, x_1 read; x_t and its trajectory slot written")}
        out = {}
        for name, (cls, nbytes, what) in spec.items():
            if prof_n[cls] > 0 and prof_ms[cls] > 0:
                gbs = nbytes * int(prof_n[cls]) / (prof_ms[cls] * 1e-3) / 1e9
                out[name] = {"GB_per_s": gbs, "frac_of_8TBps": gbs / PEAK_HBM_GBS, "launches": int(prof_n[cls]),
                             "avg_launch_us": 1e3 * prof_ms[cls] / int(prof_n[cls]), "algorithmic_bytes_per_launch": nbytes, "bytes": what}
        return out or None

    def roofline_of(dtype, prof, elapsed, bounded=True, parts=parts):
        prof_ms, prof_n = prof
        if not (profile and prof_n[0] > 0 and prof_n[1] > 0):
            return None
        peak = {"float32": PEAK_FP32_MATRIX_TFLOPS, "float32x2": PEAK_X2_MATRIX_TFLOPS}.get(dtype, PEAK_16BIT_MATRIX_TFLOPS)
        f_part, f_samp = attention_flops_per_forward(parts)
        tokens = sum(sum(x) for x in parts)
        n_launch = int(prof_n[0] + prof_n[1])
        flops = f_part * int(prof_n[0]) + f_samp * int(prof_n[1])
        secs = (prof_ms[0] + prof_ms[1]) * 1e-3
        achieved = flops / secs / 1e12
        elem = 4 if dtype in ("float32", "float32x2") else 2          # split precision moves a head and a tail per value
        symbol = SHIPPED_ATTENTION_SYMBOL[(dtype, bool(bounded))]
        if tokens == 32 * 2 * 4096 and all(len(x) == 2 and x[0] == 4096 and x[1] == 4096 for x in parts):
            traffic, source = pmc_traffic(dtype, symbol)
        elif rank == 0 and parts == S.ragged_regime_parts(262144, seed=4321):
            traffic, source = pmc_traffic(dtype, symbol, section="ragged_kernels")
        else:
            traffic, source = None, "the committed PMC passes measured the uniform configs[1] shape and the default ragged batch only"
        r = {
            "kernel": {"float32": "attention_f32_kernel", "float32x2": "attention_x2_kernel"}.get(dtype, "attention_h16_kernel"), "kernel_symbol": symbol,
            "softmax": "bounded, offset-free (every logit bound <= 40)" if bounded and dtype not in ("float16", "float32x2") else "online (running maximum)",
            "bound": "mfma",
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": source,
            # rocprofv3 --pmc cannot run inside this process: the figure is a committed constant of the builder's separate PMC passes
            # (scripts/evidence.sh, refreshed on the round-6 tree) for exactly this kernel symbol at the uniform configs[1] shape
            # (profiles/pmc_traffic.json), not a measurement of this run
            "traffic_measured_in_this_run": False,
            "algorithmic_bytes_per_launch": 4 * elem * tokens * 512,
            "launches": n_launch, "avg_launch_ms": 1e3 * secs / n_launch, "flops_per_launch_avg": flops / n_launch,
            "per_part": {"launches": int(prof_n[0]), "avg_ms": prof_ms[0] / max(1, prof_n[0]),
                         "tflops": f_part * int(prof_n[0]) / (prof_ms[0] * 1e-3) / 1e12},
            "per_sample": {"launches": int(prof_n[1]), "avg_ms": prof_ms[1] / max(1, prof_n[1]),
                           "tflops": f_samp * int(prof_n[1]) / (prof_ms[1] * 1e-3) / 1e12},
            "gemm": {"launches": int(prof_n[2]), "total_ms": float(prof_ms[2]),
                     "tflops": (tokens * DENSE_FLOPS_PER_TOKEN_LAYER * (int(prof_n[2]) / 6))
                               / (prof_ms[2] * 1e-3) / 1e12 if prof_n[2] else None},
            "fraction_of_step_time": {"attention": secs / elapsed, "gemm": prof_ms[2] * 1e-3 / elapsed},
            "measured_over": "ONE instrumented single-stream call after the K timed (un-instrumented) calls",
            "instrumented_over_clean": run_mode.instrumented_over_clean,
        }
        if dtype == "float32x2":
            r["peak_is"] = ("2 500 TFLOP/s dense fp16 MFMA / 3: every product term of the algorithmic FLOP count costs three "
                            "v_mfma_f32_32x32x16_f16 (head x head, head x tail, tail x head)")
            r["matrix_pipe_tflops_issued"] = 3 * achieved          # what the fp16 pipe actually executes
        return r

    elapsed, prof, last = run_mode(args.dtype, args.steps, args.warmup, idle_probe=not args.no_profile)     # (--no-profile: profiler-driven runs want exactly K calls)
    host_enqueue_ms, host_first_enqueue_ms = run_mode.host_enqueue_ms, run_mode.host_first_enqueue_ms
    host_idle_unprofiled_ms = run_mode.host_idle_unprofiled_ms
    prof_region_s, main_streams = run_mode.prof_region_s, run_mode.streams
    main_instr_ratio = run_mode.instrumented_over_clean
    main_bounded = run_mode.bounded_launches
    main_rank_elapsed, main_gather_ms = list(run_mode.rank_elapsed), run_mode.gather_ms
    if main_bounded != 2 * args.layers:
        raise SystemExit(f"seeded weights: {main_bounded} of {2 * args.layers} attention launches bounded -- the headline expects all")
    # multi-GPU runs: the first pair of RANK 1 (seed 1234 + 32) has a fixture of the unmodified reference too -- parity evidence from a
    # rank that is not rank 0, gathered with one tiny collective every rank takes part in (the condition depends on the arguments only)
    rank1_parity = None
    if distributed and args.dtype == "float32" and args.workload == "uniform" and args.batch == 32 and args.rigidity:
        vals = torch.full((4,), float("nan"), dtype=torch.float64, device=dev)
        if rank == 1:
            try:
                g1 = golden_parity(args, last, data, fixture="headline_c2_rank1")
                if g1:
                    vals = torch.tensor([g1["final_cloud_max_abs"], g1["R_frob_max"], g1["t_max_abs"], g1["per_step_max_abs"]["max"]],
                                        dtype=torch.float64, device=dev)
            except Exception:      # never let the evidence leg take the job down (the collective below still runs)
                pass
        allv = torch.zeros(4 * world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allv, vals)
        r1 = allv[4:8].tolist() if world > 1 else [float("nan")] * 4        # (a forced 1-rank group exercises the collective only)
        if all(x == x for x in r1):
            rank1_parity = {"fixture": "tests/golden/headline_c2_rank1.npz", "source": "unmodified reference modules, fp32 CPU, all 20 flow steps; "
                            "computed ON rank 1 for the first pair it owns (input seed 1234 + 32)",
                            "final_cloud_max_abs": r1[0], "R_frob_max": r1[1], "t_max_abs": r1[2], "per_step_max_abs_max": r1[3]}
    # ---- the same workload in the other arithmetic modes, reported beside the fp32 headline (each at most 5 timed + 2 warm-up calls:
    # secondary measurements, and the default run has to stay within minutes):
    #   emulated_fp32     split precision ("float32x2", round 5): fp32-accurate blocks on the fp16 matrix pipe
    #   reduced_precision bf16 MFMA blocks (BASELINE configs[2]'s per-GPU shard)
    #   f16               fp16 MFMA blocks (the reference's shipped GPU precision, trainer/infer.yaml:6)
    WHAT = {"float32x2": "same batch, SPLIT-PRECISION transformer blocks: every operand an fp16 head + tail, three products per contraction on "
                         "the 16-bit matrix pipe, fp32 accumulate; residual stream / LN / qk-norm / softmax state / GEGLU / head fp32 as in the headline",
            "bfloat16": "same batch, bf16 MFMA transformer blocks (fp32 accumulate / LN statistics / softmax / head)",
            "float16": "same batch, fp16 MFMA transformer blocks, online softmax (fp32 accumulate / LN statistics / softmax / head)"}

    def mode_leg(dtype, max_steps):
        k_steps, k_warm = min(args.steps, max_steps), min(args.warmup, 2)
        e2, p2, l2 = run_mode(dtype, k_steps, k_warm)
        a, b = l2["end_point_trajectory"][-1], last["end_point_trajectory"][-1]
        leg = {
            "dtype": DTYPE_TAG[dtype], "value": job_pts * k_steps / e2, "unit": "points/s", "ms_per_step": 1e3 * e2 / k_steps,
            "steps": k_steps, "warmup": k_warm, "instrumented": False,
            "host_call_ms_per_step": run_mode.host_enqueue_ms,
            "workload": WHAT[dtype] + f"; residual stream held in {run_mode.residual_dtype}",
            "residual_stream": run_mode.residual_dtype,
            "achieved_tflops_whole_call": job_flops_per_call * k_steps / e2 / 1e12,
            "roofline": roofline_of(dtype, p2, run_mode.prof_region_s), "streams": run_mode.streams,
            "hbm_kernels": hbm_kernels_of(dtype, p2, parts) if profile else None,
            "bounded_attention_launches": f"{run_mode.bounded_launches} of {2 * args.layers}",
            "deviation_from_fp32_path": {"final_cloud_max_abs": float((a - b).abs().max()),
                                         "R_frob_max": float(torch.linalg.matrix_norm(l2["R"] - last["R"]).max()),
                                         "t_max_abs": float((l2["t"] - last["t"]).abs().max())}}
        gp2 = golden_parity(args, l2, data) if args.workload == "uniform" else None
        if gp2:
            leg["parity_vs_reference_golden" if dtype == "float32x2" else "deviation_from_reference_golden"] = (
                gp2 if dtype == "float32x2" else {k: gp2[k] for k in ("final_cloud_max_abs", "R_frob_max", "t_max_abs")})
        if dtype == "float32x2" and args.workload == "uniform" and world == 1 and not args.no_cpu_baseline:
            leg["parity_vs_device_checker_last_pair"] = device_checker_last_pair(cfg, sd, args, l2, inp, dev)
        del l2
        return leg

    secondary = emulated = f16_leg = None
    if args.dtype == "float32" and not args.no_secondary:
        emulated = mode_leg("float32x2", 5)
        secondary = mode_leg("bfloat16", 5)
        f16_leg = mode_leg("float16", 3)

    # ---- opt-in HIP-graph replay of the same call (bf16, a few calls): what a call costs the HOST when it is one graph launch
    graph_leg = None
    if world == 1 and args.workload == "uniform" and args.dtype == "float32" and not args.no_secondary:
        mg = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=cfg["embed_dim"], num_layers=cfg["num_layers"], num_heads=cfg["num_heads"],
                                   local_feat_dim=cfg["local_feat_dim"], attn_dtype="bfloat16", compute_dtype="bfloat16",
                                   residual_dtype=args.residual_dtype)
        mg.load_state_dict(sd); mg.to(dev)
        kw = dict(flow_model=mg, inference_sampling_steps=args.flow_steps, rigidity_forcing=bool(args.rigidity))
        fg, fe = rap_amd.RectifiedPointFlow(graph_replay=True, **kw), rap_amd.RectifiedPointFlow(**kw)
        fg.sample_and_register(data, x_1=x_1)                    # eager warm-up + capture
        torch.cuda.synchronize()
        n_g, host = 3, []
        t0 = time.perf_counter()
        for _ in range(n_g):
            te = time.perf_counter()
            og = fg.sample_and_register(data, x_1=x_1)
            host.append(1e3 * (time.perf_counter() - te))
        torch.cuda.synchronize()
        eg = time.perf_counter() - t0
        fg.check_pending()
        te = time.perf_counter()
        oe = fe.sample_and_register(data, x_1=x_1)
        eager_host = 1e3 * (time.perf_counter() - te)
        torch.cuda.synchronize()
        graph_leg = {"what": "RectifiedPointFlow(graph_replay=True): the same batch in bf16, each call = input copies + ONE graph launch + result copies",
                     "calls": n_g, "points_per_s": pts_per_rank * n_g / eg, "ms_per_step": 1e3 * eg / n_g,
                     "host_call_ms": host, "eager_host_call_ms_idle_queue": eager_host,
                     "bit_identical_to_eager": bool(all(torch.equal(og[k], oe[k]) for k in ("end_point_trajectory", "trajectory", "R", "t")))}
        del mg, fg, fe, og, oe

    # ---- few-token latency: the reference's own demo size (BASELINE configs[0] geometry: 1 pair x 2 x 1024 points, 10 flow steps), one call at a
    # time, un-instrumented, in the three modes a user would pick from (the reference ships batch_size: 1)
    few_token = None
    if world == 1 and args.workload == "uniform" and args.dtype == "float32" and not args.no_secondary:
        small = {k: v.to(dev) for k, v in S.make_uniform_inputs(1, 2, 1024, seed=1234).items()}
        few_token = {"workload": "configs[0] geometry: 1 pair x 2 x 1024 points, 10 Euler flow steps, rap_%d, rigidity forcing on; one call at a time, "
                                 "5 warm-up + 20 timed calls, un-instrumented" % args.layers, "ms_per_call": {}}
        for dt_name in ("float32", "float32x2", "bfloat16"):
            ms = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=cfg["embed_dim"], num_layers=cfg["num_layers"], num_heads=cfg["num_heads"],
                                       local_feat_dim=cfg["local_feat_dim"], attn_dtype=dt_name, compute_dtype=dt_name, residual_dtype=args.residual_dtype)
            ms.load_state_dict(sd); ms.to(dev)
            fs = rap_amd.RectifiedPointFlow(flow_model=ms, inference_sampling_steps=10, rigidity_forcing=True)
            for _ in range(5):
                fs.sample_and_register(small, x_1=small["x_1"])
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(20):
                fs.sample_and_register(small, x_1=small["x_1"])
            torch.cuda.synchronize()
            few_token["ms_per_call"][DTYPE_TAG[dt_name]] = 1e3 * (time.perf_counter() - ts) / 20
            del ms, fs
        del small

    # ---- the RAGGED reference-regime batch through the same path: fp32 (1 warm-up + 1 timed call) and bf16
    ragged = None
    uniform_call_flops = call_flops(parts, args.layers, args.flow_steps)
    if args.workload == "uniform" and not args.no_ragged and world == 1:
        rparts, rinp, rdata = make_workload("ragged")
        rflops = call_flops(rparts, args.layers, args.flow_steps)
        rpts = int(rinp["pointclouds"].shape[0])
        ragged = {"workload": f"{len(rparts)} samples with 2 / 8 / 64 parts in turn, parts of 200 ... 20 000 points "
                              f"(rap_amd.synthetic.ragged_regime_parts, seed 4321): {rpts} points (not a multiple of 256), "
                              f"{sum(len(x) for x in rparts)} parts, longest sample {max(sum(x) for x in rparts)} points; "
                              f"{args.flow_steps} flow steps, rap_{args.layers}, rigidity_forcing={'on' if args.rigidity else 'off'}",
                  "points": rpts, "samples": len(rparts), "parts": sum(len(x) for x in rparts),
                  "algorithmic_tflop_per_call": rflops / 1e12, "uniform_algorithmic_tflop_per_call": uniform_call_flops / 1e12}
        for dt_name, k_steps in ((args.dtype, 1),) + ((("float32x2", 1), ("bfloat16", 2)) if args.dtype == "float32" and not args.no_secondary else ()):
            # fp32: no warm-up call (34 s each at this size; the instrumented call after the timed one has the workspace warm)
            er, pr, lr = run_mode(dt_name, k_steps, 1 if dt_name == "bfloat16" else 0, data=rdata, x_1=rdata["x_1"])
            finite = bool(torch.isfinite(lr["end_point_trajectory"][-1]).all() and torch.isfinite(lr["R"]).all())
            ragged[DTYPE_TAG[dt_name]] = {
                "points_per_s": rpts * k_steps / er, "ms_per_step": 1e3 * er / k_steps, "steps": k_steps, "warmup": 1 if dt_name == "bfloat16" else 0,
                "achieved_tflops_whole_call": rflops * k_steps / er / 1e12, "results_finite": finite,
                "roofline": roofline_of(dt_name, pr, run_mode.prof_region_s, parts=rparts)}
            del lr
        del rdata

    # ---- the online-softmax instantiation of the dominant kernel: same batch, q/k-norm gains x gamma_scale (every bound > 40)
    online = None
    if args.gamma_scale and args.gamma_scale != 1.0 and profile and world == 1:
        online = {"gamma_scale": args.gamma_scale,
                  "what": "same batch, MultiHeadRMSNorm gains of the seeded weights multiplied by gamma_scale: every logit bound "
                          "8 max|gamma_q| max|gamma_k| exceeds 40, so every attention launch takes the online-softmax kernel "
                          "(what a trained checkpoint with large gains runs); 1 timed sample call (+ 1 warm-up in bf16)"}
        for dt_name in ([args.dtype] if (args.dtype != "float32" or args.no_secondary) else ["float32", "bfloat16"]):
            eo, po, lo = run_mode(dt_name, 1, 0 if dt_name == "float32" else 1, gamma_scale=args.gamma_scale)      # (fp32: 16 s per call -- the workspace is warm already)
            ro = roofline_of(dt_name, po, run_mode.prof_region_s, bounded=False)
            if run_mode.bounded_launches != 0:
                raise SystemExit(f"gamma scale {args.gamma_scale}: {run_mode.bounded_launches} launches still bounded")
            if ro:
                ro["points_per_s"] = pts_per_rank / eo
                ro["ms_per_step"] = 1e3 * eo
            online[DTYPE_TAG[dt_name]] = ro
            if dt_name == args.dtype and rank == 0 and not args.no_cpu_baseline:
                # parity of the online path: pair 0, first flow step, vs the CPU oracle run with the same scaled gains
                from oracle import rap_oracle as O
                torch.set_num_threads(min(CPU_BASELINE_THREADS, os.cpu_count() or 1))
                small = S.make_uniform_inputs(1, args.views, min(args.points, 1024), seed=1234)
                ref = O.sample(scaled_gains(args.gamma_scale), cfg, small, args.flow_steps, bool(args.rigidity), max_steps=1)
                m2 = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=cfg["embed_dim"], num_layers=cfg["num_layers"],
                                           num_heads=cfg["num_heads"], local_feat_dim=cfg["local_feat_dim"], attn_dtype=dt_name,
                                           compute_dtype=dt_name)
                m2.load_state_dict(scaled_gains(args.gamma_scale)); m2.to(dev)
                f2 = rap_amd.RectifiedPointFlow(flow_model=m2, inference_sampling_steps=args.flow_steps, rigidity_forcing=bool(args.rigidity))
                o2 = f2.sample_and_register({k: v.to(dev) for k, v in small.items()}, x_1=small["x_1"].to(dev))
                online["parity_vs_cpu_oracle"] = {
                    "workload": f"1 pair x {args.views} x {min(args.points, 1024)} points, first of {args.flow_steps} flow steps, {dt_name}",
                    "x0_max_abs": float((o2["end_point_trajectory"][0].cpu() - ref["end_point_trajectory"][0]).abs().max())}
                del m2, f2, o2
            del lo

    result = None
    if rank == 0:
        total_pts = job_pts * args.steps
        value = total_pts / elapsed
        result = {
            "metric": (f"registered points/sec @{args.flow_steps} flow steps, {args.views}-view N={args.points}" if args.workload == "uniform" else
                       f"registered points/sec @{args.flow_steps} flow steps, ragged reference-regime batch"), "value": value, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": DTYPE_TAG[args.dtype], "data": "synthetic",
            "config": {"workload": ((f"RAGGED reference-regime batch ({len(parts)} samples, {pts_per_rank} points; NOT BASELINE's configuration): "
                                     f"{args.flow_steps} Euler flow steps, rap_{args.layers}, {args.dtype}, rigidity_forcing={'on' if args.rigidity else 'off'}")
                                    if args.workload == "ragged" else workload_label(args, args.layers)),
                       "preset": preset_of(args) if args.workload == "uniform" else None,
                       "pairs_per_gpu": args.batch, "views": args.views, "points_per_view": args.points,
                       "flow_steps": args.flow_steps, "num_layers": args.layers, "rigidity_forcing": bool(args.rigidity),
                       "sharding": (f"independent pairs, {world} rank(s), one RCCL all-gather of clouds+poses per step" if not strong else
                                    f"ONE job of {job_samples} samples sharded by cost over {world} rank(s), one RCCL all-gather of clouds+poses "
                                    "(+ a sample-id exchange) per step, results in the job's sample order")},
        }
        if tuning:
            result["config"]["tuning"] = tuning
        # host time spent INSIDE the sample call: ~3 300 launches at ~2.6 us each while the HIP queue has room (8.5 ms for a
        # single call), the GPU's own pace once the queue is full (the driver's 20-step run: the call blocks on queue slots)
        result["host_call_ms_per_step"] = host_enqueue_ms
        # ... and of the FIRST timed call, which starts on an empty queue: what one call costs the host when nothing is ahead of it
        # (no host synchronisation on the call path since round 4: the input check is deferred)
        result["host_call_ms_first_timed_call"] = host_first_enqueue_ms
        if host_idle_unprofiled_ms is not None:
            result["host_call_ms_idle_queue_unprofiled"] = host_idle_unprofiled_ms
        result["rccl_ranks"] = world if distributed else 0        # ranks in the RCCL process group (0: single process, no group)
        result["pairs_total"] = job_samples
        result["achieved_tflops_whole_call"] = job_flops_per_call * args.steps / elapsed / 1e12
        if strong:
            result["sharding"] = {"by": "algorithmic cost (rap_amd.parallel.shard_by_cost: LPT on 12 layers x (10.486 MFLOP per token + 2048 (sum L_part^2 + "
                                        "L_sample^2)))", "samples_per_rank": [len(a) for a in shard_info["assignment"]],
                                  "points_per_rank": [sum(sum(job_parts[i]) for i in a) for a in shard_info["assignment"]],
                                  "cost_imbalance_max_over_mean_minus_1": shard_info["imbalance"]}
        result["bounded_attention_launches"] = f"{main_bounded} of {2 * args.layers}"
        if distributed:
            result["per_rank"] = {"elapsed_s": main_rank_elapsed, "all_gather_ms_per_step": main_gather_ms,
                                  "note": "elapsed_s = each rank's own K steps before the closing barrier; value uses the max"}
            result["all_gather_ms_per_step"] = main_gather_ms
        roof = roofline_of(args.dtype, prof, prof_region_s)
        result["streams"] = main_streams
        result["instrumented"] = False          # the K timed calls carry no HIP events; the roofline's events ride in one extra call
        result["instrumented_over_clean"] = main_instr_ratio
        if roof:
            result["roofline"] = roof
        hk = hbm_kernels_of(args.dtype, prof, parts) if profile else None
        if hk:
            result["hbm_kernels"] = hk
        by_mode = {DTYPE_TAG[args.dtype]: value}
        for key, leg in (("emulated_fp32", emulated), ("reduced_precision", secondary), ("f16", f16_leg)):
            if leg:
                leg["speedup_vs_fp32_path"] = leg["value"] / value
                result[key] = leg
                by_mode[leg["dtype"]] = leg["value"]
        result["points_per_s_by_mode"] = by_mode
        if online:
            result["roofline_online_softmax"] = online
        if graph_leg:
            result["graph_replay"] = graph_leg
        if few_token:
            result["few_token_latency"] = few_token
        if ragged:
            for tag in ("f32", "f32x2", "bf16"):
                if tag in ragged:
                    other = {"bf16": secondary, "f32x2": emulated}.get(tag)
                    uni = (result["achieved_tflops_whole_call"] if tag == DTYPE_TAG[args.dtype] else
                           uniform_call_flops / (other["ms_per_step"] * 1e-3) / 1e12 if other else None)
                    ragged[tag]["uniform_achieved_tflops_whole_call"] = uni
                    ragged[tag]["ragged_over_uniform_at_equal_flops"] = ragged[tag]["achieved_tflops_whole_call"] / uni if uni else None
            result["ragged"] = ragged
        if world == 1 and not args.no_cpu_baseline:
            gpu_first = None
            if args.workload == "uniform":
                n0 = args.views * args.points
                ppp0 = data["points_per_part"][:1]

                def fit_step(step):      # the GPU's poses of pair 0 fitted on ITS end point of flow step `step` (the CPU side says which)
                    R0, t0_ = rap_amd.fit_transformations(data["pointclouds"][:n0], last["end_point_trajectory"][step][:n0].contiguous(), ppp0,
                                                          data["cu_seqlens"][:2])
                    return R0.cpu()[0], t0_.cpu()[0]

                gpu_first = (last["end_point_trajectory"][:, :n0].cpu(), last["trajectory"][:, :n0].cpu(), fit_step)
            base, err = cpu_baseline(cfg, sd, args, gpu_first, full=args.cpu_full)
            rec = reference_cpu_record()
            if rec:
                base["live_reference_all_steps_build_container"] = rec
            result["cpu_baseline"] = base
            if err:
                result["se3_vs_cpu_oracle" if base["kind"] == "port" else "first_step_vs_cpu_reference"] = err
            base["gpu_over_cpu"] = value / base["value"]     # context only (bounded CPU sample, extrapolated unless --cpu-full): never a headline
        gp = golden_parity(args, last, data) if (args.dtype == "float32" and args.workload == "uniform") else None
        if gp:
            result["parity_vs_reference_golden"] = gp
        if rank1_parity:
            result["parity_vs_reference_golden_rank1"] = rank1_parity
        if args.dtype == "float32" and args.workload == "uniform" and world == 1 and not args.no_cpu_baseline:
            result["parity_vs_device_checker_last_pair"] = device_checker_last_pair(cfg, sd, args, last, inp, dev)
        # the FULL record -> the detail file + stderr; stdout carries the compact line only (<= 6 KB, compact_line above)
        detail_path = args.detail_out or os.path.join("gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(result, f)
        except OSError:
            detail_path = None
        print(json.dumps(result), file=sys.stderr, flush=True)
        print(json.dumps(compact_line(result, detail_path)), file=json_out, flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
