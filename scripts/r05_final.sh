#!/bin/bash
# round 5, final GPU call: the driver's bench command on the final tree, and rocprofv3 kernel traces of one sampling call (fp32 headline and
# split precision) whose per-kernel averages the bench line's roofline has to agree with
set -u
OUT=gpurun_out/r05_final
mkdir -p "$OUT"
export TMPDIR=/tmp
# the whole GPU suite on the final tree first
timeout 1700 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_gpu.log" | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke exit $?"; tail -3 "$OUT/smoke.log"
HEAD=12 bash scripts/prof_bench.sh "$OUT/f32_kernel_trace"
HEAD=12 bash scripts/prof_bench.sh "$OUT/x2_kernel_trace" --dtype float32x2
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"; echo "bench exit $?"
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r05_final/bench_driver_flags.json"))
    print({k: j.get(k) for k in ("value", "ms_per_step", "points_per_s_by_mode", "instrumented", "instrumented_over_clean")})
    print("roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "traffic")}, "gemm", j["roofline"]["gemm"]["tflops"])
    for k in ("emulated_fp32", "reduced_precision", "f16"):
        l = j.get(k) or {}
        print(k, l.get("value"), (l.get("roofline") or {}).get("frac"), ((l.get("roofline") or {}).get("gemm") or {}).get("tflops"), l.get("parity_vs_reference_golden", l.get("deviation_from_reference_golden")))
    print("se3", j.get("se3_vs_cpu_oracle")); print("cpu", {k: j["cpu_baseline"][k] for k in ("value", "kind", "cores")})
    print("hbm", {k: round(v["GB_per_s"]) for k, v in (j.get("hbm_kernels") or {}).items()})
    print("ragged", {k: (v.get("points_per_s"), v.get("achieved_tflops_whole_call")) for k, v in (j.get("ragged") or {}).items() if isinstance(v, dict)})
    print("parity", j.get("parity_vs_reference_golden"), j.get("parity_vs_device_checker_last_pair"))
except Exception as e:
    print("no bench json:", e)
PY
tail -3 "$OUT/bench_driver_flags.err"
# the other BASELINE geometries in split precision (configs[3]: 16 x 8 x 2048, 30 steps; configs[4]: 4 x 2 x 32768, 50 steps)
timeout 400 python bench.py --dtype float32x2 --batch 16 --views 8 --points 2048 --flow-steps 30 --steps 2 --warmup 1 --no-ragged --gamma-scale 0 --no-cpu-baseline --no-secondary > "$OUT/bench_c3_x2.json" 2> "$OUT/bench_c3_x2.err"
timeout 900 python bench.py --dtype float32x2 --batch 4 --views 2 --points 32768 --flow-steps 50 --steps 1 --warmup 1 --no-ragged --gamma-scale 0 --no-cpu-baseline --no-secondary > "$OUT/bench_c4_x2.json" 2> "$OUT/bench_c4_x2.err"
python - <<'PY'
import json
for c in ("c3", "c4"):
    try:
        j = json.load(open(f"gpurun_out/r05_final/bench_{c}_x2.json")); r = j.get("roofline") or {}
        print(c, "x2:", round(j["value"]), "points/s,", round(j["ms_per_step"]), "ms per call; attention", round(r.get("achieved", 0), 1), "TF-eq, gemm", round((r.get("gemm") or {}).get("tflops") or 0, 1))
    except Exception as e:
        print(c, "no json", e)
PY
# kernel-level half of the mutation check (the model-level half ran in call 5: profiles/r05_c5_mutation_check.txt)
MUTATION_KERNEL_ONLY=1 bash scripts/mutation_check.sh "$OUT/mutation"
echo "r05 final done"
