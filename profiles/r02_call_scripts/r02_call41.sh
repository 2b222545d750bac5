#!/bin/bash
# r02 GPU call 41: fp32 GEMM tests with the new tile rule + the default bench line as the driver runs it
set -u
OUT=gpurun_out/r02_c41; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sample_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/e.log
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02_c41/bench_default.json")); r = j["roofline"]
print(round(j["value"]), round(j["ms_per_step"], 1), r["achieved"], r["frac"], r["gemm"], r["fraction_of_step_time"])
print({k: j[k] for k in j if k not in ("roofline", "config", "reduced_precision", "cpu_baseline")})
print("cpu_baseline", j.get("cpu_baseline"))
rp = j.get("reduced_precision", {})
print("reduced", {k: rp[k] for k in rp if k != "roofline"})
PY
echo "r02 call 41 done"
