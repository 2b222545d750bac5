#!/bin/bash
# round 5, GPU call 9: the few-token forms of the split-precision path (128 x 128 tiles for every GEMM, split-K, split-KV): parity and latency
set -u
OUT=gpurun_out/r05_c9
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_x2_gpu.py tests/test_sample_gpu.py -q -k "x2 or golden or switches" > "$OUT/pytest_small.log" 2>&1; echo "tests exit $?"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_small.log" | tail -12
timeout 300 python scripts/size_sweep.py --modes=float32x2 --x2-forced 128 256 512 1000 2000 > "$OUT/size_sweep_x2.jsonl" 2> "$OUT/size_sweep.err"; echo "sweep exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/r05_c9/size_sweep_x2.jsonl"):
    r = json.loads(l); print(r["dtype"], r["tokens"], round(r["ms_per_call"], 1), "ms", round(r["achieved_tflops_whole_call"], 1), "TF")
PY
for DT in float32x2 float32; do
  timeout 300 python bench.py --dtype $DT --batch 1 --points 1024 --flow-steps 10 --steps 20 --warmup 5 --no-ragged --gamma-scale 0 --no-cpu-baseline --no-secondary --no-profile --tuning 17=0 > "$OUT/bench_c0_$DT.json" 2> "$OUT/bench_c0_$DT.err"
  python - "$OUT/bench_c0_$DT.json" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(j["dtype"], "configs[0] geometry:", round(j["ms_per_step"], 2), "ms per call,", round(j["value"]), "points/s")
except Exception as e:
    print("no json", e)
PY
done
echo "r05 call 9 done"
