#!/bin/bash
# round 5, GPU call 4: the whole GPU suite on the round-5 tree; call-size sweep of split precision against exact fp32 (where is the crossover?)
set -u
OUT=gpurun_out/r05_c4
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python scripts/size_sweep.py --modes=float32x2,float32 --x2-forced 128 256 512 1000 2000 > "$OUT/size_sweep_x2_vs_f32.jsonl" 2> "$OUT/size_sweep.err"; echo "sweep exit $?"
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r05_c4/size_sweep_x2_vs_f32.jsonl")]
for r in rows: print(r["dtype"], r["tokens"], round(r["ms_per_call"], 1), "ms", round(r["achieved_tflops_whole_call"], 1), "TF")
PY
timeout 1700 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_gpu.log" | tail -20
tail -30 "$OUT/pytest_gpu.log" | grep -E "Error|assert|error" | head -20
echo "r05 call 4 done"
