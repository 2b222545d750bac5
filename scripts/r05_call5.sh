#!/bin/bash
# round 5, GPU call 5: the few-token residual GEMMs on 128 x 128 split-precision tiles (parity + size sweep), the mutation check
set -u
OUT=gpurun_out/r05_c5
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_x2_gpu.py -q -s -k "gemm or small_calls or model_forward" > "$OUT/pytest_x2.log" 2>&1; echo "x2 tests exit $?"
grep -E "^(FAILED|ERROR)|passed|failed|x2 gemm" "$OUT/pytest_x2.log" | tail -14
timeout 300 python scripts/size_sweep.py --modes=float32x2 --x2-forced 256 320 384 512 1000 2000 4000 > "$OUT/size_sweep_x2.jsonl" 2> "$OUT/size_sweep.err"; echo "sweep exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/r05_c5/size_sweep_x2.jsonl"):
    r = json.loads(l); print(r["dtype"], r["tokens"], round(r["ms_per_call"], 1), "ms", round(r["achieved_tflops_whole_call"], 1), "TF")
PY
timeout 200 python scripts/size_sweep.py --modes=float32 320 384 > "$OUT/size_sweep_f32.jsonl" 2>> "$OUT/size_sweep.err"
python - <<'PY'
import json
for l in open("gpurun_out/r05_c5/size_sweep_f32.jsonl"):
    r = json.loads(l); print(r["dtype"], r["tokens"], round(r["ms_per_call"], 1), "ms", round(r["achieved_tflops_whole_call"], 1), "TF")
PY
bash scripts/mutation_check.sh "$OUT/mutation"
echo "r05 call 5 done"
