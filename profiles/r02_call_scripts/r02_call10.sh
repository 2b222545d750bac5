#!/bin/bash
# r02 GPU call 10: whole GPU suite, 2-rank test, fp32 GEMM per-shape choice A/B, bench line at default flags
set -u
OUT=gpurun_out/r02_c10; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
: > $OUT/kb_f32.jsonl
for v in 16 48 16 48; do timeout 200 python scripts/kernel_bench.py --only gemm --gemm-variant $v >> $OUT/kb_f32.jsonl 2>> $OUT/kb.err; done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c10/kb_f32.jsonl"):
    j = json.loads(l); print(j["variant"], j["kernel"][:30], round(j["ms"], 3), round(j["tflops"], 1))
PY
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-1500 $OUT/bench.json
echo "r02 call 10 done"
