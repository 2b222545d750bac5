#!/bin/bash
# r02 GPU call 2: pipelined five-stage ring main loop (variants 6/7/8) vs the two-stage default (1): parity + per-shape timing
set -u
OUT=gpurun_out/r02_c2; mkdir -p $OUT
export TMPDIR=/tmp
for v in 6 8; do
  RAP_TEST_GEMM_H16_VARIANT=$v timeout 200 python -m pytest tests/test_h16_gpu.py -m gpu -q -k "gemm or qkv or geglu" > $OUT/pytest_h16_variant$v.log 2>&1
  tail -2 $OUT/pytest_h16_variant$v.log
done
: > $OUT/kb.jsonl
for v in 1 6 7 8 1 6 7 8; do
  timeout 120 python scripts/kernel_bench.py --dtype bfloat16 --only gemm --h16-gemm-variant $v >> $OUT/kb.jsonl 2>> $OUT/kb.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c2/kb.jsonl"):
    j = json.loads(l); print(j["variant"], j["kernel"][:28], round(j["ms"], 3), round(j["tflops"]))
PY
echo "r02 call 2 done"
