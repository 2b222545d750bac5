#!/bin/bash
# r02 GPU call 48: persistent two-stage 16-bit GEMM with next-tile prefetch (variant 16): parity, per-shape timing vs 14 and 1
set -u
OUT=gpurun_out/r02_c48; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_h16_gpu.py -m gpu -x -q -k "gemm and 16" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for V in 14 16 1; do
  timeout 300 python scripts/kernel_bench.py --dtype bfloat16 --only gemm --h16-gemm-variant $V 2>> $OUT/kb.err | sed "s/^/{\"variant\": $V, \"row\": /; s/$/}/" >> $OUT/kb.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c48/kb.jsonl"):
    try: j = json.loads(l)
    except Exception: continue
    r = j["row"]; print(j["variant"], r.get("kernel", "")[:46], round(r.get("ms"), 4), round(r.get("tflops"), 1))
PY
timeout 400 python scripts/gemm_square.py 16 > $OUT/gemm_square_v16.jsonl 2>> $OUT/kb.err; cat $OUT/gemm_square_v16.jsonl
echo "r02 call 48 done"
