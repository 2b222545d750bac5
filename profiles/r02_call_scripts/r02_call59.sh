#!/bin/bash
# r02 GPU call 59 (last seconds of the budget): 512-query attention blocks (variant 24): parity, one timing
set -u
OUT=gpurun_out/r02_c59; mkdir -p $OUT
timeout 60 python -m pytest tests/test_h16_gpu.py -m gpu -q -x -k "blocks-of-512 or (pipelined_variants and 24)" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for V in 0 24; do
  timeout 40 python scripts/kernel_bench.py --dtype bfloat16 --only attention --h16-attn-variant $V 2>> $OUT/kb.err | sed "s/^/{\"variant\": $V, \"row\": /; s/$/}/" >> $OUT/kb.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c59/kb.jsonl"):
    try: j = json.loads(l)
    except Exception: continue
    r = j["row"]; print(j["variant"], r.get("kernel", "")[:44], round(r.get("ms"), 4), round(r.get("tflops"), 1))
PY
echo "r02 call 59 done"
