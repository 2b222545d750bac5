#!/bin/bash
# r02 GPU call 31: bf16 attention time vs segment length at constant token count (TP = 262144): per-block fixed cost vs per-tile cost
set -u
OUT=gpurun_out/r02_c31; mkdir -p $OUT
export TMPDIR=/tmp
for V in 0 13; do
for PB in "1024 128" "2048 64" "4096 32" "8192 16" "16384 8" "32768 4"; do
  set -- $PB
  timeout 200 python scripts/kernel_bench.py --dtype bfloat16 --only attention --h16-attn-variant $V --points $1 --batch $2 --views 2 2>> $OUT/kb.err | sed "s/^/{\"variant\": $V, \"points\": $1, \"row\": /; s/$/}/" >> $OUT/kb.jsonl
done
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c31/kb.jsonl"):
    try: j = json.loads(l)
    except Exception as e: print("bad", l[:80]); continue
    r = j["row"]; print(j["variant"], j["points"], r.get("kernel", "")[:44], round(r.get("ms"), 3), round(r.get("tflops"), 1))
PY
tail -3 $OUT/kb.err
echo "r02 call 31 done"
