#!/bin/bash
# r02 GPU call 54: bf16 attention with the output stored as whole rows through an LDS slab (variant 22): parity, timing, bench A/B
set -u
OUT=gpurun_out/r02_c54; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_h16_gpu.py -m gpu -x -q -k "row-stores or (pipelined_variants and 22)" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for V in 0 22 0 22; do
  timeout 200 python scripts/kernel_bench.py --dtype bfloat16 --only attention --h16-attn-variant $V 2>> $OUT/kb.err | sed "s/^/{\"variant\": $V, \"row\": /; s/$/}/" >> $OUT/kb.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c54/kb.jsonl"):
    try: j = json.loads(l)
    except Exception: continue
    r = j["row"]; print(j["variant"], r.get("kernel", "")[:44], round(r.get("ms"), 4), round(r.get("tflops"), 1))
PY
for V in 0 22; do
  timeout 300 python bench.py --dtype bfloat16 --no-cpu-baseline --no-profile --steps 3 --warmup 1 --tuning 3=$V > $OUT/bench_bf16_v$V.json 2> $OUT/e$V.log
done
python - <<'PY'
import json
for v in (0, 22):
    try:
        j = json.load(open(f"gpurun_out/r02_c54/bench_bf16_v{v}.json")); print(v, round(j["value"]), round(j["ms_per_step"], 1))
    except Exception as e: print(v, "failed", e)
PY
echo "r02 call 54 done"
