#!/bin/bash
# r02 GPU call 34: rotated key-tile walk (variant 20): parity, timing vs default at three segment lengths, bench A/B
set -u
OUT=gpurun_out/r02_c34; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_h16_gpu.py -m gpu -x -q -k "schedule_variants or pipelined_variants" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for V in 0 20; do
for PB in "1024 128" "4096 32" "16384 8"; do
  set -- $PB
  timeout 200 python scripts/kernel_bench.py --dtype bfloat16 --only attention --h16-attn-variant $V --points $1 --batch $2 --views 2 2>> $OUT/kb.err | sed "s/^/{\"variant\": $V, \"points\": $1, \"row\": /; s/$/}/" >> $OUT/kb.jsonl
done
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c34/kb.jsonl"):
    try: j = json.loads(l)
    except Exception as e: print("bad", l[:80]); continue
    r = j["row"]; print(j["variant"], j["points"], r.get("kernel", "")[:44], round(r.get("ms"), 3), round(r.get("tflops"), 1))
PY
for V in 0 20; do
  timeout 300 python bench.py --dtype bfloat16 --no-cpu-baseline --steps 3 --warmup 1 --tuning 3=$V > $OUT/bench_bf16_v$V.json 2> $OUT/e$V.log
done
python - <<'PY'
import json
for v in (0, 20):
    try:
        j = json.load(open(f"gpurun_out/r02_c34/bench_bf16_v{v}.json")); r = j["roofline"]
        print(v, round(j["value"]), round(j["ms_per_step"], 1), round(r["achieved"], 1), round(r["frac"], 3), r["gemm"]["tflops"], r["fraction_of_step_time"])
    except Exception as e:
        print(v, "failed", e)
PY
echo "r02 call 34 done"
