#!/bin/bash
# r02 GPU call 27: the compiler-scheduled, two-tiles-per-barrier software-pipelined bf16 attention (variant 13): parity, then timing vs the default
set -u
OUT=gpurun_out/r02_c27; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_h16_gpu.py -m gpu -x -q -k "schedule_variants or pipelined_variants" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for V in 0 13 12; do
  timeout 200 python scripts/kernel_bench.py --dtype bfloat16 --only attention --h16-attn-variant $V > $OUT/kb_v$V.jsonl 2>> $OUT/kb.err
done
python - <<'PY'
import json
for v in (0, 13, 12):
    for l in open(f"gpurun_out/r02_c27/kb_v{v}.jsonl"):
        try: j = json.loads(l)
        except Exception: continue
        print(v, j.get("kernel", "")[:50], j.get("ms"), j.get("tflops"))
PY
for V in 0 13; do
  timeout 300 python bench.py --dtype bfloat16 --no-cpu-baseline --steps 3 --warmup 1 --tuning 3=$V > $OUT/bench_bf16_v$V.json 2> $OUT/e$V.log
done
python - <<'PY'
import json
for v in (0, 13):
    try:
        j = json.load(open(f"gpurun_out/r02_c27/bench_bf16_v{v}.json")); r = j["roofline"]
        print(v, round(j["value"]), round(j["ms_per_step"], 1), round(r["achieved"], 1), round(r["frac"], 3), r["gemm"]["tflops"], r["fraction_of_step_time"])
    except Exception as e:
        print(v, "failed", e)
PY
tail -3 $OUT/e13.log
echo "r02 call 27 done"
