#!/bin/bash
# r02 GPU call 39: 16-bit GEMM epilogues: packed-pair GEGLU, rolling residual prefetch + full-tile stores: parity + per-shape timing + bench
set -u
OUT=gpurun_out/r02_c39; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_h16_gpu.py -m gpu -x -q -k "gemm" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python scripts/kernel_bench.py --dtype bfloat16 --only gemm > $OUT/kb.jsonl 2> $OUT/kb.err
python - <<'PY'
import json
for l in open("gpurun_out/r02_c39/kb.jsonl"):
    try: j = json.loads(l)
    except Exception: continue
    print(j.get("kernel", "")[:50], round(j.get("ms"), 4), round(j.get("tflops"), 1))
PY
timeout 300 python bench.py --dtype bfloat16 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_bf16.json 2> $OUT/e.log
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02_c39/bench_bf16.json")); r = j["roofline"]
print(round(j["value"]), round(j["ms_per_step"], 1), round(r["achieved"], 1), round(r["frac"], 3), r["gemm"]["tflops"], r["fraction_of_step_time"])
PY
echo "r02 call 39 done"
