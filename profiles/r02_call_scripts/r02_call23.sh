#!/bin/bash
# r02 GPU call 23: fp32 GEMM epilogues with batched residual loads / unguarded full-tile stores: parity + per-shape TF + headline
set -u
OUT=gpurun_out/r02_c23; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_sample_gpu.py -m gpu -x -q -k "gemm or sample or forward" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python scripts/kernel_bench.py --only gemm > $OUT/kb_f32.jsonl 2> $OUT/kb.err
python - <<'PY'
import json
for l in open("gpurun_out/r02_c23/kb_f32.jsonl"):
    try: j = json.loads(l)
    except Exception: continue
    print(j.get("kernel", "")[:44], j.get("ms"), j.get("tflops"))
PY
timeout 400 python bench.py --no-secondary --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_f32.json 2> $OUT/e1.log
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02_c23/bench_f32.json")); r = j["roofline"]
print(round(j["value"]), round(j["ms_per_step"], 1), r["gemm"], r["fraction_of_step_time"])
PY
echo "r02 call 23 done"
