#!/bin/bash
# r02 GPU call 52: fp32 qkv epilogue (DPP row sums, hoisted head-major addressing): parity + timing + headline
set -u
OUT=gpurun_out/r02_c52; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sample_gpu.py tests/test_headline_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python scripts/kernel_bench.py --only gemm > $OUT/kb.jsonl 2> $OUT/kb.err
python - <<'PY'
import json
for l in open("gpurun_out/r02_c52/kb.jsonl"):
    try: j = json.loads(l)
    except Exception: continue
    print(j.get("kernel", "")[:40], round(j.get("ms"), 4), round(j.get("tflops"), 1))
PY
timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench_f32.json 2> $OUT/e.log
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02_c52/bench_f32.json")); r = j["roofline"]
print(round(j["value"]), round(j["ms_per_step"], 1), r["gemm"]["tflops"], j["parity_vs_reference_golden"]["final_cloud_max_abs"], j["parity_vs_reference_golden"]["per_step_max_abs"])
PY
echo "r02 call 52 done"
