#!/bin/bash
# r02 GPU call 44: concurrent batch shards in the product path: parity tests, scaling with the shard count, bench lines
set -u
OUT=gpurun_out/r02_c44; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sample_gpu.py tests/test_h16_gpu.py tests/test_headline_gpu.py tests/test_parallel_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q -k "shards or model or headline or parallel or pipeline or sample or graph" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python scripts/stream_shards_bench.py --dtype bfloat16 --reps 3 > $OUT/shards_bf16.jsonl 2> $OUT/e1.log; cat $OUT/shards_bf16.jsonl
timeout 600 python scripts/stream_shards_bench.py --dtype float16 --reps 2 --streams 1,2 > $OUT/shards_f16.jsonl 2>> $OUT/e1.log; cat $OUT/shards_f16.jsonl
timeout 900 python scripts/stream_shards_bench.py --dtype float32 --reps 1 --streams 1,2,4 > $OUT/shards_f32.jsonl 2>> $OUT/e1.log; cat $OUT/shards_f32.jsonl
timeout 400 python bench.py --dtype bfloat16 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_bf16.json 2> $OUT/e2.log
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02_c44/bench_bf16.json")); r = j["roofline"]
print(round(j["value"]), round(j["ms_per_step"], 1), j.get("streams"), round(r["achieved"], 1), round(r["frac"], 3), r["gemm"]["tflops"], r["fraction_of_step_time"], r["measured_over"][:40])
PY
tail -3 $OUT/e2.log
echo "r02 call 44 done"
