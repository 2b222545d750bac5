#!/bin/bash
# r02 GPU call 40: fp32 GEMM tile choice per shape after the epilogue rewrite; RCCL path of bench.py with one rank; bf16 GEMM per shape after the revert
set -u
OUT=gpurun_out/r02_c40; mkdir -p $OUT
export TMPDIR=/tmp
for V in 48 16 32; do
  timeout 300 python scripts/kernel_bench.py --only gemm --gemm-variant $V 2>> $OUT/kb.err | sed "s/^/{\"variant\": $V, \"row\": /; s/$/}/" >> $OUT/kb_f32.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c40/kb_f32.jsonl"):
    try: j = json.loads(l)
    except Exception: continue
    r = j["row"]; print(j["variant"], r.get("kernel", "")[:44], round(r.get("ms"), 4), round(r.get("tflops"), 1))
PY
timeout 300 python scripts/kernel_bench.py --dtype bfloat16 --only gemm > $OUT/kb_bf16.jsonl 2>> $OUT/kb.err
python - <<'PY'
import json
for l in open("gpurun_out/r02_c40/kb_bf16.jsonl"):
    try: j = json.loads(l)
    except Exception: continue
    print(j.get("kernel", "")[:50], round(j.get("ms"), 4), round(j.get("tflops"), 1))
PY
RAP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --dtype bfloat16 --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench_bf16_rccl_1rank.json 2> $OUT/e_dist.log
tail -c 600 $OUT/bench_bf16_rccl_1rank.json; echo; tail -3 $OUT/e_dist.log
echo "r02 call 40 done"
