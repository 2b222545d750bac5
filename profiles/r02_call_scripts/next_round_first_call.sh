#!/bin/bash
# First GPU call of the next round (about 3 GPU-minutes): re-establish the state the round ended in, on the box of the day.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash scripts/next_round_first_call.sh'
# What round 2 left open, in order of expected value:
#   1. overlap of the 16-bit GEMM epilogues (GEGLU math, qk-norm, the fp32 residual read-modify-write) with another tile's k-loop;
#   2. a 16-bit attention kernel in the instruction-mix microbenchmark's best arrangement (scripts/attn_mix.py) whose K/V stream does
#      not cost the 20 % the register-staged one does; 512-query blocks (half the stream per MFMA) are the cheap first probe;
#   3. small epilogue diets that are left: the V^T image of the QKV GEMM is written with 2-byte LDS stores (four consecutive tokens are
#      contiguous in vt_pos order: one ds_write_b64), the fp32 qk-norm divides twice per row and lane.
set -u
OUT=gpurun_out/r03_first; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python scripts/kernel_bench.py --only gemm > $OUT/kb_f32.jsonl 2> $OUT/kb.err
timeout 200 python scripts/kernel_bench.py --dtype bfloat16 > $OUT/kb_bf16.jsonl 2>> $OUT/kb.err
timeout 200 python scripts/attn_mix.py > $OUT/attn_mix.jsonl 2>> $OUT/kb.err
timeout 400 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
for f in ("kb_f32", "kb_bf16"):
    for l in open(f"gpurun_out/r03_first/{f}.jsonl"):
        try: j = json.loads(l)
        except Exception: continue
        print(f, j.get("kernel", "")[:48], round(j.get("ms", 0), 4), round(j.get("tflops", 0), 1) if "tflops" in j else round(j.get("GBps", 0)))
j = json.load(open("gpurun_out/r03_first/bench.json")); r = j["roofline"]
print("fp32", round(j["value"]), round(r["frac"], 4), r["gemm"]["tflops"], "| bf16", round(j["reduced_precision"]["value"]))
PY
