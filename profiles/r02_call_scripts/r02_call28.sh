#!/bin/bash
# r02 GPU call 28: variant 13 after the trans-use hazard fix: parity; attention loop microbenchmark with the K/V stream added
set -u
OUT=gpurun_out/r02_c28; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_h16_gpu.py -m gpu -x -q -k "schedule_variants or pipelined_variants" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python scripts/attn_mix.py > $OUT/attn_mix.jsonl 2> $OUT/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r02_c28/attn_mix.jsonl"):
    j = json.loads(l); print(j["query_blocks_per_wave"], j["mode"], j["waves_per_block"], j["waves_per_simd"], j["blocks_per_cu"], j["tflops"], j["what"])
PY
for V in 0 13; do
  timeout 200 python scripts/kernel_bench.py --dtype bfloat16 --only attention --h16-attn-variant $V > $OUT/kb_v$V.jsonl 2>> $OUT/kb.err
done
python - <<'PY'
import json
for v in (0, 13):
    for l in open(f"gpurun_out/r02_c28/kb_v{v}.jsonl"):
        try: j = json.loads(l)
        except Exception: continue
        print(v, j.get("kernel", "")[:50], j.get("ms"), j.get("tflops"))
PY
echo "r02 call 28 done"
