#!/bin/bash
# r02 GPU call 26: instruction-mix microbenchmark of the 16-bit attention loop (no global traffic): which structure keeps the matrix pipe busy?
set -u
OUT=gpurun_out/r02_c26; mkdir -p $OUT
timeout 300 python scripts/attn_mix.py > $OUT/attn_mix.jsonl 2> $OUT/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r02_c26/attn_mix.jsonl"):
    j = json.loads(l); print(j["query_blocks_per_wave"], j["mode"], j["waves_per_block"], j["waves_per_simd"], j["blocks_per_cu"], j["tflops"], j["what"])
PY
tail -3 $OUT/err.log
echo "r02 call 26 done"
