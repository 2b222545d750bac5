#!/bin/bash
# r02 GPU call 29: decomposition of the K/V stream cost in the attention loop microbenchmark
set -u
OUT=gpurun_out/r02_c29; mkdir -p $OUT
timeout 300 python scripts/attn_mix.py > $OUT/attn_mix.jsonl 2> $OUT/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r02_c29/attn_mix.jsonl"):
    j = json.loads(l); print(j["query_blocks_per_wave"], j["mode"], j["waves_per_block"], j["waves_per_simd"], j["blocks_per_cu"], j["tflops"], j["what"])
PY
tail -2 $OUT/err.log
echo "r02 call 29 done"
