#!/bin/bash
# r02 GPU call 3: is the 16-bit GEMM's k-tile fetch camping on a few L2 channels?  Row strides of 1 KB / 4 KB (K = 512 / 2048 bf16) vs
# padded strides (K + 64 elements = +128 B, K + 8 = +16 B), default kernel (1) and the five-stage ring (6).
set -u
OUT=gpurun_out/r02_c3; mkdir -p $OUT
: > $OUT/kb.jsonl
for pad in 0 64 0 64 8 72; do
  for v in 1 6; do
    timeout 120 python scripts/kernel_bench.py --dtype bfloat16 --only gemm --h16-gemm-variant $v --pad-lda $pad >> $OUT/kb.jsonl 2>> $OUT/kb.err
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c3/kb.jsonl"):
    j = json.loads(l); print(j["variant"], j["pad_lda"], j["kernel"][:28], round(j["ms"], 3), round(j["tflops"]))
PY
echo "r02 call 3 done"
