#!/bin/bash
# First GPU call of the next round (≈ 1.5 GPU-minutes): is the 128x512 16-bit GEMM tile (variant 5, built blind at the end of
# round 1 -- DESIGN.md section 8.1) correct, and what does it buy per shape?
#   /usr/local/graft/bin/gpurun --timeout 400 -- 'bash scripts/next_round_first_call.sh'
set -u
OUT=gpurun_out/r02_first; mkdir -p $OUT
RAP_TEST_GEMM_H16_VARIANT=5 timeout 200 python -m pytest tests/test_h16_gpu.py -m gpu -q -k "gemm or qkv or geglu" > $OUT/pytest_h16_variant5.log 2>&1
tail -5 $OUT/pytest_h16_variant5.log
: > $OUT/kb.jsonl
for v in 1 5 1 5; do
  timeout 120 python scripts/kernel_bench.py --dtype bfloat16 --only gemm --h16-gemm-variant $v >> $OUT/kb.jsonl 2>> $OUT/kb.err
done
cut -c1-200 $OUT/kb.jsonl
