#!/bin/bash
# r02 GPU call 38: the 16-bit GEMM on the guide's square benchmark shapes vs the model's shapes
set -u
OUT=gpurun_out/r02_c38; mkdir -p $OUT
timeout 400 python scripts/gemm_square.py 14 1 > $OUT/gemm_square.jsonl 2> $OUT/err.log
cat $OUT/gemm_square.jsonl; tail -3 $OUT/err.log
echo "r02 call 38 done"
