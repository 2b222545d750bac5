#!/bin/bash
# r02 GPU call 45: the batch as sequential chunks on one stream (smaller per-layer working set vs the 256 MB Infinity Cache)
set -u
OUT=gpurun_out/r02_c45; mkdir -p $OUT
timeout 900 python scripts/stream_shards_bench.py --dtype bfloat16 --reps 2 --sequential --streams 1,2,4,8,16 > $OUT/chunks_bf16.jsonl 2> $OUT/e1.log; cat $OUT/chunks_bf16.jsonl; tail -2 $OUT/e1.log
echo "r02 call 45 done"
