#!/bin/bash
# r02 GPU call 43: one batch on one stream vs two half-batches on two streams
set -u
OUT=gpurun_out/r02_c43; mkdir -p $OUT
timeout 600 python scripts/two_stream_bench.py --dtype bfloat16 --reps 3 > $OUT/two_stream_bf16.json 2> $OUT/e1.log; cat $OUT/two_stream_bf16.json; tail -2 $OUT/e1.log
timeout 900 python scripts/two_stream_bench.py --dtype float32 --reps 1 > $OUT/two_stream_f32.json 2> $OUT/e2.log; cat $OUT/two_stream_f32.json; tail -2 $OUT/e2.log
echo "r02 call 43 done"
