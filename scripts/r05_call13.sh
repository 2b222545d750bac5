#!/bin/bash
# round 5, GPU call 13: kernel traces of ONE configs[0]-size call (1 pair x 2 x 1024 points, 10 flow steps) on the final tree: split
# precision in its few-token forms, and bf16 -- the per-kernel durations behind the small-call figures of DESIGN section 5
set -u
OUT=gpurun_out/r05_c13
mkdir -p "$OUT"
export TMPDIR=/tmp
HEAD=32 timeout 200 bash scripts/prof_bench.sh "$OUT/c0_geometry_f32x2_kernel_trace" --dtype float32x2 --batch 1 --points 1024 --flow-steps 10 --tuning 17=0
HEAD=32 timeout 120 bash scripts/prof_bench.sh "$OUT/c0_geometry_bf16_kernel_trace" --dtype bfloat16 --batch 1 --points 1024 --flow-steps 10
echo "r05 call 13 done"
