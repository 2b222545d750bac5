#!/bin/bash
# round 5, GPU call 14: the kernel-level GEMM / attention tests against the library with the host-side stride validation (the last
# change of the round): no legitimate call may be refused
set -u
OUT=gpurun_out/r05_c14
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_x2_gpu.py tests/test_h16_gpu.py -x -q -m gpu > "$OUT/pytest_kernel_level.log" 2>&1; echo "tests exit $?"
tail -4 "$OUT/pytest_kernel_level.log"
echo "r05 call 14 done"
