#!/bin/bash
# r02 GPU call 56: last call of the round: the whole GPU suite on the final tree
set -u
OUT=gpurun_out/r02_c56; mkdir -p $OUT
export TMPDIR=/tmp
timeout 330 python -m pytest tests/ -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
echo "r02 call 56 done"
