#!/bin/bash
# r02 GPU call 42: end-to-end pipeline test (raw scans -> transform files)
set -u
OUT=gpurun_out/r02_c42; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -s > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log
echo "r02 call 42 done"
