#!/bin/bash
# r02 GPU call 46: O(N)-memory voxel paths (radix sort) vs the dense tables
set -u
OUT=gpurun_out/r02_c46; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_spinnet_gpu.py tests/test_pipeline_gpu.py tests/test_abi.py -m "gpu or not gpu" -x -q > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
echo "r02 call 46 done"
