#!/bin/bash
# r02 GPU call 57: selection (NaN semantics) + collate tests on the final tree
set -u
OUT=gpurun_out/r02_c57; mkdir -p $OUT
timeout 150 python -m pytest tests/test_sample_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "selection or select or collate or rigidity" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
echo "r02 call 57 done"
