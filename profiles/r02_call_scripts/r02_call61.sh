#!/bin/bash
# r02 GPU call 61: rap_16 / rap_10 fixtures on the GPU (2 tests)
set -u
OUT=gpurun_out/r02_c61; mkdir -p $OUT
timeout 25 python -m pytest tests/test_sample_gpu.py -m gpu -q -x -s -k "other_model_sizes" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
echo "r02 call 61 done"
