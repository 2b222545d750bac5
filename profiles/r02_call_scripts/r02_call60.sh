#!/bin/bash
# r02 GPU call 60: the 16-bit suite on the final tree (the default attention kernel gained a template parameter after the last full run)
set -u
OUT=gpurun_out/r02_c60; mkdir -p $OUT
timeout 75 python -m pytest tests/test_h16_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
echo "r02 call 60 done"
