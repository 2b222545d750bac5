#!/bin/bash
# r02 GPU call 49: parity of the persistent GEMM variant (every epilogue, both dtypes)
set -u
OUT=gpurun_out/r02_c49; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_h16_gpu.py -m gpu -x -q -k "persistent_prefetch" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
echo "r02 call 49 done"
