#!/bin/bash
# r02 GPU call 55: the 16-bit suites with the row-store epilogue as the attention default
set -u
OUT=gpurun_out/r02_c55; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_h16_gpu.py tests/test_headline_gpu.py tests/test_parallel_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
echo "r02 call 55 done"
