#!/bin/bash
# r02 GPU call 37: the whole GPU suite on the current tree + smoke + default bench
set -u
OUT=gpurun_out/r02_c37; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
echo "r02 call 37 done"
