#!/bin/bash
# r02 GPU call 51: whole GPU suite with the packed-pair GEGLU as the fp32 default
set -u
OUT=gpurun_out/r02_c51; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
echo "r02 call 51 done"
