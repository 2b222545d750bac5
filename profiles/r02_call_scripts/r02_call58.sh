#!/bin/bash
# r02 GPU call 58: tuning knobs after the atomic change: a variant A/B test and the smoke path
set -u
OUT=gpurun_out/r02_c58; mkdir -p $OUT
timeout 100 python -m pytest tests/test_h16_gpu.py tests/test_sample_gpu.py -m gpu -q -x -k "schedule_variants or fused_qknorm or splits or split" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
echo "r02 call 58 done"
