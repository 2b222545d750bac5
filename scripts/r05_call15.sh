#!/bin/bash
# round 5, GPU call 15: smoke() and the golden-fixture sampling tests against the library as committed last (header / validation changes)
set -u
OUT=gpurun_out/r05_c15
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke exit $?"; tail -4 "$OUT/smoke.txt"
timeout 110 python -m pytest tests/test_sample_gpu.py -x -q -m gpu -k "golden" > "$OUT/pytest_golden.log" 2>&1; echo "tests exit $?"
tail -3 "$OUT/pytest_golden.log"
echo "r05 call 15 done"
