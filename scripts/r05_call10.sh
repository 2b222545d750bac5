#!/bin/bash
# round 5, GPU call 10: the whole GPU suite + smoke on the final tree (after the few-token forms of the split-precision path)
set -u
OUT=gpurun_out/r05_c10
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_gpu.log" | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke exit $?"; tail -2 "$OUT/smoke.log"
echo "r05 call 10 done"
