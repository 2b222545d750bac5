#!/bin/bash
# rocprofv3 kernel trace of ONE sample call of bench.py (no inline HIP-event profiling, no CPU leg) -> per-kernel summary text.
# Usage (on the GPU box, from the repo root):  bash scripts/prof_bench.sh <out-prefix> [bench.py args...]
set -u
OUT=$1; shift
export TMPDIR=/tmp
D=$(mktemp -d /tmp/prof.XXXXXX)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$D" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-ragged --no-profile --gamma-scale 0 "$@" > "$GRAFT_REPO_ROOT/$OUT.log" 2>&1 )
DB=$(find "$D" -name '*.db' | head -1)
if [ -n "$DB" ]; then python "$GRAFT_REPO_ROOT/scripts/rocpd_summary.py" "$DB" > "$GRAFT_REPO_ROOT/$OUT.txt"; head -${HEAD:-14} "$GRAFT_REPO_ROOT/$OUT.txt"; else echo "no rocprof db"; tail -5 "$GRAFT_REPO_ROOT/$OUT.log"; fi
rm -rf "$D"
