#!/bin/bash
# MFMA-pipe utilisation of the kernels the MODEL path launches (one flow step of the bench workload, fp32 and bf16), from SQ counters.
# A PMC-only pass (--kernel-trace, no other trace domain).  Usage (GPU box, repo root): bash scripts/mfma_util.sh <outdir> [bench.py args, e.g. --workload ragged]
set -u
OUT=$1; shift; mkdir -p "$OUT"; : > "$OUT/mfma_utilisation.txt"
export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
for dt in ${DTYPES:-float32 bfloat16}; do
  D=$(mktemp -d /tmp/mfu.XXXXXX)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$D" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --dtype $dt --steps 1 --warmup 0 \
      --flow-steps 1 --no-cpu-baseline --no-secondary --no-ragged --no-profile --gamma-scale 0 "$@" > "$GRAFT_REPO_ROOT/$OUT/mfma_$dt.log" 2>&1 )
  DB=$(find "$D" -name '*.db' | head -1)
  if [ -n "$DB" ]; then python "$GRAFT_REPO_ROOT/scripts/mfma_util.py" "$DB" $dt >> "$OUT/mfma_utilisation.txt"; else echo "$dt: no db" >> "$OUT/mfma_utilisation.txt"; tail -5 "$OUT/mfma_$dt.log"; fi
  rm -rf "$D"
done
cat "$OUT/mfma_utilisation.txt"
