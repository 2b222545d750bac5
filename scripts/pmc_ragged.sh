#!/bin/bash
# HBM-side traffic of the attention launches of the RAGGED reference-regime batch (round 4): does a 30-40 k-token sample, whose K / V
# of one head (20 MB in fp32) exceed an XCD's 4 MB L2, still move only its algorithmic bytes?  rocprofv3 --pmc FETCH_SIZE and --pmc
# WRITE_SIZE in SEPARATE passes with --kernel-trace only (MI355X_MICROARCH.md: HBM / rocprofv3 section), MODEL path, one flow step.
# Usage (GPU box, repo root): bash scripts/pmc_ragged.sh <outdir>
set -u
OUT=$1; mkdir -p "$OUT"; : > "$OUT/pmc_traffic_ragged.txt"
export TMPDIR=/tmp
run() {   # tag counter cmd...
  local tag=$1 c=$2; shift 2
  local D; D=$(mktemp -d /tmp/pmc.XXXXXX)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d "$D" -o pmc -- "$@" > "$GRAFT_REPO_ROOT/$OUT/${tag}_$c.log" 2>&1 )
  local DB; DB=$(find "$D" -name '*.db' | head -1)
  if [ -n "$DB" ]; then python "$GRAFT_REPO_ROOT/scripts/rocpd_summary.py" "$DB" --pmc | grep -E "^PMC.*attention" | sed "s/^/$tag /" >> "$OUT/pmc_traffic_ragged.txt"; else echo "$tag $c: no db" >> "$OUT/pmc_traffic_ragged.txt"; fi
  rm -rf "$D"
}
for c in FETCH_SIZE WRITE_SIZE; do
  run ragged_f32_model_path $c python "$GRAFT_REPO_ROOT/bench.py" --workload ragged --steps 1 --warmup 0 --flow-steps 1 --no-cpu-baseline --no-secondary --no-profile --gamma-scale 0
  run ragged_bf16_model_path $c python "$GRAFT_REPO_ROOT/bench.py" --workload ragged --dtype bfloat16 --steps 1 --warmup 0 --flow-steps 1 --no-cpu-baseline --no-secondary --no-profile --gamma-scale 0
done
cat "$OUT/pmc_traffic_ragged.txt"
