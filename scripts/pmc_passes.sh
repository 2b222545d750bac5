#!/bin/bash
# HBM traffic of the dominant kernel (attention), per launch, for every instantiation rap_sample can launch: rocprofv3 --pmc FETCH_SIZE
# and --pmc WRITE_SIZE in SEPARATE passes with --kernel-trace only (MI355X_MICROARCH.md: HBM / rocprofv3 section).
#   fp32  bounded + online          : scripts/kernel_bench.py --only attention --pmc --bounded 2      (the symbols the model path launches)
#   bf16  online                    : same with --dtype bfloat16 --bounded 0
#   bf16  bounded, pre-scaled q     : the MODEL path (bench.py, 1 flow step): that instantiation is only reachable through rap_sample
# Usage (GPU box, repo root): bash scripts/pmc_passes.sh <outdir>
set -u
OUT=$1; mkdir -p "$OUT"; : > "$OUT/pmc_traffic.txt"
export TMPDIR=/tmp
run() {   # tag counter cmd...
  local tag=$1 c=$2; shift 2
  local D; D=$(mktemp -d /tmp/pmc.XXXXXX)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d "$D" -o pmc -- "$@" > "$GRAFT_REPO_ROOT/$OUT/${tag}_$c.log" 2>&1 )
  local DB; DB=$(find "$D" -name '*.db' | head -1)
  if [ -n "$DB" ]; then python "$GRAFT_REPO_ROOT/scripts/rocpd_summary.py" "$DB" --pmc | grep -E "^PMC.*attention" | sed "s/^/$tag /" >> "$OUT/pmc_traffic.txt"; else echo "$tag $c: no db" >> "$OUT/pmc_traffic.txt"; fi
  rm -rf "$D"
}
for c in FETCH_SIZE WRITE_SIZE; do
  run f32_kernel_level $c python "$GRAFT_REPO_ROOT/scripts/kernel_bench.py" --only attention --pmc --bounded 2
  run bf16_kernel_level $c python "$GRAFT_REPO_ROOT/scripts/kernel_bench.py" --dtype bfloat16 --only attention --pmc --bounded 2
  run f16_kernel_level $c python "$GRAFT_REPO_ROOT/scripts/kernel_bench.py" --dtype float16 --only attention --pmc --bounded 0
  run bf16_model_path $c python "$GRAFT_REPO_ROOT/bench.py" --dtype bfloat16 --steps 1 --warmup 0 --flow-steps 1 --no-cpu-baseline --no-secondary --no-ragged --no-profile --gamma-scale 0
done
cat "$OUT/pmc_traffic.txt"
