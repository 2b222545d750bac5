#!/bin/bash
# ONE script for the evidence the measurement rows cite, for the library that is in the tree when it runs:
#   1. rocprofv3 --kernel-trace of ONE configs[1] sampling call in each of the four arithmetic modes     -> <out>/trace_<mode>.txt
#   2. PMC FETCH_SIZE and WRITE_SIZE (separate passes, --kernel-trace only) of one FLOW STEP of the configs[1] batch through the MODEL path,
#      fp32 / split precision / bf16 / fp16 -- every attention and GEMM symbol rap_sample launches        -> <out>/pmc_<mode>_<counter>.txt
#      and their summary                                                                                   -> <out>/pmc_traffic.json
#   3. SQ pass (MFMA busy cycles, clock) of the same one-step run                                          -> <out>/mfma_utilisation.txt
#   4. the few-token call (configs[0] geometry) traced in bf16 and split precision                         -> <out>/trace_c0_<mode>.txt
# Usage (GPU box, repo root, through gpurun):  bash scripts/evidence.sh gpurun_out/<tag> [stages: trace pmc mfma c0]
# Copy the files to profiles/r<NN>_final_* and profiles/pmc_traffic.json afterwards (bench.py reads the latter for roofline.traffic).
set -u
OUT=${1:?outdir}; shift || true
STAGES=${*:-"trace pmc mfma c0"}      # + "ragged": the same two PMC passes on the ragged reference-regime batch (fp32, split precision, bf16)
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
MODES="float32 float32x2 bfloat16 float16"
ONE_STEP="--steps 1 --warmup 0 --flow-steps 1 --no-cpu-baseline --light --no-profile"

if has trace; then
  for m in $MODES; do HEAD=14 timeout 600 bash scripts/prof_bench.sh "$OUT/trace_$m" --dtype $m; done
fi
if has ragged; then
  for m in float32 float32x2 bfloat16; do
    for c in FETCH_SIZE WRITE_SIZE; do
      D=$(mktemp -d /tmp/pmc.XXXXXX)
      ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d "$D" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --workload ragged --dtype $m $ONE_STEP > "$GRAFT_REPO_ROOT/$OUT/pmc_ragged_${m}_$c.log" 2>&1 )
      DB=$(find "$D" -name '*.db' | head -1)
      if [ -n "$DB" ]; then python scripts/rocpd_summary.py "$DB" --pmc | grep -E "^PMC.*attention" > "$OUT/pmc_ragged_${m}_$c.txt"; else echo "no db" > "$OUT/pmc_ragged_${m}_$c.txt"; fi
      rm -rf "$D"
    done
  done
  cat "$OUT"/pmc_ragged_*_FETCH_SIZE.txt "$OUT"/pmc_ragged_*_WRITE_SIZE.txt
fi
if has pmc; then
  for m in $MODES; do
    for c in FETCH_SIZE WRITE_SIZE; do
      D=$(mktemp -d /tmp/pmc.XXXXXX)
      ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d "$D" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --dtype $m $ONE_STEP > "$GRAFT_REPO_ROOT/$OUT/pmc_${m}_$c.log" 2>&1 )
      DB=$(find "$D" -name '*.db' | head -1)
      if [ -n "$DB" ]; then python scripts/rocpd_summary.py "$DB" --pmc | grep -E "^PMC" > "$OUT/pmc_${m}_$c.txt"; else echo "no db" > "$OUT/pmc_${m}_$c.txt"; tail -3 "$OUT/pmc_${m}_$c.log"; fi
      rm -rf "$D"
    done
  done
  python scripts/pmc_to_json.py "$OUT" > "$OUT/pmc_traffic.json" && python - "$OUT/pmc_traffic.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for k, v in j["kernels"].items():
    print(f"{k[:72]:72s} {v['measured_on']:10s} dispatches {v['dispatches']:4d}  HBM {v['hbm_bytes_per_launch'] / 1e9:8.3f} GB/launch" + (f"  algorithmic {v['algorithmic_bytes_per_launch'] / 1e9:.3f}" if v.get("algorithmic_bytes_per_launch") else ""))
PY
fi
if has mfma; then
  DTYPES="$MODES" bash scripts/mfma_util.sh "$OUT" > /dev/null; cat "$OUT/mfma_utilisation.txt"
fi
if has c0; then
  HEAD=16 timeout 300 bash scripts/prof_bench.sh "$OUT/trace_c0_bfloat16" --dtype bfloat16 --config 0
  HEAD=16 timeout 300 bash scripts/prof_bench.sh "$OUT/trace_c0_float32x2" --dtype float32x2 --config 0 --tuning 17=0
fi
echo "evidence done: $OUT"
