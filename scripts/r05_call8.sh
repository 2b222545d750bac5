#!/bin/bash
# round 5, GPU call 8: HBM traffic of the split-precision attention on the RAGGED batch (separate FETCH / WRITE passes), the amended test
set -u
OUT=gpurun_out/r05_c8
mkdir -p "$OUT"; : > "$OUT/pmc_traffic_ragged_x2.txt"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  D=$(mktemp -d /tmp/pmc.XXXXXX)
  ( cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace -d "$D" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --workload ragged --dtype float32x2 --steps 1 --warmup 0 --flow-steps 1 \
      --no-cpu-baseline --no-secondary --no-profile --gamma-scale 0 > "$GRAFT_REPO_ROOT/$OUT/pmc_$c.log" 2>&1 )
  DB=$(find "$D" -name '*.db' | head -1)
  if [ -n "$DB" ]; then python scripts/rocpd_summary.py "$DB" --pmc | grep -E "^PMC.*attention" | sed "s/^/ragged_x2_model_path /" >> "$OUT/pmc_traffic_ragged_x2.txt"; else echo "$c: no db" >> "$OUT/pmc_traffic_ragged_x2.txt"; fi
  rm -rf "$D"
done
cat "$OUT/pmc_traffic_ragged_x2.txt"
timeout 300 python -m pytest tests/test_x2_gpu.py -q -k "graph_replay" > "$OUT/pytest_x2_graph.log" 2>&1; echo "test exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_x2_graph.log" | tail -3
echo "r05 call 8 done"
