#!/bin/bash
# round 5, last GPU call: the tests added after the final suite run, the restored idle-queue host probe, the RCCL path in strong-scaling mode with a one-rank group
set -u
OUT=gpurun_out/r05_c7
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_x2_gpu.py -q -k "graph_replay or small_calls" > "$OUT/pytest_x2_new.log" 2>&1; echo "new x2 tests exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_x2_new.log" | tail -5
timeout 600 python bench.py --steps 3 --warmup 2 --no-ragged --gamma-scale 0 --no-secondary --no-cpu-baseline > "$OUT/bench_idle_probe.json" 2> "$OUT/bench_idle_probe.err"; echo "bench exit $?"
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r05_c7/bench_idle_probe.json")); print({k: j.get(k) for k in ("value", "host_call_ms_first_timed_call", "host_call_ms_idle_queue_unprofiled", "host_call_ms_per_step")})
except Exception as e:
    print("no json", e)
PY
RAP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --scaling strong --batch 8 --steps 2 --warmup 1 --no-ragged --gamma-scale 0 --no-secondary --no-cpu-baseline > "$OUT/bench_rccl_strong_one_rank.json" 2> "$OUT/bench_rccl_strong.err"; echo "strong exit $?"
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r05_c7/bench_rccl_strong_one_rank.json")); print({k: j.get(k) for k in ("value", "scaling", "rccl_ranks", "pairs_total", "sharding", "all_gather_ms_per_step")})
except Exception as e:
    print("no json", e)
PY
RAP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29578 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --scaling strong --workload ragged --dtype float32x2 --ragged-points 131072 --steps 1 --warmup 1 --gamma-scale 0 --no-secondary --no-cpu-baseline > "$OUT/bench_rccl_strong_ragged_x2.json" 2> "$OUT/bench_rccl_strong_ragged.err"; echo "strong ragged exit $?"
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r05_c7/bench_rccl_strong_ragged_x2.json")); print({k: j.get(k) for k in ("value", "dtype", "scaling", "pairs_total", "sharding")})
except Exception as e:
    print("no json", e)
PY
tail -3 "$OUT/bench_rccl_strong_ragged.err"
echo "r05 call 7 done"
