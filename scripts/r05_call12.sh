#!/bin/bash
# round 5, GPU call 12: the DRIVER's bench command (--gpus 1 --steps 20 --warmup 5) on the final tree, every leg, timed end to end
set -u
OUT=gpurun_out/r05_c12
mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
timeout 1000 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "bench exit $?"
T1=$(date +%s)
echo "wall_s $((T1 - T0))" | tee "$OUT/wall.txt"
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r05_c12/bench_driver_cmd.json"))
    print({k: j.get(k) for k in ("value", "ms_per_step", "points_per_s_by_mode", "few_token_latency", "instrumented_over_clean")})
    print({k: j["roofline"].get(k) for k in ("achieved", "frac", "avg_launch_ms", "traffic")})
except Exception as e:
    print("no json", e)
PY
tail -2 "$OUT/bench_driver_cmd.err"
echo "r05 call 12 done"
