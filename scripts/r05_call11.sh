#!/bin/bash
# round 5, GPU call 11: the bench line on the final tree with the few-token leg (short run)
set -u
OUT=gpurun_out/r05_c11
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python bench.py --steps 2 --warmup 1 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench exit $?"
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r05_c11/bench_default.json"))
    print({k: j.get(k) for k in ("value", "points_per_s_by_mode", "few_token_latency", "host_call_ms_idle_queue_unprofiled", "instrumented_over_clean")})
    print({k: (v.get("points_per_s") if isinstance(v, dict) else None) for k, v in (j.get("ragged") or {}).items() if k in ("f32", "f32x2", "bf16")})
    print((j["ragged"]["f32x2"]["roofline"] or {}).get("traffic"))
except Exception as e:
    print("no json", e)
PY
tail -2 "$OUT/bench_default.err"
echo "r05 call 11 done"
