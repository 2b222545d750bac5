# the other BASELINE geometries on the final tree of round 3 (parity-test cases, not bench lines): configs[3] fp32 + bf16, configs[4] bf16
OUT=gpurun_out/r03_c40; mkdir -p $OUT
timeout 600 python bench.py --steps 2 --warmup 1 --batch 16 --views 8 --points 2048 --flow-steps 30 --no-cpu-baseline --gamma-scale 0 > $OUT/bench_c3_f32_and_bf16.json 2> $OUT/c3.err; echo "c3 exit $?"
timeout 600 python bench.py --steps 2 --warmup 1 --dtype bfloat16 --batch 4 --views 2 --points 32768 --flow-steps 50 --no-cpu-baseline --gamma-scale 0 > $OUT/bench_c4_bf16.json 2> $OUT/c4.err; echo "c4 exit $?"
python - <<'PY'
import json
for f in ("bench_c3_f32_and_bf16", "bench_c4_bf16"):
    d = json.load(open(f"gpurun_out/r03_c40/{f}.json"))
    r = d["roofline"]; print(f, d["dtype"], round(d["value"]), round(d["ms_per_step"]), round(r["achieved"], 1), round(r["frac"], 3), round(r["gemm"]["tflops"], 1))
    s = d.get("reduced_precision")
    if s: print("   bf16", round(s["value"]), round(s["ms_per_step"]), round(s["roofline"]["achieved"], 1), round(s["roofline"]["gemm"]["tflops"], 1))
PY
