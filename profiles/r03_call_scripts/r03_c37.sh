# full GPU suite + smoke + few-token latency on the tree with split-K ff2 and the one-block-per-CU rule of the fp32 split attention
OUT=gpurun_out/r03_c37; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-profile --gamma-scale 0"
for dt in float32 bfloat16; do
for geo in "--batch 1 --points 1024 --flow-steps 10" "--batch 1 --points 2048 --flow-steps 20" "--batch 1 --points 4096 --flow-steps 20"; do
    echo "== $dt $geo" >> $OUT/lat.txt
    timeout 300 $B --dtype $dt $geo 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $OUT/lat.txt
done
done
cat $OUT/lat.txt
HEAD=8 bash scripts/prof_bench.sh $OUT/demo_f32_trace --dtype float32 --batch 1 --points 1024 --flow-steps 10
