# fp32 few-token calls: split-K also for the K = 512 out-projection when its tile grid covers at most a quarter of the CUs
OUT=gpurun_out/r03_c38; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sample_gpu.py tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-profile --gamma-scale 0 --dtype float32"
for geo in "--batch 1 --points 1024 --flow-steps 10" "--batch 1 --points 2048 --flow-steps 20"; do
  for t in "" "--tuning 6=0"; do
    echo "== float32 $geo | $t" >> $OUT/lat.txt
    timeout 300 $B $geo $t 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $OUT/lat.txt
  done
done
cat $OUT/lat.txt
