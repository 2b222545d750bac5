OUT=gpurun_out/r03_c34; mkdir -p $OUT
timeout 900 python -m pytest tests/test_h16_gpu.py -x -q -k "splitk or sub_blocks or ragged or lds_dma or forward_h16 or sample_h16 or single_token or sharp" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
B="python bench.py --dtype bfloat16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-profile --gamma-scale 0"
for geo in "--batch 1 --points 1024 --flow-steps 10" "--batch 1 --points 2048 --flow-steps 20" "--batch 1 --points 4096 --flow-steps 20"; do
  for t in "" "--tuning 5=0" "--tuning 6=0" "--tuning 5=0 --tuning 6=0"; do
    echo "== $geo | $t" >> $OUT/lat.txt
    timeout 300 $B $geo $t 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $OUT/lat.txt
  done
done
cat $OUT/lat.txt
