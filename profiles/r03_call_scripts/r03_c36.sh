# few-token calls: does forcing ONE block per CU (extra dynamic LDS) for launches with <= 320 blocks help?  (env-gated experiment)
OUT=gpurun_out/r03_c36; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-profile --gamma-scale 0"
for dt in bfloat16 float32; do
for geo in "--batch 1 --points 1024 --flow-steps 10" "--batch 1 --points 2048 --flow-steps 20"; do
  for e in "A=0 G=0" "A=1 G=0" "A=0 G=1" "A=1 G=1"; do
    eval $e
    echo "== $dt $geo | attn_one_per_cu=$A gemm_one_per_cu=$G" >> $OUT/lat.txt
    RAP_ATTN_ONE_PER_CU=$A RAP_GEMM_ONE_PER_CU=$G timeout 300 $B --dtype $dt $geo 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $OUT/lat.txt
  done
done
done
cat $OUT/lat.txt
