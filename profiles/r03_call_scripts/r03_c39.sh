# fp32 few-token calls: one block per CU (unused dynamic LDS) for the 128 x 128 fp32 GEMM when its grid has at most N blocks (env-gated experiment)
OUT=gpurun_out/r03_c39; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-profile --gamma-scale 0 --dtype float32"
for geo in "--batch 1 --points 1024 --flow-steps 10" "--batch 1 --points 2048 --flow-steps 20"; do
  for n in 0 256 384 512; do
    echo "== float32 $geo | one block per CU up to $n blocks" >> $OUT/lat.txt
    RAP_GEMM32_ONE_PER_CU=$n timeout 300 $B $geo 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $OUT/lat.txt
  done
done
cat $OUT/lat.txt
