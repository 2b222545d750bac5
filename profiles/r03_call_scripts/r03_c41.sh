# prototype (scripts/gemm_w4.hip): bf16 GEMM with four waves per 256 x 256 tile (one wave per SIMD, 128 x 128 per wave) vs the shipped kernel, same shapes, plain bf16 store
OUT=gpurun_out/r03_c41; mkdir -p $OUT
( cd scripts && timeout 400 python gemm_w4.py > ../$OUT/gemm_w4.jsonl 2> ../$OUT/gemm_w4.err ); cat $OUT/gemm_w4.jsonl; tail -3 $OUT/gemm_w4.err
timeout 300 python scripts/gemm_square.py 1 0 > $OUT/gemm_square_shipped.jsonl 2> $OUT/gemm_square.err; cat $OUT/gemm_square_shipped.jsonl
