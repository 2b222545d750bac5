# prototype (scripts/gemm_w4.hip): epilogue through a per-wave LDS slab vs direct 8-byte stores vs main loop only
OUT=gpurun_out/r03_c43; mkdir -p $OUT
( cd scripts && timeout 400 python gemm_w4.py --epilogue > ../$OUT/gemm_w4_epilogue.jsonl 2> ../$OUT/gemm_w4.err ); cat $OUT/gemm_w4_epilogue.jsonl; tail -3 $OUT/gemm_w4.err
