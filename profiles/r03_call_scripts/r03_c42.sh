# ablation of the prototype's main loop (scripts/gemm_w4.hip): full / no LDS-DMA / no fragment reads / MFMA + barrier only
OUT=gpurun_out/r03_c42; mkdir -p $OUT
( cd scripts && timeout 400 python gemm_w4.py --ablation > ../$OUT/gemm_w4_ablation.jsonl 2> ../$OUT/gemm_w4.err ); cat $OUT/gemm_w4_ablation.jsonl; tail -3 $OUT/gemm_w4.err
