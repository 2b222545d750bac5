# the power argument: the prototype's MFMA-only loop and full main loop on random vs all-zero operands (8192^3)
OUT=gpurun_out/r03_c44; mkdir -p $OUT
( cd scripts && timeout 300 python gemm_w4.py --zeros > ../$OUT/gemm_w4_zeros.jsonl 2> ../$OUT/gemm_w4.err ); cat $OUT/gemm_w4_zeros.jsonl; tail -3 $OUT/gemm_w4.err
