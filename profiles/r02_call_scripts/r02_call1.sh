#!/bin/bash
# r02 GPU call 1: what bounds the 16-bit GEMM?  (a) 128x512 tile (variant 5) parity + timing, (b) L2 hit / fabric-read counters of
# the default and the 128x512 kernels, (c) do same-line requests from the CUs of one XCD merge in the L2 (ldsdma_fill --shared).
set -u
OUT=gpurun_out/r02_c1; mkdir -p $OUT
export TMPDIR=/tmp
RAP_TEST_GEMM_H16_VARIANT=5 timeout 200 python -m pytest tests/test_h16_gpu.py -m gpu -q -k "gemm or qkv or geglu" > $OUT/pytest_h16_variant5.log 2>&1
tail -3 $OUT/pytest_h16_variant5.log
: > $OUT/kb.jsonl
for v in 1 5 1 5; do
  timeout 120 python scripts/kernel_bench.py --dtype bfloat16 --only gemm --h16-gemm-variant $v >> $OUT/kb.jsonl 2>> $OUT/kb.err
done
cut -c1-220 $OUT/kb.jsonl
timeout 200 python scripts/ldsdma_fill.py --shared > $OUT/ldsdma_shared.jsonl 2> $OUT/ldsdma.err
cat $OUT/ldsdma_shared.jsonl
for v in 1 5; do
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace -d "$GRAFT_REPO_ROOT/$OUT/pmc_v${v}_$tag" -o pmc -- \
        python "$GRAFT_REPO_ROOT/scripts/kernel_bench.py" --dtype bfloat16 --only gemm --h16-gemm-variant $v > "$GRAFT_REPO_ROOT/$OUT/pmc_v${v}_$tag.log" 2>&1 )
    DB=$(find "$OUT/pmc_v${v}_$tag" -name '*.db' | head -1)
    if [ -n "$DB" ]; then python scripts/rocpd_summary.py "$DB" --pmc | grep -E "^PMC|gemm_h16" | sed "s/^/v$v /" >> "$OUT/pmc_gemm_h16.txt"; fi
    find "$OUT/pmc_v${v}_$tag" -name '*.db' -delete
  done
done
cat $OUT/pmc_gemm_h16.txt
echo "r02 call 1 done"
