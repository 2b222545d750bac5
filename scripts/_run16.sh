set -u
OUT=gpurun_out/r01_run16; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for b in 1 0 1 0; do timeout 300 python scripts/kernel_bench.py --only attention --bounded $b >> $OUT/kb.jsonl 2>> $OUT/kb.err; done
cut -c1-200 $OUT/kb.jsonl
