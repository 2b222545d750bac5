set -u
OUT=gpurun_out/r01_run35; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $R/$OUT/p -o pmc -- python $R/scripts/kernel_bench.py --dtype bfloat16 --only gemm --pmc > $R/$OUT/p.log 2>&1
DB=$(find $R/$OUT/p -name '*.db' | head -1); [ -n "$DB" ] && python $R/scripts/rocpd_summary.py $DB --pmc | grep -E "^PMC.*gemm_h16" | cut -c1-190
find $R/$OUT/p -name '*.db' -delete
tail -3 $R/$OUT/p.log
