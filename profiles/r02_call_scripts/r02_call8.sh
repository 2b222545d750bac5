#!/bin/bash
# r02 GPU call 8: SQ counters of the phase-split kernel (13) and the default (1) on the four GEMM shapes
set -u
OUT=gpurun_out/r02_c8; mkdir -p $OUT
export TMPDIR=/tmp
for v in 13 1; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
    tag=$(echo $set | cut -d' ' -f1)
    ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace -d "$GRAFT_REPO_ROOT/$OUT/pmc_v${v}_$tag" -o pmc -- \
        python "$GRAFT_REPO_ROOT/scripts/kernel_bench.py" --dtype bfloat16 --only gemm --h16-gemm-variant $v > "$GRAFT_REPO_ROOT/$OUT/pmc_v${v}_$tag.log" 2>&1 )
    DB=$(find "$OUT/pmc_v${v}_$tag" -name '*.db' | head -1)
    if [ -n "$DB" ]; then python scripts/rocpd_summary.py "$DB" --pmc | grep -E "gemm_h16" | sed "s/^/v$v /" >> "$OUT/pmc_sq.txt"; else tail -5 "$OUT/pmc_v${v}_$tag.log"; fi
    find "$OUT/pmc_v${v}_$tag" -name '*.db' -delete
  done
done
cat $OUT/pmc_sq.txt
