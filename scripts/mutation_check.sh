#!/bin/bash
# Does the test-suite notice a precision regression of the bf16 path?    On the GPU box: rebuild gemm_h16.hip with a
# deliberately injected extra bf16 rounding in the residual epilogues (-DRAP_MUTATION=1: of the new residual-stream value; =2: of the GEMM
# output before the residual add), relink librapflow.so, run the bf16 deviation tests, expect FAILURES, restore the shipped library.
# Usage (repo root, GPU box): bash scripts/mutation_check.sh <outdir>
set -u
OUT=$1; mkdir -p "$OUT"
CS=rap_amd/csrc
cp $CS/librapflow.so /tmp/librapflow.orig.so
OBJS=$(ls $CS/*.o | grep -v "\.abl\.o" | grep -v "gemm_h16.o" | grep -v "\.mut")
for MUT in 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRAP_MUTATION=$MUT -c $CS/gemm_h16.hip -o /tmp/gemm_h16.mut$MUT.o || { echo "mutant $MUT: build failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm_h16.mut$MUT.o -o $CS/librapflow.so
  # kernel level: the residual epilogues are pinned to ONE rounding from an fp64 evaluation on the same operands
  timeout 300 python -m pytest tests/test_h16_gpu.py -q -k "fp32_out_matches or fp16_residual_epilogue" > "$OUT/pytest_kernel_mutant$MUT.log" 2>&1
  echo "mutant $MUT, kernel-level epilogue tests: pytest exit $? -- $(grep -E "passed|failed" "$OUT/pytest_kernel_mutant$MUT.log" | tail -1)" | tee -a "$OUT/mutation_summary.txt"
  if [ "${MUTATION_KERNEL_ONLY:-0}" = "1" ]; then continue; fi
  timeout 600 python -m pytest tests/test_headline_gpu.py tests/test_fullconfig_gpu.py -q -k "(16bit_all_steps and bfloat16) or c1_all_32" > "$OUT/pytest_mutant$MUT.log" 2>&1
  echo "mutant $MUT (extra bf16 rounding of the $( [ $MUT = 1 ] && echo 'residual-stream value' || echo 'GEMM output' )): pytest exit $? -- $(grep -E "passed|failed" "$OUT/pytest_mutant$MUT.log" | tail -1)" | tee -a "$OUT/mutation_summary.txt"
  grep -E "^FAILED" "$OUT/pytest_mutant$MUT.log" | head -12 >> "$OUT/mutation_summary.txt"
  python - "$MUT" >> "$OUT/mutation_summary.txt" <<'PY'
import json, sys
rows = [json.loads(l) for l in open("gpurun_out/headline_parity.jsonl")][-10:]
for r in rows:
    if r.get("dtype") == "bfloat16":
        print(f"  mutant {sys.argv[1]}: {r['case']} [{r.get('residual_stream')}]: final cloud {r['final_end_point']:.2e}  |dR|_F {r['R_frob']:.2e}  t {r['t']:.2e}")
PY
done
cp /tmp/librapflow.orig.so $CS/librapflow.so
if [ "${MUTATION_KERNEL_ONLY:-0}" = "1" ]; then exit 0; fi
timeout 300 python -m pytest tests/test_headline_gpu.py -q -k "16bit_all_steps and bfloat16 and c1_rigid" > "$OUT/pytest_restored.log" 2>&1
echo "shipped library restored: pytest exit $? -- $(grep -E "passed|failed" "$OUT/pytest_restored.log" | tail -1)" | tee -a "$OUT/mutation_summary.txt"
