#!/bin/bash
# r02 GPU call 30: timing-only ablations of the pipelined bf16 attention (an ablation build made ON the GPU box, never shipped):
# variant 13 vs 14 (no K/V stream) vs 15 (no stream, no barriers), plus the r01 kernel's ablations for reference
set -u
OUT=gpurun_out/r02_c30; mkdir -p $OUT
export TMPDIR=/tmp
cd rap_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DRAP_ABLATION_BUILD"
( /opt/rocm/bin/hipcc $F -c attn_h16.hip -o attn_h16.o & /opt/rocm/bin/hipcc $F -c api.hip -o api.o & wait )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o librapflow.so
cd ../..
for V in 0 13 14 15 2; do
  timeout 200 python scripts/kernel_bench.py --dtype bfloat16 --only attention --h16-attn-variant $V > $OUT/kb_v$V.jsonl 2>> $OUT/kb.err
done
python - <<'PY'
import json
for v in (0, 13, 14, 15, 2):
    for l in open(f"gpurun_out/r02_c30/kb_v{v}.jsonl"):
        try: j = json.loads(l)
        except Exception: continue
        print(v, j.get("kernel", "")[:50], j.get("ms"), j.get("tflops"))
PY
tail -3 $OUT/kb.err
echo "r02 call 30 done"
