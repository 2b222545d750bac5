#!/bin/bash
# r02 GPU call 32: where does the per-block fixed cost of the bf16 attention go?  (ablation build made on the GPU box)
set -u
OUT=gpurun_out/r02_c32; mkdir -p $OUT
export TMPDIR=/tmp
cd rap_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DRAP_ABLATION_BUILD"
( /opt/rocm/bin/hipcc $F -c attn_h16.hip -o attn_h16.o & /opt/rocm/bin/hipcc $F -c api.hip -o api.o & wait )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o librapflow.so
cd ../..
for V in 0 16 17 18; do
for PB in "1024 128" "4096 32"; do
  set -- $PB
  timeout 200 python scripts/kernel_bench.py --dtype bfloat16 --only attention --h16-attn-variant $V --points $1 --batch $2 --views 2 2>> $OUT/kb.err | sed "s/^/{\"variant\": $V, \"points\": $1, \"row\": /; s/$/}/" >> $OUT/kb.jsonl
done
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c32/kb.jsonl"):
    try: j = json.loads(l)
    except Exception as e: print("bad", l[:80]); continue
    r = j["row"]; print(j["variant"], j["points"], r.get("kernel", "")[:44], round(r.get("ms"), 3), round(r.get("tflops"), 1))
PY
echo "r02 call 32 done"
