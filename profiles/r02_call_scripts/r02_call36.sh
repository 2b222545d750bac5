#!/bin/bash
# r02 GPU call 36: per-block time stamps inside the bf16 attention kernel (ablation build made on the GPU box)
set -u
OUT=gpurun_out/r02_c36; mkdir -p $OUT
export TMPDIR=/tmp
cd rap_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DRAP_ABLATION_BUILD"
( /opt/rocm/bin/hipcc $F -c attn_h16.hip -o attn_h16.o & /opt/rocm/bin/hipcc $F -c api.hip -o api.o & wait )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o librapflow.so
cd ../..
timeout 300 python scripts/attn_ts.py > $OUT/attn_ts.jsonl 2> $OUT/err.log
cat $OUT/attn_ts.jsonl; tail -5 $OUT/err.log
echo "r02 call 36 done"
