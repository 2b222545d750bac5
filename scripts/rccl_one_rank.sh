#!/bin/bash
# The torch.distributed / RCCL code path of bench.py on a ONE-GPU box (the 8-GPU tier is the driver's to run): a forced one-rank process
# group in the weak (uniform, equal-shapes all-gather) and the strong ragged mode (gather by shard plan), and the driver's torchrun form.
# Usage (GPU box, repo root): bash scripts/rccl_one_rank.sh gpurun_out/<tag>
set -u
OUT=${1:?outdir}; mkdir -p "$OUT"
export MASTER_ADDR=127.0.0.1
( export RAP_BENCH_FORCE_DIST=1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
  timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --light > "$OUT/bench_rccl_one_rank_weak.json" 2> "$OUT/weak.err"; echo "weak exit $?"
  MASTER_PORT=29512 timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --light --scaling strong --workload ragged --dtype bfloat16 \
    > "$OUT/bench_rccl_one_rank_strong_ragged.json" 2> "$OUT/strong.err"; echo "strong exit $?" )
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 1 --warmup 1 \
  --light --dtype bfloat16 > "$OUT/bench_torchrun_one_rank.json" 2> "$OUT/torchrun.err"; echo "torchrun exit $?"
for f in weak strong_ragged; do cut -c1-420 "$OUT/bench_rccl_one_rank_$f.json"; echo; done
cut -c1-300 "$OUT/bench_torchrun_one_rank.json"; echo
tail -2 "$OUT/strong.err" | cut -c1-300
