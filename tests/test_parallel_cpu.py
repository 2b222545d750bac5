"""CPU, world_size 2, gloo: the N > 1 path of bench.py -- contiguous pair sharding and the single all-gather of
registered clouds + poses -- exercised with two real processes (127.0.0.1 rendezvous)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from rap_amd.parallel import gather_registrations, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 32, 256, 257):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in shard_range(n, w, r)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, w, r)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import rap_oracle as O     # stands in for the per-rank GPU sampler in this CPU test
        from rap_amd import synthetic as S
        cfg = dict(S.RAP_12); cfg["num_layers"] = 1
        sd = S.make_weights(cfg, 0)
        n_pairs, views, pts = 4, 2, 48
        mine = shard_range(n_pairs, world, rank)
        inp = S.make_inputs([[pts] * views for _ in mine], seed=1234 + mine.start)
        res = O.sample(sd, cfg, inp, 2, True)
        final, R, t = gather_registrations(res["end_point_trajectory"][-1], res["R"], res["t"])
        torch.save({"final": final, "R": R, "t": t}, os.path.join(tmpdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_equals_single_process_batch(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from oracle import rap_oracle as O
    from rap_amd import synthetic as S
    cfg = dict(S.RAP_12); cfg["num_layers"] = 1
    sd = S.make_weights(cfg, 0)
    inp = S.make_inputs([[48, 48] for _ in range(4)], seed=1234)
    ref = O.sample(sd, cfg, inp, 2, True)
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"))
        # samples are independent: the sharded + gathered job equals the one-process 4-pair batch
        assert (got["final"] - ref["end_point_trajectory"][-1]).abs().max().item() < 1e-5
        assert (got["R"] - ref["R"]).abs().max().item() < 1e-5
        assert (got["t"] - ref["t"]).abs().max().item() < 1e-5


def _worker_uneven(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 3 pairs over 2 ranks (2 + 1) with ragged point counts: per-rank TP and B both differ  (the equal-shape
        # all_gather_into_tensor would hang or corrupt here)
        mine = shard_range(3, world, rank)
        g = torch.Generator().manual_seed(100 + rank)
        sizes = [[40, 25], [33, 60], [17, 52]]
        tp = sum(sum(sizes[i]) for i in mine)
        final = torch.randn(tp, 3, generator=g)
        R = torch.randn(len(mine), 2, 3, 3, generator=g); t = torch.randn(len(mine), 2, 3, generator=g)
        gp, gR, gt = gather_registrations(final, R, t)
        torch.save({"final": final, "R": R, "t": t, "gp": gp, "gR": gR, "gt": gt}, os.path.join(tmpdir, f"u{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_with_uneven_shards(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_uneven, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(os.path.join(str(tmp_path), f"u{r}.pt")) for r in range(2)]
    for key, gkey in (("final", "gp"), ("R", "gR"), ("t", "gt")):
        want = torch.cat([got[0][key], got[1][key]])
        for r in range(2):
            assert torch.equal(got[r][gkey], want), (key, r)


def _run_bench(extra_args, env_extra, timeout=600):
    import subprocess
    env = dict(os.environ); env.update(env_extra)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_launches_its_own_ranks_when_asked_for_more_than_one_gpu():
    """`python bench.py --gpus 2` WITHOUT torchrun: the file re-launches itself under torch.distributed.run with
    2 ranks on 127.0.0.1, every rank asserts WORLD_SIZE == --gpus, rank 0 prints ONE JSON line with n_gpus = 2.  Launcher self-test
    mode: gloo on CPU ranks with a stub in place of the sampler (rap_amd has no CPU path) -- launch, rendezvous, sharding, the
    all-gather, per-rank timing and the JSON are the code under test."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3", "--points", "16"],
                   {"RAP_BENCH_LAUNCHER_SELFTEST": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["pairs_total"] == 6 and j["gather_ok"] is True and j["stub"] is True
    assert len(j["per_rank"]["elapsed_s"]) == 2


def test_bench_refuses_a_rank_count_it_was_not_asked_for():
    """--gpus N must equal the number of ranks that actually run: a torchrun world of 1 with --gpus 2 (or the reverse) exits non-zero
    before anything is timed, and `--gpus 2` on a box with fewer than 2 GPUs refuses instead of timing one (here: no GPU at all)."""
    import subprocess
    env = dict(os.environ); env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        r = _run_bench(["--gpus", "2"], {})
        assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()
