"""Two ranks, ONE MI355X: the N > 1 path of bench.py with the REAL rap_amd sampler (not the CPU oracle) -- shard_range, one
sampling call per rank on its shard, gather_registrations -- must reproduce the single-process batch.

The test box has one GPU, so both ranks use cuda:0.  RCCL refuses two ranks on one device ("Duplicate GPU detected"); the
workers then fall back to gloo and stage the gather through host memory -- the sharding / packing / un-padding logic under test
is the same.  Which backend carried the collective is recorded in the result files."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

PAIRS = [[700, 650], [512, 512], [300, 901]]          # ragged: 3 pairs over 2 ranks = 2 + 1, different TP per rank


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port_nccl, port_gloo, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import rap_amd
    from rap_amd import synthetic as S
    from rap_amd.parallel import gather_registrations, shard_range
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    backend = "nccl"
    try:
        os.environ["MASTER_PORT"] = str(port_nccl)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        assert probe.item() == world
    except Exception:                                   # RCCL: duplicate GPU -> gloo, gather staged through the host
        backend = "gloo"
        if dist.is_initialized():
            dist.destroy_process_group()
        os.environ["MASTER_PORT"] = str(port_gloo)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = dict(S.RAP_12); cfg["num_layers"] = 2
        sd = S.make_weights(cfg, 0)
        model = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=2, num_heads=8, local_feat_dim=32)
        model.load_state_dict(sd); model.to(dev)
        flow = rap_amd.RectifiedPointFlow(flow_model=model, inference_sampling_steps=3, rigidity_forcing=True)
        mine = shard_range(len(PAIRS), world, rank)
        inp = S.make_inputs([PAIRS[i] for i in mine], seed=1234 + mine.start)
        d = {k: v.to(dev) for k, v in inp.items()}
        out = flow.sample_and_register(d, x_1=d["x_1"])
        final, R, t = out["end_point_trajectory"][-1], out["R"], out["t"]
        if backend == "gloo":
            final, R, t = final.cpu(), R.cpu(), t.cpu()
        gp, gR, gt = gather_registrations(final, R, t)
        torch.save({"backend": backend, "final": gp.cpu(), "R": gR.cpu(), "t": gt.cpu()}, os.path.join(tmpdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_single_process_batch(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), _free_port(), str(tmp_path)), nprocs=2, join=True)
    import rap_amd
    from rap_amd import synthetic as S
    dev = torch.device("cuda", 0)
    cfg = dict(S.RAP_12); cfg["num_layers"] = 2
    model = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=2, num_heads=8, local_feat_dim=32)
    model.load_state_dict(S.make_weights(cfg, 0)); model.to(dev)
    flow = rap_amd.RectifiedPointFlow(flow_model=model, inference_sampling_steps=3, rigidity_forcing=True)
    inp = S.make_inputs(PAIRS, seed=1234)
    d = {k: v.to(dev) for k, v in inp.items()}
    ref = flow.sample_and_register(d, x_1=d["x_1"])
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"))
        print("rank", r, "collective backend:", got["backend"])
        # samples are independent; few-token calls may take the split-KV / split-K paths (another fp32 summation order), so
        # the comparison is to rounding, not bitwise
        assert got["final"].shape == ref["end_point_trajectory"][-1].shape
        assert (got["final"] - ref["end_point_trajectory"][-1].cpu()).abs().max().item() < 2e-5
        assert (got["R"] - ref["R"].cpu()).abs().max().item() < 2e-5 and (got["t"] - ref["t"].cpu()).abs().max().item() < 2e-5


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` with N above the number of visible GPUs must exit non-zero and print NO result line (VERDICT r02:
    it used to time one rank and print n_gpus: 1) -- both through its own launcher and when a torchrun world of 1 is given --gpus 2."""
    import subprocess
    n = torch.cuda.device_count()
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip(), (r.returncode, r.stdout, r.stderr[-500:])
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29998")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
