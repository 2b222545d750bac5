#!/usr/bin/env python3
"""MiniSpinNet descriptor throughput: implicit-GEMM convolutions (r02 default) vs the r01 im2col + GEMM path, same process, interleaved."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rap_amd import _lib
from rap_amd.spinnet import MiniSpinNet, make_spinnet_weights
dev = torch.device("cuda:0")
lib = _lib.load()
net = MiniSpinNet(des_r=0.2); net.load_state_dict(make_spinnet_weights(0)); net.to(dev)
g = torch.Generator(device=dev).manual_seed(0)
cloud = torch.rand(65536, 3, device=dev, generator=g) * torch.tensor([4.0, 4.0, 0.3], device=dev)
kp = cloud[:4096].clone()
perm = np.random.RandomState(0).permutation(65536)
def run(): return net(cloud[None], kp[None], 0.2, True, perm=perm)["desc"]
outs = {}
for rep in range(3):
    for mode in (1, 0):
        net.im2col_path = not mode                     # per-call flag RAP_SPINNET_IM2COL_PATH (rapflow.h)
        run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): d = run()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        outs[mode] = d
        print(json.dumps({"conv_path": "implicit GEMM" if mode else "im2col + GEMM (r01)", "keypoints": 4096, "ms": round(ms, 3), "keypoints_per_s": round(4096 / ms * 1e3),
                          "algorithmic_TFLOPs": round(4096 * 119e6 / (ms * 1e-3) / 1e12, 1), "frac_of_157.3TF": round(4096 * 119e6 / (ms * 1e-3) / 157.3e12, 3)}), flush=True)
net.im2col_path = False
print(json.dumps({"max_abs_descriptor_difference_between_paths": float((outs[1] - outs[0]).abs().max())}))
