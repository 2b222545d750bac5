#!/usr/bin/env python3
"""How many host threads should the CPU-oracle baseline use on this box?  Times one velocity-network forward of a
2-layer model on 1 pair (2 x 4096) for several thread counts."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rap_oracle as O
from rap_amd import synthetic as S
cfg = dict(S.RAP_12); cfg["num_layers"] = 2
sd = S.make_weights(cfg, 0)
inp = S.make_uniform_inputs(1, 2, 4096, seed=1234)
for n in (8, 16, 32, 64, 128, 256):
    if n > (os.cpu_count() or 1):
        break
    torch.set_num_threads(n)
    t0 = time.perf_counter()
    O.sample(sd, cfg, inp, 20, False, max_steps=1)
    print(json.dumps({"threads": n, "seconds_2layer_1step": time.perf_counter() - t0}), flush=True)
