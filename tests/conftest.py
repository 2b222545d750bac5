import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a -m gpu test that takes minutes (still part of -m gpu; deselect with -m 'gpu and not slow')")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    import numpy as np
    import torch
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    g = {k: z[k] for k in z.files}
    inputs = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("in_")}
    return g, inputs


GOLDEN_CASES = ["l2_ragged_rigid", "l2_ragged_free", "l2_emptypart_rigid", "l12_small_rigid", "l12_pair512_free"]
# the reference's other two model sizes (rap_10, rap_16; config/model/flow_model/point_cloud_dit_{10,16}.yaml), fixtures from the unmodified
# reference like the ones above (oracle/make_golden.py --case=...)
MODEL_SIZE_CASES = ["l16_small_rigid", "l10_small_free"]
# the constructor switches of PointCloudDiT every shipped config leaves True (point_cloud_dit.py:28,33-34), one fixture each from the
# unmodified reference (round 5; oracle/make_golden.py CASE_SWITCHES): name -> keyword overrides
SWITCH_CASES = {"l2_noqknorm_rigid": {"qk_norm": False}, "l2_noscale_free": {"scale_emb_on": False},
                "l2_nofeat_rigid": {"local_feat_concat_on": False}}
# round 6: a model with in_dim = 64 (latent point features concatenated into the embedding, embedding.py:163-166); the fixture carries
# `in_latent_features` (TP, 64) next to the usual inputs
LATENT_CASES = {"l2_latent64_rigid": {"in_dim": 64}}
