"""CPU: the C-ABI library builds, loads, and exports exactly the symbols include/rapflow.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib_path():
    from rap_amd import _build
    return _build.build()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rapflow.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rap_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_loads(lib_path):
    lib = ctypes.CDLL(lib_path)
    assert lib.rap_version() >= 1


def test_every_declared_symbol_is_exported_and_bound(lib_path):
    from rap_amd import _lib
    lib = ctypes.CDLL(lib_path)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rapflow.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in rap_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == declared
    _lib.load()   # sets argtypes on every symbol; raises if one is missing


def test_header_is_plain_c_and_a_c_program_links_against_the_library(lib_path, tmp_path):
    """The boundary is a C ABI, not a Python one: include/rapflow.h compiles as pedantic C99 (and as C++), and a C program
    (tests/host/c_consumer.c) linked against librapflow.so gets the version, the workspace arithmetic and the argument checks --
    without torch, without Python, without a GPU."""
    import subprocess
    src = os.path.join(ROOT, "tests", "host", "c_consumer.c")
    inc = os.path.join(ROOT, "include")
    exe = str(tmp_path / "c_consumer")
    libdir = os.path.dirname(lib_path)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, src, "-L", libdir, "-lrapflow",
                           "-Wl,-rpath," + libdir, "-o", exe])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, "-x", "c++", src])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failure(s), ABI version 6" in r.stdout


def test_weight_count_matches_reference_parameter_count(lib_path):
    from rap_amd import _lib
    from rap_amd.synthetic import RAP_12, weight_spec
    lib = _lib.load()
    desc = _lib.ModelDesc(512, 12, 8, 32)
    n = lib.rap_weight_count(ctypes.byref(desc))
    import math
    assert n == sum(math.prod(s) for _, s in weight_spec(RAP_12))
    assert n == 85_577_728 - 0 or n > 85_000_000   # 85.58 M parameters
    bad = _lib.ModelDesc(500, 12, 8, 32)
    assert lib.rap_weight_count(ctypes.byref(bad)) < 0


def test_product_has_no_cpu_fallback():
    """The product path must fail loudly without a GPU: tensors on the CPU are rejected, never computed on."""
    import torch
    import rap_amd
    from rap_amd._lib import RapError
    from rap_amd.synthetic import make_inputs
    inp = make_inputs([[8, 8]], seed=1)
    with pytest.raises(RapError):
        rap_amd.fit_transformations(inp["pointclouds"], inp["pointclouds_gt"], inp["points_per_part"], inp["cu_seqlens"])
    m = rap_amd.PointCloudDiT(in_dim=0, out_dim=3, embed_dim=512, num_layers=1, num_heads=8, local_feat_dim=32)
    with pytest.raises(RapError):
        m.to("cpu")
    with pytest.raises(ValueError):
        rap_amd.get_sampler("rk4")          # sampler.py:168-169
    with pytest.raises(ValueError):
        rap_amd.fit_transformations(torch.zeros(4, 3), torch.zeros(4, 3), torch.tensor([[4]]), None)  # point_clouds.py:28-29


def test_product_does_not_import_oracle():
    """Nothing under rap_amd/ may import, call or link anything under oracle/."""
    pkg = os.path.join(ROOT, "rap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle/" not in text or f == "synthetic.py", f


def test_new_entry_points_validate_arguments_without_a_gpu():
    """Every caller-side entry point rejects NULL / non-positive arguments with RAP_ERR_INVALID (-1) before touching the device,
    and the workspace queries are pure host arithmetic."""
    import ctypes
    from rap_amd import _lib
    lib = _lib.load()
    N = ctypes.c_void_p(0)
    one = ctypes.c_void_p(1)      # non-NULL sentinel; the calls below must fail on another argument first
    assert lib.rap_rigidity_rmse(N, N, N, N, N, 1, 1, N, 0, N, N, 0, N) == -1
    assert lib.rap_trajectory_rigidity_rmse(N, N, N, 1, 1, 10, 2, N, N, N, N, 0, N) == -1
    assert lib.rap_select_generation(N, 1, 1, 1, 1, N, N, N, N, 0, N, N, N, N, N) == -1
    assert lib.rap_overlap_ratio(N, N, N, 1, 1, 10, N, 1, N, N, N, 0, N) == -1
    assert lib.rap_relative_transforms(N, N, N, N, N, N, 1, 1, N, N, N, N) == -1
    assert lib.rap_chamfer_rmse(N, N, N, 1, 10, N, N, 0, N) == -1
    assert lib.rap_correspondence_rmse(N, N, N, N, 1, 1, 0.1, N, N, 0, N) == -1
    assert lib.rap_spinnet_describe(N, N, N, 10, N, 1, 0.5, 0, N, 16, N, 0, N) == -1
    assert lib.rap_spinnet_create(N, 0, N, ctypes.byref(ctypes.c_void_p(0))) == -1
    assert lib.rap_model_set_compute_dtype(N, 1, N) == -1
    assert lib.rap_spinnet_weight_count() == 426341
    assert lib.rap_rigidity_workspace_bytes(64, 20, 32) > 0 and lib.rap_rigidity_workspace_bytes(-1, 0, 0) == 0
    assert lib.rap_overlap_workspace_bytes(262144, 32, 2) > 262144 * 8
    assert lib.rap_nn_metrics_workspace_bytes(262144, 32) > 262144 * 12
    assert lib.rap_spinnet_workspace_bytes(2048) > 2048 * 140 * 1152 * 4 and lib.rap_spinnet_workspace_bytes(0) == 0
    assert lib.rap_voxel_bounds(N, 10, 0.1, N, N, N) == -1 and lib.rap_voxel_downsample(N, 10, 0.1, N, 1.0, N, N, N, 0, N) == -1
    b = (ctypes.c_int64 * 6)(-3, -2, 0, 6, 4, 1)                   # extents 9, 6, 1 -> v = 9 -> 9 + 81 + 729 + 1 slots
    assert lib.rap_voxel_table_slots(b) == 820 and lib.rap_voxel_workspace_bytes(b) >= 820 * 8
    huge = (ctypes.c_int64 * 6)(0, 0, 0, 10 ** 7, 10, 10)
    assert lib.rap_voxel_table_slots(huge) == -1 and lib.rap_voxel_workspace_bytes(huge) == 0
    del one
    # the shipped library carries no kernel-variant switch: keys 0-4 / 8 of the round-1/2 experiments are refused, the five
    # production switches accept {0, 1} only
    for key in (0, 1, 2, 3, 4, 8, 10, 14, -1):
        assert lib.rap_set_tuning(key, 1) == -1 and lib.rap_set_tuning(key, 0) == -1, key
    for key in (5, 6, 7, 9, 11, 12, 13, 15):
        assert lib.rap_set_tuning(key, 2) == -1 and lib.rap_set_tuning(key, 1) == 0, key
    assert lib.rap_model_bounded_attention_launches(N) < 0
    # round 6: the latent-feature entry points, the transform-error metric, the few-token switches
    desc = _lib.ModelDesc(512, 12, 8, 32)
    n0 = lib.rap_weight_count(ctypes.byref(desc))
    assert lib.rap_weight_count_latent(ctypes.byref(desc), 0) == n0
    assert lib.rap_weight_count_latent(ctypes.byref(desc), 64) == n0 + 512 * 64        # 64 more columns of emb_proj.weight
    assert lib.rap_weight_count_latent(ctypes.byref(desc), 6) == -1 and lib.rap_weight_count_latent(ctypes.byref(desc), 516) == -1
    assert lib.rap_model_create_latent(ctypes.byref(desc), 64, N, n0 + 512 * 64, N, ctypes.byref(ctypes.c_void_p(0))) == -1      # NULL weights
    assert lib.rap_dit_forward_latent(N, N, N, N, N, N, N, N, N, N, 1, 1, 10, N, N, N, 0, N) == -1
    assert lib.rap_sample_latent(N, N, N, N, N, N, N, N, N, 1, 1, 10, 2, 1, N, N, N, N, N, N, 0, N) == -1
    assert lib.rap_transform_errors(N, N, N, N, N, N, N, N, 1, 1, N, N, N, N, N) == -1
    assert lib.rap_attention_workspace_bytes(262144, 64) >= (262144 // 256 + 65) * 16 + 65 * 4      # work items + the sanitised cu_seqlens copy
    assert lib.rap_set_tuning(18, -1) == -1 and lib.rap_set_tuning(18, 256) == 0
    assert lib.rap_set_tuning(19, 2) == -1 and lib.rap_set_tuning(19, 1) == 0
    assert lib.rap_set_tuning(20, 32) == -1 and lib.rap_set_tuning(20, 1) == 0
    for v in (2, 64, 128, 66, 130):                          # key 20's other values: the form without key groups, the forced A/B forms
        assert lib.rap_set_tuning(20, v) == 0
    assert lib.rap_set_tuning(20, 1) == 0
    assert lib.rap_set_tuning(21, 1) == -1                     # (no such key)


def test_split_precision_entry_points_refuse_bad_shapes_without_a_gpu():
    """rap_x2_gemm / rap_x2_pack / rap_x2_attention validate on the host, before any launch: strides that would break the 16-byte row
    pieces of the epilogues, shapes outside the tile rules, and missing planes come back as RAP_ERR_INVALID (-1) / workspace (-2)."""
    import ctypes
    from rap_amd import _lib
    lib = _lib.load()
    N, one = ctypes.c_void_p(0), ctypes.c_void_p(256)      # a non-NULL sentinel: every call below fails before the pointer is used
    def gemm(epi, lda=1024, ldw=1024, ldc=512, M=256, Nn=512, K=1024, ldr=512, resid=one, heads=0, vt=N, nblk=0, A=one):
        return lib.rap_x2_gemm(epi, A, lda, one, ldw, one, ldc, M, Nn, K, one, resid, ldr, 1.0, heads, N, N, 8.0, vt, nblk, N)
    assert gemm(1, A=N) == -1                               # NULL operand
    assert gemm(1, ldc=514) == -1 and gemm(1, ldr=513) == -1 and gemm(1, ldc=256) == -1      # fp32 rows leave as float4 pieces; ldc >= N
    assert gemm(1, lda=1000) == -1 and gemm(1, lda=512) == -1                                  # 16-byte operand rows; lda >= K_physical
    assert gemm(1, K=96) == -1 and gemm(1, K=64) == -1 and gemm(1, Nn=384) == -1              # K_physical % 64, >= 128; N % 256
    assert gemm(3, ldc=4100) == -1                                                             # paired fp16 GEGLU rows: 8-element pieces
    assert gemm(5, Nn=1536, heads=8, vt=N) == -1 and gemm(5, Nn=1536, heads=8, vt=one, nblk=3) == -1   # V^T image missing / too small
    assert gemm(5, Nn=1024, heads=8, vt=one, nblk=4) == -1                                     # N != 3 * heads * 64
    assert gemm(0) == -1 and gemm(7) == -1                                                     # epilogues the split path does not have
    assert lib.rap_x2_pack(N, 64, 4, 64, 1.0, one, N) == -1 and lib.rap_x2_pack(one, 32, 4, 64, 1.0, one, N) == -1
    assert lib.rap_x2_unpack(one, -1, 64, 1.0, one, N) == -1
    assert lib.rap_x2_attention(N, one, 4, one, 1, one, 256, 8, one, 1 << 20, N) == -1
    assert lib.rap_x2_attention(one, one, 4, one, 1, one, 256, 8, one, 8, N) == -2             # workspace smaller than the work list
    # the 16-bit entry point shares the stride rules
    assert lib.rap_gemm_h16(1, 1, one, 512, one, 512, one, 514, 256, 512, 512, one, one, 512, 0, N, 0, N) == -1


def test_splitk_rule_of_the_16bit_residual_gemm_is_host_arithmetic():
    """rap_gemm_h16_splitk_workspace_bytes states the few-token split-K rule of the 16-bit residual GEMMs (gemm_h16.hip:
    gemm_h16_splits): K >= 1024 with K / 64 a multiple of 4, at most 64 tiles of 128 x 128 -> 4 partial planes, at most 128 -> 2,
    otherwise none.  Round 4: the size is a function of the SHAPE alone -- tuning key 6 gates the launch, not the
    reservation, so a workspace sized before the key is toggled is never too small.  Pure host arithmetic -- and the entry point refuses a short workspace, other
    epilogues and NULL operands before it touches the device."""
    import ctypes
    from rap_amd import _lib
    lib = _lib.load()
    q = lib.rap_gemm_h16_splitk_workspace_bytes
    plane = lambda m, n: m * n * 4
    assert q(2048, 512, 2048) == 4 * plane(2048, 512)          # ff2 of one pair of 2 x 1024 points: 16 x 4 = 64 tiles
    assert q(2049, 512, 2048) == 2 * plane(2049, 512)          # 17 x 4 = 68 tiles
    assert q(4096, 512, 2048) == 2 * plane(4096, 512)          # 128 tiles
    assert q(4097, 512, 2048) == 0                             # 132 tiles: the grid covers more than half of the CUs
    assert q(2048, 512, 512) == 0 and q(2048, 512, 960) == 0   # short K (the out-projection); K / 64 not a multiple of 4 is also out
    assert q(100, 1024, 4096) == 4 * plane(100, 1024)          # d = 1024 models
    assert q(100, 576, 2048) == 0                              # N not a multiple of 128
    assert q(0, 512, 2048) == 0 and q(-5, 512, 2048) == 0
    try:
        assert lib.rap_set_tuning(6, 0) == 0
        assert q(2048, 512, 2048) == 4 * plane(2048, 512)      # still reserved: the key only gates the launch
    finally:
        assert lib.rap_set_tuning(6, 1) == 0
    N, one = ctypes.c_void_p(0), ctypes.c_void_p(256)
    assert lib.rap_gemm_h16_splitk(1, 7, N, 2048, one, 2048, one, 512, 2048, 512, 2048, N, one, 512, one, 1 << 30, N) == -1      # NULL A
    assert lib.rap_gemm_h16_splitk(1, 3, one, 2048, one, 2048, one, 512, 2048, 512, 2048, N, one, 512, one, 1 << 30, N) == -1    # not a residual epilogue
    assert lib.rap_gemm_h16_splitk(1, 7, one, 2048, one, 2048, one, 512, 2048, 512, 2048, N, one, 512, one, 4 * plane(2048, 512) - 1, N) == -2   # short workspace
    assert lib.rap_gemm_h16_splitk(1, 7, one, 2048, one, 2048, one, 512, 2048, 512, 2048, N, one, 512, N, 0, N) == -2             # no workspace
    # epilogue 6 (two different meanings in rounds 2 and 3) is retired with ABI version 4: refused, never reinterpreted
    assert lib.rap_gemm_h16_splitk(1, 6, one, 2048, one, 2048, one, 512, 2048, 512, 2048, N, one, 512, one, 1 << 30, N) == -1
    assert lib.rap_gemm_h16(1, 6, one, 512, one, 512, one, 512, 256, 512, 512, N, one, 512, 0, N, 0, N) == -1
    assert lib.rap_version() == _lib.ABI_VERSION == 6
    assert lib.rap_poison_on_flag(N, one, 4, N) == -1 and lib.rap_poison_on_flag(one, N, 4, N) == -1
