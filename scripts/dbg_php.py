import sys, os, torch
sys.path.insert(0, os.getcwd())
from rap_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
st = lambda: _lib.current_stream(dev)
def run(dtc, epi, M, N, K, pz, seed=101):
    g = torch.Generator(device=dev).manual_seed(seed)
    tdt = {1: torch.bfloat16, 2: torch.float16}[dtc]
    A = torch.randn(M, K, device=dev, generator=g).to(tdt); W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(tdt)
    bias = torch.randn(N, device=dev, generator=g)
    g2 = torch.Generator(device=dev).manual_seed(5)
    C = torch.randn(M, N, device=dev, generator=g2)
    if epi == 6: C = C.to(torch.float16)
    assert lib.rap_set_tuning(11, pz) == 0
    rc = lib.rap_gemm_h16(dtc, epi, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(C), N, M, N, K, _lib.ptr(bias), _lib.ptr(C), N, 0, _lib.ptr(None), 0, st())
    assert rc == 0
    torch.cuda.synchronize()
    ref = None
    return C
for dtc in (1, 2):
  for (epi, M, N, K) in ((1, 65536, 512, 2048), (1, 65536, 512, 512), (1, 131072, 512, 2048), (6, 65536, 512, 2048)):
    outs = [run(dtc, epi, M, N, K, pz) for pz in (1, 0, 1)]
    d = (outs[0].float() - outs[1].float()).abs()
    d2 = (outs[0].float() - outs[2].float()).abs()
    bad = (d > 0).nonzero()
    print("dt", dtc, "epi", epi, M, N, K, "mismatch elems", bad.shape[0], "max", d.max().item(), "persist-vs-persist mismatches", int((d2 > 0).sum()))
    if bad.shape[0]:
        rows = bad[:, 0]; cols = bad[:, 1]
        tiles = torch.unique(rows // 256 * (N // 256) + cols // 256)
        print("   tiles affected", tiles.numel(), tiles[:20].tolist(), "rows mod 256 range", int((rows % 256).min()), int((rows % 256).max()), "cols mod 256", int((cols % 256).min()), int((cols % 256).max()))
        print("   rows mod 32 hist", torch.bincount(rows % 32, minlength=32).tolist())
        print("   rows//32 mod 8 hist", torch.bincount((rows // 32) % 8, minlength=8).tolist(), " cols//64 mod 4 hist", torch.bincount((cols // 64) % 4, minlength=4).tolist())
lib.rap_set_tuning(11, 1)
