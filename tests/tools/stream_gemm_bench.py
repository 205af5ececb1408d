"""The step's streaming (short-K) NT products in the exact forms the step launches them, as achieved HBM TB/s of their algorithmic
bytes, next to a plain streaming pass over the same tensors on this box (the ceiling such a launch can reach).
usage: python tests/tools/stream_gemm_bench.py [tune ...]       (measurement tool; cold caches: 1 GiB flush between launches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import ops
from stil_tta_amd._lib import lib
from stil_tta_amd.ops import _p, _stream
L = lib()
tunes = [int(t) for t in sys.argv[1:]] or [0]
big = torch.empty(1 << 28, device="cuda")

def timeit(fn, reps=5):
    ts = []
    for _ in range(reps):
        big.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return min(ts)

for (M, N, K) in ((802816, 256, 64), (200704, 512, 128), (802816, 64, 256), (200704, 128, 512)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1; out = torch.empty(M, N, device="cuda")
    R = torch.randn(M, N, device="cuda"); Z = torch.randn(M, N, device="cuda"); Y = torch.randn(M, N, device="cuda")
    ab = torch.rand(3, N, device="cuda")
    stats = torch.stack([Y.mean(0), 1.0 / Y.std(0), 1.0 / Y.std(0), torch.zeros(N, device="cuda")]).contiguous()
    nt = (M + 63) // 64
    ts_ = torch.empty(2 * nt, N, device="cuda"); part = torch.empty(2 * nt, N, device="cuda")
    # reference: one streaming pass reading two [M,N] tensors and writing one (bn_apply with a residual)
    z = torch.empty(M, N, device="cuda")
    ms = timeit(lambda: L.bn_train_fwd_tiles(_p(Y), _p(ts_), 64, _p(ab[0]), _p(ab[1]), None, None, None, _p(R), None, _p(z), _p(stats), M, N, 1, 1e-5, 0.1,
                                             _p(torch.empty(L.bn_tiles_workspace_bytes(M, N, 64) + 64, dtype=torch.uint8, device="cuda")), L.bn_tiles_workspace_bytes(M, N, 64), _stream())) if False else None
    t_pass = timeit(lambda: torch.add(Y, R, out=z))
    print(f"[{M},{N},{K}] streaming pass (2 reads + 1 write of [M,N]): {t_pass*1e3:7.1f} us = {12.0*M*N/t_pass/1e9:5.2f} TB/s")
    forms = [("student fwd (colstats)", dict(colstats=ts_), 4.0 * (M * K + M * N)),
             ("teacher fwd (affine+resid+relu)", dict(sub=ab[2], scale=ab[0], shift=ab[1], resid=R, act=1), 4.0 * (M * K + 2 * M * N)),
             ("dgrad (resid+mask)", dict(resid=R, relu_mask=Z), 4.0 * (M * K + 3 * M * N)),
             ("dgrad (resid+mask+bstats)", dict(resid=R, relu_mask=Z, bstats=(Y, stats, part, 0, 0)), 4.0 * (M * K + 4 * M * N)),
             ("dgrad inner (bstats mode 2)", dict(bstats=(Y, stats, part, 2, 0)), 4.0 * (M * K + 2 * M * N))]
    for tune in tunes:
        ops.TUNE["gemm"] = tune
        for name, kw, nbytes in forms:
            if tune % 100 not in (0, 11) and "bstats" in kw:
                continue
            ms = timeit(lambda: ops.gemm_nt(A, W, M, N, K, out=out, **kw))
            print(f"   tune {tune:5d} {name:34s} {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TF  {nbytes/ms/1e9:5.2f} TB/s")
    ops.TUNE["gemm"] = 0
    del A, W, out, R, Z, Y, z, ts_, part
