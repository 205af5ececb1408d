import sys, os
sys.path.insert(0, "/root/repo")
import torch
from stil_tta_amd import ops
from stil_tta_amd._lib import lib
L = lib()
for (M, N, K, res) in ((802816, 256, 64, False), (802816, 256, 64, True), (200704, 512, 128, True), (802816, 64, 256, False)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    R = torch.randn(M, N, device="cuda") if res else None
    big = torch.empty(1 << 28, device="cuda")  # 1 GiB: flushes the caches between launches
    for v in (11, 21, 22):
        ops.TUNE["gemm"] = v
        ts = []
        for _ in range(5):
            big.zero_()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.gemm_nt(A, W, M, N, K, out=out, resid=R, act=1 if res else 0); e.record()
            torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        ms = min(ts)
        gb = 4.0 * (M * K + N * K + M * N * (2 if res else 1)) / 1e9
        print(f"({M},{N},{K}) resid={res} v{v}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TF  {gb/ms:5.2f} TB/s")
ops.TUNE["gemm"] = 0
