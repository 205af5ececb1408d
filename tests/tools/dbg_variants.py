"""Diagnostic (GPU box): the NT GEMM with parts of its main loop compiled out (tests/tools/dbg_libs/lib_<VARIANT>.so: apply
tests/tools/dbg_variants.patch to a scratch copy of stil_tta_amd/csrc and build stil_hip.hip with -DDBG_NOGLOBAL / -DDBG_NOSTORE /
-DDBG_NOLDSREAD / -DDBG_NOBARRIER / -DDBG_SAMEADDR / -DDBG_LINE; the results are numerically meaningless) -- which part of the loop costs what.
usage: python tests/tools/dbg_variants.py <variant>   (one process per variant: the library is loaded once)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import stil_tta_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "tests", "tools", "dbg_libs", f"lib_{sys.argv[1]}.so")
import torch
from stil_tta_amd import ops

out = []
for (M, N, K, k, H) in ((98304, 256, 2304, 1, 0), (50176, 256, 2304, 3, 14), (50176, 1024, 256, 1, 0), (802816, 256, 64, 1, 0), (200704, 512, 128, 1, 0)):
    if k == 1:
        A = torch.randn(M, K, device="cuda"); geom = None
    else:
        C = K // 9; A = torch.randn(M // (H * H), H, H, C, device="cuda"); geom = (H, H, C, H, H, 3, 3, 1, 1, 0)
    W = torch.randn(N, K, device="cuda"); o = torch.empty(M, N, device="cuda")
    for _ in range(3):
        ops.gemm_nt(A, W, M, N, K, geom=geom, out=o)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(10):
        ops.gemm_nt(A, W, M, N, K, geom=geom, out=o)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    out.append(f"{2.0 * M * N * K / ms / 1e9:6.1f}")
print(f"{sys.argv[1]:44s} " + "  ".join(out), flush=True)
