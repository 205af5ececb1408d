"""NT GEMM time vs number of resident workgroups per CU (fixed N, K; M chosen so the grid is 1, 2, 4, 8, 12, 16 tiles per CU):
separates per-workgroup pipeline latency from MFMA saturation.  usage: python tests/tools/occupancy_curve.py [variant] [bk]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import ops
from stil_tta_amd._lib import lib

L = lib()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 11
ops.TUNE["gemm"] = variant
if len(sys.argv) > 2:
    ops.TUNE["gemm"] += 100 * (int(sys.argv[2]) == 32)
BM = 64 if variant == 11 else 128
BN = 128 if variant == 22 else 64
for (N, K) in ((256, 2304), (256, 256), (1024, 512)):
    print(f"variant {variant} N={N} K={K}")
    for per_cu in (1, 2, 3, 4, 6, 8, 12, 16, 24):
        tiles = 256 * per_cu
        M = tiles // (N // BN) * BM
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
        for _ in range(2):
            ops.gemm_nt(A, W, M, N, K, out=out)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        R = 10
        for _ in range(R):
            ops.gemm_nt(A, W, M, N, K, out=out)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / R
        k_iters = K // 16
        print(f"  {per_cu:3d} tiles/CU  M={M:7d}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF   {ms*1e6*2.3/k_iters/per_cu:8.0f} cycles per (tile,k16) per CU-slot @2.3GHz")
