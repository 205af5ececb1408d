"""NT GEMM on the step's dominant shapes with the automatic configuration (one library build per process: STIL_LIB_PATH for A/B).
usage: python tests/tools/nt_bench.py [rounds] [tune]   (measurement tool)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import ops
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ops.TUNE["gemm"] = int(sys.argv[2]) if len(sys.argv) > 2 else 0
NT = [(50176, 256, 1024, 1, 14), (50176, 1024, 256, 1, 14), (12544, 512, 2048, 1, 7), (12544, 2048, 512, 1, 7), (16640, 512, 2048, 1, 0), (16640, 2048, 512, 1, 0),
      (200704, 128, 512, 1, 28), (50176, 256, 2304, 3, 14), (200704, 128, 1152, 3, 28), (802816, 64, 576, 3, 56), (12544, 512, 4608, 3, 7), (802816, 64, 256, 1, 56)]
tot = 0.0
for (M, N, K, k, H) in NT:
    if k == 1:
        A = torch.randn(M, K, device="cuda"); geom = None
    else:
        C = K // 9; A = torch.randn(M // (H * H), H, H, C, device="cuda"); geom = (H, H, C, H, H, 3, 3, 1, 1, 0)
    W = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    best = 1e9
    for rep in range(3):
        ops.gemm_nt(A, W, M, N, K, geom=geom, out=out); torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(rounds):
            ops.gemm_nt(A, W, M, N, K, geom=geom, out=out)
        e0.record(); torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(e0) / rounds)
    tot += best
    print(f"({M},{N},{K},k{k}) {best*1e3:8.1f} us {2.0*M*N*K/best/1e9:7.1f} TF")
print(f"sum {tot*1e3:.1f} us")
