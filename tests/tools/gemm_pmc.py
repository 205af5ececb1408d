"""One launch per dominant NT shape (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import ops
dev = "cuda"
for (M, N, K, k, H) in [(50176, 256, 2304, 3, 14), (802816, 256, 64, 1, 56), (16640, 2048, 512, 1, 0), (802816, 64, 576, 3, 56), (16640, 512, 2048, 1, 0)]:
    if k == 1:
        A = torch.randn(M, K, device=dev); geom = None
    else:
        C = K // 9; A = torch.randn(M // (H * H), H, H, C, device=dev); geom = (H, H, C, H, H, 3, 3, 1, 1, 0)
    W = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm_nt(A, W, M, N, K, geom=geom, out=out)
    torch.cuda.synchronize()
