"""A/B of the register prefetch depth of the short plain GEMMs (STIL_PD builds under stil_tta_amd/lib/exp): TFLOP/s on the
K < 512 shapes of the step and a checksum of the outputs (must be identical across builds).  Run once per build with
STIL_LIB_PATH set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import ops
dev = "cuda"
shapes = [(802816, 256, 64), (802816, 64, 64), (200704, 512, 128), (802816, 64, 256), (802816, 256, 128), (200704, 128, 256), (50176, 1024, 256), (16640, 512, 384)]
torch.manual_seed(0)
line = []
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    R = torch.randn(M, N, device=dev)
    for res in (False, True):
        ops.gemm_nt(A, W, M, N, K, out=out, resid=R if res else None, act=1 if res else 0); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            ops.gemm_nt(A, W, M, N, K, out=out, resid=R if res else None, act=1 if res else 0)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        line.append(f"({M},{N},{K}){'+res' if res else ''}: {2.0 * M * N * K / ms / 1e9:6.1f} TF {ms * 1e3:6.0f} us  sum {float(out.double().sum()):.6e}")
print(os.environ.get("STIL_LIB_PATH", "default"))
print("\n".join(line))
