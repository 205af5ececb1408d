import sys, os
sys.path.insert(0, "/root/repo")
import torch
from stil_tta_amd import ops
dev="cuda"
TN = [(50176, 256, 2304, 3, 14), (802816, 64, 576, 3, 56), (802816, 256, 64, 1, 56), (200704, 128, 1152, 3, 28), (50176, 1024, 256, 1, 14),
      (16640, 512, 2048, 1, 0), (12544, 512, 4608, 3, 7), (200704, 512, 128, 1, 28), (3211264, 64, 160, 1, 0), (50176, 256, 1024, 1, 14), (12544, 2048, 512, 1, 7)]
out=[]
for M, N, K, k, H in TN:
    dY = torch.randn(M, N, device=dev)
    if k == 1:
        X = torch.randn(M, K, device=dev); geom = None
    else:
        C = K // 9; X = torch.randn(M // (H * H), H, H, C, device=dev); geom = (H, H, C, H, H, 3, 3, 1, 1)
    dW = torch.empty(N, K, device=dev) if k == 1 else torch.empty(N, K // 9, 3, 3, device=dev)
    ops.wgrad_tn(dY, X, dW, M, N, K, geom=geom); torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(5): ops.wgrad_tn(dY, X, dW, M, N, K, geom=geom)
    e0.record(); torch.cuda.synchronize()
    ms = s0.elapsed_time(e0) / 5
    out.append(f"{2.0*M*N*K/ms/1e9:6.1f}")
print(os.environ.get("STIL_W11"), os.environ.get("STIL_W22"), " ".join(out))
