"""Micro-benchmark (GPU box) of the NT / TN GEMM kernels on the step's dominant shapes; A/B knobs in one process.
usage: python tests/tools/gemm_bench.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import ops
from stil_tta_amd._lib import lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
L = lib()
dev = "cuda"
# (M, N, K, k, stride, mode, H)  -- H = source spatial size for conv gathers
NT = [(50176, 256, 2304, 3, 1, 0, 14), (802816, 256, 64, 1, 1, 0, 56), (50176, 1024, 256, 1, 1, 0, 14), (16640, 512, 2048, 1, 1, 0, 0),
      (200704, 512, 128, 1, 1, 0, 28), (16640, 2048, 512, 1, 1, 0, 0), (802816, 64, 576, 3, 1, 0, 56), (200704, 128, 1152, 3, 1, 0, 28),
      (12544, 512, 4608, 3, 1, 0, 7), (802816, 64, 256, 1, 1, 0, 56)]

def run_nt(shape):
    M, N, K, k, s, mode, H = shape
    if k == 1:
        A = torch.randn(M, K, device=dev); geom = None
    else:
        C = K // (k * k); Nb = M // (H * H)
        A = torch.randn(Nb, H, H, C, device=dev); geom = (H, H, C, H, H, k, k, s, 1, 0)
    W = torch.randn(N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    ops.gemm_nt(A, W, M, N, K, geom=geom, out=out)
    torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(rounds):
        ops.gemm_nt(A, W, M, N, K, geom=geom, out=out)
    e0.record(); torch.cuda.synchronize()
    ms = s0.elapsed_time(e0) / rounds
    return ms, 2.0 * M * N * K / ms / 1e9

NT += [(50176, 256, 1024, 1, 1, 0, 14), (12544, 512, 2048, 1, 1, 0, 7), (12544, 2048, 512, 1, 1, 0, 7), (802816, 64, 64, 1, 1, 0, 56), (802816, 256, 128, 1, 1, 0, 56)]
def tune(v):
    return lambda: ops.TUNE.__setitem__("gemm", v)
# tune = variant + 100 * bk32 + 1000 * acc2 (include/stil_hip.h): auto / single-chain accumulation / per tile variant
variants = [("auto", tune(0)), ("bk16", tune(200)), ("bk32x2", tune(100)), ("bk32x1", tune(300))]
res = {}
for r in range(2):  # interleaved rounds
    for name, setter in variants:
        setter()
        for sh in NT:
            ms, tf = run_nt(sh)
            res.setdefault((name, sh), []).append(tf)
print(f"{'shape':46s} " + " ".join(f"{n:>11s}" for n, _ in variants))
for sh in NT:
    print(f"{str(sh):46s} " + " ".join(f"{max(res[(n, sh)]):11.1f}" for n, _ in variants))
ops.TUNE["gemm"] = 0

# ---- weight-gradient (TN) kernel: tile variants
TN = [(50176, 256, 2304, 3, 14), (802816, 64, 576, 3, 56), (802816, 256, 64, 1, 56), (200704, 128, 1152, 3, 28), (50176, 1024, 256, 1, 14),
      (16640, 512, 2048, 1, 0), (12544, 512, 4608, 3, 7), (200704, 512, 128, 1, 28), (3211264, 64, 160, 1, 0)]
def run_tn(shape):
    M, N, K, k, H = shape
    dY = torch.randn(M, N, device=dev)
    if k == 1:
        X = torch.randn(M, K, device=dev); geom = None
    else:
        C = K // 9; X = torch.randn(M // (H * H), H, H, C, device=dev); geom = (H, H, C, H, H, 3, 3, 1, 1)
    dW = torch.empty(N, K, device=dev) if k == 1 else torch.empty(N, K // 9, 3, 3, device=dev)
    ops.wgrad_tn(dY, X, dW, M, N, K, geom=geom)
    torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(rounds):
        ops.wgrad_tn(dY, X, dW, M, N, K, geom=geom)
    e0.record(); torch.cuda.synchronize()
    ms = s0.elapsed_time(e0) / rounds
    return ms, 2.0 * M * N * K / ms / 1e9
res = {}
for r in range(2):
    for v in (22, 11):
        ops.TUNE["wgrad"] = v
        for sh in TN:
            res.setdefault((v, sh), []).append(run_tn(sh)[1])
print(f"{'wgrad shape':46s}       t22        t11")
for sh in TN:
    print(f"{str(sh):46s} {max(res[(22, sh)]):10.1f} {max(res[(11, sh)]):10.1f}")
ops.TUNE["wgrad"] = 0

# ---- tile variants must agree bit for bit (same k order per output element)
for sh in [(50176, 256, 2304, 3, 1, 0, 14), (200704, 512, 128, 1, 1, 0, 28), (5000, 200, 96, 1, 1, 0, 0)]:
    M, N, K, k, s, mode, H = sh
    if k == 1:
        A = torch.randn(M, K, device=dev); geom = None
    else:
        C = K // (k * k); Nb = M // (H * H)
        A = torch.randn(Nb, H, H, C, device=dev); geom = (H, H, C, H, H, k, k, s, 1, 0)
    W = torch.randn(N, K, device=dev)
    outs = []
    for v in (0, 200, 100, 300):
        ops.TUNE["gemm"] = v
        outs.append(ops.gemm_nt(A, W, M, N, K, geom=geom).clone())
    torch.cuda.synchronize()
    print("bit-identical across tile variants", sh, [bool(torch.equal(outs[0], o)) for o in outs[1:]])
ops.TUNE["gemm"] = 0
