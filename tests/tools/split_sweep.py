"""Split-K sweep of the NT products a SMALL per-GPU batch launches (B = 32 at 224 px; the cardiac share of 16 at 128 px): every shape
with s = 1 (unsplit), the library's policy, and forced slice counts (stil_gemm_nt_force_splits), back-to-back launches on one stream
as a hipGraph replay issues them.   usage: python tests/tools/split_sweep.py [b32|c16] [nt|tn]        (measurement tool)
tn: the weight-gradient products dW = dY^T X of the same layers (stil_wgrad_tn: slab partials over M + the ordered reduce), block tile
11 (64x64) / 22 (128x128) x forced slab counts (stil_wgrad_force_splits)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import ops
from stil_tta_amd._lib import lib
which = sys.argv[1] if len(sys.argv) > 1 else "b32"
B, H1 = (32, 56) if which == "b32" else (16, 32)
T = 65 if which == "b32" else 76          # tokens per sample of the tabular transformer
SH = []
for li, (c, H) in enumerate(((64, H1), (128, H1 // 2), (256, H1 // 4), (512, H1 // 8))):
    M = B * H * H
    SH += [(M, c, 4 * c, 1, H), (M, c, 9 * c, 3, H), (M, 4 * c, c, 1, H)]
    if li:
        SH += [(M, c, 2 * c, 1, H)]
SH += [(B * T, 512, 2048, 1, 0), (B * T, 2048, 512, 1, 0), (B * T, 1536, 512, 1, 0), (B * T, 512, 512, 1, 0)]
SWEEP = (1, 2, 3, 4, 6, 8, 12, 16)
L = lib()
def run(M, N, K, k, H, rounds=16):
    if k == 1:
        A = torch.randn(M, K, device="cuda"); geom = None
    else:
        C = K // 9; A = torch.randn(M // (H * H), H, H, C, device="cuda"); geom = (H, H, C, H, H, 3, 3, 1, 1, 0)
    W = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    # `rounds` launches captured into a hipGraph and replayed: GPU-side back-to-back time (eager launches through ctypes are host-bound
    # below ~15 us per launch, which is what the small shapes take)
    ops.gemm_nt(A, W, M, N, K, geom=geom, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gemm_nt(A, W, M, N, K, geom=geom, out=out)
        with torch.cuda.graph(g, stream=side):
            for _ in range(rounds):
                ops.gemm_nt(A, W, M, N, K, geom=geom, out=out)
    torch.cuda.current_stream().wait_stream(side)
    best = 1e9
    for rep in range(4):
        g.replay(); torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        g.replay()
        e0.record(); torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(e0) / rounds)
    del g
    return best * 1e3
if len(sys.argv) > 2 and sys.argv[2] == "tn":
    def run_tn(M, N, K, k, H, tune, rounds=16):
        dY = torch.randn(M, N, device="cuda")
        if k == 1:
            X = torch.randn(M, K, device="cuda"); geom = None
        else:
            C = K // 9; X = torch.randn(M // (H * H), H, H, C, device="cuda"); geom = (H, H, C, H, H, 3, 3, 1, 1)
        dW = torch.empty(N, K, device="cuda") if k == 1 else torch.empty(N, K // 9, 3, 3, device="cuda")
        ops.TUNE["wgrad"] = tune
        ops.wgrad_tn(dY, X, dW, M, N, K, geom=geom); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.wgrad_tn(dY, X, dW, M, N, K, geom=geom)
            with torch.cuda.graph(g, stream=side):
                for _ in range(rounds):
                    ops.wgrad_tn(dY, X, dW, M, N, K, geom=geom)
        torch.cuda.current_stream().wait_stream(side)
        best = 1e9
        for rep in range(4):
            g.replay(); torch.cuda.synchronize()
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record(); g.replay(); e0.record(); torch.cuda.synchronize()
            best = min(best, s0.elapsed_time(e0) / rounds)
        del g
        ops.TUNE["wgrad"] = 0
        return best * 1e3
    TS = (1, 2, 4, 8, 16, 32, 64)
    print(f"{which} tn: us per wgrad (partials + reduce); the forward shape (M, N, K) -> dW[N, K] summed over M;  policy | tile 11: s = " + " ".join(map(str, TS)) + " | tile 22: same")
    tp = tb = 0.0
    for (M, N, K, k, H) in SH:
        L.wgrad_force_splits(0)
        pol = run_tn(M, N, K, k, H, 0)
        rows = {}
        for tv in (11, 22):
            rows[tv] = []
            for s_ in TS:
                if s_ > max(1, M // 64):
                    rows[tv].append(None); continue
                L.wgrad_force_splits(s_)
                rows[tv].append(run_tn(M, N, K, k, H, tv))
        L.wgrad_force_splits(0)
        b = min(x for tv in rows for x in rows[tv] if x is not None)
        tp += pol; tb += b
        fmt = lambda r: " ".join("   -- " if x is None else f"{x:6.1f}" for x in r)
        print(f"({M:6d},{N:5d},{K:5d},k{k}) policy s={L.wgrad_splits(M, N, K, 0):3d} {pol:6.1f} | {fmt(rows[11])} | {fmt(rows[22])} | best {b:6.1f} = {2.0 * M * N * K / b / 1e6:5.1f} TF")
    print(f"sum: policy {tp:.1f} us, best per shape {tb:.1f} us")
    sys.exit(0)
print(f"{which}: us per launch;  tiles = 64x64 tiles, kt = 32-deep k-tiles;  columns: policy | forced s = " + " ".join(str(s) for s in SWEEP))
tot_pol = tot_best = 0.0
for (M, N, K, k, H) in SH:
    tiles = -(-M // 64) * -(-N // 64)
    L.gemm_nt_force_splits(0)
    pol = run(M, N, K, k, H)
    row = []
    for s in SWEEP:
        if s > 1 and (s > K // 64 or tiles > 4096):
            row.append(None); continue
        L.gemm_nt_force_splits(s)                           # 1 = unsplit
        row.append(run(M, N, K, k, H))
    L.gemm_nt_force_splits(0)
    b = min(x for x in row if x is not None)
    tot_pol += pol; tot_best += b
    ideal = 2.0 * M * N * K / 105e12 * 1e6
    print(f"({M:6d},{N:5d},{K:5d},k{k}) tiles {tiles:5d} kt {K // 32:4d} | {pol:6.1f} | " + " ".join("   -- " if x is None else f"{x:6.1f}" for x in row) +
          f" | best s={SWEEP[row.index(b)]:2d}  {2.0 * M * N * K / b / 1e6:6.1f} TF  (105 TF = {ideal:5.1f} us)")
print(f"sum: policy {tot_pol:.1f} us, best per shape {tot_best:.1f} us")
