"""Do an MFMA-bound GEMM and an HBM-bound streaming pass overlap when they run on two streams?  (GPU box)
Times, per pair: GEMM alone, stream pass alone, both launched together (GEMM on one stream, the pass on another), for a
few sizes of the streaming pass.  perfect overlap -> t_both ~ max; none -> t_both ~ sum.
The second part needs tests/tools/libstream_probe.so (a probe kernel with a chosen grid / loads in flight; build it in the
build container: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tests/tools/libstream_probe.so tests/tools/stream_probe.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import ops

dev = "cuda"
M, N, K, H = 50176, 256, 2304, 14
A = torch.randn(M // (H * H), H, H, K // 9, device=dev)
W = torch.randn(N, K, device=dev)
out = torch.empty(M, N, device=dev)
geom = (H, H, K // 9, H, H, 3, 3, 1, 1, 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def gemm(n=4):
    for _ in range(n):
        ops.gemm_nt(A, W, M, N, K, geom=geom, out=out)


for rows in (802816, 200704):
    x = torch.randn(rows, 256, device=dev); y = torch.randn(rows, 256, device=dev); z = torch.empty_like(x)

    def stream_pass(n=4):
        for _ in range(n):
            ops.axpby(x.view(-1), y.view(-1), 1.0, 1.0, out=z.view(-1))

    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            gemm()
        with torch.cuda.stream(s2):
            stream_pass()
        cur.wait_stream(s1); cur.wait_stream(s2)

    tg, ts, tb = timed(gemm), timed(stream_pass), timed(both)
    gb = 3 * rows * 256 * 4 * 4 / 1e9
    print(f"stream pass {rows}x256 (4 launches, {gb:.1f} GB): gemm x4 alone {tg:.3f} ms ({4 * 2 * M * N * K / tg / 1e9:.0f} TF), pass alone {ts:.3f} ms "
          f"({gb / ts:.2f} TB/s), together {tb:.3f} ms  -> sum {tg + ts:.3f}, max {max(tg, ts):.3f}, overlap gain {(tg + ts - tb) / min(tg, ts) * 100:.0f} % of the shorter")


# ---- the same with a probe streaming kernel of chosen grid / loads in flight (tests/tools/stream_probe.hip)
import ctypes
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstream_probe.so")
if os.path.exists(so):
    P = ctypes.CDLL(so)
    P.probe_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rows = 802816
    x = torch.randn(rows, 256, device=dev); y = torch.randn(rows, 256, device=dev); z = torch.empty_like(x)
    n4 = x.numel() // 4
    tg = timed(gemm)
    print(f"gemm x4 alone {tg:.3f} ms; streaming pass = 4 launches of z = x + y over {rows}x256 (9.9 GB)")
    for grid, U in [(16384, 1), (8192, 1), (2048, 1), (2048, 4), (1024, 4), (1024, 8), (512, 8), (256, 8), (512, 4), (256, 4)]:
        def stream_pass(n=4):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(n):
                P.probe_stream(x.data_ptr(), y.data_ptr(), z.data_ptr(), n4, grid, U, st)

        def both():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                gemm()
            with torch.cuda.stream(s2):
                stream_pass()
            cur.wait_stream(s1); cur.wait_stream(s2)
        ts, tb = timed(stream_pass), timed(both)
        print(f"  grid {grid:6d} x {U} loads: pass alone {ts:.3f} ms ({9.866 / ts:.2f} TB/s), together {tb:.3f} ms, sum {tg + ts:.3f}, "
              f"overlap gain {(tg + ts - tb) / min(tg, ts) * 100:4.0f} % of the shorter")
