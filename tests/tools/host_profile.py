"""Where the HOST time of an eager step goes (cProfile over a few steps after warm-up): the small per-GPU batches are bound by ~1000
launches x the Python / autograd / ctypes cost of each.   usage: python tests/tools/host_profile.py [batch] [img]   (measurement tool)"""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import STiLModel
from stil_tta_amd.driver import synthetic_batch, train_step
from stil_tta_amd.flat import StilAdam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
img = int(sys.argv[2]) if len(sys.argv) > 2 else 224
fl = [8] * 16 + [1] * 48
torch.manual_seed(0)
m = STiLModel(dict(field_lengths=fl, num_classes=286, img_size=img, batch_size=B, start_epoch=35, repeat_ratio=1.0))
m.setup_device("cuda"); m.train(); m.current_epoch = 36
m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(286, 128)).cuda())
opt = StilAdam(m.flat, lr=1e-4)
batch = synthetic_batch(fl, 286, B, img, seed=1, device="cuda")
for _ in range(4):
    train_step(m, opt, batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    train_step(m, opt, batch)
t_host = time.perf_counter() - t0          # the host's time to ISSUE ten steps (no sync inside)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"B={B} {img}px: host issues a step in {t_host * 100:.2f} ms; with the final sync {t_all * 100:.2f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    train_step(m, opt, batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
# backward runs in the autograd engine's device thread, which cProfile does not see: once more with the engine single-threaded
print("---- backward in the calling thread (torch.autograd.set_multithreading_enabled(False)) ----")
with torch.autograd.set_multithreading_enabled(False):
    train_step(m, opt, batch)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        train_step(m, opt, batch)
    pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
