"""How the oracle's step time on the GPU box's host cores depends on torch's thread count (the box reports 128 logical CPUs
but a job's share may be smaller): one oracle full_step (B=32, bench shape) per thread count.  Test infrastructure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import stil_oracle as O

fl = [8] * 16 + [1] * 48
hp = O.default_hparams(field_lengths=fl, num_classes=286, img_size=224, batch_size=32, start_epoch=0)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "default threads", torch.get_num_threads(), flush=True)
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except Exception as e:
    print("cpu.max unreadable", e)
for n in [int(x) for x in (sys.argv[1:] or ["16", "32", "64", "128"])]:
    torch.set_num_threads(n)
    sd = O.init_state(hp, seed=0)
    b = O.synthetic_batch(hp, 32, seed=2022)
    opt = {}
    t0 = time.perf_counter()
    O.full_step(sd, opt, 1, b, hp, 1)
    t1 = time.perf_counter()
    O.full_step(sd, opt, 2, b, hp, 1)
    t2 = time.perf_counter()
    print(f"threads {n}: first step {t1 - t0:.2f} s, second {t2 - t1:.2f} s", flush=True)
