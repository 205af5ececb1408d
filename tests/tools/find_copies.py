"""Which host lines issue the step's memcpy / memset / short ATen kernels?  One profiled step (torch.profiler, with stacks) of the
bench configuration; prints, per GPU activity name, the count and the innermost repo frame of the launching CPU op.
usage: python tests/tools/find_copies.py [--batch 256] [--img 224] [--variant dvm]      (measurement tool, not product code)"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256); ap.add_argument("--img", type=int, default=224)
ap.add_argument("--variant", default="dvm")
a = ap.parse_args()
from stil_tta_amd import STiLModel
from stil_tta_amd.driver import train_step, synthetic_batch
from stil_tta_amd.flat import StilAdam
extra, ncat, ncon, K = {}, 16, 48, 286
if a.variant == "cardiac":
    ncat, ncon, K = 26, 49, 2
    extra = dict(target="CAD", th1=0.85, beta=1.0, gamma=1.0, rate_pseudo=0.95, ema_momentum=0.4, lr_eval=1e-3)
if a.variant == "saint":
    extra = dict(tabular_encoder="saint")
fl = [(4 if a.variant == "cardiac" else 8)] * ncat + [1] * ncon
torch.manual_seed(2022)
m = STiLModel(dict(field_lengths=fl, num_classes=K, img_size=a.img, batch_size=a.batch, start_epoch=35, repeat_ratio=1.0, seed=2022, **extra))
m.setup_device("cuda"); m.train(); m.current_epoch = 36
m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(K, 128)).cuda())
opt = StilAdam(m.flat, lr=1e-4)
batch = synthetic_batch(fl, K, a.batch, a.img, seed=2022, device="cuda")
for _ in range(3):
    train_step(m, opt, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(m, opt, batch)
    torch.cuda.synchronize()
ev = prof.events()
# map: launching CPU op (by correlation: kernels are children of cpu ops in the event tree)
by = collections.defaultdict(lambda: collections.Counter())
n_gpu = 0
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        continue
    for k in e.kernels:
        n_gpu += 1
        name = k.name
        short = name.startswith("Memcpy") or name.startswith("Memset") or "at::native" in name or "rocclr" in name
        if not short:
            continue
        frame = "?"
        for fr in (e.stack or []):
            if "/stil_tta_amd/" in fr or "bench.py" in fr:
                frame = fr.strip().replace(ROOT + "/", "")
                break
        by[name[:70]][(e.name[:40], frame[:110])] += 1
print("GPU activities attributed:", n_gpu)
for name, c in sorted(by.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{sum(c.values()):5d}  {name}")
    for (op, frame), n in c.most_common(12):
        print(f"        {n:4d}  {op:40s} {frame}")
