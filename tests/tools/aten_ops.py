"""Which ATen ops (device kernels AND device-to-device memcpys) does one training step still issue, and from which line of the
package?  (GPU box)  usage: python tests/tools/aten_ops.py [batch] [img]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from stil_tta_amd import STiLModel
from stil_tta_amd.driver import synthetic_batch, train_step
from stil_tta_amd.flat import StilAdam

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = int(sys.argv[2]) if len(sys.argv) > 2 else 128
fl = [8] * 16 + [1] * 48
torch.manual_seed(0)
m = STiLModel(dict(field_lengths=fl, num_classes=286, img_size=P, batch_size=B, start_epoch=35, repeat_ratio=1.0))
m.setup_device("cuda"); m.train(); m.current_epoch = 36
m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(286, 128)).cuda())
opt = StilAdam(m.flat, lr=1e-4)
batch = synthetic_batch(fl, 286, B, P, seed=1, device="cuda")
for _ in range(3):
    train_step(m, opt, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train_step(m, opt, batch)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
names = collections.Counter()
for ev in prof.events():
    names[ev.name] += 1
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue   # count the innermost op only
    frame = "?"
    for fr in (ev.stack or []):
        if "stil_tta_amd/" in fr:
            frame = fr.split("stil_tta_amd/")[-1].strip()
            break
    kinds = ",".join(sorted({k.name[:28] for k in ev.kernels})) if ev.kernels else "-"
    k = (ev.name, frame, kinds, str(ev.input_shapes)[:60])
    agg[k][0] += 1
    agg[k][1] += ev.device_time_total
for (name, frame, kinds, shp), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{n:5d} x {name:22s} {us:8.1f} us  {kinds:30s} {frame[:60]:60s} {shp}")
print("memcpy-like events:", {k: v for k, v in names.items() if "emcpy" in k or "copyBuffer" in k})
