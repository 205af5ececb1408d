"""Diagnostic (GPU box): is one training step bit-reproducible across processes / allocator histories?
usage: det_probe.py [poison]   poison in {none, nan, big}: pre-fill the caching allocator's blocks with that value."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import STiLModel
from stil_tta_amd.driver import train_step, synthetic_batch
from stil_tta_amd.flat import StilAdam
poison = sys.argv[1] if len(sys.argv) > 1 else "none"
FL = [3, 4] + [1] * 3
torch.manual_seed(0)
m = STiLModel(dict(model="resnet18", embedding_dim=512, field_lengths=FL, num_classes=5, start_epoch=0, batch_size=16, th1=0.3, mi_dropout=False))
m.setup_device("cuda"); m.train(); m.current_epoch = 1
m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(5, 128, generator=torch.Generator().manual_seed(1))).cuda())
opt = StilAdam(m.flat, lr=1e-3)
batch = synthetic_batch(FL, 5, 16, 64, seed=3, device="cuda")
mr = (torch.arange(14) % 2 == 0).cuda()
if poison != "none":
    val = float("nan") if poison == "nan" else 3.0e30
    blocks = [torch.full((n,), val, device="cuda") for n in [1 << 28] * 8 + [1 << 24] * 16 + [1 << 20] * 64 + [1 << 14] * 256 + [256] * 512]
    torch.cuda.synchronize()
    del blocks   # stays cached: the step's torch.empty() buffers now start out poisoned
sums = []
for step in range(2):
    train_step(m, opt, batch, mask_random=mr)
    torch.cuda.synchronize()
    h = hashlib.sha1(m.flat.params.cpu().numpy().tobytes()).hexdigest()[:12]
    hg = hashlib.sha1(m.flat.grads.cpu().numpy().tobytes()).hexdigest()[:12]
    sums.append((h, hg, float(m.last["loss"]), bool(torch.isfinite(m.flat.params).all())))
print(poison, os.environ.get("STIL_WGRAD_STREAM", "1"), sums)
