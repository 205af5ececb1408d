"""Wall time of the phases of one step on the main stream (HIP events): forward+losses (incl. waiting for the teacher),
backward (incl. joining the weight-gradient stream), optimiser.  usage: python tests/tools/phase_times.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stil_tta_amd import STiLModel
from stil_tta_amd.driver import synthetic_batch
from stil_tta_amd.flat import StilAdam

fl = [8] * 16 + [1] * 48
torch.manual_seed(0)
m = STiLModel(dict(field_lengths=fl, num_classes=286, img_size=224, batch_size=256, start_epoch=35, repeat_ratio=1.0))
m.setup_device("cuda"); m.train(); m.current_epoch = 36
m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(286, 128)).cuda())
opt = StilAdam(m.flat, lr=1e-4)
batch = synthetic_batch(fl, 286, 256, 224, seed=1, device="cuda")
ev = lambda: torch.cuda.Event(enable_timing=True)
for it in range(6):
    e = [ev() for _ in range(4)]
    opt.zero_grad()
    e[0].record()
    loss = m.training_step(batch, 0)
    e[1].record()
    loss.backward()
    g = m.flat.grads  # joins the side stream
    e[2].record()
    opt.step()
    e[3].record()
    torch.cuda.synchronize()
    if it >= 2:
        print(f"forward+losses {e[0].elapsed_time(e[1]):7.2f} ms   backward {e[1].elapsed_time(e[2]):7.2f} ms   adam {e[2].elapsed_time(e[3]):5.2f} ms")
