"""Forward noise of the whole step at the BASELINE shape (golden case dvm_r50_b32_224): per forward quantity, the device's and the
fp32 oracle's relative L2 distance from the float64 oracle evaluated on the device's ReLU / max-pool decisions; and the same for
the most ill-conditioned gradients.  usage: python tests/tools/step_noise.py [case]      (measurement tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import test_gpu_step as T
from oracle import stil_oracle as O
from oracle.make_golden import build_case, run_oracle64
from stil_tta_amd.driver import train_step, host_cpu_share
from stil_tta_amd.flat import StilAdam
torch.set_num_threads(host_cpu_share())
name = sys.argv[1] if len(sys.argv) > 1 else "dvm_r50_b32_224"
hp, sd, batch, epoch, mr, mm = build_case(name)
m = T._make_model(hp, sd); m.current_epoch = epoch
opt = StilAdam(m.flat, lr=hp.lr_eval)
with T._trace_decisions() as trace:
    train_step(m, opt, T._to_dev(batch), mask_random=mr, mi_masks=mm)
    torch.cuda.synchronize()
    dec = T._device_decisions(m, trace)
o64 = run_oracle64(hp, sd, batch, epoch, mr, mm, decisions=dec)
with O.force_decisions(*dec):
    o32 = O.full_step({k: v.clone() for k, v in sd.items()}, {}, 1, batch, hp, epoch, mr, mm)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
print(f"{'quantity':22s} {'device vs f64':>14s} {'fp32 oracle vs f64':>20s}")
for k in T.SCALARS + T.FWD_KEYS:
    if k in o64 and torch.is_tensor(o64[k]) and o64[k].is_floating_point():
        print(f"{k:22s} {rel(m.last[k].detach().cpu(), o64[k]):14.2e} {rel(o32[k], o64[k]):20.2e}")
params = T._named_params(m)
rows = []
for k, g in o64["grads"].items():
    if g is None: continue
    rows.append((rel(params[k]._gslot.cpu(), g), rel(o32["grads"][k], g), k))
rows.sort(reverse=True)
print("gradients with the largest device error:")
for d, r, k in rows[:12]:
    print(f"  {k:60s} device {d:.2e}  fp32 oracle {r:.2e}")
