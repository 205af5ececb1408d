"""Which operator of the ITC path makes projector_imaging.bias' gradient noisier on the device than in ATen?  The golden case
dvm_r50_b32_224 is stepped with single operators of that path replaced by their ATen-on-GPU equivalents; per variant the relative
L2 distance of the ITC heads' gradients from the float64 oracle (device decisions).  usage: python tests/tools/itc_noise.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import test_gpu_step as T
from oracle import stil_oracle as O
from oracle.make_golden import build_case, run_oracle64
from stil_tta_amd import ops
from stil_tta_amd.driver import train_step, host_cpu_share
from stil_tta_amd.flat import StilAdam
torch.set_num_threads(host_cpu_share())
hp, sd, batch, epoch, mr, mm = build_case("dvm_r50_b32_224")
keys = ["projector_imaging.bias", "projector_imaging.weight", "projector_tabular.bias", "projector_tabular.weight", "model.projection_ai.model.2.bias"]
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
o64 = None

def step(tag):
    global o64
    m = T._make_model(hp, {k: v.clone() for k, v in sd.items()}); m.current_epoch = epoch
    opt = StilAdam(m.flat, lr=hp.lr_eval)
    with T._trace_decisions() as trace:
        train_step(m, opt, T._to_dev(batch), mask_random=mr, mi_masks=mm)
        torch.cuda.synchronize()
        dec = T._device_decisions(m, trace)
    if o64 is None:
        o64 = run_oracle64(hp, sd, batch, epoch, mr, mm, decisions=dec)
        with O.force_decisions(*dec):
            o32 = O.full_step({k: v.clone() for k, v in sd.items()}, {}, 1, batch, hp, epoch, mr, mm)
        print(f"{'fp32 oracle (ATen CPU)':34s}", {k.split('.')[-2][-8:] + '.' + k.split('.')[-1]: f"{rel(o32['grads'][k], o64['grads'][k]):.2e}" for k in keys})
    p = T._named_params(m)
    print(f"{tag:34s}", {k.split('.')[-2][-8:] + '.' + k.split('.')[-1]: f"{rel(p[k]._gslot.cpu(), o64['grads'][k]):.2e}" for k in keys}, flush=True)

step("device (as shipped)")
orig_l2, orig_clip, orig_colsum = ops.l2norm, ops.clip_loss, ops.colsum
ops.l2norm = lambda x: F.normalize(x, dim=1)
step("l2norm -> ATen")
ops.l2norm = orig_l2
def clip_aten(f0, f1, Tt, lam0, gather=False):
    n0, n1 = F.normalize(f0, dim=1), F.normalize(f1, dim=1)
    Z = n0 @ n1.t() / Tt
    lab = torch.arange(len(Z), device=Z.device)
    return lam0 * F.cross_entropy(Z, lab) + (1 - lam0) * F.cross_entropy(Z.t(), lab), Z
ops.clip_loss = clip_aten
step("clip_loss -> ATen")
ops.l2norm = lambda x: F.normalize(x, dim=1)
step("clip_loss + l2norm -> ATen")
ops.clip_loss, ops.l2norm = orig_clip, orig_l2
def colsum_aten(X, out, M, N, *, ld=None, accumulate=0, scale=1.0):
    s = X.view(M, -1)[:, :N].sum(0) * scale
    out.copy_(out + s if accumulate else s)
ops.colsum = colsum_aten
step("colsum -> ATen")
ops.colsum = orig_colsum
