"""Diagnostic (GPU box): per-loss-term gradient error of the HIP path vs the float64 oracle for one golden case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import stil_oracle as O
from oracle.make_golden import build_case
from stil_tta_amd import STiLModel

name = sys.argv[1] if len(sys.argv) > 1 else "cardiac_r50"
watch = ["model.reduce.bias", "model.transformer.0.attn.qkv.bias", "model.classifier_multimodal.weight", "projector_multimodal.layers.2.weight",
         "model.projection_ai.model.0.bias", "model.encoder_tabular.norm.weight"]
hp, sd, batch, epoch, mr, mm = build_case(name)
f64 = torch.float64
s64 = {k: (v.clone().to(f64) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
b64 = {k: ([v[0][0].to(f64), v[0][1].to(f64)], [v[1][0].to(f64), v[1][1].to(f64)], v[2], v[3].to(f64), v[4]) for k, v in batch.items()}
keys = O.trainable_keys(s64)
for k in keys:
    s64[k].requires_grad_(True)
o = O.training_step(s64, b64, hp, epoch, mr, mm)
d = dict(vars(hp)); d["mi_dropout"] = False
m = STiLModel(d); m.load_state_dict(sd); m.setup_device("cuda"); m.train(); m.current_epoch = epoch
dev_batch = {k: ([v[0][0].cuda(), v[0][1].cuda()], [v[1][0].cuda(), v[1][1].cuda()], v[2].cuda(), v[3].cuda(), v[4].cuda()) for k, v in batch.items()}
m.flat.zero_grad()
m.training_step(dev_batch, 0, mask_random=mr, mi_masks=mm)
params = {n: p for n, p in m.named_parameters() if not n.startswith("ema.")}
terms = ["loss_ce", "loss_itc", "loss_club_i", "loss_club_i_est", "loss_club_t", "loss_club_t_est", "loss_pt", "loss_m_u", "loss_i_u", "loss_t_u"]
for t in terms:
    g64 = torch.autograd.grad(o[t], [s64[k] for k in watch], retain_graph=True, allow_unused=True)
    m.flat.zero_grad()
    if m.last[t].requires_grad:
        m.last[t].backward(retain_graph=True)
    torch.cuda.synchronize()
    line = f"{t:16s} val hip {float(m.last[t]):.6f} f64 {float(o[t]):.6f} |"
    for k, g in zip(watch, g64):
        if g is None or float(g.norm()) == 0:
            line += "      -  "
            continue
        e = float((params[k]._gslot.cpu().double() - g).norm() / g.norm())
        line += f" {e:8.1e}"
    print(line)
print("columns:", watch)
