"""Local-error ladder of the ITC head ON THE DEVICE: every operator of projector -> l2norm -> l2norm -> logits -> CLIP loss and of its
backward is checked against a float64 evaluation of THAT operator on the device's own inputs to it, so the operator that carries
the device's excess noise (tests/tools/itc_noise.py, decomposition (a)) shows up as the one rung whose local error is not ~1e-7.
Inputs: the fp32 oracle's x_ai / x_at of golden case dvm_r50_b32_224.      usage: python tests/tools/itc_ladder.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, torch.nn.functional as F
import test_gpu_step as T
from oracle import stil_oracle as O
from oracle.make_golden import build_case
from stil_tta_amd import ops
from stil_tta_amd.driver import host_cpu_share
from itc_head import head_grads, rel
torch.set_num_threads(host_cpu_share())
hp, sd, batch, epoch, mr, mm = build_case("dvm_r50_b32_224")
with torch.no_grad():
    o32 = O.training_step({k: v.clone() for k, v in sd.items()}, batch, hp, epoch, mr, mm)
Tt, lam0 = float(hp.temperature), float(hp.lambda_0)
m = T._make_model(hp, {k: v.clone() for k, v in sd.items()})
m.flat.zero_grad()
d64 = lambda t: t.detach().cpu().double()
x0 = o32["x_ai"].cuda().requires_grad_(True)
x1 = o32["x_at"].cuda().requires_grad_(True)
z0 = m.projector_imaging.run(x0); z1 = m.projector_tabular.run(x1)
f0, f1 = ops.l2norm(z0), ops.l2norm(z1)
n0, n1 = ops.l2norm(f0), ops.l2norm(f1)
Z = ops.MatmulNTFn.apply(n0, n1, 1.0 / Tt)
loss = ops.ClipFromLogitsFn.apply(Z, lam0)
for t in (z0, z1, f0, f1, n0, n1, Z):
    t.retain_grad()
loss.backward()
ops.join_side()
torch.cuda.synchronize()
Wi, bi = sd["projector_imaging.weight"].double(), sd["projector_imaging.bias"].double()
B = len(x0)
eye = torch.eye(B, dtype=torch.float64)
def l2f(x):
    n = x.pow(2).sum(1, keepdim=True).sqrt().clamp_min(1e-12)
    return x / n, n
def l2b(g, xin):   # backward of y = x / |x| at x = xin, exact
    y, n = l2f(xin)
    return (g - y * (g * y).sum(1, keepdim=True)) / n
print("rung: local relative L2 error of the DEVICE operator, float64 operator on the device's own inputs")
print(f"  z0 = Linear(x0)                 {rel(z0, F.linear(d64(x0), Wi, bi)):.2e}")
print(f"  f0 = normalize(z0)              {rel(f0, l2f(d64(z0))[0]):.2e}")
print(f"  n0 = normalize(f0)              {rel(n0, l2f(d64(f0))[0]):.2e}")
Zr = d64(n0) @ d64(n1).t() / Tt
print(f"  Z = n0 n1^T / T                 {rel(Z, Zr):.2e}   (max |Z| {float(Z.abs().max()):.3f}, spread of Z {float(Z.max() - Z.min()):.3e})")
Zd = d64(Z)
lab = torch.arange(B)
print(f"  loss(Z)                         {abs(float(loss) - float(lam0 * F.cross_entropy(Zd, lab) + (1 - lam0) * F.cross_entropy(Zd.t(), lab))):.2e} (absolute)")
dZr = (lam0 * (torch.softmax(Zd, 1) - eye) + (1 - lam0) * (torch.softmax(Zd.t(), 1).t() - eye)) / B
print(f"  dZ = dloss/dZ                   {rel(Z.grad, dZr):.2e}   row sums of dZ: device {float(d64(Z.grad).sum(1).abs().max()):.2e}, float64 {float(dZr.sum(1).abs().max()):.2e}")
print(f"  grand total of dZ (exactly 0 in exact arithmetic): device {float(d64(Z.grad).sum()):+.3e}, float64 {float(dZr.sum()):+.3e};  |dZ|_1 = {float(dZr.abs().sum()):.3e}")
print(f"  dn0 = dZ n1 / T                 {rel(n0.grad, d64(Z.grad) @ d64(n1) / Tt):.2e}")
print(f"  dn1 = dZ^T n0 / T               {rel(n1.grad, d64(Z.grad).t() @ d64(n0) / Tt):.2e}")
print(f"  df0 = normalize'(f0)^T dn0      {rel(f0.grad, l2b(d64(n0.grad), d64(f0))):.2e}   |df0| / |dn0| = {float(f0.grad.norm() / n0.grad.norm()):.3e}")
print(f"  dz0 = normalize'(z0)^T df0      {rel(z0.grad, l2b(d64(f0.grad), d64(z0))):.2e}   |dz0| |z0| / |df0| = {float(z0.grad.norm() * z0.norm(dim=1).mean() / f0.grad.norm()):.3e}")
p = T._named_params(m)
gb, gw = p["projector_imaging.bias"]._gslot, p["projector_imaging.weight"]._gslot
print(f"  db = colsum(dz0)                {rel(gb, d64(z0.grad).sum(0)):.2e}   |colsum| / |dz0| = {float(d64(z0.grad).sum(0).norm() / d64(z0.grad).norm()):.3e}")
print(f"  dW = dz0^T x0                   {rel(gw, d64(z0.grad).t() @ d64(x0)):.2e}")
print(f"  dx0 = dz0 W                     {rel(x0.grad, d64(z0.grad) @ Wi):.2e}")
h64 = head_grads(sd, o32["x_ai"], o32["x_at"], Tt, lam0, torch.float64)
h32 = head_grads(sd, o32["x_ai"], o32["x_at"], Tt, lam0, torch.float32)
print("end to end on these inputs (distance from the float64 head):")
print(f"  device   bias {rel(gb, h64['projector_imaging.bias']):.2e}  weight {rel(gw, h64['projector_imaging.weight']):.2e}  dz0 {rel(z0.grad, h64['dz_i']):.2e}")
print(f"  ATen CPU bias {rel(h32['projector_imaging.bias'], h64['projector_imaging.bias']):.2e}  weight {rel(h32['projector_imaging.weight'], h64['projector_imaging.weight']):.2e}  dz0 {rel(h32['dz_i'], h64['dz_i']):.2e}")
# propagated: each stage's device value pushed through the REST of the chain in float64 -> where the end-to-end error enters
def finish_from(stage, val):
    """float64 continuation of the backward chain from `stage` (device value `val`) down to dz0"""
    if stage == "dZ":
        val = val @ d64(n1) / Tt; stage = "dn0"
    if stage == "dn0":
        val = l2b(val, d64(f0)); stage = "df0"
    if stage == "df0":
        val = l2b(val, d64(z0)); stage = "dz0"
    return val
ref = h64["dz_i"]
print("dz0 error when the chain is float64 from the given device quantity on (where the error enters):")
for stage, val in (("dZ", d64(Z.grad)), ("dn0", d64(n0.grad)), ("df0", d64(f0.grad)), ("dz0", d64(z0.grad))):
    v = finish_from(stage, val)
    print(f"  from device {stage:4s}: dz0 {rel(v, ref):.2e}   bias {rel(v.sum(0), h64['projector_imaging.bias']):.2e}")
v = finish_from("dZ", dZr)
print(f"  from float64 dZ(Z_device): dz0 {rel(v, ref):.2e}   bias {rel(v.sum(0), h64['projector_imaging.bias']):.2e}   <- the forward's rounding alone (Z, n0, f0, z0 of the device)")
