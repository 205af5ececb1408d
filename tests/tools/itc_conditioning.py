"""CPU-only study of the ITC head's conditioning at the BASELINE shape (golden case dvm_r50_b32_224): how the error of
projector_imaging.bias' gradient splits into (a) the head's own operator rounding given its inputs and (b) the input features'
rounding noise propagated through the exact head -- and how differently the head reacts to a BATCH-COHERENT perturbation of x_ai
(the same vector added to every row) and to a per-row one of the same norm.  usage: python tests/tools/itc_conditioning.py
(measurement tool; the device side of the same decomposition is tests/tools/itc_noise.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from oracle import stil_oracle as O
from oracle.make_golden import build_case, run_oracle64
from itc_head import head_grads, rel, coherent_split

name = sys.argv[1] if len(sys.argv) > 1 else "dvm_r50_b32_224"
hp, sd, batch, epoch, mr, mm = build_case(name)
with O.record_decisions() as d:
    o32 = O.full_step({k: v.clone() for k, v in sd.items()}, {}, 1, batch, hp, epoch, mr, mm)
dec = (d["relu"], d["pool"])
o64 = run_oracle64(hp, sd, batch, epoch, mr, mm, decisions=dec)
KEYS = ("projector_imaging.bias", "projector_imaging.weight", "projector_tabular.bias", "projector_tabular.weight")
T, lam0, beta = float(hp.temperature), float(hp.lambda_0), float(hp.beta)
print("full step, fp32 oracle vs float64:", {k: f"{rel(o32['grads'][k], o64['grads'][k]):.2e}" for k in KEYS})
h64 = head_grads(sd, o64["x_ai"], o64["x_at"], T, lam0, torch.float64)
print("isolated head reproduces the step's gradient (beta * head):", {k: f"{rel(beta * h64[k], o64['grads'][k]):.1e}" for k in KEYS})
h32_on32 = head_grads(sd, o32["x_ai"], o32["x_at"], T, lam0, torch.float32)
h64_on32 = head_grads(sd, o32["x_ai"], o32["x_at"], T, lam0, torch.float64)
print("(a) operator noise  : head fp32 vs head f64, both on the fp32 features :", {k: f"{rel(h32_on32[k], h64_on32[k]):.2e}" for k in KEYS})
print("(b) feature noise   : head f64 on fp32 features vs on f64 features     :", {k: f"{rel(h64_on32[k], h64[k]):.2e}" for k in KEYS})
for nm in ("x_ai", "x_at"):
    e = o32[nm].double() - o64[nm]
    c, r = coherent_split(e)
    print(f"{nm}: relL2 error {rel(o32[nm], o64[nm]):.2e}; batch-coherent part {c / float(o64[nm].norm()):.2e}, per-row part {r / float(o64[nm].norm()):.2e}")
e = h64_on32["dz_i"].double() - h64["dz_i"]
c, r = coherent_split(e)
print(f"dz_i (f64 head, fp32 vs f64 features): coherent {c / float(h64['dz_i'].norm()):.2e}, per-row {r / float(h64['dz_i'].norm()):.2e}; "
      f"|colsum dz_i| / |dz_i| = {float(h64['dz_i'].sum(0).norm() / h64['dz_i'].norm()):.3f}")
# sensitivity of the head to perturbations of x_ai of relative size 2e-5 (what fp32 leaves on that tensor)
g = torch.Generator().manual_seed(0)
x = o64["x_ai"]
B, D = x.shape
eps = 2e-5 * float(x.norm())
res = {"coherent": [], "per-row": []}
for trial in range(8):
    v = torch.randn(1, D, generator=g, dtype=torch.float64).expand(B, D)
    w = torch.randn(B, D, generator=g, dtype=torch.float64)
    w = w - w.mean(0, keepdim=True)
    for tag, p in (("coherent", v), ("per-row", w)):
        hp_ = head_grads(sd, x + p * (eps / float(p.norm())), o64["x_at"], T, lam0, torch.float64)
        res[tag].append([rel(hp_[k], h64[k]) for k in KEYS])
for tag, rows in res.items():
    t = torch.tensor(rows)
    print(f"x_ai + 2e-5 {tag:9s} perturbation -> gradient relL2 change (mean over 8 draws):", {k: f"{float(t[:, i].mean()):.2e}" for i, k in enumerate(KEYS)})
