"""CPU emulation of the DEVICE's formulas for the ITC head (csrc/loss.hip clip_* kernels, l2norm_*, ops.MatmulNTFn) in fp32, one
variation at a time, against the float64 head on the same fp32 features: which formula carries the device's operator noise
(tests/tools/itc_noise.py, decomposition (a))?   usage: python tests/tools/itc_emulate.py          (measurement tool, CPU only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from oracle import stil_oracle as O
from oracle.make_golden import build_case
from itc_head import head_grads, rel

hp, sd, batch, epoch, mr, mm = build_case("dvm_r50_b32_224")
o32 = O.full_step({k: v.clone() for k, v in sd.items()}, {}, 1, batch, hp, epoch, mr, mm)
T, lam0 = float(hp.temperature), float(hp.lambda_0)
xa, xt = o32["x_ai"], o32["x_at"]
h64 = head_grads(sd, xa, xt, T, lam0, torch.float64)
h32 = head_grads(sd, xa, xt, T, lam0, torch.float32)
KEYS = ["projector_imaging.bias", "projector_imaging.weight", "projector_tabular.bias", "projector_tabular.weight"]
print(f"{'ATen autograd fp32':46s}", {k.split('.')[0][-7:] + '.' + k.split('.')[1]: f"{rel(h32[k], h64[k]):.2e}" for k in KEYS})


def emulate(lse_form="device", dtype=torch.float32, alpha_mul=True, dz_form="device"):
    f = dtype
    Wi, bi = sd["projector_imaging.weight"].to(f), sd["projector_imaging.bias"].to(f)
    Wt, bt = sd["projector_tabular.weight"].to(f), sd["projector_tabular.bias"].to(f)
    x0, x1 = xa.to(f), xt.to(f)
    z0, z1 = x0 @ Wi.t() + bi, x1 @ Wt.t() + bt
    def l2f(x):
        n = x.pow(2).sum(1, keepdim=True).sqrt().clamp_min(1e-12)
        return x / n, n
    f0, nz0 = l2f(z0); f1, nz1 = l2f(z1)
    n0, nf0 = l2f(f0); n1, nf1 = l2f(f1)
    B = len(x0)
    Z = (n0 @ n1.t()) * torch.tensor(1.0 / T, dtype=f) if alpha_mul else (n0 @ n1.t()) / T
    eye = torch.eye(B, dtype=f)
    def probs(Zm):      # row softmax of Zm in the chosen formulation
        m = Zm.max(1, keepdim=True)[0]
        s = (Zm - m).exp().sum(1, keepdim=True)
        if lse_form == "device":            # lse = m + log(s);  p = exp(z - lse)             (loss.hip lse_rows_kernel / clip_dz_kernel)
            return (Zm - (m + s.log())).exp()
        if lse_form == "split":             # p = exp((z - m) - log(s))                       (ATen's log_softmax)
            return ((Zm - m) - s.log()).exp()
        return (Zm - m).exp() / s           # "div": exp(z - m) / s
    pr, pc = probs(Z), probs(Z.t()).t()
    dZ = (lam0 * (pr - eye) + (1 - lam0) * (pc - eye)) / B
    a = torch.tensor(1.0 / T, dtype=f)
    dn0, dn1 = (dZ @ n1) * a, (dZ.t() @ n0) * a
    def l2b(g, y, n):
        s = (g * y).sum(1, keepdim=True)
        return (g - y * s) * (1.0 / n)
    df0, df1 = l2b(dn0, n0, nf0), l2b(dn1, n1, nf1)
    dz0, dz1 = l2b(df0, f0, nz0), l2b(df1, f1, nz1)
    return {"projector_imaging.bias": dz0.sum(0), "projector_imaging.weight": dz0.t() @ x0,
            "projector_tabular.bias": dz1.sum(0), "projector_tabular.weight": dz1.t() @ x1}


for tag, kw in (("device formulas, fp32", {}), ("... lse split: exp((z-m) - log s)", dict(lse_form="split")), ("... exp(z-m)/s", dict(lse_form="div")),
                ("device formulas, float64 (sanity)", dict(dtype=torch.float64))):
    e = emulate(**kw)
    print(f"{tag:46s}", {k.split('.')[0][-7:] + '.' + k.split('.')[1]: f"{rel(e[k], h64[k]):.2e}" for k in KEYS})
