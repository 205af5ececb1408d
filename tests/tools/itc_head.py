"""The ITC head of the step in isolation (projector_imaging / projector_tabular -> F.normalize -> CLIP loss, STiLModel.py:182-192,
utils/clip_loss.py:27-39): its parameter and input gradients as a function of (x_ai, x_at) alone, in any dtype, plus the
decomposition of a [B, D] error matrix into its batch-coherent (batch-mean) and per-row parts.  Shared by tests/tools/itc_noise.py
(device) and tests/tools/itc_conditioning.py (CPU only).  Measurement helper, not product code."""
import torch
import torch.nn.functional as F


def head_grads(sd, x_ai, x_at, T, lam0, dtype):
    """Gradients of CLIP(normalize(P_i x_ai), normalize(P_t x_at)) w.r.t. the two projectors' parameters, their outputs z and
    the inputs, evaluated in `dtype` on the CPU (ATen).  sd: state dict with projector_{imaging,tabular}.{weight,bias}."""
    p = {k: sd[k].detach().to(dtype).clone().requires_grad_(True) for k in
         ("projector_imaging.weight", "projector_imaging.bias", "projector_tabular.weight", "projector_tabular.bias")}
    xa = x_ai.detach().to(dtype).clone().requires_grad_(True)
    xt = x_at.detach().to(dtype).clone().requires_grad_(True)
    zi = F.linear(xa, p["projector_imaging.weight"], p["projector_imaging.bias"]); zi.retain_grad()
    zt = F.linear(xt, p["projector_tabular.weight"], p["projector_tabular.bias"]); zt.retain_grad()
    # feat = F.normalize(head(x)) (STiLModel.py:182-192), and CLIPLoss normalises its inputs once more (utils/clip_loss.py:30-31)
    f0, f1 = F.normalize(F.normalize(zi, dim=1), dim=1), F.normalize(F.normalize(zt, dim=1), dim=1)
    Z = f0 @ f1.t() / T
    lab = torch.arange(len(Z))
    loss = lam0 * F.cross_entropy(Z, lab) + (1 - lam0) * F.cross_entropy(Z.t(), lab)
    loss.backward()
    out = {k: v.grad.detach() for k, v in p.items()}
    out.update(dz_i=zi.grad.detach(), dz_t=zt.grad.detach(), dx_ai=xa.grad.detach(), dx_at=xt.grad.detach(), loss=loss.detach())
    return out


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def coherent_split(err):
    """err [B, D] -> (norm of the batch-mean part replicated over the rows, norm of the per-row remainder).  The column SUM of
    `err` (what a bias gradient sees) is B * mean: its norm is sqrt(B) * the first number."""
    err = err.double()
    mean = err.mean(0, keepdim=True)
    return float((mean.expand_as(err)).norm()), float((err - mean).norm())
