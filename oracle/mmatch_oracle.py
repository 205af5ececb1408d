"""TEST INFRASTRUCTURE ONLY -- CPU restatement (functional PyTorch, fp32) of the MMatch baseline of the reference
(SURVEY.md 8f rank 4): `models/SemiMultimodal/MMatch.py` + `models/SemiMultimodal/Multimodal_model.py`.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import anything under oracle/.

Pinned against the real reference by oracle/make_golden_mmatch.py (same stub recipe as make_golden.py); the state is a
flat dict whose keys and ORDER equal the reference MMatch.state_dict():
    embed_queue [Dp, 640], embed_queue_ptr [1], probs_queue [K, 640], DA_queue [256, K], DA_ptr [1],
    model.encoder_imaging.*, model.encoder_tabular.*, model.image_proj.*, model.multimodal_proj.*,
    model.classifier_multimodal.*, model.classifier_imaging.*, model.classifier_tabular.*
(`tabular_proj` is nn.Identity because tabular_embedding_dim == multimodal_embedding_dim, Multimodal_model.py:53).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import stil_oracle as O

Tensor = torch.Tensor
BANK = 640  # MMatch.py:52


def default_hparams(**over):
    hp = O.default_hparams(DA=True, alpha=1.0, start_epoch=0, th1=0.9)
    hp.mmatch_lambda = 5.0          # configs/config_dvm_MMatch.yaml:165
    hp.prototype_momentum = 0.9
    hp.th2 = 0.5
    hp.th_contrast = 0.8
    for k, v in over.items():
        setattr(hp, k, v)
    return hp


def init_state(hp, seed: int = 0) -> Dict[str, Tensor]:
    gen = torch.Generator().manual_seed(seed)
    K, Dp, C = hp.num_classes, hp.projection_dim, hp.multimodal_embedding_dim
    sd: Dict[str, Tensor] = {}
    sd["embed_queue"] = F.normalize(torch.randn(Dp, BANK, generator=gen), dim=0)  # MMatch.py:59-60
    sd["embed_queue_ptr"] = torch.zeros(1, dtype=torch.long)
    sd["probs_queue"] = torch.zeros(K, BANK)
    sd["DA_queue"] = torch.zeros(256, K)
    sd["DA_ptr"] = torch.zeros(1, dtype=torch.long)
    bb = O.init_backbone_state(hp, gen)
    for k, v in bb.items():
        if k.startswith("encoder_imaging.") or k.startswith("encoder_tabular."):
            sd["model." + k] = v
    O._linear_init(sd, "model.image_proj", C, hp.embedding_dim, gen)
    O._linear_init(sd, "model.multimodal_proj", Dp, 2 * C, gen)
    O._linear_init(sd, "model.classifier_multimodal", K, Dp, gen)
    O._linear_init(sd, "model.classifier_imaging", K, hp.embedding_dim, gen)
    O._linear_init(sd, "model.classifier_tabular", K, hp.tabular_embedding_dim, gen)
    return sd


def trainable_keys(sd):
    """Adam([model.parameters()]) -- MMatch.py:385-387."""
    out = []
    for k, v in sd.items():
        if not k.startswith("model.") or not v.is_floating_point():
            continue
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            continue
        out.append(k)
    return out


def backbone_forward(sd, x_img, x_tab, hp, train: bool, p: str = "model."):
    """MultimodalBackbone.forward (Multimodal_model.py:114-122): -> out_m, out_i, out_t, x_m."""
    x_i = O.resnet_forward(sd, p + "encoder_imaging.", x_img, hp.model, train).mean(dim=(2, 3))  # avgpool + flatten (resnets.py:266-267)
    x_t = O.tabular_forward(sd, p + "encoder_tabular.", x_tab, hp)
    cls = x_t[:, 0, :]
    xi_p = F.linear(x_i, sd[p + "image_proj.weight"], sd[p + "image_proj.bias"])
    x_m = F.linear(torch.cat([xi_p, cls], dim=1), sd[p + "multimodal_proj.weight"], sd[p + "multimodal_proj.bias"])
    out_m = F.linear(x_m, sd[p + "classifier_multimodal.weight"], sd[p + "classifier_multimodal.bias"])
    out_i = F.linear(x_i, sd[p + "classifier_imaging.weight"], sd[p + "classifier_imaging.bias"])
    out_t = F.linear(cls, sd[p + "classifier_tabular.weight"], sd[p + "classifier_tabular.bias"])
    return out_m, out_i, out_t, x_m


def distribution_alignment(sd, probs):
    """MMatch.py:136-148 (single process)."""
    ptr = int(sd["DA_ptr"])
    sd["DA_queue"][ptr] = probs.mean(0).detach()
    sd["DA_ptr"][0] = (ptr + 1) % sd["DA_queue"].shape[0]
    probs = probs / sd["DA_queue"].mean(0)
    probs = probs / probs.sum(dim=1, keepdim=True)
    return probs.detach()


def training_step(sd, batch, hp, current_epoch: int) -> Dict[str, Tensor]:
    """MMatch.training_step (MMatch.py:191-262); mutates BN buffers, the DA queue and the memory bank like the module."""
    im_l, tab_l, y_l = batch["l"][0][1], batch["l"][1][1], batch["l"][2]
    im_u, tab_u = batch["u"][0][1], batch["u"][1][1]
    B_l = len(y_l)
    K, T = hp.num_classes, hp.temperature
    y_m, y_i, y_t, x_m = backbone_forward(sd, torch.cat((im_l, im_u)), torch.cat((tab_l, tab_u)), hp, train=True)
    prob_m = torch.softmax(y_m.detach(), dim=1)
    feat_m = F.normalize(x_m.detach(), dim=1)
    feat_m_u = feat_m[B_l:]
    ce = F.cross_entropy
    loss_ce = ce(y_m[:B_l], y_l) + ce(y_i[:B_l], y_l) + ce(y_t[:B_l], y_l)
    pseudo_label = distribution_alignment(sd, torch.softmax(y_m[B_l:], dim=1))
    pseudo_label_orig = pseudo_label.clone()
    if current_epoch > 0:
        with torch.no_grad():
            A = torch.exp(torch.mm(feat_m_u, sd["embed_queue"]) / T)
            A = A / A.sum(dim=1, keepdim=True)
            pseudo_label = 0.9 * pseudo_label_orig + 0.1 * torch.mm(A, sd["probs_queue"].t())
    max_prob, max_idx = torch.max(pseudo_label, dim=1)
    mask1 = max_prob.ge(hp.th1)
    hard_label = torch.zeros_like(pseudo_label)
    hard_label[torch.arange(len(pseudo_label)), max_idx] = 1
    loss_i_u = (ce(y_i[B_l:], hard_label, reduction="none") * mask1).mean()
    loss_t_u = (ce(y_t[B_l:], hard_label, reduction="none") * mask1).mean()
    loss = hp.alpha * loss_ce
    if current_epoch > hp.start_epoch:
        loss = loss + hp.mmatch_lambda * (loss_i_u + loss_t_u)
    pseudo_label_all = torch.cat((F.one_hot(y_l, K).float(), pseudo_label), dim=0)
    with torch.no_grad():  # _dequeue_and_enqueue (MMatch.py:102-117)
        ptr = int(sd["embed_queue_ptr"])
        n = min(feat_m.shape[0], BANK - ptr)
        sd["embed_queue"][:, ptr:ptr + n] = feat_m[:n].T
        sd["probs_queue"][:, ptr:ptr + n] = pseudo_label_all[:n].T
        sd["embed_queue_ptr"][0] = (ptr + n) % BANK
    return dict(loss=loss, loss_ce=loss_ce, loss_i_u=loss_i_u, loss_t_u=loss_t_u, y_hat_m=y_m, y_hat_i=y_i, y_hat_t=y_t, x_m=x_m,
                prob_m=prob_m, feat_m=feat_m, pseudo_label_orig=pseudo_label_orig, pseudo_label=pseudo_label, mask1=mask1,
                hard_idx=max_idx)


def full_step(sd, opt, step_idx, batch, hp, current_epoch, lr=None):
    keys = trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    out = training_step(sd, batch, hp, current_epoch)
    gl = torch.autograd.grad(out["loss"], [sd[k] for k in keys], allow_unused=True)
    for k in keys:
        sd[k].requires_grad_(False)
    grads = dict(zip(keys, gl))
    O.adam_step(sd, grads, opt, step_idx, hp.lr_eval if lr is None else lr, hp.weight_decay_eval)
    out = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    out["grads"] = grads
    return out


def validation_step(sd, x_img, x_tab, y, hp):
    """MMatch.validation_step (MMatch.py:279-308): -> loss = alpha * CE(y_hat, y) and the three softmax score sets."""
    with torch.no_grad():
        y_m, y_i, y_t, _ = backbone_forward(sd, x_img, x_tab, hp, train=False)
        loss_ce = F.cross_entropy(y_m, y)
        return dict(loss=hp.alpha * loss_ce, loss_ce=loss_ce, probs_m=torch.softmax(y_m, 1), probs_i=torch.softmax(y_i, 1),
                    probs_t=torch.softmax(y_t, 1))


# ------------------------------------------------------------------------------------------------------------------
# CoTraining baseline ("CoTrain_Pseudo", models/SemiMultimodal/CoTraining.py): same backbone + EMA teacher; each
# unimodal head learns from the OTHER modality's confident teacher distribution (soft cross-entropy).
# State = model.* [+ ema.* when use_ema], no module-level buffers.
# ------------------------------------------------------------------------------------------------------------------
def cotrain_hparams(**over):
    hp = O.default_hparams(alpha=0.2, rate_uce=0.2, start_epoch=0, use_ema=True, eman=True, ema_momentum=0.996)
    hp.co_threshold = 0.9   # configs/config_dvm_CoTrain.yaml:160
    for k, v in over.items():
        setattr(hp, k, v)
    return hp


def cotrain_init_state(hp, seed: int = 0) -> Dict[str, Tensor]:
    full = init_state(hp, seed)
    sd = {k: v for k, v in full.items() if k.startswith("model.")}
    if hp.use_ema:
        for k in list(sd.keys()):
            sd["ema." + k[len("model."):]] = sd[k].clone()
    return sd


def cotrain_training_step(sd, batch, hp, current_epoch: int) -> Dict[str, Tensor]:
    """CoTraining.training_step (CoTraining.py:112-172)."""
    im_l, tab_l, y_l = batch["l"][0][1], batch["l"][1][1], batch["l"][2]
    im_u, tab_u = batch["u"][0][1], batch["u"][1][1]
    B_l = len(y_l)
    x_img, x_tab = torch.cat((im_l, im_u)), torch.cat((tab_l, tab_u))
    y_m, y_i, y_t, _ = backbone_forward(sd, x_img, x_tab, hp, train=True)
    with torch.no_grad():
        if hp.use_ema:
            O.ema_update(sd, hp.ema_momentum, hp.eman)
            ym_e, yi_e, yt_e, _ = backbone_forward(sd, x_img, x_tab, hp, train=False, p="ema.")
        else:
            ym_e, yi_e, yt_e = y_m.detach().clone(), y_i.detach().clone(), y_t.detach().clone()
    ce = F.cross_entropy
    loss_ce = ce(y_m[:B_l], y_l) + ce(y_i[:B_l], y_l) + ce(y_t[:B_l], y_l)
    pl_i = torch.softmax(yi_e[B_l:].detach(), dim=1)
    pl_t = torch.softmax(yt_e[B_l:].detach(), dim=1)
    mask_i = pl_i.max(dim=1).values.ge(hp.co_threshold)
    mask_t = pl_t.max(dim=1).values.ge(hp.co_threshold)
    loss_i_u = (ce(y_i[B_l:], pl_t, reduction="none") * mask_t).mean()
    loss_t_u = (ce(y_t[B_l:], pl_i, reduction="none") * mask_i).mean()
    loss = hp.alpha * loss_ce
    if current_epoch > hp.start_epoch:
        loss = loss + hp.rate_uce * (loss_i_u + loss_t_u)
    return dict(loss=loss, loss_ce=loss_ce, loss_i_u=loss_i_u, loss_t_u=loss_t_u, y_hat_m=y_m, y_hat_i=y_i, y_hat_t=y_t,
                y_hat_i_e=yi_e, y_hat_t_e=yt_e, pseudo_label_i=pl_i, pseudo_label_t=pl_t, mask_i=mask_i, mask_t=mask_t)


def cotrain_full_step(sd, opt, step_idx, batch, hp, current_epoch, lr=None):
    keys = trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    out = cotrain_training_step(sd, batch, hp, current_epoch)
    gl = torch.autograd.grad(out["loss"], [sd[k] for k in keys], allow_unused=True)
    for k in keys:
        sd[k].requires_grad_(False)
    grads = dict(zip(keys, gl))
    O.adam_step(sd, grads, opt, step_idx, hp.lr_eval if lr is None else lr, hp.weight_decay_eval)
    out = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    out["grads"] = grads
    return out


# ------------------------------------------------------------------------------------------------------------------
# CoTraining with SAINT's tabular encoder ("CoTrain_Pseudo_SAINT", models/SemiMultimodal/CoTraining_SAINT.py +
# Multimodal_model_SAINT.py): the SAINT row/column encoder of STiL_SAINT (embedding 32) replaces the tabular transformer,
# `tabular_proj` becomes Linear(32, C) and `classifier_tabular` Linear(32, K); state = model.cls_token, model.encoder_imaging.*,
# model.encoder_tabular.* (SAINT), projections / classifiers [+ ema.*].
# NB CoTraining_SAINT.momentum_update_ema (eman) applies `v_ema.copy_(v_ema * m + (1 - m) * v_main)` to SAINT's int64
# *_offset buffers too (only num_batches_tracked is copied, CoTraining_SAINT.py:102-105): the float32 result is truncated
# back to int64, so the TEACHER's categories_offset entries drift by one for some values (29, 39, 58, ... at m = 0.996).
# Restated literally: parity is with the reference as shipped.
# ------------------------------------------------------------------------------------------------------------------
def cotrain_saint_hparams(**over):
    hp = cotrain_hparams(tabular_encoder="saint")
    for k, v in over.items():
        setattr(hp, k, v)
    return hp


def cotrain_saint_init_state(hp, seed: int = 0) -> Dict[str, Tensor]:
    gen = torch.Generator().manual_seed(seed)
    K, Dp, C, Dt = hp.num_classes, hp.projection_dim, hp.multimodal_embedding_dim, O.SAINT_DIM
    sd: Dict[str, Tensor] = {"model.cls_token": torch.zeros(1, 1)}
    bb = O.init_backbone_state(hp, gen)
    for k, v in bb.items():
        if k.startswith("encoder_imaging."):
            sd["model." + k] = v
    for k, v in O.init_saint_state(hp, gen).items():
        sd["model.encoder_tabular." + k] = v
    O._linear_init(sd, "model.image_proj", C, hp.embedding_dim, gen)
    O._linear_init(sd, "model.tabular_proj", C, Dt, gen)
    O._linear_init(sd, "model.multimodal_proj", Dp, 2 * C, gen)
    O._linear_init(sd, "model.classifier_multimodal", K, Dp, gen)
    O._linear_init(sd, "model.classifier_imaging", K, hp.embedding_dim, gen)
    O._linear_init(sd, "model.classifier_tabular", K, Dt, gen)
    if hp.use_ema:
        for k in list(sd.keys()):
            sd["ema." + k[len("model."):]] = sd[k].clone()
    return sd


def saint_backbone_forward(sd, x_img, x_tab, hp, train: bool, p: str = "model.", masks=None):
    """MultimodalBackbone.forward of Multimodal_model_SAINT.py:187-195: -> out_m, out_i, out_t, x_m."""
    x_i = O.resnet_forward(sd, p + "encoder_imaging.", x_img, hp.model, train).mean(dim=(2, 3))
    x_t = O.saint_tabular_forward(sd, p, x_tab, hp, masks if train else None)
    cls = x_t[:, 0, :]
    lin = lambda t, n: F.linear(t, sd[p + n + ".weight"], sd[p + n + ".bias"])  # noqa: E731
    x_m = lin(torch.cat([lin(x_i, "image_proj"), lin(cls, "tabular_proj")], dim=1), "multimodal_proj")
    return lin(x_m, "classifier_multimodal"), lin(x_i, "classifier_imaging"), lin(cls, "classifier_tabular"), x_m


def cotrain_saint_ema_update(sd, m: float, eman: bool):
    """CoTraining_SAINT.momentum_update_ema (:95-109), literally (see the note above for the integer buffers)."""
    with torch.no_grad():
        for k in list(sd.keys()):
            if not k.startswith("model."):
                continue
            ke = "ema." + k[len("model."):]
            is_buf = not sd[k].is_floating_point() or k.endswith("running_mean") or k.endswith("running_var")
            if eman:
                if "num_batches_tracked" in k:
                    sd[ke].copy_(sd[k])
                else:
                    sd[ke].copy_(sd[ke] * m + (1.0 - m) * sd[k].detach())
            elif not is_buf:
                sd[ke] = sd[ke] * m + sd[k].detach() * (1.0 - m)


def cotrain_saint_training_step(sd, batch, hp, current_epoch: int, masks=None) -> Dict[str, Tensor]:
    """CoTraining.training_step of CoTraining_SAINT.py:112-172 (identical to CoTraining.py but for the backbone)."""
    im_l, tab_l, y_l = batch["l"][0][1], batch["l"][1][1], batch["l"][2]
    im_u, tab_u = batch["u"][0][1], batch["u"][1][1]
    B_l = len(y_l)
    x_img, x_tab = torch.cat((im_l, im_u)), torch.cat((tab_l, tab_u))
    y_m, y_i, y_t, _ = saint_backbone_forward(sd, x_img, x_tab, hp, train=True, masks=masks)
    with torch.no_grad():
        if hp.use_ema:
            cotrain_saint_ema_update(sd, hp.ema_momentum, hp.eman)
            ym_e, yi_e, yt_e, _ = saint_backbone_forward(sd, x_img, x_tab, hp, train=False, p="ema.")
        else:
            ym_e, yi_e, yt_e = y_m.detach().clone(), y_i.detach().clone(), y_t.detach().clone()
    ce = F.cross_entropy
    loss_ce = ce(y_m[:B_l], y_l) + ce(y_i[:B_l], y_l) + ce(y_t[:B_l], y_l)
    pl_i = torch.softmax(yi_e[B_l:].detach(), dim=1)
    pl_t = torch.softmax(yt_e[B_l:].detach(), dim=1)
    mask_i = pl_i.max(dim=1).values.ge(hp.co_threshold)
    mask_t = pl_t.max(dim=1).values.ge(hp.co_threshold)
    loss_i_u = (ce(y_i[B_l:], pl_t, reduction="none") * mask_t).mean()
    loss_t_u = (ce(y_t[B_l:], pl_i, reduction="none") * mask_i).mean()
    loss = hp.alpha * loss_ce
    if current_epoch > hp.start_epoch:
        loss = loss + hp.rate_uce * (loss_i_u + loss_t_u)
    return dict(loss=loss, loss_ce=loss_ce, loss_i_u=loss_i_u, loss_t_u=loss_t_u, y_hat_m=y_m, y_hat_i=y_i, y_hat_t=y_t,
                y_hat_i_e=yi_e, y_hat_t_e=yt_e, pseudo_label_i=pl_i, pseudo_label_t=pl_t, mask_i=mask_i, mask_t=mask_t)


def cotrain_saint_full_step(sd, opt, step_idx, batch, hp, current_epoch, masks=None, lr=None):
    keys = [k for k in trainable_keys(sd) if "offset" not in k]
    for k in keys:
        sd[k].requires_grad_(True)
    out = cotrain_saint_training_step(sd, batch, hp, current_epoch, masks)
    gl = torch.autograd.grad(out["loss"], [sd[k] for k in keys], allow_unused=True)
    for k in keys:
        sd[k].requires_grad_(False)
    grads = dict(zip(keys, gl))
    O.adam_step(sd, grads, opt, step_idx, hp.lr_eval if lr is None else lr, hp.weight_decay_eval)
    out = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    out["grads"] = grads
    return out
