"""TEST INFRASTRUCTURE ONLY -- CPU restatement (functional PyTorch) of the CoMatch and SimMatch baselines of the reference
(SURVEY.md 8f rank 4): `models/MatchModel/CoMatch.py` + `comatch_model.py`, `models/MatchModel/SimMatch.py` +
`simmatch_model.py`, `models/MatchModel/FreeMatchFolder/*.py`, all on `models/MatchModel/multimodal_backbone.py` (eval_datatype imaging_and_tabular) or on the
image-only `ResNet` wrapper of the two model files (eval_datatype imaging).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import anything under oracle/.

Pinned against the real reference by oracle/make_golden_match.py.  The state is a flat dict whose keys and ORDER equal the
reference module's state_dict():
  CoMatch : model.queue_s [Dp,Q], model.queue_ptr_s [1], model.probs_u [K,Q], model.queue_w [Dp,Q], model.queue_ptr_w [1],
            model.probs_xu [K,Q], model.encoder.*, model.m_encoder.*
  SimMatch: model.bank [Dp,N_l], model.labels [N_l], (model.DA_queue [256,K], model.DA_ptr [1] when DA), model.main.*, model.ema.*
  FreeMatch: model.main.*, model.ema.*   (p_model / label_hist / time_p are plain module attributes: `aux`)
  encoder (multimodal): encoder_imaging.*, encoder_tabular.*, image_proj.*, head.0.*, head.2.*, classifier_multimodal.*
  encoder (imaging)   : backbone.*, classifier.*, head.0.*, head.2.*
CoMatch's distribution-alignment history (`hist_prob`, a Python list on the module, comatch_model.py:94) is not part of
the state_dict; it travels in `aux["hist_prob"]`.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from . import stil_oracle as O

Tensor = torch.Tensor


def default_hparams(**over):
    hp = O.default_hparams(start_epoch=0)
    hp.eval_datatype = "imaging_and_tabular"
    hp.ema_momentum = 0.996
    # CoMatch (configs/config_dvm_MultiCoMatch.yaml:140-148)
    hp.K = 64                      # queue length (2560 in the config)
    hp.co_temperature = 0.1
    hp.co_threshold = 0.9
    hp.contrast_th = 0.8
    hp.alpha = 0.9
    hp.lam_c = 10.0
    hp.lam_u = 10.0
    # SimMatch (configs/config_dvm_MultiSimMatch.yaml:140-149); K = len(labelled dataset) there (trainers/evaluate.py:72)
    hp.DA = True
    hp.tt = 0.1
    hp.st = 0.1
    hp.c_smooth = 0.9
    hp.sim_threshold = 0.9
    hp.lambda_u = 10.0
    hp.lambda_in = 5.0
    hp.lambda_e = 0.001            # FreeMatch (configs/config_dvm_MultiFreeMatch.yaml:140-143; lambda_u 1.0 there)
    for k, v in over.items():
        setattr(hp, k, v)
    return hp


# ------------------------------------------------------------------------------------------------------------------ encoders
def init_encoder_state(hp, gen, p: str) -> Dict[str, Tensor]:
    sd: Dict[str, Tensor] = {}
    bb = O.init_backbone_state(hp, gen)
    K, Dp = hp.num_classes, hp.projection_dim
    if hp.eval_datatype == "imaging_and_tabular":   # multimodal_backbone.py:46-66
        C = hp.multimodal_embedding_dim
        for k, v in bb.items():
            if k.startswith("encoder_imaging.") or k.startswith("encoder_tabular."):
                sd[p + k] = v
        O._linear_init(sd, p + "image_proj", C, hp.embedding_dim, gen)
        O._linear_init(sd, p + "head.0", C, 2 * C, gen)
        O._linear_init(sd, p + "head.2", Dp, C, gen)
        O._linear_init(sd, p + "classifier_multimodal", K, 2 * C, gen)
    else:                                           # comatch_model.py:15-31 / simmatch_model.py:20-36
        E = hp.embedding_dim
        for k, v in bb.items():
            if k.startswith("encoder_imaging."):
                sd[p + "backbone." + k[len("encoder_imaging."):]] = v
        O._linear_init(sd, p + "classifier", K, E, gen)
        O._linear_init(sd, p + "head.0", E, E, gen)
        O._linear_init(sd, p + "head.2", Dp, E, gen)
    return sd


def encoder_forward(sd, p: str, x, hp, train: bool):
    """-> logits, F.normalize(embedding)   (multimodal_backbone.py:118-126 / comatch_model.py:26-31)."""
    lin = lambda t, n: F.linear(t, sd[p + n + ".weight"], sd[p + n + ".bias"])  # noqa: E731
    if hp.eval_datatype == "imaging_and_tabular":
        x_i = O.resnet_forward(sd, p + "encoder_imaging.", x[0], hp.model, train).mean(dim=(2, 3))
        x_t = O.tabular_forward(sd, p + "encoder_tabular.", x[1], hp)
        x_m = torch.cat([lin(x_i, "image_proj"), x_t[:, 0, :]], dim=1)
        emb = lin(O._relu(lin(x_m, "head.0"), p + "head.0"), "head.2")
        return lin(x_m, "classifier_multimodal"), F.normalize(emb)
    x_i = O.resnet_forward(sd, p + "backbone.", x, hp.model, train).mean(dim=(2, 3))
    emb = lin(O._relu(lin(x_i, "head.0"), p + "head.0"), "head.2")
    return lin(x_i, "classifier"), F.normalize(emb)


def _cat(parts, hp):
    if hp.eval_datatype == "imaging_and_tabular":
        return (torch.cat([q[0] for q in parts], dim=0), torch.cat([q[1] for q in parts], dim=0))
    return torch.cat(list(parts), dim=0)


def _rows(x, hp):
    return x[0].shape[0] if hp.eval_datatype == "imaging_and_tabular" else x.shape[0]


def _is_buffer(k: str) -> bool:
    return k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")


def trainable_keys(sd, student: str) -> List[str]:
    """Adam([model.parameters()]) (CoMatch.py:238-240): the momentum copy has requires_grad False, so only the student moves."""
    return [k for k, v in sd.items() if k.startswith(student) and v.is_floating_point() and not _is_buffer(k)]


# ------------------------------------------------------------------------------------------------------------------ CoMatch
def comatch_init_state(hp, seed: int = 0) -> Dict[str, Tensor]:
    gen = torch.Generator().manual_seed(seed)
    K, Dp, Q = hp.num_classes, hp.projection_dim, hp.K
    sd: Dict[str, Tensor] = {}
    sd["model.queue_s"] = F.normalize(torch.randn(Dp, Q, generator=gen), dim=0)      # comatch_model.py:80-82
    sd["model.queue_ptr_s"] = torch.zeros(1, dtype=torch.long)
    sd["model.probs_u"] = torch.zeros(K, Q)
    sd["model.queue_w"] = torch.randn(Dp, Q, generator=gen)                           # :88 (not normalised)
    sd["model.queue_ptr_w"] = torch.zeros(1, dtype=torch.long)
    sd["model.probs_xu"] = torch.zeros(K, Q)
    enc = init_encoder_state(hp, gen, "model.encoder.")
    sd.update(enc)
    for k, v in enc.items():
        sd["model.m_encoder." + k[len("model.encoder."):]] = v.clone()                # :75-77
    return sd


def _enqueue(sd, qname, pname, ptrname, z, t):
    """_dequeue_and_enqueue (comatch_model.py:114-145), single process: truncated at the end of the ring."""
    Q = sd[qname].shape[1]
    ptr = int(sd[ptrname])
    n = min(z.shape[0], Q - ptr)
    sd[qname][:, ptr:ptr + n] = z[:n].T
    sd[pname][:, ptr:ptr + n] = t[:n].T
    sd[ptrname][0] = (ptr + n) % Q


def comatch_training_step(sd, batch, hp, current_epoch: int, aux: dict) -> Dict[str, Tensor]:
    """CoMatch.training_step (CoMatch.py:76-140) + CoMatchModel.forward (comatch_model.py:213-319)."""
    x_l, y_l = batch["l"][0], batch["l"][1]
    u_w, u_s0, u_s1 = batch["u"][0][0], batch["u"][0][1], batch["u"][0][2]
    btx, btu = _rows(x_l, hp), _rows(u_w, hp)
    T = hp.co_temperature
    outputs, features = encoder_forward(sd, "model.encoder.", _cat([x_l, u_s0], hp), hp, train=True)
    outputs_x, outputs_u_s0, features_u_s0 = outputs[:btx], outputs[btx:], features[btx:]
    with torch.no_grad():
        for k in list(sd.keys()):                                                     # _update_momentum_encoder: parameters only
            if k.startswith("model.encoder.") and sd[k].is_floating_point() and not _is_buffer(k):
                km = "model.m_encoder." + k[len("model.encoder."):]
                sd[km] = sd[km] * hp.ema_momentum + sd[k].detach() * (1.0 - hp.ema_momentum)
        # the momentum encoder is never put in eval mode: BatchNorm uses (and tracks) the statistics of ITS batch
        outputs_m, features_m = encoder_forward(sd, "model.m_encoder.", _cat([x_l, u_w, u_s1], hp), hp, train=True)
        outputs_u_w = outputs_m[btx:btx + btu]
        feature_u_w, feature_xu_w, features_u_s1 = features_m[btx:btx + btu], features_m[:btx + btu], features_m[btx + btu:]
        probs = torch.softmax(outputs_u_w, dim=1)
        hist = aux.setdefault("hist_prob", [])                                         # distribution alignment (:262-276)
        hist.append(probs.mean(0))
        if len(hist) > 128:
            hist.pop(0)
        probs = probs / torch.stack(hist, dim=0).mean(0)
        probs = probs / probs.sum(dim=1, keepdim=True)
        probs_orig = probs.clone()
        if current_epoch > hp.start_epoch:                                             # memory-smoothed refinement (:279-284)
            A = torch.exp(torch.mm(feature_u_w, sd["model.queue_w"]) / T)
            A = A / A.sum(1, keepdim=True)
            probs = hp.alpha * probs + (1 - hp.alpha) * torch.mm(A, sd["model.probs_xu"].t())
        Q_self = torch.mm(probs, probs.t())
        Q_self.fill_diagonal_(1)
        Q = torch.cat([Q_self, torch.mm(probs, sd["model.probs_u"])], dim=1)
    sim_self = torch.exp(torch.mm(features_u_s0, features_u_s1.t()) / T)
    sim_past = torch.exp(torch.mm(features_u_s0, sd["model.queue_s"].clone()) / T)
    sim = torch.cat([sim_self, sim_past], dim=1)
    with torch.no_grad():
        _enqueue(sd, "model.queue_s", "model.probs_u", "model.queue_ptr_s", features_u_s1, probs)
        onehot = torch.zeros(btx, hp.num_classes, dtype=probs.dtype).scatter(1, y_l.view(-1, 1), 1)
        _enqueue(sd, "model.queue_w", "model.probs_xu", "model.queue_ptr_w", feature_xu_w, torch.cat([onehot, probs_orig], dim=0))
    # ---- CoMatch.py:90-121
    loss_x = F.cross_entropy(outputs_x, y_l)
    scores, _ = torch.max(probs, dim=1)
    mask = scores.ge(hp.co_threshold).to(probs.dtype)
    loss_u = (-torch.sum(F.log_softmax(outputs_u_s0, dim=1) * probs, dim=1) * mask).mean()
    pos_mask = (Q >= hp.contrast_th)
    Q_mask = Q * pos_mask
    Q_mask = Q_mask / Q_mask.sum(1, keepdim=True)
    pos_probs = (sim * pos_mask) / sim.sum(1, keepdim=True)
    log_probs = torch.log(pos_probs + 1e-7) * pos_mask
    loss_contrast = (-(log_probs * Q_mask).sum(1)).mean()
    lam_c = min(current_epoch + 1, hp.lam_c)
    loss = loss_x if current_epoch <= hp.start_epoch else loss_x + hp.lam_u * loss_u + lam_c * loss_contrast
    return dict(loss=loss, loss_x=loss_x, loss_u=loss_u, loss_contrast=loss_contrast, outputs_x=outputs_x, outputs_u_s0=outputs_u_s0,
                features_u_s0=features_u_s0, probs=probs, probs_orig=probs_orig, mask=mask, Q=Q, sim=sim)


# ------------------------------------------------------------------------------------------------------------------ SimMatch
def simmatch_init_state(hp, seed: int = 0) -> Dict[str, Tensor]:
    gen = torch.Generator().manual_seed(seed)
    K, Dp, N = hp.num_classes, hp.projection_dim, hp.K
    sd: Dict[str, Tensor] = {}
    sd["model.bank"] = F.normalize(torch.randn(Dp, N, generator=gen), dim=0)          # simmatch_model.py:68-70
    sd["model.labels"] = torch.zeros(N, dtype=torch.long)
    if hp.DA:
        sd["model.DA_queue"] = torch.zeros(256, K)
        sd["model.DA_ptr"] = torch.zeros(1, dtype=torch.long)
    enc = init_encoder_state(hp, gen, "model.main.")
    sd.update(enc)
    for k, v in enc.items():
        sd["model.ema." + k[len("model.main."):]] = v.clone()
    return sd


def simmatch_training_step(sd, batch, hp, current_epoch: int, aux=None) -> Dict[str, Tensor]:
    """SimMatch.training_step (SimMatch.py:74-121) + SimMatchModel.forward (simmatch_model.py:246-316), start_unlabel=True."""
    x_l, y_l, index = batch["l"][0], batch["l"][1], batch["l"][2]
    u_w, u_s = batch["u"][0][0], batch["u"][0][1]
    bx, bu = _rows(x_l, hp), _rows(u_w, hp)
    bank = sd["model.bank"].clone()
    logits_q, feat_q = encoder_forward(sd, "model.main.", _cat([x_l, u_s], hp), hp, train=True)
    logits_qx, logits_qu, feat_qu = logits_q[:bx], logits_q[bx:], feat_q[bx:]
    with torch.no_grad():
        m = hp.ema_momentum                                                            # momentum_update_ema: whole state_dict (:126-134)
        for k in list(sd.keys()):
            if not k.startswith("model.main."):
                continue
            ke = "model.ema." + k[len("model.main."):]
            if k.endswith("num_batches_tracked"):
                sd[ke] = sd[k].clone()
            else:
                sd[ke] = sd[ke] * m + (1.0 - m) * sd[k].detach()
        logits_k, feat_k = encoder_forward(sd, "model.ema.", _cat([x_l, u_w], hp), hp, train=False)   # self.ema.eval()
        logits_ku, feat_kx, feat_ku = logits_k[bx:], feat_k[:bx], feat_k[bx:]
        prob_ku_orig = F.softmax(logits_ku, dim=-1)
        if hp.DA:                                                                      # distribution_alignment (:150-162)
            ptr = int(sd["model.DA_ptr"])
            sd["model.DA_queue"][ptr] = prob_ku_orig.mean(0)
            sd["model.DA_ptr"][0] = (ptr + 1) % sd["model.DA_queue"].shape[0]
            prob_ku_orig = prob_ku_orig / sd["model.DA_queue"].mean(0)
            prob_ku_orig = prob_ku_orig / prob_ku_orig.sum(dim=1, keepdim=True)
        labels = sd["model.labels"]
        teacher_prob_orig = F.softmax((feat_ku @ bank) / hp.tt, dim=1)
        factor = prob_ku_orig.gather(1, labels.expand([bu, -1]))
        teacher_prob = teacher_prob_orig * factor
        teacher_prob = teacher_prob / torch.sum(teacher_prob, dim=1, keepdim=True)
        if hp.c_smooth < 1:
            agg = torch.zeros([bu, hp.num_classes], dtype=teacher_prob_orig.dtype).scatter_add(1, labels.expand([bu, -1]), teacher_prob_orig)
            prob_ku = prob_ku_orig * hp.c_smooth + agg * (1 - hp.c_smooth)
        else:
            prob_ku = prob_ku_orig
    student_prob = F.softmax((feat_qu @ bank) / hp.st, dim=1)
    loss_in_rows = torch.sum(-teacher_prob.detach() * torch.log(student_prob), dim=1)
    with torch.no_grad():                                                              # _update_bank (:137-144)
        sd["model.bank"][:, index] = feat_kx.t()
        sd["model.labels"][index] = y_l
    max_probs, _ = torch.max(prob_ku, dim=-1)
    mask = max_probs.ge(hp.sim_threshold).to(prob_ku.dtype)
    loss_x = F.cross_entropy(logits_qx, y_l, reduction="mean")
    loss_u = (torch.sum(-F.log_softmax(logits_qu, dim=1) * prob_ku.detach(), dim=1) * mask).mean()
    loss_in = loss_in_rows.mean()
    loss = loss_x if current_epoch <= hp.start_epoch else loss_x + hp.lambda_u * loss_u + hp.lambda_in * loss_in
    return dict(loss=loss, loss_x=loss_x, loss_u=loss_u, loss_in=loss_in, logits_x=logits_qx, logits_u_s=logits_qu, feat_qu=feat_qu,
                pseudo_label=prob_ku, prob_ku_orig=prob_ku_orig, teacher_prob=teacher_prob, mask=mask)


# ------------------------------------------------------------------------------------------------------------------ FreeMatch
def freematch_init_state(hp, seed: int = 0) -> Dict[str, Tensor]:
    """FreeMatchModel (freematch_model.py:39-100): main / ema encoders only; p_model, label_hist and time_p are plain
    attributes of the module (not in its state_dict) and travel in `aux`."""
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    enc = init_encoder_state(hp, gen, "model.main.")
    sd.update(enc)
    for k, v in enc.items():
        sd["model.ema." + k[len("model.main."):]] = v.clone()
    return sd


def freematch_aux(hp, dtype=torch.float32):
    K = hp.num_classes
    p = torch.ones(K, dtype=dtype) / K                                                 # freematch_model.py:50-52
    return {"p_model": p.clone(), "label_hist": p.clone(), "time_p": p.mean()}


def _inf_to_zero(v):
    v = v.clone()
    v[v == float("inf")] = 0.0
    return v


def freematch_entropy_loss(mask, logits_s, p_model, label_hist):
    """freematch_utils.py:18-47 (returns the loss only)."""
    logits_s = logits_s[mask.bool()]
    prob_s = logits_s.softmax(dim=-1)
    _, pred = torch.max(prob_s, dim=-1)
    hist_s = torch.bincount(pred, minlength=logits_s.shape[1]).to(logits_s.dtype)
    hist_s = hist_s / hist_s.sum()
    mod_p = p_model.reshape(1, -1) * _inf_to_zero(1 / label_hist.reshape(1, -1)).detach()
    mod_p = mod_p / mod_p.sum(dim=-1, keepdim=True)
    mod_mean = prob_s.mean(dim=0, keepdim=True) * _inf_to_zero(1 / hist_s).detach()
    mod_mean = mod_mean / mod_mean.sum(dim=-1, keepdim=True)
    return (mod_p * torch.log(mod_mean + 1e-12)).sum(dim=1).mean()


def freematch_training_step(sd, batch, hp, current_epoch: int, aux: dict) -> Dict[str, Tensor]:
    """FreeMatch.training_step (FreeMatch.py:76-123) + FreeMatchModel.forward (freematch_model.py:170-206)."""
    x_l, y_l = batch["l"][0], batch["l"][1]
    u_w, u_s = batch["u"][0][0], batch["u"][0][1]
    bx, bu = _rows(x_l, hp), _rows(u_w, hp)
    mm = 0.999                                                                         # self.m (freematch_model.py:48)
    logits_q, _ = encoder_forward(sd, "model.main.", _cat([x_l, u_s], hp), hp, train=True)
    logits_x_lb, logits_x_ulb_s = logits_q[:bx], logits_q[bx:]
    with torch.no_grad():
        m = hp.ema_momentum
        for k in list(sd.keys()):                                                      # momentum_update_ema (:113-121)
            if not k.startswith("model.main."):
                continue
            ke = "model.ema." + k[len("model.main."):]
            sd[ke] = sd[k].clone() if k.endswith("num_batches_tracked") else sd[ke] * m + (1.0 - m) * sd[k].detach()
        logits_w, _ = encoder_forward(sd, "model.ema.", u_w, hp, train=False)          # weak unlabelled views only
        probs = torch.softmax(logits_w, dim=-1)
        max_probs, max_idx = probs.max(dim=-1)                                         # update (:132-147)
        aux["time_p"] = aux["time_p"] * mm + (1 - mm) * max_probs.mean()
        aux["p_model"] = aux["p_model"] * mm + (1 - mm) * probs.mean(dim=0)
        hist = torch.bincount(max_idx, minlength=hp.num_classes).to(probs.dtype)
        aux["label_hist"] = aux["label_hist"] * mm + (1 - mm) * (hist / hist.sum())
        mod = aux["p_model"] / torch.max(aux["p_model"], dim=-1)[0]                    # masking (:165-167)
        mask = max_probs.ge(aux["time_p"] * mod[max_idx]).to(max_probs.dtype)
        pseudo_label = torch.zeros_like(logits_w)
        pseudo_label[torch.arange(bu), max_idx] = 1
    ent_loss = freematch_entropy_loss(mask, logits_x_ulb_s, aux["p_model"], aux["label_hist"]) if mask.sum() > 0 else logits_q.new_zeros(())
    sup_loss = F.cross_entropy(logits_x_lb, y_l)
    unsup_loss = F.cross_entropy(logits_x_ulb_s, pseudo_label)                         # every unlabelled sample (the mask is not applied)
    loss = sup_loss if current_epoch <= hp.start_epoch else sup_loss + hp.lambda_u * unsup_loss + hp.lambda_e * ent_loss
    return dict(loss=loss, sup_loss=sup_loss, unsup_loss=unsup_loss, ent_loss=ent_loss, logits_x_lb=logits_x_lb, logits_x_ulb_s=logits_x_ulb_s,
                pseudo_label=pseudo_label, mask=mask, p_model=aux["p_model"], label_hist=aux["label_hist"], time_p=aux["time_p"])


# ------------------------------------------------------------------------------------------------------------------ drivers
def full_step(kind: str, sd, opt, step_idx, batch, hp, current_epoch, aux=None, lr=None):
    """zero_grad -> training_step -> backward -> Adam on the student's parameters."""
    step_fn, student = {"comatch": (comatch_training_step, "model.encoder."), "simmatch": (simmatch_training_step, "model.main."),
                        "freematch": (freematch_training_step, "model.main.")}[kind]
    keys = trainable_keys(sd, student)
    for k in keys:
        sd[k].requires_grad_(True)
    out = step_fn(sd, batch, hp, current_epoch, {} if aux is None else aux)
    gl = torch.autograd.grad(out["loss"], [sd[k] for k in keys], allow_unused=True)
    for k in keys:
        sd[k].requires_grad_(False)
    grads = dict(zip(keys, gl))
    O.adam_step(sd, grads, opt, step_idx, hp.lr_eval if lr is None else lr, hp.weight_decay_eval)
    out = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    out["grads"] = grads
    return out


def eval_logits(kind: str, sd, x, hp):
    """validation / test forward (CoMatch.py:164-169, SimMatch.py:143-147): the STUDENT's logits in eval mode."""
    with torch.no_grad():
        return encoder_forward(sd, "model.encoder." if kind == "comatch" else "model.main.", x, hp, train=False)[0]


def synthetic_batch(hp, B: int, seed: int, views: int):
    """{'l': (x, y, index), 'u': ((weak, strong[, strong2]), y_u)} in the layout of ImagingAndTabularDataset(return_index=True)
    / StrongWeakImagingAndTabularDataset (datasets/StrongWeakImagingAndTabularDataset.py:166-196); x = (image, table) or image."""
    g = torch.Generator().manual_seed(seed)
    P = hp.img_size
    cat, con = O.split_field_lengths(hp.field_lengths)
    B_l = max(B // 8, 1)
    B_u = B - B_l

    def sample(n, base=None):
        img = torch.rand(n, 3, P, P, generator=g) if base is None else (base[0] * 0.8 + 0.2 * torch.rand(n, 3, P, P, generator=g))
        if base is None:
            cols = [torch.randint(0, c, (n, 1), generator=g).float() for c in cat]
            cols.append(torch.randn(n, len(con), generator=g))
            tab = torch.cat(cols, dim=1)[:, O.field_order_permutation(hp.field_lengths)]
        else:  # a corrupted view: resample ~30 % of the continuous columns
            tab = base[1].clone()
            con_pos = [i for i, c in enumerate(hp.field_lengths) if int(c) == 1]
            flip = torch.rand(n, len(con_pos), generator=g) < 0.3
            tab[:, con_pos] = torch.where(flip, torch.randn(n, len(con_pos), generator=g), tab[:, con_pos])
        return (img, tab)

    x_l = sample(B_l)
    base = sample(B_u)
    views_u = [base] + [sample(B_u, base) for _ in range(views - 1)]
    y_l = torch.randint(0, hp.num_classes, (B_l,), generator=g)
    y_u = torch.randint(0, hp.num_classes, (B_u,), generator=g)
    index = torch.randperm(hp.K, generator=g)[:B_l]
    if hp.eval_datatype != "imaging_and_tabular":
        x_l, views_u = x_l[0], [v[0] for v in views_u]
    return {"l": (x_l, y_l, index), "u": (views_u, y_u)}
