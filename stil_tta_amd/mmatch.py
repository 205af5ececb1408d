"""MMatch / CoTraining (+ its SAINT variant) baselines on the same HIP kernels: `models/SemiMultimodal/MMatch.py` (module) and
`models/SemiMultimodal/Multimodal_model.py` (concatenation backbone) of the reference, same class names, constructor,
hooks, `state_dict` keys (187 for ResNet-18 / 5 columns, asserted against the reference when the golden vectors are generated).

training_step (MMatch.py:191-262): ResNet (global-pooled) + tabular Transformer CLS token -> three classifiers;
labelled CE x3; pseudo-labels = distribution-aligned softmax of the multimodal head, smoothed with a 640-deep memory bank
of (embedding, label distribution) pairs from epoch 1 on; hard-label CE on the imaging and tabular heads for confident
samples; the bank is a ring buffer fed with every sample of the batch.  No EMA teacher.  There is no CPU path.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from ._lib import lib
from .flat import FlatState
from .modules import ResNet, TabularTransformerEncoder
from .ops import _p, _stream
from .stil_model import _HAVE_PL, STiLModel, _as_namespace, _Base, load_tip_weights

BANK = 640  # MMatch.py:52


class MultimodalBackbone(nn.Module):
    """Multimodal_model.py:43-122 (parameter holder; `run` is the HIP path)."""

    def __init__(self, hp, field_lengths):
        super().__init__()
        cat = [int(c) for c in field_lengths if int(c) != 1]
        con = [int(c) for c in field_lengths if int(c) == 1]
        self.encoder_imaging = ResNet(hp.model)
        self.encoder_tabular = TabularTransformerEncoder(hp, cat, con)
        C, Dt = hp.multimodal_embedding_dim, hp.tabular_embedding_dim
        if Dt != C:
            raise NotImplementedError("tabular_embedding_dim != multimodal_embedding_dim (the reference's tabular_proj branch has a typo)")
        self.image_proj = nn.Linear(hp.embedding_dim, C)
        self.tabular_proj = nn.Identity()
        self.multimodal_proj = nn.Linear(2 * C, hp.projection_dim)
        self.classifier_multimodal = nn.Linear(hp.projection_dim, hp.num_classes)
        self.classifier_imaging = nn.Linear(hp.embedding_dim, hp.num_classes)
        self.classifier_tabular = nn.Linear(Dt, hp.num_classes)
        if getattr(hp, "checkpoint", None):   # TIP pre-training: both encoders (Multimodal_model.py:62-81)
            load_tip_weights(hp, [(self.encoder_imaging, "encoder_imaging."), (self.encoder_tabular, "encoder_tabular.")])

    def run(self, x, train: bool):
        """-> out_m, out_i, out_t, x_m  (Multimodal_model.py:114-122)"""
        x_i = ops.tokmean(self.encoder_imaging.run(x[0], train))            # avgpool + flatten of the last map
        cls = self.encoder_tabular.run(x[1])[:, 0, :].contiguous()
        lin = lambda t, m: ops.linear(t, m.weight, m.bias)  # noqa: E731
        x_m = lin(torch.cat([lin(x_i, self.image_proj), cls], dim=1), self.multimodal_proj)
        return lin(x_m, self.classifier_multimodal), lin(x_i, self.classifier_imaging), lin(cls, self.classifier_tabular), x_m


class MultimodalBackboneSAINT(nn.Module):
    """Multimodal_model_SAINT.py:37-195: the concatenation backbone with SAINT's row/column tabular encoder (embedding 32):
    `tabular_proj` = Linear(32, C), `classifier_tabular` = Linear(32, K)."""

    def __init__(self, hp, field_lengths):
        super().__init__()
        from .saint import SAINT, SAINT_DIM, register_saint_meta
        self.encoder_imaging = ResNet(hp.model)
        self.cat_cols = [i for i, c in enumerate(field_lengths) if int(c) != 1]
        self.con_cols = [i for i, c in enumerate(field_lengths) if int(c) == 1]
        cats = [int(field_lengths[i]) for i in self.cat_cols]
        self.encoder_tabular = SAINT(cats, len(self.con_cols), hp.num_classes)
        self.cls_token = nn.Parameter(torch.zeros(1, 1))
        C, Dt = hp.multimodal_embedding_dim, SAINT_DIM
        self.image_proj = nn.Linear(hp.embedding_dim, C)
        self.tabular_proj = nn.Linear(Dt, C) if Dt != C else nn.Identity()
        self.multimodal_proj = nn.Linear(2 * C, hp.projection_dim)
        self.classifier_multimodal = nn.Linear(hp.projection_dim, hp.num_classes)
        self.classifier_imaging = nn.Linear(hp.embedding_dim, hp.num_classes)
        self.classifier_tabular = nn.Linear(Dt, hp.num_classes)
        register_saint_meta(self, cats)
        if getattr(hp, "checkpoint_SAINT", None):   # Multimodal_model_SAINT.py:137-139
            self.encoder_tabular.load_state_dict(torch.load(hp.checkpoint_SAINT, map_location="cpu", weights_only=False))
        if getattr(hp, "checkpoint", None):         # TIP: the image encoder only (:64-83)
            load_tip_weights(hp, [(self.encoder_imaging, "encoder_imaging.")])

    def run(self, x, train: bool, masks=None):
        """-> out_m, out_i, out_t, x_m  (Multimodal_model_SAINT.py:187-195); masks: the two feed-forward dropout keep-masks."""
        from .saint import saint_forward_tabular
        x_i = ops.tokmean(self.encoder_imaging.run(x[0], train))
        cls = saint_forward_tabular(self, x[1], masks if train else None)[:, 0, :].contiguous()
        lin = lambda t, m: ops.linear(t, m.weight, m.bias)  # noqa: E731
        t_p = lin(cls, self.tabular_proj) if isinstance(self.tabular_proj, nn.Linear) else cls
        x_m = lin(torch.cat([lin(x_i, self.image_proj), t_p], dim=1), self.multimodal_proj)
        return lin(x_m, self.classifier_multimodal), lin(x_i, self.classifier_imaging), lin(cls, self.classifier_tabular), x_m


class MMatch(STiLModel):
    def __init__(self, hparams):  # noqa: D401 -- deliberately NOT STiLModel.__init__: different backbone and buffers
        _Base.__init__(self)
        hp = _as_namespace(hparams)
        hp.mmatch_lambda = float(getattr(hp, "mmatch_lambda", 5.0))
        if _HAVE_PL:
            self.save_hyperparameters(vars(hp))
        else:
            self.hparams = hp
        self._epoch = 0                                  # private epoch / log store (stil_model.STiLModel plumbing)
        self.logged: Dict[str, torch.Tensor] = {}
        self.hp = hp
        fl = getattr(hp, "field_lengths", None)
        if fl is None:
            fl = torch.load(hp.field_lengths_tabular)
        self.field_lengths = [int(v) for v in fl]
        K, Dp = hp.num_classes, hp.projection_dim
        # buffers first: the reference registers them on the LightningModule, so they lead its state_dict ... but it
        # creates `model` before them; nn.Module orders own buffers before children regardless (MMatch.py:58-67)
        self.model = MultimodalBackbone(hp, self.field_lengths)
        self.register_buffer("embed_queue", nn.functional.normalize(torch.randn(Dp, BANK), dim=0))
        self.register_buffer("embed_queue_ptr", torch.zeros(1, dtype=torch.long))
        self.register_buffer("probs_queue", torch.zeros(K, BANK))
        if not hp.DA:
            raise ValueError("MMatch calls distribution_alignment unconditionally (MMatch.py:213): set DA: True")
        self.DA_len = 256
        self.register_buffer("DA_queue", torch.zeros(self.DA_len, K))
        self.register_buffer("DA_ptr", torch.zeros(1, dtype=torch.long))
        self.use_ema = False
        self.initialize_metrics(hp.batch_size, hp.batch_size)
        self.best_val_score = 0
        self.flat: Optional[FlatState] = None
        self.last: Dict[str, torch.Tensor] = {}
        self._ptr: Optional[int] = None  # host mirror of embed_queue_ptr (read once; the reference syncs every step)

    # ------------------------------------------------------------------ plumbing
    @property
    def prototypes(self):  # device anchor used by the inherited helpers
        return self.embed_queue

    def setup_device(self, device=None):
        if self.flat is not None:
            return self
        if lib().device_count() < 1:
            raise RuntimeError("stil_tta_amd: no HIP device visible; the training step has no CPU path")
        device = torch.device(device or "cuda")
        nn.Module.to(self, device)
        mirror = MultimodalBackbone(self.hp, self.field_lengths).to(device)  # FlatState wants a teacher-shaped mirror; unused
        self.flat = FlatState(self.model, mirror, [], device)
        return self

    def load_state_dict(self, sd, strict=True):
        self._ptr = None
        return super().load_state_dict(sd, strict)

    def optimizer_groups(self):
        return [self.model]   # Adam([{'params': self.model.parameters()}]) -- MMatch.py:385-387

    def forward(self, x):
        return self.model.run(x, self.training)

    # ------------------------------------------------------------------ the step
    def training_step(self, batch, _=None):
        hp = self.hp
        self.setup_device()
        dev = self.embed_queue.device
        im_l, tab_l, y_l = batch["l"][0][1], batch["l"][1][1], batch["l"][2]
        im_u, tab_u, y_u = batch["u"][0][1], batch["u"][1][1], batch["u"][2]
        B_l, B_u = len(y_l), len(y_u)
        self._check_identify(batch)                                                              # MMatch.py:199-200
        K, T = hp.num_classes, float(hp.temperature)
        x_img = torch.cat((im_l, im_u)).to(dev, torch.float32).contiguous()
        x_tab = torch.cat((tab_l, tab_u)).to(dev, torch.float32).contiguous()
        y_l = y_l.to(dev)
        y_m, y_i, y_t, x_m = self.model.run((x_img, x_tab), True)

        ce = ops.CEHardFn.apply
        loss_ce = ce(y_m[:B_l].contiguous(), y_l) + ce(y_i[:B_l].contiguous(), y_l) + ce(y_t[:B_l].contiguous(), y_l)
        with torch.no_grad():
            feat_m = ops.l2norm(x_m.detach())
            pseudo_orig = self.distribution_alignment(y_m[B_l:].detach().contiguous())       # MMatch.py:213
            pseudo = pseudo_orig
            if self.current_epoch > 0:                                                           # MMatch.py:215-221
                bank_t = ops.transpose(self.embed_queue)                                         # [640, Dp]
                A = ops.softmax_rows(ops.gemm_nt(feat_m[B_l:].contiguous(), bank_t, B_u, BANK, hp.projection_dim, alpha=1.0 / T))
                smooth = ops.gemm_nt(A, self.probs_queue, B_u, K, BANK)                          # A @ probs_bank^T
                pseudo = ops.axpby(pseudo_orig, smooth, 0.9, 0.1)
            onehot = torch.empty((B_u, K), dtype=torch.float32, device=dev)
            mask1 = torch.empty((B_u,), dtype=torch.float32, device=dev)
            hard_idx = torch.empty((B_u,), dtype=torch.int32, device=dev)
            lib().onehot_argmax(_p(pseudo), B_u, K, float(hp.th1), _p(onehot), _p(mask1), _p(hard_idx), _stream())
        loss_i_u = ops.CESoftFn.apply(y_i[B_l:].contiguous(), onehot, mask1)
        loss_t_u = ops.CESoftFn.apply(y_t[B_l:].contiguous(), onehot, mask1)
        loss = hp.alpha * loss_ce
        if self.current_epoch > hp.start_epoch:
            loss = loss + hp.mmatch_lambda * (loss_i_u + loss_t_u)

        with torch.no_grad():
            if hp.train_metrics and not torch.cuda.is_current_stream_capturing():
                prob_m = self._metric_probs(y_m)
                y_u_dev = y_u.to(dev)
                self.acc_train(prob_m[:B_l], y_l); self.auc_train(prob_m[:B_l], y_l)
                self.acc_train_unlabelled(prob_m[B_l:], y_u_dev); self.auc_train_unlabelled(prob_m[B_l:], y_u_dev)
            # _dequeue_and_enqueue (MMatch.py:102-117): every sample of the batch, truncated at the end of the ring
            if self._ptr is None:
                self._ptr = int(self.embed_queue_ptr)
            z = feat_m
            t = torch.cat((torch.nn.functional.one_hot(y_l, K).to(torch.float32), pseudo), dim=0)   # pseudo_label_all (MMatch.py:243)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:       # concat_all_gather (MMatch.py:104-106)
                zs = [torch.empty_like(z) for _ in range(dist.get_world_size())]
                ts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
                dist.all_gather(zs, z.contiguous())
                dist.all_gather(ts, t.contiguous())
                z, t = torch.cat(zs), torch.cat(ts)
            n = min(z.shape[0], BANK - self._ptr)
            self.embed_queue[:, self._ptr:self._ptr + n] = z[:n].t()
            self.probs_queue[:, self._ptr:self._ptr + n] = t[:n].t()
            self._ptr = (self._ptr + n) % BANK
            self.embed_queue_ptr.fill_(self._ptr)

        bs = B_l + B_u
        for name, v in (("CEloss", loss_ce), ("CEloss_unlabelled_i", loss_i_u), ("CEloss_unlabelled_t", loss_t_u), ("loss", loss)):
            self.log(f"multimodal.train.{name}", v.detach(), on_epoch=True, on_step=False, batch_size=bs)
        self.last = dict(loss=loss, loss_ce=loss_ce, loss_i_u=loss_i_u, loss_t_u=loss_t_u, y_hat_m=y_m, y_hat_i=y_i, y_hat_t=y_t, x_m=x_m,
                         feat_m=feat_m, pseudo_label_orig=pseudo_orig, pseudo_label=pseudo, mask1=mask1, hard_idx=hard_idx)
        return loss

    def training_epoch_end(self, _=None):
        """MMatch.py:265-276: epoch metrics only (no prototypes to commit)."""
        if self.hp.train_metrics and self.auc_train.preds:  # something was accumulated this epoch
            for name, met in (("eval.train.acc", self.acc_train), ("eval.train.auc", self.auc_train),
                              ("eval.train_unlabelled.acc", self.acc_train_unlabelled), ("eval.train_unlabelled.auc", self.auc_train_unlabelled)):
                self.log(name, met.compute(), on_epoch=True, on_step=False)
                met.reset()

    @torch.no_grad()
    def validation_step(self, batch, _=None):
        """MMatch.py:279-308."""
        x, y = batch
        self.setup_device()
        dev = self.prototypes.device
        y = y.to(dev)
        y_hat, y_i_hat, y_t_hat, _ = self.model.run((x[0].to(dev, torch.float32).contiguous(), x[1].to(dev, torch.float32).contiguous()), False)
        loss_ce = ops.CEHardFn.apply(y_hat.contiguous(), y)
        loss = self.hp.alpha * loss_ce
        self.log("multimodal.val.CEloss", loss_ce, on_epoch=True, on_step=False)
        self.log("multimodal.val.loss", loss, on_epoch=True, on_step=False)
        for head, logits in (("val", y_hat), ("val_imaging", y_i_hat), ("val_tabular", y_t_hat)):
            pr = self._metric_probs(logits)
            getattr(self, f"acc_{head}")(pr, y)
            getattr(self, f"auc_{head}")(pr, y)
        return loss

    @torch.no_grad()
    def test_step(self, batch, _=None):
        """MMatch.py:343-355."""
        x, y = batch
        self.setup_device()
        dev = self.prototypes.device
        y_hat = self.model.run((x[0].to(dev, torch.float32).contiguous(), x[1].to(dev, torch.float32).contiguous()), False)[0]
        p = self._metric_probs(y_hat)
        self.acc_test(p, y.to(dev))
        self.auc_test(p, y.to(dev))
        return p


class CoTraining(STiLModel):
    """`CoTrain_Pseudo` baseline -- models/SemiMultimodal/CoTraining.py: the concatenation backbone plus an (optional) EMA
    teacher of it; each unimodal head is trained on the OTHER modality's teacher distribution where that teacher is
    confident (soft cross-entropy, CoTraining.py:141-149).  state_dict = model.* [+ ema.*]."""

    def __init__(self, hparams):  # noqa: D401 -- not STiLModel.__init__ (different backbone, no module buffers)
        _Base.__init__(self)
        hp = _as_namespace(hparams)
        hp.co_threshold = float(getattr(hp, "co_threshold", 0.9))
        if _HAVE_PL:
            self.save_hyperparameters(vars(hp))
        else:
            self.hparams = hp
        self._epoch = 0                                  # private epoch / log store (stil_model.STiLModel plumbing)
        self.logged: Dict[str, torch.Tensor] = {}
        self.hp = hp
        fl = getattr(hp, "field_lengths", None)
        if fl is None:
            fl = torch.load(hp.field_lengths_tabular)
        self.field_lengths = [int(v) for v in fl]
        # CoTrain_Pseudo_SAINT (CoTraining_SAINT.py, trainers/evaluate.py:163-165): the same module on the SAINT backbone
        self.saint = hp.tabular_encoder == "saint" or getattr(hp, "algorithm_name", None) == "CoTrain_Pseudo_SAINT"
        self._backbone = MultimodalBackboneSAINT if self.saint else MultimodalBackbone
        self.model = self._backbone(hp, self.field_lengths)
        self.use_ema = bool(hp.use_ema)
        if self.use_ema:  # CoTraining.py:43-51
            self.ema = self._backbone(hp, self.field_lengths)
            self.ema.load_state_dict(self.model.state_dict())
            for q in self.ema.parameters():
                q.requires_grad = False
        self.initialize_metrics(hp.num_classes, hp.num_classes)
        self.best_val_score = 0
        self.flat: Optional[FlatState] = None
        self.last: Dict[str, torch.Tensor] = {}

    @property
    def prototypes(self):  # device anchor used by the inherited helpers
        return self.model.image_proj.weight

    def setup_device(self, device=None):
        if self.flat is not None:
            return self
        if lib().device_count() < 1:
            raise RuntimeError("stil_tta_amd: no HIP device visible; the training step has no CPU path")
        device = torch.device(device or "cuda")
        nn.Module.to(self, device)
        teacher = self.ema if self.use_ema else self._backbone(self.hp, self.field_lengths).to(device)
        self.flat = FlatState(self.model, teacher, [], device)
        self._rng_offset = 0
        self._rng_step = torch.zeros(1, dtype=torch.int64, device=device)
        self._int_pairs = []
        if self.saint and self.use_ema:   # SAINT's persistent int64 *_offset buffers, teacher / student
            keys = set(self.ema.state_dict().keys())
            self._int_pairs = [(bt, bs) for (nt, bt), (_, bs) in zip(self.ema.named_buffers(), self.model.named_buffers())
                               if bt.dtype == torch.int64 and "offset" in nt and nt in keys]
        return self

    def forward(self, x):
        return self.model.run(x, self.training)

    def _saint_masks(self, B, masks):
        """Keep-masks of SAINT's two feed-forward dropouts (p = 0.8, Multimodal_model_SAINT.py:112-115): injected (parity
        tests, oracle layout) or drawn on the device."""
        dev = self.prototypes.device
        nf, h4 = len(self.field_lengths) + 1, 4 * 32
        if masks is not None:
            return {"ff_col": masks["ff_col"].to(device=dev, dtype=torch.uint8).contiguous(),
                    "ff_row": masks["ff_row"].reshape(B, -1).to(device=dev, dtype=torch.uint8).contiguous()}
        out = {"ff_col": ops.rng_mask((B, nf, h4), 0.8, self.hp.seed + 2, self._rng_offset, dev, self._rng_step),
               "ff_row": ops.rng_mask((B, nf * h4), 0.8, self.hp.seed + 3, self._rng_offset, dev, self._rng_step)}
        self._rng_offset += B * nf * h4
        return out

    def _ema_update(self):
        """CoTraining.py:95-109.  SAINT + eman: the reference's EMA also runs over the int64 *_offset buffers of the SAINT
        encoder (float32 arithmetic, truncating copy, CoTraining_SAINT.py:102-105); reproduced as shipped."""
        hp = self.hp
        if self.saint and bool(hp.eman):
            old = [bt.clone() for bt, _ in self._int_pairs]
            self.flat.ema_update(hp.ema_momentum, True)          # floats + counters (copies every int64 buffer)
            for (bt, bs), o in zip(self._int_pairs, old):
                bt.copy_(o)
                lib().ema_int_trunc(_p(bt), _p(bs), bt.numel(), float(hp.ema_momentum), _stream())
        else:
            self.flat.ema_update(hp.ema_momentum, bool(hp.eman))

    def _confidence(self, probs, th):
        """(max_k p >= th) as a 0/1 float row mask."""
        R, K = probs.shape
        dev = probs.device
        onehot = torch.empty((R, K), dtype=torch.float32, device=dev)
        mask = torch.empty((R,), dtype=torch.float32, device=dev)
        idx = torch.empty((R,), dtype=torch.int32, device=dev)
        lib().onehot_argmax(_p(probs), R, K, float(th), _p(onehot), _p(mask), _p(idx), _stream())
        return mask

    def training_step(self, batch, _=None, saint_masks=None):
        hp = self.hp
        self.setup_device()
        dev = self.prototypes.device
        im_l, tab_l, y_l = batch["l"][0][1], batch["l"][1][1], batch["l"][2]
        im_u, tab_u, y_u = batch["u"][0][1], batch["u"][1][1], batch["u"][2]
        B_l = len(y_l)
        self._check_identify(batch)                                                              # CoTraining.py:121-122
        x = (torch.cat((im_l, im_u)).to(dev, torch.float32).contiguous(), torch.cat((tab_l, tab_u)).to(dev, torch.float32).contiguous())
        y_l = y_l.to(dev)
        if self.saint:
            y_m, y_i, y_t, _x = self.model.run(x, True, self._saint_masks(x[0].shape[0], saint_masks))
        else:
            y_m, y_i, y_t, _x = self.model.run(x, True)
        with torch.no_grad():
            if self.use_ema:  # CoTraining.py:128-133
                self._ema_update()
                _, yi_e, yt_e, _ = self.ema.run(x, False)
            else:
                yi_e, yt_e = y_i.detach(), y_t.detach()
            pl_i = ops.softmax_rows(yi_e[B_l:].contiguous())
            pl_t = ops.softmax_rows(yt_e[B_l:].contiguous())
            mask_i, mask_t = self._confidence(pl_i, hp.co_threshold), self._confidence(pl_t, hp.co_threshold)
        ce = ops.CEHardFn.apply
        loss_ce = ce(y_m[:B_l].contiguous(), y_l) + ce(y_i[:B_l].contiguous(), y_l) + ce(y_t[:B_l].contiguous(), y_l)
        loss_i_u = ops.CESoftFn.apply(y_i[B_l:].contiguous(), pl_t, mask_t)
        loss_t_u = ops.CESoftFn.apply(y_t[B_l:].contiguous(), pl_i, mask_i)
        loss = hp.alpha * loss_ce
        if self.current_epoch > hp.start_epoch:
            loss = loss + hp.rate_uce * (loss_i_u + loss_t_u)
        with torch.no_grad():
            if hp.train_metrics and not torch.cuda.is_current_stream_capturing():
                prob_m = self._metric_probs(y_m)
                y_u_dev = y_u.to(dev)
                self.acc_train(prob_m[:B_l], y_l); self.auc_train(prob_m[:B_l], y_l)
                self.acc_train_unlabelled(prob_m[B_l:], y_u_dev); self.auc_train_unlabelled(prob_m[B_l:], y_u_dev)
        bs = B_l + len(y_u)
        for name, v in (("CEloss", loss_ce), ("CEloss_unlabelled_i", loss_i_u), ("CEloss_unlabelled_t", loss_t_u), ("loss", loss)):
            self.log(f"multimodal.train.{name}", v.detach(), on_epoch=True, on_step=False, batch_size=bs)
        self.last = dict(loss=loss, loss_ce=loss_ce, loss_i_u=loss_i_u, loss_t_u=loss_t_u, y_hat_m=y_m, y_hat_i=y_i, y_hat_t=y_t,
                         y_hat_i_e=yi_e, y_hat_t_e=yt_e, pseudo_label_i=pl_i, pseudo_label_t=pl_t, mask_i=mask_i, mask_t=mask_t)
        return loss

    optimizer_groups = MMatch.optimizer_groups
    training_epoch_end = MMatch.training_epoch_end
    on_train_epoch_end = MMatch.training_epoch_end   # the reference uses the newer hook name (CoTraining.py:175)
    validation_step = MMatch.validation_step
    test_step = MMatch.test_step
