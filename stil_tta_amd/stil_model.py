"""STiLModel: drop-in for the reference's LightningModule (models/Disentangle/STiLModel.py:29-588).

Same constructor (`STiLModel(hparams)`), same hooks (`training_step(batch, batch_idx) -> loss`,
`training_epoch_end`, `validation_step`, `test_step`, `configure_optimizers`), same state_dict keys,
same `batch` layout.  pytorch-lightning is optional: when it is importable the class
derives from pl.LightningModule, otherwise from nn.Module with a minimal `log`/`current_epoch` shim and
`stil_tta_amd.driver` runs the zero_grad -> training_step -> backward -> step loop.

All arithmetic runs in the HIP kernels of libstil_hip.so (ops.py); there is no CPU fallback.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.distributed as dist

from . import ops
from .metrics import AUROC, Accuracy
from .flat import FlatState, StilAdam
from .modules import DisCoAttentionBackbone, TeacherPipe, fuse_mi_masks, random_mi_masks, set_teacher_pipe
from .ops import _p, _stream
from ._lib import lib

try:  # optional
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    _HAVE_PL = True
except Exception:  # pragma: no cover - pl is absent in the build image
    _Base = nn.Module
    _HAVE_PL = False

_DEFAULTS = dict(
    model="resnet50", embedding_dim=2048, img_size=128, num_classes=286, target="dvm",
    tabular_embedding_dim=512, tabular_transformer_num_layers=4, embedding_dropout=0.0, drop_rate=0.0,
    multimodal_embedding_dim=512, multimodal_transformer_num_layers=1, projection_dim=128, temperature=0.1,
    lambda_0=0.5, alpha=0.2, beta=3.0, gamma=0.5, rate_pt=1.0, rate_uce=0.2, th1=0.9, th2=0.95, th_contrast=0.8,
    start_epoch=35, rate_pseudo=0.9, use_ema=True, eman=True, ema_momentum=0.996, DA=False, repeat_ratio=1.0,
    batch_size=512, lr_eval=1e-4, weight_decay_eval=0.0, scheduler="anneal", warmup_epochs=10, max_epochs=500,
    checkpoint=None, pretrained_model="TIP", finetune_strategy="trainable", pretrain=False, logdir=None,
    mi_dropout=True, seed=2022, train_metrics=True,
    lr=3e-4, cosine_anneal_mult=1, dataset_length=1, check_val_every_n_epoch=1,  # only read by scheduler: cosine / linear
    global_contrast=False,  # data parallel: ITC / CLUB over the global batch (all-gather of embeddings)
    tabular_encoder="transformer",  # "saint": the STiLModel_SAINT.py variant (also selected by algorithm_name == "STiL_SAINT")
)


def _as_namespace(hp) -> SimpleNamespace:
    d = dict(_DEFAULTS)
    if isinstance(hp, dict):
        d.update(hp)
    else:
        try:
            d.update({k: hp[k] for k in hp.keys()})  # DictConfig / AttributeDict
        except Exception:
            d.update(vars(hp))
    if d.get("repeat_ratio") in (None, 0):
        d["repeat_ratio"] = 1.0
    return SimpleNamespace(**d)


def load_tip_weights(hp, pairs):
    """`load_weights` of the reference's backbones (STiLModel_backbone.py:108-115, Multimodal_model.py:96-112,
    multimodal_backbone.py:97-116, comatch_model.py:102-110): every (module, prefix) pair takes the `prefix*` entries of the
    TIP checkpoint `hp.checkpoint` (projection heads / prototypes skipped, strict); finetune_strategy 'frozen' stops their
    gradients."""
    ck = torch.load(hp.checkpoint, map_location="cpu", weights_only=False)
    sd = ck["state_dict"]
    if hp.pretrained_model != "TIP":
        raise ValueError(f"Unknown pretrain model: {hp.pretrained_model}")  # STiLModel_backbone.py:89-90
    if hp.finetune_strategy not in ("frozen", "trainable"):
        raise ValueError(f"Unknown finetune strategy {hp.finetune_strategy}")
    for mod, prefix in pairs:
        sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix) and "projection_head" not in k and "prototypes" not in k}
        mod.load_state_dict(sub, strict=True)
        if hp.finetune_strategy == "frozen":
            for p in mod.parameters():
                p.requires_grad = False


class SimCLRProjectionHead(nn.Module):
    """lightly==1.2.22 SimCLRProjectionHead restated: Linear -> ReLU -> Linear under `.layers` (unpinned)."""

    def __init__(self, i, h, o):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(i, h), nn.ReLU(), nn.Linear(h, o))

    def run(self, x):
        return ops.linear(ops.linear(x, self.layers[0].weight, self.layers[0].bias, act=1), self.layers[2].weight, self.layers[2].bias)


class _LinearHead(nn.Linear):
    def run(self, x):
        return ops.linear(x, self.weight, self.bias)


class CLUBMean(nn.Module):  # models/Disentangle/utils/club.py:88-130
    def __init__(self, x_dim, y_dim, hidden_size=512):
        super().__init__()
        self.p_mu = nn.Sequential(nn.Linear(x_dim, hidden_size), nn.ReLU(), nn.Linear(hidden_size, y_dim))

    def both(self, x, y, global_stats=False):
        """-> (forward(x, y), learning_loss(x, y)) sharing one p_mu evaluation."""
        mu = ops.linear(ops.linear(x, self.p_mu[0].weight, self.p_mu[0].bias, act=1), self.p_mu[2].weight, self.p_mu[2].bias)
        return ops.ClubFn.apply(mu, y.contiguous(), global_stats)


class STiLModel(_Base):
    def __init__(self, hparams):
        super().__init__()
        hp = _as_namespace(hparams)
        if _HAVE_PL:
            self.save_hyperparameters(vars(hp))
        else:
            self.hparams = hp
        self._epoch = 0                                  # private epoch / log store: used whenever no Lightning trainer is attached
        self.logged: Dict[str, torch.Tensor] = {}
        self.hp = hp
        fl = getattr(hp, "field_lengths", None)
        if fl is None:
            fl = torch.load(hp.field_lengths_tabular)  # STiLModel_backbone.py:97
        self.field_lengths = [int(v) for v in fl]
        C, Dp = hp.multimodal_embedding_dim, hp.projection_dim
        if str(getattr(hp, "algorithm_name", "")).upper().endswith("SAINT"):
            hp.tabular_encoder = "saint"
        if hp.tabular_encoder == "saint":
            from .saint import SaintBackbone
            self._backbone_cls = SaintBackbone
        else:
            self._backbone_cls = DisCoAttentionBackbone
        self.model = self._backbone_cls(hp, self.field_lengths)
        self.projector_multimodal = SimCLRProjectionHead(C * 3, C * 3, Dp)
        if hp.target == "dvm":  # STiLModel.py:57-63
            self.projector_imaging = _LinearHead(C, Dp)
            self.projector_tabular = _LinearHead(C, Dp)
        else:
            self.projector_imaging = SimCLRProjectionHead(C, C, Dp)
            self.projector_tabular = SimCLRProjectionHead(C, C, Dp)
        self.CLUB_imaging = CLUBMean(C, C)
        self.CLUB_tabular = CLUBMean(C, C)
        self.use_ema = bool(hp.use_ema)
        if self.use_ema:
            self.ema = self._backbone_cls(hp, self.field_lengths)
            self.ema.load_state_dict(self.model.state_dict())
            for p in self.ema.parameters():
                p.requires_grad = False
        self.register_buffer("prototypes", torch.zeros(hp.num_classes, Dp))
        self.register_buffer("prototypes_sum", torch.zeros(hp.num_classes, Dp))
        self.register_buffer("prototypes_count_sum", torch.zeros(hp.num_classes, 1))
        if hp.DA:
            self.DA_len = 256
            self.register_buffer("DA_queue", torch.zeros(self.DA_len, hp.num_classes))
            self.register_buffer("DA_ptr", torch.zeros(1, dtype=torch.long))
        self.initialize_metrics(hp.batch_size, hp.batch_size)  # STiLModel.py:67,79: nclasses = hparams.batch_size
        self.best_val_score = 0
        self.flat: Optional[FlatState] = None
        self._rng_offset = 0
        self.last: Dict[str, torch.Tensor] = {}
        if hp.checkpoint:
            self._load_tip_checkpoint(hp)
        if hp.tabular_encoder == "saint" and getattr(hp, "checkpoint_SAINT", None):  # STiLModel_SAINT_backbone.py:144-146
            self.model.encoder_tabular.load_state_dict(torch.load(hp.checkpoint_SAINT, map_location="cpu", weights_only=False))
            if self.use_ema:
                self.ema.load_state_dict(self.model.state_dict())

    # ------------------------------------------------------------------ plumbing
    # The epoch counter / `log` store below work with and without pytorch-lightning as the base class: the repo's own
    # driver (driver.py, fit.py, bench.py) sets `current_epoch` and reads `logged`; under a Lightning trainer the
    # trainer's epoch wins and `log` is forwarded to Lightning as well.
    def _attached_trainer(self):
        if not _HAVE_PL:
            return None
        try:
            return self.trainer
        except Exception:  # newer Lightning raises when no trainer is attached
            return None

    @property
    def current_epoch(self):
        tr = self._attached_trainer()
        return int(tr.current_epoch) if tr is not None else self._epoch

    @current_epoch.setter
    def current_epoch(self, v):
        self._epoch = int(v)

    def log(self, name, value, **kw):
        self.logged[name] = value
        if self._attached_trainer() is not None:
            super().log(name, value, **kw)

    def print(self, *a, **k):
        if self._attached_trainer() is not None:
            return super().print(*a, **k)
        print(*a, **k)

    def initialize_metrics(self, nclasses_train, nclasses_val):
        """STiLModel.py:120-146 with the device-side metrics of metrics.py (torchmetrics call surface)."""
        K = self.hp.num_classes
        task = "binary" if K == 2 else "multiclass"
        self.top1_acc_train = Accuracy("multiclass", nclasses_train, top_k=1)
        self.top1_acc_val = Accuracy("multiclass", nclasses_val, top_k=1)
        self.top5_acc_train = Accuracy("multiclass", nclasses_train, top_k=5)
        self.top5_acc_val = Accuracy("multiclass", nclasses_val, top_k=5)
        for n in ("train", "train_unlabelled", "train_pseudo", "val", "val_imaging", "val_tabular", "test"):
            setattr(self, f"acc_{n}", Accuracy(task, K))
            setattr(self, f"auc_{n}", AUROC(task, K))

    def _metric_probs(self, logits):
        """softmax(logits) for the metrics; column 1 for binary tasks (STiLModel.py:352-357, 450-457, 526-528)."""
        p = ops.softmax_rows(logits.detach().contiguous())
        return p[:, 1].contiguous() if self.hp.num_classes == 2 else p

    def freeze(self):
        """LightningModule.freeze(): no parameter requires grad, eval mode (trainers/evaluate.py:206, trainers/test.py:85)."""
        for q in self.parameters():
            q.requires_grad = False
        return self.eval()

    def _load_tip_checkpoint(self, hp):
        """STiLModel_backbone.py:69-90,108-115: load encoder_imaging.* / encoder_tabular.* from a TIP checkpoint."""
        pairs = [(self.model.encoder_imaging, "encoder_imaging.")]
        if hp.tabular_encoder != "saint":  # the SAINT backbone takes only the image encoder from TIP (STiLModel_SAINT_backbone.py:74-76)
            pairs.append((self.model.encoder_tabular, "encoder_tabular."))
        load_tip_weights(hp, pairs)
        if self.use_ema:
            self.ema.load_state_dict(self.model.state_dict())

    def setup_device(self, device=None):
        """Move to the GPU and carve the flat parameter / gradient / Adam / EMA slabs (idempotent)."""
        if self.flat is not None:
            return self
        if lib().device_count() < 1:
            raise RuntimeError("stil_tta_amd: no HIP device visible; the training step has no CPU path")
        device = torch.device(device or "cuda")
        nn.Module.to(self, device)
        teacher = self.ema if self.use_ema else self._backbone_cls(self.hp, self.field_lengths).to(device)
        self._rng_step = torch.zeros(1, dtype=torch.int64, device=device)  # device-side step counter of the mask RNG
        self.flat = FlatState(self.model, teacher, [self.projector_imaging, self.projector_tabular, self.projector_multimodal,
                                                    self.CLUB_imaging, self.CLUB_tabular], device)
        return self

    def configure_optimizers(self):
        """STiLModel.py:557-577: Adam(lr_eval, weight_decay_eval) over model + projectors + CLUBs (EMA excluded)."""
        tr = self._attached_trainer()
        if tr is not None and int(getattr(tr, "world_size", 1) or 1) > 1:
            # gradients live in the flat slab, not in .grad: Lightning's DDP wrapper would synchronise nothing.
            raise NotImplementedError("multi-GPU training of stil_tta_amd.STiLModel goes through stil_tta_amd.fit / driver "
                                      "(one process per GPU, comm.GradExchange), not through a Lightning DDP strategy")
        self.setup_device(self.prototypes.device if self.prototypes.is_cuda else None)
        opt = StilAdam(self.flat, lr=self.hp.lr_eval, weight_decay=self.hp.weight_decay_eval)
        hp = self.hp
        if hp.scheduler == "anneal":  # STiLModel.py:582-583 (pl_bolts LinearWarmupCosineAnnealingLR, closed form)
            from .driver import anneal_lambda
            sched = torch.optim.lr_scheduler.LambdaLR(opt, anneal_lambda(hp.warmup_epochs, hp.max_epochs))
        elif hp.scheduler == "cosine":  # STiLModel.py:580-581
            sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=int(hp.dataset_length * hp.cosine_anneal_mult), eta_min=0, last_epoch=-1)
        elif hp.scheduler == "linear":  # STiLModel.py:584-585
            sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=int(10 / hp.check_val_every_n_epoch), min_lr=hp.lr * 0.0001)
        elif hp.scheduler in (None, "none"):  # the repo's own tests: constant learning rate
            return {"optimizer": opt}
        else:
            raise ValueError('Valid schedulers are "cosine" and "anneal"')  # STiLModel.py:587
        return {"optimizer": opt, "lr_scheduler": sched}

    def optimizer_groups(self):
        """The modules whose parameters form the reference optimizer's param_groups, in order (STiLModel.py:563-570)."""
        return [self.model, self.projector_imaging, self.projector_tabular, self.projector_multimodal, self.CLUB_imaging, self.CLUB_tabular]

    def grad_signature(self):
        """What decides which parameters receive gradients in backward (comm.GradExchange learns one plan per value)."""
        return (self.current_epoch > self.hp.start_epoch, bool(self.training))

    def project_3features(self, feat_m=None, feat_i=None, feat_t=None):  # STiLModel.py:182-192
        fm = ops.l2norm(self.projector_multimodal.run(feat_m)) if feat_m is not None else None
        fi = ops.l2norm(self.projector_imaging.run(feat_i)) if feat_i is not None else None
        ft = ops.l2norm(self.projector_tabular.run(feat_t)) if feat_t is not None else None
        return fm, fi, ft

    @torch.no_grad()
    def distribution_alignment(self, logits_u):
        """STiLModel.py:171-180 on softmax(y_hat_m_ue): running queue of batch-mean probabilities (256 deep, zero rows
        included in its mean, as in the reference), probs / queue.mean(0), rows renormalised.  The reference calls
        all_reduce unguarded (crashes single-process); here the collective runs only when a process group exists."""
        Bu, K = logits_u.shape
        dev = logits_u.device
        probs = torch.empty_like(logits_u)
        lib().row_softmax_fwd(_p(logits_u), _p(probs), Bu, K, _stream())
        mean = torch.empty((K,), dtype=torch.float32, device=dev)
        ops.colsum(probs, mean, Bu, K, scale=1.0 / Bu)
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(mean)
            mean = mean / dist.get_world_size()
        ptr = int(self.DA_ptr)  # one host sync per step, as in the reference (STiLModel.py:175); DA is off by default
        self.DA_queue[ptr].copy_(mean)
        self.DA_ptr.fill_((ptr + 1) % self.DA_len)
        qmean = torch.empty((K,), dtype=torch.float32, device=dev)
        ops.colsum(self.DA_queue, qmean, self.DA_len, K, scale=1.0 / self.DA_len)
        out = torch.empty_like(probs)
        lib().da_apply(_p(probs), _p(qmean), _p(out), Bu, K, _stream())
        return out

    def _check_identify(self, batch):
        """STiLModel.py:237-238: the 'l' part is all labelled, the 'u' part all unlabelled.  Flags already on the device are
        checked on the first step only (the reference's assert is a host sync per step)."""
        il, iu = batch["l"][4], batch["u"][4]
        if torch.is_tensor(il) and torch.is_tensor(iu) and (not il.is_cuda or not getattr(self, "_identify_checked", False)):
            assert int(il.sum()) == len(il), "batch['l'] contains unlabelled samples"
            assert int(iu.sum()) == 0, "batch['u'] contains labelled samples"
            self._identify_checked = True

    def _mi_masks(self, B, mi_masks):
        dev = self.prototypes.device
        saint = self.hp.tabular_encoder == "saint"
        if mi_masks is not None:  # injected (parity tests): oracle layout -> fused layout
            out = {li: fuse_mi_masks(m, dev) for li, m in mi_masks.items() if li != "saint"}
            if "saint" in mi_masks:
                sm = mi_masks["saint"]
                out["saint"] = {"ff_col": sm["ff_col"].to(device=dev, dtype=torch.uint8).contiguous(),
                                "ff_row": sm["ff_row"].reshape(B, -1).to(device=dev, dtype=torch.uint8).contiguous()}
            return out
        if not self.hp.mi_dropout:
            return None
        Ni = (self._img_tokens)
        Nt = len(self.field_lengths)
        C = self.hp.multimodal_embedding_dim
        out = {}
        for li in range(self.hp.multimodal_transformer_num_layers):
            out[li] = random_mi_masks(B, Ni, Nt, C, 4, 0.1, self.hp.seed, self._rng_offset, dev, self._rng_step)
            self._rng_offset += 4 * B * (1 + Ni + Nt) * (C + 4 * (1 + Ni + Nt))
        if saint:  # ff_dropout = 0.8 of the SAINT column / row feed-forwards (STiLModel_SAINT_backbone.py:120-122)
            nf, h4 = Nt + 1, 4 * 32
            out["saint"] = {"ff_col": ops.rng_mask((B, nf, h4), 0.8, self.hp.seed + 2, self._rng_offset, dev, self._rng_step),
                            "ff_row": ops.rng_mask((B, nf * h4), 0.8, self.hp.seed + 3, self._rng_offset, dev, self._rng_step)}
            self._rng_offset += B * nf * h4
        return out

    # ------------------------------------------------------------------ the hot path
    def training_step(self, batch, _=None, mask_random: Optional[torch.Tensor] = None, mi_masks=None):
        """STiLModel.training_step (STiLModel.py:228-386).  `mask_random` / `mi_masks` optionally inject the
        step's randomness (parity tests); otherwise it is drawn on the device."""
        hp = self.hp
        self.setup_device()
        dev = self.prototypes.device
        current_epoch = self.current_epoch
        im_l, tab_l, y_l = batch["l"][0][1], batch["l"][1][1], batch["l"][2]
        im_u, tab_u, y_u = batch["u"][0][1], batch["u"][1][1], batch["u"][2]
        B_l, B_u = len(y_l), len(y_u)
        B = B_l + B_u
        self._check_identify(batch)
        x_img = torch.cat((im_l, im_u)).to(dev, torch.float32).contiguous()
        x_tab = torch.cat((tab_l, tab_u)).to(dev, torch.float32).contiguous()
        y_l = y_l.to(dev)
        th, tw = x_img.shape[-2], x_img.shape[-1]
        for _ in range(5):  # five stride-2 stages of the ResNet trunk (conv1, maxpool, layer2-4), each ceil(s / 2)
            th, tw = (th + 1) // 2, (tw + 1) // 2
        self._img_tokens = th * tw
        K, T, th = hp.num_classes, float(hp.temperature), float(hp.th1)
        use_pseudo = current_epoch > hp.start_epoch
        cache = {}
        self._rng_offset = 0  # call-site offsets inside this step; the step itself is counted on the device (_rng_step)

        masks = self._mi_masks(B, mi_masks)
        # The EMA teacher runs beside the student on the side stream, one BN layer behind it: parameters are averaged up
        # front (the forward pass does not change them); each teacher BN waits for the student's statistics update of
        # the same layer, averages its two running buffers and goes on (modules.TeacherPipe) -- same kernels, same
        # operands, same results as "student, then momentum_update_ema, then teacher" (STiLModel.py:248-257).
        side = ops.side_stream(dev) if self.use_ema else None
        pipe = None
        if side is not None:
            with torch.no_grad():
                self.flat.ema_update_params(hp.ema_momentum)
            # every GEMM operand layout of the student and of the (just averaged) teacher: two launches on the side stream, beside
            # the stem of the step
            self.flat.refresh_layouts(student=True, teacher=True, side=side)
            pipe = TeacherPipe(self.flat, hp.ema_momentum, bool(hp.eman))
            pipe.start.record()
            set_teacher_pipe(pipe)
        else:
            self.flat.refresh_layouts(student=True, teacher=False)
        try:
            if pipe is None:
                s = self.model.forward_all((x_img, x_tab), train=True, mi_masks=masks, cache=cache)
            else:
                main = torch.cuda.current_stream()
                side.wait_event(pipe.start)
                # the student's tabular encoder on the branch stream, beside the image encoder (its backward follows it there: ops._SideStream)
                branch = ops.branch_stream(dev)
                xt_s = None
                if branch is not None:
                    branch.wait_stream(main)
                    with torch.cuda.stream(branch):
                        xt_s = self.model.tabular_tokens(x_tab, True, masks)
                xi_s, xi_t = self.model.encoder_imaging.run_pair(self.ema.encoder_imaging, x_img, cache, side)
                with torch.no_grad(), torch.cuda.stream(side):  # the rest of the teacher has no hand-over: issue it first
                    t = self.ema.forward_all((x_img, x_tab), train=False, cache=cache, x_i=xi_t)
                    feat_m_e, _, _ = self.project_3features(torch.cat((t[3], t[9], t[6]), dim=1))
                if branch is not None:
                    main.wait_stream(branch)
                    xt_s.record_stream(main)
                s = self.model.forward_all((x_img, x_tab), train=True, mi_masks=masks, cache=cache, x_i=xi_s, x_t=xt_s)
                main.wait_stream(side)
                for tt in (*t, feat_m_e):
                    tt.record_stream(main)
                self.flat.copy_counters(bool(hp.eman))
        finally:
            set_teacher_pipe(None)
        y_m, y_i, y_t, si_e, si_m, ai, st_e, st_m, at, xc = s
        feat_m, feat_i, feat_t = self.project_3features(torch.cat((si_e, xc, st_e), dim=1), ai, at)

        with torch.no_grad():
            if pipe is not None:
                ym_e, yi_e, yt_e = t[0], t[1], t[2]
            elif self.use_ema:
                self.flat.ema_update(hp.ema_momentum, bool(hp.eman))
                self.flat.refresh_layouts(student=False, teacher=True)
                t = self.ema.forward_all((x_img, x_tab), train=False, cache=cache)
                feat_m_e, _, _ = self.project_3features(torch.cat((t[3], t[9], t[6]), dim=1))
                ym_e, yi_e, yt_e = t[0], t[1], t[2]
            else:
                ym_e, yi_e, yt_e, feat_m_e = y_m.detach(), y_i.detach(), y_t.detach(), feat_m.detach()
            if mask_random is None:
                mask_random = ops.rng_mask((B_u,), 0.5, hp.seed + 1, self._rng_offset, dev, self._rng_step)
                self._rng_offset += B_u
            else:
                if mask_random.numel() != B_u:
                    raise ValueError(f"mask_random has {mask_random.numel()} entries for {B_u} unlabelled samples")
                mask_random = mask_random.to(device=dev, dtype=torch.uint8).contiguous()
            prototypes = self.prototypes.clone()
            pred_in = self.distribution_alignment(ym_e[B_l:]) if hp.DA else None
            pl_, po_, pred, flags, hard_u, w3 = ops.cgpl_pgls(
                ym_e[B_l:], yi_e[B_l:], yt_e[B_l:], feat_m_e[B_l:].contiguous(), prototypes, mask_random, float(hp.rate_pseudo), T, th,
                use_pseudo, want_orig=True, pred_in=pred_in)
            # hard label / confidence of pseudo_label_all = cat(one_hot(y_l), prediction)  (STiLModel.py:321)
            hard = torch.cat((y_l.to(torch.int32), hard_u))
            conf = torch.cat((torch.ones(B_l, dtype=torch.uint8, device=dev), flags[:, 2].contiguous()))

        # ---- losses (STiLModel.py:284-345)
        ce = ops.CEHardFn.apply
        loss_ce = ce(y_m[:B_l].contiguous(), y_l) + ce(y_i[:B_l].contiguous(), y_l) + ce(y_t[:B_l].contiguous(), y_l)
        loss_m_u = ops.CESoftFn.apply(y_m[B_l:].contiguous(), pl_, w3[0])
        loss_i_u = ops.CESoftFn.apply(y_i[B_l:].contiguous(), pl_, w3[1])
        loss_t_u = ops.CESoftFn.apply(y_t[B_l:].contiguous(), pl_, w3[2])
        # global_contrast (off by default: the reference's negatives are rank-local): ITC over the all-gathered batch,
        # CLUB with batch means averaged over the ranks
        glob = bool(hp.global_contrast) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        loss_itc, itc_logits = ops.clip_loss(feat_i, feat_t, T, float(hp.lambda_0), gather=glob)
        club_i, est_i = self.CLUB_imaging.both(si_m, ai, glob)
        club_t, est_t = self.CLUB_tabular.both(st_m, at, glob)
        loss_pt = ops.ProtoLossFn.apply(feat_m, prototypes, hard, conf, T)
        loss = hp.alpha * loss_ce + hp.beta * loss_itc + hp.gamma * (club_i + est_i + club_t + est_t)
        if use_pseudo:
            loss = loss + hp.rate_pt * loss_pt + hp.rate_uce * (loss_m_u + loss_i_u + loss_t_u)

        # ---- prototype accumulation from TEACHER features (STiLModel.py:374-381)
        with torch.no_grad():
            cs = torch.empty((K, hp.projection_dim + 1), dtype=torch.float32, device=dev)
            lib().proto_accum(_p(feat_m_e), _p(hard), _p(conf), _p(cs), B, B_l, K, hp.projection_dim, float(hp.repeat_ratio), _stream())
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(cs, op=dist.ReduceOp.SUM)  # ONE fused [K, Dp+1] collective instead of the reference's two
            lib().proto_add(_p(cs), _p(self.prototypes_sum), _p(self.prototypes_count_sum), K, hp.projection_dim, _stream())

            if hp.train_metrics and not torch.cuda.is_current_stream_capturing():  # STiLModel.py:242, 359-362
                prob_m = self._metric_probs(y_m)
                y_u_dev = y_u.to(dev)
                self.acc_train(prob_m[:B_l], y_l)
                self.auc_train(prob_m[:B_l], y_l)
                self.acc_train_unlabelled(prob_m[B_l:], y_u_dev)
                self.auc_train_unlabelled(prob_m[B_l:], y_u_dev)

            ratios = torch.empty(5, dtype=torch.float32, device=dev)                               # STiLModel.py:307-311
            lib().flag_ratios(_p(flags), flags.shape[1], B_u, _p(ratios), _stream())

        lib().counter_inc(_p(self._rng_step), _stream())
        bs = B
        for j, name in enumerate(("threshold1_ratio", "case1_ratio", "case2_i_ratio", "case2_t_ratio", "case3_ratio")):
            self.log(f"multimodal.train.{name}", ratios[j], on_epoch=True, on_step=False, batch_size=bs)
        for name, v in (("CEloss", loss_ce), ("CEloss_unlabelled_m", loss_m_u), ("CEloss_unlabelled_i", loss_i_u),
                        ("CEloss_unlabelled_t", loss_t_u), ("ITCloss", loss_itc), ("CLUBloss_imaging", club_i),
                        ("CLUBloss_imaging_est", est_i), ("CLUBloss_tabular", club_t), ("CLUBloss_tabular_est", est_t),
                        ("PTloss", loss_itc), ("loss", loss)):  # "PTloss" logs loss_itc in the reference too (STiLModel.py:340)
            self.log(f"multimodal.train.{name}", v.detach(), on_epoch=True, on_step=False, batch_size=bs)
        self.last = dict(
            loss=loss, loss_ce=loss_ce, loss_itc=loss_itc, loss_club_i=club_i, loss_club_i_est=est_i, loss_club_t=club_t,
            loss_club_t_est=est_t, loss_pt=loss_pt, loss_m_u=loss_m_u, loss_i_u=loss_i_u, loss_t_u=loss_t_u,
            y_hat_m=y_m, y_hat_i=y_i, y_hat_t=y_t, x_si_enhance=si_e, x_si=si_m, x_ai=ai, x_st_enhance=st_e, x_st=st_m,
            x_at=at, x_c=xc, feat_m=feat_m, feat_i=feat_i, feat_t=feat_t, y_hat_m_e=ym_e, y_hat_i_e=yi_e, y_hat_t_e=yt_e,
            feat_m_e=feat_m_e, pseudo_label_orig=po_, pseudo_label=pl_, prediction=pred, flags=flags, w3=w3,
            mask_random=mask_random, class_sum=cs[:, :-1], class_count=cs[:, -1:], itc_logits=itc_logits)
        return loss

    def training_epoch_end(self, _=None):
        """STiLModel.py:389-421: prototypes <- sum / count (every class needs a confident sample), zero accumulators."""
        K, Dp = self.hp.num_classes, self.hp.projection_dim
        if self.hp.train_metrics and self.auc_train.preds:  # something was accumulated this epoch  # STiLModel.py:393-405
            vals = {}
            for name, met in (("eval.train.acc", self.acc_train), ("eval.train.auc", self.auc_train),
                              ("eval.train_unlabelled.acc", self.acc_train_unlabelled), ("eval.train_unlabelled.auc", self.auc_train_unlabelled)):
                vals[name] = met.compute()
                self.log(name, vals[name], on_epoch=True, on_step=False)
                met.reset()
            self.print(f"Epoch {self.current_epoch}: " + ", ".join(f"{k[5:]}: {float(v):.6f}" for k, v in vals.items()))
        bad = torch.zeros(1, dtype=torch.int32, device=self.prototypes.device)
        lib().proto_commit(_p(self.prototypes), _p(self.prototypes_sum), _p(self.prototypes_count_sum), _p(bad), K, Dp, _stream())
        assert int(bad.item()) == 0, "a class received no confident sample this epoch (STiLModel.py:412)"
        self.prototypes_sum.zero_()
        self.prototypes_count_sum.zero_()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()

    # ------------------------------------------------------------------ inference-side hooks
    @torch.no_grad()
    def validation_step(self, batch, _=None):
        """STiLModel.py:424-474: losses, ITC retrieval top-1/top-5 (full batches only), task accuracy / AUROC of the
        multimodal, imaging and tabular heads."""
        x, y = batch
        self.setup_device()
        dev = self.prototypes.device
        y = y.to(dev)
        o = self.model.forward_all((x[0].to(dev, torch.float32).contiguous(), x[1].to(dev, torch.float32).contiguous()), train=False)
        y_hat, y_i_hat, y_t_hat, si_e, si_m, ai, st_e, st_m, at, xc = o
        _, fi, ft = self.project_3features(None, ai, at)
        loss_itc, itc_logits = ops.clip_loss(fi, ft, float(self.hp.temperature), float(self.hp.lambda_0))
        if len(y) == self.hp.batch_size:  # STiLModel.py:437-438
            labels = torch.arange(len(y), device=dev)
            self.top1_acc_val(itc_logits, labels)
            self.top5_acc_val(itc_logits, labels)
        ci, ei = self.CLUB_imaging.both(si_m, ai)
        ct, et = self.CLUB_tabular.both(st_m, at)
        loss_ce = ops.CEHardFn.apply(y_hat.contiguous(), y)
        loss = self.hp.alpha * loss_ce + self.hp.beta * loss_itc + self.hp.gamma * (ci + ei + ct + et)
        for name, v in (("ITCloss", loss_itc), ("CLUBloss_imaging", ci), ("CLUBloss_imaging_est", ei), ("CLUBloss_tabular", ct),
                        ("CLUBloss_tabular_est", et), ("CEloss", loss_ce), ("loss", loss)):
            self.log(f"multimodal.val.{name}", v, on_epoch=True, on_step=False)
        for head, logits in (("val", y_hat), ("val_imaging", y_i_hat), ("val_tabular", y_t_hat)):
            pr = self._metric_probs(logits)
            getattr(self, f"acc_{head}")(pr, y)
            getattr(self, f"auc_{head}")(pr, y)
        return loss

    def validation_epoch_end(self, _=None):
        """STiLModel.py:476-515: epoch metrics, best_val_score (accuracy for DVM, AUROC otherwise), reset."""
        try:
            if self.trainer.sanity_checking:
                return
        except Exception:  # no Lightning trainer attached (the repo's own driver)
            pass
        if self.acc_val.counts is None:
            return
        vals = {}
        for name, met in (("acc", self.acc_val), ("auc", self.auc_val), ("acc_imaging", self.acc_val_imaging),
                          ("auc_imaging", self.auc_val_imaging), ("acc_tabular", self.acc_val_tabular), ("auc_tabular", self.auc_val_tabular)):
            vals[name] = met.compute()
            self.log(f"eval.val.{name}", vals[name], on_epoch=True, on_step=False)
        if self.top1_acc_val.counts is not None:
            self.log("multimodal.val.top1", self.top1_acc_val.compute(), on_epoch=True, on_step=False)
            self.log("multimodal.val.top5", self.top5_acc_val.compute(), on_epoch=True, on_step=False)
            self.top1_acc_val.reset()
            self.top5_acc_val.reset()
        self.print(f"Epoch {self.current_epoch}: " + ", ".join(f"val.{k}: {float(v):.6f}" for k, v in vals.items()))
        score = float(vals["acc"] if self.hp.target == "dvm" else vals["auc"])
        if score > self.best_val_score:
            self.print(f"Best epoch: {self.current_epoch}")
        self.best_val_score = max(self.best_val_score, score)
        for met in (self.acc_val, self.auc_val, self.acc_val_imaging, self.auc_val_imaging, self.acc_val_tabular, self.auc_val_tabular):
            met.reset()

    @torch.no_grad()
    def test_step(self, batch, _=None):
        """STiLModel.py:517-533: softmax(y_hat) (column 1 for binary tasks) into acc_test / auc_test; returns the scores."""
        x, y = batch
        self.setup_device()
        dev = self.prototypes.device
        y_hat = self.model.forward((x[0].to(dev, torch.float32).contiguous(), x[1].to(dev, torch.float32).contiguous()), train=False)[0]
        p = self._metric_probs(y_hat)
        self.acc_test(p, y.to(dev))
        self.auc_test(p, y.to(dev))
        return p

    def test_epoch_end(self, _=None):
        """STiLModel.py:535-543."""
        test_acc, test_auc = self.acc_test.compute(), self.auc_test.compute()
        self.log("test.acc", test_acc)
        self.log("test.auc", test_auc)
        return {"test.acc": test_acc, "test.auc": test_auc}
