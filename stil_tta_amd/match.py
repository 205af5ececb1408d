"""CoMatch, SimMatch and FreeMatch baselines on the same HIP kernels: `models/MatchModel/CoMatch.py` +
`comatch_model.py`, `models/MatchModel/SimMatch.py` + `simmatch_model.py`, `models/MatchModel/FreeMatchFolder/*.py` and their encoder
`models/MatchModel/multimodal_backbone.py` (or the image-only `ResNet` wrapper for eval_datatype == 'imaging') of the
reference: same class names, constructor, hooks and `state_dict` keys (asserted against the reference when the golden
vectors tests/golden/comatch_*.npz, simmatch_*.npz are generated).

Batch layout (trainers/evaluate.py:50-83): batch['l'] = (x, y, index), batch['u'] = ((weak, strong[, strong2]), y_u) with
x = (image, table) or image.  Both steps run the student on [labelled ; strong] and a momentum copy of it on the weak
views; CoMatch keeps two feature/probability queues and builds a pseudo-label graph for a graph-contrastive loss, SimMatch
keeps a labelled memory bank and matches the student's similarity distribution to the teacher's, FreeMatch adapts its
confidence threshold per class from running statistics of the teacher's predictions.  There is no CPU path.
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


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _gather_rows(t):
    """concat_all_gather (comatch_model.py:325-335): no gradient."""
    if _world() == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(_world())]
    dist.all_gather(parts, t.contiguous())
    return torch.cat(parts, dim=0)


class MatchBackbone(nn.Module):
    """multimodal_backbone.py:36-126 (`MultimodalBackbone`) or comatch_model.py:15-31 (`ResNet`): parameter holder; `run` is
    the HIP path.  -> logits, F.normalize(embedding)."""

    def __init__(self, hp, field_lengths):
        super().__init__()
        K, Dp, E = hp.num_classes, hp.projection_dim, hp.embedding_dim
        self.multimodal = hp.eval_datatype == "imaging_and_tabular"
        if self.multimodal:
            cat = [int(c) for c in field_lengths if int(c) != 1]
            con = [int(c) for c in field_lengths if int(c) == 1]
            C = hp.multimodal_embedding_dim
            if hp.tabular_embedding_dim != C:
                raise NotImplementedError("tabular_embedding_dim != multimodal_embedding_dim (the reference's tabular_proj branch has a typo)")
            self.encoder_imaging = ResNet(hp.model)
            self.encoder_tabular = TabularTransformerEncoder(hp, cat, con)
            self.image_proj = nn.Linear(E, C)
            self.tabular_proj = nn.Identity()
            self.head = nn.Sequential(nn.Linear(2 * C, C), nn.ReLU(inplace=True), nn.Linear(C, Dp))
            self.classifier_multimodal = nn.Linear(2 * C, K)
        elif hp.eval_datatype == "imaging":
            self.backbone = ResNet(hp.model)
            self.classifier = nn.Linear(E, K)
            self.head = nn.Sequential(nn.Linear(E, E), nn.ReLU(inplace=True), nn.Linear(E, Dp))
        else:
            raise ValueError(f"Unknown eval_datatype {hp.eval_datatype}")
        if getattr(hp, "checkpoint", None):   # TIP pre-training (multimodal_backbone.py:64-82; comatch_model.py:60-73 for the image-only encoder)
            load_tip_weights(hp, [(self.encoder_imaging, "encoder_imaging."), (self.encoder_tabular, "encoder_tabular.")] if self.multimodal
                             else [(self.backbone, "encoder_imaging.")])

    def run(self, x, train: bool):
        lin = lambda t, m, act=0: ops.linear(t, m.weight, m.bias, act)  # noqa: E731
        if self.multimodal:
            x_i = ops.tokmean(self.encoder_imaging.run(x[0], train))
            cls = self.encoder_tabular.run(x[1])[:, 0, :].contiguous()
            x_m = torch.cat([lin(x_i, self.image_proj), cls], dim=1)
            logits = lin(x_m, self.classifier_multimodal)
        else:
            x_m = ops.tokmean(self.backbone.run(x, train))
            logits = lin(x_m, self.classifier)
        emb = lin(lin(x_m, self.head[0], 1), self.head[2])
        return logits, ops.l2norm(emb)


class _MatchBase(STiLModel):
    """What CoMatch.py and SimMatch.py share: hyper-parameter plumbing, metrics, validation / test hooks."""

    STUDENT = TEACHER = ""

    def _init_common(self, hparams, defaults):
        _Base.__init__(self)
        hp = _as_namespace(hparams)
        for k, v in defaults.items():   # keys the STiL defaults do not know (configs/config_dvm_Multi{Co,Sim}Match.yaml:140-149)
            if getattr(hp, k, None) is None:
                setattr(hp, k, v)
        if _HAVE_PL:
            self.save_hyperparameters(vars(hp))
        else:
            self.hparams = hp
        self._epoch = 0
        self.logged: Dict[str, torch.Tensor] = {}
        self.hp = hp
        fl = getattr(hp, "field_lengths", None)
        if fl is None and hp.eval_datatype == "imaging_and_tabular":
            fl = torch.load(hp.field_lengths_tabular)
        self.field_lengths = [int(v) for v in (fl or [])]
        self.use_ema = True
        self.best_val_score = 0
        self.flat: Optional[FlatState] = None
        self.last: Dict[str, torch.Tensor] = {}
        return hp

    @property
    def student(self) -> MatchBackbone:
        return getattr(self.model, self.STUDENT)

    @property
    def teacher(self) -> MatchBackbone:
        return getattr(self.model, self.TEACHER)

    @property
    def prototypes(self):  # device anchor used by the inherited helpers
        return self.student.head[2].weight

    def setup_device(self, device=None):
        if self.flat is not None:
            return self
        if lib().device_count() < 1:
            raise RuntimeError("stil_tta_amd: no HIP device visible; the training step has no CPU path")
        device = torch.device(device or "cuda")
        nn.Module.to(self, device)
        self.flat = FlatState(self.student, self.teacher, [], device)
        return self

    def optimizer_groups(self):
        return [self.model]   # Adam([{'params': self.model.parameters()}]): student and (frozen) momentum copy -- CoMatch.py:238-240

    def _to_dev(self, x, dev):
        if self.student.multimodal:
            return (x[0].to(dev, torch.float32).contiguous(), x[1].to(dev, torch.float32).contiguous())
        return x.to(dev, torch.float32).contiguous()

    def _cat(self, parts, dev):
        if self.student.multimodal:
            return (torch.cat([q[0].to(dev, torch.float32) for q in parts]).contiguous(),
                    torch.cat([q[1].to(dev, torch.float32) for q in parts]).contiguous())
        return torch.cat([q.to(dev, torch.float32) for q in parts]).contiguous()

    def _rows(self, x):
        return x[0].shape[0] if self.student.multimodal else x.shape[0]

    def _confidence(self, probs, th):
        R, K = probs.shape
        onehot = torch.empty((R, K), dtype=torch.float32, device=probs.device)
        mask = torch.empty((R,), dtype=torch.float32, device=probs.device)
        idx = torch.empty((R,), dtype=torch.int32, device=probs.device)
        lib().onehot_argmax(_p(probs), R, K, float(th), _p(onehot), _p(mask), _p(idx), _stream())
        return mask

    def _train_metrics(self, logits_x, y_l, logits_u, y_u):
        if self.hp.train_metrics and not torch.cuda.is_current_stream_capturing():
            px, pu = self._metric_probs(logits_x), self._metric_probs(logits_u)
            self.acc_train(px, y_l); self.auc_train(px, y_l)
            self.acc_train_unlabelled(pu, y_u); self.auc_train_unlabelled(pu, y_u)

    def forward(self, x):
        return self.student.run(x, self.training)[0]

    def training_epoch_end(self, _=None):
        """CoMatch.py:143-158 / SimMatch.py:124-139: epoch metrics."""
        if self.hp.train_metrics and self.auc_train.preds:
            for name, met in (("eval.train.acc", self.acc_train), ("eval.train.auc", self.auc_train),
                              ("eval.train_unlabelled.acc", self.acc_train_unlabelled), ("eval.train_unlabelled.auc", self.auc_train_unlabelled)):
                self.log(name, met.compute(), on_epoch=True, on_step=False)
                met.reset()

    @torch.no_grad()
    def validation_step(self, batch, _=None):
        """CoMatch.py:161-177 / SimMatch.py:142-158: CE of the student's logits (eval mode) + acc / auc."""
        x, y = batch
        self.setup_device()
        dev = self.prototypes.device
        y = y.to(dev)
        y_hat = self.student.run(self._to_dev(x, dev), False)[0]
        loss = ops.CEHardFn.apply(y_hat.contiguous(), y)
        self.log("multimodal.val.loss", loss, on_epoch=True, on_step=False)
        pr = self._metric_probs(y_hat)
        self.acc_val(pr, y); self.auc_val(pr, y)
        return loss

    def validation_epoch_end(self, _=None):
        """CoMatch.py:180-198: eval.val.acc / eval.val.auc, reset (best_val_score is not tracked by these modules)."""
        try:
            if self.trainer.sanity_checking:
                return
        except Exception:  # no Lightning trainer attached
            pass
        if self.acc_val.counts is None:
            return
        acc, auc = self.acc_val.compute(), self.auc_val.compute()
        self.log("eval.val.acc", acc, on_epoch=True, on_step=False)
        self.log("eval.val.auc", auc, on_epoch=True, on_step=False)
        self.print(f"Epoch {self.current_epoch}: val.acc: {float(acc):.6f}, val.auc: {float(auc):.6f}")
        self.acc_val.reset(); self.auc_val.reset()

    @torch.no_grad()
    def test_step(self, batch, _=None):
        """CoMatch.py:201-213."""
        x, y = batch
        self.setup_device()
        dev = self.prototypes.device
        p = self._metric_probs(self.student.run(self._to_dev(x, dev), False)[0])
        self.acc_test(p, y.to(dev)); self.auc_test(p, y.to(dev))
        return p


# ====================================================================================================================== CoMatch
class CoMatchModel(nn.Module):
    """comatch_model.py:33-100: buffers (queues, pointers, probability banks) + encoder + momentum encoder."""

    HIST = 128  # comatch_model.py:272

    def __init__(self, hp, field_lengths):
        super().__init__()
        K, Dp, Q = hp.num_classes, hp.projection_dim, int(hp.K)
        self.encoder = MatchBackbone(hp, field_lengths)
        self.m_encoder = MatchBackbone(hp, field_lengths)
        self.m_encoder.load_state_dict(self.encoder.state_dict())
        for q in self.m_encoder.parameters():
            q.requires_grad = False
        self.register_buffer("queue_s", nn.functional.normalize(torch.randn(Dp, Q), dim=0))
        self.register_buffer("queue_ptr_s", torch.zeros(1, dtype=torch.long))
        self.register_buffer("probs_u", torch.zeros(K, Q))
        self.register_buffer("queue_w", torch.randn(Dp, Q))
        self.register_buffer("queue_ptr_w", torch.zeros(1, dtype=torch.long))
        self.register_buffer("probs_xu", torch.zeros(K, Q))
        # distribution-alignment history: a Python list on the reference module (not in its state_dict either)
        self.register_buffer("_hist", torch.zeros(self.HIST, K), persistent=False)
        self._hist_n = 0
        self._hist_pos = 0

    @property
    def hist_prob(self):
        """The reference's list view (oldest first)."""
        n, pos = self._hist_n, self._hist_pos
        order = [(pos - n + i) % self.HIST for i in range(n)]
        return [self._hist[i].clone() for i in order]

    @hist_prob.setter
    def hist_prob(self, rows):
        rows = list(rows)[-self.HIST:]
        self._hist.zero_()
        for i, r in enumerate(rows):
            self._hist[i].copy_(r)
        self._hist_n, self._hist_pos = len(rows), len(rows) % self.HIST


class CoMatch(_MatchBase):
    STUDENT, TEACHER = "encoder", "m_encoder"

    def __init__(self, hparams):  # noqa: D401 -- not STiLModel.__init__ (different backbone and buffers)
        hp = self._init_common(hparams, dict(K=2560, co_temperature=0.1, co_threshold=0.9, contrast_th=0.8, lam_c=10, lam_u=10,
                                             eval_datatype="imaging_and_tabular"))
        self.model = CoMatchModel(hp, self.field_lengths)
        self.initialize_metrics(hp.num_classes, hp.num_classes)
        self._ptr_s: Optional[int] = None  # host mirrors of the queue pointers (read once; the reference syncs every step)
        self._ptr_w: Optional[int] = None

    def load_state_dict(self, sd, strict=True):
        self._ptr_s = self._ptr_w = None
        return super().load_state_dict(sd, strict)

    def _enqueue(self, queue, bank, ptr_buf, ptr, z, t):
        """_dequeue_and_enqueue (comatch_model.py:114-145): truncated at the end of the ring."""
        z, t = _gather_rows(z), _gather_rows(t)
        Q = queue.shape[1]
        n = min(z.shape[0], Q - ptr)
        queue[:, ptr:ptr + n] = z[:n].t()
        bank[:, ptr:ptr + n] = t[:n].t()
        ptr = (ptr + n) % Q
        ptr_buf.fill_(ptr)
        return ptr

    def training_step(self, batch, _=None):
        hp, M = self.hp, self.model
        self.setup_device()
        dev = self.prototypes.device
        x_l, y_l = batch["l"][0], batch["l"][1].to(dev)
        (u_w, u_s0, u_s1), y_u = batch["u"][0], batch["u"][1].to(dev)
        btx, btu = self._rows(x_l), self._rows(u_w)
        K, Dp, Qn, T = hp.num_classes, hp.projection_dim, M.queue_s.shape[1], float(hp.co_temperature)
        epoch = self.current_epoch
        outputs, features = M.encoder.run(self._cat([x_l, u_s0], dev), True)                        # comatch_model.py:244
        outputs_x, outputs_u_s0 = outputs[:btx].contiguous(), outputs[btx:].contiguous()
        features_u_s0 = features[btx:].contiguous()
        with torch.no_grad():
            self.flat.ema_update(hp.ema_momentum, False)                                            # parameters only (:105-111)
            # the momentum encoder is never put in eval mode by the reference: its BatchNorm uses and tracks batch statistics
            outputs_m, features_m = M.m_encoder.run(self._cat([x_l, u_w, u_s1], dev), True)
            feature_u_w = features_m[btx:btx + btu].contiguous()
            feature_xu_w, features_u_s1 = features_m[:btx + btu], features_m[btx + btu:].contiguous()
            probs0 = ops.softmax_rows(outputs_m[btx:btx + btu].contiguous())
            mean = torch.empty((K,), dtype=torch.float32, device=dev)                               # distribution alignment (:262-276)
            ops.colsum(probs0, mean, btu, K, scale=1.0 / btu)
            if _world() > 1:
                dist.all_reduce(mean)
                mean = mean / _world()
            M._hist[M._hist_pos].copy_(mean)
            M._hist_pos = (M._hist_pos + 1) % M.HIST
            M._hist_n = min(M._hist_n + 1, M.HIST)
            havg = torch.empty((K,), dtype=torch.float32, device=dev)
            ops.colsum(M._hist, havg, M._hist_n, K, scale=1.0 / M._hist_n)
            probs_orig = torch.empty_like(probs0)
            lib().da_apply(_p(probs0), _p(havg), _p(probs_orig), btu, K, _stream())
            probs = probs_orig
            if epoch > hp.start_epoch:                                                              # memory-smoothed refinement (:279-284)
                A = ops.softmax_rows(ops.gemm_nt(feature_u_w, ops.transpose(M.queue_w), btu, Qn, Dp, alpha=1.0 / T))
                probs = ops.axpby(probs_orig, ops.gemm_nt(A, M.probs_xu, btu, K, Qn), hp.alpha, 1.0 - hp.alpha)
            N = btu + Qn                                                                            # pseudo-label graph (:287-297)
            Q = torch.empty((btu, N), dtype=torch.float32, device=dev)
            ops.gemm_nt(probs, probs, btu, btu, K, out=Q, ldc=N)
            torch.diagonal(Q).fill_(1.0)
            ops.gemm_nt(probs, ops.transpose(M.probs_u), btu, Qn, K, out=Q[:, btu:], ldc=N)
            keys = torch.cat([features_u_s1, ops.transpose(M.queue_s)], dim=0)                      # [f_s1 ; queue_s^T]  (:300-303)
        S = ops.MatmulNTFn.apply(features_u_s0, keys, 1.0 / T)                                      # log(sim)
        loss_contrast = ops.ContrastGraphFn.apply(S, Q, float(hp.contrast_th))                      # CoMatch.py:104-117
        with torch.no_grad():
            if self._ptr_s is None:
                self._ptr_s, self._ptr_w = int(M.queue_ptr_s), int(M.queue_ptr_w)
            self._ptr_s = self._enqueue(M.queue_s, M.probs_u, M.queue_ptr_s, self._ptr_s, features_u_s1, probs)
            probs_xu = torch.cat([torch.nn.functional.one_hot(y_l, K).to(torch.float32), probs_orig], dim=0)
            self._ptr_w = self._enqueue(M.queue_w, M.probs_xu, M.queue_ptr_w, self._ptr_w, feature_xu_w, probs_xu)
            mask = self._confidence(probs, hp.co_threshold)                                         # CoMatch.py:92-94
        loss_x = ops.CEHardFn.apply(outputs_x, y_l)
        loss_u = ops.CESoftFn.apply(outputs_u_s0, probs, mask)                                      # CoMatch.py:97-98
        lam_c = min(epoch + 1, hp.lam_c)
        loss = loss_x if epoch <= hp.start_epoch else loss_x + hp.lam_u * loss_u + lam_c * loss_contrast
        bs = btx + btu
        self.log("multimodal.train.loss", loss.detach(), on_epoch=True, on_step=False, batch_size=bs)
        ratio = torch.empty((), dtype=torch.float32, device=dev)
        lib().reduce_sum(_p(mask), btu, 1.0 / btu, _p(ratio), 0, _stream())
        self.log("multimodal.train.threshold1_ratio", ratio, on_epoch=True, on_step=False, batch_size=bs)
        with torch.no_grad():
            self._train_metrics(outputs_x, y_l, outputs_u_s0, y_u)
        self.last = dict(loss=loss, loss_x=loss_x, loss_u=loss_u, loss_contrast=loss_contrast, outputs_x=outputs_x, outputs_u_s0=outputs_u_s0,
                         features_u_s0=features_u_s0, probs=probs, probs_orig=probs_orig, mask=mask, Q=Q, sim_logits=S)
        return loss


# ====================================================================================================================== SimMatch
class SimMatchModel(nn.Module):
    """simmatch_model.py:39-107: labelled memory bank + labels (+ DA queue) + main / ema encoders."""

    def __init__(self, hp, field_lengths):
        super().__init__()
        K, Dp, N = hp.num_classes, hp.projection_dim, int(hp.K)
        self.main = MatchBackbone(hp, field_lengths)
        self.ema = MatchBackbone(hp, field_lengths)
        self.ema.load_state_dict(self.main.state_dict())
        for q in self.ema.parameters():
            q.requires_grad = False
        self.register_buffer("bank", nn.functional.normalize(torch.randn(Dp, N), dim=0))
        self.register_buffer("labels", torch.zeros(N, dtype=torch.long))
        self.DA = bool(hp.DA)
        if self.DA:
            self.DA_len = 256
            self.register_buffer("DA_queue", torch.zeros(self.DA_len, K))
            self.register_buffer("DA_ptr", torch.zeros(1, dtype=torch.long))


class SimMatch(_MatchBase):
    STUDENT, TEACHER = "main", "ema"

    def __init__(self, hparams):  # noqa: D401 -- not STiLModel.__init__
        hp = self._init_common(hparams, dict(tt=0.1, st=0.1, c_smooth=0.9, sim_threshold=0.9, lambda_u=10.0, lambda_in=5.0,
                                             eval_datatype="imaging_and_tabular"))
        if getattr(hp, "K", None) is None:
            raise ValueError("SimMatch needs hparams.K = len(labelled dataset) (trainers/evaluate.py:72)")
        self.model = SimMatchModel(hp, self.field_lengths)
        self.initialize_metrics(hp.num_classes, hp.num_classes)

    # distribution_alignment of STiLModel reads self.DA_queue / DA_ptr / DA_len: the buffers live on the inner model here
    DA_queue = property(lambda self: self.model.DA_queue)
    DA_ptr = property(lambda self: self.model.DA_ptr)
    DA_len = property(lambda self: self.model.DA_len)

    def training_step(self, batch, _=None):
        hp, M = self.hp, self.model
        self.setup_device()
        dev = self.prototypes.device
        x_l, y_l, index = batch["l"][0], batch["l"][1].to(dev), batch["l"][2].to(dev)
        (u_w, u_s), y_u = batch["u"][0], batch["u"][1].to(dev)
        bx, bu = self._rows(x_l), self._rows(u_w)
        K, Dp, N = hp.num_classes, hp.projection_dim, M.bank.shape[1]
        epoch = self.current_epoch
        bank_t = ops.transpose(M.bank)                                                              # [N, Dp]: bank.clone().detach() (:257)
        logits_q, feat_q = M.main.run(self._cat([x_l, u_s], dev), True)
        logits_qx, logits_qu, feat_qu = logits_q[:bx].contiguous(), logits_q[bx:].contiguous(), feat_q[bx:].contiguous()
        with torch.no_grad():
            self.flat.ema_update(hp.ema_momentum, True)                                             # whole state_dict (:126-134)
            logits_k, feat_k = M.ema.run(self._cat([x_l, u_w], dev), False)                         # self.ema.eval()
            feat_kx, feat_ku = feat_k[:bx], feat_k[bx:].contiguous()
            logits_ku = logits_k[bx:].contiguous()
            prob_ku_orig = self.distribution_alignment(logits_ku) if M.DA else ops.softmax_rows(logits_ku)
            tpo = ops.softmax_rows(ops.gemm_nt(feat_ku, bank_t, bu, N, Dp, alpha=1.0 / float(hp.tt)))
            teacher_prob, prob_ku = ops.simmatch_unfold(tpo, prob_ku_orig, M.labels, hp.c_smooth)  # :289-302
        student_logits = ops.MatmulNTFn.apply(feat_qu, bank_t, 1.0 / float(hp.st))
        ones = torch.ones((bu,), dtype=torch.float32, device=dev)
        loss_in = ops.CESoftFn.apply(student_logits, teacher_prob, ones)                            # mean_r sum_j -t log softmax (:304-306)
        with torch.no_grad():                                                                       # _update_bank (:137-144)
            k_all, l_all, i_all = _gather_rows(feat_kx.contiguous()), _gather_rows(y_l), _gather_rows(index)
            M.bank[:, i_all] = k_all.t()
            M.labels[i_all] = l_all
            mask = self._confidence(prob_ku, hp.sim_threshold)                                      # SimMatch.py:89-90
        loss_x = ops.CEHardFn.apply(logits_qx, y_l)
        loss_u = ops.CESoftFn.apply(logits_qu, prob_ku, mask)
        loss = loss_x if epoch <= hp.start_epoch else loss_x + hp.lambda_u * loss_u + hp.lambda_in * loss_in
        bs = bx + bu
        self.log("multimodal.train.loss", loss.detach(), on_epoch=True, on_step=False, batch_size=bs)
        ratio = torch.empty((), dtype=torch.float32, device=dev)
        lib().reduce_sum(_p(mask), bu, 1.0 / bu, _p(ratio), 0, _stream())
        self.log("multimodal.train.threshold1_ratio", ratio, on_epoch=True, on_step=False, batch_size=bs)
        with torch.no_grad():
            self._train_metrics(logits_qx, y_l, logits_qu, y_u)
        self.last = dict(loss=loss, loss_x=loss_x, loss_u=loss_u, loss_in=loss_in, logits_x=logits_qx, logits_u_s=logits_qu, feat_qu=feat_qu,
                         pseudo_label=prob_ku, prob_ku_orig=prob_ku_orig, teacher_prob=teacher_prob, mask=mask)
        return loss


# ====================================================================================================================== FreeMatch
class FreeMatchModel(nn.Module):
    """FreeMatchFolder/freematch_model.py:39-100: main / ema encoders; the self-adaptive threshold state (p_model, label_hist,
    time_p) is a set of plain attributes on the reference module -- here non-persistent buffers, absent from the state_dict too."""

    def __init__(self, hp, field_lengths):
        super().__init__()
        K = hp.num_classes
        self.main = MatchBackbone(hp, field_lengths)
        self.ema = MatchBackbone(hp, field_lengths)
        self.ema.load_state_dict(self.main.state_dict())
        for q in self.ema.parameters():
            q.requires_grad = False
        self.m = 0.999
        self.register_buffer("p_model", torch.ones(K) / K, persistent=False)
        self.register_buffer("label_hist", torch.ones(K) / K, persistent=False)
        self.register_buffer("time_p", (torch.ones(K) / K).mean().reshape(1), persistent=False)


class FreeMatch(_MatchBase):
    STUDENT, TEACHER = "main", "ema"

    def __init__(self, hparams):  # noqa: D401 -- not STiLModel.__init__
        hp = self._init_common(hparams, dict(lambda_u=1.0, lambda_e=0.001, eval_datatype="imaging_and_tabular"))
        self.model = FreeMatchModel(hp, self.field_lengths)
        self.initialize_metrics(hp.num_classes, hp.num_classes)

    def training_step(self, batch, _=None):
        hp, M = self.hp, self.model
        self.setup_device()
        dev = self.prototypes.device
        x_l, y_l = batch["l"][0], batch["l"][1].to(dev)
        (u_w, u_s), y_u = batch["u"][0], batch["u"][1].to(dev)
        bx, bu = self._rows(x_l), self._rows(u_w)
        epoch = self.current_epoch
        logits_q, _ = M.main.run(self._cat([x_l, u_s], dev), True)                                 # freematch_model.py:182-186
        logits_x_lb, logits_x_ulb_s = logits_q[:bx].contiguous(), logits_q[bx:].contiguous()
        with torch.no_grad():
            self.flat.ema_update(hp.ema_momentum, True)                                             # whole state_dict (:113-121)
            logits_w, _ = M.ema.run(self._to_dev(u_w, dev), False)                                  # weak unlabelled views only (:191)
            probs = _gather_rows(ops.softmax_rows(logits_w.contiguous()))                           # update() sees every rank's rows (:133-134)
            mask, pseudo_label, _ = ops.freematch_update(probs, M.p_model, M.label_hist, M.time_p, M.m)
            if _world() > 1:
                r0 = dist.get_rank() * bu
                mask, pseudo_label = mask[r0:r0 + bu].contiguous(), pseudo_label[r0:r0 + bu].contiguous()
            ones = torch.ones((bu,), dtype=torch.float32, device=dev)
        ent_loss = ops.FreeMatchEntropyFn.apply(logits_x_ulb_s, mask, M.p_model, M.label_hist)      # 0 when nothing passes (:199-202)
        sup_loss = ops.CEHardFn.apply(logits_x_lb, y_l)
        unsup_loss = ops.CESoftFn.apply(logits_x_ulb_s, pseudo_label, ones)                         # every sample: the mask is not applied (FreeMatch.py:92)
        loss = sup_loss if epoch <= hp.start_epoch else sup_loss + hp.lambda_u * unsup_loss + hp.lambda_e * ent_loss
        bs = bx + bu
        self.log("multimodal.train.loss", loss.detach(), on_epoch=True, on_step=False, batch_size=bs)
        ratio = torch.empty((), dtype=torch.float32, device=dev)
        lib().reduce_sum(_p(mask), bu, 1.0 / bu, _p(ratio), 0, _stream())
        self.log("multimodal.train.threshold1_ratio", ratio, on_epoch=True, on_step=False, batch_size=bs)
        with torch.no_grad():
            self._train_metrics(logits_x_lb, y_l, logits_x_ulb_s, y_u)
        self.last = dict(loss=loss, sup_loss=sup_loss, unsup_loss=unsup_loss, ent_loss=ent_loss, logits_x_lb=logits_x_lb,
                         logits_x_ulb_s=logits_x_ulb_s, pseudo_label=pseudo_label, mask=mask, p_model=M.p_model, label_hist=M.label_hist,
                         time_p=M.time_p)
        return loss
