"""Pin the oracle against the REAL reference and write tests/golden/*.npz.

Runs only in the build container (needs /root/reference on disk).  The reference
is imported unmodified; packages that are absent offline and carry no arithmetic
on this path (pytorch_lightning, torchmetrics, timm, omegaconf, pl_bolts) are
stubbed in sys.modules, and lightly's SimCLRProjectionHead (arithmetic, absent)
is restated as Linear-ReLU-Linear (unpinned).

For every case it
  1. builds the reference STiLModel, loads the oracle's seeded state into it,
  2. runs reference training_step + backward + Adam and the oracle's full_step on
     identical inputs (identical mask_random / dropout masks, injected by patching
     torch.rand_like / nn.Dropout.forward / drop_path in the reference process),
  3. asserts oracle == reference to 2e-5 (abs+rel),
  4. stores the REFERENCE's outputs as the golden vectors (tensors only).

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [case ...]      (default: every case + the Adam layout)
"""
from __future__ import annotations

import os
import sys
import types
import tempfile

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("STIL_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import stil_oracle as O  # noqa: E402


# ------------------------------------------------------------------ stubs
class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPathStub(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    mod("timm"); mod("timm.models")
    mod("timm.models.layers", DropPath=DropPathStub, to_2tuple=lambda x: (x, x), trunc_normal_=torch.nn.init.trunc_normal_)

    class OmegaConf:
        @staticmethod
        def create(x):
            return x

    mod("omegaconf", OmegaConf=OmegaConf, DictConfig=dict, open_dict=None)

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.current_epoch = 0
            self.logged = {}

        def save_hyperparameters(self, hp):
            self.hparams = hp

        def log(self, name, value, **kw):
            self.logged[name] = value

        def print(self, *a, **k):
            pass

    mod("pytorch_lightning", LightningModule=LightningModule)

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            return None

        def compute(self):
            return torch.tensor(0.0)

        def reset(self):
            pass

    mod("torchmetrics", Accuracy=Metric, AUROC=Metric)

    class SimCLRProjectionHead(nn.Module):  # lightly 1.2.22 restated (unpinned)
        def __init__(self, i, h, o):
            super().__init__()
            self.layers = nn.Sequential(nn.Linear(i, h), nn.ReLU(), nn.Linear(h, o))

        def forward(self, x):
            return self.layers(x)

    mod("lightly"); mod("lightly.models")
    mod("lightly.models.modules", SimCLRProjectionHead=SimCLRProjectionHead)
    mod("pl_bolts"); mod("pl_bolts.optimizers")
    mod("pl_bolts.optimizers.lr_scheduler", LinearWarmupCosineAnnealingLR=object)


def ref_hparams(hp, fl_path):
    d = dict(vars(hp))
    d.update(checkpoint=None, pretrain=False, field_lengths_tabular=fl_path, pretrained_model="TIP",
             finetune_strategy="trainable", logdir=None, sharpen_temperature=0.1, momentum=0.99)
    return _AttrDict(d)


# ------------------------------------------------------------------ mask injection
class MaskProvider:
    """Feeds a fixed, named sequence of keep-masks to the reference's Dropout / drop_path calls."""

    ORDER = ["attn_i", "attn_t", "attn_c", "proj_i", "proj_t", "proj_c", "dp1_i", "dp1_t", "dp1_c",
             "fc1_i", "fc2_i", "dp2_i", "fc1_t", "fc2_t", "dp2_t", "fc1_c", "fc2_c", "dp2_c"]

    def __init__(self, masks, saint=None):
        self.masks = dict(masks)
        self.order = list(self.ORDER)
        if saint is not None:  # the SAINT encoder's two FF dropouts are called before the MI layer's
            self.masks.update(saint)
            self.order = ["ff_col", "ff_row"] + self.order
        self.i = 0

    def next(self, shape):
        name = self.order[self.i % len(self.order)]
        self.i += 1
        m = self.masks[name]
        assert tuple(m.shape) == tuple(shape), (name, m.shape, shape)
        return m


def make_mi_masks(B, Ni, Nt, C, H, p, seed):
    g = torch.Generator().manual_seed(seed)

    def bern(*shape):
        return (torch.rand(*shape, generator=g) >= p)

    m = {"attn_i": bern(B, H, Ni, Ni), "attn_t": bern(B, H, Nt, Nt), "attn_c": bern(B, H, 1, 1 + Ni + Nt)}
    for s, n in (("i", Ni), ("t", Nt), ("c", 1)):
        m["proj_" + s] = bern(B, n, C)
        m["fc1_" + s] = bern(B, n, C)
        m["fc2_" + s] = bern(B, n, C)
        m["dp1_" + s] = bern(B)
        m["dp2_" + s] = bern(B)
    return m


def randomize_state(sd, seed):
    """Make every tensor non-trivial (BN affine / running stats, biases) so parity is discriminative."""
    g = torch.Generator().manual_seed(seed)
    if "DA_queue" in sd:  # a partly filled distribution-alignment queue (rows 0..9), pointer at 10
        K = sd["DA_queue"].shape[1]
        sd["DA_queue"][:10] = torch.softmax(torch.randn(10, K, generator=g), dim=1)
        sd["DA_ptr"][0] = 10
    for k, v in sd.items():
        if k.startswith("ema."):
            continue
        if k.endswith("num_batches_tracked") or k.startswith("prototypes") or k.startswith("DA_"):
            continue
        is_bn = (k.replace("weight", "running_var") in sd) and v.ndim == 1 and k.endswith("weight")
        if k.endswith("running_var") or is_bn:
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
        elif k.endswith("running_mean"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("bias"):
            v.copy_(0.05 * torch.randn(v.shape, generator=g))
        elif "norm" in k and k.endswith("weight"):
            v.copy_(0.75 + 0.5 * torch.rand(v.shape, generator=g))
    for k in list(sd.keys()):
        if k.startswith("model."):
            sd["ema." + k[6:]] = sd[k].clone()
    return sd


def craft_heads(sd, batch, hp, epoch, mask_random, scale=6.0, fi=0.5, ft=4.0):
    """Couple the three classifiers (imaging/tabular heads reuse slices of the multimodal head) and centre
    their logits on this batch, so that the CGPL partition case1/case2_i/case2_t/case3 and the th1 mask are
    all MIXED on a 14-sample batch (with plain random weights every sample lands in case3)."""
    s = {k: v.clone() for k, v in sd.items()}
    C = hp.multimodal_embedding_dim
    for pre in ("model.", "ema."):
        Wm = s[pre + "classifier_multimodal.weight"]
        s[pre + "classifier_imaging.weight"][:, :C] = Wm[:, :C] * fi
        s[pre + "classifier_tabular.weight"][:, :C] = Wm[:, 2 * C:] * ft
        s[pre + "classifier_imaging.weight"][:, C:] *= 0.2
        s[pre + "classifier_tabular.weight"][:, C:] *= 0.2
    with torch.no_grad():
        o = O.training_step({k: v.clone() for k, v in s.items()}, batch, hp, epoch, mask_random, None)
    for nm, key in (("classifier_multimodal", "y_hat_m_e"), ("classifier_imaging", "y_hat_i_e"), ("classifier_tabular", "y_hat_t_e")):
        mean = o[key].mean(0)
        for pre in ("model.", "ema."):
            s[pre + nm + ".weight"] *= scale
            s[pre + nm + ".bias"] = (s[pre + nm + ".bias"] - mean) * scale
    return s


CASES = {
    # name: (hparam overrides, B, epoch, use dropout masks, prefill prototypes + crafted heads)
    "dvm_r50_e0": (dict(img_size=64, num_classes=7, field_lengths=[3, 4, 5, 2, 6] + [1] * 6, batch_size=16), 16, 0, False, False),
    "dvm_r50_pseudo": (dict(img_size=64, num_classes=7, field_lengths=[3, 4, 5, 2, 6] + [1] * 6, batch_size=16,
                            th1=0.5, start_epoch=1, repeat_ratio=2.0), 16, 5, False, True),
    "dvm_r50_dropout": (dict(img_size=64, num_classes=7, field_lengths=[3, 4, 5, 2, 6] + [1] * 6, batch_size=16,
                             th1=0.5, start_epoch=1), 16, 5, True, True),
    "cardiac_r50": (dict(img_size=64, num_classes=2, target="CAD", field_lengths=[4] * 6 + [1] * 9, batch_size=16,
                         th1=0.62, start_epoch=1, rate_pseudo=0.95, ema_momentum=0.4, beta=1.0, gamma=1.0, lr_eval=1e-3),
                    16, 5, False, True),
    "dvm_r18_DA": (dict(model="resnet18", embedding_dim=512, img_size=64, num_classes=5, field_lengths=[3, 4] + [1] * 3,
                        batch_size=8, th1=0.25, start_epoch=1, DA=True), 8, 3, False, True),
    "dvm_saint": (dict(img_size=64, num_classes=7, field_lengths=[3, 4, 1, 5, 1, 1, 2], batch_size=16, th1=0.5, start_epoch=1,
                       tabular_encoder="saint"), 16, 5, True, True),
    "dvm_r18_noeman": (dict(model="resnet18", embedding_dim=512, img_size=64, num_classes=5,
                            field_lengths=[3, 4] + [1] * 3, batch_size=8, th1=0.25, start_epoch=1, eman=False),
                       8, 3, False, True),
    # BASELINE.json configs[0] at its real shape: ResNet-50, 224 px, 16 categorical (cardinality 8) + 48 continuous columns,
    # K = 286, batch 32 (4 labelled + 28 unlabelled), pseudo-label phase, MI-layer dropout masks injected (see build_case)
    "dvm_r50_b32_224": (dict(img_size=224, num_classes=286, field_lengths=[8] * 16 + [1] * 48, batch_size=32, th1=0.5, start_epoch=0),
                        32, 1, True, True),
}

VAL_KEYS = ["val_loss", "val_loss_ce", "val_loss_itc", "test_probs"]
SCALARS = ["loss", "loss_ce", "loss_itc", "loss_club_i", "loss_club_i_est", "loss_club_t", "loss_club_t_est",
           "loss_pt", "loss_m_u", "loss_i_u", "loss_t_u"]
TENSORS = ["y_hat_m", "y_hat_i", "y_hat_t", "x_si_enhance", "x_si", "x_ai", "x_st_enhance", "x_st", "x_at", "x_c",
           "feat_m", "feat_i", "feat_t", "y_hat_m_e", "y_hat_i_e", "y_hat_t_e", "feat_m_e", "pseudo_label_orig",
           "pseudo_label", "prediction", "case1", "case2_i", "case2_t", "case3", "mask1", "mask_random",
           "class_sum", "class_count"]
# locals of the reference's training_step read back from its frame (see run_reference)
LOCALS = ["pseudo_label_orig", "prediction", "case1", "case2_i", "case2_t", "case3", "mask1", "mask_random", "loss_pt"]
FULL_GRADS = ["model.encoder_tabular.simple_MLP.1.layers.0.weight", "model.encoder_tabular.embeds.weight",
              "model.encoder_tabular.pos_encodings.weight", "model.encoder_tabular.transformer.layers.0.2.fn.fn.to_out.bias",
              "model.encoder_tabular.transformer.layers.0.1.fn.fn.net.3.weight", "model.classifier_multimodal.weight", "model.reduce.bias", "model.encoder_tabular.cls_token",
              "model.encoder_tabular.con_proj.weight", "model.encoder_imaging.bn1.weight",
              "model.encoder_imaging.layer1.0.bn3.bias", "projector_imaging.bias", "CLUB_imaging.p_mu.2.bias",
              "model.transformer.0.attn.qkv.bias", "model.encoder_tabular.norm.weight",
              "model.encoder_imaging.conv1.weight", "model.encoder_tabular.cat_embedding.weight"]


def build_case(name):
    over, B, epoch, use_drop, prefill = CASES[name]
    hp = O.default_hparams(**over)
    sd = randomize_state(O.init_state(hp, seed=1234), seed=99)
    if prefill:
        g = torch.Generator().manual_seed(7)
        sd["prototypes"] = torch.nn.functional.normalize(torch.randn(hp.num_classes, hp.projection_dim, generator=g))
    batch = O.synthetic_batch(hp, B, seed=2022)
    g = torch.Generator().manual_seed(11)
    mask_random = torch.rand(B - max(B // 8, 1), generator=g).ge(0.5)
    mi_masks = None
    if use_drop:
        Ni = (hp.img_size // 32) ** 2
        Nt = len(hp.field_lengths)
        mi_masks = {0: make_mi_masks(B, Ni, Nt, hp.multimodal_embedding_dim, 4, hp.mi_drop, seed=5)}
        if hp.tabular_encoder == "saint":  # FF dropout p = 0.8 of the SAINT col / row feed-forwards (GEGLU output)
            g2 = torch.Generator().manual_seed(6)
            nf = Nt + 1
            mi_masks["saint"] = {"ff_col": torch.rand(B, nf, 4 * O.SAINT_DIM, generator=g2) >= hp.saint_ff_drop,
                                 "ff_row": torch.rand(1, B, 4 * O.SAINT_DIM * nf, generator=g2) >= hp.saint_ff_drop}
    if prefill and name == "dvm_r50_b32_224":
        # couple the three classifiers (imaging / tabular heads reuse the e_si / e_st slices of the multimodal head, whose e_st
        # slice is boosted: token means over 64 columns vary little between samples) so that the CGPL cases are mixed, and put
        # th1 into the widest gap of the confidence ranking (prediction does not depend on th1): a mixed mask1 that no rounding
        # difference between two evaluations can flip
        for pre_ in ("model.", "ema."):
            sd[pre_ + "classifier_multimodal.weight"][:, 512:1024] *= 0.3
            sd[pre_ + "classifier_multimodal.weight"][:, 1024:] *= 28.0
        sd = craft_heads(sd, batch, hp, epoch, mask_random, scale=6.0, fi=1.0, ft=1.0)
        with torch.no_grad():
            pre = O.training_step({k: v.clone() for k, v in sd.items()}, batch, hp, epoch, mask_random, None)
        B_u = B - max(B // 8, 1)
        conf = pre["prediction"].max(1)[0].sort()[0]
        lo = B_u // 4
        gaps = conf[lo + 1: B_u - lo + 1] - conf[lo: B_u - lo]
        j = int(gaps.argmax()) + lo
        hp.th1 = float((conf[j] + conf[j + 1]) / 2)
        assert float(gaps.max()) > 1e-4
    elif prefill:
        sd = craft_heads(sd, batch, hp, epoch, mask_random)
    return hp, sd, batch, epoch, mask_random, mi_masks


def run_reference(hp, sd, batch, epoch, mask_random, mi_masks):
    import models.Disentangle.utils.disentangle_transformer as DT
    if hp.tabular_encoder == "saint":
        from models.Disentangle.STiLModel_SAINT import SemiDisCoPseudoSmooth as STiLModel
    else:
        from models.Disentangle.STiLModel import STiLModel

    with tempfile.TemporaryDirectory() as td:
        fl = os.path.join(td, "fl.pt")
        torch.save(list(hp.field_lengths), fl)
        rh = ref_hparams(hp, fl)
        rh["checkpoint_SAINT"] = None
        model = STiLModel(rh)
    ref_keys = list(model.state_dict().keys())
    missing = set(ref_keys) ^ set(sd.keys())
    assert not missing, f"state_dict key mismatch: {sorted(missing)[:10]}"
    assert ref_keys == list(sd.keys()), "state_dict ORDER differs from the reference"
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    if hp.DA and not torch.distributed.is_initialized():  # the reference's DA path calls all_reduce unguarded
        torch.distributed.init_process_group("gloo", init_method="tcp://127.0.0.1:29731", rank=0, world_size=1)
    model.train()
    model.current_epoch = epoch

    provider = MaskProvider(mi_masks[0], mi_masks.get("saint")) if mi_masks else None
    orig_dropout_fwd, orig_drop_path, orig_rand_like = nn.Dropout.forward, DT.drop_path, torch.rand_like

    def dropout_fwd(self, x):
        if self.p == 0.0 or not self.training:
            return x
        if provider is None:
            return x  # dropout disabled for this case
        return x * provider.next(x.shape).to(x.dtype) / (1.0 - self.p)

    def drop_path(x, drop_prob=0.0, training=False):
        if drop_prob == 0.0 or not training or provider is None:
            return x
        keep = provider.next((x.shape[0],)).to(x.dtype)
        return x.div(1 - drop_prob) * keep.reshape((x.shape[0],) + (1,) * (x.ndim - 1))

    def rand_like(t, **kw):
        assert t.shape == mask_random.shape
        return torch.where(mask_random, torch.full_like(t, 0.75), torch.full_like(t, 0.25))

    nn.Dropout.forward, DT.drop_path, torch.rand_like = dropout_fwd, drop_path, rand_like
    try:
        opt = torch.optim.Adam([
            {"params": model.model.parameters()}, {"params": model.projector_imaging.parameters()},
            {"params": model.projector_tabular.parameters()}, {"params": model.projector_multimodal.parameters()},
            {"params": model.CLUB_imaging.parameters()}, {"params": model.CLUB_tabular.parameters()}],
            lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)  # STiLModel.py:563-570
        opt.zero_grad()
        # capture intermediates by wrapping forward_all / project_3features
        cap = {}
        fa_s, fa_e, p3 = model.model.forward_all, model.ema.forward_all, model.project_3features

        def wrap_s(x, *a, **k):
            cap["s"] = fa_s(x, *a, **k); return cap["s"]

        def wrap_e(x, *a, **k):
            cap["e"] = fa_e(x, *a, **k); return cap["e"]

        def wrap_p3(feat_m=None, feat_i=None, feat_t=None):
            r = p3(feat_m, feat_i, feat_t)
            cap.setdefault("p3", []).append(r); return r

        model.model.forward_all, model.ema.forward_all, model.project_3features = wrap_s, wrap_e, wrap_p3
        ps0, pc0 = model.prototypes_sum.clone(), model.prototypes_count_sum.clone()
        # CGPL / PGLS intermediates are locals of the reference's training_step (STiLModel.py:259-299): read them from
        # its frame when it returns (a profile hook: nothing in the reference is edited), and take `pseudo_label`
        # (re-bound to one column for K = 2 before the return, STiLModel.py:353) from the soft-target cross_entropy call.
        import torch.nn.functional as RF
        orig_ce, orig_profile = RF.cross_entropy, sys.getprofile()
        step_code = type(model).training_step.__code__

        def ce_spy(inp, target, *a, **k):
            if target.is_floating_point() and k.get("reduction") == "none":
                cap.setdefault("pseudo_label", target.detach().clone())
            return orig_ce(inp, target, *a, **k)

        def prof(frame, event, arg):
            if event == "return" and frame.f_code is step_code:
                loc = frame.f_locals
                for nm in LOCALS:
                    cap["loc_" + nm] = loc[nm].detach().clone()

        RF.cross_entropy = ce_spy
        sys.setprofile(prof)
        try:
            loss = model.training_step(batch, 0)
        finally:
            sys.setprofile(orig_profile)
            RF.cross_entropy = orig_ce
        loss.backward()
        grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model.named_parameters()
                 if not k.startswith("ema.")}
        opt.step()
    finally:
        nn.Dropout.forward, DT.drop_path, torch.rand_like = orig_dropout_fwd, orig_drop_path, orig_rand_like
    # ---- inference hooks on the post-step model, eval mode (Lightning's validation / test loops)
    model.model.forward_all, model.ema.forward_all, model.project_3features = fa_s, fa_e, p3
    model.eval()
    vx = [torch.cat((batch["l"][0][1], batch["u"][0][1])), torch.cat((batch["l"][1][1], batch["u"][1][1]))]
    vy = torch.cat((batch["l"][2], batch["u"][2]))
    model.logged = dict(model.logged)
    with torch.no_grad():
        vloss = model.validation_step((vx, vy), 0)
        captured = {}
        for nm in ("acc_test", "auc_test"):
            getattr(model, nm).forward = (lambda p_, y_, _n=nm: captured.__setitem__(_n, p_.detach().clone()))
        model.test_step((vx, vy), 0)
    val = dict(val_loss=vloss.detach(), val_loss_ce=model.logged["multimodal.val.CEloss"], val_loss_itc=model.logged["multimodal.val.ITCloss"],
               test_probs=captured["acc_test"])
    L = model.logged
    s, e = cap["s"], cap["e"]
    out = dict(loss=loss.detach(), loss_ce=L["multimodal.train.CEloss"], loss_itc=L["multimodal.train.ITCloss"],
               loss_club_i=L["multimodal.train.CLUBloss_imaging"], loss_club_i_est=L["multimodal.train.CLUBloss_imaging_est"],
               loss_club_t=L["multimodal.train.CLUBloss_tabular"], loss_club_t_est=L["multimodal.train.CLUBloss_tabular_est"],
               loss_m_u=L["multimodal.train.CEloss_unlabelled_m"], loss_i_u=L["multimodal.train.CEloss_unlabelled_i"],
               loss_t_u=L["multimodal.train.CEloss_unlabelled_t"])
    names = ["y_hat_m", "y_hat_i", "y_hat_t", "x_si_enhance", "x_si", "x_ai", "x_st_enhance", "x_st", "x_at", "x_c"]
    for n, t in zip(names, s):
        out[n] = t.detach()
    out["y_hat_m_e"], out["y_hat_i_e"], out["y_hat_t_e"] = e[0].detach(), e[1].detach(), e[2].detach()
    out["feat_m"], out["feat_i"], out["feat_t"] = (t.detach() for t in cap["p3"][0])
    out["feat_m_e"] = cap["p3"][1][0].detach()
    for nm in LOCALS:  # the reference's own pseudo-labels, case masks and threshold mask
        out[nm] = cap["loc_" + nm]
    out["pseudo_label"] = cap["pseudo_label"]
    out["class_sum"] = model.prototypes_sum - ps0
    out["class_count"] = model.prototypes_count_sum - pc0
    out.update(val)
    out = {k: (v.detach() if isinstance(v, torch.Tensor) else torch.tensor(v)) for k, v in out.items()}
    return out, grads, {k: v.detach().clone() for k, v in model.state_dict().items()}


def run_oracle64(hp, sd, batch, epoch, mask_random, mi_masks, decisions=None):
    """The oracle's full step in float64 (ground truth for gradient conditioning).  decisions = (relu, pool) dicts:
    replay those ReLU signs / max-pool winners instead of deciding anew (stil_oracle.force_decisions); the tally of
    units that float64 would have decided differently comes back as out["flips"]."""
    f64 = torch.float64
    s = {k: (v.clone().to(f64) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    b = {k: ([v[0][0].to(f64), v[0][1].to(f64)], [v[1][0].to(f64), v[1][1].to(f64)], v[2], v[3].to(f64), v[4]) for k, v in batch.items()}
    if decisions is None:
        return O.full_step(s, {}, 1, b, hp, epoch, mask_random, mi_masks)
    with O.force_decisions(*decisions) as d:
        out = O.full_step(s, {}, 1, b, hp, epoch, mask_random, mi_masks)
    out["flips"] = d.get("flips", {})
    return out


def close(a, b, tol=2e-5):
    """|a-b| <= tol * (1 + |b| + max|b|): fp32 reassociation noise scales with the tensor's magnitude."""
    a, b = a.double(), b.double()
    scale = float(b.abs().max()) if b.numel() else 0.0
    return bool(((a - b).abs() <= tol * (1.0 + b.abs() + scale)).all())


def main():
    sys.path.insert(0, REF)
    install_stubs()
    torch.manual_seed(0)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    only = [a for a in sys.argv[1:] if a in CASES]          # `python oracle/make_golden.py [case ...]`: default every case
    for name in (only or CASES):
        hp, sd, batch, epoch, mask_random, mi_masks = build_case(name)
        ref_out, ref_grads, ref_state = run_reference(hp, {k: v.clone() for k, v in sd.items()}, batch, epoch, mask_random, mi_masks)
        sd_o = {k: v.clone() for k, v in sd.items()}
        opt = {}
        o = O.full_step(sd_o, opt, 1, batch, hp, epoch, mask_random, mi_masks)
        vx_img = torch.cat((batch["l"][0][1], batch["u"][0][1])); vx_tab = torch.cat((batch["l"][1][1], batch["u"][1][1]))
        vy = torch.cat((batch["l"][2], batch["u"][2]))
        ov = O.validation_step(sd_o, vx_img, vx_tab, vy, hp)
        o["val_loss"], o["val_loss_ce"], o["val_loss_itc"] = ov["loss"], ov["loss_ce"], ov["loss_itc"]
        o["test_probs"] = O.test_step(sd_o, vx_img, vx_tab, hp)
        # ---- pin: oracle == reference
        bad = []
        for k, v in ref_out.items():
            if k in o and not close(o[k].float(), v.float()):
                bad.append((k, float((o[k].float() - v.float()).abs().max())))
        for k, g in ref_grads.items():
            go = o["grads"].get(k)
            if g is None:
                assert go is None or float(go.abs().max()) == 0.0, k
                continue
            if not close(go, g, tol=5e-5):
                bad.append(("grad:" + k, float((go - g).abs().max())))
        tr = set(O.trainable_keys(sd))
        for k, v in ref_state.items():
            if k in tr:
                # Adam turns noise-level gradients into +-lr updates (m/sqrt(v) ~ sign g): elementwise
                # bound 2.2*lr here; Adam itself is pinned exactly below on identical gradients.
                if float((sd_o[k] - v).abs().max()) > 2.2 * hp.lr_eval:
                    bad.append(("adam:" + k, float((sd_o[k] - v).abs().max())))
            elif not close(sd_o[k].float(), v.float(), tol=2e-5):
                bad.append(("state:" + k, float((sd_o[k].float() - v.float()).abs().max())))
        # Adam pinned exactly: oracle.adam_step on the REFERENCE's gradients must reproduce torch.optim.Adam
        sd_a = {k: v.clone() for k, v in sd.items()}
        O.adam_step(sd_a, {k: g for k, g in ref_grads.items() if g is not None}, {}, 1, hp.lr_eval, hp.weight_decay_eval)
        for k in tr:
            if ref_grads.get(k) is not None and float((sd_a[k] - ref_state[k]).abs().max()) > 1e-7:
                bad.append(("adam_exact:" + k, float((sd_a[k] - ref_state[k]).abs().max())))
        assert not bad, f"[{name}] oracle != reference: {bad[:8]}"
        # ---- store the reference's numbers
        fx = {"meta_epoch": np.int64(epoch)}
        for k in SCALARS:
            fx["out_" + k] = ref_out[k].numpy().astype(np.float64)
        for k in VAL_KEYS:
            fx["out_" + k] = ref_out[k].numpy()
        for k in TENSORS:
            fx["out_" + k] = ref_out[k].numpy()  # every tensor is the REFERENCE's (CGPL / PGLS internals read from its frame)
        # Conditioning: the same step in float64.  Deep train-mode-BN backward amplifies fp32 rounding (the
        # reference's own fp32 gradients sit up to ~2e-2 relative L2 away from the fp64 truth in these cases), so
        # every gradient is stored with gerr32 = relL2(reference fp32, fp64): the GPU path must be as close to the
        # fp64 truth as the reference's CPU path is (tests allow 3x gerr32 + a floor), not bit-near one fp32 run.
        o64 = run_oracle64(hp, sd, batch, epoch, mask_random, mi_masks)
        for k, g in ref_grads.items():
            fx["gnorm_" + k] = np.float64(0.0 if g is None else g.double().norm().item())
            if g is not None:
                g64 = o64["grads"][k]
                fx["g64norm_" + k] = np.float64(g64.norm().item())
                fx["gerr32_" + k] = np.float64(((g.double() - g64).norm() / (g64.norm() + 1e-30)).item())
        for k in FULL_GRADS:
            if k in ref_grads and ref_grads[k] is not None:
                fx["grad_" + k] = ref_grads[k].numpy()
                fx["grad64_" + k] = o64["grads"][k].numpy()
        for k, v in ref_state.items():  # post-step state checksums: EMA teacher + BN buffers + prototype accumulators
            if k in tr:
                continue  # Adam-updated tensors are noise-amplified (see above); Adam is tested on fixed gradients
            fx["ssum_" + k] = np.float64(v.double().sum().item())
            fx["sabs_" + k] = np.float64(v.double().abs().sum().item())
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **fx)
        n_none = sum(1 for g in ref_grads.values() if g is None)
        print(f"[{name}] oracle==reference OK  loss={float(ref_out['loss']):.6f}  mask1={int(o['mask1'].sum())}/{len(o['mask1'])} "
              f"case1={int(o['case1'].sum())} c2i={int(o['case2_i'].sum())} c2t={int(o['case2_t'].sum())} c3={int(o['case3'].sum())} "
              f"params_without_grad={n_none}  -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def adam_layout(name="dvm_r18_noeman"):
    """What the REFERENCE's configure_optimizers (STiLModel.py:557-577) builds: torch.optim.Adam over six parameter groups.
    Its state_dict() after one step -- group sizes, group hyper-parameters, the shape of every parameter in id order,
    which ids hold state -- is the `optimizer_states[0]` entry of a Lightning checkpoint; stored as
    tests/golden/adam_layout.npz so that fit.adam_state_dict can be held to it."""
    hp, sd, batch, epoch, mask_random, mi_masks = build_case(name)
    from models.Disentangle.STiLModel import STiLModel
    import models.Disentangle.STiLModel as RM

    class Sched:  # stands in for pl_bolts' LinearWarmupCosineAnnealingLR (absent offline): constructor signature only
        def __init__(self, optimizer, warmup_epochs, max_epochs):
            self.optimizer, self.warmup_epochs, self.max_epochs = optimizer, warmup_epochs, max_epochs

    RM.LinearWarmupCosineAnnealingLR = Sched
    with tempfile.TemporaryDirectory() as td:
        fl = os.path.join(td, "fl.pt")
        torch.save(list(hp.field_lengths), fl)
        rh = ref_hparams(hp, fl)
        rh.update(scheduler="anneal", warmup_epochs=10, max_epochs=500)
        model = STiLModel(rh)
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.train()
    model.current_epoch = epoch
    conf = model.configure_optimizers()
    opt = conf["optimizer"]
    orig_rand_like = torch.rand_like
    torch.rand_like = lambda t, **kw: torch.where(mask_random, torch.full_like(t, 0.75), torch.full_like(t, 0.25))
    try:
        opt.zero_grad()
        model.training_step(batch, 0).backward()
        opt.step()
    finally:
        torch.rand_like = orig_rand_like
    osd = opt.state_dict()
    params = [p_ for g in opt.param_groups for p_ in g["params"]]
    names = {id(p_): n for n, p_ in model.named_parameters()}
    fx = {"group_sizes": np.array([len(g["params"]) for g in osd["param_groups"]], dtype=np.int64),
          "lr": np.array([g["lr"] for g in osd["param_groups"]]), "weight_decay": np.array([g["weight_decay"] for g in osd["param_groups"]]),
          "betas": np.array([g["betas"] for g in osd["param_groups"]]), "eps": np.array([g["eps"] for g in osd["param_groups"]]),
          "amsgrad": np.array([bool(g["amsgrad"]) for g in osd["param_groups"]]),
          "ids": np.array([i for g in osd["param_groups"] for i in g["params"]], dtype=np.int64),
          "numel": np.array([p_.numel() for p_ in params], dtype=np.int64), "ndim": np.array([p_.ndim for p_ in params], dtype=np.int64),
          "shapes": np.array([d for p_ in params for d in p_.shape], dtype=np.int64),
          "has_state": np.array([i in osd["state"] for i in range(len(params))]),
          "names": np.array([names[id(p_)] for p_ in params]),
          "state_keys": np.array(sorted(next(iter(osd["state"].values())).keys())),
          "steps": np.array([float(osd["state"][i]["step"]) if i in osd["state"] else 0.0 for i in range(len(params))])}
    path = os.path.join(ROOT, "tests", "golden", "adam_layout.npz")
    np.savez_compressed(path, **fx)
    print(f"[adam_layout:{name}] groups {fx['group_sizes'].tolist()}, {len(params)} parameters, {int(fx['has_state'].sum())} with state, "
          f"state keys {fx['state_keys'].tolist()} -> {os.path.relpath(path, ROOT)}")


if __name__ == "__main__":
    main()
    adam_layout()
