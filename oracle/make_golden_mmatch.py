"""Pin oracle/mmatch_oracle.py against the REAL reference MMatch (models/SemiMultimodal/MMatch.py) and write
tests/golden/mmatch_*.npz.  Build container only (needs /root/reference); same stub recipe as make_golden.py.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_mmatch.py
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import stil_oracle as O  # noqa: E402
from oracle import mmatch_oracle as MO  # noqa: E402
from oracle import make_golden as G  # noqa: E402

FL = [3, 4] + [1] * 3
CASES = {
    # name: (hparam overrides, epoch, batch, bank/DA state preset)
    "mmatch_r18_e0": (dict(model="resnet18", embedding_dim=512, img_size=64, field_lengths=FL, num_classes=5, batch_size=16), 0, 16, None),
    "mmatch_r18_bank": (dict(model="resnet18", embedding_dim=512, img_size=64, field_lengths=FL, num_classes=5, batch_size=16, th1=0.25), 2, 16,
                        dict(ptr=632, da_rows=3)),   # memory bank in use; 632 + 16 > 640: the enqueue is truncated to 8 samples
    "mmatch_r18_binary": (dict(model="resnet18", embedding_dim=512, img_size=64, field_lengths=[4, 4] + [1] * 5, num_classes=2, batch_size=16,
                               th1=0.55, target="CAD"), 3, 16, dict(ptr=100, da_rows=40)),
}
SCALARS = ["loss", "loss_ce", "loss_i_u", "loss_t_u"]
TENSORS = ["y_hat_m", "y_hat_i", "y_hat_t", "x_m", "pseudo_label", "pseudo_label_orig", "mask1", "hard_idx"]


def build_case(name):
    over, epoch, B, preset = CASES[name]
    hp = MO.default_hparams(**over)
    sd = G.randomize_state(MO.init_state(hp, seed=5), seed=6)
    sd = {k: v for k, v in sd.items() if not k.startswith("ema.")}  # MMatch keeps no EMA copy
    g = torch.Generator().manual_seed(9)
    K = hp.num_classes
    if preset:
        sd["embed_queue"] = torch.nn.functional.normalize(torch.randn(hp.projection_dim, MO.BANK, generator=g), dim=0)
        sd["probs_queue"] = torch.softmax(torch.randn(K, MO.BANK, generator=g) * 2, dim=0)
        sd["embed_queue_ptr"] = torch.tensor([preset["ptr"]])
        sd["DA_queue"] = torch.zeros(256, K)
        sd["DA_queue"][: preset["da_rows"]] = torch.softmax(torch.randn(preset["da_rows"], K, generator=g), dim=1)
        sd["DA_ptr"] = torch.tensor([preset["da_rows"]])
    else:
        sd["embed_queue"] = torch.nn.functional.normalize(torch.randn(hp.projection_dim, MO.BANK, generator=g), dim=0)
        sd["probs_queue"] = torch.zeros(K, MO.BANK)
        sd["embed_queue_ptr"] = torch.zeros(1, dtype=torch.long)
        sd["DA_queue"] = torch.zeros(256, K)
        sd["DA_ptr"] = torch.zeros(1, dtype=torch.long)
    batch = O.synthetic_batch(hp, B, seed=31)
    if name == "mmatch_r18_bank":  # put the confidence threshold at the median so that mask1 is mixed
        dry = MO.training_step({k: v.clone() for k, v in sd.items()}, batch, hp, epoch)
        hp.th1 = float(dry["pseudo_label"].max(dim=1).values.median()) - 1e-4
    return hp, sd, batch, epoch


def run_reference(hp, sd, batch, epoch):
    from models.SemiMultimodal.MMatch import MMatch
    with tempfile.TemporaryDirectory() as td:
        fl = os.path.join(td, "fl.pt")
        torch.save(list(hp.field_lengths), fl)
        model = MMatch(G.ref_hparams(hp, fl))
    ref_keys = list(model.state_dict().keys())
    assert ref_keys == list(sd.keys()), f"state_dict keys/order differ: {sorted(set(ref_keys) ^ set(sd.keys()))[:10]}"
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.train()
    model.current_epoch = epoch
    cap = {}
    fwd = model.forward

    def forward(x):
        out = fwd(x)
        cap["out"] = out
        return out

    model.forward = forward
    params = dict(model.named_parameters())
    opt = torch.optim.Adam([{"params": model.model.parameters()}], lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in params.items()}
    opt.step()
    out = dict(loss=loss.detach(), loss_i_u=model.logged["multimodal.train.CEloss_unlabelled_i"].detach(),
               loss_t_u=model.logged["multimodal.train.CEloss_unlabelled_t"].detach(),
               y_hat_m=cap["out"][0].detach(), y_hat_i=cap["out"][1].detach(), y_hat_t=cap["out"][2].detach(), x_m=cap["out"][3].detach())
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.forward = fwd
    model.eval()
    x = [torch.cat((batch["l"][0][1], batch["u"][0][1])), torch.cat((batch["l"][1][1], batch["u"][1][1]))]
    y = torch.cat((batch["l"][2], batch["u"][2]))
    with torch.no_grad():
        out["val_loss"] = model.validation_step((x, y), 0).detach()
    return out, grads, state


def install_mmatch_stubs():
    """MMatch's backbone imports pl_bolts.utils.self_supervised.torchvision_ssl_encoder (lightning-bolts==0.5.0, absent
    offline).  The reference vendors that very function and its ResNet as models/self_supervised.py + models/resnets.py
    (used by the STiL backbone); the stub forwards to the vendored copy -- unpinned with respect to the pip package."""
    import types
    from models.self_supervised import torchvision_ssl_encoder
    sys.modules["pl_bolts"].__path__ = []  # make the stub a package
    m = types.ModuleType("pl_bolts.utils"); m.__path__ = []
    sys.modules["pl_bolts.utils"] = m
    m2 = types.ModuleType("pl_bolts.utils.self_supervised")
    m2.torchvision_ssl_encoder = torchvision_ssl_encoder
    sys.modules["pl_bolts.utils.self_supervised"] = m2


CO_CASES = {
    "cotrain_r18_eman": (dict(model="resnet18", embedding_dim=512, img_size=64, field_lengths=FL, num_classes=5, batch_size=16), 1, 16),
    "cotrain_r18_noema": (dict(model="resnet18", embedding_dim=512, img_size=64, field_lengths=FL, num_classes=5, batch_size=16, use_ema=False), 1, 16),
    "cotrain_r18_binary_paramema": (dict(model="resnet18", embedding_dim=512, img_size=64, field_lengths=[4, 4] + [1] * 5, num_classes=2, batch_size=16,
                                         eman=False, target="CAD"), 2, 16),
}
CO_SCALARS = ["loss", "loss_ce", "loss_i_u", "loss_t_u"]
CO_TENSORS = ["y_hat_m", "y_hat_i", "y_hat_t", "y_hat_i_e", "y_hat_t_e", "pseudo_label_i", "pseudo_label_t", "mask_i", "mask_t"]


def build_co_case(name):
    over, epoch, B = CO_CASES[name]
    hp = MO.cotrain_hparams(**over)
    sd = G.randomize_state(MO.cotrain_init_state(hp, seed=7), seed=8)   # randomize_state re-mirrors ema.* from model.*
    if not hp.use_ema:
        sd = {k: v for k, v in sd.items() if not k.startswith("ema.")}
    else:  # a teacher that lags the student a little, so that the EMA arithmetic is visible
        g = torch.Generator().manual_seed(3)
        for k in sd:
            if k.startswith("ema.") and sd[k].is_floating_point() and not k.endswith("running_var"):
                sd[k] = sd[k] + 0.01 * torch.randn(sd[k].shape, generator=g) * (1.0 + sd[k].abs())
    batch = O.synthetic_batch(hp, B, seed=41)
    dry = MO.cotrain_training_step({k: v.clone() for k, v in sd.items()}, batch, hp, epoch)
    both = torch.cat((dry["pseudo_label_i"].max(dim=1).values, dry["pseudo_label_t"].max(dim=1).values))
    hp.co_threshold = float(both.median()) - 1e-4   # mixed confidence masks
    return hp, sd, batch, epoch


def run_co_reference(hp, sd, batch, epoch):
    from models.SemiMultimodal.CoTraining import CoTraining
    with tempfile.TemporaryDirectory() as td:
        fl = os.path.join(td, "fl.pt")
        torch.save(list(hp.field_lengths), fl)
        model = CoTraining(G.ref_hparams(hp, fl))
    ref_keys = list(model.state_dict().keys())
    assert ref_keys == list(sd.keys()), f"state_dict keys/order differ: {sorted(set(ref_keys) ^ set(sd.keys()))[:10]}"
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.train()
    model.current_epoch = epoch
    cap = {}
    for nm, mod in (("s", model.model),) + ((("t", model.ema),) if hp.use_ema else ()):
        f0 = mod.forward

        def fwd(x, f0=f0, nm=nm):
            out = f0(x)
            cap[nm] = out
            return out

        mod.forward = fwd
    params = {k: p for k, p in model.named_parameters() if k.startswith("model.")}
    opt = torch.optim.Adam([{"params": model.model.parameters()}], lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in params.items()}
    opt.step()
    out = dict(loss=loss.detach(), loss_i_u=model.logged["multimodal.train.CEloss_unlabelled_i"].detach(),
               loss_t_u=model.logged["multimodal.train.CEloss_unlabelled_t"].detach(),
               y_hat_m=cap["s"][0].detach(), y_hat_i=cap["s"][1].detach(), y_hat_t=cap["s"][2].detach())
    if hp.use_ema:
        out["y_hat_i_e"], out["y_hat_t_e"] = cap["t"][1].detach(), cap["t"][2].detach()
    return out, grads, {k: v.detach().clone() for k, v in model.state_dict().items()}


def co_main():
    for name in CO_CASES:
        hp, sd, batch, epoch = build_co_case(name)
        ref_out, ref_grads, ref_state = run_co_reference(hp, {k: v.clone() for k, v in sd.items()}, batch, epoch)
        sd_o = {k: v.clone() for k, v in sd.items()}
        o = MO.cotrain_full_step(sd_o, {}, 1, batch, hp, epoch)
        bad = []
        for k, v in ref_out.items():
            if not G.close(o[k].float(), v.float()):
                bad.append((k, float((o[k].float() - v.float()).abs().max())))
        for k, g in ref_grads.items():
            go = o["grads"].get(k)
            if g is None:
                assert go is None or float(go.abs().max()) == 0.0, k
            elif not G.close(go, g, tol=5e-5):
                bad.append(("grad:" + k, float((go - g).abs().max())))
        tr = set(MO.trainable_keys(sd))
        for k, v in ref_state.items():
            if k in tr:
                if float((sd_o[k] - v).abs().max()) > 2.2 * hp.lr_eval:
                    bad.append(("adam:" + k, float((sd_o[k] - v).abs().max())))
            elif not G.close(sd_o[k].float(), v.float(), tol=2e-5):
                bad.append(("state:" + k, float((sd_o[k].float() - v.float()).abs().max())))
        assert not bad, f"[{name}] oracle != reference: {bad[:8]}"
        sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        b64 = {kk: ([t.double() for t in vv[0]], [t.double() for t in vv[1]], vv[2], vv[3].double(), vv[4]) for kk, vv in batch.items()}
        o64 = MO.cotrain_full_step(sd64, {}, 1, b64, hp, epoch)
        fx = {"meta_epoch": np.int64(epoch), "meta_co_threshold": np.float64(hp.co_threshold)}
        for k in CO_SCALARS:
            fx["out_" + k] = (ref_out[k] if k in ref_out else o[k]).numpy().astype(np.float64)
        for k in CO_TENSORS:
            fx["out_" + k] = (ref_out[k] if k in ref_out else o[k]).numpy()
        for k, g in ref_grads.items():
            fx["gnorm_" + k] = np.float64(0.0 if g is None else g.double().norm().item())
            if g is not None:
                g64 = o64["grads"][k]
                fx["g64norm_" + k] = np.float64(g64.norm().item())
                fx["gerr32_" + k] = np.float64(((g.double() - g64).norm() / (g64.norm() + 1e-30)).item())
        for k in ("model.classifier_imaging.weight", "model.classifier_tabular.weight", "model.encoder_imaging.conv1.weight"):
            if ref_grads.get(k) is not None:
                fx["grad_" + k] = ref_grads[k].numpy(); fx["grad64_" + k] = o64["grads"][k].numpy()
        for k, v in ref_state.items():
            if k not in tr:
                fx["ssum_" + k] = np.float64(v.double().sum().item()); fx["sabs_" + k] = np.float64(v.double().abs().sum().item())
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"{name}: oracle==reference OK  loss {float(ref_out['loss']):.6f}  masks {int(o['mask_i'].sum())}+{int(o['mask_t'].sum())}/{len(o['mask_i'])}  -> {os.path.getsize(path) / 1e3:.0f} kB")


# ---- CoTraining with the SAINT tabular encoder (models/SemiMultimodal/CoTraining_SAINT.py)
FLS = [3, 25, 4] + [1] * 3   # categories_offset = (0, 1, 4, 29): 29 is one of the values the reference's EMA of the int64 buffers truncates to 28
COS_CASES = {
    "cotrain_saint_r18_eman": (dict(model="resnet18", embedding_dim=512, img_size=64, field_lengths=FLS, num_classes=5, batch_size=16), 1, 16),
    "cotrain_saint_r18_paramema": (dict(model="resnet18", embedding_dim=512, img_size=64, field_lengths=FLS, num_classes=5, batch_size=16, eman=False), 2, 16),
}


def build_cos_case(name):
    over, epoch, B = COS_CASES[name]
    hp = MO.cotrain_saint_hparams(**over)
    sd = MO.cotrain_saint_init_state(hp, seed=7)
    g = torch.Generator().manual_seed(8)
    for k in list(sd.keys()):                        # non-trivial BN statistics, a teacher that lags the student a little
        if not k.startswith("model."):
            continue
        v = sd[k]
        if k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7)
        elif k == "model.cls_token":
            sd[k] = torch.zeros(1, 1)                # forward_tabular casts it to long with the categorical columns: 0 = the CLS row of `embeds`
        elif ".bn" in k or "downsample.1" in k:
            sd[k] = v + 0.1 * torch.randn(v.shape, generator=g)
        ke = "ema." + k[len("model."):]
        if ke in sd:
            if sd[k].is_floating_point() and not k.endswith("running_var") and k != "model.cls_token":
                sd[ke] = sd[k] + 0.01 * torch.randn(sd[k].shape, generator=g) * (1.0 + sd[k].abs())
            else:
                sd[ke] = sd[k].clone()
    batch = O.synthetic_batch(hp, B, seed=43)
    g2 = torch.Generator().manual_seed(6)
    nf = len(hp.field_lengths) + 1
    masks = {"ff_col": torch.rand(B, nf, 4 * O.SAINT_DIM, generator=g2) >= hp.saint_ff_drop,
             "ff_row": torch.rand(1, B, 4 * O.SAINT_DIM * nf, generator=g2) >= hp.saint_ff_drop}
    dry = MO.cotrain_saint_training_step({k: v.clone() for k, v in sd.items()}, batch, hp, epoch, masks)
    both = torch.cat((dry["pseudo_label_i"].max(dim=1).values, dry["pseudo_label_t"].max(dim=1).values))
    hp.co_threshold = float(both.median()) - 1e-4
    return hp, sd, batch, epoch, masks


def run_cos_reference(hp, sd, batch, epoch, masks):
    import torch.nn as nn
    from models.SemiMultimodal.CoTraining_SAINT import CoTraining
    with tempfile.TemporaryDirectory() as td:
        fl = os.path.join(td, "fl.pt")
        torch.save(list(hp.field_lengths), fl)
        rh = G.ref_hparams(hp, fl)
        rh["checkpoint_SAINT"] = None
        model = CoTraining(rh)
    ref_keys = list(model.state_dict().keys())
    assert ref_keys == list(sd.keys()), f"state_dict keys/order differ: {sorted(set(ref_keys) ^ set(sd.keys()))[:10]} {[(a, b) for a, b in zip(ref_keys, sd.keys()) if a != b][:3]}"
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.train()
    model.current_epoch = epoch
    cap = {}
    for nm, mod in (("s", model.model),) + ((("t", model.ema),) if hp.use_ema else ()):
        f0 = mod.forward

        def fwd(x, f0=f0, nm=nm):
            out = f0(x)
            cap[nm] = out
            return out

        mod.forward = fwd
    provider = G.MaskProvider({}, masks)
    provider.order = ["ff_col", "ff_row"]
    orig_dropout = nn.Dropout.forward

    def dropout_fwd(self, x):
        if self.p == 0.0 or not self.training:
            return x
        return x * provider.next(x.shape).to(x.dtype) / (1.0 - self.p)

    nn.Dropout.forward = dropout_fwd
    try:
        params = {k: p for k, p in model.named_parameters() if k.startswith("model.")}
        opt = torch.optim.Adam([{"params": model.model.parameters()}], lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        assert provider.i == 2, provider.i      # exactly the student's two feed-forward dropouts drew a mask
    finally:
        nn.Dropout.forward = orig_dropout
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in params.items()}
    opt.step()
    out = dict(loss=loss.detach(), loss_i_u=model.logged["multimodal.train.CEloss_unlabelled_i"].detach(),
               loss_t_u=model.logged["multimodal.train.CEloss_unlabelled_t"].detach(),
               y_hat_m=cap["s"][0].detach(), y_hat_i=cap["s"][1].detach(), y_hat_t=cap["s"][2].detach())
    if hp.use_ema:
        out["y_hat_i_e"], out["y_hat_t_e"] = cap["t"][1].detach(), cap["t"][2].detach()
    return out, grads, {k: v.detach().clone() for k, v in model.state_dict().items()}


def cos_main():
    for name in COS_CASES:
        hp, sd, batch, epoch, masks = build_cos_case(name)
        ref_out, ref_grads, ref_state = run_cos_reference(hp, {k: v.clone() for k, v in sd.items()}, batch, epoch, masks)
        sd_o = {k: v.clone() for k, v in sd.items()}
        o = MO.cotrain_saint_full_step(sd_o, {}, 1, batch, hp, epoch, masks)
        bad = []
        for k, v in ref_out.items():
            if not G.close(o[k].float(), v.float()):
                bad.append((k, float((o[k].float() - v.float()).abs().max())))
        for k, g in ref_grads.items():
            go = o["grads"].get(k)
            if g is None:
                assert go is None or float(go.abs().max()) == 0.0, k
            elif not G.close(go, g, tol=5e-5):
                bad.append(("grad:" + k, float((go - g).abs().max())))
        tr = set(o["grads"].keys())
        for k, v in ref_state.items():
            if k in tr:
                if float((sd_o[k] - v).abs().max()) > 2.2 * hp.lr_eval:
                    bad.append(("adam:" + k, float((sd_o[k] - v).abs().max())))
            elif not v.is_floating_point():
                if not torch.equal(sd_o[k], v):
                    bad.append(("state:" + k, sd_o[k].tolist(), v.tolist()))
            elif not G.close(sd_o[k].float(), v.float(), tol=2e-5):
                bad.append(("state:" + k, float((sd_o[k].float() - v.float()).abs().max())))
        assert not bad, f"[{name}] oracle != reference: {bad[:8]}"
        sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        b64 = {kk: ([t.double() for t in vv[0]], [t.double() for t in vv[1]], vv[2], vv[3].double(), vv[4]) for kk, vv in batch.items()}
        o64 = MO.cotrain_saint_full_step(sd64, {}, 1, b64, hp, epoch, masks)
        fx = {"meta_epoch": np.int64(epoch), "meta_co_threshold": np.float64(hp.co_threshold)}
        for k in CO_SCALARS:
            fx["out_" + k] = (ref_out[k] if k in ref_out else o[k]).numpy().astype(np.float64)
        for k in CO_TENSORS:
            fx["out_" + k] = (ref_out[k] if k in ref_out else o[k]).numpy()
        for k, g in ref_grads.items():
            fx["gnorm_" + k] = np.float64(0.0 if g is None else g.double().norm().item())
            if g is not None:
                g64 = o64["grads"][k]
                fx["g64norm_" + k] = np.float64(g64.norm().item())
                fx["gerr32_" + k] = np.float64(((g.double() - g64).norm() / (g64.norm() + 1e-30)).item())
        for k, v in ref_state.items():
            if k in tr:
                continue
            if not v.is_floating_point() and v.numel() > 1:
                fx["state_" + k] = v.numpy()          # the integer offset buffers, student and teacher, as the reference leaves them
            else:
                fx["ssum_" + k] = np.float64(v.double().sum().item()); fx["sabs_" + k] = np.float64(v.double().abs().sum().item())
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **fx)
        off = ref_state.get("ema.encoder_tabular.categories_offset")
        print(f"{name}: oracle==reference OK  loss {float(ref_out['loss']):.6f}  masks {int(o['mask_i'].sum())}+{int(o['mask_t'].sum())}/{len(o['mask_i'])}"
              f"  teacher categories_offset {None if off is None else off.tolist()}  -> {os.path.getsize(path) / 1e3:.0f} kB")


def main():
    sys.path.insert(0, G.REF)
    G.install_stubs()
    install_mmatch_stubs()
    only = os.environ.get("ONLY")
    if only in (None, "saint"):
        cos_main()
    if only == "saint":
        return
    if only != "mmatch":
        co_main()
    if only == "cotrain":
        return
    for name in CASES:
        hp, sd, batch, epoch = build_case(name)
        ref_out, ref_grads, ref_state = run_reference(hp, {k: v.clone() for k, v in sd.items()}, batch, epoch)
        sd_o = {k: v.clone() for k, v in sd.items()}
        o = MO.full_step(sd_o, {}, 1, batch, hp, epoch)
        x_img = torch.cat((batch["l"][0][1], batch["u"][0][1])); x_tab = torch.cat((batch["l"][1][1], batch["u"][1][1]))
        y_all = torch.cat((batch["l"][2], batch["u"][2]))
        ov = MO.validation_step(sd_o, x_img, x_tab, y_all, hp)
        o["val_loss"] = ov["loss"]
        bad = []
        for k, v in ref_out.items():
            if not G.close(o[k].float(), v.float()):
                bad.append((k, float((o[k].float() - v.float()).abs().max())))
        for k, g in ref_grads.items():
            go = o["grads"].get(k)
            if g is None:
                assert go is None or float(go.abs().max()) == 0.0, k
            elif not G.close(go, g, tol=5e-5):
                bad.append(("grad:" + k, float((go - g).abs().max())))
        tr = set(MO.trainable_keys(sd))
        for k, v in ref_state.items():
            if k in tr:
                if float((sd_o[k] - v).abs().max()) > 2.2 * hp.lr_eval:
                    bad.append(("adam:" + k, float((sd_o[k] - v).abs().max())))
            elif not G.close(sd_o[k].float(), v.float(), tol=2e-5):
                bad.append(("state:" + k, float((sd_o[k].float() - v.float()).abs().max())))
        assert not bad, f"[{name}] oracle != reference: {bad[:8]}"
        # float64 yardstick for the gradients (see make_golden.py)
        sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        b64 = {kk: ([t.double() for t in vv[0]], [t.double() for t in vv[1]], vv[2], vv[3].double(), vv[4]) for kk, vv in batch.items()}
        o64 = MO.full_step(sd64, {}, 1, b64, hp, epoch)
        fx = {"meta_epoch": np.int64(epoch), "meta_th1": np.float64(hp.th1)}
        for k in SCALARS + ["val_loss"]:
            fx["out_" + k] = (ref_out[k] if k in ref_out else o[k]).numpy().astype(np.float64)
        for k in TENSORS:
            fx["out_" + k] = (ref_out[k] if k in ref_out else o[k]).numpy()
        for k, g in ref_grads.items():
            fx["gnorm_" + k] = np.float64(0.0 if g is None else g.double().norm().item())
            if g is not None:
                g64 = o64["grads"][k]
                fx["g64norm_" + k] = np.float64(g64.norm().item())
                fx["gerr32_" + k] = np.float64(((g.double() - g64).norm() / (g64.norm() + 1e-30)).item())
        for k in ("model.classifier_multimodal.weight", "model.classifier_tabular.weight", "model.encoder_imaging.conv1.weight"):
            if ref_grads.get(k) is not None:
                fx["grad_" + k] = ref_grads[k].numpy(); fx["grad64_" + k] = o64["grads"][k].numpy()
        for k, v in ref_state.items():
            if k in tr:
                continue
            if k in ("probs_queue", "embed_queue_ptr", "DA_ptr"):
                fx["state_" + k] = v.numpy()
            elif k in ("embed_queue", "DA_queue"):  # only the rows / columns this step wrote, plus a checksum of the rest
                fx["ssum_" + k] = np.float64(v.double().sum().item()); fx["sabs_" + k] = np.float64(v.double().abs().sum().item())
                if k == "embed_queue":
                    p0 = int(sd["embed_queue_ptr"]); fx["state_embed_queue_cols"] = v[:, p0:p0 + 16].numpy()
                else:
                    fx["state_DA_queue_row"] = v[int(sd["DA_ptr"])].numpy()
            else:
                fx["ssum_" + k] = np.float64(v.double().sum().item()); fx["sabs_" + k] = np.float64(v.double().abs().sum().item())
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"{name}: oracle==reference OK  loss {float(ref_out['loss']):.6f}  mask1 {int(o['mask1'].sum())}/{len(o['mask1'])}  -> {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
