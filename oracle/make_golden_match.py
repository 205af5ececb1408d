"""Pin oracle/match_oracle.py against the REAL reference CoMatch / SimMatch (models/MatchModel/*.py) and write
tests/golden/comatch_*.npz, simmatch_*.npz.  Build container only (needs /root/reference); same stub recipe as
make_golden.py / make_golden_mmatch.py.  Nothing of the reference is edited: outputs of the model's forward are captured
by wrapping the bound method, the step's locals (loss terms, masks) are read from its frame with a profile hook.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_match.py
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

from oracle import stil_oracle as O  # noqa: E402,F401
from oracle import match_oracle as MO  # noqa: E402
from oracle import make_golden as G  # noqa: E402

FL = [3, 4] + [1] * 3
R18 = dict(model="resnet18", embedding_dim=512, img_size=64, field_lengths=FL, num_classes=5, batch_size=16)
CASES = {
    # name: (kind, hparam overrides, epoch, batch, preset)
    "comatch_r18_e0": ("comatch", dict(R18, K=40), 0, 16, None),
    # queues in use and about to wrap (34 + 14 > 40: the 's' enqueue is truncated to 6, 33 + 16 > 40: the 'w' one to 7)
    "comatch_r18_bank": ("comatch", dict(R18, K=40), 2, 16, dict(ptr_s=34, ptr_w=33, hist=3)),
    "comatch_r18_img_binary": ("comatch", dict(R18, K=24, num_classes=2, eval_datatype="imaging", lam_c=2.0), 3, 16, dict(ptr_s=3, ptr_w=8, hist=128)),
    "simmatch_r18_e0": ("simmatch", dict(R18, K=30), 0, 16, None),
    "simmatch_r18_bank": ("simmatch", dict(R18, K=30), 2, 16, dict(da_rows=5)),
    "simmatch_r18_img_noDA": ("simmatch", dict(R18, K=12, num_classes=2, eval_datatype="imaging", DA=False, c_smooth=1.0), 1, 16, dict(da_rows=0)),
    "freematch_r18_e0": ("freematch", dict(R18, lambda_u=1.0), 0, 16, None),
    "freematch_r18_mask": ("freematch", dict(R18, lambda_u=1.0, lambda_e=0.5), 2, 16, dict(seed=5)),
    "freematch_r18_img_binary": ("freematch", dict(R18, lambda_u=1.0, lambda_e=0.5, num_classes=2, eval_datatype="imaging"), 1, 16, dict(seed=6)),
}
OUT = {"comatch": (["loss", "loss_x", "loss_u", "loss_contrast"], ["outputs_x", "outputs_u_s0", "probs", "mask", "Q", "sim"]),
       "simmatch": (["loss", "loss_x", "loss_u", "loss_in"], ["logits_x", "logits_u_s", "pseudo_label", "mask"]),
       "freematch": (["loss", "sup_loss", "unsup_loss", "ent_loss"], ["logits_x_lb", "logits_x_ulb_s", "pseudo_label", "mask", "p_model", "label_hist", "time_p"])}
LOCALS = {"comatch": ["loss_x", "loss_u", "loss_contrast", "mask"], "simmatch": ["loss_x", "loss_u", "loss_in", "mask"],
          "freematch": ["sup_loss", "unsup_loss", "ent_loss", "mask"]}
STUDENT = {"comatch": "model.encoder.", "simmatch": "model.main.", "freematch": "model.main."}
INIT = {"comatch": MO.comatch_init_state, "simmatch": MO.simmatch_init_state, "freematch": MO.freematch_init_state}
STEP = {"comatch": MO.comatch_training_step, "simmatch": MO.simmatch_training_step, "freematch": MO.freematch_training_step}


def clone_aux(aux, dtype=None):
    out = {}
    for k, v in aux.items():
        if isinstance(v, list):
            out[k] = [t.clone() if dtype is None else t.to(dtype) for t in v]
        else:
            out[k] = v.clone() if dtype is None else v.to(dtype)
    return out


def build_case(name):
    kind, over, epoch, B, preset = CASES[name]
    hp = MO.default_hparams(**over)
    sd = INIT[kind](hp, seed=11)
    g = torch.Generator().manual_seed(12)
    stu, tea = STUDENT[kind], ("model.m_encoder." if kind == "comatch" else "model.ema.")
    for k in list(sd.keys()):                        # non-trivial BN statistics / affine parameters, a teacher that lags a little
        if not k.startswith(stu):
            continue
        v = sd[k]
        if k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7)
        elif ".bn" in k or "downsample.1" in k or "norm" in k:
            sd[k] = v + 0.1 * torch.randn(v.shape, generator=g)
        kt = tea + k[len(stu):]
        if sd[k].is_floating_point() and not k.endswith("running_var"):
            sd[kt] = sd[k] + 0.01 * torch.randn(sd[k].shape, generator=g) * (1.0 + sd[k].abs())
        else:
            sd[kt] = sd[k].clone()
    K, Dp, Q = hp.num_classes, hp.projection_dim, hp.K
    aux = {}
    if kind == "comatch" and preset:
        sd["model.queue_s"] = torch.nn.functional.normalize(torch.randn(Dp, Q, generator=g), dim=0)
        sd["model.queue_w"] = torch.nn.functional.normalize(torch.randn(Dp, Q, generator=g), dim=0)
        sd["model.probs_u"] = torch.softmax(torch.randn(K, Q, generator=g) * 2, dim=0)
        sd["model.probs_xu"] = torch.softmax(torch.randn(K, Q, generator=g) * 2, dim=0)
        sd["model.queue_ptr_s"] = torch.tensor([preset["ptr_s"]])
        sd["model.queue_ptr_w"] = torch.tensor([preset["ptr_w"]])
        aux["hist_prob"] = [torch.softmax(torch.randn(K, generator=g), dim=0) for _ in range(preset["hist"])]
    if kind == "simmatch":
        sd["model.labels"] = torch.randint(0, K, (Q,), generator=g)
        if hp.DA and preset and preset["da_rows"]:
            n = preset["da_rows"]
            sd["model.DA_queue"][:n] = torch.softmax(torch.randn(n, K, generator=g), dim=1)
            sd["model.DA_ptr"] = torch.tensor([n])
    batch = MO.synthetic_batch(hp, B, seed=51, views=3 if kind == "comatch" else 2)
    if kind == "freematch":
        aux.update(MO.freematch_aux(hp))
        if preset:   # a class-probability model / label histogram away from uniform; time_p is set below
            g2 = torch.Generator().manual_seed(preset["seed"])
            aux["p_model"] = torch.softmax(torch.randn(K, generator=g2), dim=0)
            aux["label_hist"] = torch.softmax(torch.randn(K, generator=g2), dim=0)
    # thresholds at the median of a dry run, so that the confidence mask / the pseudo-label graph are mixed
    dry = STEP[kind]({k: v.clone() for k, v in sd.items()}, batch, hp, epoch, clone_aux(aux))
    if kind == "freematch":
        if preset:
            with torch.no_grad():
                lw = MO.encoder_forward({k: v.clone() for k, v in sd.items()}, "model.ema.", batch["u"][0][0], hp, train=False)[0]
            mp_, mi_ = torch.softmax(lw, dim=-1).max(dim=-1)
            mod = dry["p_model"] / dry["p_model"].max()
            aux["time_p"] = (mp_ / mod[mi_]).median() - 1e-3
    elif kind == "comatch":
        hp.co_threshold = float(dry["probs"].max(dim=1).values.median()) - 1e-4
        Qm = dry["Q"]
        off = Qm[~torch.eye(Qm.shape[0], Qm.shape[1], dtype=torch.bool)]
        hp.contrast_th = float(off.quantile(0.7))
    else:
        hp.sim_threshold = float(dry["pseudo_label"].max(dim=1).values.median()) - 1e-4
    return kind, hp, sd, batch, epoch, aux


def run_reference(kind, hp, sd, batch, epoch, aux):
    if kind == "comatch":
        from models.MatchModel.CoMatch import CoMatch as Ref
    elif kind == "simmatch":
        from models.MatchModel.SimMatch import SimMatch as Ref
    else:
        from models.MatchModel.FreeMatchFolder.FreeMatch import FreeMatch as Ref
    with tempfile.TemporaryDirectory() as td:
        fl = os.path.join(td, "fl.pt")
        torch.save(list(hp.field_lengths), fl)
        model = Ref(G.ref_hparams(hp, fl))
    ref_keys = list(model.state_dict().keys())
    assert ref_keys == list(sd.keys()), f"state_dict keys/order differ: {sorted(set(ref_keys) ^ set(sd.keys()))[:10]} / {[(a, b) for a, b in zip(ref_keys, sd.keys()) if a != b][:4]}"
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.train()
    model.current_epoch = epoch
    model.model.use_ddp = False
    if kind == "comatch":
        model.model.hist_prob = [t.clone() for t in aux.get("hist_prob", [])]
    if kind == "freematch":
        model.model.p_model, model.model.label_hist, model.model.time_p = aux["p_model"].clone(), aux["label_hist"].clone(), aux["time_p"].clone()
    cap = {}
    fwd = model.model.forward

    def forward(*a, **k):
        out = fwd(*a, **k)
        if isinstance(out, tuple):
            cap["out"] = out
        return out

    model.model.forward = forward
    stu = STUDENT[kind]
    student = model.model.encoder if kind == "comatch" else model.model.main
    params = {stu + k: p for k, p in student.named_parameters()}
    opt = torch.optim.Adam([{"params": model.model.parameters()}], lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)
    opt.zero_grad()
    step_code = type(model).training_step.__code__
    orig_profile = sys.getprofile()

    def prof(frame, event, arg):
        if event == "return" and frame.f_code is step_code:
            for nm in LOCALS[kind]:
                cap["loc_" + nm] = torch.as_tensor(frame.f_locals[nm]).detach().clone()

    sys.setprofile(prof)
    try:
        loss = model.training_step(batch, 0)
    finally:
        sys.setprofile(orig_profile)
    loss.backward()
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in params.items()}
    for k, p in model.model.named_parameters():
        assert ("model." + k) in params or p.grad is None, k     # the momentum copy receives no gradient
    opt.step()
    out = {nm: cap["loc_" + nm] for nm in LOCALS[kind]}
    out["loss"] = loss.detach()
    o = cap["out"]
    if kind == "comatch":
        out.update(outputs_x=o[0].detach(), outputs_u_s0=o[1].detach(), probs=o[3].detach(), Q=o[4].detach(), sim=o[5].detach())
    elif kind == "simmatch":
        out.update(logits_x=o[0].detach(), pseudo_label=o[1].detach(), logits_u_s=o[2].detach())
    else:
        out.update(logits_x_lb=o[0].detach(), pseudo_label=o[1].detach(), logits_x_ulb_s=o[2].detach(), p_model=model.model.p_model.clone(),
                   label_hist=model.model.label_hist.clone(), time_p=model.model.time_p.clone())
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    hist = [t.clone() for t in model.model.hist_prob] if kind == "comatch" else None
    model.model.forward = fwd
    model.eval()
    x_l, y_l = batch["l"][0], batch["l"][1]
    with torch.no_grad():
        out["val_loss"] = model.validation_step((x_l, y_l), 0).detach()
    return out, grads, state, hist


def install_match_stubs():
    """models/MatchModel/multimodal_backbone.py imports einops (installed) and omegaconf (stubbed by make_golden)."""
    from oracle import make_golden_mmatch as GM
    GM.install_mmatch_stubs()


def main():
    sys.path.insert(0, G.REF)
    G.install_stubs()
    install_match_stubs()
    only = os.environ.get("ONLY")
    for name in CASES:
        if only and only not in name:
            continue
        kind, hp, sd, batch, epoch, aux = build_case(name)
        ref_out, ref_grads, ref_state, ref_hist = run_reference(kind, hp, {k: v.clone() for k, v in sd.items()}, batch, epoch, aux)
        sd_o = {k: v.clone() for k, v in sd.items()}
        aux_o = clone_aux(aux)
        o = MO.full_step(kind, sd_o, {}, 1, batch, hp, epoch, aux=aux_o)
        o["val_loss"] = torch.nn.functional.cross_entropy(MO.eval_logits(kind, sd_o, batch["l"][0], hp), batch["l"][1])
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
        tr = set(MO.trainable_keys(sd, STUDENT[kind]))
        for k, v in ref_state.items():
            if k in tr:
                if float((sd_o[k] - v).abs().max()) > 2.2 * hp.lr_eval:
                    bad.append(("adam:" + k, float((sd_o[k] - v).abs().max())))
            elif not G.close(sd_o[k].float(), v.float(), tol=2e-5):
                bad.append(("state:" + k, float((sd_o[k].float() - v.float()).abs().max())))
        if ref_hist is not None:
            assert len(ref_hist) == len(aux_o["hist_prob"]) and all(G.close(a, b) for a, b in zip(aux_o["hist_prob"], ref_hist)), "hist_prob"
        assert not bad, f"[{name}] oracle != reference: {bad[:8]}"
        # float64 yardstick for the gradients (see make_golden.py)
        d64 = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t  # noqa: E731
        sd64 = {k: d64(v.clone()) for k, v in sd.items()}
        cv = lambda x: tuple(d64(t) for t in x) if isinstance(x, (tuple, list)) else d64(x)  # noqa: E731
        b64 = {"l": (cv(batch["l"][0]), batch["l"][1], batch["l"][2]), "u": ([cv(v) for v in batch["u"][0]], batch["u"][1])}
        o64 = MO.full_step(kind, sd64, {}, 1, b64, hp, epoch, aux=clone_aux(aux, torch.float64))
        scalars, tensors = OUT[kind]
        fx = {"meta_epoch": np.int64(epoch)}
        for nm in ("co_threshold", "contrast_th", "sim_threshold"):
            fx["meta_" + nm] = np.float64(getattr(hp, nm))
        if kind == "freematch":
            fx["meta_time_p"] = aux["time_p"].numpy()
        for k in scalars + ["val_loss"]:
            fx["out_" + k] = ref_out[k].numpy().astype(np.float64)
        for k in tensors:
            fx["out_" + k] = ref_out[k].numpy()
        if ref_hist is not None:
            fx["hist_in"] = np.stack([t.numpy() for t in aux["hist_prob"]]) if aux.get("hist_prob") else np.zeros((0, hp.num_classes), np.float32)
            fx["hist_last"] = ref_hist[-1].numpy(); fx["hist_len"] = np.int64(len(ref_hist))
        for k, g in ref_grads.items():
            fx["gnorm_" + k] = np.float64(0.0 if g is None else g.double().norm().item())
            if g is not None:
                g64 = o64["grads"][k]
                fx["g64norm_" + k] = np.float64(g64.norm().item())
                fx["gerr32_" + k] = np.float64(((g.double() - g64).norm() / (g64.norm() + 1e-30)).item())
        stu = STUDENT[kind]
        first_conv = stu + ("encoder_imaging." if hp.eval_datatype == "imaging_and_tabular" else "backbone.") + "conv1.weight"
        for k in (stu + "head.2.weight", first_conv):
            if ref_grads.get(k) is not None:
                fx["grad_" + k] = ref_grads[k].numpy(); fx["grad64_" + k] = o64["grads"][k].numpy()
        for k, v in ref_state.items():
            if k in tr:
                continue
            short = k[len("model."):]
            if short in ("queue_s", "queue_w", "probs_u", "probs_xu", "queue_ptr_s", "queue_ptr_w", "bank", "labels", "DA_queue", "DA_ptr"):
                fx["state_" + k] = v.numpy()            # small: the memory banks / rings in full
            else:
                fx["ssum_" + k] = np.float64(v.double().sum().item()); fx["sabs_" + k] = np.float64(v.double().abs().sum().item())
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"{name}: oracle==reference OK  loss {float(ref_out['loss']):.6f}  mask {int(ref_out['mask'].sum())}/{len(ref_out['mask'])}"
              f"  -> {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
