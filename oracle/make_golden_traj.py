"""Pin a short TRAJECTORY against the REAL reference and write tests/golden/traj_r18.npz (round-3 verdict item 8).

Runs only in the build container (needs /root/reference on disk); test infrastructure, never imported by the product.
The single-step goldens (make_golden.py) cannot see a slow drift: a change of reduction order that is "within rounding"
on one step could accumulate over steps unnoticed.  Here the reference `STiLModel` (models/Disentangle/STiLModel.py:228-386
training_step, :557-570 Adam) takes FIVE consecutive optimisation steps (zero_grad -> training_step -> backward ->
Adam.step, the loop Lightning runs, trainers/evaluate.py:178-179) on five different seeded batches with fixed
`mask_random` draws, in the pseudo-label phase (epoch > start_epoch, prototypes pre-filled, crafted heads so that the CGPL
cases are mixed); the oracle's `full_step` runs the same five steps; the generator asserts oracle == reference on every
step's loss terms and on the final state, and stores the REFERENCE's per-step losses, its final state (every tensor of the
small ResNet-18 case) and, per tensor, two YARDSTICKS for what correct evaluations of this trajectory differ by (oracle-held): the float64 oracle's distance from
the reference, and the spread of the fp32 oracle under ONE-ULP perturbations of the initial parameters (NPERT runs).  Adam's
update is lr * m / sqrt(v) ~ lr * sign(g) in the first steps whatever the gradient's size, so an element whose gradient
changes sign under rounding lands 2 lr away (1 % of a typical weight): measured here, one ulp in the initial weights moves
the total loss by ~1e-4 relative within five steps.  A device test can therefore pin the trajectory to a few times that
spread -- tight enough to catch mistakes in what is carried ACROSS steps (Adam moments and bias correction, the EMA teacher,
BatchNorm running statistics, prototype sums), which single-step parity cannot see.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_traj.py
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import stil_oracle as O  # noqa: E402
from oracle.make_golden import REF, SCALARS, craft_heads, install_stubs, randomize_state, ref_hparams  # noqa: E402

STEPS = 5
HP = dict(model="resnet18", embedding_dim=512, img_size=64, num_classes=5, field_lengths=[3, 4] + [1] * 3, batch_size=8,
          th1=0.25, start_epoch=1, lr_eval=1e-4)
EPOCH = 3
TOL_STEP = 2e-5
B = 8
NPERT = 4            # one-ulp perturbation runs of the fp32 oracle (sensitivity yardstick)
SAMPLE = 256        # elements kept per tensor of the final state (strided): the whole ResNet-18 state would be 120 MB


def build():
    hp = O.default_hparams(**HP)
    sd = randomize_state(O.init_state(hp, seed=4321), seed=77)
    g = torch.Generator().manual_seed(8)
    sd["prototypes"] = torch.nn.functional.normalize(torch.randn(hp.num_classes, hp.projection_dim, generator=g))
    batches = [O.synthetic_batch(hp, B, seed=3000 + s) for s in range(STEPS)]
    masks = [torch.rand(B - max(B // 8, 1), generator=g).ge(0.5) for _ in range(STEPS)]
    sd = craft_heads(sd, batches[0], hp, EPOCH, masks[0])
    return hp, sd, batches, masks


def run_reference(hp, sd, batches, masks):
    from models.Disentangle.STiLModel import STiLModel
    with tempfile.TemporaryDirectory() as td:
        fl = os.path.join(td, "fl.pt")
        torch.save(list(hp.field_lengths), fl)
        model = STiLModel(ref_hparams(hp, fl))
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.train()
    model.current_epoch = EPOCH
    orig_dropout_fwd, orig_rand_like = nn.Dropout.forward, torch.rand_like
    import models.Disentangle.utils.disentangle_transformer as DT
    orig_drop_path = DT.drop_path
    cur = {}

    def rand_like(t, **kw):
        m = cur["mask"]
        assert t.shape == m.shape
        return torch.where(m, torch.full_like(t, 0.75), torch.full_like(t, 0.25))

    nn.Dropout.forward = lambda self, x: x                     # MI-layer dropout off for this case (as mi_masks=None in make_golden)
    DT.drop_path = lambda x, drop_prob=0.0, training=False: x
    torch.rand_like = rand_like
    per_step = []
    try:
        opt = torch.optim.Adam([
            {"params": model.model.parameters()}, {"params": model.projector_imaging.parameters()},
            {"params": model.projector_tabular.parameters()}, {"params": model.projector_multimodal.parameters()},
            {"params": model.CLUB_imaging.parameters()}, {"params": model.CLUB_tabular.parameters()}],
            lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)  # STiLModel.py:563-570
        for s in range(STEPS):
            cur["mask"] = masks[s]
            opt.zero_grad()
            loss = model.training_step(batches[s], s)
            loss.backward()
            opt.step()
            L = model.logged
            per_step.append(dict(loss=float(loss.detach()), loss_ce=float(L["multimodal.train.CEloss"]), loss_itc=float(L["multimodal.train.ITCloss"]),
                                 loss_club_i=float(L["multimodal.train.CLUBloss_imaging"]), loss_club_i_est=float(L["multimodal.train.CLUBloss_imaging_est"]),
                                 loss_club_t=float(L["multimodal.train.CLUBloss_tabular"]), loss_club_t_est=float(L["multimodal.train.CLUBloss_tabular_est"]),
                                 loss_m_u=float(L["multimodal.train.CEloss_unlabelled_m"]), loss_i_u=float(L["multimodal.train.CEloss_unlabelled_i"]),
                                 loss_t_u=float(L["multimodal.train.CEloss_unlabelled_t"]),
                                 mask1_ratio=float(L["multimodal.train.threshold1_ratio"]), case1_ratio=float(L["multimodal.train.case1_ratio"]),
                                 case3_ratio=float(L["multimodal.train.case3_ratio"])))
    finally:
        nn.Dropout.forward, DT.drop_path, torch.rand_like = orig_dropout_fwd, orig_drop_path, orig_rand_like
    return per_step, {k: v.detach().clone() for k, v in model.state_dict().items()}


def run_oracle(hp, sd, batches, masks, dtype=torch.float32):
    s = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    opt, per_step = {}, []
    for i in range(STEPS):
        b = batches[i]
        if dtype != torch.float32:
            b = {k: ([v[0][0].to(dtype), v[0][1].to(dtype)], [v[1][0].to(dtype), v[1][1].to(dtype)], v[2], v[3].to(dtype), v[4]) for k, v in b.items()}
        o = O.full_step(s, opt, i + 1, b, hp, EPOCH, masks[i], None)
        d = {k: float(o[k]) for k in SCALARS if k in o}
        d.update(mask1_ratio=float(o["mask1"].float().mean()), case1_ratio=float(o["case1"].float().mean()), case3_ratio=float(o["case3"].float().mean()))
        per_step.append(d)
    return per_step, s


def main():
    sys.path.insert(0, REF)
    install_stubs()
    torch.manual_seed(0)
    hp, sd, batches, masks = build()
    ref_steps, ref_state = run_reference(hp, {k: v.clone() for k, v in sd.items()}, batches, masks)
    ora_steps, ora_state = run_oracle(hp, sd, batches, masks)
    o64_steps, o64_state = run_oracle(hp, sd, batches, masks, torch.float64)
    keys = [k for k in ref_steps[0] if k in ora_steps[0]]
    # sensitivity yardstick: the fp32 oracle from initial parameters perturbed by one ulp (relative 6e-8, seeded)
    pert_runs = []
    for trial in range(NPERT):
        g = torch.Generator().manual_seed(500 + trial)
        sd_p = {k: (v * (1 + 6e-8 * (torch.rand(v.shape, generator=g) * 2 - 1)) if (v.is_floating_point() and not k.startswith("prototypes")
                    and not k.endswith("running_var") and not k.endswith("running_mean")) else v.clone()) for k, v in sd.items()}
        for k in list(sd_p):
            if k.startswith("model."):
                sd_p["ema." + k[6:]] = sd_p[k].clone() if sd_p[k].is_floating_point() else sd_p["ema." + k[6:]]
        pert_runs.append(run_oracle(hp, sd_p, batches, masks))
    worst = 0.0
    for s in range(STEPS):
        for k in keys:
            a, b = ora_steps[s][k], ref_steps[s][k]
            d = abs(a - b) / (1.0 + abs(b))
            worst = max(worst, d)
            if os.environ.get("TRAJ_DEBUG"):
                print(s, k, a, b, o64_steps[s][k], f"{d:.2e}")
            else:
                assert d <= TOL_STEP, (s, k, a, b)
    assert any(0 < r["mask1_ratio"] < 1 for r in ref_steps), "mask1 should be mixed somewhere on the trajectory"
    tr = set(O.trainable_keys(sd))
    out = {"steps": np.int64(STEPS), "epoch": np.int64(EPOCH), "B": np.int64(B), "sample": np.int64(SAMPLE)}
    dist, d64 = {}, {}
    for k, v in ref_state.items():
        o = ora_state[k]
        if not v.is_floating_point():
            assert torch.equal(v, o), k
            out["state/" + k] = v.numpy()
            continue
        if k in tr:
            moved = float((v - sd[k]).abs().max())
            assert moved <= 1.2 * hp.lr_eval * STEPS + 1e-7, (k, moved)   # |Adam update| <= ~lr per step
        r, a, h = sample(v).double(), sample(o).double(), sample(o64_state[k])
        dist[k] = float((a - r).norm() / (r.norm() + 1e-30))
        d64[k] = float((h - r).norm() / (r.norm() + 1e-30))
        # reference-held: strided sample, sum and L2 norm of the tensor; oracle-held yardsticks: the float64 oracle's distance
        # from the reference on the same sample and on the sum
        out["state/" + k] = sample(v).numpy()
        out["sum/" + k] = np.float64(v.double().sum())
        out["norm/" + k] = np.float64(v.double().norm())
        out["dist64/" + k] = np.float64(d64[k])
        out["distp/" + k] = np.float64(max(float((sample(ps[k]).double() - a).norm() / (a.norm() + 1e-30)) for _, ps in pert_runs))
        out["dsump/" + k] = np.float64(max(abs(float(ps[k].double().sum()) - float(o.double().sum())) for _, ps in pert_runs))
        out["dsum64/" + k] = np.float64(abs(float(o64_state[k].sum()) - float(v.double().sum())))
    print(f"oracle == reference over {STEPS} steps: worst scaled loss-term distance {worst:.2e}; "
          f"final-state relL2 (sampled) oracle vs reference: median {np.median(list(dist.values())):.2e}, max {max(dist.values()):.2e} "
          f"({max(dist, key=dist.get)}); float64 oracle vs reference: median {np.median(list(d64.values())):.2e}, max {max(d64.values()):.2e}")
    for k in keys:
        out["ref_" + k] = np.array([r[k] for r in ref_steps], dtype=np.float64)          # reference-held
        out["o64_" + k] = np.array([r[k] for r in o64_steps], dtype=np.float64)          # the float64 oracle's value: a yardstick
        out["sens_" + k] = np.array([max(abs(pr[s_][k] - ora_steps[s_][k]) for pr, _ in pert_runs) for s_ in range(STEPS)], dtype=np.float64)
    path = os.path.join(ROOT, "tests", "golden", "traj_r18.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; oracle==reference OK")


def sample(t):
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // SAMPLE)][:SAMPLE].clone()


if __name__ == "__main__":
    main()
