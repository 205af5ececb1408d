"""GPU parity of the MMatch baseline (stil_tta_amd/mmatch.py) against the golden vectors recorded from the REAL reference
(models/SemiMultimodal/MMatch.py, oracle/make_golden_mmatch.py): forward quantities, pseudo-labels and masks, losses,
gradients (float64 yardstick, as in test_gpu_step.py), memory bank / DA queue / BN buffers after the step."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

from oracle import make_golden_mmatch as GM  # noqa: E402
from oracle import mmatch_oracle as MO  # noqa: E402
from test_gpu_step import _close, _to_dev  # noqa: E402


@pytest.mark.parametrize("name", list(GM.CASES))
def test_mmatch_training_step_matches_reference_golden(name):
    from stil_tta_amd import MMatch
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    hp, sd, batch, epoch = GM.build_case(name)
    hp.th1 = float(fx["meta_th1"])  # the data-dependent threshold of the generating machine (CPU rounding differs across hosts)
    m = MMatch(dict(vars(hp)))
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m.setup_device("cuda"); m.train(); m.current_epoch = epoch
    train_step(m, StilAdam(m.flat, lr=hp.lr_eval), _to_dev(batch))
    torch.cuda.synchronize()
    bad = []
    for k in GM.SCALARS:
        ok, err = _close(m.last[k].detach().cpu().numpy(), fx["out_" + k])
        if not ok:
            bad.append((k, err))
    for k in ("y_hat_m", "y_hat_i", "y_hat_t", "x_m", "pseudo_label", "pseudo_label_orig"):
        ok, err = _close(m.last[k].detach().cpu().numpy(), fx["out_" + k])
        if not ok:
            bad.append((k, err))
    assert np.array_equal(m.last["mask1"].cpu().numpy() > 0.5, fx["out_mask1"]), "confidence mask"
    assert np.array_equal(m.last["hard_idx"].cpu().numpy().astype(np.int64), fx["out_hard_idx"]), "hard labels"
    params = dict(m.named_parameters())
    ratios = []
    for key in fx.files:
        if not key.startswith("gnorm_"):
            continue
        pname = key[6:]
        p = params[pname]
        if float(fx[key]) == 0.0 and "g64norm_" + pname not in fx.files:
            assert not p._stil_touched, pname
            continue
        if "grad64_" + pname in fx.files:
            g64 = fx["grad64_" + pname].astype(np.float64)
            eg = np.linalg.norm(p._gslot.cpu().double().numpy() - g64) / (np.linalg.norm(g64) + 1e-30)
            e32 = float(fx["gerr32_" + pname])
            ratios.append(eg / (3 * e32 + 1e-4))
            if eg > 3 * e32 + 1e-2:
                bad.append(("grad " + pname, eg, e32))
        else:  # norms only: within the reference's own distance from the fp64 truth (+ floor)
            n = float(p._gslot.double().norm())
            n64, e32 = float(fx["g64norm_" + pname]), float(fx["gerr32_" + pname])
            if abs(n - n64) > (3 * e32 + 1e-2) * n64 + 1e-7:
                bad.append(("gnorm " + pname, n, n64))
    msd = m.state_dict()
    p0 = int(sd["embed_queue_ptr"])
    for got, ref, what in ((msd["embed_queue"][:, p0:p0 + 16], fx["state_embed_queue_cols"], "embed_queue"), (msd["probs_queue"], fx["state_probs_queue"], "probs_queue"),
                           (msd["DA_queue"][int(sd["DA_ptr"])], fx["state_DA_queue_row"], "DA_queue")):
        ok, err = _close(got.cpu().numpy(), ref, 5e-5)
        if not ok:
            bad.append((what, err))
    assert int(msd["embed_queue_ptr"].item()) == int(fx["state_embed_queue_ptr"].item()) and int(msd["DA_ptr"].item()) == int(fx["state_DA_ptr"].item())
    for key in fx.files:
        if key.startswith("ssum_"):
            v = msd[key[5:]].double()
            if abs(float(v.sum()) - float(fx[key])) > 5e-5 * (1.0 + float(fx["sabs_" + key[5:]])):
                bad.append(("state " + key[5:], float(v.sum()), float(fx[key])))
    assert not bad, f"{len(bad)} mismatches, first: {bad[:8]}"
    # validation hook on the post-step weights of the REFERENCE (Adam noise excluded): load the oracle's post-step state
    sd_o = {k: v.clone() for k, v in sd.items()}
    MO.full_step(sd_o, {}, 1, batch, hp, epoch)
    m.load_state_dict(sd_o)
    m.eval()
    x = (torch.cat((batch["l"][0][1], batch["u"][0][1])).cuda(), torch.cat((batch["l"][1][1], batch["u"][1][1])).cuda())
    y = torch.cat((batch["l"][2], batch["u"][2])).cuda()
    ok, err = _close(m.validation_step((x, y)).cpu().numpy(), fx["out_val_loss"])
    assert ok, ("val_loss", err)


@pytest.mark.parametrize("name", list(GM.CO_CASES))
def test_cotraining_step_matches_reference_golden(name):
    """CoTrain_Pseudo baseline (models/SemiMultimodal/CoTraining.py): EMA (state_dict or parameters only) / no EMA."""
    from stil_tta_amd import CoTraining
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    hp, sd, batch, epoch = GM.build_co_case(name)
    hp.co_threshold = float(fx["meta_co_threshold"])  # data-dependent threshold of the generating machine
    m = CoTraining(dict(vars(hp)))
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m.setup_device("cuda"); m.train(); m.current_epoch = epoch
    train_step(m, StilAdam(m.flat, lr=hp.lr_eval), _to_dev(batch))
    torch.cuda.synchronize()
    bad = []
    for k in GM.CO_SCALARS + ["y_hat_m", "y_hat_i", "y_hat_t", "y_hat_i_e", "y_hat_t_e", "pseudo_label_i", "pseudo_label_t"]:
        ok, err = _close(m.last[k].detach().cpu().numpy(), fx["out_" + k])
        if not ok:
            bad.append((k, err))
    assert np.array_equal(m.last["mask_i"].cpu().numpy() > 0.5, fx["out_mask_i"]) and np.array_equal(m.last["mask_t"].cpu().numpy() > 0.5, fx["out_mask_t"])
    params = dict(m.named_parameters())
    for key in fx.files:
        if not key.startswith("gnorm_"):
            continue
        pname = key[6:]
        p = params[pname]
        if "g64norm_" + pname not in fx.files:
            assert not p._stil_touched, pname
            continue
        n64, e32 = float(fx["g64norm_" + pname]), float(fx["gerr32_" + pname])
        if "grad64_" + pname in fx.files:
            g64 = fx["grad64_" + pname].astype(np.float64)
            eg = np.linalg.norm(p._gslot.cpu().double().numpy() - g64) / (np.linalg.norm(g64) + 1e-30)
            if eg > 3 * e32 + 1e-2:
                bad.append(("grad " + pname, eg, e32))
        elif abs(float(p._gslot.double().norm()) - n64) > (3 * e32 + 1e-2) * n64 + 1e-7:
            bad.append(("gnorm " + pname, float(p._gslot.double().norm()), n64))
    msd = m.state_dict()
    for key in fx.files:
        if key.startswith("ssum_"):
            v = msd[key[5:]].double()
            if abs(float(v.sum()) - float(fx[key])) > 5e-5 * (1.0 + float(fx["sabs_" + key[5:]])):
                bad.append(("state " + key[5:], float(v.sum()), float(fx[key])))
    assert not bad, f"{len(bad)} mismatches, first: {bad[:8]}"


@pytest.mark.parametrize("name", list(GM.COS_CASES))
def test_cotraining_saint_step_matches_reference_golden(name):
    """CoTrain_Pseudo_SAINT (models/SemiMultimodal/CoTraining_SAINT.py): SAINT tabular encoder with injected feed-forward
    dropout masks; under eman the teacher's int64 offset buffers must end up as the reference's EMA leaves them (29 -> 28)."""
    from stil_tta_amd import CoTraining
    from stil_tta_amd.flat import StilAdam
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    hp, sd, batch, epoch, masks = GM.build_cos_case(name)
    hp.co_threshold = float(fx["meta_co_threshold"])
    m = CoTraining(dict(vars(hp)))
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m.setup_device("cuda"); m.train(); m.current_epoch = epoch
    opt = StilAdam(m.flat, lr=hp.lr_eval)
    opt.zero_grad()
    loss = m.training_step(_to_dev(batch), 0, saint_masks=masks)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    bad = []
    for k in GM.CO_SCALARS + ["y_hat_m", "y_hat_i", "y_hat_t", "y_hat_i_e", "y_hat_t_e", "pseudo_label_i", "pseudo_label_t"]:
        ok, err = _close(m.last[k].detach().cpu().numpy(), fx["out_" + k])
        if not ok:
            bad.append((k, err))
    assert np.array_equal(m.last["mask_i"].cpu().numpy() > 0.5, fx["out_mask_i"]) and np.array_equal(m.last["mask_t"].cpu().numpy() > 0.5, fx["out_mask_t"])
    params = dict(m.named_parameters())
    for key in fx.files:
        if not key.startswith("gnorm_"):
            continue
        pname = key[6:]
        p = params[pname]
        if "g64norm_" + pname not in fx.files:
            assert not p._stil_touched, pname
            continue
        n64, e32 = float(fx["g64norm_" + pname]), float(fx["gerr32_" + pname])
        if abs(float(p._gslot.double().norm()) - n64) > (3 * e32 + 1e-2) * n64 + 1e-7:
            bad.append(("gnorm " + pname, float(p._gslot.double().norm()), n64))
    msd = m.state_dict()
    for key in fx.files:
        if key.startswith("ssum_"):
            v = msd[key[5:]].double()
            if abs(float(v.sum()) - float(fx[key])) > 5e-5 * (1.0 + float(fx["sabs_" + key[5:]])):
                bad.append(("state " + key[5:], float(v.sum()), float(fx[key])))
        elif key.startswith("state_"):
            assert np.array_equal(msd[key[6:]].cpu().numpy(), fx[key]), (key, msd[key[6:]].tolist(), fx[key].tolist())
    assert not bad, f"{len(bad)} mismatches, first: {bad[:8]}"
