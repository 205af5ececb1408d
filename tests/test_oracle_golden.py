"""CPU: the oracle (oracle/stil_oracle.py) reproduces the golden vectors that
oracle/make_golden.py recorded from the REAL reference (STiLModel.training_step +
backward + Adam), for every case incl. injected dropout masks."""
import os

import numpy as np
import pytest
import torch

from oracle import stil_oracle as O
from oracle.make_golden import CASES, SCALARS, TENSORS, build_case

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _close(a, b, tol):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max() if b.size else 0.0
    return np.all(np.abs(a - b) <= tol * (1.0 + np.abs(b) + scale))


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    hp, sd, batch, epoch, mask_random, mi_masks = build_case(name)
    assert int(fx["meta_epoch"]) == epoch
    out = O.full_step(sd, {}, 1, batch, hp, epoch, mask_random, mi_masks)
    for k in SCALARS:
        assert _close(out[k].numpy(), fx["out_" + k], 2e-5), k
    for k in TENSORS:
        a, b = out[k].numpy(), fx["out_" + k]
        if b.dtype == np.bool_:
            assert np.array_equal(a, b), k
        else:
            assert _close(a, b, 2e-5), k
    for key in fx.files:
        if key.startswith("gnorm_"):
            g = out["grads"].get(key[6:])
            n = 0.0 if g is None else float(g.double().norm())
            assert abs(n - float(fx[key])) <= 1e-4 * (1e-6 + float(fx[key])) + 1e-7, key
        elif key.startswith("grad_"):
            assert _close(out["grads"][key[5:]].numpy(), fx[key], 5e-5), key
        elif key.startswith("ssum_"):
            v = sd[key[5:]].double()
            ref_abs = float(fx["sabs_" + key[5:]])
            assert abs(float(v.sum()) - float(fx[key])) <= 2e-5 * (1.0 + ref_abs), key


def test_oracle_trajectory_matches_reference_golden():
    """Five consecutive optimisation steps of the oracle against the trajectory the REAL reference took
    (tests/golden/traj_r18.npz, oracle/make_golden_traj.py): every loss term of every step, the logged ratios exactly, the
    final state on the stored strided sample.  Oracle and reference share ATen's CPU kernels, so the bar is tight; the device
    test (tests/test_gpu_step.py) uses the float64 yardsticks stored beside the reference's values."""
    from oracle import make_golden_traj as T
    fx = np.load(os.path.join(GOLD, "traj_r18.npz"))
    hp, sd, batches, masks = T.build()
    steps, state = T.run_oracle(hp, sd, batches, masks)
    for s_ in range(T.STEPS):
        for k, v in steps[s_].items():
            if "ref_" + k not in fx.files:      # loss_pt is a local of the reference's step, not a logged scalar
                continue
            ref = float(fx["ref_" + k][s_])
            assert abs(v - ref) <= (1e-6 if k.endswith("_ratio") else 2e-5 * (1 + abs(ref))), (s_, k, v, ref)
    assert any(0 < r < 1 for r in fx["ref_mask1_ratio"])
    for key in fx.files:
        if key.startswith("state/") and state[key[6:]].is_floating_point():
            got, ref = T.sample(state[key[6:]]).double().numpy(), fx[key].astype(np.float64)
            err = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
            # a weight whose gradient is rounding noise moves +-lr per Adam step: thread-count-dependent reductions may flip it
            assert err <= 3 * max(float(fx["dist64/" + key[6:]]), float(fx["distp/" + key[6:]])) + 1e-4, (key, err)


def test_epoch_end_commits_prototypes():
    hp = O.default_hparams(num_classes=3, projection_dim=4)
    sd = {"prototypes": torch.zeros(3, 4), "prototypes_sum": torch.arange(12.0).reshape(3, 4),
          "prototypes_count_sum": torch.tensor([[1.0], [2.0], [4.0]])}
    O.training_epoch_end(sd)
    assert torch.allclose(sd["prototypes"][2], torch.tensor([2.0, 2.25, 2.5, 2.75]))
    assert float(sd["prototypes_sum"].abs().sum()) == 0.0
    sd["prototypes_count_sum"][1] = 0.0
    with pytest.raises(AssertionError):  # STiLModel.py:412
        O.training_epoch_end(sd)


def test_state_dict_layout_matches_reference_appendix_a():
    hp = O.default_hparams()
    sd = O.init_state(hp)
    assert len(sd) == 831
    assert sum(1 for k in sd if k.startswith("model.")) == 406
    assert sum(1 for k in sd if k.startswith("ema.")) == 406
    n_train = sum(sd[k].numel() for k in O.trainable_keys(sd))
    assert abs(n_train - 46.72e6) < 0.02e6


# ---------------------------------------------------------------- metrics oracle pinned against scikit-learn
def test_metrics_oracle_against_sklearn():
    import numpy as np
    from sklearn.metrics import roc_auc_score, top_k_accuracy_score
    from oracle import metrics_oracle as MO
    rng = np.random.default_rng(0)
    # binary, with heavy ties
    s = np.round(rng.random(500), 1).astype(np.float32)
    y = rng.integers(0, 2, 500)
    assert abs(MO.binary_auroc(s, y == 1) - roc_auc_score(y, s)) < 1e-12
    assert MO.binary_accuracy(s, y) == float(((s > 0.5) == (y == 1)).mean())
    # multiclass one-vs-rest macro (every class present), softmax rows
    K = 7
    z = rng.normal(size=(400, K)).astype(np.float32)
    p = np.exp(z) / np.exp(z).sum(1, keepdims=True)
    yk = np.concatenate([np.arange(K), rng.integers(0, K, 400 - K)])
    macro, per = MO.multiclass_auroc(p, yk)
    assert abs(macro - roc_auc_score(yk, p.astype(np.float64) / p.astype(np.float64).sum(1, keepdims=True), multi_class="ovr", average="macro")) < 1e-6
    for c in range(K):
        assert abs(per[c] - roc_auc_score(yk == c, p[:, c])) < 1e-12
    # top-k (no ties in continuous scores)
    for k in (1, 5):
        assert abs(MO.topk_accuracy(z, yk, k) - top_k_accuracy_score(yk, z, k=k, labels=np.arange(K))) < 1e-12
    # degenerate class: no positives -> 0 (torchmetrics 0.11.0 rule, unpinned)
    assert MO.binary_auroc(s, np.zeros(500, bool)) == 0.0


# ---------------------------------------------------------------- MMatch baseline
from oracle import mmatch_oracle as MO  # noqa: E402
from oracle import make_golden_mmatch as GM  # noqa: E402


@pytest.mark.parametrize("name", list(GM.CASES))
def test_mmatch_oracle_matches_reference_golden(name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    hp, sd, batch, epoch = GM.build_case(name)
    hp.th1 = float(fx["meta_th1"])  # the data-dependent threshold of the generating machine (CPU rounding differs across hosts)
    sd0 = {k: v.clone() for k, v in sd.items()}
    out = MO.full_step(sd, {}, 1, batch, hp, epoch)
    for k in GM.SCALARS:
        assert _close(out[k].numpy(), fx["out_" + k], 2e-5), k
    for k in GM.TENSORS:
        a, b = out[k].numpy(), fx["out_" + k]
        assert (np.array_equal(a, b) if b.dtype in (np.bool_, np.int64) else _close(a, b, 2e-5)), k
    for key in fx.files:
        if key.startswith("gnorm_"):
            g = out["grads"].get(key[6:])
            n = 0.0 if g is None else float(g.double().norm())
            assert abs(n - float(fx[key])) <= 1e-4 * (1e-6 + float(fx[key])) + 1e-7, key
        elif key.startswith("ssum_"):
            assert abs(float(sd[key[5:]].double().sum()) - float(fx[key])) <= 2e-5 * (1.0 + float(fx["sabs_" + key[5:]])), key
    p0 = int(sd0["embed_queue_ptr"])
    assert _close(sd["embed_queue"][:, p0:p0 + 16].numpy(), fx["state_embed_queue_cols"], 2e-5)
    assert _close(sd["probs_queue"].numpy(), fx["state_probs_queue"], 2e-5)
    assert int(sd["embed_queue_ptr"].item()) == int(fx["state_embed_queue_ptr"].item()) and int(sd["DA_ptr"].item()) == int(fx["state_DA_ptr"].item())
    assert _close(sd["DA_queue"][int(sd0["DA_ptr"])].numpy(), fx["state_DA_queue_row"], 2e-5)


def test_mmatch_state_dict_layout():
    from stil_tta_amd import MMatch
    hp = MO.default_hparams(model="resnet18", embedding_dim=512, field_lengths=[3, 4, 1, 1, 1], num_classes=5, batch_size=16)
    sd = MO.init_state(hp, seed=0)
    m = MMatch(dict(vars(hp)))
    got = m.state_dict()
    assert list(got.keys()) == list(sd.keys())
    assert all(tuple(got[k].shape) == tuple(sd[k].shape) for k in sd)


@pytest.mark.parametrize("name", list(GM.CO_CASES))
def test_cotraining_oracle_matches_reference_golden(name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    hp, sd, batch, epoch = GM.build_co_case(name)
    hp.co_threshold = float(fx["meta_co_threshold"])
    out = MO.cotrain_full_step(sd, {}, 1, batch, hp, epoch)
    for k in GM.CO_SCALARS:
        assert _close(out[k].numpy(), fx["out_" + k], 2e-5), k
    for k in GM.CO_TENSORS:
        a, b = out[k].numpy(), fx["out_" + k]
        assert (np.array_equal(a, b) if b.dtype == np.bool_ else _close(a, b, 2e-5)), k
    for key in fx.files:
        if key.startswith("ssum_"):
            assert abs(float(sd[key[5:]].double().sum()) - float(fx[key])) <= 2e-5 * (1.0 + float(fx["sabs_" + key[5:]])), key


@pytest.mark.parametrize("name", list(GM.COS_CASES))
def test_cotraining_saint_oracle_matches_reference_golden(name):
    """CoTraining_SAINT.py: incl. the teacher's int64 offset buffers as the reference's EMA leaves them (29 -> 28 under eman)."""
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    hp, sd, batch, epoch, masks = GM.build_cos_case(name)
    hp.co_threshold = float(fx["meta_co_threshold"])
    out = MO.cotrain_saint_full_step(sd, {}, 1, batch, hp, epoch, masks)
    for k in GM.CO_SCALARS:
        assert _close(out[k].numpy(), fx["out_" + k], 2e-5), k
    for k in GM.CO_TENSORS:
        a, b = out[k].numpy(), fx["out_" + k]
        assert (np.array_equal(a, b) if b.dtype == np.bool_ else _close(a, b, 2e-5)), k
    for key in fx.files:
        if key.startswith("ssum_"):
            assert abs(float(sd[key[5:]].double().sum()) - float(fx[key])) <= 2e-5 * (1.0 + float(fx["sabs_" + key[5:]])), key
        elif key.startswith("state_"):
            assert np.array_equal(sd[key[6:]].numpy(), fx[key]), key
    if hp.eman:
        assert sd["ema.encoder_tabular.categories_offset"].tolist() == [0, 1, 4, 28] and sd["model.encoder_tabular.categories_offset"].tolist() == [0, 1, 4, 29]


def test_cotraining_saint_state_dict_layout():
    from stil_tta_amd import CoTraining
    hp = MO.cotrain_saint_hparams(model="resnet18", embedding_dim=512, field_lengths=GM.FLS, num_classes=5, batch_size=16)
    sd = MO.cotrain_saint_init_state(hp, seed=0)
    got = CoTraining(dict(vars(hp))).state_dict()
    assert list(got.keys()) == list(sd.keys())
    assert all(tuple(got[k].shape) == tuple(sd[k].shape) and got[k].dtype == sd[k].dtype for k in sd)


def test_tab_corrupt_restatement_matches_reference_golden():
    """datasets/ContrastiveImagingAndTabularDataset.py:146-158: the restated `corrupt` (draws made explicit) against the
    outputs of the reference's own method under the same draws (tests/golden/tab_corrupt.npz, oracle/make_golden_data.py)."""
    import numpy as np
    from oracle.make_golden_data import corrupt_oracle
    fx = np.load(os.path.join(GOLD, "tab_corrupt.npz"))
    table = fx["table"]
    for tag in ("c030", "c000", "c100", "c005"):
        out, idx, pos = fx[tag + "_out"], fx[tag + "_idx"], fx[tag + "_pos"]
        assert idx.shape[1] == int(table.shape[1] * int(tag[1:]) / 100)
        for s in range(len(out)):
            assert len(set(idx[s].tolist())) == idx.shape[1]        # random.sample: distinct columns
            assert np.array_equal(corrupt_oracle(table[s % len(table)], table.T, idx[s], pos[s]), out[s])


# ---------------------------------------------------------------- CoMatch / SimMatch baselines
from oracle import match_oracle as XO  # noqa: E402
from oracle import make_golden_match as GX  # noqa: E402


def _match_case(name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    kind, hp, sd, batch, epoch, aux = GX.build_case(name)
    for nm in ("co_threshold", "contrast_th", "sim_threshold"):   # data-dependent thresholds of the generating machine
        setattr(hp, nm, float(fx["meta_" + nm]))
    if kind == "freematch":
        aux["time_p"] = torch.tensor(fx["meta_time_p"])
    return fx, kind, hp, sd, batch, epoch, aux


@pytest.mark.parametrize("name", list(GX.CASES))
def test_match_oracle_matches_reference_golden(name):
    fx, kind, hp, sd, batch, epoch, aux = _match_case(name)
    out = XO.full_step(kind, sd, {}, 1, batch, hp, epoch, aux=aux)
    scalars, tensors = GX.OUT[kind]
    for k in scalars:
        assert _close(out[k].numpy(), fx["out_" + k], 2e-5), k
    for k in tensors:
        assert _close(out[k].numpy(), fx["out_" + k], 2e-5), k
    for key in fx.files:
        if key.startswith("gnorm_"):
            g = out["grads"].get(key[6:])
            n = 0.0 if g is None else float(g.double().norm())
            assert abs(n - float(fx[key])) <= 1e-4 * (1e-6 + float(fx[key])) + 1e-7, key
        elif key.startswith("ssum_"):
            assert abs(float(sd[key[5:]].double().sum()) - float(fx[key])) <= 2e-5 * (1.0 + float(fx["sabs_" + key[5:]])), key
        elif key.startswith("state_"):
            a, b = sd[key[6:]].numpy(), fx[key]
            assert (np.array_equal(a, b) if b.dtype == np.int64 else _close(a, b, 2e-5)), key
    if kind == "comatch":
        assert len(aux["hist_prob"]) == int(fx["hist_len"]) and _close(aux["hist_prob"][-1].numpy(), fx["hist_last"], 2e-5)


@pytest.mark.parametrize("name", ["comatch_r18_bank", "comatch_r18_img_binary", "simmatch_r18_bank", "simmatch_r18_img_noDA", "freematch_r18_mask",
                                  "freematch_r18_img_binary"])
def test_match_state_dict_layout(name):
    import stil_tta_amd
    kind, over, _, _, _ = GX.CASES[name]
    hp = XO.default_hparams(**over)
    sd = GX.INIT[kind](hp, seed=0)
    m = {"comatch": stil_tta_amd.CoMatch, "simmatch": stil_tta_amd.SimMatch, "freematch": stil_tta_amd.FreeMatch}[kind](dict(vars(hp)))
    got = m.state_dict()
    assert list(got.keys()) == list(sd.keys())
    assert all(tuple(got[k].shape) == tuple(sd[k].shape) for k in sd)
