"""The Lightning-free fit / test loops (stil_tta_amd/fit.py) end to end on a tiny synthetic task: checkpoint-on-best,
Lightning-shaped checkpoint contents, reload into a fresh module, resume, early stopping."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
FL = [3, 4] + [1] * 3


def _data(n, seed, K=3):
    """class-dependent images and columns so that a few steps already move the validation accuracy"""
    g = torch.Generator().manual_seed(seed)
    y = torch.randint(0, K, (n,), generator=g)
    img = torch.rand(n, 3, 64, 64, generator=g) * 0.2 + (y.float() / K)[:, None, None, None]
    tab = torch.cat([torch.randint(0, 3, (n, 1), generator=g).float(), (y % 4).float()[:, None], torch.randn(n, 3, generator=g) + y.float()[:, None]], 1)
    return img, tab, y


class _Pairs:
    """DataLoader stand-in yielding the reference's per-part batch tuple."""

    def __init__(self, img, tab, y, bs, labelled):
        self.img, self.tab, self.y, self.bs, self.lab = img, tab, y, bs, labelled

    def __len__(self):
        return len(self.y) // self.bs

    def __iter__(self):
        for i in range(len(self)):
            s = slice(i * self.bs, (i + 1) * self.bs)
            n = self.bs
            yield ([torch.zeros(n), self.img[s]], [self.tab[s], self.tab[s]], self.y[s], self.img[s], torch.full((n,), self.lab, dtype=torch.bool))


class _Val:
    def __init__(self, img, tab, y, bs):
        self.img, self.tab, self.y, self.bs = img, tab, y, bs

    def __len__(self):
        return (len(self.y) + self.bs - 1) // self.bs

    def __iter__(self):
        for i in range(len(self)):
            s = slice(i * self.bs, (i + 1) * self.bs)
            yield ((self.img[s], self.tab[s]), self.y[s])


def _model(**over):
    from stil_tta_amd import STiLModel
    torch.manual_seed(0)
    hp = dict(model="resnet18", embedding_dim=512, field_lengths=FL, num_classes=3, batch_size=16, target="dvm", start_epoch=0, th1=0.5,
              lr_eval=1e-3, warmup_epochs=2, max_epochs=6, mi_dropout=False)
    hp.update(over)
    return STiLModel(hp)


def test_fit_checkpoint_reload_resume_and_test(tmp_path):
    from stil_tta_amd import fit as F
    l_bs, u_bs = F.split_batch_size(16, 7)
    li, lt, ly = _data(8, 1)
    ui, ut, uy = _data(56, 2)
    vi, vt, vy = _data(40, 3)
    loaders = {"l": _Pairs(li, lt, ly, l_bs, True), "u": _Pairs(ui, ut, uy, u_bs, False)}
    val = _Val(vi, vt, vy, 16)
    m = _model(warmup_epochs=1)
    m.setup_device("cuda")
    m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(3, 128, generator=torch.Generator().manual_seed(5))).cuda())
    out = F.fit(m, loaders, val, max_epochs=4, eval_metric="acc", logdir=str(tmp_path), verbose=False)
    assert out["stopped"] == "max_epochs" and out["epochs_run"] == 4 and out["global_step"] == 4 * 4
    assert os.path.exists(out["checkpoint"]) and out["checkpoint"].endswith("checkpoint_best_acc.ckpt")
    assert abs(out["best_score"] - m.best_val_score) < 1e-7
    ck = torch.load(out["checkpoint"], map_location="cpu", weights_only=False)
    assert set(ck) >= {"epoch", "global_step", "state_dict", "hyper_parameters", "optimizer_states", "lr_schedulers", "pytorch-lightning_version"}
    assert ck["epoch"] == out["best_epoch"] and len(ck["optimizer_states"][0]["param_groups"]) == 6  # STiLModel.py:563-570
    assert any(k.startswith("ema.") for k in ck["state_dict"]) and "prototypes" in ck["state_dict"]
    # a fresh module + the checkpoint reproduces the best validation score exactly
    m2 = _model(warmup_epochs=1)
    m2.setup_device("cuda")
    F.load_checkpoint(out["checkpoint"], m2)
    got = F.validate(m2, val)
    assert abs(got["eval.val.acc"] - out["best_score"]) < 1e-7
    res = F.test(m2, val, ckpt_path=out["checkpoint"])
    assert abs(res["test.acc"] - out["best_score"]) < 1e-7 and 0.0 <= res["test.auc"] <= 1.0
    # resume continues the epoch / step counters and restores Adam's moments
    m3 = _model(warmup_epochs=1)
    out3 = F.fit(m3, loaders, val, max_epochs=ck["epoch"] + 2, logdir=str(tmp_path / "resumed"), resume_from=out["checkpoint"], verbose=False)
    assert out3["epochs_run"] == 1 and out3["global_step"] == ck["global_step"] + 4
    # the resumed epoch trains with the learning rate an uninterrupted run uses for that epoch (Lightning steps the
    # epoch-interval scheduler before validation / checkpointing: the checkpoint carries the NEXT epoch's rate)
    e3 = ck["epoch"] + 1
    m4 = _model(warmup_epochs=1)
    out4 = F.fit(m4, loaders, val, max_epochs=e3 + 1, logdir=None, verbose=False)
    assert list(out3["lr_by_epoch"]) == [e3] and out3["lr_by_epoch"][e3] == out4["lr_by_epoch"][e3]
    assert len(set(out4["lr_by_epoch"].values())) == len(out4["lr_by_epoch"]), "the schedule must move for this to be a test"
    st = ck["optimizer_states"][0]["state"]
    steps = [float(v["step"]) for v in st.values()]  # heads that only feed the pseudo-label losses start one epoch later
    assert len(st) > 100 and max(steps) == ck["global_step"] and min(steps) >= ck["global_step"] - 4


def test_early_stopping_ends_fit(tmp_path):
    from stil_tta_amd import fit as F
    li, lt, ly = _data(4, 1)
    ui, ut, uy = _data(28, 2)
    vi, vt, vy = _data(16, 3)
    m = _model(batch_size=16, lr_eval=0.0, scheduler="none")  # nothing can improve: stops after `patience` checks
    loaders = {"l": _Pairs(li, lt, ly, 2, True), "u": _Pairs(ui, ut, uy, 14, False)}
    out = F.fit(m, loaders, _Val(vi, vt, vy, 16), max_epochs=50, val_check_interval=50.0, logdir=None, verbose=False)  # patience int(100/50) = 2
    assert out["stopped"] == "early_stopping" and out["epochs_run"] == 3


def test_device_prefetcher_preserves_batches_and_order():
    """Pinned-buffer double buffering on the copy stream hands out exactly the producer's batches, in order, with the
    structure intact, also when staging slots are reused (more batches than slots) and shapes vary (last ragged batch)."""
    from stil_tta_amd.data import DevicePrefetcher
    g = torch.Generator().manual_seed(0)
    batches = []
    for i in range(7):
        b = 5 if i == 6 else 8
        batches.append({"l": ([torch.zeros(b), torch.rand(b, 3, 16, 16, generator=g)], [torch.rand(b, 4, generator=g)] * 2,
                              torch.randint(0, 3, (b,), generator=g), torch.rand(b, 3, 16, 16, generator=g), torch.ones(b, dtype=torch.bool)),
                        "idx": i})
    seen = 0
    for i, got in enumerate(DevicePrefetcher(batches, "cuda", depth=2)):
        ref = batches[i]
        assert got["idx"] == i
        im, tab, y, orig, ident = got["l"]
        assert im[1].is_cuda and tab[0].is_cuda and y.is_cuda and ident.dtype == torch.bool
        torch.cuda.current_stream().synchronize()
        assert torch.equal(im[1].cpu(), ref["l"][0][1]) and torch.equal(tab[1].cpu(), ref["l"][1][1])
        assert torch.equal(y.cpu(), ref["l"][2]) and torch.equal(orig.cpu(), ref["l"][3])
        seen += 1
    assert seen == 7


def test_optimizer_state_layout_equals_the_references_adam():
    """`optimizer_states[0]` of a checkpoint written by fit.save_checkpoint against torch.optim.Adam.state_dict() of the
    REFERENCE's own configure_optimizers (STiLModel.py:557-577) after one step, recorded by oracle/make_golden.py
    (tests/golden/adam_layout.npz): six parameter groups with the same sizes / ids / hyper-parameters, the same parameter
    shapes in id order, state for exactly the parameters that received a gradient, the same state keys and step counts."""
    import numpy as np
    from oracle.make_golden import build_case
    from stil_tta_amd import STiLModel, fit as F
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "adam_layout.npz"))
    hp, sd, batch, epoch, mask_random, mi_masks = build_case("dvm_r18_noeman")
    m = STiLModel(dict(vars(hp), mi_dropout=False))
    m.load_state_dict(sd)
    m.setup_device("cuda"); m.train(); m.current_epoch = epoch
    opt = StilAdam(m.flat, lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)
    dev_batch = {k: ([v[0][0].cuda(), v[0][1].cuda()], [v[1][0].cuda(), v[1][1].cuda()], v[2].cuda(), v[3].cuda(), v[4].cuda()) for k, v in batch.items()}
    train_step(m, opt, dev_batch, mask_random=mask_random)
    torch.cuda.synchronize()
    osd = F.adam_state_dict(m, opt)
    groups = osd["param_groups"]
    assert [len(g["params"]) for g in groups] == fx["group_sizes"].tolist()
    assert [i for g in groups for i in g["params"]] == fx["ids"].tolist()
    for gi, g in enumerate(groups):
        assert g["lr"] == float(fx["lr"][gi]) and g["weight_decay"] == float(fx["weight_decay"][gi])
        assert tuple(g["betas"]) == tuple(fx["betas"][gi]) and g["eps"] == float(fx["eps"][gi])
    order = [p for mod in (m.model, m.projector_imaging, m.projector_tabular, m.projector_multimodal, m.CLUB_imaging, m.CLUB_tabular)
             for p in mod.parameters()]
    assert [p.numel() for p in order] == fx["numel"].tolist()
    assert [d for p in order for d in p.shape] == fx["shapes"].tolist()
    names = {id(p): n for n, p in m.named_parameters()}
    assert [names[id(p)] for p in order] == fx["names"].tolist()
    has = np.array([i in osd["state"] for i in range(len(order))])
    assert np.array_equal(has, fx["has_state"]), [fx["names"][i] for i in np.nonzero(has != fx["has_state"])[0]]
    st = osd["state"][int(np.nonzero(has)[0][0])]
    assert sorted(st.keys()) == fx["state_keys"].tolist()
    assert [float(osd["state"][i]["step"]) if i in osd["state"] else 0.0 for i in range(len(order))] == fx["steps"].tolist()
    for i in np.nonzero(has)[0][:5]:
        assert tuple(osd["state"][int(i)]["exp_avg"].shape) == tuple(order[int(i)].shape)


class _Index:
    """DataLoader stand-in over a device batch builder: yields builder(indices) for consecutive index chunks."""

    def __init__(self, builder, bs):
        self.b, self.bs = builder, bs

    def __len__(self):
        return len(self.b) // self.bs

    def __iter__(self):
        for i in range(len(self)):
            yield self.b(torch.arange(i * self.bs, (i + 1) * self.bs))


@pytest.mark.parametrize("algo", ["SimMatch", "MMatch"])
def test_fit_and_checkpoint_of_the_baselines(tmp_path, algo):
    """fit -> best checkpoint (one Adam param_group over `model`, frozen momentum copy listed without state) -> reload -> test,
    for a Match baseline fed by the device batch builders and for MMatch fed by the contrastive builder."""
    import stil_tta_amd
    from stil_tta_amd import fit as F
    from stil_tta_amd.augment import ContrastiveBatchBuilder, EvalTrainBatchBuilder, StrongWeakBatchBuilder
    li, lt, ly = _data(8, 1)
    ui, ut, uy = _data(56, 2)
    vi, vt, vy = _data(40, 3)
    hp = dict(model="resnet18", embedding_dim=512, field_lengths=FL, num_classes=3, batch_size=16, target="dvm", start_epoch=0, lr_eval=1e-3,
              scheduler="anneal", warmup_epochs=1, max_epochs=6, img_size=64, DA=True)
    if algo == "SimMatch":
        loaders = {"l": _Index(EvalTrainBatchBuilder(li, lt, ly, 64, "dvm", 0.3, 0.8), 2),
                   "u": _Index(StrongWeakBatchBuilder(ui, ut, uy, 64, "dvm", 0.3), 14)}
        make = lambda: stil_tta_amd.SimMatch(dict(hp, K=8, sim_threshold=0.4))  # noqa: E731
    else:
        loaders = {"l": _Index(ContrastiveBatchBuilder(li, lt, ly, 64, "dvm", 0.3, 0.95, labelled=True), 2),
                   "u": _Index(ContrastiveBatchBuilder(ui, ut, uy, 64, "dvm", 0.3, 0.95, labelled=False), 14)}
        make = lambda: stil_tta_amd.MMatch(dict(hp, th1=0.4, alpha=1.0))  # noqa: E731
    val = _Val(vi, vt, vy, 16)
    torch.manual_seed(0)
    m = make()
    out = F.fit(m, loaders, val, max_epochs=3, eval_metric="acc", logdir=str(tmp_path), verbose=False, prefetch=False)
    assert out["stopped"] == "max_epochs" and out["global_step"] == 3 * 4 and os.path.exists(out["checkpoint"])
    ck = torch.load(out["checkpoint"], map_location="cpu", weights_only=False)
    groups = ck["optimizer_states"][0]["param_groups"]
    assert len(groups) == 1 and len(groups[0]["params"]) == len(list(m.model.parameters()))
    st = ck["optimizer_states"][0]["state"]
    n_train = sum(1 for p in m.model.parameters() if p.requires_grad)
    assert 50 < len(st) <= n_train and max(float(v["step"]) for v in st.values()) == ck["global_step"]
    torch.manual_seed(1)
    m2 = make()
    m2.setup_device("cuda")
    F.load_checkpoint(out["checkpoint"], m2, m2.configure_optimizers()["optimizer"])
    got = F.validate(m2, val)
    assert abs(got["eval.val.acc"] - out["best_score"]) < 1e-7
    res = F.test(m2, val, ckpt_path=out["checkpoint"])
    assert abs(res["test.acc"] - out["best_score"]) < 1e-7


@pytest.mark.parametrize("algo", ["FreeMatch", "CoMatch", "STiL"])
def test_create_model_semisl_loaders_and_fit(tmp_path, algo):
    """The reference's evaluate() flow (trainers/evaluate.py:50-83,142-213) on in-memory data: create_model + semisl_loaders
    (device batch builders, shuffled index loaders, repeat_ratio / K set on the hparams) + fit + test."""
    import stil_tta_amd
    from stil_tta_amd import fit as F
    from stil_tta_amd.augment import semisl_loaders
    li, lt, ly = _data(9, 1)
    ui, ut, uy = _data(56, 2)
    vi, vt, vy = _data(40, 3)
    hp = dict(algorithm_name=algo, model="resnet18", embedding_dim=512, field_lengths=FL, num_classes=3, batch_size=16, target="dvm",
              start_epoch=0, lr_eval=1e-3, scheduler="anneal", warmup_epochs=1, max_epochs=6, img_size=64, unlabelled_ratio=7, K=40,
              th1=0.5, mi_dropout=False, co_threshold=0.4, contrast_th=0.4)
    loaders = semisl_loaders(hp, (li, lt, ly), (ui, ut, uy))
    assert hp["repeat_ratio"] == 1 and len(loaders["l"]) == 5 and len(loaders["u"]) == 4      # 9 / 2 (last chunk of 1 kept), 56 / 14
    torch.manual_seed(0)
    m = stil_tta_amd.create_model(hp)
    if algo == "STiL":
        m.setup_device("cuda")
        m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(3, 128, generator=torch.Generator().manual_seed(5))).cuda())
    out = F.fit(m, loaders, _Val(vi, vt, vy, 16), max_epochs=2, eval_metric="acc", logdir=str(tmp_path), verbose=False, prefetch=False)
    assert out["stopped"] == "max_epochs" and out["global_step"] == 2 * 5 and os.path.exists(out["checkpoint"])   # max_size_cycle: 5 steps per epoch
    res = F.test(m, _Val(vi, vt, vy, 16), ckpt_path=out["checkpoint"])
    assert abs(res["test.acc"] - out["best_score"]) < 1e-7
