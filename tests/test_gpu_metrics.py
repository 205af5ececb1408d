"""Device metrics (stil_tta_amd/metrics.py, csrc/metrics.hip) against oracle/metrics_oracle.py: counters are exact, the
AUROC is integer rank statistics on the same fp32 scores -> equal to the oracle to the final float division."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _softmax(z):
    e = np.exp(z - z.max(1, keepdims=True))
    return (e / e.sum(1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("N,K,k", [(1, 3, 1), (37, 5, 1), (256, 256, 5), (1000, 286, 5), (513, 2, 1)])
def test_topk_accuracy_counts(N, K, k):
    from oracle import metrics_oracle as MO
    from stil_tta_amd.metrics import Accuracy
    rng = np.random.default_rng(N + K)
    z = np.round(rng.normal(size=(N, K)), 1).astype(np.float32)  # rounded: ties occur
    y = rng.integers(0, K, N)
    m = Accuracy("multiclass", K, top_k=k)
    half = N // 2
    m(torch.from_numpy(z[:half]).cuda(), torch.from_numpy(y[:half]).cuda())
    m(torch.from_numpy(z[half:]).cuda(), torch.from_numpy(y[half:]).cuda())
    assert int(m.counts[1]) == N
    assert int(m.counts[0]) == round(MO.topk_accuracy(z, y, k) * N)
    assert abs(float(m.compute()) - MO.topk_accuracy(z, y, k)) < 1e-6
    m.reset()
    assert int(m.counts.sum()) == 0


def test_binary_accuracy_and_auroc_with_ties():
    from oracle import metrics_oracle as MO
    from stil_tta_amd.metrics import AUROC, Accuracy
    rng = np.random.default_rng(5)
    for N in (2, 300, 5000):
        s = np.round(rng.random(N), 2).astype(np.float32)
        y = rng.integers(0, 2, N)
        y[0], y[1] = 0, 1
        acc, auc = Accuracy("binary", 2), AUROC("binary", 2)
        for a, b in ((0, N // 3), (N // 3, N)):
            acc(torch.from_numpy(s[a:b]).cuda(), torch.from_numpy(y[a:b]).cuda())
            auc(torch.from_numpy(s[a:b]).cuda(), torch.from_numpy(y[a:b]).cuda())
        assert abs(float(acc.compute()) - MO.binary_accuracy(s, y)) < 1e-6
        assert abs(float(auc.compute()) - MO.binary_auroc(s, y == 1)) < 1e-6


@pytest.mark.parametrize("N,K", [(50, 3), (777, 13), (4096, 286), (20000, 286)])
def test_multiclass_auroc(N, K):
    from oracle import metrics_oracle as MO
    from stil_tta_amd.metrics import AUROC
    rng = np.random.default_rng(N)
    p = _softmax(rng.normal(size=(N, K)).astype(np.float32) * 2)
    y = rng.integers(0, K, N)
    if N > 3 * K:
        y[y == K - 1] = 0  # one absent class: contributes 0 to the macro mean
    m = AUROC("multiclass", K)
    step = max(1, N // 3)
    for a in range(0, N, step):
        m(torch.from_numpy(p[a:a + step]).cuda(), torch.from_numpy(y[a:a + step]).cuda())
    got = float(m.compute())
    macro, per = MO.multiclass_auroc(p, y)
    assert np.abs(m.per_class.cpu().numpy().astype(np.float64) - per).max() < 1e-6
    assert abs(got - macro) < 1e-6


def test_metric_errors():
    from stil_tta_amd._lib import lib
    from stil_tta_amd.metrics import AUROC, Accuracy
    with pytest.raises(RuntimeError):
        Accuracy("multiclass", 3)(torch.zeros(4, 3), torch.zeros(4, dtype=torch.int64))  # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        AUROC("multiclass", 3).compute()
    with pytest.raises(RuntimeError):  # workspace too small is reported, not overrun
        x = torch.zeros(8, 2, device="cuda")
        y = torch.zeros(8, dtype=torch.int64, device="cuda")
        o = torch.zeros(3, device="cuda")
        lib().auroc(x.data_ptr(), 2, y.data_ptr(), 8, 2, o.data_ptr(), o[2:].data_ptr(), x.data_ptr(), 16, None)


def test_model_eval_hooks_track_oracle_metrics():
    """validation_step / validation_epoch_end / test_step / test_epoch_end on two batches: the epoch metrics equal the
    oracle metrics of the scores the module itself produced (test_step returns them), best_val_score follows acc."""
    from oracle import metrics_oracle as MO
    from stil_tta_amd import STiLModel
    fl = [3, 4] + [1] * 3
    torch.manual_seed(0)
    m = STiLModel(dict(model="resnet18", embedding_dim=512, field_lengths=fl, num_classes=5, batch_size=16, target="dvm"))
    m.setup_device("cuda").freeze()
    assert not any(p.requires_grad for p in m.parameters()) and not m.training
    g = torch.Generator().manual_seed(1)
    probs, ys = [], []
    for b in (16, 9):
        x = (torch.rand(b, 3, 64, 64, generator=g).cuda(), torch.cat([torch.randint(0, 3, (b, 2), generator=g).float(), torch.randn(b, 3, generator=g)], 1).cuda())
        y = torch.randint(0, 5, (b,), generator=g).cuda()
        m.validation_step((x, y))
        probs.append(m.test_step((x, y)).cpu().numpy())
        ys.append(y.cpu().numpy())
    P, Y = np.concatenate(probs), np.concatenate(ys)
    assert int(m.top1_acc_val.counts[1]) == 16  # retrieval accuracy only on the full batch (STiLModel.py:437)
    m.validation_epoch_end()
    out = m.test_epoch_end()
    assert abs(float(out["test.acc"]) - MO.topk_accuracy(P, Y, 1)) < 1e-6
    assert abs(float(out["test.auc"]) - MO.multiclass_auroc(P, Y)[0]) < 1e-6
    assert abs(float(m.logged["eval.val.acc"]) - MO.topk_accuracy(P, Y, 1)) < 1e-6
    assert abs(m.best_val_score - MO.topk_accuracy(P, Y, 1)) < 1e-6
    assert int(m.acc_val.counts.sum()) == 0
