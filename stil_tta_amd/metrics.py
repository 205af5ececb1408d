"""Device-side evaluation metrics with the torchmetrics==0.11.0 call surface the reference uses
(`metric(preds, target)`, `.compute()`, `.reset()`; models/Disentangle/STiLModel.py:122-145, 360-363, 458-463, 529-545).

Accuracy keeps two int64 counters on the device (hits, samples); AUROC keeps the epoch's scores and computes the exact
(thresholds=None) area with integer rank statistics in `stil_auroc`.  Under data parallelism `.compute()` reduces the
counters / gathers the scores over the process group (torchmetrics' sync-on-compute).  No host sync until `.compute()`
is read.  There is no CPU path: the kernels live in libstil_hip.so.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from ._lib import lib
from .ops import _p, _stream, _ws


def _dist_on() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _prep(preds: torch.Tensor, target: torch.Tensor):
    if not preds.is_cuda:
        raise RuntimeError("stil_tta_amd.metrics needs CUDA(HIP) tensors: there is no CPU fallback")
    return preds.detach().to(torch.float32).contiguous(), target.detach().to(device=preds.device, dtype=torch.int64).contiguous()


class Accuracy:
    """torchmetrics.Accuracy(task='multiclass', top_k=k, num_classes=K) (micro average) or task='binary' (threshold 0.5)."""

    def __init__(self, task: str = "multiclass", num_classes: Optional[int] = None, top_k: int = 1, threshold: float = 0.5):
        assert task in ("binary", "multiclass")
        self.task, self.num_classes, self.top_k, self.threshold = task, num_classes, int(top_k), float(threshold)
        self.counts: Optional[torch.Tensor] = None

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        p, y = _prep(preds, target)
        if self.counts is None:
            self.counts = torch.zeros(2, dtype=torch.int64, device=p.device)
        if self.task == "binary":
            if p.dim() != 1 or p.shape[0] != y.shape[0]:
                raise RuntimeError(f"binary accuracy expects preds [N] and target [N], got {tuple(p.shape)} / {tuple(y.shape)}")
            lib().metric_binary(_p(p), _p(y), p.shape[0], self.threshold, _p(self.counts), _stream())
        else:
            if p.dim() != 2 or p.shape[0] != y.shape[0]:
                raise RuntimeError(f"multiclass accuracy expects preds [N, K] and target [N], got {tuple(p.shape)} / {tuple(y.shape)}")
            lib().metric_topk(_p(p), p.shape[1], _p(y), p.shape[0], p.shape[1], self.top_k, _p(self.counts), _stream())

    __call__ = update

    def compute(self) -> torch.Tensor:
        if self.counts is None:
            raise RuntimeError("Accuracy.compute() before any update")
        c = self.counts.clone()
        if _dist_on():
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
        return c[0].to(torch.float32) / c[1].to(torch.float32)

    def reset(self) -> None:
        if self.counts is not None:
            self.counts.zero_()


class AUROC:
    """torchmetrics.AUROC(task='binary') / (task='multiclass', num_classes=K, average='macro'), exact mode."""

    def __init__(self, task: str = "multiclass", num_classes: Optional[int] = None):
        assert task in ("binary", "multiclass")
        self.task, self.num_classes = task, num_classes
        self.preds: List[torch.Tensor] = []
        self.target: List[torch.Tensor] = []
        self.per_class: Optional[torch.Tensor] = None

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        p, y = _prep(preds, target)
        if self.task == "binary":
            p = p.reshape(-1, 1)
        elif p.dim() != 2 or (self.num_classes is not None and p.shape[1] != self.num_classes):
            raise RuntimeError(f"multiclass AUROC expects preds [N, {self.num_classes}], got {tuple(p.shape)}")
        if p.shape[0] != y.shape[0]:
            raise RuntimeError("AUROC: preds / target length mismatch")
        self.preds.append(p)
        self.target.append(y)

    __call__ = update

    def _gathered(self):
        p, y = torch.cat(self.preds), torch.cat(self.target)
        if not _dist_on():
            return p, y
        n = torch.tensor([p.shape[0]], dtype=torch.int64, device=p.device)
        sizes = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
        dist.all_gather(sizes, n)
        nmax = int(max(int(s) for s in sizes))
        pp = torch.zeros((nmax, p.shape[1]), dtype=p.dtype, device=p.device)
        yy = torch.zeros((nmax,), dtype=y.dtype, device=p.device)
        pp[: p.shape[0]], yy[: y.shape[0]] = p, y
        ps = [torch.empty_like(pp) for _ in sizes]
        ys = [torch.empty_like(yy) for _ in sizes]
        dist.all_gather(ps, pp)
        dist.all_gather(ys, yy)
        return (torch.cat([t[: int(s)] for t, s in zip(ps, sizes)]), torch.cat([t[: int(s)] for t, s in zip(ys, sizes)]))

    def compute(self) -> torch.Tensor:
        if not self.preds:
            raise RuntimeError("AUROC.compute() before any update")
        p, y = self._gathered()
        N, K = p.shape
        nb = lib().auroc_workspace_bytes(N, K)
        if nb == 0:
            raise RuntimeError(f"AUROC: {N} x {K} scores exceed the 2^31-element limit")
        ws = _ws.get(nb, p.device)
        per = torch.empty(K, dtype=torch.float32, device=p.device)
        macro = torch.empty(1, dtype=torch.float32, device=p.device)
        lib().auroc(_p(p), K, _p(y), N, K, _p(per), _p(macro), _p(ws), nb, _stream())
        self.per_class = per
        return macro[0]

    def reset(self) -> None:
        self.preds, self.target = [], []
