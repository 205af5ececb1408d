"""Fit / test loops for STiLModel without Lightning: what `trainers/evaluate.py:93-219` gets from
`pytorch_lightning.Trainer` (1.6.4) for this module, restated for one process per GPU.

  * train loader dict {'l', 'u'} combined in `max_size_cycle` mode (Lightning's default for a dict of loaders in fit:
    an epoch lasts as long as the longest loader, shorter ones restart) -- trainers/evaluate.py:116-119;
  * `split_batch_size` / `repeat_ratio` of trainers/evaluate.py:82-85;
  * per epoch: train steps -> training_epoch_end -> epoch-interval LR scheduler step (Lightning 1.6 steps it in
    TrainingEpochLoop.on_advance_end, BEFORE validation and checkpointing, so a checkpoint carries the learning rate of
    the NEXT epoch) -> (every check_val_every_n_epoch) validation -> validation_epoch_end ->
    ModelCheckpoint(monitor eval.val.<metric>, mode max, checkpoint_best_<metric>.ckpt) -> EarlyStopping(min_delta
    1e-4, patience int(scale / val_check_interval)) -> ReduceLROnPlateau step, which needs the validation metric
    (trainers/evaluate.py:170-179);
  * checkpoints are Lightning-shaped dicts (`state_dict` with the reference's key names, `hyper_parameters`,
    `optimizer_states` in torch.optim.Adam layout over the reference's six parameter groups, `lr_schedulers`, `epoch`,
    `global_step`), so a reference checkpoint loads here and ours loads there.
Data-parallel: every rank runs the same loop on its own shard (samplers are the caller's business, as with Lightning);
rank 0 writes checkpoints; metrics are already reduced over ranks inside `.compute()`.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Iterable, Iterator, Optional

import torch
import torch.distributed as dist

from .driver import sync_buffers, train_step, world_size


# ------------------------------------------------------------------------------------------ host-side helpers
def split_batch_size(batch_size: int, unlabelled_ratio: int):
    """(labelled, unlabelled) per-step batch sizes -- trainers/evaluate.py:84-85."""
    l_bs = batch_size // (1 + unlabelled_ratio)
    return l_bs, batch_size - l_bs


def repeat_ratio(u_N: int, l_N: int, unlabelled_ratio: int) -> int:
    """How often a labelled sample recurs per unlabelled epoch (prototype weighting) -- trainers/evaluate.py:83."""
    return max(u_N // (unlabelled_ratio * l_N) - 1, 1)


def max_size_cycle(loaders: Dict[str, Iterable]) -> Iterator[Dict[str, object]]:
    """Lightning CombinedLoader(mode='max_size_cycle'): one epoch = len(longest loader) steps; exhausted shorter
    loaders are restarted (a fresh iterator, so a shuffling loader reshuffles)."""
    lens = {k: len(v) for k, v in loaders.items()}
    its = {k: iter(v) for k, v in loaders.items()}
    for _ in range(max(lens.values())):
        out = {}
        for k in loaders:
            try:
                out[k] = next(its[k])
            except StopIteration:
                its[k] = iter(loaders[k])
                out[k] = next(its[k])
        yield out


class BestCheckpoint:
    """ModelCheckpoint(monitor=..., mode='max', save_top_k=1): keeps the best-so-far score and says when to save."""

    def __init__(self, monitor: str, dirpath: Optional[str], filename: str):
        self.monitor, self.dirpath, self.filename = monitor, dirpath, filename
        self.best = -math.inf
        self.best_epoch = -1

    @property
    def path(self) -> Optional[str]:
        return None if self.dirpath is None else os.path.join(self.dirpath, self.filename + ".ckpt")

    def improved(self, value: float, epoch: int) -> bool:
        if value > self.best:
            self.best, self.best_epoch = value, epoch
            return True
        return False


class EarlyStopping:
    """pytorch_lightning EarlyStopping(mode='max', min_delta, patience): stop once `patience` consecutive checks
    failed to beat the best score by more than min_delta (wait_count >= patience)."""

    def __init__(self, min_delta: float = 1e-4, patience: int = 100):
        self.min_delta, self.patience = float(min_delta), int(patience)
        self.best = -math.inf
        self.wait = 0

    def should_stop(self, value: float) -> bool:
        if value - self.min_delta > self.best:
            self.best, self.wait = value, 0
            return False
        self.wait += 1
        return self.wait >= self.patience


# ------------------------------------------------------------------------------------------ checkpoint I/O
def adam_state_dict(model, optimizer) -> dict:
    """torch.optim.Adam.state_dict() layout over the reference's parameter groups (model.optimizer_groups(): six for STiL,
    STiLModel.py:563-570; `[self.model]` for the baselines, e.g. MMatch.py:385-387), read out of the flat slabs."""
    flat = model.flat
    groups_of = model.optimizer_groups()
    g = optimizer.param_groups[0]
    steps = flat.steps.cpu()
    state, groups, pid = {}, [], 0
    base = {k: v for k, v in g.items() if k != "params"}
    off = {id(t): i for i, t in enumerate(flat.tensors)}
    for mod in groups_of:
        ids = []
        for p in mod.parameters():
            i = off.get(id(p))          # None: a frozen copy inside the group (the baselines' momentum encoders): listed, no state
            o = p.data_ptr() - flat.params.data_ptr()
            o //= 4
            n = p.numel()
            if i is not None and int(steps[i]) > 0:
                state[pid] = {"step": torch.tensor(float(steps[i])), "exp_avg": flat.exp_avg[o:o + n].view(p.shape).cpu().clone(),
                              "exp_avg_sq": flat.exp_avg_sq[o:o + n].view(p.shape).cpu().clone()}
            ids.append(pid)
            pid += 1
        groups.append(dict(base, params=ids))
    return {"state": state, "param_groups": groups}


def load_adam_state_dict(model, optimizer, sd: dict) -> None:
    flat = model.flat
    order = [p for mod in model.optimizer_groups() for p in mod.parameters()]
    off = {id(t): i for i, t in enumerate(flat.tensors)}
    steps = torch.zeros_like(flat.steps, device="cpu")
    flat.exp_avg.zero_()
    flat.exp_avg_sq.zero_()
    for pid, st in sd["state"].items():
        p = order[int(pid)]
        if id(p) not in off:
            # a frozen copy inside the group (the baselines' momentum encoders): torch's Adam never creates state for a
            # parameter without gradient, so an entry here means the checkpoint belongs to another parameter list
            raise KeyError(f"optimizer state entry {pid} refers to a parameter outside the gradient slab (a frozen copy): "
                           "this checkpoint was not written for this model's optimizer groups")
        o = (p.data_ptr() - flat.params.data_ptr()) // 4
        n = p.numel()
        flat.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
        flat.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
        steps[off[id(p)]] = int(float(st["step"]))
    flat.steps.copy_(steps)
    if sd.get("param_groups"):
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in sd["param_groups"][0]:
                optimizer.param_groups[0][k] = sd["param_groups"][0][k]


def save_checkpoint(path: str, model, optimizer=None, scheduler=None, epoch: int = 0, global_step: int = 0, extra: Optional[dict] = None):
    torch.cuda.synchronize()
    ck = {"epoch": epoch, "global_step": global_step, "pytorch-lightning_version": "1.6.4",
          "state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
          "hyper_parameters": {k: v for k, v in vars(model.hp).items()},
          "optimizer_states": [adam_state_dict(model, optimizer)] if optimizer is not None else [],
          "lr_schedulers": [scheduler.state_dict()] if scheduler is not None else []}
    ck.update(extra or {})
    tmp = path + ".tmp"
    torch.save(ck, tmp)
    os.replace(tmp, path)
    return path


def load_checkpoint(path: str, model, optimizer=None, scheduler=None) -> dict:
    ck = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ck["state_dict"], strict=True)
    if optimizer is not None and ck.get("optimizer_states"):
        load_adam_state_dict(model, optimizer, ck["optimizer_states"][0])
    if scheduler is not None and ck.get("lr_schedulers"):
        scheduler.load_state_dict(ck["lr_schedulers"][0])
    return ck


# ------------------------------------------------------------------------------------------ loops
def _to_device(obj, dev):
    if torch.is_tensor(obj):
        return obj.to(dev, non_blocking=True)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(o, dev) for o in obj)
    if isinstance(obj, dict):
        return {k: _to_device(v, dev) for k, v in obj.items()}
    return obj


def validate(model, val_loader, limit_batches: Optional[int] = None) -> Dict[str, float]:
    dev = model.prototypes.device
    was_training = model.training
    model.eval()
    sync_buffers(model)
    for i, batch in enumerate(val_loader):
        if limit_batches is not None and i >= limit_batches:
            break
        model.validation_step(_to_device(batch, dev), i)
    model.validation_epoch_end()
    model.train(was_training)
    return {k: float(v) for k, v in model.logged.items() if k.startswith("eval.val.")} if hasattr(model, "logged") else {}


def fit(model, train_loaders: Dict[str, Iterable], val_loader: Optional[Iterable] = None, *, max_epochs: Optional[int] = None,
        eval_metric: str = "acc", logdir: Optional[str] = None, check_val_every_n_epoch: int = 1, val_check_interval: float = 1.0,
        sweep: bool = False, limit_train_batches: Optional[int] = None, limit_val_batches: Optional[int] = None,
        resume_from: Optional[str] = None, verbose: bool = True, prefetch: bool = True) -> dict:
    """Trainer.fit(model, {'l','u'}, val_loader) of trainers/evaluate.py:178-179.  Returns the run summary
    (best score / epoch, checkpoint path, last callback metrics, why it stopped)."""
    model.setup_device()
    dev = model.prototypes.device
    conf = model.configure_optimizers()
    opt, sched = conf["optimizer"], conf.get("lr_scheduler")
    max_epochs = int(model.hp.max_epochs if max_epochs is None else max_epochs)
    rank = dist.get_rank() if world_size() > 1 else 0
    ckpt = BestCheckpoint(f"eval.val.{eval_metric}", logdir, f"checkpoint_best_{eval_metric}")
    stopper = EarlyStopping(1e-4, int((40 if sweep else 100) * (1 / val_check_interval)))
    if logdir is not None and rank == 0:
        os.makedirs(logdir, exist_ok=True)
    start_epoch, gstep = 0, 0
    if resume_from is not None:
        ck = load_checkpoint(resume_from, model, opt, sched)
        start_epoch, gstep = int(ck["epoch"]) + 1, int(ck["global_step"])
        ckpt.best, ckpt.best_epoch = float(ck.get("best_score", -math.inf)), int(ck.get("best_epoch", -1))
        stopper.best, stopper.wait = float(ck.get("stopper_best", -math.inf)), int(ck.get("stopper_wait", 0))
    stopped = "max_epochs"
    last_val: Dict[str, float] = {}
    lrs: Dict[int, float] = {}
    plateau = isinstance(sched, torch.optim.lr_scheduler.ReduceLROnPlateau)  # scheduler: linear (monitors the validation metric)
    for epoch in range(start_epoch, max_epochs):
        model.train()
        model.current_epoch = epoch
        lrs[epoch] = float(opt.param_groups[0]["lr"])
        stream = max_size_cycle(train_loaders)
        if prefetch:  # host batches: pinned staging + H2D copies on a copy stream, two batches ahead (data.DevicePrefetcher)
            from .data import DevicePrefetcher
            stream = DevicePrefetcher(stream, dev, depth=2)
        for i, batch in enumerate(stream):
            if limit_train_batches is not None and i >= limit_train_batches:
                break
            train_step(model, opt, _to_device(batch, dev))
            gstep += 1
        model.training_epoch_end()
        if sched is not None and not plateau:
            sched.step()
        if val_loader is not None and (epoch + 1) % check_val_every_n_epoch == 0:
            last_val = validate(model, val_loader, limit_val_batches)
            score = last_val[ckpt.monitor]
            extra = dict(best_score=max(ckpt.best, score), best_epoch=ckpt.best_epoch, stopper_best=stopper.best, stopper_wait=stopper.wait)
            if ckpt.improved(score, epoch) and ckpt.path is not None and rank == 0:
                extra.update(best_score=ckpt.best, best_epoch=epoch)
                save_checkpoint(ckpt.path, model, opt, sched, epoch, gstep, extra)
            if verbose and rank == 0:
                print(f"epoch {epoch}: {ckpt.monitor} {score:.6f} (best {ckpt.best:.6f} @ {ckpt.best_epoch}), lr {opt.param_groups[0]['lr']:.3e}")
            if stopper.should_stop(score):
                stopped = "early_stopping"
                break
        if plateau and last_val:
            sched.step(last_val[ckpt.monitor])
    if world_size() > 1:
        dist.barrier()
    return dict(best_score=ckpt.best, best_epoch=ckpt.best_epoch, checkpoint=ckpt.path, epochs_run=epoch + 1 - start_epoch if max_epochs > start_epoch else 0,
                global_step=gstep, stopped=stopped, callback_metrics=last_val, best_val_score=model.best_val_score, lr_by_epoch=lrs)


def test(model, test_loader, ckpt_path: Optional[str] = None, limit_batches: Optional[int] = None) -> Dict[str, float]:
    """Trainer.test(model, loader, ckpt_path=best) of trainers/evaluate.py:206-213 / trainers/test.py:85-90."""
    model.setup_device()
    if ckpt_path is not None:
        load_checkpoint(ckpt_path, model)
    model.freeze()
    dev = model.prototypes.device
    model.acc_test.reset()
    model.auc_test.reset()
    for i, batch in enumerate(test_loader):
        if limit_batches is not None and i >= limit_batches:
            break
        model.test_step(_to_device(batch, dev), i)
    return {k: float(v) for k, v in model.test_epoch_end().items()}
