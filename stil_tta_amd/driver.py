"""Minimal step driver: what Lightning's automatic optimisation does around STiLModel.training_step
(trainers/evaluate.py:178-179: zero_grad -> training_step -> backward -> optimizer.step), plus the
data-parallel gradient exchange (one RCCL all-reduce of the flat gradient slab over xGMI) and the
reference's LR schedule.  One process per GPU; torch.distributed (backend "nccl" == RCCL on ROCm).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.distributed as dist


def anneal_lambda(warmup_epochs: int, max_epochs: int):
    """pl_bolts LinearWarmupCosineAnnealingLR(warmup_start_lr=0, eta_min=0) as a LambdaLR factor (STiLModel.py:583)."""

    def f(epoch: int) -> float:
        if epoch < warmup_epochs:
            return epoch / max(1, warmup_epochs - 1)
        return 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / max(1, max_epochs - warmup_epochs)))

    return f


def host_cpu_share() -> int:
    """The host cores this job may actually use: min(scheduler affinity, cgroup CPU quota).  The GPU boxes expose every logical
    CPU of the node (os.cpu_count() = 256) under a cgroup quota of 16 cores per GPU: an OpenMP pool sized by cpu_count runs
    4.4x SLOWER there than one sized by the quota (measured on a CPU training step: 14.0 s at 128 threads, 3.2 s at 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def pick_device(backend: str, local: int, ndev: int, env=None) -> int:
    """The device index of this rank, or a RuntimeError that names the cause when two ranks of the RCCL backend would share a
    device (RCCL refuses that much later, with an opaque message).  Launch styles accepted under "nccl":
      * one process per GPU of the node, every GPU visible to every rank (torch.distributed.run, bench.py --gpus N): LOCAL_RANK
        -> GPU; LOCAL_RANK must be < the visible devices and, when the launcher exports LOCAL_WORLD_SIZE, so must be the devices;
      * ONE visible GPU per rank (SLURM --gpus-per-task=1, per-rank HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES masks): device 0,
        whatever LOCAL_RANK is -- recognised by a visibility variable in the rank's environment, or by a launcher that exports no
        LOCAL_WORLD_SIZE (srun, mpirun);
      * multi-node launchers that do not export LOCAL_WORLD_SIZE: only LOCAL_RANK < devices is checked (WORLD_SIZE counts the
        ranks of every node and says nothing about this one).
    gloo ranks may share a device (the one-GPU test box): LOCAL_RANK modulo the device count."""
    env = os.environ if env is None else env
    if ndev <= 0:
        return -1
    if backend != "nccl":
        return local % ndev
    lws = env.get("LOCAL_WORLD_SIZE")
    lws = int(lws) if lws not in (None, "") else None
    masked = any(env.get(v) not in (None, "") for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    if ndev == 1:
        if lws is not None and lws > 1 and not masked:
            raise RuntimeError(f"init_distributed: {lws} local ranks on the RCCL backend but ONE visible GPU and no per-rank visibility "
                               f"mask (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES): the ranks would share the device "
                               f"(LOCAL_RANK={local}); one process per GPU")
        return 0
    if local >= ndev:
        raise RuntimeError(f"init_distributed: LOCAL_RANK={local} on the RCCL backend needs more than the {ndev} visible GPUs; "
                           f"one process per GPU, ranks never share a device")
    if lws is not None and ndev < lws:
        raise RuntimeError(f"init_distributed: {lws} local rank(s) on the RCCL backend need {lws} visible GPUs, found {ndev} "
                           f"(LOCAL_RANK={local}); one process per GPU, ranks never share a device")
    return local


def init_distributed(backend: Optional[str] = None, timeout_s: Optional[float] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run). Returns (rank, world, local_rank).
    `timeout_s` (default STIL_DIST_TIMEOUT_S or 900): the process group's timeout -- a rank that never reaches the rendezvous
    or a collective ends the job after that long instead of waiting for the job's outer limit (gloo raises in the waiting ranks;
    under RCCL the watchdog ABORTS the process).  Under the RCCL backend every rank needs its own GPU (`pick_device`)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # "nccl" is RCCL on ROCm; STIL_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
            backend = os.environ.get("STIL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        ndev = torch.cuda.device_count()
        if os.environ.get("STIL_FAKE_DEVICE_COUNT") and backend == "nccl":   # TEST ONLY: the refusals without a GPU
            pick_device(backend, local, int(os.environ["STIL_FAKE_DEVICE_COUNT"]))
        if backend == "nccl":
            torch.cuda.set_device(pick_device(backend, local, ndev))
        elif ndev > 0 and torch.cuda.is_available():
            torch.cuda.set_device(pick_device(backend, local, ndev))        # gloo: the ranks of a test may share the one GPU of the box
        import datetime
        if timeout_s is None:
            timeout_s = float(os.environ.get("STIL_DIST_TIMEOUT_S", "900"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    return rank, world, local


from .comm import GradExchange, allreduce_flat, broadcast_state, sync_buffers, world_size  # noqa: E402,F401  (re-exported)


def _exchange_of(model):
    """The model's gradient exchange (created on first use, registered with the operator layer)."""
    from . import ops
    ex = getattr(model, "_grad_exchange", None)
    if ex is None or ex.flat is not model.flat:
        ex = model._grad_exchange = GradExchange(model.flat)
    ops._exchange = ex if world_size() > 1 else None
    return ex


def train_step(model, optimizer, batch, mask_random=None, mi_masks=None):
    """One optimisation step. Returns the (detached) loss tensor; no host sync.
    Data parallel (world > 1): rank 0's state is broadcast once before the first step (DDP's constructor), BN running
    statistics before every step (DDP broadcast_buffers, one message), and the gradient slab is all-reduced in buckets
    while backward is still running (comm.GradExchange); the 1/world of the gradient average is folded into Adam."""
    if world_size() > 1 and not getattr(model, "_state_broadcast", False):
        broadcast_state(model, optimizer)
        model._state_broadcast = True
    sync_buffers(model)
    optimizer.zero_grad()
    kw = {k: v for k, v in (('mask_random', mask_random), ('mi_masks', mi_masks)) if v is not None}  # STiL's injected randomness
    loss = model.training_step(batch, 0, **kw)
    ex = _exchange_of(model)
    ex.begin(getattr(model, "grad_signature", lambda: None)())
    loss.backward()
    optimizer.grad_scale = ex.finish()
    optimizer.step()
    return loss.detach()


def shard_batch(batch, rank: int, world: int):
    """Data-parallel shard keeping the labelled:unlabelled ratio on every rank."""
    if world == 1:
        return batch

    def cut(t, n):
        per = n // world
        return t[rank * per:(rank + 1) * per]

    out = {}
    for key in ("l", "u"):
        im, tab, y, orig, ident = batch[key]
        n = len(y)
        out[key] = ([cut(im[0], n), cut(im[1], n)], [cut(tab[0], n), cut(tab[1], n)], cut(y, n), cut(orig, n), cut(ident, n))
    return out


def synthetic_batch(field_lengths, num_classes: int, B: int, img_size: int, seed: int = 2022, device="cpu"):
    """Synthetic batch of BASELINE.md section 3 in the reference's layout; generated on the host
    with torch.Generator(seed) so every rank / the CPU baseline see identical data."""
    g = torch.Generator().manual_seed(seed)
    cat = [int(c) for c in field_lengths if int(c) != 1]
    ncon = sum(1 for c in field_lengths if int(c) == 1)
    B_l = max(B // 8, 1)
    img = torch.rand(B, 3, img_size, img_size, generator=g)
    cols = [torch.randint(0, c, (B, 1), generator=g).float() for c in cat]
    cols.append(torch.randn(B, ncon, generator=g))
    tab = torch.cat(cols, dim=1)
    cat_pos = [i for i, c in enumerate(field_lengths) if int(c) != 1]
    con_pos = [i for i, c in enumerate(field_lengths) if int(c) == 1]
    perm = [0] * len(field_lengths)
    for j, pos in enumerate(cat_pos + con_pos):
        perm[pos] = j
    tab = tab[:, perm]  # identity for categorical-first column orders (the base model requires that order)
    y = torch.randint(0, num_classes, (B,), generator=g)
    img, tab, y = img.to(device), tab.to(device), y.to(device)

    def part(sl, lab):
        n = sl.stop - sl.start
        return ([torch.zeros(n, device=device), img[sl]], [tab[sl], tab[sl]], y[sl], img[sl],
                torch.full((n,), lab, dtype=torch.bool, device=device))

    return {"l": part(slice(0, B_l), True), "u": part(slice(B_l, B), False)}


def wants_graph(batch: int, img_size: int) -> bool:
    """Is a step of this per-GPU size better replayed from a hipGraph (one queue, no host in the loop, but no second stream) than
    launched eagerly (two streams, ~1000 launches at 18-20 us of host time each)?  Measured on one box (profiles/r05y5_*): cardiac
    share of 16 samples per GPU at 128 px 18.2 ms eager / 11.0 replayed; B = 64 at 128 px 20.2 / 22.0; B = 32 at 224 px 21.1 /
    23.9; B = 64 at 224 px 35.2 / 39.1; B = 256: 119 / 138.  Replay wins while the inline GPU time is below the host's ~18 ms per
    step: the boundary is put at half a million pixels per step (32 samples at 128 px)."""
    return batch * img_size * img_size <= 524288


class GraphedTrainStep:
    """The whole optimisation step (zero_grad -> training_step -> backward -> all-reduce -> Adam) captured ONCE into a
    hipGraph and replayed: ~1500 kernel launches per step collapse into one graph launch, which is what matters when
    the per-GPU batch is small (BASELINE config 1: B = 32, config 5: B = 16 per GPU) and eager launches go host-bound.

    With STIL_GRAPH_SIDE=1 the capture keeps the step's two-stream structure (EMA teacher beside the student, weight
    gradients beside the input-gradient chain: ops._SideStream forks and joins are captured as graph dependencies);
    the default captures every launch inline (faster for the small per-GPU batches graphs are for, ops._SideStream).

    Static shapes only; the batch is copied into static device buffers before each replay.  Everything that varies per
    step lives in device memory (mask-RNG step counter, Adam step counts); what is baked in at capture -- learning rate,
    `current_epoch > start_epoch`, the set of parameters that receive gradients -- is watched, and a change triggers a
    re-capture.  Not supported: `DA: True` (host read of the queue pointer, as in the reference).
    """

    def __init__(self, model, optimizer, example_batch, warmup: int = 2):
        self.model, self.optimizer = model, optimizer
        model.setup_device()
        dev = model.prototypes.device
        self.static = {k: ([v[0][0].to(dev).clone(), v[0][1].to(dev).clone()], [v[1][0].to(dev).clone(), v[1][1].to(dev).clone()],
                           v[2].to(dev).clone(), v[3].to(dev).clone(), v[4].to(dev).clone()) for k, v in example_batch.items()}
        self.warmup = warmup
        self.graph = None
        self._key = None

    def _signature(self):
        g = self.optimizer.param_groups[0]
        return (float(g["lr"]), self.model.current_epoch > self.model.hp.start_epoch, self.model.flat._active_host)

    def _snapshot(self):
        m, f = self.model, self.model.flat
        tensors = [f.params, f.grads, f.exp_avg, f.exp_avg_sq, f.ema, f.steps, m.prototypes_sum, m.prototypes_count_sum, m._rng_step]
        tensors += list(f.s_counters) + list(f.t_counters)
        return [(t, t.clone()) for t in tensors]

    def _capture(self):
        from . import ops
        snap = self._snapshot()  # warm-up steps are real optimisation steps: their effect is rolled back below
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # gradient slab reductions are deferred to one multi-job launch per step (ops._DeferredReduce): in the warm-up too, which
        # sizes the arena the capture then bakes in
        with torch.cuda.stream(side), ops.deferring():  # eager warm-up on a side stream (allocator + lazy one-time setup), as PyTorch requires
            for _ in range(self.warmup):
                train_step(self.model, self.optimizer, self.static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), ops.deferring():  # records the step; nothing executes until replay()
            self.loss = train_step(self.model, self.optimizer, self.static)
        self._key = self._signature()
        for t, c in snap:
            t.copy_(c)

    def __call__(self, batch):
        for k in ("l", "u"):
            s, b = self.static[k], batch[k]
            s[0][1].copy_(b[0][1], non_blocking=True); s[1][1].copy_(b[1][1], non_blocking=True); s[2].copy_(b[2], non_blocking=True)
        if self.graph is None or self._key != self._signature():
            self._capture()
        self.graph.replay()
        return self.loss
