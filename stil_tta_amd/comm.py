"""Data-parallel communication of the STiL step: one process per GPU, torch.distributed (backend "nccl" == RCCL
over xGMI on ROCm; "gloo" in the CPU / shared-GPU tests).

  * GradExchange      -- the gradient slab is all-reduced bucket by bucket WHILE backward is still running: a bucket
                         (a contiguous range of the flat gradient slab, flat.py) goes out on a communication stream as
                         soon as the last of its tensors has received its gradient; DDP's bucketed overlap, rebuilt on
                         the slab (the reference trains under Lightning DDP, trainers/evaluate.py:170-179).
  * sync_buffers      -- DDP broadcast_buffers semantics in ONE broadcast (student + teacher BN running statistics).
  * broadcast_state   -- DDP's construction-time broadcast of parameters / buffers (+ Adam state, prototypes) from rank 0.
  * AllGatherFn / allreduce_mean -- autograd-aware collectives for the optional `global_contrast` mode:
                         ITC over the global batch, CLUB with global batch means.

xGMI is point-to-point (7 links per GPU, per-link bound): buckets are few and large (32 MB), the one prototype
exchange is a single [K, 129] message, and nothing else crosses the links.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# ------------------------------------------------------------------------------------------ gradient exchange
class GradExchange:
    """Bucketed all-reduce of the flat gradient slab, overlapped with backward.

    Buckets are contiguous slab ranges of ~`bucket_elems` floats cut at tensor boundaries.  How many "+=" each parameter
    receives in one backward pass (shared weights receive several) is LEARNED: the first step under a given signature
    (the model's set of live loss terms) runs with a plain post-backward exchange of EVERY bucket in index order and
    records the per-tensor touch counts and the order in which buckets completed.  The plan is then AGREED between the
    ranks: rank 0's counts and order are broadcast, every rank compares its own counts with them, and only if all ranks
    agree (one MIN all-reduce) does the signature enter overlap mode -- with rank 0's order on every rank; otherwise the
    signature stays on the post-backward exchange for good.  Later steps launch each bucket as soon as its counts are
    met, always in the agreed order, so every rank issues the same sequence of collectives whatever its autograd engine
    does.  Every step additionally opens with one 4-word MAX all-reduce of (signature hash, mode): ranks that disagree on
    either would issue different collective sequences, and all of them raise instead of hanging in RCCL.  A gradient
    arriving for a bucket that has already left, or a changed set of touched parameters, raises -- but only after this
    rank has issued the step's full collective sequence, so the other ranks are never left blocked."""

    def __init__(self, flat, bucket_elems: int = 8 << 20):
        self.flat = flat
        self.enabled = os.environ.get("STIL_OVERLAP_ALLREDUCE", "1") != "0"
        offs = [(t._gslot.data_ptr() - flat._grads.data_ptr()) // 4 for t in flat.tensors]
        ends = [o + (t.numel() + 1023) // 1024 * 1024 for o, t in zip(offs, flat.tensors)]
        self.bucket_of: List[int] = []
        self.ranges: List[Tuple[int, int]] = []
        start = None
        for i, (o, e) in enumerate(zip(offs, ends)):
            gap = start is not None and o != ends[i - 1]   # the BN-buffer region between backbone and head parameters
            if start is None or gap or e - start > bucket_elems:
                if start is not None:
                    self.ranges.append((start, ends[i - 1]))
                start = o
            self.bucket_of.append(len(self.ranges))
        self.ranges.append((start, ends[-1]))
        self.tid = {id(t): i for i, t in enumerate(flat.tensors)}
        # signature -> (expected touches per tensor, bucket order) once agreed, or None = "the ranks disagreed: never overlap"
        self.plans: Dict[object, Optional[Tuple[List[int], List[int]]]] = {}
        self.comm_stream: Optional[torch.cuda.Stream] = None
        self.active = False
        self._check = None            # (work / event, result words, what) of the step-opening agreement all-reduce, verified lazily
        self._chk = None              # pinned in / device / pinned out buffers of that agreement (GPU runs)
        self._reset(None)

    def _reset(self, sig):
        self.sig = sig
        self.active = world_size() > 1
        n = len(self.flat.tensors)
        self.counts = [0] * n
        self.last_touch = {}          # bucket -> sequence number of its latest contribution (learning step)
        self.seq = 0
        self.works = []
        self.fired = set()
        self.ready = set()
        self.bad = None               # first protocol violation of this step (raised at the end of finish())
        plan = self.plans.get(sig) if (sig is not None and self.enabled and self.active) else None
        self.expect, self.fire_order = plan if plan else (None, None)
        self.next_fire = 0
        if self.expect is not None:
            self.remaining = [0] * len(self.ranges)
            for i, c in enumerate(self.expect):
                self.remaining[self.bucket_of[i]] += c

    # ---- agreement between the ranks
    def _comm(self, fn, t):
        """Run collective `fn(t)` on the communication stream (device tensors) or inline (host tensors); returns the work."""
        if t.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(t.device)
            self.comm_stream.wait_stream(torch.cuda.current_stream(t.device))   # `t` was produced on the current stream
            with torch.cuda.stream(self.comm_stream):
                return fn(t)
        return fn(t)

    @staticmethod
    def _capturing() -> bool:
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def _open_check(self):
        """Every rank must be at the same signature and in the same mode (learning / post-backward vs overlap).
        Nothing here blocks the host or the main stream: the four words travel host -> device on the main stream (pinned,
        non-blocking), are MAX-all-reduced on the communication stream and come back into pinned host memory on that
        stream; verify() reads them one step later (or at once on a learning step)."""
        import zlib
        h = zlib.crc32(repr(self.sig).encode()) & 0x7fffffff
        mode = 0 if self.expect is None else 1
        what = f"signature {self.sig!r}, mode {'overlap' if mode else 'post-backward'}"
        dev = self.flat._grads.device
        vals = torch.tensor([h, -h, mode, -mode], dtype=torch.int64)
        if dev.type != "cuda":
            wk = dist.all_reduce(vals, op=dist.ReduceOp.MAX, async_op=True)
            self._check = (wk, vals, what)
            return
        if self._chk is None:
            self._chk = (torch.empty(4, dtype=torch.int64).pin_memory(), torch.empty(4, dtype=torch.int64, device=dev),
                         torch.empty(4, dtype=torch.int64).pin_memory())
        pin_in, t, pin_out = self._chk
        pin_in.copy_(vals)
        t.copy_(pin_in, non_blocking=True)                              # main stream, asynchronous
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(dev)
        cs = self.comm_stream
        cs.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cs):
            dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True).wait()   # cs waits for the collective
            pin_out.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
        self._check = (ev, pin_out, what)

    def verify(self):
        """Host check of the last step-opening agreement (4 words; the collective was the first one of its step)."""
        if self._check is None:
            return
        wk, t, what = self._check
        self._check = None
        wk.synchronize() if isinstance(wk, torch.cuda.Event) else wk.wait()
        v = t.tolist()
        if v[0] != -v[1] or v[2] != -v[3]:
            # A rank in overlap mode may already have bucket all-reduces in flight that its peers will never match: tear the
            # communicator down before raising, so that no rank is left spinning in a collective (on the GPU: RCCL kernels
            # that would block process teardown in the process group's destructor).
            _abort_process_group()
            raise RuntimeError(f"data-parallel ranks disagree on the gradient exchange of this step (this rank: {what}; "
                               f"hash max/min {v[0]}/{-v[1]}, mode max/min {v[2]}/{-v[3]}): refusing to issue mismatched collectives")

    def _agree_plan(self, counts, order):
        """Rank 0's (counts, order) become the plan iff every rank counted the same touches; else None (never overlap)."""
        nb = len(self.ranges)
        mine = torch.tensor(list(counts) + list(order) + [-1] * (nb - len(order)), dtype=torch.int64, device=self.flat._grads.device)
        plan = mine.clone()
        self._comm(lambda x: dist.broadcast(x, src=0), plan)
        if plan.is_cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        n = len(counts)
        same = bool((plan[:n] == mine[:n]).all().item())
        ok = torch.tensor([1 if same else 0], dtype=torch.int64, device=mine.device)
        self._comm(lambda x: dist.all_reduce(x, op=dist.ReduceOp.MIN), ok)
        if ok.is_cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if int(ok.item()) != 1:
            import warnings
            warnings.warn(f"GradExchange: the ranks' backward passes touch different parameters under signature {self.sig!r}; "
                          "this signature keeps the post-backward exchange (no overlap)")
            return None
        pl = plan.tolist()
        return pl[:n], [b for b in pl[n:] if b >= 0]

    # ---- called by the step driver
    def begin(self, sig):
        """Start of backward.  `sig` identifies the set of live loss terms (None: never overlap)."""
        capturing = self._capturing()
        if not capturing:
            self.verify()             # the previous step's agreement (long complete: no stall)
        self._reset(sig)
        if self.active and not capturing:
            self._open_check()

    def note(self, param):
        """One gradient contribution to `param` has been ISSUED (on the current or the side stream)."""
        if not self.active:
            return
        i = self.tid.get(id(param))
        if i is None:
            return
        self.counts[i] += 1
        b = self.bucket_of[i]
        self.seq += 1
        self.last_touch[b] = self.seq
        if self.expect is None:            # learning step / no overlap: everything leaves in finish()
            return
        if b in self.fired:
            self.bad = self.bad or (f"gradient for {self.flat.names[i]} arrived after its bucket was all-reduced "
                                    "(the backward graph changed under an unchanged signature)")
            return
        self.remaining[b] -= 1
        if self.remaining[b] == 0:
            self.ready.add(b)
            while self.next_fire < len(self.fire_order) and self.fire_order[self.next_fire] in self.ready:
                self._fire(self.fire_order[self.next_fire])
                self.next_fire += 1

    def _fire(self, b):
        from . import ops
        a, e = self.ranges[b]
        slab = self.flat._grads[a:e]
        if slab.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(slab.device)
            cs = self.comm_stream
            cs.wait_stream(torch.cuda.current_stream(slab.device))   # BN / LN / bias gradients (main stream)
            for st in ops._side.all_streams():                       # weight gradients (side stream), the tabular branch's own gradients
                if st.device == slab.device:
                    cs.wait_stream(st)
            with torch.cuda.stream(cs):
                self.works.append(dist.all_reduce(slab, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self.works.append(dist.all_reduce(slab, op=dist.ReduceOp.SUM, async_op=True))
        self.fired.add(b)

    def finish(self) -> float:
        """After backward: send what has not left yet, wait for everything; returns the factor Adam scales by (1/world)."""
        if not self.active:
            return 1.0
        self.flat.grads  # joins the side stream (weight gradients) into the current one
        learned = None
        if self.expect is None:
            # post-backward exchange: EVERY bucket, in index order -- the same sequence on every rank whatever its
            # autograd engine did.  On a LEARNING step first make sure all ranks are here in this mode (a rare step: the
            # host read is free); steady post-backward steps are verified one step later, like overlap steps.
            # Inside a hipGraph capture nothing may synchronise with the host: the agreement below (a broadcast, an all-reduce
            # and two host reads) cannot run, so a captured first step does NOT learn -- it keeps the post-backward exchange
            # and leaves the signature out of `plans`; the next eager step under this signature learns it.
            learning = self.sig is not None and self.enabled and self.sig not in self.plans and not self._capturing()
            if learning:
                self.verify()
            pending = list(range(len(self.ranges)))
            if learning:
                # buckets that received gradients, ordered by their LAST contribution (the order they can leave in next time)
                learned = (list(self.counts), sorted(self.last_touch, key=self.last_touch.get))
        else:
            if self.counts != self.expect:
                self.bad = self.bad or "the set of parameters receiving gradients changed under an unchanged signature"
            pending = list(self.fire_order[self.next_fire:])   # the agreed plan covers exactly the buckets every rank touches
        for b in pending:
            if b not in self.fired:
                self._fire(b)
        for wk in self.works:
            wk.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if learned is not None:
            self.plans[self.sig] = self._agree_plan(*learned)
        if self.bad:
            raise RuntimeError(self.bad)
        return 1.0 / world_size()


def _abort_process_group():
    """Best effort: abort (not destroy: destroy waits for pending work) the default process group's backend."""
    try:
        pg = dist.distributed_c10d._get_default_group()
        for dev in ("cuda", "cpu"):
            try:
                be = pg._get_backend(torch.device(dev))
            except Exception:
                continue
            for name in ("abort", "_abort", "_shutdown"):
                fn = getattr(be, name, None)
                if fn is not None:
                    try:
                        fn()
                    except Exception:
                        pass
                    break
    except Exception:
        pass


def allreduce_flat(slab: torch.Tensor, bucket_elems: int = 64 << 20) -> float:
    """SUM all-reduce of a flat fp32 slab in a few large buckets, after backward (no overlap).
    Returns the factor the consumer must scale by (1/world) -- applied inside the Adam kernel."""
    w = world_size()
    if w == 1:
        return 1.0
    n = slab.numel()
    works = [dist.all_reduce(slab[o:min(n, o + bucket_elems)], op=dist.ReduceOp.SUM, async_op=True) for o in range(0, n, bucket_elems)]
    for wk in works:
        wk.wait()
    return 1.0 / w


# ------------------------------------------------------------------------------------------ buffers / initial state
def sync_buffers(model):
    """DDP broadcast_buffers semantics: every rank starts the step with rank 0's BN running statistics, student AND EMA
    teacher (the reference's LightningModule is wrapped whole, `ema` included).  The two buffer ranges live in
    different slabs: they are packed into one staging buffer so that ONE broadcast crosses the links."""
    if world_size() == 1:
        return
    model.setup_device()
    slabs = model.flat.buffer_slabs()
    if not slabs:
        return
    if len(slabs) == 1:
        dist.broadcast(slabs[0], src=0)
        return
    stage = torch.cat([s.reshape(-1) for s in slabs])
    dist.broadcast(stage, src=0)
    if dist.get_rank() != 0:
        o = 0
        for s in slabs:
            s.copy_(stage[o:o + s.numel()])
            o += s.numel()


def broadcast_state(model, optimizer=None):
    """What DistributedDataParallel does at construction (and Lightning when it restores a checkpoint on every rank):
    rank 0's parameters, buffers, EMA teacher, Adam moments / step counts and prototype buffers become everybody's,
    so differently seeded or differently restored ranks cannot drift apart silently."""
    if world_size() == 1:
        return
    model.setup_device()
    f = model.flat
    f.invalidate_layouts()      # the slabs are about to be overwritten: no operand layout derived from them survives
    for t in (f.params, f.ema, f.exp_avg, f.exp_avg_sq, f.steps):
        dist.broadcast(t, src=0)
    for b in list(f.s_counters) + list(f.t_counters):
        dist.broadcast(b, src=0)
    slabs = {f.params.untyped_storage().data_ptr(), f.ema.untyped_storage().data_ptr()}
    for _, b in model.named_buffers():       # prototypes, DA queue, the baselines' memory banks / queues and their pointers
        if b.untyped_storage().data_ptr() not in slabs:
            dist.broadcast(b, src=0)


# ------------------------------------------------------------------------------------------ autograd-aware collectives
class AllGatherFn(torch.autograd.Function):
    """x [b, D] on every rank -> cat over ranks [world*b, D] (rank order).  Backward: the gathered tensor's gradient is
    SUM-all-reduced and this rank's rows are returned.  Every rank evaluates the same global loss redundantly, so the
    sum is world x the row gradient, which the 1/world of the gradient average (DDP semantics, folded into Adam) turns
    back into d(global loss)/d(local rows)."""

    @staticmethod
    def forward(ctx, x):
        w, r = dist.get_world_size(), dist.get_rank()
        x = x.contiguous()
        out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x) if hasattr(dist, "all_gather_into_tensor") and x.is_cuda and dist.get_backend() == "nccl" \
            else out.copy_(torch.cat(_all_gather_list(x)))
        ctx.rows = (r * x.shape[0], x.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        o, n = ctx.rows
        return g[o:o + n]


def _all_gather_list(x):
    parts = [torch.empty_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, x)
    return parts


def all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    return x if world_size() == 1 else AllGatherFn.apply(x)


def allreduce_mean_(t: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks of a statistics tensor (no autograd: used inside custom Functions)."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(dist.get_world_size())
    return t
