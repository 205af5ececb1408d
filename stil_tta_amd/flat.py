"""Flat HBM slabs for parameters / gradients / Adam moments / the EMA teacher.

Layout of the student slab (floats, every tensor padded to a 1024-element boundary):
    [ backbone parameters | backbone float buffers (BN running stats) | head parameters (projectors, CLUB) ]
The teacher slab mirrors the first two regions, so momentum_update_ema (STiLModel.py:154-168) is ONE
streaming kernel over a contiguous range (parameters only when eman is False), Adam is one kernel over the
student slab (buffer chunks carry tensor id -1 and are skipped), and the data-parallel gradient exchange is
one RCCL all-reduce of the gradient slab.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Tuple

import torch
import torch.nn as nn

from ._lib import lib
from .ops import _p, _stream, join_side

ALIGN = 1024


def _round_up(n, a=ALIGN):
    return (n + a - 1) // a * a


class FlatState:
    def __init__(self, student: nn.Module, teacher: nn.Module, heads: List[nn.Module], device):
        self._modules_s, self._module_t = [student] + list(heads), teacher
        self._plans = None            # (student layouts, teacher layouts): layouts.WeightLayouts, built on first refresh
        s_keys, t_keys = set(student.state_dict().keys()), set(teacher.state_dict().keys())
        s_params = [(n, p) for n, p in student.named_parameters()]
        s_bufs = [(n, b) for n, b in student.named_buffers() if b.dtype == torch.float32 and n in s_keys]
        h_params = []
        for hi, h in enumerate(heads):
            h_params += [(f"head{hi}.{n}", p) for n, p in h.named_parameters()]
        self.names: List[str] = []
        offs, off = [], 0
        for _, t in s_params + s_bufs + h_params:
            offs.append(off)
            off += _round_up(t.numel())
        self.total = off
        self.n_backbone_params = sum(_round_up(p.numel()) for _, p in s_params)
        self.n_backbone_state = self.n_backbone_params + sum(_round_up(b.numel()) for _, b in s_bufs)
        self.params = torch.zeros(self.total, dtype=torch.float32, device=device)
        self._grads = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.ema = torch.zeros(self.n_backbone_state, dtype=torch.float32, device=device)
        chunk2tensor = torch.full((self.total // ALIGN,), -1, dtype=torch.int32)
        self.tensors: List[torch.nn.Parameter] = []
        for i, ((name, t), o) in enumerate(zip(s_params + s_bufs + h_params, offs)):
            n = t.numel()
            view = self.params[o:o + n].view(t.shape)
            view.copy_(t.data.to(device))
            t.data = view
            is_param = i < len(s_params) or i >= len(s_params) + len(s_bufs)
            if is_param:
                t._gslot = self._grads[o:o + n].view(t.shape)
                t._stil_touched = False
                t.grad = None
                tid = len(self.tensors)
                self.tensors.append(t)
                self.names.append(name)
                chunk2tensor[o // ALIGN:(o + _round_up(n)) // ALIGN] = tid
        self.chunk2tensor = chunk2tensor.to(device)
        self.steps = torch.zeros(len(self.tensors), dtype=torch.int32, device=device)
        self.active = torch.zeros(len(self.tensors), dtype=torch.uint8, device=device)
        self._active_host: Tuple[int, ...] = ()
        # teacher: same order/shapes (asserted), re-pointed into the ema slab; initialised as a copy of the student
        t_params = [(n, p) for n, p in teacher.named_parameters()]
        t_bufs = [(n, b) for n, b in teacher.named_buffers() if b.dtype == torch.float32 and n in t_keys]
        assert [n for n, _ in t_params] == [n for n, _ in s_params] and [n for n, _ in t_bufs] == [n for n, _ in s_bufs]
        for (name, t), o in zip(t_params + t_bufs, offs):
            n = t.numel()
            view = self.ema[o:o + n].view(t.shape)
            view.copy_(t.data.to(device))
            t.data = view
            if isinstance(t, nn.Parameter):
                t.requires_grad = False
        # int64 counters (num_batches_tracked) are copied, not averaged
        self.s_counters = [b for n, b in student.named_buffers() if b.dtype == torch.int64 and n in s_keys]
        self.t_counters = [b for n, b in teacher.named_buffers() if b.dtype == torch.int64 and n in t_keys]
        for b in self.s_counters + self.t_counters:
            b.data = b.data.to(device)

    # ---- per-step weight re-layouts (layouts.py): one launch for the student (+ heads), one for the teacher
    def refresh_layouts(self, student: bool = True, teacher: bool = True, side=None):
        """Recompute the GEMM operand layouts of every conv / Linear weight from the CURRENT weights, on the current stream.
        Called by training_step (student at its start, teacher right after the EMA of its parameters); everything that
        changes weights invalidates them again (Adam, EMA, load_state_dict), and stale layouts are never used (ops.cached_layout)."""
        if os.environ.get("STIL_LAYOUT_PLAN", "1") == "0":
            return
        if self._plans is None:
            from .layouts import WeightLayouts
            self._plans = (WeightLayouts(self.params, self._modules_s, True), WeightLayouts(self.ema, [self._module_t], False))
            for m in self._modules_s:
                m.register_load_state_dict_post_hook(lambda *_: self.invalidate_layouts(True, False))
            self._module_t.register_load_state_dict_post_hook(lambda *_: self.invalidate_layouts(False, True))
        if side is not None:
            # beside the step's stem (im2col, 7x7 conv, max-pool need no layout): the launches go to the side stream, which is
            # ordered after everything issued so far; consumers on other streams wait for the recorded event at first use
            main = torch.cuda.current_stream(self.params.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if student:
                    self._plans[0].refresh(publish=True)
                if teacher:
                    self._plans[1].refresh(publish=True)
            return
        if student:
            self._plans[0].refresh()
        if teacher:
            self._plans[1].refresh()

    def invalidate_layouts(self, student: bool = True, teacher: bool = True):
        if self._plans is not None:
            if student:
                self._plans[0].invalidate()
            if teacher:
                self._plans[1].invalidate()

    # ---- EMA teacher
    @torch.no_grad()
    def ema_update(self, momentum: float, eman: bool):
        self.invalidate_layouts(False, True)
        n = self.n_backbone_state if eman else self.n_backbone_params
        lib().ema_update(_p(self.ema), _p(self.params), n, float(momentum), _stream())
        if eman and self.s_counters:
            torch._foreach_copy_(self.t_counters, self.s_counters)

    @torch.no_grad()
    def ema_update_params(self, momentum: float):
        """Parameters only (the BN running buffers follow layer by layer, see modules.TeacherPipe)."""
        self.invalidate_layouts(False, True)
        lib().ema_update(_p(self.ema), _p(self.params), self.n_backbone_params, float(momentum), _stream())

    @torch.no_grad()
    def ema_update_range(self, offset: int, n: int, momentum: float):
        """EMA of ema[offset:offset+n] (floats; both multiples of the 1024-float tensor alignment)."""
        assert offset % ALIGN == 0 and n % ALIGN == 0 and 0 <= offset and offset + n <= self.n_backbone_state
        vp = ctypes.c_void_p
        lib().ema_update(vp(self.ema.data_ptr() + 4 * offset), vp(self.params.data_ptr() + 4 * offset), n, float(momentum), _stream())

    @torch.no_grad()
    def copy_counters(self, eman: bool):
        if eman and self.s_counters:
            torch._foreach_copy_(self.t_counters, self.s_counters)

    @torch.no_grad()
    def copy_student_to_teacher(self):
        self.invalidate_layouts(False, True)
        self.ema.copy_(self.params[: self.n_backbone_state])
        if self.s_counters:
            torch._foreach_copy_(self.t_counters, self.s_counters)

    def buffer_slabs(self):
        """Student / teacher float buffers (BN running stats), each one contiguous range: what DDP's default
        broadcast_buffers=True re-sends from rank 0 before every forward (the reference trains under Lightning DDP)."""
        a, b = self.n_backbone_params, self.n_backbone_state
        return [self.params[a:b], self.ema[a:b]] if b > a else []

    # ---- gradients
    @property
    def grads(self) -> torch.Tensor:
        """The gradient slab; reading it joins the side stream the weight-gradient GEMMs run on (ops.wgrad_param)."""
        join_side()
        return self._grads

    def zero_grad(self):
        self.grads.zero_()
        for t in self.tensors:
            t._stil_touched = False
            t.grad = None

    def publish_grads(self):
        """Expose the slab views as .grad (for inspection / tests / foreign optimizers)."""
        join_side()
        for t in self.tensors:
            t.grad = t._gslot if t._stil_touched else None

    # ---- Adam
    @torch.no_grad()
    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        self.invalidate_layouts(True, False)
        act = tuple(1 if t._stil_touched else 0 for t in self.tensors)
        if act != self._active_host:  # changes only when the set of live loss terms changes (epoch boundary)
            self.active.copy_(torch.tensor(act, dtype=torch.uint8))
            self._active_host = act
        lib().adam_step(_p(self.params), _p(self.grads), _p(self.exp_avg), _p(self.exp_avg_sq), _p(self.chunk2tensor),
                        _p(self.steps), _p(self.active), len(self.tensors), self.total, float(lr), float(betas[0]),
                        float(betas[1]), float(eps), float(weight_decay), float(grad_scale), _stream())


class StilAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (STiLModel.py:563-570) on the flat slabs; usable by Lightning's automatic optimisation."""

    def __init__(self, flat: FlatState, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        super().__init__([{"params": flat.tensors}], dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps))
        self.flat = flat
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none: bool = True):
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        self.flat.adam_step(g["lr"], g["betas"], g["eps"], g["weight_decay"], self.grad_scale)
        return loss
