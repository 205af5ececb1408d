"""Per-step weight re-layouts of a whole model in ONE launch (csrc/layout.hip, stil_weight_layouts).

The weights stay in the reference's layout inside the flat slab (state_dict compatible, flat.py); the NT GEMM needs, per
layer, `[Cout][tap][Cin]` (forward operand of a k > 1 convolution), `[Cin][tap][Cout]` (input-gradient operand; the plain
transpose for 1x1 convolutions and nn.Linear) and one tap subset per output phase of a strided input-gradient
(ops.strided_dgrad).  Round 3 issued one small kernel per layer and use -- 136 launches per step; here a job table built once
per model drives a single grid over a layout slab, refreshed once per step (weights change only in Adam and in the EMA update).
The operators pick the views up from attributes of the parameter (`_stil_wf`, `_stil_wd`, `_stil_wphase`) when
`WeightLayouts.fresh` says they are current, and fall back to their own per-call kernels otherwise (operator-level tests,
foreign callers).  Pure data movement: bit-identical operands.
"""
from __future__ import annotations

import struct
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ._lib import lib
from .ops import _p, _stream

_JOB = struct.Struct("<qq12i")   # LayoutJob of csrc/layout.hip: src, dst, kind, Cout, Cin, KH, KW, stride, ky0, kx0, KHs, KWs, first_block, pad


def phase_specs(k: int, stride: int, pad: int, H: Optional[int] = None, W: Optional[int] = None):
    """The (py, px, ky0, kx0, KHs, KWs) of ops.strided_dgrad's phases that own taps (spatial emptiness is the caller's business)."""
    out = []
    for py in range(stride):
        ky0 = (py + pad) % stride
        KHs = len(range(ky0, k, stride))
        if KHs == 0:
            continue
        for px in range(stride):
            kx0 = (px + pad) % stride
            KWs = len(range(kx0, k, stride))
            if KWs == 0:
                continue
            out.append((py, px, ky0, kx0, KHs, KWs))
    return out


class WeightLayouts:
    """Layout views for every conv / Linear weight of `modules` that lives in `slab`.

    want_dgrad: also the input-gradient operands (training); False for a forward-only network (the EMA teacher)."""

    def __init__(self, slab: torch.Tensor, modules: List[nn.Module], want_dgrad: bool = True):
        assert lib().weight_layout_job_bytes() == _JOB.size, "LayoutJob layout changed"
        self.slab = slab
        dev = slab.device
        base = slab.data_ptr()
        jobs: List[Tuple] = []           # (src, dst, kind, Cout, Cin, KH, KW, stride, ky0, kx0, KHs, KWs, n_elements)
        views: List[Tuple] = []          # (param, attr, key, dst offset, shape)
        off = 0

        def add(param, attr, key, kind, Cout, Cin, KH, KW, shape, stride=1, ky0=0, kx0=0, KHs=0, KWs=0):
            nonlocal off
            src = (param.data_ptr() - base) // 4
            assert 0 <= src and src + param.numel() <= slab.numel(), "parameter outside the slab"
            n = 1
            for s_ in shape:
                n *= s_
            jobs.append((src, off, kind, Cout, Cin, KH, KW, stride, ky0, kx0, KHs, KWs, n))
            views.append((param, attr, key, off, shape))
            off += (n + 255) // 256 * 256      # 1 KB aligned views (16-byte vector loads of the GEMM)

        seen = set()
        for mod in modules:
            for m in mod.modules():
                if isinstance(m, nn.Conv2d):
                    w = m.weight
                    if id(w) in seen or not self._inside(w):
                        continue
                    seen.add(id(w))
                    Cout, Cin, KH, KW = w.shape
                    k, stride, pad = m.kernel_size[0], m.stride[0], m.padding[0]
                    if Cin % 4 != 0 or KH * KW > 9:   # the stem (3 input channels, 7x7) goes through its own im2col + padded weight
                        continue
                    if KH * KW > 1:
                        add(w, "_stil_wf", None, 0, Cout, Cin, KH, KW, (Cout, KH * KW * Cin))
                    if not want_dgrad:
                        continue
                    if stride == 1:
                        add(w, "_stil_wd", None, 1, Cout, Cin, KH, KW, (Cin, KH * KW * Cout))
                    else:
                        for (py, px, ky0, kx0, KHs, KWs) in phase_specs(k, stride, pad):
                            add(w, "_stil_wphase", (py, px), 2, Cout, Cin, KH, KW, (Cin, KHs * KWs * Cout), stride, ky0, kx0, KHs, KWs)
                elif isinstance(m, nn.Linear) and want_dgrad:
                    w = m.weight
                    if id(w) in seen or not self._inside(w) or not w.requires_grad:
                        continue
                    seen.add(id(w))
                    N, K = w.shape
                    add(w, "_stil_wd", None, 1, N, K, 1, 1, (K, N))
        self.out = torch.empty(max(off, 1), dtype=torch.float32, device=dev)
        recs, blk2job, nb = [], [], 0
        for j, (src, dst, kind, Cout, Cin, KH, KW, stride, ky0, kx0, KHs, KWs, n) in enumerate(jobs):
            recs.append(_JOB.pack(src, dst, kind, Cout, Cin, KH, KW, stride, ky0, kx0, KHs, KWs, nb, 0))
            b = lib().weight_layout_job_blocks(kind, Cout, Cin, KH, KW, KHs, KWs)
            assert b > 0, "unsupported layout job"
            blk2job += [j] * b
            nb += b
        self.n_blocks = nb
        self.n_jobs = len(jobs)
        if nb:
            self.jobs = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
            self.blk2job = torch.tensor(blk2job, dtype=torch.int32).to(dev)
        self.params = []
        for param, attr, key, o, shape in views:
            v = self.out[o:o + shape[0] * shape[1]].view(shape)
            if key is None:
                setattr(param, attr, v)
            else:
                d = getattr(param, attr, None)
                if d is None:
                    d = {}
                    setattr(param, attr, d)
                d[key] = v
            param._stil_layouts = self
            self.params.append(param)
        self.fresh = False
        self.event = None             # set by refresh(publish=True)
        self.waited = set()           # (device index, raw stream handle) of the streams that have waited for `event`
        self.version = -1             # slab._version at the last refresh
        self.pversion = {}            # id(param) -> param._version at the last refresh
        self.captured = False         # the last refresh was recorded inside a hipGraph capture

    def _inside(self, w) -> bool:
        b = self.slab.data_ptr()
        return w.is_cuda and b <= w.data_ptr() and w.data_ptr() + 4 * w.numel() <= b + 4 * self.slab.numel()

    @torch.no_grad()
    def refresh(self, publish: bool = False):
        """Recompute every view from the current weights (one launch on the current stream) and mark them usable.
        publish: the launch runs on a stream other than its consumers' (the side stream, beside the stem of the step): an event
        is recorded after it, and every other stream waits for that event once, at its first use of a view (ops.cached_layout)."""
        if self.n_blocks:
            lib().weight_layouts(_p(self.slab), _p(self.out), _p(self.jobs), _p(self.blk2job), self.n_blocks, _stream())
        self.event = None
        if publish and self.n_blocks:
            st = torch.cuda.current_stream(self.slab.device)
            self.event = torch.cuda.Event()
            self.event.record(st)
            self.waited = {(st.device_index, st.cuda_stream)}
        # freshness is tied to the DATA, not only to the call sites that remember to invalidate: writes to the slab or to a
        # parameter that go through PyTorch (checkpoint restores, collectives into the slab, tests) move their version counters
        self.version = self.slab._version
        self.pversion = {id(p): p._version for p in self.params}
        self.captured = torch.cuda.is_current_stream_capturing()
        self.fresh = True

    def current(self, param) -> bool:
        """May `param`'s views be used now?  (ops.cached_layout)"""
        if not self.fresh or self.slab._version != self.version or param._version != self.pversion.get(id(param), -2):
            return False
        if self.captured and not torch.cuda.is_current_stream_capturing():
            return False      # recorded in a capture: outside it neither the views' producer nor its event is ordered before us
        return True

    def invalidate(self):
        """The weights are about to change (optimizer step, EMA update, load_state_dict): operators fall back to their own
        per-call layouts until the next refresh()."""
        self.fresh = False
