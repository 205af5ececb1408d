"""Host-side operator layer: torch.autograd.Functions over the C ABI of libstil_hip.so.

PyTorch is used for device memory (allocator), streams and autograd bookkeeping only; every
arithmetic kernel below is hand-written HIP reached through include/stil_hip.h.  Parameter
gradients are accumulated straight into the flat gradient slab (see flat.py) when the parameter
carries a `_gslot` view, so no ATen accumulate kernels run for the 406 parameter tensors.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from ._lib import lib

_vp = ctypes.c_void_p


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()   # ctypes takes the integer for a void* argument (no c_void_p object per operand)


# The current stream's raw handle straight from the C binding: torch.cuda.current_stream() builds a Stream object through
# _get_device_index -> is_available -> os.environ on every call (5-7 us; ~1400 calls per step, a quarter of the host time of the
# launch-bound small-batch steps: tests/tools/host_profile.py).
_raw_current_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_current_device = getattr(torch._C, "_cuda_getDevice", None)


def _raw_stream(device=None) -> int:
    """hipStream_t of the current stream of `device` (a torch.device / index / None = current device) as an integer"""
    if _raw_current_stream is None or _current_device is None:
        return torch.cuda.current_stream(device).cuda_stream
    if device is None:
        idx = _current_device()
    elif isinstance(device, int):
        idx = device
    else:
        idx = device.index if device.index is not None else _current_device()
    return _raw_current_stream(idx)


def _stream():
    return _raw_stream()


_STREAM_OBJS = {}


def current_stream_obj(device=None) -> "torch.cuda.Stream":
    """torch.cuda.current_stream(device), cached by raw handle (the Stream object costs 5-7 us to build; Event.record() / .wait()
    without an explicit stream build one too)"""
    if _raw_current_stream is None or _current_device is None:
        return torch.cuda.current_stream(device)
    idx = _current_device() if device is None else (device if isinstance(device, int) else (device.index if device.index is not None else _current_device()))
    key = (idx, _raw_current_stream(idx))
    st = _STREAM_OBJS.get(key)
    if st is None:
        st = _STREAM_OBJS[key] = torch.cuda.current_stream(idx)
    return st


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("stil_tta_amd ops need CUDA(HIP) tensors: there is no CPU fallback")
        if not t.is_contiguous():
            raise RuntimeError("stil_tta_amd ops need contiguous tensors")


class _Workspace:
    """One grow-only scratch buffer per (device, stream): launches of one stream are ordered, so reuse is safe.
    A buffer that was handed out while a hipGraph was being captured is baked into that graph by address: it is never freed
    (a later regrowth keeps it alive in `pinned`), so replays never write into memory the allocator has given to someone else."""
    zero = False

    def __init__(self):
        self.buf = {}
        self.captured = set()    # keys whose current buffer a captured graph may hold
        self.pinned = []         # superseded buffers a captured graph may still address

    def get(self, nbytes: int, device, stream=None) -> torch.Tensor:
        """`stream` (a torch Stream): the stream the buffer will be used on when that is not the current one -- a (re)allocation
        then happens under it, so the block belongs to that stream's pool"""
        nbytes = max(int(nbytes), 256)
        key = (device, _raw_stream(device) if stream is None else stream.cuda_stream)
        b = self.buf.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if b is None or b.numel() < nbytes:
            if b is not None and key in self.captured:
                self.pinned.append(b)
                self.captured.discard(key)
            n = int(nbytes * (1.5 if self.zero else 1.25)) + 4096
            if stream is None:
                b = (torch.zeros if self.zero else torch.empty)(n, dtype=torch.uint8, device=device)
            else:
                with torch.cuda.stream(stream):
                    b = (torch.zeros if self.zero else torch.empty)(n, dtype=torch.uint8, device=device)
            self.buf[key] = b
        if capturing:
            self.captured.add(key)
        return b


_ws = _Workspace()


class _SplitWorkspace(_Workspace):
    """Split-K workspace of stil_gemm_nt (arrival tickets + slabs), one per (device, stream), ZERO-initialised: the library leaves
    the tickets zero after every launch, so a buffer is zeroed once, when it is (re)allocated."""
    zero = True


_split_ws = _SplitWorkspace()
# Split-K of the NT products whose grid leaves CUs idle (include/stil_hip.h `split_ws`): AUTOMATIC -- the library splits grids below
# one 64x64 workgroup per CU with K >= 256, and grids of 256-1535 tiles with K >= 1152 (gemm.hip nt_splits, from tests/tools/
# split_sweep.py), i.e. what a per-GPU batch of 16-64 samples launches (cardiac share of 16 samples per GPU under graph replay:
# 16.6 ms unsplit, 13.6 ms with release/acquire slabs, 12.1 ms with write-through slabs and this policy) and, at B = 256, only
# the heads' M = 256 products (step unchanged).  Round 4 kept it opt-in because it re-rolls the rounding of every small product,
# which moved projector_imaging.bias across its bar; that gradient's noise had another cause (csrc/loss.hip, round 5) and sits at
# 0.1-0.4 of its bar now.  STIL_SPLITK=0 turns it off.
_SPLITK = __import__("os").environ.get("STIL_SPLITK", "1") != "0"

# per-call tuning arguments of stil_gemm_nt / stil_wgrad_tn (include/stil_hip.h); 0 = automatic.  Only the measurement
# tools and bench.py's A/B environment knobs (STIL_GEMM_TUNE, STIL_WGRAD_TUNE) set them.
TUNE = {"gemm": int(__import__("os").environ.get("STIL_GEMM_TUNE", "0")), "wgrad": int(__import__("os").environ.get("STIL_WGRAD_TUNE", "0"))}
# STIL_PRECISION=bf16x3 (bench.py --precision bf16x3): the OPT-IN split-precision mode of the NT products (csrc/gemm.hip B3): + 100000 on
# every stil_gemm_nt `tune`; launches that do not qualify (ragged / unaligned operands, other tiles) run the fp32-exact kernels.
B3_FLAG = 100000
if __import__("os").environ.get("STIL_PRECISION", "fp32") == "bf16x3":
    TUNE["gemm"] += B3_FLAG


class _SideStream:
    """Weight-gradient GEMMs leave the critical path of backward: they only feed the gradient slab, so they run on a
    second HIP stream and overlap the HBM-bound BN / elementwise passes of the layers below (fork = side waits on
    main when the launch is issued; join = main waits on side before anything reads the slab, see join_side()).
    STIL_WGRAD_STREAM=0 keeps everything on one stream.  Inside a hipGraph capture the forks and joins are captured too
    with STIL_GRAPH_SIDE=1 (the side stream joins the capture through wait_stream, events become graph dependencies), so a
    replayed step keeps both streams' concurrency; the default keeps captured launches inline on the capturing stream,
    which is what measured faster where graphs pay at all (profiles/r03_experiments.txt: cardiac, 16 samples per GPU:
    18.2 ms inline / 20.0 ms two-stream / 19.3 ms eager; at B = 256 the two-stream graph beats the inline one, 138 vs 141 ms,
    and eager beats both at 130 ms)."""

    def __init__(self):
        import os
        self.enabled = os.environ.get("STIL_WGRAD_STREAM", "1") != "0"
        self.in_capture = os.environ.get("STIL_GRAPH_SIDE", "0") != "0"
        self.streams = {}
        # the BRANCH stream (OPT-IN, STIL_BRANCH_STREAM=1): the student's tabular encoder runs its forward there, beside the image
        # encoder on the main stream, and -- the autograd engine runs a node's backward on the stream of its forward -- its backward
        # too, so that the attention / LayerNorm kernels of that branch hide under the image branch's GEMMs instead of running alone
        # on the main stream (profiles/r05y8_timeline_b256.txt: ~6 ms of non-GEMM kernels run alone).  Bit-identical, and measured
        # SLOWER on the step: 2062-2082 samples/s against 2074-2098 without, paired -- a third queue of matrix-bound kernels takes
        # the dominant GEMM from 65 to 50 TF/s while it co-runs and buys nothing in aggregate (profiles/r05_experiments.txt 15).
        self.use_branch = os.environ.get("STIL_BRANCH_STREAM", "0") != "0"
        self.branch = {}
        # (event recorded on the side stream after a launch, the tensors that launch reads).  Holding the references
        # (1) keeps the autograd engine from accumulating IN PLACE into a gradient buffer a pending side-stream kernel
        # still has to read (the engine only does that when it holds the last reference), and (2) keeps the allocator
        # from handing the block to a main-stream allocation before the side stream is done with it.
        self.keep = []

    def retire(self, everything=False):
        """Drop the references whose side-stream work has completed (or all of them, after a join)."""
        k = self.keep
        if torch.cuda.is_current_stream_capturing():   # nothing executes during capture and events cannot be queried:
            if everything:                              # the references live until the join
                del k[:]
            return
        i = 0
        while i < len(k) and (everything or k[i][0].query()):
            i += 1
        del k[:i]

    def get(self, device):
        st = self.streams.get(device)
        if st is None:
            st = self.streams[device] = torch.cuda.Stream(device)
        return st

    def get_branch(self, device):
        st = self.branch.get(device)
        if st is None:
            st = self.branch[device] = torch.cuda.Stream(device)
        return st

    def all_streams(self):
        return list(self.streams.values()) + list(self.branch.values())


_side = _SideStream()


def join_side():
    """Make the current stream wait for every side-stream launch issued so far (deferred gradient reductions included)."""
    _defer.flush()
    for st in _side.all_streams():
        torch.cuda.current_stream(st.device).wait_stream(st)
    _side.retire(everything=True)   # the current stream is now ordered after every side-stream read


def branch_stream(device):
    """The branch stream of `device` (see _SideStream), or None when there is no side stream now (overlap off, single-stream
    profiling, any hipGraph capture) or STIL_BRANCH_STREAM=0."""
    if not _side.use_branch or side_stream(device) is None or torch.cuda.is_current_stream_capturing():
        return None
    return _side.get_branch(device)


def side_stream(device):
    """The side stream of `device`, or None when overlap is off (or a capture runs with STIL_GRAPH_SIDE=0)."""
    if not _side.enabled or _prof_active():
        return None
    if not _side.in_capture and torch.cuda.is_current_stream_capturing():
        return None
    return _side.get(device)


# Parity instrumentation: when set to {"relu": {}, "pool": {}}, every ReLU output of a gradient-carrying pass and
# the max-pool winners are kept (by reference, no copy) under id(parameter) so that tests can replay the device's
# piecewise-linear decisions in their float64 CPU checker (tests/test_gpu_step.py).  None in production.
_trace = None


# data-parallel runs: the comm.GradExchange that is told about every gradient contribution as it is issued, so that
# finished buckets of the gradient slab can be all-reduced while backward is still running (None: single process)
_exchange = None


def _touch(param):
    """`param`'s gradient slot has just received a contribution (the launch is already enqueued)."""
    param._stil_touched = True
    if _exchange is not None:
        _exchange.note(param)


def cached_layout(param, attr: str, key=None):
    """The per-step layout view of a weight (layouts.WeightLayouts: one launch per step for the whole model), or None when
    the parameter has no plan or the plan is stale -- the caller then runs its own per-call layout kernel.  Stale = invalidated by
    a mutator that goes around PyTorch (Adam / EMA kernels: flat.invalidate_layouts), OR the slab / the parameter has been written
    through PyTorch since the refresh (their version counters moved: checkpoint restores, collectives, tests), OR the refresh
    was recorded inside a hipGraph capture and this call is not part of one (its event cannot be waited for eagerly)."""
    plan = getattr(param, "_stil_layouts", None)
    if plan is None or not plan.current(param):
        return None
    if plan.event is not None:      # refreshed on another stream: this stream waits for it once
        dev_ = plan.slab.device
        sk = (dev_.index, _raw_stream(dev_))
        if sk not in plan.waited:
            current_stream_obj(dev_).wait_event(plan.event)
            plan.waited.add(sk)
    v = getattr(param, attr, None)
    if v is None or key is None:
        return v
    return v.get(key)


def _grad_into(param: torch.Tensor, writer):
    """Run writer(dst, accumulate) for a parameter gradient.  Slab-backed parameters get "+=" into their slot
    (returns None for autograd); plain tensors get a fresh gradient tensor (returned)."""
    slot = getattr(param, "_gslot", None)
    if slot is not None:
        writer(slot, 1, True)
        _touch(param)
        return None
    g = torch.empty_like(param)
    writer(g, 0, False)
    return g


# ------------------------------------------------------------------------------------------ deferred reductions
class _DeferredReduce:
    """Small per-GPU batches under hipGraph replay are ~1000 dependent launches of ~5 us: the slab reduction that follows every
    weight-gradient GEMM and every bias column sum (137 launches of the cardiac step's 1055) is DEFERRED -- the products leave their
    partials in an arena, one StilReduceJob each is collected, and flush() (join_side(): before anything reads the gradient slab)
    finishes them all in ceil(n / 48) launches, bit for bit what the immediate kernels write (include/stil_hip.h).
    Only gradients that go to the slab (+=) are deferred, never under a GradExchange (it all-reduces buckets as they finish).
    STIL_REDUCE_DEFER = auto (default: inside driver.GraphedTrainStep's warm-up and capture) | 1 (always) | 0 (never)."""
    _JOB = __import__("struct").Struct("<QQ7if")

    def __init__(self):
        self.mode = __import__("os").environ.get("STIL_REDUCE_DEFER", "auto")
        self.forced = 0                 # > 0 inside `deferring()`
        self.ws = _Workspace()          # the arena's backing buffer, one per (device, stream), pinned under capture
        self.st = {}                    # key -> dict(stream, size, off, jobs, dsts)
        self.high = {}                  # device -> largest arena use so far (a new stream's arena starts at 1.25 x that:
                                        # GraphedTrainStep warms up on one stream and captures on another)

    def active(self) -> bool:
        if _exchange is not None or self.mode == "0":
            return False
        return self.mode == "1" or self.forced > 0

    def _state(self, device):
        key = (device, _raw_stream(device))
        d = self.st.get(key)
        if d is None:
            st = torch.cuda.current_stream(device)
            d = self.st[key] = dict(stream=st, device=device, size=max(256 << 20, int(1.25 * self.high.get(device, 0))), off=0, jobs=[], dsts=set())
        return d

    def alloc(self, nbytes: int, device) -> int:
        """-> device address of `nbytes` (256-byte aligned) that stay valid until this stream's next flush"""
        d = self._state(device)
        n = (int(nbytes) + 255) & ~255
        if d["off"] + n > d["size"]:
            self._flush_one(d)                     # the pending partials are consumed before their memory is reused / regrown
            d["size"] = max(d["size"], 2 * n, int(1.25 * self.high.get(device, 0)))
        buf = self.ws.get(d["size"], device)
        assert buf.numel() >= d["size"]
        a = buf.data_ptr() + d["off"]
        a += (-a) % 256
        d["off"] = a - buf.data_ptr() + n
        self.high[device] = max(self.high.get(device, 0), d["off"])
        return a

    def add(self, device, P: int, dst: torch.Tensor, splits, N, K, Cin, taps, Kdst, accumulate, scale):
        d = self._state(device)
        if dst.data_ptr() in d["dsts"]:            # two contributions to one slot: in order
            self._flush_one(d)
        d["dsts"].add(dst.data_ptr())
        d["jobs"].append(self._JOB.pack(P, dst.data_ptr(), splits, N, K, Cin, taps, Kdst, accumulate, float(scale)))

    def _flush_one(self, d):
        if d["jobs"]:
            L = lib()
            assert L.reduce_job_bytes() == self._JOB.size, "StilReduceJob layout changed"
            blob = b"".join(d["jobs"])
            with torch.cuda.stream(d["stream"]):
                L.reduce_jobs(blob, len(d["jobs"]), d["stream"].cuda_stream)
        d["jobs"], d["dsts"], d["off"] = [], set(), 0

    def flush(self):
        for d in self.st.values():
            self._flush_one(d)


_defer = _DeferredReduce()


class deferring:
    """with ops.deferring(): gradient reductions of the enclosed steps are deferred (driver.GraphedTrainStep's warm-up and capture)"""

    def __enter__(self):
        _defer.forced += 1

    def __exit__(self, *exc):
        _defer.forced -= 1


# ------------------------------------------------------------------------------------------ raw wrappers
# Per-shape k-tile policy of the NT products (`tune` of stil_gemm_nt), chosen from tests/tools/gemm_lab.hip's table of the step's
# shapes and confirmed on the whole step (profiles/r05_experiments.txt): STIL_GEMM_POLICY = "" (the library's automatic choice
# everywhere) | "k2048:200" style rules "k<minK>:<tune>" applied to products without operand-staging BatchNorm.
_POLICY = []
for _r in __import__("os").environ.get("STIL_GEMM_POLICY", "").split(","):
    if _r.strip():
        _k, _t = _r.strip().split(":")
        _POLICY.append((int(_k[1:]), int(_t)))
_POLICY.sort(reverse=True)


def _shape_tune(M, N, K, a_bn, has_tile_stats):
    """The `tune` argument of this product: TUNE["gemm"] when forced (measurement tools), else the first policy rule it meets."""
    t = TUNE["gemm"]
    if t % B3_FLAG or a_bn or not _POLICY:
        return t
    for mink, tune in _POLICY:
        if K >= mink:
            if tune % 100 not in (0, 11, 44) and (has_tile_stats or M % 128 or N % 128):   # per-tile statistics (64-row tiles) / ragged tiles stay on 64x64
                return t
            return tune
    return t


def gemm_nt(A, W, M, N, K, *, lda=None, ldb=None, out=None, ldc=None, geom=None, bias=None, sub=None, scale=None, shift=None,
            resid=None, pre=None, act=0, alpha=1.0, pads=None, outmap=None, out_rows=None, colstats=None, a_bn=None, relu_mask=None,
            bstats=None, scale_var=None, var_eps=0.0):
    """C = epilogue(alpha * Agather . W^T).  geom = (srcH, srcW, srcC, OH, OW, KH, KW, stride, pad, mode);
    pads = (pad_y, pad_x) overrides geom's pad; outmap = (out_stride, py, px, out_OH, out_OW) scatters GEMM row
    (n, oy, ox) to output row (n*out_OH + oy*s + py)*out_OW + ox*s + px (out_rows = rows of `out` then).
    bstats = (y, stats, partials, relu_mode, tile0): the epilogue also leaves the BatchNorm-backward partial sums of the layer
    whose output gradient this launch produces (include/stil_hip.h `bstats`)."""
    if geom is None:
        geom = (1, 1, K, 1, 1, 1, 1, 1, 0, 0)
    pads = (geom[8], geom[8]) if pads is None else pads
    outmap = (1, 0, 0, geom[3], geom[4]) if outmap is None else outmap
    lda = geom[2] if lda is None else lda
    ldb = K if ldb is None else ldb
    ldc = N if ldc is None else ldc
    if out is None:
        out = torch.empty((M if out_rows is None else out_rows, N), dtype=torch.float32, device=A.device)
    L = lib()
    meta = None
    tune = _shape_tune(M, N, K, a_bn is not None, bstats is not None or colstats is not None)
    if L._prof is not None:  # bench bookkeeping: tile variant + ALGORITHMIC flops (strided dgrad gathers count the conv's flops)
        s2 = geom[7] * geom[7] if geom[9] == 1 else 1
        src = (M // (geom[3] * geom[4])) * geom[0] * geom[1] * geom[2]  # gather source (each element fetched once, ideally)
        # every operand once: gather source, weights, the output, and the epilogue's reads / extra stores (residual, ReLU mask of the
        # block output, the raw conv output the BatchNorm-backward sums are taken against, the stored pre-activation)
        nbytes = 4.0 * (src + N * K + M * N * (1 + (resid is not None) + (pre is not None) + (relu_mask is not None) + (bstats is not None)))
        plain = int(geom[5] * geom[6] == 1 and geom[7] == 1 and pads == (0, 0) and geom[9] == 0 and outmap[0] == 1
                    and geom[0] == geom[3] and geom[1] == geom[4])
        cfg = L.gemm_nt_config(_p(A), _p(W), M, N, K, lda, ldb, geom[2], geom[5], geom[6], plain, int(a_bn is not None), tune)
        meta = (cfg, 2.0 * M * N * K / s2, (M, N, K, geom[5], geom[7], geom[9]), nbytes)
    sw, swn = None, 0
    if _SPLITK:
        swn = L.gemm_nt_split_workspace_bytes(M, N, K, tune)
        if swn:
            sw = _split_ws.get(swn, A.device)
    L.gemm_nt(_p(A), _p(W), _p(out), M, N, K, lda, ldb, ldc, *geom[:8], pads[0], pads[1], geom[9], *outmap,
              _p(bias), _p(sub), _p(scale), _p(shift), _p(resid),
              (ldc if resid is not None else 0), _p(pre), act, float(alpha), _p(colstats), _p(a_bn), _p(relu_mask),
              (ldc if relu_mask is not None else 0), *((_p(bstats[0]), _p(bstats[1]), _p(bstats[2]), int(bstats[3]), int(bstats[4]))
                                                       if bstats is not None else (None, None, None, 0, 0)),
              _p(scale_var), float(var_eps), _p(sw), (sw.numel() if sw is not None else 0), tune, _stream(), meta=meta)
    return out


def wgrad_tn(dY, X, dW, M, N, K, *, ldy=None, ldx=None, geom=None, Kdst=None, accumulate=0, x_bn=None, slot=False, stream=None):
    """`slot`: dW is a view of the gradient slab -- its slab reduction may be deferred to the next join_side() (_DeferredReduce).
    `stream` (a torch Stream, not with a deferred reduction): launch there instead of on the current stream -- the caller orders it."""
    if geom is None:
        geom = (1, 1, K, 1, 1, 1, 1, 1, 0)
    ldy = N if ldy is None else ldy
    ldx = geom[2] if ldx is None else ldx
    Kdst = K if Kdst is None else Kdst
    L = lib()
    nb = L.wgrad_workspace_bytes(M, N, K, TUNE["wgrad"])
    meta = (0, 2.0 * M * N * K, (M, N, K, geom[5], geom[7], 2)) if L._prof is not None else None
    if slot and _defer.active():
        a = _defer.alloc(nb, dY.device)
        L.wgrad_tn_partial(_p(dY), _p(X), M, N, K, ldy, ldx, *geom, _p(x_bn), a, nb, TUNE["wgrad"], _stream(), meta=meta)
        _defer.add(dY.device, a, dW, L.wgrad_splits(M, N, K, TUNE["wgrad"]), N, K, geom[2], geom[5] * geom[6], Kdst, accumulate, 1.0)
        return
    w = _ws.get(nb, dY.device, stream)
    L.wgrad_tn(_p(dY), _p(X), _p(dW), M, N, K, ldy, ldx, *geom, Kdst, accumulate, _p(x_bn), _p(w), nb, TUNE["wgrad"],
               _stream() if stream is None else stream.cuda_stream, meta=meta)


# STIL_WGRAD_SIDE=0: weight gradients stay on the main stream while the EMA teacher keeps the side stream (A/B measurements of the
# two-stream structure: does co-running two MFMA-bound kernels pay? profiles/r05_experiments.txt)
_WGRAD_SIDE = __import__("os").environ.get("STIL_WGRAD_SIDE", "1") != "0"


def wgrad_param(param, dY, X, M, N, K, **kw):
    """Weight gradient of a parameter.  Slab-backed parameters: "+=" into the slot on the side stream (returns None);
    plain tensors: fresh gradient on the current stream (returned to autograd)."""
    slot = getattr(param, "_gslot", None)
    side = side_stream(dY.device) if (slot is not None and _WGRAD_SIDE) else None
    if side is None:
        return _grad_into(param, lambda dst, acc, slot_: wgrad_tn(dY, X, dst, M, N, K, accumulate=acc, slot=slot_, **kw))
    main = current_stream_obj(dY.device)
    _side.retire()
    if main.cuda_stream == side.cuda_stream or _defer.active():
        if main != side:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            wgrad_tn(dY, X, slot, M, N, K, accumulate=1, slot=True, **kw)
            if main != side:
                ev = torch.cuda.Event()
                ev.record(side)
                _side.keep.append((ev, (dY, X, kw.get("x_bn"))))
    else:   # the launch goes to the side stream by handle: no switch of torch's current stream (~10 us of host time per product)
        fork = torch.cuda.Event()
        fork.record(main)
        side.wait_event(fork)
        wgrad_tn(dY, X, slot, M, N, K, accumulate=1, slot=True, stream=side, **kw)
        ev = torch.cuda.Event()
        ev.record(side)
        _side.keep.append((ev, (dY, X, kw.get("x_bn"))))
    _touch(param)
    return None


def _prof_active():
    L = lib()
    return L._prof is not None and L._prof_single  # single-stream profiling: keep every launch on the caller's stream


def colsum(X, out, M, N, *, ld=None, accumulate=0, scale=1.0, slot=False):
    L = lib()
    nb = L.colsum_workspace_bytes(M, N)
    if slot and _defer.active():    # see wgrad_tn
        a = _defer.alloc(nb, X.device)
        L.colsum_partial(_p(X), M, N, N if ld is None else ld, a, nb, _stream())
        _defer.add(X.device, a, out, L.colsum_chunks(M), 1, N, N, 1, N, accumulate, scale)
        return
    w = _ws.get(nb, X.device)
    L.colsum(_p(X), _p(out), M, N, N if ld is None else ld, accumulate, float(scale), _p(w), nb, _stream())


def transpose(x2d):
    R, C = x2d.shape
    out = torch.empty((C, R), dtype=torch.float32, device=x2d.device)
    lib().transpose(_p(x2d), _p(out), R, C, _stream())
    return out


def axpby(x, y, a, b, out=None):
    out = torch.empty_like(x) if out is None else out
    lib().axpby(_p(x), _p(y), _p(out), x.numel(), float(a), float(b), _stream())
    return out


def rng_mask(shape, p, seed, offset, device, step=None):
    out = torch.empty(shape, dtype=torch.uint8, device=device)
    lib().rng_mask(_p(out), out.numel(), int(seed), int(offset), float(p), _p(step), _stream())
    return out


# ------------------------------------------------------------------------------------------ Linear
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) [+ resid]   nn.Linear (+ReLU / +GELU) -- e.g. models/Transformer.py:27-33; `resid` (act == 0 only): the
    residual sum of a pre-LN transformer block rides in the GEMM epilogue (models/Transformer.py:170-173: x = x + attn(...),
    x = x + mlp(...)) instead of in a pass of its own; same roundings in the same order."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, resid=None):
        _chk(x, weight, bias, resid)
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        assert resid is None or (act == 0 and resid.numel() == M * N), "LinearFn: a residual needs act == 0 and the output's shape"
        pre = torch.empty((M, N), dtype=torch.float32, device=x.device) if act == 2 else None
        y = gemm_nt(x2, weight, M, N, K, bias=bias, pre=pre, act=act, resid=None if resid is None else resid.reshape(M, N))
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.has_resid = resid is not None
        if _trace is not None and act == 1 and any(ctx.needs_input_grad):
            _trace["relu"][id(weight)] = y.view(*x.shape[:-1], N)
        ctx.save_for_backward(x2, weight, bias, pre if act == 2 else (y if act == 1 else None))
        ctx.xshape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x2, weight, bias, ref = ctx.saved_tensors
        M, K = x2.shape
        N = weight.shape[0]
        g = gy.contiguous().reshape(M, N)
        if ctx.act:
            gp = torch.empty_like(g)
            lib().act_bwd(_p(g), _p(ref), _p(gp), g.numel(), ctx.act, _stream())
            g = gp
        dx = None
        if ctx.needs_input_grad[0]:
            wt = cached_layout(weight, "_stil_wd")  # [K, N]
            if wt is None:
                wt = transpose(weight)
            dx = gemm_nt(g, wt, M, K, N).reshape(ctx.xshape)
        dw = wgrad_param(weight, g, x2, M, N, K) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _grad_into(bias, lambda dst, acc, slot_: colsum(g, dst, M, N, accumulate=acc, slot=slot_))
        dres = gy if (ctx.has_resid and ctx.needs_input_grad[4]) else None   # the residual branch: identity
        return dx, dw, db, None, dres


def linear(x, weight, bias=None, act=0, resid=None):
    return LinearFn.apply(x, weight, bias, act, resid)


class MatmulNTFn(torch.autograd.Function):
    """Z = alpha * A B^T with gradients to both operands (CLIP logits, utils/clip_loss.py:34)."""

    @staticmethod
    def forward(ctx, A, B, alpha):
        _chk(A, B)
        M, K = A.shape
        N = B.shape[0]
        ctx.save_for_backward(A, B)
        ctx.alpha = alpha
        return gemm_nt(A, B, M, N, K, alpha=alpha)

    @staticmethod
    def backward(ctx, gZ):
        A, B = ctx.saved_tensors
        M, K = A.shape
        N = B.shape[0]
        gZ = gZ.contiguous()
        dA = gemm_nt(gZ, transpose(B), M, K, N, alpha=ctx.alpha) if ctx.needs_input_grad[0] else None            # gZ [M,N] . B [N,K]
        dB = gemm_nt(transpose(gZ), transpose(A), N, K, M, alpha=ctx.alpha) if ctx.needs_input_grad[1] else None  # gZ^T [N,M] . A [M,K]
        return dA, dB, None


# ------------------------------------------------------------------------------------------ conv + BN (+res) (+relu)
def _conv_geom_fwd(H, W, C, OH, OW, k, stride, pad):
    return (H, W, C, OH, OW, k, k, stride, pad, 0)


_BN_EPILOGUE_STATS = __import__("os").environ.get("STIL_BN_EPILOGUE_STATS", "1") != "0"


_BN_DEFER = __import__("os").environ.get("STIL_BN_DEFER", "1") != "0"
_PREMASK = __import__("os").environ.get("STIL_PREMASK", "1") != "0"   # residual gradients leave the next block's dgrad GEMM pre-masked
_BN_BWD_EPILOGUE = __import__("os").environ.get("STIL_BN_BWD_EPILOGUE", "1") != "0"   # BatchNorm-backward sums in the producing GEMM's epilogue
_LINEAR_RESID = __import__("os").environ.get("STIL_LINEAR_RESID", "1") != "0"   # transformer residual sums in the Linear GEMM's epilogue


def can_defer_bn(Cout: int) -> bool:
    """May a conv+BN+ReLU layer with Cout channels leave its BatchNorm + ReLU to the consumer's operand staging?
    (the consumer's GEMM needs Cin % 16 == 0; STIL_BN_DEFER=0 materialises every z as before)"""
    return _BN_DEFER and _BN_EPILOGUE_STATS and Cout % 16 == 0 and Cout <= 2048


class ConvBnActFn(torch.autograd.Function):
    """z = relu?( BN_train(conv(x, w)) + residual? ) on NHWC activations.

    One node per conv+BN pair of models/resnets.py:112-132 (Bottleneck) / 71-88 (BasicBlock).
    x: [N,H,W,Cin] (or the stem's im2col matrix when `stem` is given).
    """

    @staticmethod
    def forward(ctx, x, w, gamma, beta, rmean, rvar, nbt, resid, k, stride, pad, relu, stem, passthrough=False, defer=False, xstats=None,
                premask_in=None, premask_out=None, rstats=None, bstat_send=None, bstat_recv=None):
        """passthrough: also return x itself as a second output.  A residual block routes its identity branch through
        that alias, so the branch's gradient arrives HERE and is added inside the input-gradient GEMM's epilogue
        instead of by a separate accumulation kernel of the autograd engine.
        defer (inner layers of a residual block: ReLU, no residual): BatchNorm + ReLU are NOT applied here -- the node
        returns the raw conv output y and, as a last extra output, its statistics block [4, Cout]; the consumer conv
        passes them back as `xstats` and forms z = relu((y - mean) * a + beta) while it stages its operand (forward
        GEMM and weight-gradient GEMM: stil_gemm_nt a_bn / stil_wgrad_tn x_bn), so z is never written or read.  The
        gradient this node receives is still the one w.r.t. z: its backward is unchanged (the ReLU mask is recomputed
        from y and the statistics, bn_train_bwd relu = 2).
        premask_out / premask_in (a shared dict per block output z = relu(bn3(y3) + identity)): the block's last node
        publishes the dict, the NEXT block's first node -- the only consumer of z -- receives it.  That node's input-gradient
        GEMM then writes dL/dz already multiplied by the ReLU mask (z > 0), read in its 16-byte epilogue (stil_gemm_nt
        relu_mask), and says so in the dict; this node's BatchNorm backward then takes the gradient as it comes: it neither
        reads z nor materialises the masked gradient for the identity branch (one wide-tensor pass fewer per block).
        bstat_recv / bstat_send (a shared dict per conv+BN layer whose output has ONE consumer conv): the layer publishes its raw
        output y and statistics in bstat_recv; its consumer gets the same dict as bstat_send and, in backward, lets its
        input-gradient GEMM's epilogue leave the layer's BatchNorm-backward partial sums (sum g', sum g' * xhat per 64-row tile:
        stil_gemm_nt `bstats`) in the dict -- the layer's backward then skips the reduction pass over g and y
        (stil_bn_train_bwd_tiles).  Falls back to the pass whenever the consumer could not fuse (no cell filled)."""
        _chk(x, w, gamma, beta, resid)
        ctx.set_materialize_grads(False)
        # deferred layers: the inner conv+BN+ReLU of a block (applied by the next conv's operand staging) and the shortcut's
        # conv+BN (no ReLU; applied inside the block's final BatchNorm pass, `rstats` there)
        assert not defer or resid is None, "a layer with a residual input cannot defer its BatchNorm"
        Cout = w.shape[0]
        dev = x.device
        fused = _BN_EPILOGUE_STATS and Cout % 4 == 0
        if stem is not None:  # x is col [M, Kp]; stem = (N, OH, OW, Kp, wpad)
            Nb, OH, OW, Kp, wpad = stem
            M = Nb * OH * OW
        else:
            Nb, H, W_, Cin = x.shape
            OH = (H + 2 * pad - k) // stride + 1
            OW = (W_ + 2 * pad - k) // stride + 1
            M = Nb * OH * OW
        ts = tile_rows = None
        if fused:  # the GEMM epilogue leaves per-tile (mean, M2) partials: no statistics pass over y
            tile_rows = lib().gemm_nt_tile_rows(M, Cout, TUNE["gemm"])
            ts = torch.empty((2 * ((M + tile_rows - 1) // tile_rows), Cout), dtype=torch.float32, device=dev)
        if stem is not None:
            y = gemm_nt(x, wpad, M, Cout, Kp, colstats=ts)
            wf = None
            geom = None
        else:
            if k == 1:
                wf = w.reshape(Cout, Cin)
            else:
                wf = cached_layout(w, "_stil_wf")
                if wf is None:
                    wf = torch.empty((Cout, k * k * Cin), dtype=torch.float32, device=dev)
                    lib().conv_weight_layout(_p(w), _p(wf), None, Cout, Cin, k, k, _stream())
            geom = _conv_geom_fwd(H, W_, Cin, OH, OW, k, stride, pad)
            y = gemm_nt(x, wf, M, Cout, k * k * Cin, geom=geom, colstats=ts, a_bn=xstats)
        stats = torch.empty((4, Cout), dtype=torch.float32, device=dev)
        assert fused or not defer, "deferred BatchNorm needs the epilogue statistics (can_defer_bn)"
        # deferred: no z.  Under parity tracing the tests still want this layer's ReLU decisions: z is materialised for them
        z = None if (defer and _trace is None) else torch.empty((M, Cout), dtype=torch.float32, device=dev)
        if fused:
            nb = lib().bn_tiles_workspace_bytes(M, Cout, tile_rows)
            ws = _ws.get(nb, dev)
            lib().bn_train_fwd_tiles(_p(y), _p(ts), tile_rows, _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(nbt), _p(resid), _p(rstats), _p(z),
                                     _p(stats), M, Cout, 1 if relu else 0, 1e-5, 0.1, _p(ws), nb, _stream())
        else:
            assert rstats is None
            nb = lib().bn_workspace_bytes(M, Cout)
            ws = _ws.get(nb, dev)
            lib().bn_train_fwd(_p(y), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(nbt), _p(resid), _p(z), _p(stats), M, Cout,
                               1 if relu else 0, 1e-5, 0.1, _p(ws), nb, _stream())
        if _trace is not None and relu:
            _trace["relu"][id(gamma)] = z.view(Nb, OH, OW, Cout)
        ctx.save_for_backward(x, w, gamma, beta, y, None if defer else z, stats, xstats)
        ctx.cfg = (k, stride, pad, relu, stem is not None, resid is not None, geom, (Nb, OH, OW), stem)
        ctx.has_alias = bool(passthrough)
        ctx.premask_in = premask_in if (stem is None and stride == 1) else None     # only the plain / stride-1 gather dgrad GEMMs mask
        ctx.premask_out = premask_out if (relu and resid is not None and not defer) else None
        ctx.bstat_send = bstat_send if (_BN_BWD_EPILOGUE and stem is None) else None
        ctx.bstat_recv = None
        if _BN_BWD_EPILOGUE and bstat_recv is not None and Cout % 4 == 0:
            # mode 2: inner layer (ReLU mask recomputed from y); mode 0: block output whose gradient arrives pre-masked
            if defer and relu and resid is None:
                bstat_recv.update(y=y, stats=stats, mode=2)
                ctx.bstat_recv = bstat_recv
            elif relu and resid is not None and not defer and premask_out is not None:
                bstat_recv.update(y=y, stats=stats, mode=0, premask=premask_out)
                ctx.bstat_recv = bstat_recv
        out = (y if defer else z).view(Nb, OH, OW, Cout)
        if defer:
            ctx.mark_non_differentiable(stats)
            return (out, x, stats) if passthrough else (out, stats)
        if passthrough:
            return out, x
        return out

    @staticmethod
    def backward(ctx, gz, *more):
        x, w, gamma, beta, y, z, stats, xstats = ctx.saved_tensors
        # outputs: (out[, x alias][, stats]) -- the alias gradient is the only other one that can carry a value
        gx_alias = more[0] if (len(more) and ctx.has_alias) else None
        k, stride, pad, relu, is_stem, has_res, geom, (Nb, OH, OW), stem = ctx.cfg
        dev = x.device
        Cout = w.shape[0]
        M = Nb * OH * OW
        gz = gz.contiguous()
        dy = torch.empty((M, Cout), dtype=torch.float32, device=dev)
        premasked = ctx.premask_out is not None and ctx.premask_out.get("masked", False)   # gz already carries the (z > 0) mask
        gres = torch.empty((M, Cout), dtype=torch.float32, device=dev) if (has_res and relu and not premasked) else None
        coef = torch.empty((3, Cout), dtype=torch.float32, device=dev)
        nb = lib().bn_workspace_bytes(M, Cout)
        ws = _ws.get(nb, dev)
        gslot, bslot = getattr(gamma, "_gslot", None), getattr(beta, "_gslot", None)
        dgamma = gslot if gslot is not None else torch.empty_like(gamma)
        dbeta = bslot if bslot is not None else torch.empty_like(beta)
        acc = 1 if gslot is not None else 0
        relu_mode = 0 if premasked else ((1 if has_res else 2) if relu else 0)
        cell = ctx.bstat_recv
        part = cell.get("part") if cell is not None else None
        if part is not None and gres is None and tuple(part.shape) == (2 * cell["nt"], Cout) and (cell["mode"] == 2) == (relu_mode == 2) \
                and (cell["mode"] != 0 or premasked):
            # the consumer's input-gradient GEMM left the per-tile sums: no reduction pass over gz and y
            nb2 = lib().bn_bwd_tiles_workspace_bytes(cell["nt"], Cout)
            ws2 = _ws.get(nb2 + 8, dev)
            lib().bn_train_bwd_tiles(_p(gz), _p(z), _p(y), _p(gamma), _p(stats), _p(part), cell["nt"], _p(dy), _p(dgamma), _p(dbeta), _p(coef), M,
                                     Cout, relu_mode, acc, _p(ws2), nb2, _stream())
        else:
            lib().bn_train_bwd(_p(gz), _p(z), _p(y), _p(gamma), _p(stats), _p(dy), _p(gres), _p(dgamma), _p(dbeta), _p(coef), M,
                               Cout, relu_mode, acc, _p(ws), nb, _stream())
        if gslot is not None:
            _touch(gamma)
            _touch(beta)
        if has_res:
            dres = gres if (relu and not premasked) else gz.view(M, Cout)
        else:
            dres = None
        dx = None
        if is_stem:
            Kp = stem[3]
            Kreal = w.shape[1] * k * k
            dw = wgrad_param(w, dy, x, M, Cout, Kp, Kdst=Kreal)
        else:
            _, H, W_, Cin = x.shape
            if ctx.needs_input_grad[0]:
                ga = None if gx_alias is None else gx_alias.contiguous().view(Nb * H * W_, Cin)
                # x is the previous block's output z = relu(.): its consumer is this node alone, so the gradient can leave
                # here already masked (the identity branch's share arrives through `ga` and is masked with it)
                cell = ctx.premask_in if (ctx.premask_in is not None and xstats is None and Cin % 4 == 0 and stride == 1) else None
                zmask = x.view(Nb * H * W_, Cin) if cell is not None else None
                if cell is not None:
                    cell["masked"] = True   # both stride-1 branches below apply the mask
                # BatchNorm-backward partial sums of the layer that produced x, left by this GEMM's epilogue (bstat_send)
                bs = None
                sc_ = ctx.bstat_send
                Min = Nb * H * W_
                if sc_ is not None and "y" in sc_ and Cin % 4 == 0 and _bstats_tune_ok() and (sc_["mode"] == 2 or (zmask is not None and sc_.get("premask") is cell)) \
                        and (stride == 1 or ga is None) and tuple(sc_["y"].shape) == (Min, Cin) \
                        and lib().gemm_nt_bstats_ok(None, Cin, Cin, _p(ga), Cin, _p(zmask), Cin, _p(sc_["y"])):
                    nt_ = _bstat_tiles(Nb, H, W_, k, stride, pad)
                    bs = (sc_["y"], sc_["stats"], torch.empty((2 * nt_, Cin), dtype=torch.float32, device=dev), sc_["mode"], 0)
                if k == 1 and stride == 1:
                    wd = cached_layout(w, "_stil_wd")  # [Cin, Cout]
                    if wd is None:
                        wd = transpose(w.reshape(Cout, Cin))
                    dx = gemm_nt(dy, wd, M, Cin, Cout, resid=ga, relu_mask=zmask, bstats=bs).view(Nb, H, W_, Cin)
                elif stride == 1:
                    wd = cached_layout(w, "_stil_wd")
                    if wd is None:
                        wd = torch.empty((Cin, k * k * Cout), dtype=torch.float32, device=dev)
                        lib().conv_weight_layout(_p(w), None, _p(wd), Cout, Cin, k, k, _stream())
                    g2 = (OH, OW, Cout, H, W_, k, k, stride, pad, 1)
                    dx = gemm_nt(dy, wd, Nb * H * W_, Cin, k * k * Cout, geom=g2, resid=ga, relu_mask=zmask, bstats=bs).view(Nb, H, W_, Cin)
                else:
                    dx = strided_dgrad(dy, w, Nb, H, W_, Cin, OH, OW, Cout, k, stride, pad, bstats=bs)
                    if ga is not None:
                        axpby(dx.view(-1), ga.view(-1), 1.0, 1.0, out=dx.view(-1))
                if bs is not None:
                    sc_["part"], sc_["nt"] = bs[2], bs[2].shape[0] // 2
            elif gx_alias is not None:
                dx = gx_alias
            gw = geom[:9]
            dw = wgrad_param(w, dy, x, M, Cout, k * k * Cin, geom=gw, x_bn=xstats)
        return (dx, dw, (None if gslot is not None else dgamma), (None if bslot is not None else dbeta), None, None, None,
                dres, None, None, None, None, None, None, None, None, None, None, None, None, None)


def _bstats_tune_ok() -> bool:
    """`bstats` rides in the 16-byte epilogue of 64x64 tiles only (stil_gemm_nt's STIL_REQUIRE): with a forced tile variant or the
    scalar epilogue (STIL_GEMM_TUNE, A/B measurements) the caller must take bn_train_bwd's own reduction pass instead."""
    t = TUNE["gemm"] % B3_FLAG
    return t < 10000 and t % 100 in (0, 11, 44)


def _bstat_tiles(Nb, H, W_, k, stride, pad):
    """64-row tiles of the input-gradient launch(es) of a conv over an [Nb, H, W_] input: one launch for stride 1, one per
    tap-owning output phase otherwise (strided_dgrad) -- the row count of a `bstats` partial array is twice this."""
    if stride == 1:
        return (Nb * H * W_ + 63) // 64
    n = 0
    for py in range(stride):
        if len(range((py + pad) % stride, k, stride)) == 0 or len(range(py, H, stride)) == 0:
            continue
        for px in range(stride):
            if len(range((px + pad) % stride, k, stride)) == 0 or len(range(px, W_, stride)) == 0:
                continue
            n += (Nb * len(range(py, H, stride)) * len(range(px, W_, stride)) + 63) // 64
    return n


def strided_dgrad(dy, w, Nb, H, W_, Cin, OH, OW, Cout, k, stride, pad, bstats=None):
    """Input gradient of a stride-s conv as s*s stride-1 gathers, one per output phase (py, px): phase pixels
    (s*oy'+py, s*ox'+px) only see taps ky = ky0 + s*j with ky0 = (py+pad) % s, so no multiply-by-zero work is done
    (the generic mode-1 gather computes s*s times the algorithmic MACs)."""
    dev = dy.device
    # every output pixel belongs to exactly one phase; phases without taps (1x1 s2: 3 of 4) must read as zero
    all_covered = all(len(range((p_ + pad) % stride, k, stride)) > 0 for p_ in range(stride))
    dx = (torch.empty if all_covered else torch.zeros)((Nb * H * W_, Cin), dtype=torch.float32, device=dev)
    tile0 = 0    # bstats: the phases fill consecutive tile ranges of ONE partial array (pixels no phase owns have a zero gradient)
    for py in range(stride):
        ky0 = (py + pad) % stride
        KHs = len(range(ky0, k, stride))
        Hs = len(range(py, H, stride))
        if KHs == 0 or Hs == 0:
            continue
        dy0 = (py + pad - ky0) // stride
        for px in range(stride):
            kx0 = (px + pad) % stride
            KWs = len(range(kx0, k, stride))
            Ws = len(range(px, W_, stride))
            if KWs == 0 or Ws == 0:
                continue
            dx0 = (px + pad - kx0) // stride
            wsub = cached_layout(w, "_stil_wphase", (py, px))
            if wsub is None:
                wsub = torch.empty((Cin, KHs * KWs * Cout), dtype=torch.float32, device=dev)
                lib().conv_weight_layout_phase(_p(w), _p(wsub), Cout, Cin, k, k, stride, ky0, kx0, KHs, KWs, _stream())
            # iy = oy' + dy0 - jy = oy' - pad' + ky'  with ky' = KHs-1-jy, pad' = KHs-1-dy0
            geom = (OH, OW, Cout, Hs, Ws, KHs, KWs, 1, 0, 0)
            gemm_nt(dy, wsub, Nb * Hs * Ws, Cin, KHs * KWs * Cout, geom=geom, out=dx, pads=(KHs - 1 - dy0, KWs - 1 - dx0),
                    outmap=(stride, py, px, H, W_), bstats=None if bstats is None else (*bstats[:4], tile0))
            tile0 += (Nb * Hs * Ws + 63) // 64
    return dx.view(Nb, H, W_, Cin)


def conv_bn_eval(x, w, gamma, beta, rmean, rvar, resid, k, stride, pad, relu, stem=None):
    """Teacher path: eval-mode BN folded into the conv epilogue (no grad): (y - running_mean) * (weight / sqrt(running_var + eps)) + bias,
    the scale formed in the epilogue itself (stil_gemm_nt scale_var) -- no per-layer affine kernel."""
    Cout = w.shape[0]
    dev = x.device
    bn = dict(sub=rmean, scale=gamma, scale_var=rvar, var_eps=1e-5, shift=beta)
    if stem is not None:
        Nb, OH, OW, Kp, wpad = stem
        M = Nb * OH * OW
        z = gemm_nt(x, wpad, M, Cout, Kp, resid=resid, act=1 if relu else 0, **bn)
        return z.view(Nb, OH, OW, Cout)
    Nb, H, W_, Cin = x.shape
    OH = (H + 2 * pad - k) // stride + 1
    OW = (W_ + 2 * pad - k) // stride + 1
    M = Nb * OH * OW
    if k == 1:
        wf = w.reshape(Cout, Cin)
    else:
        wf = cached_layout(w, "_stil_wf")
        if wf is None:
            wf = torch.empty((Cout, k * k * Cin), dtype=torch.float32, device=dev)
            lib().conv_weight_layout(_p(w), _p(wf), None, Cout, Cin, k, k, _stream())
    z = gemm_nt(x, wf, M, Cout, k * k * Cin, geom=_conv_geom_fwd(H, W_, Cin, OH, OW, k, stride, pad), resid=resid, act=1 if relu else 0, **bn)
    return z.view(Nb, OH, OW, Cout)


def im2col_stem(x_nchw, k, stride, pad):
    """NCHW image -> (col [M, Kp], (N, OH, OW, Kp)); K padded to a multiple of 16."""
    _chk(x_nchw)
    Nb, Cin, H, W_ = x_nchw.shape
    OH = (H + 2 * pad - k) // stride + 1
    OW = (W_ + 2 * pad - k) // stride + 1
    Kp = ((Cin * k * k + 15) // 16) * 16
    col = torch.empty((Nb * OH * OW, Kp), dtype=torch.float32, device=x_nchw.device)
    lib().im2col_nchw(_p(x_nchw), _p(col), Nb, Cin, H, W_, OH, OW, k, k, stride, pad, Kp, _stream())
    return col, (Nb, OH, OW, Kp)


def pad_stem_weight(w, Kp):
    Cout = w.shape[0]
    wp = torch.zeros((Cout, Kp), dtype=torch.float32, device=w.device)
    wp[:, : w[0].numel()] = w.detach().reshape(Cout, -1)
    return wp


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) on NHWC (models/resnets.py:252)."""

    @staticmethod
    def forward(ctx, x):
        _chk(x)
        Nb, H, W_, C = x.shape
        OH, OW = (H + 2 - 3) // 2 + 1, (W_ + 2 - 3) // 2 + 1
        y = torch.empty((Nb, OH, OW, C), dtype=torch.float32, device=x.device)
        idx = torch.empty((Nb, OH, OW, C), dtype=torch.uint8, device=x.device)
        lib().maxpool3x3s2_fwd(_p(x), _p(y), _p(idx), Nb, H, W_, C, OH, OW, _stream())
        if _trace is not None and ctx.needs_input_grad[0]:
            _trace["pool"]["maxpool"] = (idx, H, W_)
        ctx.save_for_backward(idx)
        ctx.shape = (Nb, H, W_, C, OH, OW)
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        Nb, H, W_, C, OH, OW = ctx.shape
        gy = gy.contiguous()
        dx = torch.empty((Nb, H, W_, C), dtype=torch.float32, device=gy.device)
        lib().maxpool3x3s2_bwd(_p(gy), _p(idx), _p(dx), Nb, H, W_, C, OH, OW, _stream())
        return dx


# ------------------------------------------------------------------------------------------ transformer pieces
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        _chk(x, gamma, beta)
        D = x.shape[-1]
        rows = x.numel() // D
        y = torch.empty_like(x)
        mr = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        lib().layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mr), rows, D, 1e-5, _stream())
        ctx.save_for_backward(x, gamma, beta, mr)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mr = ctx.saved_tensors
        D = x.shape[-1]
        rows = x.numel() // D
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        nb = lib().layernorm_bwd_workspace_bytes(rows, D)
        ws = _ws.get(nb, x.device)
        gslot, bslot = getattr(gamma, "_gslot", None), getattr(beta, "_gslot", None)
        dg = gslot if gslot is not None else torch.empty_like(gamma)
        db = bslot if bslot is not None else torch.empty_like(beta)
        lib().layernorm_bwd(_p(gy), _p(x), _p(gamma), _p(mr), _p(dx), _p(dg), _p(db), rows, D, 1 if gslot is not None else 0,
                            _p(ws), nb, _stream())
        if gslot is not None:
            _touch(gamma)
            _touch(beta)
            return dx, None, None
        return dx, dg, db


def layernorm(x, gamma, beta):
    return LayerNormFn.apply(x, gamma, beta)


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T * scale) (dropout) v for a list of (q_off, Sq, kv_off, Skv) token windows sharing one qkv buffer.

    qkv: [B, T, 3*H*d] laid out (3, H, d) per token, exactly nn.Linear(dim, 3*dim)'s output
    (models/Transformer.py:66, disentangle_transformer.py:54-62).  Output [B, T, H*d].
    """

    @staticmethod
    def forward(ctx, qkv, H, windows, masks, drop_p):
        _chk(qkv)
        B, T, three = qkv.shape
        d = three // (3 * H)
        scale = d ** -0.5
        covered = sum(w_[1] for w_ in windows) == T  # query windows are disjoint: every output row written once
        out = (torch.empty if covered else torch.zeros)((B, T, H * d), dtype=torch.float32, device=qkv.device)
        probs = []
        for wi, (qo, Sq, ko, Skv) in enumerate(windows):
            pr = torch.empty((B, H, Sq, Skv), dtype=torch.float32, device=qkv.device)
            mk = None if masks is None else masks[wi]
            _chk(mk)
            lib().attention_fwd(_p(qkv), _p(out), _p(pr), _p(mk), B, T, H, d, qo, Sq, ko, Skv, scale, drop_p, _stream())
            probs.append(pr)
        ctx.save_for_backward(qkv, *probs)
        ctx.cfg = (H, d, windows, masks, drop_p, scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, *probs = ctx.saved_tensors
        H, d, windows, masks, drop_p, scale = ctx.cfg
        B, T, _ = qkv.shape
        gout = gout.contiguous()
        dqkv = torch.zeros_like(qkv)
        for wi, (qo, Sq, ko, Skv) in enumerate(windows):
            mk = None if masks is None else masks[wi]
            lib().attention_bwd(_p(gout), _p(qkv), _p(probs[wi]), _p(mk), _p(dqkv), B, T, H, d, qo, Sq, ko, Skv, scale,
                                drop_p, _stream())
        return dqkv, None, None, None, None


def attention(qkv, H, windows, masks=None, drop_p=0.0):
    return AttentionFn.apply(qkv, H, tuple(windows), masks, drop_p)


class DropAddFn(torch.autograd.Function):
    """out = resid + x * emask * rowmask * scale  (nn.Dropout, drop_path and the residual add in one pass)."""

    @staticmethod
    def forward(ctx, x, resid, emask, rmask, rowlen, scale):
        _chk(x, resid, emask, rmask)
        out = torch.empty_like(x)
        lib().drop_add(_p(x), _p(resid), _p(emask), _p(rmask), _p(out), x.numel(), rowlen, scale, _stream())
        ctx.masks = (emask, rmask, rowlen, scale)
        ctx.has_res = resid is not None
        return out

    @staticmethod
    def backward(ctx, g):
        emask, rmask, rowlen, scale = ctx.masks
        g = g.contiguous()
        if emask is None and rmask is None and scale == 1.0:
            dx = g
        else:
            dx = torch.empty_like(g)
            lib().drop_add(_p(g), None, _p(emask), _p(rmask), _p(dx), g.numel(), rowlen, scale, _stream())
        return dx, (g if ctx.has_res else None), None, None, None, None


def drop_add(x, resid=None, emask=None, rmask=None, rowlen=1, scale=1.0):
    if resid is None and emask is None and rmask is None and scale == 1.0:
        return x
    return DropAddFn.apply(x, resid, emask, rmask, rowlen, float(scale))


class TabEmbedFn(torch.autograd.Function):
    """TabularTransformerEncoder.embedding before the LayerNorm (models/Transformer.py:240-256)."""

    @staticmethod
    def forward(ctx, x, cat_emb, con_w, con_b, cls, colemb, offs, rowcol, ncat):
        _chk(x, cat_emb, con_w, con_b, cls, colemb)
        B, ncols = x.shape
        D = cls.shape[-1]
        h = torch.empty((B, ncols + 1, D), dtype=torch.float32, device=x.device)
        lib().tab_embed_fwd(_p(x), _p(offs), _p(cat_emb), _p(con_w), _p(con_b), _p(cls), _p(colemb), _p(h), B, ncols, ncat, D,
                            _stream())
        ctx.save_for_backward(x, cat_emb, con_w, con_b, cls, colemb)
        ctx.cfg = (offs, rowcol, ncat)
        return h

    @staticmethod
    def backward(ctx, g):
        x, cat_emb, con_w, con_b, cls, colemb = ctx.saved_tensors
        offs, rowcol, ncat = ctx.cfg
        B, ncols = x.shape
        D = cls.shape[-1]
        g = g.contiguous()
        params = [cat_emb, con_w, con_b, cls, colemb]
        slots = [getattr(p_, "_gslot", None) if p_ is not None else None for p_ in params]
        use_slab = slots[3] is not None
        outs = [(s if use_slab else (torch.zeros_like(p_) if p_ is not None else None)) for p_, s in zip(params, slots)]
        nb = lib().tab_embed_bwd_workspace_bytes(ncols, D)
        ws = _ws.get(nb, x.device)
        nrows = 0 if cat_emb is None else cat_emb.shape[0]
        lib().tab_embed_bwd(_p(g), _p(x), _p(offs), _p(rowcol), nrows if ncat else 0, _p(outs[0]), _p(outs[1]), _p(outs[2]),
                            _p(outs[3]), _p(outs[4]), B, ncols, ncat, D, 1 if use_slab else 0, _p(ws), nb, _stream())
        if use_slab:
            for p_ in params:
                if p_ is not None:
                    _touch(p_)
            return (None,) * 9
        return (None, outs[0], outs[1], outs[2], outs[3], outs[4], None, None, None)


class TokMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        B, T, D = x.shape
        y = torch.empty((B, D), dtype=torch.float32, device=x.device)
        lib().tokmean_fwd(_p(x), _p(y), B, T, D, _stream())
        ctx.shape = (B, T, D)
        return y

    @staticmethod
    def backward(ctx, g):
        B, T, D = ctx.shape
        g = g.contiguous()
        dx = torch.empty((B, T, D), dtype=torch.float32, device=g.device)
        lib().tokmean_bwd(_p(g), _p(dx), B, T, D, _stream())
        return dx


def tokmean(x):
    return TokMeanFn.apply(x.contiguous())


# ------------------------------------------------------------------------------------------ losses
class L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=1) (STiLModel.py:185-191, utils/clip_loss.py:29-30)."""

    @staticmethod
    def forward(ctx, x):
        _chk(x)
        R, D = x.shape
        y = torch.empty_like(x)
        n = torch.empty((R,), dtype=torch.float32, device=x.device)
        lib().l2norm_fwd(_p(x), _p(y), _p(n), R, D, _stream())
        ctx.save_for_backward(y, n)
        return y

    @staticmethod
    def backward(ctx, g):
        y, n = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty_like(g)
        lib().l2norm_bwd(_p(g), _p(y), _p(n), _p(dx), y.shape[0], y.shape[1], _stream())
        return dx


def l2norm(x):
    return L2NormFn.apply(x.contiguous())


def _scale_by(unit, g, c=1.0):
    out = torch.empty_like(unit)
    lib().scale_dev(_p(unit), _p(g.contiguous()), float(c), _p(out), unit.numel(), _stream())
    return out


class CEHardFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean) -- STiLModel.py:284."""

    @staticmethod
    def forward(ctx, logits, labels):
        _chk(logits, labels)
        R, K = logits.shape
        rl = torch.empty((R,), dtype=torch.float32, device=logits.device)
        dz = torch.empty_like(logits)
        lib().ce_hard(_p(logits), K, _p(labels), _p(rl), _p(dz), K, R, K, 1.0 / R, _stream())
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        lib().reduce_sum(_p(rl), R, 1.0 / R, _p(loss), 0, _stream())
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return _scale_by(dz, g), None


class CESoftFn(torch.autograd.Function):
    """(F.cross_entropy(logits, q, reduction='none') * w).mean() -- STiLModel.py:301-303."""

    @staticmethod
    def forward(ctx, logits, targets, row_w):
        _chk(logits, targets, row_w)
        R, K = logits.shape
        rl = torch.empty((R,), dtype=torch.float32, device=logits.device)
        dz = torch.empty_like(logits)
        lib().ce_soft(_p(logits), K, _p(targets), K, _p(row_w), _p(rl), _p(dz), K, R, K, 1.0 / R, _stream())
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        lib().reduce_sum(_p(rl), R, 1.0 / R, _p(loss), 0, _stream())
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return _scale_by(dz, g), None, None


class ClipFromLogitsFn(torch.autograd.Function):
    """lambda_0 CE(Z, diag) + (1-lambda_0) CE(Z^T, diag)  (utils/clip_loss.py:35-38)."""

    @staticmethod
    def forward(ctx, Z, lam0):
        _chk(Z)
        B = Z.shape[0]
        lse = torch.empty((2, B), dtype=torch.float64, device=Z.device)   # row / column log-sum-exps in double (csrc/loss.hip: why)
        terms = torch.empty((B,), dtype=torch.float32, device=Z.device)
        loss = torch.empty((), dtype=torch.float32, device=Z.device)
        lib().clip_fwd(_p(Z), _p(lse), _p(terms), _p(loss), B, lam0, 1.0 - lam0, _stream())
        ctx.save_for_backward(Z, lse)
        ctx.lam0 = lam0
        return loss

    @staticmethod
    def backward(ctx, g):
        Z, lse = ctx.saved_tensors
        dZ = torch.empty_like(Z)
        lib().clip_bwd(_p(Z), _p(lse), _p(g.contiguous()), _p(dZ), Z.shape[0], ctx.lam0, 1.0 - ctx.lam0, _stream())
        return dZ, None


class ContrastGraphFn(torch.autograd.Function):
    """CoMatch's pseudo-label-graph contrastive loss (models/MatchModel/CoMatch.py:104-117) from S = log(sim) and the graph Q."""

    @staticmethod
    def forward(ctx, S, Q, threshold):
        _chk(S, Q)
        R, N = S.shape
        rl = torch.empty((R,), dtype=torch.float32, device=S.device)
        dS = torch.empty_like(S)
        lib().contrast_graph(_p(S), N, _p(Q), N, float(threshold), _p(rl), _p(dS), N, R, N, 1.0 / R, _stream())
        loss = torch.empty((), dtype=torch.float32, device=S.device)
        lib().reduce_sum(_p(rl), R, 1.0 / R, _p(loss), 0, _stream())
        ctx.save_for_backward(dS)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dS,) = ctx.saved_tensors
        return _scale_by(dS, g), None, None


def simmatch_unfold(tpo, probs, labels, c_smooth):
    """-> teacher_prob [R,N], pseudo-label [R,K]  (models/MatchModel/simmatch_model.py:289-302); no gradient."""
    _chk(tpo, probs, labels)
    R, N = tpo.shape
    K = probs.shape[1]
    teacher = torch.empty_like(tpo)
    pseudo = torch.empty_like(probs)
    lib().simmatch_unfold(_p(tpo), _p(probs), _p(labels), _p(teacher), _p(pseudo), R, N, K, float(c_smooth), _stream())
    return teacher, pseudo


def freematch_update(probs, p_model, label_hist, time_p, momentum=0.999):
    """Self-adaptive threshold state (updated in place) -> mask [R], one-hot pseudo-labels [R,K], argmax [R]
    (FreeMatchFolder/freematch_model.py:132-168); no gradient."""
    _chk(probs, p_model, label_hist, time_p)
    R, K = probs.shape
    dev = probs.device
    mask = torch.empty((R,), dtype=torch.float32, device=dev)
    onehot = torch.empty((R, K), dtype=torch.float32, device=dev)
    idx = torch.empty((R,), dtype=torch.int32, device=dev)
    scratch = torch.empty((R,), dtype=torch.float32, device=dev)
    lib().freematch_update(_p(probs), R, K, _p(p_model), _p(label_hist), _p(time_p), float(momentum), _p(mask), _p(onehot), _p(idx),
                           _p(scratch), _stream())
    return mask, onehot, idx


class FreeMatchEntropyFn(torch.autograd.Function):
    """FreeMatch's fairness loss over the masked rows (FreeMatchFolder/freematch_utils.py:18-47); 0 when the mask is empty."""

    @staticmethod
    def forward(ctx, logits, mask, p_model, label_hist):
        _chk(logits, mask, p_model, label_hist)
        R, K = logits.shape
        dev = logits.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dz = torch.empty_like(logits)
        P = torch.empty_like(logits)
        pred = torch.empty((R,), dtype=torch.int32, device=dev)
        vec = torch.empty((4 * K,), dtype=torch.float32, device=dev)
        lib().freematch_entropy(_p(logits), _p(mask), R, K, _p(p_model), _p(label_hist), _p(loss), _p(dz), _p(P), _p(pred), _p(vec), _stream())
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return _scale_by(dz, g), None, None, None


def clip_loss(f0, f1, T, lam0, gather=False):
    """CLIPLoss.forward (utils/clip_loss.py:27-39). Returns (loss, logits).
    gather (data-parallel `global_contrast`): both embeddings are all-gathered with autograd
    (comm.AllGatherFn) and every rank evaluates the loss over the GLOBAL batch, like MMatch.py:411-418 does for its
    memory bank; off by default because the reference's STiL keeps its ITC negatives rank-local."""
    n0, n1 = l2norm(f0), l2norm(f1)
    if gather:
        from .comm import all_gather_rows
        n0, n1 = all_gather_rows(n0), all_gather_rows(n1)
    Z = MatmulNTFn.apply(n0, n1, 1.0 / T)
    return ClipFromLogitsFn.apply(Z, float(lam0)), Z


class ClubFn(torch.autograd.Function):
    """CLUBMean.forward + learning_loss given mu = p_mu(x)  (club.py:107-130), closed form over the batch."""

    @staticmethod
    def forward(ctx, mu, y, global_stats=False):
        """global_stats (data-parallel `global_contrast`): the batch means of y and mu are averaged over the ranks, which
        makes the rank-mean of this loss (and of its gradients, after the data-parallel 1/world) equal to the closed
        form over the global batch: two [D] all-reduces instead of moving samples."""
        _chk(mu, y)
        R, D = y.shape
        dev = y.device
        ybar = torch.empty((D,), dtype=torch.float32, device=dev)
        mubar = torch.empty((D,), dtype=torch.float32, device=dev)
        colsum(y, ybar, R, D, scale=1.0 / R)
        colsum(mu, mubar, R, D, scale=1.0 / R)
        if global_stats:
            from .comm import allreduce_mean_
            allreduce_mean_(ybar)
            allreduce_mean_(mubar)
        tmp = torch.empty((2, R), dtype=torch.float32, device=dev)
        out2 = torch.empty((2,), dtype=torch.float32, device=dev)
        lib().club_fwd(_p(mu), _p(y), _p(ybar), _p(tmp), _p(out2), R, D, _stream())
        ctx.save_for_backward(mu, y, ybar, mubar)
        return out2[0], out2[1]

    @staticmethod
    def backward(ctx, gc, ge):
        mu, y, ybar, mubar = ctx.saved_tensors
        R, D = y.shape
        dmu = torch.empty_like(mu)
        dy = torch.empty_like(y)
        lib().club_bwd(_p(mu), _p(y), _p(ybar), _p(mubar), _p(gc.contiguous()), _p(ge.contiguous()), _p(dmu), _p(dy), R, D,
                       _stream())
        return dmu, dy, None


class ProtoLossFn(torch.autograd.Function):
    """PrototypeLoss.forward (utils/prototype_loss.py:24-40); hard label / confidence precomputed per row."""

    @staticmethod
    def forward(ctx, feat, prototypes, hard, conf, T):
        _chk(feat, prototypes, hard, conf)
        R, Dp = feat.shape
        K = prototypes.shape[0]
        rl = torch.empty((R,), dtype=torch.float32, device=feat.device)
        df = torch.empty_like(feat)
        lib().proto_loss(_p(feat), _p(prototypes), _p(hard), _p(conf), _p(rl), _p(df), R, K, Dp, T, _stream())
        loss = torch.empty((), dtype=torch.float32, device=feat.device)
        lib().reduce_sum(_p(rl), R, 1.0 / R, _p(loss), 0, _stream())
        ctx.save_for_backward(df)
        return loss

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        return _scale_by(df, g), None, None, None, None


@torch.no_grad()
def cgpl_pgls(zm, zi, zt, feat_u, prototypes, mask_random, rate_pseudo, T, th, use_pseudo, want_orig=False, pred_in=None):
    """CGPL case partition + PGLS smoothing (STiLModel.py:262-299,317-320) in one launch."""
    _chk(zm, zi, zt, feat_u, prototypes, mask_random)
    Bu, K = zm.shape
    Dp = feat_u.shape[1]
    dev = zm.device
    pl = torch.empty((Bu, K), dtype=torch.float32, device=dev)
    po = torch.empty((Bu, K), dtype=torch.float32, device=dev) if want_orig else None
    pred = torch.empty((Bu, K), dtype=torch.float32, device=dev)
    flags = torch.empty((Bu, 4), dtype=torch.uint8, device=dev)
    hard = torch.empty((Bu,), dtype=torch.int32, device=dev)
    w3 = torch.empty((3, Bu), dtype=torch.float32, device=dev)
    lib().cgpl_pgls(_p(zm), _p(zi), _p(zt), K, _p(feat_u), _p(prototypes), _p(mask_random), _p(pl), _p(po), _p(pred),
                    _p(flags), _p(hard), _p(w3), _p(pred_in), Bu, K, Dp, rate_pseudo, T, th, 1 if use_pseudo else 0, _stream())
    return pl, po, pred, flags, hard, w3


_ptr_tables = {}


def _ptr_table(tensors, device):
    """Device array of data pointers, cached: slab-backed tensors never move, and an H2D copy per step could not be
    captured into a hipGraph."""
    key = tuple(t.data_ptr() for t in tensors)
    tb = _ptr_tables.get(key)
    if tb is None:
        tb = torch.tensor(list(key), dtype=torch.int64, device=device)
        _ptr_tables[key] = tb
    return tb


# ------------------------------------------------------------------------------------------ SAINT tabular encoder
class SaintEmbedColMlpFn(torch.autograd.Function):
    """Token buffer [B, nfeats, d] of DisCoAttentionBackbone.forward_tabular (STiLModel_SAINT_backbone.py:159-178):
    categorical tokens = embeds[code + offset] + pos_encodings, continuous tokens = per-column simple_MLP(1->100->d)."""

    @staticmethod
    def forward(ctx, x, embeds, pos, meta, *mlp_params):
        _chk(x, embeds, pos)
        B, ncols = x.shape
        d = embeds.shape[1]
        ncat, ncon, hid = meta["ncat"], meta["ncon"], meta["hid"]
        nfeats = ncat + ncon + 1
        out = torch.empty((B, nfeats, d), dtype=torch.float32, device=x.device)
        lib().saint_embed_fwd(_p(x), _p(meta["cat_cols"]), _p(meta["offs"]), _p(embeds), _p(pos), _p(out), B, ncols, ncat,
                              nfeats, d, _stream())
        if ncon:
            ptrs = _ptr_table(mlp_params, x.device)
            lib().colmlp_fwd(_p(x), _p(meta["con_cols"]), _p(ptrs), _p(out), B, ncols, ncon, nfeats, ncat + 1, hid, d, _stream())
            ctx.ptrs = ptrs
        ctx.save_for_backward(x, embeds, pos, *mlp_params)
        ctx.meta = meta
        return out

    @staticmethod
    def backward(ctx, g):
        x, embeds, pos, *mlp_params = ctx.saved_tensors
        meta = ctx.meta
        B, ncols = x.shape
        d = embeds.shape[1]
        ncat, ncon, hid = meta["ncat"], meta["ncon"], meta["hid"]
        nfeats = ncat + ncon + 1
        g = g.contiguous()
        params = [embeds, pos] + list(mlp_params)
        use_slab = getattr(embeds, "_gslot", None) is not None
        outs = [p_._gslot if use_slab else torch.zeros_like(p_) for p_ in params]
        acc = 1 if use_slab else 0
        lib().saint_embed_bwd(_p(g), _p(x), _p(meta["cat_cols"]), _p(meta["offs"]), _p(meta["rowcol"]), embeds.shape[0],
                              _p(outs[0]), _p(outs[1]), B, ncols, ncat, nfeats, d, acc, _stream())
        if ncon:
            gptrs = _ptr_table(outs[2:], x.device) if use_slab else torch.tensor([t.data_ptr() for t in outs[2:]], dtype=torch.int64, device=x.device)
            lib().colmlp_bwd(_p(g), _p(x), _p(meta["con_cols"]), _p(ctx.ptrs), _p(gptrs), B, ncols, ncon, nfeats, ncat + 1, hid,
                             d, acc, _stream())
        if use_slab:
            for p_ in params:
                _touch(p_)
            return (None,) * (4 + len(mlp_params))
        return (None, outs[0], outs[1], None, *outs[2:])


class GegluFn(torch.autograd.Function):
    """GEGLU (SAINT/model_util.py:43-46): x, gates = h.chunk(2, -1); x * gelu(gates)."""

    @staticmethod
    def forward(ctx, h):
        _chk(h)
        H = h.shape[-1] // 2
        rows = h.numel() // (2 * H)
        out = torch.empty((*h.shape[:-1], H), dtype=torch.float32, device=h.device)
        lib().geglu_fwd(_p(h), _p(out), rows, H, _stream())
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        H = h.shape[-1] // 2
        g = g.contiguous()
        dh = torch.empty_like(h)
        lib().geglu_bwd(_p(g), _p(h), _p(dh), h.numel() // (2 * H), H, _stream())
        return dh


class RowSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        _chk(z)
        R, C = z.shape
        p = torch.empty_like(z)
        lib().row_softmax_fwd(_p(z), _p(p), R, C, _stream())
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        g = g.contiguous()
        dz = torch.empty_like(p)
        lib().row_softmax_bwd(_p(g), _p(p), _p(dz), p.shape[0], p.shape[1], _stream())
        return dz


def softmax_rows(z):
    """softmax over the last dim of a [rows, C] matrix (no autograd): metric probabilities, test_step scores."""
    _chk(z)
    R, C = z.shape
    p = torch.empty_like(z)
    lib().row_softmax_fwd(_p(z), _p(p), R, C, _stream())
    return p


class MatmulNNFn(torch.autograd.Function):
    """C = A @ B (A [M,K], B [K,N]) -- P @ V of the inter-sample attention."""

    @staticmethod
    def forward(ctx, A, B):
        _chk(A, B)
        ctx.save_for_backward(A, B)
        return gemm_nt(A, transpose(B), A.shape[0], B.shape[1], A.shape[1])

    @staticmethod
    def backward(ctx, gC):
        A, B = ctx.saved_tensors
        gC = gC.contiguous()
        M, K = A.shape
        N = B.shape[1]
        dA = gemm_nt(gC, B, M, K, N)                         # gC [M,N] . B^T  (B is [K,N] = "W[K,N]")
        dB = gemm_nt(transpose(A), transpose(gC), K, N, M)   # A^T [K,M] . gC [M,N]
        return dA, dB
