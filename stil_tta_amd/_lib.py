"""ctypes binding of libstil_hip.so, generated from include/stil_hip.h at import time.

There is NO fallback: if the shared library is missing, or a call returns an error, a
RuntimeError is raised (the product path must never silently run on something else).
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "stil_hip.h")
LIB_PATH = os.environ.get("STIL_LIB_PATH") or os.path.join(_HERE, "lib", "libstil_hip.so")  # STIL_LIB_PATH: A/B builds of the same sources (tests/tools)
CSRC = os.path.join(_HERE, "csrc")

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
}


def parse_header(path: str = HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int|size_t)\s+(stil_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = {"const char*": ctypes.c_char_p, "int": ctypes.c_int, "size_t": ctypes.c_size_t}[ret]
        argl = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argl.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    toks = a.split()
                    ty = " ".join(toks[:-1])
                    argl.append((_CTYPES[ty], toks[-1]))
        protos[name] = (restype, argl)
    return protos


# host files that decide WHAT a launch carries (operands, epilogue forms, which layers fuse what): a profiler summary taken
# before a change to one of them no longer describes the launches the bench times
_HOST_HASHED = ("ops.py", "modules.py", "layouts.py", "stil_model.py", "flat.py", "saint.py")


def source_hash() -> str:
    """Content hash of the kernel sources (csrc/*.hip, *.h) AND of the host files that decide what a launch carries
    (_HOST_HASHED): stamps profiler summaries under profiles/ with the build they were taken on (the GPU box has no .git to
    ask), so that bench.py quotes them only while kernels and launch composition are unchanged."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    for f in _HOST_HASHED:
        h.update(f.encode())
        h.update(open(os.path.join(_HERE, f), "rb").read())
    return h.hexdigest()[:12]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    if (not force) and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    r = subprocess.run(["make", "-C", CSRC] + (["-B"] if force else []), capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError("building libstil_hip.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               f"(or `make -C {CSRC}`) first; there is no fallback path")
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (restype, argl) in self.protos.items():
            fn = getattr(self._dll, name)  # AttributeError if the header declares a symbol the .so lacks
            fn.restype = restype
            fn.argtypes = [t for t, _ in argl]
        self._dll.stil_last_error.restype = ctypes.c_char_p
        self._prof = None
        self._prof_single = True
        self._prof_only = None

    # ---- optional per-entry-point GPU timing (HIP events on the launch stream); used by bench.py only
    def begin_profile(self, single_stream: bool = True, only=None):
        """single_stream: keep every launch on the caller's stream while profiling (clean per-kernel durations); False lets
        the side stream run as usual, events then bracket each launch on whichever stream it goes to (durations include
        the time a kernel shares the chip with the other stream's kernel)."""
        self._prof = []
        self._prof_single = bool(single_stream)
        self._prof_only = None if only is None else frozenset(only)  # entry points to bracket (None = all)

    def end_profile(self):
        import torch
        torch.cuda.synchronize()
        out = [(name, s.elapsed_time(e), meta) for name, s, e, meta in self._prof]
        self._prof = None
        return out

    def last_error(self) -> str:
        return (self._dll.stil_last_error() or b"").decode()

    def __getattr__(self, name):
        fn = getattr(self._dll, "stil_" + name)
        restype = self.protos["stil_" + name][0]
        if restype is not ctypes.c_int or name in ("version", "device_count", "gemm_nt_variant", "gemm_nt_tile_rows", "gemm_nt_config", "weight_layout_job_bytes", "weight_layout_job_blocks", "gemm_nt_bstats_ok", "gemm_nt_force_splits", "reduce_job_bytes", "wgrad_splits", "colsum_chunks", "wgrad_force_splits"):
            return fn

        def call(*args, meta=None):
            prof = self.__dict__.get("_prof")
            if prof is not None and (self._prof_only is None or name in self._prof_only):
                import torch
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                rc = fn(*args)
                e.record()
                prof.append((name, s, e, meta))
            else:
                rc = fn(*args)
            if rc != 0:
                raise RuntimeError(f"stil_{name} failed ({rc}): {self.last_error()}")
            return rc

        self.__dict__[name] = call
        return call


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
