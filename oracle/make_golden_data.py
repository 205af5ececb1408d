"""Golden vectors for the device data pipeline's tabular corruption, recorded from the REFERENCE's
own `ContrastiveImagingAndTabularDataset.corrupt` (datasets/ContrastiveImagingAndTabularDataset.py:146-158).

Runs only in the build container (needs /root/reference).  The dataset module imports torchvision / albumentations at
its top (absent offline, no arithmetic on this path): they are stubbed in sys.modules; `corrupt` itself is plain
`random` + numpy.  The reference draws with the global `random` / `np.random` state: every sample is generated under
`random.seed(s); np.random.seed(s)` and the same two draws (`random.sample`, `np.random.choice`) are replayed here to
record WHICH columns were replaced and by WHICH rows of the marginal table -- the device kernel is fed those draws and
must reproduce the reference's output bit for bit.  Writes tests/golden/tab_corrupt.npz (tensors only).

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_data.py
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("STIL_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.transforms = mod("torchvision.transforms.transforms", Compose=object, Resize=object, Lambda=object)
    tv.transforms.Compose = object
    tv.io = mod("torchvision.io", read_image=None)
    mod("albumentations")


def corrupt_oracle(subject, marginal, idx, pos):
    """The restated algorithm with the draws made explicit (numpy; also used by tests/test_oracle_golden.py)."""
    out = np.array(subject, dtype=np.float64).copy()
    out[idx] = marginal[idx, pos]
    return out


def main():
    sys.path.insert(0, REF)
    install_stubs()
    import importlib.util
    # by file path: an installed package named `datasets` (HuggingFace) would shadow the reference's namespace directory
    spec = importlib.util.spec_from_file_location("ref_contrastive_dataset", os.path.join(REF, "datasets", "ContrastiveImagingAndTabularDataset.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    cls = D.ContrastiveImagingAndTabularDataset
    rng = np.random.RandomState(2022)
    n_rows, n_cols = 40, 17
    table = np.concatenate([rng.randint(0, 7, size=(n_rows, 4)).astype(np.float64), rng.randn(n_rows, 13)], axis=1)
    fx = {"table": table}
    for c in (0.3, 0.0, 1.0, 0.05):
        self = types.SimpleNamespace(marginal_distributions=np.transpose(table), c=c)
        k = int(n_cols * c)
        outs, idxs, poss = [], [], []
        for s in range(12):
            subject = list(table[s % n_rows])
            random.seed(s); np.random.seed(s)
            out = cls.corrupt(self, subject)                      # the reference
            random.seed(s); np.random.seed(s)                     # replay its two draws
            idx = random.sample(list(range(n_cols)), k)
            pos = np.random.choice(n_rows, size=k)
            assert np.array_equal(out, corrupt_oracle(subject, np.transpose(table), idx, pos))
            outs.append(out); idxs.append(idx); poss.append(pos)
        tag = f"c{int(c * 100):03d}"
        fx[tag + "_out"] = np.asarray(outs)
        fx[tag + "_idx"] = np.asarray(idxs, dtype=np.int32).reshape(12, k)
        fx[tag + "_pos"] = np.asarray(poss, dtype=np.int32).reshape(12, k)
    path = os.path.join(ROOT, "tests", "golden", "tab_corrupt.npz")
    np.savez_compressed(path, **fx)
    print("reference corrupt == restatement on", 4 * 12, "samples ->", os.path.relpath(path, ROOT), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
