"""TEST INFRASTRUCTURE ONLY (see oracle/stil_oracle.py header): numpy restatement of the torchmetrics==0.11.0 metrics the
reference module keeps (environment.yaml pins torchmetrics 0.11.0; the package is absent from this image, so its
published definitions are restated and pinned against scikit-learn in tests/test_oracle_golden.py):

  Accuracy(task='multiclass', top_k=k)  micro: mean(target in top-k(preds))      STiLModel.py:122-126, 129-137
  Accuracy(task='binary')               mean((preds > 0.5) == target)            STiLModel.py:127-137 (num_classes == 2)
  AUROC(task='binary')                  exact area under the ROC curve           STiLModel.py:139-145
  AUROC(task='multiclass')              macro mean of one-vs-rest areas; a class without positives (or without negatives)
                                        contributes 0 (0.11.0 returns an all-zero rate with a warning)  -- this last rule is
                                        NOT pinned by scikit-learn (it raises instead): parity unpinned for absent classes.
"""
import numpy as np


def topk_accuracy(preds: np.ndarray, target: np.ndarray, k: int = 1) -> float:
    """ties resolved towards the lower class index (argmax's first-maximum rule for k = 1)."""
    n, K = preds.shape
    st = preds[np.arange(n), target][:, None]
    idx = np.arange(K)[None, :]
    beat = ((preds > st) | ((preds == st) & (idx < target[:, None]))).sum(1)
    return float((beat < k).mean())


def binary_accuracy(probs: np.ndarray, target: np.ndarray, threshold: float = 0.5) -> float:
    return float(((probs > threshold).astype(np.int64) == (target == 1).astype(np.int64)).mean())


def binary_auroc(scores: np.ndarray, positive: np.ndarray) -> float:
    """Mann-Whitney form of the trapezoid ROC area: P(s+ > s-) + 0.5 P(s+ == s-); 0 when a side is empty."""
    scores = np.asarray(scores, dtype=np.float32)
    positive = np.asarray(positive, dtype=bool)
    npos, nneg = int(positive.sum()), int((~positive).sum())
    if npos == 0 or nneg == 0:
        return 0.0
    neg = np.sort(scores[~positive])
    pos = scores[positive]
    less = np.searchsorted(neg, pos, side="left").astype(np.int64)
    leq = np.searchsorted(neg, pos, side="right").astype(np.int64)
    u2 = int((2 * less + (leq - less)).sum())
    return u2 / (2.0 * npos * nneg)


def multiclass_auroc(probs: np.ndarray, target: np.ndarray):
    K = probs.shape[1]
    per = np.array([binary_auroc(probs[:, c], target == c) for c in range(K)], dtype=np.float64)
    return float(per.mean()), per
