"""Device input pipeline: the per-sample CPU work of the reference's DataLoader workers
(datasets/ContrastiveImagingAndTabularDataset.py:146-213, utils/utils.py:46-70) as a few batch launches on data that is
already resident in HBM -- the table, its transposed marginals and the uint8 images of one GPU's shard fit easily in
288 GB, so nothing but the batch's row indices crosses PCIe per step.

    builder = ContrastiveBatchBuilder(images_u8 [N,H,W,3], table [N,n_cols], labels [N], img_size=128, target="dvm",
                                      corruption_rate=0.3, augmentation_rate=0.95, labelled=True)
    batch_part = builder(index_tensor)          # -> (im_views, tab_views, y, orig_im, identify), SURVEY.md 8b

Random draws are made here (numpy Generator on the host for the few scalars per image, the hash RNG of
stil_tab_corrupt_draw for the table) or injected (`draws=`) by the parity tests.  What is pinned to the reference:
`corrupt` (golden vectors recorded from the reference's own method).  The image transforms follow torchvision's float
tensor formulas (the reference's non-`augmentation_speedup` branch); albumentations / cv2 / torchvision are absent
offline, so they are tested against PyTorch restatements only (unpinned).  `kind` selects the reference's transform
family: "contrastive" (grab_image_augmentations), "hard_eval" / "soft_eval" (utils/utils.py:94-186, the labelled set of the
Match baselines), "weak" / "strong" (:187-256, their unlabelled views); rotation follows A.Rotate (bilinear, reflect-101).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from ._lib import lib
from .ops import _p, _stream, _chk


def rrc_boxes(H: int, W: int, B: int, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """torchvision RandomResizedCrop.get_params for B images at once -> int32 [B, 4] (top, left, height, width):
    up to 10 attempts of area ~ U(scale) * H * W, log-uniform aspect ratio; fallback = central crop at the clamped ratio."""
    rng = rng or np.random.default_rng()
    area = float(H * W)
    out = np.zeros((B, 4), dtype=np.int32)
    done = np.zeros(B, dtype=bool)
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        ta = area * rng.uniform(scale[0], scale[1], size=B)
        ar = np.exp(rng.uniform(lo, hi, size=B))
        w = np.rint(np.sqrt(ta * ar)).astype(np.int64)
        h = np.rint(np.sqrt(ta / ar)).astype(np.int64)
        ok = (~done) & (w > 0) & (w <= W) & (h > 0) & (h <= H)
        top = (rng.random(B) * (H - np.minimum(h, H) + 1)).astype(np.int64)
        left = (rng.random(B) * (W - np.minimum(w, W) + 1)).astype(np.int64)
        out[ok] = np.stack([top, left, h, w], 1)[ok]
        done |= ok
    if not done.all():  # fallback: whole image clamped to the ratio range, centred
        in_ratio = W / H
        if in_ratio < ratio[0]:
            w, h = W, int(round(W / ratio[0]))
        elif in_ratio > ratio[1]:
            h, w = H, int(round(H * ratio[1]))
        else:
            w, h = W, H
        out[~done] = np.array([(H - h) // 2, (W - w) // 2, h, w], dtype=np.int32)
    return out


class TabularCorruptor:
    """`corrupt` (ContrastiveImagingAndTabularDataset.py:146-158) for a batch: k = int(n_cols * corruption_rate) distinct
    columns of every row are replaced by the value of a uniformly drawn training row in that column."""

    def __init__(self, table: torch.Tensor, corruption_rate: float, device, seed: int = 2022):
        table = torch.as_tensor(table, dtype=torch.float32)
        self.n_rows, self.n_cols = table.shape
        self.marginal = table.t().contiguous().to(device)   # generate_marginal_distributions: the table transposed
        self.k = int(self.n_cols * corruption_rate)
        self.seed, self.drawn = int(seed), 0   # drawn: running total of counter values consumed (batches may differ in size)

    def draw(self, B: int):
        dev = self.marginal.device
        idx = torch.empty((B, max(self.k, 1)), dtype=torch.int32, device=dev)
        pos = torch.empty((B, max(self.k, 1)), dtype=torch.int32, device=dev)
        if self.k:
            lib().tab_corrupt_draw(_p(idx), _p(pos), B, self.n_cols, self.n_rows, self.k, self.seed, self.drawn, None, _stream())
        self.drawn += B * 2 * self.k
        return idx, pos

    def __call__(self, clean: torch.Tensor, draws=None) -> torch.Tensor:
        _chk(clean)
        B = clean.shape[0]
        idx, pos = self.draw(B) if draws is None else (draws[0].to(clean.device, torch.int32).contiguous(), draws[1].to(clean.device, torch.int32).contiguous())
        out = torch.empty_like(clean)
        lib().tab_corrupt(_p(clean), _p(self.marginal), _p(idx), _p(pos), _p(out), B, self.n_cols, self.n_rows, self.k, _stream())
        return out


def resize_crop(src: torch.Tensor, boxes, P: int, flip=None, jitter=None) -> torch.Tensor:
    """src: uint8 [B,H,W,3] (scaled by 1/255 like convert_to_ts) or float [B,3,H,W]; boxes int32 [B,4] (top, left, h, w);
    flip uint8 [B]; jitter float [B,4] = (brightness, contrast, saturation, gray).  -> float [B,3,P,P] in [0,1]."""
    _chk(src)
    dev = src.device
    u8 = src.dtype == torch.uint8
    B = src.shape[0]
    H, W = (src.shape[1], src.shape[2]) if u8 else (src.shape[2], src.shape[3])
    scale = 1.0 / 255.0 if u8 else 1.0
    if not (isinstance(boxes, torch.Tensor) and boxes.is_cuda):
        # host-side boxes (the augmenter's numpy draws): checked here, before the upload, whatever the batch size -- no
        # device round trip.  Boxes that already live on the device are clamped to the image inside the kernel instead.
        b = np.asarray(boxes.cpu() if isinstance(boxes, torch.Tensor) else boxes).reshape(-1, 4)
        if not bool(((b[:, 0] >= 0) & (b[:, 1] >= 0) & (b[:, 2] > 0) & (b[:, 3] > 0) & (b[:, 0] + b[:, 2] <= H) & (b[:, 1] + b[:, 3] <= W)).all()):
            raise ValueError("crop box outside the source image")
    boxes = torch.as_tensor(boxes, dtype=torch.int32).to(dev).contiguous()
    if boxes.shape != (B, 4):
        raise ValueError(f"boxes must be [B, 4] = {(B, 4)}, got {tuple(boxes.shape)}")
    flip = None if flip is None else torch.as_tensor(flip).to(dev, torch.uint8).contiguous()
    gmean = None
    if jitter is not None:
        jitter = torch.as_tensor(jitter, dtype=torch.float32).to(dev).contiguous()
        gmean = torch.empty((B,), dtype=torch.float32, device=dev)
        lib().aug_gray_mean(_p(src) if u8 else None, None if u8 else _p(src), _p(jitter), _p(gmean), B, H, W, scale, _stream())
    out = torch.empty((B, 3, P, P), dtype=torch.float32, device=dev)
    lib().aug_resize(_p(src) if u8 else None, None if u8 else _p(src), _p(boxes), _p(flip), _p(jitter), _p(gmean), _p(out), B, H, W, P, scale,
                     _stream())
    return out


def gaussian_blur(src: torch.Tensor, sigma, ksize: int = 29) -> torch.Tensor:
    """transforms.GaussianBlur(ksize, sigma) per sample (sigma[b] <= 0: unchanged) -> float [B,3,H,W] (uint8 sources / 255)."""
    _chk(src)
    u8 = src.dtype == torch.uint8
    B = src.shape[0]
    H, W = (src.shape[1], src.shape[2]) if u8 else (src.shape[2], src.shape[3])
    sigma = torch.as_tensor(sigma, dtype=torch.float32).to(src.device).contiguous()
    tmp = torch.empty((B, 3, H, W), dtype=torch.float32, device=src.device)
    out = torch.empty_like(tmp)
    lib().aug_blur(_p(src) if u8 else None, None if u8 else _p(src), _p(sigma), _p(tmp), _p(out), B, H, W, int(ksize), 1.0 / 255.0 if u8 else 1.0,
                   _stream())
    return out


def rotate(src: torch.Tensor, angle) -> torch.Tensor:
    """A.Rotate / RandomRotation per sample: angle [B] degrees, counter-clockwise, about the centre, same size -> float [B,3,H,W]."""
    _chk(src)
    u8 = src.dtype == torch.uint8
    B = src.shape[0]
    H, W = (src.shape[1], src.shape[2]) if u8 else (src.shape[2], src.shape[3])
    angle = torch.as_tensor(angle, dtype=torch.float32).to(src.device).contiguous()
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=src.device)
    lib().aug_rotate(_p(src) if u8 else None, None if u8 else _p(src), _p(angle), _p(out), B, H, W, 1.0 / 255.0 if u8 else 1.0, _stream())
    return out


def adjust_hue_(img: torch.Tensor, hue=None, gray=None) -> torch.Tensor:
    """In place on float [B,3,H,W] in [0,1]: torchvision adjust_hue(hue[b]) then the grey image where gray[b] != 0."""
    _chk(img)
    B, _, H, W = img.shape
    hue = None if hue is None else torch.as_tensor(hue, dtype=torch.float32).to(img.device).contiguous()
    gray = None if gray is None else torch.as_tensor(gray, dtype=torch.float32).to(img.device).contiguous()
    lib().aug_hue(_p(img), _p(hue), _p(gray), B, H, W, _stream())
    return img


# transform families of utils/utils.py (torchvision branch): RandomResizedCrop scale, rotation limit, ColorJitter (amount,
# probability, hue), grayscale probability, GaussianBlur (kernel, probability, "pre" = on the source image before the crop /
# "post" = on the cropped view).  dvm only: gray, blur.
def _policy(kind: str, dvm: bool, crop_scale_lower: float = 0.08):
    if kind == "contrastive":   # grab_image_augmentations, :46-91
        return dict(scale=(crop_scale_lower, 1.0), jitter=(0.8, 0.8, 0.0), gray=0.2, blur=(29, 0.5, "pre")) if dvm else \
            dict(scale=(0.2, 1.0), rotate=45.0, jitter=(0.5, 1.0, 0.0))
    if kind == "hard_eval":     # grab_hard_eval_image_augmentations, :139-186
        return dict(scale=(0.6, 1.0), jitter=(0.8, 0.8, 0.0), gray=0.2, blur=(29, 0.5, "pre")) if dvm else \
            dict(scale=(0.6, 1.0), rotate=45.0, jitter=(0.5, 1.0, 0.0))
    if kind == "soft_eval":     # grab_soft_eval_image_augmentations, :94-136
        return dict(scale=(0.8, 1.0), rotate=20.0, jitter=(0.25, 1.0, 0.0))
    if kind == "weak":          # grab_weak_image_augmentations, :187-216
        return dict(scale=(0.2, 1.0))
    if kind == "strong":        # grab_strong_image_augmentations, :219-256
        d = dict(scale=(0.2, 1.0), jitter=(0.4, 0.8, 0.1), blur=(19, 0.5, "post"))
        if dvm:
            d["gray"] = 0.2
        return d
    raise ValueError(f"unknown transform family {kind}")


class ImageAugmenter:
    """One of the reference's transform families (utils/utils.py:46-256, see _policy) + default_transform
    (ContrastiveImagingAndTabularDataset.py:66-90) as batch launches.  contrastive/dvm: ColorJitter(0.8, 0.8, 0.8) p=0.8,
    ToGray p=0.2, GaussianBlur(29, sigma U(0.1, 2)) p=0.5 on the source image, RandomResizedCrop(scale=(0.08, 1), ratio=(3/4,
    4/3)), HFlip p=0.5; contrastive/cardiac: HFlip, Rotate(45), ColorJitter(0.5, 0.5, 0.5), RandomResizedCrop(scale=(0.2, 1)).
    Every image is augmented with probability `augmentation_rate`, otherwise only resized (generate_imaging_views)."""

    def __init__(self, img_size: int, target: str = "dvm", augmentation_rate: float = 1.0, seed: int = 2022, kind: str = "contrastive",
                 crop_scale_lower: float = 0.08):
        self.P, self.dvm, self.rate = int(img_size), target.lower() == "dvm", float(augmentation_rate)
        self.policy = _policy(kind, self.dvm, crop_scale_lower)
        self.rng = np.random.default_rng(seed)

    def draw(self, B: int, H: int, W: int) -> Dict[str, np.ndarray]:
        r, pol = self.rng, self.policy
        aug = r.random(B) < self.rate
        boxes = rrc_boxes(H, W, B, scale=pol["scale"], rng=r)
        boxes[~aug] = np.array([0, 0, H, W], dtype=np.int32)
        d = dict(boxes=boxes, flip=((r.random(B) < 0.5) & aug).astype(np.uint8), aug=aug)
        if "jitter" in pol:
            amt, prob, hue = pol["jitter"]
            jit = np.ones((B, 4), dtype=np.float32)
            on = aug & (r.random(B) < prob)
            jit[:, :3] = np.where(on[:, None], r.uniform(max(0.0, 1.0 - amt), 1.0 + amt, size=(B, 3)), 1.0)
            jit[:, 3] = 0.0
            gray = ((r.random(B) < pol["gray"]) & aug).astype(np.float32) if "gray" in pol else np.zeros(B, np.float32)
            if hue > 0.0:   # hue comes before the grey conversion: both go to the hue kernel
                d["hue"] = np.where(on, r.uniform(-hue, hue, size=B), 0.0).astype(np.float32)
                d["gray_after_hue"] = gray
            else:
                jit[:, 3] = gray
            d["jitter"] = jit
        if "blur" in pol:
            k, prob, when = pol["blur"]
            d["sigma"] = np.where(aug & (r.random(B) < prob), r.uniform(0.1, 2.0, size=B), 0.0).astype(np.float32)
        if "rotate" in pol:
            d["angle"] = np.where(aug, r.uniform(-pol["rotate"], pol["rotate"], size=B), 0.0).astype(np.float32)
        return d

    def __call__(self, src: torch.Tensor, draws: Optional[Dict[str, np.ndarray]] = None, want_orig: bool = True):
        """-> (augmented view, unaugmented resized image or None), float [B,3,P,P]."""
        u8 = src.dtype == torch.uint8
        B = src.shape[0]
        H, W = (src.shape[1], src.shape[2]) if u8 else (src.shape[2], src.shape[3])
        d = draws or self.draw(B, H, W)
        pol = self.policy
        sg = d.get("sigma")
        blur_k, _, blur_when = pol.get("blur", (29, 0.0, "pre"))
        on = sg is not None and float(np.max(sg)) > 0.0
        x = src
        if d.get("angle") is not None and float(np.abs(d["angle"]).max()) > 0.0:
            x = rotate(x, d["angle"])
        if on and blur_when == "pre" and min(H, W) > blur_k // 2:
            x = gaussian_blur(x, sg, blur_k)
        view = resize_crop(x, d["boxes"], self.P, d.get("flip"), d.get("jitter"))
        if d.get("hue") is not None:
            adjust_hue_(view, d["hue"], d.get("gray_after_hue"))
        if on and blur_when == "post" and self.P > blur_k // 2:
            view = gaussian_blur(view, sg, blur_k)
        if not want_orig:
            return view, None
        full = np.tile(np.array([[0, 0, H, W]], dtype=np.int32), (B, 1))
        return view, resize_crop(src, full, self.P)


class ContrastiveBatchBuilder:
    """ContrastiveImagingAndTabularDataset.__getitem__ + default_collate for a batch of row indices, on the device:
    -> (im_views = [placeholder [b], augmented [b,3,P,P]], tab_views = [clean [b,n], corrupted [b,n]], y [b] int64,
        orig_im [b,3,P,P], identify [b] bool)  -- the part tuple STiLModel.training_step consumes."""

    def __init__(self, images: torch.Tensor, table: torch.Tensor, labels: torch.Tensor, img_size: int, target: str = "dvm",
                 corruption_rate: float = 0.3, augmentation_rate: float = 0.95, labelled: bool = True, device="cuda", seed: int = 2022):
        self.images = images.to(device)                      # uint8 [N,H,W,3] or float [N,3,H,W], resident in HBM
        self.table = torch.as_tensor(table, dtype=torch.float32).to(device)
        self.labels = torch.as_tensor(labels, dtype=torch.int64).to(device)
        assert len(self.images) == len(self.table) == len(self.labels)
        self.corrupt = TabularCorruptor(self.table, corruption_rate, device, seed)
        self.augment = ImageAugmenter(img_size, target, augmentation_rate, seed + 1)
        self.labelled = bool(labelled)

    def __len__(self):
        return len(self.labels)

    def __call__(self, index: torch.Tensor, draws=None):
        index = torch.as_tensor(index).to(self.table.device, torch.int64)
        b = len(index)
        clean = self.table.index_select(0, index)
        src = self.images.index_select(0, index)
        view, orig = self.augment(src, None if draws is None else draws.get("image"))
        corrupted = self.corrupt(clean, None if draws is None else draws.get("table"))
        dev = self.table.device
        return ([torch.zeros(b, device=dev), view], [clean, corrupted], self.labels.index_select(0, index), orig,
                torch.full((b,), self.labelled, dtype=torch.bool, device=dev))


class EvalTrainBatchBuilder:
    """ImagingAndTabularDataset(train=True, return_index=True).__getitem__ + default_collate (datasets/ImagingAndTabularDataset.py:
    158-190): the LABELLED part of the Match baselines' batches -> ((image [b,3,P,P], table [b,n]), y [b], index [b]).  With
    probability eval_train_augment_rate a sample gets the hard-eval transform and a corrupted table, otherwise it is only
    resized and keeps its clean table."""

    def __init__(self, images, table, labels, img_size: int, target: str = "dvm", corruption_rate: float = 0.3,
                 eval_train_augment_rate: float = 0.8, device="cuda", seed: int = 2022):
        self.images = images.to(device)
        self.table = torch.as_tensor(table, dtype=torch.float32).to(device)
        self.labels = torch.as_tensor(labels, dtype=torch.int64).to(device)
        self.corrupt = TabularCorruptor(self.table, corruption_rate, device, seed) if corruption_rate and corruption_rate > 0 else None
        self.augment = ImageAugmenter(img_size, target, eval_train_augment_rate, seed + 1, kind="hard_eval")

    def __len__(self):
        return len(self.labels)

    def __call__(self, index: torch.Tensor, draws=None):
        index = torch.as_tensor(index).to(self.table.device, torch.int64)
        clean = self.table.index_select(0, index)
        src = self.images.index_select(0, index)
        u8 = src.dtype == torch.uint8
        H, W = (src.shape[1], src.shape[2]) if u8 else (src.shape[2], src.shape[3])
        d = (draws or {}).get("image") or self.augment.draw(len(index), H, W)
        view, _ = self.augment(src, d, want_orig=False)
        tab = clean
        if self.corrupt is not None:
            aug = torch.as_tensor(d["aug"]).to(clean.device)
            tab = torch.where(aug[:, None], self.corrupt(clean, (draws or {}).get("table")), clean)
        return (view, tab), self.labels.index_select(0, index), index


class StrongWeakBatchBuilder:
    """StrongWeakImagingAndTabularDataset.__getitem__ + default_collate (datasets/StrongWeakImagingAndTabularDataset.py:166-196):
    the UNLABELLED part of the Match baselines' batches -> ([(weak image, weakly corrupted table), (strong image, strongly
    corrupted table)[, a second strong pair]], y [b]); weak corruption rate 0.1 (:75), strong = corruption_rate;
    two_strong for CoMatch (trainers/evaluate.py:53)."""

    def __init__(self, images, table, labels, img_size: int, target: str = "dvm", corruption_rate: float = 0.3, two_strong: bool = False,
                 device="cuda", seed: int = 2022):
        self.images = images.to(device)
        self.table = torch.as_tensor(table, dtype=torch.float32).to(device)
        self.labels = torch.as_tensor(labels, dtype=torch.int64).to(device)
        self.weak_corrupt = TabularCorruptor(self.table, 0.1, device, seed)
        self.strong_corrupt = TabularCorruptor(self.table, corruption_rate, device, seed + 7)
        self.weak = ImageAugmenter(img_size, target, 1.0, seed + 1, kind="weak")
        self.strong = ImageAugmenter(img_size, target, 1.0, seed + 2, kind="strong")
        self.two_strong = bool(two_strong)

    def __len__(self):
        return len(self.labels)

    def __call__(self, index: torch.Tensor):
        index = torch.as_tensor(index).to(self.table.device, torch.int64)
        clean = self.table.index_select(0, index)
        src = self.images.index_select(0, index)
        views = [(self.weak(src, want_orig=False)[0], self.weak_corrupt(clean))]
        for _ in range(2 if self.two_strong else 1):
            views.append((self.strong(src, want_orig=False)[0], self.strong_corrupt(clean)))
        return views, self.labels.index_select(0, index)


class IndexLoader:
    """torch DataLoader(dataset, batch_size, shuffle=True) over a device batch builder: one epoch = a fresh permutation of the
    builder's rows in chunks of `batch_size` (the last partial chunk is kept, like DataLoader's drop_last=False)."""

    def __init__(self, builder, batch_size: int, shuffle: bool = True, seed: int = 2022, drop_last: bool = False, rank: int = 0,
                 world: int = 1):
        """rank / world: the DistributedSampler Lightning installs under DDP -- every rank draws the SAME permutation (common
        seed), pads it to a multiple of `world` by wrapping around and takes every world-th row starting at its rank."""
        self.builder, self.bs, self.shuffle, self.drop_last = builder, int(batch_size), bool(shuffle), bool(drop_last)
        self.rank, self.world = int(rank), int(world)
        self.gen = torch.Generator().manual_seed(seed)

    def _rows(self):
        n = len(self.builder)
        return (n + self.world - 1) // self.world

    def __len__(self):
        n = self._rows()
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self):
        n = len(self.builder)
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        if self.world > 1:
            total = self._rows() * self.world
            order = torch.cat([order, order[: total - n]])[self.rank::self.world]
        for i in range(len(self)):
            yield self.builder(order[i * self.bs:(i + 1) * self.bs])


def semisl_loaders(hparams, labelled, unlabelled, device="cuda"):
    """load_datasets_separate (trainers/evaluate.py:50-83) on data that is already in memory: labelled / unlabelled =
    (images, table, labels) with images uint8 [N,H,W,3] or float [N,3,H,W].  Sets hparams.repeat_ratio (and hparams.K for
    SimMatch) like the reference, splits the batch 1 : unlabelled_ratio, and returns {'l': loader, 'u': loader} of device batch
    builders in the batch layout the selected algorithm consumes."""
    from .fit import repeat_ratio, split_batch_size
    get = (lambda k, d=None: hparams.get(k, d)) if isinstance(hparams, dict) else (lambda k, d=None: getattr(hparams, k, d))

    def put(k, v):
        if isinstance(hparams, dict):
            hparams[k] = v
        else:
            setattr(hparams, k, v)

    algo, seed = get("algorithm_name", "STiL"), int(get("seed", 2022))
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    sampler_seed = seed           # the sampler permutation is COMMON to the ranks (DistributedSampler); the augmentation / corruption
    seed = seed + 1000 * rank     # draws are per rank (Lightning's seed_everything(workers=True) seeds workers by global rank)
    common = dict(img_size=get("img_size"), target=get("target", "dvm"), corruption_rate=get("corruption_rate", 0.3), device=device)
    (il, tl, yl), (iu, tu, yu) = labelled, unlabelled
    if algo in ("CoMatch", "SimMatch", "FreeMatch"):
        lab = EvalTrainBatchBuilder(il, tl, yl, eval_train_augment_rate=get("eval_train_augment_rate", 0.8), seed=seed, **common)
        unl = StrongWeakBatchBuilder(iu, tu, yu, two_strong=(algo == "CoMatch"), seed=seed + 100, **common)
        if algo == "SimMatch":
            put("K", len(lab))
    else:
        lab = ContrastiveBatchBuilder(il, tl, yl, augmentation_rate=get("augmentation_rate", 0.95), labelled=True, seed=seed, **common)
        unl = ContrastiveBatchBuilder(iu, tu, yu, augmentation_rate=get("augmentation_rate", 0.95), labelled=False, seed=seed + 100, **common)
    ratio = int(get("unlabelled_ratio", 7))
    put("repeat_ratio", repeat_ratio(len(unl), len(lab), ratio))
    l_bs, u_bs = split_batch_size(int(get("batch_size")), ratio)
    return {"l": IndexLoader(lab, l_bs, seed=sampler_seed + 1, rank=rank, world=world),
            "u": IndexLoader(unl, u_bs, seed=sampler_seed + 2, rank=rank, world=world)}
